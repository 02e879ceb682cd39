#!/bin/bash
# steady-state kernel stats of one config (GPU box): two rocprofv3 --kernel-trace --stats runs differing in --steps, differenced.
#   bash tools/profile_config.sh <yaml> <batch-per-gpu> <tag> [short long]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cfg=$1; bpg=$2; tag=$3; n0=${4:-3}; n1=${5:-13}
cd /tmp && export TMPDIR=/tmp
for n in $n0 $n1; do rm -rf /tmp/prof_$n
  timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python $R/bench.py --config $R/$cfg --batch-per-gpu $bpg --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass > /tmp/prof_$n.log 2>&1 || tail -5 /tmp/prof_$n.log
  cp $(ls /tmp/prof_$n/*/*kernel_stats.csv | head -1) $O/${tag}_kernel_stats_steps$n.csv
done
cd $R && python tools/prof_diff.py $O/${tag}_kernel_stats_steps$n0.csv $O/${tag}_kernel_stats_steps$n1.csv $((n1 - n0)) $O/${tag}_rocprofv3_steady_state.csv
