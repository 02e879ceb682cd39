#!/bin/bash
# In-call A/B of a bench.py switch: bash tools/ab_flag.sh "--flag" config.yaml batch [reps]   -> ms/step alternating without / with the flag
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
flag=$1; cfg=$2; b=$3; reps=${4:-2}
for r in $(seq 1 $reps); do for f in "" "$flag"; do
  timeout 900 python bench.py --config $cfg --batch-per-gpu $b --steps 20 --warmup 6 --no-cpu-baseline --no-kernel-timing --no-host-pass $f 2>gpurun_out/ab_flag.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', '$cfg', 'ms/step %.2f' % d['ms_per_step'], 'value %.2f' % d['value'])" || tail -5 gpurun_out/ab_flag.err
done; done
