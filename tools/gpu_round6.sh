#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r02_pytest6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_pytest6.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_bits.json 2> gpurun_out/r02_bench_bits.err
cd /tmp && export TMPDIR=/tmp
for n in 5 25; do
  rm -rf /tmp/prof_$n
  timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python $R/bench.py --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r02_prof_$n.json 2> $R/gpurun_out/r02_prof_$n.err
  cp $(ls /tmp/prof_$n/*/*kernel_stats.csv | head -1) $R/gpurun_out/r02_kernel_stats_steps$n.csv
done
cd $R
python tools/prof_diff.py gpurun_out/r02_kernel_stats_steps5.csv gpurun_out/r02_kernel_stats_steps25.csv 20 gpurun_out/r02_bench_rocprofv3_steady_state.csv
grep -E "passed|failed|^FAILED|parity|fused vs|DDP\(" gpurun_out/r02_pytest6.log | cut -c1-600
head -c 400 gpurun_out/r02_bench_bits.json
