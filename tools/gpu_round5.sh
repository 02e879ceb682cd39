#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/r02_pytest5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02_pytest5.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/r02_bench_under_rocprof.err
cp $(ls /tmp/prof_ks/*/*kernel_stats.csv | head -1) $R/gpurun_out/r02_bench_rocprofv3_kernel_stats.csv
cd $R
grep -E "passed|failed|FAILED|parity|fused vs|DDP\(" gpurun_out/r02_pytest5.log | cut -c1-400
head -c 600 gpurun_out/r02_bench_under_rocprof.json
