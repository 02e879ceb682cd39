#!/bin/bash
# two counter passes over the bench's own launch mix; ONE command writes the per-kernel CSV and the JSON bench.py reads (steps 2 + warmup 2 = 4 steps
# profiled: the JSON records it so that bench.py can check launches per step against its own run)
# (run on the GPU box): FETCH_SIZE and WRITE_SIZE separately
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export MIOPEN_FIND_MODE=2   # the library's conv search is irrelevant to the counters of the hand-written kernels
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-pass > /tmp/pmc_$c.log 2>&1 || { tail -5 /tmp/pmc_$c.log; exit 1; }
done
F=$(ls /tmp/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1)
W=$(ls /tmp/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1)
mkdir -p $R/gpurun_out
STEPS_PROFILED=4 python $R/tools/pmc_summary.py $F $W $R/gpurun_out/r06_bench_pmc_fetch_write.csv $R/gpurun_out/r06_pmc_traffic.json "python bench.py --steps 2 --warmup 2 --no-kernel-timing --no-host-pass (BASELINE configs[1], B=8 800x1333; round 6 launch mix: F(6x6,3x3), 3x3 channel products on h2.hip, 1x1 on gemm2h, no packed fp32)"
cat $R/gpurun_out/r06_bench_pmc_fetch_write.csv
