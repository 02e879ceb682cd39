#!/bin/bash
# round 6, final measurements part C: PMC traffic and steady-state rocprof summaries of the final fork policy; trajectory check
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 bash tools/gpu_checks.sh pmc
timeout 2400 bash tools/gpu_checks.sh profile
timeout 1200 python bench.py > $O/r06_bench_2.json 2> $O/bench.err; head -c 300 $O/r06_bench_2.json; echo
timeout 2400 bash tools/gpu_checks.sh traject
