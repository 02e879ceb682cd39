#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/r02_pytest9.log 2>&1
grep -E "passed|failed|^FAILED|^E  " $O/r02_pytest9.log | head -20
timeout 900 python bench.py --no-cpu-baseline > $O/r02_bench_d.json 2> $O/r02_bench_d.err
timeout 900 python bench.py --config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2 --no-cpu-baseline > $O/r02_bench_config4_d.json 2> $O/r02_bench_config4_d.err
timeout 900 python bench.py --config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --no-cpu-baseline > $O/r02_bench_config5_d.json 2> $O/r02_bench_config5_d.err
timeout 900 python bench.py --config configs/lgd_fcos_r50.yaml --batch-per-gpu 16 --no-cpu-baseline > $O/r02_bench_config3_d.json 2> $O/r02_bench_config3_d.err
for f in $O/r02_bench_d.json $O/r02_bench_config4_d.json $O/r02_bench_config5_d.json $O/r02_bench_config3_d.json; do head -c 230 $f; echo; done
timeout 600 python tools/cpu_issue_time.py lgd_retinanet_r101 2 2>&1 | grep -v amdgpu | tee $O/r02_cpu_issue_time_config4.txt
