#!/bin/bash
# round 6, final measurements part A: PMC traffic, steady-state rocprof summaries (shipped / one stream), the other BASELINE configs, trajectories
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 bash tools/gpu_checks.sh pmc
timeout 2400 bash tools/gpu_checks.sh profile
for c in "configs/lgd_fcos_r50.yaml 16 config3_fcos_r50_b16" "configs/lgd_retinanet_r101.yaml 2 config4_r101_b2_per_gpu" "configs/lgd_retinanet_r101_dcnv2.yaml 2 config5_r101_dcnv2_b2"; do set -- $c
  extra=""; [[ $3 == config5* ]] && extra="--multiscale"
  timeout 900 python bench.py --config $1 --batch-per-gpu $2 --no-cpu-baseline $extra > $O/r06_bench_$3.json 2> $O/bench_$3.err; head -c 200 $O/r06_bench_$3.json; echo
done
bash tools/ab_envval.sh GPU_MAX_HW_QUEUES "4 8" configs/lgd_retinanet_r50.yaml 8 2 2>&1 | tee $O/r06_ab_hw_queues_c2.txt
timeout 2400 bash tools/gpu_checks.sh traject
