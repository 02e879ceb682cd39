#!/bin/bash
# round 6, final measurements part B (after the fork change): full GPU suite, smoke, the driver's bench line, other configs
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 3000 python -m pytest tests -m gpu -q > $O/r06_pytest_gpu.log 2>&1; tail -4 $O/r06_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.log 2>&1; tail -3 $O/r06_smoke.log
timeout 1200 python bench.py > $O/r06_bench.json 2> $O/bench.err; head -c 300 $O/r06_bench.json; echo
for c in "configs/lgd_fcos_r50.yaml 16 config3_fcos_r50_b16" "configs/lgd_retinanet_r101.yaml 2 config4_r101_b2_per_gpu" "configs/lgd_retinanet_r101_dcnv2.yaml 2 config5_r101_dcnv2_b2"; do set -- $c
  extra=""; [[ $3 == config5* ]] && extra="--multiscale"
  timeout 900 python bench.py --config $1 --batch-per-gpu $2 --no-cpu-baseline $extra > $O/r06_bench_$3.json 2> $O/bench_$3.err; head -c 200 $O/r06_bench_$3.json; echo
done
