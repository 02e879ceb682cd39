#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out
# 1. TunableOp: tune the GEMM shapes that are not in the table yet (all four configs), merge
cp lgd_amd/tuning/tunableop_gfx950.csv $O/tunable_r02b0.csv
for cfgargs in "" "--config configs/lgd_fcos_r50.yaml --batch-per-gpu 16" "--config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2" "--config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2"; do
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunable_r02b.csv \
    timeout 1500 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing $cfgargs > $O/r02_tune_b.log 2>&1
done
python tools/merge_tunable.py lgd_amd/tuning/tunableop_gfx950.csv lgd_amd/tuning/tunableop_gfx950.csv $O/tunable_r02b0.csv >> $O/r02_tune_b.log 2>&1
cp lgd_amd/tuning/tunableop_gfx950.csv $O/tunableop_gfx950_merged_b.csv
# 2. bench lines
timeout 900 python bench.py > $O/r02_bench_c.json 2> $O/r02_bench_c.err
timeout 900 python bench.py --config configs/lgd_fcos_r50.yaml --batch-per-gpu 16 --no-cpu-baseline > $O/r02_bench_config3_fcos_r50_b16.json 2> $O/r02_bench_config3.err
timeout 900 python bench.py --config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2 --no-cpu-baseline > $O/r02_bench_config4_r101_b2_per_gpu.json 2> $O/r02_bench_config4.err
timeout 900 python bench.py --config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --no-cpu-baseline > $O/r02_bench_config5_r101_dcnv2_b2.json 2> $O/r02_bench_config5.err
# 3. op-level view + host issue time
timeout 600 python tools/step_profile_ops.py > $O/r02_step_profile_ops.txt 2>&1
timeout 600 python tools/cpu_issue_time.py > $O/r02_cpu_issue_time.txt 2>&1
# 4. tests
timeout 2400 python -m pytest tests -m gpu -q > $O/r02_pytest7.log 2>&1
tail -3 $O/r02_pytest7.log
for f in $O/r02_bench_c.json $O/r02_bench_config3_fcos_r50_b16.json $O/r02_bench_config4_r101_b2_per_gpu.json $O/r02_bench_config5_r101_dcnv2_b2.json; do head -c 260 $f; echo; done
cat $O/r02_cpu_issue_time.txt | grep -v amdgpu
