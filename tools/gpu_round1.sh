#!/bin/bash
# round-2 GPU call 1: parity tests, first bench line (default table), TunableOp pass for the new GEMM shapes, bench again
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r02_pytest.log
echo "pytest rc=$?" >> gpurun_out/r02_pytest.log
timeout 900 python bench.py > gpurun_out/r02_bench_untuned.json 2> gpurun_out/r02_bench_untuned.err
cp lgd_amd/tuning/tunableop_gfx950.csv gpurun_out/tunable_r020.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunable_r02.csv \
  timeout 1500 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r02_tune.log 2>&1
python tools/merge_tunable.py lgd_amd/tuning/tunableop_gfx950.csv lgd_amd/tuning/tunableop_gfx950.csv gpurun_out/tunable_r020.csv >> gpurun_out/r02_tune.log 2>&1
cp lgd_amd/tuning/tunableop_gfx950.csv gpurun_out/tunableop_gfx950_merged.csv
timeout 900 python bench.py > gpurun_out/r02_bench_tuned.json 2> gpurun_out/r02_bench_tuned.err
tail -3 gpurun_out/r02_pytest.log
cat gpurun_out/r02_bench_tuned.json | cut -c1-600
