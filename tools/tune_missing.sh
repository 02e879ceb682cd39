#!/bin/bash
# TunableOp pass for the GEMM shapes missing from lgd_amd/tuning/tunableop_gfx950.csv (run on the MI355X box): the table seeds
# TunableOp's result file, so only shapes it does not hold are tuned; the merged file lands in gpurun_out/tunableop_gfx950_new.csv.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
cp lgd_amd/tuning/tunableop_gfx950.csv $O/tunable0.csv   # TunableOp appends the device ordinal to the file name
for c in "configs/lgd_retinanet_r50.yaml 8" "configs/lgd_fcos_r50.yaml 16" "configs/lgd_retinanet_r101.yaml 2" "configs/lgd_retinanet_r101_dcnv2.yaml 2"; do set -- $c
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunable.csv \
    timeout 1500 python bench.py --config $1 --batch-per-gpu $2 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/tune.log 2>&1
  tail -c 200 $O/tune.log; echo; wc -l $O/tunable*.csv
done
ls $O/tunable*
cat $O/tunable*.csv | grep -v "^Validator" | sort -u | wc -l
cp $O/tunable0.csv $O/tunableop_gfx950_new.csv
