#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for hp in 1 2; do
  for c in "configs/lgd_retinanet_r50.yaml 8" "configs/lgd_retinanet_r101.yaml 2" "configs/lgd_fcos_r50.yaml 16"; do
    set -- $c
    timeout 600 python bench.py --config $1 --batch-per-gpu $2 --steps 15 --warmup 3 --no-cpu-baseline --no-kernel-timing --head-passes $hp 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rep $rep head-passes $hp', '$1', 'B=$2', '%.2f ms/step'%d['ms_per_step'], 'peak %.1f GB'%d['hbm_peak_alloc_gb'])"
  done
done
done | tee gpurun_out/r02_headpass_ab.log
