#!/bin/bash
# config 3 (FCOS R-50, 16 img/GPU): 22 optimizer steps from the same initial weights / batch: mean losses of the last 20 for the shipped
# path, the unfolded GroupNorm + ReLU passes, the composed regression losses, F(4x4,3x3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r03_training_trajectory_check_fcos.txt
echo "# python bench.py --config configs/lgd_fcos_r50.yaml --batch-per-gpu 16 --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-pass --batches 1, MI355X" > $out
run() {
  name=$1; shift
  timeout 900 python bench.py --config configs/lgd_fcos_r50.yaml --batch-per-gpu 16 --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-pass --batches 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['losses']
print('%-52s %6.1f ms/step  '%('$name',d['ms_per_step'])+'  '.join('%s %.6f'%(k,l[k]) for k in ('loss_cls','loss_box_reg','loss_centerness','loss_cls.tea','loss_box_reg.tea','loss_centerness.tea','loss_distill','total_loss')))" >> $out
}
run "shipped: F(6x6,3x3), GroupNorm folded into the next convolution"
run "GroupNorm(32) + ReLU as its own passes" --no-gn-fold
run "GIoU + centerness losses as composed torch ops" --no-fcos-fused-loss
run "F(4x4,3x3)" --wino-tile 4
cat $out
