#!/bin/bash
# same seed / synthetic batch, 42 optimizer steps from the same initial weights: mean losses of the last 40 for the shipped path,
# the two-pass head, the library convolutions (round 1's file: profiles/r01_training_trajectory_check.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r06_training_trajectory_check.txt
echo "# python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-pass --batches 1 (BASELINE configs[1], same seed / synthetic batch), MI355X" > $out
echo "# mean losses over the 40 timed optimizer steps (42 steps from the same initial weights)" >> $out
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-pass --batches 1 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['losses']
print('%-52s %6.1f ms/step  '%('$name',d['ms_per_step'])+'  '.join('%s %.6f'%(k,l[k]) for k in ('loss_cls','loss_box_reg','loss_cls.tea','loss_box_reg.tea','loss_distill','total_loss')))" >> $out
}
EXTRA="" run "shipped: F(6x6,3x3), products on h2.hip / gemm2h (f16x2)" LGD_X=0
EXTRA="" run "round-4 path: gemm3.hip bf16x3 + library dW" LGD_H2=0 LGD_GEMM2H=0
EXTRA="" run "f16x2 with every bound by its own pass (no tags)" LGD_H2_TAGS=0
EXTRA="" run "stem on the library convolution + pooling pass" LGD_STEM7=0
EXTRA="--one-stream" run "everything on one stream (no forks)" LGD_X=0
EXTRA="" run "1x1 products: bf16x3 / library (no gemm2h)" LGD_GEMM2H=0
EXTRA="--library-gemms" run "3x3 products h2 off, all on the library fp32 GEMMs" LGD_H2=0 LGD_GEMM2H=0
EXTRA="--no-teacher-fold" run "teacher activations as their own passes" LGD_X=0
EXTRA="--wino-tile 4" run "F(4x4,3x3)" LGD_X=0
EXTRA="--head-passes 2" run "two head passes (reference order)" LGD_X=0
EXTRA="--torch-optimizers" run "torch clamp_ + SGD foreach" LGD_X=0
EXTRA="--library-convs" run "library convolutions" LGD_X=0
cat $out
