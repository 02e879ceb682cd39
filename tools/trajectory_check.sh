#!/bin/bash
# same seed / synthetic batch, 42 optimizer steps from the same initial weights: mean losses of the last 40 for the shipped path,
# the two-pass head, the library convolutions (round 1's file: profiles/r01_training_trajectory_check.txt)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/r02_training_trajectory_check.txt
echo "# python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-kernel-timing (BASELINE configs[1], same seed / synthetic batch), MI355X" > $out
echo "# mean losses over the 40 timed optimizer steps (42 steps from the same initial weights)" >> $out
run() {
  name=$1; shift
  env "$@" timeout 900 python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-kernel-timing $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['losses']
print('%-52s %6.1f ms/step  '%('$name',d['ms_per_step'])+'  '.join('%s %.6f'%(k,l[k]) for k in ('loss_cls','loss_box_reg','loss_cls.tea','loss_box_reg.tea','loss_distill','total_loss')))" >> $out
}
EXTRA="" run "shipped: single head pass, adjoint backward, chains" LGD_X=0
EXTRA="" run "LGD_CONV_CHAIN=0 (every conv its own autograd node)" LGD_CONV_CHAIN=0
EXTRA="--head-passes 2" run "two head passes (reference order)" LGD_X=0
EXTRA="" run "LGD_FUSED_SGD=0 (torch clamp_ + SGD foreach)" LGD_FUSED_SGD=0
EXTRA="" run "LGD_WINO=0 (library convolutions)" LGD_WINO=0
cat $out
