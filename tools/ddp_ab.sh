#!/bin/bash
# DDP overhead at world size 1 (GPU box, one call): the plain trainer vs the same step under DistributedDataParallel over RCCL
# (LGD_FORCE_DDP=1) vs launched through torch.distributed.run, for BASELINE configs 2 and 4 [ref: train.py:279-281].
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
one() { python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-58s %7.2f ms/step %7.1f img/s' % (sys.argv[1], r['ms_per_step'], r['value']))" "$1"; }
for c in "configs/lgd_retinanet_r50.yaml 8 20 5" "configs/lgd_retinanet_r101.yaml 2 40 8"; do set -- $c
  A="--config $1 --batch-per-gpu $2 --steps $3 --warmup $4 --no-cpu-baseline --no-kernel-timing --no-host-pass"
  for rep in 1 2; do
    python bench.py $A 2>/dev/null | one "$1 B=$2 plain"
    LGD_FORCE_DDP=1 MASTER_PORT=29611 python bench.py $A 2>/dev/null | one "$1 B=$2 DDP (LGD_FORCE_DDP=1, world 1)"
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 $A 2>/dev/null | one "$1 B=$2 torchrun --nproc-per-node 1"
  done
done
python tools/ddp_bucket_times.py lgd_retinanet_r50 8
python tools/ddp_bucket_times.py lgd_retinanet_r101 2
