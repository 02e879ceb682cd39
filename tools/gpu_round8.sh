#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
mkdir -p $O
bash tools/pmc_bench.sh > $O/r02_pmc.log 2>&1
tail -3 $O/r02_pmc.log
# the DDP / RCCL bench path on the one GPU of this box (world size 1 through torchrun, and forced DDP)
LGD_FORCE_DDP=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/r02_bench_force_ddp.json 2> $O/r02_bench_force_ddp.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/r02_bench_torchrun1.json 2> $O/r02_bench_torchrun1.err
head -c 300 $O/r02_bench_force_ddp.json; echo; head -c 300 $O/r02_bench_torchrun1.json; echo
tail -2 $O/r02_bench_force_ddp.err $O/r02_bench_torchrun1.err
