#!/bin/bash
# round 6: all forks on ONE shared side stream (streams._ONE_SIDE) -- fork subsets at config 2 / 3 under 4 and 8 hardware queues, against a stream per fork
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b27; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for one in 1 0; do for q in 4 8; do
  echo "== LGD_ONE_SIDE_STREAM=$one"; LGD_ONE_SIDE_STREAM=$one GPU_MAX_HW_QUEUES=$q timeout 900 python tools/fork_subsets.py 2>$O/err.txt || tail -5 $O/err.txt
done; done 2>&1 | tee $O/fork_subsets_c2.txt
for q in 4 8; do
  echo "== LGD_ONE_SIDE_STREAM=1"; LGD_ONE_SIDE_STREAM=1 GPU_MAX_HW_QUEUES=$q timeout 900 python tools/fork_subsets.py --config configs/lgd_fcos_r50.yaml --batch 16 --steps 10 2>$O/err.txt || tail -5 $O/err.txt
done 2>&1 | tee $O/fork_subsets_c3.txt
