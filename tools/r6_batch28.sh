#!/bin/bash
# round 6: one shared side stream + event joins -- fork subsets (all 16) at config 2 and the usual 10 at config 3
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b28; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 900 python tools/fork_subsets.py --all-subsets 2>$O/err.txt | tee $O/fork_subsets_c2.txt || tail -5 $O/err.txt
GPU_MAX_HW_QUEUES=8 timeout 900 python tools/fork_subsets.py 2>$O/err.txt | tee $O/fork_subsets_c2_hwq8.txt || tail -5 $O/err.txt
timeout 900 python tools/fork_subsets.py --config configs/lgd_fcos_r50.yaml --batch 16 --steps 10 --all-subsets 2>$O/err.txt | tee $O/fork_subsets_c3.txt || tail -5 $O/err.txt
timeout 900 python tools/fork_subsets.py --config configs/lgd_retinanet_r101.yaml --batch 2 --steps 30 2>$O/err.txt | tee $O/fork_subsets_c4.txt || tail -5 $O/err.txt
