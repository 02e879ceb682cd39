#!/bin/bash
# round 6, GPU batch 3: forks with the per-call gate (side streams carry this library's kernels only) -- tests, then the forks' worth per config
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b3; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 2400 python -m pytest tests -m gpu -q -s -k "two_streams or f4x4 or competing or config2_step or fpn or meta_arch or single_head or side_stream or multiscale or two_ranks_on_one or single_rank" > $O/pytest_sel.log 2>&1
grep -E "passed|failed|^FAILED|Error|300 steps" $O/pytest_sel.log | cut -c1-400 | tail -12
bash tools/ab_env.sh LGD_SIDE_STREAMS configs/lgd_retinanet_r50.yaml 8 2 2>&1 | tee $O/ab_forks_c2.txt
bash tools/ab_env.sh LGD_SIDE_STREAMS configs/lgd_fcos_r50.yaml 16 1 2>&1 | tee $O/ab_forks_c3.txt
bash tools/ab_env.sh LGD_SIDE_STREAMS configs/lgd_retinanet_r101.yaml 2 2 2>&1 | tee $O/ab_forks_c4.txt
