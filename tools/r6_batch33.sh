#!/bin/bash
# round 6: the deformable block of res4 at config 5's per-rank size, regular grid (initialisation) against learned offsets (the sigma = 0.01 case in the tracked profiles)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b33; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for s in 0 0.01 0.03; do LGD_DCN_OFFSET_SIGMA=$s timeout 300 python tools/block_prof.py dcn 2>/dev/null | head -14; echo; done | tee $O/dcn_learned_offsets.txt
