#!/bin/bash
# round 6, GPU batch 5: which fork changes a bit (tools/fork_bisect.py); de-phasing the two workgroups of a CU in h2_fwd (LGD_H2_SKEW)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b5; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python tools/fork_bisect.py 2>&1 | grep -v amdgpu.ids | tee $O/fork_bisect.log | cut -c1-420
for s in 0 2 4 6 8; do echo "--- LGD_H2_SKEW=$s"; LGD_H2_SKEW=$s python tools/h2_rounds.py 5248 10496 2>&1 | grep "^T"; done | tee $O/h2_fwd_skew.log
