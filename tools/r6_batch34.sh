#!/bin/bash
# round 6: dcn_im2col through an LDS window -- exactness tests, then the res4 block at config 5's per-rank size, regular grid / learned offsets, window on / off
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b34; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "deform" 2>&1 | tail -4
for w in 1 0; do for s in 0 0.01 0.03; do
  echo "LGD_DCN_WINDOW=$w"; LGD_DCN_WINDOW=$w LGD_DCN_OFFSET_SIGMA=$s timeout 300 python tools/block_prof.py dcn 2>/dev/null | grep "block C=\|dcn_" ; done; done | tee $O/dcn_window.txt
for w in 1 0; do echo "LGD_DCN_WINDOW=$w res3"; LGD_DCN_WINDOW=$w LGD_DCN_OFFSET_SIGMA=0.01 timeout 300 python tools/block_prof.py dcn 512 128 100 168 2 2>/dev/null | grep "block C=\|dcn_"; echo "LGD_DCN_WINDOW=$w res5"; LGD_DCN_WINDOW=$w LGD_DCN_OFFSET_SIGMA=0.01 timeout 300 python tools/block_prof.py dcn 2048 512 25 42 2 2>/dev/null | grep "block C=\|dcn_"; done | tee -a $O/dcn_window.txt
