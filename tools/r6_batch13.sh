#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b13; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ timeout 300 python tools/canary_probe.py 2>&1 | grep canary
  for pad in 0 16384 81920; do echo "--- LGD_H2_LDS_PAD=$pad (h2_fwd asks for 72 KB + pad of LDS)"; LGD_H2_LDS_PAD=$pad timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor h2_fwd 2>&1 | grep -E "^y "; done; } | tee $O/coresidency.log
