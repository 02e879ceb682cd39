#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b10; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor gemm3 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo
  for a in 1 2 3 4 5; do timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor h2_fwd --lib tools/lab/liblgd_h2abl_$a.so 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo; done; } | tee $O/aggressor_parts.log
