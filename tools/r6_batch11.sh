#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b11; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor h2_fwd --lib tools/lab/liblgd_h2abl_6.so 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo
  timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor h2_fwd 2>&1 | grep -E "^(V|M|y) |library|first" | tr '\n' ' '; echo
  echo "--- victim on h2, aggressor h2_fwd:"; timeout 200 python tools/conv_stage_probe.py --h2 --rounds 60 --aggressor h2_fwd 2>&1 | grep -E "^(V|M|y) |first" | tr '\n' ' '; echo; } | tee $O/aggressor_tr.log
