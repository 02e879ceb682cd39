#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b12; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for a in gemm2h h2_fwd128 h2_dw h2_fwd; do timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor $a 2>&1 | grep -E "^y |library|Error|error" | tr '\n' ' '; echo; done | sed 's/library: [^|]*|//' | tee $O/aggressor_kinds.log
