#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b14; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for v in G H; do echo "--- variant $v"; timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor h2_fwd --lib tools/lab/liblgd_coh_$v.so 2>&1 | grep -E "^y "; done | tee $O/park_width.log
