#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b9; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for a in none copy amax_maps filters wino_in_h2 wino_in h2_fwd wino_out conv; do timeout 200 python tools/conv_stage_probe.py --rounds 150 --aggressor $a 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo; done | sed 's/library: [^|]*|//' | tee $O/aggressor_bisect.log
