#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b17; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for v in N; do echo "--- variant $v (winograd6.hip without packed fp32: -fno-slp-vectorize)"; timeout 200 python tools/conv_stage_probe.py --rounds 100 --aggressor h2_fwd --lib tools/lab/liblgd_coh_$v.so 2>&1 | grep -E "^y |first"; done | tee $O/no_packed.log
