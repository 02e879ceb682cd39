#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b8; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ echo "=== torch's own kernels as the victim"; timeout 300 python tools/conv_stage_probe.py --torch --rounds 200 | grep -E "differs|first"
  echo "=== shipped library, 200 rounds"; timeout 300 python tools/conv_stage_probe.py --rounds 200 | grep -E "^y |first"; timeout 300 python tools/conv_stage_probe.py --h2 --rounds 200 | grep -E "^y |first"
  echo "=== E: wino6_out loads M at system scope"; timeout 300 python tools/conv_stage_probe.py --lib tools/lab/liblgd_coh_E.so --rounds 200 | grep -E "^y "
  echo "=== B: gemm3 stores C write-through (sc0 sc1)"; timeout 300 python tools/conv_stage_probe.py --lib tools/lab/liblgd_coh_B.so --rounds 200 | grep -E "^y "
  echo "=== F: h2_fwd stores C write-through (sc0 sc1), victim on h2"; timeout 300 python tools/conv_stage_probe.py --h2 --lib tools/lab/liblgd_coh_F.so --rounds 200 | grep -E "^y "
  echo "=== F timing"; timeout 300 python tools/h2_rounds.py --lib tools/lab/liblgd_coh_F.so 5248 10496 | grep "^T"; timeout 300 python tools/h2_rounds.py 5248 10496 | grep "^T"
} 2>&1 | grep -v amdgpu.ids | tee $O/coherence_probe.log
