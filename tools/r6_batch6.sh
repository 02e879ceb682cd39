#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b6; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ python tools/conv_stage_probe.py; python tools/conv_stage_probe.py --h2
  for v in A B C D; do python tools/conv_stage_probe.py --lib tools/lab/liblgd_coh_$v.so; done; } 2>&1 | grep -v amdgpu.ids | tee $O/conv_stage_probe.log
