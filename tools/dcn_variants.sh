#!/bin/bash
# variants of the DCN backward (csrc/dcn.hip) on the GPU box: rebuild with each set of defines, config 5 bench
#   usage: bash tools/dcn_variants.sh "-DLGD_DCN_GCH=16" "-DLGD_DCN_ACC64=0" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "$@"; do
  echo "=== LGD_HIPCC_DEFS='$d'"
  touch lgd_amd/csrc/dcn.hip
  LGD_HIPCC_DEFS="$d" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
  python bench.py --config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --steps 8 --warmup 3 --no-cpu-baseline --no-host-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_hip_kernels']
print('config 5 ms/step', round(d['ms_per_step'],2), {n: k[n]['avg_us'] for n in k if 'dcn' in n})"
done
touch lgd_amd/csrc/dcn.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
