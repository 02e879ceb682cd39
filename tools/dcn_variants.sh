#!/bin/bash
# rows per thread of the DCN backward kernel (csrc/dcn.hip) on the GPU box: rebuild with each LGD_DCN_ROWS, config 5 bench
#   usage: bash tools/dcn_variants.sh 2 4 6 8
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in "$@"; do
  echo "=== LGD_DCN_ROWS=$r"
  touch lgd_amd/csrc/dcn.hip
  LGD_HIPCC_DEFS="-DLGD_DCN_ROWS=$r" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
  python bench.py --config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['all_hip_kernels']
print('config 5 ms/step', round(d['ms_per_step'],2), {n: (k[n]['avg_us'], k[n]['min_us'], k[n]['max_us']) for n in k if 'dcn_col' in n})"
done
touch lgd_amd/csrc/dcn.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
