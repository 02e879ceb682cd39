#!/bin/bash
# load-pipeline depth variants of the band-streaming kernels (box_sum / gn_pool / gn_pool backward; csrc/box_ops.hip) on the GPU box:
# rebuild with each LGD_BAND_G and time them HBM-cold (tools/kbench.py).   usage: bash tools/band_variants.sh 4 8 12
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in "$@"; do
  echo "=== LGD_BAND_G=$g"
  touch lgd_amd/csrc/box_ops.hip
  LGD_HIPCC_DEFS="-DLGD_BAND_G=$g" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i "error" | head -3
  python tools/kbench.py 2>&1 | grep "box_sum_kernel\|gn_pool\|box_paint"
done
touch lgd_amd/csrc/box_ops.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
