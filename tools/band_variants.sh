#!/bin/bash
# load-pipeline depth variants of the band-streaming painting kernels (box_paint / gn_pool backward apply; csrc/box_ops.hip) on the GPU
# box: rebuild with each LGD_BAND_G and time them HBM-cold at 8 and 2 images (tools/kbench.py).   usage: bash tools/band_variants.sh 4 8 12
cd ${GRAFT_REPO_ROOT:-/root/repo}
for g in "$@"; do
  echo "=== LGD_BAND_G=$g"
  touch lgd_amd/csrc/box_ops.hip
  LGD_HIPCC_DEFS="-DLGD_BAND_G=$g" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
  for b in 8 2; do python tools/kbench.py --B $b --nset $((24 / b)) 2>&1 | grep "gn_pool_bwd_apply\|box_paint" | cut -c1-75 | tr '\n' ' '; echo " (B=$b)"; done
done
touch lgd_amd/csrc/box_ops.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
