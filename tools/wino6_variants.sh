#!/bin/bash
# occupancy variants of the F(6x6,3x3) transform kernels (GPU box): rebuild winograd6.hip with each setting of the
# LGD_W6_*_WAVES knobs and time the pyramid convolution's kernels (tools/wino_tile_ab.py).   usage: bash tools/wino6_variants.sh "<defs>" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
for defs in "$@"; do
  echo "=== $defs"
  touch lgd_amd/csrc/winograd6.hip
  LGD_HIPCC_DEFS="$defs" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i "error\|warning: " | head -5
  LGD_WINO_AB_TILES=6 python tools/wino_tile_ab.py 8 2>&1 | grep "tile 6: fwd\|rror\|fault" | sed 's/"wino_filter[^,]*, //g' | cut -c1-700
done
touch lgd_amd/csrc/winograd6.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
