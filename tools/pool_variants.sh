#!/bin/bash
# build variants of the mask-pooling kernel (csrc/box_pool.hip) on the GPU box and time them HBM-cold (tools/kbench.py)
#   usage: bash tools/pool_variants.sh "-DLGD_POOL_DEPTH=3" "-DLGD_POOL_F16MASK=1" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "$@"; do
  echo "=== LGD_HIPCC_DEFS='$d'"
  touch lgd_amd/csrc/box_pool.hip
  LGD_HIPCC_DEFS="$d" python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -i " error" | head -3
  for i in 1 2; do python tools/kbench.py 2>&1 | grep "box_sum_kernel\|gn_pool_kernel\|gn_stats_kernel" | tr '\n' ' '; echo; done
done
touch lgd_amd/csrc/box_pool.hip
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
