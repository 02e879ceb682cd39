#!/bin/bash
# round 6 lab: the F(6x6) transforms' frequency buffers as contiguous 64 KB blocks per workgroup instead of 64 runs of 1 KB (speed only: the products read garbage)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b29; mkdir -p $O
bash tools/wino6_variants.sh "" "-DLGD_LAB_BLOCK_STORE" "-DLGD_LAB_BLOCK_STORE -DLGD_LAB_BLOCK_LOAD" "" 2>&1 | tee $O/block_layout_lab.txt
