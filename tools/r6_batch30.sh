#!/bin/bash
# round 6: occupancy knobs of the F(6x6) transforms again (register counts moved when the packed-fp32 forms left the build)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b30; mkdir -p $O
bash tools/wino6_variants.sh "" "-DLGD_W6_INT_WAVES=4" "-DLGD_W6_IN_WAVES=4" "-DLGD_W6_OUT_WAVES=5" "-DLGD_W6_OUT_WAVES=4" "-DLGD_W6_OUTT_WAVES=6" "" 2>&1 | tee $O/wino6_waves.txt
