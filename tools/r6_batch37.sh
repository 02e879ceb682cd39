#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b37; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
bash tools/ab_envval.sh LGD_SPLITK_FILL "1 2 3" configs/lgd_retinanet_r50.yaml 8 2 2>&1 | tee $O/splitk_fill_c2.txt
bash tools/ab_envval.sh LGD_SPLITK_FILL "1 3" configs/lgd_fcos_r50.yaml 16 1 2>&1 | tee $O/splitk_fill_c3.txt
