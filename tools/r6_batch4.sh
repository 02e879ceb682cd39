#!/bin/bash
# round 6, GPU batch 4: where a launch of h2_fwd spends its time (ablation builds), and the h2 size gate at config 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b4; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ python tools/h2_rounds.py 5248 10496 2560
  for a in 1 2 3 4 5; do echo "--- LGD_H2_ABL=$a"; python tools/h2_rounds.py --lib tools/lab/liblgd_h2abl_$a.so 5248 10496 2560; done; } 2>&1 | grep -v amdgpu.ids | tee $O/h2_fwd_ablation.log
bash tools/ab_envval.sh LGD_H2_MIN_T "1500 1000 250" configs/lgd_retinanet_r50.yaml 8 2 2>&1 | tee $O/ab_h2_min_t_c2.txt
