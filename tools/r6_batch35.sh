#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b35; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "deform or dcn" 2>&1 | tail -3
for s in 0 0.01; do LGD_DCN_OFFSET_SIGMA=$s timeout 300 python tools/block_prof.py dcn 2>/dev/null | grep "block C=\|dcn_" ; done | tee $O/dcn_skip_zero.txt
for r in 1 2; do timeout 600 python bench.py --config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --multiscale --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 ms/step %.2f value %.2f' % (d['ms_per_step'], d['value']))"; done | tee -a $O/dcn_skip_zero.txt
