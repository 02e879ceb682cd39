python -m pytest tests/test_kernels_gpu.py tests/test_h2_gpu.py -q -x 2>&1 | tail -3
LGD_H2_DEBUG=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>&1 | grep "without a tag" | sort | uniq -c | sort -rn | cut -c1-220
b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3))"; }
CFG=""; CFGNAME=c2; b LGD_X=0; b LGD_X=0
CFG="--config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2"; CFGNAME=c4; STEPS=30; b LGD_X=0; b LGD_RELU_AMAX=0;  b LGD_X=0; b LGD_RELU_AMAX=0
