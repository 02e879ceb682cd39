b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3))"; }
CFG=""; CFGNAME="c2"; b LGD_X=0; b LGD_GEMM2H_TILE64=0; b LGD_X=0; b LGD_GEMM2H_TILE64=0
CFG="--config configs/lgd_fcos_r50.yaml --batch-per-gpu 16"; CFGNAME="c3"; STEPS=10; b LGD_X=0; b LGD_GEMM2H_TILE64=0
