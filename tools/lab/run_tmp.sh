b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3))"; }
CFG="--config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2"; CFGNAME=c4; STEPS=30
for r in 1 2 3; do b LGD_X=0; b LGD_STEP_IMAGES=0; done
for r in 1 2 3 4 5 6; do EXTRA="" b LGD_X=0; EXTRA="--no-teacher-fold" b LGD_X=0; done
CFG=""; CFGNAME=c2; STEPS=20
for r in 1 2 3 4 5 6; do EXTRA="" b LGD_X=0; EXTRA="--no-teacher-fold" b LGD_X=0; done
