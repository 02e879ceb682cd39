python -m pytest tests/test_model_gpu.py -q -x -k "golden or teacher_features or two_ranks_on_one_gpu or trainer" 2>&1 | tail -3
b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3), {k: round(v,6) for k,v in d['losses'].items()})"; }
CFG=""; CFGNAME=c2; b LGD_X=0; b LGD_TEACHER_STREAM=0; b LGD_X=0; b LGD_TEACHER_STREAM=0
CFG="--config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2"; CFGNAME=c4; STEPS=30
b LGD_X=0; b LGD_TEACHER_STREAM=0; b LGD_X=0; b LGD_TEACHER_STREAM=0
