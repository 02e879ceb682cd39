python -m pytest tests/test_model_gpu.py -q -x -k "side_stream or two_ranks or fused_sgd" 2>&1 | grep -E "passed|failed|Error" | tail -5
b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3))"; }
CFG=""; CFGNAME="c2"; b LGD_X=0
