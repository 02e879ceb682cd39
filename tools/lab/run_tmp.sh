python -m pytest tests/test_h2_gpu.py -q -x 2>&1 | tail -2
b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3))"; }
CFG=""; CFGNAME="c2 lds-epilogue"; b LGD_X=0; b LGD_X=0
export LGD_HIPCC_DEFS="-DLGD_H2_EPI_LDS=0"; touch lgd_amd/csrc/h2.hip; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
CFGNAME="c2 dword-epilogue"; b LGD_X=0; b LGD_X=0
