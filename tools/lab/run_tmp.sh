b() { env "$@" python bench.py $CFG $EXTRA --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFGNAME $EXTRA $*', round(d['ms_per_step'],3))"; }
CFG=""; CFGNAME="c2 rings A2 B6"; b LGD_X=0; b LGD_X=0
cp lgd_amd/csrc/h2.hip /tmp/h2_new.hip; cp tools/lab/h2_old.hip.txt lgd_amd/csrc/h2.hip; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
CFGNAME="c2 old 3 buffers"; b LGD_X=0; b LGD_X=0
