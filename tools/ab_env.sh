#!/bin/bash
# In-call A/B of an environment switch: bash tools/ab_env.sh VAR config.yaml batch [reps]   -> ms/step alternating VAR=1 / VAR=0
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; cfg=$2; b=$3; reps=${4:-2}
for r in $(seq 1 $reps); do for v in 1 0; do
  env $var=$v timeout 900 python bench.py --config $cfg --batch-per-gpu $b --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>gpurun_out/ab_env.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', '$cfg', 'ms/step %.2f' % d['ms_per_step'], 'value %.2f' % d['value'])" || tail -5 gpurun_out/ab_env.err
done; done
