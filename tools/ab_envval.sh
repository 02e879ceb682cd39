#!/bin/bash
# In-call A/B of an environment variable's VALUES: bash tools/ab_envval.sh VAR "v1 v2 ..." config.yaml batch [reps]
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
var=$1; vals=$2; cfg=$3; b=$4; reps=${5:-2}
for r in $(seq 1 $reps); do for v in $vals; do
  env $var=$v timeout 900 python bench.py --config $cfg --batch-per-gpu $b --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>gpurun_out/ab_env.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', '$cfg', 'ms/step %.2f' % d['ms_per_step'], 'value %.2f' % d['value'])" || tail -5 gpurun_out/ab_env.err
done; done
