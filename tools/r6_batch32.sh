#!/bin/bash
# round 6: the 300-step stress run (all forks + competing copy stream vs one-stream twins) under 8 hardware queues, and in the stream-per-fork form under 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b32; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for v in "GPU_MAX_HW_QUEUES=8 LGD_ONE_SIDE_STREAM=1" "GPU_MAX_HW_QUEUES=4 LGD_ONE_SIDE_STREAM=0"; do
  echo "== $v"
  env $v timeout 1500 python tools/stream_stress.py --steps 300 > $O/stress.json 2> $O/stress.err; echo "exit $?"
  grep -v Warning $O/stress.err | tail -4
  python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6b32/stress.json").read().strip().splitlines()[-1])
a, b = d["forked"], d["one_stream"]
worst = max(abs(p[k] - q[k]) for p, q in zip(a, b) for k in p)
print("steps %d forks/step %s competitor copies %d  ms/step forked under load %.1f, one stream %.1f  worst |loss difference| over all steps %.3g" % (
    d["steps"], d["forks_per_step"], d["competitor_copies"], d["ms_per_step_forked_under_load"], d["ms_per_step_one_stream"], worst))
PY
done 2>&1 | tee $O/stress_queues.txt
