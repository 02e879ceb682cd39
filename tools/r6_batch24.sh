#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b24; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for q in 2 4 5 8; do for sk in 0 3; do
  echo "PROBE_SKIP_STREAMS=$sk"; GPU_MAX_HW_QUEUES=$q PROBE_SKIP_STREAMS=$sk timeout 300 python tools/queue_probe.py 2>$O/err.txt || tail -5 $O/err.txt
done; done 2>&1 | tee $O/queue_probe.txt
