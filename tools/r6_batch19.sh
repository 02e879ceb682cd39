#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b19; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 600 python tools/step_determinism.py --runs 3 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/step_determinism.log
timeout 900 python tools/stream_stress.py --steps 300 > $O/stress.json 2> $O/stress.err; tail -4 $O/stress.err | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log | cut -c1-300
