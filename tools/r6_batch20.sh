#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b20; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ echo "--- shipped"; timeout 300 python tools/h2_rounds.py 5248 10496 2560 1312 | grep "^T"
  echo "--- LGD_H2_PIPE=1"; timeout 300 python tools/h2_rounds.py --lib tools/lab/liblgd_h2pipe.so 5248 10496 2560 1312 | grep "^T"; } | tee $O/h2_pipe_rounds.log
LGD_HIP_LIB=$PWD/tools/lab/liblgd_h2pipe.so timeout 900 python -m pytest tests/test_h2_gpu.py -m gpu -q -x -k "h2_fwd or conv3x3 or under_load or within_plane" 2>&1 | tail -3 | tee $O/pytest_pipe.log
