#!/bin/bash
# round 6, GPU batch 1: the new evidence tests + two lab probes (results under gpurun_out/r6b1/)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b1; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 2400 python -m pytest tests -m gpu -q -x -s -k "margin or within_plane or pinned_ring or config2_step_shipped or competing_stream or h2_dma or kloop" > $O/pytest_new.log 2>&1
grep -E "passed|failed|^FAILED|Error|margin case|within-plane|too wide|300 steps|config 2 \(8" $O/pytest_new.log | cut -c1-400 | tail -30
timeout 600 python tools/chunk_probe.py > $O/chunk_probe.log 2>&1; cat $O/chunk_probe.log | cut -c1-300
timeout 600 python tools/chunk_probe.py --both-pyramids > $O/chunk_probe_both.log 2>&1; cat $O/chunk_probe_both.log | cut -c1-300
timeout 900 python tools/gemm2h_probe.py 8 > $O/gemm2h_probe_dma.log 2>&1; cut -c1-400 $O/gemm2h_probe_dma.log
