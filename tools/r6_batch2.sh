#!/bin/bash
# round 6, GPU batch 2: the stall with the library calls ordered (must run through) and unordered (must stall: the control), the new tests, the guard's cost
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b2; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
LGD_LIBRARY_ORDER=1 STEPS=150 bash tools/stall_repro.sh "loop only_head" > $O/stall_ordered.txt 2>&1; cp gpurun_out/stall/verdict.txt $O/stall_ordered_verdict.txt
LGD_LIBRARY_ORDER=0 STEPS=30 bash tools/stall_repro.sh "only_head" > $O/stall_unordered.txt 2>&1; cp gpurun_out/stall/verdict.txt $O/stall_unordered_verdict.txt; cp gpurun_out/stall/only_head.log $O/stall_unordered_only_head.log
cat $O/stall_ordered_verdict.txt $O/stall_unordered_verdict.txt
timeout 900 python tools/stream_stress.py --steps 300 > $O/stress.json 2> $O/stress.err; tail -4 $O/stress.err | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q -s -k "pinned_ring or config2_step_shipped or multiscale_dcn or two_streams or side_stream" > $O/pytest_new.log 2>&1
grep -E "passed|failed|^FAILED|Error|config 2 \(8" $O/pytest_new.log | cut -c1-400 | tail -12
bash tools/ab_env.sh LGD_LIBRARY_ORDER configs/lgd_retinanet_r50.yaml 8 2 2>&1 | tee $O/ab_order_c2.txt
bash tools/ab_env.sh LGD_LIBRARY_ORDER configs/lgd_retinanet_r101.yaml 2 2 2>&1 | tee $O/ab_order_c4.txt
