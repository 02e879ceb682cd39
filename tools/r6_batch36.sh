#!/bin/bash
# round 6: the 3x3 convolutions' filter operands prepared ahead of the step on the side stream (ops.filters_ahead) -- tests, then same-process A/B at configs 2 / 3 / 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b36; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1500 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "stream or fork or shipped_vs_library or trainer or margin or eval" 2>&1 | tail -4
S="none;filters;teacher+head+adapter;teacher+head+adapter+filters"
timeout 900 python tools/fork_subsets.py --sets "$S" 2>$O/err.txt | tee $O/ahead_c2.txt || tail -5 $O/err.txt
timeout 900 python tools/fork_subsets.py --config configs/lgd_fcos_r50.yaml --batch 16 --steps 10 --sets "$S" 2>$O/err.txt | tee $O/ahead_c3.txt || tail -5 $O/err.txt
timeout 900 python tools/fork_subsets.py --config configs/lgd_retinanet_r101.yaml --batch 2 --steps 30 --sets "$S" 2>$O/err.txt | tee $O/ahead_c4.txt || tail -5 $O/err.txt
