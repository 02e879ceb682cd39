#!/bin/bash
# round 6: the new fork defaults (head fork off; forks off above 4 hardware queues) -- fork tests, stress run, bench under 4 / 8 queues
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b26; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 2400 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "stream or fork or shipped_vs_library or pinned_ring or side" > $O/pytest_forks.log 2>&1; tail -5 $O/pytest_forks.log
one() { lab=$1; shift; envs=(); while [[ $1 != -- ]]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py "$@" --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>$O/err.txt \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s ms/step %.2f value %.2f forks %s' % ('$lab', d['ms_per_step'], d['value'], d['config'].get('side_streams')))" || tail -3 $O/err.txt; }
for r in 1 2; do
  one "c2 default" X=1 -- --config configs/lgd_retinanet_r50.yaml --batch-per-gpu 8
  one "c2 GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8 -- --config configs/lgd_retinanet_r50.yaml --batch-per-gpu 8
  one "c2 GPU_MAX_HW_QUEUES=8 LGD_SIDE_STREAMS=force" GPU_MAX_HW_QUEUES=8 LGD_SIDE_STREAMS=force -- --config configs/lgd_retinanet_r50.yaml --batch-per-gpu 8
  one "c2 LGD_SIDE_STREAMS=0" LGD_SIDE_STREAMS=0 -- --config configs/lgd_retinanet_r50.yaml --batch-per-gpu 8
  one "c3 default" X=1 -- --config configs/lgd_fcos_r50.yaml --batch-per-gpu 16
  one "c3 LGD_HEAD_STREAMS=1" LGD_HEAD_STREAMS=1 -- --config configs/lgd_fcos_r50.yaml --batch-per-gpu 16
done 2>&1 | tee $O/new_defaults.txt
