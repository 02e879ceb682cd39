#!/bin/bash
# round 6: fork subsets at config 3 under 4 / 8 hardware queues; the weight-gradient fork (LGD_DW_STREAM) at config 2 under 4 / 8 queues
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r6b23; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for q in 4 8; do GPU_MAX_HW_QUEUES=$q timeout 900 python tools/fork_subsets.py --config configs/lgd_fcos_r50.yaml --batch 16 --steps 10 2>$O/err_c3_$q.txt | tee $O/fork_subsets_c3_hwq$q.txt; done
one() { lab=$1; shift; envs=(); while [[ $1 != -- ]]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py "$@" --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>$O/err.txt \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s ms/step %.2f value %.2f' % ('$lab', d['ms_per_step'], d['value']))" || tail -3 $O/err.txt; }
for r in 1 2; do for q in 4 8; do for dw in 0 1; do
  one "c2 HWQ=$q LGD_HEAD_STREAMS=0 LGD_DW_STREAM=$dw" GPU_MAX_HW_QUEUES=$q LGD_HEAD_STREAMS=0 LGD_DW_STREAM=$dw -- --config configs/lgd_retinanet_r50.yaml --batch-per-gpu 8
done; done; done 2>&1 | tee $O/dw_stream_c2.txt
