#!/bin/bash
# round 6: (a) config 5 got slower than round 5's line (42.7 vs 39.8 ms) -- which switch, which kernels; (b) hardware-queue count sweep at config 2
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r6b21; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
one() { # label, env..., -- args
  lab=$1; shift; envs=(); while [[ $1 != -- ]]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 600 python bench.py "$@" --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>$O/err.txt \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s ms/step %.2f value %.2f' % ('$lab', d['ms_per_step'], d['value']))" || tail -3 $O/err.txt
}
C5="--config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --multiscale"
for r in 1 2; do
  one "c5 shipped" X=1 -- $C5
  one "c5 LGD_LIBRARY_ORDER=0" LGD_LIBRARY_ORDER=0 -- $C5
  one "c5 LGD_SIDE_STREAMS=0" LGD_SIDE_STREAMS=0 -- $C5
  one "c5 fixed size (no --multiscale)" X=1 -- --config configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2
done 2>&1 | tee $O/c5_switches.txt
(cd /tmp && export TMPDIR=/tmp
 for n in 3 13; do rm -rf /tmp/p5_$n
   timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5_$n -- python $R/bench.py $C5 --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass > /dev/null 2>&1
   cp $(ls /tmp/p5_$n/*/*kernel_stats.csv | head -1) $R/$O/config5_kernel_stats_steps$n.csv
 done)
python tools/prof_diff.py $O/config5_kernel_stats_steps3.csv $O/config5_kernel_stats_steps13.csv 10 $O/config5_rocprofv3_steady_state.csv | tail -3
C2="--config configs/lgd_retinanet_r50.yaml --batch-per-gpu 8"
for r in 1 2; do for q in 1 2 3 4 5 6; do one "c2 GPU_MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q -- $C2; done; done 2>&1 | tee $O/hwq_c2.txt
C4="--config configs/lgd_retinanet_r101.yaml --batch-per-gpu 2"
for q in 1 2 4; do one "c4 GPU_MAX_HW_QUEUES=$q" GPU_MAX_HW_QUEUES=$q -- $C4; done 2>&1 | tee $O/hwq_c4.txt
