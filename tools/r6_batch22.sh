#!/bin/bash
# round 6: (a) which fork loses when every stream has a hardware queue of its own (GPU_MAX_HW_QUEUES >= 5: 70 ms against 51.6); (b) config 5 kernel table + packed-fp32 A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r6b22; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for q in 4 8; do GPU_MAX_HW_QUEUES=$q timeout 900 python tools/fork_subsets.py 2>$O/err_$q.txt | tee $O/fork_subsets_hwq$q.txt; done
C5="--config $R/configs/lgd_retinanet_r101_dcnv2.yaml --batch-per-gpu 2 --multiscale"
(cd /tmp && export TMPDIR=/tmp
 for n in 3 13; do rm -rf /tmp/p5_$n
   timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5_$n -- python $R/bench.py $C5 --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass > /tmp/p5_$n.log 2>&1 || tail -5 /tmp/p5_$n.log
   cp $(ls /tmp/p5_$n/*/*kernel_stats.csv | head -1) $R/$O/config5_kernel_stats_steps$n.csv
 done)
python tools/prof_diff.py $O/config5_kernel_stats_steps3.csv $O/config5_kernel_stats_steps13.csv 10 $O/config5_rocprofv3_steady_state.csv | tail -3
one() { lab=$1; shift
  timeout 600 python bench.py $C5 --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>$O/err.txt \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s ms/step %.2f value %.2f' % ('$lab', d['ms_per_step'], d['value']))" || tail -3 $O/err.txt; }
one "c5 no packed fp32 (shipped)" | tee $O/c5_packed_ab.txt
LGD_PACKED_FP32=1 timeout 600 python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_packed.log 2>&1
LGD_PACKED_FP32=1 one "c5 packed fp32 build" | tee -a $O/c5_packed_ab.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(force=True)" > $O/build_back.log 2>&1
one "c5 no packed fp32 (shipped) again" | tee -a $O/c5_packed_ab.txt
