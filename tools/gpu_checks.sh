#!/bin/bash
# The round's GPU recipe (run on the MI355X box, e.g. `gpurun --timeout 5400 -- 'bash tools/gpu_checks.sh all'`); results land in
# gpurun_out/, the ones worth judging are copied to profiles/ by hand.
#   tests    : pytest -m gpu
#   bench    : the driver's bench line + BASELINE configs 3 / 4 / 5
#   profile  : steady-state kernel stats (difference of two rocprofv3 --kernel-trace --stats runs: one-time find trials cancel)
#   pmc      : HBM traffic per hand-written kernel (separate FETCH_SIZE / WRITE_SIZE passes, tools/pmc_bench.sh)
#   tune     : TunableOp pass for GEMM shapes missing from lgd_amd/tuning/tunableop_gfx950.csv (all four configs), merged in place
#   traject  : 42-step loss trajectory of the shipped path vs two head passes vs the library convolutions
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out; mkdir -p $O
what=${1:-all}
CONFIGS=("configs/lgd_fcos_r50.yaml 16 config3_fcos_r50_b16" "configs/lgd_retinanet_r101.yaml 2 config4_r101_b2_per_gpu" "configs/lgd_retinanet_r101_dcnv2.yaml 2 config5_r101_dcnv2_b2")
if [[ $what == all || $what == tune ]]; then
  cp lgd_amd/tuning/tunableop_gfx950.csv $O/tunable0.csv
  for c in "configs/lgd_retinanet_r50.yaml 8 x" "${CONFIGS[@]}"; do set -- $c
    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$O/tunable.csv \
      timeout 1500 python bench.py --config $1 --batch-per-gpu $2 --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/tune.log 2>&1
  done
  python tools/merge_tunable.py lgd_amd/tuning/tunableop_gfx950.csv lgd_amd/tuning/tunableop_gfx950.csv $O/tunable0.csv
  cp lgd_amd/tuning/tunableop_gfx950.csv $O/tunableop_gfx950_merged.csv
fi
if [[ $what == all || $what == tests ]]; then
  timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1
  grep -E "passed|failed|^FAILED" $O/pytest_gpu.log | tail -5
fi
if [[ $what == all || $what == bench ]]; then
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo
  for c in "${CONFIGS[@]}"; do set -- $c
    timeout 900 python bench.py --config $1 --batch-per-gpu $2 --no-cpu-baseline > $O/bench_$3.json 2> $O/bench_$3.err; head -c 200 $O/bench_$3.json; echo
  done
fi
if [[ $what == all || $what == profile ]]; then
  (cd /tmp && export TMPDIR=/tmp
   for n in 5 25; do rm -rf /tmp/prof_$n
     timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python $R/bench.py --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass > /dev/null 2>&1
     cp $(ls /tmp/prof_$n/*/*kernel_stats.csv | head -1) $R/$O/kernel_stats_steps$n.csv
   done)
  python tools/prof_diff.py $O/kernel_stats_steps5.csv $O/kernel_stats_steps25.csv 20 $O/bench_rocprofv3_steady_state.csv
  # ... and the same with the step's side streams off (bench.py --one-stream): a kernel's OWN duration, what bench.py's roofline objects are priced on
  (cd /tmp && export TMPDIR=/tmp
   for n in 5 25; do rm -rf /tmp/prof1_$n
     timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_$n -- python $R/bench.py --one-stream --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass > /dev/null 2>&1
     cp $(ls /tmp/prof1_$n/*/*kernel_stats.csv | head -1) $R/$O/kernel_stats_one_stream_steps$n.csv
   done)
  python tools/prof_diff.py $O/kernel_stats_one_stream_steps5.csv $O/kernel_stats_one_stream_steps25.csv 20 $O/bench_rocprofv3_steady_state_one_stream.csv
fi
if [[ $what == all || $what == pmc ]]; then bash tools/pmc_bench.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log; fi
if [[ $what == all || $what == traject ]]; then bash tools/trajectory_check.sh; fi
