#!/bin/bash
# round 6, GPU batch 18: the library without packed fp32 instructions -- the probes that failed, the stress run, the cost
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b18; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ for a in h2_fwd gemm2h conv; do timeout 200 python tools/conv_stage_probe.py --rounds 100 --aggressor $a 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo; timeout 200 python tools/conv_stage_probe.py --h2 --rounds 100 --aggressor $a 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo; done
  echo "--- control: the same library built WITH packed fp32 (tools/lab/liblgd_packed.so)"; timeout 200 python tools/conv_stage_probe.py --rounds 60 --aggressor h2_fwd --lib tools/lab/liblgd_packed.so 2>&1 | grep -E "^y |library" | tr '\n' ' '; echo
  timeout 300 python tools/fpn_race_probe.py 2>&1 | grep -v amdgpu.ids; } | tee $O/probes_after.log
timeout 900 python tools/stream_stress.py --steps 300 > $O/stress.json 2> $O/stress.err; tail -4 $O/stress.err | cut -c1-300
for r in 1 2; do for l in "" tools/lab/liblgd_packed.so; do
  env LGD_HIP_LIB=$l timeout 900 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-host-pass 2>$O/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=${l:-shipped (no packed fp32)}', 'ms/step %.2f' % d['ms_per_step'])" || tail -3 $O/ab.err
done; done | tee $O/ab_packed_c2.txt
