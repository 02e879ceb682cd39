#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6b7; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
{ for e in "X=1" "AMD_OPT_FLUSH=0" "GPU_MAX_HW_QUEUES=1" "GPU_MAX_HW_QUEUES=8" "DEBUG_HIP_DYNAMIC_QUEUES=0" "ROC_SYSTEM_SCOPE_SIGNAL=0"; do echo "=== env $e"; env $e python tools/conv_stage_probe.py 2>&1 | grep -E "^y |library"; env $e python tools/conv_stage_probe.py --h2 2>&1 | grep -E "^y "; done; } 2>&1 | tee $O/conv_stage_probe_env.log
bash tools/ab_envval.sh AMD_OPT_FLUSH "1 0" configs/lgd_retinanet_r50.yaml 8 2 2>&1 | tee $O/ab_opt_flush_c2.txt
