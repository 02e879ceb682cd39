#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/overlap_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02_overlap_probe.log
