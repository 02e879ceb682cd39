#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for part in probe distill; do
  timeout 600 python tools/diag_grad_chain.py c1_ctx_stuguided winograd $part 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r02_diag_grad_chain.log
timeout 600 python tools/diag_grad_chain.py c1_ctx_stuguided library probe 2>&1 | grep -v amdgpu.ids >> gpurun_out/r02_diag_grad_chain.log
timeout 600 python tools/diag_determinism.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_diag_determinism.log
cat gpurun_out/r02_diag_grad_chain.log gpurun_out/r02_diag_determinism.log
