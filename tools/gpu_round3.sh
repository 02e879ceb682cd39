#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -s -k "fcos_targets or distill_loss_and_grads or edge_batch" > gpurun_out/r02_pytest3.log 2>&1
timeout 600 python tools/diag_headpass.py winograd > gpurun_out/r02_diag_headpass.log 2>&1
timeout 600 python tools/diag_headpass.py library >> gpurun_out/r02_diag_headpass.log 2>&1
timeout 900 python tools/chunk_bench.py > gpurun_out/r02_chunk_bench.log 2>&1
cat gpurun_out/r02_diag_headpass.log | grep -v amdgpu.ids
cat gpurun_out/r02_chunk_bench.log | grep -v amdgpu.ids
