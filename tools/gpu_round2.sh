#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | tail -120 > gpurun_out/r02_pytest2.log
echo "pytest rc=$?" >> gpurun_out/r02_pytest2.log
timeout 900 python tools/chunk_bench.py > gpurun_out/r02_chunk_bench.log 2>&1
tail -5 gpurun_out/r02_pytest2.log
cat gpurun_out/r02_chunk_bench.log
