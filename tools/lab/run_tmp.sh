timeout 2400 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/pytest_gpu_full.log; grep -E "passed|failed" gpurun_out/pytest_gpu_full.log | tail -3
