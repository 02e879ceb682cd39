python -m pytest tests/test_kernels_gpu.py -q -x -k "stem" 2>&1 | tail -2
python tools/lab/stem_time.py 2>&1 | tail -1
