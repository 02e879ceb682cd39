python -m pytest tests/test_kernels_gpu.py -q -x -k "stem" 2>&1 | tail -2
python tools/lab/stem_time.py 2>&1 | tail -1
for wv in 3 4; do export LGD_HIPCC_DEFS="-DLGD_STEM7_WAVES=$wv"; touch lgd_amd/csrc/stem.hip; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; python tools/lab/stem_time.py 2>&1 | tail -1; done
