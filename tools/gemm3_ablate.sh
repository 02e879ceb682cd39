#!/bin/bash
# lab builds of the kernel library with csrc/gemm3.hip's staging parts compiled out one at a time (LGD_GEMM3_ABL, results are garbage),
# run HERE (hipcc cross-compiles); then on the GPU box: python tools/gemm3_rounds.py --lib tools/lab/liblgd_abl_N.so T...
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" > /dev/null
OBJS=$(ls build/obj/*.o | grep -v gemm3.o)
for a in ${@:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLGD_GEMM3_ABL=$a -c lgd_amd/csrc/gemm3.hip -o /tmp/gemm3_abl_$a.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $OBJS /tmp/gemm3_abl_$a.o -o tools/lab/liblgd_abl_$a.so && echo built $a
done
