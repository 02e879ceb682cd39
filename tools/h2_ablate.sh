#!/bin/bash
# lab builds of the kernel library with parts of csrc/h2.hip's forward product compiled out one at a time (LGD_H2_ABL, results are garbage),
# run HERE (hipcc cross-compiles); then on the GPU box: python tools/h2_rounds.py --lib tools/lab/liblgd_h2abl_N.so T...
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" > /dev/null
OBJS=$(ls build/obj/*.o | grep -v "/h2.o")
for a in ${@:-1 2 3 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLGD_H2_ABL=$a -c lgd_amd/csrc/h2.hip -o /tmp/h2_abl_$a.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $OBJS /tmp/h2_abl_$a.o -o tools/lab/liblgd_h2abl_$a.so && echo built $a
done
