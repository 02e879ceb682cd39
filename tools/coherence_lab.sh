#!/bin/bash
# lab builds for the cross-XCD visibility diagnosis (tools/conv_stage_probe.py): A = wino6_out reads M with plain loads, B = gemm3 stores C write-through
# (sc0 sc1), C = gemm3 ends every tile with __threadfence(), D = gemm3 stores nt
cd $(dirname $0)/..
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mk() { name=$1; src=$2; shift 2
  OBJS=$(ls build/obj/*.o | grep -v "/$src.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c lgd_amd/csrc/$src.hip -o /tmp/coh_$name.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $OBJS /tmp/coh_$name.o -o tools/lab/liblgd_coh_$name.so && echo built $name; }
[[ -n $ONLY ]] || mk A winograd6 -DLGD_W6_SLAB_PLAIN_LOADS
[[ -n $ONLY ]] || mk B gemm3 -DLGD_GEMM3_STORE_AUX=17
[[ -n $ONLY ]] || mk C gemm3 -DLGD_GEMM3_END_FENCE=1
[[ -n $ONLY ]] || mk D gemm3 -DLGD_GEMM3_STORE_AUX=2
[[ -n $ONLY ]] || mk E winograd6 -DLGD_W6_SLAB_SYS_LOADS
[[ -n $ONLY ]] || mk F h2 -DLGD_H2_STORE_AUX=17
mk N winograd6 -fno-slp-vectorize
