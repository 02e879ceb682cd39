#!/bin/bash
# round 6: where the waves of the transform kernels spend their cycles (SQ wave-state counters over one pyramid convolution, fwd + bwd)
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r6b31; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcsq
LGD_WINO_AB_TILES=6 timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d /tmp/pmcsq -- python $R/tools/wino_tile_ab.py 8 > /tmp/pmcsq.log 2>&1 || tail -5 /tmp/pmcsq.log
F=$(ls /tmp/pmcsq/*/*counter_collection.csv | head -1)
head -2 $F
python $R/tools/pmc_sq.py $F "wino6|h2_" | tee $R/$O/sq_wave_states.txt
rm -rf /tmp/pmcsq2
LGD_WINO_AB_TILES=6 timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmcsq2 -- python $R/tools/wino_tile_ab.py 8 > /tmp/pmcsq2.log 2>&1 || tail -5 /tmp/pmcsq2.log
F2=$(ls /tmp/pmcsq2/*/*counter_collection.csv | head -1)
python - "$F2" <<'PY' | tee $R/$O/sq_insts.txt
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if not re.search("wino6|h2_", name): continue
    s = re.sub(r"\(.*", "", name).replace("lgd::(anonymous namespace)::", "").replace("lgd::", "").replace("void ", "")
    acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
    if (s, r["Dispatch_Id"]) not in seen: seen.add((s, r["Dispatch_Id"])); n[s] += 1
for k, v in acc.items():
    print("%-44s n %4d " % (k[:44], n[k]) + "  ".join("%s %.3g" % (c, x / n[k]) for c, x in sorted(v.items())))
PY
