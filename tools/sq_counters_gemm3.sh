#!/bin/bash
# SQ issue / wait / LDS / MFMA counters of the lab GEMM (tools/gemm3_lab.py): bash tools/sq_counters_gemm3.sh [args to gemm3_lab.py]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU"
P2="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
P3="SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1)); rm -rf /tmp/sq_g3_$i
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/sq_g3_$i -- python $R/tools/gemm3_lab.py --reps 2 "$@" > /tmp/sq_g3_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/sq_g3_$i.log; continue; }
  F=$(ls /tmp/sq_g3_$i/*/*counter_collection.csv | head -1)
  python - "$F" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1], newline="")):
    n = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    if "gemm3" not in n and "Cijk" not in n:
        continue
    n = n[:60]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[n] += 1
for n, c in sorted(acc.items()):
    k = max(cnt[n], 1)
    print("%-60s n=%d  " % (n, k) + "  ".join("%s=%.4g" % (a.replace("SQ_", ""), v / k) for a, v in sorted(c.items())))
PY
done
