#!/bin/bash
# SQ issue / wait counters of the kernels tools/kbench.py runs (HBM-cold, config-2 shapes): bash tools/sq_counters_kbench.sh [kernel regex]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/sq_pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU \
  --output-format csv -d /tmp/sq_pmc -- python $R/tools/kbench.py --iters 4 > /tmp/sq_pmc.log 2>&1 || { tail -5 /tmp/sq_pmc.log; exit 1; }
F=$(ls /tmp/sq_pmc/*/*counter_collection.csv | head -1)
python - "$F" "${1:-.}" <<'PY'
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1], newline="")):
    n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
    if "lgd::" not in n or not re.search(sys.argv[2], n):
        continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[n] += 1
print("%-44s %5s %10s %6s %6s %6s %6s %10s %10s" % ("kernel", "n", "wave_qcyc", "wait%", "stall%", "act%", "valu%", "VALU", "SALU"))
for n, c in sorted(acc.items()):
    k = max(cnt[n], 1); w = c["SQ_WAVE_CYCLES"] or 1
    print("%-44s %5d %10.3g %6.1f %6.1f %6.1f %6.1f %10.3g %10.3g" % (n[:44], k, w / k, 100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w,
          100 * c["SQ_ACTIVE_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_VALU"] / w, c["SQ_INSTS_VALU"] / k, c["SQ_INSTS_SALU"] / k))
PY
