"""Per-kernel means of SQ wave-state counters from a rocprofv3 --pmc counter_collection.csv: where the waves of a kernel spend their cycles
(SQ_WAIT_ANY = parked on s_waitcnt / barrier, SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three add up to SQ_WAVE_CYCLES;
units: quad-cycles -- /opt/skills/guides/MI355X_MICROARCH.md, rocprofv3 PMC slots).   python tools/pmc_sq.py counter_collection.csv [name filter]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for r in rows:
    name = r.get("Kernel_Name") or r.get("Kernel Name")
    if flt and not re.search(flt, name):
        continue
    short = re.sub(r"\(.*", "", name).replace("lgd::(anonymous namespace)::", "").replace("lgd::", "").replace("void ", "")
    acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (short, r.get("Dispatch_Id"))
    if key not in seen:
        seen.add(key)
        cnt[short] += 1
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"]
print("%-44s %6s %9s | of a wave's cycles: %8s %8s %8s %8s | %10s" % ("kernel", "n", "waves", "parked", "stalled", "issuing", "(valu)", "valu/wave"))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    w = v.get("SQ_WAVES", 0) or 1
    print("%-44s %6d %9.0f | %26.1f%% %7.1f%% %7.1f%% %7.1f%% | %10.0f" % (k[:44], cnt[k], w / cnt[k], 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc,
                                                                    100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_VALU", 0) / wc, v.get("SQ_INSTS_VALU", 0) / w))
