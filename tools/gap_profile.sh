#!/bin/bash
# GPU idle gaps inside the training step (GPU box): rocprofv3 --kernel-trace of a short bench run, then the largest gaps between
# consecutive kernels of the last steps with the kernels on either side.   usage: bash tools/gap_profile.sh [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/gap_prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_prof -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass "$@" > /tmp/gap_prof.log 2>&1 || tail -3 /tmp/gap_prof.log
F=$(ls /tmp/gap_prof/*/*kernel_trace.csv | head -1)
python - "$F" <<'PY' | tee $O/gap_profile.log
import csv, re, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1], newline=""))]
rows.sort()
n = len(rows)
rows = rows[int(n * 0.6):]          # the last steps (steady state)
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps.append((s1 - e0, n0, n1))
tot_gap = sum(max(g, 0) for g, _, _ in gaps)
short = lambda k: re.sub(r"\(.*$", "", re.sub(r"^void |at::native::|\(anonymous namespace\)::", "", k))[:70]
print("kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, tot_gap / 1e6, 100.0 * tot_gap / span))
hist = {}
for g, _, _ in gaps:
    b = "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    hist.setdefault(b, [0, 0]); hist[b][0] += 1; hist[b][1] += max(g, 0)
for b in ("<2us", "<5us", "<20us", "<100us", ">=100us"):
    if b in hist: print("  gaps %-7s n=%5d  total %.2f ms" % (b, hist[b][0], hist[b][1] / 1e6))
pairs = {}
for g, a, b in gaps:
    if g >= 5000:
        k = (short(a)[:40], short(b)[:40]); pairs.setdefault(k, [0, 0]); pairs[k][0] += 1; pairs[k][1] += g
print("gap time by (kernel before the gap, kernel after it), gaps >= 5 us:")
for (a, b), (n_, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %7.2f ms  n=%4d  %-40s -> %s" % (t / 1e6, n_, a, b))
print("largest gaps:")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us  after %-70s before %s" % (g / 1e3, short(a), short(b)))
PY
