"""steady-state per-kernel time of bench.py from TWO rocprofv3 --kernel-trace --stats runs that differ only in --steps: the
one-time work (MIOpen's find trials, warm-up) cancels in the difference.
    python tools/prof_diff.py short_kernel_stats.csv long_kernel_stats.csv d_steps out.csv"""
import csv
import sys


def load(f):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}


a, b, d = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for n, (c1, t1) in b.items():
    c0, t0 = a.get(n, (0, 0.0))
    if c1 - c0 > 0 and t1 - t0 > 0:
        rows.append((n, (c1 - c0) / d, (t1 - t0) / d / 1e6, (t1 - t0) / (c1 - c0) / 1e3))
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
with open(sys.argv[4], "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerStep", "MsPerStep", "AverageUs", "Percentage"])
    for n, c, ms, us in rows:
        w.writerow([n, "%.2f" % c, "%.4f" % ms, "%.2f" % us, "%.2f" % (100 * ms / tot)])
print("steady state: %.2f ms of kernel time per step over %d kernel names" % (tot, len(rows)))
