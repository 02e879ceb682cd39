#!/bin/bash
# round 6: where do the 19 ms go when the head fork gets a hardware queue of its own -- slower kernels or idle gaps?  kernel stats at GPU_MAX_HW_QUEUES=8, head fork on / off
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD; O=gpurun_out/r6b25; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
(cd /tmp && export TMPDIR=/tmp
 for h in 0 1; do for n in 3 13; do rm -rf /tmp/pq_$h_$n
   GPU_MAX_HW_QUEUES=8 LGD_HEAD_STREAMS=$h timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq_${h}_$n -- python $R/bench.py --steps $n --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-pass > /tmp/pq_${h}_$n.log 2>&1 || tail -5 /tmp/pq_${h}_$n.log
   cp $(ls /tmp/pq_${h}_$n/*/*kernel_stats.csv | head -1) $R/$O/hwq8_head${h}_kernel_stats_steps$n.csv
   grep -o '"ms_per_step": [0-9.]*' /tmp/pq_${h}_$n.log | head -1
 done; done
 # the trace itself of the 13-step head-on run: start / end per kernel and queue (for the gaps)
 f=$(ls /tmp/pq_1_13/*/*kernel_trace.csv | head -1); python - "$f" > $R/$O/hwq8_head1_trace_summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
q = collections.Counter(r["Queue_Id"] for r in rows)
print("kernels", len(rows), "span ms", (t1 - t0) / 1e6, "queues", dict(q))
# busy time (union of intervals) over the last 40% of the trace (steady state)
cut = t0 + int(0.6 * (t1 - t0))
iv = sorted((max(int(r["Start_Timestamp"]), cut), int(r["End_Timestamp"])) for r in rows if int(r["End_Timestamp"]) > cut)
busy, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("steady part: span ms %.2f, some kernel running ms %.2f (%.1f%%), sum of kernel durations ms %.2f" % ((t1 - cut) / 1e6, busy / 1e6, 100.0 * busy / (t1 - cut), sum(e - s for s, e in iv) / 1e6))
PY
)
for h in 0 1; do python tools/prof_diff.py $O/hwq8_head${h}_kernel_stats_steps3.csv $O/hwq8_head${h}_kernel_stats_steps13.csv 10 $O/hwq8_head${h}_steady_state.csv | tail -1; done
cat $O/hwq8_head1_trace_summary.txt
