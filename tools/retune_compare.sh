#!/bin/bash
# Fresh TunableOp pass for one config (GPU box) next to the shipped table: per shape, the faster of the two entries wins.
#   bash tools/retune_compare.sh <yaml> <batch-per-gpu> <tag>      -> gpurun_out/tunable_fresh_<tag>.csv, gpurun_out/tunableop_min_<tag>.csv
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out; mkdir -p $O
cfg=$1; b=$2; tag=$3
rm -f $O/tunable_fresh_$tag.csv /tmp/fresh*.csv
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/fresh.csv LGD_TUNED_GEMM=0 \
  timeout 2400 python bench.py --config $cfg --batch-per-gpu $b --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-pass > $O/retune_$tag.log 2>&1
cp /tmp/fresh*.csv $O/tunable_fresh_$tag.csv 2>/dev/null || { tail -3 $O/retune_$tag.log; exit 1; }
python - $O/tunable_fresh_$tag.csv lgd_amd/tuning/tunableop_gfx950.csv $O/tunableop_min_$tag.csv <<'PY'
import sys
fresh, cur, out = sys.argv[1:4]
def load(f):
    v, rows = [], {}
    for line in open(f):
        line = line.strip()
        if not line: continue
        if line.startswith("Validator,"): v.append(line)
        else:
            op, shape, sol, t = line.split(",")
            rows[(op, shape)] = (sol, float(t))
    return v, rows
vf, rf = load(fresh); vc, rc = load(cur)
assert vf == vc, "validators differ"
better = same = new = 0; gain = 0.0
for k, (sol, t) in rf.items():
    if k not in rc: rc[k] = (sol, t); new += 1
    elif t < rc[k][1] * 0.97 and sol != rc[k][0]:
        gain += rc[k][1] - t; better += 1; rc[k] = (sol, t)
    else: same += 1
with open(out, "w") as fo:
    fo.write("\n".join(vc) + "\n")
    for (op, shape), (sol, t) in sorted(rc.items()): fo.write("%s,%s,%s,%s\n" % (op, shape, sol, t))
print("fresh %d shapes: %d new, %d faster by >3%% (sum of per-call gains %.3f ms), %d kept" % (len(rf), new, better, gain, same))
PY
