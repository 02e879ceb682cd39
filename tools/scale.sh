#!/bin/bash
# Weak-scaling curve of bench.py on ONE node: N = 1, 2, 4, 8 (or the list given), one process per GPU over RCCL.
#   tools/scale.sh [-o out.json] [N ...] [-- extra bench.py flags]
# bench.py launches its own ranks (lgd_amd/launch.py), so nothing here knows about the launcher.  Every line bench.py prints is kept;
# the summary holds value / ms_per_step / rccl_ranks per N.  Efficiency is the reader's to compute (value_N / (N * value_1)).
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/scale.json
ns=()
extra=()
while [ $# -gt 0 ]; do
  case "$1" in
    -o) out=$2; shift 2;;
    --) shift; extra=("$@"); break;;
    *) ns+=("$1"); shift;;
  esac
done
[ ${#ns[@]} -eq 0 ] && ns=(1 2 4 8)
have=$(python -c 'import torch; print(torch.cuda.device_count())')
mkdir -p "$(dirname "$out")"
: > "$out.lines"
for n in "${ns[@]}"; do
  if [ "$n" -gt "$have" ]; then echo "[scale] skip N=$n: $have GPU(s) visible" >&2; continue; fi
  echo "[scale] N=$n" >&2
  python bench.py --gpus "$n" --steps 20 --warmup 5 --no-cpu-baseline "${extra[@]}" >> "$out.lines" || echo "[scale] N=$n failed" >&2
done
python - "$out" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1] + ".lines") if l.strip().startswith("{")]
summary = [{"n_gpus": r["n_gpus"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "rccl_ranks": r.get("rccl_ranks"),
            "host_threads_pinned": r.get("host_threads_pinned"), "workload": r["config"]["workload"]} for r in rows]
json.dump({"scaling": "weak", "curve": summary}, open(sys.argv[1], "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
