"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected separately) per hand-written kernel.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> [<pmc_traffic.json>] [note]
HBM bytes per launch = 2 * FETCH_SIZE KB (gfx950: FETCH_SIZE counts 128-B read requests as 64 B,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE KB, averaged over all launches of the kernel in the profiled command
(the same launch mix bench.py times)."""
import collections
import csv
import json
import re
import sys


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            if "lgd::" not in name:
                continue
            name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
            name = re.sub(r"\(.*$", "", name)
            name = re.sub(r"^gemm3_kernel<.*", "gemm3_kernel", name.replace("lgd::", ""))
            name = name if name.startswith("lgd::") else "lgd::" + name
            acc[name][0] += 1
            acc[name][1] += float(r["Counter_Value"])
    return acc


# profiler kernel name (template instance) -> name used by the library's launch timer / bench.py
TIMER_NAMES = {
    "gn_stats_kernel<0>": ["gn_stats_kernel"], "gn_stats_kernel<1>": ["gn_bwd_stats_kernel"],
    "gn_apply_kernel<0>": ["gn_apply_kernel"], "gn_apply_kernel<1>": ["gn_bwd_apply_kernel"],
    "gn_finalize_kernel<0>": ["gn_finalize_kernel"], "gn_finalize_kernel<1>": ["gn_bwd_finalize_kernel"],
    "ctx_relu_kernel<0>": ["ctx_relu_kernel"], "ctx_relu_kernel<1>": ["ctx_relu_bwd_kernel"],
    "focal_kernel<0>": ["focal_fwd_kernel"], "focal_kernel<1>": ["focal_bwd_kernel"], "focal_kernel<2>": ["focal_fwd_grad_kernel"],
    "rowln_kernel<0>": ["rowln_kernel"], "rowln_kernel<1>": ["rowln_bwd_kernel"],
    "gg_stats_kernel<0>": ["gn_group_stats_kernel"], "gg_stats_kernel<1>": ["gn_group_bwd_stats_kernel"],
    "gg_apply_kernel<0>": ["gn_group_apply_kernel"], "gg_apply_kernel<1>": ["gn_group_bwd_apply_kernel"],
    "gg_finalize_kernel<0>": ["gn_group_finalize_kernel"], "gg_finalize_kernel<1>": ["gn_group_bwd_finalize_kernel"],
    "wino4_out_kernel": ["wino_out_kernel"], "wino6_out_kernel": ["wino_out_kernel"], "wino6_out_t_kernel": ["wino_out_t_kernel"],
    "bias_act_kernel<4>": ["bias_act_kernel"], "bias_act_kernel<1>": ["bias_act_kernel"],
    "relu_mask_kernel<4>": ["relu_mask_kernel"], "relu_mask_kernel<1>": ["relu_mask_kernel"], "wino4_out_t_kernel": ["wino_out_t_kernel"],
    "wino4_in_t_kernel": ["wino_in_t_kernel"],
}


def short(name):
    n = name.replace("lgd::", "")
    if n in TIMER_NAMES:
        return TIMER_NAMES[n]
    m = re.match(r"paint_kernel<([12]),", n)                                          # <MODE, SPLIT>
    if m:
        return ["gn_pool_bwd_apply_kernel" if m.group(1) == "1" else "box_paint_kernel"]
    m = re.match(r"box_pool_kernel<([01]),", n)                                       # <GN, CH>
    if m:
        return ["gn_pool_kernel" if m.group(1) == "1" else "box_sum_kernel"]
    if re.match(r"wino[46]_in_kernel<", n):                                           # <PRE[, H2]>
        return ["wino_in_kernel"]
    if re.match(r"wino6_out_t_kernel<", n):                                           # <H2>
        return ["wino_out_t_kernel"]
    m = re.match(r"wino[46]_in_t_kernel<(true|false)", n)                           # <FUSE[, H2]>
    if m:
        return ["wino_in_t_out_t_kernel" if m.group(1) == "true" else "wino_in_t_kernel"]
    base = re.sub(r"<.*$", "", n)
    return [{"stem_pool_pair_kernel": "stem_pool_kernel", "wino4_filter_fwd_kernel": "wino_filter_kernel", "wino6_filter_fwd_kernel": "wino_filter_kernel",
             "wino4_filter_bwd_kernel": "wino_filter_bwd_kernel", "wino6_filter_bwd_kernel": "wino_filter_bwd_kernel", "relu_bits_kernel": "relu_bits_kernel"}.get(base, base)]


def traffic_json(summary_csv, out_json, note):
    """per timer name: launch-weighted mean of (2*FETCH + WRITE) over the template instances it covers."""
    acc = collections.defaultdict(lambda: [0, 0.0])
    for line in open(summary_csv):
        if line.startswith("#") or line.startswith("kernel,"):
            continue
        k, rest = line.rsplit(",", 5)[0], line.strip().rsplit(",", 5)[1:]
        n, rd, wr = int(rest[0]), float(rest[3]), float(rest[4])
        for t in short(k):
            acc[t][0] += n
            acc[t][1] += n * (rd + wr) * 1e6
    import os
    steps = int(os.environ.get("STEPS_PROFILED", "0")) or None
    js = {t: {"hbm_bytes_per_launch": int(b / n), "launches": n, "steps_profiled": steps,
              "source": "%s (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; mean over the launches of: %s)"
                        % (summary_csv.split("/")[-1], note)} for t, (n, b) in sorted(acc.items())}
    json.dump(js, open(out_json, "w"), indent=1)


def main():
    fetch, write, out = sys.argv[1:4]
    tj = sys.argv[4] if len(sys.argv) > 4 else None
    note = sys.argv[5] if len(sys.argv) > 5 else ""
    F, Wr = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    rows = []
    for k in sorted(set(F) | set(Wr)):
        nf, sf = F.get(k, [0, 0.0])
        nw, sw = Wr.get(k, [0, 0.0])
        fkb, wkb = sf / max(nf, 1), sw / max(nw, 1)
        rows.append((k, max(nf, nw, 1), fkb, wkb, 2 * fkb * 1024 / 1e6, wkb * 1024 / 1e6))
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2); %s\n" % note)
        f.write("# gfx950: hbm_read_bytes = 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md, HBM)\n")
        f.write("kernel,launches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_read_MB_corrected,hbm_write_MB\n")
        for r in rows:
            f.write("%s,%d,%.1f,%.1f,%.1f,%.1f\n" % r)
    if tj:
        traffic_json(out, tj, note)


if __name__ == "__main__":
    if sys.argv[1] == "--from-summary":  # pmc_summary.py --from-summary <summary.csv> <pmc_traffic.json> [note]
        traffic_json(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    else:
        main()
