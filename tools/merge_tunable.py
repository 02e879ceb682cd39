"""Merge TunableOp result files (same validators) into lgd_amd/tuning/tunableop_gfx950.csv: later files win per (op, shape)."""
import sys

out, files = sys.argv[1], sys.argv[2:]
validators, rows = None, {}
for f in files:
    v = []
    for line in open(f):
        line = line.strip()
        if not line:
            continue
        if line.startswith("Validator,"):
            v.append(line)
        else:
            op, shape, rest = line.split(",", 2)
            rows[(op, shape)] = rest
    if validators is None:
        validators = v
    elif v != validators:
        raise SystemExit("validator mismatch in %s" % f)
with open(out, "w") as fo:
    fo.write("\n".join(validators) + "\n")
    for (op, shape), rest in sorted(rows.items()):
        fo.write("%s,%s,%s\n" % (op, shape, rest))
print(len(rows), "entries")
