"""run-to-run and fused-vs-two-pass gradient differences of the meta-arch (diagnostic)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import config, ops  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "winograd"
ops.conv3x3_backend(winograd=(backend == "winograd"), min_tiles=0)
cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
model = build_model(cfg).train()
model.distill_flag = 1
data = synthetic_batch(2, 256, 320, 5, seed=5)


def run(fused):
    model.fused_head_pass = fused
    model.student.loss_normalizer = torch.tensor(100.0, device="cuda")
    model.zero_grad(set_to_none=True)
    losses = model(data)
    sum(losses.values()).backward()
    return {k: float(v.detach()) for k, v in losses.items()}, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


runs = {"A1": run(True), "A2": run(True), "B1": run(False), "B2": run(False)}
print("losses", {k: v[0] for k, v in runs.items()})
names = list(runs["A1"][1])
groups = {}
for n in names:
    g = ".".join(n.split(".")[:3])
    groups.setdefault(g, []).append(n)
for g, ns in groups.items():
    w = {}
    for a, b in (("A1", "A2"), ("B1", "B2"), ("A1", "B1")):
        w[a + b] = max(rel(runs[a][1][n], runs[b][1][n]) for n in ns)
    print("%-45s run-to-run fused %.1e  two-pass %.1e | fused vs two-pass %.1e" % (g, w["A1A2"], w["B1B2"], w["A1B1"]))
