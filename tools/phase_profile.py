"""Forward-phase / backward / optimizer split of one steady-state step with HIP events (methods wrapped in place)."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import config, ops  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402
from lgd_amd.engine import Trainer  # noqa: E402

cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda"])
model = build_model(cfg)
tr = Trainer(cfg, model)
data = synthetic_batch(8, 800, 1333, 10, seed=1)
acc = collections.OrderedDict()
pending = []


def wrap(obj, name, tag):
    fn = getattr(obj, name)
    call = fn.forward if isinstance(fn, torch.nn.Module) else fn

    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = call(*a, **k)
        e1.record()
        pending.append((tag, e0, e1))
        return out
    if isinstance(fn, torch.nn.Module):
        fn.forward = timed  # wrap the module's forward instead of replacing the child
        return
    setattr(obj, name, timed)


m = tr.raw_model
s = m.student
wrap(s, "raw_backbone", "fwd resnet")
rb = s.raw_backbone
wrap(rb, "stem", "  resnet stem")
wrap(rb.stem, "conv1", "    stem conv7x7+bias+relu")
for nm in ("res2", "res3", "res4", "res5"):
    wrap(rb, nm, "  resnet " + nm)
wrap(s, "backbone", "fwd fpn")
wrap(s, "predict", "fwd head (x2: student + teacher feats)")
wrap(s, "losses", "fwd detection losses (x2)")
wrap(m.teacher, "forward", "fwd dynamic teacher")
wrap(m, "distill", "fwd adapter + distill loss")
wrap(tr.stu_optimizer, "step", "optimizer (student)")
wrap(tr.tea_optimizer, "step", "optimizer (teacher)")
wrap(tr, "_clip", "clip")
wrap(tr, "model", "forward total")
for i in range(5):
    tr.step(data, 40000 + i)
torch.cuda.synchronize()
pending.clear()
N = 5
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for i in range(N):
    tr.step(data, 40005 + i)
t1.record()
torch.cuda.synchronize()
for tag, e0, e1 in pending:
    acc[tag] = acc.get(tag, 0.0) + e0.elapsed_time(e1)
total = t0.elapsed_time(t1) / N
print("step %.2f ms" % total)
for k, v in acc.items():
    print("  %-44s %7.2f ms" % (k, v / N))
fw = acc.get("forward total", 0) / N
op = sum(v for k, v in acc.items() if k.startswith("optimizer") or k == "clip") / N
print("  %-44s %7.2f ms" % ("backward (step - forward - optimizer)", total - fw - op))
