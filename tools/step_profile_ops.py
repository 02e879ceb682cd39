"""aten-op level view of one steady-state step: self device time per (op, input shapes), library GEMM/conv ops excluded."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from lgd_amd import config  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402
from lgd_amd.engine import Trainer  # noqa: E402

cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda"])
model = build_model(cfg)
tr = Trainer(cfg, model)
data = synthetic_batch(8, 800, 1333, 10, seed=1)
for i in range(4):
    tr.step(data, 40000 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as p:
    tr.step(data, 40004)
    torch.cuda.synchronize()
ka = p.key_averages(group_by_input_shape=True)
skip = ("conv", "bmm", "mm", "_Conv3x3", "autograd::engine", "Optimizer")
rows = [e for e in ka if e.self_device_time_total > 0 and not any(s in e.key for s in skip)]
tot = sum(e.self_device_time_total for e in rows)
print("non-GEMM/conv aten ops: %.2f ms" % (tot / 1e3))
for e in sorted(rows, key=lambda e: -e.self_device_time_total)[:60]:
    print("%7.2f ms n=%3d %-40s %s" % (e.self_device_time_total / 1e3, e.count, e.key[:40], str(e.input_shapes)[:110]))
aten = [e for e in rows if e.key.startswith("aten::")]
print("aten:: ops (torch's own elementwise / reduction / copy kernels): %.2f ms" % (sum(e.self_device_time_total for e in aten) / 1e3))
for e in sorted(aten, key=lambda e: -e.self_device_time_total)[:50]:
    print("%7.3f ms n=%3d %-28s %s" % (e.self_device_time_total / 1e3, e.count, e.key[:28], str(e.input_shapes)[:150]))
