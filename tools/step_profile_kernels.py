"""Steady-state device-time breakdown of one training step by kernel (torch profiler, after warm-up)."""
import collections
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

cfg = config.setup_cfg(os.path.join(ROOT, "configs", sys.argv[1] if len(sys.argv) > 1 else "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda"])
model = build_model(cfg)
tr = Trainer(cfg, model)
data = synthetic_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 8, 800, 1333, 10, seed=1)
for i in range(4):
    tr.step(data, 40000 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as p:
    tr.step(data, 40004)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in p.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        agg[e.name][0] += 1
        agg[e.name][1] += e.device_time
tot = sum(v[1] for v in agg.values())
print("total device time %.2f ms" % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%8.2f ms %5.1f%% n=%4d avg %8.1f us  %s" % (v[1] / 1e3, 100 * v[1] / tot, v[0], v[1] / v[0], k[:110]))
