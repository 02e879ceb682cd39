"""host-side time to ISSUE one training step (no sync inside) vs its GPU time: how far the step is from launch-bound."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import config  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402
from lgd_amd.engine import Trainer  # noqa: E402

yaml = sys.argv[1] if len(sys.argv) > 1 else "lgd_retinanet_r50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = config.setup_cfg(os.path.join(ROOT, "configs", yaml + ".yaml"), ["MODEL.DEVICE", "cuda"])
tr = Trainer(cfg, build_model(cfg))
data = synthetic_batch(B, 800, 1333, 10, seed=1, pin=True)
for i in range(5):
    tr.step(data, 40000 + i)
torch.cuda.synchronize()
issue, total = [], []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(data, 40005 + i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append(t1 - t0)
    total.append(t2 - t0)
print(yaml, "B=%d:" % B, "host issue %.1f ms/step, issue+drain %.1f ms/step (GPU idle at start of each step in this measurement)"
      % (1e3 * sum(issue) / len(issue), 1e3 * sum(total) / len(total)))
