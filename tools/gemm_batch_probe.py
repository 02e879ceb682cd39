"""Every lgd_gemm_batch launch of one teacher forward + backward (config 2 boxes: 8 images x 11 rows): problem shapes and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import config, ops, hip
from lgd_amd.data import synthetic_batch
from lgd_amd.distillator import build_model
from lgd_amd.engine import Trainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda"])
tr = Trainer(cfg, build_model(cfg))
data = synthetic_batch(8, 800, 1333, 10, seed=1, pin=True)
d = cfg.MODEL.DISTILLATOR
it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
for i in range(3):
    tr.step(data, it0 + i)
torch.cuda.synchronize()
log = []
orig = ops._gemm_batch


def timed(problems):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(problems); e1.record()
    log.append(([(p.M, p.N, p.K) for p in problems], e0, e1))


ops._gemm_batch = timed
tr.step(data, it0 + 3)
torch.cuda.synchronize()
tot = 0.0
for shapes, e0, e1 in log:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    print("%7.1f us  %s" % (us, shapes))
print("launches", len(log), "total %.1f us" % tot)
