"""Host time of the sections of Trainer.step (no synchronisation inside): forward / backward / clip / optimizers / bookkeeping."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import config  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402
from lgd_amd.engine import Trainer  # noqa: E402

yaml = sys.argv[1] if len(sys.argv) > 1 else "lgd_retinanet_r101"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = config.setup_cfg(os.path.join(ROOT, "configs", yaml + ".yaml"), ["MODEL.DEVICE", "cuda"])
tr = Trainer(cfg, build_model(cfg))
data = synthetic_batch(B, 800, 1333, 10, seed=1, device="cuda")
d = cfg.MODEL.DISTILLATOR
it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
for i in range(5):
    tr.step(data, it0 + i)
torch.cuda.synchronize()
acc = {}


def lap(name, t):
    now = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + now - t
    return now


for i in range(n):
    torch.cuda.synchronize()      # every section starts with an EMPTY queue: pure host cost, no back-pressure
    t = time.perf_counter()
    it = it0 + 5 + i
    tr.set_phase(it)
    t = lap("set_phase", t)
    loss_dict = tr.model(data)
    losses = sum(loss_dict.values())
    t = lap("forward", t)
    torch.cuda.synchronize(); t = time.perf_counter()
    tr.stu_optimizer.zero_grad(set_to_none=True); tr.tea_optimizer.zero_grad(set_to_none=True)
    t = lap("zero_grad", t)
    losses.backward()
    t = lap("backward", t)
    torch.cuda.synchronize(); t = time.perf_counter()
    if tr.clip.ENABLED:
        tr._clip()
    t = lap("clip", t)
    tr.stu_optimizer.step(); tr.tea_optimizer.step()
    t = lap("optimizers", t)
    tr.stu_scheduler.step(); tr.tea_scheduler.step()
    t = lap("schedulers", t)
    vals = torch.stack([v.detach() for v in loss_dict.values()])
    t = lap("loss stack", t)
torch.cuda.synchronize()
print("%s B=%d, host ms per step by section (queue drained before each group):" % (yaml, B))
for k, v in acc.items():
    print("  %-12s %7.2f" % (k, 1e3 * v / n))
print("  %-12s %7.2f" % ("total", 1e3 * sum(acc.values()) / n))
