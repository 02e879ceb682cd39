"""Host-side cost of ISSUING training steps (cProfile over N steps without synchronisation): where the Python time goes when the
step is launch-bound (2 images per GPU).  usage: python tools/host_profile.py [yaml-name] [batch] [steps]"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import config  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402
from lgd_amd.engine import Trainer  # noqa: E402

yaml = sys.argv[1] if len(sys.argv) > 1 else "lgd_retinanet_r101"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = config.setup_cfg(os.path.join(ROOT, "configs", yaml + ".yaml"), ["MODEL.DEVICE", "cuda"])
tr = Trainer(cfg, build_model(cfg))
data = synthetic_batch(B, 800, 1333, 10, seed=1, device="cuda")
d = cfg.MODEL.DISTILLATOR
it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
for i in range(5):
    tr.step(data, it0 + i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    tr.step(data, it0 + 5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s B=%d: host issue %.1f ms/step, issue+drain %.1f ms/step" % (yaml, B, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    tr.step(data, it0 + 5 + n + i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print("\n".join(l[:170] for l in s.getvalue().splitlines()[:70]))
