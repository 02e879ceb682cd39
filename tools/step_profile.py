import sys, os, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import profile, ProfilerActivity
from lgd_amd import config
from lgd_amd.data import synthetic_batch
from lgd_amd.distillator import build_model
from lgd_amd.engine import Trainer
cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda"])
model = build_model(cfg); tr = Trainer(cfg, model)
data = synthetic_batch(8, 800, 1333, 10, seed=1)
for i in range(2): tr.step(data, 40000 + i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as p:
    tr.step(data, 40002); torch.cuda.synchronize()
ev = [e for e in p.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = sum(e.device_time for e in ev)
groups = collections.defaultdict(lambda: [0.0, 0])
def cls(n):
    n = n.lower()
    if "conv" in n or "igemm" in n or "cijk" in n or "xdlops" in n or "sp3asm" in n or "winograd" in n or "im2" in n or "transpose" in n or "gemm" in n and "lgd" not in n: return "conv/gemm (MIOpen/rocBLAS/CK)"
    if "lgd::" in n: return "lgd HIP kernels"
    return n[:70]
for e in ev:
    g = groups[cls(e.name)]; g[0] += e.device_time; g[1] += 1
print("total device time %.1f ms" % (tot / 1e3))
for k, (t, c) in sorted(groups.items(), key=lambda kv: -kv[1][0])[:32]:
    print("%8.2f ms %5.1f%%  n=%4d  %s" % (t / 1e3, 100 * t / tot, c, k))
