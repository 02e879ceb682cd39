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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as p:
    tr.step(data, 40002); torch.cuda.synchronize()
ka = p.key_averages(group_by_input_shape=True)
rows = [e for e in ka if e.self_device_time_total > 0 and "conv" not in e.key.lower() and not e.key.startswith("autograd::engine")]
rows = sorted(rows, key=lambda e: -e.self_device_time_total)[:45]
for e in rows:
    print("%8.2f ms n=%3d %-34s %s" % (e.self_device_time_total / 1e3, e.count, e.key[:34], str(e.input_shapes)[:120]))
