"""When do the gradient buckets of DistributedDataParallel become ready relative to the end of backward?  Single-rank RCCL process
group, the real model, a communication hook that stamps every bucket with a HIP event before handing it to the default all-reduce:
what is still in flight when backward ends is what an N-rank run cannot hide behind compute [ref: train.py:279-281, 200-204]."""
import os
import sys

import torch
import torch.distributed as dist
from torch.distributed.algorithms.ddp_comm_hooks import default_hooks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lgd_amd import config  # noqa: E402
from lgd_amd.data import synthetic_batch  # noqa: E402
from lgd_amd.distillator import build_model  # noqa: E402
from lgd_amd.engine import Trainer  # noqa: E402

yaml = sys.argv[1] if len(sys.argv) > 1 else "lgd_retinanet_r50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29613")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
cfg = config.setup_cfg(os.path.join(ROOT, "configs", yaml + ".yaml"), ["MODEL.DEVICE", "cuda:0"])
torch.manual_seed(0)
tr = Trainer(cfg, build_model(cfg), distributed=True)
data = synthetic_batch(B, 800, 1333, 10, seed=1, device="cuda")
it0 = 40000
for i in range(4):
    tr.step(data, it0 + i)
stamps = []


def hook(state, bucket):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    stamps.append((bucket.index(), bucket.buffer().numel() * 4 / 2 ** 20, bucket.is_last(), ev))
    return default_hooks.allreduce_hook(state, bucket)


tr.model.register_comm_hook(None, hook)
rows = []
for i in range(6):
    stamps.clear()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tr.set_phase(it0 + 10 + i)
    loss = sum(tr.model(data).values())
    tr._fused_sgd.zero_grad() if tr._fused_sgd is not None else None
    e0.record()
    loss.backward()
    e1.record()
    torch.cuda.synchronize()
    rows.append((e0.elapsed_time(e1), [(idx, mb, last, e0.elapsed_time(ev)) for idx, mb, last, ev in stamps]))
bw = sum(r[0] for r in rows[1:]) / (len(rows) - 1)
print("%s B=%d: backward %.2f ms on the device; buckets (index, MiB, ready at ms after backward start, ms before its end):" % (yaml, B, bw))
for j, (idx, mb, last, _) in enumerate(rows[-1][1]):
    t = sum(r[1][j][3] for r in rows[1:]) / (len(rows) - 1)
    print("   bucket %2d  %6.1f MiB  ready at %7.2f ms  (%6.2f ms before the end of backward)%s" % (idx, mb, t, bw - t, "  <- last" if last else ""))
dist.destroy_process_group()
