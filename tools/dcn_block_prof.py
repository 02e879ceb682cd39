import torch, sys
sys.path.insert(0, '.')
from lgd_amd.student.resnet import DeformBottleneck
from lgd_amd import ops
ops.enable_tuned_gemms()
torch.manual_seed(0)
blk = DeformBottleneck(1024, 1024, 256, 1).cuda()
for p in blk.parameters(): p.requires_grad_(True)
torch.nn.init.normal_(blk.conv2_offset.weight, std=0.01)
x = torch.randn(2, 1024, 50, 84, device='cuda', requires_grad=True)
for _ in range(3):
    y = blk(x); y.sum().backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        y = blk(x); y.sum().backward()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
agg = {}
for e in evs:
    a = agg.setdefault(e.name[:110], [0, 0.0]); a[0] += 1; a[1] += e.device_time if hasattr(e, 'device_time') else e.cuda_time
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%5.1f calls %8.1f us/call  %s" % (n / 5, t / n, k))
