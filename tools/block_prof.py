"""kernel list of ONE student block, forward + backward (torch.profiler on the GPU box): python tools/block_prof.py [dcn|plain] [C mid H W N]"""
import sys
import torch
sys.path.insert(0, '.')
from lgd_amd import ops
from lgd_amd.student.resnet import Bottleneck, DeformBottleneck
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
C, mid, H, W, N = (int(v) for v in (sys.argv[2:7] if len(sys.argv) > 6 else (1024, 256, 50, 84, 2)))
ops.enable_tuned_gemms()
torch.manual_seed(0)
blk = (DeformBottleneck if kind == "dcn" else Bottleneck)(C, C, mid, 1).cuda()
import os
sigma = float(os.environ.get("LGD_DCN_OFFSET_SIGMA", "0"))   # > 0: LEARNED offsets -- the zero-initialised offset convolution's weights drawn from N(0, sigma^2): fractional sampling positions
if kind == "dcn" and sigma > 0:
    torch.nn.init.normal_(blk.conv2_offset.weight, std=sigma)
for p in blk.parameters():
    p.requires_grad_(True)
x = torch.randn(N, C, H, W, device='cuda', requires_grad=True)
for _ in range(3):
    blk(x).sum().backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        blk(x).sum().backward()
    torch.cuda.synchronize()
agg = {}
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        a = agg.setdefault(e.name[:110], [0, 0.0])
        a[0] += 1
        a[1] += e.device_time
tot = sum(t for _, t in agg.values()) / 5
print("%s block C=%d mid=%d %dx%d N=%d%s: %.1f us of kernels per fwd+bwd, %d launches" % (kind, C, mid, H, W, N, " offset-conv weights N(0, %g^2)" % sigma if sigma > 0 else "", tot, sum(n for n, _ in agg.values()) / 5))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%5.1f calls %8.1f us/call  %s" % (n / 5, t / n, k))
