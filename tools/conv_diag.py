import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
def prof(tag, fn):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        fn(); torch.cuda.synchronize()
    names = sorted(((e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.key) for e in p.key_averages()), reverse=True)[:4]
    print(tag, [(round(t), k[:60]) for t, k in names], flush=True)
x = torch.randn(8, 256, 100, 168, device="cuda")
w1 = torch.randn(128, 256, 1, 1, device="cuda", requires_grad=True)
w3 = torch.randn(128, 128, 3, 3, device="cuda", requires_grad=True)
s = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda")
def bwd(y): y.sum().backward()
prof("1x1 plain        ", lambda: bwd(F.conv2d(x, w1)))
prof("1x1 folded*scale ", lambda: bwd(F.conv2d(x, w1 * s.view(-1, 1, 1, 1), sh)))
prof("1x1 folded view  ", lambda: bwd(F.conv2d(x, (w1.view(128, -1) * s[:, None]).view_as(w1), sh)))
prof("1x1 stride2      ", lambda: bwd(F.conv2d(x, w1, stride=2)))
h = F.conv2d(x, w1).detach()
prof("3x3 plain        ", lambda: bwd(F.conv2d(h, w3, padding=1)))
prof("3x3 folded       ", lambda: bwd(F.conv2d(h, w3 * s.view(-1, 1, 1, 1), sh, padding=1)))
h2 = F.relu_(F.conv2d(x, w1 * s.view(-1, 1, 1, 1), sh)).detach()
print("h2 strides", h2.stride(), h2.is_contiguous())
prof("3x3 on folded out", lambda: bwd(F.conv2d(h2, w3, padding=1)))
from lgd_amd.student.resnet import Bottleneck
blk = Bottleneck(256, 512, 128, 2).cuda()
xx = x.clone().requires_grad_(True)
prof("Bottleneck s2    ", lambda: bwd(blk(xx)))
