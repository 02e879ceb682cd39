import time, torch, torch.nn.functional as F, sys, os
bench = sys.argv[1] == "1"
torch.backends.cudnn.benchmark = bench
dev = "cuda"
def t(fn, n=10):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); first = time.time() - t0
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return first, (time.time() - t0) / n
B = 8
shapes = [(256, 256, 3, 100, 168, 1), (256, 256, 3, 50, 84, 1), (256, 720, 3, 100, 168, 1), (256, 36, 3, 100, 168, 1), (128, 128, 3, 100, 168, 1), (512, 128, 1, 100, 168, 1), (256, 1024, 1, 50, 84, 1),
          (1024, 256, 1, 50, 84, 1), (512, 512, 3, 25, 42, 1), (2048, 256, 3, 25, 42, 2), (512, 256, 1, 100, 168, 1)]
print("benchmark =", bench)
for (ci, co, k, h, w, s) in shapes:
    x = torch.randn(B, ci, h, w, device=dev, requires_grad=True)
    wt = torch.randn(co, ci, k, k, device=dev, requires_grad=True)
    bias = torch.randn(co, device=dev, requires_grad=True)
    f = lambda: F.conv2d(x, wt, bias, padding=k // 2, stride=s)
    first, avg = t(f)
    y = f(); g = torch.randn_like(y)
    fb = lambda: torch.autograd.grad(f(), (x, wt, bias), g)
    firstb, avgb = t(fb)
    ho, wo = y.shape[-2:]
    fl = 2 * B * ci * co * k * k * ho * wo
    print("ci%4d co%4d k%d s%d %4dx%4d  fwd %.3f ms %.1f TF (first %.1fs) | fwd+bwd %.3f ms %.1f TF (first %.1fs)" % (ci, co, k, s, h, w, avg*1e3, fl/avg/1e12, first, avgb*1e3, 3*fl/avgb/1e12, firstb), flush=True)
