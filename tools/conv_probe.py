import time, torch, torch.nn.functional as F, sys
torch.backends.cudnn.benchmark = False
dev = "cuda"
def t(fn, n=5):
    torch.cuda.synchronize(); t0 = time.time(); fn(); torch.cuda.synchronize(); first = time.time() - t0
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return first, (time.time() - t0) / n
B = 8
shapes = [(256, 256, 3, 100, 168), (256, 256, 3, 50, 84), (256, 256, 3, 25, 42), (256,256,3,13,21), (256,256,3,7,11), (256, 720, 3, 100, 168),
          (64, 64, 3, 200, 336), (256, 64, 1, 200, 336), (128,128,3,100,168), (512, 2048, 1, 25, 42), (3, 64, 7, 800, 1344)]
for fmt in ("nchw", "nhwc"):
  for dt in (torch.float32, torch.bfloat16):
    for (ci, co, k, h, w) in shapes:
        x = torch.randn(B, ci, h, w, device=dev, dtype=dt, requires_grad=True)
        wt = torch.randn(co, ci, k, k, device=dev, dtype=dt, requires_grad=True)
        if fmt == "nhwc":
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wt = wt.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        stride = 2 if k == 7 else 1
        f = lambda: F.conv2d(x, wt, padding=k // 2, stride=stride)
        first, avg = t(f)
        y = f(); g = torch.randn_like(y)
        fb = lambda: torch.autograd.grad(f(), (x, wt), g)
        firstb, avgb = t(fb)
        ho, wo = y.shape[-2:]
        fl = 2 * B * ci * co * k * k * ho * wo
        print("%s %s ci%4d co%4d k%d %4dx%4d  fwd first %.2fs avg %.3f ms %.1f TF | fwd+bwd first %.2fs avg %.3f ms %.1f TF" % (fmt, str(dt)[6:], ci, co, k, h, w, first, avg*1e3, fl/avg/1e12, firstb, avgb*1e3, 3*fl/avgb/1e12), flush=True)
