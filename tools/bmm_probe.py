import torch, time
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
for (Co, Ci, tiles) in ((256, 256, 33600), (256, 256, 8400), (720, 256, 33600), (256, 256, 2100)):
    U = torch.randn(16, Co, Ci, device="cuda"); V = torch.randn(16, Ci, tiles, device="cuda")
    dt = t(lambda: torch.bmm(U, V))
    fl = 2 * 16 * Co * Ci * tiles
    print("bmm 16x[%dx%d]x[%dx%d]: %.3f ms %.1f TF" % (Co, Ci, Ci, tiles, dt * 1e3, fl / dt / 1e12))
    Vt = torch.randn(16, tiles, Ci, device="cuda")
    dt = t(lambda: torch.bmm(Vt, U.transpose(1, 2)))
    print("bmm 16x[%dx%d]x[%dx%d] (tiles-major): %.3f ms %.1f TF" % (tiles, Ci, Ci, Co, dt * 1e3, fl / dt / 1e12))
    # single big GEMM for reference
    A = torch.randn(Co, Ci, device="cuda"); Bm = torch.randn(Ci, 16 * tiles, device="cuda")
    dt = t(lambda: A @ Bm)
    print("   mm [%dx%d]x[%dx%d]: %.3f ms %.1f TF" % (Co, Ci, Ci, 16 * tiles, dt * 1e3, fl / dt / 1e12))
