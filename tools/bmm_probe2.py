"""GEMM shapes of the Winograd weight-gradient / per-image batched layouts (fp32, hipBLASLt via torch)."""
import time

import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda"


def bench(f, flop, reps=8):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e3, flop / dt / 1e12


def run(tag, N, Timg, Ci, Co):
    T = N * Timg
    fl = 2.0 * 16 * Co * Ci * T
    U = torch.randn(16, Co, Ci, device=dev)
    V = torch.randn(16, Ci, T, device=dev)
    dM = torch.randn(16, Co, T, device=dev)
    print(tag, "fwd  16 x (Co,Ci)@(Ci,T)          %.3f ms %.1f TF" % bench(lambda: torch.bmm(U, V), fl))
    print(tag, "dgrad 16 x (Ci,Co)@(Co,T) U^T view %.3f ms %.1f TF" % bench(lambda: torch.bmm(U.transpose(1, 2), dM), fl))
    print(tag, "wgrad 16 x (Co,T)@(T,Ci) NT        %.3f ms %.1f TF" % bench(lambda: torch.bmm(dM, V.transpose(1, 2)), fl))
    Vn = torch.randn(N, 16, Ci, Timg, device=dev)
    dMn = torch.randn(N, 16, Co, Timg, device=dev)
    Ue = U[None].expand(N, 16, Co, Ci)
    print(tag, "fwd  N*16 x (Co,Ci)@(Ci,Timg) matmul %.3f ms %.1f TF" % bench(lambda: torch.matmul(Ue, Vn), fl))
    Uc = Ue.reshape(N * 16, Co, Ci)
    print(tag, "fwd  N*16 bmm (U pre-expanded)       %.3f ms %.1f TF" % bench(lambda: torch.bmm(Uc, Vn.view(N * 16, Ci, Timg)), fl))
    print(tag, "wgrad N*16 x (Co,Timg)@(Timg,Ci) +sum %.3f ms %.1f TF" % bench(
        lambda: torch.bmm(dMn.view(N * 16, Co, Timg), Vn.view(N * 16, Ci, Timg).transpose(1, 2)).view(N, 16, Co, Ci).sum(0), fl))
    # split T into S chunks by a strided 4-D matmul
    del Vn, dMn
    torch.cuda.empty_cache()


run("p3 256->256", 8, 50 * 84, 256, 256)
run("p4 256->256", 8, 25 * 42, 256, 256)
run("p3 256->720", 8, 50 * 84, 256, 720)
run("res3 128->128", 8, 50 * 84, 128, 128)
run("res4 256->256", 8, 25 * 42, 256, 256)
run("res5 512->512", 8, 13 * 21, 512, 512)
