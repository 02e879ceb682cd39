"""Winograd channel-GEMM shapes in isolation (HBM-cold operands, tuned solution table on): TFLOP/s of the forward / dx / dW
products at one pyramid (T tiles), both pyramids (2T) and the 2T problem issued as two T-column halves of the same buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import ops

tuned = ops.enable_tuned_gemms()
print("tuned table:", tuned)
nf, C = 36, 256
NSET = 3


def bench(fn, flop, reps=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, flop / us / 1e6


for T in (11440, 22880):
    for Co in (256, 512, 720):
        U = [torch.randn(nf, Co, C, device="cuda") for _ in range(NSET)]
        Ut = [u.transpose(1, 2).contiguous() for u in U]
        V = [ops._freq_buf(nf, C, T, "cuda").normal_() for _ in range(NSET)]
        M = [ops._freq_buf(nf, Co, T, "cuda").normal_() for _ in range(NSET)]
        flop = 2.0 * nf * Co * C * T
        r = {}
        r["fwd"] = bench(lambda i: torch.bmm(U[i % NSET], V[i % NSET], out=M[i % NSET]), flop)
        r["dx"] = bench(lambda i: torch.bmm(Ut[i % NSET], M[i % NSET], out=V[i % NSET]), flop)
        r["dw"] = bench(lambda i: torch.bmm(M[i % NSET], V[i % NSET].transpose(1, 2)), flop)
        h = T // 2
        if h % 4 == 0:
            def fwd_halves(i):
                torch.bmm(U[i % NSET], V[i % NSET][:, :, :h], out=M[i % NSET][:, :, :h])
                torch.bmm(U[i % NSET], V[i % NSET][:, :, h:], out=M[i % NSET][:, :, h:])
            def dx_halves(i):
                torch.bmm(Ut[i % NSET], M[i % NSET][:, :, :h], out=V[i % NSET][:, :, :h])
                torch.bmm(Ut[i % NSET], M[i % NSET][:, :, h:], out=V[i % NSET][:, :, h:])
            r["fwd_halves"] = bench(fwd_halves, flop)
            r["dx_halves"] = bench(dx_halves, flop)
        print("T=%5d Co=%3d  " % (T, Co) + "  ".join("%s %.0f us %.0f TF" % (k, v[0], v[1]) for k, v in r.items()), flush=True)
        del U, Ut, V, M
