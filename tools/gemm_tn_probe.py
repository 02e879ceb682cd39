"""dV[f] = U[f]^T dM[f] of the Winograd backward: pre-transposed operand (NN, what ships: the filter transform writes U AND U^T) against
the transposed view of U (TN: the filter transform would write half the bytes).  TunableOp picks the best solution for both."""
import os
os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tn_probe.csv")
import torch
nf = 64
for Ct, Ci, T in ((256, 256, 21456), (256, 256, 42896), (720, 256, 42896), (64, 64, 15232), (128, 128, 3808), (256, 256, 1008), (512, 512, 288)):
    U = torch.randn(nf, Ct, Ci, device="cuda")
    Ut = U.transpose(1, 2).contiguous()
    buf = torch.randn(Ct, nf, T, device="cuda")
    dM = buf.permute(1, 0, 2)                       # [nf][Ct][T] view of the [C][nf][T] frequency buffer
    out = torch.empty(Ci, nf, T, device="cuda").permute(1, 0, 2)
    res = []
    for name, a in (("NN (U^T stored)", Ut), ("TN (view of U)", U.transpose(1, 2))):
        for _ in range(3):
            torch.bmm(a, dM, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            torch.bmm(a, dM, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append("%s %.3f ms %.0f TF" % (name, ms, 2 * nf * Ct * Ci * T / ms / 1e9))
    print("Ct %4d Ci %4d T %6d: %s" % (Ct, Ci, T, " | ".join(res)))
