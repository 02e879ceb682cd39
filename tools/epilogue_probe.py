"""conv3 of the bottleneck blocks at BASELINE config 2 (8 images of 800x1344): the product + bias_act pass against the product with the
epilogue inside csrc/gemm3.hip (tools/gpu_checks.sh; numbers in DESIGN.md K9).  HBM-cold: rotating operand sets."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import hip, ops


def _warm_clocks(seconds=1.0):
    """the first second of work on an idle GPU runs at ramping clocks: shapes measured first would read 10-20 % slow"""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            a @ a
        torch.cuda.synchronize()


lib = hip.load()
_warm_clocks()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SH = [("res2 conv3 64->256", 64, 256, 200 * 336), ("res3 conv3 128->512", 128, 512, 100 * 168), ("res4 conv3 256->1024", 256, 1024, 50 * 84),
      ("res5 conv3 512->2048", 512, 2048, 25 * 42), ("res3 conv1 512->128", 512, 128, 100 * 168), ("res4 conv1 1024->256", 1024, 256, 50 * 84)]
NSET = 3
for name, Ci, Co, HW in SH:
    g = torch.Generator(device="cuda").manual_seed(0)
    w = torch.randn(Co, Ci, device="cuda", generator=g) * 0.05
    xs = [torch.randn(N, Ci, HW, device="cuda", generator=g) for _ in range(NSET)]
    rs = [torch.randn(N, Co, HW, device="cuda", generator=g) for _ in range(NSET)]
    shift = torch.randn(Co, device="cuda", generator=g)
    a = w.view(1, Co, Ci).expand(N, Co, Ci)
    ys = [torch.empty(N, Co, HW, device="cuda") for _ in range(NSET)]
    os_ = [torch.empty(N, Co, HW, device="cuda") for _ in range(NSET)]
    bits = torch.empty(int(lib.lgd_relu_bits_words(N * Co * HW)), dtype=torch.int32, device="cuda")
    rbits = torch.empty(int(lib.lgd_relu_rowbits_words(N * Co, HW)), dtype=torch.int32, device="cuda")

    def unfused(i):
        y = ops.gemm3_bmm(a, xs[i % NSET], ys[i % NSET])
        hip.check(lib.lgd_bias_act_fwd(hip.ptr(y), hip.ptr(shift), hip.ptr(rs[i % NSET]), N, Co, HW, 1, hip.ptr(os_[i % NSET]), hip.ptr(bits), hip.stream_ptr()), "b")

    def prod(i):
        ops.gemm3_bmm(a, xs[i % NSET], ys[i % NSET])

    def fused(i):
        ops.gemm3_bmm(a, xs[i % NSET], os_[i % NSET], residual=rs[i % NSET], shift=shift, relu=True, relu_bits=rbits)

    def fused_r(i):
        ops.gemm3_bmm(a, xs[i % NSET], os_[i % NSET], residual=rs[i % NSET])

    def fused_s(i):
        ops.gemm3_bmm(a, xs[i % NSET], os_[i % NSET], shift=shift, relu=True, relu_bits=rbits)

    def fused_s0(i):
        ops.gemm3_bmm(a, xs[i % NSET], os_[i % NSET], shift=shift)

    def fused_s1(i):
        ops.gemm3_bmm(a, xs[i % NSET], os_[i % NSET], shift=shift, relu=True)

    def lib_(i):
        y = torch.bmm(a, xs[i % NSET], out=ys[i % NSET])
        hip.check(lib.lgd_bias_act_fwd(hip.ptr(y), hip.ptr(shift), hip.ptr(rs[i % NSET]), N, Co, HW, 1, hip.ptr(os_[i % NSET]), hip.ptr(bits), hip.stream_ptr()), "b")

    def t(fn, reps=20):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    tl, tp, tu, tf, tr, ts = t(lib_), t(prod), t(unfused), t(fused), t(fused_r), t(fused_s)
    mb = 4 * N * HW * (Ci + 2 * Co) / 1e6
    print("%-24s library+bias_act %7.1f us | gemm3 %7.1f + bias_act = %7.1f us | fused %7.1f us (%.2f TB/s of %d MB) | x%.2f | R only %.1f, shift + ReLU + mask only %.1f us" %
          (name, tl, tp, tu, tf, mb / tf, mb, tu / tf, tr, ts), flush=True)
    print("   shift only %.1f, shift + ReLU %.1f us" % (t(fused_s0), t(fused_s1)), flush=True)
