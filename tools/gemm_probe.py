import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import ops
def run(name, probs, n=20):
    for _ in range(3): ops._gemm_batch(probs())
    torch.cuda.synchronize()
    ops.kernel_timer_enable(True)
    for _ in range(n): ops._gemm_batch(probs())
    torch.cuda.synchronize()
    t = ops.kernel_timer_collect(); ops.kernel_timer_enable(False)
    c, ms = t["gemm_batch_kernel"]
    print("%-40s %.1f us" % (name, 1e3 * ms / c))
E = 256
for M in (16, 88, 440):
    A = torch.randn(M, E, device="cuda"); W = torch.randn(E, E, device="cuda"); C = torch.empty(M, E, device="cuda")
    run("NT M=%d N=256 K=256" % M, lambda: [ops._gemm((A, 0), (E, 1), (W, 0), (E, 1), (C, 0), (E, 1), M, E, E)])
    run("NN M=%d N=256 K=256" % M, lambda: [ops._gemm((A, 0), (E, 1), (W, 0), (1, E), (C, 0), (E, 1), M, E, E)])
    D = torch.empty(E, E, device="cuda")
    run("TN M=256 N=256 K=%d" % M, lambda: [ops._gemm((A, 0), (1, E), (A, 0), (1, E), (D, 0), (E, 1), E, E, M)])
x = torch.randn(440, 256, device="cuda"); w = torch.randn(256, 256, device="cuda")
import time
for _ in range(5): x @ w.t()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): x @ w.t()
e1.record(); torch.cuda.synchronize(); print("torch matmul 440x256x256: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
