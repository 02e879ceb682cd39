"""csrc/gemm3.hip against the number of workgroup ROUNDS a product fills (tiles / resident slots): the channel product of one pyramid
(64 x [256 x 256] . [256 x T]) for T around BASELINE config 2's 5232 -- how much of a launch is the last, partly filled round."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import hip, ops

if "--lib" in sys.argv:   # a lab build of the kernel library (tools/gemm3_ablate.sh)
    i = sys.argv.index("--lib")
    hip._LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]


def _warm_clocks(seconds=1.0):
    """the first second of work on an idle GPU runs at ramping clocks: shapes measured first read 10-20 % slow (seen as a spurious
    'row pitch' effect in tools/gemm3_rounds.py before this was here)"""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            a @ a
        torch.cuda.synchronize()


_warm_clocks()
g = torch.Generator(device="cuda").manual_seed(0)
U = torch.randn(64, 256, 256, device="cuda", generator=g) * 0.05
NSET = 3
print("library:", hip.lib_path())
for T in (int(a) for a in (sys.argv[1:] or "2560 3840 5120 5232 5248 5376 5632 6144 10240 10464 11264".split())):
    V = [ops._freq_buf(64, 256, T, "cuda").normal_(generator=g) for _ in range(NSET)]
    M = [ops._freq_buf(64, 256, T, "cuda") for _ in range(NSET)]
    for i in range(3):
        ops.gemm3_bmm(U, V[i % NSET], M[i % NSET])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for i in range(reps):
        ops.gemm3_bmm(U, V[i % NSET], M[i % NSET])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tiles = 64 * ((T + 127) // 128)
    print("T %6d: %4d tiles = %.3f rounds of 512: %7.1f us, %.1f TF-eq, %.2f us per FULL round-equivalent" %
          (T, tiles, tiles / 512, us, 2 * 64 * 256 * 256 * T / us / 1e6, us / (tiles / 512)), flush=True)
