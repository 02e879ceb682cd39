"""csrc/h2.hip's forward product (lgd_h2_fwd: 64 x [M x 256] . [256 x T], both operands by LDS-DMA) in isolation, HBM-cold: us, algorithmic TB/s and
the number of workgroup ROUNDS the launch fills (tiles / 512 resident slots) for T around BASELINE config 2's 5248 (one pyramid) and 10496 (both);
--lib: a lab build with parts of the kernel compiled out (tools/h2_ablate.sh: where the time of a launch goes).
    python tools/h2_rounds.py [--lib tools/lab/liblgd_h2abl_1.so] [T ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lgd_amd import hip  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    hip._LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common as cm  # noqa: E402

lib = hip.load()
dev = "cuda"
a = torch.randn(4096, 4096, device=dev)
for _ in range(60):   # (clocks up)
    a @ a
torch.cuda.synchronize()
g = torch.Generator(device=dev).manual_seed(0)
nb, M, K = 64, 256, 256
U = torch.randn((nb, M, K), device=dev, generator=g) * 0.05
sa = cm.h2_pow2_scale(U.abs().amax((1, 2)))
img, ia = cm.h2_split_image(U, sa), (1 / sa).contiguous()
NSET = 3
print("library:", hip.lib_path())
for T in (int(t) for t in (sys.argv[1:] or "2560 3840 5120 5248 5376 6144 10496".split())):
    Vs = []
    for _ in range(NSET):
        v = torch.randn((K, nb, T), device=dev, generator=g)
        sv = cm.h2_pow2_scale(v.abs().amax((0, 2)))
        Vs.append((cm.h2_split_rows(v, sv), (1 / sv).contiguous()))
        del v
    Cs = [torch.empty((M, nb, T), device=dev) for _ in range(NSET)]

    def run(i):
        vs, iv = Vs[i % NSET]
        hip.check(lib.lgd_h2_fwd(hip.ptr(img), hip.ptr(vs), 4 * T, 4 * nb * T, 4 * vs.numel(), hip.ptr(Cs[i % NSET]), T, nb * T, hip.ptr(ia), hip.ptr(iv), 1, None,
                                 nb, M, T, K, hip.stream_ptr()), "lgd_h2_fwd")
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for i in range(reps):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tiles = nb * ((T + 127) // 128)
    print("T %6d: %4d tiles = %.3f rounds of 512: %7.1f us  %.2f TB/s algorithmic  %.1f us per full-round equivalent" % (
        T, tiles, tiles / 512, us, 4.0 * nb * T * (K + M) / us / 1e6, us / (tiles / 512)), flush=True)
    del Vs, Cs
