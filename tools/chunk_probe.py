"""LAB (round 6): one 3x3 convolution of the path (256 -> 256 over the config-2 pyramid, the shipped F(6x6,3x3) + h2 pipeline:
wino6_in -> h2_fwd -> wino6_out; backward wino6_out_t -> h2_fwd (dx) + h2_dw -> wino6_in_t) run over the WHOLE mini-batch at once (shipped) against
the same convolution issued over image CHUNKS one after the other, so that a chunk's V and M (2 x 43 MB at 1 image ... 2 x 171 MB at 4) may still be
in the 256 MB memory-side cache when the next kernel of the chain reads them.  Round 2 measured this with library GEMMs (MFMA-bound, N = T/2..T/8
lost more than the transforms gained); the products are HBM-bound now.  Inputs rotate over > 1 GB (HBM-cold), forward and forward + backward.
    python tools/chunk_probe.py [--B 8] [--both-pyramids]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lgd_amd import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--both-pyramids", action="store_true", help="the head's shape: 2 x 5 maps per call")
a = ap.parse_args()
B, C = a.B, 256
hws = synth.pyramid_shapes(800, 1344) * (2 if a.both_pyramids else 1)
nset = 4
sets = [[torch.randn(B, C, h, w, device="cuda") for h, w in hws] for _ in range(nset)]
grads = [[torch.randn(B, C, h, w, device="cuda") for h, w in hws] for _ in range(nset)]
w = (torch.randn(C, C, 3, 3, device="cuda") * 0.02).requires_grad_(True)
b = torch.zeros(C, device="cuda", requires_grad=True)


def conv(xs, k):
    """the convolution over chunks of k images (k = B: one call)"""
    outs = [[] for _ in xs]
    for i in range(0, B, k):
        ys = ops._Conv3x3.apply(w, b, True, ops._WINO_TILE, *[x[i:i + k] for x in xs])
        for o, y in zip(outs, ys):
            o.append(y)
    return outs


def timeit(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.reps):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / a.reps


for k in [B] + [c for c in (4, 2, 1) if c < B]:
    def fwd(i):
        with torch.no_grad():
            conv(sets[i % nset], k)

    def fwd_bwd(i):
        xs = [x.detach().requires_grad_(True) for x in sets[i % nset]]
        outs = conv(xs, k)
        ys = [y for o in outs for y in o]
        gs = [g[j:j + k] for g, o in zip(grads[i % nset], outs) for j in range(0, B, k)]
        torch.autograd.backward(ys, gs)
        w.grad = b.grad = None
    tiles = sum(k * ((h + 5) // 6) * ((ww + 5) // 6) for h, ww in hws)
    print("chunks of %d image(s) (%5d tiles, V = %5.1f MB per chunk): forward %.3f ms   forward + backward %.3f ms" % (
        k, tiles, tiles * 64 * C * 4 / 1e6, timeit(fwd), timeit(fwd_bwd)), flush=True)
# where the time goes, whole batch vs chunks of 2: the library's own per-kernel event timers
for k in (B, 2):
    ops.kernel_timer_enable(True)
    for i in range(6):
        with torch.no_grad():
            conv(sets[i % nset], k)
    torch.cuda.synchronize()
    t = ops.kernel_timer_collect()
    ops.kernel_timer_enable(False)
    print("forward, chunks of %d: " % k + "  ".join("%s %d x %.1f us" % (n.replace("_kernel", ""), v[0] // 6, 1e3 * v[1] / v[0]) for n, v in sorted(t.items()) if v[1] / 6 > 0.02))
