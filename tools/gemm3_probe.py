"""csrc/gemm3.hip against the library's tuned fp32 GEMM, shape by shape (HBM-cold operands, rotating buffer sets): the Winograd channel
products and the student's 1x1 convolutions of the BASELINE configs at 8 and at 2 images per GPU.  Guides ops._gemm3_ok.
  python tools/gemm3_probe.py [--imgs 8,2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from lgd_amd import hip, ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--imgs", default="8,2")
ap.add_argument("--reps", type=int, default=12)
args = ap.parse_args()

def _warm_clocks(seconds=1.0):
    """the first second of work on an idle GPU runs at ramping clocks: shapes measured first would read 10-20 % slow"""
    import time
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            a @ a
        torch.cuda.synchronize()


print("tuned table:", ops.enable_tuned_gemms())
_warm_clocks()
NSET = 3


def bench(fn):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.reps


def tiles(hws, n, tile=6):
    hw = [d for s in hws for d in s]
    return hip.load().lgd_wino_tiles(hip.int_array(hw), len(hws), n, tile)


def ab(tag, a, b, o, flop, accumulate=False):
    ok = ops._gemm3_ok(a[0], b[0], o[0], accumulate)
    t_lib = bench((lambda i: torch.baddbmm(o[i % NSET], a[i % NSET], b[i % NSET], out=o[i % NSET])) if accumulate else
                  (lambda i: torch.bmm(a[i % NSET], b[i % NSET], out=o[i % NSET])))
    prev = ops.gemm3_backend(True)[0]
    t_new = bench(lambda i: ops.gemm3_bmm(a[i % NSET], b[i % NSET], o[i % NSET], accumulate)) if a[0].shape[2] % 16 == 0 else float("nan")
    ops.gemm3_backend(prev)
    print("%-58s library %7.1f us %6.1f TF | gemm3 %7.1f us %6.1f TF-eq | x%.2f %s" % (tag, t_lib, flop / t_lib / 1e6, t_new, flop / t_new / 1e6, t_lib / t_new,
                                                                                  "" if ok else "(gated off)"), flush=True)


for n_img in [int(x) for x in args.imgs.split(",")]:
    pyr = synth.pyramid_shapes(800, 1344)
    print("== %d images per GPU" % n_img)
    # Winograd channel products: (name, Co, Ci, maps)
    res = {"res3": (100, 168), "res4": (50, 84), "res5": (25, 42)}
    cases = [("teacher / adapter / FPN 256->256, one pyramid", 256, 256, tiles(pyr, n_img)),
             ("head towers 256->256, both pyramids", 256, 256, tiles(pyr + pyr, n_img)),
             ("head first convs 256->512, both pyramids", 512, 256, tiles(pyr + pyr, n_img)),
             ("cls_score 256->720, both pyramids", 720, 256, tiles(pyr + pyr, n_img)),
             ("res3 conv2 128->128", 128, 128, tiles([res["res3"]], n_img)),
             ("res4 conv2 256->256", 256, 256, tiles([res["res4"]], n_img)),
             ("res5 conv2 512->512", 512, 512, tiles([res["res5"]], n_img))]
    for name, Co, Ci, T in cases:
        U = [torch.randn(64, Co, Ci, device="cuda") * 0.05 for _ in range(NSET)]
        V = [ops._freq_buf(64, Ci, T, "cuda").normal_() for _ in range(NSET)]
        M = [ops._freq_buf(64, Co, T, "cuda").normal_() for _ in range(NSET)]
        flop = 2.0 * 64 * Co * Ci * T
        ab("wino fwd %s T=%d" % (name, T), U, V, M, flop)
        ab("wino dx  %s T=%d" % (name, T), [u.transpose(1, 2) for u in U], M, V, flop)
        del U, V, M
    # 1x1 convolutions: (name, Co, Ci, (H, W))
    pw = [("res3 conv1 512->128", 128, 512, res["res3"]), ("res3 conv3 128->512", 512, 128, res["res3"]),
          ("res4 conv1 1024->256", 256, 1024, res["res4"]), ("res4 conv3 256->1024", 1024, 256, res["res4"]),
          ("res5 conv1 2048->512", 512, 2048, res["res5"]), ("res5 conv3 512->2048", 2048, 512, res["res5"]),
          ("res3 shortcut 256->512 (stride 2 in)", 512, 256, res["res3"]), ("FPN lateral 2048->256", 256, 2048, res["res5"]),
          ("FPN lateral 1024->256", 256, 1024, res["res4"]), ("FPN lateral 512->256", 256, 512, res["res3"])]
    for name, Co, Ci, (H, W) in pw:
        w = [torch.randn(Co, Ci, device="cuda") * 0.05 for _ in range(NSET)]
        x = [torch.randn(n_img, Ci, H * W, device="cuda") for _ in range(NSET)]
        y = [torch.randn(n_img, Co, H * W, device="cuda") for _ in range(NSET)]
        flop = 2.0 * n_img * Co * Ci * H * W
        ab("1x1 fwd %s" % name, [t.view(1, Co, Ci).expand(n_img, Co, Ci) for t in w], x, y, flop)
        ab("1x1 dx  %s" % name, [t.t().unsqueeze(0).expand(n_img, Ci, Co) for t in w], y, x, flop)
        ab("1x1 dx+ %s" % name, [t.t().unsqueeze(0).expand(n_img, Ci, Co) for t in w], y, x, flop, accumulate=True)
        del w, x, y
