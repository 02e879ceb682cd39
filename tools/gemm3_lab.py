"""LAB driver for tools/lab/gemm3_lab.hip (VERDICT r3 #1 stage a): the bf16x3 split-operand batched GEMM against the library's tuned
fp32 GEMM (torch.bmm + the TunableOp table) on the Winograd channel-product shapes of BASELINE config 2 -- time (HBM-cold operands,
rotating buffer sets) and error against an fp64 product of the same fp32 operands.

  python tools/gemm3_lab.py [--build] [--shapes fwd,dx,dw] [--T 5248] [--reps 20]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tools", "lab", "gemm3_lab.hip")
LIB = os.path.join(ROOT, "tools", "lab", "libgemm3_lab.so")


def build(defs=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB] + list(defs)
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def load(path=None):
    lib = ctypes.CDLL(path or LIB)
    L, P = ctypes.c_long, ctypes.c_void_p
    lib.gemm3_lab.argtypes = [P, L, L, L, P, L, L, L, P, L, L, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, L, P]
    lib.gemm3_lab.restype = ctypes.c_int
    lib.gemm3_image_bytes.argtypes = [ctypes.c_int] * 3
    lib.gemm3_image_bytes.restype = L
    lib.gemm3_split_a.argtypes = [P, L, L, L, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P]
    lib.gemm3_split_a.restype = ctypes.c_int
    lib.gemm3s_lab.argtypes = [P, P, L, L, P, L, L, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    lib.gemm3s_lab.restype = ctypes.c_int
    return lib


def split_a(lib, a, img=None):
    """the bf16x3 fragment image of a (nb, M, K) fp32 operand with arbitrary strides (a filter: done once per step)"""
    import torch
    nb, M, K = a.shape
    if img is None:
        img = torch.empty(lib.gemm3_image_bytes(nb, M, K), dtype=torch.uint8, device=a.device)
    rc = lib.gemm3_split_a(a.data_ptr(), a.stride(0), a.stride(1), a.stride(2), nb, M, K, img.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return img


def gemm3s(lib, img, a_shape, b, out):
    """out[b] = A[b] @ b[b] with A given as its pre-split image; b (nb, K, N) and out (nb, M, N) with the last axis contiguous"""
    import torch
    nb, M, K = a_shape
    N = b.shape[2]
    assert b.stride(2) == 1 and out.stride(2) == 1
    rc = lib.gemm3s_lab(img.data_ptr(), b.data_ptr(), b.stride(0), b.stride(1), out.data_ptr(), out.stride(0), out.stride(1), nb, M, N, K,
                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    return out


def gemm3(lib, a, b, out, ksplit=1, ws=None):
    """out[b] = a[b] @ b[b] for 3-D fp32 tensors with arbitrary (supported) strides; ksplit > 1: partials into ws (ksplit, *out.shape), summed."""
    import torch
    nb, M, K = a.shape
    N = b.shape[2]
    assert out.stride(2) == 1
    tgt = out if ksplit == 1 else ws
    rc = lib.gemm3_lab(a.data_ptr(), a.stride(0), a.stride(1), a.stride(2), b.data_ptr(), b.stride(0), b.stride(1), b.stride(2),
                       tgt.data_ptr(), out.stride(0) if ksplit == 1 else ws.stride(1), out.stride(1) if ksplit == 1 else ws.stride(2),
                       nb, M, N, K, ksplit, 0 if ksplit == 1 else ws.stride(0), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    if ksplit > 1:
        torch.sum(ws, 0, out=out)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--shapes", default="fwd,dx,dw")
    ap.add_argument("--T", type=int, default=0, help="tiles (0: BASELINE config 2's pyramid, 8 images of 800x1344, F(6x6))")
    ap.add_argument("--co", type=int, default=256)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ksplit", type=int, default=4)
    ap.add_argument("--lib", default=None, help="another build of the lab library (ablation variants)")
    ap.add_argument("--v3-only", action="store_true", help="time only the pre-split-A kernel (no error checks: ablation builds compute garbage)")
    args = ap.parse_args()
    if args.build or not os.path.exists(LIB):
        build()
        if args.build and len(sys.argv) == 2:
            return
    import torch
    from lgd_amd import hip, ops, synth
    lib = load(args.lib)
    print("tuned GEMM table:", ops.enable_tuned_gemms())
    nf, C, Co = 64, 256, args.co
    T = args.T
    if not T:
        hw = [d for s in synth.pyramid_shapes(800, 1344) for d in s]
        T = hip.load().lgd_wino_tiles(hip.int_array(hw), 5, 8, 6)
    print("shape: nf %d, C %d, Co %d, T %d" % (nf, C, Co, T))
    NSET = 3
    g = torch.Generator(device="cuda").manual_seed(0)
    U = [torch.randn(nf, Co, C, device="cuda", generator=g) * 0.05 for _ in range(NSET)]
    V = [ops._freq_buf(nf, C, T, "cuda").normal_(generator=g) for _ in range(NSET)]
    M = [ops._freq_buf(nf, Co, T, "cuda").normal_(generator=g) for _ in range(NSET)]

    def bench(fn, flop):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        return us, flop / us / 1e6

    def err(c, a, b):
        ref = torch.bmm(a[:4].double(), b[:4].double())
        d = (c[:4].double() - ref)
        return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())

    flop = 2.0 * nf * Co * C * T
    for shape in args.shapes.split(","):
        if shape == "fwd":      # M[f] = U[f] . V[f]
            a, b, o = U, V, [ops._freq_buf(nf, Co, T, "cuda") for _ in range(NSET)]
            ks = 1
        elif shape == "dx":     # dV[f] = U[f]^T . dM[f]
            a, b, o = [u.transpose(1, 2) for u in U], M, [ops._freq_buf(nf, C, T, "cuda") for _ in range(NSET)]
            ks = 1
        else:                   # dU[f] = dM[f] . V[f]^T
            a, b, o = M, [v.transpose(1, 2) for v in V], [torch.empty(nf, Co, C, device="cuda") for _ in range(NSET)]
            ks = args.ksplit
        if args.v3_only:
            imgs = [split_a(lib, x) for x in a]
            t3 = bench(lambda i: gemm3s(lib, imgs[i % NSET], a[0].shape, b[i % NSET], o[i % NSET]), flop)
            print("%-3s v3 %s: %.0f us" % (shape, os.path.basename(args.lib or LIB), t3[0]), flush=True)
            continue
        ws = torch.empty((ks,) + tuple(o[0].shape), device="cuda") if ks > 1 else None
        c_lib = torch.bmm(a[0], b[0])
        c_new = gemm3(lib, a[0], b[0], o[0], ks, ws).clone()
        e_lib, e_new = err(c_lib, a[0], b[0]), err(c_new, a[0], b[0])
        d = float((c_new - c_lib).abs().max() / c_lib.abs().max())
        t_lib = bench(lambda i: torch.bmm(a[i % NSET], b[i % NSET], out=o[i % NSET]) if o[0].stride(2) == 1 and shape != "dw"
                      else torch.bmm(a[i % NSET], b[i % NSET]), flop)
        t_new = bench(lambda i: gemm3(lib, a[i % NSET], b[i % NSET], o[i % NSET], ks, ws), flop)
        if shape in ("fwd", "dx"):   # v3: A pre-split (a filter), B split in the kernel
            imgs = [split_a(lib, x) for x in a]
            c3 = gemm3s(lib, imgs[0], a[0].shape, b[0], o[0]).clone()
            e3 = err(c3, a[0], b[0])
            t3 = bench(lambda i: gemm3s(lib, imgs[i % NSET], a[0].shape, b[i % NSET], o[i % NSET]), flop)
            ts = bench(lambda i: split_a(lib, a[i % NSET], imgs[i % NSET]), flop)
            print("%-3s v3 (A pre-split image + LDS-DMA, 256x128x16): %.0f us %.1f TF-equivalent (err max %.2e rms %.2e; vs library %.2e) | speed-up %.2fx | "
                  "split of A: %.1f us per call" % (shape, t3[0], t3[1], e3[0], e3[1], float((c3 - c_lib).abs().max() / c_lib.abs().max()),
                                                    t_lib[0] / t3[0], ts[0]), flush=True)
        print("%-3s library %.0f us %.1f TF (err max %.2e rms %.2e) | bf16x3 %.0f us %.1f TF-equivalent (err max %.2e rms %.2e; vs library %.2e) "
              "| speed-up %.2fx%s" % (shape, t_lib[0], t_lib[1], e_lib[0], e_lib[1], t_new[0], t_new[1], e_new[0], e_new[1], d,
                                      t_lib[0] / t_new[0], " (ksplit %d incl. the sum)" % ks if ks > 1 else ""), flush=True)


if __name__ == "__main__":
    main()
