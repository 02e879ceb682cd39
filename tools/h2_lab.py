"""LAB driver for tools/lab/h2_lab.hip (round 5): the Winograd channel products from f16x2 split operands that are
already split in HBM -- forward / input-gradient product (filter image x split rows, transposing LDS reads) and the weight-gradient
product (split rows x split rows, split-K) -- against an fp64 product of the same fp32 operands (error) and against the library's
fp32 GEMM and csrc/gemm3.hip (time, HBM-cold: rotating operand sets).

  python tools/h2_lab.py [--build] [--T 5248] [--reps 10] [--what probe,fwd,dw]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tools", "lab", "h2_lab.hip")
LIB = os.path.join(ROOT, "tools", "lab", "libh2_lab.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", LIB]
    print(" ".join(cmd))
    subprocess.check_call(cmd)


def load():
    lib = ctypes.CDLL(LIB)
    L, P, I = ctypes.c_long, ctypes.c_void_p, ctypes.c_int
    lib.h2_image_bytes.argtypes = [I, I, I]; lib.h2_image_bytes.restype = L
    lib.h2_split_rows.argtypes = [P, P, P, L, I, I, P]
    lib.h2_split_image.argtypes = [P, L, L, L, P, I, I, I, P, P]
    lib.h2_fwd.argtypes = [P, P, L, L, L, P, L, L, P, P, P, I, I, I, I, P]
    lib.h2_dw.argtypes = [P, L, L, L, P, L, L, L, P, P, P, P, I, I, I, I, I, I, P]
    lib.h2_tr_probe.argtypes = [P, P]
    lib.h2_fwd2.argtypes = [P, P, L, L, L, P, L, L, P, P, I, I, I, I, I, P]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--T", type=int, default=5248)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--what", default="probe,fwd,dw")
    ap.add_argument("--M", type=int, default=256)
    ap.add_argument("--variants", default="0,8,9")
    args = ap.parse_args()
    if args.build or not os.path.exists(LIB):
        build()
    import torch
    lib = load()
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    what = args.what.split(",")
    nf, C, T, M = 64, 256, args.T, args.M

    if "probe" in what:
        out = torch.zeros(256, device=dev)
        lib.h2_tr_probe(out.data_ptr(), st())
        torch.cuda.synchronize()
        o = out.view(64, 4).cpu().int().tolist()
        print("ds_read_b64_tr_b16 probe: lane -> 4 values (LDS element indices; lane l read at element 4 l)")
        for l in range(0, 64, 1):
            print("  lane %2d: %s" % (l, o[l]))

    def timeit(fn, sets, reps):
        for i in range(2):
            fn(i % sets)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(reps):
            fn(i % sets)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / reps

    g = torch.Generator(device=dev); g.manual_seed(1)
    # per-frequency magnitudes over 6 decades and per-channel magnitudes over 2: what the transforms produce
    fmag = torch.exp2(torch.linspace(-10, 10, nf, device=dev))[torch.randperm(nf, device=dev, generator=g)]
    cmag = torch.exp2(torch.linspace(-3, 3, C, device=dev))[torch.randperm(C, device=dev, generator=g)]

    def pow2_scale(x_amax):   # multiplier 2^e with |x| 2^e < 2^15
        e = 14 - torch.floor(torch.log2(x_amax.clamp_min(1e-30)))
        return torch.exp2(e)

    def split_rows(x):   # x (rows, nf, T) fp32 contiguous -> (bytes tensor, scale multiplier (nf), inverse)
        sc = pow2_scale(x.abs().amax(dim=(0, 2)))
        out = torch.empty(x.numel() * 4, dtype=torch.uint8, device=dev)
        lib.h2_split_rows(x.data_ptr(), sc.data_ptr(), out.data_ptr(), x.shape[0], nf, x.shape[2], st())
        return out, sc, (1.0 / sc).contiguous()

    def split_image(a):  # a (nf, M, K)
        sc = pow2_scale(a.abs().amax(dim=(1, 2)))
        img = torch.empty(lib.h2_image_bytes(a.shape[0], a.shape[1], a.shape[2]), dtype=torch.uint8, device=dev)
        lib.h2_split_image(a.data_ptr(), a.stride(0), a.stride(1), a.stride(2), sc.data_ptr(), a.shape[0], a.shape[1], a.shape[2], img.data_ptr(), st())
        return img, (1.0 / sc).contiguous()

    def err(c, ref):
        e = (c.double() - ref).abs().amax(dim=(1, 2)) / ref.abs().amax(dim=(1, 2))
        return e.max().item(), e.mean().item()

    if "fwd" in what:
        print("== forward product  M[f] = U[f] V[f]:  %d x [%d x %d].[%d x %d]" % (nf, M, C, C, T))
        U = torch.randn((nf, M, C), device=dev, generator=g) * fmag.view(-1, 1, 1) * 0.05
        V = torch.randn((C, nf, T), device=dev, generator=g) * fmag.flip(0).view(1, -1, 1) * cmag.view(-1, 1, 1)
        img, a_inv = split_image(U)
        Vs, _, b_inv = split_rows(V)
        Cout = torch.empty((M, nf, T), device=dev)
        rc = lib.h2_fwd(img.data_ptr(), Vs.data_ptr(), T * 4, nf * T * 4, Vs.numel(), Cout.data_ptr(), T, nf * T, a_inv.data_ptr(), b_inv.data_ptr(),
                        None, nf, M, T, C, st())
        torch.cuda.synchronize()
        assert rc == 0, rc
        Vp = V.permute(1, 0, 2)
        ref = torch.bmm(U.double(), Vp.double())
        c32 = torch.bmm(U, Vp)
        print("   error vs fp64 (max over batches of max|d| / max|ref|, mean):  h2 %.3e %.3e   library fp32 %.3e %.3e" %
              (*err(Cout.permute(1, 0, 2), ref), *err(c32, ref)))
        amax = torch.zeros(nf, dtype=torch.int32, device=dev)
        lib.h2_fwd(img.data_ptr(), Vs.data_ptr(), T * 4, nf * T * 4, Vs.numel(), Cout.data_ptr(), T, nf * T, a_inv.data_ptr(), b_inv.data_ptr(),
                   amax.data_ptr(), nf, M, T, C, st())
        torch.cuda.synchronize()
        am = amax.view(torch.float32)
        print("   amax epilogue: max rel dev from the true per-batch max %.2e" % ((am - ref.abs().amax(dim=(1, 2)).float()).abs() / am).max().item())
        del ref, c32
        sets = 3
        Vss = [Vs] + [Vs.clone() for _ in range(sets - 1)]
        Cs = [Cout] + [torch.empty_like(Cout) for _ in range(sets - 1)]
        t_h2 = timeit(lambda i: lib.h2_fwd(img.data_ptr(), Vss[i].data_ptr(), T * 4, nf * T * 4, Vs.numel(), Cs[i].data_ptr(), T, nf * T, a_inv.data_ptr(),
                                           b_inv.data_ptr(), None, nf, M, T, C, st()), sets, args.reps)
        Vf = [V] + [V.clone() for _ in range(sets - 1)]
        t_lib = timeit(lambda i: torch.bmm(U, Vf[i].permute(1, 0, 2), out=Cs[i].permute(1, 0, 2)), sets, args.reps)
        flop = 2.0 * nf * M * C * T
        byt = 4.0 * nf * T * (C + M)
        print("   h2_fwd %.1f us (%.0f TFLOP/s fp32-eq, %.2f TB/s algorithmic)   library fp32 bmm %.1f us" % (t_h2, flop / t_h2 * 1e-6, byt / t_h2 * 1e-6, t_lib))
        names = {0: "256x128 3 buffers (shipped)", 1: "256x256 8 waves 4 buffers", 2: "128x256 3 buffers", 3: "256x128 2 buffers", 4: "256x256 3 buffers",
                 5: "256x256 5 buffers", 6: "128x256 4 buffers", 7: "128x128 2 waves 4 buffers", 8: "256x128x3 + nt stores of C", 9: "256x128x3 + nt stores + nt DMA of B"}
        for variant in [int(v) for v in args.variants.split(",")]:
            Cs[0].fill_(float("nan"))
            rc = lib.h2_fwd2(img.data_ptr(), Vs.data_ptr(), T * 4, nf * T * 4, Vs.numel(), Cs[0].data_ptr(), T, nf * T, a_inv.data_ptr(), b_inv.data_ptr(),
                             nf, M, T, C, variant, st())
            torch.cuda.synchronize()
            ref = torch.bmm(U.double(), V.permute(1, 0, 2).double())
            e = err(Cs[0].permute(1, 0, 2), ref)
            del ref
            t_v = timeit(lambda i: lib.h2_fwd2(img.data_ptr(), Vss[i].data_ptr(), T * 4, nf * T * 4, Vs.numel(), Cs[i].data_ptr(), T, nf * T, a_inv.data_ptr(),
                                               b_inv.data_ptr(), nf, M, T, C, variant, st()), sets, args.reps)
            print("   variant %d %-28s rc %d error %.2e: %.1f us (%.2f TB/s algorithmic)" % (variant, names[variant], rc, e[0], t_v, byt / t_v * 1e-6))
        # layout experiment (timing only: one filter image for every batch): the same bytes with every tile's B operand CONTIGUOUS -- [tile block][f][C][128 tiles]
        # emulated as 64 x 41 "batches" of one 128-column tile each, k-row stride 512 B, batch stride 128 KB; C likewise [batch][M][128]
        nbx = nf * (T // 128)
        Vx = torch.empty(nbx * C * 128, dtype=torch.int32, device=dev).random_(0, 2 ** 14)
        Vxs = [Vx] + [Vx.clone() for _ in range(sets - 1)]
        Cx = [torch.empty(nbx * M * 128, device=dev) for _ in range(sets)]
        ainv = torch.ones(nbx, device=dev)
        for variant in (0, 100):
            if variant == 0:
                fnx = lambda i: lib.h2_fwd2(img.data_ptr(), Vss[i].data_ptr(), T * 4, nf * T * 4, Vs.numel(), Cs[i].data_ptr(), T, nf * T, a_inv.data_ptr(), b_inv.data_ptr(), nf, M, T, C, 0, st())  # noqa: E731
            else:
                fnx = lambda i: lib.h2_fwd2(img.data_ptr(), Vxs[i].data_ptr(), C * 512, 512, Vx.numel() * 4, Cx[i].data_ptr(), M * 128, 128, ainv.data_ptr(), ainv.data_ptr(), nbx, M, 128, C, 100, st())  # noqa: E731
            print("   layout experiment %s: %.1f us" % ("strided rows [C][f][T] (shipped)" if variant == 0 else "contiguous tiles [T/128][f][C][128]", timeit(fnx, sets, args.reps)))
        del Vxs, Cx
        try:
            from lgd_amd import ops
            ops.gemm3_backend(True, force=True)
            t_g3 = timeit(lambda i: ops.gemm3_bmm(U, Vf[i].permute(1, 0, 2), out=Cs[i].permute(1, 0, 2)), sets, args.reps)
            print("   gemm3 (bf16x3, split pass of U included) %.1f us" % t_g3)
        except Exception as e:  # noqa: BLE001
            print("   gemm3 not timed:", e)
        del Vss, Cs, Vf, V, Vs, Cout

    if "dw" in what:
        print("== weight-gradient product  dU[f] = dM[f] V[f]^T:  %d x [%d x %d].[%d x %d]" % (nf, M, T, T, C))
        dM = torch.randn((M, nf, T), device=dev, generator=g) * fmag.view(1, -1, 1) * 1e-3
        V = torch.randn((C, nf, T), device=dev, generator=g) * fmag.flip(0).view(1, -1, 1) * cmag.view(-1, 1, 1)
        As, _, a_inv = split_rows(dM)
        Bs, _, b_inv = split_rows(V)
        ref = torch.bmm(dM.permute(1, 0, 2).double(), V.permute(1, 2, 0).double())
        c32 = torch.bmm(dM.permute(1, 0, 2), V.permute(1, 2, 0))
        print("   library fp32 error vs fp64: %.3e %.3e" % err(c32, ref))
        rs = nf * T * 4
        for variant in (0, 1):
            for S in (1, 2, 4, 8):
                out = torch.empty((nf, M, C), device=dev)
                part = torch.empty((S, nf, M, C), device=dev)
                rc = lib.h2_dw(As.data_ptr(), rs, T * 4, As.numel(), Bs.data_ptr(), rs, T * 4, Bs.numel(), part.data_ptr(), out.data_ptr(), a_inv.data_ptr(),
                               b_inv.data_ptr(), nf, M, C, T, S, variant, st())
                torch.cuda.synchronize()
                assert rc == 0, rc
                e = err(out, ref)
                sets = 3
                Ass = [As] + [As.clone() for _ in range(sets - 1)]
                Bss = [Bs] + [Bs.clone() for _ in range(sets - 1)]
                t = timeit(lambda i: lib.h2_dw(Ass[i].data_ptr(), rs, T * 4, As.numel(), Bss[i].data_ptr(), rs, T * 4, Bs.numel(), part.data_ptr(), out.data_ptr(),
                                               a_inv.data_ptr(), b_inv.data_ptr(), nf, M, C, T, S, variant, st()), sets, args.reps)
                flop = 2.0 * nf * M * C * T
                byt = 4.0 * nf * T * (C + M)
                print("   variant %d (%s) S=%d: error %.3e %.3e   %.1f us (%.0f TFLOP/s fp32-eq, %.2f TB/s algorithmic)" %
                      (variant, "16 tiles x 4 stages" if variant == 0 else "32 tiles x 2 stages", S, *e, t, flop / t * 1e-6, byt / t * 1e-6))
                del Ass, Bss
        dMs = [dM] + [dM.clone() for _ in range(2)]
        Vf = [V] + [V.clone() for _ in range(2)]
        t_lib = timeit(lambda i: torch.bmm(dMs[i].permute(1, 0, 2), Vf[i].permute(1, 2, 0)), 3, args.reps)
        print("   library fp32 bmm %.1f us" % t_lib)


if __name__ == "__main__":
    main()
