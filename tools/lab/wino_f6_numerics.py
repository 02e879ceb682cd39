"""fp32 conditioning of F(m x m, 3x3) for m = 2, 4, 6 (numpy simulation of transform -> per-frequency channel product -> inverse
transform with every intermediate rounded to fp32, against the fp64 direct convolution).  VERDICT r02 item 5: the go / no-go for
F(6x6,3x3) is its error, so the error is measured first, on the CPU, before any kernel is written.
usage: python tools/lab/wino_f6_numerics.py [C=256] [HW=48]"""
import sys
from fractions import Fraction

import numpy as np


def toom_cook(points, m, r=3):
    """A^T (m x n), G (n x r), B^T (n x n) for F(m, r) on the finite `points` + infinity (n = m + r - 1), exact rationals.
    Convention y = A^T [(G g) . (B^T d)]."""
    n = m + r - 1
    pts = [Fraction(p) for p in points]
    assert len(pts) == n - 1
    # Vandermonde-type construction (Lavin / Barabasz): A^T rows = powers of the points, last column for infinity
    AT = [[(p ** i) for p in pts] + [Fraction(1 if i == m - 1 else 0)] for i in range(m)]
    G = []
    for p in pts:
        denom = Fraction(1)
        for q in pts:
            if q != p:
                denom *= (p - q)
        G.append([(p ** j) / denom for j in range(r)])
    G.append([Fraction(0)] * (r - 1) + [Fraction(1)])
    # B^T from the Lagrange basis: row i = coefficients of prod_{q != p_i} (x - q), last row = coefficients of prod (x - q)
    def polymul(a, b):
        out = [Fraction(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out
    BT = []
    for p in pts:
        poly = [Fraction(1)]
        for q in pts:
            if q != p:
                poly = polymul(poly, [-q, Fraction(1)])
        BT.append(poly + [Fraction(0)] * (n - len(poly)))
    poly = [Fraction(1)]
    for q in pts:
        poly = polymul(poly, [-q, Fraction(1)])
    BT.append(poly)
    f = lambda M: np.array([[float(x) for x in row] for row in M], dtype=np.float64)  # noqa: E731
    return f(AT), f(G), f(BT)


def check_exact(AT, G, BT, m):
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal(m + 2), rng.standard_normal(3)
    y = AT @ ((G @ g) * (BT @ d))
    ref = np.array([d[i] * g[0] + d[i + 1] * g[1] + d[i + 2] * g[2] for i in range(m)])
    return np.abs(y - ref).max()


def conv_direct64(x, w):
    C, H, W = x.shape
    xp = np.zeros((C, H + 2, W + 2)); xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W))
    for i in range(3):
        for j in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, i, j], xp[:, i:i + H, j:j + W])
    return out


def conv_wino32(x, w, AT, G, BT, m, filter64=True):
    """every stage in fp32 (filter transform optionally in fp64 then rounded: it runs once per step per filter)."""
    n = m + 2
    C, H, W = x.shape
    f32 = np.float32
    A32, B32 = AT.astype(f32), BT.astype(f32)
    if filter64:
        U = np.einsum("ai,ocij,bj->aboc", G, w.astype(np.float64), G).astype(f32)
    else:
        G32 = G.astype(f32)
        U = np.einsum("ai,ocij,bj->aboc", G32, w.astype(f32), G32).astype(f32)
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((C, th * m + 2, tw * m + 2), dtype=f32); xp[:, 1:H + 1, 1:W + 1] = x.astype(f32)
    tiles = np.stack([xp[:, i * m:i * m + n, j * m:j * m + n] for i in range(th) for j in range(tw)], 1)   # C, T, n, n
    # B^T d B as two fp32 passes (rows, then columns), like a kernel would
    t1 = np.einsum("ai,ctij->ctaj", B32, tiles).astype(f32)
    V = np.einsum("bj,ctaj->abct", B32, t1).astype(f32)
    M = np.einsum("aboc,abct->abot", U, V).astype(f32)      # fp32 sgemm per frequency
    t2 = np.einsum("ia,abot->ibot", A32, M).astype(f32)
    Y = np.einsum("jb,ibot->otij", A32, t2).astype(f32)      # O, T, m, m
    out = np.zeros((w.shape[0], th * m, tw * m), dtype=f32)
    k = 0
    for i in range(th):
        for j in range(tw):
            out[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = Y[:, k]; k += 1
    return out[:, :H, :W]


if __name__ == "__main__":
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    HW = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    rng = np.random.default_rng(1)
    x = np.maximum(rng.standard_normal((C, HW, HW)), 0)       # post-ReLU activations, like the path's maps
    w = rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5
    ref = conv_direct64(x, w)
    scale = np.abs(ref).max()
    d32 = conv_direct64(x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64))
    print("C=%d map %dx%d   output scale %.3f, rms %.3f" % (C, HW, HW, scale, ref.std()))
    cands = {
        "F(2x2) {0,1,-1}": (2, [0, 1, -1]),
        "F(4x4) {0,1,-1,2,-2}": (4, [0, 1, -1, 2, -2]),
        "F(4x4) {0,1,-1,1/2,-1/2}": (4, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2)]),
        "F(6x6) {0,1,-1,2,-2,1/2,-1/2}": (6, [0, 1, -1, 2, -2, Fraction(1, 2), Fraction(-1, 2)]),
        "F(6x6) {0,1,-1,1/2,-1/2,3/2,-3/2}": (6, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(3, 2), Fraction(-3, 2)]),
        "F(6x6) {0,1,-1,1/2,-1/2,1/4,-1/4}": (6, [0, 1, -1, Fraction(1, 2), Fraction(-1, 2), Fraction(1, 4), Fraction(-1, 4)]),
        "F(6x6) {0,1,-1,2,-2,1/4,-1/4}": (6, [0, 1, -1, 2, -2, Fraction(1, 4), Fraction(-1, 4)]),
        "F(6x6) {0,3/4,-3/4,3/2,-3/2,1/3,-1/3}": (6, [0, Fraction(3, 4), Fraction(-3, 4), Fraction(3, 2), Fraction(-3, 2), Fraction(1, 3), Fraction(-1, 3)]),
    }
    for name, (m, pts) in cands.items():
        AT, G, BT = toom_cook(pts, m)
        ex = check_exact(AT, G, BT, m)
        for f64 in (True, False):
            y = conv_wino32(x, w, AT, G, BT, m, filter64=f64)
            e = np.abs(y - ref)
            print("%-40s exact %.1e  filter %s: max err / scale %.2e   rms err / rms %.2e" % (name, ex, "fp64" if f64 else "fp32", e.max() / scale, e.std() / ref.std()))


def lavin_f6():
    """the matrices the kernels use (F(6x6,3x3) on the points 0, +-1, +-2, +-1/2, inf; Lavin & Gray's scaling)."""
    AT = np.array([[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, .5, -.5, 0], [0, 1, 1, 4, 4, .25, .25, 0], [0, 1, -1, 8, -8, .125, -.125, 0],
                   [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]], dtype=np.float64)
    G = np.array([[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
                  [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0, 0, 1]], dtype=np.float64)
    BT = np.array([[1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
                   [0, .5, .25, -2.5, -1.25, 2, 1, 0], [0, -.5, .25, 2.5, -1.25, -2, 1, 0], [0, 2, 4, -2.5, -5, .5, 1, 0],
                   [0, -2, 4, 2.5, -5, -.5, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]], dtype=np.float64)
    return AT, G, BT


if __name__ == "__main__":
    AT, G, BT = lavin_f6()
    print("Lavin F(6,3) exactness:", check_exact(AT, G, BT, 6))
    for f64 in (True, False):
        y = conv_wino32(x, w, AT, G, BT, 6, filter64=f64)
        e = np.abs(y - ref)
        print("%-40s filter %s: max err / scale %.2e   rms err / rms %.2e" % ("F(6x6) Lavin matrices", "fp64" if f64 else "fp32", e.max() / scale, e.std() / ref.std()))
