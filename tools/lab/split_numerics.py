import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from wino_f6_numerics import lavin_f6, conv_direct64
torch.set_num_threads(8)
f32 = np.float32

def split(x, dt, n):
    x = torch.from_numpy(np.ascontiguousarray(x)).float()
    ps = []; r = x.clone()
    for _ in range(n):
        p = r.to(dt).float(); ps.append(p); r = r - p
    return ps

def prod(U, V, mode):
    # U [F,O,C], V [F,C,T]  fp32 ->  fp32
    Ut, Vt = torch.from_numpy(U).float(), torch.from_numpy(V).float()
    if mode == 'fp32':
        return torch.bmm(Ut, Vt).numpy()
    if mode == 'fp64':
        return torch.bmm(Ut.double(), Vt.double()).numpy()
    if mode == 'bf16x3':
        a = split(U, torch.bfloat16, 3); b = split(V, torch.bfloat16, 3)
        pairs = [(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]
    elif mode == 'bf16x2':
        a = split(U, torch.bfloat16, 2); b = split(V, torch.bfloat16, 2)
        pairs = [(0,0),(0,1),(1,0)]
    elif mode == 'fp16x2':
        # power-of-two scale per frequency so that max -> < 2^15
        sa = torch.exp2(14 - torch.floor(torch.log2(Ut.abs().amax((1,2), keepdim=True))))
        sb = torch.exp2(14 - torch.floor(torch.log2(Vt.abs().amax((1,2), keepdim=True))))
        a = split((Ut*sa).numpy(), torch.float16, 2); b = split((Vt*sb).numpy(), torch.float16, 2)
        pairs = [(0,0),(0,1),(1,0)]
        acc = None
        for i,j in reversed(pairs):
            t = torch.bmm(a[i], b[j]); acc = t if acc is None else acc + t
        return (acc/(sa*sb)).numpy()
    elif mode == 'fp16x2_4':
        sa = torch.exp2(14 - torch.floor(torch.log2(Ut.abs().amax((1,2), keepdim=True))))
        sb = torch.exp2(14 - torch.floor(torch.log2(Vt.abs().amax((1,2), keepdim=True))))
        a = split((Ut*sa).numpy(), torch.float16, 2); b = split((Vt*sb).numpy(), torch.float16, 2)
        acc = None
        for i,j in [(1,1),(1,0),(0,1),(0,0)]:
            t = torch.bmm(a[i], b[j]); acc = t if acc is None else acc + t
        return (acc/(sa*sb)).numpy()
    acc = None
    for i,j in reversed(pairs):
        t = torch.bmm(a[i], b[j]); acc = t if acc is None else acc + t
    return acc.numpy()

def conv_wino(x, w, mode):
    AT, G, BT = lavin_f6(); m = 6; n = 8
    C, H, W = x.shape
    A32, B32 = AT.astype(f32), BT.astype(f32)
    U = np.einsum("ai,ocij,bj->aboc", G, w.astype(np.float64), G).astype(f32)
    th, tw = (H + m - 1) // m, (W + m - 1) // m
    xp = np.zeros((C, th * m + 2, tw * m + 2), dtype=f32); xp[:, 1:H + 1, 1:W + 1] = x.astype(f32)
    tiles = np.stack([xp[:, i * m:i * m + n, j * m:j * m + n] for i in range(th) for j in range(tw)], 1)
    t1 = np.einsum("ai,ctij->ctaj", B32, tiles).astype(f32)
    V = np.einsum("bj,ctaj->abct", B32, t1).astype(f32)
    O = w.shape[0]; T = V.shape[-1]
    Uf = np.ascontiguousarray(U.reshape(64, O, C)); Vf = np.ascontiguousarray(V.reshape(64, C, T))
    Mf = prod(Uf, Vf, mode)
    M64 = prod(Uf, Vf, 'fp64')
    gerr = np.abs(Mf - M64).max(axis=(1,2)) / np.abs(M64).max(axis=(1,2))
    grms = (Mf - M64).std() / M64.std()
    M = Mf.astype(f32).reshape(8, 8, O, T)
    t2 = np.einsum("ia,abot->ibot", A32, M).astype(f32)
    Y = np.einsum("jb,ibot->otij", A32, t2).astype(f32)
    out = np.zeros((O, th * m, tw * m), dtype=f32); k = 0
    for i in range(th):
        for j in range(tw):
            out[:, i * m:(i + 1) * m, j * m:(j + 1) * m] = Y[:, k]; k += 1
    return out[:, :H, :W], gerr.max(), grms

C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
HW = int(sys.argv[2]) if len(sys.argv) > 2 else 48
rng = np.random.default_rng(1)
x = np.maximum(rng.standard_normal((C, HW, HW)), 0)
w = rng.standard_normal((C, C, 3, 3)) * (2.0 / (9 * C)) ** 0.5
ref = conv_direct64(x, w); scale = np.abs(ref).max()
for mode in ['fp64', 'fp32', 'bf16x3', 'fp16x2', 'fp16x2_4', 'bf16x2']:
    y, ge, gr = conv_wino(x, w, mode)
    e = np.abs(y - ref)
    print("%-9s gemm max err/scale %.2e rms/rms %.2e | conv max err/scale %.2e  rms err/rms %.2e" % (mode, ge, gr, e.max()/scale, e.std()/ref.std()))
