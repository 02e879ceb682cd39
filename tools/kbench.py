"""Kernel micro-benchmark at BASELINE config-2 shapes (B=8, 800x1344, C=256, 11 boxes/img incl. ctx).
Buffers rotate over > 1 GB so every launch reads HBM-cold data (Infinity Cache is 256 MiB); per-kernel
times come from the library's own HIP-event timing (same facility bench.py uses)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--n", type=int, default=10)
ap.add_argument("--ctx", type=int, default=1)
ap.add_argument("--iters", type=int, default=12)
ap.add_argument("--nset", type=int, default=3)
a = ap.parse_args()
B, H, W, C = a.B, 800, 1344, 256
level_hw = synth.pyramid_shapes(H, W)
P = B * C * sum(h * w for h, w in level_hw) * 4
bl = []
for b, _ in synth.synth_gt(B, H, W, a.n, seed=0):
    bb = torch.from_numpy(b).clone()
    if a.ctx:
        bb = torch.cat([bb, torch.tensor([[0., 0., W, H]])])
    bb[:, [0, 2]] = bb[:, [0, 2]].clamp(0, W - 1); bb[:, [1, 3]] = bb[:, [1, 3]].clamp(0, H - 1)
    bl.append(bb)
counts = [len(x) for x in bl]
boxes = torch.cat(bl).cuda()
sets = [[torch.randn(B, C, h, w, device="cuda") for h, w in level_hw] for _ in range(2 * a.nset)]
geom = ops.BoxGeometry(boxes, counts, (H, W), level_hw)
vals = torch.randn(len(level_hw), sum(counts), C, device="cuda")
cvec = torch.randn(len(level_hw), B, C, device="cuda", requires_grad=True)
q = torch.randn(5, sum(counts), C, device="cuda", requires_grad=True)
kv = torch.randn(1, sum(counts), C, device="cuda", requires_grad=True)
mh = torch.nn.MultiheadAttention(C, 8).cuda()
wconv = (torch.randn(C, C, 3, 3, device="cuda") * 0.02).requires_grad_(True)
bconv = torch.zeros(C, device="cuda")


def one(it):
    x, y = sets[2 * (it % a.nset)], sets[2 * (it % a.nset) + 1]
    ops.BoxGeometry(boxes, counts, (H, W), level_hw)
    ops._box_sum(geom, x, True, False)
    ops._box_paint(geom, vals, False, bool(a.ctx))
    fr = [f.detach().requires_grad_(True) for f in x]
    loss = ops.distill_in_mse(fr, y, 1.0)
    torch.autograd.grad(loss, fr)
    ys = ops.gn1(fr, True)
    torch.autograd.grad(sum(v.sum() for v in ys), fr)
    pl = ops.gn_relu_mask_pool(geom, fr)
    torch.autograd.grad(pl.sum(), fr)
    zs = ops.bias_ctx_relu(fr, cvec)
    torch.autograd.grad(sum(v.sum() for v in zs), fr)
    cy = ops._Conv3x3.apply(wconv, bconv, True, ops._WINO_TILE, *fr)  # one 256->256 filter + ReLU over the pyramid: wino_in / wino_out / wino_out_t / wino_in_t
    torch.autograd.grad(sum(v.sum() for v in cy), [wconv] + fr)
    o = ops.mha_blockdiag(q, kv, counts, mh.in_proj_weight, mh.in_proj_bias, mh.out_proj.weight, mh.out_proj.bias, 8, geom.img_off)
    torch.autograd.grad(o.sum(), [q, kv])


for i in range(3):
    one(i)
torch.cuda.synchronize()
ops.kernel_timer_enable(True)
for i in range(a.iters):
    one(i)
torch.cuda.synchronize()
t = ops.kernel_timer_collect()
ops.kernel_timer_enable(False)
alg = {"in_moments_kernel": 2 * P, "in_mse_bwd_kernel": 3 * P, "box_sum_kernel": P, "box_paint_kernel": P, "gn_stats_kernel": P,
       "gn_apply_kernel": 2 * P, "gn_bwd_stats_kernel": 2 * P, "gn_bwd_apply_kernel": 3 * P, "ctx_relu_kernel": 2 * P, "ctx_relu_bwd_kernel": 3 * P,
       "gn_pool_kernel": P, "gn_pool_bwd_apply_kernel": 2 * P}
tl = ops._WINO_TILE
FB = 4 * (tl + 2) ** 2 * C * sum(B * ((h + tl - 1) // tl) * ((w + tl - 1) // tl) for h, w in level_hw)  # one frequency buffer
MB = ops._WINO_MASK_DTYPE[tl].itemsize * C * (FB // (4 * (tl + 2) ** 2 * C))  # ReLU mask table: one entry per tile
alg.update({"wino_in_kernel": P + FB, "wino_out_kernel": P + FB + MB, "wino_out_t_kernel": P + MB + FB, "wino_in_t_kernel": P + FB})
res = {}
for k, (n, ms, _lo, _hi) in sorted(t.items(), key=lambda kv: -kv[1][1]):
    us = 1e3 * ms / n
    res[k] = {"us": us, "GBps": alg[k] / us / 1e3 if k in alg else None}
    print("%-24s n=%3d avg %8.1f us   %s" % (k, n, us, ("%6.0f GB/s (%.0f%% of 8 TB/s)" % (alg[k] / us / 1e3, alg[k] / us / 1e3 / 80)) if k in alg else ""))
print(json.dumps(res))
