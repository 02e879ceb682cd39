"""Kernel micro-benchmark at BASELINE config-2 shapes (B=8, 800x1344, C=256, 11 boxes/img incl. ctx).
Prints per-kernel time and algorithmic GB/s (SURVEY.md section 8d byte counts)."""
import argparse
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import ops, synth


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(iters))
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--n", type=int, default=10)
    ap.add_argument("--ctx", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    B, H, W, C = a.B, 800, 1344, 256
    level_hw = synth.pyramid_shapes(H, W)
    P = B * C * sum(h * w for h, w in level_hw) * 4
    gt = synth.synth_gt(B, H, W, a.n, seed=0)
    bl = []
    for b, _ in gt:
        bb = torch.from_numpy(b).clone()
        if a.ctx:
            bb = torch.cat([bb, torch.tensor([[0., 0., W, H]])])
        bb[:, [0, 2]] = bb[:, [0, 2]].clamp(0, W - 1)
        bb[:, [1, 3]] = bb[:, [1, 3]].clamp(0, H - 1)
        bl.append(bb)
    counts = [len(x) for x in bl]
    boxes = torch.cat(bl).cuda()
    feats = [torch.randn(B, C, h, w, device="cuda") for h, w in level_hw]
    feats2 = [torch.randn(B, C, h, w, device="cuda") for h, w in level_hw]
    geom = ops.BoxGeometry(boxes, counts, (H, W), level_hw)
    vals = torch.randn(len(level_hw), sum(counts), C, device="cuda")
    res = {}

    def rep(name, fn, nbytes):
        med, best = timeit(fn, a.iters)
        res[name] = dict(us=med * 1e6, best_us=best * 1e6, GBps=nbytes / med / 1e9, bytes=nbytes)
        print("%-22s %9.1f us (best %9.1f)  %8.1f GB/s algorithmic  (%.1f MB)" % (name, med * 1e6, best * 1e6, nbytes / med / 1e9, nbytes / 1e6))

    rep("box_prep", lambda: ops.BoxGeometry(boxes, counts, (H, W), level_hw), 1)
    rep("mask_pool_fwd", lambda: ops._box_sum(geom, feats, True, False), P)
    rep("render_paint_fwd", lambda: ops._box_paint(geom, vals, False, bool(a.ctx)), P)
    rep("mask_pool_bwd(paint)", lambda: ops._box_paint(geom, vals, True, False), P)
    rep("render_bwd(sum)", lambda: ops._box_sum(geom, feats, False, bool(a.ctx)), P)
    rep("distill_fwd", lambda: ops.distill_in_mse(feats, feats2, 1.0), 2 * P)
    fr = [f.clone().requires_grad_(True) for f in feats]
    loss = ops.distill_in_mse(fr, feats2, 1.0)
    rep("distill_bwd", lambda: torch.autograd.grad(loss, fr, retain_graph=True), 3 * P)
    # torch references for context
    rep("torch copy (2P)", lambda: [f.clone() for f in feats], 2 * P)
    import torch.nn.functional as F
    rep("torch IN+mse (ref ops)", lambda: sum(F.mse_loss(F.instance_norm(x), F.instance_norm(y), reduction="sum") for x, y in zip(feats, feats2)), 2 * P)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
