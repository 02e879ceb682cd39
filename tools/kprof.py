"""Launch every hand-written HIP kernel a few times at BASELINE config-2 shapes (for rocprofv3 passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lgd_amd import ops, synth
B, H, W, C = 8, 800, 1344, 256
level_hw = synth.pyramid_shapes(H, W)
gt = synth.synth_gt(B, H, W, 10, seed=0)
bl = []
for b, _ in gt:
    bb = torch.cat([torch.from_numpy(b), torch.tensor([[0., 0., W, H]])])
    bb[:, [0, 2]] = bb[:, [0, 2]].clamp(0, W - 1); bb[:, [1, 3]] = bb[:, [1, 3]].clamp(0, H - 1)
    bl.append(bb)
counts = [len(x) for x in bl]
boxes = torch.cat(bl).cuda()
NSET = 3  # rotate over > 256 MiB of distinct buffers so reads come from HBM, not the Infinity Cache
sets = [[torch.randn(B, C, h, w, device="cuda") for h, w in level_hw] for _ in range(2 * NSET)]
geom = ops.BoxGeometry(boxes, counts, (H, W), level_hw)
vals = torch.randn(len(level_hw), sum(counts), C, device="cuda")
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    a, b = sets[2 * (it % NSET)], sets[2 * (it % NSET) + 1]
    ops._box_sum(geom, a, True, False)
    ops._box_paint(geom, vals, False, True)
    fr = [f.detach().requires_grad_(True) for f in a]
    loss = ops.distill_in_mse(fr, b, 1.0)
    torch.autograd.grad(loss, fr)
    ys = ops.gn1(fr, True)
    torch.autograd.grad(sum(y.sum() for y in ys), fr)
    pl = ops.gn_relu_mask_pool(geom, fr)
    torch.autograd.grad(pl.sum(), fr)
    cvec = torch.randn(len(level_hw), B, C, device="cuda", requires_grad=True)
    zs = ops.bias_ctx_relu(fr, cvec)
    torch.autograd.grad(sum(z.sum() for z in zs), fr)
torch.cuda.synchronize()
print("done")
