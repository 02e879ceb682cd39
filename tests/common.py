"""Shared helpers: rebuild the closed-form inputs of each golden case (same recipe as
tests/golden/make_golden.py) and load the reference outputs stored in tests/golden/*.npz."""
import os

import numpy as np
import torch

from lgd_amd import synth
from oracle import lgd_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLE_STRIDE = 37

CASES = {
    # name: (B, H, W, add_ctx, interact, box_format, coef, feat_seed)
    "c1_ctx_stuguided": (2, 512, 512, True, "stuGuided", "x1y1x2y2", 1.0, 11),
    "c1b_noctx_labelguided_wh": (3, 384, 512, False, "labelGuided", "x1y1wh", 1.0, 11),
    "c1c_noctx_stuguided": (2, 320, 480, False, "stuGuided", "x1y1x2y2", 2.5, 13),
    # the full BASELINE shape (configs 2 / 3), B=2, 10 boxes/img; losses stored, no gradients
    "c2_full_ctx_800x1344": (2, 800, 1344, True, "stuGuided", "x1y1x2y2", 1.0, 17),
    "c3_full_noctx_800x1344": (2, 800, 1344, False, "stuGuided", "x1y1x2y2", 1.0, 19),
}
# a case whose ReLU inputs / max-pool candidates all keep a MARGIN (tests/golden/make_golden_margin.py nudges the biases in fp64
# until they do; the nudged biases are inputs stored in the fixture): the reference's own gradients are a hard 1e-4 target there
MARGIN_CASES = {"c4_margin": (2, 192, 256, True, "stuGuided", "x1y1x2y2", 1.0, 23)}
FULL_STRIDE = 997
SMALL_CASES = [k for k in CASES if "_full_" not in k]


def stride_of(name):
    return FULL_STRIDE if "_full_" in name else SAMPLE_STRIDE


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def case_gt(name):
    if name == "c1_ctx_stuguided":
        gt = synth.synth_gt(2, 512, 512, 10, table=True)
    elif name == "c1b_noctx_labelguided_wh":
        gt = synth.synth_gt(3, 384, 512, 7, seed=9)
        gt[1] = (np.zeros((0, 4), np.float32), np.zeros((0,), np.int64))
        gt[2] = (gt[2][0][:4], gt[2][1][:4])
    elif name == "c1c_noctx_stuguided":
        gt = synth.synth_gt(2, 320, 480, 6, seed=4)
    elif name == "c2_masks_800x1344":
        gt = synth.synth_gt(8, 800, 1344, 10, seed=0)
    elif name == "c4_margin":
        gt = synth.synth_gt(2, 192, 256, 5, seed=21)
    elif "_full_" in name:
        gt = synth.synth_gt(2, 800, 1344, 10, seed=0)
    else:
        raise KeyError(name)
    return [(torch.from_numpy(b.copy()), torch.from_numpy(c.copy())) for b, c in gt]


def case_feats(name, requires_grad=False):
    B, H, W, _, _, _, _, seed = (CASES.get(name) or MARGIN_CASES[name])
    return {k: torch.from_numpy(v.copy()).requires_grad_(requires_grad)
            for k, v in synth.synth_features(B, H, W, seed=seed).items()}


def _nudged(p, case):
    """a margin case's parameters: the closed-form ones with the biases its fixture stores (`bias_<state_dict name>`)"""
    if case in MARGIN_CASES:
        g = golden(case)
        for k in p:
            if "bias_" + k in g.files:
                p[k] = g["bias_" + k].astype(np.float32)
    return p


def teacher_params(requires_grad=False, case=None):
    p = _nudged(synth.closed_form_params(O.teacher_param_shapes()), case)
    return {k: torch.from_numpy(v.copy()).requires_grad_(requires_grad) for k, v in p.items()}


def adapter_params(requires_grad=False, case=None):
    p = _nudged(synth.closed_form_params(O.adapter_param_shapes()), case)
    return {k: torch.from_numpy(v.copy()).requires_grad_(requires_grad) for k, v in p.items()}


def probes(tea):
    return {k: torch.from_numpy(synth.det_uniform(tuple(tea[k].shape), 900 + i, -1e-3, 1e-3)) for i, k in enumerate(tea.keys())}


def rel_err(a, b):
    a = torch.as_tensor(a).detach().cpu().to(torch.float64).reshape(-1)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64).reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def sample(t, stride=SAMPLE_STRIDE):
    f = t.detach().reshape(-1).double().cpu()
    return f[::stride].float().numpy(), float(f.sum()), float((f * f).sum())


def kink_robust_close(a, b, tol=1e-4, max_outlier_frac=5e-4, max_rel=2e-2):
    """Gradient comparison that tolerates ReLU-kink flips.

    Backward masks are `activation > 0`; forward values that differ in the last ulp between two
    fp32 platforms (MKL vs MIOpen/rocBLAS summation order) flip the mask of the few activations that
    sit within an ulp of zero, which changes isolated gradient elements by O(1) while everything else
    agrees to ~1e-6 (measured on MI355X: oracle-on-GPU vs oracle-on-CPU, identical torch ops, shows
    1e-3 relative L2 on exactly the levels where a flip happened and 1e-6 elsewhere).  So: all but a
    tiny fraction of elements must agree to `tol` (relative to the tensor's RMS), and the whole tensor
    to `max_rel`.  Returns (ok, message)."""
    a = torch.as_tensor(a).detach().cpu().double().reshape(-1)
    b = torch.as_tensor(b).detach().cpu().double().reshape(-1)
    rms = float(b.pow(2).mean().sqrt()) + 1e-30
    bad = ((a - b).abs() > tol * (b.abs() + rms)).double().mean().item()
    rel = float((a - b).norm() / (b.norm() + 1e-30))
    return (bad <= max_outlier_frac and rel <= max_rel), "outlier frac %.2e (max %.0e), rel L2 %.2e (max %.0e)" % (bad, max_outlier_frac, rel, max_rel)


def oracle_grads_on_device(name, device):
    """the ORACLE's fp32 torch ops executed on `device` (the reference's own arithmetic on this platform): feature and
    parameter gradients of  loss_distill(flag=1) + sum(teacher feats * probe)  for golden case `name`.
    Used to calibrate model-level gradient bounds: a ReLU input within rounding noise of zero gets its backward mask from the
    platform's summation order, so ANY fp32 evaluation on another platform (this one included) leaves the CPU reference's
    gradient by ~1e-3 on the levels where that happens (measured on MI355X, tools/diag_grad_chain.py: oracle-fp32-on-GPU vs
    fp64: 1e-3 at p3, 3..5e-3 at p5, 5e-6 at p4/p6/p7 -- and the HIP path shows the same numbers on the same levels)."""
    B, H, W, ctx, interact, fmt, coef, _ = CASES[name]
    p = {k: v.to(device).requires_grad_(True) for k, v in teacher_params().items()}
    pa = {k: v.to(device).requires_grad_(True) for k, v in adapter_params().items()}
    feats = {k: v.to(device).requires_grad_(True) for k, v in case_feats(name).items()}
    tea, _, _ = O.teacher_forward(p, feats, case_gt(name), (H, W), ctx, interact, False, fmt)
    loss = O.distill_loss(pa, feats, tea, coef, 1)
    pr = probes(tea)
    (loss + sum((tea[k] * pr[k].to(device)).sum() for k in tea)).backward()
    grads = {n: v.grad for n, v in p.items()}
    grads.update({"adapter." + n: v.grad for n, v in pa.items()})
    return {k: feats[k].grad for k in feats}, grads


# ---- FCOS fixtures (tests/golden/make_golden.py: run_fcos_gt_case / run_fcos_head_case, generated from the reference's in-tree FCOS)
FCOS_STRIDES = [8, 16, 32, 64, 128]
FCOS_SOI = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, float("inf")]]
FCOS_GT_CASES = {"fcos_gt_small_r15": ("small", 1.5), "fcos_gt_small_r0": ("small", 0.0), "fcos_gt_full_r15": ("full", 1.5)}
FCOS_HEAD_LEVELS = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]


def fcos_gt_inputs(case):
    """ground truth of the FCOS target-assignment fixtures: `small` = three 512x640 images -- random boxes, nested boxes of equal
    centre + a duplicate (min-area ties, first index wins), a crowd of 40 overlapping boxes; `full` = two images at the config-3
    shape 800x1344.  (An image WITHOUT boxes cannot go through the reference: its `gt_positions_area.min(dim=0)` raises on an empty
    dimension, thirdparty_heads/fcos.py:259.)"""
    if case == "small":
        H, W = 512, 640
        gts = synth.synth_gt(3, H, W, 8, seed=9)
        nested = np.array([[100.0, 100.0, 420.0, 400.0], [180.0, 175.0, 340.0, 325.0], [180.0, 175.0, 340.0, 325.0]], np.float32)
        gts[1] = (np.concatenate([gts[1][0], nested]), np.concatenate([gts[1][1], np.array([3, 4, 5], np.int64)]))
        u = synth.det_uniform((40, 5), 7771, 0.0, 1.0).astype(np.float64)
        x1, y1 = u[:, 0] * 560, u[:, 1] * 440
        crowd = np.stack([x1, y1, np.minimum(W - 1.0, x1 + 8 + u[:, 2] * 300), np.minimum(H - 1.0, y1 + 8 + u[:, 3] * 260)], 1)
        gts[2] = (crowd.astype(np.float32), np.minimum((u[:, 4] * 80).astype(np.int64), 79))
        return H, W, gts
    H, W = 800, 1344
    return H, W, synth.synth_gt(2, H, W, 10, seed=0)


def fcos_head_inputs(B=2):
    """closed-form features and output probes of the `fcos_head` fixture."""
    feats = [synth.det_uniform((B, 256, h, w), 4100 + i, -1.0, 1.0) for i, (h, w) in enumerate(FCOS_HEAD_LEVELS)]
    co = {"logits": 80, "reg": 4, "ctr": 1}
    probes = {kind: [synth.det_uniform((B, co[kind], h, w), 4200 + 10 * i + len(kind), -1.0, 1.0) for i, (h, w) in enumerate(FCOS_HEAD_LEVELS)]
              for kind in co}
    return feats, probes


# ---- f16x2 split operands of csrc/h2.hip, built with torch ops (an independent statement of the two formats: include/lgd_hip.h, K10)
def h2_pow2_scale(amax):
    """per-batch power-of-two multipliers 2^e with amax * 2^e < 2^15 (amax > 0)"""
    return torch.exp2(14 - torch.floor(torch.log2(amax)))


def h2_split_rows(x, scale):
    """x (rows, nb, T) fp32, scale (nb,) -> int32 (rows, nb, T) split rows: per row blocks of 32 tiles, 32 h values then 32 m values"""
    rows, nb, T = x.shape
    assert T % 32 == 0
    t = x * scale.view(1, -1, 1)
    h = t.half()
    m = (t - h.float()).half()
    pk = torch.stack([h.view(rows, nb, T // 32, 32), m.view(rows, nb, T // 32, 32)], dim=3).contiguous()
    return pk.view(torch.int32).view(rows, nb, T)


def h2_unsplit_rows(buf, inv):
    """int32 (rows, nb, T) split rows -> fp32 (rows, nb, T) values (h + m) * inv[b] (inv: (nb,) or (1,))"""
    rows, nb, T = buf.shape
    pk = buf.contiguous().view(torch.float16).view(rows, nb, T // 32, 2, 32).float()
    v = (pk[:, :, :, 0] + pk[:, :, :, 1]).reshape(rows, nb, T)
    return v * (inv.view(1, -1, 1) if inv.numel() > 1 else inv)


def h2_split_image(a, scale):
    """a (nb, M, K) fp32, scale (nb,) -> uint8 image [nb][K/16][2 pieces][ceil(M/32)][lane = (k % 16 / 8) * 32 + m % 32][8 f16]"""
    nb, M, K = a.shape
    rbp, ktp = (M + 31) // 32, (K + 15) // 16
    t = torch.zeros((nb, rbp * 32, ktp * 16), dtype=torch.float32, device=a.device)
    t[:, :M, :K] = a * scale.view(-1, 1, 1)
    h = t.half()
    m = (t - h.float()).half()
    frag = lambda p: p.view(nb, rbp, 32, ktp, 2, 8).permute(0, 3, 1, 4, 2, 5)   # noqa: E731   (nb, ktp, rbp, k-group, row, 8)
    img = torch.stack([frag(h), frag(m)], dim=2).contiguous()                  # (nb, ktp, piece, rbp, k-group, row, 8)
    return img.view(torch.uint8).view(-1)


def h2_unsplit_image(img, nb, M, K, inv, padded=False):
    """uint8 image -> fp32 (nb, M, K) values (h + m) * inv[b] (padded: the whole (nb, 32 ceil(M/32), 16 ceil(K/16)) array the image holds)"""
    rbp, ktp = (M + 31) // 32, (K + 15) // 16
    t = img.view(torch.float16).view(nb, ktp, 2, rbp, 2, 32, 8).float()
    v = (t[:, :, 0] + t[:, :, 1]).permute(0, 2, 4, 1, 3, 5).reshape(nb, rbp * 32, ktp * 16)   # (nb, rbp, row, ktp, k-group, 8)
    return (v if padded else v[:, :M, :K]) * inv.view(-1, 1, 1)
