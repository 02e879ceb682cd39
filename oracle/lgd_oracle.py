"""CPU oracle for the LGD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch (CPU, fp32, autograd-capable) restatement of the reference's
dynamic-teacher forward and feature-distillation loss, written functionally
over a flat {state_dict name -> tensor} parameter dict so the same closed-form
weights drive the reference (golden generator), this oracle and the HIP path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; lgd_amd/ never does (the product path fails loudly when
the HIP library is missing).

Parity status: PINNED.  tests/golden/make_golden.py imports the real reference
modules from /root/reference (build container only) and stores their outputs
on closed-form inputs in tests/golden/*.npz; tests/test_oracle_golden.py checks
every function below against those vectors.

Each function cites the reference file:line it follows (paths relative to
/root/reference/models/customized_detectors/dynamic_teacher/ unless noted).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NUM_CLASSES = 80
C = 256
LEVELS = ("p3", "p4", "p5", "p6", "p7")


# --------------------------------------------------------------------------- shapes
def teacher_param_shapes(k=84):
    """state_dict names/shapes of the reference DynamicTeacher."""
    s = {}
    for stn, kk in (("stn_desc", k), ("stn_feat", 64)):
        p = "label_encoder_.%s." % stn
        s[p + "conv1.weight"] = (64, kk, 1)
        s[p + "conv2.weight"] = (128, 64, 1)
        s[p + "conv3.weight"] = (1024, 128, 1)
        s[p + "fc1.weight"] = (512, 1024)
        s[p + "fc2.weight"] = (256, 512)
        s[p + "fc3.weight"] = (kk * kk, 256)
        for n in ("conv1", "conv2", "conv3", "fc1", "fc2", "fc3"):
            s[p + n + ".bias"] = (s[p + n + ".weight"][0],)
    s["label_encoder_.conv1.weight"] = (64, k, 1)
    s["label_encoder_.conv2.weight"] = (128, 64, 1)
    s["label_encoder_.conv3.weight"] = (1024, 128, 1)
    s["label_encoder_.conv4.weight"] = (256, 1088, 1)
    for n in ("conv1", "conv2", "conv3", "conv4"):
        s["label_encoder_.%s.bias" % n] = (s["label_encoder_.%s.weight" % n][0],)
    s["canoni_proj_1D.0.0.weight"] = (C, C)
    s["canoni_proj_1D.0.0.bias"] = (C,)
    s["student_proj_2D.0.0.weight"] = (C, C, 3, 3)
    s["student_proj_2D.0.0.bias"] = (C,)
    s["local_inst_proj_2D.weight"] = (C, C, 3, 3)
    s["local_inst_proj_2D.bias"] = (C,)
    s["global_ctx_proj_1D.weight"] = (C, C)
    s["global_ctx_proj_1D.bias"] = (C,)
    s["local_inst_proj_1D.weight"] = (C, C)
    s["local_inst_proj_1D.bias"] = (C,)
    for i in (0, 3, 6):
        s["refinement_module.%d.weight" % i] = (C, C, 3, 3)
        s["refinement_module.%d.bias" % i] = (C,)
    s["multi_head_attn.in_proj_weight"] = (3 * C, C)
    s["multi_head_attn.in_proj_bias"] = (3 * C,)
    s["multi_head_attn.out_proj.weight"] = (C, C)
    s["multi_head_attn.out_proj.bias"] = (C,)
    return s


def adapter_param_shapes():
    """state_dict of the reference SequentialConvs (models/adapters/sequential_convs.py:8-15)."""
    s = {}
    for i in (0, 2, 4):
        s["adapter.%d.weight" % i] = (C, C, 3, 3)
        s["adapter.%d.bias" % i] = (C,)
    return s


# --------------------------------------------------------------------------- a-1 descriptors
@torch.no_grad()
def encode_box_descriptors(gt, img_h, img_w, add_ctx, box_format="x1y1x2y2"):
    """label_encoder.py:12-115 (one_hot category format).

    gt: list of (boxes (Ni,4) float32 tensor, classes (Ni,) int64 tensor), absolute px.
    Returns (descriptors list B x (Ni',84) in [-1,1], boxlists list B x Ni' x 4 python floats
    (clamped, un-normalised), inst_labels list B x (Ni,)).
    """
    descs, boxlists, labels_out = [], [], []
    for boxes, classes in gt:
        n = int(boxes.shape[0])
        if n > 0:
            bb = boxes.reshape(n, 4).to(torch.float32).clone()
            lab = classes.reshape(n, 1)
        else:  # label_encoder.py:64-66: empty image -> one unit box, class vector all zero
            bb = torch.tensor([[0.0, 0.0, 1.0, 1.0]])
            lab = torch.zeros((1, 1))
        if box_format == "x1y1wh":  # utils.py:26-38
            bb = torch.stack([bb[:, 0], bb[:, 1], bb[:, 0] + bb[:, 2] - 1.0, bb[:, 1] + bb[:, 3] - 1.0], 1)
        if add_ctx and n > 0:  # label_encoder.py:75-77, context box appended LAST
            bb = torch.cat([bb, torch.tensor([[0.0, 0.0, float(img_w), float(img_h)]])], 0)
        # utils.py:40-51 clamp to [0, w-1] x [0, h-1]
        bb = torch.stack([bb[:, 0].clamp(0, img_w - 1), bb[:, 1].clamp(0, img_h - 1),
                          bb[:, 2].clamp(0, img_w - 1), bb[:, 3].clamp(0, img_h - 1)], 1)
        boxlists.append(bb.clone().tolist())
        norm = bb.clone()
        norm[:, [0, 2]] /= img_w
        norm[:, [1, 3]] /= img_h
        rows = norm.shape[0]
        onehot = torch.zeros(rows, NUM_CLASSES)
        if n > 0:
            assert bool(((lab >= 0) & (lab <= NUM_CLASSES - 1)).all())
            onehot[:n].scatter_(1, lab.to(torch.int64), 1.0)  # ctx row stays all-zero
        d = torch.cat([norm, onehot], 1)
        assert bool(((d >= 0) & (d <= 1)).all())
        # utils.py:16-24 with a=-1,b=1,Min=0,Max=1:  (b-a)/(Max-Min)*(x-Min)+a
        d = (1.0 - (-1.0)) / (1.0 - 0.0) * (d - 0.0) + (-1.0)
        descs.append(d)
        labels_out.append(lab.reshape(-1))
    return descs, boxlists, labels_out


# --------------------------------------------------------------------------- a-2 / a-3 label encoder
def _pw(p, name, x):
    """pointwise conv1d on length-1 sequences == row-wise linear (weight (out,in,1))."""
    w = p[name + ".weight"]
    return F.linear(x, w.reshape(w.shape[0], -1), p[name + ".bias"])


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), eps=1e-5)


def stn_forward(p, prefix, x, k):
    """spatial_transformer.py:30-47; x (T,k) -> (T,k,k); no identity shortcut."""
    h = F.relu(_ln(_pw(p, prefix + "conv1", x)))
    h = F.relu(_ln(_pw(p, prefix + "conv2", h)))
    h = F.relu(_ln(_pw(p, prefix + "conv3", h)))  # max over the length-1 axis is a no-op
    h = F.relu(_ln(F.linear(h, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"])))
    h = F.relu(_ln(F.linear(h, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])))
    h = F.linear(h, p[prefix + "fc3.weight"], p[prefix + "fc3.bias"])
    return h.view(-1, k, k)


def label_encoder_forward(p, descs, return_all=False):
    """label_encoder.py:216-276 with R=1, noise 0.  descs: list B x (Ni,84) -> (T,256)."""
    counts = [int(d.shape[0]) for d in descs]
    x = torch.cat(descs, 0).to(p["label_encoder_.conv1.weight"].dtype)  # (T,84)
    k = x.shape[1]
    pre = "label_encoder_."
    m1 = stn_forward(p, pre + "stn_desc.", x, k)
    x1 = torch.bmm(x.unsqueeze(1), m1).squeeze(1)  # (x^T M)^T
    h = F.relu(_ln(_pw(p, pre + "conv1", x1)))
    m2 = stn_forward(p, pre + "stn_feat.", h, 64)
    xf = torch.bmm(h.unsqueeze(1), m2).squeeze(1)
    h2 = F.relu(_ln(_pw(p, pre + "conv2", xf)))
    h3 = F.relu(_ln(_pw(p, pre + "conv3", h2)))
    g = torch.stack([t.max(dim=0)[0] for t in h3.split(counts, 0)], 0)  # per-image max (B,1024)
    g_rows = torch.cat([g[b:b + 1].expand(n, -1) for b, n in enumerate(counts)], 0)
    out = F.relu(_ln(_pw(p, pre + "conv4", torch.cat([xf, g_rows], 1))))
    if return_all:
        return out, m1, m2
    return out


# --------------------------------------------------------------------------- a-4 masks
@torch.no_grad()
def inside_box_mask(boxlist, src_hw, dst_hw):
    """utils.py:53-89.  boxlist: N x 4 python floats (clamped xyxy in src px) -> (N, H*W) float {0,1}."""
    bt = torch.tensor(boxlist, dtype=torch.float32).reshape(-1, 4)
    r_h, r_w = dst_hw[0] / src_hw[0], dst_hw[1] / src_hw[1]  # python doubles, cast to fp32 by the multiply
    bt[:, [0, 2]] = bt[:, [0, 2]] * r_w
    bt[:, [1, 3]] = bt[:, [1, 3]] * r_h
    xc = (bt[:, 0] + bt[:, 2]) * 0.5
    yc = (bt[:, 1] + bt[:, 3]) * 0.5
    w_ = bt[:, 2] - bt[:, 0]
    h_ = bt[:, 3] - bt[:, 1]
    ys = torch.arange(dst_hw[0])
    xs = torch.arange(dst_hw[1])
    iny = (torch.abs(yc[:, None] - ys[None, :]) / h_[:, None]) <= 0.5  # (N,H)
    inx = (torch.abs(xc[:, None] - xs[None, :]) / w_[:, None]) <= 0.5  # (N,W)
    return (iny[:, :, None] & inx[:, None, :]).flatten(1).float()


@torch.no_grad()
def box_rects(boxlist, src_hw, dst_hw):
    """Integer form of inside_box_mask: (N,4) int32 [x0,x1,y0,y1] inclusive; empty -> x0>x1 or y0>y1.

    The predicate |c-p|/s <= 0.5 is monotone in |c-p| under fp32 rounding, so each axis'
    true-set is one interval; this is what the HIP box-prep kernel emits.
    """
    n = len(boxlist)
    out = np.zeros((n, 4), np.int32)
    bt = torch.tensor(boxlist, dtype=torch.float32).reshape(-1, 4)
    r_h, r_w = dst_hw[0] / src_hw[0], dst_hw[1] / src_hw[1]
    x1 = bt[:, 0] * r_w
    x2 = bt[:, 2] * r_w
    y1 = bt[:, 1] * r_h
    y2 = bt[:, 3] * r_h
    iny = (torch.abs(((y1 + y2) * 0.5)[:, None] - torch.arange(dst_hw[0])[None]) / (y2 - y1)[:, None]) <= 0.5
    inx = (torch.abs(((x1 + x2) * 0.5)[:, None] - torch.arange(dst_hw[1])[None]) / (x2 - x1)[:, None]) <= 0.5
    for i in range(n):
        xs = torch.nonzero(inx[i]).flatten()
        ys = torch.nonzero(iny[i]).flatten()
        out[i] = [int(xs[0]) if len(xs) else 0, int(xs[-1]) if len(xs) else -1,
                  int(ys[0]) if len(ys) else 0, int(ys[-1]) if len(ys) else -1]
        if len(xs):
            assert len(xs) == int(xs[-1]) - int(xs[0]) + 1
        if len(ys):
            assert len(ys) == int(ys[-1]) - int(ys[0]) + 1
    return out


# --------------------------------------------------------------------------- a-5 mask pooling
def mask_pool(feat, masks):
    """dynamic_teacher.py:81-103.  feat (B,C,H,W); masks list B x (Ni,HW) -> (T,C)."""
    flat = feat.flatten(2)
    out = []
    for b, m in enumerate(masks):
        m = m.to(feat.dtype)  # fp64 "truth" runs reuse the fp32-exact masks
        pooled = m @ flat[b].T
        cnt = torch.maximum(m.sum(-1), torch.ones((), dtype=feat.dtype))
        out.append(pooled / cnt[:, None])
    return torch.cat(out, 0)


# --------------------------------------------------------------------------- a-7 attention
def mha_blockdiag(p, q_in, kv_in, counts, heads=8):
    """nn.MultiheadAttention(256, 8) semantics as called at dynamic_teacher.py:270 (seq-first,
    batch 1, bool mask True=blocked between different images, dropout 0).
    q_in (T,C) queries; kv_in (T,C) keys=values; counts = boxes per image."""
    w, bias = p["multi_head_attn.in_proj_weight"], p["multi_head_attn.in_proj_bias"]
    E = q_in.shape[1]
    d = E // heads
    q = F.linear(q_in, w[:E], bias[:E]) * (1.0 / math.sqrt(d))
    k = F.linear(kv_in, w[E:2 * E], bias[E:2 * E])
    v = F.linear(kv_in, w[2 * E:], bias[2 * E:])
    T = q.shape[0]
    img = torch.repeat_interleave(torch.arange(len(counts)), torch.tensor(counts))
    blocked = (img[:, None] != img[None, :]).to(q.device)
    qh = q.view(T, heads, d).transpose(0, 1)
    kh = k.view(T, heads, d).transpose(0, 1)
    vh = v.view(T, heads, d).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2)).masked_fill(blocked[None], float("-inf"))
    a = torch.softmax(s, -1)
    o = torch.bmm(a, vh).transpose(0, 1).reshape(T, E)
    return F.linear(o, p["multi_head_attn.out_proj.weight"], p["multi_head_attn.out_proj.bias"])


# --------------------------------------------------------------------------- a-8 rendering
def _gn1(x):
    return F.group_norm(x, 1, eps=1e-5)


def render(p, attn_out, masks, counts, hw, add_ctx):
    """dynamic_teacher.py:106-190 for one level.  attn_out (T,C); masks list B x (Ni,HW)."""
    B = len(counts)
    rows = attn_out.split(counts, 0)
    painted = []
    for b in range(B):
        inst = rows[b][:-1] if add_ctx else rows[b]
        m = masks[b][:-1] if add_ctx else masks[b]
        proj = F.linear(inst, p["local_inst_proj_1D.weight"], p["local_inst_proj_1D.bias"])
        painted.append(proj.T @ m.to(proj.dtype))  # (C,HW): per-pixel SUM over covering boxes
    fmap = torch.cat(painted, 0).reshape(B, -1, hw[0], hw[1])
    fmap = F.conv2d(fmap, p["local_inst_proj_2D.weight"], p["local_inst_proj_2D.bias"], padding=1)
    if add_ctx:
        ctx = torch.stack([r[-1] for r in rows], 0)
        ctx = F.linear(ctx, p["global_ctx_proj_1D.weight"], p["global_ctx_proj_1D.bias"])
        fmap = fmap + ctx[:, :, None, None]
    return F.relu(fmap)


def refine(p, x):
    """dynamic_teacher.py:67-73."""
    for i, act in ((0, True), (3, True), (6, False)):
        x = _gn1(F.conv2d(x, p["refinement_module.%d.weight" % i], p["refinement_module.%d.bias" % i], padding=1))
        if act:
            x = F.relu(x)
    return x


# --------------------------------------------------------------------------- a-10 teacher forward
def teacher_forward(p, feats, gt, img_hw, add_ctx=True, interact="stuGuided", detach_app=False,
                    box_format="x1y1x2y2", return_intermediates=False):
    """dynamic_teacher.py:209-301.  feats: dict p3..p7 of (B,C,H,W); gt as encode_box_descriptors.
    Returns (teacher feature dict, inst_labels, masks[level][image])."""
    descs, boxlists, inst_labels = encode_box_descriptors(gt, img_hw[0], img_hw[1], add_ctx, box_format)
    label_embed = label_encoder_forward(p, [d.to(p["canoni_proj_1D.0.0.weight"].device) for d in descs])
    counts = [len(b) for b in boxlists]
    if detach_app:
        feats = {k: v.detach() for k, v in feats.items()}
    canoni = F.relu(_ln(F.linear(label_embed, p["canoni_proj_1D.0.0.weight"], p["canoni_proj_1D.0.0.bias"])))
    keys = list(feats.keys())
    proj = {k: F.relu(_gn1(F.conv2d(feats[k], p["student_proj_2D.0.0.weight"], p["student_proj_2D.0.0.bias"],
                                    padding=1))) for k in keys}
    hws = [tuple(feats[k].shape[-2:]) for k in keys]
    dev = feats[keys[0]].device  # (the oracle also runs on a GPU for diagnostics; masks are always built on the CPU)
    masks = [[inside_box_mask(bl, img_hw, hw).to(dev) for bl in boxlists] for hw in hws]
    app = [mask_pool(proj[k], masks[i]) for i, k in enumerate(keys)]
    if interact == "stuGuided":
        att = [mha_blockdiag(p, a, canoni, counts) for a in app]
    elif interact == "labelGuided":
        att = [mha_blockdiag(p, canoni, a, counts) for a in app]
    elif interact == "student_fill":
        att = app
    elif interact == "teacher_fill":
        att = [canoni for _ in app]
    else:
        raise ValueError(interact)
    raw = [render(p, att[i], masks[i], counts, hws[i], add_ctx) for i in range(len(keys))]
    tea = {k: refine(p, raw[i]) for i, k in enumerate(keys)}
    if return_intermediates:
        return tea, inst_labels, masks, dict(descs=descs, boxlists=boxlists, label_embed=label_embed,
                                             canoni=canoni, proj=proj, app=app, att=att, raw=raw)
    return tea, inst_labels, masks


# --------------------------------------------------------------------------- a-11 / a-12 distill
def adapter_forward(p, x):
    """models/adapters/sequential_convs.py:8-15."""
    x = F.relu(F.conv2d(x, p["adapter.0.weight"], p["adapter.0.bias"], padding=1))
    x = F.relu(F.conv2d(x, p["adapter.2.weight"], p["adapter.2.bias"], padding=1))
    return F.conv2d(x, p["adapter.4.weight"], p["adapter.4.bias"], padding=1)


def distill_loss(p_adapter, stu, tea, coef=1.0, distill_flag=1):
    """models/base_distillator.py:34-64."""
    keys = sorted(stu.keys() & tea.keys())
    bs = tea[keys[0]].shape[0]
    s = [stu[k].detach() if distill_flag == 0 else stu[k] for k in keys]
    t = [tea[k].detach() for k in keys]
    s = [F.instance_norm(adapter_forward(p_adapter, f), eps=1e-5) for f in s]
    t = [F.instance_norm(f, eps=1e-5) for f in t]
    s = torch.cat([f.reshape(bs, -1) for f in s], 1)
    t = torch.cat([f.reshape(bs, -1) for f in t], 1)
    return coef * F.mse_loss(t, s)


def in_mse(a_list, b_list, coef=1.0):
    """The loss tail of base_distillator.py:59-64 on already-adapted maps (what the HIP kernel K4 computes)."""
    bs = a_list[0].shape[0]
    s = torch.cat([F.instance_norm(f, eps=1e-5).reshape(bs, -1) for f in a_list], 1)
    t = torch.cat([F.instance_norm(f, eps=1e-5).reshape(bs, -1) for f in b_list], 1)
    return coef * F.mse_loss(t, s)
