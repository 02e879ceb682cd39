"""CPU/torch restatements of the STUDENT-side arithmetic the HIP kernels replace -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Parity status: UNPINNED.  These functions restate third-party code that is NOT under /root/reference and is not
installed here (detectron2 v0.3: Matcher / RetinaNet.losses / Box2BoxTransform / ModulatedDeformConv; fvcore:
sigmoid_focal_loss_jit, smooth_l1_loss; cvpods: iou_loss) from their public definitions, plus the
reference's own FCOS target assignment, which IS in the tree and is cited per line.  No golden vector of the reference
exists for them; the HIP kernels (anchor_match.hip, focal.hip, box_reg.hip, dcn.hip, fcos_target.hip, gn.hip groups > 1)
are held to these restatements, and the restatements to hand-computed cases in tests/test_host_cpu.py.

Only tests/ may import this module; lgd_amd/ never does.
"""
import torch
import torch.nn.functional as F

INF = float("inf")


# ------------------------------------------------------------------------------------------------ RetinaNet side
def pairwise_iou(a, b):
    """detectron2 pairwise_iou on raw (M,4) / (R,4) xyxy tensors -> (M,R); empty intersection -> 0."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b[None, :] - inter), torch.zeros_like(inter))


def label_anchors(anchors, gt, num_classes, iou_thresholds=(0.4, 0.5), iou_labels=(0, -1, 1)):
    """detectron2 RetinaNet.label_anchors = Matcher(thresholds, labels, allow_low_quality_matches=True) per image
    [call site ref: models/customized_detectors/retinanet.py:66-67; thresholds configs/.../Base-RetinaNet.yaml:12-13].
    anchors (R,4); gt = list of (boxes (M,4), classes (M,)).  Returns per image: class ids (R,) int64 with
    num_classes = background and -1 = ignore, and the matched GT box of every anchor (R,4) (zeros if the image is empty)."""
    lo, hi = iou_thresholds
    labels_out, boxes_out = [], []
    for gb, gc in gt:
        if gb.shape[0] == 0:
            labels_out.append(torch.full((anchors.shape[0],), num_classes, dtype=torch.int64, device=anchors.device))
            boxes_out.append(torch.zeros_like(anchors))
            continue
        iou = pairwise_iou(gb, anchors)                      # (M,R)
        best, arg = iou.max(0)                               # first arg-max on ties
        lab = torch.full_like(arg, iou_labels[0])
        lab = torch.where(best >= lo, torch.full_like(arg, iou_labels[1]), lab)
        lab = torch.where(best >= hi, torch.full_like(arg, iou_labels[2]), lab)
        # low-quality matches: every anchor that attains some GT's best IoU becomes positive (incl. the all-zero-row quirk)
        per_gt_best = iou.max(1, keepdim=True).values
        lab = torch.where((iou == per_gt_best).any(0), torch.ones_like(lab), lab)
        cls = gc[arg].to(torch.int64)
        cls = torch.where(lab == 0, torch.full_like(cls, num_classes), cls)
        cls = torch.where(lab == -1, torch.full_like(cls, -1), cls)
        labels_out.append(cls)
        boxes_out.append(gb[arg])
    return labels_out, boxes_out


def sigmoid_focal_sum(logits, labels, num_classes, alpha, gamma):
    """fvcore sigmoid_focal_loss(reduction='sum') over the non-ignored anchors; the one-hot target is implied by the integer
    labels (num_classes = background, -1 = ignored).  logits (B,R,K), labels (B,R)."""
    valid = labels >= 0
    t = (labels[..., None] == torch.arange(num_classes, device=labels.device)).to(logits.dtype)
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return (loss * valid[..., None].to(loss.dtype)).sum()


def box2box_deltas(src, dst, weights=(1.0, 1.0, 1.0, 1.0)):
    """detectron2 Box2BoxTransform.get_deltas"""
    sw, sh = src[..., 2] - src[..., 0], src[..., 3] - src[..., 1]
    sx, sy = src[..., 0] + 0.5 * sw, src[..., 1] + 0.5 * sh
    dw, dh = dst[..., 2] - dst[..., 0], dst[..., 3] - dst[..., 1]
    dx, dy = dst[..., 0] + 0.5 * dw, dst[..., 1] + 0.5 * dh
    wx, wy, ww, wh = weights
    return torch.stack((wx * (dx - sx) / sw, wy * (dy - sy) / sh, ww * torch.log(dw / sw), wh * torch.log(dh / sh)), -1)


def box_reg_sum(deltas, labels, anchors, matched, num_classes, beta, weights=(1.0, 1.0, 1.0, 1.0)):
    """sum over positive anchors of fvcore smooth_l1_loss(pred, Box2BoxTransform deltas of the matched box, beta).
    deltas (B,R,4), labels (B,R), anchors (R,4), matched (B,R,4)."""
    pos = (labels >= 0) & (labels != num_classes)
    target = box2box_deltas(anchors[None], matched, weights)
    diff = (deltas - torch.where(pos[..., None], target, deltas.detach())).abs()
    if beta >= 1e-5:
        diff = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return (diff * pos[..., None].to(diff.dtype)).sum()


def flatten_head_output(t, K):
    """(N, A*K, H, W) -> (N, H*W*A, K)  [ref: models/customized_detectors/retinanet.py:13-22]"""
    N, _, H, W = t.shape
    return t.view(N, -1, K, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, K)


# ------------------------------------------------------------------------------------------------ DCNv2
def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1):
    """DCNv2 from its definition: out[n,o,y,x] = sum_{c,k} W[o,c,k] * mask[n,k,y,x] * bilinear(in[n,c], y*s-p+ky*d+dy_k, x*s-p+kx*d+dx_k),
    zero outside the input; offsets are (dy, dx) channel pairs per tap k = ky*kw + kx.  One bilinear grid_sample per tap + ONE GEMM."""
    N, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    ys = torch.arange(Ho, device=x.device, dtype=x.dtype) * stride - padding
    xs = torch.arange(Wo, device=x.device, dtype=x.dtype) * stride - padding
    base_y, base_x = torch.meshgrid(ys, xs, indexing="ij")
    cols = []
    for k in range(kh * kw):
        ky, kx = divmod(k, kw)
        py = base_y + ky * dilation + offset[:, 2 * k]
        px = base_x + kx * dilation + offset[:, 2 * k + 1]
        gx = 2.0 * px / max(W - 1, 1) - 1.0  # align_corners=True: [-1,1] <-> pixel centres 0..W-1
        gy = 2.0 * py / max(H - 1, 1) - 1.0
        s = F.grid_sample(x, torch.stack((gx, gy), -1), mode="bilinear", padding_mode="zeros", align_corners=True)
        cols.append(s * mask[:, k:k + 1] if mask is not None else s)
    col = torch.stack(cols, 2).reshape(N, C * kh * kw, Ho * Wo)
    out = torch.matmul(weight.reshape(O, C * kh * kw), col).reshape(N, O, Ho, Wo)
    return out if bias is None else out + bias.view(1, -1, 1, 1)


# ------------------------------------------------------------------------------------------------ FCOS side
def fcos_targets(shifts, strides, sizes_of_interest, gt, num_classes, radius):
    """FCOS ground-truth assignment [ref: models/customized_detectors/thirdparty_heads/fcos.py:177-284]:
    shifts = per-level (HW,2) centres; gt = list of (boxes (M,4), classes (M,)).  A location is a candidate of box m if it
    lies strictly inside the centre-sampling box (centre +- radius*stride, clipped to the GT box; radius <= 0: inside the GT
    box itself) [fcos.py:222-246] and its largest ltrb distance falls in the level's size range [fcos.py:248-252]; among
    candidates the smallest-area box wins [fcos.py:254-259]; no candidate -> background = num_classes.  Centerness
    sqrt(min(l,r)/max(l,r) * min(t,b)/max(t,b)) [fcos.py:268-276].
    Returns classes (B,R) int64, ltrb deltas (B,R,4), centerness (B,R)."""
    pts = torch.cat(shifts, 0)
    R = pts.shape[0]
    lo = torch.cat([pts.new_full((len(s),), float(r[0])) for s, r in zip(shifts, sizes_of_interest)])
    hi = torch.cat([pts.new_full((len(s),), float(r[1])) for s, r in zip(shifts, sizes_of_interest)])
    rad = torch.cat([pts.new_full((len(s),), float(st) * radius) for s, st in zip(shifts, strides)])
    cls_out, dl_out, ct_out = [], [], []
    for gb, gc in gt:
        M = gb.shape[0]
        if M == 0:
            cls_out.append(torch.full((R,), num_classes, dtype=torch.int64, device=pts.device))
            dl_out.append(pts.new_zeros((R, 4)))
            ct_out.append(pts.new_zeros((R,)))
            continue
        px, py = pts[None, :, 0], pts[None, :, 1]
        ltrb = torch.stack((px - gb[:, None, 0], py - gb[:, None, 1], gb[:, None, 2] - px, gb[:, None, 3] - py), -1)  # (M,R,4)
        if radius > 0:
            cx, cy = (gb[:, 0] + gb[:, 2]) / 2, (gb[:, 1] + gb[:, 3]) / 2
            x0 = torch.max(cx[:, None] - rad[None], gb[:, None, 0])
            y0 = torch.max(cy[:, None] - rad[None], gb[:, None, 1])
            x1 = torch.min(cx[:, None] + rad[None], gb[:, None, 2])
            y1 = torch.min(cy[:, None] + rad[None], gb[:, None, 3])
            inside = torch.stack((px - x0, py - y0, x1 - px, y1 - py), -1).min(-1).values > 0
        else:
            inside = ltrb.min(-1).values > 0
        far = ltrb.max(-1).values
        cand = inside & (far >= lo[None]) & (far <= hi[None])
        area = ((gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1]))[:, None].expand(M, R)
        area = torch.where(cand, area, torch.full_like(area, INF))
        best, arg = area.min(0)
        d = ltrb[arg, torch.arange(R, device=pts.device)]
        cls = torch.where(best == INF, torch.full((R,), num_classes, dtype=torch.int64, device=pts.device), gc[arg].to(torch.int64))
        lr_min, lr_max = torch.min(d[:, 0], d[:, 2]), torch.max(d[:, 0], d[:, 2])
        tb_min, tb_max = torch.min(d[:, 1], d[:, 3]), torch.max(d[:, 1], d[:, 3])
        ctr = torch.sqrt((lr_min / lr_max).clamp(min=0) * (tb_min / tb_max).clamp(min=0))
        cls_out.append(cls)
        dl_out.append(d)
        ct_out.append(ctr)
    return torch.stack(cls_out), torch.stack(dl_out), torch.stack(ct_out)


def giou_ltrb_loss(pred, target):
    """cvpods iou_loss(box_mode='ltrb', loss_type='giou', reduction='none') [call site ref: thirdparty_heads/fcos.py:151-157]."""
    eps = torch.finfo(torch.float32).eps
    p = torch.cat((-pred[..., :2], pred[..., 2:]), -1)
    t = torch.cat((-target[..., :2], target[..., 2:]), -1)
    pa = (p[..., 2] - p[..., 0]).clamp(min=0) * (p[..., 3] - p[..., 1]).clamp(min=0)
    ta = (t[..., 2] - t[..., 0]).clamp(min=0) * (t[..., 3] - t[..., 1]).clamp(min=0)
    wi = (torch.min(p[..., 2], t[..., 2]) - torch.max(p[..., 0], t[..., 0])).clamp(min=0)
    hi = (torch.min(p[..., 3], t[..., 3]) - torch.max(p[..., 1], t[..., 1])).clamp(min=0)
    inter = wi * hi
    union = ta + pa - inter
    iou = inter / union.clamp(min=eps)
    gw = torch.max(p[..., 2], t[..., 2]) - torch.min(p[..., 0], t[..., 0])
    gh = torch.max(p[..., 3], t[..., 3]) - torch.min(p[..., 1], t[..., 1])
    ac = gw * gh
    return 1 - (iou - (ac - union) / ac.clamp(min=eps))


def group_norm_relu(x, groups, weight, bias, relu=True, eps=1e-5):
    """nn.GroupNorm(groups, C)(x) [+ ReLU] of the FCOS towers [ref: thirdparty_heads/fcos.py:455-470]."""
    y = F.group_norm(x, groups, weight, bias, eps)
    return F.relu(y) if relu else y


def fcos_head_param_shapes(C=256, num_classes=80, num_convs=4, num_levels=5):
    """state_dict names / shapes of the reference FCOSHead [ref: thirdparty_heads/fcos.py:453-512]: towers of (conv3x3, GroupNorm(32),
    ReLU) at Sequential indices 3k, 3k+1, 3k+2; cls_score / bbox_pred / centerness convs; one learnable scalar per level."""
    s = {}
    for sub in ("cls_subnet", "bbox_subnet"):
        for k in range(num_convs):
            s["%s.%d.weight" % (sub, 3 * k)] = (C, C, 3, 3)
            s["%s.%d.bias" % (sub, 3 * k)] = (C,)
            s["%s.%d.weight" % (sub, 3 * k + 1)] = (C,)
            s["%s.%d.bias" % (sub, 3 * k + 1)] = (C,)
    for name, co in (("cls_score", num_classes), ("bbox_pred", 4), ("centerness", 1)):
        s[name + ".weight"] = (co, C, 3, 3)
        s[name + ".bias"] = (co,)
    for i in range(num_levels):
        s["scales.%d.scale" % i] = (1,)
    return s


def fcos_head_forward(p, features, strides, num_convs=4, centerness_on_reg=True, norm_reg_targets=True):
    """FCOSHead.forward [ref: thirdparty_heads/fcos.py:514-546] on a dict of parameters under the reference's state_dict names: per
    level, cls / bbox towers of conv3x3 -> GroupNorm(32) -> ReLU [:455-470], cls_score on the cls tower, centerness on the bbox tower
    (CENTERNESS_ON_REG) [:534-538], bbox_pred * scales[level] then ReLU(.) * stride (NORM_REG_TARGETS) or exp(.) [:540-544].
    Returns (logits, bbox_reg, centerness), lists over levels."""
    def tower(x, sub):
        for k in range(num_convs):
            x = F.conv2d(x, p["%s.%d.weight" % (sub, 3 * k)], p["%s.%d.bias" % (sub, 3 * k)], 1, 1)
            x = F.relu(F.group_norm(x, 32, p["%s.%d.weight" % (sub, 3 * k + 1)], p["%s.%d.bias" % (sub, 3 * k + 1)], 1e-5))
        return x
    logits, regs, ctrs = [], [], []
    for level, x in enumerate(features):
        c, b = tower(x, "cls_subnet"), tower(x, "bbox_subnet")
        logits.append(F.conv2d(c, p["cls_score.weight"], p["cls_score.bias"], 1, 1))
        ctrs.append(F.conv2d(b if centerness_on_reg else c, p["centerness.weight"], p["centerness.bias"], 1, 1))
        r = F.conv2d(b, p["bbox_pred.weight"], p["bbox_pred.bias"], 1, 1) * p["scales.%d.scale" % level]
        regs.append(F.relu(r) * strides[level] if norm_reg_targets else torch.exp(r))
    return logits, regs, ctrs


def fcos_shifts(level_hw, strides, offset=0.5, device="cpu"):
    """cvpods ShiftGenerator (NUM_SHIFTS = 1) [cvpods-memory, SURVEY.md appendix B]: per level the cell centres
    (x, y) = (j * s + offset * s, i * s + offset * s), row-major -> list of (H*W, 2)."""
    out = []
    for (h, w), s in zip(level_hw, strides):
        sx = torch.arange(0, w * s, s, dtype=torch.float32, device=device) + offset * s
        sy = torch.arange(0, h * s, s, dtype=torch.float32, device=device) + offset * s
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        out.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), 1))
    return out
