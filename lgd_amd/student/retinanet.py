"""RetinaNetCT: the student detector the reference builds by subclassing detectron2's RetinaNet
[ref: models/customized_detectors/retinanet.py:24-95].  detectron2 is not available here, so its
v0.3 RetinaNet is restated from the public definition ([d2-memory]) with the
reference's surface: `forward` returns (losses, raw_features, features, images, (gt_labels, gt_boxes)),
`predict(features)`, `losses(...)`, `inference(...)`, attributes `fpn/backbone/raw_backbone/head/
head_in_features`.  The loss path is free of host syncs (d2 calls `.item()` on the positive count)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..registry import CUSTOMIZED_DETECTORS_REGISTRY
from ..structures import ImageList
from .fpn import FPN, LastLevelP6P7
from .resnet import ResNet


def permute_to_N_HWA_K(t, K):
    """(N, A*K, H, W) -> (N, H*W*A, K)  [ref: retinanet.py:13-22]"""
    N, _, H, W = t.shape
    return t.view(N, -1, K, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, K)


class RetinaNetHead(nn.Module):
    def __init__(self, cin, num_classes, num_anchors, num_convs=4, prior_prob=0.01):
        super().__init__()
        cls, box = [], []
        for _ in range(num_convs):
            cls += [ops.Conv3x3(cin, cin), nn.ReLU()]
            box += [ops.Conv3x3(cin, cin), nn.ReLU()]
        self.cls_subnet = nn.Sequential(*cls)
        self.bbox_subnet = nn.Sequential(*box)
        self.cls_score = ops.Conv3x3(cin, num_anchors * num_classes)
        self.bbox_pred = ops.Conv3x3(cin, num_anchors * 4)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.01)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_score.bias, -math.log((1 - prior_prob) / prior_prob))

    def forward(self, features):
        # the towers share their filters across levels: every conv is ONE pass over the concatenated pyramid, ReLU fused
        c = b = list(features)
        for i in range(0, len(self.cls_subnet), 2):
            c = self.cls_subnet[i].levels(c, relu=True)
            b = self.bbox_subnet[i].levels(b, relu=True)
        return self.cls_score.levels(c), self.bbox_pred.levels(b)


class AnchorGenerator(nn.Module):
    """cell anchors ordered size-major then ratio; grid ordered (y, x, a); offset 0."""

    def __init__(self, sizes, ratios, strides, offset=0.0):
        super().__init__()
        self.strides, self.offset = list(strides), offset
        if len(ratios) == 1:
            ratios = list(ratios) * len(sizes)
        self.num_cell_anchors = [len(s) * len(r) for s, r in zip(sizes, ratios)]
        for i, (ss, rr) in enumerate(zip(sizes, ratios)):
            cell = []
            for s in ss:
                area = float(s) ** 2
                for r in rr:
                    w = math.sqrt(area / r)
                    h = r * w
                    cell.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
            self.register_buffer("cell_%d" % i, torch.tensor(cell, dtype=torch.float32), persistent=False)
        self._cache = {}

    def forward(self, features):
        key = tuple(tuple(f.shape[-2:]) for f in features) + (str(features[0].device),)
        if key not in self._cache:
            out = []
            for i, f in enumerate(features):
                H, W = f.shape[-2:]
                s = self.strides[i]
                sx = torch.arange(self.offset * s, W * s, s, dtype=torch.float32, device=f.device)
                sy = torch.arange(self.offset * s, H * s, s, dtype=torch.float32, device=f.device)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), 1)
                cell = getattr(self, "cell_%d" % i).to(f.device)
                out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
            self._cache = {key: out}
        return self._cache[key]


def pairwise_iou(a, b):
    """a (M,4), b (R,4) xyxy -> (M,R)"""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[None, :, 2:]) - torch.max(a[:, None, :2], b[None, :, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return torch.where(inter > 0, inter / (area_a[:, None] + area_b[None, :] - inter), torch.zeros_like(inter))


def box_deltas(src, dst, weights=(1.0, 1.0, 1.0, 1.0)):
    """Box2BoxTransform.get_deltas"""
    sw, sh = src[..., 2] - src[..., 0], src[..., 3] - src[..., 1]
    sx, sy = src[..., 0] + 0.5 * sw, src[..., 1] + 0.5 * sh
    dw, dh = dst[..., 2] - dst[..., 0], dst[..., 3] - dst[..., 1]
    dx, dy = dst[..., 0] + 0.5 * dw, dst[..., 1] + 0.5 * dh
    wx, wy, ww, wh = weights
    return torch.stack((wx * (dx - sx) / sw, wy * (dy - sy) / sh, ww * torch.log(dw / sw), wh * torch.log(dh / sh)), -1)


def apply_deltas(deltas, boxes, weights=(1.0, 1.0, 1.0, 1.0), clamp=math.log(1000.0 / 16)):
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    dx, dy = deltas[:, 0] / weights[0], deltas[:, 1] / weights[1]
    dw, dh = (deltas[:, 2] / weights[2]).clamp(max=clamp), (deltas[:, 3] / weights[3]).clamp(max=clamp)
    pcx, pcy, pw, ph = dx * w + cx, dy * h + cy, torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), 1)


def sigmoid_focal_sum(logits, labels, valid, num_classes, alpha, gamma):
    """sum over valid anchors and classes of fvcore's sigmoid focal loss, the one-hot target given
    implicitly by integer labels (num_classes = background).  logits (B,R,K), labels (B,R)."""
    t = (labels[..., None] == torch.arange(num_classes, device=labels.device)).to(logits.dtype)
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return (loss * valid[..., None].to(loss.dtype)).sum()


class HeadOutputs(list):
    """What `predict()` hands to `losses()` / `inference()`: behaves as the reference's list of per-level
    (N, HWA, K) tensors (materialised lazily, on first element access) and keeps the head's raw (N, A*K, H, W)
    outputs in `.raw`, which the fused HIP loss kernels read in place (no 516 MB permute copy per pass)."""

    def __init__(self, raw, K):
        super().__init__([None] * len(raw))
        self.raw, self.K = list(raw), K

    def _get(self, i):
        v = list.__getitem__(self, i)
        if v is None:
            v = permute_to_N_HWA_K(self.raw[i], self.K)
            list.__setitem__(self, i, v)
        return v

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self)))]
        return self._get(i if i >= 0 else len(self) + i)

    def __iter__(self):
        return (self._get(i) for i in range(len(self)))


def batched_nms(boxes, scores, idxs, thresh):
    """plain greedy class-aware NMS (inference only, not on the training hot path)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    off = idxs.to(boxes) * (boxes.max() + 1)
    b = boxes + off[:, None]
    order = scores.argsort(descending=True)
    b = b[order]
    iou = pairwise_iou(b, b)
    keep = []
    suppressed = torch.zeros(len(b), dtype=torch.bool, device=b.device)
    for i in range(len(b)):
        if suppressed[i]:
            continue
        keep.append(i)
        suppressed |= iou[i] > thresh
    return order[torch.tensor(keep, dtype=torch.int64, device=b.device)]


def build_resnet_fpn(cfg, top_in="res5"):
    r = cfg.MODEL.RESNETS
    bottom_up = ResNet(r.DEPTH, r.OUT_FEATURES, cfg.MODEL.BACKBONE.FREEZE_AT, r.STRIDE_IN_1X1, r.NUM_GROUPS,
                       r.WIDTH_PER_GROUP, r.RES2_OUT_CHANNELS, r.STEM_OUT_CHANNELS, tuple(r.DEFORM_ON_PER_STAGE),
                       r.DEFORM_MODULATED)
    feats = cfg.MODEL.FPN.IN_FEATURES
    cout = cfg.MODEL.FPN.OUT_CHANNELS
    top_c = bottom_up.out_channels[top_in] if top_in.startswith("res") else cout
    return FPN(bottom_up, feats, [bottom_up.out_channels[f] for f in feats], cout, LastLevelP6P7(top_c, cout, top_in))


@CUSTOMIZED_DETECTORS_REGISTRY.register()
class RetinaNetCT(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        rc = cfg.MODEL.RETINANET
        self.num_classes = rc.NUM_CLASSES
        self.head_in_features = list(rc.IN_FEATURES)
        self.in_features = self.head_in_features
        self.focal_loss_alpha, self.focal_loss_gamma = rc.FOCAL_LOSS_ALPHA, rc.FOCAL_LOSS_GAMMA
        self.smooth_l1_beta = rc.SMOOTH_L1_LOSS_BETA
        self.iou_thresholds, self.iou_labels = list(rc.IOU_THRESHOLDS), list(rc.IOU_LABELS)
        self.bbox_reg_weights = tuple(rc.BBOX_REG_WEIGHTS)
        self.test_score_thresh, self.test_topk = rc.SCORE_THRESH_TEST, rc.TOPK_CANDIDATES_TEST
        self.test_nms_thresh, self.max_detections = rc.NMS_THRESH_TEST, cfg.TEST.DETECTIONS_PER_IMAGE
        self.vis_period = cfg.VIS_PERIOD
        # separate fpn and backbone exactly like the reference (retinanet.py:29-34)
        self.backbone = build_resnet_fpn(cfg)
        self.fpn = self.backbone
        self.raw_backbone = self.fpn.bottom_up
        self.fpn.bottom_up = nn.Sequential()
        strides = [8, 16, 32, 64, 128][:len(self.head_in_features)]
        self.anchor_generator = AnchorGenerator(cfg.MODEL.ANCHOR_GENERATOR.SIZES, cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                                                strides, cfg.MODEL.ANCHOR_GENERATOR.OFFSET)
        self.head = RetinaNetHead(cfg.MODEL.FPN.OUT_CHANNELS, self.num_classes, self.anchor_generator.num_cell_anchors[0],
                                  rc.NUM_CONVS, rc.PRIOR_PROB)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), persistent=False)
        # EMA of the positive-anchor count; a device tensor so that no step needs a host sync
        self.register_buffer("loss_normalizer", torch.tensor(100.0), persistent=False)
        self.loss_normalizer_momentum = 0.9

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        imgs = [(x["image"].to(self.device, non_blocking=True).float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        return ImageList.from_tensors(imgs, self.backbone.size_divisibility)

    def predict(self, features):
        """[ref: retinanet.py:36-43]"""
        anchors = self.anchor_generator(features)
        logits, deltas = self.head(features)
        return anchors, HeadOutputs(logits, self.num_classes), HeadOutputs(deltas, 4)

    @torch.no_grad()
    def label_anchors(self, anchors, gt_instances):
        """IoU matcher, thresholds [0.4,0.5] -> labels [0,-1,1], low-quality matches allowed;
        background -> num_classes, ignore -> -1.  All images in one batched IoU when they have
        the same number of boxes is not assumed: the loop is over images, tensors stay on device."""
        A = torch.cat(anchors, 0)
        lo, hi = self.iou_thresholds
        if A.is_cuda and tuple(self.iou_labels) == (0, -1, 1):  # the whole mini-batch in two HIP launches, no IoU matrix
            counts = [len(inst) for inst in gt_instances]
            if sum(counts):
                gb = torch.cat([inst.gt_boxes.tensor for inst in gt_instances if len(inst)], 0)
                gc = torch.cat([inst.gt_classes for inst in gt_instances if len(inst)], 0)
            else:
                gb = gc = None
            labels, matched = ops.anchor_match(A, gb, gc, counts, lo, hi, self.num_classes, True)
            return list(labels.unbind(0)), list(matched.unbind(0))
        return self._label_anchors_torch(A, gt_instances)

    def _label_anchors_torch(self, A, gt_instances):
        """the same matching as elementwise torch ops per image (CPU tests; the HIP path is checked against it)."""
        gt_labels, gt_boxes = [], []
        lo, hi = self.iou_thresholds
        for inst in gt_instances:
            if len(inst) == 0:
                gt_labels.append(torch.full((A.shape[0],), self.num_classes, dtype=torch.int64, device=A.device))
                gt_boxes.append(torch.zeros_like(A))
                continue
            gb = inst.gt_boxes.tensor
            iou = pairwise_iou(gb, A)
            vals, idx = iou.max(0)
            l0, l1, l2 = (torch.full_like(idx, v) for v in self.iou_labels)
            lab = torch.where(vals >= hi, l2, torch.where(vals >= lo, l1, l0))
            best_per_gt = iou.max(1, keepdim=True)[0]
            lab = torch.where((iou == best_per_gt).any(0), torch.ones_like(lab), lab)  # allow_low_quality_matches
            cls = inst.gt_classes[idx].to(torch.int64)
            cls = torch.where(lab == 0, torch.full_like(cls, self.num_classes), cls)
            cls = torch.where(lab == -1, torch.full_like(cls, -1), cls)
            gt_labels.append(cls)
            gt_boxes.append(gb[idx])
        return gt_labels, gt_boxes

    def losses(self, anchors, pred_logits, gt_labels, pred_anchor_deltas, gt_boxes):
        """[d2-memory RetinaNet.losses]; the EMA normaliser advances on EVERY call -- the distillator
        calls this twice per iteration (student and teacher features, distillator.py:110)."""
        labels = torch.stack(gt_labels)  # (B,R)
        A = torch.cat(anchors, 0)
        valid = labels >= 0
        pos = valid & (labels != self.num_classes)
        num_pos = pos.sum().to(torch.float32)
        self.loss_normalizer = (self.loss_normalizer_momentum * self.loss_normalizer
                                + (1 - self.loss_normalizer_momentum) * num_pos.clamp(min=1.0)).detach()
        raw, raw_d = getattr(pred_logits, "raw", None), getattr(pred_anchor_deltas, "raw", None)
        if raw is not None and raw_d is not None and raw[0].is_cuda:
            # fused HIP kernels on the head's NCHW outputs: no permute copies, no one-hot, no target deltas for all anchors
            from .. import ops
            nA = raw[0].shape[1] // self.num_classes
            key = (id(gt_labels[0]), len(gt_labels))
            if getattr(self, "_label_plane_key", None) != key:  # student and teacher passes share the same targets
                hw = [tuple(x.shape[-2:]) for x in raw]
                self._label_planes = ops.label_planes(labels, hw, nA)
                self._matched = torch.stack(gt_boxes)
                self._label_plane_key = key
            loss_cls = ops.focal_loss_sum(raw, self._label_planes, nA, self.num_classes, self.focal_loss_alpha, self.focal_loss_gamma)
            loss_box = ops.box_reg_loss_sum(raw_d, self._label_planes, A, self._matched, nA, self.num_classes,
                                            self.smooth_l1_beta, self.bbox_reg_weights)
        else:
            gt_deltas = box_deltas(A[None], torch.stack(gt_boxes), self.bbox_reg_weights)
            deltas = torch.cat(list(pred_anchor_deltas), 1)
            loss_cls = sigmoid_focal_sum(torch.cat(list(pred_logits), 1), labels, valid, self.num_classes,
                                         self.focal_loss_alpha, self.focal_loss_gamma)
            diff = (deltas - torch.where(pos[..., None], gt_deltas, deltas.detach())).abs()
            if self.smooth_l1_beta >= 1e-5:
                b = self.smooth_l1_beta
                diff = torch.where(diff < b, 0.5 * diff * diff / b, diff - 0.5 * b)
            loss_box = (diff * pos[..., None].to(diff.dtype)).sum()
        return {"loss_cls": loss_cls / self.loss_normalizer, "loss_box_reg": loss_box / self.loss_normalizer}

    def forward(self, batched_inputs):
        """[ref: retinanet.py:45-81]"""
        images = self.preprocess_image(batched_inputs)
        raw_features = self.raw_backbone(images.tensor)
        features = self.fpn(raw_features)
        features = [features[f] for f in self.head_in_features]
        anchors, pred_logits, pred_anchor_deltas = self.predict(features)
        features = dict(zip(self.head_in_features, features))
        if self.training:
            assert "instances" in batched_inputs[0], "Instance annotations are missing in training!"
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
            gt_labels, gt_boxes = self.label_anchors(anchors, gt_instances)
            losses = self.losses(anchors, pred_logits, gt_labels, pred_anchor_deltas, gt_boxes)
            return losses, raw_features, features, images, (gt_labels, gt_boxes)
        results = self.inference(anchors, pred_logits, pred_anchor_deltas, images.image_sizes)
        return self.get_processed_results(results, batched_inputs, images), raw_features, features, images

    @torch.no_grad()
    def inference(self, anchors, pred_logits, pred_anchor_deltas, image_sizes):
        from ..structures import Boxes, Instances
        results = []
        for i, size in enumerate(image_sizes):
            boxes_all, scores_all, cls_all = [], [], []
            for lg, dl, an in zip(pred_logits, pred_anchor_deltas, anchors):
                sc = lg[i].flatten().sigmoid()
                k = min(self.test_topk, sc.numel())
                sc, idx = sc.sort(descending=True)
                sc, idx = sc[:k], idx[:k]
                keep = sc > self.test_score_thresh
                sc, idx = sc[keep], idx[keep]
                a_idx, c_idx = idx // self.num_classes, idx % self.num_classes
                boxes_all.append(apply_deltas(dl[i][a_idx], an[a_idx], self.bbox_reg_weights))
                scores_all.append(sc)
                cls_all.append(c_idx)
            b, s, c = torch.cat(boxes_all), torch.cat(scores_all), torch.cat(cls_all)
            keep = batched_nms(b, s, c, self.test_nms_thresh)[:self.max_detections]
            bb = b[keep]
            bb = torch.stack((bb[:, 0].clamp(0, size[1]), bb[:, 1].clamp(0, size[0]),
                              bb[:, 2].clamp(0, size[1]), bb[:, 3].clamp(0, size[0])), 1)
            results.append(Instances(size, pred_boxes=Boxes(bb), scores=s[keep], pred_classes=c[keep]))
        return results

    def get_processed_results(self, results, batched_inputs, images):
        """[ref: retinanet.py:84-95]; detector_postprocess = rescale boxes to the requested output size."""
        from ..structures import Boxes, Instances
        out = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            sx, sy = w / size[1], h / size[0]
            b = r.pred_boxes.tensor * torch.tensor([sx, sy, sx, sy], device=r.pred_boxes.tensor.device)
            out.append({"instances": Instances((h, w), pred_boxes=Boxes(b), scores=r.scores, pred_classes=r.pred_classes)})
        return out
