"""FCOSCT: cvpods-style FCOS student with the reference's surface
[ref: models/customized_detectors/fcos.py:17-77, thirdparty_heads/fcos.py:68-546].
cvpods is not available: ShiftGenerator / Shift2BoxTransform / iou_loss / focal loss follow their
public definitions (parity unpinned); GT assignment, losses and the head
follow the in-repo reference code cited per function.  Loss path: no host syncs, and the two
scalar all-reduces of the reference are packed into one 2-element all-reduce per call."""
import math

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, streams
from ..registry import CUSTOMIZED_DETECTORS_REGISTRY
from ..structures import ImageList
from . import retinanet as _rn
from .retinanet import batched_nms, build_resnet_fpn


class Scale(nn.Module):
    """[ref: thirdparty_heads/scale.py]"""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([init_value], dtype=torch.float32))

    def forward(self, x):
        return x * self.scale


class RawRegMaps(list):
    """the head's bbox_pred maps BEFORE the per-level Scale / ReLU / stride epilogue (FCOSHead.forward(raw_reg=True)): FCOSCT.losses applies the
    epilogue inside the fused loss kernel for these, and expects decoded distances for a plain list."""


class FCOSHead(nn.Module):
    """towers of conv3x3 + GN(32) + ReLU; per-level Scale; ReLU(.)*stride regression
    [ref: thirdparty_heads/fcos.py:433-546]"""

    def __init__(self, cfg):
        super().__init__()
        f = cfg.MODEL.FCOS
        C = cfg.MODEL.FPN.OUT_CHANNELS
        self.fpn_strides = list(f.FPN_STRIDES)
        self.centerness_on_reg, self.norm_reg_targets = f.CENTERNESS_ON_REG, f.NORM_REG_TARGETS
        cls, box = [], []
        for _ in range(f.NUM_CONVS):
            cls += [ops.Conv3x3(C, C), nn.GroupNorm(32, C), nn.ReLU()]
            box += [ops.Conv3x3(C, C), nn.GroupNorm(32, C), nn.ReLU()]
        self.cls_subnet, self.bbox_subnet = nn.Sequential(*cls), nn.Sequential(*box)
        self.cls_score = ops.Conv3x3(C, f.NUM_CLASSES)
        self.bbox_pred = ops.Conv3x3(C, 4)
        self.centerness = ops.Conv3x3(C, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.01)
                nn.init.constant_(m.bias, 0)
        nn.init.constant_(self.cls_score.bias, -math.log((1 - f.PRIOR_PROB) / f.PRIOR_PROB))
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.fpn_strides])
        self.fold_group_norm = True
        self.fold_group_norm_bwd = True   # False: the GroupNorm backward as its own statistics + apply passes (A/B: bench.py --no-gn-bwd-fold)

    def _forward_two_streams(self, feats, raw_reg):
        """forward() with the two towers on two streams (the shipped fold path): after the first layer, which reads the shared input, the class
        tower runs on a second stream beside the box tower -- see RetinaNetHead.forward"""
        nl = len(self.fpn_strides)
        gc, gb = self.cls_subnet[1], self.bbox_subnet[1]
        wc, wb = self.cls_subnet[0], self.bbox_subnet[0]
        (pc, c), (pb, b) = ops.conv3x3_gn(feats, [(wc.weight, wc.bias, gc.weight, gc.bias), (wb.weight, wb.bias, gb.weight, gb.bias)], gc.num_groups)

        def tower(sub, x, p):
            for i in range(3, len(sub), 3):
                w, g = sub[i], sub[i + 1]
                (p, x), = ops.conv3x3_gn(x, [(w.weight, w.bias, g.weight, g.bias)], g.num_groups, pre=p)
            return x, p
        # the CLASS tower's convolution + GroupNorm layers go to the side stream: 256 -> 256 convolutions on this library's kernels only (the gate of
        # forward(): ops.convs_on_own_kernels).  The score convolutions (C' = 80 / 4 + 1: products of the vendor library) run on the main stream, the
        # class one behind the join -- a side stream never carries a library call (round 6: the root cause of round 5's stall, lgd_amd/streams.py)
        side_mods = [self.cls_subnet[i] for i in range(3, len(self.cls_subnet)) if not isinstance(self.cls_subnet[i], nn.ReLU)]
        main, side = streams.fork(c[0].device, "head", inputs=list(c) + [pc])
        streams.join_on_grad([q for m in side_mods for q in m.parameters()], "head")
        with torch.cuda.stream(side):
            c, pc = tower(self.cls_subnet, c, pc)
        b, pb = tower(self.bbox_subnet, b, pb)
        if self.centerness_on_reg:
            regs, ctr = ops.conv3x3_shared_input(b, [(self.bbox_pred.weight, self.bbox_pred.bias), (self.centerness.weight, self.centerness.bias)], pre=pb)
        else:
            regs, ctr = self.bbox_pred.levels(b, pre=pb), None
        streams.join(main, side, outputs=list(c) + [pc])
        if self.centerness_on_reg:
            logits = self.cls_score.levels(c, pre=pc)
        else:
            logits, ctr = ops.conv3x3_shared_input(c, [(self.cls_score.weight, self.cls_score.bias), (self.centerness.weight, self.centerness.bias)], pre=pc)
        if raw_reg:
            return logits, RawRegMaps(regs), ctr
        reg = []
        for i, r in enumerate(regs):
            lvl = i % nl
            r = self.scales[lvl](r)
            reg.append(F.relu(r) * self.fpn_strides[lvl] if self.norm_reg_targets else torch.exp(r))
        return logits, reg, ctr

    def forward(self, features, raw_reg=False):
        """features: the L pyramid levels, or 2L maps (student + teacher pyramids, one pass).  raw_reg: return the bbox_pred maps without
        the per-level Scale / ReLU * stride epilogue (the training losses apply it inside their kernel, ops.fcos_reg_ctr_loss).  Every tower layer is ONE
        Winograd conv over all maps + ONE GroupNorm(32) statistics pass over all maps; the normalisation + ReLU itself runs inside the
        next convolution's input transform."""
        nl = len(self.fpn_strides)
        c = b = list(features)
        pc = pb = None   # (scale, shift) of the previous layer's GroupNorm + ReLU, applied by the next convolution's input transform
        if (_rn._HEAD_STREAMS and c[0].is_cuda and ops.side_streams_ok() and self.fold_group_norm and self.fold_group_norm_bwd
                and ops.convs_on_own_kernels(c, [[self.cls_subnet[i].weight] for i in range(3, len(self.cls_subnet), 3)])):
            return self._forward_two_streams(c, raw_reg)
        for i in range(0, len(self.cls_subnet), 3):
            gc, gb = self.cls_subnet[i + 1], self.bbox_subnet[i + 1]
            wc, wb = self.cls_subnet[i], self.bbox_subnet[i]
            if self.fold_group_norm and self.fold_group_norm_bwd:
                # conv + the GroupNorm statistics as one node: the GroupNorm's backward apply runs inside the conv's adjoint output transform
                fc, fb = (wc.weight, wc.bias, gc.weight, gc.bias), (wb.weight, wb.bias, gb.weight, gb.bias)
                if i == 0:  # the towers' first convs read the same maps: one input transform / stacked GEMM / summed input gradient
                    (pc, c), (pb, b) = ops.conv3x3_gn(c, [fc, fb], gc.num_groups)
                else:
                    (pc, c), = ops.conv3x3_gn(c, [fc], gc.num_groups, pre=pc)
                    (pb, b), = ops.conv3x3_gn(b, [fb], gb.num_groups, pre=pb)
                continue
            if i == 0:
                c, b = ops.conv3x3_shared_input(c, [(wc.weight, wc.bias), (wb.weight, wb.bias)])
            else:
                c, b = wc.levels(c, pre=pc), wb.levels(b, pre=pb)
            if self.fold_group_norm:
                # GroupNorm(32) + ReLU: statistics only; the apply pass is folded into the next convolution's load (ops.group_norm_fold)
                pc, c = ops.group_norm_fold(c, gc.num_groups, gc.weight, gc.bias)
                pb, b = ops.group_norm_fold(b, gb.num_groups, gb.weight, gb.bias)
            else:   # the reference's literal three passes per layer (A/B runs: bench.py --no-gn-fold)
                c = ops.group_norm_relu(c, gc.num_groups, gc.weight, gc.bias, relu=True)
                b = ops.group_norm_relu(b, gb.num_groups, gb.weight, gb.bias, relu=True)
        # centerness shares its input with bbox_pred (or cls_score): same sharing
        if self.centerness_on_reg:
            logits = self.cls_score.levels(c, pre=pc)
            regs, ctr = ops.conv3x3_shared_input(b, [(self.bbox_pred.weight, self.bbox_pred.bias), (self.centerness.weight, self.centerness.bias)], pre=pb)
        else:
            regs = self.bbox_pred.levels(b, pre=pb)
            logits, ctr = ops.conv3x3_shared_input(c, [(self.cls_score.weight, self.cls_score.bias), (self.centerness.weight, self.centerness.bias)], pre=pc)
        if raw_reg:
            return logits, RawRegMaps(regs), ctr
        reg = []
        for i, r in enumerate(regs):
            lvl = i % nl
            r = self.scales[lvl](r)
            reg.append(F.relu(r) * self.fpn_strides[lvl] if self.norm_reg_targets else torch.exp(r))
        return logits, reg, ctr


def _flatten_levels(per_level, K):
    """list of (N, K, H, W) -> (N, sum HW, K)"""
    return torch.cat([t.permute(0, 2, 3, 1).reshape(t.shape[0], -1, K) for t in per_level], 1)


def giou_ltrb_loss(pred, target):
    """cvpods iou_loss(box_mode='ltrb', loss_type='giou') per element (no reduction)."""
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


@CUSTOMIZED_DETECTORS_REGISTRY.register()
class FCOSCT(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        f = cfg.MODEL.FCOS
        self.num_classes = f.NUM_CLASSES
        self.in_features = list(f.IN_FEATURES)
        self.head_in_features = self.in_features
        self.fpn_strides = list(f.FPN_STRIDES)
        self.focal_loss_alpha, self.focal_loss_gamma = f.FOCAL_LOSS_ALPHA, f.FOCAL_LOSS_GAMMA
        self.iou_loss_type = f.IOU_LOSS_TYPE
        assert self.iou_loss_type == "giou"
        self.center_sampling_radius = f.CENTER_SAMPLING_RADIUS
        self.object_sizes_of_interest = [list(map(float, s)) for s in f.OBJECT_SIZES_OF_INTEREST]
        self.score_threshold, self.topk_candidates = f.SCORE_THRESH_TEST, f.TOPK_CANDIDATES_TEST
        self.nms_threshold, self.max_detections_per_image = f.NMS_THRESH_TEST, cfg.TEST.DETECTIONS_PER_IMAGE
        self.shift_offset = cfg.MODEL.SHIFT_GENERATOR.OFFSET
        # registration order = the reference's (thirdparty_heads/fcos.py:93-97 backbone, head; customized_detectors/fcos.py:23-27
        # then fpn alias, raw_backbone): named_parameters() runs FPN -> head -> bottom-up ResNet, the index order of the reference's
        # one-group-per-parameter optimizer state
        self.backbone = build_resnet_fpn(cfg)
        raw_backbone = self.backbone.bottom_up
        self.backbone.bottom_up = nn.Sequential()
        self.head = FCOSHead(cfg)
        self.fpn = self.backbone  # [ref: customized_detectors/fcos.py:23-27]
        self.raw_backbone = raw_backbone
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), persistent=False)
        self._shift_cache = {}
        self.fused_reg_loss = True   # training: GIoU + centerness losses on the raw head outputs in one kernel (False: the composed torch form)

    @property
    def device(self):
        return self.pixel_mean.device

    def preprocess_image(self, batched_inputs):
        imgs = [(x["image"].to(self.device, non_blocking=True).float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        return ImageList.from_tensors(imgs, self.backbone.size_divisibility)

    def shift_generator(self, features):
        """cvpods ShiftGenerator(NUM_SHIFTS=1, OFFSET=0.5): level-wise (HW,2) centres (x,y), row-major.
        The per-image replication of cvpods is dropped: shifts do not depend on the image."""
        key = tuple(tuple(x.shape[-2:]) for x in features) + (str(features[0].device),)
        if key not in self._shift_cache:
            out = []
            for x, s in zip(features, self.fpn_strides):
                H, W = x.shape[-2:]
                sx = torch.arange(0, W * s, s, dtype=torch.float32, device=x.device) + self.shift_offset * s
                sy = torch.arange(0, H * s, s, dtype=torch.float32, device=x.device) + self.shift_offset * s
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                out.append(torch.stack((xx.reshape(-1), yy.reshape(-1)), 1))
            self._shift_cache = {key: out}
        return self._shift_cache[key]

    def predict(self, features):
        """[ref: customized_detectors/fcos.py:29-33]"""
        box_cls, box_delta, box_center = self.head(features, raw_reg=self.training and self.fused_reg_loss)
        return self.shift_generator(features), box_cls, box_delta, box_center

    @torch.no_grad()
    def get_ground_truth(self, shifts, targets):
        """centre sampling + scale ranges + min-area tie break for the whole mini-batch in ONE HIP launch
        [ref: thirdparty_heads/fcos.py:177-284] -> (gt_classes (B,R), gt_shifts_deltas (B,R,4), gt_centerness (B,R))."""
        counts = [len(t) for t in targets]
        if sum(counts):
            gb = torch.cat([t.gt_boxes.tensor for t in targets if len(t)], 0)
            gc = torch.cat([t.gt_classes for t in targets if len(t)], 0)
        else:
            gb = gc = None
        return ops.fcos_targets(shifts, self.fpn_strides, self.object_sizes_of_interest, gb, gc, counts, self.num_classes,
                                self.center_sampling_radius)

    @staticmethod
    def reduce_counts(counts):
        """rank-mean of (num_fg, sum of foreground centerness): ONE packed all-reduce instead of the reference's two
        scalar ones (fcos.py:141,143)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(counts)
            counts = counts / dist.get_world_size()
        return counts

    def losses(self, gt_classes, gt_shifts_deltas, gt_centerness, pred_class_logits, pred_shift_deltas, pred_centerness):
        """[ref: thirdparty_heads/fcos.py:107-175] without boolean-index gathers / host syncs; the focal loss is the fused
        HIP kernel on the raw (N, K, H, W) logits."""
        fg = (gt_classes >= 0) & (gt_classes != self.num_classes)
        # which representation the regression maps are in travels WITH them (RawRegMaps: the raw bbox_pred maps of head(..., raw_reg=True);
        # a plain list: decoded distances, the reference's losses() contract) -- not re-derived from module state at a different time
        if isinstance(pred_shift_deltas, RawRegMaps):
            gt_ctr = torch.where(fg, gt_centerness, torch.zeros_like(gt_centerness))
            counts = self.reduce_counts(torch.stack((fg.sum().to(torch.float32), gt_ctr.sum())))
            num_fg, num_targets = counts[0].clamp(min=1.0), counts[1].clamp(min=1.0)
            hw = [tuple(x.shape[-2:]) for x in pred_class_logits]
            loss_cls = ops.focal_loss_sum(pred_class_logits, ops.label_planes(gt_classes, hw, 1), 1, self.num_classes,
                                          self.focal_loss_alpha, self.focal_loss_gamma, normalizer=num_fg)
            loss_box, loss_ctr = ops.fcos_reg_ctr_loss(pred_shift_deltas, pred_centerness, torch.cat([m.scale for m in self.head.scales]),
                                                       self.fpn_strides, gt_classes, gt_shifts_deltas, gt_centerness,
                                                       num_targets.reciprocal(), num_fg.reciprocal(), self.num_classes,
                                                       self.head.norm_reg_targets)
            return {"loss_cls": loss_cls, "loss_box_reg": loss_box, "loss_centerness": loss_ctr}
        deltas = _flatten_levels(pred_shift_deltas, 4)
        ctr = _flatten_levels(pred_centerness, 1).squeeze(-1)
        gt_ctr = torch.where(fg, gt_centerness, torch.zeros_like(gt_centerness))
        counts = self.reduce_counts(torch.stack((fg.sum().to(torch.float32), gt_ctr.sum())))
        num_fg, num_targets = counts[0].clamp(min=1.0), counts[1].clamp(min=1.0)
        hw = [tuple(x.shape[-2:]) for x in pred_class_logits]
        loss_cls = ops.focal_loss_sum(pred_class_logits, ops.label_planes(gt_classes, hw, 1), 1, self.num_classes,
                                      self.focal_loss_alpha, self.focal_loss_gamma, normalizer=num_fg)
        safe_t = torch.where(fg[..., None], gt_shifts_deltas, torch.ones_like(gt_shifts_deltas))
        safe_p = torch.where(fg[..., None], deltas, torch.ones_like(deltas))
        loss_box = (torch.where(fg, giou_ltrb_loss(safe_p, safe_t) * gt_ctr, torch.zeros_like(gt_ctr))).sum() / num_targets
        bce = F.binary_cross_entropy_with_logits(ctr, gt_ctr, reduction="none")
        loss_ctr = torch.where(fg, bce, torch.zeros_like(bce)).sum() / num_fg
        return {"loss_cls": loss_cls, "loss_box_reg": loss_box, "loss_centerness": loss_ctr}

    def backbone_features(self, batched_inputs, after_preprocess=None):
        """bottom-up + FPN only (see RetinaNetCT.backbone_features)."""
        images = self.preprocess_image(batched_inputs)
        if after_preprocess is not None:   # (the distillator starts the teacher's label encoder here, on its side stream, ahead of the backbone)
            after_preprocess(images)
        raw_features = self.raw_backbone(images.tensor)
        features = self.fpn(raw_features)
        features = {f: features[f] for f in self.in_features}
        gt_instances = None
        if self.training:
            assert "instances" in batched_inputs[0], "Instance annotations are missing in training!"
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        return raw_features, features, images, gt_instances

    def predict_pair(self, feats_a, feats_b):
        """predict() on two pyramids with ONE head pass (see RetinaNetCT.predict_pair):
        shifts, (cls_a, delta_a, center_a), (cls_b, delta_b, center_b)."""
        L = len(feats_a)
        cls, reg, ctr = self.head(list(feats_a) + list(feats_b), raw_reg=self.training and self.fused_reg_loss)
        raw = isinstance(reg, RawRegMaps)
        ra, rb = (RawRegMaps(reg[:L]), RawRegMaps(reg[L:])) if raw else (reg[:L], reg[L:])
        return self.shift_generator(feats_a), (cls[:L], ra, ctr[:L]), (cls[L:], rb, ctr[L:])

    def forward(self, batched_inputs):
        """[ref: customized_detectors/fcos.py:36-63]"""
        raw_features, features, images, gt_instances = self.backbone_features(batched_inputs)
        shifts, box_cls, box_delta, box_center = self.predict([features[f] for f in self.in_features])
        if self.training:
            gt = self.get_ground_truth(shifts, gt_instances)
            losses = self.losses(*gt, box_cls, box_delta, box_center)
            return losses, raw_features, features, images, gt
        results = self.inference(box_cls, box_delta, box_center, shifts, images)
        return self.get_processed_results(results, batched_inputs, images), raw_features, features, images

    @torch.no_grad()
    def inference(self, box_cls, box_delta, box_center, shifts, images):
        """[ref: thirdparty_heads/fcos.py:286-394] score = sqrt(cls * centerness), top-k per level, NMS."""
        from ..structures import Boxes, Instances
        results = []
        for i, size in enumerate(images.image_sizes):
            bs, ss, cs = [], [], []
            for cl, dl, ce, sh in zip(box_cls, box_delta, box_center, shifts):
                K = self.num_classes
                sc = (cl[i].permute(1, 2, 0).reshape(-1, K).sigmoid() * ce[i].permute(1, 2, 0).reshape(-1, 1).sigmoid()).flatten()
                k = min(self.topk_candidates, sc.numel())
                sc, idx = sc.sort(descending=True)
                sc, idx = sc[:k], idx[:k]
                keep = sc > self.score_threshold
                sc, idx = sc[keep], idx[keep]
                pos, c = idx // K, idx % K
                d = dl[i].permute(1, 2, 0).reshape(-1, 4)[pos]
                p = sh[pos]
                bs.append(torch.cat((p - d[:, :2], p + d[:, 2:]), 1))
                ss.append(torch.sqrt(sc))
                cs.append(c)
            b, s, c = torch.cat(bs), torch.cat(ss), torch.cat(cs)
            keep = batched_nms(b, s, c, self.nms_threshold)[:self.max_detections_per_image]
            results.append(Instances(size, pred_boxes=Boxes(b[keep]), scores=s[keep], pred_classes=c[keep]))
        return results

    def get_processed_results(self, results, batched_inputs, images):
        from ..structures import Boxes, Instances
        out = []
        for r, inp, size in zip(results, batched_inputs, images.image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            sx, sy = w / size[1], h / size[0]
            b = r.pred_boxes.tensor * torch.tensor([sx, sy, sx, sy], device=r.pred_boxes.tensor.device)
            out.append({"instances": Instances((h, w), pred_boxes=Boxes(b), scores=r.scores, pred_classes=r.pred_classes)})
        return out
