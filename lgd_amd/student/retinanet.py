"""RetinaNetCT: the student detector the reference builds by subclassing detectron2's RetinaNet
[ref: models/customized_detectors/retinanet.py:24-95].  detectron2 is not available here, so its
v0.3 RetinaNet is restated from the public definition ([d2-memory]) with the
reference's surface: `forward` returns (losses, raw_features, features, images, (gt_labels, gt_boxes)),
`predict(features)`, `losses(...)`, `inference(...)`, attributes `fpn/backbone/raw_backbone/head/
head_in_features`.  The loss path is free of host syncs (d2 calls `.item()` on the positive count)."""
import math
import os

import torch
import torch.nn as nn

from .. import ops, streams
from ..registry import CUSTOMIZED_DETECTORS_REGISTRY
from ..structures import ImageList
from .fpn import FPN, LastLevelP6P7
from .resnet import ResNet


def permute_to_N_HWA_K(t, K):
    """(N, A*K, H, W) -> (N, H*W*A, K)  [ref: retinanet.py:13-22]"""
    N, _, H, W = t.shape
    return t.view(N, -1, K, H, W).permute(0, 3, 4, 1, 2).reshape(N, -1, K)


# The class tower of the head on the step's side stream beside the box tower (LGD_HEAD_STREAMS=0: one stream).  Round 6
# (profiles/r06_hw_queues_and_forks.txt): with a stream PER fork this one gained nothing under HIP's default 4 hardware queues (its stream shared the
# main stream's queue) and cost 19-22 ms per step under more (71 ms against 52: cross-queue waits stalling, not the kernels); on the ONE side stream
# all forks now share (streams._ONE_SIDE) it is worth 0.5-0.6 ms at config 2, 0.8-1.1 at config 3, the same under 4 and 8 queues.
_HEAD_STREAMS = os.environ.get("LGD_HEAD_STREAMS", "1") != "0"


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
        # the first convs of the two towers read the same maps: one input transform / stacked GEMM / summed input gradient
        c0, b0 = self.cls_subnet[0], self.bbox_subnet[0]
        c, b = ops.conv3x3_shared_input(list(features), [(c0.weight, c0.bias), (b0.weight, b0.bias)], relu=True)
        # the rest of each tower incl. its score conv is a chain whose intermediate maps nobody else reads: the backward crosses every
        # conv -> ReLU -> conv link in the frequency domain (no gradient map written / re-read)
        cl = [self.cls_subnet[i] for i in range(2, len(self.cls_subnet), 2)] + [self.cls_score]
        bl = [self.bbox_subnet[i] for i in range(2, len(self.bbox_subnet), 2)] + [self.bbox_pred]
        relus = [True] * (len(cl) - 1) + [False]
        if _HEAD_STREAMS and c[0].is_cuda and ops.side_streams_ok() and ops.convs_on_own_kernels(c, [[m.weight] for m in cl]):
            # the two chains do not depend on each other: the CLASS tower runs on a second stream beside the box tower (tails and small launches of
            # one under the other's kernels; round 5, same call: 52.32 / 52.11 -> 51.60 / 51.53 ms, losses identical; round 6: see _HEAD_STREAMS above).
            # Which one goes aside is not a matter of taste: a side stream carries this library's kernels only (ops.convs_on_own_kernels) -- bbox_pred's
            # C' = 36 products are calls of the vendor library and stay on the main stream (round 6: the root cause of round 5's stall, streams.py)
            main, side = streams.fork(c[0].device, "head", inputs=c)
            streams.join_on_grad([q for m in cl for q in (m.weight, m.bias)], "head")
            with torch.cuda.stream(side):
                cout = ops.conv3x3_chain(c, [(m.weight, m.bias) for m in cl], relus)
            bout = ops.conv3x3_chain(b, [(m.weight, m.bias) for m in bl], relus)
            streams.join(main, side, outputs=cout)
            return cout, bout
        return (ops.conv3x3_chain(c, [(m.weight, m.bias) for m in cl], relus),
                ops.conv3x3_chain(b, [(m.weight, m.bias) for m in bl], relus))


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


def apply_deltas(deltas, boxes, weights=(1.0, 1.0, 1.0, 1.0), clamp=math.log(1000.0 / 16)):
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    dx, dy = deltas[:, 0] / weights[0], deltas[:, 1] / weights[1]
    dw, dh = (deltas[:, 2] / weights[2]).clamp(max=clamp), (deltas[:, 3] / weights[3]).clamp(max=clamp)
    pcx, pcy, pw, ph = dx * w + cx, dy * h + cy, torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), 1)


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
        # separate fpn and backbone exactly like the reference (retinanet.py:29-34).  Registration ORDER is the reference's too:
        # detectron2's RetinaNet registers `backbone` (the FPN) and `head`; RetinaNetCT then adds the alias `fpn` and, last,
        # `raw_backbone` -- so named_parameters() runs FPN -> head -> bottom-up ResNet, the order in which the reference's
        # optimizers index their one-group-per-parameter state (utils/build.py:494-512); checkpoints map by that index.
        self.backbone = build_resnet_fpn(cfg)
        raw_backbone = self.backbone.bottom_up
        self.backbone.bottom_up = nn.Sequential()
        strides = [8, 16, 32, 64, 128][:len(self.head_in_features)]
        self.anchor_generator = AnchorGenerator(cfg.MODEL.ANCHOR_GENERATOR.SIZES, cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                                                strides, cfg.MODEL.ANCHOR_GENERATOR.OFFSET)   # buffers only: no place in the order
        self.head = RetinaNetHead(cfg.MODEL.FPN.OUT_CHANNELS, self.num_classes, self.anchor_generator.num_cell_anchors[0],
                                  rc.NUM_CONVS, rc.PRIOR_PROB)
        self.fpn = self.backbone
        self.raw_backbone = raw_backbone
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
        """detectron2 Matcher (thresholds [0.4,0.5] -> labels [0,-1,1], low-quality matches allowed; background ->
        num_classes, ignore -> -1) for the whole mini-batch in two HIP launches, no IoU matrix, tensors stay on the device
        [call site ref: retinanet.py:66-67].  The int32 label planes the fused loss kernels read are built here, once per
        iteration, and shared by the student and the teacher `losses()` calls."""
        if tuple(self.iou_labels) != (0, -1, 1):
            raise ValueError("IOU_LABELS %s not supported (shipped configs use [0, -1, 1])" % (self.iou_labels,))
        A = torch.cat(anchors, 0)
        lo, hi = self.iou_thresholds
        counts = [len(inst) for inst in gt_instances]
        if sum(counts):
            gb = torch.cat([inst.gt_boxes.tensor for inst in gt_instances if len(inst)], 0)
            gc = torch.cat([inst.gt_classes for inst in gt_instances if len(inst)], 0)
        else:
            gb = gc = None
        labels, matched = ops.anchor_match(A, gb, gc, counts, lo, hi, self.num_classes, True)
        gt_labels, gt_boxes = list(labels.unbind(0)), list(matched.unbind(0))
        self._remember_targets(gt_labels, labels, matched)
        return gt_labels, gt_boxes

    def _remember_targets(self, gt_labels, labels, matched):
        """this iteration's stacked targets.  The cache holds a strong reference to the first label tensor and is matched
        by identity (`is`), so a recycled object address can never alias a later batch's targets."""
        self._target_cache = {"src": gt_labels[0], "n": len(gt_labels), "labels": labels, "matched": matched,
                              "hw": None, "planes": None}

    def _targets_for(self, gt_labels, gt_boxes, raw):
        """stacked labels / matched boxes / int32 label planes for these targets (built once, shared by both losses() calls)."""
        c = getattr(self, "_target_cache", None)
        if c is None or c["src"] is not gt_labels[0] or c["n"] != len(gt_labels):
            self._remember_targets(gt_labels, torch.stack(gt_labels), torch.stack(gt_boxes))
            c = self._target_cache
        hw = [tuple(x.shape[-2:]) for x in raw]
        if c["planes"] is None or c["hw"] != hw:
            c["planes"], c["hw"] = ops.label_planes(c["labels"], hw, raw[0].shape[1] // self.num_classes), hw
        return c

    def losses(self, anchors, pred_logits, gt_labels, pred_anchor_deltas, gt_boxes):
        """[d2-memory RetinaNet.losses]; the EMA normaliser advances on EVERY call -- the distillator
        calls this twice per iteration (student and teacher features, distillator.py:110).
        Fused HIP kernels on the head's raw NCHW outputs: no permute copies, no one-hot, no target deltas for all anchors,
        no host sync (d2 calls `.item()` on the positive count)."""
        raw, raw_d = getattr(pred_logits, "raw", None), getattr(pred_anchor_deltas, "raw", None)
        if raw is None or raw_d is None:
            raise TypeError("losses() expects the HeadOutputs returned by predict()")
        c = self._targets_for(gt_labels, gt_boxes, raw)
        labels = c["labels"]
        A = torch.cat(anchors, 0)
        num_pos = ((labels >= 0) & (labels != self.num_classes)).sum().to(torch.float32)
        self.loss_normalizer = (self.loss_normalizer_momentum * self.loss_normalizer
                                + (1 - self.loss_normalizer_momentum) * num_pos.clamp(min=1.0)).detach()
        nA = raw[0].shape[1] // self.num_classes
        loss_cls = ops.focal_loss_sum(raw, c["planes"], nA, self.num_classes, self.focal_loss_alpha, self.focal_loss_gamma,
                                      normalizer=self.loss_normalizer)   # = sum / normalizer, gradient written in the same pass
        loss_box = ops.box_reg_loss_sum(raw_d, c["planes"], A, c["matched"], nA, self.num_classes,
                                        self.smooth_l1_beta, self.bbox_reg_weights)
        return {"loss_cls": loss_cls, "loss_box_reg": loss_box / self.loss_normalizer}

    def backbone_features(self, batched_inputs, after_preprocess=None):
        """bottom-up + FPN only (no head pass): (raw_features, features dict, images, gt_instances | None).  The
        distillator defers the student's head pass until the teacher features exist and runs the head ONCE over both
        (ref: distillator.py:107-112 re-runs student.predict on the teacher features)."""
        images = self.preprocess_image(batched_inputs)
        if after_preprocess is not None:   # (the distillator starts the teacher's label encoder here, on its side stream, ahead of the backbone)
            after_preprocess(images)
        raw_features = self.raw_backbone(images.tensor)
        features = self.fpn(raw_features)
        features = {f: features[f] for f in self.head_in_features}
        gt_instances = None
        if self.training:
            assert "instances" in batched_inputs[0], "Instance annotations are missing in training!"
            gt_instances = [x["instances"].to(self.device) for x in batched_inputs]
        return raw_features, features, images, gt_instances

    def predict_pair(self, feats_a, feats_b):
        """predict() on two pyramids with ONE head pass: the 2 x L maps go through every tower conv as one Winograd problem
        (one filter transform, one GEMM per frequency over both pyramids' tiles, one weight-gradient GEMM), no concat copy.
        Returns anchors, (logits_a, deltas_a), (logits_b, deltas_b)."""
        L = len(feats_a)
        anchors = self.anchor_generator(feats_a)
        logits, deltas = self.head(list(feats_a) + list(feats_b))
        K = self.num_classes
        return (anchors, (HeadOutputs(logits[:L], K), HeadOutputs(deltas[:L], 4)),
                (HeadOutputs(logits[L:], K), HeadOutputs(deltas[L:], 4)))

    def forward(self, batched_inputs):
        """[ref: retinanet.py:45-81]"""
        raw_features, features, images, gt_instances = self.backbone_features(batched_inputs)
        anchors, pred_logits, pred_anchor_deltas = self.predict([features[f] for f in self.head_in_features])
        if self.training:
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
