"""Meta-architectures `DistillatorRetinaNet` / `DistillatorFCOS` (the drop-in API boundary)
[ref: models/distillator.py:23-114 and 201-297].  Training forward = student -> dynamic teacher
-> student head re-run on the teacher features (losses suffixed '.tea') -> distillation loss."""
from .base_distillator import BaseDistillator
from .registry import META_ARCH_REGISTRY


class _Distillator(BaseDistillator):
    """Training forward [ref: distillator.py:39-68 / 216-243]: student -> dynamic teacher -> student head on the teacher
    features -> distillation loss.  The student's own head pass does not feed the teacher, so it is deferred until the
    teacher features exist and the head then runs ONCE over both pyramids (`student.predict_pair`);
    `forward_student` / `forward_teacher` keep the reference's two-pass signatures for callers that use them directly."""

    def __init__(self, cfg=None):
        super().__init__(cfg)
        self.flag_seg_map = cfg.MODEL.DISTILLATOR.LABEL_ENCODER.LOAD_LABELMAP
        self.fused_head_pass = True  # False: the reference's literal two head passes (tests compare the two)

    def forward_student(self, batched_inputs, **kwargs):
        return self.student(batched_inputs)

    def forward(self, batched_inputs, **kwargs):
        if self.training and self.fused_head_pass:
            s = self.student
            r_features, features, images, gt_instances = s.backbone_features(
                batched_inputs, after_preprocess=lambda images: self.teacher.encode_ahead(batched_inputs, images))
            adapted = self.adapt_ahead(features)   # (on its side stream, beside the teacher)
            features_tea, inst_labels, geom = self.teacher((batched_inputs, images, r_features, features))
            losses, losses_tea = self._pair_losses([features[f] for f in s.head_in_features],
                                                   [features_tea[f] for f in s.head_in_features], gt_instances)
            losses.update({k + ".tea": v for k, v in losses_tea.items()})
            losses.update(self.distill_loss({"stu": features, "tea": features_tea}, images, batched_inputs, geom, inst_labels, adapted=adapted))
            return losses
        if self.training:
            losses, r_features, features, images, gt = self.forward_student(batched_inputs)
            losses_tea, _, features_tea, geom, inst_labels = self.forward_teacher(
                batched_inputs, images=images, r_features=r_features, features=features, **{self._gt_kw: gt})
            losses_distill = self.distill_loss({"stu": features, "tea": features_tea}, images, batched_inputs, geom, inst_labels)
            losses.update(losses_tea)
            losses.update(losses_distill)
            return losses
        processed, r_features, features, images = self.forward_student(batched_inputs)
        feats = [features[f] for f in self.student.head_in_features]
        if kwargs.get("eval_teacher", False):  # upper-bound probe: needs GT at test time
            features_tea, _, _ = self.teacher((batched_inputs, images, r_features, features))
            feats = [features_tea[f] for f in self.student.head_in_features]
        return self._infer(feats, batched_inputs, images)


@META_ARCH_REGISTRY.register()
class DistillatorRetinaNet(_Distillator):
    _gt_kw = "gt_labels_boxes"

    def _pair_losses(self, feats_stu, feats_tea, gt_instances):
        """student.losses on both halves of one head pass; student first, so the EMA loss normaliser advances in the
        reference's order (student forward, then forward_teacher: distillator.py:44-52, 110)."""
        s = self.student
        anchors, (logits, deltas), (logits_t, deltas_t) = s.predict_pair(feats_stu, feats_tea)
        gt_labels, gt_boxes = s.label_anchors(anchors, gt_instances)
        return (s.losses(anchors, logits, gt_labels, deltas, gt_boxes),
                s.losses(anchors, logits_t, gt_labels, deltas_t, gt_boxes))

    def forward_teacher(self, batched_inputs, **kwargs):
        """[ref: distillator.py:96-114]"""
        images, r_features, features = kwargs["images"], kwargs["r_features"], kwargs["features"]
        gt_labels, gt_boxes = kwargs["gt_labels_boxes"]
        features_tea, inst_labels, geom = self.teacher((batched_inputs, images, r_features, features))
        anchors, logits_tea, deltas_tea = self.student.predict([features_tea[f] for f in self.student.head_in_features])
        losses_tea = self.student.losses(anchors, logits_tea, gt_labels, deltas_tea, gt_boxes)
        return {k + ".tea": v for k, v in losses_tea.items()}, None, features_tea, geom, inst_labels

    def _infer(self, feats, batched_inputs, images):
        anchors, logits, deltas = self.student.predict(feats)
        results = self.student.inference(anchors, logits, deltas, images.image_sizes)
        return self.student.get_processed_results(results, batched_inputs, images)


@META_ARCH_REGISTRY.register()
class DistillatorFCOS(_Distillator):
    _gt_kw = "gt_targets"

    def _pair_losses(self, feats_stu, feats_tea, gt_instances):
        s = self.student
        shifts, out_s, out_t = s.predict_pair(feats_stu, feats_tea)
        gt = s.get_ground_truth(shifts, gt_instances)
        return s.losses(*gt, *out_s), s.losses(*gt, *out_t)

    def forward_teacher(self, batched_inputs, **kwargs):
        """[ref: distillator.py:277-297]"""
        images, r_features, features = kwargs["images"], kwargs["r_features"], kwargs["features"]
        gt_classes, gt_deltas, gt_centerness = kwargs["gt_targets"]
        features_tea, inst_labels, geom = self.teacher((batched_inputs, images, r_features, features))
        _, box_cls, box_delta, box_center = self.student.predict([features_tea[f] for f in self.student.in_features])
        losses_tea = self.student.losses(gt_classes, gt_deltas, gt_centerness, box_cls, box_delta, box_center)
        return {k + ".tea": v for k, v in losses_tea.items()}, None, features_tea, geom, inst_labels

    def _infer(self, feats, batched_inputs, images):
        shifts, box_cls, box_delta, box_center = self.student.predict(feats)
        results = self.student.inference(box_cls, box_delta, box_center, shifts, images)
        return self.student.get_processed_results(results, batched_inputs, images)


def build_model(cfg):
    """[ref: train.py:262 build_model(cfg)] -> META_ARCH_REGISTRY['Distillator'+X](cfg)."""
    from . import dynamic_teacher, student  # noqa: F401  (populate the registries)
    import torch
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg)
    return model.to(torch.device(cfg.MODEL.DEVICE))
