"""BaseDistillator: student + dynamic teacher + adapter and the feature-distillation loss
[ref: models/base_distillator.py:11-77, models/customized_detectors/build.py:14-17,40-43]."""
from abc import abstractmethod

import torch
import torch.nn as nn

from . import ops
from .adapters import build_adapter
from .registry import CUSTOMIZED_DETECTORS_REGISTRY


def get_model(cfg, meta_arch):
    return CUSTOMIZED_DETECTORS_REGISTRY.get(meta_arch)(cfg).to(torch.device(cfg.MODEL.DEVICE))


def build_customized_detector(cfg):
    d = cfg.MODEL.DISTILLATOR
    return get_model(cfg, d.STUDENT.META_ARCH), get_model(cfg, d.TEACHER.META_ARCH)


class BaseDistillator(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        # kept for state_dict/attribute compatibility; affine-free and statistics-free, so they
        # own no tensors -- the normalisation itself runs inside the fused HIP loss kernel.
        self.norm_stu = nn.InstanceNorm2d(256, affine=False)
        self.norm_tea = nn.InstanceNorm2d(256, affine=False)
        self.student, self.teacher = build_customized_detector(cfg)
        self.coef = cfg.MODEL.DISTILLATOR.LAMBDA
        self.add_bg_box = cfg.MODEL.DISTILLATOR.TEACHER.ADD_CONTEXT_BOX
        self.adapter = nn.ModuleDict({"distill": build_adapter(cfg)})
        # the reference leaves distill_flag to the training loop (train.py:266); default = OFF like there
        self.distill_flag = cfg.MODEL.DISTILLATOR.DISTILL_OFF

    def distill_loss(self, features, images, batched_inputs, batchified_inside_masks, inst_labels):
        return {"loss_distill": self.distill(features, images, batched_inputs, batchified_inside_masks, inst_labels)}

    def distill(self, features, images, batched_inputs, batchified_inside_masks, fg_labels):
        """coef * mse(IN(tea), IN(adapter(stu))) over all shared levels; teacher always detached,
        student detached while distill_flag == 0 (the adapter still trains).  `images`,
        `batched_inputs`, masks and labels are accepted and unused, as in the reference
        [ref: base_distillator.py:34-64].  InstanceNorm x2 + flatten/cat + MSE = ONE HIP kernel pass."""
        keys = sorted(features["stu"].keys() & features["tea"].keys())
        stu = [features["stu"][k] for k in keys]
        tea = [features["tea"][k].detach() for k in keys]
        if self.distill_flag == 0:
            stu = [f.detach() for f in stu]
        adapter = self.adapter["distill"]
        stu = adapter.levels(stu) if hasattr(adapter, "levels") else [adapter(f) for f in stu]
        return ops.distill_in_mse(stu, tea, self.coef)

    @abstractmethod
    def forward(self, batched_inputs, **kwargs):
        pass

    @abstractmethod
    def forward_student(self, batched_inputs, **kwargs):
        pass

    @abstractmethod
    def forward_teacher(self, batched_inputs, **kwargs):
        pass
