"""BaseDistillator: student + dynamic teacher + adapter and the feature-distillation loss
[ref: models/base_distillator.py:11-77, models/customized_detectors/build.py:14-17,40-43]."""
import os
from abc import abstractmethod

import torch
import torch.nn as nn

from . import ops, streams
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

    def distill_loss(self, features, images, batched_inputs, batchified_inside_masks, inst_labels, adapted=None):
        return {"loss_distill": self.distill(features, images, batched_inputs, batchified_inside_masks, inst_labels, adapted=adapted)}

    # the adapter reads the student's features only: issued on a second stream as soon as they exist (adapt_ahead), it runs beside the teacher's
    # kernels, and its backward beside the head's (lgd_amd/streams.py).  LGD_ADAPTER_STREAM=0: inside distill(), on the caller's stream.
    adapter_stream = os.environ.get("LGD_ADAPTER_STREAM", "1") != "0"

    def adapt_ahead(self, features_stu):
        """the adapter over the student's pyramid, issued as soon as the features exist: on the side stream where that is allowed (the fork), else at
        the SAME point of the program on the caller's stream -- so that the autograd graph, and with it the order in which the feature gradients of
        the teacher, the adapter and the head are summed, does not depend on whether the step forks: the forked and the one-stream step are
        bit-identical (tools/step_determinism.py).  distill(adapted=...) picks the result up.  None where it does not apply."""
        adapter = self.adapter["distill"]
        keys = sorted(features_stu.keys())
        stu = [features_stu[k] for k in keys]
        if not (hasattr(adapter, "levels") and stu and stu[0].is_cuda):
            return None
        if self.distill_flag == 0:
            stu = [f.detach() for f in stu]
        convs = [m for m in adapter.modules() if isinstance(m, torch.nn.Conv2d)]
        # (a side stream carries this library's kernels only: ops.convs_on_own_kernels / streams.library_call)
        if not (self.adapter_stream and ops.side_streams_ok() and ops.convs_on_own_kernels(stu, [[m.weight] for m in convs])):
            return keys, adapter.levels(stu), None, None, self.distill_flag, None
        main, side = streams.fork(stu[0].device, "adapter", inputs=stu)
        streams.join_on_grad(list(adapter.parameters()), "adapter")
        with torch.cuda.stream(side):
            out = adapter.levels(stu)
        return keys, out, main, side, self.distill_flag, streams.done(side)

    def distill(self, features, images, batched_inputs, batchified_inside_masks, fg_labels, adapted=None):
        """coef * mse(IN(tea), IN(adapter(stu))) over all shared levels; teacher always detached,
        student detached while distill_flag == 0 (the adapter still trains).  `images`,
        `batched_inputs`, masks and labels are accepted and unused, as in the reference
        [ref: base_distillator.py:34-64].  InstanceNorm x2 + flatten/cat + MSE = ONE HIP kernel pass."""
        keys = sorted(features["stu"].keys() & features["tea"].keys())
        stu = [features["stu"][k] for k in keys]
        tea = [features["tea"][k].detach() for k in keys]
        if self.distill_flag == 0:
            stu = [f.detach() for f in stu]
        if adapted is not None and adapted[0] == keys and adapted[4] == self.distill_flag:
            _, stu, main, side, _, ev = adapted
            if side is not None:
                streams.join(main, side, outputs=stu, event=ev)
        else:
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
