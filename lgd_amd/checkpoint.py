"""Checkpoint surface of the reference: detectron2 `DetectionCheckpointer` files
[ref: train.py:155-167 (checkpointer with stu/tea optimizers + schedulers, resume_or_load(cfg.MODEL.WEIGHTS, resume)),
 README.md:20-48 (released `.pth` models), models/customized_detectors/retinanet.py:29-34 (alias modules)].

A reference checkpoint is `{"model": state_dict, "stu_optimizer", "tea_optimizer", "stu_scheduler", "tea_scheduler",
"iteration"}` where the model dict
  * repeats the FPN under two names (`student.backbone.*` and `student.fpn.*` are the same module) next to
    `student.raw_backbone.*` (the bottom-up ResNet moved out of the FPN),
  * carries detectron2 buffers this build does not persist (`student.pixel_mean/std`, `student.anchor_generator.cell_anchors.*`),
  * carries FrozenBN buffers (`*.norm.{weight,bias,running_mean,running_var}`),
  * may be prefixed with `module.` (saved from a DDP wrapper) and may hold numpy arrays (converted pickles).
ImageNet backbones (`MODEL.WEIGHTS: detectron2://ImageNetPretrained/MSRA/R-50.pkl`) are Caffe2-named pickles
(`{"model": {...}, "__author__": "Caffe2", "matching_heuristics": True}`); their names are mapped onto
`student.raw_backbone.*` by `convert_c2_backbone_names` ([d2-memory] of detectron2's c2 name conversion, backbone part only).

No file of either kind can be fetched here (no network): tests/test_host_cpu.py round-trips synthetic files written under the
key names of SURVEY.md section 8c.
"""
import os
import pickle
import re

import numpy as np
import torch

# reference-side buffers that have no persistent counterpart here (rebuilt from the config at construction)
_IGNORABLE = (re.compile(r"^student\.pixel_(mean|std)$"), re.compile(r"^student\.anchor_generator\.cell_anchors\.\d+$"),
              re.compile(r"^student\.shift_generator\."), re.compile(r"\.num_batches_tracked$"))


class CheckpointReport:
    def __init__(self):
        self.loaded, self.missing, self.unexpected, self.ignored, self.alias_conflicts = [], [], [], [], []
        self.iteration = None

    def __repr__(self):
        return ("CheckpointReport(loaded=%d, missing=%d, unexpected=%d, ignored=%d, iteration=%s)"
                % (len(self.loaded), len(self.missing), len(self.unexpected), len(self.ignored), self.iteration))


def read_file(path):
    """`.pth` (torch.save) or `.pkl` (pickle, detectron2 model zoo) -> dict with at least "model"."""
    if not os.path.isfile(path):
        raise FileNotFoundError("checkpoint %r is not a local file (remote detectron2:// / https:// weights cannot be fetched "
                                "here: download them and point MODEL.WEIGHTS at the file)" % path)
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if "model" not in data:  # bare {name: array} dict of old model-zoo files
            data = {"model": data, "__author__": "Caffe2", "matching_heuristics": True}
        return data
    data = torch.load(path, map_location="cpu", weights_only=False)
    return data if isinstance(data, dict) and "model" in data else {"model": data}


def convert_c2_backbone_names(sd):
    """Caffe2 / MSRA ResNet names -> `student.raw_backbone.*` ([d2-memory] convert_basic_c2_names, backbone part):
    conv1_w -> stem.conv1.weight; res2_0_branch2a_w -> res2.0.conv1.weight (2b -> conv2, 2c -> conv3, branch1 -> shortcut);
    *_bn_s / *_bn_b -> .norm.weight / .norm.bias (the affine Caffe2 folded the batch statistics into; running_mean = 0 and
    running_var = 1 stay at their constructed values); fc1000 / *_momentum blobs are dropped."""
    out = {}
    for k, v in sd.items():
        if k.startswith("fc1000") or k.endswith("_momentum"):
            continue
        n = k.replace("res_conv1_bn_", "conv1_bn_")  # the stem's affine is stored as res_conv1_bn_{s,b}
        n = n.replace("_bn_s", ".norm.weight").replace("_bn_b", ".norm.bias")
        n = re.sub(r"_w$", ".weight", n)
        n = re.sub(r"_b$", ".bias", n)
        n = re.sub(r"^conv1\.", "stem.conv1.", n)
        n = re.sub(r"^res(\d)_(\d+)_branch1", r"res\1.\2.shortcut", n)
        for c2, d2 in (("branch2a", "conv1"), ("branch2b", "conv2"), ("branch2c", "conv3")):
            n = re.sub(r"^res(\d)_(\d+)_%s" % c2, r"res\1.\2.%s" % d2, n)
        out["student.raw_backbone." + n] = v
    return out


def _as_tensor(v):
    if isinstance(v, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(v))
    return v


def load_model_state(model, sd, c2_backbone=False):
    """copy a reference-format model dict into `model` (any device); returns a CheckpointReport.  Alias keys
    (`student.backbone.*` == `student.fpn.*`) are both accepted; if the two copies in the FILE differ the conflict is
    reported and the `student.backbone.*` copy wins (it is the name detectron2's own RetinaNet registers).  Shape mismatches
    raise."""
    rep = CheckpointReport()
    sd = {(k[len("module."):] if k.startswith("module.") else k): _as_tensor(v) for k, v in sd.items()}
    if c2_backbone:
        sd = convert_c2_backbone_names(sd)
    own = model.state_dict()
    # the file's two FPN copies must agree
    for k, v in sd.items():
        if k.startswith("student.fpn."):
            twin = "student.backbone." + k[len("student.fpn."):]
            if twin in sd and torch.is_tensor(v) and torch.is_tensor(sd[twin]) and v.shape == sd[twin].shape \
                    and not torch.equal(v, sd[twin]):
                rep.alias_conflicts.append(k)
    for k in sorted(sd, key=lambda n: (n.startswith("student.backbone."), n)):  # backbone.* last: it wins a conflict
        v = sd[k]
        if k not in own:
            (rep.ignored if any(p.search(k) for p in _IGNORABLE) else rep.unexpected).append(k)
            continue
        if not torch.is_tensor(v):
            raise TypeError("checkpoint entry %s is %s, expected a tensor / ndarray" % (k, type(v).__name__))
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError("shape mismatch for %s: checkpoint %s vs model %s" % (k, tuple(v.shape), tuple(own[k].shape)))
        with torch.no_grad():
            own[k].copy_(v.to(own[k].dtype))
        rep.loaded.append(k)
    have = set(rep.loaded)
    for k in own:
        if k in have:
            continue
        # an alias whose twin was loaded shares its storage: not missing
        twin = None
        if k.startswith("student.fpn."):
            twin = "student.backbone." + k[len("student.fpn."):]
        elif k.startswith("student.backbone."):
            twin = "student.fpn." + k[len("student.backbone."):]
        if twin in have:
            continue
        rep.missing.append(k)
    return rep


def load_checkpoint(path, model, trainer=None, resume=False):
    """[ref: train.py:160 `checkpointer.resume_or_load(cfg.MODEL.WEIGHTS, resume=resume)`]: model weights always; optimizers,
    schedulers and the iteration counter only when resuming and `trainer` is given."""
    data = read_file(path)
    c2 = bool(data.get("matching_heuristics", False)) or data.get("__author__") == "Caffe2"
    rep = load_model_state(model, data["model"], c2_backbone=c2)
    if resume and trainer is not None and "stu_optimizer" in data:
        from .engine import load_scheduler_state, optimizer_state_from_reference
        trainer.stu_optimizer.load_state_dict(optimizer_state_from_reference(data["stu_optimizer"], trainer.stu_optimizer))
        trainer.tea_optimizer.load_state_dict(optimizer_state_from_reference(data["tea_optimizer"], trainer.tea_optimizer))
        load_scheduler_state(trainer.stu_scheduler, data["stu_scheduler"])
        load_scheduler_state(trainer.tea_scheduler, data["tea_scheduler"])
        trainer.iteration = data.get("iteration", -1) + 1
        rep.iteration = trainer.iteration
    return rep
