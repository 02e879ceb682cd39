"""Import the reference LGD hot-path modules from /root/reference WITHOUT detectron2/cvpods.

Used ONLY by tests/golden/make_golden.py (in the build container, where
/root/reference exists) to emit golden input/output vectors.  Nothing in the
product, the `-m gpu` tests, smoke() or bench.py imports this file: the
reference does not exist on the GPU box.

How: the hot-path modules only need detectron2's
`Registry` and `META_ARCH_REGISTRY` symbols at import time, so tiny dict-backed
stand-ins are registered under those module names; the package __init__ files
that pull in detectron2's RetinaNet / cvpods are bypassed by pre-seeding empty
package objects whose __path__ points into the reference tree.
"""
import importlib
import sys
import types

REF_ROOT = "/root/reference"


class _Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self[o.__name__] = o
                return o
            return deco
        self[obj.__name__] = obj
        return obj

    def get(self, name):
        return self[name]


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Returns a namespace with the reference classes/functions on the hot path."""
    if "models.base_distillator" in sys.modules and hasattr(sys.modules["models"], "_lgd_ref"):
        return sys.modules["models"]._lgd_ref

    _mod("detectron2")
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("detectron2.modeling", META_ARCH_REGISTRY=_Registry("META_ARCH"))
    _mod("detectron2.structures")
    _mod("detectron2.structures.masks")

    pkg = _mod("models")
    pkg.__path__ = [REF_ROOT + "/models"]
    cd = _mod("models.customized_detectors")
    cd.__path__ = [REF_ROOT + "/models/customized_detectors"]

    build = importlib.import_module("models.customized_detectors.build")
    dt = importlib.import_module("models.customized_detectors.dynamic_teacher.dynamic_teacher")
    le = importlib.import_module("models.customized_detectors.dynamic_teacher.label_encoder")
    ut = importlib.import_module("models.customized_detectors.dynamic_teacher.utils")
    st = importlib.import_module("models.customized_detectors.dynamic_teacher.spatial_transformer")
    cd.build_customized_detector = build.build_customized_detector
    cd.CUSTOMIZED_DETECTORS_REGISTRY = build.CUSTOMIZED_DETECTORS_REGISTRY
    adapters = importlib.import_module("models.adapters")
    base = importlib.import_module("models.base_distillator")

    ns = types.SimpleNamespace(
        DynamicTeacher=dt.DynamicTeacher,
        LabelEncoder=le.LabelEncoder,
        box_descriptor_encode=le.box_descriptor_encode,
        STN=st.STN,
        get_inside_gt_mask=ut.get_inside_gt_mask,
        resolution=ut.resolution,
        BaseDistillator=base.BaseDistillator,
        SequentialConvs=adapters.SequentialConvs,
    )
    pkg._lgd_ref = ns
    return ns


def make_cfg(add_ctx=True, interact="stuGuided", detach_app=False, box_format="x1y1x2y2", coef=1.0):
    """SimpleNamespace cfg carrying exactly the keys the hot path reads."""
    NS = types.SimpleNamespace
    return NS(
        NUM_CLASSES=80,
        MODEL=NS(
            DEVICE="cpu",
            FPN=NS(OUT_CHANNELS=256),
            RECIPROCAL_FPN_STRIDES=[1 / 8, 1 / 16, 1 / 32, 1 / 64, 1 / 128],
            DISTILLATOR=NS(
                LAMBDA=coef,
                TEACHER=NS(INTERACT_PATTERN=interact, ADD_CONTEXT_BOX=add_ctx,
                           DETACH_APPEARANCE_EMBED=detach_app, NR_TRANSFORMER_HEADS=8,
                           META_ARCH="DynamicTeacher"),
                LABEL_ENCODER=NS(BOX_FORMAT=box_format, CATEGORY_FORMAT="one_hot", LOAD_LABELMAP=False),
                ADAPTER=NS(META_ARCH="SequentialConvs"),
            ),
        ),
    )


def import_reference_fcos():
    """The reference's in-tree FCOS (models/customized_detectors/thirdparty_heads/fcos.py: `FCOSHead` :433-546, `FCOS.get_ground_truth`
    :177-284, `FCOS.losses` :105-175) imported with stand-ins for the cvpods / detectron2 names its module header pulls in.

    Stand-ins WITHOUT arithmetic (names only; nothing they compute reaches a fixture):
      cvpods.layers.ShapeSpec / generalized_batched_nms, cvpods.utils.comm / log_first_n, cvpods.modeling.losses.iou_loss /
      sigmoid_focal_loss_jit (only `FCOS.losses` calls them -- not run here: their arithmetic stays UNPINNED),
      detectron2.modeling.build_backbone, detectron2.structures.ImageList / Instances,
      cvpods.modeling.anchor_generator.ShiftGenerator (FCOSHead.__init__ reads `.num_cell_shifts`: one shift per cell, the
      configs' NUM_SHIFTS = 1), cvpods.layers.cat (= torch.cat).
    Stand-ins WITH arithmetic -- public one-line definitions restated from memory, so what they compute is NOT pinned by the fixtures
    that flow through them (the fixtures pin everything the reference does AROUND them: centre sampling, size ranges, min-area ties,
    background, centerness):
      cvpods.modeling.box_regression.Shift2BoxTransform.get_deltas(shifts, boxes) = cat(shifts - boxes[..., :2], boxes[..., 2:] - shifts)
      detectron2.structures.Boxes.get_centers() = (xy1 + xy2) / 2, .area() = (x2 - x1) * (y2 - y1), indexing."""
    import torch
    if "models.customized_detectors.thirdparty_heads.fcos" in sys.modules:
        return sys.modules["models"]._lgd_ref_fcos
    import_reference()

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    class ShiftGenerator:
        def __init__(self, cfg, input_shape):
            self.num_cell_shifts = [1 for _ in input_shape]

    class Shift2BoxTransform:
        def __init__(self, weights):
            self.weights = weights

        def get_deltas(self, shifts, boxes):
            return torch.cat((shifts - boxes[..., :2], boxes[..., 2:] - shifts), dim=-1)

    class Boxes:
        def __init__(self, tensor):
            self.tensor = tensor

        def get_centers(self):
            return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

        def area(self):
            b = self.tensor
            return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

        def __getitem__(self, item):
            return Boxes(self.tensor[item])

        def __len__(self):
            return self.tensor.shape[0]

    def _unused(*a, **k):
        raise RuntimeError("stand-in without arithmetic was called")

    sys.modules["detectron2.modeling"].build_backbone = _unused
    st = sys.modules["detectron2.structures"]
    st.ImageList, st.Instances, st.Boxes = object, object, Boxes
    _mod("cvpods")
    _mod("cvpods.modeling")
    _mod("cvpods.modeling.anchor_generator", ShiftGenerator=ShiftGenerator)
    _mod("cvpods.layers", ShapeSpec=ShapeSpec, cat=torch.cat, generalized_batched_nms=_unused)
    _mod("cvpods.modeling.box_regression", Shift2BoxTransform=Shift2BoxTransform)
    _mod("cvpods.modeling.losses", iou_loss=_unused, sigmoid_focal_loss_jit=_unused)
    _mod("cvpods.utils", comm=types.SimpleNamespace(), log_first_n=_unused)
    th = _mod("models.customized_detectors.thirdparty_heads")
    th.__path__ = [REF_ROOT + "/models/customized_detectors/thirdparty_heads"]
    fc = importlib.import_module("models.customized_detectors.thirdparty_heads.fcos")
    ns = types.SimpleNamespace(FCOS=fc.FCOS, FCOSHead=fc.FCOSHead, Boxes=Boxes, ShapeSpec=ShapeSpec, Shift2BoxTransform=Shift2BoxTransform)
    sys.modules["models"]._lgd_ref_fcos = ns
    return ns
