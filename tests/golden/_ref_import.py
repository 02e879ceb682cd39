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
