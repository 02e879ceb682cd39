"""Config tree for the LGD path: a small yacs-free CfgNode that accepts the reference's
`configs/Distillation/**.yaml` files unchanged (`_BASE_` inheritance, `KEY VALUE` overrides,
string literals such as `1e03` / `(120000, 160000)` parsed like yacs does, and the
`!!python/object/apply:eval` anchor-size tag of configs/Base-RetinaNet.yaml:8).

Defaults: the detectron2 v0.3 keys the student restatement reads (SURVEY.md appendix A,
[d2-memory]) + `build_distillator_configs` [ref: utils/build.py:557-653] + `build_fcos`
[ref: utils/build.py:671-703].
"""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    """dict with attribute access, recursive merge and freeze."""

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self._frozen:
            raise AttributeError("config is frozen; cannot set %s" % k)
        self[k] = v

    def freeze(self, flag=True):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze(flag)
        return self

    def defrost(self):
        return self.freeze(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out

    # ---- merging ---------------------------------------------------------------------------
    def merge_from_other(self, other, path=""):
        for k, v in other.items():
            full = path + k
            if isinstance(v, dict):
                if k not in self:
                    self[k] = CfgNode()
                if not isinstance(self[k], CfgNode):
                    raise KeyError("%s: cannot merge a mapping into a leaf" % full)
                self[k].merge_from_other(v, full + ".")
            else:
                if k not in self and not full.startswith("OUTPUT_DIR"):
                    raise KeyError("unknown config key: %s" % full)
                self[k] = _coerce(v, self.get(k), full)

    def merge_from_file(self, path):
        self.merge_from_other(load_yaml_with_base(path))

    def merge_from_list(self, opts):
        if len(opts) % 2:
            raise ValueError("override list must be KEY VALUE pairs: %s" % (opts,))
        for key, val in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("unknown config key: %s" % key)
            node[parts[-1]] = _coerce(_literal(val), node[parts[-1]], key)


def _literal(v):
    """yacs semantics: a string that parses as a python literal becomes that literal."""
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def _coerce(new, old, key):
    new = _literal(new)
    if old is None or new is None:
        return new
    if isinstance(old, bool) or isinstance(new, bool):
        return new
    if isinstance(old, float) and isinstance(new, int):
        return float(new)
    if isinstance(old, int) and isinstance(new, float) and float(new).is_integer():
        return int(new)  # WARMUP_ITERS: 1e03
    if isinstance(old, (list, tuple)) and isinstance(new, (list, tuple)):
        return type(old)(new) if isinstance(old, tuple) else list(new)
    return new


class _Loader(yaml.SafeLoader):
    pass


def _apply_eval(loader, node):
    # Base-RetinaNet.yaml:8 builds the anchor sizes with a python expression.  Only arithmetic
    # list comprehensions are accepted: evaluated with no builtins and no names.
    args = loader.construct_sequence(node)
    if len(args) != 1 or not isinstance(args[0], str):
        raise yaml.YAMLError("unsupported python/object/apply:eval payload")
    tree = ast.parse(args[0], mode="eval")
    allowed = (ast.Expression, ast.ListComp, ast.List, ast.Tuple, ast.BinOp, ast.UnaryOp, ast.Constant, ast.Name,
               ast.comprehension, ast.Load, ast.Store, ast.Mult, ast.Div, ast.Add, ast.Sub, ast.Pow, ast.USub)
    for n in ast.walk(tree):
        if not isinstance(n, allowed):
            raise yaml.YAMLError("disallowed expression in eval tag: %s" % type(n).__name__)
    return eval(compile(tree, "<cfg>", "eval"), {"__builtins__": {}}, {})


_Loader.add_constructor("tag:yaml.org,2002:python/object/apply:eval", _apply_eval)


def load_yaml_with_base(path):
    with open(path) as f:
        cfg = yaml.load(f, Loader=_Loader) or {}
    base = cfg.pop("_BASE_", None)
    if base is None:
        return cfg
    if not os.path.isabs(base):
        base = os.path.join(os.path.dirname(path), base)
    merged = load_yaml_with_base(base)

    def rec(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                rec(dst[k], v)
            else:
                dst[k] = v
    rec(merged, cfg)
    return merged


# --------------------------------------------------------------------------------------------- defaults
def _solver_node(lr):
    return dict(OPTIMIZER="SGD", BASE_LR=lr, MOMENTUM=0.9, WEIGHT_DECAY=1e-4, LR_SCHEDULER_NAME=None, STEPS=None,
                GAMMA=None, WARMUP_FACTOR=None, WARMUP_ITERS=None, WARMUP_METHOD=None, AMP=dict(ENABLED=False))


def get_cfg():
    """detectron2-style defaults for the keys this build reads ([d2-memory])."""
    return CfgNode(dict(
        VERSION=2,
        OUTPUT_DIR="./output",
        SEED=-1,
        VIS_PERIOD=0,
        MODEL=dict(
            DEVICE="cuda",
            META_ARCHITECTURE="RetinaNet",
            WEIGHTS="",
            MASK_ON=False,
            PIXEL_MEAN=[103.530, 116.280, 123.675],
            PIXEL_STD=[1.0, 1.0, 1.0],
            BACKBONE=dict(NAME="build_retinanet_resnet_fpn_backbone", FREEZE_AT=2),
            RESNETS=dict(DEPTH=50, OUT_FEATURES=["res3", "res4", "res5"], NUM_GROUPS=1, WIDTH_PER_GROUP=64,
                         NORM="FrozenBN", STRIDE_IN_1X1=True, RES5_DILATION=1, RES2_OUT_CHANNELS=256,
                         STEM_OUT_CHANNELS=64, DEFORM_ON_PER_STAGE=[False, False, False, False],
                         DEFORM_MODULATED=False, DEFORM_NUM_GROUPS=1),
            FPN=dict(IN_FEATURES=["res3", "res4", "res5"], OUT_CHANNELS=256, NORM="", FUSE_TYPE="sum"),
            ANCHOR_GENERATOR=dict(NAME="DefaultAnchorGenerator", SIZES=[[32, 64, 128, 256, 512]],
                                  ASPECT_RATIOS=[[0.5, 1.0, 2.0]], ANGLES=[[-90, 0, 90]], OFFSET=0.0),
            RETINANET=dict(NUM_CLASSES=80, IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], NUM_CONVS=4,
                           IOU_THRESHOLDS=[0.4, 0.5], IOU_LABELS=[0, -1, 1], PRIOR_PROB=0.01,
                           SCORE_THRESH_TEST=0.05, TOPK_CANDIDATES_TEST=1000, NMS_THRESH_TEST=0.5,
                           BBOX_REG_WEIGHTS=(1.0, 1.0, 1.0, 1.0), FOCAL_LOSS_GAMMA=2.0, FOCAL_LOSS_ALPHA=0.25,
                           SMOOTH_L1_LOSS_BETA=0.1, BBOX_REG_LOSS_TYPE="smooth_l1", NORM=""),
        ),
        INPUT=dict(MIN_SIZE_TRAIN=(800,), MIN_SIZE_TRAIN_SAMPLING="choice", MAX_SIZE_TRAIN=1333, MIN_SIZE_TEST=800,
                   MAX_SIZE_TEST=1333, FORMAT="BGR", RANDOM_FLIP="horizontal"),
        DATASETS=dict(TRAIN=(), TEST=()),
        DATALOADER=dict(NUM_WORKERS=4, ASPECT_RATIO_GROUPING=True),
        SOLVER=dict(IMS_PER_BATCH=16, BASE_LR=0.001, STEPS=(30000,), MAX_ITER=40000, MOMENTUM=0.9, NESTEROV=False,
                    WEIGHT_DECAY=1e-4, GAMMA=0.1, WARMUP_FACTOR=1e-3, WARMUP_ITERS=1000, WARMUP_METHOD="linear",
                    CHECKPOINT_PERIOD=5000,
                    CLIP_GRADIENTS=dict(ENABLED=False, CLIP_TYPE="value", CLIP_VALUE=1.0, NORM_TYPE=2.0)),
        TEST=dict(EVAL_PERIOD=0, DETECTIONS_PER_IMAGE=100),
    ))


def build_distillator_configs(cfg):
    """[ref: utils/build.py:557-653] -- same keys and defaults."""
    cfg.NUM_CLASSES = 80
    cfg.MODEL.DISTILLATOR = CfgNode(dict(
        STUDENT=dict(SOLVER=_solver_node(0.02), META_ARCH=None),
        TEACHER=dict(SOLVER=_solver_node(0.02), META_ARCH=None, INTERACT_PATTERN="stuGuided", NR_TRANSFORMER_HEADS=8,
                     DETACH_APPEARANCE_EMBED=False, ADD_CONTEXT_BOX=False, AFFINE=False),
        ADAPTER=dict(META_ARCH="SequentialConvs"),
        PRE_NONDISTILL_ITERS=40000, POST_NONDISTILL_ITERS=0, PRE_FREEZE_STUDENT_BACKBONE_ITERS=10000,
        DISTILL_OFF=0, DISTILL_ON=1, HIDDEN_DIM=64, SMOOTH=0, EVAL_TEACHER=True,
        LABEL_ENCODER=dict(LOAD_LABELMAP=False, BOX_FORMAT="x1y1x2y2", CATEGORY_FORMAT="one_hot"),
        KNOWLEDGE_MAPPER=dict(),
        LAMBDA=1.0, TOWER_DISTILL_COEF=1.0, USE_MTH_HEAD=1, DETACH_TEA_WHEN_DISTILL=True, ADAIN_BEFORE_DISTILL=False,
    ))
    cfg.MODEL.RECIPROCAL_FPN_STRIDES = [1 / 8, 1 / 16, 1 / 32, 1 / 64, 1 / 128]
    cfg.MODEL.LOAD_BOXMAP = False
    cfg.MODEL.STRONGER_AUGS = False
    cfg.MODEL.LOAD_BOX_MASK = False
    # [ref: utils/build.py:671-703] cvpods-style FCOS keys
    cfg.MODEL.FCOS = CfgNode(dict(
        NUM_CLASSES=80, IN_FEATURES=["p3", "p4", "p5", "p6", "p7"], NUM_CONVS=4, FPN_STRIDES=[8, 16, 32, 64, 128],
        PRIOR_PROB=0.01, CENTERNESS_ON_REG=True, NORM_REG_TARGETS=True, SCORE_THRESH_TEST=0.05,
        TOPK_CANDIDATES_TEST=1000, NMS_THRESH_TEST=0.6, BBOX_REG_WEIGHTS=(1.0, 1.0, 1.0, 1.0), FOCAL_LOSS_GAMMA=2.0,
        FOCAL_LOSS_ALPHA=0.25, IOU_LOSS_TYPE="giou", CENTER_SAMPLING_RADIUS=1.5,
        OBJECT_SIZES_OF_INTEREST=[[-1, 64], [64, 128], [128, 256], [256, 512], [512, float("inf")]],
        NORM_SYNC=True, REG_WEIGHT=2.0))
    cfg.MODEL.SHIFT_GENERATOR = CfgNode(dict(NUM_SHIFTS=1, OFFSET=0.5))
    cfg.MODEL.NMS_TYPE = "normal"
    cfg.MODEL.FPN.TOP_LEVELS = 2
    return cfg


def setup_cfg(config_file=None, opts=()):
    """[ref: train.py:237-256]: defaults -> distillator keys -> yaml -> 'Distillator'+META_ARCHITECTURE -> overrides."""
    cfg = build_distillator_configs(get_cfg())
    if config_file:
        cfg.merge_from_file(config_file)
    if not cfg.MODEL.META_ARCHITECTURE.startswith("Distillator"):
        cfg.MODEL.META_ARCHITECTURE = "Distillator" + cfg.MODEL.META_ARCHITECTURE
    cfg.merge_from_list(list(opts))
    return cfg.freeze()
