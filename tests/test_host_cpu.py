"""CPU tests of the host-side mirror of the reference interface: config loader, registries, state_dict
names, LR schedule, phase logic, synthetic data.  (No kernel is called: the HIP path has no CPU fallback.)"""
import os
import textwrap

import pytest
import torch

from lgd_amd import config, registry
from lgd_amd.engine import warmup_multistep_factor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_style_yaml_loads(tmp_path):
    """the constructs the reference's YAML files use: _BASE_, python/object/apply:eval, 1e03, tuple strings."""
    base = tmp_path / "Base.yaml"
    base.write_text(textwrap.dedent('''
        MODEL:
          META_ARCHITECTURE: "RetinaNet"
          ANCHOR_GENERATOR:
            SIZES: !!python/object/apply:eval ["[[x, x * 2**(1.0/3), x * 2**(2.0/3) ] for x in [32, 64, 128, 256, 512 ]]"]
          RETINANET:
            IOU_THRESHOLDS: [0.4, 0.5]
            SMOOTH_L1_LOSS_BETA: 0.0
        SOLVER:
          STEPS: (60000, 80000)
          CLIP_GRADIENTS: {"ENABLED": True}
        VERSION: 2
    '''))
    child = tmp_path / "sub" / "child.yaml"
    child.parent.mkdir()
    child.write_text(textwrap.dedent('''
        _BASE_: "../Base.yaml"
        MODEL:
          RESNETS:
            DEPTH: 101
          DISTILLATOR:
            TEACHER:
              META_ARCH: 'DynamicTeacher'
              ADD_CONTEXT_BOX: True
              SOLVER:
                STEPS: (120000, 160000)
                WARMUP_FACTOR: 1e-3
                WARMUP_ITERS:  1e03
            STUDENT:
              META_ARCH: 'RetinaNetCT'
            PRE_NONDISTILL_ITERS: 30000
        OUTPUT_DIR: 'outputs/x/'
    '''))
    cfg = config.setup_cfg(str(child), ["MODEL.DISTILLATOR.LAMBDA", "0.5", "SOLVER.MAX_ITER", "100"])
    assert cfg.MODEL.META_ARCHITECTURE == "DistillatorRetinaNet"  # train.py:247-248
    assert cfg.MODEL.RESNETS.DEPTH == 101 and cfg.SOLVER.CLIP_GRADIENTS.ENABLED is True
    assert abs(cfg.MODEL.ANCHOR_GENERATOR.SIZES[1][2] - 64 * 2 ** (2 / 3)) < 1e-9
    assert tuple(cfg.MODEL.DISTILLATOR.TEACHER.SOLVER.STEPS) == (120000, 160000)
    assert cfg.MODEL.DISTILLATOR.TEACHER.SOLVER.WARMUP_ITERS == 1000 and cfg.MODEL.DISTILLATOR.LAMBDA == 0.5
    assert tuple(cfg.SOLVER.STEPS) == (60000, 80000) and cfg.SOLVER.MAX_ITER == 100
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cpu"  # frozen
    with pytest.raises(KeyError):
        config.setup_cfg(str(child), ["MODEL.NO_SUCH_KEY", "1"])


def test_eval_tag_rejects_code():
    import yaml
    with pytest.raises(yaml.YAMLError):
        yaml.load('a: !!python/object/apply:eval ["__import__(\'os\').system(\'true\')"]', Loader=config._Loader)


@pytest.mark.parametrize("name,meta,ctx", [("lgd_retinanet_r50", "DistillatorRetinaNet", True),
                                           ("lgd_retinanet_r101", "DistillatorRetinaNet", True),
                                           ("lgd_fcos_r50", "DistillatorFCOS", False)])
def test_shipped_configs(name, meta, ctx):
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", name + ".yaml"))
    d = cfg.MODEL.DISTILLATOR
    assert cfg.MODEL.META_ARCHITECTURE == meta and d.TEACHER.ADD_CONTEXT_BOX is ctx
    assert (d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS, d.LAMBDA) == (30000, 20000, 1.0)
    assert cfg.SOLVER.MAX_ITER == 180000 and d.STUDENT.SOLVER.BASE_LR == 0.01


def test_registries_resolve_reference_names():
    from lgd_amd import adapters, distillator, dynamic_teacher, student  # noqa: F401
    for n in ("DistillatorRetinaNet", "DistillatorFCOS"):
        assert registry.META_ARCH_REGISTRY.get(n).__name__ == n
    for n in ("RetinaNetCT", "FCOSCT", "DynamicTeacher"):
        assert registry.CUSTOMIZED_DETECTORS_REGISTRY.get(n).__name__ == n
    assert registry.ADAPTERS_REGISTRY.get("SequentialConvs").__name__ == "SequentialConvs"
    with pytest.raises(KeyError):
        registry.META_ARCH_REGISTRY.get("DistillatorPOTO")  # out of scope


def test_state_dict_names_match_reference():
    """names/shapes recorded from the real reference modules (tests/golden/make_golden.py loads the same dict strict=True)."""
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.dynamic_teacher import DynamicTeacher
    from oracle import lgd_oracle as O
    cfg = config.setup_cfg(None, ["MODEL.DEVICE", "cpu", "MODEL.DISTILLATOR.TEACHER.ADD_CONTEXT_BOX", "True"])
    sd = DynamicTeacher(cfg).state_dict()
    want = O.teacher_param_shapes()
    assert set(sd) == set(want) and all(tuple(sd[k].shape) == tuple(want[k]) for k in want)
    assert sum(v.numel() for v in sd.values()) == 8304336
    sa = SequentialConvs(cfg).state_dict()
    assert set(sa) == set(O.adapter_param_shapes()) and sum(v.numel() for v in sa.values()) == 1770240


def test_meta_arch_builds_with_reference_attribute_surface():
    from lgd_amd.distillator import build_model
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cpu"])
    m = build_model(cfg)
    assert m.distill_flag == cfg.MODEL.DISTILLATOR.DISTILL_OFF and m.coef == 1.0
    assert m.student.fpn is m.student.backbone and len(list(m.student.fpn.bottom_up.children())) == 0  # retinanet.py:29-34
    keys = m.state_dict().keys()
    assert "student.backbone.fpn_lateral3.weight" in keys and "student.fpn.fpn_lateral3.weight" in keys
    assert "student.raw_backbone.res4.5.conv3.norm.running_var" in keys and "student.head.cls_score.bias" in keys
    assert "adapter.distill.adapter.4.bias" in keys and "teacher.multi_head_attn.in_proj_weight" in keys
    frozen = [n for n, p in m.student.raw_backbone.named_parameters() if not p.requires_grad]
    assert all(n.startswith(("stem", "res2")) for n in frozen) and len(frozen) > 0  # FREEZE_AT=2
    assert sum(p.numel() for p in m.student.parameters()) == 37915572


def test_warmup_multistep_schedule():
    f = lambda it: warmup_multistep_factor(it, (120000, 160000), 0.1, 1e-3, 1000, "linear")  # noqa: E731
    assert abs(f(0) - 1e-3) < 1e-12 and abs(f(500) - (1e-3 * 0.5 + 0.5)) < 1e-12 and f(1000) == 1.0
    assert f(119999) == 1.0 and abs(f(120000) - 0.1) < 1e-12 and abs(f(160000) - 0.01) < 1e-12


def test_synthetic_batch_format():
    from lgd_amd.data import synthetic_batch
    b = synthetic_batch(2, 64, 96, 3, seed=1)
    assert len(b) == 2 and b[0]["image"].shape == (3, 64, 96) and len(b[0]["instances"]) == 3
    assert b[0]["instances"].gt_boxes.tensor.shape == (3, 4) and b[0]["instances"].gt_classes.dtype == torch.int64
    again = synthetic_batch(2, 64, 96, 3, seed=1)
    assert torch.equal(b[1]["image"], again[1]["image"])  # hash-based: reproducible everywhere


def test_retinanet_anchor_labels_and_losses_cpu():
    """student-side restatement (parity unpinned): self-consistency of anchors, matching and the sync-free losses."""
    from lgd_amd.student.retinanet import AnchorGenerator, box_deltas, apply_deltas, pairwise_iou, sigmoid_focal_sum
    ag = AnchorGenerator([[32, 40.3, 50.8]], [[0.5, 1.0, 2.0]], [8])
    a = ag([torch.zeros(1, 1, 4, 6)])[0]
    assert a.shape == (4 * 6 * 9, 4)
    assert torch.allclose(a[0], torch.tensor([-22.6274, -11.3137, 22.6274, 11.3137]), atol=1e-3)  # size 32, ratio .5 at (0,0)
    assert torch.allclose(a[9, :2] - a[0, :2], torch.tensor([8.0, 0.0]))                           # next cell along x
    gt = torch.tensor([[4.0, 4.0, 40.0, 30.0]])
    d = box_deltas(a, gt.expand_as(a))
    assert torch.allclose(apply_deltas(d, a), gt.expand_as(a), atol=1e-3)
    assert abs(float(pairwise_iou(gt, gt)) - 1.0) < 1e-6
    logits = torch.randn(2, 7, 5)
    labels = torch.tensor([[0, 5, 5, -1, 2, 5, 5], [5, 5, 5, 5, 5, 5, 4]])
    valid = labels >= 0
    got = sigmoid_focal_sum(logits, labels, valid, 5, 0.25, 2.0)
    t = torch.nn.functional.one_hot(labels.clamp(min=0), 6)[..., :5].float()
    p = torch.sigmoid(logits)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(logits, t, reduction="none")
    ref = ((0.25 * t + 0.75 * (1 - t)) * ce * (1 - (p * t + (1 - p) * (1 - t))) ** 2)[valid].sum()
    assert abs(float(got - ref)) < 1e-5


def test_modulated_deform_conv_matches_definition():
    """DCNv2 (BASELINE config 5): zero offsets + unit mask == plain conv; random offsets == direct bilinear sampling."""
    import torch.nn.functional as F
    from lgd_amd.student.deform import modulated_deform_conv2d
    torch.manual_seed(0)
    x, w, b = torch.randn(2, 5, 9, 11), torch.randn(7, 5, 3, 3), torch.randn(7)
    for stride in (1, 2):
        Ho, Wo = (9 + 2 - 3) // stride + 1, (11 + 2 - 3) // stride + 1
        y = modulated_deform_conv2d(x, torch.zeros(2, 18, Ho, Wo), torch.ones(2, 9, Ho, Wo), w, b, stride, 1, 1)
        assert float((y - F.conv2d(x, w, b, stride, 1)).abs().max()) < 1e-4
    off, m = torch.randn(2, 18, 9, 11) * 1.5, torch.rand(2, 9, 9, 11)
    y = modulated_deform_conv2d(x, off, m, w, b)

    def bil(img, py, px):
        H, W = img.shape[-2:]
        y0, x0 = int(torch.floor(py)), int(torch.floor(px))
        r = torch.zeros(img.shape[0])
        for yy, wy in ((y0, 1 - (py - y0)), (y0 + 1, py - y0)):
            for xx, wx in ((x0, 1 - (px - x0)), (x0 + 1, px - x0)):
                if 0 <= yy < H and 0 <= xx < W:
                    r = r + img[:, yy, xx] * wy * wx
        return r
    for n, yy, xx in ((0, 0, 0), (1, 4, 5), (0, 8, 10), (1, 3, 0)):
        acc = b.clone()
        for k in range(9):
            ky, kx = divmod(k, 3)
            acc = acc + (w[:, :, ky, kx] @ bil(x[n], yy - 1 + ky + off[n, 2 * k, yy, xx], xx - 1 + kx + off[n, 2 * k + 1, yy, xx])) * m[n, k, yy, xx]
        assert float((y[n, :, yy, xx] - acc).abs().max()) < 1e-4


def test_dcnv2_config_builds():
    from lgd_amd.distillator import build_model
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r101_dcnv2.yaml"), ["MODEL.DEVICE", "cpu"])
    m = build_model(cfg)
    keys = m.state_dict().keys()
    assert "student.raw_backbone.res3.0.conv2_offset.weight" in keys and "student.raw_backbone.res5.2.conv2_offset.bias" in keys
    assert "student.raw_backbone.res2.0.conv2_offset.weight" not in keys
    assert m.state_dict()["student.raw_backbone.res4.22.conv2_offset.weight"].shape == (27, 256, 3, 3)


@pytest.mark.parametrize("tile", [2, 4])
def test_winograd_matrices_define_the_convolution(tile):
    """host side of K8: the filter-transform matrix G the wrapper ships (ops._WINO_G, as kron(G, G)) together with the
    B^T / A^T the transforms hard-code (restated here) satisfies the minimal-filtering identity
    A^T [ (G g G^T) * (B^T d B) ] A == valid 3x3 correlation of the (tile+2)^2 window, in fp64."""
    from lgd_amd import ops
    if tile == 2:
        BT = [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]]
        AT = [[1, 1, 1, 0], [0, 1, -1, -1]]
    else:
        BT = [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
              [0, 4, 0, -5, 0, 1]]
        AT = [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]
    BT, AT = torch.tensor(BT, dtype=torch.float64), torch.tensor(AT, dtype=torch.float64)
    G = torch.tensor(ops._WINO_G[tile], dtype=torch.float64)
    n = tile + 2
    gen = torch.Generator().manual_seed(3)
    d = torch.randn(n, n, dtype=torch.float64, generator=gen)
    g = torch.randn(3, 3, dtype=torch.float64, generator=gen)
    U = (torch.kron(G, G) @ g.reshape(9)).reshape(n, n)          # what ops._wino_gg applies to the flattened filter
    assert torch.allclose(U, G @ g @ G.t(), atol=1e-12)
    y = AT @ (U * (BT @ d @ BT.t())) @ AT.t()
    ref = torch.nn.functional.conv2d(d[None, None], g[None, None])[0, 0]
    assert torch.allclose(y, ref, atol=1e-10)
    # the adjoint used for the weight gradient: frequency (1,1) of A g A^T is the plain sum of the tile's gradient
    gy = torch.randn(tile, tile, dtype=torch.float64, generator=gen)
    dM = AT.t() @ gy @ AT
    assert abs(float(dM[1, 1] - gy.sum())) < 1e-12
    assert 1 * n + 1 == tile + 3  # ... which ops._Conv3x3.backward indexes as dM[tile + 3]
