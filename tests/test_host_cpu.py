"""CPU tests of the host-side mirror of the reference interface: config loader, registries, state_dict
names, LR schedule, phase logic, synthetic data.  (No kernel is called: the HIP path has no CPU fallback.)"""
import os
import textwrap

import pytest
import torch

from lgd_amd import config, registry
from lgd_amd.engine import warmup_cosine_factor, warmup_multistep_factor

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


def test_warmup_cosine_schedule():
    """[ref: utils/build.py:544-551 WarmupCosineLR]"""
    import math
    f = lambda it: warmup_cosine_factor(it, 1000, 1e-3, 100, "linear")  # noqa: E731
    assert abs(f(0) - 1e-3) < 1e-12 and abs(f(500) - 0.5) < 1e-12 and abs(f(1000)) < 1e-12
    assert abs(f(50) - (1e-3 * 0.5 + 0.5) * 0.5 * (1 + math.cos(math.pi * 0.05))) < 1e-12
    from lgd_amd.engine import build_distillator_lr_scheduler
    cfg = config.setup_cfg(None, ["MODEL.DISTILLATOR.STUDENT.SOLVER.LR_SCHEDULER_NAME", "WarmupCosineLR",
                                  "MODEL.DISTILLATOR.STUDENT.SOLVER.WARMUP_FACTOR", "0.001",
                                  "MODEL.DISTILLATOR.STUDENT.SOLVER.WARMUP_ITERS", "100",
                                  "MODEL.DISTILLATOR.STUDENT.SOLVER.WARMUP_METHOD", "linear"])
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], 0.1)
    sch = build_distillator_lr_scheduler(cfg.MODEL.DISTILLATOR.STUDENT.SOLVER, opt, max_iter=1000)
    for _ in range(500):
        opt.step()
        sch.step()
    assert abs(opt.param_groups[0]["lr"] - 0.05) < 1e-9


def test_optimizer_layout_matches_reference_and_round_trips():
    """[ref: utils/build.py:494-512] one param group per parameter, every parameter that requires grad at build time in
    named_parameters order -- incl. the never-trained global_ctx_proj_1D; Trainer.state_dict() writes that layout and
    load_state_dict() reads it back into the fused single-group optimizers."""
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_fcos_r50.yaml"), ["MODEL.DEVICE", "cpu"])  # ctx box off
    m = build_model(cfg)
    tr = Trainer(cfg, m, device=torch.device("cpu"), distributed=False)
    want_tea = [n for n, _ in m.teacher.named_parameters()]
    assert not m.teacher.global_ctx_proj_1D.weight.requires_grad  # frozen statically (never receives a gradient)
    assert len(tr.tea_optimizer.param_groups) == 1 and len(tr.tea_optimizer.param_groups[0]["params"]) == len(want_tea)
    seen, want_stu = set(), []
    for mod in (m.student, m.adapter):
        for n, p in mod.named_parameters():
            if id(p) not in seen and (p.requires_grad or getattr(p, "_lgd_phase_frozen", False)):
                seen.add(id(p))
                want_stu.append(p)
    assert [id(p) for p in tr.stu_optimizer.param_groups[0]["params"]] == [id(p) for p in want_stu]
    # ... and that order is the REFERENCE's, written out: its FCOS registers `backbone` (the FPN, whose bottom_up the CT wrapper
    # empties) and `head` (thirdparty_heads/fcos.py:93-97; FCOSHead.__init__ :433-512), then the alias `fpn` (skipped by the
    # memo) and `raw_backbone` (customized_detectors/fcos.py:23-27); the adapter follows (utils/build.py:511).  A reference
    # checkpoint indexes its one-group-per-parameter optimizer state by exactly this sequence.
    fpn = ["backbone.%s.%s" % (m_, wb) for m_ in ("fpn_lateral3", "fpn_output3", "fpn_lateral4", "fpn_output4", "fpn_lateral5",
                                                  "fpn_output5", "top_block.p6", "top_block.p7") for wb in ("weight", "bias")]
    head = ["head.%s.%d.%s" % (sub, 3 * i + j, wb) for sub in ("cls_subnet", "bbox_subnet") for i in range(4) for j in (0, 1)
            for wb in ("weight", "bias")]
    head += ["head.%s.%s" % (m_, wb) for m_ in ("cls_score", "bbox_pred", "centerness") for wb in ("weight", "bias")]
    head += ["head.scales.%d.scale" % i for i in range(5)]
    names = {id(p): n for n, p in m.student.named_parameters(remove_duplicate=False) if not n.startswith("fpn.")}
    got = [names[id(p)] for p in tr.stu_optimizer.param_groups[0]["params"] if id(p) in names]
    assert got[:len(fpn) + len(head)] == fpn + head
    rest = got[len(fpn) + len(head):]
    assert rest[0] == "raw_backbone.res3.0.shortcut.weight" and all(n.startswith("raw_backbone.res") for n in rest)
    n_adapter = len(list(m.adapter.parameters()))
    assert [id(p) for p in tr.stu_optimizer.param_groups[0]["params"][-n_adapter:]] == [id(p) for p in m.adapter.parameters()]
    for p in tr.tea_optimizer.param_groups[0]["params"][:3]:  # give some state to carry
        p.grad = torch.ones_like(p)
    tr.tea_optimizer.step()
    sd = tr.state_dict()
    assert len(sd["tea_optimizer"]["param_groups"]) == len(want_tea)
    assert all(len(g["params"]) == 1 and g["weight_decay"] == 1e-4 for g in sd["tea_optimizer"]["param_groups"])
    assert [g["params"][0] for g in sd["tea_optimizer"]["param_groups"]] == list(range(len(want_tea)))
    tr2 = Trainer(cfg, build_model(cfg), device=torch.device("cpu"), distributed=False)
    tr2.load_state_dict(sd)
    st = tr2.tea_optimizer.state_dict()
    assert len(st["param_groups"]) == 1 and sorted(st["state"]) == [0, 1, 2]
    assert torch.equal(st["state"][0]["momentum_buffer"], tr.tea_optimizer.state_dict()["state"][0]["momentum_buffer"])


def test_resume_from_reference_shaped_optimizer_and_scheduler_state():
    """a checkpoint as the REFERENCE writes it: one param group per parameter in the reference's named_parameters order
    (utils/build.py:494-512) and detectron2 WarmupMultiStepLR scheduler states (no `lr_lambdas`, one base_lr per parameter).
    Momentum buffers must land on the parameter of the same reference index -- checked by NAME with a buffer that encodes it."""
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cpu"])
    m = build_model(cfg)
    tr = Trainer(cfg, m, device=torch.device("cpu"), distributed=False)
    # the reference's own enumeration, restated from utils/build.py:494-512 over the reference's registration order
    ref_order, memo = [], set()
    stu = m.student
    for prefix, mod in (("student.backbone.", stu.backbone), ("student.head.", stu.head), ("student.raw_backbone.", stu.raw_backbone),
                        ("adapter.", m.adapter)):
        for n, p in mod.named_parameters():
            if (p.requires_grad or getattr(p, "_lgd_phase_frozen", False)) and id(p) not in memo:
                memo.add(id(p))
                ref_order.append((prefix + n, p))
    state = {i: {"momentum_buffer": torch.full_like(p, float(i))} for i, (_, p) in enumerate(ref_order)}
    groups = [{"lr": 0.01, "momentum": 0.9, "dampening": 0, "weight_decay": 1e-4, "nesterov": False, "params": [i]}
              for i in range(len(ref_order))]
    d2_sched = {"milestones": [120000, 160000], "gamma": 0.1, "warmup_factor": 0.001, "warmup_iters": 1000, "warmup_method": "linear",
                "base_lrs": [0.01] * len(ref_order), "last_epoch": 130000, "_step_count": 130001,
                "_get_lr_called_within_step": False, "_last_lr": [0.001] * len(ref_order)}
    sd = tr.state_dict()
    sd["stu_optimizer"] = {"state": state, "param_groups": groups}
    sd["stu_scheduler"], sd["tea_scheduler"] = dict(d2_sched), dict(d2_sched)
    sd["iteration"] = 129999
    tr.load_state_dict(sd)
    for i, (n, p) in enumerate(ref_order):
        buf = tr.stu_optimizer.state[p]["momentum_buffer"]
        assert float(buf.flatten()[0]) == float(i), n
    assert tr.stu_scheduler.last_epoch == 130000 and tr.iteration == 130000
    assert abs(tr.stu_optimizer.param_groups[0]["lr"] - 0.001) < 1e-12   # past the first milestone: base_lr * gamma
    tr.stu_optimizer.step()
    tr.stu_scheduler.step()
    assert tr.stu_scheduler.last_epoch == 130001 and abs(tr.stu_scheduler.get_last_lr()[0] - 0.001) < 1e-12


def test_reference_checkpoint_loader(tmp_path):
    """a detectron2 DetectionCheckpointer file written under the reference's key names (section 8c):
    alias copies of the FPN, `module.` prefix, persisted pixel/anchor buffers, numpy payloads; and a Caffe2-named ImageNet
    backbone pickle (`MODEL.WEIGHTS: detectron2://ImageNetPretrained/MSRA/R-50.pkl`)."""
    import pickle
    import numpy as np
    from lgd_amd.checkpoint import load_checkpoint, load_model_state
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    cfg = config.setup_cfg(os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cpu"])
    torch.manual_seed(1)
    src = build_model(cfg)
    tr = Trainer(cfg, src, device=torch.device("cpu"), distributed=False)
    tr.iteration = 1235
    sd = tr.state_dict()
    model_sd = {}
    for i, (k, v) in enumerate(sd["model"].items()):
        model_sd["module." + k] = v.numpy().copy() if i % 7 == 0 else v.clone()
    model_sd["module.student.pixel_mean"] = torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)
    model_sd["module.student.pixel_std"] = torch.ones(3, 1, 1)
    for i in range(5):
        model_sd["module.student.anchor_generator.cell_anchors.%d" % i] = torch.zeros(9, 4)
    model_sd["module.student.some_new_head.weight"] = torch.zeros(3)
    f = tmp_path / "model_0001234.pth"
    torch.save({**sd, "model": model_sd}, str(f))
    torch.manual_seed(2)
    dst = build_model(cfg)
    tr2 = Trainer(cfg, dst, device=torch.device("cpu"), distributed=False)
    rep = load_checkpoint(str(f), dst, tr2, resume=True)
    assert rep.missing == [] and rep.unexpected == ["student.some_new_head.weight"] and len(rep.ignored) == 7
    assert rep.alias_conflicts == [] and tr2.iteration == 1235
    for (n, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), n
    # the two FPN names are one module here: loading either copy fills both; a file whose copies disagree is reported
    only_fpn = {k: v for k, v in sd["model"].items() if not k.startswith("student.backbone.")}
    rep = load_model_state(build_model(cfg), only_fpn)
    assert rep.missing == []
    bad = dict(sd["model"])
    bad["student.fpn.fpn_lateral3.bias"] = bad["student.fpn.fpn_lateral3.bias"] + 1.0
    assert load_model_state(build_model(cfg), bad).alias_conflicts == ["student.fpn.fpn_lateral3.bias"]
    with pytest.raises(ValueError):
        load_model_state(build_model(cfg), {"teacher.local_inst_proj_1D.weight": torch.zeros(3, 3)})
    # Caffe2-named backbone pickle
    bb = {k[len("student.raw_backbone."):]: v for k, v in sd["model"].items() if k.startswith("student.raw_backbone.")}
    c2 = {"fc1000_w": np.zeros((1000, 2048), np.float32), "fc1000_b": np.zeros(1000, np.float32)}
    for k, v in bb.items():
        if k.endswith(("running_mean", "running_var")) or "conv2_offset" in k:
            continue
        n = k.replace(".norm.weight", "_bn_s").replace(".norm.bias", "_bn_b").replace(".weight", "_w").replace(".bias", "_b")
        n = n.replace("stem.conv1_bn", "res_conv1_bn").replace("stem.conv1", "conv1")
        for d2, cc in (("shortcut", "branch1"), ("conv1", "branch2a"), ("conv2", "branch2b"), ("conv3", "branch2c")):
            n = n.replace("." + d2, "_" + cc) if n.startswith("res") else n
        n = n.replace(".", "_", 1) if n.startswith("res") and not n.startswith("res_conv1") else n
        c2[n] = v.numpy()
    pk = tmp_path / "R-50.pkl"
    with open(str(pk), "wb") as fh:
        pickle.dump({"model": c2, "__author__": "Caffe2", "matching_heuristics": True}, fh)
    torch.manual_seed(3)
    fresh = build_model(cfg)
    rep = load_checkpoint(str(pk), fresh)
    assert rep.unexpected == [] and len(rep.loaded) == len(c2) - 2
    assert torch.equal(fresh.state_dict()["student.raw_backbone.res4.5.conv3.weight"], sd["model"]["student.raw_backbone.res4.5.conv3.weight"])
    assert torch.equal(fresh.state_dict()["student.raw_backbone.stem.conv1.norm.bias"], sd["model"]["student.raw_backbone.stem.conv1.norm.bias"])
    assert all(not k.startswith("student.raw_backbone.") or k.endswith(("running_mean", "running_var")) for k in rep.missing)


def test_clip_and_finite_flag_cpu():
    """per-parameter gradient clipping (d2 maybe_add_gradient_clipping, 'norm' type) and the sync-free non-finite flag
    (the reference asserts isfinite every iteration, train.py:194)."""
    from types import SimpleNamespace
    from lgd_amd.engine import Trainer
    t = Trainer.__new__(Trainer)
    a, b = torch.nn.Parameter(torch.zeros(4)), torch.nn.Parameter(torch.zeros(2))
    a.grad, b.grad = torch.full((4,), 3.0), torch.tensor([0.1, 0.0])
    t.stu_optimizer, t.tea_optimizer = torch.optim.SGD([a], 0.1), torch.optim.SGD([b], 0.1)
    t.clip = SimpleNamespace(CLIP_TYPE="norm", CLIP_VALUE=1.0, NORM_TYPE=2.0)
    t._clip()
    assert abs(float(a.grad.norm()) - 1.0) < 1e-5 and torch.equal(b.grad, torch.tensor([0.1, 0.0]))  # each on its own
    t.clip = SimpleNamespace(CLIP_TYPE="value", CLIP_VALUE=0.05, NORM_TYPE=2.0)
    t._clip()
    assert float(b.grad.max()) == pytest.approx(0.05)
    t.distributed, t.iteration = False, 7
    t._finite = torch.tensor(True) & torch.isfinite(torch.tensor(float("nan")))
    with pytest.raises(FloatingPointError):
        t.check_finite()
    t._finite = torch.tensor(True)
    t.check_finite()


def test_synthetic_batch_format():
    from lgd_amd.data import synthetic_batch
    b = synthetic_batch(2, 64, 96, 3, seed=1)
    assert len(b) == 2 and b[0]["image"].shape == (3, 64, 96) and len(b[0]["instances"]) == 3
    assert b[0]["instances"].gt_boxes.tensor.shape == (3, 4) and b[0]["instances"].gt_classes.dtype == torch.int64
    again = synthetic_batch(2, 64, 96, 3, seed=1)
    assert torch.equal(b[1]["image"], again[1]["image"])  # hash-based: reproducible everywhere


def test_retinanet_anchors_and_student_restatements_cpu():
    """student side (parity unpinned): anchor grid of the product, and the restatements in
    oracle/student_oracle.py (what the HIP kernels are held to) against hand-computed cases."""
    from lgd_amd.student.retinanet import AnchorGenerator, apply_deltas, pairwise_iou
    from oracle import student_oracle as SO
    ag = AnchorGenerator([[32, 40.3, 50.8]], [[0.5, 1.0, 2.0]], [8])
    a = ag([torch.zeros(1, 1, 4, 6)])[0]
    assert a.shape == (4 * 6 * 9, 4)
    assert torch.allclose(a[0], torch.tensor([-22.6274, -11.3137, 22.6274, 11.3137]), atol=1e-3)  # size 32, ratio .5 at (0,0)
    assert torch.allclose(a[9, :2] - a[0, :2], torch.tensor([8.0, 0.0]))                           # next cell along x
    gt = torch.tensor([[4.0, 4.0, 40.0, 30.0]])
    d = SO.box2box_deltas(a, gt.expand_as(a))
    assert torch.allclose(apply_deltas(d, a), gt.expand_as(a), atol=1e-3)
    assert abs(float(pairwise_iou(gt, gt)) - 1.0) < 1e-6 and abs(float(SO.pairwise_iou(gt, gt)) - 1.0) < 1e-6
    # focal: explicit one-hot formula
    logits = torch.randn(2, 7, 5)
    labels = torch.tensor([[0, 5, 5, -1, 2, 5, 5], [5, 5, 5, 5, 5, 5, 4]])
    got = SO.sigmoid_focal_sum(logits, labels, 5, 0.25, 2.0)
    t = torch.nn.functional.one_hot(labels.clamp(min=0), 6)[..., :5].float()
    p = torch.sigmoid(logits)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(logits, t, reduction="none")
    ref = ((0.25 * t + 0.75 * (1 - t)) * ce * (1 - (p * t + (1 - p) * (1 - t))) ** 2)[labels >= 0].sum()
    assert abs(float(got - ref)) < 1e-5
    # matcher: IoU 1.0 -> positive; 0.45 -> ignore; 0.1 -> background, unless it is some GT's best anchor (low-quality match)
    anchors = torch.tensor([[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 4.5], [100.0, 100.0, 110.0, 110.0], [50.0, 50.0, 60.0, 60.0]])
    g = [(torch.tensor([[0.0, 0.0, 10.0, 10.0], [50.0, 50.0, 60.0, 51.0]]), torch.tensor([3, 9]))]
    lab, mb = SO.label_anchors(anchors, g, 80)
    assert lab[0].tolist() == [3, -1, 80, 9]  # anchor 3 has IoU 0.1 with GT 1 but is its best anchor
    assert torch.equal(mb[0][0], g[0][0][0]) and torch.equal(mb[0][3], g[0][0][1])
    lab, mb = SO.label_anchors(anchors, [(torch.zeros(0, 4), torch.zeros(0, dtype=torch.int64))], 80)
    assert lab[0].tolist() == [80] * 4 and float(mb[0].abs().max()) == 0.0
    # box-regression loss: one positive anchor, |pred - target| summed over 4 coordinates
    deltas = torch.zeros(1, 4, 4)
    tgt = SO.box2box_deltas(anchors[0], torch.tensor([1.0, 1.0, 9.0, 11.0]))
    loss = SO.box_reg_sum(deltas, torch.tensor([[3, 80, 80, -1]]), anchors, torch.tensor([[[1.0, 1.0, 9.0, 11.0]] * 4]), 80, 0.0)
    assert abs(float(loss) - float(tgt.abs().sum())) < 1e-6


def test_fcos_target_restatement_cpu():
    """oracle/student_oracle.py::fcos_targets on hand cases [ref: thirdparty_heads/fcos.py:177-284]."""
    from oracle import student_oracle as SO
    shifts = [torch.tensor([[4.0, 4.0], [12.0, 4.0], [4.0, 12.0], [12.0, 12.0]]), torch.tensor([[8.0, 8.0]])]
    big = torch.tensor([[0.0, 0.0, 16.0, 16.0]])
    # level 0 accepts max-ltrb in [-1, 64]: all four locations are inside the centre-sampling box (8 +- 12, clipped to the box)
    c, d, t = SO.fcos_targets(shifts, [8, 16], [[-1, 64], [64, 1e9]], [(big, torch.tensor([7]))], 80, 1.5)
    assert c[0].tolist() == [7, 7, 7, 7, 80]       # level 1 needs max-ltrb >= 64
    assert d[0, 0].tolist() == [4.0, 4.0, 12.0, 12.0] and abs(float(t[0, 0]) - 1.0 / 3.0) < 1e-6
    # a smaller box covering location 0 wins it (min area); radius 0 = anywhere strictly inside the box
    two = torch.tensor([[0.0, 0.0, 16.0, 16.0], [2.0, 2.0, 7.0, 7.0]])
    c, d, _ = SO.fcos_targets(shifts, [8, 16], [[-1, 64], [64, 1e9]], [(two, torch.tensor([7, 9]))], 80, 0.0)
    assert c[0].tolist() == [9, 7, 7, 7, 80] and d[0, 0].tolist() == [2.0, 2.0, 3.0, 3.0]
    c, d, t = SO.fcos_targets(shifts, [8, 16], [[-1, 64], [64, 1e9]], [(torch.zeros(0, 4), torch.zeros(0, dtype=torch.int64))], 80, 1.5)
    assert c[0].tolist() == [80] * 5 and float(d.abs().max()) == 0.0 and float(t.abs().max()) == 0.0


def test_modulated_deform_conv_matches_definition():
    """DCNv2 (BASELINE config 5): zero offsets + unit mask == plain conv; random offsets == direct bilinear sampling."""
    import torch.nn.functional as F
    from oracle.student_oracle import modulated_deform_conv2d
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


_WINO_BT = {4: [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                [0, 4, 0, -5, 0, 1]],
            6: [[1, 0, -21 / 4, 0, 21 / 4, 0, -1, 0], [0, 1, 1, -17 / 4, -17 / 4, 1, 1, 0], [0, -1, 1, 17 / 4, -17 / 4, -1, 1, 0],
                [0, .5, .25, -2.5, -1.25, 2, 1, 0], [0, -.5, .25, 2.5, -1.25, -2, 1, 0], [0, 2, 4, -2.5, -5, .5, 1, 0],
                [0, -2, 4, 2.5, -5, -.5, 1, 0], [0, -1, 0, 21 / 4, 0, -21 / 4, 0, 1]]}
_WINO_AT = {4: [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]],
            6: [[1, 1, 1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, .5, -.5, 0], [0, 1, 1, 4, 4, .25, .25, 0], [0, 1, -1, 8, -8, .125, -.125, 0],
                [0, 1, 1, 16, 16, 1 / 16, 1 / 16, 0], [0, 1, -1, 32, -32, 1 / 32, -1 / 32, 1]]}


@pytest.mark.parametrize("tile", [4, 6])
def test_winograd_matrices_define_the_convolution(tile):
    """host side of K8: the filter-transform matrix G (ops._WINO_G) together with the B^T / A^T the transforms hard-code (restated
    here: csrc/winograd.hip bt6 / at6, csrc/winograd6.hip bt8 / at8) satisfies the minimal-filtering identity
    A^T [ (G g G^T) * (B^T d B) ] A == valid 3x3 correlation of the (tile+2)^2 window, in fp64."""
    from lgd_amd import ops
    BT, AT = torch.tensor(_WINO_BT[tile], dtype=torch.float64), torch.tensor(_WINO_AT[tile], dtype=torch.float64)
    G = torch.tensor(ops._WINO_G[tile], dtype=torch.float64)
    n = tile + 2
    gen = torch.Generator().manual_seed(3)
    d = torch.randn(n, n, dtype=torch.float64, generator=gen)
    g = torch.randn(3, 3, dtype=torch.float64, generator=gen)
    U = G @ g @ G.t()
    y = AT @ (U * (BT @ d @ BT.t())) @ AT.t()
    ref = torch.nn.functional.conv2d(d[None, None], g[None, None])[0, 0]
    assert torch.allclose(y, ref, atol=1e-10)
    # the adjoint used for the weight gradient: frequency (1,1) of A g A^T is the plain sum of the tile's gradient
    gy = torch.randn(tile, tile, dtype=torch.float64, generator=gen)
    dM = AT.t() @ gy @ AT
    assert abs(float(dM[1, 1] - gy.sum())) < 1e-12
    assert 1 * n + 1 == tile + 3  # ... which ops._Conv3x3K.backward indexes as dM[tile + 3]
    # what the gather form of the adjoint input transform rests on: window row / column 0 depends on frequency 0 only, n-1 on n-1 only
    assert [float(v) for v in BT[:, 0]] == [float(BT[0, 0])] + [0.0] * (n - 1)
    assert [float(v) for v in BT[:, n - 1]] == [0.0] * (n - 1) + [1.0]


def test_winograd6_adjoint_input_transform_as_a_gather():
    """The identity lgd_wino_in_t (tile = 6) is built on (winograd6.hip: wino6_in_t_phase / vacc8): the adjoint of the F(6x6,3x3) input
    transform -- overlap-add of the 8x8 windows Z_t = B G_t B^T at stride 6 -- equals, per 6x6 block, a GATHER of the tile's own 64
    values, 8 of each edge neighbour and 1 of each corner, evaluated ROW FIRST: every frequency row a goes through the horizontal
    transform h = (B g_a)[1..6] (+ the left tile's G[a][7], + the right tile's G[a][0]) and is then accumulated with B^T[a][r+1]."""
    import numpy as np
    BT = np.array(_WINO_BT[6], dtype=np.float64)
    B = BT.T
    rng = np.random.default_rng(0)
    TH, TW, H, W = 3, 4, 16, 21
    G = rng.standard_normal((TH, TW, 8, 8))
    dx = np.zeros((6 * TH + 2, 6 * TW + 2))
    for ty in range(TH):
        for tx in range(TW):
            dx[6 * ty:6 * ty + 8, 6 * tx:6 * tx + 8] += B @ G[ty, tx] @ B.T     # window (ty, tx) starts at pixel (6 ty - 1, 6 tx - 1)
    ref = dx[1:1 + H, 1:1 + W]

    def mid(g):   # entries 1..6 of B g (the kernel's b8mid)
        d12, s12, d34, s34, d56, s56 = g[1] - g[2], g[1] + g[2], g[3] - g[4], g[3] + g[4], g[5] - g[6], g[5] + g[6]
        return np.array([d12 + .5 * d34 + 2 * d56 - g[7], -5.25 * g[0] + s12 + .25 * s34 + 4 * s56,
                         -4.25 * d12 - 2.5 * (d34 + d56) + 5.25 * g[7], 5.25 * g[0] - 4.25 * s12 - 1.25 * s34 - 5 * s56,
                         d12 + 2 * d34 + .5 * d56 - 5.25 * g[7], s12 + s34 + s56 - g[0]])
    for g in rng.standard_normal((3, 8)):
        assert np.abs(mid(g) - (B @ g)[1:7]).max() < 1e-12
    C = BT[:, 1:7]   # vacc8's coefficient table: B^T[a][r+1]
    out = np.zeros((6 * TH, 6 * TW))
    for ty in range(TH):
        for tx in range(TW):
            up, dn, lf, rt = ty > 0, ty < TH - 1, tx > 0, tx < TW - 1

            def hrow(t_y, a):   # horizontal transform of frequency row a of tile (t_y, tx) incl. its left / right neighbours' columns
                h = mid(G[t_y, tx][a])
                if lf:
                    h[0] += G[t_y, tx - 1][a, 7]
                if rt:
                    h[5] += G[t_y, tx + 1][a, 0]
                return h
            z = np.zeros((6, 6))
            for a in range(8):
                z += np.outer(C[a], hrow(ty, a))
            if dn:
                z[5] += hrow(ty + 1, 0)   # the lower tile's window row 0 = its frequency row 0
            if up:
                z[0] += hrow(ty - 1, 7)   # the upper tile's window row 7 = its frequency row 7
            out[6 * ty:6 * ty + 6, 6 * tx:6 * tx + 6] = z
    assert np.abs(out[:H, :W] - ref).max() < 1e-12


def test_winograd_adjoint_input_transform_as_a_gather():
    """The identity lgd_wino_in_t is built on (winograd.hip: wino4_in_t): the adjoint of the F(4x4,3x3) input transform -- overlap-add
    of the 6x6 windows Z_t = B G_t B^T at stride 4 -- equals, per 4x4 block, a GATHER of the tile's own 36 values, 6 values of each edge
    neighbour (frequency row / column 0 or 5 only, because B's first row is [4 0 0 0 0 0] and its last [0 0 0 0 0 1]) and 1 of each corner."""
    import numpy as np
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                   [0, 4, 0, -5, 0, 1]], dtype=np.float64)
    B = BT.T
    assert list(B[0]) == [4, 0, 0, 0, 0, 0] and list(B[5]) == [0, 0, 0, 0, 0, 1]
    rng = np.random.default_rng(0)
    TH, TW, H, W = 3, 4, 11, 14
    G = rng.standard_normal((TH, TW, 6, 6))
    dx = np.zeros((4 * TH + 2, 4 * TW + 2))
    for ty in range(TH):
        for tx in range(TW):
            dx[4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6] += B @ G[ty, tx] @ B.T     # window (ty, tx) starts at pixel (4 ty - 1, 4 tx - 1)
    ref = dx[1:1 + H, 1:1 + W]

    def mid(g):   # rows 1..4 of B g (the kernel's b6mid)
        return np.array([4 * (g[2] - g[1]) + 2 * (g[4] - g[3]) + 4 * g[5], -5 * g[0] - 4 * (g[1] + g[2]) - (g[3] + g[4]),
                         (g[1] - g[2]) + 2 * (g[3] - g[4]) - 5 * g[5], g[0] + g[1] + g[2] + g[3] + g[4]])
    out = np.zeros((4 * TH, 4 * TW))
    for ty in range(TH):
        for tx in range(TW):
            g = G[ty, tx]
            up, dn, lf, rt = ty > 0, ty < TH - 1, tx > 0, tx < TW - 1
            t = np.stack([mid(g[:, b]) for b in range(6)], 1)             # (4, 6): block rows x frequency columns
            if up:
                t[0] += G[ty - 1, tx][5]                                   # the upper tile's window row 5 = its frequency row 5
            if dn:
                t[3] += 4 * G[ty + 1, tx][0]                               # the lower tile's window row 0 = 4 x its frequency row 0
            tl = mid(G[ty, tx - 1][:, 5]) if lf else np.zeros(4)
            tr = mid(G[ty, tx + 1][:, 0]) if rt else np.zeros(4)
            if up and lf:
                tl[0] += G[ty - 1, tx - 1][5, 5]
            if dn and lf:
                tl[3] += 4 * G[ty + 1, tx - 1][0, 5]
            if up and rt:
                tr[0] += G[ty - 1, tx + 1][5, 0]
            if dn and rt:
                tr[3] += 4 * G[ty + 1, tx + 1][0, 0]
            for r in range(4):
                y = mid(t[r])
                y[0] += tl[r]
                y[3] += 4 * tr[r]
                out[4 * ty + r, 4 * tx:4 * tx + 4] = y
    assert np.abs(out[:H, :W] - ref).max() < 1e-12


# ------------------------------------------------------------------------------------------------ one process per GPU: launch + host-thread slices
def test_host_thread_slices_partition_the_allowed_cpus():
    """lgd_amd/launch.py: eight ranks of one node get disjoint, equal, contiguous slices of the allowed CPUs; with topology, the ranks whose GPUs
    hang off one NUMA node share THAT node's CPUs [ref: train.py:296-310 one process per GPU]."""
    from lgd_amd import launch
    allowed = set(range(256))
    got = [launch.plan_affinity(r, 8, allowed) for r in range(8)]
    assert all(len(g) == 32 for g in got) and sorted(c for g in got for c in g) == list(range(256))
    assert got[3] == list(range(96, 128))
    # two sockets: CPUs 0-63 + 128-191 on node 0, 64-127 + 192-255 on node 1; GPUs 0-3 on node 0, 4-7 on node 1
    cpus_of_node = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    node_of_rank = {r: r // 4 for r in range(8)}
    got = [launch.plan_affinity(r, 8, allowed, node_of_rank, cpus_of_node) for r in range(8)]
    assert all(len(g) == 32 for g in got) and sorted(c for g in got for c in g) == list(range(256))
    assert all(set(got[r]) <= set(cpus_of_node[r // 4]) for r in range(8))
    # a cgroup that allows fewer CPUs than ranks: every rank keeps the whole set instead of an empty one
    assert launch.plan_affinity(5, 8, {3, 4}) == [3, 4]
    # unknown topology for this rank: equal slices
    assert launch.plan_affinity(1, 2, set(range(8)), {0: None, 1: None}, {}) == [4, 5, 6, 7]
    assert launch._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_bench_and_train_self_launch_command(monkeypatch):
    """`python bench.py --gpus 4` / `python train.py --num-gpus 4` started without a launcher re-execute as 4 ranks of one node on 127.0.0.1;
    under a launcher (RANK / WORLD_SIZE present) they do not."""
    import importlib.util
    import subprocess
    from lgd_amd import launch
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert not launch.launched()
    rc = launch.self_launch("/x/bench.py", 4, ["--gpus", "4", "--steps", "3"])
    assert rc == 7
    c = seen["cmd"]
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and c[c.index("--nproc-per-node") + 1] == "4"
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "4")
    assert launch.launched()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for name, fn, flag in (("bench.py", "_gpus_requested", "--gpus"), ("train.py", "_num_gpus", "--num-gpus")):
        spec = importlib.util.spec_from_file_location("m_" + name[:-3], os.path.join(root, name))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)     # not __main__: importing must not launch anything
        f = getattr(mod, fn)
        assert f([flag, "8", "--x"]) == 8 and f([flag + "=2"]) == 2 and f(["--steps", "3"]) == 1


def test_side_streams_switches(monkeypatch):
    """ops.side_streams_ok(): LGD_SIDE_STREAMS=0 switches every fork off; with a stream PER fork (LGD_ONE_SIDE_STREAM=0) the forks are also off above HIP's
    default 4 hardware queues (round 6: cross-queue waits stall there, profiles/r06_hw_queues_and_forks.txt), LGD_SIDE_STREAMS=force overrides; the shipped form
    -- one shared side stream -- does not depend on the queue count."""
    from lgd_amd import ops, streams
    for k in ("LGD_SIDE_STREAMS", "GPU_MAX_HW_QUEUES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(streams, "_ONE_SIDE", True)
    assert ops.side_streams_ok()
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert ops.side_streams_ok()
    monkeypatch.setattr(streams, "_ONE_SIDE", False)
    assert not ops.side_streams_ok()
    monkeypatch.setenv("LGD_SIDE_STREAMS", "force")
    assert ops.side_streams_ok()
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "4")
    monkeypatch.setenv("LGD_SIDE_STREAMS", "1")
    assert ops.side_streams_ok()
    monkeypatch.setenv("LGD_SIDE_STREAMS", "0")
    assert not ops.side_streams_ok()
    monkeypatch.setattr(streams, "_ONE_SIDE", True)
    assert not ops.side_streams_ok()
