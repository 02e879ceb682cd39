"""GPU parity of the PRODUCT modules (lgd_amd.DynamicTeacher / BaseDistillator.distill / meta-archs) against
the reference golden vectors and the CPU oracle.  Bar: 1e-4 relative (BASELINE.json north_star)."""
import os
import time

import numpy as np
import pytest
import torch

import common as cm
from lgd_amd import synth
from oracle import lgd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _cfg(ctx, interact, fmt, coef=1.0, student="RetinaNetCT", meta="RetinaNet"):
    from lgd_amd import config
    return config.setup_cfg(None, [
        "MODEL.DEVICE", DEV, "MODEL.META_ARCHITECTURE", meta,
        "MODEL.DISTILLATOR.STUDENT.META_ARCH", student, "MODEL.DISTILLATOR.TEACHER.META_ARCH", "DynamicTeacher",
        "MODEL.DISTILLATOR.TEACHER.ADD_CONTEXT_BOX", str(ctx), "MODEL.DISTILLATOR.TEACHER.INTERACT_PATTERN", interact,
        "MODEL.DISTILLATOR.LABEL_ENCODER.BOX_FORMAT", fmt, "MODEL.DISTILLATOR.LAMBDA", str(coef)])


def _batched_inputs(gt, H, W):
    from lgd_amd.structures import Boxes, Instances
    return [{"image": torch.zeros(3, H, W), "instances": Instances((H, W), gt_boxes=Boxes(b.clone()), gt_classes=c.clone())}
            for b, c in gt]


def _teacher(name):
    from lgd_amd.dynamic_teacher import DynamicTeacher
    B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
    t = DynamicTeacher(_cfg(ctx, interact, fmt, coef))
    # strict load under the REFERENCE's state_dict names (tests/golden/make_golden.py loads the same dict into the reference)
    missing, unexpected = t.load_state_dict({k: v for k, v in cm.teacher_params().items()}, strict=True)
    assert not missing and not unexpected
    return t.to(DEV).train()


# every golden case on the Winograd kernels (forced: the small cases have fewer tiles than the production threshold), the
# small ones once more on the library convolutions (ops.conv3x3_backend(winograd=False))
# round 5: every case once more with ALL channel products (forward, input and weight gradient) forced onto csrc/h2.hip (f16x2 split operands: the
# shipped path at production sizes), and once with the f16x2 pipeline off and the products forced onto csrc/gemm3.hip
_RUNS = ([(n, "winograd") for n in cm.CASES] + [(n, "winograd+h2") for n in cm.CASES] + [(n, "winograd+gemm3") for n in cm.CASES]
         + [(n, "library") for n in cm.SMALL_CASES])


@pytest.fixture(scope="module", params=_RUNS, ids=["%s-%s" % r for r in _RUNS])
def run(request):
    from lgd_amd import ops
    from lgd_amd.structures import ImageList
    name, backend = request.param
    prev = ops.conv3x3_backend(winograd=backend.startswith("winograd"), min_tiles=0)
    prev_h2 = ops.h2_backend(backend != "winograd+gemm3", force=(backend == "winograd+h2"))
    prev_g3 = ops.gemm3_backend(True, force=(backend == "winograd+gemm3"))
    B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
    teacher = _teacher(name)
    feats = {k: v.to(DEV).requires_grad_(True) for k, v in cm.case_feats(name).items()}
    images = ImageList(torch.zeros(B, 3, H, W, device=DEV), [(H, W)] * B)
    bi = _batched_inputs(cm.case_gt(name), H, W)
    cap = {}
    h = teacher.label_encoder_.register_forward_hook(lambda m, i, o: cap.__setitem__("le", o))
    # the same taps tests/golden/make_golden.py puts on the reference: output of aggregate_per_level and of the MHA
    real_pool, real_mha = ops.gn_relu_mask_pool, ops.mha_blockdiag
    ops.gn_relu_mask_pool = lambda *a, **k: cap.setdefault("app", real_pool(*a, **k))
    ops.mha_blockdiag = lambda *a, **k: cap.setdefault("att", real_mha(*a, **k))
    try:
        tea, inst_labels, geom = teacher((bi, images, None, feats))
    finally:
        ops.gn_relu_mask_pool, ops.mha_blockdiag = real_pool, real_mha
        h.remove()
    yield dict(name=name, backend=backend, g=cm.golden(name), teacher=teacher, feats=feats, tea=tea, geom=geom, le=cap["le"],
               coef=coef, inst_labels=inst_labels, app=cap.get("app"), att=cap.get("att"), stride=cm.stride_of(name))
    ops.gemm3_backend(*prev_g3)
    ops.h2_backend(*prev_h2)
    ops.conv3x3_backend(*prev)


def test_label_encoder_and_geometry(run):
    g = run["g"]
    embed, m1, m2, boxes, _, _, counts = run["le"]
    assert np.array_equal(np.array(counts, np.int32), g["counts"])
    assert np.array_equal(boxes.cpu().double().numpy(), g["boxlists"])  # clamped boxes: bit-exact
    assert cm.rel_err(embed, g["label_embed"]) < TOL
    assert cm.rel_err(m1.reshape(-1)[::cm.SAMPLE_STRIDE], g["stn_desc_sample"]) < TOL
    assert cm.rel_err(m2.reshape(-1)[::cm.SAMPLE_STRIDE], g["stn_feat_sample"]) < TOL
    rects = run["geom"].rects().cpu().numpy()
    for i, k in enumerate(O.LEVELS):
        ref = g["rects_" + k].copy()
        ref[(ref[:, 0] > ref[:, 1]) | (ref[:, 2] > ref[:, 3])] = [0, -1, 0, -1]
        assert np.array_equal(rects[i], ref), k
    lab = torch.cat([l.reshape(-1) for l in run["inst_labels"]]).cpu().numpy().astype(np.int64)
    assert np.array_equal(lab, g["inst_labels"])


def test_appearance_and_attention_match_reference(run):
    """the HIP mask pooling (fused GN+ReLU+pool) and block-diagonal MHA against the reference's own intermediate tensors
    (golden app_* = aggregate_per_level output, attn_* = nn.MultiheadAttention output, per level)."""
    g = run["g"]
    assert run["app"] is not None and run["att"] is not None
    for i, k in enumerate(O.LEVELS):
        assert cm.rel_err(run["app"][i], g["app_" + k]) < TOL, k
        assert cm.rel_err(run["att"][i], g["attn_" + k]) < TOL, k


def test_teacher_features_match_reference(run):
    g, tea = run["g"], run["tea"]
    errs = {k: cm.rel_err(cm.sample(tea[k], run["stride"])[0], g["tea_s_" + k]) for k in O.LEVELS}
    print("teacher features vs the reference's golden [%s]: %s (bar %.0e)" % (run["name"] + " / " + str(run["backend"]), " ".join("%s %.1e" % kv for kv in errs.items()), TOL))
    for k in O.LEVELS:
        s, _, sq = cm.sample(tea[k], run["stride"])
        assert cm.rel_err(s, g["tea_s_" + k]) < TOL, k
        assert abs(sq - float(g["tea_sq_" + k])) / float(g["tea_sq_" + k]) < 2 * TOL, k
        if k in ("p6", "p7") and "tea_full_" + k in g:
            assert cm.rel_err(tea[k], g["tea_full_" + k]) < TOL, k


def test_distill_loss_and_grads_match_reference(run):
    g = run["g"]
    if "total_loss" not in g:
        pytest.skip("case stored without grads")
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.base_distillator import BaseDistillator

    class D(BaseDistillator):  # distill() only needs coef / adapter / distill_flag
        def __init__(self, coef):
            torch.nn.Module.__init__(self)
            self.coef = coef
            self.adapter = torch.nn.ModuleDict({"distill": SequentialConvs(None)})
    d = D(run["coef"])
    d.adapter["distill"].load_state_dict(cm.adapter_params(), strict=True)
    d.to(DEV)
    tea, feats, teacher = run["tea"], run["feats"], run["teacher"]
    for flag in (0, 1):
        d.distill_flag = flag
        loss = d.distill({"stu": feats, "tea": tea}, None, None, None, None)
        ref = float(g["loss_distill_flag%d" % flag])
        assert abs(loss.item() - ref) / ref < TOL
    pr = cm.probes({k: tea[k] for k in O.LEVELS})
    total = loss + sum((tea[k] * pr[k].to(DEV)).sum() for k in O.LEVELS)
    assert abs(total.item() - float(g["total_loss"])) < TOL * abs(float(g["total_loss"])) + 1e-6
    total.backward()
    # Model-level gradients and ReLU kinks.  The CPU reference agrees with an fp64 evaluation to 2..4e-6 on every level (no
    # conditioning problem), but a pre-activation within rounding noise of zero takes its backward mask from the summation order
    # of whoever computed it: the ORACLE's own fp32 torch ops, run on this GPU, leave the reference by 1e-3 (p3) / 3..5e-3 (p5) on
    # case c1 and by ~5e-6 on the other levels (tools/diag_grad_chain.py, profiles/r02_diag_*); the HIP path shows the same on
    # whichever levels ITS rounding flips (a level of 64 x 64 x 256 units per ReLU layer almost surely has one within 1e-7 of zero,
    # the 100 x 168 levels of the full-size cases have dozens).  One flipped unit reaches 7x7x256 inputs through the refinement convs
    # (20 % of a 16x16 level), so no element-wise criterion survives it.  The HARD assert on the model-level gradients is therefore
    # test_teacher_gradients_fp64_under_product_masks below (fp64 oracle under the product's own masks: ALL levels and parameters to
    # 1e-4); here the deviation from the reference's fp32 gradients is printed next to that of torch's own ops on this device and
    # only bounded at the size a flip can have (2e-2).
    dev_feat, dev_w = cm.oracle_grads_on_device(run["name"], DEV)
    errs = {k: cm.rel_err(cm.sample(feats[k].grad, run["stride"])[0], g["gfeat_s_" + k]) for k in O.LEVELS}
    print("feature-gradient parity vs reference [%s]: %s" % (run["backend"], "; ".join(
        "%s product %.1e / torch-ops-on-this-GPU %.1e" % (k, errs[k], cm.rel_err(cm.sample(dev_feat[k], run["stride"])[0], g["gfeat_s_" + k])) for k in O.LEVELS)))
    # measured: <= 1.6e-3 (p3) .. 4e-3 (p5) on the 512 x 512 cases; the 800 x 1344 cases have dozens of near-zero units per level
    # (8.4e-3 at p3).  Bounds at ~2x the measured flip noise: they trip on regressions, not on flips
    bound = 5e-3 if run["name"] in cm.SMALL_CASES else 1.5e-2
    assert all(e <= bound for e in errs.values()), errs
    bound = max(bound, 1e-2)   # parameter gradients sum the flips of all levels: torch's own ops on this GPU leave the reference by 7.5e-3
    named = list(teacher.named_parameters()) + [("adapter." + n, p) for n, p in d.adapter["distill"].named_parameters()]
    worst = (0.0, 0.0, "")
    for n, prm in named:
        if "gnone_" + n in g:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, n
            continue
        ref_s, ref_sq = g["gw_s_" + n], float(g["gw_sq_" + n])
        if ref_sq <= 1e-10:  # adapter.4.bias sits right before an InstanceNorm: its gradient is analytically 0 (fp noise)
            assert cm.sample(prm.grad)[2] <= 1e-9, n
            continue
        # parameter gradients sum over all levels, the flipped ones included
        scale = float(np.abs(ref_s).max()) + 1e-12
        e_dev = float(np.abs(cm.sample(dev_w[n])[0][:64] - ref_s).max()) / scale
        e_prod = float(np.abs(cm.sample(prm.grad)[0][:64] - ref_s).max()) / scale
        worst = max(worst, (e_prod, e_dev, n))
        assert e_prod <= bound, (n, e_prod, e_dev)
        assert abs(cm.sample(prm.grad)[2] - ref_sq) <= 4e-2 * ref_sq, n
    print("weight-gradient parity: worst sampled deviation %.1e (torch ops on this GPU: %.1e) at %s" % worst)


@pytest.mark.parametrize("backend", ["winograd", "winograd+h2", "winograd+gemm3", "library"])
def test_margin_case_gradients_match_reference_to_1e_4(backend):
    """VERDICT r5 weak 1 -- the chain closes on the REFERENCE's own gradient vectors.  tests/golden/c4_margin.npz (make_golden_margin.py): a case
    whose every ReLU / LayerNorm-ReLU / GroupNorm-ReLU input and every per-image max-pool candidate keeps a margin of 2e-4 of its tensor's scale
    (biases nudged in fp64 until they do; re-checked on the real reference's fp32 run), so no backward mask depends on anybody's summation order.
    The PRODUCT teacher + adapter + distill on the HIP path -- Winograd kernels forced onto the small maps, once per product back-end, and the
    library's convolutions -- against the reference's outputs and gradients of  loss_distill + <teacher features, probe>  directly:
    teacher features, loss, ALL feature gradients (whole levels p4..p7, samples of p3) and ALL parameter gradients to 1e-4.
    [ref: base_distillator.py:34-64, dynamic_teacher.py:209-283]"""
    from lgd_amd import ops
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.base_distillator import BaseDistillator
    from lgd_amd.dynamic_teacher import DynamicTeacher
    from lgd_amd.structures import ImageList
    name = "c4_margin"
    B, H, W, ctx, interact, fmt, coef, _ = cm.MARGIN_CASES[name]
    g = cm.golden(name)
    prev = ops.conv3x3_backend(winograd=backend.startswith("winograd"), min_tiles=0)
    prev_h2 = ops.h2_backend(backend != "winograd+gemm3", force=(backend == "winograd+h2"))
    prev_g3 = ops.gemm3_backend(True, force=(backend == "winograd+gemm3"))
    try:
        t = DynamicTeacher(_cfg(ctx, interact, fmt, coef))
        missing, unexpected = t.load_state_dict(cm.teacher_params(case=name), strict=True)
        assert not missing and not unexpected
        t.to(DEV).train()

        class D(BaseDistillator):
            def __init__(self):
                torch.nn.Module.__init__(self)
                self.coef = coef
                self.adapter = torch.nn.ModuleDict({"distill": SequentialConvs(None)})
        d = D()
        d.adapter["distill"].load_state_dict(cm.adapter_params(case=name), strict=True)
        d.to(DEV)
        d.distill_flag = 1
        feats = {k: v.to(DEV).requires_grad_(True) for k, v in cm.case_feats(name).items()}
        images = ImageList(torch.zeros(B, 3, H, W, device=DEV), [(H, W)] * B)
        tea, _, _ = t((_batched_inputs(cm.case_gt(name), H, W), images, None, feats))
        loss = d.distill({"stu": feats, "tea": tea}, None, None, None, None)
        pr = cm.probes({k: tea[k] for k in O.LEVELS})
        total = loss + sum((tea[k] * pr[k].to(DEV)).sum() for k in O.LEVELS)
        total.backward()
    finally:
        ops.gemm3_backend(*prev_g3)
        ops.h2_backend(*prev_h2)
        ops.conv3x3_backend(*prev)
    assert abs(loss.item() - float(g["loss_distill_flag1"])) < TOL * float(g["loss_distill_flag1"])
    assert abs(total.item() - float(g["total_loss"])) < TOL * abs(float(g["total_loss"]))
    ef = {k: cm.rel_err(cm.sample(tea[k])[0], g["tea_s_" + k]) for k in O.LEVELS}
    eg = {}
    for k in O.LEVELS:
        eg[k] = cm.rel_err(cm.sample(feats[k].grad)[0], g["gfeat_s_" + k])
        if g["gfeat_" + k].size:
            eg[k] = max(eg[k], cm.rel_err(feats[k].grad, g["gfeat_" + k]))
    worst = (0.0, "")
    for n, prm in list(t.named_parameters()) + [("adapter." + n, q) for n, q in d.adapter["distill"].named_parameters()]:
        ref_s, ref_sq = g["gw_s_" + n], float(g["gw_sq_" + n])
        if ref_sq <= 1e-10:   # adapter.4.bias sits right before an InstanceNorm: analytically 0
            continue
        s, _, sq = cm.sample(prm.grad)
        e = float(np.abs(s[:256] - ref_s).max()) / float(np.abs(ref_s).max())
        worst = max(worst, (e, n))
        assert abs(sq - ref_sq) <= 4 * TOL * ref_sq, (n, sq, ref_sq)
    print("margin case [%s] vs the REFERENCE: teacher features %s | feature gradients %s | worst parameter gradient %.1e (%s)  (bar %.0e)" % (
        backend, " ".join("%s %.1e" % kv for kv in ef.items()), " ".join("%s %.1e" % kv for kv in eg.items()), worst[0], worst[1], TOL))
    assert all(e < TOL for e in ef.values()), ef
    assert all(e < TOL for e in eg.values()), eg
    assert worst[0] < TOL, worst


_RN_KEYS = {"loss_cls", "loss_box_reg", "loss_cls.tea", "loss_box_reg.tea", "loss_distill"}
_FC_KEYS = _RN_KEYS | {"loss_centerness", "loss_centerness.tea"}


@pytest.fixture
def conv_backend(request):
    """'winograd': winograd.hip forced onto the small test problems (production threshold: 500 tiles); 'library': ops.conv3x3_backend(winograd=False)."""
    from lgd_amd import ops
    prev = ops.conv3x3_backend(winograd=(request.param == "winograd"), min_tiles=0)
    yield request.param
    ops.conv3x3_backend(*prev)


@pytest.mark.parametrize("yaml_name,keys,conv_backend", [
    ("lgd_retinanet_r50", _RN_KEYS, "winograd"),
    ("lgd_retinanet_r50", _RN_KEYS, "library"),
    ("lgd_fcos_r50", _FC_KEYS, "winograd"),
    ("lgd_retinanet_r101", _RN_KEYS, "winograd"),        # BASELINE configs[3] (per-rank workload of the DDP 8x run)
    ("lgd_retinanet_r101_dcnv2", _RN_KEYS, "winograd"),  # BASELINE configs[4]
], indirect=["conv_backend"])
def test_meta_arch_train_step(yaml_name, keys, conv_backend):
    """one full training iteration (both optimizers) on a small synthetic batch; loss keys as in the reference
    [ref: distillator.py:66-68,110-112,292-295]."""
    import os
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", yaml_name + ".yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    model = build_model(cfg)
    tr = Trainer(cfg, model)
    data = synthetic_batch(2, 256, 320, 5, seed=1)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    for it in (0, 25000, 40000):  # frozen-backbone phase, non-distill phase, distill phase
        losses = tr.step(data, it)
        assert set(losses) == keys
    m = tr.fetch_metrics()
    assert all(np.isfinite(v) for v in m.values()), m
    changed = {n for n, p in model.named_parameters() if not torch.equal(p, before[n])}
    assert any(n.startswith("teacher.") for n in changed)
    assert any(n.startswith("adapter.") for n in changed)
    assert any(n.startswith("student.raw_backbone.res5") for n in changed)
    assert not any(n.startswith("student.raw_backbone.stem") or n.startswith("student.raw_backbone.res2") for n in changed)
    # multi-scale batch (INPUT.MIN_SIZE_TRAIN sampling): images of different sizes are padded to one tensor, the
    # teacher works in the padded frame [ref: label_encoder.py:167]
    ragged = synthetic_batch(1, 224, 288, 4, seed=3) + synthetic_batch(1, 256, 320, 6, seed=4)
    losses = tr.step(ragged, 40001)
    assert set(losses) == keys and all(np.isfinite(v) for v in tr.fetch_metrics().values())
    # eval branch (incl. teacher upper-bound probe) runs and returns one result per image
    model.eval()
    with torch.no_grad():
        out = model(data, eval_teacher=True)
    assert len(out) == 2 and "instances" in out[0]


def test_train_cli_checkpoint_resume(tmp_path):
    """train.py with the reference's CLI surface: trains a few synthetic iterations, writes metrics.json with the
    reference's scalar names and a checkpoint with the reference's keys, resumes from it, runs --eval-only."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "train.py"), "--config-file", os.path.join(root, "configs", "lgd_retinanet_r50.yaml"),
            "--num-gpus", "1", "--image-size", "256", "320"]
    opts = ["OUTPUT_DIR", str(tmp_path), "SOLVER.IMS_PER_BATCH", "2", "SOLVER.CHECKPOINT_PERIOD", "3",
            "MODEL.DISTILLATOR.PRE_NONDISTILL_ITERS", "2", "MODEL.DISTILLATOR.PRE_FREEZE_STUDENT_BACKBONE_ITERS", "1"]
    subprocess.check_call(base + ["--max-iter", "4"] + opts)
    ck = torch.load(os.path.join(str(tmp_path), "model_final.pth"), map_location="cpu")
    assert set(ck) == {"model", "stu_optimizer", "tea_optimizer", "stu_scheduler", "tea_scheduler", "iteration"} and ck["iteration"] == 3
    m = [json.loads(l) for l in open(os.path.join(str(tmp_path), "metrics.json"))]
    assert {"total_loss", "stu_lr", "tea_lr", "loss_distill", "loss_cls.tea"} <= set(m[-1])
    subprocess.check_call(base + ["--max-iter", "6", "--resume"] + opts)
    m2 = [json.loads(l) for l in open(os.path.join(str(tmp_path), "metrics.json"))]
    assert m2[-1]["iteration"] == 5 and len(m2) == len(m) + 1
    subprocess.check_call(base + ["--eval-only", "--resume"] + opts)


def test_teacher_edge_batch_vs_oracle():
    """ragged batch through the PRODUCT teacher vs the (reference-pinned) oracle: a crowded image (90 boxes: two
    64-box passes in the box kernels, attention tiling), an EMPTY-GT image (substitute unit box), a single-box image,
    non-square odd-sized pyramid (p5..p7 widths not multiples of 4)."""
    from lgd_amd.dynamic_teacher import DynamicTeacher
    from lgd_amd.structures import ImageList
    B, H, W = 3, 416, 608
    rng = np.random.default_rng(11)
    def boxes(n):
        x1, y1 = rng.uniform(0, W - 40, n), rng.uniform(0, H - 40, n)
        return torch.tensor(np.stack([x1, y1, np.minimum(W - 1, x1 + rng.uniform(4, 300, n)), np.minimum(H - 1, y1 + rng.uniform(4, 200, n))], 1),
                            dtype=torch.float32)
    gt = [(boxes(90), torch.from_numpy(rng.integers(0, 80, 90))), (torch.zeros(0, 4), torch.zeros(0, dtype=torch.int64)),
          (boxes(1), torch.tensor([17]))]
    feats = {k: torch.from_numpy(v) for k, v in synth.synth_features(B, H, W, seed=31).items()}
    for ctx in (True, False):
        t = DynamicTeacher(_cfg(ctx, "stuGuided", "x1y1x2y2"))
        t.load_state_dict(cm.teacher_params(), strict=True)
        t.to(DEV).train()
        images = ImageList(torch.zeros(B, 3, H, W, device=DEV), [(H, W)] * B)
        fg = {k: v.to(DEV).requires_grad_(True) for k, v in feats.items()}
        tea, labels, geom = t((_batched_inputs(gt, H, W), images, None, fg))
        assert geom.counts == ([91, 1, 2] if ctx else [90, 1, 1])
        fc = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
        ref, _, _ = O.teacher_forward(cm.teacher_params(), fc, gt, (H, W), ctx, "stuGuided")
        for k in O.LEVELS:
            assert cm.rel_err(tea[k], ref[k]) < TOL, (ctx, k)
        pr = cm.probes({k: ref[k] for k in O.LEVELS})  # (sum tea^2 would be constant: the last op is a GroupNorm)
        sum((tea[k] * pr[k].to(DEV)).sum() for k in O.LEVELS).backward()
        sum((ref[k] * pr[k]).sum() for k in O.LEVELS).backward()
        # kink flips: see test_distill_loss_and_grads_match_reference -- at least three levels to 1e-4, the rest to 2e-2
        errs = {k: cm.rel_err(fg[k].grad, fc[k].grad) for k in O.LEVELS}
        print("edge batch ctx=%s feature-gradient parity: %s" % (ctx, {k: "%.1e" % v for k, v in errs.items()}))
        assert sum(e <= 1e-4 for e in errs.values()) >= 3, (ctx, errs)
        assert all(e <= 2e-2 for e in errs.values()), (ctx, errs)


@pytest.mark.parametrize("B,ctx", [(8, True), (16, False)], ids=["config2-b8-ctx", "config3-b16-noctx"])
def test_full_size_properties(B, ctx):
    """size-independent properties at BASELINE config-2 size (B=8, 800x1344, C=256, context box) and at config 3's (B=16, no context
    box, T=160) where the oracle is too slow: distill(a, a) == 0 and is invariant to per-plane affine maps; GN(1) output has zero
    mean / unit variance per sample; box_sum is linear; the box sum of ones is the pixel count of the bit-exact rectangle."""
    from lgd_amd import ops
    H, W, C = 800, 1344, 256
    level_hw = synth.pyramid_shapes(H, W)
    g = torch.Generator(device=DEV).manual_seed(0)
    a = [torch.randn(B, C, h, w, device=DEV, generator=g) * 2 + 0.5 for h, w in level_hw]
    b = [torch.randn(B, C, h, w, device=DEV, generator=g) for h, w in level_hw]
    assert abs(ops.distill_in_mse(a, a, 1.0).item()) < 1e-6
    base = ops.distill_in_mse(a, b, 1.0).item()
    assert 1.9 < base < 2.1                                    # two independent unit-variance fields: E = 2
    scaled = ops.distill_in_mse([x * 3.0 - 7.0 for x in a], [y * 0.25 + 1.0 for y in b], 1.0).item()
    assert abs(scaled - base) / base < 1e-4                    # InstanceNorm removes per-plane affine maps
    ys = ops.gn1(a, False)
    for y in ys:
        m = y.double().mean(dim=(1, 2, 3))
        v = y.double().var(dim=(1, 2, 3), unbiased=False)
        assert float(m.abs().max()) < 1e-5 and float((v - 1).abs().max()) < 1e-4
    gt = synth.synth_gt(B, H, W, 10, seed=0)
    _, boxlists, _ = O.encode_box_descriptors([(torch.from_numpy(x), torch.from_numpy(c)) for x, c in gt], H, W, ctx)
    boxes = torch.tensor([r for bl in boxlists for r in bl], dtype=torch.float32).to(DEV)
    assert len(boxes) == B * (11 if ctx else 10)
    geom = ops.BoxGeometry(boxes, [len(bl) for bl in boxlists], (H, W), level_hw)
    s1, s2 = ops._box_sum(geom, a, False, False), ops._box_sum(geom, b, False, False)
    s12 = ops._box_sum(geom, [x * 2.0 - y for x, y in zip(a, b)], False, False)
    assert cm.rel_err(s12, 2.0 * s1 - s2) < 1e-5               # linearity of the box reduction
    ones = ops._box_sum(geom, [torch.ones_like(x) for x in a], False, False)
    r = geom.rects()
    cnt = torch.where(r[..., 1] >= r[..., 0], (r[..., 1] - r[..., 0] + 1) * (r[..., 3] - r[..., 2] + 1), torch.zeros_like(r[..., 0]))
    assert torch.equal(ones[..., 0], cnt.float())              # box sum of ones == pixel count of the bit-exact rectangle



@pytest.mark.timeout(900)
def test_config3_batch16_equals_two_batches_of_8():
    """BASELINE config 3 at ITS batch size: 16 images of 800x1344, no context box, 10 boxes each (T = 160) through the product
    DynamicTeacher + adapter + distill loss on the shipped Winograd path.  Nothing on the path mixes images (label-encoder pooling,
    block-diagonal attention, mask pooling / rendering, GroupNorm(1), InstanceNorm are all per image: SURVEY.md section 8e), so the
    teacher features of the batch of 16 must equal those of its two halves of 8 run on their own, the distill loss must be the mean
    of the halves' losses, and so must the gradients w.r.t. the student features (x 1/2: the loss is a mean over the batch)
    [ref: dynamic_teacher.py:209-283, base_distillator.py:34-64, configs/Distillation/FCOS/fcos_R_50...yaml:24]."""
    from lgd_amd import ops
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.dynamic_teacher import DynamicTeacher
    from lgd_amd.structures import ImageList
    B, H, W = 16, 800, 1344
    t = DynamicTeacher(_cfg(False, "stuGuided", "x1y1x2y2", student="FCOSCT", meta="FCOS"))
    t.load_state_dict(cm.teacher_params(), strict=True)
    t.to(DEV).train()
    ad = SequentialConvs(None)
    ad.load_state_dict(cm.adapter_params(), strict=True)
    ad.to(DEV).train()
    feats = {k: torch.from_numpy(v).to(DEV) for k, v in synth.synth_features(B, H, W, seed=5).items()}
    gt = [(torch.from_numpy(b), torch.from_numpy(c)) for b, c in synth.synth_gt(B, H, W, 10, seed=4)]
    keys = sorted(feats)

    def run(lo, hi):
        n = hi - lo
        f = {k: v[lo:hi].clone().requires_grad_(True) for k, v in feats.items()}
        images = ImageList(torch.zeros(n, 3, H, W, device=DEV), [(H, W)] * n)
        tea, labels, geom = t((_batched_inputs(gt[lo:hi], H, W), images, None, f))
        assert geom.counts == [10] * n and len(labels) == n
        loss = ops.distill_in_mse(ad.levels([f[k] for k in keys]), [tea[k] for k in keys], 1.0)
        loss.backward()
        return {k: v.detach() for k, v in tea.items()}, float(loss), {k: f[k].grad for k in keys}
    prev = ops.conv3x3_backend(winograd=True)    # (a module-scoped golden fixture may still hold the library back-end)
    try:
        assert ops._WINO_TILE == 6
        tea16, loss16, g16 = run(0, 16)
        assert 0.5 < loss16 < 4.0 and loss16 == loss16
        halves = [run(0, 8), run(8, 16)]
    finally:
        ops.conv3x3_backend(*prev)
    for h, (tea8, _, g8) in enumerate(halves):
        for k in keys:
            assert cm.rel_err(tea16[k][8 * h:8 * h + 8], tea8[k]) < 1e-5, (h, k)
            # the gradient of a mean over 16 images w.r.t. one image's features is half that of the mean over its 8
            assert cm.rel_err(2.0 * g16[k][8 * h:8 * h + 8], g8[k]) < 1e-4, (h, k)
    mean_halves = 0.5 * (halves[0][1] + halves[1][1])
    assert abs(loss16 - mean_halves) <= 1e-6 * abs(mean_halves), (loss16, mean_halves)
    # GroupNorm(1, no affine) is the last op of the teacher: every sample of every level has zero mean / unit variance
    for k in keys:
        y = tea16[k].double()
        assert float(y.mean(dim=(1, 2, 3)).abs().max()) < 1e-5 and float((y.var(dim=(1, 2, 3), unbiased=False) - 1).abs().max()) < 1e-4

def test_full_size_conv3x3_properties():
    """the Winograd convolution at BASELINE config-2 size (B=8, whole pyramid, 256 -> 256): agrees with the library's direct
    convolution (same fp32 inputs, a different summation order) within 5e-5; is linear in the input; an impulse filter
    (centre tap = identity matrix) reproduces the input; <conv(x), g> == <x, conv^T(g)> (backward is the exact adjoint)."""
    import torch.nn.functional as F
    from lgd_amd import ops
    B, C = 8, 256
    level_hw = synth.pyramid_shapes(800, 1344)
    g = torch.Generator(device=DEV).manual_seed(1)
    xs = [torch.randn(B, C, h, w, device=DEV, generator=g) for h, w in level_hw]
    w = torch.randn(C, C, 3, 3, device=DEV, generator=g) * 0.02
    b = torch.randn(C, device=DEV, generator=g) * 0.1
    ys = ops.conv3x3_levels(xs, w, b)
    for x, y in zip(xs, ys):
        ref = F.conv2d(x, w, b, 1, 1)
        assert float((y - ref).abs().max()) <= 5e-5 * float(ref.abs().max())
    x2 = [torch.randn_like(x) for x in xs]
    y2 = ops.conv3x3_levels(x2, w, None)
    y12 = ops.conv3x3_levels([2.0 * p - q for p, q in zip(xs, x2)], w, None)
    y1 = ops.conv3x3_levels(xs, w, None)
    for p, q, r in zip(y1, y2, y12):
        assert cm.rel_err(r, 2.0 * p - q) < 2e-5
    wi = torch.zeros(C, C, 3, 3, device=DEV)
    wi[torch.arange(C), torch.arange(C), 1, 1] = 1.0
    for x, y in zip(xs, ops.conv3x3_levels(xs, wi, None)):
        assert float((y - x).abs().max()) <= 2e-5 * float(x.abs().max())
    xr = [x.clone().requires_grad_(True) for x in xs]
    gy = [torch.randn_like(y) for y in ys]
    out = ops.conv3x3_levels(xr, w, None)
    torch.autograd.backward(out, gy)
    lhs = sum(float((o.detach().double() * q.double()).sum()) for o, q in zip(out, gy))
    rhs = sum(float((x.detach().double() * x.grad.double()).sum()) for x in xr)
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs)


# ------------------------------------------------------------------------------------------- f-1: one head pass over student + teacher
@pytest.mark.parametrize("yaml_name,conv_backend", [("lgd_retinanet_r50", "winograd"), ("lgd_fcos_r50", "winograd"),
                                                    ("lgd_retinanet_r50", "library")], indirect=["conv_backend"])
def test_single_head_pass_equals_two_passes(yaml_name, conv_backend):
    """the head runs ONCE over the 2 x L maps of the student and teacher pyramids
    (`student.predict_pair`) instead of twice [ref: distillator.py:88-91 + 107-112].
    (1) same function: on fixed pyramids, predict_pair == two predict calls (outputs to 1e-5 -- the GEMM over twice the tiles
        may sum in another order);
    (2) same training signal: the seven / five loss values of the whole meta-arch to 1e-6, every parameter gradient within the
        run-to-run noise of this device (the library's backbone convolutions are not bitwise reproducible, and one ReLU mask that
        flips moves a gradient by ~1e-3: tools/diag_headpass.py measures 2e-4 .. 8e-4 between two IDENTICAL runs)."""
    import os
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", yaml_name + ".yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    model = build_model(cfg).train()
    model.distill_flag = 1
    s = model.student
    gen = torch.Generator(device=DEV).manual_seed(3)
    fa = [torch.randn(2, 256, h, w, device=DEV, generator=gen) for h, w in synth.pyramid_shapes(256, 320)]
    fb = [torch.randn(2, 256, h, w, device=DEV, generator=gen) for h, w in synth.pyramid_shapes(256, 320)]
    with torch.no_grad():
        pair = s.predict_pair(fa, fb)
        one_a, one_b = s.predict(fa), s.predict(fb)
    for got, want in ((pair[1], one_a[1:]), (pair[2], one_b[1:])):
        for go, wo in zip(got, want):
            for x, y in zip(getattr(go, "raw", go), getattr(wo, "raw", wo)):
                # (the pair's products may run on csrc/h2.hip -- twice the tiles pass its speed gate -- and the single passes on gemm3 / the library:
                #  two fp32-class evaluations of five stacked convolutions)
                assert cm.rel_err(x, y) < 3e-5
    data = synthetic_batch(2, 256, 320, 5, seed=5)

    def run(fused):
        model.fused_head_pass = fused
        if hasattr(s, "loss_normalizer"):
            s.loss_normalizer = torch.tensor(100.0, device=DEV)
        model.zero_grad(set_to_none=True)
        losses = model(data)
        sum(losses.values()).backward()
        return ({k: float(v.detach()) for k, v in losses.items()},
                {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    a1, a2, b1 = run(True), run(True), run(False)
    assert set(a1[0]) == set(b1[0])
    for k in a1[0]:
        assert abs(a1[0][k] - b1[0][k]) <= 1e-6 * abs(b1[0][k]) + 1e-7, (k, a1[0][k], b1[0][k])
    assert set(a1[1]) == set(b1[1])
    worst = (0.0, 0.0, "")
    for n in a1[1]:
        if float(b1[1][n].abs().max()) < 1e-9:
            continue
        noise, diff = cm.rel_err(a1[1][n], a2[1][n]), cm.rel_err(a1[1][n], b1[1][n])
        worst = max(worst, (diff, noise, n))
        assert diff <= max(5e-3, 5 * noise), (n, diff, noise)
    print("fused vs two-pass gradients: worst %.1e (run-to-run noise of that tensor %.1e) at %s" % worst)


# ------------------------------------------------------------------------------------------- (e): the real model under torch.distributed / RCCL
def _init_single_rank_nccl():
    import socket
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))


def test_trainer_step_under_single_rank_rccl():
    """the REAL DistillatorRetinaNet wrapped in DistributedDataParallel over the `nccl` (= RCCL) backend [ref: train.py:279-281],
    world size 1: three Trainer.steps across both phase switches (frozen backbone -> trainable rebuilds the reducer; distill
    flag off -> on) must leave the same parameters as the non-DDP trainer started from the same weights."""
    import copy
    import os
    import torch.distributed as dist
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    from torch.nn.parallel import DistributedDataParallel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    data = synthetic_batch(2, 256, 320, 5, seed=6)
    its = (0, 25000, 40000)
    plain = Trainer(cfg, base, distributed=False)
    for it in its:
        plain.step(data, it)
    _init_single_rank_nccl()
    try:
        ddp = Trainer(cfg, twin, distributed=True)
        assert isinstance(ddp.model, DistributedDataParallel)
        first = ddp.model
        for it in its:
            ddp.step(data, it)
        assert ddp.model is not first  # the frozen -> trainable switch rebuilt the reducer
        m = ddp.fetch_metrics()        # one all-reduce over RCCL
        assert all(np.isfinite(v) for v in m.values())
    finally:
        dist.destroy_process_group()
    worst = 0.0
    for (n, a), (_, b) in zip(base.named_parameters(), twin.named_parameters()):
        worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-12)))
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), n
    print("DDP(world 1, RCCL) vs plain trainer: worst relative parameter difference %.2e" % worst)


def _nccl2_worker(rank, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=dev)
    try:
        from lgd_amd import config
        from lgd_amd.data import synthetic_batch
        from lgd_amd.distillator import build_model
        from lgd_amd.engine import Trainer
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", "cuda:%d" % rank])
        torch.manual_seed(0)
        tr = Trainer(cfg, build_model(cfg), device=dev)
        for it in (19999, 20000, 40000):
            tr.step(synthetic_batch(1, 256, 320, 4, seed=10 + rank), it)
        m = tr.fetch_metrics()
        w = torch.cat([p.detach().reshape(-1) for p in tr.raw_model.parameters()])
        ws = [torch.empty_like(w) for _ in range(2)]
        dist.all_gather(ws, w)
        q.put((rank, bool(torch.equal(ws[0], ws[1])), m["loss_distill"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_trainer_two_ranks_rccl():
    """two processes, two GPUs, RCCL: replicas fed different images stay bit-identical (gradients were averaged) and report
    the same rank-averaged metrics.  Skips on a single-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (the driver's multi-GPU node)")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl2_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=800) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == res[1][2]


def _gloo2_one_gpu_worker(rank, port, q, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from lgd_amd import config
        from lgd_amd.data import synthetic_batch
        from lgd_amd.distillator import build_model
        from lgd_amd.engine import Trainer
        from torch.nn.parallel import DistributedDataParallel
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
        torch.manual_seed(0)
        tr = Trainer(cfg, build_model(cfg))
        assert isinstance(tr.model, DistributedDataParallel) and tr._fused_sgd is not None
        for it in (19999, 20000, 40000):
            tr.step(synthetic_batch(1, 256, 320, 4, seed=10 + rank), it)
        m = tr.fetch_metrics()
        w = torch.cat([p.detach().reshape(-1) for p in tr.raw_model.parameters()]).cpu()
        ws = [torch.empty_like(w) for _ in range(2)]
        dist.all_gather(ws, w)
        if rank == 0:
            torch.save(w, os.path.join(out_dir, "w.pt"))
        q.put((rank, bool(torch.equal(ws[0], ws[1])), m["loss_distill"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_trainer_two_ranks_on_one_gpu(tmp_path):
    """world size 2 on the box there is: two processes share cuda:0, the exchange runs over `gloo` (RCCL refuses two ranks on one
    device).  The REAL DistillatorRetinaNet on the HIP path under DistributedDataParallel with bucket-view gradients and the fused
    clip + SGD launch, each rank fed its own image, across the frozen -> trainable and distill off -> on switches
    [ref: train.py:279-281, 303-310]: (i) the replicas stay bit-identical, (ii) both report the same rank-averaged metrics, (iii)
    the parameters equal a single process that keeps one replica per rank, feeds each its rank's image and steps both on the MEAN
    of the two gradients (clip after the mean, like the reducer's hooks leave it) with torch's optimizers."""
    import socket
    import torch.multiprocessing as mp
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gloo2_one_gpu_worker, args=(r, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=800) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert res[0][2] == res[1][2]
    w_ddp = torch.load(os.path.join(str(tmp_path), "w.pt"))

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
    import copy
    torch.manual_seed(0)
    models = [build_model(cfg)]
    models.append(copy.deepcopy(models[0]))          # a replica per rank: each keeps its own buffers (RetinaNet's loss_normalizer EMA)
    trs = [Trainer(cfg, m, distributed=False, fused_sgd=False) for m in models]
    batches = [synthetic_batch(1, 256, 320, 4, seed=10 + r) for r in range(2)]
    for it in (19999, 20000, 40000):
        for tr, m, b in zip(trs, models, batches):
            tr.set_phase(it)
            m.train()
            tr.stu_optimizer.zero_grad(set_to_none=True)
            tr.tea_optimizer.zero_grad(set_to_none=True)
            sum(m(b).values()).backward()
        for p0, p1 in zip(models[0].parameters(), models[1].parameters()):
            assert (p0.grad is None) == (p1.grad is None)
            if p0.grad is not None:
                g = (p0.grad + p1.grad) * 0.5
                p0.grad, p1.grad = g, g.clone()
        for tr in trs:
            tr._clip()
            tr.stu_optimizer.step()
            tr.tea_optimizer.step()
            tr.stu_scheduler.step()
            tr.tea_scheduler.step()
    model = models[0]
    w_one = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    err = float((w_ddp - w_one).abs().max())
    print("2 ranks (gloo, one GPU) vs one process on the mean gradient: max |dw| %.3e" % err)
    assert torch.allclose(w_ddp, w_one, rtol=1e-4, atol=1e-6)


def test_small_uploads_through_the_pinned_ring():
    """hip.to_device: short host lists reach the device intact through the ring of pinned staging slots, also after the ring has
    wrapped several times (a slot is rewritten only long after its copy has executed)."""
    from lgd_amd import hip
    got, want = [], []
    for i in range(3 * 1024 + 7):
        vals = [i, i + 1, 2 * i, -i]
        want.append(vals)
        got.append(hip.to_device(vals, torch.int32 if i % 2 else torch.int64, DEV))
        if i % 256 == 255:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert g.tolist() == w
    big = list(range(1000))   # larger than a slot: plain copy
    assert hip.to_device(big, torch.int64, DEV).tolist() == big
    assert hip.to_device([[1, 2], [3, 4]], torch.int32, DEV).tolist() == [[1, 2], [3, 4]]


def test_forks_share_one_side_stream():
    """round 6 (profiles/r06_hw_queues_and_forks.txt): every fork of the step runs on ONE physical side stream per device -- two compute streams in the
    process, two hardware queues whatever else it created -- and a fork's join waits for the event recorded where ITS work ends, not for whatever a later
    fork put on the shared stream (the adapter is issued between the label encoder's fork and its join)."""
    from lgd_amd import streams
    dev = torch.device(DEV)
    assert streams._ONE_SIDE
    names = ("teacher", "head", "adapter", "fpn")
    assert len({id(streams.side(dev, n)) for n in names}) == 1
    prev, streams._ONE_SIDE = streams._ONE_SIDE, False
    try:
        assert len({id(streams.side(dev, n)) for n in names}) == 4      # (the round-5 form, behind LGD_ONE_SIDE_STREAM=0)
    finally:
        streams._ONE_SIDE = prev
    main, s = streams.fork(dev, "teacher")
    with torch.cuda.stream(s):
        a = torch.full((1 << 20,), 3.0, device=dev)
        ev = streams.done(s)
        torch.cuda._sleep(200_000_000)                                  # "a later fork": ~0.1 s of work behind the event on the same stream
        late = torch.full((4,), 1.0, device=dev)
    streams.join(main, s, [a], event=ev)
    t0 = time.time()
    assert float(a.sum()) == 3.0 * (1 << 20)                            # the first fork's result is there ...
    assert not s.query()                                                # ... while the shared stream is still busy with the later one
    assert time.time() - t0 < 0.05
    streams.join(main, s, [late])
    assert float(late.sum()) == 4.0


def test_pinned_ring_records_every_stream_of_a_segment():
    """the step uploads from its main stream AND from side streams (lgd_amd/streams.py); a segment of the ring must hold an event for
    every stream that copied out of it, not only for the stream of its last upload -- otherwise a slot could be rewritten under a pending copy."""
    from lgd_amd import hip
    ring = hip._PinnedRing()
    side = torch.cuda.Stream(DEV)
    got = []
    for i in range(2 * ring.SLOTS + 5):
        t = torch.tensor([i, -i], dtype=torch.int64)
        if i % 3 == 0:
            with torch.cuda.stream(side):
                got.append((i, ring.upload(t, torch.device(DEV))))
        else:
            got.append((i, ring.upload(t, torch.device(DEV))))
    assert all(evs is not None and len(evs) == 2 for evs in ring.events), [None if e is None else len(e) for e in ring.events]
    assert ring.waits >= ring.SLOTS // ring.SEGMENT
    torch.cuda.synchronize()
    for i, g in got:
        assert g.tolist() == [i, -i]


def test_bench_stdout_is_one_json_record():
    """the driver contract: `python bench.py ...` prints exactly ONE line on stdout, the JSON record -- also when RCCL is initialised
    (its version banner goes to the C stdout and would otherwise follow the record)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, LGD_FORCE_DDP="1", MASTER_PORT="29577")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                        "--config", os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), "--batch-per-gpu", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_mfma", "roofline_lgd_forward"):
        assert k in rec, k
    # the dominant hand-written kernel: gemm3 (MFMA bound, priced against the bf16 peak) where it runs, a Winograd transform (HBM) otherwise
    assert rec["roofline"]["bound"] in ("hbm", "mfma") and 0 < rec["roofline"]["frac"] < 1
    assert rec["roofline"]["unit"] == ("GB/s" if rec["roofline"]["bound"] == "hbm" else "TFLOP/s")
    hb = rec["roofline_hbm"] or rec["roofline"]
    assert hb["bound"] == "hbm" and 0 < hb["frac"] < 1
    assert rec["config"]["workload"] and "model" not in rec["config"]


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("form", ["plain", "torchrun"])
def test_bench_two_ranks_on_one_gpu(form):
    """bench.py at N = 2 on the box there is, in both forms it can be started in: `python bench.py --gpus 2` PLAINLY (it re-executes itself
    as 2 ranks, lgd_amd/launch.py) and the driver's `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`.  Both ranks
    on cuda:0, exchange over gloo (LGD_BENCH_SHARE_GPU / LGD_BENCH_BACKEND: test knobs, never a performance number).  The barriers, the
    max-over-ranks time, DDP, the host-thread slices and the rank-0 record: ONE JSON line on stdout, n_gpus 2, the communicator's size read
    back from an all-reduce, whole-job value = 2 ranks' images [ref: train.py:296-310]."""
    import json
    import subprocess
    import sys
    from lgd_amd import launch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LGD_BENCH_SHARE_GPU="1", LGD_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-per-gpu", "2", "--height", "256", "--width", "320"]
    head = [sys.executable] if form == "plain" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                      "--master-addr", "127.0.0.1", "--master-port", str(launch.free_port())]
    r = subprocess.run(head + tail, env=env, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["parallelism"] == "dp2" and rec["config"]["global_batch"] == 4
    assert rec["rccl_ranks"] == 2 and rec["collective_backend"] == "gloo"
    assert abs(rec["value"] - 4 * 1e3 / rec["ms_per_step"]) <= 1e-6 * rec["value"]      # whole-job images per second over the slowest rank's time
    assert "cpu_baseline" not in rec                                                      # rank 0 at N = 1 only
    if len(os.sched_getaffinity(0)) >= 2:   # equal slices of the CPUs of the GPU's NUMA node (or of all allowed CPUs when sysfs does not tell)
        n_cpus = int(rec["host_threads_pinned"].split()[0])
        assert 1 <= n_cpus <= len(os.sched_getaffinity(0)) // 2 and "CPUs per rank" in rec["host_threads_pinned"]


def test_trainer_fused_sgd_equals_torch_optimizers():
    """Trainer.step with the one-launch clip + SGD (csrc/optim.hip, the default) against the same trainer on torch's
    clamp_ / SGD(foreach) path (Trainer(fused_sgd=False)) from the same weights, three steps across both phase switches
    [ref: train.py:191-207]; the checkpoint payload keeps the reference's optimizer layout either way."""
    import copy
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    data = synthetic_batch(2, 256, 320, 5, seed=6)
    its = (0, 25000, 40000)
    fused = Trainer(cfg, base, distributed=False)
    assert fused._fused_sgd is not None
    plain = Trainer(cfg, twin, distributed=False, fused_sgd=False)
    assert plain._fused_sgd is None
    for it in its:
        fused.step(data, it)
        plain.step(data, it)
    worst = 0.0
    for (n, a), (_, b) in zip(base.named_parameters(), twin.named_parameters()):
        worst = max(worst, float((a - b).abs().max() / (a.abs().max() + 1e-12)))
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), n
    print("fused clip+SGD trainer vs torch optimizers: worst relative parameter difference %.2e" % worst)
    sa, sb = fused.state_dict(), plain.state_dict()
    for k in ("stu_optimizer", "tea_optimizer"):
        assert len(sa[k]["param_groups"]) == len(sb[k]["param_groups"])
        assert sorted(sa[k]["state"].keys()) == sorted(sb[k]["state"].keys())
    # resume: the fused step picks the loaded momentum buffers up (a copy: Optimizer.load_state_dict keeps the tensors it is
    # handed when dtype and device already match, and `sb` holds plain's LIVE buffers)
    fused.load_state_dict(copy.deepcopy(sb))
    fused.step(data, 40000)
    plain.step(data, 40000)
    for (n, a), (_, b) in zip(base.named_parameters(), twin.named_parameters()):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), n


def test_label_encoder_on_the_side_stream_equals_in_line():
    """DynamicTeacher.encode_ahead: the label encoder (annotations only) issued on a side stream ahead of the backbone, its backward on that stream
    under the backbone's -- against the in-line order (side_stream = False) from the same weights: losses and parameters after three trainer steps
    across the phase switches equal to the run-to-run noise of the step (fp32 atomics in the pooling reductions), and the side stream is really used."""
    import copy
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    assert base.teacher.side_stream
    twin.teacher.side_stream = False
    data = synthetic_batch(2, 256, 320, 5, seed=6)
    a = Trainer(cfg, base, distributed=False)
    b = Trainer(cfg, twin, distributed=False)
    seen = []
    enc = base.teacher.label_encoder_.forward
    base.teacher.label_encoder_.forward = lambda x0: (seen.append(torch.cuda.current_stream() != torch.cuda.default_stream()), enc(x0))[1]
    for it in (0, 25000, 40000):
        la, lb = a.step(data, it), b.step(data, it)
        for k in la:
            va, vb = float(la[k].detach()), float(lb[k].detach())
            assert abs(va - vb) <= 1e-5 * max(1.0, abs(vb)), (it, k, va, vb)
    assert seen == [True, True, True]
    from lgd_amd import dynamic_teacher
    assert dynamic_teacher._RUNTIME.get(twin.teacher, {}).get("side") is None and dynamic_teacher._RUNTIME[base.teacher].get("side") is not None
    copy.deepcopy(base)   # (the stream lives outside the module: the model still copies)
    for (n, p), (_, q) in zip(base.named_parameters(), twin.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), (n, float((p - q).abs().max()))


@pytest.mark.parametrize("yaml_name", ["lgd_retinanet_r50.yaml", "lgd_fcos_r50.yaml"])
def test_head_towers_on_two_streams_equal_one_stream(yaml_name):
    """the head's box tower on a second stream beside the class tower and the adapter on its own stream beside the teacher (lgd_amd/streams.py;
    autograd runs their backward there too) against everything on one stream, from the same weights: losses and parameters after three trainer steps across the phase switches equal to the run-to-run noise"""
    import copy
    from lgd_amd import config, streams
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    from lgd_amd.student import retinanet
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", yaml_name), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    data = synthetic_batch(2, 256, 320, 5, seed=6)
    a = Trainer(cfg, base, distributed=False)
    b = Trainer(cfg, twin, distributed=False)
    assert base.adapter_stream
    from lgd_amd.student import fpn
    shipped = (retinanet._HEAD_STREAMS, fpn._FPN_STREAM)
    retinanet._HEAD_STREAMS = fpn._FPN_STREAM = True   # (every fork the code has: the FPN's ships off since round 6 -- student/fpn.py -- and stays under test)
    twin.adapter_stream = False          # (the adapter beside the teacher on ITS side stream: switched off in the twin as well)
    forks = []
    real_fork = streams.fork
    streams.fork = lambda dev, name, inputs=(): (forks.append(name), real_fork(dev, name, inputs))[1]
    from lgd_amd import ops
    prev_conv = ops.conv3x3_backend(winograd=True, tile=6)   # (a module-scoped golden fixture may still hold the library back-end)
    # the maps of this test are far below the size gates: the per-call gate of the forks (ops.convs_on_own_kernels) would keep every chain on one
    # stream.  Forced on, the side chains run on the vendor library's GEMMs -- ordered against the main stream's by streams.library_call
    os.environ["LGD_SIDE_STREAMS_ANY"] = "1"
    assert ops.side_streams_ok()
    try:
        for it in (0, 25000, 40000):
            la = a.step(data, it)
            n_forks = len(forks)
            retinanet._HEAD_STREAMS = fpn._FPN_STREAM = False
            try:
                lb = b.step(data, it)
            finally:
                retinanet._HEAD_STREAMS = fpn._FPN_STREAM = True
            assert len(forks) == n_forks and "head" in forks and "adapter" in forks and "fpn" in forks
            for k in la:
                va, vb = float(la[k].detach()), float(lb[k].detach())
                assert abs(va - vb) <= 1e-5 * max(1.0, abs(vb)), (it, k, va, vb)
    finally:
        streams.fork = real_fork
        ops.conv3x3_backend(*prev_conv)
        del os.environ["LGD_SIDE_STREAMS_ANY"]
        retinanet._HEAD_STREAMS, fpn._FPN_STREAM = shipped
    for (n, p), (_, q) in zip(base.named_parameters(), twin.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), (n, float((p - q).abs().max()))


@pytest.mark.timeout(600)
def test_f4x4_variant_with_forks_makes_progress():
    """Regression test of round 5's two-stream stall (profiles/r06_stall_root_cause.txt): BASELINE config 5 (R-101-DCNv2, 2 multi-scale images) on the
    F(4x4) A/B variant with EVERY fork forced on -- two long-K rocBLAS products of the vendor library side by side on two streams, which stopped
    making progress on the second step when nothing ordered them -- now runs through (streams.library_call orders every library call of this
    package).  A subprocess with its own watchdog: a stall is exit code 3 after 60 s, not a hung suite."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("LGD_")}
    env["LGD_SIDE_STREAMS_ANY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stall_repro.py"), "--steps", "12", "--tile", "4"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0 and "12 steps done" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])


@pytest.mark.timeout(1800)
def test_step_forks_under_a_competing_stream():
    """VERDICT r5 item 2c / ADVICE r5: 300 optimizer steps of the shipped path at BASELINE config 2 (RetinaNet R-50 + LGD, 8 images of 800 x 1333)
    with ALL forks of the step on, while a further stream saturates HBM with 256 MB copies -- the place RCCL's kernels take in a data-parallel job
    -- against the same steps on ONE stream with nothing beside them (tools/stream_stress.py, run as a subprocess with a deadline: a stall fails the
    test instead of hanging the suite).  Progress: every step returns its losses.  Results: the forked run is held to the drift the tool measures
    between two IDENTICAL one-stream runs up to the same step (x10) -- at this size the one-stream step is bit-reproducible, so that means
    BIT-EQUAL losses over all 300 steps, which is what round 6 measures since the library holds no packed-fp32 instruction any more (they return
    wrong low bits beside another kernel's f16 MFMAs: test_abi.py::test_no_packed_fp32_in_the_library) and the adapter is issued at the same point
    of the program whether it forks or not (same gradient summation order).  [ref: the step is train.py:182-215]"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("LGD_")}
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_stress.py"), "--steps", "300", "--deadline", "300"], env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    fa, fb, fc = rec["forked"], rec["one_stream"], rec["one_stream_again"]
    assert len(fa) == len(fb) == len(fc) == 300 and rec["competitor_copies"] == 300 * 24
    assert {"head", "adapter", "fpn"} <= set(rec["forks_per_step"]) and all(v >= 1.0 for v in rec["forks_per_step"].values()), rec["forks_per_step"]

    def dev(x, y):
        assert all(set(p) == set(q) and all(np.isfinite(v) for v in list(p.values()) + list(q.values())) for p, q in zip(x, y))
        return [max(abs(p[k] - q[k]) / max(abs(q[k]), 1e-6) for k in p) for p, q in zip(x, y)]
    dab, dbc = dev(fa, fb), dev(fb, fc)
    table = r.stderr[r.stderr.find("step:"):][:700]
    assert max(dab[:20]) <= 1e-5, (max(dab[:20]), table)   # (measured 0 .. 2e-6: equal to the last bits of fp32 sums; the forks change no arithmetic)
    for i in range(300):   # (where the one-stream step is bit-reproducible -- measured at this size: all 300 steps -- this demands BIT-equal losses of the forked run)
        assert max(dab[:i + 1]) <= 10.0 * max(dbc[:i + 1]), (i, max(dab[:i + 1]), max(dbc[:i + 1]), table)
    print("300 steps at config 2: forks on + competing stream %.1f ms/step, one stream %.1f ms/step; worst loss deviation over the first 20 steps %.1e "
          "(two one-stream runs: %.1e), over the first 50 %.1e (%.1e)" % (rec["ms_per_step_forked_under_load"], rec["ms_per_step_one_stream"], max(dab[:20]),
                                                                          max(dbc[:20]), max(dab[:50]), max(dbc[:50])))


def test_step_folds_equal_per_op_folds():
    """StepFolds (student/resnet.py: w * scale of every trainable 1x1 ConvBN in ONE launch per step) against the fold inside each op:
    the folded filters are bit-identical, the trainer uses them for every trainable 1x1 ConvBN in every phase (losses and the parameters
    after three steps across the phase switches equal a trainer without them to the run-to-run noise of the step), and a weight
    written after prepare() falls back to the per-op fold."""
    import copy
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    from lgd_amd.student.resnet import ConvBN, StepFolds
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    sf = StepFolds(base)
    sf.prepare()
    mods = [m for m in base.modules() if isinstance(m, ConvBN) and (m._pointwise or m._pointwise_s2) and m.weight.requires_grad]
    assert len(mods) >= 20 and len(sf.mods) == len(mods)
    for m in mods:
        scale = m.norm.scale_shift()[0]
        assert torch.equal(m._cached_fold(scale), m.weight.detach() * scale.view(-1, 1, 1, 1))
    # the same launch leaves max |w * scale| of every fold (the f16x2 scale of the filter's image: ops.gemm2h_bmm finds it by the fold's offset)
    table, version = sf.flat._lgd_w_amax_table
    assert version == sf.flat._version and len(table) == len(mods)
    for v in sf.views:
        word, numel = table[v.storage_offset()]
        assert numel == v.numel() and word.view(torch.float32).item() == v.abs().max().item()
    # ... and one more launch the f16x2 images of both products of every fold (W, W^T): byte for byte what lgd_gemm2h_split makes of each
    from lgd_amd import hip
    lib = hip.load()
    images, version = sf.flat._lgd_w_img_table
    assert version == sf.flat._version and len(images) == 2 * len(mods)
    for v in sf.views[:3] + sf.views[-3:]:
        w2 = v.view(v.shape[0], -1)
        for a0, tr in ((w2, False), (w2.t(), True)):
            img, inv, numel = images[(v.storage_offset(), tr)]
            M, K = a0.shape
            ref = torch.empty(lib.lgd_gemm2h_image_bytes(1, M, K), dtype=torch.uint8, device=DEV)
            rinv = torch.empty(1, device=DEV)
            hip.check(lib.lgd_gemm2h_split(hip.ptr(a0), 0, a0.stride(0), a0.stride(1), 1, M, K, hip.ptr(table[v.storage_offset()][0]), hip.ptr(ref), hip.ptr(rinv),
                                           hip.stream_ptr()), "lgd_gemm2h_split")
            assert numel == v.numel() and torch.equal(img, ref) and torch.equal(inv, rinv), (tuple(v.shape), tr)
    with torch.no_grad():
        mods[0].weight.mul_(1.5)                      # written after prepare(): the cached fold is stale and must not be used
    assert mods[0]._cached_fold(mods[0].norm.scale_shift()[0]) is None
    with torch.no_grad():
        mods[0].weight.div_(1.5)
    data = synthetic_batch(2, 256, 320, 5, seed=6)
    a = Trainer(cfg, base, distributed=False)
    assert a._step_folds is not None
    os.environ["LGD_STEP_FOLDS"] = "0"
    try:
        b = Trainer(cfg, twin, distributed=False)
    finally:
        del os.environ["LGD_STEP_FOLDS"]
    assert b._step_folds is None
    hits = []
    cached = ConvBN._cached_fold

    def counting(self, scale):
        r = cached(self, scale)
        if r is not None and self.weight.requires_grad:
            hits.append(self)
        return r
    ConvBN._cached_fold = counting
    try:
        for it in (0, 25000, 40000):
            la = a.step(data, it)
            want = sum(1 for m in base.modules() if isinstance(m, ConvBN) and (m._pointwise or m._pointwise_s2) and m.weight.requires_grad)
            # every trainable 1x1 ConvBN took the step's fold (iteration 0: the backbone is frozen, SOLVER.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
            assert (want > 0 or it == 0) and len(set(map(id, hits))) == want, (it, len(hits), want)
            hits.clear()
            lb = b.step(data, it)
            assert not hits                              # the trainer without StepFolds folds inside each op
            for k in la:
                va, vb = float(la[k].detach()), float(lb[k].detach())
                assert abs(va - vb) <= 1e-5 * max(1.0, abs(vb)), (it, k, va, vb)
    finally:
        ConvBN._cached_fold = cached
    # (the step is not bit-reproducible run to run -- fp32 atomics in the pooling / matching reductions -- so the parameters are compared to the noise
    #  of those, not bit for bit; the folds themselves are, above)
    for (n, p), (_, q) in zip(base.named_parameters(), twin.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-4, atol=1e-5), (n, float((p - q).abs().max()))


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("yaml_name", ["lgd_retinanet_r50.yaml", "lgd_fcos_r50.yaml", "lgd_retinanet_r101.yaml"])   # (R-101-DCNv2 passes too: 3.5 min of library conv search)
def test_full_size_step_shipped_path_vs_library_convolutions(yaml_name):
    """Two training steps of the distillator meta-arch (BASELINE configs 2 / 3, and config 4's exact per-rank workload: R-101 at 2 images) at the
    BASELINE image size (2 x 800 x 1333, 10 boxes; every 3x3 convolution of the backbone, FPN, head and teacher is above the Winograd
    threshold) on the SHIPPED path -- whatever `ops` ships as its default tile (F(6x6,3x3) transforms with the folded pre-activations), conv1 +
    shortcut nodes with beta = 1 accumulation, FPN laterals as GEMMs, fused stem epilogue, one-launch clip + SGD -- against the same model
    on the library's convolutions and torch's optimizers: same losses (the loss of step 2 goes through every gradient and the parameter
    update) [ref: train.py:184-207]."""
    import copy
    from lgd_amd import config, ops
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", yaml_name), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    data = synthetic_batch(2, 800, 1333, 10, seed=3)
    d = cfg.MODEL.DISTILLATOR
    it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
    prev = ops.conv3x3_backend(winograd=True)           # the shipped tile, whatever it is: no `tile=` here
    assert ops._WINO_TILE == 6                          # ... and today that is F(6x6,3x3)
    try:
        a = Trainer(cfg, base, distributed=False)
        assert a._fused_sgd is not None
        la = [{k: float(v) for k, v in a.step(data, it0 + i).items()} for i in range(2)]
        ops.conv3x3_backend(winograd=False)
        b = Trainer(cfg, twin, distributed=False, fused_sgd=False)
        assert b._fused_sgd is None
        lb = [{k: float(v) for k, v in b.step(data, it0 + i).items()} for i in range(2)]
    finally:
        ops.conv3x3_backend(*prev)
    for i in range(2):
        for k in la[i]:
            assert abs(la[i][k] - lb[i][k]) <= 2e-4 * abs(lb[i][k]) + 1e-6, (i, k, la[i][k], lb[i][k])
    print("full-size step, shipped vs library path: step-2 losses", la[1], lb[1])


@pytest.mark.timeout(1500)
def test_config2_step_shipped_vs_library_b8():
    """BASELINE config 2's REAL step -- RetinaNet R-50 + LGD, 8 images of 800 x 1333, production policy (every product above its
    size gate on h2.hip / gemm2h, F(6x6,3x3) transforms) with all forks of the step ON (label encoder, box tower, adapter, FPN small levels on side
    streams) -- against an independent path: the library's convolutions (MIOpen) and torch's optimizers on ONE stream.  Two optimizer steps from the
    same weights; the second step's losses go through every gradient and the parameter update [ref: train.py:182-215].  What the 42-step tool run
    (tools/trajectory_check.sh) shows over means, here per step inside the committed suite: every loss to 1e-5, the total to 3e-6."""
    import copy
    from lgd_amd import config, ops, streams
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r50.yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    data = synthetic_batch(8, 800, 1333, 10, seed=3)
    d = cfg.MODEL.DISTILLATOR
    it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
    forks = []
    real_fork = streams.fork
    streams.fork = lambda dev, name, inputs=(): (forks.append(name), real_fork(dev, name, inputs))[1]
    prev = ops.conv3x3_backend(winograd=True)
    try:
        assert ops._WINO_TILE == 6 and ops.side_streams_ok()
        a = Trainer(cfg, base, distributed=False)
        assert a._fused_sgd is not None and base.teacher.side_stream
        la = [{k: float(v) for k, v in a.step(data, it0 + i).items()} for i in range(2)]
        assert {"head", "adapter"} <= set(forks), forks     # the step as shipped: every shipped fork taken (+ the label encoder's: base.teacher.side_stream)
        nf = len(forks)
        ops.conv3x3_backend(winograd=False)
        os.environ["LGD_SIDE_STREAMS"] = "0"                         # every fork off ...
        twin.teacher.side_stream = False                             # (... and the label encoder's own side stream: an instance switch, as bench.py --one-stream)
        try:
            assert not ops.side_streams_ok()
            b = Trainer(cfg, twin, distributed=False, fused_sgd=False)
            lb = [{k: float(v) for k, v in b.step(data, it0 + i).items()} for i in range(2)]
        finally:
            del os.environ["LGD_SIDE_STREAMS"]
        assert len(forks) == nf, forks[nf:]                          # ... against ONE stream
    finally:
        streams.fork = real_fork
        ops.conv3x3_backend(*prev)
    for i in range(2):
        ta, tb = sum(la[i].values()), sum(lb[i].values())
        for k in la[i]:
            assert np.isfinite(la[i][k]) and abs(la[i][k] - lb[i][k]) <= 1e-5 * abs(lb[i][k]) + 1e-7, (i, k, la[i][k], lb[i][k])
        assert abs(ta - tb) <= 3e-6 * abs(tb), (i, ta, tb)
    print("config 2 (8 x 800 x 1333), shipped path with forks vs library convolutions on one stream: step-2 losses", la[1], lb[1])


@pytest.mark.parametrize("tile", [4, 6])
def test_fcos_head_matches_reference_golden(tile):
    """the product FCOSHead (lgd_amd/student/fcos.py: shared-input Winograd convolutions, gn_group.hip GroupNorm(32) + ReLU over all
    maps, per-level scale) against outputs and gradients of the reference's OWN FCOSHead [thirdparty_heads/fcos.py:433-546]
    (tests/golden/fcos_head.npz, generated from /root/reference): same state_dict names, closed-form parameters and features."""
    from lgd_amd import config, ops
    from lgd_amd.student.fcos import FCOSHead
    from oracle import student_oracle as SO
    g = cm.golden("fcos_head")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_fcos_r50.yaml"), ["MODEL.DEVICE", DEV])
    head = FCOSHead(cfg)
    sd = {k: torch.from_numpy(v) for k, v in synth.fcos_head_params(SO.fcos_head_param_shapes()).items()}
    missing, unexpected = head.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    head.to(DEV).train()
    feats_np, probes = cm.fcos_head_inputs()
    feats = [torch.from_numpy(f).to(DEV).requires_grad_(True) for f in feats_np]
    prev = ops.conv3x3_backend(winograd=True, min_tiles=0, tile=tile)
    try:
        outs = dict(zip(("logits", "reg", "ctr"), head(feats)))
        total = 0.0
        for kind, maps in outs.items():
            for i, t in enumerate(maps):
                e = cm.rel_err(t.detach().reshape(-1)[::7], g["%s_s_%d" % (kind, i)])
                assert e < 1e-4, (kind, i, e)
                if i >= 3:
                    assert cm.rel_err(t.detach(), g["%s_full_%d" % (kind, i)]) < 1e-4
                total = total + (t * torch.from_numpy(probes[kind][i]).to(DEV)).sum()
        assert abs(float(total.detach()) - float(g["total"])) <= 1e-4 * abs(float(g["total"]))
        total.backward()
    finally:
        ops.conv3x3_backend(*prev)
    # Gradients against the reference's fp32 run are a DIAGNOSTIC: four GroupNorm(32) + ReLU layers per tower, and a unit within
    # rounding of zero takes its mask from the summation order -- through the group statistics one flip moves every element of its
    # group (measured: 4e-3 relative L2 at the 24 x 32 level).  The hard assert is test_fcos_head_gradients_fp64_under_product_masks.
    worst = 0.0
    for i, f in enumerate(feats):
        e = cm.rel_err(f.grad.reshape(-1)[::7], g["gfeat_s_%d" % i])
        assert e < 5e-3 + (5e-3 if tile == 6 else 0.0), (i, e)   # measured 2..4e-3 with F(4x4), twice that with F(6x6)
        worst = max(worst, e)
    for n, prm in head.named_parameters():
        e = cm.rel_err(prm.grad.reshape(-1)[::cm.SAMPLE_STRIDE][:64], g["gw_s_" + n])
        assert e < 5e-3 + (5e-3 if tile == 6 else 0.0), (n, e)
        worst = max(worst, e)
    print("FCOS head vs reference: worst gradient deviation %.2e (tile %d)" % (worst, tile))


# ---------------------------------------------------------------------------------------------------------------------------------
# Model-level gradients under PINNED activation masks.  A pre-activation within fp32 rounding of zero takes its
# backward mask from the summation order of whoever computed it, so the reference's fp32 gradients (golden) and the product's differ
# by O(1e-3) on whichever levels have such a unit (test_distill_loss_and_grads_match_reference prints both deviations).  Here the
# question "are the product's kernels and wiring right?" is asked without that noise: the ORACLE is evaluated in fp64 with every ReLU
# mask taken from the product's own forward (15 row-LN ReLUs of the label encoder / canoni_proj_1D, student_proj_2D, rendering, the
# two refinement ReLUs, the two adapter ReLUs, per level) -- then ALL five levels and ALL parameters must agree to 1e-4.
class _PinnedReluF:
    """stands in for torch.nn.functional inside oracle/lgd_oracle.py: relu(x) = x * (the product's mask for this site)."""

    def __init__(self, masks):
        import torch.nn.functional as F
        self._F, self._masks, self.used, self.flip_frac = F, list(masks), 0, []

    def __getattr__(self, name):
        return getattr(self._F, name)

    def relu(self, x):
        m = self._masks[self.used].to(x.device)
        self.used += 1
        assert tuple(m.shape) == tuple(x.shape), (self.used, tuple(m.shape), tuple(x.shape))
        self.flip_frac.append(float(((x.detach() > 0) != m).double().mean()))
        return x * m.to(x.dtype)


def _run_product_capturing_masks(name, coef):
    """the product DynamicTeacher + adapter + distill loss with every ReLU-bearing op observable: the fused forms whose activation
    never reaches memory are replaced by their unfused kernels (gn_pool -> gn1 + mask_pool; the adapter's conv chain -> one node per
    conv, whose forward kernels are the chain's), everything else is the shipped path."""
    from lgd_amd import ops
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.base_distillator import BaseDistillator
    from lgd_amd.structures import ImageList
    B, H, W, ctx, interact, fmt, _, _ = cm.CASES[name]
    rec = {"ln": [], "proj": None, "render": None, "gn": [], "adapter": []}
    real = {k: getattr(ops, k) for k in ("row_ln", "gn_relu_mask_pool", "bias_ctx_relu", "gn1", "conv3x3_chain", "conv3x3_levels", "ctx_shift_fold",
                                         "conv3x3_gn")}
    on = lambda ys: [(y.detach() > 0).cpu() for y in ys]  # noqa: E731

    def row_ln(x, relu):
        y = real["row_ln"](x, relu)
        if relu:
            rec["ln"].append((y.detach() > 0).cpu())
        return y

    def pool(geom, xs):
        ys = real["gn1"](xs, True)
        rec["proj"] = on(ys)
        return ops.mask_pool(geom, ys)

    def ctx_relu(xs, c):
        ys = real["bias_ctx_relu"](xs, c)
        rec["render"] = on(ys)
        return ys

    def gn1(xs, relu):
        ys = real["gn1"](xs, relu)
        if relu:
            rec["gn"].append(on(ys))
        return ys

    # the folded activations (DynamicTeacher.fold_activations): the next convolution's input transform applies relu(fma(x, scale, shift)) with
    # the raw maps x and the per-(map, sample, channel) affine these calls return; the sign of an fp32 fma is the sign of the exact value,
    # which fp64 holds (the product of two floats is exact)
    def folded(aff, xs):
        n = xs[0].shape[0]
        a64 = aff.double()
        return [((x.detach().double() * a64[lv * n:(lv + 1) * n, :, 0, None, None] + a64[lv * n:(lv + 1) * n, :, 1, None, None]) > 0).cpu()
                for lv, x in enumerate(xs)]

    def ctx_shift_fold(xs, c):
        aff, ys = real["ctx_shift_fold"](xs, c)
        rec["render"] = folded(aff, ys)
        return aff, ys

    def conv_gn(xs, *a, **k):
        out = real["conv3x3_gn"](xs, *a, **k)
        for aff, ys in out:
            rec["gn"].append(folded(aff, ys))
        return out

    def levels(xs, w, b=None, relu=False, **kw):
        ys = real["conv3x3_levels"](xs, w, b, relu, **kw)
        if relu:   # the rendering conv of the configurations without a context box
            rec["render"] = on(ys)
        return ys

    def chain(xs, filters, relus):
        for (w, b), r in zip(filters, relus):
            xs = real["conv3x3_levels"](xs, w, b, r)
            if r:
                rec["adapter"].append(on(xs))
        return xs

    class D(BaseDistillator):
        def __init__(self, c):
            torch.nn.Module.__init__(self)
            self.coef = c
            self.adapter = torch.nn.ModuleDict({"distill": SequentialConvs(None)})
    prev = ops.conv3x3_backend(winograd=True, min_tiles=0)
    patched = {"row_ln": row_ln, "gn_relu_mask_pool": pool, "bias_ctx_relu": ctx_relu, "gn1": gn1, "conv3x3_chain": chain,
               "conv3x3_levels": levels, "ctx_shift_fold": ctx_shift_fold, "conv3x3_gn": conv_gn}
    try:
        for k, f in patched.items():
            setattr(ops, k, f)
        ops._MLP_FUSED = False   # the label encoder's ladders layer by layer (same launches: test_mlp_ladder_equals_layers), so that row_ln is seen
        teacher = _teacher(name)
        d = D(coef)
        d.adapter["distill"].load_state_dict(cm.adapter_params(), strict=True)
        d.to(DEV)
        d.distill_flag = 1
        feats = {k: v.to(DEV).requires_grad_(True) for k, v in cm.case_feats(name).items()}
        images = ImageList(torch.zeros(B, 3, H, W, device=DEV), [(H, W)] * B)
        tea, _, _ = teacher((_batched_inputs(cm.case_gt(name), H, W), images, None, feats))
        loss = d.distill({"stu": feats, "tea": tea}, None, None, None, None)
        pr = cm.probes({k: tea[k] for k in O.LEVELS})
        total = loss + sum((tea[k] * pr[k].to(DEV)).sum() for k in O.LEVELS)
        total.backward()
    finally:
        ops._MLP_FUSED = True
        for k, f in real.items():
            setattr(ops, k, f)
        ops.conv3x3_backend(*prev)
    grads = {n: p.grad for n, p in teacher.named_parameters()}
    grads.update({"adapter." + n: p.grad for n, p in d.adapter["distill"].named_parameters()})
    return rec, {k: feats[k].grad for k in O.LEVELS}, grads, float(total.detach())


@pytest.mark.parametrize("name", list(cm.CASES))
def test_teacher_gradients_fp64_under_product_masks(name):
    B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
    rec, gfeat, gw, total = _run_product_capturing_masks(name, coef)
    L = len(O.LEVELS)
    assert len(rec["ln"]) == 15 and len(rec["gn"]) == 2 and len(rec["adapter"]) == 2 and rec["proj"] and rec["render"]
    # the oracle's ReLU sites in ITS call order (oracle/lgd_oracle.py: label encoder 14, canoni 1, student_proj_2D per level, rendering
    # per level, refinement level by level (2 ReLUs each), then the adapter level by level (2 ReLUs each))
    order = rec["ln"] + rec["proj"] + rec["render"]
    for lv in range(L):
        order += [rec["gn"][0][lv], rec["gn"][1][lv]]
    for lv in range(L):
        order += [rec["adapter"][0][lv], rec["adapter"][1][lv]]
    pinned = _PinnedReluF(order)
    p = {k: v.double().requires_grad_(True) for k, v in cm.teacher_params().items()}
    pa = {k: v.double().requires_grad_(True) for k, v in cm.adapter_params().items()}
    feats = {k: v.double().requires_grad_(True) for k, v in cm.case_feats(name).items()}
    real_F = O.F
    O.F = pinned
    try:
        tea, _, _ = O.teacher_forward(p, feats, cm.case_gt(name), (H, W), ctx, interact, False, fmt)
        loss = O.distill_loss(pa, feats, tea, coef, 1)
    finally:
        O.F = real_F
    assert pinned.used == len(order)
    assert max(pinned.flip_frac) < 2e-3, pinned.flip_frac   # the masks ARE the oracle's, up to the units within rounding of zero
    pr = cm.probes(tea)
    terms = [(tea[k] * pr[k].double()).sum() for k in tea]
    ref_total = loss + sum(terms)
    # the probe sums are signed and cancel against the loss (c3_full: total -0.155 from a loss of ~2): the bar is relative to the
    # magnitude of what is summed, not to the residue
    scale = abs(float(loss.detach())) + sum(abs(float(t.detach())) for t in terms)
    assert abs(total - float(ref_total.detach())) <= 1e-5 * scale, (total, float(ref_total.detach()), scale)
    ref_total.backward()
    errs = {k: cm.rel_err(gfeat[k], feats[k].grad) for k in O.LEVELS}
    assert all(e <= 1e-4 for e in errs.values()), errs   # ALL five levels (measured ~1e-5)
    worst = (0.0, "")
    refs = dict(p)
    refs.update({"adapter." + n: v for n, v in pa.items()})
    for n, g in gw.items():
        r = refs[n].grad
        if r is None or float(r.abs().max()) < 1e-9:   # never-used parameters; adapter.4.bias in front of an InstanceNorm (analytically 0)
            assert g is None or float(g.abs().max()) < 1e-6, n
            continue
        e = cm.rel_err(g, r)
        worst = max(worst, (e, n))
        assert e <= 1e-4, (n, e)
    print("gradients under pinned masks [%s]: features %s; worst parameter %.1e (%s); flipped units <= %.1e of a site"
          % (name, " ".join("%s %.1e" % kv for kv in errs.items()), worst[0], worst[1], max(pinned.flip_frac)))


@pytest.mark.parametrize("tile", [4, 6])
def test_fcos_head_gradients_fp64_under_product_masks(tile):
    """the product FCOSHead's gradients against the oracle head in fp64 evaluated under the PRODUCT's activation masks (8 GroupNorm +
    ReLU calls over all levels, the ReLU of the regression branch): every feature and parameter gradient to 1e-4."""
    from lgd_amd import config, ops
    from lgd_amd.student.fcos import FCOSHead
    from oracle import student_oracle as SO
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_fcos_r50.yaml"), ["MODEL.DEVICE", DEV])
    head = FCOSHead(cfg)
    head.load_state_dict({k: torch.from_numpy(v) for k, v in synth.fcos_head_params(SO.fcos_head_param_shapes()).items()}, strict=True)
    head.to(DEV).train()
    feats_np, probes = cm.fcos_head_inputs()
    feats = [torch.from_numpy(f).to(DEV).requires_grad_(True) for f in feats_np]
    L = len(feats)
    gn_masks = []
    real_conv_gn = ops.conv3x3_gn

    def conv_gn(xs, *a, **k):
        # the GroupNorm + ReLU runs inside the next convolution's input transform as relu(fma(y, scale, shift)) with the raw maps y and
        # the (scale, shift) this call returns per filter (tile 6: one conv + GroupNorm node whose backward applies the GroupNorm gradient
        # in the adjoint output transform; tile 4: conv and group_norm_fold nodes); the sign of an fp32 fma is the sign of the exact
        # value, which fp64 holds (products of two floats are exact)
        out = real_conv_gn(xs, *a, **k)
        for aff, ys in out:
            N = ys[0].shape[0]
            a64 = aff.double()
            gn_masks.append([((y.detach().double() * a64[lv * N:(lv + 1) * N, :, 0, None, None] + a64[lv * N:(lv + 1) * N, :, 1, None, None]) > 0).cpu()
                             for lv, y in enumerate(ys)])
        return out
    prev = ops.conv3x3_backend(winograd=True, min_tiles=0, tile=tile)
    ops.conv3x3_gn = conv_gn
    from lgd_amd.student import retinanet
    streams_on, retinanet._HEAD_STREAMS = retinanet._HEAD_STREAMS, False   # (one stream: the masks are recorded in the interleaved call order below)
    try:
        outs = dict(zip(("logits", "reg", "ctr"), head(feats)))
        total = sum((t * torch.from_numpy(probes[kind][i]).to(DEV)).sum() for kind, maps in outs.items() for i, t in enumerate(maps))
        total.backward()
    finally:
        ops.conv3x3_gn = real_conv_gn
        ops.conv3x3_backend(*prev)
        retinanet._HEAD_STREAMS = streams_on
    assert len(gn_masks) == 8   # per tower layer: the cls tower's call, then the bbox tower's
    reg_masks = [(t.detach() > 0).cpu() for t in outs["reg"]]
    # the oracle's ReLU sites in its call order: per level -- cls tower (4), bbox tower (4), then the regression branch
    order = []
    for lv in range(L):
        order += [gn_masks[2 * k][lv] for k in range(4)] + [gn_masks[2 * k + 1][lv] for k in range(4)] + [reg_masks[lv]]
    pinned = _PinnedReluF(order)
    p = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in synth.fcos_head_params(SO.fcos_head_param_shapes()).items()}
    f64 = [torch.from_numpy(f).double().requires_grad_(True) for f in feats_np]
    real_F = SO.F
    SO.F = pinned
    try:
        ref = dict(zip(("logits", "reg", "ctr"), SO.fcos_head_forward(p, f64, cm.FCOS_STRIDES)))
    finally:
        SO.F = real_F
    assert pinned.used == len(order) and max(pinned.flip_frac) < 2e-3, pinned.flip_frac
    rt = sum((t * torch.from_numpy(probes[kind][i]).double()).sum() for kind, maps in ref.items() for i, t in enumerate(maps))
    assert abs(float(total.detach()) - float(rt.detach())) <= 1e-4 * abs(float(rt.detach()))   # a signed sum of 290k fp32 outputs
    rt.backward()
    errs = {"feature level %d" % i: cm.rel_err(a.grad, b.grad) for i, (a, b) in enumerate(zip(feats, f64))}
    errs.update({n: cm.rel_err(prm.grad, p[n].grad) for n, prm in head.named_parameters()})
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    print("FCOS head gradients under pinned masks (tile %d): worst %s; flipped units <= %.1e of a site"
          % (tile, ", ".join("%s %.1e" % kv for kv in top), max(pinned.flip_frac)))
    # the per-level `scales.i.scale` are scalars: their gradient is ONE signed sum over the level's regression map (probe in [-1, 1]),
    # so cancellation carries the fp32 rounding of its terms into the relative error (measured 1.3e-4 at the 12 x 16 level): 5e-4
    bad = {n: e for n, e in errs.items() if e > (5e-4 if n.endswith(".scale") else 1e-4)}
    assert not bad, bad


@pytest.mark.timeout(1500)
def test_full_size_multiscale_dcn_step():
    """BASELINE config 5 as its recipe states it: RetinaNet R-101-DCNv2 + LGD, 2 images per GPU whose short side is drawn per image from
    INPUT.MIN_SIZE_TRAIN (640..800, max 1333; configs/Base-RetinaNet.yaml:26), so the two images differ in size and the batch is padded
    to its own maximum.  Two optimizer steps of the shipped path (dcn.hip deformable convolutions in res3-5, F(6x6,3x3) everywhere
    else) from the same weights as the same model on the F(4x4,3x3) kernels -- two independent implementations of every 3x3
    convolution around the SAME deformable kernels, whose own arithmetic is held to the definition in test_kernels_gpu.py: every
    loss finite and equal to 2e-4 after the second step (i.e. after one full update through all gradients)."""
    import copy
    from lgd_amd import config, ops
    from lgd_amd.data import multiscale_sizes, synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_retinanet_r101_dcnv2.yaml"), ["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    base = build_model(cfg)
    twin = copy.deepcopy(base)
    sizes = multiscale_sizes(2, 800, 1333, tuple(cfg.INPUT.MIN_SIZE_TRAIN), cfg.INPUT.MAX_SIZE_TRAIN, seed=5)
    assert sizes[0] != sizes[1] and all(min(s) in cfg.INPUT.MIN_SIZE_TRAIN for s in sizes), sizes
    data = synthetic_batch(2, 800, 1333, 10, seed=3, sizes=sizes)
    d = cfg.MODEL.DISTILLATOR
    it0 = max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
    prev = ops.conv3x3_backend(winograd=True, tile=6)
    try:
        a = Trainer(cfg, base, distributed=False)
        la = [{k: float(v) for k, v in a.step(data, it0 + i).items()} for i in range(2)]
        ops.conv3x3_backend(tile=4)
        b = Trainer(cfg, twin, distributed=False)
        lb = [{k: float(v) for k, v in b.step(data, it0 + i).items()} for i in range(2)]
    finally:
        ops.conv3x3_backend(*prev)
    for i in range(2):
        for k in la[i]:
            assert np.isfinite(la[i][k]) and abs(la[i][k] - lb[i][k]) <= 2e-4 * abs(lb[i][k]) + 1e-6, (i, k, la[i][k], lb[i][k])
    print("config 5, multi-scale %s: step-2 losses F(6x6) %s / F(4x4) %s" % (sizes, la[1], lb[1]))


@pytest.mark.gpu
def test_device_prefetcher_hands_over_the_loaders_batches():
    """lgd_amd.data.DevicePrefetcher (the host -> device copy of batch k + 1 on a side stream under step k): the batches arrive on the device in
    order, equal to the host tensors, while the step's stream is kept busy; exhausted after the last one [ref: retinanet.py:48 copies inside the step]."""
    from lgd_amd.data import DevicePrefetcher, synthetic_batch
    host = [synthetic_batch(2, 64, 96, 3, seed=100 + i, pin=True) for i in range(4)]
    busy = torch.empty(1 << 24, device=DEV)
    pf = DevicePrefetcher(iter(host), DEV)
    n = 0
    for b, h in zip(pf, host):
        busy.mul_(1.0001)   # (work on the step's stream between hand-overs)
        assert len(b) == len(h)
        for x, y in zip(b, h):
            assert x["image"].is_cuda and torch.equal(x["image"].cpu(), y["image"])
            assert torch.equal(x["instances"].gt_boxes.tensor.cpu(), y["instances"].gt_boxes.tensor)
            assert torch.equal(x["instances"].gt_classes.cpu(), y["instances"].gt_classes)
            assert (x["height"], x["width"]) == (y["height"], y["width"])
        n += 1
    assert n == 4
    with pytest.raises(StopIteration):
        next(pf)
