"""Generate golden vectors by running the REAL reference hot path (imported from
/root/reference with detectron2 stubs, see _ref_import.py) on closed-form inputs.

Run in the build container only:   python tests/golden/make_golden.py
Outputs tests/golden/*.npz (data only: inputs are regenerated from lgd_amd.synth,
the files hold the reference's outputs).  The reference itself never travels.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from _ref_import import import_reference, import_reference_fcos, make_cfg  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
import common as cm  # noqa: E402  (tests/common.py: the closed-form inputs the tests regenerate)
from lgd_amd import synth  # noqa: E402
from oracle import lgd_oracle as O  # noqa: E402

torch.set_num_threads(8)
SAMPLE_STRIDE = 37
ONLY = set()  # --only a,b: regenerate just these cases
FULL_STRIDE = 997  # full-size cases (11.5 M teacher-feature elements)


class _Boxes:
    def __init__(self, t):
        self.tensor = t
        self.device = t.device


class _Instances:
    def __init__(self, boxes, classes):
        self.gt_boxes = _Boxes(boxes)
        self.gt_classes = classes

    def __len__(self):
        return int(self.gt_boxes.tensor.shape[0])


def to_batched_inputs(gt, h, w):
    return [{"image": torch.zeros(3, h, w), "instances": _Instances(torch.from_numpy(b.copy()), torch.from_numpy(c.copy()))}
            for b, c in gt]


def load_params(module, shapes, gain=1.0):
    params = synth.closed_form_params(shapes, gain)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    missing, unexpected = module.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    return sd


def sample(t, stride=SAMPLE_STRIDE):
    f = t.detach().reshape(-1).double()
    return dict(s=f[::stride].float().numpy(), sum=np.float64(f.sum()), sq=np.float64((f * f).sum()))


def pack_rects(masks_level, hw):
    """reference masks (N,HW) -> int32 (N,4) [x0,x1,y0,y1]; asserts each mask is an axis-aligned rectangle."""
    out = []
    for m in masks_level:
        m2 = m.reshape(-1, hw[0], hw[1]) > 0
        for i in range(m2.shape[0]):
            ys = torch.nonzero(m2[i].any(1)).flatten()
            xs = torch.nonzero(m2[i].any(0)).flatten()
            if len(ys) == 0:
                out.append([0, -1, 0, -1])
                continue
            x0, x1, y0, y1 = int(xs[0]), int(xs[-1]), int(ys[0]), int(ys[-1])
            assert int(m2[i].sum()) == (x1 - x0 + 1) * (y1 - y0 + 1), "reference mask is not a rectangle"
            out.append([x0, x1, y0, y1])
    return np.array(out, np.int32).reshape(-1, 4)


def run_teacher_case(ref, name, B, H, W, gt, add_ctx, interact, box_format="x1y1x2y2", with_grads=False, coef=1.0,
                     feat_seed=11, stride=SAMPLE_STRIDE, full_small_levels=True, with_loss=False):
    """with_grads: distill losses + total loss + gradients; with_loss: the losses only (full-size cases: no backward);
    stride: sampling stride of the stored teacher features; full_small_levels: also store p6/p7 and their masks whole."""
    if ONLY and name not in ONLY:
        return
    cfg = make_cfg(add_ctx=add_ctx, interact=interact, box_format=box_format, coef=coef)
    teacher = ref.DynamicTeacher(cfg)
    load_params(teacher, O.teacher_param_shapes())
    teacher.train()
    feats_np = synth.synth_features(B, H, W, seed=feat_seed)
    feats = {k: torch.from_numpy(v.copy()).requires_grad_(with_grads) for k, v in feats_np.items()}
    images = types.SimpleNamespace(tensor=torch.zeros(B, 3, H, W), image_sizes=[(H, W)] * B)
    bi = to_batched_inputs(gt, H, W)

    cap = {}
    h1 = teacher.label_encoder_.register_forward_hook(lambda m, i, o: cap.__setitem__("le", o))
    attn_outs = []
    h2 = teacher.multi_head_attn.register_forward_hook(lambda m, i, o: attn_outs.append(o[0].squeeze(1)))
    app_outs = []
    orig_agg = teacher.aggregate_per_level

    def agg(f, m):
        r = orig_agg(f, m)
        app_outs.append(torch.cat(r, 0))
        return r
    teacher.aggregate_per_level = agg

    tea, inst_labels, masks = teacher((bi, images, None, feats))
    h1.remove()
    h2.remove()

    label_embed, m1, m2, boxlists, _, _ = cap["le"]
    out = {}
    descs, _, _ = ref.box_descriptor_encode([b["instances"] for b in bi], H, W, 80, "one_hot", box_format, add_ctx)
    out["descs"] = torch.cat(descs, 0).numpy()
    out["counts"] = np.array([len(b) for b in boxlists], np.int32)
    out["boxlists"] = np.array([r for bl in boxlists for r in bl], np.float64)
    out["inst_labels"] = torch.cat([l.reshape(-1) for l in inst_labels]).numpy().astype(np.int64)
    out["label_embed"] = label_embed.detach().numpy()
    out["stn_desc_sample"] = m1.detach().reshape(-1)[::SAMPLE_STRIDE].numpy()
    out["stn_feat_sample"] = m2.detach().reshape(-1)[::SAMPLE_STRIDE].numpy()
    keys = list(feats.keys())
    hws = [tuple(feats[k].shape[-2:]) for k in keys]
    for i, k in enumerate(keys):
        out["rects_" + k] = pack_rects(masks[i], hws[i])
        out["app_" + k] = app_outs[i].detach().numpy()
        if attn_outs:
            out["attn_" + k] = attn_outs[i].detach().numpy()
        if k in ("p6", "p7") and full_small_levels:
            out["tea_full_" + k] = tea[k].detach().numpy()
            out["mask_bits_" + k] = np.packbits(torch.cat(list(masks[i]), 0).numpy().astype(np.uint8), axis=None)
        s = sample(tea[k], stride)
        out["tea_s_" + k], out["tea_sum_" + k], out["tea_sq_" + k] = s["s"], s["sum"], s["sq"]

    if with_grads or with_loss:
        # loss = loss_distill(flag=1) + sum(teacher feats * probe)   (SURVEY.md section 8c viii)
        class D(ref.BaseDistillator):
            def __init__(self):
                torch.nn.Module.__init__(self)
                self.norm_stu = torch.nn.InstanceNorm2d(256, affine=False)
                self.norm_tea = torch.nn.InstanceNorm2d(256, affine=False)
                self.coef = coef
                self.adapter = torch.nn.ModuleDict({"distill": ref.SequentialConvs(cfg)})
        d = D()
        load_params(d.adapter["distill"], O.adapter_param_shapes())
        for flag in (0, 1):
            d.distill_flag = flag
            loss = d.distill({"stu": feats, "tea": tea}, None, None, None, None)
            out["loss_distill_flag%d" % flag] = np.float64(loss.item())
        probe = {k: torch.from_numpy(synth.det_uniform(tuple(tea[k].shape), 900 + i, -1e-3, 1e-3)) for i, k in enumerate(keys)}
        total = loss + sum((tea[k] * probe[k]).sum() for k in keys)
        out["total_loss"] = np.float64(total.item())
        if not with_grads:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
            print(name, "T=%d" % int(out["counts"].sum()), "saved", len(out), "arrays (no grads)")
            return
        total.backward()
        for k in keys:
            g = sample(feats[k].grad, stride)
            out["gfeat_s_" + k], out["gfeat_sq_" + k] = g["s"], g["sq"]
        for n, prm in list(teacher.named_parameters()) + [("adapter." + n, q) for n, q in d.adapter["distill"].named_parameters()]:
            if prm.grad is None:
                out["gnone_" + n] = np.int32(1)
                continue
            g = sample(prm.grad)
            out["gw_s_" + n], out["gw_sq_" + n] = g["s"][:64], g["sq"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "T=%d" % int(out["counts"].sum()), "saved", len(out), "arrays")


def run_mask_case(ref, name, H, W, gt, add_ctx):
    """masks only at a BASELINE config-2 shape: rect bounds for every (level, box)."""
    if ONLY and name not in ONLY:
        return
    descs, boxlists, _ = ref.box_descriptor_encode([b["instances"] for b in to_batched_inputs(gt, H, W)], H, W, 80,
                                                   "one_hot", "x1y1x2y2", add_ctx)
    out = {"counts": np.array([len(b) for b in boxlists], np.int32),
           "boxlists": np.array([r for bl in boxlists for r in bl], np.float64)}
    for i, hw in enumerate(synth.pyramid_shapes(H, W)):
        masks = [ref.get_inside_gt_mask(bl, ref.resolution(H, W), ref.resolution(*hw), "cpu") for bl in boxlists]
        out["rects_p%d" % (i + 3)] = pack_rects(masks, hw)
        out["cnt_p%d" % (i + 3)] = torch.cat([m.sum(-1) for m in masks]).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "saved")


def run_distill_case(ref, name, B, H, W, coef):
    """BaseDistillator.distill alone on independent closed-form student/teacher pyramids."""
    if ONLY and name not in ONLY:
        return
    cfg = make_cfg(coef=coef)

    class D(ref.BaseDistillator):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.norm_stu = torch.nn.InstanceNorm2d(256, affine=False)
            self.norm_tea = torch.nn.InstanceNorm2d(256, affine=False)
            self.coef = coef
            self.adapter = torch.nn.ModuleDict({"distill": ref.SequentialConvs(cfg)})
    d = D()
    load_params(d.adapter["distill"], O.adapter_param_shapes())
    stu = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in synth.synth_features(B, H, W, seed=21).items()}
    tea = {k: torch.from_numpy(v.copy() * 2.0 + 0.5) for k, v in synth.synth_features(B, H, W, seed=22).items()}
    out = {}
    for flag in (0, 1):
        d.distill_flag = flag
        d.zero_grad()
        for v in stu.values():
            v.grad = None
        loss = d.distill({"stu": stu, "tea": tea}, None, None, None, None)
        loss.backward()
        out["loss_flag%d" % flag] = np.float64(loss.item())
        out["stu_grad_is_none_flag%d" % flag] = np.int32(all(v.grad is None for v in stu.values()))
        if flag == 1:
            for k, v in stu.items():
                g = sample(v.grad)
                out["gstu_s_" + k], out["gstu_sq_" + k] = g["s"], g["sq"]
        for n, prm in d.adapter["distill"].named_parameters():
            g = sample(prm.grad)
            out["gw%d_s_%s" % (flag, n)], out["gw%d_sq_%s" % (flag, n)] = g["s"][:64], g["sq"]
    # loss tail alone (IN + MSE on given maps): the HIP kernel K4 contract
    a = {k: torch.from_numpy(v.copy() * 1.5 - 0.25) for k, v in synth.synth_features(B, H, W, seed=23).items()}
    keys = sorted(a.keys())
    d.adapter["distill"] = torch.nn.Identity()
    d.distill_flag = 1
    out["in_mse"] = np.float64(d.distill({"stu": a, "tea": tea}, None, None, None, None).item())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "saved", {k: float(v) for k, v in out.items() if k.startswith("loss") or k == "in_mse"})


def run_fcos_gt_case(fc, name, case, radius):
    """the REAL FCOS.get_ground_truth [thirdparty_heads/fcos.py:177-284] (through the stand-ins _ref_import documents: the arithmetic of
    Shift2BoxTransform.get_deltas and Boxes.get_centers / area is restated there, everything else is the reference's)."""
    if ONLY and name not in ONLY:
        return
    from oracle import student_oracle as SO
    H, W, gts = cm.fcos_gt_inputs(case)
    level_hw = synth.pyramid_shapes(H, W)
    shifts = SO.fcos_shifts(level_hw, cm.FCOS_STRIDES)
    me = types.SimpleNamespace(object_sizes_of_interest=cm.FCOS_SOI, shift2box_transform=fc.Shift2BoxTransform((1.0, 1.0, 1.0, 1.0)),
                               center_sampling_radius=radius, fpn_strides=cm.FCOS_STRIDES, num_classes=80)

    class T:
        def __init__(self, b, c):
            self.gt_boxes, self.gt_classes = fc.Boxes(torch.from_numpy(b.copy())), torch.from_numpy(c.copy())

        def __len__(self):
            return len(self.gt_boxes)
    targets = [T(b, c) for b, c in gts]
    cls, deltas, ctr = fc.FCOS.get_ground_truth(me, [shifts for _ in gts], targets)
    fg = (cls >= 0) & (cls != 80)
    out = {"classes": cls.numpy().astype(np.uint8), "fg_deltas": deltas[fg].numpy(), "fg_centerness": ctr[fg].numpy(),
           "n_fg": np.int64(int(fg.sum()))}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "R=%d" % cls.shape[1], "foreground", int(fg.sum()), "saved")


def run_fcos_head_case(fc, name, B):
    """the REAL FCOSHead [thirdparty_heads/fcos.py:433-546] (pure torch; ShiftGenerator is a name-only stand-in) on closed-form
    parameters and features: outputs of every level and the gradients of loss = sum_l <out_l, probe_l>."""
    if ONLY and name not in ONLY:
        return
    from oracle import student_oracle as SO
    NS = types.SimpleNamespace
    cfg = NS(MODEL=NS(FCOS=NS(NUM_CLASSES=80, NUM_CONVS=4, PRIOR_PROB=0.01, FPN_STRIDES=cm.FCOS_STRIDES, CENTERNESS_ON_REG=True,
                              NORM_REG_TARGETS=True)))
    head = fc.FCOSHead(cfg, [fc.ShapeSpec(channels=256) for _ in cm.FCOS_HEAD_LEVELS])
    shapes = SO.fcos_head_param_shapes()
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.fcos_head_params(shapes).items()}
    missing, unexpected = head.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    head.train()
    feats_np, probes = cm.fcos_head_inputs(B)
    feats = [torch.from_numpy(f.copy()).requires_grad_(True) for f in feats_np]
    logits, regs, ctrs = head(feats)
    out, total = {}, 0.0
    for kind, maps in (("logits", logits), ("reg", regs), ("ctr", ctrs)):
        for i, t in enumerate(maps):
            s = sample(t, 7)
            out["%s_s_%d" % (kind, i)], out["%s_sum_%d" % (kind, i)], out["%s_sq_%d" % (kind, i)] = s["s"], s["sum"], s["sq"]
            if i >= 3:
                out["%s_full_%d" % (kind, i)] = t.detach().numpy()
            total = total + (t * torch.from_numpy(probes[kind][i])).sum()
    out["total"] = np.float64(total.item())
    total.backward()
    for i, f in enumerate(feats):
        g = sample(f.grad, 7)
        out["gfeat_s_%d" % i], out["gfeat_sq_%d" % i] = g["s"], g["sq"]
    for n, prm in head.named_parameters():
        g = sample(prm.grad)
        out["gw_s_" + n], out["gw_sq_" + n] = g["s"][:64], g["sq"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "saved", len(out), "arrays; total %.6f" % out["total"])


def to_t(gt):
    return [(b, c) for b, c in gt]


def main():
    ref = import_reference()
    # C1: BASELINE config 1 -- B=2, 512x512, 10 boxes/img from the edge-case table, ctx=YES, stuGuided, with grads
    gt = synth.synth_gt(2, 512, 512, 10, table=True)
    run_teacher_case(ref, "c1_ctx_stuguided", 2, 512, 512, gt, True, "stuGuided", with_grads=True, coef=1.0)
    # C1b: ctx=NO (FCOS R-50 config), labelGuided, unequal Ni, one EMPTY-GT image, x1y1wh boxes
    gt_b = synth.synth_gt(3, 384, 512, 7, seed=9)
    gt_b[1] = (np.zeros((0, 4), np.float32), np.zeros((0,), np.int64))
    gt_b[2] = (gt_b[2][0][:4], gt_b[2][1][:4])
    run_teacher_case(ref, "c1b_noctx_labelguided_wh", 3, 384, 512, gt_b, False, "labelGuided", box_format="x1y1wh", with_grads=True)
    # C1c: ctx=NO stuGuided (the shipped FCOS R-50 configuration), non-square, with grads, coef != 1
    gt_c = synth.synth_gt(2, 320, 480, 6, seed=4)
    run_teacher_case(ref, "c1c_noctx_stuguided", 2, 320, 480, gt_c, False, "stuGuided", with_grads=True, coef=2.5, feat_seed=13)
    # mask-only at the config-2 shape (800x1344), random + table boxes
    gt2 = synth.synth_gt(8, 800, 1344, 10, seed=0)
    run_mask_case(ref, "c2_masks_800x1344", 800, 1344, gt2, True)
    # the full BASELINE shape, B=2 800x1344, 10 boxes/img: config 2 (ctx=YES) and config 3 (ctx=NO)
    gt_f = synth.synth_gt(2, 800, 1344, 10, seed=0)
    run_teacher_case(ref, "c2_full_ctx_800x1344", 2, 800, 1344, gt_f, True, "stuGuided", with_grads=True, feat_seed=17,
                     stride=FULL_STRIDE, full_small_levels=False)
    run_teacher_case(ref, "c3_full_noctx_800x1344", 2, 800, 1344, gt_f, False, "stuGuided", with_grads=True, feat_seed=19,
                     stride=FULL_STRIDE, full_small_levels=False)
    # distill alone
    run_distill_case(ref, "distill_c1", 2, 512, 512, coef=1.0)
    run_distill_case(ref, "distill_coef", 2, 256, 320, coef=0.37)
    # the FCOS code that IS in the reference tree: target assignment and the head
    fc = import_reference_fcos()
    for name, (case, radius) in cm.FCOS_GT_CASES.items():
        run_fcos_gt_case(fc, name, case, radius)
    run_fcos_head_case(fc, "fcos_head", 2)


if __name__ == "__main__":
    if "--only" in sys.argv:
        ONLY = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    main()
