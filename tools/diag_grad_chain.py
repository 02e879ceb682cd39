"""where does the product's feature gradient leave the exact (fp64) gradient?  Stage-by-stage gradients of
loss = distill + sum(tea * probe) on golden case c1: product (HIP) vs oracle fp32 on the GPU vs oracle fp64 on the GPU."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common as cm  # noqa: E402
from lgd_amd import config, ops  # noqa: E402
from oracle import lgd_oracle as O  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c1_ctx_stuguided"
backend = sys.argv[2] if len(sys.argv) > 2 else "winograd"
part = sys.argv[3] if len(sys.argv) > 3 else "both"  # probe | distill | both
ops.conv3x3_backend(winograd=(backend == "winograd"), min_tiles=0)
B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
DEV = "cuda"


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def oracle(dt):
    p = {k: v.to(DEV, dt).requires_grad_(True) for k, v in cm.teacher_params().items()}
    pa = {k: v.to(DEV, dt).requires_grad_(True) for k, v in cm.adapter_params().items()}
    feats = {k: v.to(DEV, dt).requires_grad_(True) for k, v in cm.case_feats(name).items()}
    tea, _, _, inter = O.teacher_forward(p, feats, cm.case_gt(name), (H, W), ctx, interact, False, fmt, return_intermediates=True)
    stages = {"label_embed": inter["label_embed"]}
    for i, k in enumerate(O.LEVELS):
        stages["proj_" + k], stages["app_" + k], stages["att_" + k], stages["raw_" + k] = inter["proj"][k], inter["app"][i], inter["att"][i], inter["raw"][i]
        stages["tea_" + k] = tea[k]
    for t in stages.values():
        t.retain_grad()
    total = 0
    if part in ("distill", "both"):
        total = total + O.distill_loss(pa, feats, tea, coef, 1)
    if part in ("probe", "both"):
        pr = cm.probes(tea)
        total = total + sum((tea[k] * pr[k].to(DEV, dt)).sum() for k in tea)
    total.backward()
    out = {k: v.grad for k, v in stages.items()}
    out.update({"feat_" + k: feats[k].grad for k in feats})
    out.update({"w_" + k: v.grad for k, v in p.items() if v.grad is not None})
    out.update({"wa_" + k: v.grad for k, v in pa.items() if v.grad is not None})
    return out, {k: v.detach() for k, v in stages.items()}


def product():
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.base_distillator import BaseDistillator
    from lgd_amd.dynamic_teacher import DynamicTeacher
    from lgd_amd.structures import Boxes, ImageList, Instances
    cfg = config.setup_cfg(None, ["MODEL.DEVICE", DEV, "MODEL.META_ARCHITECTURE", "RetinaNet",
                                  "MODEL.DISTILLATOR.STUDENT.META_ARCH", "RetinaNetCT", "MODEL.DISTILLATOR.TEACHER.META_ARCH", "DynamicTeacher",
                                  "MODEL.DISTILLATOR.TEACHER.ADD_CONTEXT_BOX", str(ctx), "MODEL.DISTILLATOR.TEACHER.INTERACT_PATTERN", interact,
                                  "MODEL.DISTILLATOR.LABEL_ENCODER.BOX_FORMAT", fmt, "MODEL.DISTILLATOR.LAMBDA", str(coef)])
    t = DynamicTeacher(cfg)
    t.load_state_dict(cm.teacher_params(), strict=True)
    t.to(DEV).train()
    feats = {k: v.to(DEV).requires_grad_(True) for k, v in cm.case_feats(name).items()}
    images = ImageList(torch.zeros(B, 3, H, W, device=DEV), [(H, W)] * B)
    bi = [{"image": torch.zeros(3, H, W), "instances": Instances((H, W), gt_boxes=Boxes(b.clone()), gt_classes=c.clone())} for b, c in cm.case_gt(name)]
    cap = {}
    real_pool, real_mha, real_render = ops.gn_relu_mask_pool, ops.mha_blockdiag, t.rendering
    ops.gn_relu_mask_pool = lambda *a, **k: cap.setdefault("app", real_pool(*a, **k))
    ops.mha_blockdiag = lambda *a, **k: cap.setdefault("att", real_mha(*a, **k))
    t.rendering = lambda *a, **k: cap.setdefault("raw", real_render(*a, **k))
    h = t.label_encoder_.register_forward_hook(lambda m, i, o: cap.__setitem__("le", o[0]))
    tea, _, _ = t((bi, images, None, feats))
    h.remove()
    ops.gn_relu_mask_pool, ops.mha_blockdiag = real_pool, real_mha
    for k in ("app", "att", "le"):
        cap[k].retain_grad()
    for r in cap["raw"]:
        r.retain_grad()
    for k in tea:
        tea[k].retain_grad()

    class D(BaseDistillator):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.coef = coef
            self.adapter = torch.nn.ModuleDict({"distill": SequentialConvs(None)})
    d = D()
    d.adapter["distill"].load_state_dict(cm.adapter_params(), strict=True)
    d.to(DEV)
    d.distill_flag = 1
    total = 0
    if part in ("distill", "both"):
        total = total + d.distill({"stu": feats, "tea": tea}, None, None, None, None)
    if part in ("probe", "both"):
        pr = cm.probes({k: tea[k] for k in O.LEVELS})
        total = total + sum((tea[k] * pr[k].to(DEV)).sum() for k in O.LEVELS)
    total.backward()
    out = {"label_embed": cap["le"].grad}
    for i, k in enumerate(O.LEVELS):
        out["app_" + k], out["att_" + k], out["raw_" + k], out["tea_" + k] = cap["app"].grad[i], cap["att"].grad[i], cap["raw"][i].grad, tea[k].grad
        out["feat_" + k] = feats[k].grad
    out.update({"w_" + n: q.grad for n, q in t.named_parameters() if q.grad is not None})
    out.update({"wa_" + n: q.grad for n, q in d.adapter["distill"].named_parameters() if q.grad is not None})
    vals = {"label_embed": cap["le"].detach()}
    for i, k in enumerate(O.LEVELS):
        vals["app_" + k], vals["att_" + k], vals["raw_" + k], vals["tea_" + k] = cap["app"][i].detach(), cap["att"][i].detach(), cap["raw"][i].detach(), tea[k].detach()
    return out, vals


g64, v64 = oracle(torch.float64)
g32, v32 = oracle(torch.float32)
gp, vp = product()
print("case %s backend %s loss part %s" % (name, backend, part))
print("%-34s | values: oracle32-gpu  product | grads: oracle32-gpu  product   (all vs oracle fp64 on the GPU)" % "stage")
for k in g64:
    if k.startswith(("w_", "wa_")):
        continue
    line = "%-34s |" % k
    line += "  %9.1e %9.1e |" % (rel(v32[k], v64[k]), rel(vp[k], v64[k])) if k in vp and k in v64 else "  %9s %9s |" % ("", "")
    line += "  %9.1e %9.1e" % (rel(g32[k], g64[k]), rel(gp[k], g64[k]) if k in gp else float("nan"))
    print(line)
worst = sorted(((rel(gp[k], g64[k]), rel(g32[k], g64[k]), k) for k in g64 if k.startswith(("w_", "wa_")) and k in gp and float(g64[k].abs().max()) > 1e-12), reverse=True)
for e, e32, k in worst[:12]:
    print("weight grad %-45s product %.1e   oracle32-gpu %.1e" % (k, e, e32))
