"""Find the first backward intermediate where oracle-on-GPU departs from oracle-on-CPU."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common as cm
from oracle import lgd_oracle as O
name = "c1_ctx_stuguided"
B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
def run(dev, which):
    p = {k: v.to(dev).requires_grad_(True) for k, v in cm.teacher_params().items()}
    pa = {k: v.to(dev).requires_grad_(True) for k, v in cm.adapter_params().items()}
    feats = {k: v.to(dev).requires_grad_(True) for k, v in cm.case_feats(name).items()}
    tea, _, _, inter = O.teacher_forward(p, feats, cm.case_gt(name), (H, W), ctx, interact, False, fmt, return_intermediates=True)
    keep = {}
    for nm in ("app", "att", "raw"):
        for i, t in enumerate(inter[nm]):
            t.retain_grad(); keep["%s[%d]" % (nm, i)] = t
    for k, t in inter["proj"].items():
        t.retain_grad(); keep["proj[%s]" % k] = t
    for k, t in tea.items():
        t.retain_grad(); keep["tea[%s]" % k] = t
    inter["canoni"].retain_grad(); keep["canoni"] = inter["canoni"]
    inter["label_embed"].retain_grad(); keep["label_embed"] = inter["label_embed"]
    loss = O.distill_loss(pa, feats, tea, coef, 1)
    pr = cm.probes(tea)
    total = (loss if which != "probe" else 0) + (sum((tea[k] * pr[k].to(dev)).sum() for k in tea) if which != "distill" else 0)
    total.backward()
    out = {k: v.grad.detach().cpu() for k, v in keep.items() if v.grad is not None}
    out.update({"feat[%s]" % k: v.grad.detach().cpu() for k, v in feats.items()})
    return out
for which in ("distill", "probe"):
    a, b = run("cpu", which), run("cuda", which)
    print("====", which)
    for k in a:
        print("%-14s gpu-vs-cpu rel err %.2e   |g| %.3e" % (k, cm.rel_err(b[k], a[k]), float(a[k].norm())))
