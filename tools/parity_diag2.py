"""Run the ORACLE's torch ops on the GPU (MIOpen/rocBLAS/native kernels) and compare its grads to the golden:
separates 'torch-on-ROCm numerical behaviour' from 'our HIP kernels'."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common as cm
from oracle import lgd_oracle as O
name = "c1_ctx_stuguided"
B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
g = cm.golden(name)
dev = sys.argv[1] if len(sys.argv) > 1 else "cuda"
p = {k: v.to(dev).requires_grad_(True) for k, v in cm.teacher_params().items()}
pa = {k: v.to(dev).requires_grad_(True) for k, v in cm.adapter_params().items()}
feats = {k: v.to(dev).requires_grad_(True) for k, v in cm.case_feats(name).items()}
tea, _, _ = O.teacher_forward(p, feats, cm.case_gt(name), (H, W), ctx, interact, False, fmt)
loss = O.distill_loss(pa, feats, tea, coef, 1)
pr = cm.probes(tea)
which = sys.argv[2] if len(sys.argv) > 2 else "both"
total = (loss if which in ("both", "distill") else 0) + (sum((tea[k] * pr[k].to(dev)).sum() for k in tea) if which in ("both", "probe") else 0)
total.backward()
for k in O.LEVELS:
    print(k, "oracle-on-%s gfeat rel err vs golden %.2e  |grad| %.3e" % (dev, cm.rel_err(cm.sample(feats[k].grad)[0], g["gfeat_s_" + k]), float(feats[k].grad.norm())))
