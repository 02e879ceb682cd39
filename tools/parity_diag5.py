"""Count ReLU-mask disagreements between the oracle's forward on CPU and on GPU (same torch ops)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import common as cm
from oracle import lgd_oracle as O
name = "c1_ctx_stuguided"
B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
def run(dev):
    p = {k: v.to(dev) for k, v in cm.teacher_params().items()}
    feats = {k: v.to(dev) for k, v in cm.case_feats(name).items()}
    with torch.no_grad():
        tea, _, _, inter = O.teacher_forward(p, feats, cm.case_gt(name), (H, W), ctx, interact, False, fmt, return_intermediates=True)
        masks = {}
        for i, k in enumerate(O.LEVELS):
            masks["proj[%s]" % k] = (inter["proj"][k] > 0).cpu()
            masks["raw[%s]" % k] = (inter["raw"][i] > 0).cpu()
            x = inter["raw"][i]
            for j, idx in enumerate((0, 3)):
                x = F.relu(F.group_norm(F.conv2d(x, p["refinement_module.%d.weight" % idx], p["refinement_module.%d.bias" % idx], padding=1), 1, eps=1e-5))
                masks["refine%d[%s]" % (j, k)] = (x > 0).cpu()
    return masks
a, b = run("cpu"), run("cuda")
for k in a:
    n = int((a[k] != b[k]).sum())
    if n: print("%-14s %d of %d activation masks differ" % (k, n, a[k].numel()))
print("total flips:", sum(int((a[k] != b[k]).sum()) for k in a))
