"""Print GPU-vs-golden relative errors of the product teacher (forward samples, loss, feature grads)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common as cm
import test_model_gpu as T
from oracle import lgd_oracle as O
from lgd_amd.structures import ImageList
name = sys.argv[1] if len(sys.argv) > 1 else "c1_ctx_stuguided"
B, H, W, ctx, interact, fmt, coef, _ = cm.CASES[name]
g = cm.golden(name)
teacher = T._teacher(name)
feats = {k: v.cuda().requires_grad_(True) for k, v in cm.case_feats(name).items()}
images = ImageList(torch.zeros(B, 3, H, W, device="cuda"), [(H, W)] * B)
tea, _, geom = teacher((T._batched_inputs(cm.case_gt(name), H, W), images, None, feats))
for k in O.LEVELS:
    print(k, "tea feat rel err %.2e" % cm.rel_err(cm.sample(tea[k])[0], g["tea_s_" + k]))
if "total_loss" in g:
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd import ops
    ad = SequentialConvs(None); ad.load_state_dict(cm.adapter_params()); ad.cuda()
    keys = sorted(tea)
    loss = ops.distill_in_mse([ad(feats[k]) for k in keys], [tea[k].detach() for k in keys], coef)
    print("loss rel err %.2e" % (abs(loss.item() - float(g["loss_distill_flag1"])) / float(g["loss_distill_flag1"])))
    pr = cm.probes({k: tea[k] for k in O.LEVELS})
    total = loss + sum((tea[k] * pr[k].cuda()).sum() for k in O.LEVELS)
    total.backward()
    for k in O.LEVELS:
        print(k, "gfeat rel err %.2e" % cm.rel_err(cm.sample(feats[k].grad)[0], g["gfeat_s_" + k]))
    for n, p in teacher.named_parameters():
        if p.grad is not None and "gw_s_" + n in g and n.endswith("weight") and ("proj" in n or "refine" in n or "attn" in n):
            print(n, "gw rel err %.2e" % cm.rel_err(cm.sample(p.grad)[0][:64], g["gw_s_" + n]))
