"""Per-op backward accuracy on REAL chain data: refine(x) = conv-GN-ReLU-conv-GN-ReLU-conv-GN.
CPU chain gives each op's (input, upstream grad, input grad); each op is then replayed alone on the GPU."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import common as cm
from lgd_amd import synth
p = cm.teacher_params()
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
for hw in (64, 32, 16):
    x0 = torch.relu(torch.from_numpy(synth.det_uniform((2, 256, hw, hw), 5)) * 3)
    dy = torch.from_numpy(synth.det_uniform((2, 256, hw, hw), 6, -1e-3, 1e-3))
    stages = []
    x = x0.clone().requires_grad_(True)
    cur = x
    for i, act in ((0, True), (3, True), (6, False)):
        w, b = p["refinement_module.%d.weight" % i], p["refinement_module.%d.bias" % i]
        c = F.conv2d(cur, w, b, padding=1); c.retain_grad()
        n = F.group_norm(c, 1, eps=1e-5); n.retain_grad()
        a = F.relu(n) if act else n
        if act: a.retain_grad()
        stages.append((cur, w, b, c, n, a, act))
        cur = a
    cur.backward(dy)
    print("hw", hw)
    for si, (inp, w, b, c, n, a, act) in enumerate(stages):
        # conv dgrad/wgrad alone on the GPU with the CPU's input and upstream grad
        ig = inp.detach().cuda().requires_grad_(True); wg = w.detach().cuda().requires_grad_(True)
        cg = F.conv2d(ig, wg, b.cuda(), padding=1)
        cg.backward(c.grad.cuda())
        inp_grad_cpu = inp.grad if inp.grad is not None else None
        # GN(+ReLU) backward alone on the GPU
        c2 = c.detach().cuda().requires_grad_(True)
        n2 = F.group_norm(c2, 1, eps=1e-5)
        up = (a.grad if act else dy)
        (F.relu(n2) if act else n2).backward(up.cuda())
        print("  stage %d: conv fwd %.1e dgrad %.1e | gn fwd %.1e gn(+relu) bwd %.1e" % (
            si, rel(cg, c), rel(ig.grad, inp_grad_cpu) if inp_grad_cpu is not None else -1, rel(n2, n), rel(c2.grad, c.grad)))
