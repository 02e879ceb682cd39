import torch, torch.nn.functional as F
torch.manual_seed(0)
for hw in (64, 32, 16, 8):
    x = torch.randn(2, 256, hw, hw); w = torch.randn(256, 256, 3, 3) * 0.02; b = torch.randn(256) * 0.1
    g = torch.randn(2, 256, hw, hw)
    def run(dev, dt):
        xx = x.detach().clone().to(dev, dt).requires_grad_(True); ww = w.detach().clone().to(dev, dt).requires_grad_(True); bb = b.detach().clone().to(dev, dt).requires_grad_(True)
        y = F.conv2d(xx, ww, bb, padding=1)
        y.backward(g.to(dev, dt))
        x2 = x.detach().clone().to(dev, dt).requires_grad_(True)
        z = F.relu(F.group_norm(x2, 1, eps=1e-5)); z.backward(g.to(dev, dt))
        return [t.detach().double().cpu() for t in (y, xx.grad, ww.grad, bb.grad, z, x2.grad)]
    ref = run("cpu", torch.float64)
    for dev in ("cpu", "cuda"):
        out = run(dev, torch.float32)
        names = ["conv fwd", "conv dgrad", "conv wgrad", "conv bgrad", "gn fwd", "gn bwd"]
        print(hw, dev, "  ".join("%s %.1e" % (n, float((o - r).norm() / r.norm())) for n, o, r in zip(names, out, ref)))
