import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch.nn.functional as F
from lgd_amd import ops, synth
DEV='cuda'
def run(h2, N, C, hws, seed=1):
    ops.conv3x3_backend(winograd=True, min_tiles=0, tile=6); ops.h2_backend(h2, force=h2); ops.gemm3_backend(True, force=True)
    xs=[torch.from_numpy(synth.det_uniform((N,C,h,w), seed+i, -2.0, 3.0)).to(DEV).requires_grad_(True) for i,(h,w) in enumerate(hws)]
    w=(torch.from_numpy(synth.det_uniform((C,C,3,3), seed+50, -1.0, 1.0))*(2.0/(9*C))**0.5).to(DEV).requires_grad_(True)
    gys=[torch.from_numpy(synth.det_uniform((N,C,h,w_), seed+70+i, -1.0, 1.0)).to(DEV) for i,(h,w_) in enumerate(hws)]
    ys=ops.conv3x3_levels(xs, w, None)
    torch.autograd.backward(ys, gys)
    xr=[x.detach().double().cpu().requires_grad_(True) for x in xs]; wr=w.detach().double().cpu().requires_grad_(True)
    yr=[F.conv2d(x, wr, None, 1, 1) for x in xr]
    torch.autograd.backward(yr, [g.double().cpu() for g in gys])
    e=lambda a,b: float((a.detach().cpu().double()-b.detach()).abs().max()/b.detach().abs().max())
    r=lambda a,b: float((a.detach().cpu().double()-b.detach()).norm()/b.detach().norm())
    print("h2=%d N=%d C=%d %s: y max %.2e l2 %.2e | dx max %.2e l2 %.2e | dw max %.2e l2 %.2e" % (h2,N,C,hws, max(e(a,b) for a,b in zip(ys,yr)), max(r(a,b) for a,b in zip(ys,yr)),
          max(e(x.grad,q.grad) for x,q in zip(xs,xr)), max(r(x.grad,q.grad) for x,q in zip(xs,xr)), e(w.grad, wr.grad), r(w.grad, wr.grad)))
for cfg in [(3,256,[(12,16),(6,7),(2,3)]), (2,64,[(20,28),(13,21),(7,11)]), (2,256,[(64,64),(32,32)]), (1,256,[(2,3)]), (3,256,[(12,16)])]:
    for h2 in (True, False): run(h2,*cfg)
