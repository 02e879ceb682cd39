"""bitwise run-to-run reproducibility of the building blocks of one conv3x3 (diagnostic)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lgd_amd import ops  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


torch.manual_seed(0)
for tag, N, hws in (("test-size pyramid", 2, [(32, 40), (16, 20), (8, 10), (4, 5), (2, 3)]), ("p3 only", 2, [(32, 40)]),
                    ("full pyramid B=8", 8, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)])):
    xs = [torch.randn(N, 256, h, w, device="cuda", requires_grad=True) for h, w in hws]
    w = (torch.randn(256, 256, 3, 3, device="cuda") * 0.02).requires_grad_(True)
    b = torch.zeros(256, device="cuda", requires_grad=True)
    gys = [torch.randn_like(x) for x in xs]
    for backend in ("winograd", "library"):
        ops.conv3x3_backend(winograd=(backend == "winograd"), min_tiles=0)
        outs = []
        for rep in range(3):
            for t in xs + [w, b]:
                t.grad = None
            ys = ops.conv3x3_levels(xs, w, b, relu=True)
            torch.autograd.backward(ys, gys)
            outs.append(([y.detach().clone() for y in ys], [x.grad.clone() for x in xs], w.grad.clone(), b.grad.clone()))
        y_eq = all(torch.equal(a, c) for a, c in zip(outs[1][0], outs[2][0]))
        dx_eq = all(torch.equal(a, c) for a, c in zip(outs[1][1], outs[2][1]))
        print("%-18s %-8s fwd bitwise %s | dx bitwise %s (rel %.1e) | dw rel run-to-run %.2e (bitwise %s) | db rel %.2e" % (
            tag, backend, y_eq, dx_eq, max(rel(a, c) for a, c in zip(outs[1][1], outs[2][1])), rel(outs[1][2], outs[2][2]),
            torch.equal(outs[1][2], outs[2][2]), rel(outs[1][3], outs[2][3])), flush=True)
    # the weight-gradient GEMM alone
    T = 4 * ((N * sum(((h + 3) // 4) * ((w_ + 3) // 4) for h, w_ in hws) + 3) // 4)
    dM = torch.randn(256, 36, T, device="cuda").permute(1, 0, 2)
    V = torch.randn(256, 36, T, device="cuda").permute(1, 0, 2)
    r = [torch.bmm(dM, V.transpose(1, 2)) for _ in range(3)]
    ref = torch.bmm(dM.double(), V.double().transpose(1, 2))
    print("%-18s bmm dW (T=%d): run-to-run bitwise %s rel %.2e | vs fp64 %.2e" % (tag, T, torch.equal(r[1], r[2]), rel(r[1], r[2]), rel(r[1], ref)), flush=True)
