import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lgd_amd import ops, hip
lib = hip.load()
x = torch.randn(8, 3, 800, 1344, device='cuda'); w = torch.randn(64, 3, 7, 7, device='cuda') * 0.05; sh = torch.randn(64, device='cuda')
ops.stem_conv_pool(x, w, sh)
img, winv, _ = w._lgd_stem7
xa = x.abs().max().reshape(1).view(torch.int32)
out = torch.empty(8, 64, 200, 336, device='cuda')
def k(): hip.check(lib.lgd_stem7_conv_pool(hip.ptr(x), hip.ptr(img), hip.ptr(winv), hip.ptr(xa), hip.ptr(sh), 8, 800, 1344, hip.ptr(out), None, hip.stream_ptr()), "k")
for _ in range(3): k()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(20): k()
e1.record(); torch.cuda.synchronize(); print("%s stem7_kernel alone %.1f us" % (os.environ.get("LGD_HIPCC_DEFS", "shipped"), e0.elapsed_time(e1) * 1e3 / 20))
