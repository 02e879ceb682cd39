"""LAB driver for tools/lab/canary_lab.hip: the canary kernel on the main stream while csrc/h2.hip's forward product (or another kernel) runs on a side
stream; reports register and LDS mismatches.   python tools/canary_probe.py [--lib lab.so]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from lgd_amd import hip, ops  # noqa: E402

if "--lib" in sys.argv:
    i = sys.argv.index("--lib")
    hip._LIB_PATH = os.path.abspath(sys.argv[i + 1])
    del sys.argv[i:i + 2]
import common as cm  # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
can = ctypes.CDLL(os.path.join(root, "tools", "lab", "libcanary_lab.so"))
can.canary_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib = hip.load()
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
nb, M, K, T = 64, 256, 256, 3808
U = torch.randn((nb, M, K), device=dev, generator=g) * 0.05
v = torch.randn((K, nb, T), device=dev, generator=g)
sa, sv = cm.h2_pow2_scale(U.abs().amax((1, 2))), cm.h2_pow2_scale(v.abs().amax((0, 2)))
img, vs, ia, iv = cm.h2_split_image(U, sa), cm.h2_split_rows(v, sv), (1 / sa).contiguous(), (1 / sv).contiguous()
C = torch.empty((M, nb, T), device=dev)
ops.gemm3_backend(True, force=True)
Vf = torch.randn((nb, K, T), device=dev, generator=g)


def h2():
    hip.check(lib.lgd_h2_fwd(hip.ptr(img), hip.ptr(vs), 4 * T, 4 * nb * T, 4 * vs.numel(), hip.ptr(C), T, nb * T, hip.ptr(ia), hip.ptr(iv), 1, None, nb, M, T, K,
                             hip.stream_ptr()), "lgd_h2_fwd")


aggr = {"nothing": lambda: None, "h2_fwd": h2, "gemm3 (bf16x3)": lambda: ops.gemm3_bmm(U, Vf, out=C.permute(1, 0, 2))}
side = torch.cuda.Stream()
for name, fn in aggr.items():
    out = torch.zeros(16, dtype=torch.int32, device=dev)
    for _ in range(20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        can.canary_launch(out.data_ptr(), 2048, 40, 4, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    o = out.cpu().tolist()
    print("canary beside %-16s: register mismatches %d, LDS mismatches %d | first register: index %d thread %d block %d got %#x want %#x | first LDS: word %d block %d got %#x want %#x"
          % (name, o[0], o[1], o[4], o[5], o[6], o[7] & 0xffffffff, o[8] & 0xffffffff, o[9], o[10], o[11] & 0xffffffff, o[12] & 0xffffffff), flush=True)
