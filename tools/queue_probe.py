"""Two streams of HEAVY kernels under different numbers of hardware queues: csrc/h2.hip's forward product (64 x [256 x 256] . [256 x T]: fills the chip
for ~150 us) issued (a) 2 R times on one stream, (b) R times on each of two streams with no dependency between them, (c) as in (b) but every launch
of stream B waits for an event behind the matching launch of stream A and A's next launch waits for B's (the per-layer fork / join pattern), and
(d) = (b) with a streaming elementwise kernel (a copy) on stream B instead of the product.  Wall time per product launch.
Why: at GPU_MAX_HW_QUEUES >= 5 the training step with the head fork on takes 71 ms instead of 52 (profiles/r06_hw_queues_and_forks.txt).
    GPU_MAX_HW_QUEUES=8 python tools/queue_probe.py [T]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from lgd_amd import hip  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import common as cm  # noqa: E402

lib = hip.load()
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5248
g = torch.Generator(device=dev).manual_seed(0)
nb, M, K = 64, 256, 256
U = torch.randn((nb, M, K), device=dev, generator=g) * 0.05
sa = cm.h2_pow2_scale(U.abs().amax((1, 2)))
img, ia = cm.h2_split_image(U, sa), (1 / sa).contiguous()
sets = []
for _ in range(2):
    v = torch.randn((K, nb, T), device=dev, generator=g)
    sv = cm.h2_pow2_scale(v.abs().amax((0, 2)))
    sets.append((cm.h2_split_rows(v, sv), (1 / sv).contiguous(), torch.empty((M, nb, T), device=dev)))
    del v
big = torch.empty(64 << 20, dtype=torch.float32, device=dev).normal_()
big2 = torch.empty_like(big)


def product(i):
    vs, iv, C = sets[i]
    hip.check(lib.lgd_h2_fwd(hip.ptr(img), hip.ptr(vs), 4 * T, 4 * nb * T, 4 * vs.numel(), hip.ptr(C), T, nb * T, hip.ptr(ia), hip.ptr(iv), 1, None,
                             nb, M, T, K, hip.stream_ptr()), "lgd_h2_fwd")


A = torch.cuda.current_stream(dev)
extra = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("PROBE_SKIP_STREAMS", "0")))]   # (moves B to a later slot of torch's stream pool)
B = torch.cuda.Stream(dev)
R = 40


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    fn()
    torch.cuda.synchronize()
    return 1e6 * (time.time() - t0)


def one_stream():
    for i in range(2 * R):
        product(i & 1)


def two_independent():
    B.wait_stream(A)
    for i in range(R):
        product(0)
        with torch.cuda.stream(B):
            product(1)
    A.wait_stream(B)


def two_ping_pong():
    for i in range(R):
        product(0)
        B.wait_stream(A)
        with torch.cuda.stream(B):
            product(1)
        A.wait_stream(B)


def product_beside_copy():
    B.wait_stream(A)
    for i in range(R):
        product(0)
        with torch.cuda.stream(B):
            big2.copy_(big, non_blocking=True)
    A.wait_stream(B)


def copies_only():
    for i in range(R):
        big2.copy_(big, non_blocking=True)


print("GPU_MAX_HW_QUEUES=%s  T=%d  (%d tiles = %.2f rounds of 512)" % (os.environ.get("GPU_MAX_HW_QUEUES", "(default)"), T, nb * ((T + 127) // 128), nb * ((T + 127) // 128) / 512))
for rep in range(2):
    t1 = timed(one_stream) / (2 * R)
    t2 = timed(two_independent) / (2 * R)
    t3 = timed(two_ping_pong) / (2 * R)
    tc = timed(copies_only) / R
    t4 = timed(product_beside_copy) / R
    print("per product launch: one stream %.1f us | two streams, independent %.1f us | two streams, fork/join per launch %.1f us || a 256 MB copy alone %.1f us, product + copy on two streams %.1f us per pair (sum alone %.1f)"
          % (t1, t2, t3, tc, t4, t1 + tc), flush=True)
