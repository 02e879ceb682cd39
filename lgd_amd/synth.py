"""Deterministic synthetic inputs and closed-form weights for the LGD hot path.

Everything here is integer-hash based (splitmix64 -> 24-bit mantissa), so the
same arrays are regenerated bit-for-bit by the golden generator (build
container), by the CPU tests and on the GPU box -- no torch RNG, no libm.
Workload shapes follow BASELINE.md section 3 (C1..C3).
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def det_uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array, uniform on [lo, hi) from a counter-based hash (exactly reproducible)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32))
        h = _splitmix64(idx)
    u = (h >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1), 24 bits: exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def name_seed(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def pyramid_shapes(img_h, img_w, strides=(8, 16, 32, 64, 128)):
    """FPN level (H, W) for a padded image: p3..p5 by ceil-halving from stride 8, p6/p7 stride-2 convs."""
    out = []
    h, w = img_h, img_w
    for i, s in enumerate(strides):
        if i == 0:
            h, w = -(-img_h // s), -(-img_w // s)
        else:
            h, w = (h + 1) // 2, (w + 1) // 2  # conv3x3/s2/p1 and the ResNet strided stages
        out.append((h, w))
    return out


def synth_features(B, img_h, img_w, seed=11, C=256, scale=1.0):
    """dict p3..p7 -> float32 (B, C, Hi, Wi)."""
    feats = {}
    for i, (h, w) in enumerate(pyramid_shapes(img_h, img_w)):
        feats["p%d" % (i + 3)] = det_uniform((B, C, h, w), seed * 16 + i, -scale, scale)
    return feats


# Fixed box table for config C1 (512x512, 10 boxes/img).  Includes the mask
# edge cases of SURVEY.md section 8c(iii): stride-aligned boundaries, a sub-pixel box, a
# zero-extent box, an out-of-bounds box (clamped), a near-full-image box.
_C1_BOXES = [
    [8.0, 8.0, 24.0, 24.0],        # stride-8 aligned: columns {1,2,3} at p3
    [16.0, 32.0, 232.0, 200.0],
    [100.5, 40.25, 180.75, 300.0],
    [300.0, 300.0, 303.0, 303.0],  # 3 px box: empty at every level
    [64.0, 64.0, 64.0, 200.0],     # zero width: empty everywhere (x/0)
    [-20.0, 400.0, 90.0, 600.0],   # out of bounds: clamped to the image
    [0.0, 0.0, 511.0, 511.0],      # full image
    [128.0, 128.0, 384.0, 384.0],  # exact multiples of 128 (boundary at p7)
    [33.3, 250.1, 250.7, 260.9],   # thin horizontal
    [470.0, 10.0, 500.0, 480.0],   # thin vertical
]


def synth_gt(B, img_h, img_w, n_per_img=10, seed=5, table=False):
    """list of (boxes float32 (Ni,4) xyxy abs px, classes int64 (Ni,)).

    table=True: the fixed C1 edge-case table (jittered per image by +3*b px).
    Otherwise sides log-uniform 16..600 px clipped to the image, classes
    uniform 0..79, from the deterministic hash.
    """
    out = []
    for b in range(B):
        if table:
            bx = np.array(_C1_BOXES[:n_per_img], dtype=np.float32)
            bx = bx + np.float32(3.0 * b) * np.array([1, 1, 1, 1], np.float32) * (np.arange(len(bx))[:, None] % 2 == 1)
            cls = np.array([(7 * j + 3 * b) % 80 for j in range(len(bx))], dtype=np.int64)
        else:
            u = det_uniform((n_per_img, 5), seed * 131 + b, 0.0, 1.0).astype(np.float64)
            bw = np.minimum(16.0 * (600.0 / 16.0) ** u[:, 0], img_w - 2.0)
            bh = np.minimum(16.0 * (600.0 / 16.0) ** u[:, 1], img_h - 2.0)
            x1 = u[:, 2] * (img_w - 1.0 - bw)
            y1 = u[:, 3] * (img_h - 1.0 - bh)
            bx = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
            cls = np.minimum((u[:, 4] * 80).astype(np.int64), 79)
        out.append((bx, cls))
    return out


def synth_images(B, h, w, seed=3):
    """uint8 images (B,3,h,w), BGR order, as detectron2's DatasetMapper hands them to the model."""
    return np.floor(det_uniform((B, 3, h, w), seed, 0.0, 256.0)).astype(np.uint8)


def closed_form_params(shapes, gain=1.0):
    """name -> float32 array.  weights ~ U(-a,a), a = gain*sqrt(3/fan_in); biases ~ U(-0.1,0.1)."""
    out = {}
    for name, shp in shapes.items():
        shp = tuple(shp)
        if name.endswith("bias") or len(shp) == 1:
            out[name] = det_uniform(shp, name_seed(name), -0.1, 0.1)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = gain * (3.0 / fan_in) ** 0.5
            out[name] = det_uniform(shp, name_seed(name), -a, a)
    return out


def fcos_head_params(shapes, gain=2.0 ** 0.5):
    """closed-form parameters for an FCOS head state_dict (thirdparty_heads/fcos.py:433-512 names): conv weights / biases as
    closed_form_params (gain sqrt(2): the towers keep their signal through four conv + GroupNorm + ReLU layers), GroupNorm weights and
    the per-level `scales.i.scale` ~ U(0.5, 1.5) (a closed-form U(-0.1, 0.1) 'bias-like' gamma would switch the norm layers off)."""
    out = closed_form_params(shapes, gain)
    for name, shp in shapes.items():
        is_gn_weight = name.endswith(".weight") and len(tuple(shp)) == 1
        if is_gn_weight or name.endswith(".scale"):
            out[name] = det_uniform(tuple(shp), name_seed(name), 0.5, 1.5)
    return out
