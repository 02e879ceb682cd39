"""torch.autograd wrappers over the HIP kernels (forward AND backward are HIP; torch only owns
device memory and the stream).  Shapes follow include/lgd_hip.h:
  pyramids  : list of L tensors (B, C, H_l, W_l) fp32 NCHW
  box tables: (L, T, C) fp32, boxes concatenated image-major, context box last per image
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

from . import hip, streams


class BoxGeometry:
    """Integer rectangles + row bands of every (level, box), built once per step on the GPU.

    Replaces the reference's dense masks `batchified_inside_masks` (F x B x (Ni, HiWi) floats)
    [ref: dynamic_teacher.py:241-242, utils.py:53-89].
    boxes: (T,4) fp32 device tensor of CLAMPED xyxy boxes in padded-image pixels
           [ref: label_encoder.py:83-85]; counts: python list of boxes per image (host-known,
           no device sync); level_hw: list of (H, W).
    """

    def __init__(self, boxes, counts, img_hw, level_hw):
        hip.require_gpu(boxes)
        lib = hip.load()
        self.counts = [int(c) for c in counts]
        self.B = len(self.counts)
        self.T = int(sum(self.counts))
        self.max_n = max(self.counts) if self.counts else 0
        self.level_hw = [(int(h), int(w)) for h, w in level_hw]
        self.L = len(self.level_hw)
        self.img_hw = (int(img_hw[0]), int(img_hw[1]))
        if boxes.shape != (self.T, 4):
            raise hip.LgdHipError("boxes must be (T,4) with T=sum(counts)=%d, got %s" % (self.T, tuple(boxes.shape)))
        self.boxes = hip.dense_f32(boxes.detach())
        off = [0]
        for c in self.counts:
            off.append(off[-1] + c)
        self.img_off = hip.to_device(off, torch.int32, boxes.device)
        self._hw = hip.int_array([v for hw in self.level_hw for v in hw])
        n_ints = lib.lgd_geom_ints(self.L, self.B, self.T, self.max_n)
        self.geom = torch.empty(n_ints, dtype=torch.int32, device=boxes.device)
        hip.check(lib.lgd_box_prep(hip.ptr(self.boxes), hip.ptr(self.img_off), self.B, self.T, self.max_n,
                                   self.img_hw[0], self.img_hw[1], self._hw, self.L, hip.ptr(self.geom),
                                   hip.stream_ptr()), "lgd_box_prep")

    # --- debugging / tests -------------------------------------------------------------------
    def rects(self):
        """(L, T, 4) int32 [x0, x1, y0, y1] inclusive (empty: x1 < x0)."""
        lib = hip.load()
        o = lib.lgd_geom_rects_off(self.L, self.B, self.T, self.max_n)
        padded = self.geom[o:o + self.L * self.B * self.max_n * 4].view(self.L, self.B * self.max_n, 4)
        rows = [b * self.max_n + j for b, n in enumerate(self.counts) for j in range(n)]
        return padded[:, hip.to_device(rows, torch.int64, padded.device)]

    def bands(self):
        """list[L][B] of python lists of row breakpoints."""
        lib = hip.load()
        on = lib.lgd_geom_nbp_off(self.L, self.B, self.T, self.max_n)
        ob = lib.lgd_geom_bands_off(self.L, self.B, self.T, self.max_n)
        mb = 2 * self.max_n + 2
        g = self.geom.cpu()
        nbp = g[on:on + self.L * self.B].view(self.L, self.B)
        bands = g[ob:ob + self.L * self.B * mb].view(self.L, self.B, mb)
        return [[bands[l, b, :int(nbp[l, b])].tolist() for b in range(self.B)] for l in range(self.L)]

    def _check(self, maps):
        if len(maps) != self.L:
            raise hip.LgdHipError("expected %d pyramid levels, got %d" % (self.L, len(maps)))
        for m, hw in zip(maps, self.level_hw):
            if m.dim() != 4 or m.shape[0] != self.B or tuple(m.shape[-2:]) != hw:
                raise hip.LgdHipError("level shape %s does not match geometry (B=%d, HW=%s)" % (tuple(m.shape), self.B, hw))


def _box_sum(geom, maps, normalize, skip_last):
    lib = hip.load()
    maps = [hip.dense_f32(m) for m in maps]
    geom._check(maps)
    C = maps[0].shape[1]
    out = torch.empty((geom.L, geom.T, C), dtype=torch.float32, device=maps[0].device)
    ws = torch.empty(lib.lgd_box_pool_ws_floats(geom._hw, geom.L, geom.B, C, geom.max_n, 1), dtype=torch.float32, device=maps[0].device)
    hip.check(lib.lgd_box_sum(hip.ptr_array(maps), geom._hw, geom.L, geom.B, C, geom.T, geom.max_n,
                              hip.ptr(geom.img_off), hip.ptr(geom.geom), hip.ptr(ws), hip.ptr(out), int(normalize), int(skip_last),
                              hip.stream_ptr()), "lgd_box_sum")
    return out


def _box_paint(geom, vals, normalize, skip_last):
    lib = hip.load()
    vals = hip.dense_f32(vals)
    C = vals.shape[2]
    if tuple(vals.shape) != (geom.L, geom.T, C):
        raise hip.LgdHipError("vals must be (L,T,C)=(%d,%d,C), got %s" % (geom.L, geom.T, tuple(vals.shape)))
    outs = [torch.empty((geom.B, C, h, w), dtype=torch.float32, device=vals.device) for h, w in geom.level_hw]
    hip.check(lib.lgd_box_paint(hip.ptr(vals), geom._hw, geom.L, geom.B, C, geom.T, geom.max_n, hip.ptr(geom.img_off),
                                hip.ptr(geom.geom), hip.ptr_array(outs), int(normalize), int(skip_last),
                                hip.stream_ptr()), "lgd_box_paint")
    return outs


class _MaskPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, *maps):
        hip.require_gpu(*maps)
        ctx.geom = geom
        return _box_sum(geom, maps, True, False)

    @staticmethod
    def backward(ctx, g):
        grads = _box_paint(ctx.geom, g, True, False)
        return (None, *grads)


class _RenderPaint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, skip_last, vals):
        hip.require_gpu(vals)
        ctx.geom, ctx.skip_last = geom, skip_last
        return tuple(_box_paint(geom, vals, False, skip_last))

    @staticmethod
    def backward(ctx, *gmaps):
        return None, None, _box_sum(ctx.geom, gmaps, False, ctx.skip_last)


def mask_pool(geom, maps):
    """Appearance embeddings: per-box mean of each level's map -> (L, T, C).
    [ref: dynamic_teacher.py:81-103, 249-253]"""
    return _MaskPool.apply(geom, *maps)


def render_paint(geom, vals, skip_last):
    """Per-pixel sum of the covering boxes' rows of vals (L,T,C) -> list of L maps (B,C,H_l,W_l).
    [ref: dynamic_teacher.py:137-143 / 173-179]; skip_last: the context box is not painted."""
    return list(_RenderPaint.apply(geom, bool(skip_last), vals))


class _DistillInMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coef, n_levels, *maps):
        lib = hip.load()
        hip.require_gpu(*maps)
        a = [hip.dense_f32(m) for m in maps[:n_levels]]
        b = [hip.dense_f32(m.detach()) for m in maps[n_levels:]]
        B, C = a[0].shape[0], a[0].shape[1]
        for x, y in zip(a, b):
            if x.shape != y.shape or x.shape[0] != B or x.shape[1] != C:
                raise hip.LgdHipError("student/teacher level shapes differ: %s vs %s" % (tuple(x.shape), tuple(y.shape)))
        hw = hip.int_array([v for m in a for v in m.shape[-2:]])
        dev = a[0].device
        ws = torch.empty(lib.lgd_distill_ws_doubles(hw, n_levels, B, C), dtype=torch.float64, device=dev)
        stats = torch.empty((n_levels * B * C, 8), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_distill_fwd(hip.ptr_array(a), hip.ptr_array(b), hw, n_levels, B, C, float(coef), hip.ptr(ws),
                                      hip.ptr(stats), hip.ptr(loss), hip.stream_ptr()), "lgd_distill_fwd")
        ctx.save_for_backward(stats, *a, *b)
        ctx.meta = (float(coef), n_levels, B, C, hw)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = hip.load()
        coef, L, B, C, hw = ctx.meta
        stats = ctx.saved_tensors[0]
        a = ctx.saved_tensors[1:1 + L]
        b = ctx.saved_tensors[1 + L:]
        g = g.contiguous().to(torch.float32)
        ga = [torch.empty_like(x) for x in a]
        hip.check(lib.lgd_distill_bwd(hip.ptr_array(a), hip.ptr_array(b), hw, L, B, C, coef, hip.ptr(stats), hip.ptr(g),
                                      hip.ptr_array(ga), hip.stream_ptr()), "lgd_distill_bwd")
        return (None, None, *ga, *([None] * L))


def distill_in_mse(stu_maps, tea_maps, coef):
    """coef * mse(InstanceNorm(tea), InstanceNorm(stu)) over all levels; the teacher side is
    detached.  [ref: models/base_distillator.py:55-64]"""
    stu_maps, tea_maps = list(stu_maps), list(tea_maps)
    return _DistillInMse.apply(float(coef), len(stu_maps), *stu_maps, *tea_maps)


# ------------------------------------------------------------------------------------------------ K5 / K2 / K6 wrappers
def _offsets(counts):
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    return off


def segment_ids(counts, device):
    """(T,) int64 image index of every box row; built on the host (counts are host-known)."""
    ids = [b for b, n in enumerate(counts) for _ in range(n)]
    return hip.to_device(ids, torch.int64, device)


def _levels_meta(maps):
    B, C = maps[0].shape[0], maps[0].shape[1]
    for m in maps:
        if m.dim() != 4 or m.shape[0] != B or m.shape[1] != C:
            raise hip.LgdHipError("pyramid levels must share (B, C); got %s" % [tuple(x.shape) for x in maps])
    return B, C, hip.int_array([v for m in maps for v in m.shape[-2:]])


class _Gn1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, relu, *xs):
        lib = hip.load()
        hip.require_gpu(*xs)
        xs = [hip.dense_f32(x) for x in xs]
        L = len(xs)
        B, C, hw = _levels_meta(xs)
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn1_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        stats = torch.empty((L * B, 2), dtype=torch.float32, device=dev)
        ys = [torch.empty_like(x) for x in xs]
        hip.check(lib.lgd_gn1_fwd(hip.ptr_array(xs), hw, L, B, C, int(relu), hip.ptr(ws), hip.ptr(stats),
                                  hip.ptr_array(ys), hip.stream_ptr()), "lgd_gn1_fwd")
        ctx.save_for_backward(stats, *xs)
        ctx.meta = (bool(relu), L, B, C, hw)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = hip.load()
        relu, L, B, C, hw = ctx.meta
        stats, xs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        dys = [hip.dense_f32(d) for d in dys]
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn1_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        bstats = torch.empty((L * B, 2), dtype=torch.float32, device=dev)
        dxs = [torch.empty_like(x) for x in xs]
        hip.check(lib.lgd_gn1_bwd(hip.ptr_array(xs), hip.ptr_array(dys), hw, L, B, C, int(relu), hip.ptr(stats), hip.ptr(ws),
                                  hip.ptr(bstats), hip.ptr_array(dxs), hip.stream_ptr()), "lgd_gn1_bwd")
        return (None, *dxs)


def gn1(xs, relu):
    """GroupNorm(num_groups=1, affine=False, eps=1e-5) [+ ReLU] on every pyramid level in one call.
    xs: list of (B,C,H_l,W_l) -> list.  [ref: layers.py:6-7, 22-32]"""
    return list(_Gn1.apply(bool(relu), *xs))


class _GnReluPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, geom, *xs):
        lib = hip.load()
        hip.require_gpu(*xs)
        xs = [hip.dense_f32(x) for x in xs]
        geom._check(xs)
        L = len(xs)
        B, C, hw = _levels_meta(xs)
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn1_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        stats = torch.empty((L * B, 2), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_gn1_stats(hip.ptr_array(xs), hw, L, B, C, hip.ptr(ws), hip.ptr(stats), hip.stream_ptr()), "lgd_gn1_stats")
        out = torch.empty((L, geom.T, C), dtype=torch.float32, device=dev)
        # per (box, channel): sum of relu(xhat) and number of active pixels -- the backward's GroupNorm means without a pass over x
        raw = torch.empty((2, L, geom.T, C), dtype=torch.float32, device=dev)
        pws = torch.empty(lib.lgd_box_pool_ws_floats(hw, L, B, C, geom.max_n, 2), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_gn_pool_fwd(hip.ptr_array(xs), hip.ptr(stats), hw, L, B, C, geom.T, geom.max_n, hip.ptr(geom.img_off),
                                      hip.ptr(geom.geom), hip.ptr(pws), hip.ptr(out), hip.ptr(raw), hip.stream_ptr()), "lgd_gn_pool_fwd")
        ctx.save_for_backward(stats, raw, *xs)
        ctx.geom, ctx.meta = geom, (L, B, C, hw)
        return out

    @staticmethod
    def backward(ctx, dpool):
        lib = hip.load()
        L, B, C, hw = ctx.meta
        geom = ctx.geom
        stats, raw, xs = ctx.saved_tensors[0], ctx.saved_tensors[1], ctx.saved_tensors[2:]
        dpool = hip.dense_f32(dpool)
        dev = xs[0].device
        bstats = torch.empty((L * B, 2), dtype=torch.float32, device=dev)
        dxs = [torch.empty_like(x) for x in xs]
        hip.check(lib.lgd_gn_pool_bwd(hip.ptr_array(xs), hip.ptr(stats), hip.ptr(dpool), hip.ptr(raw), hw, L, B, C, geom.T, geom.max_n,
                                      hip.ptr(geom.img_off), hip.ptr(geom.geom), hip.ptr(bstats), hip.ptr_array(dxs),
                                      hip.stream_ptr()), "lgd_gn_pool_bwd")
        return (None, *dxs)


def gn_relu_mask_pool(geom, xs):
    """mask_pool(geom, gn1(xs, relu=True)) without ever writing the normalised maps: appearance embeddings (L,T,C)
    straight from the conv outputs.  [ref: dynamic_teacher.py:57,235 student_proj_2D + 249-253 aggregate_per_level]"""
    return _GnReluPool.apply(geom, *xs)


class _GroupNormRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, groups, relu, weight, bias, *xs):
        lib = hip.load()
        hip.require_gpu(*xs)
        xs = [hip.dense_f32(x) for x in xs]
        L = len(xs)
        B, C, hw = _levels_meta(xs)
        weight = hip.dense_f32(weight) if weight is not None else None
        bias = hip.dense_f32(bias) if bias is not None else None
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        stats = torch.empty((L * B * groups, 2), dtype=torch.float32, device=dev)
        ys = [torch.empty_like(x) for x in xs]
        nb = 4 * sum(x.numel() for x in xs)
        _count_bytes("gn_group_stats_kernel", nb)
        _count_bytes("gn_group_apply_kernel", 2 * nb)
        hip.check(lib.lgd_gn_group_fwd(hip.ptr_array(xs), hw, L, B, C, groups, hip.ptr(weight) if weight is not None else None,
                                       hip.ptr(bias) if bias is not None else None, int(relu), hip.ptr(ws), hip.ptr(stats),
                                       hip.ptr_array(ys), hip.stream_ptr()), "lgd_gn_group_fwd")
        ctx.save_for_backward(stats, weight, bias, *xs)
        ctx.meta = (groups, bool(relu), L, B, C, hw)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = hip.load()
        groups, relu, L, B, C, hw = ctx.meta
        stats, weight, bias, *xs = ctx.saved_tensors
        dys = [hip.dense_f32(d) for d in dys]
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        bstats = torch.empty((L * B * groups, 2), dtype=torch.float32, device=dev)
        psums = torch.empty((L * B, C, 2), dtype=torch.float32, device=dev)
        dxs = [torch.empty_like(x) for x in xs]
        nb = 4 * sum(x.numel() for x in xs)
        _count_bytes("gn_group_bwd_stats_kernel", 2 * nb)
        _count_bytes("gn_group_bwd_apply_kernel", 3 * nb)
        hip.check(lib.lgd_gn_group_bwd(hip.ptr_array(xs), hip.ptr_array(dys), hw, L, B, C, groups,
                                       hip.ptr(weight) if weight is not None else None, hip.ptr(bias) if bias is not None else None,
                                       int(relu), hip.ptr(stats), hip.ptr(ws), hip.ptr(bstats), hip.ptr(psums), hip.ptr_array(dxs),
                                       hip.stream_ptr()), "lgd_gn_group_bwd")
        s = psums.sum(0) if (weight is not None or bias is not None) else None
        dw = s[:, 1].contiguous() if weight is not None and ctx.needs_input_grad[2] else None
        db = s[:, 0].contiguous() if bias is not None and ctx.needs_input_grad[3] else None
        return (None, None, dw, db, *dxs)


def group_norm_relu(xs, groups, weight=None, bias=None, relu=True):
    """nn.GroupNorm(groups, C)(x) [+ ReLU] with ONE module applied to a list of maps (B,C,H_l,W_l) in one call
    [ref: thirdparty_heads/fcos.py:455-470 tower layers]."""
    return list(_GroupNormRelu.apply(int(groups), bool(relu), weight, bias, *xs))


class _GroupNormFold(torch.autograd.Function):
    """nn.GroupNorm(groups, C) + ReLU in front of a 3x3 convolution WITHOUT its apply pass: the statistics are folded with gamma / beta
    into a per-(map, sample, channel) scale and shift (lgd_gn_group_stats_affine) that the convolution's input transform applies while it
    loads (conv3x3_levels(..., pre=affine)); the normalised maps are never written or re-read (2 map transfers of 3 per tower layer).
    forward -> (affine (L*B, C, 2), the maps themselves); backward: the convolution returns the gradient w.r.t. the GroupNorm OUTPUT
    (its adjoint input transform applies the activation bits the forward transform wrote), lgd_gn_group_bwd(relu = 0) turns it into the
    gradient of the raw maps, d gamma and d beta."""

    @staticmethod
    def forward(ctx, groups, weight, bias, *xs):
        lib = hip.load()
        hip.require_gpu(*xs)
        xs = [hip.dense_f32(x) for x in xs]
        L = len(xs)
        B, C, hw = _levels_meta(xs)
        weight = hip.dense_f32(weight) if weight is not None else None
        bias = hip.dense_f32(bias) if bias is not None else None
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        stats = torch.empty((L * B * groups, 2), dtype=torch.float32, device=dev)
        affine = torch.empty((L * B, C, 2), dtype=torch.float32, device=dev)
        _count_bytes("gn_group_stats_kernel", 4 * sum(x.numel() for x in xs))
        hip.check(lib.lgd_gn_group_stats_affine(hip.ptr_array(xs), hw, L, B, C, groups, hip.ptr(weight) if weight is not None else None,
                                                hip.ptr(bias) if bias is not None else None, hip.ptr(ws), hip.ptr(stats), hip.ptr(affine),
                                                hip.stream_ptr()), "lgd_gn_group_stats_affine")
        ctx.save_for_backward(stats, weight, bias, *xs)
        ctx.meta = (groups, L, B, C, hw)
        ctx.mark_non_differentiable(affine)
        return (affine, *[x.view_as(x) for x in xs])

    @staticmethod
    def backward(ctx, _gaff, *dys):
        lib = hip.load()
        groups, L, B, C, hw = ctx.meta
        stats, weight, bias, *xs = ctx.saved_tensors
        dys = [hip.dense_f32(d) for d in dys]
        dev = xs[0].device
        ws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        bstats = torch.empty((L * B * groups, 2), dtype=torch.float32, device=dev)
        psums = torch.empty((L * B, C, 2), dtype=torch.float32, device=dev)
        dxs = [torch.empty_like(x) for x in xs]
        nb = 4 * sum(x.numel() for x in xs)
        _count_bytes("gn_group_bwd_stats_kernel", 2 * nb)
        _count_bytes("gn_group_bwd_apply_kernel", 3 * nb)
        hip.check(lib.lgd_gn_group_bwd(hip.ptr_array(xs), hip.ptr_array(dys), hw, L, B, C, groups,
                                       hip.ptr(weight) if weight is not None else None, hip.ptr(bias) if bias is not None else None,
                                       0, hip.ptr(stats), hip.ptr(ws), hip.ptr(bstats), hip.ptr(psums), hip.ptr_array(dxs),
                                       hip.stream_ptr()), "lgd_gn_group_bwd")
        s = psums.sum(0) if (weight is not None or bias is not None) else None
        dw = s[:, 1].contiguous() if weight is not None and ctx.needs_input_grad[1] else None
        db = s[:, 0].contiguous() if bias is not None and ctx.needs_input_grad[2] else None
        return (None, dw, db, *dxs)


def _tag_folded(affine, maps):
    """the raw maps of a folded activation are only meaningful together with their affine: their autograd gradient is the gradient w.r.t.
    the activation's OUTPUT, which only a convolution called with pre=affine returns.  The tag lets this library's consumers refuse a
    mismatch; a foreign consumer (a hook, an auxiliary loss on tower features) must apply the affine itself."""
    for m in maps:
        m._lgd_needs_pre = affine
    return affine, maps


def _check_pre(xs, pre):
    for x in xs:
        need = getattr(x, "_lgd_needs_pre", None)
        if need is not None and need is not pre:
            raise hip.LgdHipError("these maps are the RAW outputs of a folded activation (group_norm_fold / conv3x3_gn / ctx_shift_fold): pass the "
                                  "affine returned with them as pre=")


def group_norm_fold(xs, groups, weight=None, bias=None):
    """GroupNorm(groups, C) + ReLU of a list of maps, to be applied by the NEXT 3x3 convolution's input transform: returns (affine, maps);
    pass both on: conv3x3_levels(maps, w, b, pre=affine) / conv3x3_shared_input(maps, filters, pre=affine)
    [ref: thirdparty_heads/fcos.py:455-470 tower layers conv -> GroupNorm(32) -> ReLU -> conv]."""
    out = _GroupNormFold.apply(int(groups), weight, bias, *xs)
    return _tag_folded(out[0], list(out[1:]))


class _CtxRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cvec, *xs):
        lib = hip.load()
        hip.require_gpu(cvec, *xs)
        xs = [hip.dense_f32(x) for x in xs]
        L = len(xs)
        B, C, hw = _levels_meta(xs)
        cvec = hip.dense_f32(cvec)
        if tuple(cvec.shape) != (L, B, C):
            raise hip.LgdHipError("ctx must be (L,B,C)=(%d,%d,%d), got %s" % (L, B, C, tuple(cvec.shape)))
        ys = [torch.empty_like(x) for x in xs]
        hip.check(lib.lgd_ctx_relu_fwd(hip.ptr_array(xs), hip.ptr(cvec), hw, L, B, C, hip.ptr_array(ys), hip.stream_ptr()),
                  "lgd_ctx_relu_fwd")
        ctx.save_for_backward(*ys)
        ctx.meta = (L, B, C, hw)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        lib = hip.load()
        L, B, C, hw = ctx.meta
        ys = ctx.saved_tensors
        dys = [hip.dense_f32(d) for d in dys]
        dxs = [torch.empty_like(y) for y in ys]
        dctx = torch.empty((L, B, C), dtype=torch.float32, device=ys[0].device)
        hip.check(lib.lgd_ctx_relu_bwd(hip.ptr_array(ys), hip.ptr_array(dys), hw, L, B, C, hip.ptr_array(dxs), hip.ptr(dctx),
                                       hip.stream_ptr()), "lgd_ctx_relu_bwd")
        return (dctx, *dxs)


def bias_ctx_relu(xs, ctx):
    """ReLU(x_l + ctx[l][:, :, None, None]) for every level; ctx (L,B,C).  [ref: dynamic_teacher.py:151]"""
    return list(_CtxRelu.apply(ctx, *xs))


class _CtxShiftFold(torch.autograd.Function):
    """ReLU(x_l + ctx[l][:, :, None, None]) [ref: dynamic_teacher.py:151] handed to the NEXT 3x3 convolution as a per-(map, sample, channel)
    scale 1 / shift ctx that its input transform applies while it loads (the `pre` affine of conv3x3_levels / conv3x3_gn): the activated maps
    are neither written nor re-read.  That convolution returns, for its input maps, the gradient w.r.t. the affine OUTPUT with the ReLU
    mask applied -- which for scale 1 is the gradient of x itself, and whose per-plane sum is the gradient of ctx."""

    @staticmethod
    def forward(ctx, cvec, *xs):
        L = len(xs)
        B, C, hw = _levels_meta(xs)
        if tuple(cvec.shape) != (L, B, C):
            raise hip.LgdHipError("ctx must be (L,B,C)=(%d,%d,%d), got %s" % (L, B, C, tuple(cvec.shape)))
        aff = torch.stack((torch.ones_like(cvec), cvec), -1).view(L * B, C, 2)
        ctx.mark_non_differentiable(aff)
        ctx.meta = (L, B, C, hw)
        return (aff, *[x.view_as(x) for x in xs])

    @staticmethod
    def backward(ctx, _daff, *dxs):
        if not ctx.needs_input_grad[0]:
            return (None, *dxs)
        # d ctx[l, b, c] = the plane sum of the gradient: per-plane means in ONE pass over all levels (the GroupNorm statistics kernel with one
        # group per channel, fp64 partials) times the plane sizes
        L, B, C, hw = ctx.meta
        if not dxs[0].is_cuda:   # (the forward is pure torch and runs anywhere; so does this form of its adjoint)
            return (torch.stack([d.sum((2, 3)) for d in dxs]), *dxs)
        lib = hip.load()
        gs = [hip.dense_f32(d) for d in dxs]
        dev = gs[0].device
        ws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, B, C), dtype=torch.float64, device=dev)
        stats = torch.empty((L * B * C, 2), dtype=torch.float32, device=dev)
        aff = torch.empty((L * B, C, 2), dtype=torch.float32, device=dev)
        _count_bytes("gn_group_stats_kernel", 4 * sum(g.numel() for g in gs))
        hip.check(lib.lgd_gn_group_stats_affine(hip.ptr_array(gs), hw, L, B, C, C, None, None, hip.ptr(ws), hip.ptr(stats), hip.ptr(aff),
                                                hip.stream_ptr()), "lgd_gn_group_stats_affine")
        sizes = hip.to_device([float(g.shape[2] * g.shape[3]) for g in gs], torch.float32, dev)
        dctx = stats[:, 0].view(L, B, C) * sizes.view(L, 1, 1)
        return (dctx, *dxs)


def ctx_shift_fold(xs, cvec):
    """(affine, maps): ReLU(x_l + ctx[l]) to be applied by the next convolution -- conv3x3_levels(maps, w, b, pre=affine) /
    conv3x3_gn(maps, filters, groups, pre=affine).  cvec (L,B,C)."""
    out = _CtxShiftFold.apply(cvec, *xs)
    return _tag_folded(out[0], list(out[1:]))


def _gemm(A, sa, B, sb, C, sc, M, N, K, bias=None, alpha=1.0, rowsum=None):
    """one lgd_gemm_problem: C[m,n] = alpha*(sum_k A(m,k)B(n,k) + bias[n]); strides in elements; A/B/C are (tensor, offset)."""
    def addr(t):
        ten, off = t
        return ten.data_ptr() + 4 * off
    return hip.GemmProblem(addr(A), addr(B), addr(bias) if bias is not None else None, addr(C),
                           addr(rowsum) if rowsum is not None else None, M, N, K, 0,
                           sa[0], sa[1], sb[0], sb[1], sc[0], sc[1], float(alpha), 0)


def _gemm_batch(problems):
    arr = (hip.GemmProblem * len(problems))(*problems)
    hip.check(hip.load().lgd_gemm_batch(ctypes.cast(arr, ctypes.c_void_p), len(problems), hip.stream_ptr()), "lgd_gemm_batch")


class _MhaBlockDiag(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_in, kv_in, in_w, in_b, out_w, out_b, img_off, heads):
        lib = hip.load()
        hip.require_gpu(q_in, kv_in, in_w)
        q_in, kv_in = hip.dense_f32(q_in), hip.dense_f32(kv_in)
        in_w, in_b, out_w, out_b = (hip.dense_f32(t) for t in (in_w, in_b, out_w, out_b))
        Lq, T, E = q_in.shape
        Lk = kv_in.shape[0]
        L = max(Lq, Lk)
        B = img_off.numel() - 1
        scale = 1.0 / math.sqrt(E // heads)
        dev = q_in.device
        Q = torch.empty((Lq, T, E), dtype=torch.float32, device=dev)
        K = torch.empty((Lk, T, E), dtype=torch.float32, device=dev)
        V = torch.empty((Lk, T, E), dtype=torch.float32, device=dev)
        _gemm_batch([
            _gemm((q_in, 0), (E, 1), (in_w, 0), (E, 1), (Q, 0), (E, 1), Lq * T, E, E, bias=(in_b, 0), alpha=scale),
            _gemm((kv_in, 0), (E, 1), (in_w, E * E), (E, 1), (K, 0), (E, 1), Lk * T, E, E, bias=(in_b, E)),
            _gemm((kv_in, 0), (E, 1), (in_w, 2 * E * E), (E, 1), (V, 0), (E, 1), Lk * T, E, E, bias=(in_b, 2 * E)),
        ])
        O = torch.empty((L, T, E), dtype=torch.float32, device=dev)
        lse = torch.empty((L, T, heads), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_attn_fwd(hip.ptr(Q), hip.ptr(K), hip.ptr(V), hip.ptr(img_off), Lq, Lk, B, T, E, heads, hip.ptr(O),
                                   hip.ptr(lse), hip.stream_ptr()), "lgd_attn_fwd")
        out = torch.empty((L, T, E), dtype=torch.float32, device=dev)
        _gemm_batch([_gemm((O, 0), (E, 1), (out_w, 0), (E, 1), (out, 0), (E, 1), L * T, E, E, bias=(out_b, 0))])
        ctx.save_for_backward(q_in, kv_in, in_w, out_w, Q, K, V, O, lse, img_off)
        ctx.meta = (Lq, Lk, L, B, T, E, heads, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = hip.load()
        q_in, kv_in, in_w, out_w, Q, K, V, O, lse, img_off = ctx.saved_tensors
        Lq, Lk, L, B, T, E, heads, scale = ctx.meta
        dout = hip.dense_f32(dout)
        dev = dout.device
        dO = torch.empty((L, T, E), dtype=torch.float32, device=dev)
        d_out_w = torch.empty((E, E), dtype=torch.float32, device=dev)
        d_out_b = torch.empty((E,), dtype=torch.float32, device=dev)
        _gemm_batch([
            _gemm((dout, 0), (E, 1), (out_w, 0), (1, E), (dO, 0), (E, 1), L * T, E, E),                       # dO = dY Wo
            _gemm((dout, 0), (1, E), (O, 0), (1, E), (d_out_w, 0), (E, 1), E, E, L * T, rowsum=(d_out_b, 0)),  # dWo = dY^T O
        ])
        dQ = torch.empty((L, T, E), dtype=torch.float32, device=dev)   # per-level partials
        dK = torch.empty((L, T, E), dtype=torch.float32, device=dev)
        dV = torch.empty((L, T, E), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_attn_bwd(hip.ptr(Q), hip.ptr(K), hip.ptr(V), hip.ptr(O), hip.ptr(lse), hip.ptr(dO), hip.ptr(img_off),
                                   Lq, Lk, B, T, E, heads, hip.ptr(dQ), hip.ptr(dK), hip.ptr(dV), hip.stream_ptr()),
                  "lgd_attn_bwd")
        if Lq == 1 and L > 1:
            dQ = dQ.sum(0, keepdim=True)
        if Lk == 1 and L > 1:
            dK, dV = dK.sum(0, keepdim=True), dV.sum(0, keepdim=True)
        d_q_in = torch.empty((Lq, T, E), dtype=torch.float32, device=dev)
        d_k_in = torch.empty((Lk, T, E), dtype=torch.float32, device=dev)
        d_v_in = torch.empty((Lk, T, E), dtype=torch.float32, device=dev)
        d_in_w = torch.empty((3 * E, E), dtype=torch.float32, device=dev)
        d_in_b = torch.empty((3 * E,), dtype=torch.float32, device=dev)
        Mq, Mk = Lq * T, Lk * T
        _gemm_batch([
            _gemm((dQ, 0), (E, 1), (in_w, 0), (1, E), (d_q_in, 0), (E, 1), Mq, E, E, alpha=scale),
            _gemm((dK, 0), (E, 1), (in_w, E * E), (1, E), (d_k_in, 0), (E, 1), Mk, E, E),
            _gemm((dV, 0), (E, 1), (in_w, 2 * E * E), (1, E), (d_v_in, 0), (E, 1), Mk, E, E),
            _gemm((dQ, 0), (1, E), (q_in, 0), (1, E), (d_in_w, 0), (E, 1), E, E, Mq, alpha=scale, rowsum=(d_in_b, 0)),
            _gemm((dK, 0), (1, E), (kv_in, 0), (1, E), (d_in_w, E * E), (E, 1), E, E, Mk, rowsum=(d_in_b, E)),
            _gemm((dV, 0), (1, E), (kv_in, 0), (1, E), (d_in_w, 2 * E * E), (E, 1), E, E, Mk, rowsum=(d_in_b, 2 * E)),
        ])
        return d_q_in, d_k_in + d_v_in, d_in_w, d_in_b, d_out_w, d_out_b, None, None


def mha_blockdiag(q_in, kv_in, counts, in_w, in_b, out_w, out_b, heads, img_off=None):
    """nn.MultiheadAttention(E, heads) forward as called at [ref: dynamic_teacher.py:255-273]: sequence-first,
    batch 1, boolean mask blocking attention between boxes of different images -- computed per image block.
    q_in (Lq,T,E), kv_in (Lk,T,E) with Lq == Lk or one of them 1 (operand shared by all levels) -> (max(Lq,Lk),T,E).
    In/out projections: fp32 MFMA GEMM kernel; softmax(QK^T)V: one wave per (image, head)."""
    if img_off is None:
        img_off = hip.to_device(_offsets(counts), torch.int32, q_in.device)
    return _MhaBlockDiag.apply(q_in, kv_in, in_w, in_b, out_w, out_b, img_off, int(heads))


# ------------------------------------------------------------------------------------------------ K6: label-encoder ops
class _Linear(torch.autograd.Function):
    """y = x W^T + b on the fp32 MFMA GEMM kernel; backward = one launch with dX, dW (+ db as a row sum)."""

    @staticmethod
    def forward(ctx, x, w, b):
        hip.require_gpu(x, w)
        x, w = hip.dense_f32(x), hip.dense_f32(w)
        b = hip.dense_f32(b) if b is not None else None
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        _gemm_batch([_gemm((x, 0), (K, 1), (w, 0), (K, 1), (y, 0), (N, 1), M, N, K, bias=(b, 0) if b is not None else None)])
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = hip.dense_f32(dy)
        M, K = x.shape
        N = w.shape[0]
        dw = torch.empty_like(w)
        db = torch.empty((N,), dtype=torch.float32, device=x.device) if ctx.has_bias else None
        probs = [_gemm((dy, 0), (1, N), (x, 0), (1, K), (dw, 0), (K, 1), N, K, M, rowsum=(db, 0) if db is not None else None)]
        # dX = dY W reduces over N; a wide layer on few rows (T-Net fc3: N = 7056, M ~ 10^2) would be ONE long serial
        # MFMA chain per tile, so the reduction is cut into <= 14 slices (separate problems of the same launch) whose
        # partial products are summed in a fixed order.
        S = min(14, max(1, N // 512)) if (N > 1024 and ((M + 63) // 64) * ((K + 63) // 64) < 64) else 1
        step = (N + S - 1) // S
        step = (step + 3) // 4 * 4  # keep 16-byte alignment of the slices
        S = (N + step - 1) // step
        dxp = torch.empty((S, M, K), dtype=torch.float32, device=x.device)
        for i in range(S):
            n0, n1 = i * step, min(N, (i + 1) * step)
            probs.append(_gemm((dy, n0), (N, 1), (w, n0 * K), (1, K), (dxp, i * M * K), (K, 1), M, K, n1 - n0))
        _gemm_batch(probs)
        dx = dxp[0] if S == 1 else dxp.sum(0)
        return dx, dw, db


def linear(x, w, b=None):
    """F.linear for (..., K) inputs; weights may be Conv1d-shaped (N, K, 1)."""
    w2 = w.reshape(w.shape[0], -1)
    lead = x.shape[:-1]
    return _Linear.apply(x.reshape(-1, x.shape[-1]), w2, b).reshape(*lead, w2.shape[0])


class _RowLn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, relu):
        lib = hip.load()
        hip.require_gpu(x)
        x = hip.dense_f32(x)
        T, F_ = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((T, 2), dtype=torch.float32, device=x.device)
        hip.check(lib.lgd_rowln_fwd(hip.ptr(x), T, F_, int(relu), hip.ptr(y), hip.ptr(stats), hip.stream_ptr()), "lgd_rowln_fwd")
        ctx.save_for_backward(x, stats)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        x, stats = ctx.saved_tensors
        dy = hip.dense_f32(dy)
        dx = torch.empty_like(x)
        hip.check(lib.lgd_rowln_bwd(hip.ptr(x), hip.ptr(dy), hip.ptr(stats), x.shape[0], x.shape[1], int(ctx.relu), hip.ptr(dx),
                                    hip.stream_ptr()), "lgd_rowln_bwd")
        return dx, None


def row_ln(x, relu):
    """LayerNorm over the last axis (no affine, eps 1e-5) [+ ReLU] of a (T, F) tensor."""
    return _RowLn.apply(x, bool(relu))


class _MlpLnRelu(torch.autograd.Function):
    """A ladder of Linear -> LayerNorm(no affine) -> ReLU layers [+ a last Linear without normalisation] as ONE autograd node: the same
    launches as linear() / row_ln() per layer (lgd_gemm_batch, lgd_rowln_fwd / _bwd), issued back to back from one Python frame.  The
    label encoder is 15 such pairs on ~10^2 rows -- kernels of 7-20 us whose per-layer autograd.Function round trips (~30 us of host
    each, forward and backward) were the largest recurring gap of the step at 2 images per GPU.
    apply(n_ln, has_last, x, w_1, b_1, ..., w_n, b_n): n_ln normalised layers, then one plain layer if has_last."""

    @staticmethod
    def forward(ctx, n_ln, has_last, x, *params):
        lib = hip.load()
        hip.require_gpu(x, *[t for t in params if t is not None])
        x = hip.dense_f32(x)
        n = n_ln + (1 if has_last else 0)
        ws = [hip.dense_f32(params[2 * i]).reshape(params[2 * i].shape[0], -1) for i in range(n)]
        bs = [hip.dense_f32(params[2 * i + 1]) if params[2 * i + 1] is not None else None for i in range(n)]
        M = x.shape[0]
        dev = x.device
        st = hip.stream_ptr()
        ins, pres, stats = [], [], []
        cur = x
        for i in range(n):
            N, K = ws[i].shape
            if cur.shape[1] != K:
                raise hip.LgdHipError("layer %d expects %d inputs, got %d" % (i, K, cur.shape[1]))
            y = torch.empty((M, N), dtype=torch.float32, device=dev)
            _gemm_batch([_gemm((cur, 0), (K, 1), (ws[i], 0), (K, 1), (y, 0), (N, 1), M, N, K, bias=(bs[i], 0) if bs[i] is not None else None)])
            ins.append(cur)
            if i < n_ln:
                z = torch.empty_like(y)
                sv = torch.empty((M, 2), dtype=torch.float32, device=dev)
                hip.check(lib.lgd_rowln_fwd(hip.ptr(y), M, N, 1, hip.ptr(z), hip.ptr(sv), st), "lgd_rowln_fwd")
                pres.append(y)
                stats.append(sv)
                cur = z
            else:
                cur = y
        ctx.save_for_backward(*ins, *pres, *stats, *ws)
        ctx.meta = (n_ln, n, [b is not None for b in bs])
        return cur

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        n_ln, n, has_b = ctx.meta
        t = ctx.saved_tensors
        ins, pres, stats, ws = t[:n], t[n:n + n_ln], t[n + n_ln:n + 2 * n_ln], t[n + 2 * n_ln:]
        dy = hip.dense_f32(dy)
        dev = dy.device
        st = hip.stream_ptr()
        M = dy.shape[0]
        grads = [None] * (2 * n)
        need_x = ctx.needs_input_grad[2]
        for i in range(n - 1, -1, -1):
            N, K = ws[i].shape
            if i < n_ln:   # through ReLU + LayerNorm to the gradient of the layer's GEMM output
                d = torch.empty_like(dy)
                hip.check(lib.lgd_rowln_bwd(hip.ptr(pres[i]), hip.ptr(dy), hip.ptr(stats[i]), M, N, 1, hip.ptr(d), st), "lgd_rowln_bwd")
                dy = d
            x = ins[i]
            probs = []
            if ctx.needs_input_grad[3 + 2 * i]:
                dw = torch.empty((N, K), dtype=torch.float32, device=dev)
                db = torch.empty((N,), dtype=torch.float32, device=dev) if has_b[i] and ctx.needs_input_grad[4 + 2 * i] else None
                probs.append(_gemm((dy, 0), (1, N), (x, 0), (1, K), (dw, 0), (K, 1), N, K, M, rowsum=(db, 0) if db is not None else None))
                grads[2 * i], grads[2 * i + 1] = dw, db
            dxp = None
            if i > 0 or need_x:
                # dX = dY W reduces over N; a wide layer on few rows (T-Net fc3: N = 7056) is cut into slices (see _Linear.backward)
                S = min(14, max(1, N // 512)) if (N > 1024 and ((M + 63) // 64) * ((K + 63) // 64) < 64) else 1
                step = ((N + S - 1) // S + 3) // 4 * 4
                S = (N + step - 1) // step
                dxp = torch.empty((S, M, K), dtype=torch.float32, device=dev)
                for j in range(S):
                    n0, n1 = j * step, min(N, (j + 1) * step)
                    probs.append(_gemm((dy, n0), (N, 1), (ws[i], n0 * K), (1, K), (dxp, j * M * K), (K, 1), M, K, n1 - n0))
            if probs:
                _gemm_batch(probs)
            if dxp is not None:
                dy = dxp[0] if dxp.shape[0] == 1 else dxp.sum(0)
        dx = dy if need_x else None
        return (None, None, dx, *grads)


def mlp_ln_relu(x, layers, last=None):
    """row_ln(linear(x, w, b), relu=True) for every (w, b) of `layers` in sequence, then linear(., *last) if given, as one autograd node
    (weights may be Conv1d-shaped (N, K, 1): pointwise convolutions on length-1 sequences).  [ref: spatial_transformer.py:30-40,
    label_encoder.py:243-270]"""
    if not _MLP_FUSED:   # the same launches, one autograd node per op (tests observe the per-layer activations through row_ln)
        for w, b in layers:
            x = row_ln(linear(x, w, b), True)
        return linear(x, *last) if last is not None else x
    params = []
    for w, b in list(layers) + ([last] if last is not None else []):
        params += [w.reshape(w.shape[0], -1) if w.dim() != 2 else w, b]
    return _MlpLnRelu.apply(len(layers), last is not None, x, *params)


_MLP_FUSED = True


class _RowVecMat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, M):
        lib = hip.load()
        hip.require_gpu(x, M)
        x, M = hip.dense_f32(x), hip.dense_f32(M)
        T, k = x.shape
        out = torch.empty_like(x)
        hip.check(lib.lgd_rowvecmat_fwd(hip.ptr(x), hip.ptr(M), T, k, hip.ptr(out), hip.stream_ptr()), "lgd_rowvecmat_fwd")
        ctx.save_for_backward(x, M)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = hip.load()
        x, M = ctx.saved_tensors
        dout = hip.dense_f32(dout)
        dx, dM = torch.empty_like(x), torch.empty_like(M)
        hip.check(lib.lgd_rowvecmat_bwd(hip.ptr(x), hip.ptr(M), hip.ptr(dout), x.shape[0], x.shape[1], hip.ptr(dx), hip.ptr(dM),
                                        hip.stream_ptr()), "lgd_rowvecmat_bwd")
        return dx, dM


def row_vecmat(x, M):
    """out[t] = x[t] @ M[t]; x (T,k), M (T,k,k)  [ref: label_encoder.py:241,248]"""
    return _RowVecMat.apply(x, M)


class _SegMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, off):
        lib = hip.load()
        hip.require_gpu(x, off)
        x = hip.dense_f32(x)
        B, F_ = off.numel() - 1, x.shape[1]
        out = torch.empty_like(x)
        arg = torch.empty((B, F_), dtype=torch.int32, device=x.device)
        hip.check(lib.lgd_segmax_fwd(hip.ptr(x), hip.ptr(off), B, F_, hip.ptr(out), hip.ptr(arg), hip.stream_ptr()), "lgd_segmax_fwd")
        ctx.save_for_backward(off, arg)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = hip.load()
        off, arg = ctx.saved_tensors
        dout = hip.dense_f32(dout)
        dx = torch.empty_like(dout)
        hip.check(lib.lgd_segmax_bwd(hip.ptr(dout), hip.ptr(off), hip.ptr(arg), arg.shape[0], arg.shape[1], hip.ptr(dx),
                                     hip.stream_ptr()), "lgd_segmax_bwd")
        return dx, None


def segment_max_broadcast(x, img_off):
    """per-image max over the image's rows, broadcast back to those rows: (T,F) -> (T,F); img_off (B+1) int32 device.
    [ref: label_encoder.py:195-213 hier_pool + 262-264 repeat]"""
    return _SegMax.apply(x, img_off)


def box_descriptors(boxes_in, classes, in_counts, out_counts, img_h, img_w, num_classes, add_ctx, wh_format):
    """[ref: label_encoder.py:12-115] on the device, one launch: returns (desc (T,4+K), clamped boxes (T,4), out_off (B+1) int32).
    boxes_in (T0,4) / classes (T0,) are the concatenated annotations (device); counts are host lists."""
    lib = hip.load()
    hip.require_gpu(boxes_in)
    dev = boxes_in.device
    B, T = len(out_counts), int(sum(out_counts))
    offs = hip.to_device([_offsets(in_counts), _offsets(out_counts)], torch.int32, dev)
    boxes_in = hip.dense_f32(boxes_in.reshape(-1, 4)) if boxes_in.numel() else torch.zeros((1, 4), device=dev)
    classes = classes.to(torch.int32).contiguous() if classes.numel() else torch.zeros((1,), dtype=torch.int32, device=dev)
    desc = torch.empty((T, 4 + num_classes), dtype=torch.float32, device=dev)
    boxes = torch.empty((T, 4), dtype=torch.float32, device=dev)
    hip.check(lib.lgd_box_descriptors(hip.ptr(boxes_in), hip.ptr(classes), hip.ptr(offs[0]), hip.ptr(offs[1]), B, T, num_classes,
                                      int(img_h), int(img_w), int(add_ctx), int(wh_format), hip.ptr(desc), hip.ptr(boxes),
                                      hip.stream_ptr()), "lgd_box_descriptors")
    return desc, boxes, offs[1]


# ------------------------------------------------------------------------------------------------ focal loss
class _FocalSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, K, alpha, gamma, n_levels, *tensors):
        lib = hip.load()
        logits = [hip.dense_f32(t) for t in tensors[:n_levels]]
        labels = list(tensors[n_levels:])
        hip.require_gpu(*logits, *labels)
        N = logits[0].shape[0]
        for x, y in zip(logits, labels):
            if x.shape[0] != N or x.shape[1] != A * K or y.dtype != torch.int32 or tuple(y.shape) != (N, A) + tuple(x.shape[-2:]) \
                    or not y.is_contiguous():
                raise hip.LgdHipError("focal loss: logits (N,A*K,H,W) / int32 labels (N,A,H,W) expected, got %s / %s %s"
                                      % (tuple(x.shape), tuple(y.shape), y.dtype))
        hw = hip.int_array([v for m in logits for v in m.shape[-2:]])
        dev = logits[0].device
        ws = torch.empty(lib.lgd_focal_ws_doubles(hw, n_levels, N, A, K), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_focal_loss_fwd(hip.ptr_array(logits), hip.ptr_array(labels), hw, n_levels, N, A, K, float(alpha),
                                         float(gamma), hip.ptr(ws), hip.ptr(loss), hip.stream_ptr()), "lgd_focal_loss_fwd")
        ctx.save_for_backward(*logits, *labels)
        ctx.meta = (A, K, float(alpha), float(gamma), n_levels, N, hw)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = hip.load()
        A, K, alpha, gamma, L, N, hw = ctx.meta
        logits, labels = ctx.saved_tensors[:L], ctx.saved_tensors[L:]
        g = g.contiguous().to(torch.float32)
        grads = [torch.empty_like(x) for x in logits]
        hip.check(lib.lgd_focal_loss_bwd(hip.ptr_array(logits), hip.ptr_array(labels), hw, L, N, A, K, alpha, gamma, hip.ptr(g),
                                         hip.ptr_array(grads), hip.stream_ptr()), "lgd_focal_loss_bwd")
        return (None, None, None, None, None, *grads, *([None] * L))


class _FocalSumNorm(torch.autograd.Function):
    """focal sum / normaliser with the gradient written by the FORWARD pass (lgd_focal_loss_fwd_grad): the normaliser is a buffer
    (no gradient) known when the loss is evaluated and the upstream gradient of a loss term in the training step is 1, so
    (1 / normaliser) dsum/dlogits is final when the logits stream for the sum; the backward only rescales if its upstream != 1."""

    @staticmethod
    def forward(ctx, A, K, alpha, gamma, n_levels, normalizer, *tensors):
        import ctypes
        lib = hip.load()
        logits = [hip.dense_f32(t) for t in tensors[:n_levels]]
        labels = list(tensors[n_levels:])
        hip.require_gpu(*logits, *labels)
        N = logits[0].shape[0]
        for x, y in zip(logits, labels):
            if x.shape[0] != N or x.shape[1] != A * K or y.dtype != torch.int32 or tuple(y.shape) != (N, A) + tuple(x.shape[-2:]) \
                    or not y.is_contiguous():
                raise hip.LgdHipError("focal loss: logits (N,A*K,H,W) / int32 labels (N,A,H,W) expected, got %s / %s %s"
                                      % (tuple(x.shape), tuple(y.shape), y.dtype))
        hw = hip.int_array([v for m in logits for v in m.shape[-2:]])
        dev = logits[0].device
        inv = torch.reciprocal(normalizer.detach().to(torch.float32).reshape(()))
        ws = torch.empty(lib.lgd_focal_ws_doubles(hw, n_levels, N, A, K), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grads = [torch.empty_like(x) for x in logits]
        bound = torch.empty(1, dtype=torch.int32, device=dev)   # bound of |gradient| (the class convolution's f16x2 backward takes its scale from it)
        hip.check(lib.lgd_focal_loss_fwd_grad(hip.ptr_array(logits), hip.ptr_array(labels), hw, n_levels, N, A, K, float(alpha),
                                              float(gamma), hip.ptr(inv), hip.ptr(ws), hip.ptr(loss), hip.ptr_array(grads), hip.ptr(bound),
                                              hip.stream_ptr()), "lgd_focal_loss_fwd_grad")
        ctx.save_for_backward(*grads, bound)
        ctx.sizes = (ctypes.c_longlong * n_levels)(*[g.numel() for g in grads])
        return loss * inv

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        lib = hip.load()
        _consume_once(ctx, "focal_loss_sum")
        *grads, bound = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        hip.check(lib.lgd_scale_unless_one(hip.ptr_array(grads), ctx.sizes, len(grads), hip.ptr(g), hip.ptr(bound), hip.stream_ptr()), "lgd_scale_unless_one")
        _amax_tag(grads, bound)
        return (None, None, None, None, None, None, *grads, *([None] * len(grads)))


def focal_loss_sum(raw_logits, label_planes, A, K, alpha, gamma, normalizer=None):
    """sum of fvcore's sigmoid focal loss over non-ignored anchors and classes, evaluated in place on the head's raw
    (N, A*K, H, W) outputs; label_planes: per level (N, A, H, W) int32 (K = background, < 0 = ignore).
    normalizer (a scalar tensor without gradient: detectron2's EMA of the positive count, FCOS's foreground count): returns
    sum / normalizer and writes the logits' gradient during the forward pass (one pass over the logits instead of two)."""
    raw_logits, label_planes = list(raw_logits), list(label_planes)
    if normalizer is not None and torch.is_grad_enabled() and any(x.requires_grad for x in raw_logits):
        return _FocalSumNorm.apply(int(A), int(K), float(alpha), float(gamma), len(raw_logits), normalizer, *raw_logits, *label_planes)
    out = _FocalSum.apply(int(A), int(K), float(alpha), float(gamma), len(raw_logits), *raw_logits, *label_planes)
    return out if normalizer is None else out / normalizer


def _consume_once(ctx, what):
    """the one-pass losses write their gradient buffers in the forward pass and rescale them IN PLACE by the upstream scalar: a second
    backward over the same graph (retain_graph, a per-loss torch.autograd.grad) would scale them again and alias what the first one
    returned -- refuse it instead of returning g^2-scaled gradients silently."""
    if getattr(ctx, "_lgd_consumed", False):
        raise RuntimeError("%s: its gradient buffers were consumed by a previous backward pass (they are written by the forward pass and "
                           "scaled in place); run the forward again instead of backpropagating twice through the same graph" % what)
    ctx._lgd_consumed = True


class _FcosRegCtrLoss(torch.autograd.Function):
    """FCOS loss_box_reg (GIoU, centerness-weighted) and loss_centerness (BCE) on the head's RAW outputs in one launch, the head's
    Scale / ReLU / stride epilogue folded in and the gradients written by the forward pass (lgd_fcos_loss_fwd_grad; the normalisers are
    known when the loss is evaluated, the upstream gradients are 1 in the training step; lgd_scale_unless_one rescales otherwise).
    apply(K, norm_reg, strides, scales (L,), gt_classes (N,R), gt_deltas (N,R,4), gt_centerness (N,R), inv_norm (2,), reg_1..L, ctr_1..L)
    -> (loss_box_reg, loss_centerness)."""

    @staticmethod
    def forward(ctx, K, norm_reg, strides, scales, gt_classes, gt_deltas, gt_ctr, inv_norm, *maps):
        lib = hip.load()
        L = len(maps) // 2
        regs = [hip.dense_f32(t) for t in maps[:L]]
        ctrs = [hip.dense_f32(t) for t in maps[L:]]
        hip.require_gpu(*regs, *ctrs)
        N = regs[0].shape[0]
        R = sum(t.shape[2] * t.shape[3] for t in regs)
        for r_, c_ in zip(regs, ctrs):
            if r_.shape[0] != N or r_.shape[1] != 4 or tuple(c_.shape) != (N, 1) + tuple(r_.shape[2:]):
                raise hip.LgdHipError("FCOS loss: regression (N,4,H,W) / centerness (N,1,H,W) expected, got %s / %s" % (tuple(r_.shape), tuple(c_.shape)))
        gt_classes = gt_classes.contiguous()
        gt_deltas, gt_ctr = hip.dense_f32(gt_deltas), hip.dense_f32(gt_ctr)
        if gt_classes.dtype != torch.int64 or tuple(gt_classes.shape) != (N, R) or tuple(gt_deltas.shape) != (N, R, 4) or tuple(gt_ctr.shape) != (N, R):
            raise hip.LgdHipError("FCOS loss: targets (N,R) int64 / (N,R,4) / (N,R) expected for N=%d R=%d" % (N, R))
        scales = hip.dense_f32(scales.detach()).reshape(-1)
        inv_norm = hip.dense_f32(inv_norm.detach()).reshape(-1)
        if scales.numel() != L or inv_norm.numel() != 2 or len(strides) != L:
            raise hip.LgdHipError("FCOS loss: %d scales / strides and 2 normalisers expected" % L)
        hw = hip.int_array([d for t in regs for d in t.shape[2:]])
        st = (ctypes.c_float * L)(*[float(v) for v in strides])
        dev = regs[0].device
        ws = torch.empty(lib.lgd_fcos_loss_ws_doubles(hw, L, N), dtype=torch.float64, device=dev)
        out = torch.empty(2 + L, dtype=torch.float32, device=dev)
        g_reg = [torch.empty_like(t) for t in regs]
        g_ctr = [torch.empty_like(t) for t in ctrs]
        hip.check(lib.lgd_fcos_loss_fwd_grad(hip.ptr_array(regs), hip.ptr_array(ctrs), hw, st, L, N, int(K), R, hip.ptr(scales), int(norm_reg),
                                             hip.ptr(gt_classes), hip.ptr(gt_deltas), hip.ptr(gt_ctr), hip.ptr(inv_norm), hip.ptr(ws),
                                             hip.ptr(out), hip.ptr_array(g_reg), hip.ptr_array(g_ctr), hip.stream_ptr()), "lgd_fcos_loss_fwd_grad")
        ctx.save_for_backward(out, *g_reg, *g_ctr)
        ctx.L = L
        ctx.sizes = ((ctypes.c_longlong * L)(*[t.numel() for t in g_reg]), (ctypes.c_longlong * L)(*[t.numel() for t in g_ctr]))
        return out[0], out[1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_box, g_c):
        lib = hip.load()
        _consume_once(ctx, "fcos_reg_ctr_loss")
        L = ctx.L
        out, *g = ctx.saved_tensors
        g_reg, g_ctr = list(g[:L]), list(g[L:])
        g_box = (g_box if g_box is not None else torch.zeros((), device=out.device)).contiguous().to(torch.float32)
        g_c = (g_c if g_c is not None else torch.zeros((), device=out.device)).contiguous().to(torch.float32)
        hip.check(lib.lgd_scale_unless_one(hip.ptr_array(g_reg), ctx.sizes[0], L, hip.ptr(g_box), None, hip.stream_ptr()), "lgd_scale_unless_one")
        hip.check(lib.lgd_scale_unless_one(hip.ptr_array(g_ctr), ctx.sizes[1], L, hip.ptr(g_c), None, hip.stream_ptr()), "lgd_scale_unless_one")
        d_scales = out[2:] * g_box if ctx.needs_input_grad[3] else None
        return (None, None, None, d_scales, None, None, None, None, *g_reg, *g_ctr)


def fcos_reg_ctr_loss(raw_regs, ctrs, scales, strides, gt_classes, gt_deltas, gt_centerness, inv_num_targets, inv_num_fg, num_classes,
                      norm_reg_targets=True):
    """(loss_box_reg, loss_centerness) of FCOS from the RAW bbox_pred maps (N,4,H_l,W_l) and the centerness logits (N,1,H_l,W_l):
    per-level Scale, ReLU * stride (or exp), GIoU against the ltrb targets weighted by the centerness targets, BCE of the centerness,
    both normalised, in one launch [ref: thirdparty_heads/fcos.py:533-546, 107-175].  scales: (L,) tensor (gradient flows)."""
    inv = torch.stack((inv_num_targets.reshape(()), inv_num_fg.reshape(())))
    return _FcosRegCtrLoss.apply(int(num_classes), bool(norm_reg_targets), tuple(float(s) for s in strides), scales, gt_classes, gt_deltas,
                                 gt_centerness, inv, *raw_regs, *ctrs)


class _BoxRegSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, K, beta, weights, n_levels, anchors, matched, *tensors):
        lib = hip.load()
        deltas = [hip.dense_f32(t) for t in tensors[:n_levels]]
        labels = list(tensors[n_levels:])
        hip.require_gpu(*deltas, *labels)
        anchors, matched = hip.dense_f32(anchors), hip.dense_f32(matched)
        N, R = deltas[0].shape[0], anchors.shape[0]
        for x, y in zip(deltas, labels):
            if x.shape[0] != N or x.shape[1] != A * 4 or y.dtype != torch.int32 or tuple(y.shape) != (N, A) + tuple(x.shape[-2:]) \
                    or not y.is_contiguous():
                raise hip.LgdHipError("box-reg loss: deltas (N,A*4,H,W) / int32 labels (N,A,H,W) expected, got %s / %s %s"
                                      % (tuple(x.shape), tuple(y.shape), y.dtype))
        if tuple(matched.shape) != (N, R, 4):
            raise hip.LgdHipError("box-reg loss: matched boxes (N,R,4) expected, got %s" % (tuple(matched.shape),))
        hw = hip.int_array([v for m in deltas for v in m.shape[-2:]])
        w4 = (ctypes.c_float * 4)(*[float(v) for v in weights])
        dev = deltas[0].device
        ws = torch.empty(lib.lgd_box_reg_ws_doubles(hw, n_levels, N, A), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        hip.check(lib.lgd_box_reg_loss_fwd(hip.ptr_array(deltas), hip.ptr_array(labels), hw, n_levels, N, A, K, hip.ptr(anchors),
                                           hip.ptr(matched), R, float(beta), ctypes.cast(w4, ctypes.c_void_p), hip.ptr(ws),
                                           hip.ptr(loss), hip.stream_ptr()), "lgd_box_reg_loss_fwd")
        ctx.save_for_backward(anchors, matched, *deltas, *labels)
        ctx.meta = (A, K, float(beta), [float(v) for v in weights], n_levels, N, R, hw)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = hip.load()
        A, K, beta, weights, L, N, R, hw = ctx.meta
        anchors, matched = ctx.saved_tensors[:2]
        deltas, labels = ctx.saved_tensors[2:2 + L], ctx.saved_tensors[2 + L:]
        w4 = (ctypes.c_float * 4)(*weights)
        g = g.contiguous().to(torch.float32)
        grads = [torch.empty_like(x) for x in deltas]
        hip.check(lib.lgd_box_reg_loss_bwd(hip.ptr_array(deltas), hip.ptr_array(labels), hw, L, N, A, K, hip.ptr(anchors),
                                           hip.ptr(matched), R, beta, ctypes.cast(w4, ctypes.c_void_p), hip.ptr(g),
                                           hip.ptr_array(grads), hip.stream_ptr()), "lgd_box_reg_loss_bwd")
        return (None, None, None, None, None, None, None, *grads, *([None] * L))


def box_reg_loss_sum(raw_deltas, label_planes, anchors, matched_boxes, A, K, beta, weights=(1.0, 1.0, 1.0, 1.0)):
    """sum over positive anchors of smooth-L1(pred deltas, Box2BoxTransform deltas of the matched box), evaluated in place on
    the head's raw (N, A*4, H, W) outputs with the focal loss's int32 label planes; anchors (R,4), matched_boxes (N,R,4)."""
    raw_deltas, label_planes = list(raw_deltas), list(label_planes)
    return _BoxRegSum.apply(int(A), int(K), float(beta), tuple(weights), len(raw_deltas), anchors, matched_boxes,
                            *raw_deltas, *label_planes)


def label_planes(labels, level_hw, A):
    """(N, R) integer anchor labels ordered (level, y, x, a) -> per level (N, A, H, W) int32 planes."""
    out, off = [], 0
    N = labels.shape[0]
    for h, w in level_hw:
        n = h * w * A
        out.append(labels[:, off:off + n].view(N, h, w, A).permute(0, 3, 1, 2).to(torch.int32).contiguous())
        off += n
    return out


# ------------------------------------------------------------------------------------------------ K8: 3x3 convolutions
# G of F(tile x tile, 3x3) (tests/test_host_cpu.py::test_winograd_matrices_define_the_convolution holds the kernels' matrices to the
# definition; the device code carries its own copies in csrc/winograd.hip / winograd6.hip)
_WINO_G = {4: [[1 / 4, 0.0, 0.0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
               [1 / 24, -1 / 12, 1 / 6], [0.0, 0.0, 1.0]],
           6: [[1.0, 0.0, 0.0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
               [32 / 45, 16 / 45, 8 / 45], [32 / 45, -16 / 45, 8 / 45], [0.0, 0.0, 1.0]]}
_WINO_MASK_DTYPE = {4: torch.int16, 6: torch.int64}   # per-tile activation masks: 16 bits per 4x4 tile, 36 of 64 bits per 6x6 tile


def _freq_buf(nf, C, T, device):
    """frequency buffer in the kernels' layout [C][nf][T], returned as its (nf, C, T) view: the per-frequency GEMMs see
    (C x T) matrices with row stride nf*T (a plain leading dimension for the BLAS), nothing is copied."""
    return torch.empty((C, nf, T), dtype=torch.float32, device=device).permute(1, 0, 2)


class _FilterImage:
    """the filter operand of one channel product as csrc/gemm3.hip takes it: the bf16x3 image (a uint8 tensor) of an (nf, M, K) fp32 operand"""
    __slots__ = ("t", "shape")

    def __init__(self, t, shape):
        self.t, self.shape = t, tuple(shape)

    @property
    def device(self):
        return self.t.device


def _pack_filter(a):
    """(tensor for ctx.save_for_backward, shape or None) of a filter operand that is an fp32 tensor or a _FilterImage"""
    return (a.t, a.shape) if isinstance(a, _FilterImage) else (a, None)


def _unpack_filter(t, shape):
    return _FilterImage(t, shape) if shape is not None else t


def _wino_filters(lib, ws, scales, Ci, dev, tile, T=None, need_dx=True):
    """The transformed filter of K convolutions stacked along C_out, as the two channel products take it:
         A_fwd for M = U V            -- U (nf, sum Co, Ci) fp32, or its gemm3 image,
         A_dx  for dV = U^T dM        -- the (Co, Ci)-transposed VIEW of U (tuned TN library solutions are as fast as NN on a materialised
                                         U^T, tools/gemm_tn_probe.py), or the gemm3 image of U^T; None when no input gradient is needed.
    Where csrc/gemm3.hip will run the product (T given, shape gate as in _gemm3_ok) lgd_wino_filter_images writes the bf16x3 image straight
    from the filter transform: fp32 U is neither written nor re-read by a split pass.  The frozen per-channel scale of a FrozenBN that
    follows the conv is folded in on the way (no scaled copy of the weights)."""
    Cos = [w.shape[0] for w in ws]
    Ct = sum(Cos)
    nf = (tile + 2) ** 2
    img_ok = (tile == 6 and T is not None and _GEMM3_ON and _FILTER_IMAGES and Ci % 16 == 0 and all(c % 16 == 0 for c in Cos) and ws[0].is_cuda)
    fwd_img = img_ok and _gemm3_shape_ok(nf, Ct, Ci, T, dev)
    dx_img = img_ok and need_dx and _gemm3_shape_ok(nf, Ci, Ct, T, dev)
    need_u = (not fwd_img) or (need_dx and not dx_img)
    U = torch.empty((nf, Ct, Ci), dtype=torch.float32, device=dev) if need_u else None
    imf = torch.empty(lib.lgd_gemm3_image_bytes(nf, Ct, Ci), dtype=torch.uint8, device=dev) if fwd_img else None
    imb = torch.empty(lib.lgd_gemm3_image_bytes(nf, Ci, Ct), dtype=torch.uint8, device=dev) if dx_img else None
    c0 = 0
    for w, sc, Co in zip(ws, scales, Cos):
        if need_u:
            hip.check(lib.lgd_wino_filter_fwd(hip.ptr(w), hip.ptr(sc) if sc is not None else None, Co, Ci, tile,
                                              ctypes.c_void_p(U.data_ptr() + 4 * c0 * Ci), Ct * Ci, None, 0, 0, hip.stream_ptr()),
                      "lgd_wino_filter_fwd")
        if fwd_img or dx_img:
            hip.check(lib.lgd_wino_filter_images(hip.ptr(w), hip.ptr(sc) if sc is not None else None, Co, Ci, tile, c0, Ct,
                                                 hip.ptr(imf) if fwd_img else None, hip.ptr(imb) if dx_img else None, hip.stream_ptr()),
                      "lgd_wino_filter_images")
        c0 += Co
    a_fwd = _FilterImage(imf, (nf, Ct, Ci)) if fwd_img else U
    a_dx = (_FilterImage(imb, (nf, Ci, Ct)) if dx_img else U.transpose(1, 2)) if need_dx else None
    return a_fwd, a_dx


def _wino_filter_grads(lib, dU, scales, Cos, need, Ci, tile):
    """dw_k = scale_k . G^T dU_k G for the filters that need it; dU (nf, sum Co, Ci) from the weight-gradient GEMM, or (S, nf, sum Co, Ci): the
    split-K partials of csrc/h2.hip, summed in fixed order by the transform while it reads them."""
    Ct = sum(Cos)
    out, c0 = [], 0
    parts = dU.dim() == 4
    for sc, Co, nd in zip(scales, Cos, need):
        dw = None
        if nd and parts:
            dw = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=dU.device)
            hip.check(lib.lgd_wino_filter_bwd_parts(ctypes.c_void_p(dU.data_ptr() + 4 * c0 * Ci), Ct * Ci, dU.stride(0), dU.shape[0],
                                                    hip.ptr(sc) if sc is not None else None, Co, Ci, hip.ptr(dw), hip.stream_ptr()), "lgd_wino_filter_bwd_parts")
        elif nd:
            dw = torch.empty((Co, Ci, 3, 3), dtype=torch.float32, device=dU.device)
            hip.check(lib.lgd_wino_filter_bwd(ctypes.c_void_p(dU.data_ptr() + 4 * c0 * Ci), Ct * Ci, hip.ptr(sc) if sc is not None else None,
                                              Co, Ci, tile, hip.ptr(dw), hip.stream_ptr()), "lgd_wino_filter_bwd")
        out.append(dw)
        c0 += Co
    return out


def _wino_out(lib, Mk, bias, hw, L, N, Co, tile, relu, ys, bits):
    """output transform of Co channels of M (+ bias, ReLU, mask bits); with the f16x2 pipeline on (tile 6) it also leaves max |y| for the
    next convolution's scale and tags the maps with it"""
    if tile == 6 and _tags_wanted(sum(N * ((y.shape[2] + 5) // 6) * ((y.shape[3] + 5) // 6) for y in ys), ys):
        amax = _zero_words(ys[0].device)
        hip.check(lib.lgd_wino_out_amax(hip.ptr(Mk), hip.ptr(bias) if bias is not None else None, hw, L, N, Co, int(relu), hip.ptr_array(ys),
                                        hip.ptr(bits) if bits is not None else None, hip.ptr(amax), hip.stream_ptr()), "lgd_wino_out_amax")
        _amax_tag(ys, amax)
    else:
        hip.check(lib.lgd_wino_out(hip.ptr(Mk), hip.ptr(bias) if bias is not None else None, hw, L, N, Co, tile, int(relu), hip.ptr_array(ys),
                                   hip.ptr(bits) if bits is not None else None, hip.stream_ptr()), "lgd_wino_out")


def _wino_in_t(lib, dV, hw, L, N, Ci, tile, dxs, pre_bits):
    """adjoint input transform (+ activation mask); tile 6 with the f16x2 pipeline on: also max |dx|, tagged on the gradient maps"""
    if tile == 6 and _tags_wanted(sum(N * ((d.shape[2] + 5) // 6) * ((d.shape[3] + 5) // 6) for d in dxs), dxs):
        amax = _zero_words(dxs[0].device)
        hip.check(lib.lgd_wino_in_t_amax(hip.ptr(dV), hw, L, N, Ci, hip.ptr_array(dxs), hip.ptr(pre_bits) if pre_bits is not None else None,
                                         hip.ptr(amax), hip.stream_ptr()), "lgd_wino_in_t_amax")
        _amax_tag(dxs, amax)
    else:
        hip.check(lib.lgd_wino_in_t(hip.ptr(dV), hw, L, N, Ci, tile, hip.ptr_array(dxs), hip.ptr(pre_bits) if pre_bits is not None else None,
                                    hip.stream_ptr()), "lgd_wino_in_t")


class _Conv3x3K(torch.autograd.Function):
    """K filters nn.Conv2d(Ci, Co_k, 3, stride 1, padding 1) [+ ReLU] applied to the SAME L maps (the pyramid levels; K = 1: one
    conv, K = 2: e.g. the first convs of the cls / bbox towers, which read the same features) in the minimal-filtering form
    F(tile x tile, 3x3), tile = 6 or 4: HIP data transforms (lgd_wino_in / lgd_wino_out / lgd_wino_out_t / lgd_wino_in_t) around
    per-frequency channel GEMMs (hipBLASLt / rocBLAS fp32 MFMA through torch.bmm) over the concatenated tiles of all levels.  The
    input is transformed ONCE for all K filters (their U are stacked along C_out: one GEMM), and the backward sums their input
    gradients inside the dV GEMM (K = sum Co_k) -- one adjoint input transform, no gradient-accumulation pass.  Forward, input
    gradient and weight gradient all run at 64/324 (tile 6) or 1/4 (tile 4) of the direct multiplies.  The backward is the autograd of
    the pipeline itself: dy is expanded ONCE (dM = A dy A^T), dV[f] = U[f]^T dM[f] comes back through the adjoint of the input
    transform, the weight gradient is dU[f] = dM[f] V[f]^T.
    apply(K, relu, tile, scales, pre, w_1, b_1, ..., w_K, b_K, x_1, ..., x_L) -> K * L maps, filter-major; scales: None or one per-output-
    channel factor (a buffer, no gradient) per filter, applied to the filter inside its transform; pre: None, or a per-INPUT-
    channel bias (a buffer): the maps are then pre-activations and the convolution runs on relu(x + pre[c]) -- the bias + ReLU epilogue
    of the producing 1x1 convolution folded into the input transform, its backward mask into the adjoint transform, so the
    gradient returned for x is the gradient of the RAW map.  pre of shape (L*N, Ci, 2): per (map, sample, channel) scale and shift of a
    GroupNorm + ReLU that precedes the convolution (group_norm_fold); the gradient returned for x is then the gradient w.r.t. the
    GroupNorm OUTPUT, which group_norm_fold's backward turns into the gradient of the raw map."""

    @staticmethod
    def forward(ctx, K, relu, tile, scales, pre, *args):
        ws, bs, xs = list(args[0:2 * K:2]), list(args[1:2 * K:2]), list(args[2 * K:])
        if tile not in _WINO_MASK_DTYPE:
            raise hip.LgdHipError("Winograd output tile must be 4 or 6")
        scales = list(scales) if scales is not None else [None] * K
        hip.require_gpu(*ws, *xs)
        lib = hip.load()
        ws = [hip.dense_f32(w) for w in ws]
        xs = [_dense_tagged(x) for x in xs]
        bs = [hip.dense_f32(b) if b is not None else None for b in bs]
        L, N, Ci = len(xs), xs[0].shape[0], xs[0].shape[1]
        Cos = [w.shape[0] for w in ws]
        Ct = sum(Cos)
        dev = ws[0].device
        nf = (tile + 2) ** 2
        mdt, mb = _WINO_MASK_DTYPE[tile], _WINO_MASK_DTYPE[tile].itemsize
        hw = hip.int_array([d for x in xs for d in x.shape[2:]])
        T = lib.lgd_wino_tiles(hw, L, N, tile)
        need_dx = any(ctx.needs_input_grad[5 + 2 * K:])
        h2 = _h2_ok(tile, Ci, Cos, T, dev)
        pre = hip.dense_f32(pre) if pre is not None else None
        affine = pre is not None and pre.dim() == 3   # (L*N, Ci, 2) scale / shift per (map, sample, channel): a folded GroupNorm + ReLU
        if affine and tuple(pre.shape) != (L * N, Ci, 2):
            raise hip.LgdHipError("affine pre-activation must be (L*N, C, 2) = (%d, %d, 2), got %s" % (L * N, Ci, tuple(pre.shape)))
        pre_bits = (torch.empty((Ci, T), dtype=mdt, device=dev)
                    if pre is not None and need_dx else None)
        px = 4 * N * sum(x.shape[2] * x.shape[3] for x in xs)  # bytes of one channel of the maps
        fb = 4 * nf * T                                        # bytes of one channel of a frequency buffer
        p_bias, p_aff = (hip.ptr(pre) if pre is not None and not affine else None), (hip.ptr(pre) if affine else None)
        uinv = vinv = None
        if h2:   # K10: V written as f16x2 split rows, the filter as an f16x2 image, the product on csrc/h2.hip
            filt = _h2_filters(lib, ws, scales, Ci, dev, need_dx)
            amax = _amax_bits(lib, xs, hw, pre, affine)
            V, vinv, uinv = _h2_buf(Ci, T, dev), torch.empty(1, dtype=torch.float32, device=dev), filt.inv
            hip.check(lib.lgd_wino_in_h2(hip.ptr_array(xs), hw, L, N, Ci, hip.ptr(V), p_bias, p_aff, hip.ptr(pre_bits) if pre_bits is not None else None,
                                         hip.ptr(amax), hip.ptr(vinv), hip.stream_ptr()), "lgd_wino_in_h2")
            _count_bytes("wino_in_kernel", (px + fb) * Ci)
            M = _h2_product(lib, "fwd", filt.fwd, Ct, Ci, V, vinv, False, uinv, _freq_buf(nf, Ct, T, dev))
            Ut = filt.bwd
        else:
            U, Ut = _wino_filters(lib, ws, scales, Ci, dev, tile, T, need_dx)
            V = _freq_buf(nf, Ci, T, dev)
            hip.check(lib.lgd_wino_in(hip.ptr_array(xs), hw, L, N, Ci, tile, hip.ptr(V), p_bias, p_aff,
                                      hip.ptr(pre_bits) if pre_bits is not None else None, hip.stream_ptr()), "lgd_wino_in")
            _count_bytes("wino_in_kernel", (px + fb) * Ci)
            M = _wino_gemm("wino_gemm_fwd", U, V, out=_freq_buf(nf, Ct, T, dev))
        # ReLU mask for the backward: one bit per pixel, a table entry per tile, written by the output transform, so the backward
        # reads 1 bit instead of 4 bytes per pixel and the forward output is not kept alive
        bits = torch.empty((Ct, T), dtype=mdt, device=dev) if relu else None
        ys, c0 = [], 0
        for k in range(K):   # one output transform per filter: its channels are a contiguous slab of M ([C][nf][T])
            yk = [torch.empty((N, Cos[k]) + tuple(x.shape[2:]), dtype=torch.float32, device=dev) for x in xs]
            _count_bytes("wino_out_kernel", (px + fb + (mb * T if bits is not None else 0)) * Cos[k])
            _wino_out(lib, M[:, c0], bs[k], hw, L, N, Cos[k], tile, relu, yk, bits[c0] if bits is not None else None)
            ys += yk
            c0 += Cos[k]
        need_w = any(ctx.needs_input_grad[5:5 + 2 * K:2])
        ut_t, ctx.ut_shape = (Ut, None) if h2 else _pack_filter(Ut)
        ctx.dev, ctx.h2 = dev, h2
        ctx.save_for_backward(ut_t, V if need_w else None, bits, pre_bits, uinv, vinv)   # the backward needs U^T (dV = U^T dM)
        ctx.scales = scales
        ctx.meta = (K, L, N, Ci, Cos, hw, T, [b is not None for b in bs], [tuple(x.shape[2:]) for x in xs], tile, px, fb)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        Ut, V, bits, pre_bits, uinv, vinv = ctx.saved_tensors
        if not ctx.h2:
            Ut = _unpack_filter(Ut, ctx.ut_shape)
        K, L, N, Ci, Cos, hw, T, has_bias, shapes, tile, px, fb = ctx.meta
        Ct = sum(Cos)
        lib = hip.load()
        dev = ctx.dev
        nf = (tile + 2) ** 2
        mb = _WINO_MASK_DTYPE[tile].itemsize
        # an output nothing downstream used arrives as None
        dys = [_dense_tagged(g) if g is not None else torch.zeros((N, Cos[i // L]) + shapes[i % L], dtype=torch.float32, device=dev)
               for i, g in enumerate(dys)]
        need_ws = list(ctx.needs_input_grad[5:5 + 2 * K:2])
        need_bs = [hb and nb for hb, nb in zip(has_bias, ctx.needs_input_grad[6:6 + 2 * K:2])]
        need_w, need_x = any(need_ws), any(ctx.needs_input_grad[5 + 2 * K:])
        dws, dbs = [None] * K, [None] * K
        dxs = [None] * L
        if not (need_x or need_w or any(need_bs)):
            return (None, None, None, None, None, *[None] * (2 * K), *dxs)
        if ctx.h2:
            dM, dminv = _h2_buf(Ct, T, dev), torch.empty(64, dtype=torch.float32, device=dev)
            amax = _amax_bits_groups(lib, [dys[k * L:(k + 1) * L] for k in range(K)], hw)   # ONE scale for the stacked gradients: K = C_out in dV = U^T dM
        else:
            dM = _freq_buf(nf, Ct, T, dev)
        c0 = 0
        for k in range(K):
            gk = dys[k * L:(k + 1) * L]
            _count_bytes("wino_out_t_kernel", (px + fb + (mb * T if bits is not None else 0)) * Cos[k])
            if ctx.h2:
                hip.check(lib.lgd_wino_out_t_h2(hip.ptr_array(gk), hip.ptr(bits[c0]) if bits is not None else None, hw, L, N, Cos[k],
                                                ctypes.c_void_p(dM.data_ptr() + 4 * c0 * nf * T), hip.ptr(amax), hip.ptr(dminv), hip.stream_ptr()),
                          "lgd_wino_out_t_h2")
            else:
                hip.check(lib.lgd_wino_out_t(hip.ptr_array(gk), hip.ptr(bits[c0]) if bits is not None else None, hw, L, N, Cos[k], tile,
                                             hip.ptr(dM[:, c0]), hip.stream_ptr()), "lgd_wino_out_t")
            c0 += Cos[k]
        if need_x:
            _count_bytes("wino_in_t_kernel", (px + fb) * Ci)
            if ctx.h2:
                dV = _h2_product(lib, "dx", Ut, Ci, Ct, dM, dminv, True, uinv, _freq_buf(nf, Ci, T, dev))
            else:
                dV = _wino_gemm("wino_gemm_dx", Ut, dM, out=_freq_buf(nf, Ci, T, dev))
            dxs = [torch.empty((N, Ci) + s, dtype=torch.float32, device=dev) for s in shapes]
            _wino_in_t(lib, dV, hw, L, N, Ci, tile, dxs, pre_bits)
            del dV
        if need_w:
            dU = _h2_dw(lib, dM, dminv, V, vinv, Ct, Ci) if ctx.h2 else _timed_bmm("wino_gemm_dw", dM, V.transpose(1, 2))
            dws = _wino_filter_grads(lib, dU, ctx.scales, Cos, need_ws, Ci, tile)
        if any(need_bs):
            # A's row of the interpolation point 1 is all ones: that frequency of dM = A g A^T is the tile's gradient sum
            db = _h2_plane_sums(dM, tile + 3, dminv) if ctx.h2 else dM[tile + 3].sum(1)
            c0 = 0
            for k in range(K):
                dbs[k] = db[c0:c0 + Cos[k]] if need_bs[k] else None
                c0 += Cos[k]
        return (None, None, None, None, None, *[g for pair in zip(dws, dbs) for g in pair], *dxs)


class _Conv3x3GN(torch.autograd.Function):
    """_Conv3x3K (tile 6, no ReLU, no filter scale) whose K outputs each feed a GroupNorm(groups) + ReLU that the NEXT convolution applies
    while it loads (group_norm_fold's forward half: statistics -> per-(map, sample, channel) scale / shift), as ONE autograd node, so
    that the GroupNorm's backward apply pass disappears as well: the node receives the gradients w.r.t. the GroupNorm OUTPUTS (the next
    convolution's adjoint input transform has applied the ReLU bits), one statistics pass over (y, g) yields the coefficients
    (lgd_gn_group_bwd_coef) and the adjoint output transform computes dy = ca * g - cm - (y - mean) * cb on the blocks it loads
    (lgd_wino_out_t_gn): per tower layer the backward moves 2 + 2 maps instead of 2 + 3 + 1 and launches one kernel less
    [ref: thirdparty_heads/fcos.py:455-470, 520-531].
    apply(K, tile, groups, pre, w_1, b_1, gamma_1, beta_1, ..., x_1, ..., x_L) -> (affine_1, ..., affine_K, K * L raw maps, filter-major)."""

    @staticmethod
    def forward(ctx, K, tile, groups, pre, *args):
        ws, bs = list(args[0:4 * K:4]), list(args[1:4 * K:4])
        gammas, betas = list(args[2:4 * K:4]), list(args[3:4 * K:4])
        xs = list(args[4 * K:])
        if tile != 6:
            raise hip.LgdHipError("the fused GroupNorm backward exists for F(6x6,3x3) only")
        hip.require_gpu(*ws, *xs)
        lib = hip.load()
        ws = [hip.dense_f32(w) for w in ws]
        xs = [_dense_tagged(x) for x in xs]
        bs = [hip.dense_f32(b) if b is not None else None for b in bs]
        gammas = [hip.dense_f32(g) if g is not None else None for g in gammas]
        betas = [hip.dense_f32(b) if b is not None else None for b in betas]
        L, N, Ci = len(xs), xs[0].shape[0], xs[0].shape[1]
        Cos = [w.shape[0] for w in ws]
        Ct = sum(Cos)
        dev = ws[0].device
        nf = (tile + 2) ** 2
        mdt = _WINO_MASK_DTYPE[tile]
        hw = hip.int_array([d for x in xs for d in x.shape[2:]])
        T = lib.lgd_wino_tiles(hw, L, N, tile)
        need_dx = any(ctx.needs_input_grad[4 + 4 * K:])
        h2 = _h2_ok(tile, Ci, Cos, T, dev)
        pre = hip.dense_f32(pre) if pre is not None else None
        affine_in = pre is not None and pre.dim() == 3
        if affine_in and tuple(pre.shape) != (L * N, Ci, 2):
            raise hip.LgdHipError("affine pre-activation must be (L*N, C, 2) = (%d, %d, 2), got %s" % (L * N, Ci, tuple(pre.shape)))
        pre_bits = (torch.empty((Ci, T), dtype=mdt, device=dev) if pre is not None and need_dx else None)
        px = 4 * N * sum(x.shape[2] * x.shape[3] for x in xs)
        fb = 4 * nf * T
        p_bias, p_aff = (hip.ptr(pre) if pre is not None and not affine_in else None), (hip.ptr(pre) if affine_in else None)
        uinv = vinv = None
        if h2:   # K10: as _Conv3x3K
            filt = _h2_filters(lib, ws, [None] * K, Ci, dev, need_dx)
            amax = _amax_bits(lib, xs, hw, pre, affine_in)
            V, vinv, uinv = _h2_buf(Ci, T, dev), torch.empty(1, dtype=torch.float32, device=dev), filt.inv
            hip.check(lib.lgd_wino_in_h2(hip.ptr_array(xs), hw, L, N, Ci, hip.ptr(V), p_bias, p_aff, hip.ptr(pre_bits) if pre_bits is not None else None,
                                         hip.ptr(amax), hip.ptr(vinv), hip.stream_ptr()), "lgd_wino_in_h2")
            _count_bytes("wino_in_kernel", (px + fb) * Ci)
            M = _h2_product(lib, "fwd", filt.fwd, Ct, Ci, V, vinv, False, uinv, _freq_buf(nf, Ct, T, dev))
            Ut = filt.bwd
        else:
            U, Ut = _wino_filters(lib, ws, [None] * K, Ci, dev, tile, T, need_dx)
            V = _freq_buf(nf, Ci, T, dev)
            hip.check(lib.lgd_wino_in(hip.ptr_array(xs), hw, L, N, Ci, tile, hip.ptr(V), p_bias, p_aff,
                                      hip.ptr(pre_bits) if pre_bits is not None else None, hip.stream_ptr()), "lgd_wino_in")
            _count_bytes("wino_in_kernel", (px + fb) * Ci)
            M = _wino_gemm("wino_gemm_fwd", U, V, out=_freq_buf(nf, Ct, T, dev))
        ys, affs, stats, yamax, c0 = [], [], [], [], 0
        for k in range(K):
            yk = [torch.empty((N, Cos[k]) + tuple(x.shape[2:]), dtype=torch.float32, device=dev) for x in xs]
            _count_bytes("wino_out_kernel", (px + fb) * Cos[k])
            _wino_out(lib, M[:, c0], bs[k], hw, L, N, Cos[k], tile, False, yk, None)
            tag = getattr(yk[0], "_lgd_amax", None)
            yamax.append(tag[0] if tag is not None else None)   # max |y|: the backward's bound of the GroupNorm gradient needs it
            # (the statistics in the output transform's epilogue -- per-plane fp64 sums by atomics -- were measured: the transform slows down by
            #  what the pass costs)
            gws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, N, Cos[k]), dtype=torch.float64, device=dev)
            st = torch.empty((L * N * groups, 2), dtype=torch.float32, device=dev)
            aff = torch.empty((L * N, Cos[k], 2), dtype=torch.float32, device=dev)
            _count_bytes("gn_group_stats_kernel", px * Cos[k])
            hip.check(lib.lgd_gn_group_stats_affine(hip.ptr_array(yk), hw, L, N, Cos[k], groups,
                                                    hip.ptr(gammas[k]) if gammas[k] is not None else None,
                                                    hip.ptr(betas[k]) if betas[k] is not None else None, hip.ptr(gws), hip.ptr(st), hip.ptr(aff),
                                                    hip.stream_ptr()), "lgd_gn_group_stats_affine")
            ys += yk
            affs.append(aff)
            stats.append(st)
            c0 += Cos[k]
        need_w = any(ctx.needs_input_grad[4:4 + 4 * K:4])
        ut_t, ctx.ut_shape = (Ut, None) if h2 else _pack_filter(Ut)
        ctx.dev, ctx.h2, ctx.yamax = dev, h2, yamax
        ctx.save_for_backward(ut_t, V if need_w else None, pre_bits, uinv, vinv, *stats, *gammas, *ys)
        ctx.meta = (K, L, N, Ci, Cos, hw, T, [b is not None for b in bs], [b is not None for b in betas],
                    [tuple(x.shape[2:]) for x in xs], tile, groups, px, fb)
        ctx.mark_non_differentiable(*affs)
        return (*affs, *ys)

    @staticmethod
    def backward(ctx, *grads):
        K, L, N, Ci, Cos, hw, T, has_bias, has_beta, shapes, tile, groups, px, fb = ctx.meta
        Ut, V, pre_bits, uinv, vinv = ctx.saved_tensors[:5]
        if not ctx.h2:
            Ut = _unpack_filter(Ut, ctx.ut_shape)
        stats = ctx.saved_tensors[5:5 + K]
        gammas = ctx.saved_tensors[5 + K:5 + 2 * K]
        ys = ctx.saved_tensors[5 + 2 * K:]
        gs = grads[K:]
        Ct = sum(Cos)
        lib = hip.load()
        dev = ctx.dev
        nf = (tile + 2) ** 2
        gs = [_dense_tagged(g) if g is not None else torch.zeros((N, Cos[i // L]) + shapes[i % L], dtype=torch.float32, device=dev)
              for i, g in enumerate(gs)]
        nig = ctx.needs_input_grad
        need_ws = list(nig[4:4 + 4 * K:4])
        need_bs = [hb and nb for hb, nb in zip(has_bias, nig[5:5 + 4 * K:4])]
        need_w, need_x = any(need_ws), any(nig[4 + 4 * K:])
        dws, dbs, dgs, dbes = [None] * K, [None] * K, [None] * K, [None] * K
        dxs = [None] * L
        h2 = ctx.h2
        if h2:
            dM, dminv, bound = _h2_buf(Ct, T, dev), torch.empty(64, dtype=torch.float32, device=dev), _zero_words(dev)
        else:
            dM = _freq_buf(nf, Ct, T, dev)
        coefs, c0 = [], 0
        for k in range(K):
            gk, yk = gs[k * L:(k + 1) * L], list(ys[k * L:(k + 1) * L])
            gws = torch.empty(lib.lgd_gn_group_ws_doubles(hw, L, N, Cos[k]), dtype=torch.float64, device=dev)
            bst = torch.empty((L * N * groups, 2), dtype=torch.float32, device=dev)
            psums = torch.empty((L * N, Cos[k], 2), dtype=torch.float32, device=dev)
            coef = torch.empty((L * N, Cos[k], 4), dtype=torch.float32, device=dev)
            _count_bytes("gn_group_bwd_stats_kernel", 2 * px * Cos[k])
            wmax = torch.empty(gws.numel(), dtype=torch.float32, device=dev) if h2 else None   # chunk maxima of |g| and |xhat|: the gradient's bound per plane
            hip.check(lib.lgd_gn_group_bwd_coef(hip.ptr_array(yk), hip.ptr_array(gk), hw, L, N, Cos[k], groups,
                                                hip.ptr(gammas[k]) if gammas[k] is not None else None, hip.ptr(stats[k]), hip.ptr(gws),
                                                hip.ptr(bst), hip.ptr(psums), hip.ptr(coef), hip.ptr(wmax) if h2 else None, hip.ptr(bound) if h2 else None,
                                                hip.stream_ptr()), "lgd_gn_group_bwd_coef")
            _count_bytes("wino_out_t_gn_kernel", (2 * px + fb) * Cos[k])
            if h2:   # ONE scale for the stacked gradients: the bound (lgd_gn_group_bwd_coef) accumulates over the K filters, the transforms run behind the loop
                coefs.append(coef)
            else:
                hip.check(lib.lgd_wino_out_t_gn(hip.ptr_array(gk), hip.ptr_array(yk), hip.ptr(coef), hw, L, N, Cos[k], tile, hip.ptr(dM[:, c0]),
                                                hip.stream_ptr()), "lgd_wino_out_t_gn")
            if (gammas[k] is not None and nig[6 + 4 * k]) or (has_beta[k] and nig[7 + 4 * k]):
                s = psums.sum(0)
                dgs[k] = s[:, 1].contiguous() if gammas[k] is not None and nig[6 + 4 * k] else None
                dbes[k] = s[:, 0].contiguous() if has_beta[k] and nig[7 + 4 * k] else None
            c0 += Cos[k]
        if h2:
            c0 = 0
            for k in range(K):
                gk, yk = gs[k * L:(k + 1) * L], list(ys[k * L:(k + 1) * L])
                hip.check(lib.lgd_wino_out_t_gn_h2(hip.ptr_array(gk), hip.ptr_array(yk), hip.ptr(coefs[k]), hw, L, N, Cos[k],
                                                   ctypes.c_void_p(dM.data_ptr() + 4 * c0 * nf * T), hip.ptr(bound), hip.ptr(dminv), hip.stream_ptr()),
                          "lgd_wino_out_t_gn_h2")
                c0 += Cos[k]
        if need_x:
            _count_bytes("wino_in_t_kernel", (px + fb) * Ci)
            if h2:
                dV = _h2_product(lib, "dx", Ut, Ci, Ct, dM, dminv, True, uinv, _freq_buf(nf, Ci, T, dev))
            else:
                dV = _wino_gemm("wino_gemm_dx", Ut, dM, out=_freq_buf(nf, Ci, T, dev))
            dxs = [torch.empty((N, Ci) + sh, dtype=torch.float32, device=dev) for sh in shapes]
            _wino_in_t(lib, dV, hw, L, N, Ci, tile, dxs, pre_bits)
            del dV
        if need_w:
            dU = _h2_dw(lib, dM, dminv, V, vinv, Ct, Ci) if h2 else _timed_bmm("wino_gemm_dw", dM, V.transpose(1, 2))
            dws = _wino_filter_grads(lib, dU, [None] * K, Cos, need_ws, Ci, tile)
        if any(need_bs):
            db = _h2_plane_sums(dM, tile + 3, dminv) if h2 else dM[tile + 3].sum(1)   # the frequency of the interpolation point 1: the tile's gradient sum
            c0 = 0
            for k in range(K):
                dbs[k] = db[c0:c0 + Cos[k]] if need_bs[k] else None
                c0 += Cos[k]
        return (None, None, None, None, *[g for quad in zip(dws, dbs, dgs, dbes) for g in quad], *dxs)


class _Conv3x3Chain(torch.autograd.Function):
    """K convolutions 3x3 / stride 1 / padding 1 in SEQUENCE over the same L maps, conv k [+ ReLU if relus[k]] feeding conv k+1 and
    nothing else (the head towers after their first conv incl. the score conv, the adapter: distillator.py:107-109 ->
    retinanet.py:36-43, sequential_convs.py:10-12).  The forward is the per-conv Winograd pipeline of _Conv3x3K.  The backward keeps
    the gradient in the FREQUENCY domain across a link: dV_k = U_k^T dM_k goes through ONE kernel (lgd_wino_in_t_out_t: adjoint input
    transform, ReLU mask of conv k-1, A . A^T) into dM_{k-1}; the intermediate gradient maps are neither written nor re-read.
    apply(K, relus, tile, w_1, b_1, ..., w_K, b_K, x_1, ..., x_L) -> the L maps of the last conv."""

    @staticmethod
    def forward(ctx, K, relus, tile, *args):
        ws, bs, xs = list(args[0:2 * K:2]), list(args[1:2 * K:2]), list(args[2 * K:])
        hip.require_gpu(*ws, *xs)
        lib = hip.load()
        nf = (tile + 2) ** 2
        mdt, mb = _WINO_MASK_DTYPE[tile], _WINO_MASK_DTYPE[tile].itemsize
        ws = [hip.dense_f32(w) for w in ws]
        xs = [_dense_tagged(x) for x in xs]
        bs = [hip.dense_f32(b) if b is not None else None for b in bs]
        L, N = len(xs), xs[0].shape[0]
        dev = ws[0].device
        hw = hip.int_array([d for x in xs for d in x.shape[2:]])
        shapes = [tuple(x.shape[2:]) for x in xs]
        T = lib.lgd_wino_tiles(hw, L, N, tile)
        px = 4 * N * sum(h * w_ for h, w_ in shapes)   # bytes of one channel of the maps
        fb = 4 * nf * T                                # bytes of one channel of a frequency buffer
        need_ws = list(ctx.needs_input_grad[3:3 + 2 * K:2])
        saved, cur, ut_shapes, dims, h2s = [], xs, [], [], []
        for k in range(K):
            Co, Ci = ws[k].shape[0], ws[k].shape[1]
            need_dx = k > 0 or any(ctx.needs_input_grad[3 + 2 * K:])
            h2 = _h2_ok(tile, Ci, [Co], T, dev)
            uinv = vinv = None
            if h2:
                filt = _h2_filters(lib, [ws[k]], [None], Ci, dev, need_dx)
                amax = _amax_bits(lib, cur, hw)   # (inside the chain: the tag the previous output transform left)
                V, vinv, uinv = _h2_buf(Ci, T, dev), torch.empty(1, dtype=torch.float32, device=dev), filt.inv
                hip.check(lib.lgd_wino_in_h2(hip.ptr_array(cur), hw, L, N, Ci, hip.ptr(V), None, None, None, hip.ptr(amax), hip.ptr(vinv),
                                             hip.stream_ptr()), "lgd_wino_in_h2")
                _count_bytes("wino_in_kernel", (px + fb) * Ci)
                M = _h2_product(lib, "fwd", filt.fwd, Co, Ci, V, vinv, False, uinv, _freq_buf(nf, Co, T, dev))
                ut_t, ut_shape = filt.bwd, None
            else:
                U, Ut = _wino_filters(lib, [ws[k]], [None], Ci, dev, tile, T, need_dx)
                V = _freq_buf(nf, Ci, T, dev)
                hip.check(lib.lgd_wino_in(hip.ptr_array(cur), hw, L, N, Ci, tile, hip.ptr(V), None, None, None, hip.stream_ptr()), "lgd_wino_in")
                _count_bytes("wino_in_kernel", (px + fb) * Ci)
                M = _wino_gemm("wino_gemm_fwd", U, V, out=_freq_buf(nf, Co, T, dev))
                ut_t, ut_shape = _pack_filter(Ut)
            bits = torch.empty((Co, T), dtype=mdt, device=dev) if relus[k] else None
            cur = [torch.empty((N, Co) + s, dtype=torch.float32, device=dev) for s in shapes]
            _count_bytes("wino_out_kernel", (px + fb + (mb * T if bits is not None else 0)) * Co)
            _wino_out(lib, M, bs[k], hw, L, N, Co, tile, relus[k], cur, bits)
            del M
            ut_shapes.append(ut_shape)
            dims.append((Ci, Co))
            h2s.append(h2)
            saved += [ut_t, V if need_ws[k] else None, bits, uinv, vinv]
        ctx.ut_shapes, ctx.dims, ctx.dev, ctx.h2s = ut_shapes, dims, dev, h2s
        ctx.save_for_backward(*saved)
        ctx.meta = (K, L, N, hw, T, shapes, [b is not None for b in bs], px, fb, tile)
        return tuple(cur)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        K, L, N, hw, T, shapes, has_bias, px, fb, tile = ctx.meta
        lib = hip.load()
        nf = (tile + 2) ** 2
        mb = _WINO_MASK_DTYPE[tile].itemsize
        dev = ctx.dev
        h2s = ctx.h2s
        need_ws = list(ctx.needs_input_grad[3:3 + 2 * K:2])
        need_bs = [hb and nb for hb, nb in zip(has_bias, ctx.needs_input_grad[4:4 + 2 * K:2])]
        need_x = any(ctx.needs_input_grad[3 + 2 * K:])
        dws, dbs, dxs = [None] * K, [None] * K, [None] * L
        Co = ctx.dims[K - 1][1]
        dys = [_dense_tagged(g) if g is not None else torch.zeros((N, Co) + shapes[i], dtype=torch.float32, device=dev) for i, g in enumerate(dys)]

        def out_t(maps, bits_k, C, h2):
            """dM = A (g . mask) A^T of C channels, as split rows (h2) or fp32; returns (dM, per-frequency inverse scales or None)"""
            _count_bytes("wino_out_t_kernel", (px + fb + (mb * T if bits_k is not None else 0)) * C)
            if h2:
                buf, inv = _h2_buf(C, T, dev), torch.empty(64, dtype=torch.float32, device=dev)
                amax = _amax_bits(lib, maps, hw)
                hip.check(lib.lgd_wino_out_t_h2(hip.ptr_array(maps), hip.ptr(bits_k) if bits_k is not None else None, hw, L, N, C, hip.ptr(buf),
                                                hip.ptr(amax), hip.ptr(inv), hip.stream_ptr()), "lgd_wino_out_t_h2")
                return buf, inv
            buf = _freq_buf(nf, C, T, dev)
            hip.check(lib.lgd_wino_out_t(hip.ptr_array(maps), hip.ptr(bits_k) if bits_k is not None else None, hw, L, N, C, tile, hip.ptr(buf),
                                         hip.stream_ptr()), "lgd_wino_out_t")
            return buf, None

        dM, dminv = out_t(dys, saved[5 * (K - 1) + 2], Co, h2s[K - 1])
        for k in range(K - 1, -1, -1):
            Ut, V, uinv, vinv = saved[5 * k], saved[5 * k + 1], saved[5 * k + 3], saved[5 * k + 4]
            if not h2s[k]:
                Ut = _unpack_filter(Ut, ctx.ut_shapes[k])
            Ci, Co = ctx.dims[k]
            if need_ws[k]:
                dU = _h2_dw(lib, dM, dminv, V, vinv, Co, Ci) if h2s[k] else _timed_bmm("wino_gemm_dw", dM, V.transpose(1, 2))
                dws[k] = _wino_filter_grads(lib, dU, [None], [Co], [True], Ci, tile)[0]
            if need_bs[k]:   # A's row of the interpolation point 1 is all ones: the tile's gradient sum
                dbs[k] = _h2_plane_sums(dM, tile + 3, dminv) if h2s[k] else dM[tile + 3].sum(1)
            if k == 0 and not need_x:
                break
            fused_h2 = k > 0 and h2s[k] and h2s[k - 1]
            if h2s[k]:
                amax64 = _zero_words(dev, 64) if fused_h2 else None   # max |dV[f]|: the bound of the link's dx
                dV = _h2_product(lib, "dx", Ut, Ci, Co, dM, dminv, True, uinv, _freq_buf(nf, Ci, T, dev), amax64)
            else:
                dV = _wino_gemm("wino_gemm_dx", Ut, dM, out=_freq_buf(nf, Ci, T, dev))
            del dM
            if k > 0:   # the link to conv k-1: dM_{k-1} = A (in_t(dV) . relu mask) A^T without the map in between
                pb = saved[5 * (k - 1) + 2]
                if fused_h2:
                    bound = torch.empty(1, dtype=torch.int32, device=dev)
                    hip.check(lib.lgd_h2_link_bound(hip.ptr(amax64), hip.ptr(bound), hip.stream_ptr()), "lgd_h2_link_bound")
                    dM, dminv = _h2_buf(Ci, T, dev), torch.empty(64, dtype=torch.float32, device=dev)
                    _count_bytes("wino_in_t_out_t_kernel", (2 * fb + (mb * T if pb is not None else 0)) * Ci)
                    hip.check(lib.lgd_wino_in_t_out_t_h2(hip.ptr(dV), hip.ptr(pb) if pb is not None else None, hw, L, N, Ci, hip.ptr(dM), hip.ptr(bound),
                                                         hip.ptr(dminv), hip.stream_ptr()), "lgd_wino_in_t_out_t_h2")
                elif h2s[k - 1]:   # (conv k on the fp32 path: no per-frequency maxima of dV) through the gradient map
                    mid = [torch.empty((N, Ci) + s, dtype=torch.float32, device=dev) for s in shapes]
                    _count_bytes("wino_in_t_kernel", (px + fb) * Ci)
                    _wino_in_t(lib, dV, hw, L, N, Ci, tile, mid, None)
                    dM, dminv = out_t(mid, pb, Ci, True)
                    del mid
                else:
                    dM, dminv = _freq_buf(nf, Ci, T, dev), None
                    _count_bytes("wino_in_t_out_t_kernel", (2 * fb + (mb * T if pb is not None else 0)) * Ci)
                    hip.check(lib.lgd_wino_in_t_out_t(hip.ptr(dV), hip.ptr(pb) if pb is not None else None, hw, L, N, Ci, tile, hip.ptr(dM),
                                                      hip.stream_ptr()), "lgd_wino_in_t_out_t")
            else:
                _count_bytes("wino_in_t_kernel", (px + fb) * Ci)
                dxs = [torch.empty((N, Ci) + s, dtype=torch.float32, device=dev) for s in shapes]
                _wino_in_t(lib, dV, hw, L, N, Ci, tile, dxs, None)
            del dV
        return (None, None, None, *[g for pair in zip(dws, dbs) for g in pair], *dxs)


class _Conv3x3:
    """single-filter form of _Conv3x3K with the historical argument order: apply(w, b, relu, tile, *xs)."""

    @staticmethod
    def apply(w, b, relu, tile, *xs, scale=None, pre=None):
        return _Conv3x3K.apply(1, bool(relu), tile, None if scale is None else (scale,), pre, w, b, *xs)


def enable_tuned_gemms(path=None):
    """Library GEMM solution selection for the Winograd channel products: lgd_amd/tuning/tunableop_gfx950.csv holds the
    fastest rocBLAS / hipBLASLt solution per GEMM shape of the BASELINE configs, measured on an MI355X with torch's
    TunableOp (regenerate: PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=... python bench.py).  Lookup only --
    nothing is tuned at run time; shapes not in the table (or a table from another ROCm build) use the default."""
    if os.environ.get("LGD_TUNED_GEMM", "1") == "0" or "PYTORCH_TUNABLEOP_ENABLED" in os.environ or not torch.cuda.is_available():
        return False
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")
    if not os.path.exists(path):
        return False
    global _TUNED_GEMM
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(False)
    _TUNED_GEMM = bool(tunable.read_file(path))  # TunableOp rejects a table whose validators (ROCm / BLAS versions) differ
    if not _TUNED_GEMM:
        tunable.enable(False)
    return _TUNED_GEMM


_TUNED_GEMM = False  # set by enable_tuned_gemms(); Trainer / bench.py opt in, importing this module changes nothing
# output tile of the minimal-filtering form: 6 -> F(6x6,3x3) (64 frequencies, 1.78 multiplies per output pixel), 4 -> F(4x4,3x3) (36, 2.25)
_WINO_TILE = 6
# smallest problem (2x2-pixel blocks over all maps of the call) that takes the Winograd path; measured at config 4 (R-101, 2 img/GPU,
# whose res5 3x3 convolutions have 546): 2000 -> 35.7, 500 -> 35.1, 100 -> 35.3 ms/step in one call
_WINO_MIN_TILES = 500
_WINO_MIN_CH = 64
_WINO_MIN_CO = 16   # single-map convolutions with fewer output channels stay on the library (16: the 27-channel offset convolution of a DCNv2 block
                    # runs here -- config 5 43.4 -> 42.75 ms in one call against MIOpen's Winograd forward / data + implicit-GEMM weight kernels)
_WINO_ON = True


class _LibConv2d(torch.autograd.Function):
    """the vendor library's convolution (MIOpen through torch) as ONE autograd node whose forward AND backward run inside
    streams.library_call: what stays on the library -- tiny maps under the Winograd threshold, strided / 7x7 / grouped convolutions -- is
    never in flight on two streams at once, also when autograd runs the backward on a side stream (a plain F.conv2d's backward is issued
    by the engine, outside anybody's guard)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dilation, groups):
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, padding, dilation, groups, b is not None)
        with streams.library_call(x.device):
            return F.conv2d(x, w, b, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups, has_b = ctx.geom
        mask = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2])
        with streams.library_call(x.device):
            dx, dw, db = torch.ops.aten.convolution_backward(dy, x, w, [w.shape[0]] if has_b else None, list(stride), list(padding), list(dilation),
                                                             False, [0, 0], groups, list(mask))
        return dx, dw, (db if mask[2] else None), None, None, None, None


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(e) for e in v)


def lib_conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    """F.conv2d on the vendor library, ordered against every other library call of the step (see _LibConv2d / streams.library_call)"""
    if not x.is_cuda:
        return F.conv2d(x, w, b, stride, padding, dilation, groups)
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        return _LibConv2d.apply(x, w, b, _pair(stride), _pair(padding), _pair(dilation), int(groups))
    with streams.library_call(x.device):
        return F.conv2d(x, w, b, stride, padding, dilation, groups)


def conv3x3_backend(winograd=None, min_tiles=None, tile=None):
    """which implementation the 3x3 convolutions take: tests force the Winograd kernels onto problems below the production
    threshold (min_tiles = 0), or the library's convolutions onto large ones (winograd = False) to compare the two; tile selects
    F(6x6,3x3) or F(4x4,3x3).  Returns the previous (winograd, min_tiles, tile)."""
    global _WINO_ON, _WINO_MIN_TILES, _WINO_TILE
    prev = (_WINO_ON, _WINO_MIN_TILES, _WINO_TILE)
    if winograd is not None:
        _WINO_ON = bool(winograd)
    if min_tiles is not None:
        _WINO_MIN_TILES = int(min_tiles)
    if tile is not None:
        if int(tile) not in _WINO_MASK_DTYPE:
            raise ValueError("tile must be 4 or 6")
        _WINO_TILE = int(tile)
    return prev


def _wino_ok(xs, w):
    x = xs[0]
    tiles = sum(x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2) for x in xs)
    return (_WINO_ON and x.is_cuda and x.dtype == torch.float32 and tiles >= _WINO_MIN_TILES and x.shape[1] >= _WINO_MIN_CH
            and (len(xs) > 1 or w.shape[0] >= _WINO_MIN_CO))


def conv3x3_levels(xs, w, b=None, relu=False, scale=None, pre=None):
    """one 3x3 / stride 1 / padding 1 filter [+ ReLU] over a list of maps (the FPN levels): a single Winograd pass over
    the concatenated tiles.  Tiny problems stay on the library's direct kernels.  scale: per-output-channel factor on the filter
    (the frozen affine of a FrozenBN after the conv), folded into the filter transform on the Winograd path.  pre: per-input-channel
    bias of a bias + ReLU that precedes the convolution (the maps are its pre-activations), folded into the input transform on the
    Winograd path (see _Conv3x3K); elsewhere applied as its own pass."""
    xs = list(xs)
    _check_pre(xs, pre)
    if _wino_ok(xs, w):
        return list(_Conv3x3.apply(w, b, bool(relu), _WINO_TILE, *xs, scale=scale, pre=pre))
    if pre is not None:
        xs = _apply_pre(xs, pre)
    if scale is not None:
        w = w * scale.view(-1, 1, 1, 1)
    ys = [lib_conv2d(x, w, b, 1, 1) for x in xs]
    return [F.relu_(y) for y in ys] if relu else ys


def _apply_pre(xs, pre):
    """the pre-activation of conv3x3_levels as its own pass (problems that stay on the library's convolutions): per-channel bias + ReLU,
    or the (L*N, C, 2) scale / shift of a folded GroupNorm + ReLU"""
    if pre.dim() == 1:
        return [bias_act(x, pre, None, True) for x in xs]
    # the gradient handed back for x must be the gradient w.r.t. the affine OUTPUT (mask only, as the adjoint input transform of the
    # Winograd path returns it; group_norm_fold's backward does the rest): value of the affine, derivative 1
    N = xs[0].shape[0]
    out = []
    for l, x in enumerate(xs):
        t = x.detach() * pre[l * N:(l + 1) * N, :, 0, None, None] + pre[l * N:(l + 1) * N, :, 1, None, None]
        out.append(F.relu(x + (t - x.detach())))
    return out


def conv3x3_shared_input(xs, filters, relu=False, pre=None):
    """several 3x3 / stride 1 / padding 1 filters [(w, b), ...] [+ ReLU] on the SAME list of maps: one input transform, one
    stacked GEMM, one adjoint input transform for the summed input gradient (see _Conv3x3K).  Returns one list of maps per filter."""
    xs = list(xs)
    _check_pre(xs, pre)
    if len(filters) > 1 and all(_wino_ok(xs, w) for w, _ in filters):
        ys = _Conv3x3K.apply(len(filters), bool(relu), _WINO_TILE, None, pre, *[t for wb in filters for t in wb], *xs)
        return [list(ys[k * len(xs):(k + 1) * len(xs)]) for k in range(len(filters))]
    if pre is not None:
        xs = _apply_pre(xs, pre)
    return [conv3x3_levels(xs, w, b, relu) for w, b in filters]


_GN_FUSED_BWD = True   # False: conv + group_norm_fold as two autograd nodes (tests compare the two)


def conv3x3_gn(xs, filters, groups, pre=None):
    """filters [(w, b, gamma, beta), ...] on the SAME maps, each followed by GroupNorm(groups, gamma, beta) + ReLU that the NEXT convolution
    applies while it loads: returns [(affine, raw maps), ...] -- pass both on as conv3x3_levels(maps, ..., pre=affine) / conv3x3_gn(maps,
    ..., pre=affine).  One autograd node per call on the F(6x6,3x3) path (_Conv3x3GN: no GroupNorm apply pass in either direction);
    elsewhere conv3x3_levels / conv3x3_shared_input + group_norm_fold."""
    xs = list(xs)
    _check_pre(xs, pre)
    L = len(xs)
    if _GN_FUSED_BWD and _WINO_TILE == 6 and all(_wino_ok(xs, f[0]) for f in filters):
        K = len(filters)
        out = _Conv3x3GN.apply(K, _WINO_TILE, int(groups), pre, *[t for f in filters for t in f], *xs)
        return [_tag_folded(out[k], list(out[K + k * L:K + (k + 1) * L])) for k in range(K)]
    if len(filters) > 1:
        ys = conv3x3_shared_input(xs, [(f[0], f[1]) for f in filters], pre=pre)
    else:
        ys = [conv3x3_levels(xs, filters[0][0], filters[0][1], pre=pre)]
    return [group_norm_fold(y, groups, f[2], f[3]) for y, f in zip(ys, filters)]


def conv3x3_chain(xs, filters, relus, fused_links=True):
    """filters [(w, b), ...] applied in sequence to a list of maps, ReLU after conv k where relus[k]; the maps between two convs have no
    other consumer, so the backward crosses each link in the frequency domain (see _Conv3x3Chain).  fused_links = False: every conv
    as its own autograd node (what the chain is tested against)."""
    xs = list(xs)
    relus = tuple(bool(r) for r in relus)
    if fused_links and len(filters) > 1 and all(_wino_ok(xs, w) and w.shape[1] >= _WINO_MIN_CH for w, _ in filters):
        return list(_Conv3x3Chain.apply(len(filters), relus, _WINO_TILE, *[t for wb in filters for t in wb], *xs))
    for (w, b), r in zip(filters, relus):
        xs = conv3x3_levels(xs, w, b, r)
    return xs


def conv3x3(x, w, b=None, relu=False, scale=None, pre=None):
    """single-map form of conv3x3_levels."""
    return conv3x3_levels([x], w, b, relu, scale, pre)[0]


def conv3x3_folds_pre(N, C, H, W, Cout, device, dtype=torch.float32):
    """True if conv3x3 on a (N, C, H, W) map would take the Winograd path, i.e. can fold a preceding bias + ReLU into its input
    transform: the producer may then hand over its raw output (student/resnet.py::Bottleneck)."""
    tiles = N * ((H + 1) // 2) * ((W + 1) // 2)
    return (_WINO_ON and device.type == "cuda" and dtype == torch.float32 and tiles >= _WINO_MIN_TILES
            and C >= _WINO_MIN_CH and Cout >= _WINO_MIN_CH)


def side_streams_ok():
    """whether the independent chains of the step may fork onto the side stream at all (lgd_amd/streams.py); LGD_SIDE_STREAMS=0 switches every fork
    off (A/B runs).  WHICH chain may fork is decided per call by convs_on_own_kernels().
    With a stream PER fork (LGD_ONE_SIDE_STREAM=0, the round-5 form) the forks are also off when the process asks HIP for more than its default 4
    hardware queues (GPU_MAX_HW_QUEUES > 4): a fork whose two chains wait for each other then costs 19-22 ms per step (config 2: 71 ms against 52)
    -- not in the kernels (two streams of lgd_h2_fwd overlap 5 % FASTER than one under any queue count, tools/queue_probe.py) and not under
    rocprofv3 (every dispatch carries a completion signal there): the cross-queue waits themselves stall (profiles/r06_hw_queues_and_forks.txt).
    The shipped form -- ONE side stream for all forks, two streams in the process -- measures the same under 4 and 8 queues.
    LGD_SIDE_STREAMS=force overrides."""
    v = os.environ.get("LGD_SIDE_STREAMS", "1")
    if v == "0":
        return False
    if v == "force" or streams._ONE_SIDE:
        return True
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) <= 4
    except ValueError:
        return True


def convs_on_own_kernels(xs, filter_groups):
    """True if every 3x3 convolution over the maps xs -- filter_groups: one list of weights per convolution call, several weights = filters stacked on
    one input (conv3x3_shared_input / conv3x3_gn) -- runs on THIS library's kernels only: F(6x6) transforms and the f16x2 products of csrc/h2.hip,
    forward, input AND weight gradient.  The per-call gate of the step's forks: a chain goes to a side stream only then.

    Why: round 5's two-stream stall, reproduced and named in round 6 (profiles/r06_stall_root_cause.txt) -- two rocBLAS GEMMs in flight on two streams
    share the single workspace of torch's per-thread handle, and the split-K kernel of one spins for ever on flags the other reset.  Chains that
    contain calls of the vendor library (problems under the size gates, C' = 36 / 80 score convolutions, the F(4x4) and library A/B variants) stay on
    the step's main stream, where stream order serialises them; streams.library_call additionally ORDERS any library call that does reach a side
    stream, so nothing depends on this predicate for correctness -- only for speed (an ordered call on a side stream waits for the main stream).
    LGD_SIDE_STREAMS_ANY=1: every chain may fork (tools/stall_repro.sh: with LGD_LIBRARY_ORDER=0 the stall reproduces)."""
    if os.environ.get("LGD_SIDE_STREAMS_ANY") == "1":
        return True
    xs = list(xs)
    if not xs or not xs[0].is_cuda or _WINO_TILE != 6:
        return False
    hw = hip.int_array([d for x in xs for d in x.shape[2:]])
    T = hip.load().lgd_wino_tiles(hw, len(xs), xs[0].shape[0], 6)
    Ci = xs[0].shape[1]
    for ws in filter_groups:
        if not all(_wino_ok(xs, w) and w.shape[1] == Ci for w in ws):
            return False
        if not _h2_ok(6, Ci, [w.shape[0] for w in ws], T, xs[0].device):
            return False
        Ci = ws[0].shape[0] if len(ws) == 1 else Ci   # (a chain: the next convolution reads this one's output; stacked filters end a chain)
    return True


def conv3x3_stride2(x, w, b=None):
    """3x3 / stride 2 / padding 1 convolution (the FPN's extra levels p6 / p7 [d2-memory: LastLevelP6P7]).  Output pixel (i, j) of the
    strided convolution is output pixel (2i, 2j) of the stride-1 one, and F(4x4,3x3) spends 36 / 16 = 2.25 (F(6x6,3x3): 1.78) multiplies
    per stride-1 output pixel against 9 / 4 = 2.25 per input pixel for the direct strided form: the same GEMM work or less, so where the Winograd path
    applies (p6 over res5: 2048 -> 256 channels) the convolution runs as the stride-1 Winograd convolution + every other output,
    forward, input and weight gradient on the tuned channel GEMMs.  The library's strided implicit-GEMM kernels cost 2.1 ms/step at
    config 2 (forward, two split weight-gradient passes and NCHW <-> NHWC transposes of the 2048-channel map)."""
    if _wino_ok([x], w):
        y = conv3x3(x, w, b)
        z = y[:, :, ::2, ::2]
        tag = getattr(y, "_lgd_amax", None)
        if tag is not None and tag[1] == y._version:   # (a subset of the pixels: the bound of the whole map holds)
            _amax_tag([z], tag[0])
        return z
    return lib_conv2d(x, w, b, 2, 1)


class Conv3x3(torch.nn.Conv2d):
    """nn.Conv2d(cin, cout, 3, 1, 1) with the same parameters / state_dict keys; `levels` applies it to a pyramid."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, 3, 1, 1)

    def forward(self, x):
        return conv3x3(x, self.weight, self.bias)

    def levels(self, xs, relu=False, pre=None):
        return conv3x3_levels(xs, self.weight, self.bias, relu, pre=pre)


# ------------------------------------------------------------------------------------------------ timing
# ------------------------------------------------------------------------------------------------ anchor matching
def anchor_match(anchors, gt_boxes, gt_classes, counts, iou_lo, iou_hi, num_classes, allow_low_quality=True, img_off=None):
    """detectron2 Matcher + label assignment for the whole mini-batch in two launches (no IoU matrix):
    anchors (R,4), gt_boxes (T,4) / gt_classes (T,) concatenated image-major, counts = boxes per image.
    Returns labels (B,R) int64 (class id, num_classes = background, -1 = ignore) and matched boxes (B,R,4)."""
    hip.require_gpu(anchors)
    anchors = hip.dense_f32(anchors)
    B, R, T = len(counts), anchors.shape[0], int(sum(counts))
    dev = anchors.device
    if img_off is None:
        img_off = hip.to_device(_offsets(counts), torch.int32, dev)
    labels = torch.empty((B, R), dtype=torch.int64, device=dev)
    matched = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    if T:
        gt_boxes = hip.dense_f32(gt_boxes)
        gt_classes = gt_classes.to(torch.int64).contiguous()
        ws = torch.empty((T,), dtype=torch.int32, device=dev)
    hip.check(hip.load().lgd_anchor_match(hip.ptr(anchors), R, hip.ptr(gt_boxes) if T else None,
                                          hip.ptr(gt_classes) if T else None, hip.ptr(img_off), B, T, float(iou_lo), float(iou_hi),
                                          int(num_classes), int(bool(allow_low_quality)), hip.ptr(ws) if T else None,
                                          hip.ptr(labels), hip.ptr(matched), hip.stream_ptr()), "lgd_anchor_match")
    return labels, matched


def fcos_targets(shifts, strides, sizes_of_interest, gt_boxes, gt_classes, counts, num_classes, radius, img_off=None):
    """FCOS ground-truth assignment for the mini-batch in one launch [ref: thirdparty_heads/fcos.py:177-284]: shifts = list of
    per-level (HW,2) centres; gt_boxes (T,4) / gt_classes (T,) concatenated image-major (None when T = 0), counts = boxes
    per image.  Returns classes (B,R) int64 (num_classes = background), ltrb deltas (B,R,4), centerness (B,R)."""
    hip.require_gpu(*shifts)
    dev = shifts[0].device
    pts = hip.dense_f32(torch.cat(list(shifts), 0))
    L, R, B, T = len(shifts), pts.shape[0], len(counts), int(sum(counts))
    if img_off is None:
        img_off = hip.to_device(_offsets(counts), torch.int32, dev)
    locs = hip.int_array([s.shape[0] for s in shifts])
    fa = lambda v: (ctypes.c_float * len(v))(*[float(x) for x in v])  # noqa: E731
    lo, hi = fa([s[0] for s in sizes_of_interest]), fa([s[1] for s in sizes_of_interest])
    rad = fa([float(st) * float(radius) for st in strides])
    cls = torch.empty((B, R), dtype=torch.int64, device=dev)
    deltas = torch.empty((B, R, 4), dtype=torch.float32, device=dev)
    ctr = torch.empty((B, R), dtype=torch.float32, device=dev)
    if T:
        gt_boxes = hip.dense_f32(gt_boxes)
        gt_classes = gt_classes.to(torch.int64).contiguous()
    c = lambda a: ctypes.cast(a, ctypes.c_void_p)  # noqa: E731
    hip.check(hip.load().lgd_fcos_targets(hip.ptr(pts), c(locs), c(lo), c(hi), c(rad), L, R, hip.ptr(gt_boxes) if T else None,
                                          hip.ptr(gt_classes) if T else None, hip.ptr(img_off), B, T, int(num_classes),
                                          int(radius > 0), hip.ptr(cls), hip.ptr(deltas), hip.ptr(ctr), hip.stream_ptr()),
              "lgd_fcos_targets")
    return cls, deltas, ctr


# ------------------------------------------------------------------------------------------------ DCNv2 (config 5)
class _DeformConv(torch.autograd.Function):
    """modulated deformable 3x3 convolution: one gather kernel builds the column matrix, the channel contraction is a library
    GEMM; backward = GEMM for d col, one kernel for dx / d offset / d mask, GEMM for dW (the column matrix is kept)."""

    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias, stride, padding, dilation, packed=False):
        # packed: `offset` is the offset convolution's own (N, 27, Ho, Wo) output -- channels 0..17 the offsets, 18..26 the mask LOGITS
        # (the kernels read it in place and apply the sigmoid; the backward returns ONE (N, 27, Ho, Wo) gradient: no chunk / cat / sigmoid passes)
        hip.require_gpu(x, offset, weight)
        lib = hip.load()
        x, offset, weight = hip.dense_f32(x), hip.dense_f32(offset), hip.dense_f32(weight)
        mask = hip.dense_f32(mask) if mask is not None else None
        N, C, H, W = x.shape
        O = weight.shape[0]
        Ho = (H + 2 * padding - 2 * dilation - 1) // stride + 1
        Wo = (W + 2 * padding - 2 * dilation - 1) // stride + 1
        if tuple(weight.shape[1:]) != (C, 3, 3) or tuple(offset.shape) != (N, 27 if packed else 18, Ho, Wo) or (packed and mask is not None):
            raise hip.LgdHipError("deform conv: weight (O,C,3,3) / offset (N,18,Ho,Wo) [packed: (N,27,Ho,Wo), no mask] expected, got %s / %s"
                                  % (tuple(weight.shape), tuple(offset.shape)))
        col = torch.empty((N, C * 9, Ho * Wo), dtype=torch.float32, device=x.device)
        if packed:
            hip.check(lib.lgd_dcn_im2col_packed(hip.ptr(x), hip.ptr(offset), N, C, H, W, stride, padding, dilation, hip.ptr(col),
                                                hip.stream_ptr()), "lgd_dcn_im2col_packed")
        else:
            hip.check(lib.lgd_dcn_im2col(hip.ptr(x), hip.ptr(offset), hip.ptr(mask) if mask is not None else None, N, C, H, W,
                                         stride, padding, dilation, hip.ptr(col), hip.stream_ptr()), "lgd_dcn_im2col")
        # batched GEMM with the filter as a stride-0 batch: torch.matmul(2-D, 3-D) folds the batch by transposing + copying the whole
        # column matrix (and the result back): 135 strided copies = 8 of config 5's 58 ms of kernels per step (rocprofv3)
        with streams.library_call(x.device):
            out = torch.bmm(weight.view(1, O, C * 9).expand(N, O, C * 9), col).view(N, O, Ho, Wo)
        if bias is not None:
            out = out + bias.view(1, -1, 1, 1)
        ctx.save_for_backward(x, offset, mask, weight, col)
        ctx.geom = (stride, padding, dilation, bias is not None, bool(packed))
        return out

    @staticmethod
    def backward(ctx, dy):
        x, offset, mask, weight, col = ctx.saved_tensors
        stride, padding, dilation, has_bias, packed = ctx.geom
        lib = hip.load()
        N, C, H, W = x.shape
        O = weight.shape[0]
        dy = hip.dense_f32(dy).view(N, O, -1)
        dx = doff = dmask = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or (mask is not None and ctx.needs_input_grad[2]):
            with streams.library_call(x.device):
                dcol = torch.bmm(weight.view(1, O, C * 9).transpose(1, 2).expand(N, C * 9, O), dy)
            # dx | d offset | d mask | per-cell contribution lists (dx by gather instead of the L2's float atomics, csrc/dcn.hip; _DCN_GATHER =
            # False: atomic scatter) carved out of ONE allocation in the order the entry point zeroes them: one fill instead of four
            def r4(n):
                return (n + 3) // 4 * 4
            n_dx, n_off, n_m = x.numel(), offset.numel(), (mask.numel() if mask is not None else 0)
            n_ws = (int(lib.lgd_dcn_ws_bytes(N, H, W)) + 3) // 4 if _DCN_GATHER else 0
            gap = 0 if _DCN_ONE_FILL else 64    # A/B: 256-byte holes between the regions = four fills, as with four allocations
            o_off = r4(n_dx) + gap
            o_m = o_off + n_off + gap
            o_ws = r4(o_m + n_m) + gap
            buf = torch.empty(o_ws + n_ws, dtype=torch.float32, device=x.device)
            dx = buf[:n_dx].view_as(x)
            doff = buf[o_off:o_off + n_off].view_as(offset)
            dmask = buf[o_m:o_m + n_m].view_as(mask) if mask is not None else None
            ws = buf[o_ws:] if _DCN_GATHER else None
            if packed:
                hip.check(lib.lgd_dcn_col2im_packed(hip.ptr(x), hip.ptr(offset), hip.ptr(dcol), N, C, H, W, stride, padding, dilation,
                                                    hip.ptr(dx), hip.ptr(doff), hip.ptr(ws) if ws is not None else None,
                                                    hip.stream_ptr()), "lgd_dcn_col2im_packed")
            else:
                hip.check(lib.lgd_dcn_col2im(hip.ptr(x), hip.ptr(offset), hip.ptr(mask) if mask is not None else None, hip.ptr(dcol),
                                             N, C, H, W, stride, padding, dilation, hip.ptr(dx), hip.ptr(doff),
                                             hip.ptr(dmask) if dmask is not None else None, hip.ptr(ws) if ws is not None else None,
                                             hip.stream_ptr()), "lgd_dcn_col2im")
        if ctx.needs_input_grad[3]:
            with streams.library_call(x.device):
                dw = torch.bmm(dy, col.transpose(1, 2)).sum(0).view_as(weight)
        if has_bias and ctx.needs_input_grad[4]:
            db = dy.sum((0, 2))
        return dx, doff, dmask, dw, db, None, None, None, None


_DCN_GATHER = True
_DCN_ONE_FILL = os.environ.get("LGD_DCN_ONE_FILL", "1") != "0"


def deform_conv3x3(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1):
    """DCNv2 (mask given) / DCNv1 (mask None) 3x3 convolution on the GPU."""
    return _DeformConv.apply(x, offset, mask, weight, bias, int(stride), int(padding), int(dilation))


def deform_conv3x3_packed(x, om, weight, bias=None, stride=1, padding=1, dilation=1):
    """DCNv2 driven directly by the offset convolution's (N, 27, Ho, Wo) output: = deform_conv3x3(x, om[:, :18], sigmoid(om[:, 18:]), ...)
    (chunk(3) -> cat(o1, o2), sigmoid(m) of detectron2's DeformBottleneckBlock) without the cat / sigmoid passes and their backward."""
    return _DeformConv.apply(x, om, None, weight, bias, int(stride), int(padding), int(dilation), True)


# ------------------------------------------------------------------------------------------------ student conv epilogues
def _relu_bits(lib, numel, device):
    """workspace of the 1-bit ReLU mask bias_act writes (element e = bit e % 32 of word e // 32)."""
    return torch.empty(int(lib.lgd_relu_bits_words(int(numel))), dtype=torch.int32, device=device)


class _BiasAct(torch.autograd.Function):
    """relu(x + bias[c] (+ residual)) in one pass; backward = the ReLU mask on the incoming gradient (shared by x and the
    residual) from the 1-bit-per-element bitmap the forward wrote (2 map transfers instead of 3, and the output is not kept
    alive for the mask), bias gradient only if the bias is a parameter (FrozenBN shifts are buffers)."""

    @staticmethod
    def forward(ctx, x, bias, residual, relu):
        hip.require_gpu(x)
        lib = hip.load()
        x = hip.dense_f32(x)
        bias = hip.dense_f32(bias) if bias is not None else None
        residual = hip.dense_f32(residual) if residual is not None else None
        N, C = x.shape[0], x.shape[1]
        HW = x.numel() // (N * C)
        out = torch.empty_like(x)
        bits = _relu_bits(lib, x.numel(), x.device) if relu and any(ctx.needs_input_grad) else None
        hip.check(lib.lgd_bias_act_fwd(hip.ptr(x), hip.ptr(bias) if bias is not None else None,
                                       hip.ptr(residual) if residual is not None else None, N, C, HW, int(relu),
                                       hip.ptr(out), hip.ptr(bits) if bits is not None else None, hip.stream_ptr()), "lgd_bias_act_fwd")
        ctx.relu = bool(relu)
        if relu:
            ctx.save_for_backward(bits)
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = hip.dense_f32(dy)
        if ctx.relu:
            (bits,) = ctx.saved_tensors
            dz = torch.empty_like(dy)
            hip.check(hip.load().lgd_relu_bits_bwd(hip.ptr(bits), hip.ptr(dy), dy.numel(), hip.ptr(dz), hip.stream_ptr()),
                      "lgd_relu_bits_bwd")
        else:
            dz = dy
        db = dz.sum((0, 2, 3)) if ctx.needs_input_grad[1] else None
        return dz, db, (dz if ctx.needs_input_grad[2] else None), None


def bias_act(x, bias=None, residual=None, relu=True):
    """relu(x + bias.view(1,-1,1,1) + residual) for NCHW fp32 maps in one pass."""
    return _BiasAct.apply(x, bias, residual, bool(relu))


def _valid_tag(t):
    tag = getattr(t, "_lgd_amax", None) if _H2_TAGS else None
    return tag[0] if tag is not None and tag[1] == t._version else None


def _pointwise_dw(dz, x, scale=None):
    """dW (Co, Ci, 1, 1) = scale[o] * sum_n dz[n] (Co x HW) @ x[n]^T (HW x Ci).  Where both maps carry their producers' maxima: lgd_h2_pwdw (both
    operands split into f16 pairs in registers, split-K over (image, pixel) ranges) -- otherwise per-image NT GEMMs of the library on the NCHW maps
    (one batched launch); the partials are added, with the frozen per-row scale, by one small kernel either way."""
    N, Co, Ci = dz.shape[0], dz.shape[1], x.shape[1]
    HW = dz.shape[2] * dz.shape[3]
    lib = hip.load()
    ta, tb = (_valid_tag(dz), _valid_tag(x)) if (_GEMM2H_ON and _H2_ON) else (None, None)
    if ta is not None and tb is not None and HW % 4 == 0 and Co >= 64 and Ci >= 64 and dz.is_contiguous() and x.is_contiguous():
        S = lib.lgd_h2_pwdw_splits(N, Co, Ci, HW)
        part = torch.empty((S, Co, Ci), dtype=torch.float32, device=dz.device)
        fn = lambda: hip.check(lib.lgd_h2_pwdw(hip.ptr(dz), hip.ptr(x), hip.ptr(ta), hip.ptr(tb), hip.ptr(part), S, N, Co, Ci, HW, hip.stream_ptr()), "lgd_h2_pwdw")   # noqa: E731
        if _TIMER_ON:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            _GEMM_EVENTS.append(("pw_gemm2h_dw", e0, e1))
            _GEMM_FLOPS["pw_gemm2h_dw"] = _GEMM_FLOPS.get("pw_gemm2h_dw", 0) + _pw_flops(x, Co)
        else:
            fn()
        nparts = S
    else:
        part = _timed_gemm("pw_gemm_dw", _pw_flops(x, Co), torch.bmm, dz.view(N, Co, -1), x.view(N, Ci, -1).transpose(1, 2))
        nparts = N
    dw = torch.empty((Co, Ci, 1, 1), dtype=torch.float32, device=dz.device)
    hip.check(lib.lgd_sum_batch_scale(hip.ptr(part), hip.ptr(scale) if scale is not None else None, nparts, Co, Ci, hip.ptr(dw),
                                      hip.stream_ptr()), "lgd_sum_batch_scale")
    return dw


class _Subsample2(torch.autograd.Function):
    """x[:, :, ::2, ::2] as a contiguous map, one pass forward and one backward (torch: a strided copy, and two zero fills + two strided
    copies over the full-resolution gradient)."""

    @staticmethod
    def forward(ctx, x):
        hip.require_gpu(x)
        x = hip.dense_f32(x)
        N, C, H, W = x.shape
        ctx.dims = (N, C, H, W)
        y = torch.empty((N, C, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=x.device)
        hip.check(hip.load().lgd_subsample2_fwd(hip.ptr(x), N * C, H, W, hip.ptr(y), hip.stream_ptr()), "lgd_subsample2_fwd")
        return y

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.dims
        g = hip.dense_f32(g)
        dx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        hip.check(hip.load().lgd_subsample2_bwd(hip.ptr(g), N * C, H, W, hip.ptr(dx), hip.stream_ptr()), "lgd_subsample2_bwd")
        return dx


def subsample2(x):
    """every other pixel of every other row of an NCHW map [d2-memory: the stride of a 1x1 / stride 2 convolution, applied ahead of it]"""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and os.environ.get("LGD_SUBSAMPLE2", "1") != "0":
        y = _Subsample2.apply(x)
    else:
        y = x[:, :, ::2, ::2].contiguous()
    tag = getattr(x, "_lgd_amax", None)
    if tag is not None and tag[1] == x._version:   # (a subset of the pixels: the bound of the whole map holds)
        _amax_tag([y], tag[0])
    return y


def stem_bias_relu_maxpool(y, bias):
    """max_pool2d(relu(y + bias[c]), 3, 2, 1) of the frozen stem convolution's output in one pass (no autograd: the stem is frozen)."""
    hip.require_gpu(y, bias)
    y, bias = hip.dense_f32(y.detach()), hip.dense_f32(bias.detach())
    N, C, H, W = y.shape
    out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=y.device)
    hip.check(hip.load().lgd_stem_bias_relu_maxpool(hip.ptr(y), hip.ptr(bias), N, C, H, W, hip.ptr(out), hip.stream_ptr()),
              "lgd_stem_bias_relu_maxpool")
    return out


_STEM7_ON = os.environ.get("LGD_STEM7", "1") != "0"   # 0: the library's convolution + the pooling pass (A/B runs)


def stem_conv_pool_ok(x, wf):
    """whether csrc/stem.hip takes the stem: the frozen 7x7 / stride 2 / padding 3 convolution of 3 -> 64 channels on an fp32 CUDA batch"""
    return (_STEM7_ON and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and tuple(wf.shape) == (64, 3, 7, 7)
            and wf.dtype == torch.float32 and x.shape[2] * x.shape[3] <= 1 << 27)


def stem_conv_pool(x, wf, shift):
    """max_pool2d(relu(conv2d(x, wf, stride 2, padding 3) + shift[c]), 3, 2, 1) in one kernel (lgd_stem7_conv_pool; no autograd: the stem is frozen).
    The filter's f16x2 image is made once per filter tensor and version."""
    hip.require_gpu(x, wf, shift)
    lib = hip.load()
    x, shift = hip.dense_f32(x.detach()), hip.dense_f32(shift.detach())
    st = hip.stream_ptr()
    kept = getattr(wf, "_lgd_stem7", None)
    if kept is None or kept[2] != wf._version:
        wc = hip.dense_f32(wf.detach())
        wa = torch.linalg.vector_norm(wc, float("inf")).reshape(1).view(torch.int32)
        img = torch.empty(lib.lgd_stem7_image_bytes(), dtype=torch.uint8, device=x.device)
        winv = torch.empty(1, dtype=torch.float32, device=x.device)
        hip.check(lib.lgd_stem7_image(hip.ptr(wc), hip.ptr(wa), hip.ptr(img), hip.ptr(winv), st), "lgd_stem7_image")
        kept = (img, winv, wf._version)
        try:
            wf._lgd_stem7 = kept
        except AttributeError:
            pass
    N, _, H, W = x.shape
    xa = _zero_words(x.device)
    hip.check(lib.lgd_h2_amax_maps(hip.ptr_array([x]), hip.int_array([H, W]), 1, N, 3, None, None, hip.ptr(xa), 1, st), "lgd_h2_amax_maps")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((N, 64, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    tag = _zero_words(x.device) if (_GEMM2H_ON and _H2_TAGS) else None
    hip.check(lib.lgd_stem7_conv_pool(hip.ptr(x), hip.ptr(kept[0]), hip.ptr(kept[1]), hip.ptr(xa), hip.ptr(shift), N, H, W, hip.ptr(out),
                                      hip.ptr(tag) if tag is not None else None, st), "lgd_stem7_conv_pool")
    if tag is not None:
        _amax_tag([out], tag)
    return out


class _PointwiseConvBN(torch.autograd.Function):
    """1x1 / stride 1 convolution with a folded frozen per-channel affine, as ONE autograd node:
        out = relu?( conv(x, w * scale[:, None, None, None]) + shift[c] (+ residual) )
    [d2-memory: conv -> FrozenBN (-> += shortcut) -> relu of the bottleneck blocks, SURVEY.md appendix A].  The filter fold, the
    library GEMM, the bias / residual / ReLU epilogue kernel (which also writes the 1-bit ReLU mask) and, backward, the mask kernel, the input-gradient GEMM and the
    weight gradient as per-image NT GEMMs on the NCHW maps run under a single node -- three autograd nodes and their host
    bookkeeping less per convolution than fold * conv1x1 * bias_act, which is what bounds the step at 2 images per GPU."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, residual, relu, wf=None):
        hip.require_gpu(x, w)
        lib = hip.load()
        x = hip.dense_f32(x)
        if wf is None:   # wf given: the folded filter of a FROZEN convolution, cached by the caller until the weight is written
            wf = w * scale.view(-1, 1, 1, 1)
        if shift is None and residual is None and not relu:   # raw output: the consumer folds the bias + ReLU into its own load
            ctx.relu = False
            ctx.save_for_backward(x, wf, scale, None)
            return _conv1x1_fwd(x, wf)
        residual = hip.dense_f32(residual) if residual is not None else None
        fused = _conv1x1_epilogue(ctx, x, wf, scale, shift, residual, relu)   # csrc/gemm3.hip with the epilogue inside the product kernel
        if fused is not None:
            return fused
        y = _conv1x1_fwd(x, wf)
        N, C = y.shape[0], y.shape[1]
        out = torch.empty_like(y)
        bits = _relu_bits(lib, y.numel(), y.device) if relu and any(ctx.needs_input_grad) else None
        hip.check(lib.lgd_bias_act_fwd(hip.ptr(y), hip.ptr(shift), hip.ptr(residual) if residual is not None else None, N, C,
                                       y.numel() // (N * C), int(relu), hip.ptr(out), hip.ptr(bits) if bits is not None else None,
                                       hip.stream_ptr()), "lgd_bias_act_fwd")
        ctx.relu = bool(relu)
        ctx.rowbits = False
        ctx.save_for_backward(x, wf, scale, bits)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, wf, scale, bits = ctx.saved_tensors
        dy = hip.dense_f32(dy)
        if ctx.relu:
            dz = _relu_bits_bwd(ctx, bits, dy)
            dz._lgd_exclusive = True   # fresh; after the GEMMs below its only reader is whoever receives the residual gradient
        else:
            dz = dy
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _conv1x1_dx(dz, x, wf)
        if ctx.needs_input_grad[1]:
            dw = _pointwise_dw(dz, x, scale)
        return dx, dw, None, None, (dz if ctx.needs_input_grad[4] else None), None, None


def _conv1x1_epilogue(ctx, x, wf, scale, shift, residual, relu):
    """relu?(conv1x1(x, wf) + shift[c] + residual) as ONE launch of csrc/gemm3.hip (accumulators initialised from residual + shift, ReLU
    and its row-padded 1-bit mask in the kernel's epilogue), where the shape gate lets the 128-row tile run; None otherwise (the caller
    runs the product and the bias_act pass).  Saves (x, wf, scale, bits) on ctx like the unfused path."""
    if not _GEMM3_EPILOGUE:
        return None
    N, Ci, H, W = x.shape
    Co = wf.shape[0]
    a, b = wf.view(1, Co, Ci).expand(N, Co, Ci), x.view(N, Ci, H * W)
    if not _gemm3_ok(a, b, None, accumulate=True):
        return None
    if residual is not None and tuple(residual.shape) != (N, Co, H, W):
        return None
    lib = hip.load()
    bits = None
    if relu and any(ctx.needs_input_grad):
        bits = torch.empty(int(lib.lgd_relu_rowbits_words(N * Co, H * W)), dtype=torch.int32, device=x.device)
    r, amax = _pw_product("pw_gemm3_fwd", a, b, x, None, residual=residual.view(N, Co, H * W) if residual is not None else None,
                          shift=hip.dense_f32(shift) if shift is not None else None, relu=bool(relu), relu_bits=bits)
    out = _TaggedView(r, amax).view(N, Co, H, W)
    ctx.relu = bool(relu)
    ctx.rowbits = True
    ctx.save_for_backward(x, wf, scale, bits)
    return out


def _relu_bits_bwd(ctx, bits, dy):
    """dy where the forward output was > 0, from the flat bitmap bias_act wrote or the row-padded one of the gemm3 epilogue"""
    dz = torch.empty_like(dy)
    tag = getattr(dy, "_lgd_amax", None)
    tagged = tag is not None and tag[1] == dy._version
    if getattr(ctx, "rowbits", False):
        # a gradient without a tag (a sum autograd built: the first block of a stage, the FPN laterals' sources): the mask kernel leaves max |dz|, so
        # that the products reading dz keep their f16x2 forms
        # (only where the input-gradient product that reads dz -- C -> C / 4, the block's last 1x1 convolution -- passes lgd_gemm2h's gate: at 2 images
        #  per GPU it does not, and the weight gradient alone on lgd_h2_pwdw measured slower than the library there: config 4 27.6 against 27.4 ms)
        word = None
        if not tagged and _GEMM2H_ON and _H2_TAGS and _RELU_AMAX and _gemm3_shape_ok(dy.shape[0], max(dy.shape[1] // 4, 1), dy.shape[1],
                                                                                    dy.shape[2] * dy.shape[3], dy.device, shared=True):
            word = _zero_words(dy.device)
        hip.check(hip.load().lgd_relu_rowbits_bwd(hip.ptr(bits), hip.ptr(dy), dy.shape[0] * dy.shape[1], dy.shape[2] * dy.shape[3], hip.ptr(dz),
                                                  hip.ptr(word) if word is not None else None, hip.stream_ptr()), "lgd_relu_rowbits_bwd")
        if word is not None:
            _amax_tag([dz], word)
    else:
        hip.check(hip.load().lgd_relu_bits_bwd(hip.ptr(bits), hip.ptr(dy), dy.numel(), hip.ptr(dz), hip.stream_ptr()), "lgd_relu_bits_bwd")
    if tagged:   # (a mask only removes elements: the bound of dy holds for dz)
        _amax_tag([dz], tag[0])
    return dz


class _PointwiseConvBNSkip(torch.autograd.Function):
    """conv1 of an identity-shortcut bottleneck block together with the shortcut itself:
        out, skip = relu(conv1x1(x, w * scale) + shift), x
    [d2-memory: BottleneckBlock.forward -- out = conv1(x) ...; out += shortcut (= x); SURVEY.md appendix A].  `skip` goes into conv3's
    residual add, so BOTH gradients of x arrive at this node, and the input-gradient GEMM accumulates onto the shortcut's gradient
    (beta = 1: dx = W^T dz + d_skip) instead of autograd adding the two maps in a separate pass over the block's input-size map
    (3 map transfers; measured per block at config 2, tools/skip_accum_probe.py: res3 350 -> 200 us, res4 255 -> 208 us)."""

    @staticmethod
    def forward(ctx, x, w, scale, shift, raw=False, wf=None):
        hip.require_gpu(x, w)
        lib = hip.load()
        x = hip.dense_f32(x)
        if wf is None:   # wf given: this step's fold from StepFolds.prepare() (one launch for all trainable 1x1 convolutions)
            wf = w * scale.view(-1, 1, 1, 1)
        ctx.raw = bool(raw)
        if raw:   # the 3x3 convolution that follows folds + shift and the ReLU into its input transform (and the mask into its adjoint)
            ctx.save_for_backward(x, wf, scale, None)
            return _conv1x1_fwd(x, wf), x.view_as(x)
        fused = _conv1x1_epilogue(ctx, x, wf, scale, shift, None, True)
        if fused is not None:
            return fused, x.view_as(x)
        y = _conv1x1_fwd(x, wf)
        N, C = y.shape[0], y.shape[1]
        out = torch.empty_like(y)
        bits = _relu_bits(lib, y.numel(), y.device) if any(ctx.needs_input_grad) else None
        hip.check(lib.lgd_bias_act_fwd(hip.ptr(y), hip.ptr(shift), None, N, C, y.numel() // (N * C), 1, hip.ptr(out),
                                       hip.ptr(bits) if bits is not None else None, hip.stream_ptr()), "lgd_bias_act_fwd")
        ctx.rowbits = False
        ctx.save_for_backward(x, wf, scale, bits)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        x, wf, scale, bits = ctx.saved_tensors
        N, Ci, Co = x.shape[0], x.shape[1], wf.shape[0]
        dx = dw = None
        dz = None
        if dy is not None and ctx.raw:
            dz = hip.dense_f32(dy)
        elif dy is not None:
            dz = _relu_bits_bwd(ctx, bits, hip.dense_f32(dy))
        if ctx.needs_input_grad[0]:
            if dz is None:
                dx = dskip
            elif dskip is None:
                dx = _conv1x1_dx(dz, x, wf)
            else:
                # W^T dz accumulated onto the shortcut's gradient inside the GEMM.  In place when the incoming tensor is the fresh,
                # otherwise unreferenced ReLU-masked gradient that conv3's node (_PointwiseConvBN.backward) produced for its
                # residual input and tagged as such; any other tensor (a sum autograd built, a caller's buffer) is left untouched:
                # torch.baddbmm out of place = a copy of it + the same GEMM (measured: the copy costs what the add pass did).
                acc = hip.dense_f32(dskip)
                own = getattr(dskip, "_lgd_exclusive", False) and acc is dskip
                a3 = acc.view(N, Ci, -1)
                wt, dz3 = wf.view(Co, Ci).t().unsqueeze(0).expand(N, Ci, Co), dz.view(N, Co, -1)
                if own and _gemm3_ok(wt, dz3, a3, accumulate=True):   # csrc/gemm3.hip with its accumulators initialised from the gradient
                    r, amax = _pw_product("pw_gemm3_dx", wt, dz3, dz, a3, accumulate=True)
                    dx = _TaggedView(r, amax).view(*x.shape)
                else:
                    dx = _timed_gemm("pw_gemm_dx", _pw_flops(x, Co), torch.baddbmm, a3, wt, dz3, **({"out": a3} if own else {})).view_as(x)
        if ctx.needs_input_grad[1] and dz is not None:
            dw = _pointwise_dw(dz, x, scale)
        return dx, dw, None, None, None, None


def pointwise_conv_bn_skip(x, w, scale, shift, raw=False, wf=None):
    """(relu(conv1x1(x, w * scale) + shift), x): conv1 + identity shortcut of a bottleneck block as one node (see above);
    raw: (conv1x1(x, w * scale), x) -- bias and ReLU are left to the consumer (conv3x3(..., pre=shift))."""
    return _PointwiseConvBNSkip.apply(x, w, scale, shift, bool(raw), wf)


def pointwise_conv_bn(x, w, scale, shift, residual=None, relu=True, wf=None):
    """relu?(conv1x1(x, w * scale) + shift (+ residual)); scale / shift are the frozen affine of the FrozenBN that follows;
    wf: the pre-folded filter w * scale of a frozen convolution (skips the fold launch)."""
    return _PointwiseConvBN.apply(x, w, scale, shift, residual, bool(relu), wf)


class _Conv1x1(torch.autograd.Function):
    """pointwise (1x1, stride 1) convolution without bias: forward and input gradient on the library's GEMM path, the
    weight gradient as per-image NT GEMMs on the NCHW maps, dW = sum_n dy[n] (Co x HW) @ x[n]^T (HW x Ci) -- the library's
    implicit-GEMM weight-gradient kernels transpose both operands to NHWC first."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return _conv1x1_fwd(x, w)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dw = None
        dy = dy.contiguous()
        if ctx.needs_input_grad[0]:
            dx = _conv1x1_dx(dy, x, w)
        if ctx.needs_input_grad[1]:
            dw = _pointwise_dw(dy, x)
        return dx, dw


def conv1x1(x, w):
    """bias-free pointwise convolution (see _Conv1x1); stride 1 only."""
    if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and torch.is_grad_enabled() and w.requires_grad:
        return _Conv1x1.apply(x, w)
    return lib_conv2d(x, w)


_TIMER_ON = False
_ALG_BYTES = {}


def _count_bytes(name, nbytes):
    """algorithmic HBM bytes of kernels whose launches differ in shape (the Winograd transforms), summed while the
    timer runs so that bench.py can price total bytes / total time."""
    if _TIMER_ON:
        _ALG_BYTES[name] = _ALG_BYTES.get(name, 0) + int(nbytes)


def kernel_timer_enable(on):
    """bracket every HIP kernel launch of the library with an event pair on its stream (bench.py)."""
    global _TIMER_ON
    hip.check(hip.load().lgd_timing_enable(int(bool(on))), "lgd_timing_enable")
    _TIMER_ON = bool(on)
    if on:
        _ALG_BYTES.clear()
        _GEMM_FLOPS.clear()
        _GEMM_EVENTS.clear()


def kernel_alg_bytes():
    """{kernel name: algorithmic bytes summed over the launches since kernel_timer_enable(True)} (shape-varying kernels)."""
    return dict(_ALG_BYTES)


def kernel_timer_collect():
    """{kernel name: (launches, total_ms, min_ms, max_ms)} since the last collect (synchronises the recorded events); the
    channel GEMMs of the Winograd convolutions (rocBLAS, issued and timed inside the library) appear as 'wino_gemm_fwd' /
    'wino_gemm_dx' / 'wino_gemm_dw'; the student's pointwise convolutions (torch.bmm, timed with events on torch's current stream =
    their launch stream) as 'pw_gemm_*'."""
    import ctypes
    lib = hip.load()
    names = ctypes.create_string_buffer(8192)
    ms, mn, mx = (ctypes.c_double * 128)(), (ctypes.c_double * 128)(), (ctypes.c_double * 128)()
    cnt = (ctypes.c_int32 * 128)()
    c = lambda a: ctypes.cast(a, ctypes.c_void_p)  # noqa: E731
    n = lib.lgd_timing_collect_ex(names, 8192, c(ms), c(mn), c(mx), c(cnt), 128)
    out, off = {}, 0
    raw = names.raw
    for i in range(n):
        end = raw.index(b"\0", off)
        out[raw[off:end].decode()] = (int(cnt[i]), float(ms[i]), float(mn[i]), float(mx[i]))
        off = end + 1
    if _GEMM_EVENTS:
        torch.cuda.synchronize()
        for name, a, b in _GEMM_EVENTS:
            t = a.elapsed_time(b)
            n0, s0, lo, hi = out.get(name, (0, 0.0, 1e300, 0.0))
            out[name] = (n0 + 1, s0 + t, min(lo, t), max(hi, t))
        _GEMM_EVENTS.clear()
    return out


_GEMM_EVENTS = []
_GEMM_FLOPS = {}


def _timed_bmm(name, a, b, out=None):
    """torch.bmm, bracketed by an event pair on the current stream while the kernel timer is on (bench.py's MFMA roofline)."""
    if not _TIMER_ON:
        with streams.library_call(a.device):   # (one rocBLAS handle and workspace per host thread: never two library GEMMs at once -- streams.py)
            return torch.bmm(a, b, out=out) if out is not None else torch.bmm(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with streams.library_call(a.device):
        e0.record()
        r = torch.bmm(a, b, out=out) if out is not None else torch.bmm(a, b)
        e1.record()
    _GEMM_EVENTS.append((name, e0, e1))
    _GEMM_FLOPS[name] = _GEMM_FLOPS.get(name, 0) + 2 * a.shape[0] * a.shape[1] * a.shape[2] * b.shape[2]
    return r


# ---- K9: the Winograd channel products on the bf16 MFMA pipe (csrc/gemm3.hip: three-way split fp32 operands, fp32 accumulate)
_GEMM3_ON = os.environ.get("LGD_GEMM3", "1") != "0"
_FILTER_IMAGES = os.environ.get("LGD_FILTER_IMAGES", "1") != "0"   # 0: fp32 U + the split pass of gemm3_bmm (A/B runs)
_GEMM3_ONE_ROUND = os.environ.get("LGD_GEMM3_ONE_ROUND", "1") != "0"   # 0: two rounds of workgroups for every shape (A/B runs)
_GEMM3_EPILOGUE = os.environ.get("LGD_GEMM3_EPILOGUE", "1") != "0"   # 0: product + a bias_act pass (A/B runs)


_GEMM3_FORCE = False   # tests: take csrc/gemm3.hip wherever the kernel CAN run (K % 16 == 0), whatever the speed policy says


def gemm3_backend(on=None, force=None):
    """whether the forward / input-gradient channel products of the Winograd convolutions (and the student's 1x1 convolutions) run on
    csrc/gemm3.hip (default) or on the library's fp32 GEMM (A/B runs, tests); force: bypass the speed policy so that small test problems
    take the kernel too.  Returns the previous (on, force)."""
    global _GEMM3_ON, _GEMM3_FORCE
    prev = (_GEMM3_ON, _GEMM3_FORCE)
    if on is not None:
        _GEMM3_ON = bool(on)
    if force is not None:
        _GEMM3_FORCE = bool(force)
    return prev


_CU_COUNT = {}


def _cu_count(device):
    i = device.index if device.index is not None else torch.cuda.current_device()
    if i not in _CU_COUNT:
        _CU_COUNT[i] = torch.cuda.get_device_properties(i).multi_processor_count
    return _CU_COUNT[i]


def _gemm3_shape_ok(nb, M, K, N, device, accumulate=False, shared=False, splitk_ok=False):
    """the speed policy of csrc/gemm3.hip on plain sizes (tile choice as in lgd_gemm3 / lgd_gemm2h); shared: one filter for the whole batch (the
    student's 1x1 convolutions, which take the f16x2 form and its 128-row tiles)"""
    if not _GEMM3_ON:
        return False
    if _GEMM3_FORCE:
        return K % 16 == 0
    # the kernel's tile spans 256 (or 128) rows of A: shapes that would leave more than ~30 % of the MFMA rows empty (C' = 36, 64, 320 ...) and
    # tiny problems stay on the library; K % 16: the k-step
    small = (shared and _GEMM2H_ON) or ((M + 255) // 256 * 256 - M >= 64 and (M + 127) // 128 * 128 - M < 64)
    bm = 128 if small else 256
    if K % 16 or K < 32 or N < 256 or M < 0.7 * bm * ((M + bm - 1) // bm):
        return False
    # at least one full round of workgroups (2 per CU with 256-row tiles, 3 with 128-row ones): below that a tile's prologue, its short
    # k-loop without a co-resident partner and the filter split are the whole launch and the library's smaller tiles win
    # (tools/gemm3_probe.py, profiles/r04_gemm3_probe.log: res5's 288 tiles x0.61, the 2048 -> 256 lateral x0.55; from one round up x1.05-1.6)
    wgs, cus = nb * ((N + 127) // 128) * ((M + bm - 1) // bm), _cu_count(device)
    if wgs >= cus * (3 if small else 2) * _PW_MIN_FILL:
        return True
    # one workgroup per CU is enough when the k-loop is long (64+ steps amortise a tile's prologue and epilogue): the 1024 -> 256
    # convolutions of res4 and the 1024-channel lateral at 8 images, x1.05-1.10 (profiles/r04_gemm3_probe_buffer_addressing.log)
    if _GEMM3_ONE_ROUND and wgs >= cus * _PW_ONE_ROUND_FILL and K >= 1024:
        return True
    # plain products of the f16x2 form below that: split-K brings them to two workgroups per CU (_splitk) -- the caller says whether it is one
    return bool(splitk_ok) and shared and _GEMM2H_ON and not accumulate and K >= 1024 and wgs * _splitk(nb, M, K, N, device) >= cus


_SPLITK = os.environ.get("LGD_GEMM2H_SPLITK", "1") != "0"   # 0: no split-K (A/B runs)


def _splitk(nb, M, K, N, device):
    """number of K splits of a plain lgd_gemm2h product (1 = none): the 128-row tiles fill less than one workgroup per CU and the k-loop has at least
    16 steps per split -- a workgroup's time is its k-steps (a latency chain each), so S splits take 1 / S of it plus the partials' round trip"""
    if not (_SPLITK and _GEMM2H_ON) or (nb * M * N) % 4:
        return 1
    wgs, cus = nb * ((N + 127) // 128) * ((M + 127) // 128), _cu_count(device)
    if wgs >= cus:
        return 1
    return max(1, min(4, (K // 16) // 16, (2 * cus + wgs - 1) // wgs))


def _gemm3_ok(a, b, out, accumulate=False, plain=False):
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 3 and b.dim() == 3):
        return False
    if not _gemm3_shape_ok(a.shape[0], a.shape[1], a.shape[2], b.shape[2], a.device, accumulate, shared=a.stride(0) == 0, splitk_ok=plain and out is None):
        return False
    if b.stride(2) != 1 or 17 * b.stride(1) + b.shape[2] >= 1 << 30:   # the kernel's 32-bit byte offsets (lgd_gemm3 returns LGD_EINVAL beyond)
        return False
    return out is None or (out.stride(2) == 1 and out.dtype == torch.float32 and 0 <= 256 * out.stride(1) < 1 << 30)


def gemm3_bmm(a, b, out=None, accumulate=False, residual=None, shift=None, relu=False, relu_bits=None, amax_out=None):
    """out[i] = a[i] @ b[i] (accumulate: out[i] += ...) for fp32 (nb, M, K) x (nb, K, N): an fp32-class product (error vs fp64 as the
    library's fp32 GEMM) computed on v_mfma_f32_32x32x16_bf16 from three-way split operands.  a (the filter operand, any strides) is split
    ahead of the product into an MFMA-ordered image -- ONE image when a is the same matrix for every batch (stride 0: the student's 1x1
    convolutions); b and out have their last axis contiguous.  Epilogue inside the kernel: out = relu?(a @ b + residual + shift[:, None]),
    relu_bits (int32, lgd_relu_rowbits_words(nb * M, N)) receives the ReLU mask for lgd_relu_rowbits_bwd."""
    lib = hip.load()
    nb, M, K = a.shape
    N = b.shape[2]
    if out is None:
        if accumulate:
            raise hip.LgdHipError("accumulate needs the tensor to accumulate onto")
        out = torch.empty((nb, M, N), dtype=torch.float32, device=a.device)
    if accumulate:
        if residual is not None:
            raise hip.LgdHipError("accumulate and residual are the same slot of the epilogue")
        residual = out
    if residual is not None and (tuple(residual.shape) != (nb, M, N) or residual.stride(2) != 1 or residual.dtype != torch.float32):
        raise hip.LgdHipError("residual must be an fp32 (nb, M, N) map with its last axis contiguous")
    if shift is not None and (shift.numel() != M or not shift.is_contiguous() or shift.dtype != torch.float32):
        raise hip.LgdHipError("shift must be M contiguous fp32 values")
    shared = a.stride(0) == 0 and nb > 1
    ni = 1 if shared else nb
    img = torch.empty(lib.lgd_gemm3_image_bytes(ni, M, K), dtype=torch.uint8, device=a.device)
    st = hip.stream_ptr()
    hip.check(lib.lgd_gemm3_split(hip.ptr(a), a.stride(0), a.stride(1), a.stride(2), ni, M, K, hip.ptr(img), st), "lgd_gemm3_split")
    hip.check(lib.lgd_gemm3(hip.ptr(img), 1 if shared else 0, hip.ptr(b), b.stride(0), b.stride(1), hip.ptr(out), out.stride(0), out.stride(1),
                            hip.ptr(residual) if residual is not None else None, residual.stride(0) if residual is not None else 0,
                            residual.stride(1) if residual is not None else 0, hip.ptr(shift) if shift is not None else None, 1 if relu else 0,
                            hip.ptr(relu_bits) if relu_bits is not None else None, hip.ptr(amax_out) if amax_out is not None else None, nb, M, N, K, st),
              "lgd_gemm3")
    return out


_GEMM2H_ON = os.environ.get("LGD_GEMM2H", "1") != "0"
_PW_MIN_FILL = float(os.environ.get("LGD_PW_MIN_FILL", "1.0"))   # fraction of "two rounds of workgroups" a product must fill to leave the library (experiments)
_PW_ONE_ROUND_FILL = float(os.environ.get("LGD_PW_ONE_ROUND_FILL", "1.0"))   # workgroups per CU a long-K product must reach (experiments)
_RELU_AMAX = os.environ.get("LGD_RELU_AMAX", "1") != "0"   # 0: the ReLU-mask kernel leaves no maximum for untagged gradients (A/B runs)
_PW_TAGS_ALWAYS = os.environ.get("LGD_PW_TAGS_ALWAYS", "0") != "0"   # 1: every output transform leaves its maximum, whatever the map's size (experiments)


_SPLIT_CALLS = [0]   # filter images made in front of a product (tests)


def gemm2h_bmm(a, b, b_amax, out=None, accumulate=False, residual=None, shift=None, relu=False, relu_bits=None, amax_out=None):
    """gemm3_bmm in the f16x2 form (lgd_gemm2h: three MFMAs per k-step instead of six): a must be ONE matrix for the whole batch (stride 0: the
    student's 1x1 convolutions); b_amax: int32[1], float bits of a bound of max |b| (the tag its producer left, or _amax_bits)."""
    lib = hip.load()
    nb, M, K = a.shape
    N = b.shape[2]
    if out is None:
        if accumulate:
            raise hip.LgdHipError("accumulate needs the tensor to accumulate onto")
        out = torch.empty((nb, M, N), dtype=torch.float32, device=a.device)
    if accumulate:
        if residual is not None:
            raise hip.LgdHipError("accumulate and residual are the same slot of the epilogue")
        residual = out
    if residual is not None and (tuple(residual.shape) != (nb, M, N) or residual.stride(2) != 1 or residual.dtype != torch.float32):
        raise hip.LgdHipError("residual must be an fp32 (nb, M, N) map with its last axis contiguous")
    if shift is not None and (shift.numel() != M or not shift.is_contiguous() or shift.dtype != torch.float32):
        raise hip.LgdHipError("shift must be M contiguous fp32 values")
    if not (a.stride(0) == 0 or nb == 1):
        raise hip.LgdHipError("lgd_gemm2h takes one filter matrix for the whole batch")
    a0 = a[0]
    # max |w|: one small launch per filter and step -- W and its transposed view (forward and input gradient) are views of ONE tensor object, which
    # carries the word with the version it was taken at (keyed by the object, never by its address: the allocator hands the same address to next
    # step's filter)
    root = a0._base if a0._base is not None else a0
    cached = getattr(root, "_lgd_w_amax", None)
    table = getattr(root, "_lgd_w_amax_table", None)   # this step's FrozenBN folds (student.resnet.StepFolds): the fold launch left every maximum
    hit = table[0].get(a0.storage_offset()) if table is not None and table[1] == root._version else None
    if hit is not None and hit[1] == a0.numel():
        a_amax = hit[0]
    elif cached is not None and cached[1] == root._version and cached[2] == root.numel() and a0.numel() == root.numel():
        a_amax = cached[0]
    else:
        a_amax = torch.linalg.vector_norm(a0, float("inf")).reshape(1).view(torch.int32)
        if a0.numel() == root.numel():
            root._lgd_w_amax = (a_amax, root._version, root.numel())
    st = hip.stream_ptr()
    # the filter's image: this step's, where student.resnet.StepFolds split all of them in one launch; a frozen filter's from its first product;
    # otherwise made here
    images = getattr(root, "_lgd_w_img_table", None)
    key = (a0.storage_offset(), a0.stride(0) == 1 and M > 1)
    hit = images[0].get(key) if images is not None and images[1] == root._version else None
    kept = getattr(root, "_lgd_w_img", None) if hit is None else None
    if hit is not None and hit[2] == a0.numel() and a0.is_contiguous() != key[1]:
        img, a_inv = hit[0], hit[1]
    elif kept is not None and key in kept and kept[key][2] == root._version and kept[key][3] == tuple(a0.shape) + tuple(a0.stride()):
        img, a_inv = kept[key][0], kept[key][1]
    else:
        img = torch.empty(lib.lgd_gemm2h_image_bytes(1, M, K), dtype=torch.uint8, device=a.device)
        a_inv = torch.empty(1, dtype=torch.float32, device=a.device)
        hip.check(lib.lgd_gemm2h_split(hip.ptr(a0), 0, a0.stride(0), a0.stride(1), 1, M, K, hip.ptr(a_amax), hip.ptr(img), hip.ptr(a_inv), st), "lgd_gemm2h_split")
        _SPLIT_CALLS[0] += 1
        if not root.requires_grad and a0.numel() == root.numel():   # a frozen filter (res2, a frozen backbone): split once, until somebody writes it
            if kept is None:
                kept = root._lgd_w_img = {}
            kept[key] = (img, a_inv, root._version, tuple(a0.shape) + tuple(a0.stride()))
    # split-K where a plain product's tiles leave most of the chip idle behind a long k-loop (_splitk)
    S = _splitk(nb, M, K, N, a.device) if (residual is None and shift is None and not relu and relu_bits is None and out.is_contiguous()) else 1
    ws = torch.empty((S, nb, M, N), dtype=torch.float32, device=a.device) if S > 1 else None
    hip.check(lib.lgd_gemm2h(hip.ptr(img), 1, hip.ptr(a_inv), hip.ptr(b), hip.ptr(b_amax), b.stride(0), b.stride(1), hip.ptr(out), out.stride(0), out.stride(1),
                             hip.ptr(residual) if residual is not None else None, residual.stride(0) if residual is not None else 0,
                             residual.stride(1) if residual is not None else 0, hip.ptr(shift) if shift is not None else None, 1 if relu else 0,
                             hip.ptr(relu_bits) if relu_bits is not None else None, hip.ptr(amax_out) if amax_out is not None else None,
                             hip.ptr(ws) if ws is not None else None, S, nb, M, N, K, st), "lgd_gemm2h")
    return out


def gemm3_image_bmm(img, b, out):
    """out[i] = A[i] @ b[i] with A given as its gemm3 image (_FilterImage of an (nb, M, K) operand)"""
    nb, M, K = img.shape
    if b.shape[0] != nb or b.shape[1] != K or out.shape[1] != M or b.stride(2) != 1 or out.stride(2) != 1:
        raise hip.LgdHipError("gemm3 image %s does not match B %s / C %s" % (img.shape, tuple(b.shape), tuple(out.shape)))
    hip.check(hip.load().lgd_gemm3(hip.ptr(img.t), 0, hip.ptr(b), b.stride(0), b.stride(1), hip.ptr(out), out.stride(0), out.stride(1),
                                   None, 0, 0, None, 0, None, None, nb, M, b.shape[2], K, hip.stream_ptr()), "lgd_gemm3")
    return out


def _timed_gemm3(name, a, b, out=None, accumulate=False, **epi):
    fn = (lambda: gemm3_image_bmm(a, b, out)) if isinstance(a, _FilterImage) else (lambda: gemm3_bmm(a, b, out, accumulate, **epi))
    if not _TIMER_ON:
        return fn()
    nb_, M_, K_ = a.shape
    # algorithmic bytes of the product: B read once, C written once (a residual / accumulated map read as well), the filter image once per batch
    reads_c = accumulate or epi.get("residual") is not None
    _count_bytes("gemm3_kernel", 4 * nb_ * b.shape[2] * (K_ + M_ * (2 if reads_c else 1)) + 6 * M_ * K_ * (1 if not isinstance(a, _FilterImage) and a.stride(0) == 0 else nb_))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    _GEMM_EVENTS.append((name, e0, e1))
    _GEMM_FLOPS[name] = _GEMM_FLOPS.get(name, 0) + 2 * a.shape[0] * a.shape[1] * a.shape[2] * b.shape[2]
    return r


class _TaggedView:
    """the (N, C, H, W) view of a product that carries the product's magnitude tag"""

    def __init__(self, t, amax):
        self.t, self.amax = t, amax

    def view(self, *shape):
        v = self.t.view(*shape)
        if self.amax is not None:
            _amax_tag([v], self.amax)
        return v


def _pw_product(name, a, b, bmap, out=None, accumulate=False, **epi):
    """a product of a 1x1 convolution on csrc/gemm3.hip -- a (nb, M, K) the filter as a stride-0 batch, b (nb, K, HW) the view of the NCHW map
    `bmap` (its magnitude tag is looked up on it) -- in the f16x2 form (lgd_gemm2h) unless switched off, with the epilogue's max |C| recorded
    for the consumer.  Returns (out, amax word or None)."""
    amax = _zero_words(b.device) if (_tags_wanted(b.shape[0] * b.shape[2] // 36) or _GEMM2H_ON) else None   # (the next 1x1 product's f16x2 form needs it whatever the map's size)
    tag = getattr(bmap, "_lgd_amax", None) if (_GEMM2H_ON and _H2_TAGS) else None
    if tag is not None and tag[1] == bmap._version and (a.stride(0) == 0 or a.shape[0] == 1):
        # the f16x2 form needs B's bound: taken where the producing kernel left it (every product and output transform of this library does); a map
        # without one (a sum autograd built, the stem's output) runs the bf16x3 form, which needs none -- never a pass over the map for a 1x1 product
        b_amax = tag[0]
        fn = lambda: gemm2h_bmm(a, b, b_amax, out, accumulate, amax_out=amax, **epi)   # noqa: E731
        name = name.replace("_gemm3_", "_gemm2h_")
    else:
        if _H2_DEBUG:
            import traceback
            fr = [f for f in traceback.extract_stack()[:-1] if "ops.py" not in f.filename and "torch" not in f.filename][-2:]
            print("[1x1 product without a tag -> bf16x3] %s M=%d K=%d N=%d x %d  <- %s" % (name, a.shape[1], a.shape[2], b.shape[2], b.shape[0],
                  " / ".join("%s:%d %s" % (f.filename.split("/")[-1], f.lineno, f.name) for f in fr)))
        fn = lambda: gemm3_bmm(a, b, out, accumulate, amax_out=amax, **epi)   # noqa: E731
    if not _TIMER_ON:
        return fn(), amax
    nb_, M_, K_ = a.shape
    reads_c = accumulate or epi.get("residual") is not None
    _count_bytes("gemm2h_kernel" if "_gemm2h_" in name else "gemm3_kernel", 4 * nb_ * b.shape[2] * (K_ + M_ * (2 if reads_c else 1)) + 4 * M_ * K_)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    _GEMM_EVENTS.append((name, e0, e1))
    _GEMM_FLOPS[name] = _GEMM_FLOPS.get(name, 0) + 2 * nb_ * M_ * K_ * b.shape[2]
    return r, amax


def _tagged_gemm3(name, a, b, bmap):
    """_pw_product as a view factory: the (N, C, H, W) view of the result carries the epilogue's max |C| (the f16x2 scale of the convolution that
    consumes the map: conv1 -> conv2 of a bottleneck forward, conv3 -> conv2 backward)"""
    r, amax = _pw_product(name, a, b, bmap)
    return _TaggedView(r, amax)


def _wino_gemm(name, a, b, out=None):
    """one of the per-frequency channel products (forward M = U V, input gradient dV = U^T dM): csrc/gemm3.hip where its tile fits the
    shape, the library's fp32 GEMM otherwise.  Timed under `name` + '3' (its launches also appear as gemm3_kernel / gemm3_split_kernel)."""
    if isinstance(a, _FilterImage):   # decided (same gate) where the filter was transformed: its image came straight from the transform
        return _timed_gemm3(name.replace("wino_gemm_", "wino_gemm3_"), a, b, out)
    if not _gemm3_ok(a, b, out):
        return _timed_bmm(name, a, b, out)
    return _timed_gemm3(name.replace("wino_gemm_", "wino_gemm3_"), a, b, out)


# ---- K10: the Winograd channel products from f16x2 split operands (csrc/h2.hip): V / dM are WRITTEN split by the F(6x6) transforms, the
# filter comes as an f16x2 image, forward / input-gradient / weight-gradient products run on v_mfma_f32_32x32x16_f16
_H2_ON = os.environ.get("LGD_H2", "1") != "0"
_H2_FORCE = False    # tests: take the h2 path wherever the kernels CAN run, whatever the speed policy says
_H2_TAGS = os.environ.get("LGD_H2_TAGS", "1") != "0"
# fewest tiles (all levels, padded) of a convolution that takes the f16x2 pipeline: at 2 images per GPU (BASELINE configs 4 / 5) one pyramid is 1376
# tiles and its products are ~40 us launches -- the bound's bookkeeping costs what the faster products gain (config 4, same call, 30 steps: every
# convolution on h2 30.1 ms, from 1500 tiles 29.1, none 29.6 with the producers still leaving their maxima / 29.0 without)
_H2_MIN_T = int(os.environ.get("LGD_H2_MIN_T", "1500"))
_H2_DEBUG = os.environ.get("LGD_H2_DEBUG", "0") != "0"   # print every bound that takes its own pass over the maps, with the call site   # 0: every bound by its own pass over the maps (A/B runs)


def h2_backend(on=None, force=None):
    """whether the F(6x6,3x3) convolutions run their channel products on csrc/h2.hip (default) or on csrc/gemm3.hip / the library (A/B runs,
    tests); force: bypass the speed policy so that small test problems take the kernels too.  Returns the previous (on, force)."""
    global _H2_ON, _H2_FORCE
    prev = (_H2_ON, _H2_FORCE)
    if on is not None:
        _H2_ON = bool(on)
    if force is not None:
        _H2_FORCE = bool(force)
    return prev


def _h2_ok(tile, Ci, Cos, T, dev):
    """the f16x2 pipeline takes a convolution when the kernels can run it (F(6x6); channel counts multiples of 16: the image kernel's
    16 x 16 blocks and the products' 16-deep k-steps) and, unless forced, when the products fill the chip (the speed policy of gemm3)"""
    if not (_H2_ON and tile == 6 and dev.type == "cuda" and Ci % 16 == 0 and all(c % 16 == 0 for c in Cos)):
        return False
    if _H2_FORCE:
        return True
    Ct = sum(Cos)
    bm = 128 if ((Ct + 255) // 256 * 256 - Ct >= 64 and (Ct + 127) // 128 * 128 - Ct < 64) else 256
    if Ci < 32 or T < _H2_MIN_T or Ct < 0.7 * bm * ((Ct + bm - 1) // bm):
        return False
    wgs = 64 * ((T + 127) // 128) * ((Ct + bm - 1) // bm)
    return wgs >= _cu_count(dev)


_ZERO_POOL = {}


def _zero_words(dev, n=1):
    """n int32 words that are zero and that nobody has written: the running maxima of the bounds start from them.  Cut from a pool that is
    filled once per 4096 words -- a fill launch per bound was 142 launches and 0.7 ms per step at BASELINE config 2"""
    # (a pool per stream: the fill and the kernels that start from the words are ordered by the stream they were issued on)
    key = (dev.type, dev.index, torch._C._cuda_getCurrentRawStream(dev.index) if dev.type == "cuda" and dev.index is not None else 0)
    pool = _ZERO_POOL.get(key)
    if pool is None or pool[1] + n > pool[0].numel():
        pool = [torch.zeros(4096, dtype=torch.int32, device=dev), 0]
        _ZERO_POOL[key] = pool
    w = pool[0][pool[1]:pool[1] + n]
    pool[1] += n
    return w


def _tags_wanted(n_tiles, maps=None):
    """whether a kernel that writes maps of n_tiles 6x6 tiles should leave their maximum: only where a consumer could take the f16x2 pipeline (the
    one-pass head runs over two pyramids: twice the tiles of the maps it reads) -- or, for a single map of C channels (a bottleneck's 3x3
    convolution), where the 1x1 product that reads it (C -> 4 C) passes lgd_gemm2h's gate (res5 at 8 images: 280 tiles, but 1152 workgroups)"""
    if not (_H2_ON and _H2_TAGS):
        return False
    if _H2_FORCE or _PW_TAGS_ALWAYS or 2 * n_tiles >= _H2_MIN_T:
        return True
    if maps is not None and len(maps) == 1 and _GEMM2H_ON:
        m = maps[0]
        return _gemm3_shape_ok(m.shape[0], 4 * m.shape[1], m.shape[1], m.shape[2] * m.shape[3], m.device, shared=True)
    return False


def _amax_tag(maps, amax):
    """record, on tensors a kernel of this library has just written, the device word holding (the float bits of) a bound of their
    magnitude: the next convolution derives its f16 scale from it instead of passing over the maps once more.  The tensor's version
    is recorded with it: an in-place write invalidates the tag."""
    if _H2_TAGS:
        for m in maps:
            m._lgd_amax = (amax, m._version)


def _dense_tagged(t):
    """hip.dense_f32 that carries a valid magnitude tag over to the copy it may have to make (a contiguous copy of a strided view)"""
    d = hip.dense_f32(t)
    if d is not t:
        tag = getattr(t, "_lgd_amax", None)
        if tag is not None and tag[1] == t._version:
            _amax_tag([d], tag[0])
    return d


def _amax_bits(lib, xs, hw_levels, pre=None, affine=False):
    """int32[1] device tensor: float bits of a bound of max |act(x)| over the maps xs (all (N, C, H_l, W_l); act = identity, relu(x + pre[c]) or the
    (L*N, C, 2) scale / shift form).  Tags left by the producing kernels are used where every map has one; otherwise one pass over the maps."""
    dev = xs[0].device
    # (an affine pre-activation takes its own pass: max|x| max|scale| + max shift multiplies maxima of DIFFERENT channels -- a GroupNorm channel with a
    #  tiny variance has a huge scale and small values -- and a bound 2^10 too loose spends the f16 pair's precision window on nothing)
    tags = [getattr(x, "_lgd_amax", None) for x in xs] if (_H2_TAGS and not affine) else [None] * len(xs)
    tags = [t if t is not None and t[1] == x._version else None for t, x in zip(tags, xs)]
    uniq = []
    for t in tags:
        if t is not None and not any(t[0] is u for u in uniq):
            uniq.append(t[0])
    rest = [i for i, t in enumerate(tags) if t is None]
    if not rest and len(uniq) == 1 and pre is None:
        return uniq[0]
    out = _zero_words(dev)
    if uniq and len(uniq) <= 16:   # the tagged maps' bounds, folded into one word by one small launch (non-negative floats order like their bits)
        hip.check(lib.lgd_h2_words_max(hip.ptr_array(uniq), len(uniq), hip.ptr(out), hip.stream_ptr()), "lgd_h2_words_max")
    elif uniq:
        rest = list(range(len(xs)))
    if not rest:
        if pre is None:
            return out
        b = (out.view(torch.float32) + pre.max()).clamp_min(0.0)   # |relu(x + b[c])| <= max(0, max|x| + max b)
        return (b * 1.000001).view(torch.int32)   # (the roundings of the bound's own arithmetic)
    if pre is not None and len(rest) < len(xs):   # (a pre-activation's bound is taken over all maps or from tags alone)
        rest = list(range(len(xs)))
        out = _zero_words(dev)
    # the maps without a (valid) tag -- the small levels the library's convolutions produced, sums autograd built -- take one pass, added onto the
    # tagged ones' bound
    L, N, C = len(rest), xs[0].shape[0], xs[0].shape[1]
    if _H2_DEBUG:
        import traceback
        fr = [f for f in traceback.extract_stack()[:-1] if "ops.py" not in f.filename and "torch" not in f.filename][-2:]
        node = torch._C._current_autograd_node()
        print("[h2 amax pass] L=%d of %d N=%d C=%d hw=%s pre=%s <- %s%s" % (L, len(xs), N, C, [tuple(xs[i].shape[2:]) for i in rest][:2], None if pre is None else ("affine" if affine else "bias"),
                                                                           " / ".join("%s:%d %s" % (f.filename.split("/")[-1], f.lineno, f.name) for f in fr),
                                                                           "" if node is None else " [backward of %s; maps from %s]" % (node.name(), ", ".join(sorted({type(getattr(xs[i], "grad_fn", None)).__name__ for i in rest})))))
    hw_rest = hw_levels if len(rest) == len(xs) else hip.int_array([v for i in rest for v in xs[i].shape[-2:]])
    hip.check(lib.lgd_h2_amax_maps(hip.ptr_array([xs[i] for i in rest]), hw_rest, L, N, C, hip.ptr(pre) if pre is not None and not affine else None,
                                   hip.ptr(pre) if pre is not None and affine else None, hip.ptr(out), 1, hip.stream_ptr()), "lgd_h2_amax_maps")
    _count_bytes("h2_amax_maps_kernel", 4 * sum(xs[i].numel() for i in rest))
    return out


def _amax_bits_groups(lib, groups, hw_levels):
    """one bound over several groups of level maps with different channel counts (the K gradients of stacked filters)"""
    if len(groups) == 1:
        return _amax_bits(lib, groups[0], hw_levels)
    parts = [_amax_bits(lib, g, hw_levels) for g in groups]
    out = _zero_words(parts[0].device)
    hip.check(lib.lgd_h2_words_max(hip.ptr_array(parts), len(parts), hip.ptr(out), hip.stream_ptr()), "lgd_h2_words_max")
    return out


class _H2Filter:
    """the filter operand of the f16x2 products: images of U (forward) and U^T (input gradient) and the 64 per-frequency inverse scales"""
    __slots__ = ("fwd", "bwd", "inv", "Ct", "Ci")

    def __init__(self, fwd, bwd, inv, Ct, Ci):
        self.fwd, self.bwd, self.inv, self.Ct, self.Ci = fwd, bwd, inv, Ct, Ci


def _h2_filters(lib, ws, scales, Ci, dev, need_dx):
    """f16x2 images of the transformed filters of K convolutions stacked along C_out (lgd_wino_filter_images_h2): one scale per frequency for the
    whole stack, from max |w * scale| over all K filters"""
    Cos = [w.shape[0] for w in ws]
    Ct = sum(Cos)
    st = hip.stream_ptr()
    amax = _zero_words(dev)
    sc_arr = (ctypes.c_void_p * len(ws))(*[sc.data_ptr() if sc is not None else None for sc in scales])
    hip.check(lib.lgd_h2_amax_filters(hip.ptr_array(ws), sc_arr, hip.int_array(Cos), len(ws), Ci * 9, hip.ptr(amax), st), "lgd_h2_amax_filters")
    imf = torch.empty(lib.lgd_h2_image_bytes(64, Ct, Ci), dtype=torch.uint8, device=dev)
    imb = torch.empty(lib.lgd_h2_image_bytes(64, Ci, Ct), dtype=torch.uint8, device=dev) if need_dx else None
    inv = torch.empty(64, dtype=torch.float32, device=dev)
    c0 = 0
    for w, sc, Co in zip(ws, scales, Cos):
        hip.check(lib.lgd_wino_filter_images_h2(hip.ptr(w), hip.ptr(sc) if sc is not None else None, Co, Ci, c0, Ct, hip.ptr(imf),
                                                hip.ptr(imb) if imb is not None else None, hip.ptr(amax), hip.ptr(inv), st), "lgd_wino_filter_images_h2")
        c0 += Co
    return _H2Filter(imf, imb, inv, Ct, Ci)


def _h2_buf(C, T, dev):
    """a split frequency buffer [C][64][T], 4 bytes per element (int32 storage)"""
    return torch.empty((C, 64, T), dtype=torch.int32, device=dev)


def _h2_timed(name, flops, nbytes, fn):
    if not _TIMER_ON:
        return fn()
    _count_bytes(name, nbytes)
    _GEMM_FLOPS[name] = _GEMM_FLOPS.get(name, 0) + int(flops)
    return fn()


def _h2_product(lib, name, img, M, K, B, b_inv, b_inv_per_batch, a_inv, out, amax_out=None):
    """out (64, M, T view of [M][64][T]) = image (64, M, K) . B ([K][64][T] split rows)"""
    T = B.shape[2]
    fn = lambda: hip.check(lib.lgd_h2_fwd(hip.ptr(img), hip.ptr(B), 4 * T, 4 * 64 * T, 4 * B.numel(), hip.ptr(out), out.stride(0), out.stride(1),   # noqa: E731
                                          hip.ptr(a_inv), hip.ptr(b_inv), 1 if b_inv_per_batch else 0, hip.ptr(amax_out) if amax_out is not None else None,
                                          64, M, T, K, hip.stream_ptr()), "lgd_h2_fwd")
    _h2_timed("h2_fwd_kernel", 2.0 * 64 * M * K * T, 4.0 * 64 * T * (K + M) + 4.0 * 64 * M * K, fn)
    return out


def _h2_dw(lib, dM, dm_inv, V, v_inv, Ct, Ci):
    """dU (64, Ct, Ci) = dM . V^T over the tiles, both split rows"""
    T = V.shape[2]
    dev = V.device
    S = lib.lgd_h2_dw_splits(64, Ct, Ci, T)
    out = torch.empty((64, Ct, Ci), dtype=torch.float32, device=dev)
    # (the split-K partials are added by lgd_h2_dw's own 12 us reduce launch: reading them inside the filter transform's adjoint instead
    #  -- lgd_wino_filter_bwd_parts, one launch less -- measured 37 us against 10 + 12: its 256 workgroups walk S x 64 strided planes each)
    part = torch.empty((S, 64, Ct, Ci), dtype=torch.float32, device=dev) if S > 1 else None
    fn = lambda: hip.check(lib.lgd_h2_dw(hip.ptr(dM), 4 * 64 * T, 4 * T, 4 * dM.numel(), hip.ptr(dm_inv), 1, hip.ptr(V), 4 * 64 * T, 4 * T, 4 * V.numel(),   # noqa: E731
                                         hip.ptr(v_inv), 0, hip.ptr(out), hip.ptr(part) if part is not None else None, S, 64, Ct, Ci, T, hip.stream_ptr()),
                           "lgd_h2_dw")
    _h2_timed("h2_dw_kernel", 2.0 * 64 * Ct * Ci * T, 4.0 * 64 * T * (Ct + Ci) + 4.0 * 64 * Ct * Ci * (2 * S if S > 1 else 1), fn)
    return out


def _h2_plane_sums(buf, f, inv):
    """sum over the tiles of frequency plane f of a split buffer [C][64][T], per channel (the bias gradient: the frequency of the interpolation
    point 1 of dM = A g A^T is the tile's gradient sum)"""
    C, _, T = buf.shape
    out = torch.empty(C, dtype=torch.float32, device=buf.device)
    hip.check(hip.load().lgd_h2_plane_sums(ctypes.c_void_p(buf.data_ptr() + 4 * T * f), 4 * 64 * T, C, T, ctypes.c_void_p(inv.data_ptr() + 4 * f), hip.ptr(out),
                                           hip.stream_ptr()),
              "lgd_h2_plane_sums")
    return out


def _timed_gemm(name, flops, fn, *args, **kw):
    """fn(*args, **kw) -- a library GEMM (or a convolution the library runs as one) -- bracketed by an event pair on the current
    stream while the kernel timer is on: the student's pointwise convolutions in bench.py's second MFMA roofline object."""
    if not _TIMER_ON:
        with streams.library_call(args[0].device):
            return fn(*args, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with streams.library_call(args[0].device):
        e0.record()
        r = fn(*args, **kw)
        e1.record()
    _GEMM_EVENTS.append((name, e0, e1))
    _GEMM_FLOPS[name] = _GEMM_FLOPS.get(name, 0) + int(flops)
    return r


def _pw_flops(x, co):
    """2 * N * HW * Ci * Co of a pointwise convolution on the map x"""
    return 2 * x.shape[0] * x.shape[1] * co * x.shape[2] * x.shape[3]


def _conv1x1_fwd(x, wf):
    """pointwise convolution of a contiguous NCHW map as per-image GEMMs [Co x Ci] . [Ci x HW] with the filter as a stride-0 batch: the
    library's strided-batched GEMM with the solution the tuning table holds for the shape (torch.bmm -> TunableOp).  F.conv2d took
    MIOpen's own 1x1 kernels: 100 TFLOP/s at config 2 against 125-140 for the table's GEMMs."""
    N, Ci, H, W = x.shape
    Co = wf.shape[0]
    a, b = wf.view(1, Co, Ci).expand(N, Co, Ci), x.view(N, Ci, H * W)
    if _gemm3_ok(a, b, None, plain=_valid_tag(x) is not None):   # csrc/gemm3.hip: one image of the filter for the whole batch (split-K only in the f16x2 form: the map must carry its bound)
        return _tagged_gemm3("pw_gemm3_fwd", a, b, x).view(N, Co, H, W)
    return _timed_gemm("pw_gemm_fwd", _pw_flops(x, Co), torch.bmm, a, b).view(N, Co, H, W)


def _conv1x1_dx(dz, x, wf):
    """input gradient of the pointwise convolution: W^T [Ci x Co] . dz [Co x HW] per image (same form as the forward)"""
    N, Ci, H, W = x.shape
    Co = wf.shape[0]
    a, b = wf.view(1, Co, Ci).transpose(1, 2).expand(N, Ci, Co), dz.view(N, Co, H * W)
    if _gemm3_ok(a, b, None, plain=_valid_tag(dz) is not None):
        return _tagged_gemm3("pw_gemm3_dx", a, b, dz).view(N, Ci, H, W)
    return _timed_gemm("pw_gemm_dx", _pw_flops(x, Co), torch.bmm, a, b).view(N, Ci, H, W)


def kernel_gemm_flops():
    """{gemm name: floating-point operations summed over the launches since kernel_timer_enable(True)}."""
    return dict(_GEMM_FLOPS)
