"""ctypes binding of the C-ABI in include/lgd_hip.h (lgd_amd/_lib/liblgd_hip.so).

There is NO CPU fallback: importing succeeds without the library (so CPU-only host
logic stays testable) but any kernel call raises LgdHipError when the library or a
GPU is missing -- the product path must fail loudly rather than silently compute
elsewhere.
"""
import ctypes
import os

import torch

_LIB_PATH = os.environ.get("LGD_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "liblgd_hip.so")   # LGD_HIP_LIB: a lab build
_lib = None

ABI_VERSION = 26

c_fp = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/lgd_hip.h one-to-one (checked by tests/test_abi.py)
SIGNATURES = {
    "lgd_abi_version": (c_i, []),
    "lgd_arch": (ctypes.c_char_p, []),
    "lgd_last_error": (ctypes.c_char_p, []),
    "lgd_geom_ints": (c_sz, [c_i, c_i, c_i, c_i]),
    "lgd_geom_rects_off": (c_sz, [c_i, c_i, c_i, c_i]),
    "lgd_geom_nbp_off": (c_sz, [c_i, c_i, c_i, c_i]),
    "lgd_geom_bands_off": (c_sz, [c_i, c_i, c_i, c_i]),
    "lgd_box_prep": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_i, c_fp, c_fp]),
    "lgd_box_pool_ws_floats": (c_sz, [c_fp, c_i, c_i, c_i, c_i, c_i]),
    "lgd_box_sum": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    "lgd_box_paint": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    "lgd_distill_ws_doubles": (c_sz, [c_fp, c_i, c_i, c_i]),
    "lgd_distill_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_fp]),
    "lgd_distill_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn1_ws_doubles": (c_sz, [c_fp, c_i, c_i, c_i]),
    "lgd_gn1_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn1_stats": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_gn_pool_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn_pool_bwd": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn1_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn_group_ws_doubles": (c_sz, [c_fp, c_i, c_i, c_i]),
    "lgd_gn_group_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn_group_stats_affine": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn_group_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gn_group_bwd_coef": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_fcos_targets": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_ctx_relu_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_ctx_relu_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_gemm_batch": (c_i, [c_fp, c_i, c_fp]),
    "lgd_attn_fwd": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_attn_bwd": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_box_descriptors": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_rowln_fwd": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_rowln_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_rowvecmat_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
    "lgd_rowvecmat_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_segmax_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_segmax_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
    "lgd_focal_ws_doubles": (c_sz, [c_fp, c_i, c_i, c_i, c_i]),
    "lgd_focal_loss_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_f, c_fp, c_fp, c_fp]),
    "lgd_focal_loss_fwd_grad": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_f, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_scale_unless_one": (c_i, [c_fp, c_fp, c_i, c_fp, c_fp, c_fp]),
    "lgd_fcos_loss_ws_doubles": (c_sz, [c_fp, c_i, c_i]),
    "lgd_fcos_loss_fwd_grad": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_focal_loss_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_f, c_fp, c_fp, c_fp]),
    "lgd_wino_tiles": (c_sz, [c_fp, c_i, c_i, c_i]),
    "lgd_wino_mask_bytes": (c_sz, [c_i]),
    "lgd_wino_in": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_out": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_wino_out_t": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_wino_in_t": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_wino_out_t_gn": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_wino_filter_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, ctypes.c_longlong, c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp]),
    "lgd_wino_filter_images": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_wino_filter_bwd": (c_i, [c_fp, ctypes.c_longlong, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_wino_filter_bwd_parts": (c_i, [c_fp, ctypes.c_longlong, ctypes.c_longlong, c_i, c_fp, c_i, c_i, c_fp, c_fp]),
    "lgd_wino_in_t_out_t": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_gemm3_image_bytes": (c_sz, [c_i, c_i, c_i]),
    "lgd_gemm3_split": (c_i, [c_fp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_gemm3": (c_i, [c_fp, c_i, c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp, ctypes.c_longlong, ctypes.c_longlong,
                        c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "lgd_h2_image_bytes": (c_sz, [c_i, c_i, c_i]),
    "lgd_h2_fwd": (c_i, [c_fp, c_fp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "lgd_h2_dw_splits": (c_i, [c_i, c_i, c_i, c_i]),
    "lgd_h2_dw": (c_i, [c_fp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_fp, c_i, c_fp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_fp, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "lgd_h2_pwdw_splits": (c_i, [c_i, c_i, c_i, c_i]),
    "lgd_h2_pwdw": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "lgd_h2_amax_maps": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_fp]),
    "lgd_h2_amax_filters": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
    "lgd_h2_link_bound": (c_i, [c_fp, c_fp, c_fp]),
    "lgd_h2_words_max": (c_i, [c_fp, c_i, c_fp, c_fp]),
    "lgd_h2_plane_sums": (c_i, [c_fp, ctypes.c_longlong, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_wino_out_t_gn_h2": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_in_h2": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_out_t_h2": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_in_t_out_t_h2": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_out_amax": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_in_t_amax": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_wino_filter_images_h2": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gemm2h_image_bytes": (c_sz, [c_i, c_i, c_i]),
    "lgd_gemm2h_split_multi": (c_i, [c_fp, c_fp, c_i, c_i, c_fp]),
    "lgd_gemm2h_split": (c_i, [c_fp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_gemm2h": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp, ctypes.c_longlong, ctypes.c_longlong, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "lgd_relu_rowbits_words": (c_sz, [ctypes.c_longlong, c_i]),
    "lgd_relu_rowbits_bwd": (c_i, [c_fp, c_fp, ctypes.c_longlong, c_i, c_fp, c_fp, c_fp]),
    "lgd_relu_bits_words": (c_sz, [ctypes.c_longlong]),
    "lgd_bias_act_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_stem_bias_relu_maxpool": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_stem7_image_bytes": (c_sz, []),
    "lgd_stem7_image": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_stem7_conv_pool": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "lgd_sum_batch_scale": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_scale_rows_multi": (c_i, [c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
    "lgd_relu_bits_bwd": (c_i, [c_fp, c_fp, ctypes.c_longlong, c_fp, c_fp]),
    "lgd_relu_mask_bwd": (c_i, [c_fp, c_fp, ctypes.c_longlong, c_fp, c_fp]),
    "lgd_subsample2_fwd": (c_i, [c_fp, ctypes.c_longlong, c_i, c_i, c_fp, c_fp]),
    "lgd_subsample2_bwd": (c_i, [c_fp, ctypes.c_longlong, c_i, c_i, c_fp, c_fp]),
    "lgd_anchor_match": (c_i, [c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_f, c_f, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_box_reg_ws_doubles": (c_sz, [c_fp, c_i, c_i, c_i]),
    "lgd_box_reg_loss_fwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_fp]),
    "lgd_box_reg_loss_bwd": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_fp]),
    "lgd_dcn_im2col": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_dcn_im2col_packed": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "lgd_dcn_col2im_packed": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "lgd_dcn_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "lgd_dcn_col2im": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "lgd_sgd_chunk_elems": (c_i, []),
    "lgd_sgd_clip_step": (c_i, [c_fp, c_fp, c_i, c_i, c_f, c_fp]),
    "lgd_timing_enable": (c_i, [c_i]),
    "lgd_timing_collect": (c_i, [ctypes.c_char_p, c_sz, c_fp, c_fp, c_i]),
    "lgd_timing_collect_ex": (c_i, [ctypes.c_char_p, c_sz, c_fp, c_fp, c_fp, c_fp, c_i]),
    "lgd_timing_pending": (c_i, [ctypes.c_char_p, c_sz]),
}


class GemmProblem(ctypes.Structure):
    """mirror of `lgd_gemm_problem` (include/lgd_hip.h)."""
    _fields_ = [("A", c_fp), ("B", c_fp), ("bias", c_fp), ("C", c_fp), ("rowsum", c_fp),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("reserved0", ctypes.c_int32),
                ("sa_m", ctypes.c_int64), ("sa_k", ctypes.c_int64), ("sb_n", ctypes.c_int64), ("sb_k", ctypes.c_int64),
                ("sc_m", ctypes.c_int64), ("sc_n", ctypes.c_int64), ("alpha", ctypes.c_float), ("reserved1", ctypes.c_int32)]


class LgdHipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """dlopen the kernel library (idempotent). Raises LgdHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LgdHipError("HIP kernel library missing: %s -- run `python -c 'import __graft_entry__ as g; g.build()'`"
                          % _LIB_PATH)
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.lgd_abi_version() != ABI_VERSION:
        raise LgdHipError("liblgd_hip.so ABI %d != binding ABI %d: rebuild" % (lib.lgd_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_HAS_GPU = None


def require_gpu(*tensors):
    global _HAS_GPU
    if _HAS_GPU is None:   # asked once: torch.cuda.is_available() costs microseconds per call and every op passes through here
        _HAS_GPU = torch.cuda.is_available()
    if not _HAS_GPU:
        raise LgdHipError("LGD HIP kernels need a ROCm GPU (gfx950); no device is visible and there is no CPU fallback")
    for t in tensors:
        if not t.is_cuda:
            raise LgdHipError("expected a device tensor, got %s" % t.device)


def check(rc, what):
    if rc != 0:
        err = load().lgd_last_error().decode()
        raise LgdHipError("%s failed with code %d %s" % (what, rc, err))


def stream_ptr():
    """the current HIP stream of the current device as a void* (the raw getter: torch.cuda.current_stream() builds a Stream object
    and resolves the device index in Python, ~9 us per call x ~300 calls per step)."""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """host array of device pointers (const float* const*)."""
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class _PinnedRing:
    """Small host -> device uploads (per-image offsets, index lists: a few dozen bytes, several per step) through a ring of PINNED
    staging slots: `torch.tensor(list).to(device)` reads pageable memory, which the runtime copies synchronously with the host and
    only after the stream has drained -- every such call let the GPU run dry for 0.2-0.7 ms (tools/gap_profile.sh).  The ring is cut
    into SEGMENTS of slots; the last upload of a segment records an event on EVERY stream that issued a copy out of the segment (the step uploads
    from its main stream and from side streams alike: lgd_amd/streams.py) and the first upload of the ring's next pass over that segment waits
    for all of them -- a slot is never rewritten before the copy that read it has executed (in practice the events are hundreds of steps old and
    the wait is a flag test)."""
    SLOTS, SLOT_BYTES, SEGMENT = 1024, 1024, 128

    def __init__(self):
        self.buf = torch.empty(self.SLOTS * self.SLOT_BYTES, dtype=torch.uint8).pin_memory()
        self.i = 0
        self.events = [None] * (self.SLOTS // self.SEGMENT)   # per segment: list of events, one per stream that copied out of it
        self.used = {}    # raw stream handle -> torch stream object: the streams that copied out of the current segment
        self.waits = 0    # how many segment re-entries found events to wait for (tests)

    def upload(self, t, device):
        n = t.numel() * t.element_size()
        if n == 0 or n > self.SLOT_BYTES:
            return t.to(device, non_blocking=True)
        seg, first = divmod(self.i, self.SEGMENT)
        if first == 0:
            if self.events[seg]:
                for e in self.events[seg]:
                    e.synchronize()
                self.waits += 1
            self.used = {}
        o = self.i * self.SLOT_BYTES
        slot = self.buf[o:o + n].view(t.dtype).view(t.shape)
        slot.copy_(t)
        out = slot.to(device, non_blocking=True)
        raw = torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch._C._cuda_getDevice())
        if raw not in self.used:
            self.used[raw] = torch.cuda.current_stream(device)
        if first == self.SEGMENT - 1:
            evs = []
            for st in self.used.values():   # every stream that read a slot of this segment, not only the one of the last upload
                e = torch.cuda.Event()
                e.record(st)
                evs.append(e)
            self.events[seg] = evs
        self.i = (self.i + 1) % self.SLOTS
        return out


_ring = None


def to_device(values, dtype, device):
    """device tensor of a short Python list / nested list, uploaded without blocking the host or draining the stream."""
    global _ring
    t = torch.tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    if _ring is None:
        _ring = _PinnedRing()
    return _ring.upload(t, device)


def int_array(vals):
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


def dense_f32(t):
    """fp32, contiguous, 16-byte aligned view/copy of t (the kernels use dwordx4 accesses)."""
    if t.dtype != torch.float32:
        raise LgdHipError("LGD kernels compute in fp32, got %s" % t.dtype)
    t = t.contiguous()
    if t.data_ptr() % 16:
        t = t.clone()
    return t
