"""Gradient clipping + the two SGD updates of the distillation step as ONE HIP launch (csrc/optim.hip).

[ref: train.py:200-204 -- `stu_optimizer.step(); tea_optimizer.step()`; utils/build.py:494-529 -- torch.optim.SGD with momentum and
weight decay, wrapped by detectron2's per-parameter gradient clipping (CLIP_TYPE "value").]

The torch.optim.SGD objects stay the owners of the state (momentum buffers under the reference's keys, param groups, the LR
schedulers drive their `lr`), so checkpoints and `Trainer.state_dict()` are unchanged; only the arithmetic of `step()` moves into
`lgd_sgd_clip_step`.  torch's multi-tensor path costs ~1.7 ms of Python per step (grouping ~330-530 tensors for each of six foreach
ops and two optimizers) and ~40 launches over 13 tensor transfers per element; at 2 images per GPU the GPU ran dry for 0.7 ms at the
optimizer entry.  Here the host builds one table (three pointers, length, lr / wd / mu per tensor) in pinned memory and launches once.
"""
import ctypes

import numpy as np
import torch

from . import hip

_TENSOR_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4"), ("mu", "<f4"),
                       ("reserved", "<i4")])  # mirror of lgd_sgd_tensor (include/lgd_hip.h), 48 bytes
assert _TENSOR_DT.itemsize == 48


def supported(optimizers, clip):
    """the fused step covers what the reference's configs use: SGD (dampening 0, no Nesterov, not maximize) on fp32 CUDA parameters,
    gradient clipping off or CLIP_TYPE 'value'; anything else stays on torch's own multi-tensor path."""
    if clip.ENABLED and clip.CLIP_TYPE != "value":
        return False
    for o in optimizers:
        if type(o) is not torch.optim.SGD:
            return False
        for g in o.param_groups:
            if g.get("dampening", 0) != 0 or g.get("nesterov", False) or g.get("maximize", False):
                return False
            for p in g["params"]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    return False
    return True


class FusedClipSGD:
    """step() = clip + update of every parameter that has a gradient, over all the given torch.optim.SGD objects."""
    SLOTS = 8  # pinned staging tables in flight (a slot is reused SLOTS steps later, after its upload event has completed)

    def __init__(self, optimizers, clip_value=None):
        self.optimizers = list(optimizers)
        self.clip_value = float("inf") if clip_value is None else float(clip_value)
        self.params, self.owner = [], []
        for k, o in enumerate(self.optimizers):
            for gi, g in enumerate(o.param_groups):
                for p in g["params"]:
                    self.params.append(p)
                    self.owner.append((k, gi))
        n = len(self.params)
        self._numel = np.array([p.numel() for p in self.params], dtype=np.int64)
        self._m_ptr = np.zeros(n, dtype=np.uint64)
        self._m_ref = [None] * n   # the buffer each cached pointer belongs to
        self._state_ids = None     # Optimizer.load_state_dict installs a new state dict: the cached buffers are then stale
        self._grp = np.array([k * 4096 + gi for k, gi in self.owner], dtype=np.int64)
        self.chunk = None
        self._checked_idx = None
        rec = 48 * n + 4 * (n + 1)
        self._slot_bytes = (rec + 255) // 256 * 256
        self._pinned = None
        self._events = [None] * self.SLOTS
        self._slot = 0

    def zero_grad(self):
        """`optimizer.zero_grad(set_to_none=True)` of all the optimizers (plain loop: no per-optimizer bookkeeping)."""
        for p in self.params:
            p.grad = None

    def _buffers(self, idx):
        """momentum buffers of the parameters `idx` (created zero-filled on first use: torch's first step sets buf = d_p, which
        is what mu * 0 + d_p gives); the cache is dropped when an optimizer's state dict has been replaced (load_state_dict)."""
        ids = [id(o.state) for o in self.optimizers]
        if ids != self._state_ids:
            self._state_ids = ids
            self._m_ref = [None] * len(self.params)
        for i in idx:
            if self._m_ref[i] is not None:
                continue
            p = self.params[i]
            st = self.optimizers[self.owner[i][0]].state[p]
            m = st.get("momentum_buffer")
            if m is None:
                m = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if not (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous() and m.numel() == p.numel()):
                raise hip.LgdHipError("momentum buffer of a fused-SGD parameter must be a dense fp32 device tensor")
            self._m_ref[i] = m
            self._m_ptr[i] = m.data_ptr()

    @torch.no_grad()
    def step(self):
        lib = hip.load()
        if self.chunk is None:
            self.chunk = int(lib.lgd_sgd_chunk_elems())
        grads = [p.grad for p in self.params]
        idx = [i for i, g in enumerate(grads) if g is not None]
        if not idx:
            return
        hip.require_gpu(grads[idx[0]])
        if idx != self._checked_idx:   # layout checks once per set of parameters that receive gradients (autograd hands out the
            for i in idx:              # same kind of tensor every step; the checks cost ~0.12 ms of the ~0.4 ms this call takes)
                g = grads[i]
                if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
                    raise hip.LgdHipError("fused SGD needs dense contiguous fp32 gradients")
            self._checked_idx = idx
        self._buffers(idx)
        ii = np.asarray(idx, dtype=np.int64)
        k = len(idx)
        device = grads[idx[0]].device
        if self._pinned is None:
            self._pinned = torch.empty(self.SLOTS * self._slot_bytes, dtype=torch.uint8).pin_memory()
        s = self._slot
        self._slot = (s + 1) % self.SLOTS
        if self._events[s] is not None:
            self._events[s].synchronize()   # SLOTS steps old: long complete unless the host ran that far ahead
        host = self._pinned[s * self._slot_bytes:(s + 1) * self._slot_bytes]
        raw = host.numpy()
        tab = raw[:48 * k].view(_TENSOR_DT)
        blk = raw[48 * k:48 * k + 4 * (k + 1)].view(np.int32)
        ptr_of = torch.Tensor.data_ptr
        params = self.params
        tab["p"] = np.fromiter(map(ptr_of, [params[i] for i in idx]), dtype=np.uint64, count=k)
        tab["g"] = np.fromiter(map(ptr_of, [grads[i] for i in idx]), dtype=np.uint64, count=k)
        tab["m"] = self._m_ptr[ii]
        tab["n"] = self._numel[ii]
        hyper = {}
        for key in np.unique(self._grp[ii]).tolist():
            g = self.optimizers[key // 4096].param_groups[key % 4096]
            hyper[key] = (float(g["lr"]), float(g["weight_decay"]), float(g["momentum"]))
        hy = np.array([hyper[key] for key in self._grp[ii].tolist()], dtype=np.float32)
        tab["lr"], tab["wd"], tab["mu"] = hy[:, 0], hy[:, 1], hy[:, 2]
        tab["reserved"] = 0
        blk[0] = 0
        blk[1:] = np.cumsum((self._numel[ii] + self.chunk - 1) // self.chunk)
        n_blocks = int(blk[k])
        nbytes = 48 * k + 4 * (k + 1)
        dev = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dev.copy_(host[:nbytes], non_blocking=True)
        ev = self._events[s]
        if ev is None:
            ev = self._events[s] = torch.cuda.Event()
        ev.record()
        base = dev.data_ptr()
        hip.check(lib.lgd_sgd_clip_step(ctypes.c_void_p(base), ctypes.c_void_p(base + 48 * k), k, n_blocks,
                                        ctypes.c_float(self.clip_value), hip.stream_ptr()), "lgd_sgd_clip_step")
        # the kernel wrote p, g and the momentum buffers through raw pointers: tell autograd's version counters, which the fold
        # caches (ConvBN._frozen_fold, FrozenBatchNorm2d.scale_shift) key on and the saved-tensor check reads
        torch.autograd.graph.increment_version([params[i] for i in idx] + [grads[i] for i in idx] + [self._m_ref[i] for i in idx])
        for o in self.optimizers:
            o._opt_called = True   # what the LR schedulers' "scheduler.step() before optimizer.step()" check looks at
