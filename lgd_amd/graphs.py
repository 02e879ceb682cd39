"""hipGraph replay of the student's bottom-up backbone + FPN (forward AND backward) for the launch-bound small-batch configs.

[ref: the reference's DDP runs feed 2 images per GPU (README.md / BASELINE configs 4-5: bs 16 over 8 GPUs); its student backbone
is detectron2's ResNet + FPN, models/customized_detectors/retinanet.py:29-53.]  At 2 images per GPU an R-101 step is ~1,450 kernel
launches of 10-50 us against ~28 ms of Python + dispatch: the host is barely ahead of the GPU, so every host-heavy stretch (label
encoder, start of the backward, optimizer entry, step boundary) lets the GPU run dry (~10 % idle, tools/gap_profile.sh).  The
backbone is the shape-static two thirds of those launches: for a given padded image shape it always issues the same kernels on the
same buffers.  `GraphedBackbone` captures it once per (input shape, requires_grad pattern) with torch.cuda.make_graphed_callables
-- the forward launches into one hipGraph, the backward (captured through autograd, i.e. through the custom Functions of ops.py, whose
ctypes launches go to the capturing stream) into another -- and replays both with one host call each.  The box-count dependent part
of the step (teacher, losses) stays eager.

Opt-in (`Trainer(..., graph_backbone=True)`, `LGD_GRAPH_BACKBONE=1`, `bench.py --graph-backbone`): at 8+ images per GPU the step is
GPU-bound and replay buys nothing, and kernels inside a replayed graph are invisible to the per-launch event timers of bench.py.
"""
from collections import OrderedDict

import torch
import torch.nn as nn


class _BackboneFPN(nn.Module):
    """x -> (raw res features..., FPN features...) as a flat tuple (make_graphed_callables wants tensors in, tensors out)."""

    def __init__(self, raw_backbone, fpn):
        super().__init__()
        self.raw_backbone = raw_backbone
        self.fpn = fpn
        self.raw_keys = None
        self.fpn_keys = None

    def forward(self, x):
        raw = self.raw_backbone(x)
        feats = self.fpn(raw)
        if self.raw_keys is None:
            self.raw_keys, self.fpn_keys = tuple(raw.keys()), tuple(feats.keys())
        # the raw res features are handed on for the reference's signature only (DynamicTeacher ignores them,
        # dynamic_teacher.py:285-301): detached, so that the captured backward does not push zero gradients through them
        return tuple(raw[k].detach() for k in self.raw_keys) + tuple(feats[k] for k in self.fpn_keys)


class GraphedBackbone:
    """callable (images tensor) -> (raw_features dict, fpn features dict), replaying captured graphs in training mode."""

    def __init__(self, raw_backbone, fpn, max_graphs=16, warmup_iters=3):
        self.raw_backbone, self.fpn = raw_backbone, fpn
        self.max_graphs, self.warmup_iters = max_graphs, warmup_iters
        self._graphs = OrderedDict()
        self._frozen_print = None
        self.captures = 0

    def _state(self):
        """(requires_grad pattern, fingerprint of everything the captured kernels read that is NOT recomputed inside the graph):
        frozen parameters and buffers feed caches (FrozenBN scale / shift, folded filters of frozen convs) that are filled
        outside the capture; writing one of them (load_state_dict) must drop the graphs."""
        pattern, frozen = [], []
        for m in (self.raw_backbone, self.fpn):
            for p in m.parameters():
                pattern.append(p.requires_grad)
                if not p.requires_grad:
                    frozen.append((p.data_ptr(), p._version))
            for b in m.buffers():
                frozen.append((b.data_ptr(), b._version))
        return tuple(pattern), tuple(frozen)

    def reset(self):
        self._graphs.clear()

    def __call__(self, x):
        pattern, frozen = self._state()
        if frozen != self._frozen_print:
            self._graphs.clear()
            self._frozen_print = frozen
        key = (tuple(x.shape), x.dtype, x.device, pattern)
        entry = self._graphs.get(key)
        if entry is None:
            mod = _BackboneFPN(self.raw_backbone, self.fpn)
            mod.train()
            sample = x.detach().clone()
            torch.cuda.make_graphed_callables(mod, (sample,), num_warmup_iters=self.warmup_iters, allow_unused_input=True)
            self.captures += 1
            entry = mod
            self._graphs[key] = entry
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        outs = entry(x)
        nr = len(entry.raw_keys)
        return dict(zip(entry.raw_keys, outs[:nr])), dict(zip(entry.fpn_keys, outs[nr:]))
