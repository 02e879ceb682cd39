"""DynamicTeacher -- the LGD label-appearance encoder, inter-object relation adapter and
intra-object knowledge mapper, MI355X-native.

Same constructor (`DynamicTeacher(cfg)`), same `forward(info_list)` contract and the same
state_dict names/shapes as the reference module
  [ref: models/customized_detectors/dynamic_teacher/dynamic_teacher.py:16-301,
        .../label_encoder.py:119-276, .../spatial_transformer.py:9-47, .../layers.py:6-32]
so released checkpoints load and the meta-arch above it is unchanged.  What differs is HOW:
  * no dense masks, no per-image python loops, no D2H syncs: boxes stay on the device, the mask of
    every (level, box) is an integer rectangle built by one HIP kernel (ops.BoxGeometry);
  * mask pooling / rendering / GN(1) / attention run as hand-written HIP kernels over all levels
    and images at once (lgd_amd/csrc/*.hip); only the dense 3x3 convolutions go to MIOpen.
The third value returned by forward() is the BoxGeometry (the reference returns the dense
masks there; nothing downstream reads them -- base_distillator.py:34-64 ignores the argument).
"""
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip, ops, streams
from .registry import CUSTOMIZED_DETECTORS_REGISTRY


# ----------------------------------------------------------------------------------------- layers.py
def get_norm(channels, nr_groups=1, affine_flag=False):
    return nn.GroupNorm(num_groups=nr_groups, num_channels=channels, affine=affine_flag)


def get_MLP(nr_layers, channels, has_norm, has_relu=True, affine_flag=False):
    def unit():
        mods = [nn.Linear(channels, channels)]
        if has_norm:
            mods.append(nn.LayerNorm([channels], elementwise_affine=affine_flag))
        if has_relu:
            mods.append(nn.ReLU())
        return nn.Sequential(*mods)
    return nn.Sequential(*[unit() for _ in range(nr_layers)])


def get_CONVS(nr_layers, channels, has_norm, has_relu=True, nr_groups=1, affine_flag=False):
    def unit():
        mods = [ops.Conv3x3(channels, channels)]
        if has_norm:
            mods.append(get_norm(channels, nr_groups, affine_flag))
        if has_relu:
            mods.append(nn.ReLU())
        return nn.Sequential(*mods)
    return nn.Sequential(*[unit() for _ in range(nr_layers)])


def _lin_ln_relu(layer, x):
    """Linear / pointwise Conv1d (length-1 sequences == row-wise linear map) -> LayerNorm(no affine) -> ReLU:
    fp32 MFMA GEMM kernel + one fused row-LN + ReLU kernel, one autograd node."""
    return ops.mlp_ln_relu(x, [(layer.weight, layer.bias)])


# ----------------------------------------------------------------------------------------- STN
class STN(nn.Module):
    """T-Net without identity shortcut [ref: spatial_transformer.py:9-47]; x (T,k) -> (T,k,k)."""

    def __init__(self, k=64):
        super().__init__()
        self.conv1 = nn.Conv1d(k, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.fc1 = nn.Linear(1024, 512)
        self.fc2 = nn.Linear(512, 256)
        self.fc3 = nn.Linear(256, k * k)
        self.k = k

    def forward(self, x):
        # the max over the length-1 axis between conv3 and fc1 (spatial_transformer.py:34) is a no-op: five Linear -> LN -> ReLU layers
        # and fc3 as ONE autograd node (ops.mlp_ln_relu)
        h = ops.mlp_ln_relu(x, [(m.weight, m.bias) for m in (self.conv1, self.conv2, self.conv3, self.fc1, self.fc2)],
                            last=(self.fc3.weight, self.fc3.bias))
        return h.view(-1, self.k, self.k)


# ----------------------------------------------------------------------------------------- label encoder
class LabelEncoder(nn.Module):
    """PointNet over per-box descriptors [ref: label_encoder.py:119-276], R=1, no noise."""

    def __init__(self, category_format="one_hot", box_format="x1y1x2y2", nr_fg_classes=80, noise_std=0.0,
                 add_context_box=False, parse_mask=False):
        super().__init__()
        if category_format != "one_hot":
            raise ValueError("category_format %r not supported (shipped configs use 'one_hot')" % category_format)
        if parse_mask:
            raise ValueError("LOAD_LABELMAP / mask descriptors are Mask R-CNN only (out of scope)")
        if box_format not in ("x1y1x2y2", "x1y1wh"):
            raise AssertionError(box_format)
        self.category_format, self.box_format = category_format, box_format
        self.nr_fg_classes, self.add_context_box = nr_fg_classes, add_context_box
        self.R, self.noise_std = 1, noise_std
        self.inp = 4 + nr_fg_classes
        self.stn_desc = STN(self.inp)
        self.stn_feat = STN(64)
        self.conv1 = nn.Conv1d(self.inp, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, 1024, 1)
        self.conv4 = nn.Conv1d(1088, 256, 1)

    @torch.no_grad()
    def encode_descriptors(self, targets, img_h, img_w, device):
        """[ref: label_encoder.py:12-115] without the per-image host round trips: the annotations are concatenated,
        moved to the device once, and ONE kernel emits descriptors + clamped boxes for the whole mini-batch.
        Returns descriptors (T,84) in [-1,1], clamped boxes (T,4), counts (host list), img_off (B+1 int32, device),
        inst_labels (list of per-image class tensors)."""
        K = self.nr_fg_classes
        in_counts, out_counts, boxes, classes, inst_labels = [], [], [], [], []
        for inst in targets:
            n = len(inst)
            in_counts.append(n)
            out_counts.append(n + 1 if (n > 0 and self.add_context_box) else max(n, 1))
            if n > 0:
                cls = inst.gt_classes.reshape(n)
                if not cls.is_cuda:  # validate where it is free (label_encoder.py:98)
                    assert bool(((cls >= 0) & (cls <= K - 1)).all()), "gt_classes outside [0, %d]" % (K - 1)
                boxes.append(inst.gt_boxes.tensor.reshape(n, 4).to(torch.float32))
                classes.append(cls)
                inst_labels.append(cls)
            else:
                inst_labels.append(torch.zeros(1))  # label_encoder.py:66,114
        if boxes:
            bb = torch.cat(boxes, 0).to(device, non_blocking=True)
            cc = torch.cat(classes, 0).to(device, non_blocking=True)
        else:
            bb, cc = torch.zeros((0, 4), device=device), torch.zeros((0,), dtype=torch.int64, device=device)
        desc, clamped, img_off = ops.box_descriptors(bb, cc, in_counts, out_counts, img_h, img_w, K, self.add_context_box,
                                                     self.box_format == "x1y1wh")
        return desc, clamped, out_counts, img_off, inst_labels

    def forward(self, x0):
        batched_inputs, images, _, fpn = x0
        device = fpn if isinstance(fpn, torch.device) else (fpn["p3"].device if isinstance(fpn, dict) else fpn[0].device)
        _, _, h, w = images.tensor.shape
        targets = [x["instances"] for x in batched_inputs]
        x, boxes, counts, img_off, inst_labels = self.encode_descriptors(targets, h, w, device)
        m1 = self.stn_desc(x)
        x1 = ops.row_vecmat(x, m1)                       # (x^T M)^T, label_encoder.py:241
        hfeat = _lin_ln_relu(self.conv1, x1)
        m2 = self.stn_feat(hfeat)
        xf = ops.row_vecmat(hfeat, m2)                   # label_encoder.py:248
        h3 = ops.mlp_ln_relu(xf, [(self.conv2.weight, self.conv2.bias), (self.conv3.weight, self.conv3.bias)])
        g = ops.segment_max_broadcast(h3, img_off)       # per-image max, broadcast back to the image's rows
        out = _lin_ln_relu(self.conv4, torch.cat([xf, g], 1))
        return out, m1, m2, boxes, {"h": h, "w": w}, inst_labels, counts


# ----------------------------------------------------------------------------------------- teacher
_RUNTIME = weakref.WeakKeyDictionary()   # DynamicTeacher -> side-stream state (DynamicTeacher._rt)


@CUSTOMIZED_DETECTORS_REGISTRY.register()
class DynamicTeacher(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.nr_fpn_channels = cfg.MODEL.FPN.OUT_CHANNELS
        self.num_classes = cfg.NUM_CLASSES
        assert self.nr_fpn_channels == 256
        assert self.num_classes == 80
        d = cfg.MODEL.DISTILLATOR
        self.interact_pattern = d.TEACHER.INTERACT_PATTERN
        self.strides = cfg.MODEL.RECIPROCAL_FPN_STRIDES
        self.box_format = d.LABEL_ENCODER.BOX_FORMAT
        self.category_format = d.LABEL_ENCODER.CATEGORY_FORMAT
        self.use_seg_map = d.LABEL_ENCODER.LOAD_LABELMAP
        self.add_context_box = d.TEACHER.ADD_CONTEXT_BOX
        self.detach_appearance_embed = d.TEACHER.DETACH_APPEARANCE_EMBED
        if self.interact_pattern not in ("stuGuided", "labelGuided", "student_fill", "teacher_fill"):
            raise ValueError("interact pattern: {} not supported !".format(self.interact_pattern))
        C = self.nr_fpn_channels
        self.label_encoder_ = LabelEncoder(self.category_format, self.box_format, self.num_classes,
                                           add_context_box=self.add_context_box, parse_mask=self.use_seg_map)
        self.canoni_proj_1D = get_MLP(1, C, has_norm=True, has_relu=True, affine_flag=False)
        self.student_proj_2D = get_CONVS(1, C, has_norm=True, has_relu=True, nr_groups=1, affine_flag=False)
        self.local_inst_proj_2D = ops.Conv3x3(C, C)
        self.global_ctx_proj_1D = nn.Linear(C, C)
        self.local_inst_proj_1D = nn.Linear(C, C)
        self.refinement_module = nn.Sequential(
            ops.Conv3x3(C, C), get_norm(C, 1, False), nn.ReLU(),
            ops.Conv3x3(C, C), get_norm(C, 1, False), nn.ReLU(),
            ops.Conv3x3(C, C), get_norm(C, 1, False))
        self.nr_transformer_heads = d.TEACHER.NR_TRANSFORMER_HEADS
        self.multi_head_attn = nn.MultiheadAttention(C, self.nr_transformer_heads)

    # rendering's "+ ctx, ReLU" and the GroupNorm(1) + ReLU pairs of the refinement module run INSIDE the next convolution's input transform
    # (ops.ctx_shift_fold / ops.conv3x3_gn -> the `pre` affine of the Winograd input transform, its adjoint applies the activation bits):
    # the activated / normalised maps are never written or re-read -- 2 + 2 x 2 map transfers forward, 3 + 2 x 3 backward per level.
    # False: every activation as its own pass (ops.bias_ctx_relu, ops.gn1), what the folded form is tested against.
    fold_activations = True

    # -- a-8: per-box projected embeddings painted back through the box rectangles, conv, +ctx, ReLU
    def rendering(self, attn_out, geom):
        """attn_out (L,T,C) -> (list of L maps, pre): pre is None (the maps are final) or the affine whose ReLU(scale x + shift) the next
        convolution applies while it loads.  [ref: dynamic_teacher.py:106-190]
        The context row of every image is projected too (one GEMM for all rows) but never painted."""
        proj = ops.linear(attn_out, self.local_inst_proj_1D.weight, self.local_inst_proj_1D.bias)
        painted = ops.render_paint(geom, proj, skip_last=self.add_context_box)
        conv = self.local_inst_proj_2D
        if self.add_context_box:
            last = hip.to_device([o - 1 for o in _offsets(geom.counts)[1:]], torch.int64, attn_out.device)
            ctx = ops.linear(attn_out[:, last], self.global_ctx_proj_1D.weight, self.global_ctx_proj_1D.bias)  # (L,B,C)
            if self.fold_activations:
                pre, maps = ops.ctx_shift_fold(conv.levels(painted), ctx)
                return maps, pre
            return ops.bias_ctx_relu(conv.levels(painted), ctx), None
        return conv.levels(painted, relu=True), None

    def refine(self, xs, pre=None):
        """[ref: dynamic_teacher.py:67-73,280-281] on all levels: conv -> GN(1) -> ReLU -> conv -> GN(1) -> ReLU -> conv -> GN(1); each
        convolution one Winograd pass over the concatenated levels."""
        m = self.refinement_module
        if self.fold_activations:
            (a0, r0), = ops.conv3x3_gn(xs, [(m[0].weight, m[0].bias, None, None)], 1, pre=pre)
            (a1, r1), = ops.conv3x3_gn(r0, [(m[3].weight, m[3].bias, None, None)], 1, pre=a0)
            return ops.gn1(m[6].levels(r1, pre=a1), relu=False)   # the last GroupNorm has no ReLU behind it and its output IS the teacher feature
        for idx, relu in ((0, True), (3, True), (6, False)):
            xs = ops.gn1(m[idx].levels(xs, pre=pre if idx == 0 else None), relu=relu)
        return xs

    # The label encoder (PointNet over the box descriptors: two spatial transformers, ~30 small GEMM / LayerNorm launches of a handful of
    # workgroups each, a serial chain that leaves the chip idle for ~0.5 ms forward and ~0.7 ms backward) depends on the annotations only, not
    # on the student's features: encode_ahead() issues it on a SIDE stream before the backbone, where it runs under the backbone's kernels;
    # autograd runs its backward on the same stream, under the backbone's backward.  LGD_TEACHER_STREAM=0: in line, on the caller's stream.
    side_stream = os.environ.get("LGD_TEACHER_STREAM", "1") != "0"

    def _rt(self):
        """runtime state of the side stream (stream objects, hook handles, the result in flight), kept OUTSIDE the module's attributes: a
        copy.deepcopy / pickle of the model must not meet a stream"""
        st = _RUNTIME.get(self)
        if st is None:
            st = _RUNTIME[self] = {}
        return st

    def encode_ahead(self, batched_inputs, images):
        """label encoder + canonical projection of this mini-batch on the side stream (a no-op off the GPU or when switched off); forward()
        picks the result up.  Called by the distillator between the student's preprocessing and its backbone."""
        rt = self._rt()
        rt["ahead"] = None
        dev = images.tensor.device
        if not (self.side_stream and dev.type == "cuda" and ops.side_streams_ok()):
            return
        main = rt["main"] = torch.cuda.current_stream(dev)
        side = rt["side"] = streams.side(dev, "teacher")
        self._join_hooks()
        side.wait_stream(main)   # the weights last step's optimizer wrote, the annotations the loader copied
        with torch.cuda.stream(side):
            enc = self.label_encoder_((batched_inputs, images, None, dev))
            canoni = _lin_ln_relu(self.canoni_proj_1D[0][0], enc[0])
        rt["ahead"] = (id(batched_inputs), enc, canoni, side, streams.done(side))

    def _join_hooks(self):
        """Data-parallel runs: DistributedDataParallel starts a bucket's all-reduce from the gradient hook of the bucket's LAST parameter and orders it
        behind the stream THAT hook runs on.  The encoder's parameter gradients are written on the side stream, those of the other parameters of
        the same bucket on the main one: as each encoder gradient is accumulated the two streams are joined (each waits for the other's work so
        far), so whichever hook comes last, its stream has seen every gradient of the bucket.  The encoder's backward is the last thing the engine
        issues, so the joins cost no overlap.  Single-process runs need none of this (the engine joins the streams when backward() returns)."""
        rt = self._rt()
        if rt.get("handles") or not streams._multi_rank():   # (asked on every forward: registered on the first one that runs in a process group of > 1 ranks)
            return
        ref = weakref.ref(self)

        def join(_param):
            me = ref()
            rt = (_RUNTIME.get(me) if me is not None else None) or {}
            side, main = rt.get("side"), rt.get("main")
            if side is not None and main is not None:
                side.wait_stream(main)
                main.wait_stream(side)
        rt["handles"] = [p.register_post_accumulate_grad_hook(join) for m in (self.label_encoder_, self.canoni_proj_1D) for p in m.parameters()]

    def interactive_remapping(self, label_embed, boxes, counts, feats, img_size_dict, canoni=None):
        """[ref: dynamic_teacher.py:209-283]"""
        if self.detach_appearance_embed:
            feats = {k: v.detach() for k, v in feats.items()}
        keys = list(feats.keys())
        if canoni is None:
            canoni = _lin_ln_relu(self.canoni_proj_1D[0][0], label_embed)
        sp = self.student_proj_2D[0][0]
        geom = ops.BoxGeometry(boxes, counts, (img_size_dict["h"], img_size_dict["w"]),
                               [tuple(feats[k].shape[-2:]) for k in keys])
        # student_proj_2D = conv3x3 -> GN(1) -> ReLU, consumed only by the mask pooling: GN+ReLU are applied inside
        # the pooling kernel, the normalised maps are never written (and never re-read by a separate pooling pass)
        app = ops.gn_relu_mask_pool(geom, sp.levels([feats[k] for k in keys]))  # (L,T,C)
        a = self.multi_head_attn
        if self.interact_pattern == "student_fill":
            att = app
        elif self.interact_pattern == "teacher_fill":
            att = canoni.unsqueeze(0).expand(len(keys), -1, -1)
        elif self.interact_pattern == "stuGuided":  # Q = appearance, K = V = label embeddings
            att = ops.mha_blockdiag(app, canoni.unsqueeze(0), counts, a.in_proj_weight, a.in_proj_bias,
                                    a.out_proj.weight, a.out_proj.bias, self.nr_transformer_heads, geom.img_off)
        else:  # labelGuided: Q = label embeddings, K = V = appearance
            att = ops.mha_blockdiag(canoni.unsqueeze(0), app, counts, a.in_proj_weight, a.in_proj_bias,
                                    a.out_proj.weight, a.out_proj.bias, self.nr_transformer_heads, geom.img_off)
        tea = dict(zip(keys, self.refine(*self.rendering(att, geom))))
        return tea, geom

    def forward(self, info_list):
        rt = _RUNTIME.get(self)
        ahead = rt.pop("ahead", None) if rt else None
        canoni = None
        if ahead is not None and ahead[0] == id(info_list[0]):
            _, enc, canoni, side, ev = ahead
            main = torch.cuda.current_stream(canoni.device)
            streams.join(main, side, (canoni, enc), event=ev)   # made on the side stream (uploads and magnitude tags included), read on this one from here on
        else:
            enc = self.label_encoder_(info_list)
        x, _, _, boxes, img_size_dict, inst_labels, counts = enc
        tea, geom = self.interactive_remapping(x, boxes, counts, info_list[-1], img_size_dict, canoni=canoni)
        return tea, inst_labels, geom


def _offsets(counts):
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    return off
