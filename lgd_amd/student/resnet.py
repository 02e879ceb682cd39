"""ResNet-50/101 bottom-up backbone with FrozenBN, detectron2 parameter names
(`stem.conv1.weight`, `res3.0.conv2.norm.running_var`, `res4.5.shortcut.weight`, ...)
[d2-memory: detectron2/modeling/backbone/resnet.py @ v0.3; SURVEY.md appendix A]."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class FrozenBatchNorm2d(nn.Module):
    """y = x * w * rsqrt(var + eps) + (b - mean * w * rsqrt(var + eps)); all four are buffers."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(c))
        self.register_buffer("bias", torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c) - eps)

    def scale_shift(self):
        """(scale, shift) of the frozen affine; the four buffers never change during training, so the pair is computed once
        and reused until a buffer is written (load_state_dict / .to(): tracked by the tensors' identity and version counters)."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((id(b), b._version) for b in bufs)
        if getattr(self, "_ss_key", None) != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                self._ss = (scale, self.bias - self.running_mean * scale)
            self._ss_key = key
        return self._ss

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class ConvBN(nn.Conv2d):
    """bias-free conv followed by FrozenBN [+ReLU].  The frozen affine is folded into the conv
    (scaled weights + bias) so no separate normalisation pass touches the activation."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1):
        super().__init__(cin, cout, k, stride, padding, dilation, groups, bias=False)
        self.norm = FrozenBatchNorm2d(cout)
        self._plain3x3 = (k == 3 and stride == 1 and padding == 1 and dilation == 1 and groups == 1)
        self._pointwise = (k == 1 and stride == 1 and padding == 0 and groups == 1)
        # 1x1 / stride 2 (STRIDE_IN_1X1 bottlenecks, projection shortcuts) = take every other pixel, then a pointwise conv:
        # forward, input and weight gradient then run as plain GEMMs instead of the library's strided implicit-GEMM kernels,
        # whose weight gradient transposes both operands to NHWC first (4.5 ms/step at config 2)
        self._pointwise_s2 = (k == 1 and stride == 2 and padding == 0 and groups == 1)

    def _frozen_fold(self, scale):
        """weight * scale of a frozen convolution, computed once and reused until the weight (or the FrozenBN) is written."""
        key = (id(self.weight), self.weight._version, id(scale))
        if getattr(self, "_fold_key", None) != key:
            with torch.no_grad():
                self._fold = self.weight * scale.view(-1, 1, 1, 1)
            self._fold_key = key
        return self._fold

    def _cached_fold(self, scale):
        """the folded filter this step's forward may use without a launch of its own: the frozen fold, or the fold StepFolds.prepare()
        wrote for the current version of a trainable weight (None: fold inside the op)."""
        if not self.weight.requires_grad:
            return self._frozen_fold(scale)
        sf = getattr(self, "_step_fold", None)
        if sf is not None and sf[0] == (self.weight._version, id(scale)):
            return sf[1]
        return None

    @staticmethod
    def subsample2(x):
        return ops.subsample2(x)

    def forward(self, x, relu=False, residual=None, subsampled=False, raw=False, pre=None):
        """conv -> FrozenBN [-> += residual] [-> ReLU]: the affine is folded into the filter; bias, residual and ReLU are ONE
        pass over the conv output (ops.bias_act) instead of three.  raw (1x1 only): return the scaled convolution alone -- the 3x3
        convolution that consumes it applies this layer's shift + ReLU inside its input transform (its `pre` = this shift)."""
        scale, shift = self.norm.scale_shift()
        if self._pointwise or self._pointwise_s2:  # fold + GEMM + epilogue (and their backward) as one autograd node
            if self._pointwise_s2 and not subsampled:
                x = self.subsample2(x)
            return ops.pointwise_conv_bn(x, self.weight, scale, None if raw else shift, residual, relu and not raw, self._cached_fold(scale))
        if self.weight.requires_grad and self._plain3x3 and residual is None:
            # trainable 3x3 / stride 1: the scale is folded inside the Winograd filter transform (no scaled copy of the weights, and the
            # backward returns the gradient of the RAW filter); bias + ReLU ride in the output transform
            return ops.conv3x3(x, self.weight, shift, relu=relu, scale=scale, pre=pre)
        if self.weight.requires_grad:
            w = self.weight * scale.view(-1, 1, 1, 1)
        else:  # frozen (FREEZE_AT prefix, or the backbone-freeze phase): the folded filter is reused until the weight is written
            w = self._frozen_fold(scale)
        if self._plain3x3 and residual is None:  # 3x3 / stride 1: Winograd transforms + GEMMs (bias + ReLU fused in the output transform)
            return ops.conv3x3(x, w, shift, relu=relu, pre=pre)
        y = ops.lib_conv2d(x, w, None, self.stride, self.padding, self.dilation, self.groups)
        return ops.bias_act(y, shift, residual, relu)


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, mid, stride, stride_in_1x1=True, groups=1, dilation=1):
        super().__init__()
        self.shortcut = ConvBN(cin, cout, 1, stride) if (cin != cout or stride != 1) else None
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = ConvBN(cin, mid, 1, s1)
        self.conv2 = ConvBN(mid, mid, 3, s3, padding=dilation, dilation=dilation, groups=groups)
        self.conv3 = ConvBN(mid, cout, 1)

    def forward(self, x):
        shared = self.conv1._pointwise_s2 and self.shortcut is not None and self.shortcut._pointwise_s2
        if shared:  # conv1 and the projection shortcut read the same every-other-pixel view of x: take it once
            x = ConvBN.subsample2(x)
        # conv1's FrozenBN shift + ReLU folded into conv2's Winograd input transform (and the mask into its adjoint): conv1 hands
        # over its raw GEMM output, no epilogue pass over the mid-size map in either direction
        s1 = 2 if (self.conv1._pointwise_s2 and not shared) else 1
        fold = self.conv2._plain3x3 and ops.conv3x3_folds_pre(x.shape[0], self.conv1.out_channels, (x.shape[2] + s1 - 1) // s1,
                                                              (x.shape[3] + s1 - 1) // s1, self.conv2.out_channels, x.device, x.dtype)
        pre = self.conv1.norm.scale_shift()[1] if fold else None
        if (self.shortcut is None and self.conv1._pointwise and x.is_cuda and x.dtype == torch.float32
                and torch.is_grad_enabled() and x.requires_grad):
            # identity block: conv1 and the shortcut as one node, so that conv1's input-gradient GEMM accumulates onto the
            # shortcut's gradient instead of a separate add pass (ops._PointwiseConvBNSkip)
            scale, shift = self.conv1.norm.scale_shift()
            out, sc = ops.pointwise_conv_bn_skip(x, self.conv1.weight, scale, shift, raw=fold, wf=self.conv1._cached_fold(scale))
            return self.conv3(self.conv2(out, relu=True, pre=pre), relu=True, residual=sc)
        out = self.conv1(x, relu=True, subsampled=shared, raw=fold)
        out = self.conv2(out, relu=True, pre=pre)
        sc = self.shortcut(x, subsampled=shared) if self.shortcut is not None else x
        return self.conv3(out, relu=True, residual=sc)


class DeformBottleneck(Bottleneck):
    """Bottleneck whose 3x3 conv is a (modulated) deformable conv driven by a zero-initialised offset conv
    (`conv2_offset`: 18 offsets [+ 9 masks, sigmoid]) [d2-memory: DeformBottleneckBlock]."""

    def __init__(self, cin, cout, mid, stride, stride_in_1x1=True, groups=1, dilation=1, modulated=True):
        super().__init__(cin, cout, mid, stride, stride_in_1x1, groups, dilation)
        self.modulated = modulated
        self.conv2_offset = nn.Conv2d(mid, 27 if modulated else 18, 3, self.conv2.stride, dilation, dilation)
        nn.init.constant_(self.conv2_offset.weight, 0)
        nn.init.constant_(self.conv2_offset.bias, 0)

    def forward(self, x):
        shared = self.conv1._pointwise_s2 and self.shortcut is not None and self.shortcut._pointwise_s2
        if shared:
            x = ConvBN.subsample2(x)
        sc = None
        if (self.shortcut is None and self.conv1._pointwise and x.is_cuda and x.dtype == torch.float32
                and torch.is_grad_enabled() and x.requires_grad):
            # identity block: conv1 and the shortcut as one node, as in Bottleneck (no add pass for the block input's two gradients)
            scale1, shift1 = self.conv1.norm.scale_shift()
            out, sc = ops.pointwise_conv_bn_skip(x, self.conv1.weight, scale1, shift1, wf=self.conv1._cached_fold(scale1))
        else:
            out = self.conv1(x, relu=True, subsampled=shared)
        co = self.conv2_offset
        if out.is_cuda and co.stride == (1, 1) and co.dilation == (1, 1) and co.padding == (1, 1):
            om = ops.conv3x3(out, co.weight, co.bias)
        else:
            om = co(out)
        c2 = self.conv2
        scale, shift = c2.norm.scale_shift()
        # modulated deformable convolution (DCNv2) [ref: configs/Distillation/RetinaNet/retinanet_R_101_dcnv2_*.yaml:7-8; d2-memory:
        # ModulatedDeformConv]: out[n,o,y,x] = sum_{c,k} W[o,c,k] mask[n,k,y,x] bilinear(in[n,c], y s - p + ky d + dy_k, x s - p + kx d + dx_k),
        # offsets stored as (dy, dx) channel pairs per tap k = 3 ky + kx; HIP gather kernel -> column matrix -> library GEMM (csrc/dcn.hip)
        wf = c2.weight * scale.view(-1, 1, 1, 1)
        if self.modulated and out.is_cuda:
            # chunk(3) -> offset = cat(o1, o2) = channels 0..17, mask = sigmoid(channels 18..26): read in place by the kernels
            out = ops.deform_conv3x3_packed(out, om, wf, None, c2.stride[0], c2.padding[0], c2.dilation[0])
        else:
            if self.modulated:
                o1, o2, m = torch.chunk(om, 3, dim=1)
                offset, mask = torch.cat((o1, o2), 1), m.sigmoid()
            else:
                offset, mask = om, None
            out = ops.deform_conv3x3(out, offset, mask, wf, None, c2.stride[0], c2.padding[0], c2.dilation[0])
        out = ops.bias_act(out, shift, None, True) if out.is_cuda else F.relu_(out + shift.view(1, -1, 1, 1))   # one pass
        if sc is None:
            sc = self.shortcut(x, subsampled=shared) if self.shortcut is not None else x
        return self.conv3(out, relu=True, residual=sc)


class Stem(nn.Module):
    def __init__(self, cin=3, cout=64):
        super().__init__()
        self.conv1 = ConvBN(cin, cout, 7, 2, 3)

    def forward(self, x):
        c = self.conv1
        if x.is_cuda and x.dtype == torch.float32 and not (torch.is_grad_enabled() and (x.requires_grad or c.weight.requires_grad)):
            # frozen stem (FREEZE_AT >= 1, the reference's configs): conv with the folded filter, then bias + ReLU + 3x3/2 max-pool
            # in ONE pass over the 550 MB conv output (ops.stem_bias_relu_maxpool) instead of an epilogue pass and a pooling pass
            scale, shift = c.norm.scale_shift()
            with torch.no_grad():
                wf = c._frozen_fold(scale)
                if ops.stem_conv_pool_ok(x, wf) and c.stride == (2, 2) and c.padding == (3, 3):
                    return ops.stem_conv_pool(x, wf, shift)   # csrc/stem.hip: convolution, shift, ReLU and the pool in one kernel
                return ops.stem_bias_relu_maxpool(ops.lib_conv2d(x, wf, None, c.stride, c.padding), shift)
        return F.max_pool2d(self.conv1(x, relu=True), 3, 2, 1)


class ResNet(nn.Module):
    """returns {"res3","res4","res5"} (strides 8/16/32; 512/1024/2048 channels)."""
    BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth=50, out_features=("res3", "res4", "res5"), freeze_at=2, stride_in_1x1=True,
                 num_groups=1, width_per_group=64, res2_out=256, stem_out=64, deform_on_per_stage=(False,) * 4,
                 deform_modulated=False):
        super().__init__()
        self.stem = Stem(3, stem_out)
        self.out_features = tuple(out_features)
        cin, cout, mid = stem_out, res2_out, num_groups * width_per_group
        self.stage_names = []
        for i, n in enumerate(self.BLOCKS[depth]):
            blocks = []
            for j in range(n):
                stride = 2 if (j == 0 and i > 0) else 1
                if deform_on_per_stage[i]:
                    blocks.append(DeformBottleneck(cin, cout, mid, stride, stride_in_1x1, num_groups, modulated=deform_modulated))
                else:
                    blocks.append(Bottleneck(cin, cout, mid, stride, stride_in_1x1, num_groups))
                cin = cout
            name = "res%d" % (i + 2)
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            cout, mid = cout * 2, mid * 2
        self.out_channels = {"res2": res2_out, "res3": res2_out * 2, "res4": res2_out * 4, "res5": res2_out * 8}
        self.freeze_at = freeze_at
        self.freeze(freeze_at)

    def freeze(self, freeze_at):
        """FREEZE_AT=k: stem and res2..res(k) get requires_grad=False [d2-memory: ResNet.freeze]."""
        if freeze_at >= 1:
            for p in self.stem.parameters():
                p.requires_grad = False
        for idx, name in enumerate(self.stage_names, start=2):
            if freeze_at >= idx:
                for p in getattr(self, name).parameters():
                    p.requires_grad = False

    def forward(self, x):
        outs = {}
        # frozen prefix: no autograd graph, nothing saved for backward
        frozen = torch.no_grad() if self.freeze_at >= 1 else torch.enable_grad()
        with frozen:
            x = self.stem(x)
        for idx, name in enumerate(self.stage_names, start=2):
            if self.freeze_at >= idx:
                with torch.no_grad():
                    x = getattr(self, name)(x)
            else:
                x = getattr(self, name)(x)
            if name in self.out_features:
                outs[name] = x
        return outs


_STEP_IMAGES = os.environ.get("LGD_STEP_IMAGES", "1") != "0"   # 0: every 1x1 product splits its filter in front of itself (A/B runs)


class StepFolds:
    """w * scale of EVERY trainable pointwise ConvBN of a model in one launch per step (lgd_scale_rows_multi) instead of one 5 us launch
    per convolution inside its op (62 per step for R-101): at 2 images per GPU the step is bound by the device's ~1,100 short launches.
    prepare() before the forward pass; a convolution whose weight has been written since (or that prepare() never saw) folds inside
    its op as before (ConvBN._cached_fold)."""

    def __init__(self, model):
        self.model = model
        self._key = None
        self._candidates = None

    def _build(self, mods):
        import numpy as np
        dev = mods[0].weight.device
        offs, off = [], 0
        for m in mods:
            offs.append(off)
            off += (m.weight.numel() + 3) // 4 * 4      # every fold starts on a 16-byte boundary
        self.flat = torch.empty(off, dtype=torch.float32, device=dev)
        dt = np.dtype([("w", "<u8"), ("scale", "<u8"), ("out", "<u8"), ("rows", "<i4"), ("cols", "<i4")])
        tab = np.zeros(len(mods), dtype=dt)
        blk0 = np.zeros(len(mods), dtype=np.int32)
        blk = 0
        self.views, self.scales = [], []
        for i, m in enumerate(mods):
            scale = m.norm.scale_shift()[0]
            n = m.weight.numel()
            v = self.flat[offs[i]:offs[i] + n].view_as(m.weight)
            tab[i] = (m.weight.data_ptr(), scale.data_ptr(), v.data_ptr(), m.weight.shape[0], n // m.weight.shape[0])
            blk0[i] = blk
            blk += (n + 1023) // 1024
            self.views.append(v)
            self.scales.append(scale)
        self.tab = torch.from_numpy(tab.view(np.uint8).copy()).to(dev)     # built once per set of trainable convolutions (phase changes)
        self.blk0 = torch.from_numpy(blk0).to(dev)
        self.nblk = blk
        self.mods = mods
        # max |w * scale| per fold, left by the same launch (the f16x2 scale of the filter images: ops.gemm2h_bmm looks a filter up by its offset
        # in `flat`)
        self.amax = torch.zeros(len(mods), dtype=torch.int32, device=dev)
        self._amax_of = {v.storage_offset(): (self.amax[i:i + 1], v.numel()) for i, v in enumerate(self.views)}
        # ... and ONE more launch splits every fold into the f16x2 images of its two products (W: forward, W^T: input gradient) instead of a ~5 us
        # launch in front of each of them (lgd_gemm2h_split_multi; ops.gemm2h_bmm finds an image by the fold's offset and orientation)
        from .. import hip
        lib = hip.load()
        st = np.dtype([("a", "<u8"), ("img", "<u8"), ("amax", "<u8"), ("inv", "<u8"), ("M", "<i4"), ("K", "<i4"), ("sm", "<i4"), ("sk", "<i4")])
        shapes = []
        for v in self.views:
            co, ci = v.shape[0], v.numel() // v.shape[0]
            shapes += [(co, ci, ci, 1), (ci, co, 1, ci)]     # (M, K, stride of m, stride of k)
        sizes = [int(lib.lgd_gemm2h_image_bytes(1, m_, k_)) for m_, k_, _, _ in shapes]
        self.images = torch.empty(sum(sizes), dtype=torch.uint8, device=dev)
        self.img_inv = torch.empty(len(shapes), dtype=torch.float32, device=dev)
        stab = np.zeros(len(shapes), dtype=st)
        sblk0 = np.zeros(len(shapes), dtype=np.int32)
        self._image_of = {}
        ioff = blk = 0
        for j, (m_, k_, sm_, sk_) in enumerate(shapes):
            v = self.views[j // 2]
            stab[j] = (v.data_ptr(), self.images.data_ptr() + ioff, self.amax.data_ptr() + 4 * (j // 2), self.img_inv.data_ptr() + 4 * j, m_, k_, sm_, sk_)
            sblk0[j] = blk
            blk += ((k_ + 15) // 16 * ((m_ + 31) // 32) * 64 + 255) // 256
            self._image_of[(v.storage_offset(), sm_ == 1)] = (self.images[ioff:ioff + sizes[j]], self.img_inv[j:j + 1], v.numel())
            ioff += sizes[j]
        self.stab = torch.from_numpy(stab.view(np.uint8).copy()).to(dev)
        self.sblk0 = torch.from_numpy(sblk0).to(dev)
        self.snblk = blk

    @torch.no_grad()
    def prepare(self):
        from .. import hip
        if self._candidates is None:   # the module tree does not change after construction: walk it once, not every step
            self._candidates = [m for m in self.model.modules() if isinstance(m, ConvBN) and (m._pointwise or m._pointwise_s2)]
        mods = [m for m in self._candidates if m.weight.requires_grad and m.weight.is_cuda]
        if not mods:
            return
        key = tuple((id(m), m.weight.data_ptr(), id(m.norm.scale_shift()[0])) for m in mods)
        if key != self._key:
            self._build(mods)
            self._key = key
        lib = hip.load()
        self.amax.zero_()
        hip.check(lib.lgd_scale_rows_multi(hip.ptr(self.tab), hip.ptr(self.blk0), len(self.mods), self.nblk, hip.ptr(self.amax), hip.stream_ptr()),
                  "lgd_scale_rows_multi")
        # the views were rewritten behind autograd's back (a raw kernel on their storage): bump their version counters, so that a graph that
        # saved LAST step's folds (gradient accumulation, an evaluation forward between step and backward) raises instead of running its
        # backward with this step's filters
        torch.autograd.graph.increment_version(self.flat)   # (one bump: the views share their base's version counter -- ADVICE r4)
        self.flat._lgd_w_amax_table = (self._amax_of, self.flat._version)
        if _STEP_IMAGES:
            hip.check(lib.lgd_gemm2h_split_multi(hip.ptr(self.stab), hip.ptr(self.sblk0), len(self._image_of), self.snblk, hip.stream_ptr()),
                      "lgd_gemm2h_split_multi")
            self.flat._lgd_w_img_table = (self._image_of, self.flat._version)
        for m, v, sc in zip(self.mods, self.views, self.scales):
            m._step_fold = ((m.weight._version, id(sc)), v)
