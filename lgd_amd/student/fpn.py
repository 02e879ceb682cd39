"""FPN with LastLevelP6P7, detectron2 names (`fpn_lateral3`, `fpn_output5`, `top_block.p6`)
[d2-memory: detectron2/modeling/backbone/fpn.py @ v0.3].  `forward` takes the bottom-up feature
dict: the reference replaces `fpn.bottom_up` by an identity nn.Sequential() and feeds it the raw
ResNet features (models/customized_detectors/retinanet.py:29-34,52-53)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, streams


# p3's output convolution on the side stream beside the small levels: OFF since all forks share one side stream (round 6,
# profiles/r06_hw_queues_and_forks.txt: alone +0.2 ms at config 2, with the other three forks +0.2 / +0.4 at configs 2 / 3; it was worth
# -0.2 ms on a stream of its own, which the shared stream's robustness is not traded for).  LGD_FPN_STREAM=1 switches it on; under test.
_FPN_STREAM = os.environ.get("LGD_FPN_STREAM", "0") == "1"


class LastLevelP6P7(nn.Module):
    def __init__(self, cin, cout, in_feature="res5"):
        super().__init__()
        self.num_levels = 2
        self.in_feature = in_feature
        self.p6 = nn.Conv2d(cin, cout, 3, 2, 1)
        self.p7 = nn.Conv2d(cout, cout, 3, 2, 1)
        for m in (self.p6, self.p7):
            nn.init.kaiming_uniform_(m.weight, a=1)
            nn.init.constant_(m.bias, 0)

    def forward(self, c5):
        p6 = ops.conv3x3_stride2(c5, self.p6.weight, self.p6.bias)
        return [p6, ops.conv3x3_stride2(F.relu(p6), self.p7.weight, self.p7.bias)]


class FPN(nn.Module):
    def __init__(self, bottom_up, in_features, in_channels, out_channels=256, top_block=None):
        super().__init__()
        self.bottom_up = bottom_up
        self.in_features = tuple(in_features)
        self.stages = []
        for f, c in zip(in_features, in_channels):
            idx = int(f[3:]) if f.startswith("res") else int(f[-1])
            lat = nn.Conv2d(c, out_channels, 1)
            out = ops.Conv3x3(out_channels, out_channels)
            for m in (lat, out):
                nn.init.kaiming_uniform_(m.weight, a=1)
                nn.init.constant_(m.bias, 0)
            self.add_module("fpn_lateral%d" % idx, lat)
            self.add_module("fpn_output%d" % idx, out)
            self.stages.append(idx)
        self.top_block = top_block
        self.size_divisibility = 32
        self.out_features = ["p%d" % i for i in self.stages] + \
            (["p%d" % (self.stages[-1] + 1 + k) for k in range(top_block.num_levels)] if top_block else [])

    def forward(self, x):
        feats = self.bottom_up(x)
        # top-down path first: lateral 1x1 convolutions + the upsampled sum, smallest level first
        lats, prev = {}, None
        for f, idx in zip(reversed(self.in_features), reversed(self.stages)):
            m, x = getattr(self, "fpn_lateral%d" % idx), feats[f]
            up = F.interpolate(prev, scale_factor=2.0, mode="nearest") if prev is not None else None
            if x.is_cuda and x.dtype == torch.float32:
                # lateral 1x1 conv as a GEMM whose weight gradient runs as per-image NT GEMMs on the NCHW maps (ops.conv1x1: the
                # library's implicit-GEMM weight gradient transposes both operands to NHWC first, 0.9 ms/step for the three laterals
                # at config 2), bias + top-down sum in ONE pass over the GEMM output (ops.bias_act with the upsampled map as residual)
                lat = ops.bias_act(ops.conv1x1(x, m.weight), m.bias, up, relu=False)
            else:
                lat = m(x)
                if up is not None:
                    lat = lat + up
            lats[idx] = prev = lat
        # The output convolution of the LARGEST level (p3: 3/4 of the pyramid's pixels, this library's kernels only) goes to a second stream, where
        # it runs beside the small levels' output convolutions and the extra levels p6 / p7 (lgd_amd/streams.py; round 5, the roles the other way
        # round, same call at config 2: 51.17 / 50.92 -> 50.73 / 50.50 ms).  Round 6 turned the roles round: the small levels and p6 / p7 are problems
        # under the size gates -- calls of the vendor library -- and a side stream carries this library's kernels only (ops.convs_on_own_kernels;
        # the root cause of round 5's stall: streams.library_call).  Opt-in since round 6: see _FPN_STREAM.
        big = self.stages[0]
        out_big = getattr(self, "fpn_output%d" % big)
        two = (_FPN_STREAM and ops.side_streams_ok() and len(self.stages) > 1 and lats[big].is_cuda and lats[big].dtype == torch.float32
               and ops.convs_on_own_kernels([lats[big]], [[out_big.weight]]))
        results = {}
        forked = None
        if two:
            forked = streams.fork(lats[big].device, "fpn", inputs=[lats[big]])
            streams.join_on_grad(list(out_big.parameters()), "fpn")
            with torch.cuda.stream(forked[1]):
                results[big] = out_big(lats[big])
        for idx in reversed(self.stages):
            if idx not in results:
                results[idx] = getattr(self, "fpn_output%d" % idx)(lats[idx])
        results = [results[idx] for idx in self.stages]
        if self.top_block is not None:
            results.extend(self.top_block(feats[self.top_block.in_feature]))
        if forked is not None:
            streams.join(forked[0], forked[1], outputs=[results[0]])
        return dict(zip(self.out_features, results))
