"""FPN with LastLevelP6P7, detectron2 names (`fpn_lateral3`, `fpn_output5`, `top_block.p6`)
[d2-memory: detectron2/modeling/backbone/fpn.py @ v0.3].  `forward` takes the bottom-up feature
dict: the reference replaces `fpn.bottom_up` by an identity nn.Sequential() and feeds it the raw
ResNet features (models/customized_detectors/retinanet.py:29-34,52-53)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, streams


_FPN_STREAM = os.environ.get("LGD_FPN_STREAM", "1") != "0"   # 0: the whole FPN on one stream (A/B runs)


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
        results = []
        prev = None
        # the extra levels (p6 / p7) and the output convolutions of the small levels do not feed the top-down path: on a second stream they run beside
        # the laterals and the large level's output convolution (lgd_amd/streams.py; same call, config 2: 51.17 / 50.92 -> 50.73 / 50.50 ms)
        two = _FPN_STREAM and ops.side_streams_ok() and all(v.is_cuda and v.dtype == torch.float32 for v in feats.values())
        top = None
        forked = None
        if two and self.top_block is not None and self.top_block.in_feature in feats:
            src = feats[self.top_block.in_feature]
            forked = streams.fork(src.device, "fpn", inputs=[src])
            streams.join_on_grad(list(self.top_block.parameters()), "fpn")
            with torch.cuda.stream(forked[1]):
                top = self.top_block(src)
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
            prev = lat
            out = getattr(self, "fpn_output%d" % idx)
            if two and idx != self.stages[0]:   # (every level but the largest)
                if forked is None:
                    forked = streams.fork(prev.device, "fpn", inputs=[prev])
                else:
                    forked[1].wait_stream(forked[0])
                    prev.record_stream(forked[1])
                streams.join_on_grad(list(out.parameters()), "fpn")
                with torch.cuda.stream(forked[1]):
                    results.insert(0, out(prev))
            else:
                results.insert(0, out(prev))
        if self.top_block is not None:
            results.extend(top if top is not None else self.top_block(feats[self.top_block.in_feature]))
        if forked is not None:
            streams.join(forked[0], forked[1], outputs=results[1:])
        return dict(zip(self.out_features, results))
