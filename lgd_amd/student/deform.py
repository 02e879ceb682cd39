"""Modulated deformable convolution (DCNv2) for BASELINE config 5 (RetinaNet R-101-DCNv2)
[ref: configs/Distillation/RetinaNet/retinanet_R_101_dcnv2_*.yaml:7-8 DEFORM_ON_PER_STAGE / DEFORM_MODULATED;
detectron2's ModulatedDeformConv CUDA op is not available in this environment].

    out[n, o, y, x] = sum_{c, k} W[o, c, k] * mask[n, k, y, x] * bilinear(in[n, c], y*s - p + ky*d + dy_k, x*s - p + kx*d + dx_k)
with zero padding outside the input, offsets stored as (dy, dx) channel pairs per tap k = ky*3 + kx ([d2-memory], SURVEY.md
appendix A).  One path: `ops.deform_conv3x3` (HIP gather kernel -> column matrix -> library GEMM; backward kernel for dx /
d offset / d mask).  The definition-level restatement it is tested against lives in oracle/student_oracle.py."""
from .. import ops


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1):
    """x (N,C,H,W); offset (N, 18, Ho, Wo), channel 2k = dy_k, 2k+1 = dx_k; mask (N, 9, Ho, Wo) or None; weight (O, C, 3, 3)."""
    if tuple(weight.shape[2:]) != (3, 3):
        raise ValueError("deformable convolution: only 3x3 filters are used by the shipped configs, got %s" % (tuple(weight.shape[2:]),))
    return ops.deform_conv3x3(x, offset, mask, weight, bias, stride, padding, dilation)
