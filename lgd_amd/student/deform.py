"""Modulated deformable convolution (DCNv2) for BASELINE config 5 (RetinaNet R-101-DCNv2)
[ref: configs/Distillation/RetinaNet/retinanet_R_101_dcnv2_*.yaml:7-8 DEFORM_ON_PER_STAGE / DEFORM_MODULATED;
detectron2's ModulatedDeformConv CUDA op is not available in this environment].

Restated from the public DCNv2 definition ([d2-memory]):
    out[n, o, y, x] = sum_{c, k} W[o, c, k] * mask[n, k, y, x] * bilinear(in[n, c], y*s - p + ky*d + dy_k, x*s - p + kx*d + dx_k)
with zero padding outside the input, offsets stored as (dy, dx) channel pairs per tap k = ky*3 + kx.
On the GPU: `ops.deform_conv3x3` (HIP gather kernel -> column matrix -> library GEMM; backward kernel for dx / d offset /
d mask).  The restatement below (bilinear `grid_sample` per tap + ONE GEMM, autograd backward) is the CPU form the HIP
path is tested against."""
import torch
import torch.nn.functional as F


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1):
    """x (N,C,H,W); offset (N, 2*K, Ho, Wo) with K = kh*kw, channel 2k = dy_k, 2k+1 = dx_k; mask (N, K, Ho, Wo);
    weight (O, C, kh, kw)."""
    if x.is_cuda and x.dtype == torch.float32 and tuple(weight.shape[2:]) == (3, 3):  # fused gather kernel + library GEMM
        from .. import ops
        return ops.deform_conv3x3(x, offset, mask, weight, bias, stride, padding, dilation)
    return modulated_deform_conv2d_torch(x, offset, mask, weight, bias, stride, padding, dilation)


def modulated_deform_conv2d_torch(x, offset, mask, weight, bias=None, stride=1, padding=1, dilation=1):
    """the same operator as elementwise torch ops (any device): what the HIP path is tested against."""
    N, C, H, W = x.shape
    O, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    ys = torch.arange(Ho, device=x.device, dtype=x.dtype) * stride - padding
    xs = torch.arange(Wo, device=x.device, dtype=x.dtype) * stride - padding
    base_y, base_x = torch.meshgrid(ys, xs, indexing="ij")
    cols = []
    for k in range(kh * kw):
        ky, kx = divmod(k, kw)
        py = base_y + ky * dilation + offset[:, 2 * k]        # (N,Ho,Wo) sampling rows
        px = base_x + kx * dilation + offset[:, 2 * k + 1]
        # grid_sample(align_corners=True) maps [-1,1] to pixel centres 0..W-1; out-of-range samples read zeros
        gx = 2.0 * px / max(W - 1, 1) - 1.0
        gy = 2.0 * py / max(H - 1, 1) - 1.0
        s = F.grid_sample(x, torch.stack((gx, gy), -1), mode="bilinear", padding_mode="zeros", align_corners=True)
        cols.append(s * mask[:, k:k + 1])
    col = torch.stack(cols, 2).reshape(N, C * kh * kw, Ho * Wo)  # (N, C*K, HoWo), K fastest within a channel
    out = torch.matmul(weight.reshape(O, C * kh * kw), col).reshape(N, O, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
