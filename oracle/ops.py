"""Oracle (test infrastructure): resampling / mask / ROI primitives of the hot path.

Each function restates one reference call site; see SURVEY.md Appendix C for the
index arithmetic the HIP kernels reproduce.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def bilinear_ac(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(..., mode='bilinear', align_corners=True) -- used by every decoder
    (util/blocks.py:144, fusion_model.py:16-18, bi_directional_fusion_model.py:139,393,421)."""
    return F.interpolate(x, size=tuple(int(s) for s in size), mode="bilinear", align_corners=True)


def nearest(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate default mode (baseline_pretrain.py:210, utils.py:42)."""
    return F.interpolate(x, size=tuple(int(s) for s in size))


def bilinear_ac_explicit(x: torch.Tensor, size) -> torch.Tensor:
    """Index-level restatement of bilinear/align_corners=True in float32
    (the rule the HIP gather kernels implement)."""
    B, C, H, W = x.shape
    oh, ow = int(size[0]), int(size[1])

    def axis(n_in, n_out):
        scale = np.float32(0.0) if n_out <= 1 else np.float32(n_in - 1) / np.float32(n_out - 1)
        src = np.arange(n_out, dtype=np.float32) * scale
        i0 = np.floor(src).astype(np.int64)
        i0 = np.minimum(i0, n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        w1 = (src - i0.astype(np.float32)).astype(np.float32)
        w0 = (np.float32(1.0) - w1).astype(np.float32)
        return (torch.from_numpy(i0), torch.from_numpy(i1), torch.from_numpy(w0), torch.from_numpy(w1))

    y0, y1, wy0, wy1 = axis(H, oh)
    x0, x1, wx0, wx1 = axis(W, ow)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return top * wy0[:, None] + bot * wy1[:, None]


def nearest_explicit(x: torch.Tensor, size) -> torch.Tensor:
    """Legacy 'nearest': src = min(floor(dst * float32(in/out)), in-1) (Appendix C row 2)."""
    B, C, H, W = x.shape
    oh, ow = int(size[0]), int(size[1])

    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        idx = np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64)
        return torch.from_numpy(np.minimum(idx, n_in - 1))

    return x[:, :, axis(H, oh)][:, :, :, axis(W, ow)]


# ---------------------------------------------------------------------------------
# Resize transforms used for image_lr and for every patch crop
# ---------------------------------------------------------------------------------
def resize_da_size(width: int, height: int, multiple_of: int = 14):
    """external/depth_anything/transform.py:56-125 with keep_aspect_ratio=False,
    resize_method='minimal': each side rounded to a multiple of ``multiple_of``."""

    def constrain(v):
        return int(np.round(v / multiple_of) * multiple_of)

    return constrain(width), constrain(height)


def resize_da(x: torch.Tensor, width: int, height: int, multiple_of: int = 14) -> torch.Tensor:
    """ResizeDA.__call__ (external/depth_anything/transform.py:127-129).  The target
    does not depend on the input size when keep_aspect_ratio=False."""
    in_w, in_h = x.shape[-1], x.shape[-2]
    nw, nh = resize_da_size((width / in_w) * in_w, (height / in_h) * in_h, multiple_of)
    return bilinear_ac(x, (nh, nw))


def resize_zoe(x: torch.Tensor) -> torch.Tensor:
    """ResizeZoe.__call__ hard-codes 384x512 (external/zoedepth/models/base_models/midas.py:171-174)."""
    return bilinear_ac(x, (384, 512))


# ---------------------------------------------------------------------------------
# cv2.GaussianBlur restatement + generatemask
# ---------------------------------------------------------------------------------
def gaussian_kernel1d(ksize: int, sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel for ksize > 7: exp(-(i-c)^2 / (2 sigma^2)) normalised (float64
    maths, float32 taps for CV_32F images)."""
    c = (ksize - 1) * 0.5
    i = np.arange(ksize, dtype=np.float64)
    k = np.exp(-((i - c) ** 2) / (2.0 * float(sigma) ** 2))
    k /= k.sum()
    return k.astype(np.float32)


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """Separable Gaussian, BORDER_REFLECT_101 (cv2 default).  Parity unpinned: cv2 is absent;
    on this path the zero band of the box exceeds the half-kernel, so the border mode never
    matters."""
    k = torch.from_numpy(gaussian_kernel1d(ksize, sigma))
    x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))[None, None]
    r = ksize // 2
    x = F.pad(x, (r, r, r, r), mode="reflect")
    x = F.conv2d(x, k.view(1, 1, 1, -1))
    x = F.conv2d(x, k.view(1, 1, -1, 1))
    return x[0, 0].numpy()


def generatemask(size, border: float = 0.1) -> np.ndarray:
    """estimator/models/utils.py:51-60."""
    mask = np.zeros(size, dtype=np.float32)
    sigma = int(size[0] / 16)
    k_size = int(2 * np.ceil(2 * int(size[0] / 16)) + 1)
    mask[int(border * size[0]): size[0] - int(border * size[0]),
         int(border * size[1]): size[1] - int(border * size[1])] = 1
    mask = gaussian_blur(mask, int(k_size), sigma)
    mask = (mask - mask.min()) / (mask.max() - mask.min())
    return mask.astype(np.float32)


# ---------------------------------------------------------------------------------
# torchvision.ops.roi_align(aligned=True, sampling_ratio=-1) restatement
# ---------------------------------------------------------------------------------
def roi_align(feat: torch.Tensor, rois: torch.Tensor, output_size, spatial_scale: float,
              aligned: bool = True) -> torch.Tensor:
    """torchvision 0.16.2 roi_align forward (CPU kernel semantics), float32 throughout.

    feat [N,C,H,W]; rois [K,5] = (batch_idx, x1, y1, x2, y2); returns [K,C,oh,ow].
    Call sites: patchrefinerplus.py:271,276; patchrefiner.py:206,210.  Parity unpinned
    (torchvision absent here); cross-checked against grid_sample in tests."""
    N, C, H, W = feat.shape
    oh, ow = int(output_size[0]), int(output_size[1])
    K = rois.shape[0]
    out = feat.new_zeros((K, C, oh, ow))
    f32 = np.float32
    offset = f32(0.5) if aligned else f32(0.0)
    rois_np = rois.detach().cpu().numpy().astype(np.float32)
    for k in range(K):
        b = int(rois_np[k, 0])
        x1 = rois_np[k, 1] * f32(spatial_scale) - offset
        y1 = rois_np[k, 2] * f32(spatial_scale) - offset
        x2 = rois_np[k, 3] * f32(spatial_scale) - offset
        y2 = rois_np[k, 4] * f32(spatial_scale) - offset
        roi_w, roi_h = f32(x2 - x1), f32(y2 - y1)
        if not aligned:
            roi_w, roi_h = max(roi_w, f32(1.0)), max(roi_h, f32(1.0))
        bin_h, bin_w = f32(roi_h / f32(oh)), f32(roi_w / f32(ow))
        gh = int(math.ceil(float(roi_h) / oh))
        gw = int(math.ceil(float(roi_w) / ow))
        # (no clamp of the grid: torchvision's "when the grid is empty, output zeros" -- a degenerate box, roi_w or roi_h <= 0)
        count = f32(max(gh * gw, 1))
        acc = feat.new_zeros((C, oh, ow))
        ph = np.arange(oh, dtype=np.float32)
        pw = np.arange(ow, dtype=np.float32)
        for iy in range(gh):
            yy = (y1 + ph * bin_h + f32(iy + 0.5) * bin_h / f32(gh)).astype(np.float32)
            for ix in range(gw):
                xx = (x1 + pw * bin_w + f32(ix + 0.5) * bin_w / f32(gw)).astype(np.float32)
                acc += _bilinear_sample(feat[b], yy, xx, H, W)
        out[k] = acc / float(count)
    return out


def _bilinear_sample(f: torch.Tensor, yy: np.ndarray, xx: np.ndarray, H: int, W: int) -> torch.Tensor:
    f32 = np.float32
    vy = ~((yy < -1.0) | (yy > H))
    vx = ~((xx < -1.0) | (xx > W))
    y = np.maximum(yy, f32(0.0))
    x = np.maximum(xx, f32(0.0))
    yl = y.astype(np.int64)
    xl = x.astype(np.int64)
    yh = np.where(yl >= H - 1, H - 1, yl + 1)
    xh = np.where(xl >= W - 1, W - 1, xl + 1)
    y = np.where(yl >= H - 1, (H - 1), y).astype(np.float32)
    x = np.where(xl >= W - 1, (W - 1), x).astype(np.float32)
    yl = np.minimum(yl, H - 1)
    xl = np.minimum(xl, W - 1)
    ly = (y - yl.astype(np.float32)).astype(np.float32)
    lx = (x - xl.astype(np.float32)).astype(np.float32)
    hy, hx = (f32(1.0) - ly).astype(np.float32), (f32(1.0) - lx).astype(np.float32)
    t = torch.from_numpy
    yl_t, yh_t, xl_t, xh_t = t(yl), t(yh), t(xl), t(xh)
    w1 = t(np.outer(hy, hx).astype(np.float32))
    w2 = t(np.outer(hy, lx).astype(np.float32))
    w3 = t(np.outer(ly, hx).astype(np.float32))
    w4 = t(np.outer(ly, lx).astype(np.float32))
    v1 = f[:, yl_t][:, :, xl_t]
    v2 = f[:, yl_t][:, :, xh_t]
    v3 = f[:, yh_t][:, :, xl_t]
    v4 = f[:, yh_t][:, :, xh_t]
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    valid = t(np.outer(vy, vx))
    return val * valid
