"""Oracle (test infrastructure): LightWeightRefiner with the ConvNeXt-L encoder (v2_convx_u4k config).

LightWeightRefiner.forward, convnext branch   estimator/models/blocks/lightweight_refiner.py:277-283,307-313
4-channel stem surgery                        estimator/models/patchrefinerplus.py:194-200
The encoder is timm's ``convnext_large`` (features_only) -- timm is NOT in the reference tree nor installed.
The block arithmetic is pinned against HuggingFace ``transformers``' ConvNext, an independent implementation of
the same published architecture (tests/golden/convnext_tiny.npz, made by oracle/make_golden.py::g_convnext);
the timm state-dict key names are unpinned except ``stem_0`` (used by the reference's stem surgery).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from patchrefinerv2_amd.weights import CONVNEXT_LARGE

from .ops import bilinear_ac

LN_EPS = 1e-6


def _ln_cf(x, w, b):  # LayerNorm2d: over channels of an NCHW map
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, LN_EPS).permute(0, 3, 1, 2)


def convnext_block(sd, p, x):
    c = x.shape[1]
    t = F.conv2d(x, sd[p + "conv_dw.weight"], sd[p + "conv_dw.bias"], padding=3, groups=c)
    t = t.permute(0, 2, 3, 1)
    t = F.layer_norm(t, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], LN_EPS)
    t = F.linear(t, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    t = F.gelu(t)
    t = F.linear(t, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    t = t * sd[p + "gamma"]
    return x + t.permute(0, 3, 1, 2)


def convnext_features(sd, p, x, arch=CONVNEXT_LARGE):
    x = F.conv2d(x, sd[p + "stem_0.weight"], sd[p + "stem_0.bias"], stride=4)
    x = _ln_cf(x, sd[p + "stem_1.weight"], sd[p + "stem_1.bias"])
    feats = []
    for i, n in enumerate(arch["depths"]):
        st = f"{p}stages_{i}."
        if i > 0:
            x = _ln_cf(x, sd[st + "downsample.0.weight"], sd[st + "downsample.0.bias"])
            x = F.conv2d(x, sd[st + "downsample.1.weight"], sd[st + "downsample.1.bias"], stride=2)
        for j in range(n):
            x = convnext_block(sd, f"{st}blocks.{j}.", x)
        feats.append(x)
    return feats


def lightweight_refiner_convnext(sd, p, crop_image, coarse_depth, arch=CONVNEXT_LARGE):
    """Reference order ``refiner_features[::-1]``: low -> high resolution; the two prepended maps are
    relu(upsample_convx(map0)) (stride 2) and its 2x bilinear copy (stride 1); out_depth = 0."""
    mean = torch.tensor(arch["mean"], dtype=crop_image.dtype).view(-1, 1, 1)
    std = torch.tensor(arch["std"], dtype=crop_image.dtype).view(-1, 1, 1)
    x = (crop_image - mean) / std
    feats = convnext_features(sd, p + "refiner_encoder.", torch.cat([x, coarse_depth], dim=1), arch)
    up = F.relu(F.conv_transpose2d(feats[0], sd[p + "upsample_convx.0.weight"], sd[p + "upsample_convx.0.bias"], stride=2))
    up2 = bilinear_ac(up, (up.shape[-2] * 2, up.shape[-1] * 2))
    feats = [up2, up] + feats
    return feats[::-1], torch.zeros_like(crop_image[:, :1])
