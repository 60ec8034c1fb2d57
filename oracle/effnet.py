"""Oracle (test infrastructure): LightWeightRefiner with the EfficientNet-B5 encoder (v2_eff_u4k config).

LightWeightRefiner.forward    estimator/models/blocks/lightweight_refiner.py:285-322 (the non-convnext branch)
4-channel stem surgery         estimator/models/patchrefinerplus.py:152-158 (Conv2dSame 4 -> 48, k3 s2)
The encoder is timm's ``tf_efficientnet_b5_ap`` (features_only) -- timm is NOT in the reference tree nor installed.
The block arithmetic is pinned against HuggingFace ``transformers``' EfficientNet, an independent port of the same
TensorFlow model (tests/golden/effnet_refiner.npz, made by oracle/make_golden.py::g_effnet, even sizes only: its
stride-2 padding is static); the timm state-dict key names are unpinned except ``conv_stem``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from patchrefinerv2_amd.weights import EFFNET_B5, effnet_blocks

from .ops import bilinear_ac


def _bn(sd, b, x, eps):
    return F.batch_norm(x, sd[b + "running_mean"], sd[b + "running_var"], sd[b + "weight"], sd[b + "bias"], False, 0.0, eps)


def conv_same(x, w, stride, groups=1):
    """timm Conv2dSame == TensorFlow 'SAME': out = ceil(in / s); the odd padding pixel goes to the bottom / right"""
    k = w.shape[-1]
    ih, iw = x.shape[-2:]
    ph = max((-(-ih // stride) - 1) * stride + k - ih, 0)
    pw = max((-(-iw // stride) - 1) * stride + k - iw, 0)
    x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return F.conv2d(x, w, None, stride=stride, groups=groups)


def _se(sd, p, x):
    s = x.mean((2, 3), keepdim=True)
    s = F.silu(F.conv2d(s, sd[p + "conv_reduce.weight"], sd[p + "conv_reduce.bias"]))
    s = torch.sigmoid(F.conv2d(s, sd[p + "conv_expand.weight"], sd[p + "conv_expand.bias"]))
    return x * s


def effnet_features(sd, p, x, arch=EFFNET_B5):
    eps = arch["bn_eps"]
    x = F.silu(_bn(sd, p + "bn1.", conv_same(x, sd[p + "conv_stem.weight"], 2), eps))
    feats = []
    for B in effnet_blocks(arch):
        b = p + B["name"]
        h = x
        if B["kind"] == "ds":
            h = F.silu(_bn(sd, b + "bn1.", conv_same(h, sd[b + "conv_dw.weight"], B["s"], groups=B["cin"]), eps))
            h = _se(sd, b + "se.", h)
            h = _bn(sd, b + "bn2.", F.conv2d(h, sd[b + "conv_pw.weight"]), eps)
        else:
            h = F.silu(_bn(sd, b + "bn1.", F.conv2d(h, sd[b + "conv_pw.weight"]), eps))
            h = F.silu(_bn(sd, b + "bn2.", conv_same(h, sd[b + "conv_dw.weight"], B["s"], groups=B["cmid"]), eps))
            h = _se(sd, b + "se.", h)
            h = _bn(sd, b + "bn3.", F.conv2d(h, sd[b + "conv_pwl.weight"]), eps)
        x = x + h if B["res"] else h
        if B["tap"]:
            feats.append(x)
    return feats


def lightweight_refiner_effnet(sd, p, crop_image, coarse_depth, arch=EFFNET_B5):
    """Reference order ``refiner_features[::-1]``: low -> high resolution with the 2x bilinear copy of map 0 last."""
    mean = torch.tensor(arch["mean"], dtype=crop_image.dtype).view(-1, 1, 1)
    std = torch.tensor(arch["std"], dtype=crop_image.dtype).view(-1, 1, 1)
    x = (crop_image - mean) / std
    feats = effnet_features(sd, p + "refiner_encoder.", torch.cat([x, coarse_depth], dim=1), arch)
    hi = feats[0]
    feats = [bilinear_ac(hi, (hi.shape[-2] * 2, hi.shape[-1] * 2))] + feats
    return feats[::-1], torch.zeros_like(crop_image[:, :1])
