"""Oracle (test infrastructure): LightWeightRefiner with the MobileNetV4-conv-small encoder.

LightWeightRefiner.forward   estimator/models/blocks/lightweight_refiner.py:285-322
4-channel stem surgery       estimator/models/patchrefinerplus.py:159-165
The encoder itself is timm's ``mobilenetv4_conv_small`` (features_only) -- timm is NOT in
the reference tree nor installed: PARITY UNPINNED, restated from the public MobileNetV4
definition (see patchrefinerv2_amd/weights.py::MNV4_SMALL).  BatchNorm in eval mode.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from patchrefinerv2_amd.weights import MNV4_SMALL, mnv4_layers

from .ops import bilinear_ac

BN_EPS = 1e-5


def mnv4_features(sd, p, x, arch=MNV4_SMALL):
    layers, taps = mnv4_layers(arch, in_chans=x.shape[1])
    feats = []
    skip = None
    for i, L in enumerate(layers):
        if L.get("res_begin"):
            skip = x
        x = F.conv2d(x, sd[p + L["conv"] + ".weight"], None, stride=L["s"], padding=L["k"] // 2, groups=L["g"])
        b = p + L["bn"] + "."
        x = F.batch_norm(x, sd[b + "running_mean"], sd[b + "running_var"], sd[b + "weight"], sd[b + "bias"],
                         False, 0.0, BN_EPS)
        if L["act"]:
            x = F.relu(x)
        if L.get("res_end"):
            x = x + skip
            skip = None
        if i in taps:
            feats.append(x)
    return feats


def lightweight_refiner(sd, p, crop_image, coarse_depth, arch=MNV4_SMALL):
    """Returns (feats high -> low ... wait: reference returns ``refiner_features[::-1]`` =
    low -> high resolution with the 2x-upsampled copy of map 0 LAST) and out_depth = 0."""
    mean = torch.tensor(arch["mean"], dtype=crop_image.dtype).view(-1, 1, 1)
    std = torch.tensor(arch["std"], dtype=crop_image.dtype).view(-1, 1, 1)
    x = (crop_image - mean) / std
    feats = mnv4_features(sd, p + "refiner_encoder.", torch.cat([x, coarse_depth], dim=1), arch)
    hi = feats[0]
    up = bilinear_ac(hi, (hi.shape[-2] * 2, hi.shape[-1] * 2))  # scale_factor=2 (lightweight_refiner.py:316)
    feats = [up] + feats
    return feats[::-1], torch.zeros_like(crop_image[:, :1])
