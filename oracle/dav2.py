"""Oracle (test infrastructure): Depth-Anything-V2 forward, restated functionally.

Follows external/depth_anything_v2/dpt.py:182-203 (DepthAnythingV2.forward),
dinov2.py:179-231,271-321 (token preparation, pos-embed interpolation, block loop,
intermediate taps), dinov2_layers/{attention.py:49-62, mlp.py:35-41, block.py:82-107,
layer_scale.py:27, patch_embed.py:69-82}, dpt.py:116-150 (DPTHead.forward) and
util/blocks.py:57-80,123-148 (ResidualConvUnit / FeatureFusionBlock).
Weights come from a flat state dict with the reference's parameter names.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .ops import bilinear_ac

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def interpolate_pos_encoding(pos_embed: torch.Tensor, npatch: int, w: int, h: int, patch: int,
                             offset: float = 0.1) -> torch.Tensor:
    """dinov2.py:179-210.  NB the caller passes (H, W) under the names (w, h) (dinov2.py:213)."""
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    class_pos = pos_embed[:, 0]
    patch_pos = pos_embed[:, 1:]
    dim = pos_embed.shape[-1]
    w0 = w // patch + offset
    h0 = h // patch + offset
    sqrt_n = math.sqrt(N)
    sx, sy = float(w0) / sqrt_n, float(h0) / sqrt_n
    patch_pos = F.interpolate(
        patch_pos.reshape(1, int(sqrt_n), int(sqrt_n), dim).permute(0, 3, 1, 2),
        scale_factor=(sx, sy), mode="bicubic", antialias=False)
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((class_pos.unsqueeze(0), patch_pos), dim=1)


def attention(sd, p, x, heads):
    """attention.py:49-62: q pre-scaled, softmax over keys, no mask."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((C // heads) ** -0.5), qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])


def block(sd, p, x, heads):
    """block.py:82-107 (eval branch) with LayerScale (layer_scale.py:27)."""
    D = x.shape[-1]
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    x = x + attention(sd, p + "attn.", h, heads) * sd[p + "ls1.gamma"]
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + h * sd[p + "ls2.gamma"]


def dinov2_intermediate(sd, p, x, vit):
    """get_intermediate_layers(x, taps, return_class_token=True) (dinov2.py:297-321)."""
    B, _, H, W = x.shape
    patch = vit["patch"]
    t = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd[p + "cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + interpolate_pos_encoding(sd[p + "pos_embed"], t.shape[1] - 1, H, W, patch)
    outs = []
    for i in range(vit["depth"]):
        t = block(sd, f"{p}blocks.{i}.", t, vit["heads"])
        if i in vit["taps"]:
            outs.append(t)
    D = t.shape[-1]
    outs = [F.layer_norm(o, (D,), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6) for o in outs]
    return [(o[:, 1:], o[:, 0]) for o in outs]


def residual_conv_unit(sd, p, x):
    """util/blocks.py:57-80 (bn=False)."""
    out = F.relu(x)
    out = F.conv2d(out, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    out = F.relu(out)
    out = F.conv2d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return out + x


def feature_fusion_block(sd, p, xs, size=None):
    """util/blocks.py:123-148."""
    out = xs[0]
    if len(xs) == 2:
        out = out + residual_conv_unit(sd, p + "resConfUnit1.", xs[1])
    out = residual_conv_unit(sd, p + "resConfUnit2.", out)
    if size is None:
        size = (out.shape[-2] * 2, out.shape[-1] * 2)
    out = bilinear_ac(out, size)
    return F.conv2d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


def dpt_head(sd, p, feats, ph, pw):
    """dpt.py:116-150 (use_clstoken=False)."""
    out = []
    for i, (x, _cls) in enumerate(feats):
        x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], ph, pw)
        x = F.conv2d(x, sd[f"{p}projects.{i}.weight"], sd[f"{p}projects.{i}.bias"])
        if i == 0:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.0.weight"], sd[p + "resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, sd[p + "resize_layers.1.weight"], sd[p + "resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = F.conv2d(x, sd[p + "resize_layers.3.weight"], sd[p + "resize_layers.3.bias"], stride=2, padding=1)
        out.append(x)
    l1, l2, l3, l4 = out
    s = p + "scratch."
    l1rn = F.conv2d(l1, sd[s + "layer1_rn.weight"], None, padding=1)
    l2rn = F.conv2d(l2, sd[s + "layer2_rn.weight"], None, padding=1)
    l3rn = F.conv2d(l3, sd[s + "layer3_rn.weight"], None, padding=1)
    l4rn = F.conv2d(l4, sd[s + "layer4_rn.weight"], None, padding=1)
    path4 = feature_fusion_block(sd, s + "refinenet4.", [l4rn], size=l3rn.shape[2:])
    path3 = feature_fusion_block(sd, s + "refinenet3.", [path4, l3rn], size=l2rn.shape[2:])
    path2 = feature_fusion_block(sd, s + "refinenet2.", [path3, l2rn], size=l1rn.shape[2:])
    path1 = feature_fusion_block(sd, s + "refinenet1.", [path2, l1rn])
    o = F.conv2d(path1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
    out_feat = bilinear_ac(o, (int(ph * 14), int(pw * 14)))
    o = F.conv2d(out_feat, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1)
    o = F.relu(o)
    o = F.conv2d(o, sd[s + "output_conv2.2.weight"], sd[s + "output_conv2.2.bias"])
    o = torch.sigmoid(o)
    return o, [l4rn, path4, path3, path2, path1, out_feat]


def dav2_forward(sd, prefix, x, cfg):
    """DepthAnythingV2.forward (dpt.py:182-203).  Returns dict(metric_depth, temp_features)."""
    mean = torch.tensor(IMAGENET_MEAN, dtype=x.dtype).view(-1, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=x.dtype).view(-1, 1, 1)
    x = (x - mean) / std
    ph, pw = x.shape[-2] // 14, x.shape[-1] // 14
    feats = dinov2_intermediate(sd, prefix + "pretrained.", x, cfg["vit"])
    depth, f = dpt_head(sd, prefix + "depth_head.", feats, ph, pw)
    depth = depth * cfg["max_depth"]
    return dict(
        metric_depth=depth,
        temp_features=dict(x_d0=f[0], x_blocks_feat_0=f[1], x_blocks_feat_1=f[2],
                           x_blocks_feat_2=f[3], x_blocks_feat_3=f[4], midas_final_feat=f[5]))


def coarse_features(out):
    """patchrefinerplus.py:226-237: low -> high resolution pyramid."""
    t = out["temp_features"]
    return [t["x_d0"], t["x_blocks_feat_0"], t["x_blocks_feat_1"], t["x_blocks_feat_2"],
            t["x_blocks_feat_3"], t["midas_final_feat"]], out["metric_depth"]
