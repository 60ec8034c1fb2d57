"""Oracle (test infrastructure): the per-patch fusion networks, restated functionally.

FusionUnet            estimator/models/blocks/fusion_model.py:84-122
UpSample.forward_hardcode / DoubleConv   fusion_model.py:15-24, convs.py:31-45
SingleConvCNNLN + channels-first LayerNorm   convs.py:21-29,64-75
BiDirectionalFusion   estimator/models/blocks/bi_directional_fusion_model.py:379-446
C2FModule             bi_directional_fusion_model.py:184-208
GatedFusionBlock / GatedConvUnit   bi_directional_fusion_model.py:116-146, 56-82
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .ops import bilinear_ac

TRACE = None  # tests set this to a dict: intermediate maps of the last call (c2f_depth, dec_last, offset)


def ln_cf(x, w, b, eps=1e-6):
    """channels-first LayerNorm (convs.py:24-29)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def single_conv_ln(sd, p, x):
    """SingleConvCNNLN (convs.py:67-72): conv -> LN -> GELU; with the parameters of SingleConvCNNLNHeavy (BiDirectionalFusionHeavy,
    bi_directional_fusion_model.py:449-463): conv -> LN -> conv -> LN -> conv -> GELU (no activation behind the LayerNorms)."""
    x = F.conv2d(x, sd[p + "single_conv.0.weight"], None, padding=1)
    x = ln_cf(x, sd[p + "single_conv.1.weight"], sd[p + "single_conv.1.bias"])
    if p + "single_conv.4.weight" in sd:
        x = F.conv2d(x, sd[p + "single_conv.2.weight"], None, padding=1)
        x = ln_cf(x, sd[p + "single_conv.3.weight"], sd[p + "single_conv.3.bias"])
        x = F.conv2d(x, sd[p + "single_conv.4.weight"], None, padding=1)
    return F.gelu(x)


def double_conv(sd, p, x):
    """DoubleConv (convs.py:31-45): two conv + GELU; DoubleConvHeavy (bi_directional_fusion_model.py:465-485): five."""
    for i in (0, 2, 4, 6, 8):
        if p + f"double_conv.{i}.weight" in sd:
            x = F.gelu(F.conv2d(x, sd[p + f"double_conv.{i}.weight"], None, padding=1))
    return x


def upsample_hardcode(sd, p, x1, x2, pred1, pred2):
    size = x2.shape[-2:]
    x = torch.cat([bilinear_ac(x1, size), x2, bilinear_ac(pred1, size), bilinear_ac(pred2, size)], dim=1)
    return double_conv(sd, p + "conv.", x)


def _encode_decode(sd, p, enc1, enc2, dec, c_feat, f_feat, pred1, pred2, update_base):
    temp = []
    for idx, (c, f) in enumerate(zip(c_feat, f_feat)):
        f = single_conv_ln(sd, f"{p}{enc1}.{idx}.", torch.cat([c, f], dim=1))
        size = f.shape[-2:]
        f = torch.cat([f, bilinear_ac(pred1, size), bilinear_ac(pred2, size)], dim=1)
        temp.append(single_conv_ln(sd, f"{p}{enc2}.{idx}.", f))
    dec_feat = temp[0]
    temp = temp[::-1]
    _feat = temp[0]
    for j, feat in enumerate(temp[1:]):
        if f"{p}{dec}.{j}.conv.double_conv.0.weight" not in sd:
            break
        dec_feat = upsample_hardcode(sd, f"{p}{dec}.{j}.", _feat, feat, pred1, pred2)
        _feat = dec_feat
    off = F.conv2d(dec_feat, sd[p + "final_conv.weight"], None, padding=1)
    if TRACE is not None:
        TRACE["dec_last"], TRACE["offset"] = dec_feat, off
    if update_base is not None:
        return torch.clamp(update_base + off, min=0)
    return off


def fusion_unet(sd, p, c_feat, f_feat, pred1, pred2, update_base=None):
    """FusionUnet.forward (fusion_model.py:84-122).  c_feat / f_feat: high -> low resolution."""
    return _encode_decode(sd, p, "encoder_layers_1", "encoder_layers_2", "decoder_layers",
                          c_feat, f_feat, pred1, pred2, update_base)


def gated_conv_unit(sd, p, x, c_feat, fusion=True, gate=True):
    """GatedConvUnit.forward (:56-82): ``fusion=False`` ('self-agg') stops after the residual conv; ``gate=False`` ('coarse-fusion')
    returns the fusion_conv output itself instead of gating ``out`` with its sigmoid (:76-80)."""
    out = F.conv2d(F.relu(x), sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1) + x
    if not fusion:
        return out
    fused = torch.cat([out, c_feat], dim=1)
    fused = F.conv2d(fused, sd[p + "fusion_conv.0.weight"], sd[p + "fusion_conv.0.bias"], padding=1)
    fused = F.relu(ln_cf(fused, sd[p + "fusion_conv.1.weight"], sd[p + "fusion_conv.1.bias"]))
    fused = F.conv2d(fused, sd[p + "fusion_conv.3.weight"], None)
    return out * torch.sigmoid(fused) if gate else fused


def gated_fusion_block(sd, p, xs, coarse_feat, size=None, upscale=True, fusion=True, gate=True):
    out = xs[0]
    if len(xs) == 2:
        out = out + gated_conv_unit(sd, p + "GateresConfUnit1.", xs[1], coarse_feat, fusion, gate)
    out = gated_conv_unit(sd, p + "GateresConfUnit2.", out, coarse_feat, fusion, gate)
    if upscale:
        if size is None:
            size = (out.shape[-2] * 2, out.shape[-1] * 2)
        out = bilinear_ac(out, size)
    return F.conv2d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


C2F_TYPES = {"coarse-gated": (True, True), "coarse-fusion": (True, False), "self-agg": (False, False)}  # :355-372 -> (fusion, gate)


def c2f_noenc_module(sd, p, fine, coarse):
    """C2FNOENCModule.forward (bi_directional_fusion_model.py:253-286; coarse2fine_type='only-gate', built with fusion=True, gate=False):
    per pyramid level two GatedConvUnits on the projected refiner map, no top-down path; level 0 = ConvTranspose2d(k2, s2) + ReLU +
    conv3x3 of the highest refiner map.  fine: 5 maps, coarse: 6 maps, high -> low.  Returns ([path_5 .. path_0], depth)."""
    s = p + "scratch."
    rn = [F.conv2d(fine[i], sd[f"{s}layer{i + 1}_rn.weight"], None, padding=1) for i in range(5)]
    up = F.relu(F.conv_transpose2d(fine[0], sd[s + "upsample_conv.0.weight"], sd[s + "upsample_conv.0.bias"], stride=2))
    rn0 = F.conv2d(up, sd[s + "upsample_conv.2.weight"], None, padding=1)
    paths = []
    for k, (x, lvl) in enumerate(zip([rn[4], rn[3], rn[2], rn[1], rn[0], rn0], [5, 4, 3, 2, 1, 0]), start=1):  # layer1_gate* <-> level 5
        x = gated_conv_unit(sd, f"{s}layer{k}_gate1.", x, coarse[lvl], True, False)
        paths.append(gated_conv_unit(sd, f"{s}layer{k}_gate2.", x, coarse[lvl], True, False))
    out = F.conv2d(paths[-1], sd[s + "output_conv.weight"], sd[s + "output_conv.bias"], padding=1)
    return paths, out


def c2f_module(sd, p, fine, coarse, fusion=True, gate=True):
    """C2FModule.forward (bi_directional_fusion_model.py:184-208); fine & coarse high -> low."""
    s = p + "scratch."
    kw = dict(fusion=fusion, gate=gate)
    rn = [F.conv2d(fine[i], sd[f"{s}layer{i + 1}_rn.weight"], None, padding=1) for i in range(5)]
    path5 = gated_fusion_block(sd, s + "refinenet5.", [rn[4]], coarse[5], size=rn[3].shape[2:], **kw)
    path4 = gated_fusion_block(sd, s + "refinenet4.", [path5, rn[3]], coarse[4], size=rn[2].shape[2:], **kw)
    path3 = gated_fusion_block(sd, s + "refinenet3.", [path4, rn[2]], coarse[3], size=rn[1].shape[2:], **kw)
    path2 = gated_fusion_block(sd, s + "refinenet2.", [path3, rn[1]], coarse[2], size=rn[0].shape[2:], **kw)
    path1 = gated_fusion_block(sd, s + "refinenet1.", [path2, rn[0]], coarse[1], **kw)
    out = F.conv2d(path1, sd[s + "output_conv1.weight"], sd[s + "output_conv1.bias"], padding=1)
    last = F.relu(F.conv2d(out, sd[s + "output_conv2.0.weight"], sd[s + "output_conv2.0.bias"], padding=1))
    last = gated_fusion_block(sd, s + "output_conv2_fusion.", [last], coarse[0], upscale=False, **kw)
    out = F.conv2d(last, sd[s + "output_conv3.0.weight"], sd[s + "output_conv3.0.bias"])
    return [rn[4], path5, path4, path3, path2, last], out


def bidirectional_fusion(sd, p, c_feat, f_feat, pred1, pred2, update_base=None, coarse2fine_type="coarse-gated", coarse2fine=True):
    """BiDirectionalFusion.forward, coarse2fine_type in C2F_TYPES, glb_att=False
    (bi_directional_fusion_model.py:379-446).  c_feat: 6 maps high -> low; f_feat: 6 maps
    high -> low (index 0 = the 2x-upsampled copy that is dropped at :408)."""
    c_feat = list(c_feat)
    if c_feat[-1].shape[-2:] != f_feat[-1].shape[-2:]:
        c_feat = [bilinear_ac(c, f.shape[-2:]) for c, f in zip(c_feat, f_feat)]
    if coarse2fine:  # (:407-414; without it all six refiner maps and the caller's pred2 go on)
        if coarse2fine_type == "only-gate":
            f_feat, out_depth = c2f_noenc_module(sd, p + "c2f.", list(f_feat[1:]), c_feat)
        else:
            f_feat, out_depth = c2f_module(sd, p + "c2f.", list(f_feat[1:]), c_feat, *C2F_TYPES[coarse2fine_type])
        f_feat, pred2 = f_feat[::-1], out_depth
        if TRACE is not None:
            TRACE["c2f_depth"], TRACE["c2f_last"] = out_depth, f_feat[0]
    return _encode_decode(sd, p, "fusion_layers_1", "fusion_layers_2", "f2r_agg",
                          c_feat, f_feat, pred1, pred2, update_base)
