"""Oracle (test infrastructure): the MiDaS v3.1 ``DPT_BEiT_L_384`` core of ``type='ZoeDepth'`` with the MidasCore hooks.

The reference fetches this network with ``torch.hub.load("AyaanShah2204/MiDaS", "DPT_BEiT_L_384")``
(external/zoedepth/models/base_models/midas.py:342-347) over timm's ``beit_large_patch16_384`` -- neither is vendored,
so the arithmetic below restates the PUBLISHED MiDaS 3.1 code (midas/dpt_model.py DPT / DPTDepthModel, midas/blocks.py
FeatureFusionBlock_custom / ResidualConvUnit_custom / _make_scratch, midas/backbones/{utils.py, beit.py}):
  * BEiT-L/16: cls token, no absolute position embedding, 24 pre-LN blocks with layer scale (gamma_1 / gamma_2),
    qkv bias = [q_bias, 0, v_bias], per-block relative position bias whose 47 x 47 table is bilinearly resized to the
    input's window (beit.py::_get_rel_pos_bias, incl. its (old_width, old_height) reshape), LayerNorm eps 1e-6, GELU MLP;
  * taps after blocks 5 / 11 / 17 / 23 (hooks, no final norm), 'project' readout (Linear(2D -> D) + GELU on
    [token | cls]), act_postprocess 1..4 (1x1 conv; ConvT 4x4 s4 / ConvT 2x2 s2 / - / 3x3 s2);
  * scratch: layer{1..4}_rn (3x3, no bias), refinenet4..1 (resConfUnit1/2, bilinear align_corners=True to the next level's
    size, out_conv 1x1), output_conv = conv3x3(256->128), x2 bilinear(align_corners=True), conv3x3(128->32), ReLU,
    conv1x1(32->1), ReLU.
What the reference itself fixes and this file follows: PrepForMidas = (x - 0.5) / 0.5 (midas.py:176-188), the six hooked
tensors ('out_conv' = output_conv[3], 'l4_rn', 'r4'..'r1': midas.py:296-318), rel_depth = the squeezed network output.

PINNED THROUGH A SECOND IMPLEMENTATION: HuggingFace ``transformers`` carries an independent port of exactly this network
(DPTForDepthEstimation over a BeitBackbone, converted from the MiDaS 3.1 checkpoints); oracle/make_golden.py::g_midas_beit
puts it behind a ``torch.hub.load`` stand-in, runs the REFERENCE's own MidasCore + ZoeDepth classes over it with
synthetic weights mapped by name and requires oracle == that.  Unpinned: timm's / MiDaS's state-dict key names.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import dav2
from .ops import bilinear_ac


def relative_position_index(wh: int, ww: int) -> torch.Tensor:
    """beit.py::gen_relative_position_index: [wh*ww+1, wh*ww+1] indices into the (2wh-1)(2ww-1)+3 table"""
    nrd = (2 * wh - 1) * (2 * ww - 1) + 3
    area = wh * ww
    coords = torch.stack(torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wh - 1
    rel[:, :, 1] += ww - 1
    rel[:, :, 0] *= 2 * ww - 1
    idx = torch.zeros((area + 1, area + 1), dtype=rel.dtype)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = nrd - 3
    idx[0:, 0] = nrd - 2
    idx[0, 0] = nrd - 1
    return idx


def rel_pos_bias(table: torch.Tensor, old_window, window) -> torch.Tensor:
    """beit.py::_get_rel_pos_bias -> [heads, N, N] for an input of window = (h // 16, w // 16) patches"""
    oh, ow = 2 * old_window[0] - 1, 2 * old_window[1] - 1
    nh, nw = 2 * window[0] - 1, 2 * window[1] - 1
    n_old = oh * ow + 3
    sub = table[:n_old - 3].reshape(1, ow, oh, -1).permute(0, 3, 1, 2)
    new = F.interpolate(sub, size=(nh, nw), mode="bilinear")
    new = new.permute(0, 2, 3, 1).reshape(nh * nw, -1)
    full = torch.cat([new, table[n_old - 3:]])
    n = window[0] * window[1] + 1
    return full[relative_position_index(*window).view(-1)].view(n, n, -1).permute(2, 0, 1).contiguous()


def beit_block(sd, p, x, heads, bias):
    D = x.shape[-1]
    B, N, _ = x.shape
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv_bias = torch.cat((sd[p + "attn.q_bias"], torch.zeros_like(sd[p + "attn.v_bias"]), sd[p + "attn.v_bias"]))
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], qkv_bias).reshape(B, N, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((D // heads) ** -0.5), qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1) + bias.unsqueeze(0)).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, D)
    x = x + sd[p + "gamma_1"] * F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + sd[p + "gamma_2"] * h


def midas_beit_core(sd, p, x, cfg):
    """MidasCore.forward(x, return_rel_depth=True) with do_resize=False.  x: [B,3,H,W] in [0,1], H and W multiples of 32.
    Returns rel_depth [B,H,W] and [out_conv, l4_rn, r4, r3, r2, r1]."""
    b = cfg["beit"]
    x = (x - 0.5) / 0.5
    B, _, H, Wd = x.shape
    gh, gw = H // b["patch"], Wd // b["patch"]
    m = p + "pretrained.model."
    t = F.conv2d(x, sd[m + "patch_embed.proj.weight"], sd[m + "patch_embed.proj.bias"], stride=b["patch"]).flatten(2).transpose(1, 2)
    t = torch.cat((sd[m + "cls_token"].expand(B, -1, -1), t), dim=1)
    taps = []
    for i in range(b["depth"]):
        bp = f"{m}blocks.{i}."
        bias = rel_pos_bias(sd[bp + "attn.relative_position_bias_table"], b["window"], (gh, gw))
        t = beit_block(sd, bp, t, b["heads"], bias)
        if i in b["taps"]:
            taps.append(t)
    layers = []
    for i, tok in enumerate(taps):
        a = f"{p}pretrained.act_postprocess{i + 1}."
        readout = tok[:, 0].unsqueeze(1).expand_as(tok[:, 1:])
        f = F.gelu(F.linear(torch.cat((tok[:, 1:], readout), -1), sd[a + "0.project.0.weight"], sd[a + "0.project.0.bias"]))
        f = f.transpose(1, 2).reshape(B, -1, gh, gw)
        f = F.conv2d(f, sd[a + "3.weight"], sd[a + "3.bias"])
        if i == 0:
            f = F.conv_transpose2d(f, sd[a + "4.weight"], sd[a + "4.bias"], stride=4)
        elif i == 1:
            f = F.conv_transpose2d(f, sd[a + "4.weight"], sd[a + "4.bias"], stride=2)
        elif i == 3:
            f = F.conv2d(f, sd[a + "4.weight"], sd[a + "4.bias"], stride=2, padding=1)
        layers.append(f)
    s = p + "scratch."
    rn = [F.conv2d(layers[i], sd[f"{s}layer{i + 1}_rn.weight"], None, padding=1) for i in range(4)]
    r4 = dav2.feature_fusion_block(sd, s + "refinenet4.", [rn[3]], size=rn[2].shape[2:])
    r3 = dav2.feature_fusion_block(sd, s + "refinenet3.", [r4, rn[2]], size=rn[1].shape[2:])
    r2 = dav2.feature_fusion_block(sd, s + "refinenet2.", [r3, rn[1]], size=rn[0].shape[2:])
    r1 = dav2.feature_fusion_block(sd, s + "refinenet1.", [r2, rn[0]])
    o = F.conv2d(r1, sd[s + "output_conv.0.weight"], sd[s + "output_conv.0.bias"], padding=1)
    o = bilinear_ac(o, (o.shape[-2] * 2, o.shape[-1] * 2))
    out_conv = F.relu(F.conv2d(o, sd[s + "output_conv.2.weight"], sd[s + "output_conv.2.bias"], padding=1))
    d = F.relu(F.conv2d(out_conv, sd[s + "output_conv.4.weight"], sd[s + "output_conv.4.bias"]))
    return d.squeeze(1), [out_conv, rn[3], r4, r3, r2, r1]
