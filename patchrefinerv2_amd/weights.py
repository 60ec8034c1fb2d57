"""State-dict contract of the hot path + deterministic synthetic weights.

The reference ships no checkpoints (README.md:14,45), so parity and benchmarks
run on synthetic weights.  What IS a contract is the parameter naming/shape of
the reference modules; the ``*_spec`` functions below
enumerate it so that a real checkpoint, once released, loads by name.

Reference modules mirrored (names + shapes only, no code):
  * DepthAnythingV2          external/depth_anything_v2/dpt.py:153-181
    DinoVisionTransformer    external/depth_anything_v2/dinov2.py:44-176
    DPTHead                  external/depth_anything_v2/dpt.py:38-114
  * FusionUnet               estimator/models/blocks/fusion_model.py:53-82
  * BiDirectionalFusion      estimator/models/blocks/bi_directional_fusion_model.py:290-377
    C2FModule                ...bi_directional_fusion_model.py:148-182
  * LightWeightRefiner       estimator/models/blocks/lightweight_refiner.py:242-283
    (timm ``mobilenetv4_conv_small`` features_only encoder -- un-vendored,
    names follow timm's MobileNetV3Features module tree; parity unpinned)
"""
from __future__ import annotations

import hashlib
import re
from collections import OrderedDict
from typing import Sequence, Tuple

import numpy as np
import torch

Spec = "OrderedDict[str, Tuple[int, ...]]"

VIT_CFG = {
    # encoder: (embed_dim, depth, heads, taps)  dinov2.py:340-395, dpt.py:165-170
    "vits": dict(dim=384, depth=12, heads=6, taps=[2, 5, 8, 11]),
    "vitb": dict(dim=768, depth=12, heads=12, taps=[2, 5, 8, 11]),
    "vitl": dict(dim=1024, depth=24, heads=16, taps=[4, 11, 17, 23]),
}


def vit_cfg(encoder: str, **over) -> dict:
    cfg = dict(VIT_CFG[encoder]) if encoder in VIT_CFG else {}
    cfg.update(over)
    cfg.setdefault("patch", 14)
    cfg.setdefault("img_size", 518)  # DINOv2() builder, dinov2.py:407
    cfg.setdefault("mlp_ratio", 4)
    return cfg


def dinov2_spec(prefix: str, cfg: dict) -> Spec:
    D, L, p = cfg["dim"], cfg["depth"], cfg["patch"]
    n_pos = (cfg["img_size"] // p) ** 2 + 1
    hid = int(D * cfg["mlp_ratio"])
    s: Spec = OrderedDict()
    s[prefix + "cls_token"] = (1, 1, D)
    s[prefix + "pos_embed"] = (1, n_pos, D)
    s[prefix + "mask_token"] = (1, D)
    s[prefix + "patch_embed.proj.weight"] = (D, 3, p, p)
    s[prefix + "patch_embed.proj.bias"] = (D,)
    for i in range(L):
        b = f"{prefix}blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "ls1.gamma"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (hid, D)
        s[b + "mlp.fc1.bias"] = (hid,)
        s[b + "mlp.fc2.weight"] = (D, hid)
        s[b + "mlp.fc2.bias"] = (D,)
        s[b + "ls2.gamma"] = (D,)
    s[prefix + "norm.weight"] = (D,)
    s[prefix + "norm.bias"] = (D,)
    return s


def dpt_head_spec(prefix: str, D: int, features: int, out_channels: Sequence[int]) -> Spec:
    F_, oc = features, list(out_channels)
    s: Spec = OrderedDict()
    for i in range(4):
        s[f"{prefix}projects.{i}.weight"] = (oc[i], D, 1, 1)
        s[f"{prefix}projects.{i}.bias"] = (oc[i],)
    s[prefix + "resize_layers.0.weight"] = (oc[0], oc[0], 4, 4)  # ConvTranspose2d [in,out,k,k]
    s[prefix + "resize_layers.0.bias"] = (oc[0],)
    s[prefix + "resize_layers.1.weight"] = (oc[1], oc[1], 2, 2)
    s[prefix + "resize_layers.1.bias"] = (oc[1],)
    s[prefix + "resize_layers.3.weight"] = (oc[3], oc[3], 3, 3)
    s[prefix + "resize_layers.3.bias"] = (oc[3],)
    for i in range(4):
        s[f"{prefix}scratch.layer{i + 1}_rn.weight"] = (F_, oc[i], 3, 3)
    for r in range(1, 5):
        b = f"{prefix}scratch.refinenet{r}."
        s[b + "out_conv.weight"] = (F_, F_, 1, 1)
        s[b + "out_conv.bias"] = (F_,)
        for u in (1, 2):
            for c in (1, 2):
                s[f"{b}resConfUnit{u}.conv{c}.weight"] = (F_, F_, 3, 3)
                s[f"{b}resConfUnit{u}.conv{c}.bias"] = (F_,)
    s[prefix + "scratch.output_conv1.weight"] = (F_ // 2, F_, 3, 3)
    s[prefix + "scratch.output_conv1.bias"] = (F_ // 2,)
    s[prefix + "scratch.output_conv2.0.weight"] = (32, F_ // 2, 3, 3)
    s[prefix + "scratch.output_conv2.0.bias"] = (32,)
    s[prefix + "scratch.output_conv2.2.weight"] = (1, 32, 1, 1)
    s[prefix + "scratch.output_conv2.2.bias"] = (1,)
    return s


def dav2_spec(prefix: str, model_cfg: dict) -> Spec:
    """DepthAnythingV2(encoder, features, out_channels) parameter table."""
    cfg = dav2_cfg(model_cfg)
    s = dinov2_spec(prefix + "pretrained.", cfg["vit"])
    s.update(dpt_head_spec(prefix + "depth_head.", cfg["vit"]["dim"], cfg["features"], cfg["out_channels"]))
    return s


def dav2_cfg(model_cfg: dict) -> dict:
    """Normalise a reference ``model_cfg`` dict (configs/patchrefiner_dav2/pr_u4k.py:31-35).

    ``vit`` may be given explicitly for reduced test models (the reference has
    no such knob; tests build the reference module with the same dims)."""
    model_cfg = dict(model_cfg)
    enc = model_cfg.get("encoder", "vitl")
    vit = vit_cfg(enc, **model_cfg.get("vit", {}))
    return dict(
        encoder=enc,
        vit=vit,
        features=model_cfg.get("features", 256),
        out_channels=list(model_cfg.get("out_channels", [256, 512, 1024, 1024])),
        max_depth=float(model_cfg.get("max_depth", 20.0)),
    )


DA_CORE = {  # DepthAnythingCore.build (external/zoedepth/models/base_models/depth_anything.py:340-365)
    "vits": dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384]),
    "vitb": dict(encoder="vitb", features=128, out_channels=[96, 192, 384, 768]),
    "vitl": dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024]),
}


# MiDaS v3.1 DPT_BEiT_L_384 (torch.hub "AyaanShah2204/MiDaS", midas.py:342-347; timm beit_large_patch16_384): the core of
# type='ZoeDepth'.  ``beit`` overrides give the reduced nets of the parity tests.
MIDAS_BEIT = {
    "DPT_BEiT_L_384": dict(dim=1024, depth=24, heads=16, taps=[5, 11, 17, 23], patch=16, window=(24, 24), mlp_ratio=4,
                           features=256, out_channels=[256, 512, 1024, 1024]),
}


def midas_beit_cfg(model_type: str = "DPT_BEiT_L_384", **over) -> dict:
    c = dict(MIDAS_BEIT[model_type])
    c.update(over)
    c["window"] = tuple(c["window"])
    return c


def midas_beit_spec(prefix: str, b: dict) -> Spec:
    """MidasCore.core = DPTDepthModel(backbone='beitl16_384') parameter table: pretrained.model.* (timm BEiT),
    pretrained.act_postprocess{1..4}.* (midas/backbones/utils.py make_backbone_default), scratch.* (midas/blocks.py).
    Only what the forward reads; a real checkpoint also carries fc_norm / head / relative_position_index (ignored: the
    reference loads ZoeDepth checkpoints with strict=False, patchrefinerplus.py:108)."""
    D, L, p, F_, oc = b["dim"], b["depth"], b["patch"], b["features"], list(b["out_channels"])
    hid = int(D * b["mlp_ratio"])
    nrd = (2 * b["window"][0] - 1) * (2 * b["window"][1] - 1) + 3
    s: Spec = OrderedDict()
    m = prefix + "pretrained.model."
    s[m + "cls_token"] = (1, 1, D)
    s[m + "patch_embed.proj.weight"] = (D, 3, p, p)
    s[m + "patch_embed.proj.bias"] = (D,)
    for i in range(L):
        k = f"{m}blocks.{i}."
        s[k + "gamma_1"] = (D,)
        s[k + "gamma_2"] = (D,)
        s[k + "norm1.weight"] = (D,)
        s[k + "norm1.bias"] = (D,)
        s[k + "attn.q_bias"] = (D,)
        s[k + "attn.v_bias"] = (D,)
        s[k + "attn.relative_position_bias_table"] = (nrd, b["heads"])
        s[k + "attn.qkv.weight"] = (3 * D, D)
        s[k + "attn.proj.weight"] = (D, D)
        s[k + "attn.proj.bias"] = (D,)
        s[k + "norm2.weight"] = (D,)
        s[k + "norm2.bias"] = (D,)
        s[k + "mlp.fc1.weight"] = (hid, D)
        s[k + "mlp.fc1.bias"] = (hid,)
        s[k + "mlp.fc2.weight"] = (D, hid)
        s[k + "mlp.fc2.bias"] = (D,)
    for i in range(4):
        a = f"{prefix}pretrained.act_postprocess{i + 1}."
        s[a + "0.project.0.weight"] = (D, 2 * D)
        s[a + "0.project.0.bias"] = (D,)
        s[a + "3.weight"] = (oc[i], D, 1, 1)
        s[a + "3.bias"] = (oc[i],)
        if i == 0:
            s[a + "4.weight"] = (oc[0], oc[0], 4, 4)  # ConvTranspose2d [in, out, k, k]
            s[a + "4.bias"] = (oc[0],)
        elif i == 1:
            s[a + "4.weight"] = (oc[1], oc[1], 2, 2)
            s[a + "4.bias"] = (oc[1],)
        elif i == 3:
            s[a + "4.weight"] = (oc[3], oc[3], 3, 3)
            s[a + "4.bias"] = (oc[3],)
    c = prefix + "scratch."
    for i in range(4):
        s[f"{c}layer{i + 1}_rn.weight"] = (F_, oc[i], 3, 3)
    for r in range(1, 5):
        k = f"{c}refinenet{r}."
        s[k + "out_conv.weight"] = (F_, F_, 1, 1)
        s[k + "out_conv.bias"] = (F_,)
        for u in (1, 2):
            for cc in (1, 2):
                s[f"{k}resConfUnit{u}.conv{cc}.weight"] = (F_, F_, 3, 3)
                s[f"{k}resConfUnit{u}.conv{cc}.bias"] = (F_,)
    s[c + "output_conv.0.weight"] = (F_ // 2, F_, 3, 3)
    s[c + "output_conv.0.bias"] = (F_ // 2,)
    s[c + "output_conv.2.weight"] = (32, F_ // 2, 3, 3)
    s[c + "output_conv.2.bias"] = (32,)
    s[c + "output_conv.4.weight"] = (1, 32, 1, 1)
    s[c + "output_conv.4.bias"] = (1,)
    return s


def zoedepth_cfg(cfg: dict) -> dict:
    """Normalise a reference ZoeDepth config dict (configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py:10-66)
    for the DepthAnything-core flavour (type='DA-ZoeDepth')."""
    mt = cfg.get("midas_model_type", "DPT_BEiT_L_384")  # ZoeDepth.build default (zoedepth_v1.py:297)
    if mt not in DA_CORE and mt not in MIDAS_BEIT:
        raise NotImplementedError(
            f"midas_model_type={mt!r}: built are the vendored DepthAnything cores (vits/vitb/vitl) and MiDaS "
            f"{sorted(MIDAS_BEIT)} (midas.py:377-385 lists others no shipped config uses)")
    if cfg.get("bin_centers_type", "softplus") != "softplus" or cfg.get("attractor_type", "inv") != "inv" or \
            cfg.get("attractor_kind", "mean") != "mean":
        raise NotImplementedError("only bin_centers_type='softplus', attractor_type='inv', attractor_kind='mean' "
                                  "(every shipped config)")
    if mt in MIDAS_BEIT:
        beit = midas_beit_cfg(mt, **cfg.get("beit", {}))
        core = dict(beit=beit, features=beit["features"], out_channels=beit["out_channels"])
    else:
        core = dav2_cfg({**DA_CORE[mt], "vit": cfg.get("vit", {})})
    return dict(core=core, core_type=mt, n_bins=int(cfg.get("n_bins", 64)),
                bin_embedding_dim=int(cfg.get("bin_embedding_dim", 128)),
                n_attractors=list(cfg.get("n_attractors", [16, 8, 4, 1])), min_temp=float(cfg.get("min_temp", 5)),
                max_temp=float(cfg.get("max_temp", 50)), min_depth=float(cfg.get("min_depth", 1e-3)),
                max_depth=float(cfg.get("max_depth", 10)))


def zoedepth_spec(prefix: str, cfg: dict) -> Spec:
    """ZoeDepth(core=DepthAnythingCore) parameter table (zoedepth_v1.py:39-123)."""
    z = zoedepth_cfg(cfg)
    F_ = z["core"]["features"]
    nb, emb = z["n_bins"], z["bin_embedding_dim"]
    s: Spec = OrderedDict()
    if "beit" in z["core"]:
        s.update(midas_beit_spec(prefix + "core.core.", z["core"]["beit"]))
    else:
        s.update(dinov2_spec(prefix + "core.core.pretrained.", z["core"]["vit"]))
        s.update(dpt_head_spec(prefix + "core.core.depth_head.", z["core"]["vit"]["dim"], F_, z["core"]["out_channels"]))

    def mlp(name, cin, mid, cout):
        s[f"{prefix}{name}._net.0.weight"] = (mid, cin, 1, 1)
        s[f"{prefix}{name}._net.0.bias"] = (mid,)
        s[f"{prefix}{name}._net.2.weight"] = (cout, mid, 1, 1)
        s[f"{prefix}{name}._net.2.bias"] = (cout,)

    s[prefix + "conv2.weight"] = (F_, F_, 1, 1)
    s[prefix + "conv2.bias"] = (F_,)
    mlp("seed_bin_regressor", F_, 256, nb)
    mlp("seed_projector", F_, 128, emb)
    for i in range(4):
        mlp(f"projectors.{i}", F_, 128, emb)
    for i in range(4):
        mlp(f"attractors.{i}", emb, 128, z["n_attractors"][i])
    last_in = 32 + 1
    bott = (last_in + emb) // 2
    s[prefix + "conditional_log_binomial.log_binomial_transform.k_idx"] = (1, nb, 1, 1)
    s[prefix + "conditional_log_binomial.log_binomial_transform.K_minus_1"] = (1, 1, 1, 1)
    s[prefix + "conditional_log_binomial.mlp.0.weight"] = (bott, last_in + emb, 1, 1)
    s[prefix + "conditional_log_binomial.mlp.0.bias"] = (bott,)
    s[prefix + "conditional_log_binomial.mlp.2.weight"] = (4, bott, 1, 1)
    s[prefix + "conditional_log_binomial.mlp.2.bias"] = (4,)
    return s


def fusion_unet_spec(prefix: str, input_chl, temp_chl, dec_chl) -> Spec:
    s: Spec = OrderedDict()
    for l, (ic, tc) in enumerate(zip(input_chl, temp_chl)):
        s[f"{prefix}encoder_layers_1.{l}.single_conv.0.weight"] = (tc, ic, 3, 3)
        s[f"{prefix}encoder_layers_1.{l}.single_conv.1.weight"] = (tc,)
        s[f"{prefix}encoder_layers_1.{l}.single_conv.1.bias"] = (tc,)
        s[f"{prefix}encoder_layers_2.{l}.single_conv.0.weight"] = (tc, tc + 2, 3, 3)
        s[f"{prefix}encoder_layers_2.{l}.single_conv.1.weight"] = (tc,)
        s[f"{prefix}encoder_layers_2.{l}.single_conv.1.bias"] = (tc,)
    t = list(temp_chl)[::-1]
    ch = t[0]
    for j, (tc, dc) in enumerate(zip(t[1:], dec_chl)):
        n = tc + ch + 2
        s[f"{prefix}decoder_layers.{j}.conv.double_conv.0.weight"] = (n, n, 3, 3)
        s[f"{prefix}decoder_layers.{j}.conv.double_conv.2.weight"] = (dc, n, 3, 3)
        ch = dc
    last = dec_chl[-1] if len(dec_chl) else ch
    s[prefix + "final_conv.weight"] = (1, last, 3, 3)
    return s


def _gated_unit_spec(s: Spec, b: str, F_: int, fusion: bool = True):
    s[b + "conv.weight"] = (F_, F_, 3, 3)
    s[b + "conv.bias"] = (F_,)
    if not fusion:  # 'self-agg': GatedConvUnit(fusion=False) has no fusion_conv (bi_directional_fusion_model.py:45-51)
        return
    s[b + "fusion_conv.0.weight"] = (F_, 2 * F_, 3, 3)
    s[b + "fusion_conv.0.bias"] = (F_,)
    s[b + "fusion_conv.1.weight"] = (F_,)
    s[b + "fusion_conv.1.bias"] = (F_,)
    s[b + "fusion_conv.3.weight"] = (F_, F_, 1, 1)


def _gated_block_spec(s: Spec, b: str, F_: int, fusion: bool = True):
    s[b + "out_conv.weight"] = (F_, F_, 1, 1)
    s[b + "out_conv.bias"] = (F_,)
    _gated_unit_spec(s, b + "GateresConfUnit1.", F_, fusion)
    _gated_unit_spec(s, b + "GateresConfUnit2.", F_, fusion)


# -> (fusion, gate) of the GatedConvUnits; 'only-gate' = C2FNOENCModule(fusion=True, gate=False) (bi_directional_fusion_model.py:355-372)
C2F_TYPES = {"coarse-gated": (True, True), "coarse-fusion": (True, False), "self-agg": (False, False), "only-gate": (True, False)}


def bidir_fusion_spec(prefix: str, coarse_chl, fine_chl, fine_chl_after_coarse2fine, temp_chl, dec_chl,
                      features: int = 256, coarse2fine_type: str = "coarse-gated", coarse2fine: bool = True, heavy: bool = False) -> Spec:
    """BiDirectionalFusion parameter table for the C2FModule types (bi_directional_fusion_model.py:355-372): 'coarse-gated' and
    'coarse-fusion' hold the same parameters, 'self-agg' drops every fusion_conv.  ``heavy``: BiDirectionalFusionHeavy (:519-560) --
    three convs per encoder layer (SingleConvCNNLNHeavy, :449-463) and five per decoder stage (DoubleConvHeavy, :465-485)."""
    fusion = C2F_TYPES[coarse2fine_type][0]
    s: Spec = OrderedDict()

    def enc(b, cin, tc):
        s[b + "single_conv.0.weight"] = (tc, cin, 3, 3)
        s[b + "single_conv.1.weight"] = (tc,)
        s[b + "single_conv.1.bias"] = (tc,)
        if heavy:
            s[b + "single_conv.2.weight"] = (tc, tc, 3, 3)
            s[b + "single_conv.3.weight"] = (tc,)
            s[b + "single_conv.3.bias"] = (tc,)
            s[b + "single_conv.4.weight"] = (tc, tc, 3, 3)

    for l, (cc, fc, tc) in enumerate(zip(coarse_chl, fine_chl_after_coarse2fine, temp_chl)):
        enc(f"{prefix}fusion_layers_1.{l}.", cc + fc, tc)
        enc(f"{prefix}fusion_layers_2.{l}.", tc + 2, tc)
    t = list(temp_chl)[::-1]
    ch = t[0]
    for j, (tc, dc) in enumerate(zip(t[1:], dec_chl)):
        n = tc + ch + 2
        s[f"{prefix}f2r_agg.{j}.conv.double_conv.0.weight"] = (n, n, 3, 3)
        if heavy:
            for i in (2, 4, 6):
                s[f"{prefix}f2r_agg.{j}.conv.double_conv.{i}.weight"] = (n, n, 3, 3)
        s[f"{prefix}f2r_agg.{j}.conv.double_conv.{8 if heavy else 2}.weight"] = (dc, n, 3, 3)
        ch = dc
    last = dec_chl[-1] if len(dec_chl) else ch
    s[prefix + "final_conv.weight"] = (1, last, 3, 3)
    if not coarse2fine:  # no c2f module (bi_directional_fusion_model.py:364): the refiner's pyramid goes straight to fusion_layers_1
        return s
    c = prefix + "c2f.scratch."
    for i in range(5):
        s[f"{c}layer{i + 1}_rn.weight"] = (features, fine_chl[i], 3, 3)
    if coarse2fine_type == "only-gate":  # C2FNOENCModule (:211-251): two units per level, no refinenet blocks / head convs
        for k in range(1, 6):
            _gated_unit_spec(s, f"{c}layer{k}_gate1.", features)
            _gated_unit_spec(s, f"{c}layer{k}_gate2.", features)
        s[c + "upsample_conv.0.weight"] = (fine_chl[0], 32, 2, 2)  # ConvTranspose2d: [in, out, k, k]
        s[c + "upsample_conv.0.bias"] = (32,)
        s[c + "upsample_conv.2.weight"] = (32, 32, 3, 3)
        _gated_unit_spec(s, c + "layer6_gate1.", 32)
        _gated_unit_spec(s, c + "layer6_gate2.", 32)
        s[c + "output_conv.weight"] = (1, 32, 3, 3)
        s[c + "output_conv.bias"] = (1,)
        return s
    for r in range(1, 6):
        _gated_block_spec(s, f"{c}refinenet{r}.", features, fusion)
    h2 = coarse_chl[0]
    s[c + "output_conv1.weight"] = (features // 2, features, 3, 3)
    s[c + "output_conv1.bias"] = (features // 2,)
    s[c + "output_conv2.0.weight"] = (h2, features // 2, 3, 3)
    s[c + "output_conv2.0.bias"] = (h2,)
    _gated_block_spec(s, c + "output_conv2_fusion.", h2, fusion)
    s[c + "output_conv3.0.weight"] = (1, h2, 1, 1)
    s[c + "output_conv3.0.bias"] = (1,)
    return s


# ----------------------------------------------------------------------------
# timm mobilenetv4_conv_small (features_only) -- un-vendored third party.
# Architecture restated from the public MobileNetV4 definition (conv-small):
#   stem 3x3 s2 -> 32
#   stage0: cn 3x3 s2 ->32, cn 1x1 ->32            (/4,  32)
#   stage1: cn 3x3 s2 ->96, cn 1x1 ->64            (/8,  64)
#   stage2: uir(5,5,s2,e3)->96, 4x uir(0,3,e2)->96, uir(3,0,e4)->96   (/16, 96)
#   stage3: uir(3,3,s2,e6)->128, uir(5,5,e4), uir(0,5,e4), uir(0,5,e3),
#           uir(0,3,e4), uir(0,3,e4) ->128         (/32, 128)
#   stage4: cn 1x1 -> 960                          (/32, 960)
# feature taps (features_only): stem act (/2, 32), stage0, stage1, stage2, stage4.
# Every conv is followed by BatchNorm (+ReLU except the projection conv).
# ----------------------------------------------------------------------------
MNV4_SMALL = dict(
    stem=32,
    stages=[
        [("cn", 3, 2, 32), ("cn", 1, 1, 32)],
        [("cn", 3, 2, 96), ("cn", 1, 1, 64)],
        [("uir", 5, 5, 2, 3.0, 96), ("uir", 0, 3, 1, 2.0, 96), ("uir", 0, 3, 1, 2.0, 96),
         ("uir", 0, 3, 1, 2.0, 96), ("uir", 0, 3, 1, 2.0, 96), ("uir", 3, 0, 1, 4.0, 96)],
        [("uir", 3, 3, 2, 6.0, 128), ("uir", 5, 5, 1, 4.0, 128), ("uir", 0, 5, 1, 4.0, 128),
         ("uir", 0, 5, 1, 3.0, 128), ("uir", 0, 3, 1, 4.0, 128), ("uir", 0, 3, 1, 4.0, 128)],
        [("cn", 1, 1, 960)],
    ],
    feature_stages=[0, 1, 2, 4],  # + the stem
    mean=(0.485, 0.456, 0.406),
    std=(0.229, 0.224, 0.225),
)


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def _bn_spec(s: Spec, b: str, c: int):
    s[b + "weight"] = (c,)
    s[b + "bias"] = (c,)
    s[b + "running_mean"] = (c,)
    s[b + "running_var"] = (c,)


def mnv4_layers(arch: dict = MNV4_SMALL, in_chans: int = 4):
    """Flatten the architecture into a list of primitive (conv+bn[+relu]) layers.

    Returns (layers, taps): each layer is a dict(name, kind, cin, cout, k, stride,
    groups, act, block_start, block_end/residual); taps = indices after which a
    feature map is emitted."""
    layers = []
    taps = []
    c = arch["stem"]
    layers.append(dict(conv="conv_stem", bn="bn1", cin=in_chans, cout=c, k=3, s=2, g=1, act=True))
    taps.append(len(layers) - 1)
    for si, stage in enumerate(arch["stages"]):
        for bi, blk in enumerate(stage):
            b = f"blocks.{si}.{bi}."
            if blk[0] == "cn":
                _, k, st, co = blk
                layers.append(dict(conv=b + "conv", bn=b + "bn1", cin=c, cout=co, k=k, s=st, g=1, act=True))
                c = co
            else:
                _, k0, k1, st, e, co = blk
                mid = make_divisible(c * e)
                first = len(layers)
                if k0:
                    # start depthwise: strided only if there is no mid depthwise
                    layers.append(dict(conv=b + "dw_start.conv", bn=b + "dw_start.bn", cin=c, cout=c, k=k0,
                                       s=(1 if k1 else st), g=c, act=False))
                layers.append(dict(conv=b + "pw_exp.conv", bn=b + "pw_exp.bn", cin=c, cout=mid, k=1, s=1, g=1, act=True))
                if k1:
                    layers.append(dict(conv=b + "dw_mid.conv", bn=b + "dw_mid.bn", cin=mid, cout=mid, k=k1, s=st, g=mid, act=True))
                layers.append(dict(conv=b + "pw_proj.conv", bn=b + "pw_proj.bn", cin=mid, cout=co, k=1, s=1, g=1, act=False))
                if st == 1 and c == co:
                    layers[first]["res_begin"] = True
                    layers[-1]["res_end"] = True
                c = co
        if si in arch["feature_stages"]:
            taps.append(len(layers) - 1)
    return layers, taps


def mnv4_spec(prefix: str, arch: dict = MNV4_SMALL, in_chans: int = 4) -> Spec:
    s: Spec = OrderedDict()
    layers, _ = mnv4_layers(arch, in_chans)
    for L in layers:
        s[prefix + L["conv"] + ".weight"] = (L["cout"], L["cin"] // L["g"], L["k"], L["k"])
        _bn_spec(s, prefix + L["bn"] + ".", L["cout"])
    return s

# ConvNeXt-L (timm ``convnext_large``, features_only): stem = Conv 4x4 stride 4 + LayerNorm2d, four stages of
# [LayerNorm2d + Conv 2x2 stride 2 (stages 1..3)] + depth x block(dw 7x7 -> LN -> Linear 4C -> GELU -> Linear C ->
# gamma -> + x), every LayerNorm with eps 1e-6, feature taps = the four stage outputs (no final norm).  timm is not
# vendored in the reference: the arithmetic is pinned against the independent implementation in HuggingFace
# ``transformers`` (oracle/make_golden.py::g_convnext); the flattened timm key names (``stem_0`` is confirmed by the
# reference's stem surgery, patchrefinerplus.py:194-200; the rest follows timm's FeatureListNet) are unpinned.
CONVNEXT_LARGE = dict(dims=(192, 384, 768, 1536), depths=(3, 3, 27, 3), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))
CONVNEXT_TINY_TEST = dict(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225))


def convnext_spec(prefix: str, arch: dict = CONVNEXT_LARGE, in_chans: int = 4) -> Spec:
    s: Spec = OrderedDict()
    d = arch["dims"]
    s[prefix + "stem_0.weight"] = (d[0], in_chans, 4, 4)
    s[prefix + "stem_0.bias"] = (d[0],)
    s[prefix + "stem_1.weight"] = (d[0],)
    s[prefix + "stem_1.bias"] = (d[0],)
    for i, (c, n) in enumerate(zip(d, arch["depths"])):
        st = f"{prefix}stages_{i}."
        if i > 0:
            s[st + "downsample.0.weight"] = (d[i - 1],)
            s[st + "downsample.0.bias"] = (d[i - 1],)
            s[st + "downsample.1.weight"] = (c, d[i - 1], 2, 2)
            s[st + "downsample.1.bias"] = (c,)
        for j in range(n):
            b = f"{st}blocks.{j}."
            s[b + "gamma"] = (c,)
            s[b + "conv_dw.weight"] = (c, 1, 7, 7)
            s[b + "conv_dw.bias"] = (c,)
            s[b + "norm.weight"] = (c,)
            s[b + "norm.bias"] = (c,)
            s[b + "mlp.fc1.weight"] = (4 * c, c)
            s[b + "mlp.fc1.bias"] = (4 * c,)
            s[b + "mlp.fc2.weight"] = (c, 4 * c)
            s[b + "mlp.fc2.bias"] = (c,)
    return s

# EfficientNet (timm ``tf_efficientnet_b5_ap``, features_only; v2_eff_u4k.py:94-101): compound-scaled MBConv stages with
# squeeze-excite, SiLU, BatchNorm eps 1e-3 and TensorFlow "SAME" padding (timm Conv2dSame -- the reference's stem surgery
# builds one: patchrefinerplus.py:152-158).  Stage 0 = DepthwiseSeparable blocks (dw -> SE -> pw), the rest
# InvertedResidual (pw expand x6 -> dw -> SE -> pw-linear); residual when stride 1 and cin == cout; the SE bottleneck is
# 1/4 of the BLOCK INPUT width.  Feature taps = outputs of stages 0, 1, 2, 4, 6 (strides 2..32): b5 = [24, 40, 64, 176, 512]
# (the config's fine_chl).  timm is not vendored: the arithmetic is pinned against HuggingFace ``transformers``'
# EfficientNet (oracle/make_golden.py::g_effnet); key names and the AdvProp mean/std (0.5) are recollection.
_EFF_BASE = [("ds", 3, 1, 1, 16, 1), ("ir", 3, 2, 6, 24, 2), ("ir", 5, 2, 6, 40, 2), ("ir", 3, 2, 6, 80, 3),
             ("ir", 5, 1, 6, 112, 3), ("ir", 5, 2, 6, 192, 4), ("ir", 3, 1, 6, 320, 1)]


def _round_filters(f, width, divisor=8):
    f = f * width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def effnet_arch(width: float, depth: float, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)) -> dict:
    import math
    stages = [(kind, k, s, e, _round_filters(c, width), int(math.ceil(depth * r))) for kind, k, s, e, c, r in _EFF_BASE]
    return dict(stem=_round_filters(32, width), stages=stages, taps=(0, 1, 2, 4, 6), se_ratio=0.25, bn_eps=1e-3, mean=mean, std=std,
                width=width, depth=depth)


EFFNET_B5 = effnet_arch(1.6, 2.2)
EFFNET_TINY_TEST = effnet_arch(0.5, 0.4)


def effnet_blocks(arch: dict):
    """flat list of blocks: dict(name, kind, cin, cmid, cout, k, s, cse, res, tap)"""
    out = []
    cin = arch["stem"]
    for si, (kind, k, s, e, c, r) in enumerate(arch["stages"]):
        for j in range(r):
            st = s if j == 0 else 1
            out.append(dict(name=f"blocks.{si}.{j}.", kind=kind, cin=cin, cmid=cin * e, cout=c, k=k, s=st,
                            cse=max(1, int(cin * arch["se_ratio"])), res=(st == 1 and cin == c),
                            tap=(j == r - 1 and si in arch["taps"])))
            cin = c
    return out


def effnet_spec(prefix: str, arch: dict = None, in_chans: int = 4) -> Spec:
    arch = arch or EFFNET_B5
    s: Spec = OrderedDict()
    s[prefix + "conv_stem.weight"] = (arch["stem"], in_chans, 3, 3)
    _bn_spec(s, prefix + "bn1.", arch["stem"])
    for B in effnet_blocks(arch):
        b = prefix + B["name"]
        if B["kind"] == "ds":
            s[b + "conv_dw.weight"] = (B["cin"], 1, B["k"], B["k"])
            _bn_spec(s, b + "bn1.", B["cin"])
        else:
            s[b + "conv_pw.weight"] = (B["cmid"], B["cin"], 1, 1)
            _bn_spec(s, b + "bn1.", B["cmid"])
            s[b + "conv_dw.weight"] = (B["cmid"], 1, B["k"], B["k"])
            _bn_spec(s, b + "bn2.", B["cmid"])
        s[b + "se.conv_reduce.weight"] = (B["cse"], B["cmid"], 1, 1)
        s[b + "se.conv_reduce.bias"] = (B["cse"],)
        s[b + "se.conv_expand.weight"] = (B["cmid"], B["cse"], 1, 1)
        s[b + "se.conv_expand.bias"] = (B["cmid"],)
        if B["kind"] == "ds":
            s[b + "conv_pw.weight"] = (B["cout"], B["cmid"], 1, 1)
            _bn_spec(s, b + "bn2.", B["cout"])
        else:
            s[b + "conv_pwl.weight"] = (B["cout"], B["cmid"], 1, 1)
            _bn_spec(s, b + "bn3.", B["cout"])
    return s


# ----------------------------------------------------------------------------
# synthetic weights
# ----------------------------------------------------------------------------
def _rng(name: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """Deterministic value for one named parameter (numpy PCG64 keyed by name).

    Scaling keeps activations O(1) through the ~45-layer chain so that neither
    the sigmoid depth head nor the clamp saturates."""
    g = _rng(name, seed)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "k_idx":  # LogBinomial buffers (dist_layers.py:45-48): fixed values, not random
        return torch.arange(0, shape[1]).view(shape)
    if leaf == "K_minus_1":  # filled from its sibling k_idx by synth_state_dict
        return torch.zeros(shape)
    n = int(np.prod(shape)) if len(shape) else 1
    z = g.standard_normal(n).astype(np.float32).reshape(shape)
    if leaf in ("gamma_1", "gamma_2"):  # BEiT layer scale (init_values 0.1 in timm; O(1) here like DINOv2's LayerScale test values)
        v = 1.0 + 0.05 * z
    elif leaf in ("q_bias", "v_bias"):
        v = 0.02 * z
    elif leaf == "relative_position_bias_table":
        v = 0.5 * z  # O(1) logit offsets: the bias must matter next to q.k
    elif leaf == "gamma" and ".stages_" in name:  # ConvNeXt layer scale: 36 residual blocks deep -- keep the trunk O(1)
        v = 0.15 + 0.02 * z
    elif leaf == "gamma":  # LayerScale
        v = 1.0 + 0.05 * z
    elif leaf == "running_var":
        v = 1.0 + 0.1 * np.abs(z)
    elif leaf == "running_mean":
        v = 0.1 * z
    elif leaf in ("cls_token", "pos_embed"):
        v = 0.02 * z
    elif leaf == "mask_token":
        v = 0.0 * z
    elif leaf == "bias":
        if "seed_bin_regressor._net.2" in name:
            v = 1.5 * z  # spread the 64 seed bin centres (softplus) over ~0.1 .. 5
        else:
            v = (0.1 if len(shape) == 1 and _is_norm(name) else 0.02) * z
    elif len(shape) == 1:  # norm weight
        v = 1.0 + 0.1 * z
    else:
        if ("resize_layers.0." in name or "resize_layers.1." in name or "upsample_convx" in name
                or "act_postprocess1.4." in name or "act_postprocess2.4." in name):
            fan_in = shape[0]  # ConvTranspose2d [in, out, k, k], k == stride
        else:
            fan_in = int(np.prod(shape[1:]))
        if name.endswith("final_conv.weight"):
            gain = 0.5  # offset head: O(1) metres on top of the coarse depth
        elif "seed_bin_regressor._net.2" in name or "conditional_log_binomial.mlp.2" in name:
            gain = 3.0
        elif "output_conv2.2." in name or "scratch.output_conv.4." in name:
            gain = 1.5  # pre-sigmoid logits O(1): the depth head must not saturate
        elif re.search(r"resConfUnit\d\.conv2\.|GateresConfUnit\d\.conv\.", name):
            gain = 0.4  # residual branches: keep the pyramid's scale flat across levels
        else:
            gain = 1.0
        v = z * (gain / np.sqrt(max(fan_in, 1)))
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def _is_norm(name: str) -> bool:
    return any(t in name for t in (".norm", "norm1", "norm2", "single_conv.1", "fusion_conv.1", ".bn", "stem_1.", "downsample.0."))


def synth_state_dict(spec: Spec, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    sd = OrderedDict((k, synth_tensor(k, tuple(shp), seed)) for k, shp in spec.items())
    for k in sd:
        if k.endswith("K_minus_1"):
            sd[k] = torch.tensor([float(sd[k[:-len("K_minus_1")] + "k_idx"].numel() - 1)]).view(sd[k].shape)
    return sd


# ------------------------------------------------------------------------------------------------------------------
# checkpoint readiness: what to do when a released checkpoint's key names differ from this build's table
# ------------------------------------------------------------------------------------------------------------------
# The reference's own modules (estimator/*, external/depth_anything*, external/zoedepth) are checked name by name against the
# imported reference (oracle/make_golden.py).  Two un-vendored packages contribute names that could only be restated from their
# published conventions: timm (``refiner_encoder.*``: MobileNetV4 / ConvNeXt / EfficientNet under ``features_only=True``, whose
# FeatureListNet flattens the top-level Sequentials -- ``blocks.3`` may appear as ``blocks_3``, ``stem.0`` as ``stem_0``,
# ``stages.2`` as ``stages_2``) and the torch.hub MiDaS fork (``core.core.pretrained.model.*`` = timm's BEiT).  These rules
# rewrite such variants; ``diagnose_state_dict`` reports what they cannot explain.
REMAP_RULES = (
    ("DistributedDataParallel prefix", r"^module\.", ""),
    ("timm FeatureListNet flattening (blocks_N -> blocks.N)", r"(^|\.)blocks_(\d+)(\.|$)", r"\1blocks.\2\3"),
    ("timm unflattened stem (stem.N -> stem_N)", r"(^|\.)stem\.(\d+)\.", r"\1stem_\2."),
    ("timm unflattened stages (stages.N -> stages_N)", r"(^|\.)stages\.(\d+)\.", r"\1stages_\2."),
)
IGNORABLE = (r"\.num_batches_tracked$", r"\.relative_position_index$", r"\.attn\.k_bias$")  # bookkeeping / derived buffers


def remap_state_dict(sd, spec: Spec):
    """Apply REMAP_RULES to the keys of ``sd`` that are not in ``spec`` (a rule is used for a key only if the rewritten name IS
    in ``spec`` with the same shape); drop IGNORABLE bookkeeping keys.  Returns (new_sd, applied) with applied = {rule: count}."""
    out, applied = OrderedDict(), {}
    for k, v in sd.items():
        if k in spec:
            out[k] = v
            continue
        if any(re.search(p, k) for p in IGNORABLE):
            applied["dropped bookkeeping buffers"] = applied.get("dropped bookkeeping buffers", 0) + 1
            continue
        cands, names = [k], [[]]
        for name, pat, rep in REMAP_RULES:  # rules compose (a DDP prefix on top of a flattened name)
            for c, used in list(zip(cands, names)):
                n = re.sub(pat, rep, c)
                if n != c and n not in cands:
                    cands.append(n)
                    names.append(used + [name])
        hit = next(((c, used) for c, used in zip(cands, names) if c in spec and tuple(spec[c]) == tuple(v.shape)), None)
        if hit is None:
            out[k] = v
        else:
            out[hit[0]] = v
            for name in hit[1]:
                applied[name] = applied.get(name, 0) + 1
    return out, applied


def diagnose_state_dict(spec: Spec, sd, depth: int = 4) -> dict:
    """What of ``sd`` this build can and cannot use: matched / missing / unexpected / shape-mismatched keys grouped by module
    prefix (``depth`` name components, indices folded to N), plus, for every unexpected key, the missing keys with the same
    leaf name and shape inside the same top-level module (candidate renames a maintainer can turn into a REMAP_RULES line)."""
    def group(keys):
        g = OrderedDict()
        for k in keys:
            p = ".".join(re.sub(r"\.\d+(?=\.|$)", ".N", k).split(".")[:depth])
            g.setdefault(p, []).append(k)
        return OrderedDict((p, dict(count=len(ks), example=ks[0])) for p, ks in g.items())
    missing = [k for k in spec if k not in sd]
    unexpected = [k for k in sd if k not in spec]
    mism = [(k, tuple(sd[k].shape), tuple(spec[k])) for k in spec if k in sd and tuple(sd[k].shape) != tuple(spec[k])]
    cands = OrderedDict()
    by_leaf = {}
    for k in missing:
        by_leaf.setdefault((k.split(".")[0], k.rsplit(".", 1)[-1], tuple(spec[k])), []).append(k)
    for k in unexpected:
        c = by_leaf.get((k.split(".")[0], k.rsplit(".", 1)[-1], tuple(sd[k].shape)), [])
        if c:
            cands[k] = c[:3]
    return dict(matched=len(spec) - len(missing) - len(mism), total=len(spec), missing=group(missing), unexpected=group(unexpected),
                shape_mismatch=mism[:20], rename_candidates=OrderedDict(list(cands.items())[:40]))


def format_diagnosis(d: dict) -> str:
    lines = [f"state dict: {d['matched']} of {d['total']} parameters matched"]
    for title, key in (("MISSING (in this model, not in the checkpoint)", "missing"), ("UNEXPECTED (in the checkpoint, unknown here)", "unexpected")):
        if d[key]:
            lines.append(f"  {title}:")
            lines += [f"    {v['count']:5d}  {p}.*   e.g. {v['example']}" for p, v in d[key].items()]
    if d["shape_mismatch"]:
        lines.append("  SHAPE MISMATCH: " + "; ".join(f"{k}: checkpoint {a} vs model {b}" for k, a, b in d["shape_mismatch"][:8]))
    if d["rename_candidates"]:
        lines.append("  rename candidates (same leaf name + shape inside the same top-level module; add a rule to weights.REMAP_RULES):")
        lines += [f"    {k}  ->  {' | '.join(c)}" for k, c in list(d["rename_candidates"].items())[:12]]
    return "\n".join(lines)
