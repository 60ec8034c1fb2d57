"""Oracle (test infrastructure): the seeded parity cases shared by make_golden.py and tests/.

A case = model dims + weight seed + input seed.  Inputs/weights are regenerated from seeds
everywhere; only the reference's OUTPUTS are stored under tests/golden.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from patchrefinerv2_amd import weights as W


def rand_image(seed: int, b: int, h: int, w: int) -> torch.Tensor:
    """image_hr = torch.rand(B,3,H,W, generator=manual_seed(f)) in [0,1)."""
    return torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(seed))


def randn(seed: int, *shape) -> torch.Tensor:
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


# -- tiny DepthAnythingV2: patch 14, pos-embed grid 5x5 (img 70), D 64, 4 blocks, 2 heads --------
TINY_DAV2 = dict(
    model_cfg=dict(encoder="vits", features=32, out_channels=[16, 32, 64, 64], max_depth=20.0,
                   vit=dict(dim=128, depth=4, heads=2, taps=[0, 1, 2, 3], img_size=70)),
    seed=1,
    inputs=dict(rect=(56, 84), square=(70, 70), big=(112, 98)),
)


def tiny_dav2_sd(prefix: str = "", seed: int = 7):
    return W.synth_state_dict(W.dav2_spec(prefix, TINY_DAV2["model_cfg"]), seed=seed)


# -- FusionUnet, reduced channels (real cfg: configs/patchrefiner_dav2/pr_u4k.py:44-48) -----------
def _pyramid(seed, b, chl, sizes):
    return [randn(seed + i, b, c, h, w) for i, (c, (h, w)) in enumerate(zip(chl, sizes))]


def _fu_inputs():
    sizes = [(28, 42), (16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    c = _pyramid(100, 2, [16, 32, 32, 32, 32, 32], sizes)
    f = _pyramid(200, 2, [16, 32, 32, 32, 32, 32], sizes)
    pred1 = torch.rand(2, 1, 28, 42, generator=torch.Generator().manual_seed(301)) * 10
    pred2 = torch.rand(2, 1, 28, 42, generator=torch.Generator().manual_seed(302)) * 10
    return dict(c_feat=c, f_feat=f, pred1=pred1, pred2=pred2)


TINY_FUSION_UNET = dict(input_chl=[32, 64, 64, 64, 64, 64], temp_chl=[16, 34, 32, 32, 32, 32],
                        dec_chl=[32, 32, 32, 32, 16], seed=21, make_inputs=_fu_inputs)


# -- BiDirectionalFusion 'coarse-gated' (C2F features=256 is hard-wired -> tiny spatial sizes) -----
def _bd_inputs(tag):
    f_sizes = [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    c_sizes = f_sizes if tag == "same" else [(32, 48), (18, 26), (9, 13), (5, 7), (3, 4), (2, 2)]
    c = _pyramid(400, 1, [32, 256, 256, 256, 256, 256], c_sizes)
    f = _pyramid(500, 1, [32, 32, 32, 64, 96, 960], f_sizes)  # index 0 = dropped 2x copy
    pred1 = torch.rand(1, 1, 32, 48, generator=torch.Generator().manual_seed(601)) * 10
    pred2 = torch.zeros(1, 1, 32, 48)
    return dict(c_feat=c, f_feat=f, pred1=pred1, pred2=pred2)


TINY_BIDIR = dict(coarse_chl=[32, 256, 256, 256, 256, 256], fine_chl=[32, 32, 64, 96, 960],
                  fine_chl_after=[32, 256, 256, 256, 256, 256], temp_chl=[32, 64, 64, 128, 256, 512],
                  dec_chl=[512, 256, 128, 64, 32], seed=31, make_inputs=_bd_inputs)


# BiDirectionalFusion(coarse2fine=False) -- the "base" ablations (configs/patchrefinerv2_zoedepth_ablation/plus_*_u4k_base_coarse.py): the
# refiner pyramid enters fusion_layers_1 directly; level 0 is the x2 bilinear copy of level 1 (lightweight_refiner.py:314-316), pred2 zeros
def _bd_noc2f_inputs(tag):
    import torch.nn.functional as F
    i = _bd_inputs(tag)
    f = i["f_feat"]
    f[0] = F.interpolate(f[1], scale_factor=2, mode="bilinear", align_corners=True)
    return i


TINY_BIDIR_NOC2F = dict(TINY_BIDIR, fine_chl_after=[32, 32, 32, 64, 96, 960], make_inputs=_bd_noc2f_inputs)


# -- end-to-end V1: PatchRefiner(DA2 tiny x2 + FusionUnet), 216x384 frame, 2x2 -------------------
_E2E_DA2 = dict(encoder="vits", features=32, out_channels=[16, 32, 64, 64],
                vit=dict(dim=128, depth=4, heads=2, taps=[0, 1, 2, 3], img_size=70))
E2E_V1 = dict(
    raw=[216, 384], split=[2, 2], pps=[56, 84], max_depth=80.0, seed=0, modes=["m1", "m2", "r8"],
    da2_cfg=_E2E_DA2,
    fusion=dict(input_chl=[32, 64, 64, 64, 64, 64], temp_chl=[16, 32, 32, 32, 32, 32], dec_chl=[32, 32, 32, 32, 16]),
)
E2E_V1["ref_config"] = dict(
    image_raw_shape=E2E_V1["raw"], patch_process_shape=E2E_V1["pps"], patch_raw_shape=[108, 192],
    patch_split_num=E2E_V1["split"], fusion_feat_level=6, min_depth=1e-3, max_depth=80.0,
    pretrain_coarse_model=None, pretrain_fine_model=None, strategy_refiner_target="offset_coarse",
    coarse_branch=dict(type="DA2", pretrained="dummy_da2", model_cfg=dict(encoder="vits")),
    refiner=dict(fine_branch=dict(type="DA2", pretrained="dummy_da2", model_cfg=dict(encoder="vits")),
                 fusion_model=dict(type="FusionUnet", **E2E_V1["fusion"])),
    sigloss=dict(type="SILogLoss"), pretrained=None, pre_norm_bbox=True)


def e2e_v1_sd(seed: int = 41):
    spec = OrderedDict()
    spec.update(W.dav2_spec("coarse_branch.", _E2E_DA2))
    spec.update(W.dav2_spec("refiner_fine_branch.", _E2E_DA2))
    spec.update(W.fusion_unet_spec("refiner_fusion_model.", **E2E_V1["fusion"]))
    return W.synth_state_dict(spec, seed=seed)


# -- end-to-end V2: PatchRefinerPlus(DA2 tiny w/ 256 features + MNv4-S + BiDirectionalFusion) ----
# NB the reference only resizes the coarse pyramid to the refiner's sizes when the LOWEST level
# differs (bi_directional_fusion_model.py:389-393); P must make (P/28) != ceil(P/32) on one axis
# (true for the real 448x448 config: 16 vs 14), otherwise torch.cat fails at a middle level.
_E2E2_DA2 = dict(encoder="vits", features=256, out_channels=[16, 32, 64, 64],
                 vit=dict(dim=128, depth=4, heads=2, taps=[0, 1, 2, 3], img_size=70))
E2E_V2 = dict(
    raw=[256, 512], split=[2, 2], pps=[112, 224], max_depth=80.0, seed=0, modes=["m1", "r4"],
    da2_cfg=_E2E2_DA2,
    fusion=dict(coarse_chl=[128, 256, 256, 256, 256, 256], fine_chl=[32, 32, 64, 96, 960],
                fine_chl_after_coarse2fine=[128, 256, 256, 256, 256, 256], temp_chl=[32, 64, 64, 128, 256, 512],
                dec_chl=[512, 256, 128, 64, 32]),
)
E2E_V2["ref_config"] = dict(
    e2e_training=True, pretrain_stage=False, image_raw_shape=E2E_V2["raw"], patch_process_shape=E2E_V2["pps"],
    patch_raw_shape=[128, 256], patch_split_num=E2E_V2["split"], fusion_feat_level=6, min_depth=1e-3, max_depth=80.0,
    pretrain_coarse_model=None, strategy_refiner_target="offset_coarse",
    coarse_branch=dict(type="DA2", pretrained="dummy_da2", model_cfg=dict(encoder="vits")),
    refiner=dict(
        fine_branch=dict(type="LightWeightRefiner", coarse_condition=True, with_decoder=False,
                         encoder_name="mobilenetv4_conv_small.e2400_r224_in1k"),
        fusion_model=dict(type="BiDirectionalFusion", encoder_name="mobilenetv4_conv_small.e2400_r224_in1k",
                          coarse2fine=True, coarse2fine_type="coarse-gated", **E2E_V2["fusion"])),
    sigloss=dict(type="SILogLoss"), gmloss=dict(type="GradMatchLoss"), sigweight=1, pre_norm_bbox=True,
    pretrained=None, whole_pretrained=None)


def e2e_v2_sd(seed: int = 43):
    spec = OrderedDict()
    spec.update(W.dav2_spec("coarse_branch.", _E2E2_DA2))
    spec.update(W.mnv4_spec("refiner_fine_branch.refiner_encoder.", in_chans=4))
    f = E2E_V2["fusion"]
    spec.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"],
                                    f["fine_chl_after_coarse2fine"], f["temp_chl"], f["dec_chl"]))
    return W.synth_state_dict(spec, seed=seed)


# -- ZoeDepth metric-bins head over the vendored DepthAnything (v1) core, real ViT-S dims ---------------
# (type='DA-ZoeDepth'; head hyper-parameters as in configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py:10-66)
ZOE_DA = dict(
    zcfg=dict(midas_model_type="vits", min_depth=1e-3, max_depth=80, pretrained_resource=None, use_pretrained_midas=False,
              train_midas=True, freeze_midas_bn=True, do_resize=False, attractor_alpha=1000, attractor_gamma=2,
              attractor_kind="mean", attractor_type="inv", bin_centers_type="softplus", bin_embedding_dim=128,
              img_size=[56, 84], inverse_midas=False, max_temp=50.0, min_temp=0.0212, memory_efficient=True,
              n_attractors=[16, 8, 4, 1], n_bins=64, output_distribution="logbinomial", force_keep_ar=True),
    seed=51, inputs=dict(rect=(56, 84), square=(70, 70)),
)


# -- end-to-end V2 with the ZoeDepth (DepthAnything ViT-L core) coarse branch: the fully vendored sibling of
#    configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py (coarse_chl[0] = 32 = ZoeDepth's out_conv feature)
_ZOE_L = {**ZOE_DA["zcfg"], "midas_model_type": "vitl", "img_size": [112, 224]}
E2E_V2Z = dict(
    raw=[256, 512], split=[2, 2], pps=[112, 224], max_depth=80.0, seed=0, modes=["m1", "r4"], zcfg=_ZOE_L,
    fusion=dict(coarse_chl=[32, 256, 256, 256, 256, 256], fine_chl=[32, 32, 64, 96, 960],
                fine_chl_after_coarse2fine=[32, 256, 256, 256, 256, 256], temp_chl=[32, 64, 64, 128, 256, 512],
                dec_chl=[512, 256, 128, 64, 32]),
)
E2E_V2Z["ref_config"] = {**E2E_V2["ref_config"], "coarse_branch": dict(type="DA-ZoeDepth", **_ZOE_L),
                         "refiner": dict(fine_branch=E2E_V2["ref_config"]["refiner"]["fine_branch"],
                                         fusion_model=dict(type="BiDirectionalFusion",
                                                           encoder_name="mobilenetv4_conv_small.e2400_r224_in1k",
                                                           coarse2fine=True, coarse2fine_type="coarse-gated",
                                                           **E2E_V2Z["fusion"]))}


def e2e_v2z_sd(seed: int = 47):
    spec = OrderedDict()
    spec.update(W.zoedepth_spec("coarse_branch.", _ZOE_L))
    spec.update(W.mnv4_spec("refiner_fine_branch.refiner_encoder.", in_chans=4))
    f = E2E_V2Z["fusion"]
    spec.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"],
                                    f["fine_chl_after_coarse2fine"], f["temp_chl"], f["dec_chl"]))
    return W.synth_state_dict(spec, seed=seed)


# -- LightWeightRefiner with a ConvNeXt encoder (reduced dims; real cfg: configs/patchrefinerv2_zoedepth/v2_convx_u4k.py:90-97) ----
CONVNEXT_REFINER = dict(arch=W.CONVNEXT_TINY_TEST, seed=53, in_seed=540, b=2, h=64, w=96)


def convnext_refiner_sd(arch=None, seed=None, prefix=""):
    arch = arch or CONVNEXT_REFINER["arch"]
    spec = W.convnext_spec(prefix + "refiner_encoder.", arch, in_chans=4)
    spec[prefix + "upsample_convx.0.weight"] = (arch["dims"][0], arch["dims"][0] // 2, 2, 2)
    spec[prefix + "upsample_convx.0.bias"] = (arch["dims"][0] // 2,)
    return W.synth_state_dict(spec, seed=CONVNEXT_REFINER["seed"] if seed is None else seed)


def convnext_refiner_inputs(c=None):
    c = c or CONVNEXT_REFINER
    crop = rand_image(c["in_seed"], c["b"], c["h"], c["w"])
    depth = torch.rand(c["b"], 1, c["h"], c["w"], generator=torch.Generator().manual_seed(c["in_seed"] + 1)) * 10
    return crop, depth


# -- end-to-end V2 with the ConvNeXt refiner encoder (configs/patchrefinerv2_zoedepth/v2_convx_u4k.py:90-104), reduced dims
_CX = W.CONVNEXT_TINY_TEST
E2E_V2CX = dict(E2E_V2, arch=_CX, modes=["m1", "r4"],
                fusion=dict(E2E_V2["fusion"], fine_chl=[_CX["dims"][0] // 2] + list(_CX["dims"])))
E2E_V2CX["ref_config"] = {**E2E_V2["ref_config"], "refiner": dict(
    fine_branch=dict(type="LightWeightRefiner", coarse_condition=True, with_decoder=False, encoder_name="convnext_large",
                     encoder_channels=[_CX["dims"][0] // 2] + list(_CX["dims"]), arch=_CX),
    fusion_model=dict(type="BiDirectionalFusion", encoder_name="convnext_large", coarse2fine=True,
                      coarse2fine_type="coarse-gated", **E2E_V2CX["fusion"]))}


def e2e_v2cx_sd(seed: int = 59):
    spec = OrderedDict()
    spec.update(W.dav2_spec("coarse_branch.", _E2E2_DA2))
    spec.update(W.convnext_spec("refiner_fine_branch.refiner_encoder.", _CX, in_chans=4))
    spec["refiner_fine_branch.upsample_convx.0.weight"] = (_CX["dims"][0], _CX["dims"][0] // 2, 2, 2)
    spec["refiner_fine_branch.upsample_convx.0.bias"] = (_CX["dims"][0] // 2,)
    f = E2E_V2CX["fusion"]
    spec.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"],
                                    f["fine_chl_after_coarse2fine"], f["temp_chl"], f["dec_chl"]))
    return W.synth_state_dict(spec, seed=seed)


# -- LightWeightRefiner with an EfficientNet encoder (reduced width/depth; real cfg: configs/patchrefinerv2_zoedepth/v2_eff_u4k.py:90-101)
EFFNET_REFINER = dict(arch=W.EFFNET_TINY_TEST, seed=61, in_seed=620, b=2, h=64, w=96)


def effnet_refiner_sd(arch=None, seed=None, prefix=""):
    arch = arch or EFFNET_REFINER["arch"]
    return W.synth_state_dict(W.effnet_spec(prefix + "refiner_encoder.", arch, in_chans=4),
                              seed=EFFNET_REFINER["seed"] if seed is None else seed)


def effnet_refiner_inputs(c=None):
    c = c or EFFNET_REFINER
    crop = rand_image(c["in_seed"], c["b"], c["h"], c["w"])
    depth = torch.rand(c["b"], 1, c["h"], c["w"], generator=torch.Generator().manual_seed(c["in_seed"] + 1)) * 10
    return crop, depth


# -- end-to-end V2 with the EfficientNet refiner encoder (configs/patchrefinerv2_zoedepth/v2_eff_u4k.py:90-104), reduced dims
_EF = W.EFFNET_TINY_TEST
_EF_CHL = [B["cout"] for B in W.effnet_blocks(_EF) if B["tap"]]
E2E_V2EF = dict(E2E_V2, arch=_EF, modes=["m1", "r4"], fusion=dict(E2E_V2["fusion"], fine_chl=_EF_CHL))
E2E_V2EF["ref_config"] = {**E2E_V2["ref_config"], "refiner": dict(
    fine_branch=dict(type="LightWeightRefiner", coarse_condition=True, with_decoder=False, encoder_name="tf_efficientnet_b5_ap",
                     arch=_EF),
    fusion_model=dict(type="BiDirectionalFusion", encoder_name="tf_efficientnet_b5_ap", coarse2fine=True,
                      coarse2fine_type="coarse-gated", **E2E_V2EF["fusion"]))}


def e2e_v2ef_sd(seed: int = 67):
    spec = OrderedDict()
    spec.update(W.dav2_spec("coarse_branch.", _E2E2_DA2))
    spec.update(W.effnet_spec("refiner_fine_branch.refiner_encoder.", _EF, in_chans=4))
    f = E2E_V2EF["fusion"]
    spec.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"],
                                    f["fine_chl_after_coarse2fine"], f["temp_chl"], f["dec_chl"]))
    return W.synth_state_dict(spec, seed=seed)


# -- BaselinePretrain (estimator/models/baseline_pretrain.py:44-93,377-464): the bare backbone, target 'coarse' (one forward
#    on image_lr: BASELINE config[0]'s plumbing check) and target 'fine' (the backbone on every tile; N * process_num random
#    tiles for r<N>, blend mask border 0.1).  Backbone = the reference's 'DA-ZoeDepth' (ZoeDepth.build over DepthAnything ViT-S).
BASELINE = dict(raw=[216, 384], split=[2, 2], pps=[56, 84], seed=5, sd_seed=71, zcfg={**ZOE_DA["zcfg"], "img_size": [56, 84]},
                fine_modes=["m1", "m2", "r2"], max_depth=80.0)


def baseline_kwargs(target: str) -> dict:
    c = BASELINE
    branch = dict(type="DA-ZoeDepth", **c["zcfg"])
    return dict(coarse_branch=branch, fine_branch=branch, sigloss=dict(type="SILogLoss"), min_depth=1e-3, max_depth=c["max_depth"],
                image_raw_shape=c["raw"], patch_process_shape=c["pps"], patch_split_num=c["split"], target=target)


def baseline_sd(prefix: str = ""):
    return W.synth_state_dict(W.zoedepth_spec(prefix, BASELINE["zcfg"]), seed=BASELINE["sd_seed"])


# -- ZoeDepth over the MiDaS DPT_BEiT_L_384 core (type='ZoeDepth'), reduced BEiT (dim 128, 4 blocks, 2 heads, 4 x 4 window),
#    non-square inputs so that the relative-position table is resized (configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py:10-66)
_BEIT_TINY = dict(dim=128, depth=4, heads=2, taps=[0, 1, 2, 3], window=[4, 4], features=256, out_channels=[32, 64, 128, 128])
ZOE_BEIT = dict(
    zcfg={**{k: v for k, v in ZOE_DA["zcfg"].items() if k != "midas_model_type"}, "midas_model_type": "DPT_BEiT_L_384",
          "img_size": [64, 96], "beit": _BEIT_TINY},
    seed=83, inputs=dict(rect=(64, 96), square=(64, 64), big=(96, 160)),
)

# -- end-to-end V2 exactly as configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py wires it: coarse_branch type='ZoeDepth'
#    (MidasCore, BEiT), ResizeZoe (hard-coded 384 x 512, midas.py:171-174) => patch_process_shape 384 x 512; reduced BEiT
_ZOE_B = {**ZOE_BEIT["zcfg"], "img_size": [384, 512]}
E2E_V2B = dict(
    raw=[540, 960], split=[2, 2], pps=[384, 512], max_depth=80.0, seed=0, modes=["m1", "r4"], zcfg=_ZOE_B,
    fusion=E2E_V2Z["fusion"],
)
E2E_V2B["ref_config"] = {**E2E_V2Z["ref_config"], "image_raw_shape": E2E_V2B["raw"], "patch_process_shape": E2E_V2B["pps"],
                         "patch_raw_shape": [270, 480], "coarse_branch": dict(type="ZoeDepth", **_ZOE_B)}


def e2e_v2b_sd(seed: int = 89):
    spec = OrderedDict()
    spec.update(W.zoedepth_spec("coarse_branch.", _ZOE_B))
    spec.update(W.mnv4_spec("refiner_fine_branch.refiner_encoder.", in_chans=4))
    f = E2E_V2B["fusion"]
    spec.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"],
                                    f["fine_chl_after_coarse2fine"], f["temp_chl"], f["dec_chl"]))
    return W.synth_state_dict(spec, seed=seed)


# -- full-width parity on synthetic weights: make the depth RANGE exercise the bins ------------------------------------------
def widen_depth_range(sd, prefix: str = "coarse_branch.", gain: float = 4.0, lo: float = 0.5, hi: float = 60.0,
                      offset_gain: float = 6.0, fusion_prefix: str = "refiner_fusion_model."):
    """Synthetic weights leave the ZoeDepth metric-bins head nearly constant (bin centres 0.1 .. 5 m, a flat log-binomial:
    depth 0.9 .. 1.6 of max_depth 80) and the refinement offset ~0.07 m, so an AbsRel on coarse + offset is blind to the
    per-patch network.  This re-scales the HEADS only (same shapes, same graph): the 64 seed bin centres spread geometrically
    over lo .. hi metres (softplus^-1 into the regressor's bias, its input-dependent part damped), the log-binomial
    probability logits x gain (the mode then moves over the bins from pixel to pixel: 1 % .. 99 % quantiles of the coarse
    depth ~5 .. 49 m at 384 x 512) and the fusion net's final 3 x 3 conv x offset_gain (offsets of metres).  Returns a new
    state dict; the product and the oracle both load it."""
    import numpy as np
    sd = OrderedDict(sd)
    k = prefix + "seed_bin_regressor._net.2."
    if k + "bias" in sd:
        nb = sd[k + "bias"].numel()
        c = torch.from_numpy(np.geomspace(lo, hi, nb).astype(np.float32))
        sd[k + "bias"] = torch.log(torch.expm1(c))
        sd[k + "weight"] = sd[k + "weight"] * 0.1
        wt = sd[prefix + "conditional_log_binomial.mlp.2.weight"].clone()
        wt[:2] *= gain
        sd[prefix + "conditional_log_binomial.mlp.2.weight"] = wt
    if fusion_prefix + "final_conv.weight" in sd:
        sd[fusion_prefix + "final_conv.weight"] = sd[fusion_prefix + "final_conv.weight"] * offset_gain
    return sd
