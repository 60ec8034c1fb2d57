"""Named full-size workloads (BASELINE.json configs made concrete) + algorithmic
FLOP accounting (2*MAC over conv / linear / attention matmuls, what torch's FlopCounterMode reports).
"""
from __future__ import annotations

from collections import OrderedDict

from . import weights as W

DA2_L = dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024])
DA2_S = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384])

BIDIR_DAV2 = dict(coarse_chl=[128, 256, 256, 256, 256, 256], fine_chl=[32, 32, 64, 96, 960],
                  fine_chl_after_coarse2fine=[128, 256, 256, 256, 256, 256], temp_chl=[32, 64, 64, 128, 256, 512],
                  dec_chl=[512, 256, 128, 64, 32])

WORKLOADS = {
    # configs/patchrefinerv2_dav2/plus_mobile_u4k_base_coarse_e2e_c2f_pretrain.py at 4K 4x4 r32:
    # the V2 model (MNv4-S refiner + BiDirectionalFusion) over the fully vendored DAv2 ViT-L coarse
    # branch -- BASELINE config[2]'s shape (4K, 4x4, r32, 81 patches) with config[3]'s backbone.
    "v2_dav2l_4k_r32": dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[448, 448], mode="r32",
                            coarse=DA2_L, fusion=BIDIR_DAV2, patches=81),
    "v2_dav2l_4k_r64": dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[448, 448], mode="r64",
                            coarse=DA2_L, fusion=BIDIR_DAV2, patches=113),
    "v2_dav2l_4k_r128": dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[448, 448], mode="r128",
                             coarse=DA2_L, fusion=BIDIR_DAV2, patches=177),
    # BASELINE config[1]: DAv2 ViT-S, 1080x1920, 2x2, m1 -> V1 PatchRefiner
    "v1_dav2s_1080p_m1": dict(kind="PatchRefiner", raw=[1080, 1920], split=[2, 2], pps=[448, 448], mode="m1",
                              coarse=DA2_S, fine=DA2_S,
                              fusion=dict(input_chl=[64, 128, 128, 128, 128, 128], temp_chl=[32, 64, 64, 64, 64, 64],
                                          dec_chl=[64, 64, 64, 64, 32]), patches=4),
    # configs/patchrefiner_dav2/pr_u4k.py: V1 with ViT-L on every patch (the ViT-block-heavy variant)
    "v1_dav2l_4k_r32": dict(kind="PatchRefiner", raw=[2160, 3840], split=[4, 4], pps=[448, 448], mode="r32",
                            coarse=DA2_L, fine=DA2_L,
                            fusion=dict(input_chl=[256, 512, 512, 512, 512, 512], temp_chl=[128, 256, 256, 256, 256, 256],
                                        dec_chl=[256, 256, 256, 256, 128]), patches=81),
}
# BASELINE config[2] in its fully vendored form (SURVEY.md 8d C3 "pinned sibling"): configs/patchrefinerv2_zoedepth/
# v2_mobile_u4k.py (BiDirectionalFusion zoe cfg: coarse_chl[0] = 32) with the ZoeDepth metric-bins head over the
# DepthAnything ViT-L core (type='DA-ZoeDepth') instead of the un-vendored MiDaS DPT-BEiT-L; P must be a
# multiple of 14 -> 392 x 518.
ZOE_DA_L = dict(midas_model_type="vitl", min_depth=1e-3, max_depth=80, do_resize=False, attractor_alpha=1000,
                attractor_gamma=2, attractor_kind="mean", attractor_type="inv", bin_centers_type="softplus",
                bin_embedding_dim=128, img_size=[392, 518], max_temp=50.0, min_temp=0.0212, n_attractors=[16, 8, 4, 1],
                n_bins=64)
BIDIR_ZOE = dict(coarse_chl=[32, 256, 256, 256, 256, 256], fine_chl=[32, 32, 64, 96, 960],
                 fine_chl_after_coarse2fine=[32, 256, 256, 256, 256, 256], temp_chl=[32, 64, 64, 128, 256, 512],
                 dec_chl=[512, 256, 128, 64, 32])
WORKLOADS["v2_zoeda_4k_r32"] = dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[392, 518], mode="r32",
                                    coarse=None, zoe=ZOE_DA_L, fusion=BIDIR_ZOE, patches=81)
# configs/patchrefinerv2_zoedepth/v2_convx_u4k.py: the same model with the ConvNeXt-L refiner encoder (fine_chl :101).
# The 4x4-stride-4 stem needs P % 4 == 0 on top of the ViT's P % 14 == 0 (the prediction is 4 * floor(P / 4) wide, in the
# reference as well) -> 392 x 504.
BIDIR_ZOE_CONVX = dict(BIDIR_ZOE, fine_chl=[96, 192, 384, 768, 1536])
WORKLOADS["v2_convx_zoeda_4k_r32"] = dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[392, 504], mode="r32",
                                          coarse=None, zoe=dict(ZOE_DA_L, img_size=[392, 504]), fusion=BIDIR_ZOE_CONVX, patches=81,
                                          refiner_encoder="convnext_large")
# configs/patchrefinerv2_zoedepth/v2_eff_u4k.py: EfficientNet-B5-AP refiner encoder (fine_chl :101)
WORKLOADS["v2_eff_zoeda_4k_r32"] = dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[392, 518], mode="r32",
                                        coarse=None, zoe=ZOE_DA_L, fusion=dict(BIDIR_ZOE, fine_chl=[24, 40, 64, 176, 512]),
                                        patches=81, refiner_encoder="tf_efficientnet_b5_ap")
# BASELINE config[2] AS THE REFERENCE CONFIGURES IT (configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py): coarse_branch type='ZoeDepth'
# = the metric-bins head over MiDaS DPT_BEiT_L_384 (:10-17), ResizeZoe -> P = 384 x 512, MNv4-S refiner, BiDirectionalFusion zoe cfg
ZOE_BEIT_L = dict(ZOE_DA_L, midas_model_type="DPT_BEiT_L_384", img_size=[384, 512])
WORKLOADS["v2_zoe_4k_r32"] = dict(kind="PatchRefinerPlus", raw=[2160, 3840], split=[4, 4], pps=[384, 512], mode="r32",
                                  coarse=None, zoe=ZOE_BEIT_L, zoe_type="ZoeDepth", fusion=BIDIR_ZOE, patches=81)
# (bench.py runs every workload with 41 tiles per launch batch unless an entry names ``max_batch``: 81 = 41 + 40 on two streams -- the
#  small pyramid levels (48 x 64 and below: 84-336 workgroups per launch at 14 tiles) fill the chip better: 198.5 -> 194.3 ms here,
#  +1..3 % on the other 81-tile workloads, flat at r64; results are batch independent)
# configs/patchrefiner_zoedepth/pr_u4k.py (the README's example command): V1 with ZoeDepth / BEiT-L on every tile
WORKLOADS["v1_zoe_4k_r32"] = dict(kind="PatchRefiner", raw=[2160, 3840], split=[4, 4], pps=[384, 512], mode="r32",
                                  coarse=None, zoe=ZOE_BEIT_L, zoe_type="ZoeDepth", fine_zoe=ZOE_BEIT_L,
                                  fusion=dict(input_chl=[64, 512, 512, 512, 512, 512], temp_chl=[32, 256, 256, 256, 256, 256],
                                              dec_chl=[256, 256, 256, 256, 32]), patches=81)
DEFAULT_WORKLOAD = "v2_zoe_4k_r32"
MNV4_NAME = "mobilenetv4_conv_small.e2400_r224_in1k"


def model_config(name: str, prec: str = "f32", max_batch=None, n_streams=1) -> dict:
    w = WORKLOADS[name]
    raw, split = w["raw"], w["split"]
    cfg = dict(
        image_raw_shape=raw, patch_process_shape=w["pps"], patch_raw_shape=[raw[0] // split[0], raw[1] // split[1]],
        patch_split_num=split, fusion_feat_level=6, min_depth=1e-3, max_depth=80.0, pretrain_coarse_model=None,
        strategy_refiner_target="offset_coarse",
        coarse_branch=(dict(type=w.get("zoe_type", "DA-ZoeDepth"), **w["zoe"]) if w.get("zoe") else
                       dict(type="DA2", pretrained=None, model_cfg=w["coarse"])),
        sigloss=dict(type="SILogLoss"), pretrained=None, pre_norm_bbox=True, prec=prec, max_batch=max_batch, n_streams=n_streams)
    if w["kind"] == "PatchRefinerPlus":
        cfg.update(e2e_training=True, pretrain_stage=False, gmloss=dict(type="GradMatchLoss"), sigweight=1,
                   whole_pretrained=None,
                   refiner=dict(fine_branch=dict(type="LightWeightRefiner", coarse_condition=True, with_decoder=False,
                                                 encoder_name=w.get("refiner_encoder", MNV4_NAME)),
                                fusion_model=dict(type="BiDirectionalFusion",
                                                  encoder_name=w.get("refiner_encoder", MNV4_NAME),
                                                  coarse2fine=True, coarse2fine_type="coarse-gated", **w["fusion"])))
    else:
        fine = (dict(type=w["zoe_type"], **w["fine_zoe"]) if w.get("fine_zoe") else dict(type="DA2", pretrained=None, model_cfg=w["fine"]))
        cfg.update(pretrain_fine_model=None, refiner=dict(fine_branch=fine, fusion_model=dict(type="FusionUnet", **w["fusion"])))
    return dict(type=w["kind"], config=cfg)


def state_spec(name: str) -> "OrderedDict[str, tuple]":
    w = WORKLOADS[name]
    s = OrderedDict()
    if w.get("zoe"):
        s.update(W.zoedepth_spec("coarse_branch.", w["zoe"]))
    else:
        s.update(W.dav2_spec("coarse_branch.", w["coarse"]))
    if w["kind"] == "PatchRefinerPlus":
        if "efficientnet" in w.get("refiner_encoder", ""):
            s.update(W.effnet_spec("refiner_fine_branch.refiner_encoder.", W.EFFNET_B5, in_chans=4))
        elif "convnext" in w.get("refiner_encoder", ""):
            s.update(W.convnext_spec("refiner_fine_branch.refiner_encoder.", W.CONVNEXT_LARGE, in_chans=4))
            d0 = W.CONVNEXT_LARGE["dims"][0]
            s["refiner_fine_branch.upsample_convx.0.weight"] = (d0, d0 // 2, 2, 2)
            s["refiner_fine_branch.upsample_convx.0.bias"] = (d0 // 2,)
        else:
            s.update(W.mnv4_spec("refiner_fine_branch.refiner_encoder.", in_chans=4))
        f = w["fusion"]
        s.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"],
                                     f["fine_chl_after_coarse2fine"], f["temp_chl"], f["dec_chl"]))
    else:
        if w.get("fine_zoe"):
            s.update(W.zoedepth_spec("refiner_fine_branch.", w["fine_zoe"]))
        else:
            s.update(W.dav2_spec("refiner_fine_branch.", w["fine"]))
        s.update(W.fusion_unet_spec("refiner_fusion_model.", **w["fusion"]))
    return s
