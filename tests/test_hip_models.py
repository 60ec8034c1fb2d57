"""GPU parity of the assembled networks and the frame drivers against the oracle and the golden
vectors generated from the reference (tests/golden).  Tolerance: the north star's per-pixel
AbsRel <= 1e-4 (fp32 MFMA path is ~1e-6)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dav2 as o_dav2, fusion as o_fusion, mnv4 as o_mnv4, tiling as o_tiling  # noqa: E402
from oracle.cases import (E2E_V1, E2E_V2, E2E_V2Z, TINY_BIDIR, TINY_DAV2, TINY_FUSION_UNET, ZOE_DA, e2e_v1_sd,  # noqa: E402
                          e2e_v2_sd, e2e_v2z_sd, rand_image, tiny_dav2_sd)
from oracle import zoe as o_zoe  # noqa: E402
from patchrefinerv2_amd import weights as W  # noqa: E402

DEV = "cuda"
ABSREL_TOL = 1e-4
torch.set_grad_enabled(False)


def absrel(out, ref, min_depth=1e-3):
    out, ref = out.detach().cpu().double(), torch.as_tensor(ref).double()
    m = ref > min_depth
    return float(((out - ref).abs()[m] / ref[m]).mean()), float((out - ref).abs().max())


def close(got, ref, tol=2e-5, what=""):
    got, ref = got.detach().cpu(), torch.as_tensor(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = float((got - ref).abs().max())
    assert err <= tol * max(1.0, float(ref.abs().max())), f"{what}: max|d| {err:.3e}"


@pytest.fixture(scope="module")
def P():
    from patchrefinerv2_amd import ops
    ops.L.load()
    return ops


def test_dav2_tiny(P, golden):
    from patchrefinerv2_amd.dav2 import DepthAnythingV2
    g = golden("dav2_tiny")
    sd = tiny_dav2_sd()
    m = DepthAnythingV2(**TINY_DAV2["model_cfg"])
    m.load_state_dict(sd, strict=True)
    cfg = W.dav2_cfg(TINY_DAV2["model_cfg"])
    for tag, (h, w) in TINY_DAV2["inputs"].items():
        x = rand_image(TINY_DAV2["seed"], 2, h, w)
        out = m(x.to(DEV), return_final_centers=True)
        ref = o_dav2.dav2_forward(sd, "", x, cfg)
        close(out["metric_depth"], g[f"{tag}_depth"], 2e-5, f"{tag} depth vs golden")
        for k, v in ref["temp_features"].items():
            close(out["temp_features"][k].to_nchw(), v, 2e-5, f"{tag} {k}")
        ar, _ = absrel(out["metric_depth"], ref["metric_depth"])
        assert ar < ABSREL_TOL


def test_dav2_vits_block_stack(P):
    """real-width ViT-S (D=384, 6 heads) x 3 blocks + DPT head at 448x448 (1025 tokens)."""
    from patchrefinerv2_amd.dav2 import DepthAnythingV2
    mc = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], max_depth=80.0,
              vit=dict(depth=4, taps=[0, 1, 2, 3]))
    sd = W.synth_state_dict(W.dav2_spec("", mc), seed=5)
    m = DepthAnythingV2(**mc)
    m.load_state_dict(sd)
    x = rand_image(3, 1, 448, 448)
    out = m(x.to(DEV))
    ref = o_dav2.dav2_forward(sd, "", x, W.dav2_cfg(mc))
    ar, mx = absrel(out["metric_depth"], ref["metric_depth"])
    assert ar < 1e-5, (ar, mx)
    close(out["temp_features"]["x_blocks_feat_3"].to_nchw(), ref["temp_features"]["x_blocks_feat_3"], 5e-5)


def test_vit_blocks_presplit_path_is_bit_identical(P):
    """a batch of tiles (>= ops.SS_MIN_ROWS token rows) runs the ViT blocks on the pre-split operand path (LayerNorm / attention /
    GELU epilogues write the matrix pipe's operand format, gemm_ss_kernel consumes it by LDS-DMA); a single tile runs the
    fp32-operand kernels: same bits, so per-tile results do not depend on the batch (DINOv2 blocks and BEiT blocks)"""
    from patchrefinerv2_amd import ops
    from patchrefinerv2_amd.dav2 import DepthAnythingV2
    from patchrefinerv2_amd.zoedepth import ZoeDepth
    mc = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], max_depth=80.0, vit=dict(depth=4, taps=[0, 1, 2, 3]))
    m = DepthAnythingV2(**mc, prec="bf16x3")
    m.load_state_dict(W.synth_state_dict(W.dav2_spec("", mc), seed=5))
    x = rand_image(3, 3, 448, 448).to(DEV)                    # 3 x 1025 = 3075 token rows
    old_min = ops.SS_MIN_ROWS
    ops.SS_MIN_ROWS = 2048
    a = m(x)["metric_depth"]
    try:
        ops.SS_DISABLED = True
        b = m(x)["metric_depth"]
    finally:
        ops.SS_DISABLED = False
    assert torch.equal(a, b)
    assert torch.equal(a[1:2], m(x[1:2])["metric_depth"])     # one image alone (fp32-operand kernels)
    from oracle.cases import ZOE_BEIT
    zc = dict(ZOE_BEIT["zcfg"])
    z = ZoeDepth.build(**zc, prec="bf16x3")
    z.load_state_dict(W.synth_state_dict(W.zoedepth_spec("", zc), seed=6), strict=True)
    xb = rand_image(4, 6, 320, 320).to(DEV)                   # 6 x 401 = 2406 token rows
    a = z(xb)["metric_depth"]
    try:
        ops.SS_DISABLED = True
        b = z(xb)["metric_depth"]
    finally:
        ops.SS_DISABLED = False
    ops.SS_MIN_ROWS = old_min
    assert torch.equal(a, b)


def test_fusion_unet(P, golden):
    from patchrefinerv2_amd.fusion import FusionUnet
    c = TINY_FUSION_UNET
    sd = W.synth_state_dict(W.fusion_unet_spec("", c["input_chl"], c["temp_chl"], c["dec_chl"]), seed=c["seed"])
    m = FusionUnet(c["input_chl"], c["temp_chl"], c["dec_chl"])
    m.load_state_dict(sd)
    i = c["make_inputs"]()
    f = lambda ts: [P.Feat.from_nchw(t.to(DEV)) for t in ts]  # noqa: E731
    out = m(f(i["c_feat"]), f(i["f_feat"]), i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV))
    close(out, golden("fusion_unet")["out"], 2e-5)


def test_bidir_fusion(P, golden):
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    c = TINY_BIDIR
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"],
                                                c["dec_chl"]), seed=c["seed"])
    m = BiDirectionalFusion(coarse2fine_type="coarse-gated", coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                            fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"])
    m.load_state_dict(sd)
    g = golden("bidir_fusion")
    for tag in ("same", "resized"):
        i = c["make_inputs"](tag)
        f = lambda ts: [P.Feat.from_nchw(t.to(DEV)) for t in ts]  # noqa: E731
        ff = f(i["f_feat"])
        out = m(f(i["c_feat"]), [None] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV),
                f_sizes=[(t.shape[-2], t.shape[-1]) for t in i["f_feat"]])
        close(out, g[tag], 3e-5, tag)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("c2f_type,name", [("coarse-fusion", "bidir_fusion_coarse_fusion"), ("self-agg", "bidir_fusion_self_agg"),
                                           ("only-gate", "bidir_fusion_only_gate")])
def test_bidir_fusion_c2f_ablation_types(P, golden, c2f_type, name, prec):
    """coarse2fine_type 'coarse-fusion' (the fusion_conv output replaces the gated product) and 'self-agg' (no fusion_conv, the coarse
    pyramid only enters at fusion_layers_1) against the reference's outputs (tests/golden, oracle/make_golden.py g_bidir)"""
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    c = TINY_BIDIR
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"],
                                                coarse2fine_type=c2f_type), seed=c["seed"])
    m = BiDirectionalFusion(coarse2fine_type=c2f_type, coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                            fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec=prec)
    m.load_state_dict(sd)
    g = golden(name)
    for tag in ("same", "resized"):
        i = c["make_inputs"](tag)
        f = lambda ts: [P.Feat.from_nchw(t.to(DEV)) for t in ts]  # noqa: E731
        ff = f(i["f_feat"])
        out = m(f(i["c_feat"]), [None] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV),
                f_sizes=[(t.shape[-2], t.shape[-1]) for t in i["f_feat"]])
        close(out, g[tag], 3e-5, f"{c2f_type}/{prec}/{tag}")


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_bidir_fusion_without_c2f(P, golden, prec):
    """BiDirectionalFusion(coarse2fine=False) (configs/patchrefinerv2_zoedepth_ablation/plus_*_u4k_base_coarse.py) against the reference's
    outputs; level 0 of the refiner pyramid is handed over as None and built here as the x2 copy of level 1 (lightweight_refiner.py:314-316)"""
    from oracle.cases import TINY_BIDIR_NOC2F as c
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"],
                                                coarse2fine=False), seed=c["seed"])
    m = BiDirectionalFusion(coarse2fine=False, coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                            fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec=prec)
    m.load_state_dict(sd)
    g = golden("bidir_fusion_no_c2f")
    for tag in ("same", "resized"):
        i = c["make_inputs"](tag)
        f = lambda ts: [P.Feat.from_nchw(t.to(DEV)) for t in ts]  # noqa: E731
        ff = f(i["f_feat"])
        sizes = [(t.shape[-2], t.shape[-1]) for t in i["f_feat"]]
        for first in (None, ff[0]):
            out = m(f(i["c_feat"]), [first] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV), f_sizes=sizes)
            close(out, g[tag], 3e-5, f"no c2f/{prec}/{tag}")


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_bidir_fusion_heavy(P, golden, prec):
    """BiDirectionalFusionHeavy (three convs per encoder layer, five per decoder stage; coarse2fine=False as its configs have it) against
    the reference's outputs: the sum with the base and the raw offset (~1e-2 with these weights: tolerance relative to ITS range)"""
    from oracle.cases import TINY_BIDIR_NOC2F as c
    from patchrefinerv2_amd.fusion import BiDirectionalFusionHeavy
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"],
                                                coarse2fine=False, heavy=True), seed=c["seed"])
    m = BiDirectionalFusionHeavy(coarse2fine=False, coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                                 fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec=prec)
    m.load_state_dict(sd)
    g = golden("bidir_fusion_heavy")
    for tag in ("same", "resized"):
        i = c["make_inputs"](tag)
        f = lambda ts: [P.Feat.from_nchw(t.to(DEV)) for t in ts]  # noqa: E731
        ff = f(i["f_feat"])
        sizes = [(t.shape[-2], t.shape[-1]) for t in i["f_feat"]]
        out = m(f(i["c_feat"]), [None] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV), f_sizes=sizes)
        close(out, g[tag], 3e-5, f"heavy/{prec}/{tag}")
        off = m(f(i["c_feat"]), [None] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), f_sizes=sizes)
        ref = torch.as_tensor(g[tag + "_offset"])
        err = float((off.cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err < (2e-5 if prec == "f32" else 1e-4), (prec, tag, err)


def test_bidir_fusion_x2_format_is_bit_identical(P):
    """BiDirectionalFusion with the GatedConvUnits' concat buffers in the pre-split X2 operand format (default when the coarse
    pyramid arrives as ROI sources at the refiner's sizes) == the same network on fp32 buffers (PRV2_X2=0), bit for bit"""
    from patchrefinerv2_amd import ops
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    c = TINY_BIDIR
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"],
                                                c["dec_chl"]), seed=c["seed"])
    m = BiDirectionalFusion(coarse2fine_type="coarse-gated", coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                            fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec="bf16x3")
    m.load_state_dict(sd)
    i = c["make_inputs"]("same")
    ff = [P.Feat.from_nchw(t.to(DEV)) for t in i["f_feat"]]
    sizes = [(t.shape[-2], t.shape[-1]) for t in i["f_feat"]]
    K = i["pred1"].shape[0]
    # the coarse pyramid as ROI sources (what the frame driver hands over): whole-map boxes at each level's own size
    frame = [P.Feat.from_nchw(t[:1].to(DEV)) for t in i["c_feat"]]
    boxes = torch.tensor([[0.0, 0.0, float(sizes[0][1]), float(sizes[0][0])]] * K, device=DEV)
    rois = lambda: [ops.RoiSource(f, boxes, f.h / sizes[0][0], f.h, f.w) for f in frame]  # noqa: E731

    def run():
        return m(rois(), [None] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV), f_sizes=sizes).clone()

    used = []
    real = ops.conv3x3_ln_gate

    def spy(x, *a, **k):
        used.append(bool(x.x2))
        return real(x, *a, **k)
    ops.conv3x3_ln_gate = spy
    try:
        a = run()
        assert any(used), "the X2 path was not taken"
        ops.X2_FORMAT = False
        used.clear()
        b = run()
        assert not any(used)
    finally:
        ops.X2_FORMAT = True
        ops.conv3x3_ln_gate = real
    assert torch.equal(a, b)


def test_bidir_fusion_upconv_route_matches_the_loader_route(P, golden):
    """BiDirectionalFusion in bf16x3 with the 3x3 convs of upsampled tensors on csrc/upconv.hip (output_conv1 and every decoder stage
    with >= 256 interpolated channels, split by weight columns) against the same network with PRV2_UPCONV off (the fused-upsample
    loader kernel) and against the reference's golden output: same tolerance on both routes"""
    from patchrefinerv2_amd import ops
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    c = TINY_BIDIR
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"],
                                                c["dec_chl"]), seed=c["seed"])
    i = c["make_inputs"]("same")
    sizes = [(t.shape[-2], t.shape[-1]) for t in i["f_feat"]]
    outs, calls = {}, {}
    real = ops.upconv3x3
    for on in (True, False):
        ops.UPCONV = on
        n = []

        def spy(*a, **k):
            n.append(1)
            return real(*a, **k)
        ops.upconv3x3 = spy
        try:
            m = BiDirectionalFusion(coarse2fine_type="coarse-gated", coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                                    fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec="bf16x3")
            m.load_state_dict(sd)   # (the decoder's weight split is decided when the weights are packed)
            f = lambda ts: [P.Feat.from_nchw(t.to(DEV)) for t in ts]  # noqa: E731
            ff = f(i["f_feat"])
            outs[on] = m(f(i["c_feat"]), [None] + ff[1:], i["pred1"].to(DEV), i["pred2"].to(DEV), update_base=i["pred1"].to(DEV), f_sizes=sizes).clone()
        finally:
            ops.UPCONV = True
            ops.upconv3x3 = real
        calls[on] = len(n)
    assert calls[True] >= 3 and calls[False] == 0, calls   # output_conv1 + the 770- and 642-channel decoder stages (+ 322)
    close(outs[True], golden("bidir_fusion")["same"], 3e-5, "upconv route vs golden")
    close(outs[False], golden("bidir_fusion")["same"], 3e-5, "loader route vs golden")
    close(outs[True], outs[False].cpu(), 3e-5, "upconv route vs loader route")


def test_lightweight_refiner(P):
    from patchrefinerv2_amd.refiner import LightWeightRefiner
    sd = W.synth_state_dict(W.mnv4_spec("refiner_encoder.", in_chans=4), seed=9)
    m = LightWeightRefiner("mobilenetv4_conv_small.e2400_r224_in1k", coarse_condition=True)
    m.load_state_dict(sd)
    img = rand_image(2, 2, 96, 128)
    depth = torch.rand(2, 1, 96, 128, generator=torch.Generator().manual_seed(3)) * 40
    ref_feats, _ = o_mnv4.lightweight_refiner(sd, "", img, depth)  # low -> high, 2x copy last
    mean = torch.tensor(W.MNV4_SMALL["mean"]).view(1, 3, 1, 1)
    std = torch.tensor(W.MNV4_SMALL["std"]).view(1, 3, 1, 1)
    x4 = torch.cat([(img - mean) / std, depth], dim=1)
    feats, sizes = m(P.Feat.from_nchw(x4.to(DEV)))
    assert feats[0] is None and sizes[0] == tuple(ref_feats[-1].shape[-2:])
    for got, ref in zip(feats[1:], ref_feats[::-1][1:]):
        close(got.to_nchw(), ref, 3e-5, f"mnv4 {tuple(ref.shape)}")


def _convnext_input(arch, img, depth):
    mean = torch.tensor(arch["mean"]).view(1, 3, 1, 1)
    std = torch.tensor(arch["std"]).view(1, 3, 1, 1)
    return torch.cat([(img - mean) / std, depth], dim=1)


def test_lightweight_refiner_convnext_golden(P, golden):
    """reduced-width ConvNeXt encoder + upsample_convx against the reference's LightWeightRefiner (over transformers' ConvNext)."""
    from oracle.cases import CONVNEXT_REFINER, convnext_refiner_inputs, convnext_refiner_sd
    from patchrefinerv2_amd.refiner import LightWeightRefiner
    arch = CONVNEXT_REFINER["arch"]
    m = LightWeightRefiner("convnext_large", coarse_condition=True, arch=arch)
    m.load_state_dict(convnext_refiner_sd(), strict=True)
    img, depth = convnext_refiner_inputs()
    feats, sizes = m(P.Feat.from_nchw(_convnext_input(arch, img, depth).to(DEV)))
    g = golden("convnext_refiner")  # feat0 = stride 32 ... feat4 = stride 2 (upsample_convx), feat5 = its 2x copy (dropped)
    assert feats[0] is None and sizes[0] == tuple(g["feat5"].shape[-2:])
    for i, got in enumerate(feats[1:]):
        ref = g[f"feat{4 - i}"]
        assert sizes[i + 1] == tuple(ref.shape[-2:])
        close(got.to_nchw(), ref, 3e-5, f"convnext feat{4 - i}")


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_lightweight_refiner_convnext_large_width(P, prec):
    """real ConvNeXt-L widths (192..1536), depths cut to (2,2,3,2), a 128x160 patch: every kernel shape of the full encoder."""
    from oracle import convnext as o_cx
    from oracle.cases import convnext_refiner_sd
    from patchrefinerv2_amd.refiner import LightWeightRefiner
    arch = dict(W.CONVNEXT_LARGE, depths=(2, 2, 3, 2))
    sd = convnext_refiner_sd(arch, seed=11)
    m = LightWeightRefiner("convnext_large", coarse_condition=True, arch=arch, prec=prec)
    m.load_state_dict(sd, strict=True)
    img = rand_image(12, 2, 128, 160)
    depth = torch.rand(2, 1, 128, 160, generator=torch.Generator().manual_seed(13)) * 40
    ref_feats, _ = o_cx.lightweight_refiner_convnext(sd, "", img, depth, arch)
    feats, _ = m(P.Feat.from_nchw(_convnext_input(arch, img, depth).to(DEV)))
    for got, ref in zip(feats[1:], ref_feats[::-1][1:]):
        close(got.to_nchw(), ref, 5e-5, f"convnext-L {tuple(ref.shape)}")


def test_lightweight_refiner_effnet_golden(P, golden):
    """reduced EfficientNet encoder against the reference's LightWeightRefiner (over transformers' EfficientNet)"""
    from oracle.cases import EFFNET_REFINER, effnet_refiner_inputs, effnet_refiner_sd
    from patchrefinerv2_amd.refiner import LightWeightRefiner
    arch = EFFNET_REFINER["arch"]
    m = LightWeightRefiner("tf_efficientnet_b5_ap", coarse_condition=True, arch=arch)
    m.load_state_dict(effnet_refiner_sd(), strict=True)
    img, depth = effnet_refiner_inputs()
    feats, sizes = m(P.Feat.from_nchw(_convnext_input(arch, img, depth).to(DEV)))
    g = golden("effnet_refiner")  # feat0 = stride 32 ... feat4 = stride 2
    assert feats[0] is None and sizes[0] == (img.shape[-2], img.shape[-1])
    for i, got in enumerate(feats[1:]):
        ref = torch.as_tensor(g[f"feat{4 - i}"])
        assert sizes[i + 1] == tuple(ref.shape[-2:])
        err = float((got.to_nchw().cpu() - ref).abs().max())
        assert err <= 3e-5 * float(ref.abs().max()), (i, err, float(ref.abs().max()))


@pytest.mark.parametrize("prec,hw", [("f32", (96, 160)), ("bf16x3", (96, 160)), ("f32", (70, 98))])
def test_lightweight_refiner_effnet_b5_width(P, prec, hw):
    """real B5 widths (48 .. 3072 expanded channels), depth cut to 0.6; the odd size exercises 'SAME' padding with an odd
    input at the stride-2 layers (35 -> 18 -> 9 -> 5), which only the oracle covers (transformers pads statically)."""
    from oracle import effnet as o_eff
    from oracle.cases import effnet_refiner_sd
    from patchrefinerv2_amd.refiner import LightWeightRefiner
    arch = W.effnet_arch(1.6, 0.6)
    sd = effnet_refiner_sd(arch, seed=17)
    m = LightWeightRefiner("tf_efficientnet_b5_ap", coarse_condition=True, arch=arch, prec=prec)
    m.load_state_dict(sd, strict=True)
    img = rand_image(18, 2, *hw)
    depth = torch.rand(2, 1, *hw, generator=torch.Generator().manual_seed(19)) * 40
    ref_feats, _ = o_eff.lightweight_refiner_effnet(sd, "", img, depth, arch)
    feats, _ = m(P.Feat.from_nchw(_convnext_input(arch, img, depth).to(DEV)))
    for got, ref in zip(feats[1:], ref_feats[::-1][1:]):
        err = float((got.to_nchw().cpu() - ref).abs().max())
        assert got.to_nchw().shape == ref.shape and err <= 1e-4 * float(ref.abs().max()), (tuple(ref.shape), err, float(ref.abs().max()))


def _build(kind, c, sd, **extra):
    from patchrefinerv2_amd.models import PatchRefiner, PatchRefinerPlus  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    cfg = dict(c["ref_config"])
    for br in ("coarse_branch",):
        cfg[br] = dict(type="DA2", pretrained=None, model_cfg={**c["da2_cfg"]})
    if kind == "PatchRefiner":
        cfg["refiner"] = dict(cfg["refiner"])
        cfg["refiner"]["fine_branch"] = dict(type="DA2", pretrained=None, model_cfg={**c["da2_cfg"]})
    cfg.update(extra)
    m = build_model(dict(type=kind, config=cfg))
    res = m.load_state_dict(sd, strict=True)
    assert not res["missing_keys"] and not res["unexpected_keys"]
    return m


def _run(m, c, mode, **kw):
    image_hr = rand_image(c["seed"], 1, *c["raw"]).to(DEV)
    image_lr = m.resizer(image_hr)
    random.seed(621)
    return m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=dict(image_raw_shape=c["raw"], patch_split_num=c["split"]),
             image_lr=image_lr, image_hr=image_hr, **kw)


def test_e2e_v1_vs_reference_golden(P, golden):
    c, g = E2E_V1, golden("e2e_v1")
    m = _build("PatchRefiner", c, e2e_v1_sd())
    for mode in c["modes"]:
        depth, log = _run(m, c, mode)
        assert not depth.is_cuda and log["coarse_prediction"].is_cuda
        assert tuple(depth.shape) == tuple(g[mode].shape)
        ar, mx = absrel(depth, g[mode])
        assert ar < ABSREL_TOL and mx < 1e-3, (mode, ar, mx)
        close(log["coarse_prediction"], g[mode + "_coarse"], 2e-5)


def test_e2e_v2_vs_reference_golden(P, golden):
    c, g = E2E_V2, golden("e2e_v2")
    m = _build("PatchRefinerPlus", c, e2e_v2_sd())
    for mode in c["modes"]:
        depth, _ = _run(m, c, mode)
        assert tuple(depth.shape) == tuple(g[mode].shape)
        ar, mx = absrel(depth, g[mode])
        assert ar < ABSREL_TOL and mx < 1e-3, (mode, ar, mx)


def test_f16f6_range_guard_runs_for_a_model_built_on_the_default_device(P):
    """a model built with the config default device ('cuda') registers its fp16 + fp6 layers under that name while the
    frame asks for its input's device ('cuda:0'): the range guard must find them (ops.F6Range._key), run after every frame, and -- on an
    image 1e5 x outside the calibrated range -- recompute the frame with moved scales instead of returning fp6-grade values."""
    from patchrefinerv2_amd import ops
    c = E2E_V2
    m = _build("PatchRefinerPlus", c, e2e_v2_sd(), prec="f16f6")  # (no explicit device: torch.device('cuda'))
    assert str(m.device) in ("cuda", "cuda:0") and m.refiner_fusion_model.f16f6
    assert ops.F6Range.active("cuda") and ops.F6Range.active("cuda:0") and ops.F6Range.active(torch.device("cuda", 0))
    hr = rand_image(c["seed"], 1, *c["raw"]).to(DEV)
    tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])

    def run(img):
        random.seed(621)
        return m(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=m.resizer(img), image_hr=img)[0]
    d0 = run(hr)
    assert getattr(m, "f6_guarded_frames", 0) == 1 and torch.isfinite(d0).all()
    n0 = getattr(m, "f6_recalibrations", 0)
    d1 = run(hr * 1e5)
    assert m.f6_guarded_frames == 2 and getattr(m, "f6_recalibrations", 0) > n0, (m.f6_guarded_frames, getattr(m, "f6_recalibrations", 0))
    assert torch.isfinite(d1).all()
    # against the same frames in bf16x3: inside the tolerance at unit scale; at x 1e5 (an image no checkpoint was trained for: the network itself
    # amplifies any arithmetic difference there -- measured 3.7e-4) still far from the fp6-grade result (~1e-2) a saturated fp16 part gives
    mb = _build("PatchRefinerPlus", c, e2e_v2_sd(), prec="bf16x3")

    def run_b(img):
        random.seed(621)
        return mb(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=mb.resizer(img), image_hr=img)[0]
    ar0, _ = absrel(d0, run_b(hr))
    ar, _ = absrel(d1, run_b(hr * 1e5))
    print(f"\nAbsRel f16f6 vs bf16x3: unit scale {ar0:.2e}; x 1e5 image ({m.f6_recalibrations - n0} recalibration(s)) {ar:.2e}")
    assert ar0 < ABSREL_TOL and ar < 2e-3, (ar0, ar)


@pytest.mark.parametrize("enc", ["convnext", "effnet"])
def test_e2e_v2_other_refiner_encoders_vs_oracle(P, enc):
    """PatchRefinerPlus with the ConvNeXt (v2_convx_u4k.py) / EfficientNet (v2_eff_u4k.py) refiner encoder end to end; the
    encoders themselves are pinned by test_lightweight_refiner_{convnext,effnet}_golden, everything around them by e2e_v2."""
    from oracle import cases
    c, sd = (cases.E2E_V2CX, cases.e2e_v2cx_sd()) if enc == "convnext" else (cases.E2E_V2EF, cases.e2e_v2ef_sd())
    m = _build("PatchRefinerPlus", c, sd)
    ora = o_tiling.OraclePatchRefinerPlus(sd, W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}),
                                          **{f"{enc}_arch": c["arch"]},
                                          patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"])
    hr = rand_image(c["seed"], 1, *c["raw"])
    tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])
    for mode in c["modes"]:
        random.seed(621)
        ref, _ = ora(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tc, image_lr=ora.resizer(hr), image_hr=hr)
        depth, _ = _run(m, c, mode)
        ar, mx = absrel(depth, ref)
        assert ar < 1e-5 and mx < 1e-3, (mode, ar, mx)


@pytest.mark.parametrize("variant", ["coarse-fusion", "self-agg", "no-c2f", "no-c2f-no-condition"])  # ('only-gate' needs coarse_chl[0] = 32: module test)
def test_e2e_v2_fusion_ablation_variants_vs_oracle(P, variant):
    """PatchRefinerPlus end to end with the reference's BiDirectionalFusion ablations (configs/patchrefinerv2_zoedepth_ablation/
    plus_mobile_c2f_wogate.py, plus_mobile_c2f_selfagg.py, plus_mobile_u4k_base_coarse.py): C2FModule without the gate, without the
    fusion conv, and no c2f module at all (the refiner's six maps, level 0 = the x2 copy, and pred2 = zeros go straight on).  The
    fusion module of each is pinned against the reference by test_bidir_fusion_c2f_ablation_types / test_bidir_fusion_without_c2f.
    'no-c2f-no-condition' = plus_mobile_u4k_base.py: additionally LightWeightRefiner(coarse_condition=False), a 3-channel encoder stem."""
    import copy
    from collections import OrderedDict
    c = copy.deepcopy(E2E_V2)
    cond = variant != "no-c2f-no-condition"
    variant = variant.replace("-no-condition", "")
    c["ref_config"]["refiner"]["fine_branch"]["coarse_condition"] = cond
    kw = dict(coarse2fine=False) if variant == "no-c2f" else dict(coarse2fine_type=variant)
    f = c["fusion"]
    if variant == "no-c2f":
        f["fine_chl_after_coarse2fine"] = [32] + f["fine_chl"]
    fm = c["ref_config"]["refiner"]["fusion_model"]
    fm.update(f)
    fm.update(coarse2fine=variant != "no-c2f", coarse2fine_type="coarse-gated" if variant == "no-c2f" else variant)
    spec = OrderedDict()
    spec.update(W.dav2_spec("coarse_branch.", c["da2_cfg"]))
    spec.update(W.mnv4_spec("refiner_fine_branch.refiner_encoder.", in_chans=4 if cond else 3))
    spec.update(W.bidir_fusion_spec("refiner_fusion_model.", f["coarse_chl"], f["fine_chl"], f["fine_chl_after_coarse2fine"], f["temp_chl"],
                                    f["dec_chl"], **kw))
    sd = W.synth_state_dict(spec, seed=43)
    m = _build("PatchRefinerPlus", c, sd)
    ora = o_tiling.OraclePatchRefinerPlus(sd, W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}), fusion_kw=kw, coarse_condition=cond,
                                          patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"])
    hr = rand_image(c["seed"], 1, *c["raw"])
    tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])
    for mode in c["modes"]:
        random.seed(621)
        ref, _ = ora(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tc, image_lr=ora.resizer(hr), image_hr=hr)
        depth, _ = _run(m, c, mode)
        ar, mx = absrel(depth, ref)
        assert ar < 1e-5 and mx < 1e-3, (variant, mode, ar, mx)


def test_save_and_from_pretrained_roundtrip(P, tmp_path):
    """PyTorchModelHubMixin's local layout (config.json + model.safetensors): the re-loaded model reproduces the frame"""
    from patchrefinerv2_amd.models import PatchRefinerPlus
    c = E2E_V2
    m = _build("PatchRefinerPlus", c, e2e_v2_sd())
    a, _ = _run(m, c, "r4")
    m.save_pretrained(str(tmp_path / "ckp"))
    m2 = PatchRefinerPlus.from_pretrained(str(tmp_path / "ckp"))
    b, _ = _run(m2, c, "r4")
    assert torch.equal(a, b)
    with pytest.raises(FileNotFoundError):
        PatchRefinerPlus.from_pretrained("zhyever/not-a-local-dir")


def test_batching_independence(P):
    """per-patch results do not depend on the mini-batch size (the licence for large batches)."""
    c = E2E_V1
    m = _build("PatchRefiner", c, e2e_v1_sd())
    a, _ = _run(m, c, "r8")
    m.max_batch = 7
    b, _ = _run(m, c, "r8")
    assert torch.equal(a, b)
    m.max_batch, m.n_streams = 3, 3  # batches spread over 3 HIP streams
    for _ in range(3):
        d, _ = _run(m, c, "r8")
        assert torch.equal(a, d)


@pytest.mark.parametrize("kind", ["PatchRefiner", "PatchRefinerPlus"])
def test_hip_graph_replay_is_bit_identical(P, kind):
    """hip_graph=True: the device side of a frame (coarse forward, tile batches on 3 streams, blend) captured once into a
    hipGraph and replayed -- the eager first frame, the captured second frame and further replays (other image, other random
    tiles: the plan's coordinates are graph INPUTS) all equal the eager model bit for bit"""
    c, sd = (E2E_V1, e2e_v1_sd()) if kind == "PatchRefiner" else (E2E_V2, e2e_v2_sd())
    mode = "r8" if kind == "PatchRefiner" else "r4"
    eager = _build(kind, c, sd, max_batch=3, n_streams=3)
    graph = _build(kind, c, sd, max_batch=3, n_streams=3, hip_graph=True)
    tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])
    for it, (img_seed, rnd_seed) in enumerate(((0, 621), (0, 621), (7, 5), (0, 621), (9, 1234))):
        hr = rand_image(img_seed, 1, *c["raw"]).to(DEV)
        outs = []
        for m in (eager, graph):
            random.seed(rnd_seed)
            d, log = m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tc, image_lr=m.resizer(hr), image_hr=hr)
            outs.append((d, log["coarse_prediction"].clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), it
    assert any(isinstance(v, dict) for v in graph._graphs.values())  # really captured
    m1a, _ = _run(eager, c, "m1")
    m1b, _ = _run(graph, c, "m1")   # another mode: its own graph (first call eager)
    m1c, _ = _run(graph, c, "m1")
    assert torch.equal(m1a, m1b) and torch.equal(m1a, m1c)


def test_predict_tiles_and_run_consistency(P, tmp_path):
    """predict_tiles (explicit crop origins, no blend: the reference's mode='train' forward per crop) gives exactly the tiles the
    m1 frame pastes; Tester.run_consistency (tester.py:211-321) on top of it: shifted overlapping crops, mean |difference| over
    the shared strips"""
    import numpy as np
    from patchrefinerv2_amd.tester import ImageDataset, RunnerInfo, Tester, write_png8
    c = E2E_V1
    m = _build("PatchRefiner", c, e2e_v1_sd())
    hr = rand_image(c["seed"], 1, *c["raw"]).to(DEV)
    m1, _ = _run(m, c, "m1")
    rh, rw = c["raw"][0] // 2, c["raw"][1] // 2
    tiles = [(0, 0), (0, rw), (rh, 0), (rh, rw)]
    preds = m.predict_tiles(m.resizer(hr), hr, tiles)
    ph, pw = c["pps"]
    for k, (i, j) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        assert torch.equal(preds[k, 0].cpu(), m1[0, 0, i * ph:(i + 1) * ph, j * pw:(j + 1) * pw])
    d = tmp_path / "rgb"
    d.mkdir()
    img = (rand_image(3, 1, *c["raw"])[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    write_png8(str(d / "f0.png"), img)
    ds = ImageDataset(str(d), image_resolution=c["raw"], network_process_size=c["pps"])
    t = Tester(None, RunnerInfo(save=True, work_dir=str(tmp_path / "out")), ds, m)
    res = t.run_consistency(image_raw_shape=c["raw"], patch_split_num=c["split"], overlap=rh // 2)
    assert len(res) == 1 and np.isfinite(res[0]["consistency_error"]) and res[0]["consistency_error"] > 0
    assert t.last_eval["consistency_error"] == res[0]["consistency_error"] and (tmp_path / "out" / "f0.png").exists()


def test_run_consistency_vs_reference_golden(P, golden, tmp_path):
    """Tester.run_consistency against the REFERENCE's own Tester.run_consistency (tester.py:211-321), which
    oracle/make_golden.py::g_consistency ran over the reference's PatchRefiner on the same synthetic 2160 x 3840 frame prepared
    as the U4K consistency dataset does (16 crops of 540 x 960, overlap 270, pre-normalised bboxs): the 16 resized crop
    predictions and the consistency error"""
    import numpy as np
    from oracle import tiling as o_tiling
    from patchrefinerv2_amd.tester import ImageDataset, RunnerInfo, Tester
    c, g = E2E_V1, golden("consistency")
    raw, split, overlap = [2160, 3840], [4, 4], int(g["overlap"])
    m = _build("PatchRefiner", c, e2e_v1_sd(), image_raw_shape=raw, patch_split_num=split, max_batch=8)
    img = rand_image(int(g["image_seed"]), 1, *raw)
    d = tmp_path / "rgb"
    d.mkdir()
    np.save(str(d / "frame0.npy"), img[0].permute(1, 2, 0).contiguous().numpy())
    ds = ImageDataset(str(d), image_resolution=raw, network_process_size=c["pps"])
    assert torch.equal(ds[0]["image_hr"].cpu(), img[0])       # same-size bicubic(align_corners) is the identity
    t = Tester(None, RunnerInfo(save=False, work_dir=str(tmp_path / "out")), ds, m)
    res = t.run_consistency(image_raw_shape=raw, patch_split_num=split, overlap=overlap)
    ce_ref = float(g["consistency_error"])
    crops = t.last_crops.cpu()
    assert tuple(crops.shape) == (16, 540, 960)
    err = float((crops[:, ::9, ::16] - torch.from_numpy(g["crops_strided"])).abs().max())
    print(f"run_consistency vs reference: consistency_error {res[0]['consistency_error']:.6f} (reference {ce_ref:.6f}); crops max|d| {err:.2e}")
    assert err < 1e-3 and abs(res[0]["consistency_error"] - ce_ref) < 2e-5 * ce_ref
    # and against the oracle's restatement run here on the full crops
    dcfg = W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]})
    ora = o_tiling.OraclePatchRefiner(e2e_v1_sd(), dcfg, dcfg, patch_process_shape=c["pps"], image_raw_shape=raw, patch_split_num=split)
    ce_o, crops_o = o_tiling.run_consistency(ora, img, overlap=overlap)
    assert abs(ce_o - ce_ref) < 1e-6 and float((crops - crops_o).abs().max()) < 1e-3


def test_compute_metrics_device_on_the_gpu_vs_reference_golden(P, golden):
    """metrics.compute_metrics_device -- how Tester.run scores a frame with ground truth -- on CUDA tensors against the
    reference's compute_metrics outputs (tests/golden/output_stage.npz, estimator/utils/metric.py:87-149), incl. the NaN / inf
    pixels, the garg crop, the resized prediction and the soft-edge error"""
    import numpy as np
    from patchrefinerv2_amd import metrics as M
    g = golden("output_stage")
    gt, pred, pred_lo, edges = (torch.from_numpy(g[k]) for k in ("gt", "pred", "pred_lo", "edges"))
    d1 = M.compute_metrics_device(gt.to(DEV), pred.to(DEV), garg_crop=False, eigen_crop=False, dataset="u4k", min_depth_eval=0.1,
                                  max_depth_eval=10, disp_gt_edges=edges.to(DEV))
    d2 = M.compute_metrics_device(gt, pred_lo.to(DEV), garg_crop=True, eigen_crop=False, dataset="kitti", min_depth_eval=0.1, max_depth_eval=10)
    for tag, d in (("m1", d1), ("m2", d2)):
        keys = [k[len(tag) + 1:] for k in g.files if k.startswith(tag + "_")]
        assert set(keys) == set(d), (sorted(keys), sorted(d))
        for k in keys:
            np.testing.assert_allclose(float(d[k]), float(g[f"{tag}_{k}"]), rtol=2e-5, atol=1e-7, err_msg=f"{tag} {k}")


def test_models_through_torch_custom_ops(P):
    """The host mirror reaches the kernels on two routes -- the PyTorch custom ops torch.ops.prv2.* (ops.DISPATCH = 'torch', the
    default) and ctypes straight on the C ABI -- for EVERY entry point a frame uses (convs with all epilogues, X2 formats, coarse taps,
    gathers, ViT pieces, blend): whole frames of V1, V2 and V2 over the ZoeDepth/BEiT coarse branch are bit-identical on both routes"""
    from oracle.cases import E2E_V2B, e2e_v2b_sd
    from patchrefinerv2_amd import ops, torch_ops
    from patchrefinerv2_amd.registry import build_model
    torch_ops.load()
    default = ops.DISPATCH

    def both(m, c, mode):
        out = {}
        try:
            for route in ("ctypes", "torch"):
                ops.DISPATCH = route
                out[route] = _run(m, c, mode)
        finally:
            ops.DISPATCH = default
        return out["ctypes"], out["torch"]

    cases = [("PatchRefiner", E2E_V1, e2e_v1_sd(), "r8"), ("PatchRefinerPlus", E2E_V2, e2e_v2_sd(), "r4")]
    for kind, c, sd, mode in cases:
        for prec in ("f32", "bf16x3"):
            m = _build(kind, c, sd, prec=prec, n_streams=2, max_batch=3)
            (a, la), (b, lb) = both(m, c, mode)
            assert torch.equal(a, b) and torch.equal(la["coarse_prediction"], lb["coarse_prediction"]), (kind, prec)
    c = E2E_V2B
    m = build_model(dict(type="PatchRefinerPlus", config={**c["ref_config"], "prec": "bf16x3"}))
    m.load_state_dict(e2e_v2b_sd(), strict=True)
    (a, _), (b, _) = both(m, c, "m1")
    assert torch.equal(a, b)


def test_next_frame_coarse_prefetch_is_bit_identical(P):
    """forward(next_image_lr=...): the next frame's coarse forward runs on a stream of its own beside this frame's tile batches and
    is picked up by the next call; frames are bit-identical with / without, a mismatching tensor is simply ignored"""
    c = E2E_V2
    m = _build("PatchRefinerPlus", c, e2e_v2_sd(), prec="bf16x3", n_streams=2, max_batch=3)
    frames = []
    for seed in (0, 1, 2):
        hr = rand_image(seed, 1, *c["raw"]).cuda()
        frames.append((hr, m.resizer(hr)))
    tc = dict(image_raw_shape=list(c["raw"]), patch_split_num=list(c["split"]))

    def run(i, nxt):
        random.seed(5)
        d, log = m(mode="infer", cai_mode="r4", process_num=2, tile_cfg=tc, image_lr=frames[i][1], image_hr=frames[i][0], next_image_lr=nxt)
        return d, log["coarse_prediction"].clone()

    ref = [run(i, None) for i in range(3)]
    got = [run(0, frames[1][1]), run(1, frames[2][1]), run(2, None)]
    for (a, ca), (b, cb) in zip(ref, got):
        assert torch.equal(a, b) and torch.equal(ca, cb)
    assert "_coarse_prefetched" not in m.__dict__  # consumed
    run(0, frames[2][1])          # prefetched for frame 2 ...
    d, cp = run(1, None)          # ... but frame 1 arrives: computed inline
    assert torch.equal(d, ref[1][0]) and torch.equal(cp, ref[1][1])
    # the announced frame is dropped and ANOTHER image lands at the same address (what the caching allocator does with a freed
    # block): the prefetch entry is keyed by the tensor object, so the stale pyramid must not be used
    ghost = frames[2][1].clone()
    run(0, ghost)
    ghost_ptr = ghost.data_ptr()
    del ghost
    other = frames[1][1].clone()
    if other.data_ptr() == ghost_ptr:
        assert other._version == 0
    frames.append((frames[1][0], other))
    d, cp = run(3, None)
    assert torch.equal(d, ref[1][0]) and torch.equal(cp, ref[1][1])
    # in-place modification of the announced tensor invalidates the prefetch as well
    lr2 = frames[2][1].clone()
    run(0, lr2)
    lr2.copy_(frames[1][1])
    frames.append((frames[1][0], lr2))
    d, cp = run(4, None)
    assert torch.equal(d, ref[1][0]) and torch.equal(cp, ref[1][1])
    # weights replaced between two calls: nothing derived from the old ones survives
    run(0, frames[1][1])
    m.load_state_dict(e2e_v2_sd(), strict=True)
    assert "_coarse_prefetched" not in m.__dict__


def test_rejects_cpu_inputs_and_bad_shapes(P):
    c = E2E_V1
    m = _build("PatchRefiner", c, e2e_v1_sd())
    hr = rand_image(0, 1, *c["raw"])
    with pytest.raises(RuntimeError):
        m(mode="infer", cai_mode="m1", image_lr=hr[:, :, :56, :84], image_hr=hr)
    with pytest.raises(ValueError):
        m.prepare_tile_cfg([217, 384], [2, 2])
    with pytest.raises(ValueError):
        _run(m, c, "x3")


def test_e2e_bf16x3_meets_absrel_target(P, golden):
    """the fast arithmetic mode (split-bf16, 3 MFMAs) stays inside the north star's AbsRel <= 1e-4"""
    for kind, c, sdf, gname in (("PatchRefiner", E2E_V1, e2e_v1_sd, "e2e_v1"), ("PatchRefinerPlus", E2E_V2, e2e_v2_sd, "e2e_v2")):
        g = golden(gname)
        m = _build(kind, c, sdf(), prec="bf16x3")
        mode = c["modes"][-1]
        depth, _ = _run(m, c, mode)
        ar, mx = absrel(depth, g[mode])
        print(kind, mode, "bf16x3 AbsRel", ar, "max|d|", mx)
        assert ar < ABSREL_TOL, (kind, ar)
        assert ar > 1e-8  # really the split path


def test_zoedepth_da_core(P, golden):
    """ZoeDepth metric-bins head over the DepthAnything ViT-S core vs the reference's ZoeDepth.build output"""
    from patchrefinerv2_amd.zoedepth import ZoeDepth
    c, g = ZOE_DA, golden("zoedepth_da")
    sd = W.synth_state_dict(W.zoedepth_spec("", c["zcfg"]), seed=c["seed"])
    m = ZoeDepth.build(**c["zcfg"])
    res = m.load_state_dict(sd, strict=True)
    assert not res["missing_keys"]
    z = W.zoedepth_cfg(c["zcfg"])
    for tag, (h, w) in c["inputs"].items():
        x = rand_image(c["seed"], 2, h, w)
        out = m(x.to(DEV), return_final_centers=True)
        ref = o_zoe.zoedepth_forward(sd, "", x, z)
        close(out["metric_depth"], g[f"{tag}_depth"], 3e-5, f"{tag} depth vs golden")
        for k, v in ref["temp_features"].items():
            close(out["temp_features"][k].to_nchw(), v, 3e-5, f"{tag} {k}")
    with pytest.raises(NotImplementedError):
        ZoeDepth.build(midas_model_type="DPT_SwinV2_L_384")


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_zoedepth_midas_beit_core(P, golden, prec):
    """type='ZoeDepth': ZoeDepth over the MiDaS DPT-BEiT core (relative-position-bias attention, project readout, MiDaS DPT
    decoder hooks) vs the reference's ZoeDepth / MidasCore classes run over transformers' DPT-BEiT (golden) and vs the oracle;
    non-square inputs resize the 7 x 7 bias table"""
    from oracle.cases import ZOE_BEIT
    from patchrefinerv2_amd.zoedepth import ZoeDepth
    c, g = ZOE_BEIT, golden("zoedepth_beit")
    sd = W.synth_state_dict(W.zoedepth_spec("", c["zcfg"]), seed=c["seed"])
    m = ZoeDepth.build(**c["zcfg"], prec=prec)
    res = m.load_state_dict(sd, strict=True)
    assert not res["missing_keys"]
    z = W.zoedepth_cfg(c["zcfg"])
    tol = 3e-5 if prec == "f32" else 3e-4
    for tag, (h, w) in c["inputs"].items():
        x = rand_image(c["seed"], 2, h, w)
        out = m(x.to(DEV), return_final_centers=True)
        ref = o_zoe.zoedepth_forward(sd, "", x, z)
        close(out["metric_depth"], g[f"{tag}_depth"], tol, f"{tag} depth vs golden")
        for k, v in ref["temp_features"].items():
            close(out["temp_features"][k].to_nchw(), v, tol, f"{tag} {k}")
        ar, _ = absrel(out["metric_depth"], ref["metric_depth"])
        assert ar < (1e-5 if prec == "f32" else ABSREL_TOL), (tag, ar)
    with pytest.raises(ValueError):
        m(rand_image(1, 1, 56, 84).to(DEV))  # not multiples of 32


def test_e2e_v2_midas_beit_coarse_vs_reference_golden(P, golden):
    """PatchRefinerPlus wired as configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py: coarse_branch type='ZoeDepth' (MidasCore / BEiT),
    ResizeZoe to 384 x 512, P = 384 x 512 -- against the reference's own classes end to end (reduced BEiT)"""
    from oracle.cases import E2E_V2B, e2e_v2b_sd
    from patchrefinerv2_amd.registry import build_model
    import patchrefinerv2_amd.models  # noqa: F401
    c, g = E2E_V2B, golden("e2e_v2b")
    m = build_model(dict(type="PatchRefinerPlus", config=dict(c["ref_config"])))
    res = m.load_state_dict(e2e_v2b_sd(), strict=True)
    assert not res["missing_keys"] and not res["unexpected_keys"]
    assert m.resizer.kind == "zoe" and m.resizer.out_hw == (384, 512)
    for mode, shape in zip(c["modes"], ((768, 1024), (540, 960))):
        depth, _ = _run(m, c, mode)
        assert tuple(depth.shape[-2:]) == shape
        ar, mx = absrel(depth[..., ::3, ::3], g[mode])  # the golden holds every third row / column of the map
        assert ar < 1e-5 and mx < 1e-3, (mode, ar, mx)
    with pytest.raises(ValueError):
        build_model(dict(type="PatchRefinerPlus", config={**c["ref_config"], "patch_process_shape": [392, 518]}))


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_baseline_config0_as_configured_beit_l_full_size(P, prec):
    """BASELINE config[0] exactly as the reference configures it: a 540 x 960 frame, BaselinePretrain(target='coarse'),
    coarse_branch type='ZoeDepth' = MiDaS DPT_BEiT_L_384 (24 blocks, 1024 wide, 769 tokens at 384 x 512) -- both arithmetic
    modes against the fp32 oracle on the same synthetic weights"""
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import ZOE_BEIT_L
    branch = dict(type="ZoeDepth", **ZOE_BEIT_L)
    m = build_model(dict(type="BaselinePretrain", coarse_branch=branch, fine_branch=branch, sigloss=dict(type="SILogLoss"),
                         min_depth=1e-3, max_depth=80, image_raw_shape=[540, 960], patch_process_shape=[384, 512],
                         patch_split_num=[1, 1], target="coarse", prec=prec))
    sd = W.synth_state_dict(W.zoedepth_spec("", ZOE_BEIT_L), seed=0)
    m.load_dict(sd)
    hr = rand_image(11, 1, 540, 960)
    lr = m.resizer(hr.to(DEV))
    assert tuple(lr.shape) == (1, 3, 384, 512)
    depth, _ = m(mode="infer", image_lr=lr, image_hr=hr.to(DEV), depth_gt=None)
    ref = o_zoe.zoedepth_forward(sd, "", lr.cpu(), W.zoedepth_cfg(ZOE_BEIT_L))["metric_depth"]
    ar, mx = absrel(depth, ref)
    print(f"config[0] as configured (DPT_BEiT_L_384, 384x512) {prec}: AbsRel {ar:.3e} max|d| {mx:.3e} (depth {float(ref.min()):.2f}..{float(ref.max()):.2f})")
    assert tuple(depth.shape) == (1, 1, 384, 512) and ar < (1e-5 if prec == "f32" else ABSREL_TOL), (ar, mx)


def test_baseline_pretrain_vs_reference_golden(P, golden):
    """BaselinePretrain (registered type, keyword constructor) against the reference's own class: target 'coarse' returns
    the device tensor of one backbone forward, target 'fine' tiles with the bare backbone (N * process_num random tiles)"""
    from oracle.cases import BASELINE, baseline_kwargs, baseline_sd
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    c, g = BASELINE, golden("baseline")
    hr = rand_image(c["seed"], 1, *c["raw"]).to(DEV)
    tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])
    for target, modes in (("coarse", ["m1"]), ("fine", c["fine_modes"])):
        m = build_model(dict(type="BaselinePretrain", **baseline_kwargs(target)))
        res = m.load_dict(baseline_sd())
        assert not res["missing_keys"] and not res["unexpected_keys"]
        for mode in modes:
            random.seed(621)
            depth, log = m(mode="infer", image_lr=m.resizer(hr), image_hr=hr, depth_gt=None, tile_cfg=tc, cai_mode=mode, process_num=4)
            assert depth.is_cuda == (target == "coarse") and sorted(log) == (["depth_gt", "depth_pred", "rgb"] if target == "coarse" else [])
            ar, mx = absrel(depth, g[f"{target}_{mode}"])
            assert tuple(depth.shape) == tuple(g[f"{target}_{mode}"].shape) and ar < 1e-5 and mx < 1e-3, (target, mode, ar, mx)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_baseline_config0_540x960_coarse_full_size(P, prec):
    """BASELINE config[0] at its real size: a 540 x 960 frame, BaselinePretrain(target='coarse') with the full ViT-L
    'DA-ZoeDepth' backbone (the pinned sibling of the un-vendored BEiT-L), image_lr 392 x 518 (1037 tokens,
    24 blocks) -- the product in both arithmetic modes against the fp32 oracle on the same synthetic weights."""
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import ZOE_DA_L
    branch = dict(type="DA-ZoeDepth", **ZOE_DA_L)
    m = build_model(dict(type="BaselinePretrain", coarse_branch=branch, fine_branch=branch, sigloss=dict(type="SILogLoss"),
                         min_depth=1e-3, max_depth=80, image_raw_shape=[540, 960], patch_process_shape=[392, 518],
                         patch_split_num=[1, 1], target="coarse", prec=prec))
    sd = W.synth_state_dict(W.zoedepth_spec("", ZOE_DA_L), seed=0)
    m.load_dict(sd)
    hr = rand_image(11, 1, 540, 960)
    lr = m.resizer(hr.to(DEV))
    assert tuple(lr.shape) == (1, 3, 392, 518)
    depth, log = m(mode="infer", image_lr=lr, image_hr=hr.to(DEV), depth_gt=None)
    ref = o_zoe.zoedepth_forward(sd, "", lr.cpu(), W.zoedepth_cfg(ZOE_DA_L))["metric_depth"]
    ar, mx = absrel(depth, ref)
    print(f"config[0] coarse-only 392x518 ViT-L {prec}: AbsRel {ar:.3e} max|d| {mx:.3e} (depth range {float(ref.min()):.2f}..{float(ref.max()):.2f})")
    assert tuple(depth.shape) == (1, 1, 392, 518) and ar < (1e-5 if prec == "f32" else ABSREL_TOL), (ar, mx)


def test_e2e_v2_zoedepth_coarse_vs_reference_golden(P, golden):
    c, g = E2E_V2Z, golden("e2e_v2z")
    from patchrefinerv2_amd.registry import build_model
    import patchrefinerv2_amd.models  # noqa: F401
    m = build_model(dict(type="PatchRefinerPlus", config=dict(c["ref_config"])))
    res = m.load_state_dict(e2e_v2z_sd(), strict=True)
    assert not res["missing_keys"] and not res["unexpected_keys"]
    for mode in c["modes"]:
        depth, _ = _run(m, c, mode)
        ar, mx = absrel(depth, g[mode])
        assert ar < ABSREL_TOL and mx < 1e-3, (mode, ar, mx)


@pytest.mark.parametrize("raw,split,mode,pn", [
    ([216, 384], [2, 2], "r8", 4),      # the golden case, against a freshly run oracle
    ([168, 504], [1, 3], "m1", 4),      # single tile row
    ([224, 336], [2, 3], "r6", 3),      # process_num 3: two random calls of 3 tiles sharing a column each
    ([112, 168], [1, 1], "m1", 4),      # one tile == whole frame
    ([240, 400], [2, 2], "r7", 4),      # N // process_num = 1 random call (7 // 4)
])
def test_tiling_variants_vs_oracle(P, raw, split, mode, pn):
    """frame driver on ragged / degenerate tilings: product == oracle run on the same seed (tile plan, ROI boxes,
    nearest/bilinear resizes of the blend, order-dependent running mean)"""
    c = dict(E2E_V1)
    c["raw"], c["split"] = raw, split
    sd = e2e_v1_sd()
    m = _build("PatchRefiner", c, sd, image_raw_shape=raw, patch_split_num=split)
    cfg = W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]})
    ora = o_tiling.OraclePatchRefiner(sd, cfg, cfg, patch_process_shape=c["pps"], image_raw_shape=raw, patch_split_num=split)
    hr = rand_image(3, 1, *raw)
    tc = dict(image_raw_shape=raw, patch_split_num=split)
    random.seed(99)
    ref, _ = ora(mode="infer", cai_mode=mode, process_num=pn, tile_cfg=tc, image_lr=ora.resizer(hr), image_hr=hr)
    random.seed(99)
    hr_d = hr.to(DEV)
    got, _ = m(mode="infer", cai_mode=mode, process_num=pn, tile_cfg=tc, image_lr=m.resizer(hr_d), image_hr=hr_d)
    assert tuple(got.shape) == tuple(ref.shape)
    ar, mx = absrel(got, ref)
    assert ar < 1e-5 and mx < 1e-3, (ar, mx)


def test_single_row_split_rejects_overlap_modes(P):
    """the reference cannot run m2 / r<N> with a 1-tile axis (empty half-offset pass): same loud failure"""
    c = dict(E2E_V1)
    m = _build("PatchRefiner", c, e2e_v1_sd(), image_raw_shape=[168, 504], patch_split_num=[1, 3])
    hr = rand_image(3, 1, 168, 504).to(DEV)
    with pytest.raises(RuntimeError):
        m(mode="infer", cai_mode="m2", process_num=4, tile_cfg=dict(image_raw_shape=[168, 504], patch_split_num=[1, 3]),
          image_lr=m.resizer(hr), image_hr=hr)


def test_full_size_4k_r32_properties(P):
    """BASELINE config[2] at its real size (4K, 4x4, r32 = 81 tiles, bf16x3, synthetic weights), checked through
    size-independent properties: bit-identical across repeated runs, across mini-batch sizes and stream counts (no atomics,
    no order dependence anywhere in the kernels), finite, inside [min_depth, max_depth] after the blend, and the r32
    result differs from the m1 result only where the extra passes contribute."""
    from patchrefinerv2_amd import models, weights as W  # noqa: F401  (models registers the types)
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
    name = "v2_zoeda_4k_r32"
    w = WORKLOADS[name]
    model = build_model(model_config(name, prec="bf16x3", max_batch=14, n_streams=3))
    model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
    hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(3)).to(DEV)
    lr = model.resizer(hr)
    tile_cfg = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])

    def run(mode="r32"):
        random.seed(621)
        d, log = model(mode="infer", cai_mode=mode, process_num=4, tile_cfg=tile_cfg, image_lr=lr, image_hr=hr)
        return d, log

    a, log = run()
    assert tuple(a.shape) == (1, 1, 2160, 3840) and a.device.type == "cpu" and bool(torch.isfinite(a).all())
    assert float(a.min()) >= 0.0 and float(a.max()) <= float(model.max_depth) * 1.0001
    assert tuple(log["coarse_prediction"].shape) == (1, 1, *w["pps"])
    b, _ = run()
    assert torch.equal(a, b)                         # run-to-run
    model.max_batch, model.n_streams = 9, 1
    c, _ = run()
    assert torch.equal(a, c)                         # batch size / streams
    model.max_batch, model.n_streams = 27, 2
    d, _ = run()
    assert torch.equal(a, d)
    m1, _ = run("m1")
    assert tuple(m1.shape) == (1, 1, 4 * w["pps"][0], 4 * w["pps"][1])   # m-modes stay at the reensemble resolution


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU path on ONE device: the sharded forward for every rank, exchanged as RCCL would
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("dst", [None, 0])
def test_patch_shard_emulation_equals_unsharded(P, world, dst):
    """run ``shard=(r, N)`` for r = 0..N-1 sequentially on one GPU, hand rank 0 the stacks exactly as all_gather / gather
    would deliver them (rank-major), and require the blended frame to be bit-identical to the unsharded one
    (tile i -> rank i mod N, blend in the reference's order)."""
    from conftest import ShardEmulation
    c = E2E_V1
    m = _build("PatchRefiner", c, e2e_v1_sd())
    full, _ = _run(m, c, "r8")
    emu = ShardEmulation(m, world)
    emu.record()
    for r in range(world):
        depth, log = _run(m, c, "r8", shard=(r, world), gather_dst=dst)
        assert depth is None and log["coarse_prediction"] is not None
    n_tiles = sum(len(p["raw"]) for p in m.last_plan)
    groups = m.last_shard_layout
    two = world < 8   # (from 8 ranks on a first group of < 8 tiles per rank merges with the random tiles: models.SHARD_MERGE_BELOW)
    assert n_tiles == 17 and [g["n"] for g in groups] == ([9, 8] if two else [17])   # [init + 3 grids | random]: two exchanges per frame
    assert sorted(emu.stacks) == [(r, g) for r in range(world) for g in range(2 if two else 1)]
    assert all(emu.stacks[(r, gi)].shape[0] == g["per"] for r in range(world) for gi, g in enumerate(groups))  # padded to a common length
    emu.deliver()
    for r in ([0] if dst is not None else range(world)):                   # all-gather: every rank blends the same map
        got, _ = _run(m, c, "r8", shard=(r, world), gather_dst=dst)
        assert torch.equal(got, full)
    # a rank seeded differently still blends at rank 0's coordinates only through the broadcast plan (tests/test_distributed.py);
    # modes without random tiles have a single gather group
    emu.restore()
    full_m2, _ = _run(m, c, "m2")
    emu = ShardEmulation(m, world)
    emu.record()
    for r in range(world):
        assert _run(m, c, "m2", shard=(r, world), gather_dst=0)[0] is None
    assert len(m.last_shard_layout) == 1
    emu.deliver()
    assert torch.equal(_run(m, c, "m2", shard=(0, world), gather_dst=0)[0], full_m2)


# ------------------------------------------------------------------------------------------------------------------
# the headline arithmetic at headline width / depth against the fp32 oracle on the same synthetic weights
# ------------------------------------------------------------------------------------------------------------------
def _one_tile(model, ora, hr, tile, tile_cfg):
    """one tile through the product's per-patch path and through the oracle's: (pred, ref, coarse, coarse_ref)"""
    from patchrefinerv2_amd.ops import Feat
    hr_d = hr.to(DEV)
    lr_d = model.resizer(hr_d)
    feats, cp = model.coarse_forward(lr_d)
    cd = Feat(cp.view(1, cp.shape[-2], cp.shape[-1], 1))
    tc = model.prepare_tile_cfg(tile_cfg["image_raw_shape"], tile_cfg["patch_split_num"])
    t_dev = torch.tensor([tile], dtype=torch.int32, device=DEV)
    boxes = torch.from_numpy(model._boxes([tile], tc)).to(DEV)
    crops, rois, droi = model._prepare_batch(hr_d[0].contiguous(), t_dev, boxes, tc, feats, cd)
    pred = model.infer_forward(crops, rois, droi)
    rh, rw = tc["patch_raw_shape"]
    lr = lr_d.cpu()
    o_feats, o_cp = ora.coarse_forward(lr)
    o_crops, bb = ora._crops(hr[0], [tile[0]], [tile[1]], rh, rw)
    post = o_tiling.coarse_postprocess_test(o_cp, o_feats, o_tiling.bboxs_to_feat(bb, tile_cfg["image_raw_shape"], ora.patch_process_shape),
                                            ora.patch_process_shape[0])
    ref = ora.infer_forward(o_crops, post)
    return pred, ref, cp, o_cp


@pytest.mark.parametrize("prec", ["bf16x3", "f32"])
def test_full_width_v2_tile_vs_fp32_oracle(P, prec):
    """ONE real tile of BASELINE config[2] (v2_zoeda_4k_r32): the full ViT-L 'DA-ZoeDepth' coarse forward at 392 x 518, the ROI
    pyramid of a half-offset 540 x 960 tile, MobileNetV4-S and the full-width BiDirectionalFusion (45 layers, 256 channels, K up
    to 770 * 9) -- the benchmarked arithmetic at the benchmarked size against the fp32 oracle.  North-star bound: per-pixel
    AbsRel <= 1e-4."""
    from oracle import dav2 as od
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
    name = "v2_zoeda_4k_r32"
    w = WORKLOADS[name]
    sd = W.synth_state_dict(state_spec(name), seed=0)
    model = build_model(model_config(name, prec=prec))
    model.load_state_dict(sd, strict=True)
    zc = W.zoedepth_cfg(w["zoe"])
    ora = o_tiling.OraclePatchRefinerPlus(sd, None, coarse_fn=lambda lr: od.coarse_features(o_zoe.zoedepth_forward(sd, "coarse_branch.", lr, zc)),
                                          patch_process_shape=w["pps"], image_raw_shape=w["raw"], patch_split_num=w["split"])
    hr = rand_image(3, 1, *w["raw"])
    pred, ref, cp, o_cp = _one_tile(model, ora, hr, (270, 1440), dict(image_raw_shape=w["raw"], patch_split_num=w["split"]))
    ar_c, mx_c = absrel(cp, o_cp)
    ar, mx = absrel(pred, ref)
    print(f"{name} one tile, {prec}: coarse AbsRel {ar_c:.3e} max|d| {mx_c:.3e}; refined tile AbsRel {ar:.3e} max|d| {mx:.3e} "
          f"(depth {float(ref.min()):.2f}..{float(ref.max()):.2f})")
    tol = ABSREL_TOL if prec == "bf16x3" else 1e-5
    assert tuple(pred.shape) == tuple(ref.shape) == (1, 1, *w["pps"]) and ar_c < tol and ar < tol, (ar_c, ar, mx)


def test_full_width_v1_tile_vs_fp32_oracle(P):
    """ONE real tile of the ViT-block-heavy V1 model (v1_dav2l_4k_r32, configs/patchrefiner_dav2/pr_u4k.py): the 24-block ViT-L
    + DPT head on the 448 x 448 crop and the full-width FusionUnet (512-channel 3x3 convs at 448^2), bf16x3 vs the fp32 oracle"""
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
    name = "v1_dav2l_4k_r32"
    w = WORKLOADS[name]
    sd = W.synth_state_dict(state_spec(name), seed=0)
    model = build_model(model_config(name, prec="bf16x3"))
    model.load_state_dict(sd, strict=True)
    cc = W.dav2_cfg({**w["coarse"], "max_depth": 80.0})
    ora = o_tiling.OraclePatchRefiner(sd, cc, W.dav2_cfg({**w["fine"], "max_depth": 80.0}), patch_process_shape=w["pps"],
                                      image_raw_shape=w["raw"], patch_split_num=w["split"])
    hr = rand_image(4, 1, *w["raw"])
    pred, ref, cp, o_cp = _one_tile(model, ora, hr, (810, 480), dict(image_raw_shape=w["raw"], patch_split_num=w["split"]))
    ar_c, mx_c = absrel(cp, o_cp)
    ar, mx = absrel(pred, ref)
    print(f"{name} one tile, bf16x3: coarse AbsRel {ar_c:.3e} max|d| {mx_c:.3e}; refined tile AbsRel {ar:.3e} max|d| {mx:.3e}")
    assert ar_c < ABSREL_TOL and ar < ABSREL_TOL, (ar_c, ar, mx)


def test_baseline_config1_1080p_m1_full_frame_vs_oracle(P):
    """BASELINE config[1] whole: Depth-Anything-V2 ViT-S, 1080 x 1920, 2 x 2 patches, cai-mode m1 (V1 PatchRefiner, SURVEY.md 8d
    C2) -- the full frame in bf16x3 against the fp32 oracle; output 896 x 896."""
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
    name = "v1_dav2s_1080p_m1"
    w = WORKLOADS[name]
    sd = W.synth_state_dict(state_spec(name), seed=0)
    model = build_model(model_config(name, prec="bf16x3"))
    model.load_state_dict(sd, strict=True)
    cfg = W.dav2_cfg({**w["coarse"], "max_depth": 80.0})
    ora = o_tiling.OraclePatchRefiner(sd, cfg, cfg, patch_process_shape=w["pps"], image_raw_shape=w["raw"], patch_split_num=w["split"])
    hr = rand_image(6, 1, *w["raw"])
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
    ref, _ = ora(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=ora.resizer(hr), image_hr=hr)
    hr_d = hr.to(DEV)
    got, _ = model(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=model.resizer(hr_d), image_hr=hr_d)
    ar, mx = absrel(got, ref)
    print(f"{name} full frame bf16x3: AbsRel {ar:.3e} max|d| {mx:.3e}")
    assert tuple(got.shape) == tuple(ref.shape) == (1, 1, 896, 896) and ar < ABSREL_TOL, (ar, mx)


@pytest.mark.parametrize("prec", ["bf16x3", "f16f6"])  # (f16f6 = bench.py's default: on these configs the gate-tail kernel's K = 512 concat instantiation)
@pytest.mark.parametrize("name,tiles", [("v2_dav2l_4k_r64", 113), ("v2_dav2l_4k_r128", 177)])
def test_baseline_config3_config4_single_gpu_properties(P, name, tiles, prec):
    """BASELINE config[3] (DAv2 ViT-L, 4K, r64) and one frame of config[4] (r128) on ONE GPU, through size-independent
    properties: tile count, output shape, finite and inside [0, max_depth], bit-identical when re-run with another batch
    size / stream count, and bit-identical when the tiles are computed as 8 rank shards and exchanged (the multi-GPU path of
    these two configs, emulated on one device)."""
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
    w = WORKLOADS[name]
    model = build_model(model_config(name, prec=prec, max_batch=14, n_streams=3))
    model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
    hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(5)).to(DEV)
    lr = model.resizer(hr)
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])

    def run(**kw):
        random.seed(621)
        return model(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tc, image_lr=lr, image_hr=hr, **kw)[0]

    a = run()
    if prec == "f16f6":  # the whole frame against the same frame in bf16x3: the north star's tolerance, with the margin printed
        assert model.refiner_fusion_model.f16f6 and getattr(model, "f6_guarded_frames", 0) >= 1
        mb = build_model(model_config(name, prec="bf16x3", max_batch=14, n_streams=3))
        mb.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
        random.seed(621)
        ref = mb(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tc, image_lr=lr, image_hr=hr)[0]
        ar, mx = absrel(a, ref)
        print(f"\n{name} whole frame f16f6 vs bf16x3: AbsRel {ar:.2e} (tolerance 1e-4: margin {1e-4 / max(ar, 1e-30):.0f}x), max |d| {mx:.2e}")
        assert ar < ABSREL_TOL / 2, ar
        del mb, ref
    assert sum(len(p["raw"]) for p in model.last_plan) == tiles == w["patches"]
    assert tuple(a.shape) == (1, 1, 2160, 3840) and bool(torch.isfinite(a).all())
    assert float(a.min()) >= 0.0 and float(a.max()) <= 80.0 * 1.0001
    model.max_batch, model.n_streams = 10, 2
    assert torch.equal(a, run())
    from conftest import ShardEmulation
    emu = ShardEmulation(model, 8)
    emu.record()
    for r in range(8):
        assert run(shard=(r, 8), gather_dst=0) is None
    emu.deliver()
    assert torch.equal(a, run(shard=(0, 8), gather_dst=0))
