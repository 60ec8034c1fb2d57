"""Generate tests/golden/*.npz by running the REFERENCE implementation (imported from
/root/reference under stubs, see refharness.py) on seeded inputs + synthetic weights,
and assert the oracle restatement agrees while doing so.

Run in the build container only:   cd /root/reference && PYTHONPATH=/root/repo python /root/repo/oracle/make_golden.py
The committed .npz files hold inputs' seeds and the reference's outputs (data only).
"""
from __future__ import annotations

import json
import os
import random
import re
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.chdir("/root/reference")

from oracle import refharness  # noqa: E402

refharness.install()

from oracle import dav2 as o_dav2, fusion as o_fusion, mnv4 as o_mnv4, ops as o_ops, tiling as o_tiling  # noqa: E402
from oracle import zoe as o_zoe  # noqa: E402
from oracle.cases import (TINY_DAV2, TINY_FUSION_UNET, TINY_BIDIR, TINY_BIDIR_NOC2F, E2E_V1, E2E_V2, E2E_V2Z, rand_image, tiny_dav2_sd,  # noqa: E402
                          e2e_v1_sd, e2e_v2_sd, e2e_v2z_sd)
from patchrefinerv2_amd import weights as W  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_grad_enabled(False)
torch.manual_seed(0)


def maxdiff(a, b):
    return float((a - b).abs().max())


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in arrs.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024:.0f} KiB)")


# --------------------------------------------------------------------------------------
def build_ref_dav2(model_cfg, sd, prefix=""):
    """Reference DepthAnythingV2 with (possibly reduced) dims, loaded strictly from ``sd``."""
    dpt = refharness.ref_module("external.depth_anything_v2.dpt")
    dinov2 = refharness.ref_module("external.depth_anything_v2.dinov2")
    cfg = W.dav2_cfg(model_cfg)
    vit = cfg["vit"]
    m = dpt.DepthAnythingV2.__new__(dpt.DepthAnythingV2)
    torch.nn.Module.__init__(m)
    m.intermediate_layer_idx = {cfg["encoder"]: vit["taps"]}
    m.max_depth = cfg["max_depth"]
    m.encoder = cfg["encoder"]
    from functools import partial
    m.pretrained = dinov2.DinoVisionTransformer(
        img_size=vit["img_size"], patch_size=vit["patch"], embed_dim=vit["dim"], depth=vit["depth"],
        num_heads=vit["heads"], mlp_ratio=vit["mlp_ratio"], init_values=1.0, ffn_layer="mlp", block_chunks=0,
        num_register_tokens=0, interpolate_antialias=False, interpolate_offset=0.1,
        block_fn=partial(dinov2.Block, attn_class=dinov2.MemEffAttention))
    m.depth_head = dpt.DPTHead(vit["dim"], cfg["features"], False, out_channels=cfg["out_channels"], use_clstoken=False)
    m.register_buffer("refiner_pixel_mean", torch.Tensor([0.485, 0.456, 0.406]).view(-1, 1, 1), False)
    m.register_buffer("refiner_pixel_std", torch.Tensor([0.229, 0.224, 0.225]).view(-1, 1, 1), False)
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    m.load_state_dict(sub, strict=True)
    return m.eval()


def g_dav2():
    print("[dav2_tiny]")
    sd = tiny_dav2_sd()
    m = build_ref_dav2(TINY_DAV2["model_cfg"], sd)
    cfg = W.dav2_cfg(TINY_DAV2["model_cfg"])
    res = {}
    for tag, (h, w) in TINY_DAV2["inputs"].items():
        x = rand_image(TINY_DAV2["seed"], 2, h, w)
        ref = m(x, return_final_centers=True)
        ora = o_dav2.dav2_forward(sd, "", x, cfg)
        d = maxdiff(ref["metric_depth"], ora["metric_depth"])
        print(f"  {tag}: depth range [{float(ref['metric_depth'].min()):.3f}, {float(ref['metric_depth'].max()):.3f}]"
              f" oracle-vs-ref max|d| {d:.2e}")
        assert d < 1e-4, d
        for k in ref["temp_features"]:
            dd = maxdiff(ref["temp_features"][k], ora["temp_features"][k])
            assert dd < 1e-3 * max(1.0, float(ref["temp_features"][k].abs().max())), (k, dd)
        res[f"{tag}_depth"] = ref["metric_depth"]
        res[f"{tag}_x_d0"] = ref["temp_features"]["x_d0"]
        res[f"{tag}_final_feat_mean"] = ref["temp_features"]["midas_final_feat"].mean(dim=1)
    # spec check against the real-size reference modules (names + shapes only)
    names = {}
    for enc, mc in (("vits", dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384])),
                    ("vitl", dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024]))):
        dpt = refharness.ref_module("external.depth_anything_v2.dpt")
        full = dpt.DepthAnythingV2(**mc)
        ref_shapes = {k: tuple(v.shape) for k, v in full.state_dict().items()}
        spec = {k: tuple(v) for k, v in W.dav2_spec("", mc).items()}
        assert ref_shapes == spec, (set(ref_shapes) ^ set(spec))
        names[enc] = {k: list(v) for k, v in ref_shapes.items()}
        print(f"  spec {enc}: {len(spec)} tensors, {sum(int(np.prod(v)) for v in spec.values()) / 1e6:.1f} M elements == reference")
    with open(os.path.join(OUT, "dav2_state_dict_shapes.json"), "w") as f:
        json.dump(names, f)
    save("dav2_tiny", **res)


def g_vit_block():
    """One real-size ViT-S block (D=384, 6 heads) on 1025 tokens; row slice stored."""
    print("[vit_block]")
    blk_mod = refharness.ref_module("external.depth_anything_v2.dinov2_layers.block")
    att_mod = refharness.ref_module("external.depth_anything_v2.dinov2_layers.attention")
    spec = {k: v for k, v in W.dinov2_spec("", W.vit_cfg("vits", depth=1)).items() if k.startswith("blocks.0.")}
    sd = W.synth_state_dict(spec, seed=11)
    blk = blk_mod.Block(384, 6, mlp_ratio=4, qkv_bias=True, proj_bias=True, ffn_bias=True, init_values=1.0,
                        attn_class=att_mod.MemEffAttention).eval()
    blk.load_state_dict({k[len("blocks.0."):]: v for k, v in sd.items()}, strict=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 1025, 384, generator=g)
    ref = blk(x)
    ora = o_dav2.block(sd, "blocks.0.", x, 6)
    print(f"  oracle-vs-ref max|d| {maxdiff(ref, ora):.2e}")
    assert maxdiff(ref, ora) < 1e-4
    save("vit_block", out_rows=ref[0, ::64].clone())


def g_fusion_unet():
    print("[fusion_unet]")
    fm = refharness.ref_module("estimator.models.blocks.fusion_model")
    c = TINY_FUSION_UNET
    spec = W.fusion_unet_spec("", c["input_chl"], c["temp_chl"], c["dec_chl"])
    sd = W.synth_state_dict(spec, seed=c["seed"])
    m = fm.FusionUnet(input_chl=list(c["input_chl"]), temp_chl=list(c["temp_chl"]), dec_chl=list(c["dec_chl"])).eval()
    m.load_state_dict(sd, strict=True)
    inp = c["make_inputs"]()
    ref = m(c_feat=[t.clone() for t in inp["c_feat"]], f_feat=[t.clone() for t in inp["f_feat"]], pred1=inp["pred1"],
            pred2=inp["pred2"], update_base=inp["pred1"])
    ora = o_fusion.fusion_unet(sd, "", inp["c_feat"], inp["f_feat"], inp["pred1"], inp["pred2"], update_base=inp["pred1"])
    print(f"  out range [{float(ref.min()):.3f},{float(ref.max()):.3f}] oracle-vs-ref max|d| {maxdiff(ref, ora):.2e}")
    assert maxdiff(ref, ora) < 1e-4
    # real-size spec check (V1 DAv2-L cfg, configs/patchrefiner_dav2/pr_u4k.py:44-48)
    with torch.device("meta"):
        full = fm.FusionUnet(input_chl=[256, 512, 512, 512, 512, 512], temp_chl=[128, 256, 256, 256, 256, 256],
                             dec_chl=[256, 256, 256, 256, 128])
    assert {k: tuple(v.shape) for k, v in full.state_dict().items()} == \
        {k: tuple(v) for k, v in W.fusion_unet_spec("", [256, 512, 512, 512, 512, 512], [128, 256, 256, 256, 256, 256],
                                                    [256, 256, 256, 256, 128]).items()}
    save("fusion_unet", out=ref)


def g_bidir():
    print("[bidir_fusion]")
    bm = refharness.ref_module("estimator.models.blocks.bi_directional_fusion_model")
    c = TINY_BIDIR
    # 'coarse-gated' = every released V2 config; 'coarse-fusion' / 'self-agg' = the C2FModule ablations (:355-372)
    for c2f_type, name in (("coarse-gated", "bidir_fusion"), ("coarse-fusion", "bidir_fusion_coarse_fusion"),
                           ("self-agg", "bidir_fusion_self_agg"), ("only-gate", "bidir_fusion_only_gate")):
        spec = W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"],
                                   coarse2fine_type=c2f_type)
        sd = W.synth_state_dict(spec, seed=c["seed"])
        m = bm.BiDirectionalFusion(coarse2fine=True, coarse2fine_type=c2f_type, coarse_chl=list(c["coarse_chl"]),
                                   fine_chl=list(c["fine_chl"]), fine_chl_after_coarse2fine=list(c["fine_chl_after"]),
                                   temp_chl=list(c["temp_chl"]), dec_chl=list(c["dec_chl"])).eval()
        missing = m.load_state_dict(sd, strict=True)
        print(f"  {c2f_type}: strict load ok:", missing)
        res = {}
        for tag in ("same", "resized"):
            inp = c["make_inputs"](tag)
            ref = m(c_feat=[t.clone() for t in inp["c_feat"]], f_feat=[t.clone() for t in inp["f_feat"]],
                    pred1=inp["pred1"], pred2=inp["pred2"], update_base=inp["pred1"])
            ora = o_fusion.bidirectional_fusion(sd, "", inp["c_feat"], inp["f_feat"], inp["pred1"], inp["pred2"],
                                                update_base=inp["pred1"], coarse2fine_type=c2f_type)
            print(f"  {c2f_type}/{tag}: out range [{float(ref.min()):.3f},{float(ref.max()):.3f}] oracle-vs-ref max|d| {maxdiff(ref, ora):.2e}")
            assert maxdiff(ref, ora) < 2e-4
            res[tag] = ref
        save(name, **res)
    # coarse2fine=False: no c2f module, six refiner maps (level 0 = the x2 copy of level 1) and the caller's pred2 (zeros)
    c = TINY_BIDIR_NOC2F
    spec = W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"], coarse2fine=False)
    sd = W.synth_state_dict(spec, seed=c["seed"])
    m = bm.BiDirectionalFusion(coarse2fine=False, coarse2fine_type="coarse-gated", coarse_chl=list(c["coarse_chl"]),
                               fine_chl=list(c["fine_chl"]), fine_chl_after_coarse2fine=list(c["fine_chl_after"]),
                               temp_chl=list(c["temp_chl"]), dec_chl=list(c["dec_chl"])).eval()
    print("  coarse2fine=False: strict load ok:", m.load_state_dict(sd, strict=True))
    res = {}
    for tag in ("same", "resized"):
        inp = c["make_inputs"](tag)
        ref = m(c_feat=[t.clone() for t in inp["c_feat"]], f_feat=[t.clone() for t in inp["f_feat"]],
                pred1=inp["pred1"], pred2=inp["pred2"], update_base=inp["pred1"])
        ora = o_fusion.bidirectional_fusion(sd, "", inp["c_feat"], inp["f_feat"], inp["pred1"], inp["pred2"],
                                            update_base=inp["pred1"], coarse2fine=False)
        print(f"  coarse2fine=False/{tag}: out range [{float(ref.min()):.3f},{float(ref.max()):.3f}] oracle-vs-ref max|d| {maxdiff(ref, ora):.2e}")
        assert maxdiff(ref, ora) < 2e-4
        res[tag] = ref
    save("bidir_fusion_no_c2f", **res)
    # BiDirectionalFusionHeavy (:519-560), as its three configs use it: coarse2fine=False
    spec = W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"], coarse2fine=False,
                               heavy=True)
    sd = W.synth_state_dict(spec, seed=c["seed"])
    m = bm.BiDirectionalFusionHeavy(coarse2fine=False, coarse2fine_type="coarse-gated", coarse_chl=list(c["coarse_chl"]),
                                    fine_chl=list(c["fine_chl"]), fine_chl_after_coarse2fine=list(c["fine_chl_after"]),
                                    temp_chl=list(c["temp_chl"]), dec_chl=list(c["dec_chl"])).eval()
    print("  heavy: strict load ok:", m.load_state_dict(sd, strict=True))
    res = {}
    for tag in ("same", "resized"):
        inp = c["make_inputs"](tag)
        ref = m(c_feat=[t.clone() for t in inp["c_feat"]], f_feat=[t.clone() for t in inp["f_feat"]],
                pred1=inp["pred1"], pred2=inp["pred2"], update_base=inp["pred1"])
        ora = o_fusion.bidirectional_fusion(sd, "", inp["c_feat"], inp["f_feat"], inp["pred1"], inp["pred2"],
                                            update_base=inp["pred1"], coarse2fine=False)
        print(f"  heavy/{tag}: out range [{float(ref.min()):.3f},{float(ref.max()):.3f}] oracle-vs-ref max|d| {maxdiff(ref, ora):.2e}")
        assert maxdiff(ref, ora) < 2e-4
        res[tag] = ref
        # the raw offset as well (update_base=None): with these synthetic weights it is ~1e-2 on a base of 0..10, so the sum hides it
        off = m(c_feat=[t.clone() for t in inp["c_feat"]], f_feat=[t.clone() for t in inp["f_feat"]], pred1=inp["pred1"], pred2=inp["pred2"])
        ora = o_fusion.bidirectional_fusion(sd, "", inp["c_feat"], inp["f_feat"], inp["pred1"], inp["pred2"], coarse2fine=False)
        print(f"  heavy/{tag}: offset range [{float(off.min()):.4f},{float(off.max()):.4f}] oracle-vs-ref max|d| {maxdiff(off, ora):.2e}")
        assert maxdiff(off, ora) < 1e-6
        res[tag + "_offset"] = off
    save("bidir_fusion_heavy", **res)


def g_tiling():
    print("[tiling]")
    utils = refharness.ref_module("estimator.models.utils")
    bp = refharness.ref_module("estimator.models.baseline_pretrain")
    res = {}
    # RunningAverageMap sequence incl. zero-weight pixels and resize
    g = torch.Generator().manual_seed(3)
    H, W_ = 48, 64
    avg0 = torch.rand(H, W_, generator=g) * 10
    cnt0 = (torch.rand(H, W_, generator=g) > 0.3).float() * torch.rand(H, W_, generator=g)
    ref = utils.RunningAverageMap(avg0.clone(), cnt0.clone())
    ora = o_tiling.RunningAverageMap(avg0.clone(), cnt0.clone())
    for i in range(3):
        p = torch.rand(H, W_, generator=g) * 10
        c = torch.rand(H, W_, generator=g)
        c[:, : 8 * (i + 1)] = 0
        ref.update(p.clone(), c.clone())
        ora.update(p.clone(), c.clone())
    assert maxdiff(ref.average_map, ora.average_map) == 0 and maxdiff(ref.count_map, ora.count_map) == 0
    res["ram_avg"], res["ram_cnt"] = ref.average_map.clone(), ref.count_map.clone()
    ref.resize((108, 100))
    ora.resize((108, 100))
    assert maxdiff(ref.average_map, ora.average_map) == 0 and maxdiff(ref.count_map, ora.count_map) == 0
    res["ram_avg_rs"], res["ram_cnt_rs"] = ref.average_map.clone(), ref.count_map.clone()
    # generatemask through the reference function (blur itself = oracle restatement; unpinned)
    for (h, w) in ((384, 512), (448, 448), (540, 960), (56, 84)):
        mk = utils.generatemask((h, w), border=0.15)
        assert np.array_equal(mk, o_ops.generatemask((h, w), border=0.15))
        res[f"mask_{h}x{w}_rowsum"] = mk.sum(axis=1)
        res[f"mask_{h}x{w}_colsum"] = mk.sum(axis=0)
        res[f"mask_{h}x{w}_zeros"] = np.array([(mk == 0).sum()])
    # prepare_tile_cfg + tile plans through the reference's own regular_tile / random_tile
    plans = {}

    class Probe(bp.BaselinePretrain):
        def __init__(self, pps):
            torch.nn.Module.__init__(self)
            self.patch_process_shape = pps
            self.log = []

            class R:
                def __call__(s, x):
                    return torch.zeros(1, 3, pps[0], pps[1])
            self.resizer = R()

        def infer_forward(self, imgs_crop, *a):
            return torch.zeros(imgs_crop.shape[0], 1, *self.patch_process_shape)

        def coarse_postprocess_test(self, bboxs, bboxs_feat, **kw):
            self.log.append((bboxs.clone(), bboxs_feat.clone()))
            return dict(coarse_depth_roi=torch.zeros(bboxs.shape[0], 1, 1, 1), coarse_feats_roi=[torch.zeros(bboxs.shape[0], 1, 1, 1)])

    for name, (raw, split, pps, mode) in dict(
            c2=((1080, 1920), (2, 2), (448, 448), "m1"), c3=((2160, 3840), (4, 4), (384, 512), "r32"),
            c4=((2160, 3840), (4, 4), (448, 448), "r64"), small=((216, 384), (2, 2), (56, 84), "r8")).items():
        pr = Probe(pps)
        tc = pr.prepare_tile_cfg(raw, split)
        otc = o_tiling.prepare_tile_cfg(pps, raw, split)
        assert {k: list(v) for k, v in tc.items()} == {k: list(v) for k, v in otc.items()}
        random.seed(621)
        img = torch.zeros(3, *raw)
        blur = torch.zeros(pps)
        tt = dict(coarse_prediction=None, coarse_features=None)
        rh, rw = tc["patch_raw_shape"]
        avg = pr.regular_tile([0, 0], [0, 0], img, init_flag=True, tile_temp=tt, blur_mask=blur, tile_cfg=tc, process_num=4)
        if mode != "m1":
            for off, offp in (([0, rw // 2], [0, pps[1] // 2]), ([rh // 2, 0], [pps[0] // 2, 0]),
                              ([rh // 2, rw // 2], [pps[0] // 2, pps[1] // 2])):
                avg = pr.regular_tile(off, offp, img, init_flag=False, tile_temp=tt, blur_mask=blur, avg_depth_map=avg,
                                      tile_cfg=tc, process_num=4)
        if mode[0] == "r":
            avg.resize(tc["image_raw_shape"])
            blur_r = torch.zeros(rh, rw)
            for _ in range(int(mode[1:]) // 4):
                avg = pr.random_tile(img, tile_temp=tt, blur_mask=blur_r, avg_depth_map=avg, tile_cfg=tc, process_num=4)
        bb = torch.cat([b for b, _ in pr.log]).numpy()
        bf = torch.cat([f for _, f in pr.log]).numpy()
        plans[name] = dict(raw=list(raw), split=list(split), pps=list(pps), mode=mode, n=int(bb.shape[0]))
        res[f"plan_{name}_bboxs"] = bb.astype(np.int32)
        res[f"plan_{name}_bboxs_feat"] = bf.astype(np.float32)
        print(f"  plan {name}: {bb.shape[0]} tiles")
    with open(os.path.join(OUT, "tile_plans.json"), "w") as f:
        json.dump(plans, f)
    # Resize (both flavours) on a small random crop
    rda = refharness.ref_module("external.depth_anything.transform").Resize
    rzoe = refharness.ref_module("external.zoedepth.models.base_models.midas").Resize
    crop = torch.rand(1, 3, 54, 96, generator=torch.Generator().manual_seed(9))
    a = rda(84, 56, keep_aspect_ratio=False, ensure_multiple_of=14, resize_method="minimal")(crop)
    assert maxdiff(a, o_ops.resize_da(crop, 84, 56)) == 0
    assert maxdiff(a, o_ops.bilinear_ac_explicit(crop, (56, 84))) < 1e-6
    b = rda(512, 384, keep_aspect_ratio=False, ensure_multiple_of=14, resize_method="minimal")(crop)
    assert tuple(b.shape[-2:]) == (378, 518)
    z = rzoe(512, 384, keep_aspect_ratio=False, ensure_multiple_of=32, resize_method="minimal")(crop)
    assert maxdiff(z, o_ops.resize_zoe(crop)) == 0
    res["resize_da_56x84"] = a
    res["resize_zoe_rowmean"] = z.mean(dim=-1)
    save("tiling", **res)


def build_ref_zoedepth(zcfg, sd, prefix=""):
    """Reference ZoeDepth over the DepthAnything core, built by the reference's own ZoeDepth.build
    (type='DA-ZoeDepth' path, zoedepth_v1.py:296-311 -> DepthAnythingCore.build), strict-loaded from ``sd``."""
    zmod = refharness.ref_module("external.zoedepth.models.zoedepth.zoedepth_v1")
    m = zmod.ZoeDepth.build(**zcfg)
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    m.load_state_dict(sub, strict=True)
    return m.eval()


def g_zoedepth():
    print("[zoedepth]")
    from oracle.cases import ZOE_DA
    c = ZOE_DA
    sd = W.synth_state_dict(W.zoedepth_spec("", c["zcfg"]), seed=c["seed"])
    m = build_ref_zoedepth(dict(c["zcfg"]), sd)
    z = W.zoedepth_cfg(c["zcfg"])
    res = {}
    for tag, (h, w) in c["inputs"].items():
        x = rand_image(c["seed"], 2, h, w)
        ref = m(x, return_final_centers=True)
        ora = o_zoe.zoedepth_forward(sd, "", x, z)
        d = maxdiff(ref["metric_depth"], ora["metric_depth"])
        print(f"  {tag}: depth range [{float(ref['metric_depth'].min()):.3f}, {float(ref['metric_depth'].max()):.3f}] "
              f"oracle-vs-ref max|d| {d:.2e}")
        assert d < 1e-4 * float(ref["metric_depth"].max()), d
        for k in ("x_d0", "x_blocks_feat_0", "x_blocks_feat_3", "midas_final_feat"):
            dd = maxdiff(ref["temp_features"][k], ora["temp_features"][k])
            assert dd < 1e-4 * max(1.0, float(ref["temp_features"][k].abs().max())), (k, dd)
        res[f"{tag}_depth"] = ref["metric_depth"]
        res[f"{tag}_x_d0"] = ref["temp_features"]["x_d0"]
        res[f"{tag}_final_feat_mean"] = ref["temp_features"]["midas_final_feat"].mean(dim=1)
    save("zoedepth_da", **res)


def _patch_torch_load(sd_for):
    real = torch.load

    def fake(path, *a, **k):
        for key, fn in sd_for.items():
            if key in str(path):
                return fn()
        return real(path, *a, **k)
    torch.load = fake
    return real


def _build_ref_e2e_v1(c, sd):
    """the reference's PatchRefiner class over reduced DA2 backbones, synthetic weights loaded by name"""
    dpt = refharness.ref_module("external.depth_anything_v2.dpt")
    # reduced DA2 dims: the reference class is built by name, so swap its ctor for the reduced builder
    orig = dpt.DepthAnythingV2
    pr_mod = refharness.ref_module("estimator.models.patchrefiner")
    refharness.ref_module("estimator.models.blocks.fusion_model")
    built = []

    def small_da2(**kw):
        prefix = "coarse_branch." if not built else "refiner_fine_branch."
        built.append(prefix)
        return build_ref_dav2({**c["da2_cfg"], "max_depth": kw["max_depth"]}, sd, prefix)
    pr_mod.DepthAnythingV2 = small_da2
    real = _patch_torch_load({"dummy_da2": lambda: None})
    try:
        torch.nn.Module.load_state_dict_orig = torch.nn.Module.load_state_dict
        cfg = refharness.AttrDict(c["ref_config"])
        # DA2 branch calls self.coarse_branch.load_state_dict(torch.load(pretrained)) -> tolerate None
        orig_lsd = torch.nn.Module.load_state_dict
        torch.nn.Module.load_state_dict = lambda self, s, strict=True, **k: (orig_lsd(self, s, strict=strict, **k) if s is not None else None)
        m = pr_mod.PatchRefiner(cfg).eval()
        torch.nn.Module.load_state_dict = orig_lsd
    finally:
        torch.load = real
        pr_mod.DepthAnythingV2 = orig
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    return m


def g_e2e_v1():
    print("[e2e_v1]")
    c = E2E_V1
    sd = e2e_v1_sd()
    m = _build_ref_e2e_v1(c, sd)
    ora = o_tiling.OraclePatchRefiner(sd, W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}),
                                      W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}),
                                      patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"])
    image_hr = rand_image(c["seed"], 1, *c["raw"])
    image_lr = m.resizer(image_hr)
    res = {}
    for mode in c["modes"]:
        random.seed(621)
        ref, log = m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=dict(image_raw_shape=c["raw"], patch_split_num=c["split"]),
                     image_lr=image_lr, image_hr=image_hr)
        random.seed(621)
        out, olog = ora(mode="infer", cai_mode=mode, process_num=4, tile_cfg=dict(image_raw_shape=c["raw"], patch_split_num=c["split"]),
                        image_lr=image_lr, image_hr=image_hr)
        d = maxdiff(ref, out)
        print(f"  {mode}: out {tuple(ref.shape)} range [{float(ref.min()):.3f},{float(ref.max()):.3f}] coarse range "
              f"[{float(log['coarse_prediction'].min()):.3f},{float(log['coarse_prediction'].max()):.3f}] oracle-vs-ref max|d| {d:.2e}")
        assert d < 2e-4, d
        res[mode] = ref
        res[mode + "_coarse"] = log["coarse_prediction"]
    save("e2e_v1", **res)


def _g_e2e_v2(tag, c, sd, zoe=False, post_build=None, unmapped_prefix=None, golden_stride=1):
    print(f"[{tag}]")
    import timm

    class Enc(torch.nn.Module):
        """timm.create_model stand-in (timm absent): the oracle's MNv4-small restatement with a
        3-channel stem; the reference then performs its 4-channel stem surgery on ``conv_stem``."""
        default_cfg = dict(mean=W.MNV4_SMALL["mean"], std=W.MNV4_SMALL["std"])

        def __init__(self):
            super().__init__()
            self.conv_stem = torch.nn.Conv2d(3, 32, 3, 2, 1, bias=False)
            self.sd = {}  # plain dict (not registered): filled from the synthetic state dict below

        def forward(self, x):
            d = dict(self.sd)
            d["conv_stem.weight"] = self.conv_stem.weight
            return o_mnv4.mnv4_features(d, "", x)

    timm.create_model = lambda name, pretrained=True, features_only=True: Enc()
    prp = refharness.ref_module("estimator.models.patchrefinerplus")
    refharness.ref_module("estimator.models.blocks.lightweight_refiner")
    refharness.ref_module("estimator.models.blocks.bi_directional_fusion_model")
    orig = prp.DepthAnythingV2
    if not zoe:
        prp.DepthAnythingV2 = lambda **kw: build_ref_dav2({**c["da2_cfg"], "max_depth": kw["max_depth"]}, sd, "coarse_branch.")
    prp.Conv2dSame = torch.nn.Conv2d
    orig_lsd = torch.nn.Module.load_state_dict
    real = _patch_torch_load({"dummy_da2": lambda: None})
    try:
        torch.nn.Module.load_state_dict = lambda self, s, strict=True, **k: (orig_lsd(self, s, strict=strict, **k) if s is not None else None)
        m = prp.PatchRefinerPlus(refharness.AttrDict(c["ref_config"])).eval()
    finally:
        torch.nn.Module.load_state_dict = orig_lsd
        torch.load = real
        prp.DepthAnythingV2 = orig
    # load everything by name (the encoder's buffers go through Enc._load_from_state_dict)
    res_load = m.load_state_dict(dict(sd), strict=False)
    ep = "refiner_fine_branch.refiner_encoder."
    if unmapped_prefix is None:
        assert not res_load.missing_keys, res_load.missing_keys[:5]
        assert all(k.startswith(ep) for k in res_load.unexpected_keys)
    else:  # a sub-module that is a stand-in with other parameter names: filled by post_build
        assert all(k.startswith(unmapped_prefix) for k in res_load.missing_keys), res_load.missing_keys[:5]
        assert all(k.startswith(ep) or k.startswith(unmapped_prefix) for k in res_load.unexpected_keys)
        post_build(m)
    m.refiner_fine_branch.refiner_encoder.sd = {k[len(ep):]: v for k, v in sd.items() if k.startswith(ep)}
    assert tuple(m.refiner_fine_branch.refiner_encoder.conv_stem.weight.shape) == (32, 4, 3, 3)
    if zoe:
        zc = W.zoedepth_cfg(c["zcfg"])
        ora = o_tiling.OraclePatchRefinerPlus(
            sd, None, coarse_fn=lambda lr: o_dav2.coarse_features(o_zoe.zoedepth_forward(sd, "coarse_branch.", lr, zc)),
            patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"],
            resizer="zoe" if c["ref_config"]["coarse_branch"]["type"] == "ZoeDepth" else "da")
    else:
        ora = o_tiling.OraclePatchRefinerPlus(sd, W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}),
                                              patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"])
    image_hr = rand_image(c["seed"], 1, *c["raw"])
    image_lr = m.resizer(image_hr)
    res = {}
    for mode in c["modes"]:
        random.seed(621)
        ref, log = m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=dict(image_raw_shape=c["raw"], patch_split_num=c["split"]),
                     image_lr=image_lr, image_hr=image_hr)
        random.seed(621)
        out, _ = ora(mode="infer", cai_mode=mode, process_num=4, tile_cfg=dict(image_raw_shape=c["raw"], patch_split_num=c["split"]),
                     image_lr=image_lr, image_hr=image_hr)
        d = maxdiff(ref, out)
        print(f"  {mode}: out {tuple(ref.shape)} range [{float(ref.min()):.3f},{float(ref.max()):.3f}] oracle-vs-ref max|d| {d:.2e}")
        assert d < 2e-4, d
        res[mode] = ref[..., ::golden_stride, ::golden_stride]  # large frames: a regular sub-grid of the map is committed
    save(tag, **res)


def g_e2e_v2():
    _g_e2e_v2("e2e_v2", E2E_V2, e2e_v2_sd())


def g_e2e_v2z():
    """PatchRefinerPlus with the reference's own 'DA-ZoeDepth' coarse branch (ZoeDepth.build -> DepthAnythingCore vitl)."""
    _g_e2e_v2("e2e_v2z", E2E_V2Z, e2e_v2z_sd(), zoe=True)


def g_output_stage():
    """Tester output stage: colour maps and evaluation metrics of the reference (estimator/utils/color.py, metric.py)."""
    print("[output_stage]")
    import types
    refharness.install()
    for name, attrs in (("skimage", {}), ("skimage.feature", {"canny": None}), ("kornia", {}), ("imageio", {})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    u = types.ModuleType("estimator.utils")
    u.__path__ = [os.path.join(refharness.REF, "estimator/utils")]
    sys.modules["estimator.utils"] = u
    import importlib
    color = importlib.import_module("estimator.utils.color")
    metric = importlib.import_module("estimator.utils.metric")
    import matplotlib
    if not hasattr(matplotlib.cm, "get_cmap"):  # removed in matplotlib 3.9; the reference pins an older one
        matplotlib.cm.get_cmap = lambda name: matplotlib.colormaps[name]
    g = torch.Generator().manual_seed(77)
    h, w = 40, 56
    gt = torch.rand(1, 1, h, w, generator=g) * 12 + 0.05          # some pixels outside (min, max) = (0.1, 10)
    pred = (gt * (1 + 0.2 * torch.randn(1, 1, h, w, generator=g))).clamp(min=-0.5)
    pred[0, 0, 3, 5] = float("nan")
    pred[0, 0, 4, 6] = float("inf")
    pred_lo = torch.nn.functional.interpolate(pred, (20, 28), mode="bilinear")  # resized inside compute_metrics
    disp = 1.0 / gt.squeeze().numpy()
    edges = metric.get_boundaries(disp * 40, th=1, dilation=0)
    out = dict(gt=gt.numpy(), pred=pred.numpy(), pred_lo=pred_lo.numpy(), edges=edges)
    c1 = color.colorize(pred.clone(), cmap="Spectral", vminp=0, vmaxp=100)
    vv = gt.clone()
    vv[0, 0, :4, :4] = -99
    c2 = color.colorize(vv, cmap="gray_r")
    out.update(color_spectral=c1, color_gray=c2)
    e = metric.compute_errors(gt.squeeze().numpy()[2:30, 2:40], np.abs(pred.squeeze().numpy()[2:30, 2:40]) + 0.01)
    out.update({f"err_{k}": np.float64(v) for k, v in e.items()})
    m1 = metric.compute_metrics(gt, pred.clone(), garg_crop=False, eigen_crop=False, dataset="u4k", min_depth_eval=0.1,
                                max_depth_eval=10, disp_gt_edges=torch.from_numpy(edges))
    out.update({f"m1_{k}": np.float64(float(v)) for k, v in m1.items()})
    m2 = metric.compute_metrics(gt, pred_lo.clone(), garg_crop=True, eigen_crop=False, dataset="kitti", min_depth_eval=0.1,
                                max_depth_eval=10)
    out.update({f"m2_{k}": np.float64(float(v)) for k, v in m2.items()})
    see = metric.soft_edge_error(np.abs(pred.squeeze().numpy()), gt.squeeze().numpy(), radius=2)
    out.update(see_r2=see)
    np.savez_compressed(os.path.join(OUT, "output_stage.npz"), **out)
    print("  wrote output_stage.npz", {k: float(v) for k, v in m1.items()})


def g_convnext():
    """LightWeightRefiner (reference class, convnext branch) over HuggingFace transformers' ConvNext as the stand-in for
    timm.create_model('convnext_large', features_only=True): pins the block arithmetic of oracle/convnext.py against an
    independent implementation of the published architecture and the wrapper (upsample_convx, 2x copy, order) against
    the reference's own code."""
    print("[convnext]")
    import timm
    from transformers import ConvNextConfig, ConvNextModel
    from oracle import convnext as o_cx
    from oracle.cases import CONVNEXT_REFINER, convnext_refiner_sd, convnext_refiner_inputs
    arch = CONVNEXT_REFINER["arch"]
    sd = convnext_refiner_sd()

    class Enc(torch.nn.Module):
        default_cfg = dict(mean=arch["mean"], std=arch["std"])

        def __init__(self):
            super().__init__()
            self.hf = ConvNextModel(ConvNextConfig(num_channels=4, hidden_sizes=list(arch["dims"]), depths=list(arch["depths"])))

        def forward(self, x):
            return list(self.hf(x, output_hidden_states=True).hidden_states[1:])  # the four stage outputs, no final norm

    timm.create_model = lambda name, pretrained=True, features_only=True: Enc()
    lwr = refharness.ref_module("estimator.models.blocks.lightweight_refiner")
    d0 = arch["dims"][0]
    m = lwr.LightWeightRefiner("convnext_large", True, encoder_channels=[d0 // 2] + list(arch["dims"])).eval()

    def hf_name(k):  # timm (flattened features_only) -> transformers
        k = k.replace("stem_0.", "embeddings.patch_embeddings.").replace("stem_1.", "embeddings.layernorm.")
        k = re.sub(r"stages_(\d)\.downsample\.", r"encoder.stages.\1.downsampling_layer.", k)
        k = re.sub(r"stages_(\d)\.blocks\.(\d+)\.", r"encoder.stages.\1.layers.\2.", k)
        return (k.replace(".gamma", ".layer_scale_parameter").replace("conv_dw.", "dwconv.").replace(".norm.", ".layernorm.")
                .replace("mlp.fc1.", "pwconv1.").replace("mlp.fc2.", "pwconv2."))

    ep = "refiner_encoder."
    hf_sd = {hf_name(k[len(ep):]): v for k, v in sd.items() if k.startswith(ep)}
    res = m.refiner_encoder.hf.load_state_dict(hf_sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith("layernorm.") for k in res.missing_keys), res  # final norm: unused
    m.upsample_convx[0].weight.data.copy_(sd["upsample_convx.0.weight"])
    m.upsample_convx[0].bias.data.copy_(sd["upsample_convx.0.bias"])
    crop, depth = convnext_refiner_inputs()
    feats, out_depth = m(crop, depth)
    o_feats, o_out = o_cx.lightweight_refiner_convnext(sd, "", crop, depth, arch)
    assert len(feats) == len(o_feats) == 6 and float(out_depth.abs().max()) == 0.0 and o_out.shape == out_depth.shape
    for i, (a, b) in enumerate(zip(feats, o_feats)):
        d = maxdiff(a, b)
        print(f"  feat {i}: {tuple(a.shape)} |x|max {float(a.abs().max()):.2f} oracle-vs-ref max|d| {d:.2e}")
        assert a.shape == b.shape and d < 1e-4 * max(1.0, float(a.abs().max())), d
    save("convnext_refiner", **{f"feat{i}": f for i, f in enumerate(feats)})


def g_effnet():
    """LightWeightRefiner (reference class) over HuggingFace transformers' EfficientNet as the stand-in for
    timm.create_model('tf_efficientnet_b5_ap', features_only=True): pins oracle/effnet.py (MBConv + SE + SiLU + BN eps 1e-3 +
    'SAME' padding, feature taps) against an independent port of the TensorFlow model."""
    print("[effnet]")
    import timm
    from transformers import EfficientNetConfig, EfficientNetModel
    from oracle import effnet as o_eff
    from oracle.cases import EFFNET_REFINER, effnet_refiner_sd, effnet_refiner_inputs
    arch = EFFNET_REFINER["arch"]
    sd = effnet_refiner_sd()
    blocks = W.effnet_blocks(arch)
    taps = [i + 1 for i, B in enumerate(blocks) if B["tap"]]  # hidden_states[0] = the stem output

    class Enc(torch.nn.Module):
        default_cfg = dict(mean=arch["mean"], std=arch["std"])

        def __init__(self):
            super().__init__()
            cfg = EfficientNetConfig(num_channels=4, width_coefficient=arch["width"], depth_coefficient=arch["depth"],
                                     depthwise_padding=[], hidden_dim=W._round_filters(1280, arch["width"]))
            self.hf = EfficientNetModel(cfg)

        def forward(self, x):
            hs = self.hf(x, output_hidden_states=True).hidden_states
            return [hs[t] for t in taps]

    timm.create_model = lambda name, pretrained=True, features_only=True: Enc()
    lwr = refharness.ref_module("estimator.models.blocks.lightweight_refiner")
    m = lwr.LightWeightRefiner("tf_efficientnet_b5_ap", True).eval()

    def hf_name(k):  # timm -> transformers
        if k.startswith("conv_stem."):
            return k.replace("conv_stem.", "embeddings.convolution.")
        if k.startswith("bn1."):
            return k.replace("bn1.", "embeddings.batchnorm.")
        mm = re.match(r"blocks\.(\d)\.(\d+)\.(.*)", k)
        si, j, rest = int(mm.group(1)), int(mm.group(2)), mm.group(3)
        idx = [i for i, B in enumerate(blocks) if B["name"] == f"blocks.{si}.{j}."][0]
        ds = blocks[idx]["kind"] == "ds"
        table = ([("conv_dw.", "depthwise_conv.depthwise_conv."), ("bn1.", "depthwise_conv.depthwise_norm."),
                  ("conv_pw.", "projection.project_conv."), ("bn2.", "projection.project_bn.")] if ds else
                 [("conv_pw.", "expansion.expand_conv."), ("bn1.", "expansion.expand_bn."),
                  ("conv_dw.", "depthwise_conv.depthwise_conv."), ("bn2.", "depthwise_conv.depthwise_norm."),
                  ("conv_pwl.", "projection.project_conv."), ("bn3.", "projection.project_bn.")])
        table += [("se.conv_reduce.", "squeeze_excite.reduce."), ("se.conv_expand.", "squeeze_excite.expand.")]
        for a, b in table:
            if rest.startswith(a):
                return f"encoder.blocks.{idx}.{b}{rest[len(a):]}"
        raise KeyError(k)

    ep = "refiner_encoder."
    hf_sd = {hf_name(k[len(ep):]): v for k, v in sd.items() if k.startswith(ep)}
    res = m.refiner_encoder.hf.load_state_dict(hf_sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys[:5]
    assert all(k.startswith("encoder.top_") or k.endswith("num_batches_tracked") for k in res.missing_keys), res.missing_keys[:8]
    crop, depth = effnet_refiner_inputs()
    feats, out_depth = m(crop, depth)
    o_feats, o_out = o_eff.lightweight_refiner_effnet(sd, "", crop, depth, arch)
    assert len(feats) == len(o_feats) == 6 and float(out_depth.abs().max()) == 0.0 and o_out.shape == out_depth.shape
    for i, (a, b) in enumerate(zip(feats, o_feats)):
        d = maxdiff(a, b)
        print(f"  feat {i}: {tuple(a.shape)} |x|max {float(a.abs().max()):.2f} oracle-vs-ref max|d| {d:.2e}")
        assert a.shape == b.shape and d < 1e-4 * max(1.0, float(a.abs().max())), d
    save("effnet_refiner", **{f"feat{i}": f for i, f in enumerate(feats[:5])})  # (feat5 = the 2x bilinear copy of feat4)


class _MidasHF(torch.nn.Module):
    """What ``torch.hub.load("AyaanShah2204/MiDaS", "DPT_BEiT_L_384")`` returns, as far as the reference touches it
    (midas.py:260-318): a module whose forward gives the relative depth [B,H,W] and which exposes ``scratch.output_conv``
    (children()[3] = the ReLU behind the 128->32 conv), ``scratch.refinenet1..4``, ``scratch.layer4_rn`` and ``pretrained``.
    The arithmetic is HuggingFace transformers' independent port of MiDaS 3.1 DPT-BEiT (DPTForDepthEstimation over BeitBackbone)."""

    def __init__(self, beit):
        super().__init__()
        from transformers import BeitConfig, DPTConfig, DPTForDepthEstimation
        b = beit
        bc = BeitConfig(image_size=[b["window"][0] * b["patch"], b["window"][1] * b["patch"]], patch_size=b["patch"], hidden_size=b["dim"],
                        num_hidden_layers=b["depth"], num_attention_heads=b["heads"], intermediate_size=b["dim"] * b["mlp_ratio"],
                        use_relative_position_bias=True, use_shared_relative_position_bias=False, use_absolute_position_embeddings=False,
                        layer_scale_init_value=0.1, use_mask_token=False, reshape_hidden_states=False, layer_norm_eps=1e-6,
                        out_features=[f"stage{t + 1}" for t in b["taps"]], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                        drop_path_rate=0.0)
        dc = DPTConfig(backbone_config=bc, neck_hidden_sizes=list(b["out_channels"]), fusion_hidden_size=b["features"],
                       reassemble_factors=[4, 2, 1, 0.5], readout_type="project", use_batch_norm_in_fusion_residual=False, add_projection=False,
                       head_in_index=-1, hidden_act="gelu", hidden_size=b["dim"], use_bias_in_fusion_residual=True)
        self.hf = DPTForDepthEstimation(dc)
        self.pretrained = self.hf.backbone
        self.scratch = torch.nn.Module()
        self.scratch.output_conv = self.hf.head.head
        for r in (1, 2, 3, 4):
            setattr(self.scratch, f"refinenet{r}", self.hf.neck.fusion_stage.layers[4 - r])
        self.scratch.layer4_rn = self.hf.neck.convs[3]

    def forward(self, x):
        return self.hf(pixel_values=x).predicted_depth

    def load_midas(self, sd, prefix, beit):
        """synthetic weights under the MiDaS / timm names (patchrefinerv2_amd/weights.py::midas_beit_spec) -> transformers' names"""
        D = beit["dim"]
        out = {}
        for k, v in sd.items():
            if not k.startswith(prefix):
                continue
            n = k[len(prefix):]
            if n.startswith("pretrained.model."):
                n = n[len("pretrained.model."):]
                if n == "cls_token":
                    out["backbone.beit.embeddings.cls_token"] = v
                elif n.startswith("patch_embed.proj."):
                    out["backbone.beit.embeddings.patch_embeddings.projection." + n.rsplit(".", 1)[1]] = v
                else:
                    _, i, rest = n.split(".", 2)
                    L = f"backbone.beit.layers.{i}."
                    m = {"gamma_1": "lambda_1", "gamma_2": "lambda_2", "attn.q_bias": "attention.q_proj.bias", "attn.v_bias": "attention.v_proj.bias",
                         "attn.relative_position_bias_table": "relative_position_bias.relative_position_bias_table",
                         "attn.proj.weight": "attention.o_proj.weight", "attn.proj.bias": "attention.o_proj.bias"}
                    if rest == "attn.qkv.weight":
                        out[L + "attention.q_proj.weight"], out[L + "attention.k_proj.weight"], out[L + "attention.v_proj.weight"] = v[:D], v[D:2 * D], v[2 * D:]
                    elif rest in m:
                        out[L + m[rest]] = v
                    else:
                        out[L + rest.replace("norm1.", "layernorm_before.").replace("norm2.", "layernorm_after.")] = v
            elif n.startswith("pretrained.act_postprocess"):
                i = int(n[len("pretrained.act_postprocess")]) - 1
                rest = n.split(".", 2)[2]
                R = "neck.reassemble_stage."
                if rest.startswith("0.project.0."):
                    out[f"{R}readout_projects.{i}.0." + rest.rsplit(".", 1)[1]] = v
                elif rest.startswith("3."):
                    out[f"{R}layers.{i}.projection." + rest[2:]] = v
                else:
                    out[f"{R}layers.{i}.resize." + rest[2:]] = v
            elif n.startswith("scratch.layer"):
                out[f"neck.convs.{int(n[len('scratch.layer')]) - 1}.weight"] = v
            elif n.startswith("scratch.refinenet"):
                r = int(n[len("scratch.refinenet")])
                rest = n.split(".", 2)[2]
                rest = rest.replace("out_conv.", "projection.").replace("resConfUnit", "residual_layer").replace(".conv", ".convolution")
                out[f"neck.fusion_stage.layers.{4 - r}.{rest}"] = v
            elif n.startswith("scratch.output_conv."):
                out["head.head." + n[len("scratch.output_conv."):]] = v
            else:
                raise KeyError(k)
        res = self.hf.load_state_dict(out, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys[:5]
        assert not res.missing_keys, res.missing_keys[:8]


def _install_midas_hub(beit_holder):
    real = torch.hub.load
    torch.hub.load = lambda repo, name, **kw: _MidasHF(beit_holder["beit"])
    return real


def g_midas_beit():
    """type='ZoeDepth': the reference's own ZoeDepth + MidasCore classes (hooks, PrepForMidas, metric-bins head) over
    transformers' DPT-BEiT behind the torch.hub stand-in, synthetic weights mapped by name: oracle/midas_beit.py == that."""
    print("[midas_beit]")
    from oracle.cases import ZOE_BEIT
    c = ZOE_BEIT
    z = W.zoedepth_cfg(c["zcfg"])
    beit = z["core"]["beit"]
    sd = W.synth_state_dict(W.zoedepth_spec("", c["zcfg"]), seed=c["seed"])
    zmod = refharness.ref_module("external.zoedepth.models.zoedepth.zoedepth_v1")
    real = _install_midas_hub(dict(beit=beit))
    try:
        m = zmod.ZoeDepth.build(**{k: v for k, v in c["zcfg"].items() if k != "beit"}).eval()
    finally:
        torch.hub.load = real
    assert type(m.core).__name__ == "MidasCore" and m.core.output_channels == (256,) * 5
    m.core.core.load_midas(sd, "core.core.", beit)
    res_load = m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("core.core.")}, strict=False)
    assert all(k.startswith("core.core.") for k in res_load.missing_keys) and not res_load.unexpected_keys
    res = {}
    for tag, (h, w) in c["inputs"].items():
        x = rand_image(c["seed"], 2, h, w)
        ref = m(x, return_final_centers=True)
        ora = o_zoe.zoedepth_forward(sd, "", x, z)
        d = maxdiff(ref["metric_depth"], ora["metric_depth"])
        print(f"  {tag}: depth range [{float(ref['metric_depth'].min()):.3f}, {float(ref['metric_depth'].max()):.3f}] oracle-vs-ref max|d| {d:.2e}")
        assert d < 1e-4 * float(ref["metric_depth"].max()), d
        for k in ("x_d0", "x_blocks_feat_0", "x_blocks_feat_1", "x_blocks_feat_2", "x_blocks_feat_3", "midas_final_feat"):
            dd = maxdiff(ref["temp_features"][k], ora["temp_features"][k])
            assert dd < 1e-4 * max(1.0, float(ref["temp_features"][k].abs().max())), (k, dd)
        res[f"{tag}_depth"] = ref["metric_depth"]
        res[f"{tag}_x_d0"] = ref["temp_features"]["x_d0"]
        res[f"{tag}_x_blocks_feat_3_mean"] = ref["temp_features"]["x_blocks_feat_3"].mean(dim=1)
        res[f"{tag}_final_feat_mean"] = ref["temp_features"]["midas_final_feat"].mean(dim=1)
    save("zoedepth_beit", **res)


def g_e2e_v2b():
    """PatchRefinerPlus as configs/patchrefinerv2_zoedepth/v2_mobile_u4k.py wires it (coarse_branch type='ZoeDepth' -> MidasCore /
    BEiT, ResizeZoe 384 x 512), reduced BEiT, the reference's own classes end to end."""
    from oracle.cases import E2E_V2B, e2e_v2b_sd
    c = E2E_V2B
    sd = e2e_v2b_sd()
    beit = W.zoedepth_cfg(c["zcfg"])["core"]["beit"]
    real = _install_midas_hub(dict(beit=beit))
    try:
        cfg = dict(c)
        cfg["ref_config"] = {**c["ref_config"], "coarse_branch": {k: v for k, v in c["ref_config"]["coarse_branch"].items() if k != "beit"}}
        _g_e2e_v2("e2e_v2b", cfg, sd, zoe=True, unmapped_prefix="coarse_branch.core.core.",
                  post_build=lambda m: m.coarse_branch.core.core.load_midas(sd, "coarse_branch.core.core.", beit), golden_stride=3)
    finally:
        torch.hub.load = real


def g_baseline():
    """The reference's own BaselinePretrain class (estimator/models/baseline_pretrain.py), both targets, with the vendored
    'DA-ZoeDepth' backbone: pins OracleBaselinePretrain (tile plan with N * process_num random tiles, border-0.1 mask)."""
    print("[baseline]")
    from oracle.cases import BASELINE, baseline_kwargs, baseline_sd
    c = BASELINE
    bp = refharness.ref_module("estimator.models.baseline_pretrain")
    sd = baseline_sd()
    z = W.zoedepth_cfg(c["zcfg"])
    branch_fn = lambda x: o_zoe.zoedepth_forward(sd, "", x, z)["metric_depth"]  # noqa: E731
    tc = dict(image_raw_shape=c["raw"], patch_split_num=c["split"])
    image_hr = rand_image(c["seed"], 1, *c["raw"])
    res = {}
    for target in ("coarse", "fine"):
        kw = baseline_kwargs(target)
        for b in ("coarse_branch", "fine_branch"):
            kw[b] = refharness.AttrDict(kw[b])
        m = bp.BaselinePretrain(**kw).eval()
        r = m.load_dict(dict(sd))
        assert not r.missing_keys and not r.unexpected_keys, (r.missing_keys[:4], r.unexpected_keys[:4])
        assert sorted(m.get_save_dict()) == sorted(sd)
        ora = o_tiling.OracleBaselinePretrain(branch_fn, target=target, patch_process_shape=c["pps"], image_raw_shape=c["raw"],
                                              patch_split_num=c["split"])
        image_lr = m.resizer(image_hr)
        assert maxdiff(image_lr, ora.resizer(image_hr)) == 0.0
        for mode in (["m1"] if target == "coarse" else c["fine_modes"]):
            with torch.no_grad():
                random.seed(621)
                ref, log = m(mode="infer", image_lr=image_lr, image_hr=image_hr, depth_gt=None, tile_cfg=tc, cai_mode=mode, process_num=4)
                random.seed(621)
                out, olog = ora(mode="infer", image_lr=image_lr, image_hr=image_hr, tile_cfg=tc, cai_mode=mode, process_num=4)
            d = maxdiff(ref, out)
            print(f"  {target} {mode}: out {tuple(ref.shape)} range [{float(ref.min()):.3f},{float(ref.max()):.3f}] oracle-vs-ref max|d| {d:.2e}; log keys {sorted(log)}")
            assert d < 1e-4 and sorted(log) == sorted(olog), (d, sorted(log), sorted(olog))
            res[f"{target}_{mode}"] = ref
    save("baseline", **res)


def g_consistency():
    """The reference's own ``Tester.run_consistency`` (estimator/tester/tester.py:211-321) driven over the reference's
    PatchRefiner (E2E_V1 reduced backbones) on one synthetic 2160 x 3840 frame prepared exactly as the U4K consistency dataset
    does (u4k_dataset.py:62-65,158-185: 16 crops of 540 x 960 shifted towards the centre by multiples of overlap / 2, each
    resized with the dataset's ResizeDA, pre-normalised bboxs): pins ``oracle.tiling.run_consistency`` and gives the product's
    ``Tester.run_consistency`` a known answer (tests/golden/consistency.npz)."""
    print("[consistency]")
    import types
    c = E2E_V1
    sd = e2e_v1_sd()
    m = _build_ref_e2e_v1(c, sd)
    # -- the modules tester.py imports at its top and never needs on this path ------------------------------------------
    class _PB:
        def __init__(self, *a, **k):
            pass

        def update(self):
            pass
    mm = sys.modules["mmengine"]
    for name in ("mmengine.analysis", "mmengine.optim", "mmengine.dist", "mmengine.utils", "mmengine.fileio", "wandb", "tqdm",
                 "skimage", "skimage.io", "kornia", "PIL", "PIL.Image", "matplotlib.pyplot"):
        if name not in sys.modules:
            sys.modules[name] = refharness._AnyAttr(name)
    sys.modules["mmengine.dist"].get_dist_info = lambda: (0, 1)
    sys.modules["mmengine.dist"].collect_results_gpu = lambda results, n: results
    sys.modules["mmengine.dist"].collect_results_cpu = lambda results, n: results
    sys.modules["mmengine.utils"].ProgressBar = _PB
    sys.modules["mmengine.utils"].mkdir_or_exist = lambda p: os.makedirs(p, exist_ok=True)
    for sub in ("analysis", "optim", "dist", "utils", "fileio"):
        setattr(mm, sub, sys.modules["mmengine." + sub])
    sys.modules["mmengine.optim"].build_optim_wrapper = None
    sys.modules["mmengine.fileio"].dump = None
    sys.modules["tqdm"].tqdm = lambda x, *a, **k: x
    sys.modules["skimage"].io = sys.modules["skimage.io"]
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    u = types.ModuleType("estimator.utils")
    for n in ("colorize", "colorize_infer_pfv1", "colorize_rescale", "extract_edges", "rescale_tensor"):
        setattr(u, n, None)
    sys.modules["estimator.utils"] = u
    mu = refharness.ref_module("estimator.models.utils")
    if not hasattr(mu, "HookTool"):
        mu.HookTool = None
    tp = types.ModuleType("estimator.tester")
    tp.__path__ = [os.path.join(refharness.REF, "estimator/tester")]
    sys.modules["estimator.tester"] = tp
    import importlib
    ref_tester = importlib.import_module("estimator.tester.tester")

    # -- one frame, prepared as U4KDataset.__getitem__ does in consistency mode -----------------------------------------
    overlap, H, Wd, h, w = 270, 2160, 3840, 540, 960
    ph, pw = c["pps"]
    image = rand_image(c["seed"] + 5, 1, H, Wd)[0]
    h_start_list = [int(0 + 3 * overlap / 2), int(540 + overlap / 2), int(1080 - overlap / 2), int(1620 - 3 * overlap / 2)]
    w_start_list = [int(0 + 3 * overlap / 2), int(960 + overlap / 2), int(1920 - overlap / 2), int(2880 - 3 * overlap / 2)]
    crops, bboxs = [], []
    for hs in h_start_list:
        for ws in w_start_list:
            crops.append(m.resizer(image[:, hs:hs + h, ws:ws + w].unsqueeze(0)).squeeze(0))
            bbox = torch.tensor([ws, hs, ws + w, hs + h])
            bboxs.append(torch.tensor([bbox[0] / Wd * pw, bbox[1] / H * ph, bbox[2] / Wd * pw, bbox[3] / H * ph]))
    batch = dict(image_lr=m.resizer(image.unsqueeze(0)), image_hr=torch.tensor([[H, Wd]]), crops_image_hr=torch.stack(crops)[None],
                 depth_gt=torch.ones(1, 1, 8, 8), crop_depths=torch.ones(1, 16, 1, ph, pw), bboxs=torch.stack(bboxs)[None],
                 img_file_basename=["frame0"])

    class _DS:
        def __len__(self):
            return 1

        def evaluate_consistency(self, results):
            return results
    ds = _DS()
    ds.h_start_list, ds.w_start_list = h_start_list, w_start_list

    class _DL:
        dataset = ds
        batch_sampler = [[0]]

        def __iter__(self):
            return iter([batch])
    cfg = refharness.AttrDict(collect_input_args=["image_lr", "image_hr", "crops_image_hr", "depth_gt", "crop_depths", "bboxs"])
    t = ref_tester.Tester(cfg, refharness.AttrDict(rank=0, save=False, work_dir="/tmp"), _DL(), m)
    got = {}
    real_cat = torch.cat
    real_cuda = torch.Tensor.cuda
    # capture what the reference computes (it only hands the scalars to dataset.evaluate_consistency)
    ds.evaluate_consistency = lambda results: got.setdefault("results", results)
    crops_seen = []
    real_interp = torch.nn.functional.interpolate

    def spy_interp(x, size=None, **k):
        y = real_interp(x, size, **k)
        if size is not None and tuple(size) == (540, 960):
            crops_seen.append(y.squeeze().clone())
        return y
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref_tester.F.interpolate = spy_interp
    try:
        t.run_consistency()
    finally:
        torch.Tensor.cuda = real_cuda
        ref_tester.F.interpolate = real_interp
    assert torch.cat is real_cat
    ce_ref = float(got["results"][0]["consistency_error"])
    ref_crops = torch.stack(crops_seen[:16])

    ora = o_tiling.OraclePatchRefiner(sd, W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}),
                                      W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]}),
                                      patch_process_shape=c["pps"], image_raw_shape=[H, Wd], patch_split_num=[4, 4])
    ce, o_crops = o_tiling.run_consistency(ora, image.unsqueeze(0), overlap=overlap)
    d = maxdiff(ref_crops, o_crops)
    print(f"  reference consistency_error {ce_ref:.6f}, oracle {ce:.6f}; crops oracle-vs-ref max|d| {d:.2e} (crop depth "
          f"{float(ref_crops.min()):.3f}..{float(ref_crops.max()):.3f})")
    assert d < 2e-4 and abs(ce - ce_ref) < 1e-6 * max(1.0, ce_ref), (d, ce, ce_ref)
    np.savez_compressed(os.path.join(OUT, "consistency.npz"), consistency_error=np.float64(ce_ref), overlap=np.int64(overlap),
                        image_seed=np.int64(c["seed"] + 5), crops_strided=ref_crops[:, ::9, ::16].numpy())
    print("  wrote consistency.npz")


if __name__ == "__main__":
    which = sys.argv[1:] or ["dav2", "vit_block", "fusion_unet", "bidir", "tiling", "e2e_v1", "e2e_v2", "zoedepth", "e2e_v2z", "baseline"]
    for w in which:
        globals()["g_" + w]()
    print("done")
