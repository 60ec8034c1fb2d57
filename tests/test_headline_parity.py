"""Parity of the BENCHMARKED workload itself (bench.py's default, BASELINE config[2] as the reference configures it):
``v2_zoe_4k_r32`` = PatchRefinerPlus over ZoeDepth / MiDaS DPT_BEiT_L_384, P = 384 x 512, MobileNetV4-S refiner, full-width
BiDirectionalFusion, bf16x3 arithmetic -- against the fp32 oracle on the same synthetic weights, at full size:

  * one real tile (coarse BEiT-L forward + ROI pyramid + refiner + fusion), with assertions on the refinement OFFSET and on two
    intermediate maps, not only on coarse + offset (whose AbsRel divides any error of the fusion network by depth / offset);
  * one whole 4K frame in m1 (16 tiles: tiling, ROI boxes, blend at 1536 x 2048) against the oracle's frame driver;
  * the 4K r32 frame at the bench's batching (41 tiles per batch, 3 streams, next-frame coarse prefetch): determinism and the
    8-rank patch-shard emulation;
  * one tile of ``v1_zoe_4k_r32`` (the README's pr_u4k.py: BEiT-L on every tile + full-width FusionUnet).

The synthetic heads are re-scaled (oracle.cases.widen_depth_range) so that the coarse depth spans an order of magnitude and the
offsets are metres.  Numbers of the last GPU run: profiles/r04_parity_numbers.txt (printed with -s).
"""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dav2 as o_dav2, fusion as o_fusion, tiling as o_tiling, zoe as o_zoe  # noqa: E402
from oracle.cases import rand_image, widen_depth_range  # noqa: E402
from patchrefinerv2_amd import weights as W  # noqa: E402

DEV = "cuda"
ABSREL_TOL = 1e-4     # north star: per-pixel AbsRel vs the PyTorch reference
REL_L2_TOL = 1e-3     # refinement offset / intermediate maps, relative L2
torch.set_grad_enabled(False)


def absrel(out, ref, min_depth=1e-3):
    out, ref = out.detach().cpu().double(), torch.as_tensor(ref).double()
    m = ref > min_depth
    return float(((out - ref).abs()[m] / ref[m]).mean()), float((out - ref).abs().max())


def rel_l2(out, ref):
    out, ref = out.detach().cpu().double(), torch.as_tensor(ref).double()
    return float((out - ref).norm() / ref.norm().clamp_min(1e-30))


def _pair(name, prec="bf16x3", **model_kw):
    """(product model, oracle driver, workload dict) of a named full-size workload on widened synthetic weights"""
    from patchrefinerv2_amd import models  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
    w = WORKLOADS[name]
    sd = widen_depth_range(W.synth_state_dict(state_spec(name), seed=0))
    model = build_model(model_config(name, prec=prec, **model_kw))
    model.load_state_dict(sd, strict=True)
    if not w.get("zoe"):  # DepthAnythingV2 coarse branch (type='DA2': sigmoid * max_depth head, no bins to widen; the offsets are scaled)
        kw = dict(patch_process_shape=w["pps"], image_raw_shape=w["raw"], patch_split_num=w["split"])
        ccfg = W.dav2_cfg({**w["coarse"], "max_depth": 80.0})
        assert w["kind"] == "PatchRefinerPlus"
        return model, o_tiling.OraclePatchRefinerPlus(sd, ccfg, **kw), w
    zc = W.zoedepth_cfg(w["zoe"])
    kw = dict(patch_process_shape=w["pps"], image_raw_shape=w["raw"], patch_split_num=w["split"],
              resizer="zoe" if w.get("zoe_type") == "ZoeDepth" else "da")
    if w["kind"] == "PatchRefinerPlus":
        ora = o_tiling.OraclePatchRefinerPlus(
            sd, None, coarse_fn=lambda lr: o_dav2.coarse_features(o_zoe.zoedepth_forward(sd, "coarse_branch.", lr, zc)), **kw)
    else:
        fz = W.zoedepth_cfg(w["fine_zoe"])
        ora = o_tiling.OraclePatchRefiner(sd, None, None, coarse_fn=lambda lr: o_zoe.zoedepth_forward(sd, "coarse_branch.", lr, zc),
                                          fine_fn=lambda x: o_zoe.zoedepth_forward(sd, "refiner_fine_branch.", x, fz), **kw)
    return model, ora, w


def _one_tile(model, ora, hr, tile, tile_cfg):
    """one tile through the product's per-patch path and through the oracle's, both with intermediate maps recorded:
    dict(pred, base, coarse, trace) for each side"""
    from patchrefinerv2_amd.ops import Feat
    hr_d = hr.to(DEV)
    lr_d = model.resizer(hr_d)
    feats, cp = model.coarse_forward(lr_d)
    cd = Feat(cp.view(1, cp.shape[-2], cp.shape[-1], 1))
    tc = model.prepare_tile_cfg(tile_cfg["image_raw_shape"], tile_cfg["patch_split_num"])
    t_dev = torch.tensor([tile], dtype=torch.int32, device=DEV)
    boxes = torch.from_numpy(model._boxes([tile], tc)).to(DEV)
    crops, rois, droi = model._prepare_batch(hr_d[0].contiguous(), t_dev, boxes, tc, feats, cd)
    model.refiner_fusion_model.trace = {}
    try:
        pred = model.infer_forward(crops, rois, droi)
        got = dict(pred=pred, base=droi.buf.view(1, 1, droi.h, droi.w).clone(), coarse=cp,
                   trace={k: v.clone() for k, v in model.refiner_fusion_model.trace.items()})
    finally:
        model.refiner_fusion_model.trace = None
    rh, rw = tc["patch_raw_shape"]
    lr = lr_d.cpu()
    o_feats, o_cp = ora.coarse_forward(lr)
    o_crops, bb = ora._crops(hr[0], [tile[0]], [tile[1]], rh, rw)
    post = o_tiling.coarse_postprocess_test(o_cp, o_feats, o_tiling.bboxs_to_feat(bb, tile_cfg["image_raw_shape"], ora.patch_process_shape),
                                            ora.patch_process_shape[0])
    o_fusion.TRACE = {}
    try:
        ref = ora.infer_forward(o_crops, post)
        want = dict(pred=ref, base=post["coarse_depth_roi"], coarse=o_cp, trace=dict(o_fusion.TRACE))
    finally:
        o_fusion.TRACE = None
    return got, want


def _report(tag, got, want, keys):
    ar_c, mx_c = absrel(got["coarse"], want["coarse"])
    ar, mx = absrel(got["pred"], want["pred"])
    r = {k: rel_l2(got["trace"][k], want["trace"][k]) for k in keys}
    # the offset as the caller sees it: prediction minus the coarse ROI it refines
    r["pred-base"] = rel_l2(got["pred"] - got["base"], want["pred"] - want["base"])
    c, o = want["coarse"], want["trace"]["offset"]
    print(f"{tag}: coarse AbsRel {ar_c:.3e} max|d| {mx_c:.3e} (coarse depth {float(c.min()):.2f}..{float(c.max()):.2f}, "
          f"1%..99% {float(c.flatten().kthvalue(max(1, c.numel() // 100)).values):.2f}..{float(c.flatten().kthvalue(c.numel() - c.numel() // 100).values):.2f}); "
          f"refined tile AbsRel {ar:.3e} max|d| {mx:.3e}; offset mean {float(o.mean()):+.3f} std {float(o.std()):.3f} m; relative L2: "
          + ", ".join(f"{k} {v:.3e}" for k, v in r.items()))
    return ar_c, ar, r


@pytest.mark.parametrize("arith", ["bf16x3", "f16f6"])
def test_headline_v2_zoe_tile_vs_fp32_oracle(arith):
    """(a) ONE real tile of the benchmarked workload, bf16x3 vs the fp32 oracle: AbsRel of the coarse map and of the refined tile
    <= 1e-4; relative L2 of the refinement offset, of the coarse-to-fine module's output (depth + last 32-channel feature) and of
    the last decoder stage <= 1e-3; the coarse depth spans more than 10x.  ``f16f6``: the same tolerances with GatedConvUnit.conv in
    the fp16 + block-scaled-fp6 arithmetic (csrc/conv3x3_f6.hip); the margins are printed."""
    name = "v2_zoe_4k_r32"
    model, ora, w = _pair(name, prec=arith)
    if arith == "f16f6":  # the mode is really on: every 256-channel unit of the c2f module carries the fp16 + fp6 image
        R = model.refiner_fusion_model._packed["refine"]
        assert all("conv_f6" in R[r][u] for r in range(1, 6) for u in ("u1", "u2")) and all("f0a_f6" in R[r]["u2"] for r in range(1, 6))
    assert model.resizer.kind == "zoe" and tuple(w["pps"]) == (384, 512)
    hr = rand_image(3, 1, *w["raw"])
    got, want = _one_tile(model, ora, hr, (270, 1440), dict(image_raw_shape=w["raw"], patch_split_num=w["split"]))
    ar_c, ar, r = _report(f"{name} one tile {arith}", got, want, ["c2f_depth", "c2f_last", "dec_last", "offset"])
    print(f"{name} one tile {arith}: margins AbsRel {ABSREL_TOL / ar:.1f}x, worst relative L2 {REL_L2_TOL / max(r.values()):.1f}x")
    c = want["coarse"].flatten()
    lo, hi = float(c.kthvalue(c.numel() // 100).values), float(c.kthvalue(c.numel() - c.numel() // 100).values)
    assert hi / lo >= 8.0 and float(c.max()) / float(c.min()) >= 10.0, (lo, hi)    # the bins are exercised
    assert float(want["trace"]["offset"].abs().mean()) > 0.1                        # the offsets are not noise-level
    assert tuple(got["pred"].shape) == tuple(want["pred"].shape) == (1, 1, 384, 512)
    assert ar_c < ABSREL_TOL and ar < ABSREL_TOL, (ar_c, ar)
    assert all(v <= REL_L2_TOL for v in r.values()), r


@pytest.mark.parametrize("arith", ["bf16x3", "f16f6"])
def test_v2_dav2l_r64_tile_vs_fp32_oracle(arith):
    """(a') ONE full-width tile of ``v2_dav2l_4k_r64`` -- the workload of BASELINE configs [3] / [4]: DepthAnythingV2 ViT-L coarse
    branch at 448 x 448 (24 blocks, 1025 tokens), MobileNetV4-S refiner, BiDirectionalFusion with coarse_chl[0] = 128 (the 128-channel
    gate kernel at full resolution, the coarse-tap tables of every level) -- bf16x3 vs the fp32 oracle with the offset / intermediate
    assertions of the headline tile"""
    name = "v2_dav2l_4k_r64"
    model, ora, w = _pair(name, prec=arith)   # (f16f6: GatedConvUnit.conv and the unit tail over the K = 512 concat, same tolerances)
    assert tuple(w["pps"]) == (448, 448) and w["fusion"]["coarse_chl"][0] == 128
    hr = rand_image(5, 1, *w["raw"])
    got, want = _one_tile(model, ora, hr, (405, 1200), dict(image_raw_shape=w["raw"], patch_split_num=w["split"]))
    ar_c, ar, r = _report(f"{name} one tile {arith}", got, want, ["c2f_depth", "c2f_last", "dec_last", "offset"])
    print(f"{name} one tile {arith}: margins AbsRel {ABSREL_TOL / ar:.1f}x, worst relative L2 {REL_L2_TOL / max(r.values()):.1f}x")
    assert tuple(got["pred"].shape) == tuple(want["pred"].shape) == (1, 1, 448, 448)
    assert float(want["trace"]["offset"].abs().mean()) > 0.01
    assert ar_c < ABSREL_TOL and ar < ABSREL_TOL, (ar_c, ar)
    assert all(v <= REL_L2_TOL for v in r.values()), r


def test_headline_v1_zoe_tile_vs_fp32_oracle():
    """(c) one tile of v1_zoe_4k_r32 (configs/patchrefiner_zoedepth/pr_u4k.py): ZoeDepth / BEiT-L on the crop + the full-width
    FusionUnet (512-channel 3x3 convs), bf16x3 vs the fp32 oracle, with the offset assertion"""
    name = "v1_zoe_4k_r32"
    model, ora, w = _pair(name)
    hr = rand_image(4, 1, *w["raw"])
    got, want = _one_tile(model, ora, hr, (810, 480), dict(image_raw_shape=w["raw"], patch_split_num=w["split"]))
    ar_c, ar, r = _report(f"{name} one tile bf16x3", got, want, ["dec_last", "offset"])
    assert ar_c < ABSREL_TOL and ar < ABSREL_TOL, (ar_c, ar)
    assert all(v <= REL_L2_TOL for v in r.values()), r


def test_headline_whole_4k_m1_frame_vs_oracle():
    """(d) one WHOLE 4K frame of the benchmarked model in m1 -- 16 tiles of 540 x 960, ROI boxes, per-tile networks, paste at
    1536 x 2048 -- through the product's frame driver (bench batching: 41 per batch -> one batch of 16, 3 streams) against the
    oracle's frame driver (the reference's regular_tile loop in batches of process_num = 4)"""
    name = "v2_zoe_4k_r32"
    model, ora, w = _pair(name, max_batch=41, n_streams=3)
    hr = rand_image(7, 1, *w["raw"])
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
    ref, rlog = ora(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=ora.resizer(hr), image_hr=hr)
    hr_d = hr.to(DEV)
    got, log = model(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=model.resizer(hr_d), image_hr=hr_d)
    assert tuple(got.shape) == tuple(ref.shape) == (1, 1, 1536, 2048)
    ar, mx = absrel(got, ref)
    ar_c, mx_c = absrel(log["coarse_prediction"], rlog["coarse_prediction"])
    off = rel_l2(got - torch.nn.functional.interpolate(log["coarse_prediction"].cpu(), (1536, 2048), mode="bilinear"),
                 ref - torch.nn.functional.interpolate(rlog["coarse_prediction"], (1536, 2048), mode="bilinear"))
    print(f"{name} whole 4K m1 frame bf16x3 vs oracle: AbsRel {ar:.3e} max|d| {mx:.3e} (depth {float(ref.min()):.2f}..{float(ref.max()):.2f}); "
          f"coarse AbsRel {ar_c:.3e}; (frame - upsampled coarse) relative L2 {off:.3e}")
    assert ar < ABSREL_TOL and ar_c < ABSREL_TOL, (ar, mx)
    assert off <= REL_L2_TOL, off


def test_headline_whole_4k_r32_frame_vs_oracle():
    """(e) the metric's own configuration: one WHOLE 4K frame in cai-mode **r32** -- 81 tiles: the grid pass, the three half-offset
    passes, 32 random tiles (nearest-upsampled 540 x 960 predictions, raw-resolution resize of the running average / count maps,
    ``+1e-3`` mask; patchrefinerplus.py:499-520, baseline_pretrain.py:149-231, utils.py:38-43) -- through the product's frame driver
    at the bench's batching against the oracle's frame driver on the same ``random`` draws; ~2.5 min of host time for the oracle"""
    name = "v2_zoe_4k_r32"
    model, ora, w = _pair(name, max_batch=41, n_streams=3)
    hr = rand_image(11, 1, *w["raw"])
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
    torch.set_num_threads(min(32, torch.get_num_threads()))
    random.seed(621)
    ref, rlog = ora(mode="infer", cai_mode="r32", process_num=4, tile_cfg=tc, image_lr=ora.resizer(hr), image_hr=hr)
    hr_d = hr.to(DEV)
    up = lambda c: torch.nn.functional.interpolate(c, (2160, 3840), mode="bilinear")  # noqa: E731
    for arith in ("bf16x3", "f16f6"):  # (one oracle frame, both arithmetics of the product: the same tolerances)
        if arith != "bf16x3":
            del model
            torch.cuda.empty_cache()
            model, _, _ = _pair(name, prec=arith, max_batch=41, n_streams=3)
        random.seed(621)
        got, log = model(mode="infer", cai_mode="r32", process_num=4, tile_cfg=tc, image_lr=model.resizer(hr_d), image_hr=hr_d)
        assert sum(len(p["raw"]) for p in model.last_plan) == 81
        assert tuple(got.shape) == tuple(ref.shape) == (1, 1, 2160, 3840)
        ar, mx = absrel(got, ref)
        ar_c, _ = absrel(log["coarse_prediction"], rlog["coarse_prediction"])
        off = rel_l2(got - up(log["coarse_prediction"].cpu()), ref - up(rlog["coarse_prediction"]))
        print(f"{name} whole 4K r32 frame (81 tiles) {arith} vs oracle: AbsRel {ar:.3e} max|d| {mx:.3e} (depth {float(ref.min()):.2f}..{float(ref.max()):.2f}); "
              f"coarse AbsRel {ar_c:.3e}; (frame - upsampled coarse) relative L2 {off:.3e}; margins {ABSREL_TOL / ar:.1f}x / {REL_L2_TOL / off:.1f}x")
        assert ar < ABSREL_TOL and ar_c < ABSREL_TOL, (arith, ar, mx)
        assert off <= REL_L2_TOL, (arith, off)


def test_headline_4k_r32_bench_batching_properties_and_shards():
    """(b) the 4K r32 frame exactly as bench.py runs it (41 tiles per batch, 3 streams, next frame's coarse forward prefetched
    beside the tiles): 81 tiles, finite, inside [0, max_depth], bit-identical run to run, with / without the prefetch, at another
    batching, and when the tiles are computed as 8 rank shards and exchanged as RCCL would deliver them"""
    name = "v2_zoe_4k_r32"
    model, _, w = _pair(name, max_batch=41, n_streams=3)
    frames = []
    for seed in (3, 4):
        hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(seed)).to(DEV)
        frames.append((hr, model.resizer(hr)))
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])

    def run(i=0, nxt=None, **kw):
        random.seed(621)
        return model(mode="infer", cai_mode="r32", process_num=4, tile_cfg=tc, image_lr=frames[i][1], image_hr=frames[i][0],
                     next_image_lr=nxt, **kw)[0]

    a = run(0)
    assert sum(len(p["raw"]) for p in model.last_plan) == 81 == w["patches"]
    assert tuple(a.shape) == (1, 1, 2160, 3840) and a.device.type == "cpu" and bool(torch.isfinite(a).all())
    assert float(a.min()) >= 0.0 and float(a.max()) <= float(model.max_depth) * 1.0001
    assert float(a.max()) / max(float(a.min()), 1e-3) >= 5.0      # the widened heads reach the blended map
    b0 = run(0, nxt=frames[1][1])       # frame 1's coarse forward rides beside frame 0's tiles ...
    b1 = run(1)                         # ... and is picked up here
    assert torch.equal(a, b0)
    model.max_batch, model.n_streams = 14, 2
    assert torch.equal(a, run(0))
    assert torch.equal(b1, run(1))      # the prefetched coarse pyramid == the one computed inline
    model.max_batch, model.n_streams = 41, 3
    from conftest import ShardEmulation
    emu = ShardEmulation(model, 8)
    emu.record()
    for r in range(8):
        assert run(0, shard=(r, 8), gather_dst=0) is None
    groups = model.last_shard_layout
    # 8 ranks: 49 fixed tiles are fewer than 8 per rank -> one gather group (one 10 / 11-tile batch per rank instead of 6 + 4: SHARD_MERGE_BELOW)
    assert [g["n"] for g in groups] == [81] and [g["per"] for g in groups] == [11]
    assert groups[0]["share"][0] < max(groups[0]["share"])       # the blending rank computes fewer tiles of the last group
    emu.deliver()
    assert torch.equal(a, run(0, shard=(0, 8), gather_dst=0))
    # ... with the next frame's coarse forward prefetched beside the sharded tiles, too
    emu.restore()
    emu = ShardEmulation(model, 2)
    emu.record()
    for r in range(2):
        assert run(0, shard=(r, 2), gather_dst=0) is None
    emu.deliver()
    assert torch.equal(a, run(0, nxt=frames[1][1], shard=(0, 2), gather_dst=0))
    emu.restore()
    assert torch.equal(b1, run(1))
