"""Where does a frame with the next frame's coarse forward riding beside it first differ from the plain frame?  Integer checksums of
every stage of every GatedConvUnit call (input, out, pre, y) recorded on the launching stream, compared between the two runs."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchrefinerv2_amd import models, fusion, ops, weights as W  # noqa: F401
from patchrefinerv2_amd.registry import build_model
from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
name = "v2_zoe_4k_r32"
w = WORKLOADS[name]
model = build_model(model_config(name, prec="bf16x3", max_batch=41, n_streams=3))
model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
frames = []
for seed in (3, 4):
    hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(seed)).cuda()
    frames.append((hr, model.resizer(hr)))
tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
REC = []
KEEP = []


def cks(t):
    b = t.buf if isinstance(t, ops.Feat) else t
    return b.view(torch.int32).sum(dtype=torch.int64)


orig = fusion.BiDirectionalFusion._gated_unit_taps


def patched(u, x, taps, coarse, F_, res=None):
    out = ops.Feat(torch.empty((x.n, x.h, x.w, F_), device=x.device, dtype=torch.float32), x2=F_ == 256)
    ops.conv2d(x, u["conv"], out, relu_in=True, res=x)
    pre = taps.gather(coarse.boxes, coarse.scale, x.h, x.w)
    y = ops.conv3x3_ln_gate(out, u["f0a"], (u["lnw"], u["lnb"]), u["f3g"], u["f3"].bias, act=ops.ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_)
    REC.append((f"{x.n}x{x.h}x{x.w}", cks(x), cks(out), cks(pre), cks(y), cks(taps.v), cks(taps.g.buf), cks(coarse.boxes)))
    if x.h <= 48:
        KEEP.append((f"{x.n}x{x.h}x{x.w}", pre.buf.clone(), coarse.boxes.clone()))
    return y


fusion.BiDirectionalFusion._gated_unit_taps = staticmethod(patched)


def run(i, nxt=None):
    random.seed(621)
    REC.clear()
    KEEP.clear()
    d = model(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tc, image_lr=frames[i][1], image_hr=frames[i][0], next_image_lr=nxt)[0]
    torch.cuda.synchronize()
    return d, [(r[0],) + tuple(int(v) for v in r[1:]) for r in REC], list(KEEP)


a, ra, ka = run(0)
for trial in range(4):
    b, rb, kb = run(0, nxt=frames[1][1])
    run(1)
    print(f"trial {trial}: frame equal={torch.equal(a, b)}; calls {len(ra)} / {len(rb)}")
    names = ["x", "out", "pre", "y", "V", "G", "boxes"]
    # calls of the two batches interleave on the host in a fixed order: compare position by position
    for i, (p, q) in enumerate(zip(ra, rb)):
        bad = [names[j] for j in range(7) if p[j + 1] != q[j + 1]]
        if bad:
            print(f"   call {i} {p[0]}: differs in {bad}")
    for (n1, p1, b1), (n2, p2, b2) in zip(ka, kb):
        if not torch.equal(p1, p2):
            d = (p1 != p2)
            idx = d.nonzero()
            print(f"   pre {n1}: {int(d.sum())} elements differ; tiles {idx[:, 0].unique().tolist()}; rows {idx[:, 1].min().item()}..{idx[:, 1].max().item()} "
                  f"cols {idx[:, 2].min().item()}..{idx[:, 2].max().item()} ch {idx[:, 3].min().item()}..{idx[:, 3].max().item()}; max|d| {float((p1 - p2).abs().max()):.3e} "
                  f"scale {float(p1.abs().max()):.3e}; boxes equal {torch.equal(b1, b2)}")
            t0, r0, c0 = (int(v) for v in idx[0, :3])
            chs = d[t0, r0, c0].nonzero().flatten().tolist()
            print(f"     tile {t0} px ({r0},{c0}) channels {chs}")
            print("     plain :", [round(float(v), 4) for v in p1[t0, r0, c0, chs[:8]]])
            print("     beside:", [round(float(v), 4) for v in p2[t0, r0, c0, chs[:8]]])
            # is the 'beside' value something that lives elsewhere in the correct buffer (a misplaced / stale row)?
            v = p2[t0, r0, c0, chs[0]]
            hits = (p1 == v).nonzero()
            print("     where the first wrong value occurs in the plain buffer:", hits[:4].tolist())
