"""Do the stages of the restructured GatedConvUnit (coarse taps) give bit-identical results when other kernels -- the next frame's
coarse-half GEMM + knot tables -- run beside them on another stream?   python tools/probes/taps_concurrency.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchrefinerv2_amd import ops as P, lib as L
PR = L.PREC_NAMES["bf16x3"]
g = torch.Generator().manual_seed(1)
F_, k = 256, int(os.environ.get("K", 41))
h, w = (int(v) for v in os.environ.get("HW", "192,256").split(","))
dev = "cuda"
x = P.Feat(torch.randn(k, h, w, F_, generator=g).to(dev))
res = P.Feat(torch.randn(k, h, w, F_, generator=g).to(dev))
coarse = P.Feat(torch.randn(1, h, w, F_, generator=g).to(dev))
coarse2 = P.Feat(torch.randn(1, h, w, F_, generator=g).to(dev))
wc, bc = torch.randn(F_, F_, 3, 3, generator=g) / (3 * F_ ** 0.5), torch.randn(F_, generator=g) * 0.1
wf, bf = torch.randn(F_, 2 * F_, 3, 3, generator=g) / (3 * (2 * F_) ** 0.5), torch.randn(F_, generator=g) * 0.1
w3, gb = torch.randn(F_, F_, 1, 1, generator=g) / 16, (torch.randn(F_, generator=g) * 0.1).to(dev)
ln = ((torch.rand(F_, generator=g) + 0.5).to(dev), (torch.randn(F_, generator=g) * 0.1).to(dev))
cw_c = P.pack_conv(wc.to(dev), bc.to(dev), pad=1, prec=PR)
cw_a = P.pack_conv(wf[:, :F_].contiguous().to(dev), bf.to(dev), pad=1, prec=PR)
tw2 = torch.cat([P.coarse_tap_weight(wf[:, F_:]), P.coarse_tap_weight(wf[:, F_:].flip(0))], 0)
cw_t = P.pack_conv(tw2.to(dev), None, prec=PR)
gw = P.pack_gate(w3.to(dev))
rng = torch.Generator().manual_seed(5)
org = torch.rand(k, 2, generator=rng) * torch.tensor([3 * w * 1.0, 3 * h * 1.0])
boxes = torch.cat([org, org + torch.tensor([w * 1.0, h * 1.0])], 1).to(dev)


def frame_prep(c):
    G = P.conv2d(c, cw_t)
    return G, [P.CoarseTaps(G.slice(i * 9 * F_, 9 * F_), F_, (0.25, 0.25)) for i in range(2)]


def unit(taps):
    out = P.Feat(torch.empty((k, h, w, F_), device=dev), x2=True)
    P.conv2d(x, cw_c, out, relu_in=True, res=x)
    pre = taps.gather(boxes, 0.25, h, w)
    y = P.conv3x3_ln_gate(out, cw_a, ln, gw, gb, act=P.ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_)
    return out, pre, y


G, taps = frame_prep(coarse)
torch.cuda.synchronize()
ref = [t.buf.clone() for t in unit(taps[0])]
refG, refV = G.buf.clone(), [t.v.buf.clone() for t in taps]
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
names = ["out (c256 conv, X2)", "pre (tap gather)", "y (gate kernel + pre)"]
bad = {n: 0 for n in names + ["G other frame", "V other frame"]}
s3 = torch.cuda.Stream()
for it in range(8):
    with torch.cuda.stream(s1):
        got = unit(taps[0])
    with torch.cuda.stream(s3):  # a second batch of tiles beside the first (the frame runs two)
        got_b = unit(taps[0])
    with torch.cuda.stream(s2):
        G2, taps2 = frame_prep(coarse if it % 2 else coarse2)  # the "next frame" prepares beside the tiles
    torch.cuda.synchronize()
    for n, a, b in zip(names, got_b, ref):
        if not torch.equal(a.buf, b):
            bad[n] += 1
            d = (a.buf.view(torch.int32) != b.view(torch.int32))
            print(f"iter {it} (second batch) {n}: {int(d.sum())} words differ")
    for n, a, b in zip(names, got, ref):
        if not torch.equal(a.buf, b):
            bad[n] += 1
            d = (a.buf.view(torch.int32) != b.view(torch.int32))
            print(f"iter {it} {n}: {int(d.sum())} words differ; images {d.flatten(1).any(1).nonzero().flatten().tolist()[:8]}")
    if it % 2:
        if not torch.equal(G2.buf, refG):
            bad["G other frame"] += 1
        if not all(torch.equal(t.v.buf, r) for t, r in zip(taps2, refV)):
            bad["V other frame"] += 1
print("mismatching iterations:", bad)
