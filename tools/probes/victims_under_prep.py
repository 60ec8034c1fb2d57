"""Which kernel family gives different bits when the per-frame coarse-tap preparation (a 4608-column 1x1 GEMM + the knot-table kernel,
256 VGPRs) runs beside it on another stream?   python tools/probes/victims_under_prep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchrefinerv2_amd import ops as P, lib as L
PR = L.PREC_NAMES["bf16x3"]
dev = "cuda"
torch.manual_seed(0)


def conv_layer(n, h, w, cin, cout, k=3, **kw):
    x = P.Feat.alloc(n, h, w, cin, dev); x.buf.normal_()
    if x.ld != cin:
        x.buf[..., cin:] = 0
    cw = P.pack_conv(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5, torch.randn(cout, device=dev), prec=PR)
    ln = ((torch.rand(cout, device=dev) + 0.5), torch.randn(cout, device=dev) * 0.1) if kw.pop("ln", False) else None
    return lambda: P.conv2d(x, cw, ln=ln, **kw).buf, f"conv {cin}->{cout} k{k} {n}x{h}x{w} {'ln' if ln else ''} {kw}"


victims = [
    conv_layer(20, 384, 512, 128, 32, act=P.ACT_RELU),
    conv_layer(20, 384, 512, 98, 32, act=P.ACT_GELU),
    conv_layer(20, 384, 512, 34, 32, act=P.ACT_GELU, ln=True),
    conv_layer(20, 192, 256, 512, 64, act=P.ACT_GELU, ln=True),
    conv_layer(20, 192, 256, 194, 64, act=P.ACT_GELU),
    conv_layer(20, 384, 512, 98, 98, act=P.ACT_GELU),
    conv_layer(20, 192, 256, 194, 194, act=P.ACT_GELU),
    conv_layer(20, 192, 256, 256, 256, relu_in=True),
    conv_layer(20, 96, 128, 512, 256, act=P.ACT_GELU, ln=True),
    conv_layer(20, 192, 256, 32, 256),
    conv_layer(20, 192, 256, 256, 256, k=1),
    conv_layer(20, 384, 512, 32, 32, k=1),
    conv_layer(20, 48, 64, 642, 642, act=P.ACT_GELU),
    conv_layer(20, 24, 32, 770, 770, act=P.ACT_GELU),
    conv_layer(20, 12, 16, 960, 256),
]
xd = P.Feat.alloc(20, 192, 256, 64, dev); xd.buf.normal_()
wdw = torch.randn(9, 64, device=dev)
victims.append((lambda: P.dwconv2d(xd, wdw, None, 3, 1, relu=True).buf, "dwconv 3x3 64ch"))
xu = P.Feat.alloc(20, 96, 128, 256, dev); xu.buf.normal_()
victims.append((lambda: P.upsample_bilinear(xu, 192, 256).buf, "upsample x2 256ch"))
xc = P.Feat.alloc(20, 384, 512, 32, dev); xc.buf.normal_()
wc1 = torch.randn(1, 32, 3, 3, device=dev)
victims.append((lambda: P.conv2d_cout1(xc, wc1, None, 3), "conv_cout1 3x3"))

# round 5: the fused 32-channel chains and the 5x5 composite (both compiled WITH packed fp32 math)
r = lambda *sh: torch.randn(*sh)  # noqa: E731
c_w1, c_w2, c_w34 = r(32, 32, 3, 3) / 17, r(32, 32, 3, 3) / 17, r(32, 34, 3, 3) / 17.5
cw_c2f = dict(w1=P.pack_chain32(c_w1, 0, dev), w2=P.pack_chain32(c_w2, 1, dev), wg=P.pack_chain32(r(32, 32) / 5.6, 1, dev), wo=P.pack_chain32(r(32, 32) / 5.6, 1, dev),
              consts=P.chain32_consts(dev, b1=r(32) * .1, ln1w=1 + .2 * r(32), ln1b=.1 * r(32), b2=r(32) * .1, bo=r(32) * .1, w3=1 + .3 * r(32)), b3=0.1)
cw_enc = dict(w1=P.pack_chain32(c_w1, 0, dev), w2=P.pack_chain32(c_w34, 1, dev), wt=P.pack_chain32(c_w34, 2, dev),
              consts=P.chain32_consts(dev, b1=r(32) * .1, ln1w=1 + .2 * r(32), ln1b=.1 * r(32), b2=r(32) * .1, ln2w=1 + .2 * r(32), ln2b=.1 * r(32)))
cx = P.Feat(torch.randn(20, 384, 512, 32, device=dev).relu_())
cpre = P.Feat(torch.randn(20, 384, 512, 32, device=dev) * 0.5)
cp1, cp2 = torch.rand(20, 1, 384, 512, device=dev) * 10, torch.rand(20, 1, 384, 512, device=dev) * 10
victims.append((lambda: torch.cat([t.reshape(-1) for t in (lambda o, d: (o.buf, d))(*P.chain32_c2f(cx, cw_c2f, cpre))]), "chain32_c2f 20x384x512"))
victims.append((lambda: P.chain32_enc(cx, cw_enc, cpre, cp1, cp2).buf, "chain32_enc 20x384x512"))
cw5 = P.compose_upconv5x5(r(128, 256, 3, 3) / 48, r(128) * .1, r(9, 128) * .1, r(32, 128, 3, 3) / 34, r(32) * .1, dev, PR)
u5 = P.Feat(torch.randn(20, 192, 256, 256, device=dev))
victims.append((lambda: P.upconv5x5(u5, 384, 512, cw5, act=P.ACT_RELU).buf, "upconv5x5 256->(128)->32 20x384x512"))

F_ = 256
coarse = P.Feat(torch.randn(1, 192, 256, F_, device=dev))
cw_t = P.pack_conv(torch.randn(18 * F_, F_, device=dev) / 16, None, prec=PR)


# round 5, second half: the fp16 + fp6 kernels (weights streamed L2 -> registers, LDS-DMA halo, counted waits)
rd = lambda *s_: r(*s_).to(dev)  # noqa: E731
x6 = P.Feat(rd(14, 96, 128, 256))
cw6 = P.pack_conv3x3_f6(rd(256, 256, 3, 3) / 48, rd(256) * .1)
victims.append((lambda: P.conv3x3_f6(x6, cw6, relu_in=True, res=x6).buf, "conv3x3_c256_f6 14x96x128"))
o6 = P.Feat(torch.empty(14, 96, 128, 256, device=dev), x2=True)
P.conv3x3_f6(x6, cw6, o6, relu_in=True, res=x6)
g6w, ln6 = P.pack_gate(rd(256, 256, 1, 1) / 16), (1 + .2 * rd(256), .1 * rd(256))
pre6 = P.Feat(rd(14, 96, 128, 256) * .5)
gb6 = torch.full((256,), .1, device=dev)
victims.append((lambda: P.conv3x3_ln_gate_f6(o6, cw6, ln6, g6w, gb6, mul=o6, res=x6, pre=pre6, pre_cin=256).buf, "conv3x3_c256_gate_f6 14x96x128"))

def frame_prep():
    G = P.conv2d(coarse, cw_t)
    return G, [P.CoarseTaps(G.slice(i * 9 * F_, 9 * F_), F_, (0.25, 0.25)) for i in range(2)]


torch.cuda.synchronize()
ref = [fn().clone() for fn, _ in victims]
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad = [0] * len(victims)
for it in range(5):
    for i, (fn, name) in enumerate(victims):
        with torch.cuda.stream(s2):
            keep = [frame_prep() for _ in range(2)]
        with torch.cuda.stream(s1):
            y = fn()
        torch.cuda.synchronize()
        if not torch.equal(y, ref[i]):
            bad[i] += 1
            d = y.view(torch.int32) != ref[i].view(torch.int32)
            print(f"iter {it} {name}: {int(d.sum())} words differ", flush=True)
for (fn, name), b in zip(victims, bad):
    print(f"{b}/5  {name}")
