"""GPU parity: every C-ABI kernel against the oracle / plain torch fp32 on the CPU, same seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import ops as o_ops  # noqa: E402
from oracle import tiling as o_tiling  # noqa: E402
from oracle.dav2 import attention as o_attention  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def P():
    from patchrefinerv2_amd import ops
    ops.L.load()
    return ops


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def close(got, ref, tol, what=""):
    got = got.detach().cpu()
    err = float((got - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err <= tol * scale, f"{what}: max|d|={err:.3e} (scale {scale:.2f})"


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, opts
    (2, 24, 32, 64, 128, 3, 1, {}),
    (1, 17, 23, 34, 32, 3, 1, {}),                      # +2 concat channels, ragged spatial, BN=64 path
    (2, 12, 16, 98, 98, 3, 1, dict(act="gelu")),        # f2r_agg-like odd channels
    (1, 16, 16, 256, 200, 3, 1, dict(bias=True, relu_in=True, res=True)),
    (1, 20, 28, 96, 256, 1, 1, dict(bias=True)),
    (1, 32, 32, 64, 64, 3, 2, dict(bias=True)),        # DPT resize_layers.3 / MNv4 strided
    (3, 9, 7, 48, 24, 1, 1, dict(bias=True, act="sigmoid", mul=True)),
    (1, 8, 8, 130, 128, 3, 1, dict(res=True, res2=True, gamma=True, bias=True)),
    # LDS-halo 3x3 kernel (W >= 24): ragged tiles, multi-slab, both channel-tile widths, fused epilogues
    (2, 30, 50, 66, 130, 3, 1, dict(bias=True, act="gelu")),
    (1, 8, 32, 32, 32, 3, 1, {}),
    (1, 19, 75, 258, 256, 3, 1, dict(bias=True, relu_in=True, res=True)),
    (3, 7, 24, 98, 40, 3, 1, dict(res=True, res2=True, gamma=True, bias=True)),
    (1, 56, 56, 512, 256, 3, 1, dict(bias=True)),
    (2, 21, 45, 98, 32, 3, 1, dict(bias=True, act="gelu")),   # cout <= 32: 8x1 wave grid variant
    (1, 40, 64, 34, 20, 3, 1, dict(res=True)),
    (2, 64, 64, 80, 4, 1, 1, dict(bias=True)),               # ZoeDepth's 80 -> 4 head conv: the few-outputs form of conv1x1_small_kernel
    (2, 160, 128, 4, 32, 3, 2, dict(bias=True, act="gelu")),  # the refiner stem on the 4-channel crop: conv_few_in_kernel (direct, <= 4 input channels)
    (1, 141, 131, 3, 24, 3, 2, dict(bias=True)),             # ... 3 channels, ragged size, 24 outputs
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(P, case):
    n, h, w, cin, cout, k, stride, o = case
    x = rnd(1, n, cin, h, w)
    wt = rnd(2, cout, cin, k, k) / np.sqrt(cin * k * k)
    bias = rnd(3, cout) if o.get("bias") else None
    act = {"gelu": P.ACT_GELU, "sigmoid": P.ACT_SIGMOID, None: P.ACT_NONE}[o.get("act")]
    xin = F.relu(x) if o.get("relu_in") else x
    ref = F.conv2d(xin, wt, bias, stride=stride, padding=k // 2)
    if o.get("act") == "gelu":
        ref = F.gelu(ref)
    if o.get("act") == "sigmoid":
        ref = torch.sigmoid(ref)
    gamma = rnd(4, cout) if o.get("gamma") else None
    if gamma is not None:
        ref = ref * gamma.view(1, -1, 1, 1)
    mul = rnd(5, *ref.shape) if o.get("mul") else None
    res = rnd(6, *ref.shape) if o.get("res") else None
    res2 = rnd(7, *ref.shape) if o.get("res2") else None
    if mul is not None:
        ref = mul * ref
    if res is not None:
        ref = ref + res
    if res2 is not None:
        ref = ref + res2
    cw = P.pack_conv(wt.to(DEV), bias.to(DEV) if bias is not None else None, stride=stride)
    f = lambda t: P.Feat.from_nchw(t.to(DEV)) if t is not None else None  # noqa: E731
    y = P.conv2d(f(x), cw, relu_in=bool(o.get("relu_in")), act=act, gamma=gamma.to(DEV) if gamma is not None else None,
                 mul=f(mul), res=f(res), res2=f(res2))
    close(y.to_nchw(), ref, 2e-5, f"conv {case}")


GATE_CASES = [  # n, h, w, cin, channels F, gate, residual
    (2, 20, 32, 64, 256, True, True), (1, 8, 16, 512, 256, True, False), (3, 13, 48, 96, 256, False, False), (1, 24, 16, 32, 256, True, True),
    (2, 13, 24, 64, 256, True, True), (1, 9, 37, 32, 256, True, False),   # ragged right edge
    # F = 128 / 32 (the full-resolution output_conv2_fusion block): the 8 x 32-pixel kernels with the gate stage in their epilogue
    (2, 20, 32, 256, 128, True, True), (1, 11, 70, 64, 128, True, False), (1, 33, 40, 32, 128, True, True),   # (70, 40: 32 k + 6 / 8 -> strip tiles)
    (2, 20, 32, 64, 32, True, True), (1, 9, 37, 32, 32, True, False), (3, 8, 64, 96, 32, True, True),
]


def _x2_reconstruct(v: torch.Tensor) -> torch.Tensor:
    """what an X2 element stands for: bf16 hi (RNE) + bf16 lo (RNE of the remainder), as split_bf16 computes them"""
    hi = v.to(torch.bfloat16).float()
    lo = (v - hi).to(torch.bfloat16).float()
    return hi + lo


@pytest.mark.parametrize("case", GATE_CASES)
@pytest.mark.parametrize("prec", ["bf16x3", "f32ref"])
def test_conv3x3_ln_gate_fused_tail(P, case, prec):
    """GatedConvUnit tail (bi_directional_fusion_model.py:44-51,70-80) as one kernel: vs plain torch fp32, and vs the unfused
    kernel sequence conv2d -> LayerNorm -> conv2d(1x1, sigmoid, mul, res) in the same arithmetic mode"""
    n, h, w, cin, C_, gate, with_res = case
    x = rnd(1, n, cin, h, w)
    w0, b0 = rnd(2, C_, cin, 3, 3) / np.sqrt(9 * cin), rnd(3, C_) * 0.1
    lnw, lnb = 1 + 0.2 * rnd(4, C_), 0.1 * rnd(5, C_)
    w3, b3 = rnd(6, C_, C_, 1, 1) / np.sqrt(C_), rnd(7, C_) * 0.1
    mul, res = rnd(8, n, C_, h, w), rnd(9, n, C_, h, w)
    y = F.conv2d(x, w0, b0, padding=1)
    u = y.mean(1, keepdim=True)
    sd = (y - u).pow(2).mean(1, keepdim=True)
    fused = F.relu((y - u) / torch.sqrt(sd + 1e-6) * lnw.view(1, -1, 1, 1) + lnb.view(1, -1, 1, 1))
    ref = mul * torch.sigmoid(F.conv2d(fused, w3, b3)) + (res if with_res else 0) if gate else fused
    PR = P.L.PREC_NAMES["bf16x3"]
    f = lambda t: P.Feat.from_nchw(t.to(DEV))  # noqa: E731
    cw0, cw3 = P.pack_conv(w0.to(DEV), b0.to(DEV), pad=1, prec=PR), P.pack_conv(w3.to(DEV), b3.to(DEV), prec=PR)
    xf, ln = f(x), (lnw.to(DEV), lnb.to(DEV))
    assert P.conv3x3_ln_gate_supported(xf, cw0)
    got = P.conv3x3_ln_gate(xf, cw0, ln, P.pack_gate(w3.to(DEV)) if gate else None, b3.to(DEV) if gate else None, act=P.ACT_RELU,
                            mul=f(mul) if gate else None, res=f(res) if gate and with_res else None)
    kname = P.L.load().prv2_last_kernel().decode()
    assert kname.startswith({256: "conv3x3_c256_gate_kernel<256" if gate else "conv3x3_c256_kernel<256", 128: "conv3x3_halo16_gate_kernel<128",
                             32: "conv3x3_halo16_gate_kernel<32"}[C_]), kname
    if prec == "f32ref":
        close(got.to_nchw(), ref, 2e-5, f"fused tail {case}")
        return
    t = P.conv2d(xf, cw0, act=P.ACT_RELU, ln=ln)
    if gate:  # (the 256-channel gate kernel multiplies by the bf16 pair hi + lo of ``mul`` -- the operand its conv reads, in either format)
        t = P.conv2d(t, cw3, act=P.ACT_SIGMOID, mul=f(_x2_reconstruct(mul) if C_ == 256 else mul), res=f(res) if with_res else None)
    close(got.to_nchw(), t.to_nchw().cpu(), 2e-6, f"fused vs unfused {case}")


@pytest.mark.parametrize("case", [(2, 20, 32, 64, dict(bias=True, relu_in=True, res=True)), (1, 9, 48, 96, dict(act="gelu")), (2, 11, 41, 64, dict(bias=True, res=True)),
                                  (3, 8, 16, 512, dict(bias=True, ln=True, act="gelu")), (1, 24, 64, 32, dict(bias=True, ln=True, res=True))])
def test_conv2d_256_channels_take_the_wide_tile_kernel(P, case):
    """prv2_conv2d sends 3x3 convs with 256 output channels (width >= 16, Cin % 32 == 0, bf16 modes) to the 8 x 16 x 256 kernel
    of conv3x3_gate.hip -- also with a fused LayerNorm, which the 128-column kernel cannot do at this width"""
    n, h, w, cin, o = case
    PR = P.L.PREC_NAMES["bf16x3"]
    x = rnd(1, n, cin, h, w)
    wt, bias = rnd(2, 256, cin, 3, 3) / np.sqrt(9 * cin), (rnd(3, 256) * 0.1 if o.get("bias") else None)
    lnw, lnb = 1 + 0.2 * rnd(4, 256), 0.1 * rnd(5, 256)
    res = rnd(6, n, 256, h, w) if o.get("res") else None
    ref = F.conv2d(F.relu(x) if o.get("relu_in") else x, wt, bias, padding=1)
    if o.get("ln"):
        u = ref.mean(1, keepdim=True)
        ref = (ref - u) / torch.sqrt((ref - u).pow(2).mean(1, keepdim=True) + 1e-6) * lnw.view(1, -1, 1, 1) + lnb.view(1, -1, 1, 1)
    if o.get("act") == "gelu":
        ref = F.gelu(ref)
    if res is not None:
        ref = ref + res
    cw = P.pack_conv(wt.to(DEV), bias.to(DEV) if bias is not None else None, pad=1, prec=PR)
    f = lambda t: P.Feat.from_nchw(t.to(DEV)) if t is not None else None  # noqa: E731
    kw = dict(relu_in=bool(o.get("relu_in")), act=P.ACT_GELU if o.get("act") else P.ACT_NONE, res=f(res))
    y = P.conv2d(f(x), cw, ln=(lnw.to(DEV), lnb.to(DEV)) if o.get("ln") else None, **kw)
    assert P.L.load().prv2_last_kernel().decode() == "conv3x3_c256_kernel<256,bf16x3>"
    close(y.to_nchw(), ref, 2e-5, f"c256 {case}")
    if not o.get("ln"):
        g = P.conv2d(f(x), cw, force_generic=True, **kw)
        close(y.to_nchw(), g.to_nchw().cpu(), 2e-6, "c256 vs generic")


def test_conv3x3_ln_gate_rejects_what_it_does_not_cover(P):
    PR = P.L.PREC_NAMES["bf16x3"]
    cw = P.pack_conv(rnd(1, 256, 64, 3, 3).to(DEV), None, pad=1, prec=PR)
    assert not P.conv3x3_ln_gate_supported(P.Feat.from_nchw(rnd(2, 1, 64, 8, 12).to(DEV)), cw)      # narrower than a tile
    cw64 = P.pack_conv(rnd(1, 64, 64, 3, 3).to(DEV), None, pad=1, prec=PR)
    assert not P.conv3x3_ln_gate_supported(P.Feat.from_nchw(rnd(2, 1, 64, 8, 32).to(DEV)), cw64)    # cout not 32 / 128 / 256
    cw128 = P.pack_conv(rnd(1, 128, 64, 3, 3).to(DEV), None, pad=1, prec=PR)
    assert not P.conv3x3_ln_gate_supported(P.Feat.from_nchw(rnd(2, 1, 64, 8, 16).to(DEV)), cw128)   # 128 channels: width >= 24
    cwf = P.pack_conv(rnd(1, 256, 64, 3, 3).to(DEV), None, pad=1)
    assert not P.conv3x3_ln_gate_supported(P.Feat.from_nchw(rnd(2, 1, 64, 8, 32).to(DEV)), cwf)     # f32 mode
    x = P.Feat.from_nchw(rnd(2, 1, 64, 8, 12).to(DEV))
    ln = (torch.ones(256, device=DEV), torch.zeros(256, device=DEV))
    with pytest.raises(RuntimeError, match="width"):
        P.conv3x3_ln_gate(x, cw, ln, None, None)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_out_conv_folds_into_the_following_3x3(P, prec):
    """GatedFusionBlock.out_conv (1x1 + bias) -> bilinear x2 -> output_conv1 (3x3 + bias) (bi_directional_fusion_model.py:139-142,201) ==
    ONE 3x3 conv with composed weights on the upsampled input + the border correction of the folded bias (prv2_conv_border_bias)"""
    x = rnd(1, 2, 64, 12, 16)
    woc, boc = rnd(2, 64, 64, 1, 1) / 8, rnd(3, 64)
    w1, b1 = rnd(4, 32, 64, 3, 3) / 24, rnd(5, 32)
    ref = F.conv2d(F.interpolate(F.conv2d(x, woc, boc), size=(24, 32), mode="bilinear", align_corners=True), w1, b1, padding=1)
    wf = torch.einsum("omyx,mi->oiyx", w1.double(), woc.double()[:, :, 0, 0]).float()
    bt = torch.einsum("omyx,m->yxo", w1.double(), boc.double())
    PR = P.L.PREC_NAMES[prec]
    cw = P.pack_conv(wf.to(DEV), (b1.double() + bt.sum((0, 1))).float().to(DEV), pad=1, prec=PR)
    up = P.upsample_bilinear(P.Feat.from_nchw(x.to(DEV)), 24, 32)
    out = P.conv2d(up, cw)
    wrong = float((out.to_nchw().cpu() - ref).abs().max())
    P.conv_border_bias(out, bt.reshape(9, -1).float().contiguous().to(DEV))
    close(out.to_nchw(), ref, 1e-5 if prec == "f32" else 3e-5, "folded out_conv")
    assert wrong > 1e-2  # (the border pixels do need the correction)


def test_conv_into_concat_slice(P):
    """producer writes into a channel slice of a wider buffer == torch.cat"""
    x = rnd(1, 1, 32, 10, 12)
    w1, w2 = rnd(2, 64, 32, 3, 3) / 17, rnd(3, 34, 32, 3, 3) / 17
    ref = torch.cat([F.conv2d(x, w1, padding=1), F.conv2d(x, w2, padding=1)], dim=1)
    xf = P.Feat.from_nchw(x.to(DEV))
    cat = P.Feat.alloc(1, 10, 12, 98, DEV)
    P.conv2d(xf, P.pack_conv(w1.to(DEV)), cat.slice(0, 64))
    P.conv2d(xf, P.pack_conv(w2.to(DEV)), cat.slice(64, 34))
    close(cat.to_nchw(), ref, 2e-5)
    # and is consumable as a 98-channel input
    w3 = rnd(4, 40, 98, 3, 3) / 30
    close(P.conv2d(cat, P.pack_conv(w3.to(DEV))).to_nchw(), F.conv2d(ref, w3, padding=1), 2e-5)


@pytest.mark.parametrize("k", [2, 4])
def test_conv_transpose(P, k):
    x = rnd(1, 2, 48, 6, 9)
    wt = rnd(2, 48, 40, k, k) / 7
    b = rnd(3, 40)
    ref = F.conv_transpose2d(x, wt, b, stride=k)
    cw = P.pack_conv(wt.to(DEV), b.to(DEV), convt_k=k)
    close(P.conv2d(P.Feat.from_nchw(x.to(DEV)), cw).to_nchw(), ref, 2e-5)


def test_linear_vit_shapes(P):
    x = rnd(1, 2 * 1025, 384)
    w, b, g = rnd(2, 1536, 384) / 20, rnd(3, 1536), None
    ref = F.gelu(F.linear(x, w, b))
    y = P.linear(x.to(DEV), P.pack_conv(w.to(DEV), b.to(DEV)), act=P.ACT_GELU)
    close(y, ref, 2e-5)
    w2, b2, g = rnd(4, 384, 1536) / 40, rnd(5, 384), rnd(6, 384)
    ref2 = x + F.linear(ref, w2, b2) * g
    y2 = P.linear(y, P.pack_conv(w2.to(DEV), b2.to(DEV)), gamma=g.to(DEV), res=x.to(DEV))
    close(y2, ref2, 2e-5)


def test_conv_cout1_and_dw(P):
    x = rnd(1, 2, 32, 20, 24)
    w = rnd(2, 1, 32, 3, 3) / 17
    base = rnd(3, 2, 1, 20, 24) + 1
    ref = torch.clamp(base + F.conv2d(x, w, padding=1), min=0)
    xf = P.Feat.from_nchw(x.to(DEV))
    close(P.conv2d_cout1(xf, w.to(DEV), None, 3, res=base.to(DEV), clamp0=True), ref, 1e-5)
    w1, b1 = rnd(4, 1, 32, 1, 1) / 6, rnd(5, 1)
    close(P.conv2d_cout1(xf, w1.to(DEV), b1.to(DEV), 1, act=P.ACT_SIGMOID, scale=80.0), torch.sigmoid(F.conv2d(x, w1, b1)) * 80, 1e-5)
    for k, s in ((3, 1), (3, 2), (5, 1), (5, 2), (7, 1), (7, 2)):
        wd, bd = rnd(6, 32, 1, k, k) / k, rnd(7, 32)
        refd = F.relu(F.conv2d(x, wd, bd, stride=s, padding=k // 2, groups=32))
        wt = wd.view(32, k * k).t().contiguous().to(DEV)
        close(P.dwconv2d(xf, wt, bd.to(DEV), k, s, True).to_nchw(), refd, 1e-5, f"dw k{k}s{s}")
    # stride-1 strip kernel (8 pixels per thread): ragged widths (W % 8 != 0, W < 8 -> naive kernel), no ReLU, no bias
    for k, (h, w_), c in ((7, (9, 31), 192), (7, (5, 8), 48), (5, (6, 13), 64), (3, (4, 7), 32), (7, (12, 15), 1536)):
        xs = rnd(8, 2, c, h, w_)
        wd = rnd(9, c, 1, k, k) / k
        refd = F.conv2d(xs, wd, None, padding=k // 2, groups=c)
        wt = wd.view(c, k * k).t().contiguous().to(DEV)
        close(P.dwconv2d(P.Feat.from_nchw(xs.to(DEV)), wt, None, k, 1, False).to_nchw(), refd, 1e-5, f"dw strip k{k} {h}x{w_}x{c}")


@pytest.mark.parametrize("n,h,w,c", [(3, 196, 259, 24), (2, 7, 5, 3072), (14, 49, 65, 384), (1, 1, 1, 8), (2, 98, 130, 240), (1, 33, 600, 12)])
def test_squeeze_excite_pieces(P, n, h, w, c):
    """global mean per (image, channel): fixed-order two-stage reduction (bit-reproducible), and the in-place channel gate"""
    x = rnd(21, n, c, h, w) + 0.5
    xf = P.Feat.from_nchw(x.to(DEV))
    m1 = P.global_avgpool(xf)
    m2 = P.global_avgpool(xf)
    assert torch.equal(m1, m2)
    ref = x.double().mean((2, 3)).float()
    assert float((m1.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    g = torch.rand(n, c, generator=torch.Generator().manual_seed(22))
    P.channel_scale_(xf, g.to(DEV).contiguous())
    assert torch.equal(xf.to_nchw().cpu(), x * g.view(n, c, 1, 1))
    cse = max(1, c // 24)
    w1, b1, w2, b2 = rnd(23, cse, c) / np.sqrt(c), rnd(24, cse), rnd(25, c, cse) / np.sqrt(cse), rnd(26, c)
    gate = P.se_gate(m1, w1.to(DEV), b1.to(DEV), w2.t().contiguous().to(DEV), b2.to(DEV)).cpu()
    ref_g = torch.sigmoid(F.linear(F.silu(F.linear(ref.double(), w1.double(), b1.double())), w2.double(), b2.double())).float()
    assert float((gate - ref_g).abs().max()) <= 2e-6


@pytest.mark.parametrize("c", [32, 98, 256, 384, 1024])
def test_layernorm(P, c):
    x = rnd(1, 3, c, 7, 5) * 3 + 1
    w, b = rnd(2, c), rnd(3, c)
    from oracle.fusion import ln_cf
    ref = F.gelu(ln_cf(x, w, b))
    xf = P.Feat.from_nchw(x.to(DEV))
    out = P.layernorm_feat(xf, w.to(DEV), b.to(DEV), 1e-6, P.ACT_GELU)
    close(out.to_nchw(), ref, 1e-5, f"ln c={c}")


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("shape", [(1, 1025, 6), (2, 197, 2), (1, 64, 1), (3, 130, 1)])
def test_attention(P, shape, prec):
    from patchrefinerv2_amd import lib as L
    b, n, heads = shape
    D = heads * 64
    x = rnd(1, b, n, D)
    sd = {"qkv.weight": rnd(2, 3 * D, D) / np.sqrt(D), "qkv.bias": rnd(3, 3 * D) * 0.1,
          "proj.weight": torch.eye(D), "proj.bias": torch.zeros(D)}
    ref = o_attention(sd, "", x, heads)
    qkv = F.linear(x, sd["qkv.weight"], sd["qkv.bias"]).reshape(b * n, 3 * D)
    out = P.attention(qkv.to(DEV).contiguous(), b, n, heads, L.PREC_NAMES[prec])
    close(out.view(b, n, D), ref, 1e-5 if prec == "f32" else 4e-5, f"attention {shape} {prec}")


def test_attention_spiky(P):
    """rows whose max jumps tile to tile (online-softmax rescale path) + large logits"""
    b, n, heads = 1, 300, 1
    q = rnd(1, n, 64)
    k = rnd(2, n, 64)
    k[70] = q[5] * 6
    k[200] = q[5] * 9
    k[299] = q[17] * 12
    v = rnd(3, n, 64)
    qkv = torch.stack([q, k, v], dim=1).reshape(n, 192)
    att = ((q * 0.125) @ k.t()).softmax(-1) @ v
    out = P.attention(qkv.to(DEV).contiguous(), b, n, heads)
    close(out, att, 1e-5)
    from patchrefinerv2_amd import lib as L
    out3 = P.attention(qkv.to(DEV).contiguous(), b, n, heads, L.PREC_BF16X3)
    close(out3, att, 1e-4)  # logits up to ~100: the 2^-17 product error is amplified by exp()


def test_patchify_tokens(P):
    img = rnd(1, 2, 3, 28, 42)
    w, bias = rnd(2, 64, 3, 14, 14) / 24, rnd(3, 64)
    cls, pos = rnd(4, 1, 1, 64), rnd(5, 1, 7, 64)
    emb = F.conv2d(img, w, bias, stride=14).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.expand(2, -1, -1), emb], 1) + pos
    f = P.Feat.from_nchw(img.to(DEV))
    rows = P.patchify(f, 14, 592)
    wl = w.permute(0, 2, 3, 1).reshape(64, 588)
    e = P.linear(rows, P.pack_conv(wl.to(DEV), bias.to(DEV)))
    tok = P.assemble_tokens(e, cls.to(DEV).contiguous(), pos.to(DEV).contiguous(), 2, 6, 64)
    close(tok, ref, 1e-5)


def test_crop_resize(P):
    img = torch.rand(3, 216, 384, generator=torch.Generator().manual_seed(1))
    tiles = torch.tensor([[0, 0], [108, 192], [54, 96], [107, 191]], dtype=torch.int32)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    out = P.Feat.alloc(4, 56, 84, 4, DEV)
    P.crop_resize(img.to(DEV), tiles.to(DEV), 108, 192, 56, 84, mean, std, out)
    got = out.to_nchw()[:, :3]
    for i, (h0, w0) in enumerate(tiles.tolist()):
        ref = o_ops.resize_da(img[None, :, h0:h0 + 108, w0:w0 + 192], 84, 56)
        ref = (ref - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
        close(got[i:i + 1], ref, 2e-6, f"crop {i}")
    # identity size == plain layout change
    out2 = P.Feat.alloc(1, 216, 384, 3, DEV, pad_to=1)
    P.crop_resize(img.to(DEV), torch.zeros(1, 2, dtype=torch.int32, device=DEV), 216, 384, 216, 384, None, None, out2)
    assert torch.equal(out2.to_nchw().cpu()[0], img)


def test_roi_align(P):
    feat = rnd(1, 1, 32, 24, 32)
    depth = rnd(2, 1, 1, 56, 84)
    bboxs = torch.tensor([[0, 0, 192, 108], [192, 108, 384, 216], [96, 54, 288, 162], [191, 107, 383, 215]])
    bf = o_tiling.bboxs_to_feat(bboxs, (216, 384), (56, 84))
    ref = o_ops.roi_align(feat.repeat(4, 1, 1, 1), bf, (24, 32), 24 / 56, aligned=True)
    got = P.roi_align(P.Feat.from_nchw(feat.to(DEV)), bf[:, 1:].contiguous().to(DEV), 24 / 56, 24, 32)
    close(got.to_nchw(), ref, 1e-5, "roi feat")
    refd = o_ops.roi_align(depth.repeat(4, 1, 1, 1), bf, (56, 84), 1.0, aligned=True)
    gotd = P.roi_align(P.Feat.from_nchw(depth.to(DEV), pad_to=1), bf[:, 1:].contiguous().to(DEV), 1.0, 56, 84)
    close(gotd.to_nchw(), refd, 1e-5, "roi depth")
    # down-sampling ROI (adaptive sampling grid > 1)
    big = torch.tensor([[0, 0.0, 0.0, 84.0, 56.0]])
    ref2 = o_ops.roi_align(feat, big, (5, 7), 24 / 56, aligned=True)
    got2 = P.roi_align(P.Feat.from_nchw(feat.to(DEV)), big[:, 1:].contiguous().to(DEV), 24 / 56, 5, 7)
    close(got2.to_nchw(), ref2, 1e-5, "roi grid")
    # sampling grid > 1 at >= 16 output rows (the row-cached kernel's general branch), and a box hanging over the map's edge
    boxes3 = torch.tensor([[0, 0.0, 0.0, 84.0, 56.0], [0, 40.0, 20.0, 100.0, 70.0]])
    ref3 = o_ops.roi_align(feat.repeat(2, 1, 1, 1), torch.cat([torch.arange(2.0)[:, None], boxes3[:, 1:]], 1), (16, 20), 24 / 56, aligned=True)
    got3 = P.roi_align(P.Feat.from_nchw(feat.to(DEV)), boxes3[:, 1:].contiguous().to(DEV), 24 / 56, 16, 20)
    close(got3.to_nchw(), ref3, 1e-5, "roi grid, rows kernel")
    ref4 = o_ops.roi_align(feat.repeat(2, 1, 1, 1), torch.cat([torch.arange(2.0)[:, None], boxes3[:, 1:]], 1), (48, 64), 24 / 56, aligned=True)
    got4 = P.roi_align(P.Feat.from_nchw(feat.to(DEV)), boxes3[:, 1:].contiguous().to(DEV), 24 / 56, 48, 64)
    close(got4.to_nchw(), ref4, 1e-5, "roi zoom, rows kernel, out-of-map samples")
    # degenerate boxes (roi_w or roi_h <= 0: an empty sampling grid) give zeros whatever kernel the output height selects
    boxes5 = torch.tensor([[0, 30.0, 10.0, 30.0, 40.0], [0, 10.0, 30.0, 50.0, 20.0], [0, 0.0, 0.0, 84.0, 56.0]])
    for oh, ow in ((8, 8), (16, 20), (48, 64)):
        ref5 = o_ops.roi_align(feat.repeat(3, 1, 1, 1), torch.cat([torch.arange(3.0)[:, None], boxes5[:, 1:]], 1), (oh, ow), 24 / 56, aligned=True)
        got5 = P.roi_align(P.Feat.from_nchw(feat.to(DEV)), boxes5[:, 1:].contiguous().to(DEV), 24 / 56, oh, ow)
        assert float(ref5[:2].abs().max()) == 0.0 and float(got5.to_nchw()[:2].abs().max()) == 0.0
        close(got5.to_nchw(), ref5, 1e-5, f"roi degenerate boxes {oh}x{ow}")


@pytest.mark.parametrize("case", [(2, 32, 12, 16, 24, 32), (1, 1, 56, 84, 28, 42), (1, 98, 7, 11, 14, 21), (1, 256, 16, 16, 14, 14)])
def test_upsample(P, case):
    n, c, h, w, oh, ow = case
    x = rnd(1, n, c, h, w)
    got = P.upsample_bilinear(P.Feat.from_nchw(x.to(DEV)), oh, ow)
    close(got.to_nchw(), o_ops.bilinear_ac(x, (oh, ow)), 2e-6)


@pytest.mark.parametrize("case", [(3, 56, 84, 56, 84, 64), (2, 56, 84, 28, 42, 32), (2, 24, 32, 48, 61, 8), (1, 12, 16, 3, 5, 0)])
def test_depth_pair_fill_is_the_two_placements(P, case):
    """[feat | pred1 | pred2 | 0 | 0]: the fused tail writer == two 1-channel bilinear placements + zeroed pads, bit for bit,
    and == the oracle's bilinear(align_corners=True) (fusion_model.py:91-118)"""
    n, h, w, oh, ow, c0 = case
    p1, p2 = rnd(3, n, 1, h, w), rnd(4, n, 1, h, w)
    f1, f2 = (P.Feat(t.to(DEV).view(n, h, w, 1).contiguous()) for t in (p1, p2))
    ref = P.Feat(torch.zeros((n, oh, ow, c0 + 4), device=DEV), c0 + 2)
    P.upsample_bilinear(f1, oh, ow, out=ref.slice(c0, 1))
    P.upsample_bilinear(f2, oh, ow, out=ref.slice(c0 + 1, 1))
    got = P.Feat.alloc_raw(n, oh, ow, c0 + 2, DEV)
    got.buf.fill_(float("nan"))
    if c0:
        got.buf[..., :c0] = 0
    P.depth_pair_fill(f1, f2, got, c0)
    assert torch.equal(got.buf, ref.buf)
    close(got.buf[..., c0].cpu(), o_ops.bilinear_ac(p1, (oh, ow))[:, 0], 1e-5, "pair fill vs oracle")
    with pytest.raises(RuntimeError, match="16-byte"):
        P.L.check(P.L.load().prv2_depth_pair_fill(f1.ptr, f2.ptr, n, h, w, oh, ow, got.ptr + 4, got.ld, 0), "x")


def test_roi_source_writes_what_roi_align_materialises(P):
    """the un-materialised pyramid level gathers the same numbers into a slice of a concat buffer (patchrefinerplus.py:263-283)"""
    feat = P.Feat.from_nchw(rnd(1, 1, 32, 24, 32).to(DEV))
    bboxs = torch.tensor([[0, 0, 192, 108], [192, 108, 384, 216], [96, 54, 288, 162]])
    boxes = o_tiling.bboxs_to_feat(bboxs, (216, 384), (56, 84))[:, 1:].contiguous().to(DEV)
    ref = P.roi_align(feat, boxes, 24 / 56, 24, 32)
    src = P.RoiSource(feat, boxes, 24 / 56, 24, 32)
    cat = P.Feat(torch.zeros(3, 24, 32, 32 + 40, device=DEV))
    src.write(cat.slice(40, 32))
    assert torch.equal(cat.buf[..., 40:72], ref.buf) and float(cat.buf[..., :40].abs().max()) == 0
    assert torch.equal(src.materialize().buf, ref.buf) and (src.n, src.h, src.w, src.c) == (3, 24, 32, 32)


def test_blend_sequence(P):
    g = torch.Generator().manual_seed(4)
    ph, pw, MH, MW = 24, 32, 48, 64
    mask = torch.tensor(o_ops.generatemask((ph, pw), border=0.15))
    assert int((mask == 0).sum()) > 0
    preds = torch.rand(4, ph, pw, generator=g) * 10
    tiles0 = torch.tensor([[0, 0], [0, 32], [24, 0], [24, 32]], dtype=torch.int32)
    pd, cnt = torch.zeros(MH, MW), torch.zeros(MH, MW)
    for i, (h0, w0) in enumerate(tiles0.tolist()):
        cnt[h0:h0 + ph, w0:w0 + pw] = mask
        pd[h0:h0 + ph, w0:w0 + pw] = preds[i]
    ram = o_tiling.RunningAverageMap(pd, cnt)
    avg_d = torch.zeros(MH, MW, device=DEV)
    cnt_d = torch.zeros(MH, MW, device=DEV)
    P.blend_paste(avg_d, cnt_d, preds.to(DEV), mask.to(DEV), tiles0.to(DEV), ph, pw)
    assert torch.equal(avg_d.cpu(), ram.average_map) and torch.equal(cnt_d.cpu(), ram.count_map)
    # second pass: offset tiles incl. two overlapping ones (order matters)
    preds2 = torch.rand(3, ph, pw, generator=g) * 10
    tiles1 = torch.tensor([[12, 16], [20, 24], [0, 16]], dtype=torch.int32)
    for i, (h0, w0) in enumerate(tiles1.tolist()):
        c, p = torch.zeros(MH, MW), torch.zeros(MH, MW)
        c[h0:h0 + ph, w0:w0 + pw] = mask
        p[h0:h0 + ph, w0:w0 + pw] = preds2[i]
        ram.update(p, c)
    P.blend_update(avg_d, cnt_d, preds2.to(DEV), mask.to(DEV), tiles1.to(DEV), ph, pw)
    assert torch.equal(avg_d.cpu(), ram.average_map) and torch.equal(cnt_d.cpu(), ram.count_map)
    # resize + raw-resolution tiles with nearest-upsampled predictions
    ram.resize((54, 96))
    a2, c2 = P.blend_resize(avg_d, cnt_d, 54, 96)
    assert torch.equal(a2.cpu(), ram.average_map)
    close(c2, ram.count_map, 1e-6)
    rh, rw = 27, 48
    mask_r = torch.tensor(o_ops.generatemask((rh, rw), border=0.15) + 1e-3)
    preds3 = torch.rand(2, ph, pw, generator=g) * 10
    tiles2 = torch.tensor([[3, 7], [20, 7]], dtype=torch.int32)
    up = o_ops.nearest(preds3[:, None], (rh, rw))[:, 0]
    ram.count_map = c2.cpu().clone()
    for i, (h0, w0) in enumerate(tiles2.tolist()):
        c, p = torch.zeros(54, 96), torch.zeros(54, 96)
        c[h0:h0 + rh, w0:w0 + rw] = mask_r
        p[h0:h0 + rh, w0:w0 + rw] = up[i]
        ram.update(p, c)
    P.blend_update(a2, c2, preds3.to(DEV), mask_r.to(DEV), tiles2.to(DEV), rh, rw)
    assert torch.equal(a2.cpu(), ram.average_map) and torch.equal(c2.cpu(), ram.count_map)


@pytest.mark.parametrize("prec,tol", [("bf16x3", 3e-5), ("bf16", 2e-2)])
def test_conv2d_split_precision(P, prec, tol):
    """bf16x3 (hi*hi + hi*lo + lo*hi, fp32 accumulate) keeps ~16 mantissa bits; plain bf16 does not."""
    from patchrefinerv2_amd import lib as L
    pr = L.PREC_NAMES[prec]
    for (n, h, w, cin, cout, k) in ((1, 24, 32, 258, 256, 3), (2, 16, 16, 64, 32, 3), (1, 40, 24, 512, 130, 1)):
        x = rnd(1, n, cin, h, w) * 3
        wt = rnd(2, cout, cin, k, k) / np.sqrt(cin * k * k)
        b = rnd(3, cout)
        ref = F.gelu(F.conv2d(x.double(), wt.double(), b.double(), padding=k // 2)).float()
        y = P.conv2d(P.Feat.from_nchw(x.to(DEV)), P.pack_conv(wt.to(DEV), b.to(DEV), prec=pr), act=P.ACT_GELU)
        close(y.to_nchw(), ref, tol, f"{prec} conv {cin}->{cout} k{k}")
        if prec == "bf16x3":
            err = float((y.to_nchw().cpu() - ref).abs().max())
            assert err > 0  # not accidentally the fp32 kernel
    wt = rnd(4, 48, 40, 2, 2) / 7
    x = rnd(5, 1, 48, 6, 9)
    ref = F.conv_transpose2d(x, wt, None, stride=2)
    close(P.conv2d(P.Feat.from_nchw(x.to(DEV)), P.pack_conv(wt.to(DEV), None, convt_k=2, prec=pr)).to_nchw(), ref, tol)


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_conv3x3_halo_equals_generic(P, prec):
    """the LDS-halo kernel and the generic per-tap kernel agree (k order differs: slab-major vs tap-major)"""
    from patchrefinerv2_amd import lib as L
    x = rnd(1, 2, 70, 21, 45)
    wt = rnd(2, 136, 70, 3, 3) / 25
    b = rnd(3, 136)
    cw = P.pack_conv(wt.to(DEV), b.to(DEV), prec=L.PREC_NAMES[prec])
    xf = P.Feat.from_nchw(x.to(DEV))
    a = P.conv2d(xf, cw, act=P.ACT_RELU).to_nchw()
    g = P.conv2d(xf, cw, act=P.ACT_RELU, force_generic=True).to_nchw()
    close(a, g.cpu(), 2e-6 if prec == "f32" else 2e-5)
    assert not torch.equal(a, g) or prec != "f32"  # really two different kernels


def test_zoe_head_ops(P):
    """softplus epilogue, add, inverse attractor (alpha=300, gamma=2) and the log-binomial expectation"""
    from oracle import zoe as o_zoe
    x = rnd(1, 2, 48, 9, 11)
    w, b = rnd(2, 16, 48, 1, 1) / 5, rnd(3, 16)
    y = P.conv2d(P.Feat.from_nchw(x.to(DEV)), P.pack_conv(w.to(DEV), b.to(DEV)), act=P.ACT_SOFTPLUS)
    close(y.to_nchw(), F.softplus(F.conv2d(x * 1.0, w, b)), 2e-5)
    big = P.conv2d(P.Feat.from_nchw((x * 50).to(DEV)), P.pack_conv(w.to(DEV), b.to(DEV)), act=P.ACT_SOFTPLUS)
    close(big.to_nchw(), F.softplus(F.conv2d(x * 50, w, b)), 2e-5)   # threshold=20 branch
    a2, b2 = rnd(4, 2, 12, 9, 11), rnd(5, 2, 12, 9, 11)
    close(P.add(P.Feat.from_nchw(a2.to(DEV)), P.Feat.from_nchw(b2.to(DEV))).to_nchw(), a2 + b2, 0)
    A = F.softplus(rnd(6, 2, 16, 9, 11))
    bins = F.softplus(rnd(7, 2, 64, 9, 11)) * 3
    ref = bins + torch.mean(o_zoe.inv_attractor(A.unsqueeze(2) - bins.unsqueeze(1)), dim=1)
    got = P.zoe_attractor(P.Feat.from_nchw(A.to(DEV)), P.Feat.from_nchw(bins.to(DEV)), 300.0)
    close(got.to_nchw(), ref, 2e-6)
    pt = F.softplus(rnd(8, 2, 4, 9, 11) * 2)
    centers = F.softplus(rnd(9, 2, 64, 9, 11)) * 10
    pp = (pt[:, 0] + 1e-4) / (pt[:, 0] + pt[:, 1] + 2e-4)
    tt = ((pt[:, 2] + 1e-4) / (pt[:, 2] + pt[:, 3] + 2e-4)).unsqueeze(1) * (50 - 0.0212) + 0.0212
    xx = pp.unsqueeze(1)
    k = torch.arange(0, 64).view(1, -1, 1, 1)
    yy = o_zoe.log_binom(torch.Tensor([63.]).view(1, 1, 1, 1), k) + k * torch.log(xx.clamp(1e-4, 1)) + \
        (63 - k) * torch.log((1 - xx).clamp(1e-4, 1))
    refd = (torch.softmax(yy / tt, dim=1) * centers).sum(1, keepdim=True)
    gotd = P.zoe_logbinom_depth(P.Feat.from_nchw(pt.to(DEV)), P.Feat.from_nchw(centers.to(DEV)), 0.0212, 50.0)
    close(gotd, refd, 2e-5)


@pytest.mark.parametrize("case", [(2, 20, 40, 66, 32, 3), (1, 9, 33, 130, 128, 3), (1, 12, 16, 64, 64, 3), (2, 7, 5, 48, 96, 1)])
def test_conv_fused_layernorm(P, case):
    """conv -> channels-first LayerNorm -> GELU in one launch (cout <= 128) == the unfused sequence"""
    from oracle.fusion import ln_cf
    n, h, w, cin, cout, k = case
    x = rnd(1, n, cin, h, w)
    wt, b = rnd(2, cout, cin, k, k) / np.sqrt(cin * k * k), rnd(3, cout)
    lw, lb = 1 + 0.1 * rnd(4, cout), 0.1 * rnd(5, cout)
    ref = F.gelu(ln_cf(F.conv2d(x, wt, b, padding=k // 2), lw, lb))
    y = P.conv2d(P.Feat.from_nchw(x.to(DEV)), P.pack_conv(wt.to(DEV), b.to(DEV)), act=P.ACT_GELU, ln=(lw.to(DEV), lb.to(DEV)))
    close(y.to_nchw(), ref, 2e-5, f"fused LN {case}")
    # cout > 128 falls back to conv + row-LN pass
    wt2 = rnd(6, 160, cin, k, k) / np.sqrt(cin * k * k)
    lw2, lb2 = 1 + 0.1 * rnd(7, 160), 0.1 * rnd(8, 160)
    ref2 = F.relu(ln_cf(F.conv2d(x, wt2, None, padding=k // 2), lw2, lb2))
    y2 = P.conv2d(P.Feat.from_nchw(x.to(DEV)), P.pack_conv(wt2.to(DEV), None), act=P.ACT_RELU, ln=(lw2.to(DEV), lb2.to(DEV)))
    close(y2.to_nchw(), ref2, 2e-5)


ONE_BY_ONE_CASES = [
    # n, h, w, cin, cout, opts   -- M = n*h*w >= 512 rows: gemm_m16.hip (cout > 64) / conv1x1_small (cin, cout <= 64, M >= 4096)
    (1, 24, 40, 256, 256, dict(bias=True, act="sigmoid", mul=True, res=True)),          # GatedConvUnit f3 gate
    (2, 20, 25, 98, 130, dict(bias=True, relu_in=True)),                                  # ragged K (98) and N (130), last tile 232 rows
    (1, 33, 31, 512, 96, dict(bias=True, act="gelu", ln=True)),                           # fused channels-first LayerNorm (cout <= 128)
    (1, 1025, 1, 160, 192, dict(bias=True, gamma=True, res=True, res2=True)),             # ViT-style linear, 5 slabs
    (3, 40, 40, 32, 32, dict(bias=True, act="sigmoid", mul=True)),                        # small-channel gate (fp32 VALU kernel)
    (2, 48, 48, 64, 40, dict(bias=True, relu_in=True, res=True, gamma=True)),
    (1, 70, 70, 36, 8, dict(act="gelu")),
]


@pytest.mark.parametrize("case", ONE_BY_ONE_CASES)
def test_conv1x1_split_precision_kernels(P, case):
    """the bf16x3 1x1 paths (rows straight into MFMA operand registers; fp32 VALU for few channels) vs fp64, and vs the
    generic kernel; input and output are channel slices of wider buffers (ldx > cin, ldy > cout)"""
    from patchrefinerv2_amd import lib as L
    n, h, w, cin, cout, o = case
    pr = L.PREC_NAMES["bf16x3"]
    x = rnd(1, n, cin, h, w) * 2
    wt = rnd(2, cout, cin, 1, 1) / np.sqrt(cin)
    bias = rnd(3, cout) if o.get("bias") else None
    ref = F.conv2d((F.relu(x) if o.get("relu_in") else x).double(), wt.double(), bias.double() if bias is not None else None)
    lnw = lnb = None
    if o.get("ln"):
        lnw, lnb = rnd(8, cout).abs() + 0.5, rnd(9, cout)
        u = ref.mean(1, keepdim=True)
        s = (ref - u).pow(2).mean(1, keepdim=True)
        ref = (ref - u) / torch.sqrt(s + 1e-6) * lnw.double().view(1, -1, 1, 1) + lnb.double().view(1, -1, 1, 1)
    if o.get("act") == "gelu":
        ref = F.gelu(ref)
    if o.get("act") == "sigmoid":
        ref = torch.sigmoid(ref)
    gamma = rnd(4, cout) if o.get("gamma") else None
    if gamma is not None:
        ref = ref * gamma.double().view(1, -1, 1, 1)
    mul, res, res2 = (rnd(s_, n, cout, h, w) if o.get(k_) else None for s_, k_ in ((5, "mul"), (6, "res"), (7, "res2")))
    if mul is not None:
        ref = mul.double() * ref
    if res is not None:
        ref = ref + res.double()
    if res2 is not None:
        ref = ref + res2.double()
    ref = ref.float()
    act = {"gelu": P.ACT_GELU, "sigmoid": P.ACT_SIGMOID, None: P.ACT_NONE}[o.get("act")]
    cw = P.pack_conv(wt.to(DEV), bias.to(DEV) if bias is not None else None, prec=pr)
    f = lambda t: P.Feat.from_nchw(t.to(DEV)) if t is not None else None  # noqa: E731
    xcat = P.Feat.alloc(n, h, w, cin + 8, DEV)      # x lives in channels [4, 4 + cin) of a wider buffer
    xcat.buf.fill_(7.0)
    P.add(f(x), f(torch.zeros_like(x)), out=xcat.slice(4, cin))
    kw = dict(relu_in=bool(o.get("relu_in")), act=act, gamma=gamma.to(DEV) if gamma is not None else None, mul=f(mul),
              res=f(res), res2=f(res2), ln=(lnw.to(DEV), lnb.to(DEV)) if lnw is not None else None)
    outs = []
    for force in (False, True):
        ycat = P.Feat.alloc(n, h, w, cout + 12, DEV)
        ycat.buf.fill_(-3.0)
        y = P.conv2d(xcat.slice(4, cin), cw, ycat.slice(8, cout), force_generic=force, **kw)
        outs.append(y.to_nchw())
        assert float(ycat.buf[..., :8].min()) == -3.0 and float(ycat.buf[..., 8 + cout:].max()) == -3.0  # neighbours untouched
    close(outs[0], ref, 3e-5, f"1x1 {case}")
    close(outs[0], outs[1].cpu(), 2e-5, f"1x1 vs generic {case}")


STRIP_CASES = [
    # n, h, w, cin, cout, opts -- widths 32k + (1..8): 32-pixel tile columns + a remainder strip (32 x 8 tiles in the bf16
    # modes, the generic kernel's column window in f32 mode); heights that are not multiples of 8 / 32
    (2, 37, 67, 64, 128, dict(bias=True, act="gelu")),
    (1, 50, 65, 98, 130, dict(bias=True, relu_in=True, res=True)),          # rem 1, tail tile (cin = 96 + 2), 2 channel tiles
    (3, 19, 72, 34, 40, dict(bias=True, ln=True, act="gelu")),              # rem 8, BN = 64, fused LayerNorm, tail tile
    (2, 33, 130, 66, 32, dict(bias=True, act="sigmoid", mul=True)),         # rem 2, BN = 32, gate
    (1, 64, 100, 256, 256, dict(res=True, res2=True, gamma=True, bias=True)),  # rem 4
    # cout <= 32 without tail tile: persistent workgroups walk several tiles each (> 256 tiles), next tile's halo prefetched
    (3, 200, 264, 64, 32, dict(bias=True, act="gelu")),                      # 2 slabs, 675 tiles + strip (rem 8)
    (2, 264, 352, 32, 32, dict(bias=True, relu_in=True, res=True)),          # 1 slab: halo buffer parity flips per tile
    (2, 150, 390, 128, 16, dict(bias=True, ln=True, act="gelu")),            # 4 slabs, cout 16, fused LN, rem 6, H % 8 != 0
    (5, 96, 224, 96, 24, dict(bias=True, act="sigmoid", mul=True)),          # 3 slabs (odd), 420 tiles: uneven tiles per workgroup
]


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("bf16x3", 3e-5)])
@pytest.mark.parametrize("case", STRIP_CASES)
def test_conv3x3_remainder_strip(P, case, prec, tol):
    from patchrefinerv2_amd import lib as L
    n, h, w, cin, cout, o = case
    x = rnd(1, n, cin, h, w)
    wt = rnd(2, cout, cin, 3, 3) / np.sqrt(cin * 9)
    bias = rnd(3, cout) if o.get("bias") else None
    ref = F.conv2d((F.relu(x) if o.get("relu_in") else x).double(), wt.double(), bias.double() if bias is not None else None,
                   padding=1)
    lnw = lnb = None
    if o.get("ln"):
        lnw, lnb = rnd(8, cout).abs() + 0.5, rnd(9, cout)
        u = ref.mean(1, keepdim=True)
        s = (ref - u).pow(2).mean(1, keepdim=True)
        ref = (ref - u) / torch.sqrt(s + 1e-6) * lnw.double().view(1, -1, 1, 1) + lnb.double().view(1, -1, 1, 1)
    if o.get("act") == "gelu":
        ref = F.gelu(ref)
    if o.get("act") == "sigmoid":
        ref = torch.sigmoid(ref)
    gamma = rnd(4, cout) if o.get("gamma") else None
    if gamma is not None:
        ref = ref * gamma.double().view(1, -1, 1, 1)
    mul, res, res2 = (rnd(s_, n, cout, h, w) if o.get(k_) else None for s_, k_ in ((5, "mul"), (6, "res"), (7, "res2")))
    if mul is not None:
        ref = mul.double() * ref
    if res is not None:
        ref = ref + res.double()
    if res2 is not None:
        ref = ref + res2.double()
    ref = ref.float()
    act = {"gelu": P.ACT_GELU, "sigmoid": P.ACT_SIGMOID, None: P.ACT_NONE}[o.get("act")]
    cw = P.pack_conv(wt.to(DEV), bias.to(DEV) if bias is not None else None, prec=L.PREC_NAMES[prec])
    f = lambda t: P.Feat.from_nchw(t.to(DEV)) if t is not None else None  # noqa: E731
    kw = dict(relu_in=bool(o.get("relu_in")), act=act, gamma=gamma.to(DEV) if gamma is not None else None, mul=f(mul),
              res=f(res), res2=f(res2), ln=(lnw.to(DEV), lnb.to(DEV)) if lnw is not None else None)
    ycat = P.Feat.alloc(n, h, w, cout + 8, DEV)
    ycat.buf.fill_(-3.0)
    y = P.conv2d(f(x), cw, ycat.slice(4, cout), **kw)
    close(y.to_nchw(), ref, tol, f"strip {case} {prec}")
    assert float(ycat.buf[..., :4].min()) == -3.0 and float(ycat.buf[..., 4 + cout:].max()) == -3.0
    close(y.to_nchw(), P.conv2d(f(x), cw, force_generic=True, **kw).to_nchw().cpu(), tol, f"strip vs generic {case} {prec}")
    # timed (profiled) launches issue tiles and strip separately in f32 mode (prv2_conv_desc.part): same result
    P.PROFILER.start(timed=True)
    y2 = P.conv2d(f(x), cw, **kw)
    P.PROFILER.stop()
    assert torch.equal(y2.to_nchw(), y.to_nchw())


@pytest.mark.parametrize("case", [(54, 96, 216, 384, "u8"), (45, 80, 90, 160, "u8"), (64, 48, 33, 97, "f32"), (30, 40, 30, 40, "u8"),
                                  (120, 200, 61, 77, "f32")])
def test_bicubic_input_stage(P, case):
    """device read_image: RGB / 255 -> bicubic(align_corners=True) -> CHW, vs the reference's float64 F.interpolate
    (general_dataset.py:55-60: cv2 image / 255.0 is a float64 array)"""
    h, w, H, W, kind = case
    g = torch.Generator().manual_seed(5)
    if kind == "u8":
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        src = img.double() / 255.0
    else:
        img = torch.rand(h, w, 3, generator=g)
        src = img.double()
    ref = F.interpolate(src.unsqueeze(0).permute(0, 3, 1, 2), (H, W), mode="bicubic", align_corners=True)[0].float()
    out = P.bicubic_resize(img.to(DEV), H, W).cpu()
    assert out.shape == (3, H, W)
    assert float((out - ref).abs().max()) <= 2.4e-7, float((out - ref).abs().max())   # <= 2 ulp at 1.0


# ---- split-swizzled operand path of the large ViT linears (csrc/gemm_ss.hip) ---------------------------------------------------
@pytest.mark.parametrize("M,K,N", [(100, 64, 128), (1037, 256, 384), (4100, 1024, 1024), (2500, 384, 1152)])
@pytest.mark.parametrize("tile", ["128", "256", "auto"])  # (auto: the library's choice -- 64 x 64 tiles with 2 / 4 LDS stages on the small grids)
def test_gemm_ss_bit_equal_to_gemm16(P, M, K, N, tile, monkeypatch):
    """gemm_ss_kernel (pre-split operands, LDS-DMA) == gemm16_kernel (fp32 operands split in the kernel), BIT FOR BIT: same
    split, same three products in the same order, same epilogue.  That equality is what lets the host pick the kernel by
    problem size without making results depend on the batch.  Also vs the fp32 reference within the bf16x3 tolerance."""
    if tile != "auto":
        monkeypatch.setenv("PRV2_GEMM_SS_TILE", tile)
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    gam, res = 1 + 0.1 * torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    cw = P.pack_conv(w.to(DEV), b.to(DEV), prec=P.L.PREC_BF16X3)
    xd, gd, rd = x.to(DEV), gam.to(DEV), res.to(DEV)
    xs = P.split_ss(xd)
    for kw in (dict(), dict(act=P.ACT_GELU), dict(gamma=gd, res=rd)):
        y16 = P.linear(xd, cw, **kw)
        yss = P.gemm_ss(xs, cw, **kw)
        assert torch.equal(y16, yss), kw.keys()
    ref = x @ w.t() + b
    assert float((yss.cpu() - (ref * gam + res)).abs().max()) < 2e-4 * float(ref.abs().max())
    # chain with a split-swizzled intermediate (fc1 -> GELU -> fc2): the producer's split == the consumer's in-kernel split
    K2 = max(K, 128)  # (a 64-column layer would go to the generic kernel on the fp32-operand side)
    w2 = torch.randn(K2, N, generator=g) / N ** 0.5
    cw2 = P.pack_conv(w2.to(DEV), None, prec=P.L.PREC_BF16X3)
    a = P.linear(P.linear(xd, cw, act=P.ACT_GELU), cw2)
    bq = P.gemm_ss(P.gemm_ss(xs, cw, act=P.ACT_GELU, out_ss=True), cw2)
    assert torch.equal(a, bq)


@pytest.mark.parametrize("M,K,N", [(9000, 96, 2048), (23000, 160, 768), (16500, 128, 1024), (9300, 32, 2048)])  # 288 (4 x 8-blocked) / 270 / 260 tiles of 256 x 256: > 256 workgroups; 37 row tiles blocked: 24 idle ids, one slab
@pytest.mark.parametrize("ppb", ["1", "2", "4", "8"])
def test_gemm_ss_persistent_workgroups_walk_several_tiles_bit_equal(P, M, K, N, ppb, monkeypatch):
    """gemm_ss_p_kernel (one persistent workgroup per CU: the next tile's first slab is DMA'd under the current tile's last slab and
    epilogue, the LDS stage parity carries over from tile to tile -- odd and even slab counts) == the one-tile-per-workgroup kernel
    == gemm16_kernel, bit for bit, for every DMA spread PPB; fp32 rows with gamma + residual (in place, as the ViT blocks call it) and
    split-swizzled rows with GELU / the qkv column scale."""
    monkeypatch.setenv("PRV2_GEMM_SS_TILE", "256")
    monkeypatch.setenv("PRV2_GSS_PPB", ppb)
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(DEV)
    cw = P.pack_conv((torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), (torch.randn(N, generator=g) * 0.1).to(DEV), prec=P.L.PREC_BF16X3)
    gam, res = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    xs = P.split_ss(x)
    got = {}
    for persist in ("1", "0"):
        monkeypatch.setenv("PRV2_GSS_PERSIST", persist)
        r_in = res.clone()
        got[persist] = (P.gemm_ss(xs, cw), P.gemm_ss(xs, cw, gamma=gam, res=r_in, out=r_in), P.gemm_ss(xs, cw, act=P.ACT_GELU, out_ss=True))
        assert P.L.load().prv2_last_kernel().decode().startswith("gemm_ss")
    for a, b in zip(got["1"], got["0"]):
        assert torch.equal(a, b)
    assert torch.equal(got["1"][0], P.linear(x, cw)) and torch.equal(got["1"][1], P.linear(x, cw, gamma=gam, res=res))


@pytest.mark.parametrize("B,N,H,bias", [(3, 259, 6, None), (2, 1025, 16, None), (2, 769, 16, "image"), (1, 130, 2, "rows"), (5, 37, 3, None)])
def test_attention_on_split_swizzled_qkv_is_bit_equal_to_the_pre_pass_path(P, B, N, H, bias):
    """prv2_gemm_ss_qkv -> prv2_attention_qkv_ss (the Linear writes bf16 hi / lo with q pre-scaled, the attention kernel reads those rows and
    transposes V with ds_read_b64_tr_b16) == prv2_gemm_ss (fp32 rows) -> qkv_split_kernel -> attention_bf16x3_kernel, bit for bit: split-swizzled
    and fp32 output rows, no bias / bias rows / the packed bias image; 1025 and 769 tokens are the DINOv2 / BEiT sequence lengths (one query in the
    last query tile, keys masked in the last key tile); rows of a batch > 1 start at odd global rows (the swizzle key is a function of the global row)."""
    g = torch.Generator().manual_seed(B * N + H)
    D, M = H * 64, B * N
    x = torch.randn(M, D, generator=g).to(DEV)
    cw = P.pack_conv((torch.randn(3 * D, D, generator=g) / D ** 0.5).to(DEV), (torch.randn(3 * D, generator=g) * 0.1).to(DEV), prec=P.L.PREC_BF16X3)
    xs = P.split_ss(x)
    bt = None
    if bias is not None:
        ld = -(-N // 64) * 64
        rows = torch.zeros(H, N, ld)
        rows[:, :, :N] = torch.randn(H, N, N, generator=g)
        bt = P.pack_attention_bias(rows.to(DEV), N) if bias == "image" else rows.to(DEV)
    qkv = P.gemm_ss(xs, cw)
    qkv_ss = P.gemm_ss_qkv(xs, cw, H)
    for out_ss in (True, False):
        ref = P.attention(qkv, B, N, H, P.L.PREC_BF16X3, bias=bt, out_ss=out_ss)
        got = P.attention_qkv_ss(qkv_ss, B, N, H, bias=bt, out_ss=out_ss)
        assert torch.equal(ref, got), (out_ss, float((ref - got).abs().max()))
    # and against float64 softmax attention (the tolerance of test_attention's bf16x3 case)
    q, k, v = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3).double().cpu() for t in qkv.reshape(B, N, 3, H, 64).unbind(2))
    sc = q @ k.transpose(-1, -2) * 0.125
    if bias is not None:
        sc = sc + rows[None, :, :, :N].double()
    want = (sc.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(M, D)
    assert float((got.double().cpu() - want).abs().max()) < 2e-4 * max(1.0, float(want.abs().max()))


def test_layernorm_ss_and_attention_ss_feed_gemm_ss_bit_equal(P):
    g = torch.Generator().manual_seed(5)
    M, D = 777, 384
    x = torch.randn(M, D, generator=g).to(DEV)
    lw, lb = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    cw = P.pack_conv((torch.randn(3 * D, D, generator=g) / D ** 0.5).to(DEV), (torch.randn(3 * D, generator=g) * 0.1).to(DEV), prec=P.L.PREC_BF16X3)
    h = torch.empty_like(x)
    P.layernorm_rows(x, M, D, D, lw, lb, 1e-6, P.ACT_NONE, h, D)
    hs = torch.empty_like(x)
    P.layernorm_ss(x, M, D, D, lw, lb, 1e-6, hs)
    qkv = P.linear(h, cw)
    assert torch.equal(qkv, P.gemm_ss(hs, cw))
    B, N, H = 3, 259, 6
    cwp = P.pack_conv((torch.randn(D, D, generator=g) / D ** 0.5).to(DEV), None, prec=P.L.PREC_BF16X3)
    a = P.attention(qkv, B, N, H, P.L.PREC_BF16X3)
    a_ss = P.attention(qkv, B, N, H, P.L.PREC_BF16X3, out_ss=True)
    assert torch.equal(P.linear(a, cwp), P.gemm_ss(a_ss, cwp))


UPS_CASES = [
    # n, (h, w) of the low-resolution source, (H, W) of the conv, source channels c1, cin, cout
    (2, (24, 32), (48, 64), 256, 256, 128),      # output_conv1: every channel comes from the upsampled path_1 (x2)
    (1, (20, 33), (40, 66), 64, 98, 98),         # f2r_agg: [up(x1) 64 | x2 32 | pred1 | pred2], tail tile, width 64 + a 2-column strip
    (2, (13, 17), (25, 33), 128, 194, 194),      # ragged pyramid sizes (not x2), ragged rows
    (1, (12, 16), (24, 32), 512, 770, 770),      # the lowest decoder stage
    (1, (6, 9), (29, 70), 32, 66, 130),          # strong zoom, strip of 6 columns, cout over one 128-column tile
]


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", UPS_CASES)
def test_conv2d_ups_equals_upsample_then_conv(P, case, prec):
    """prv2_conv2d_ups (bilinear(align_corners) upsample of the first c1 input channels formed inside the 3x3 conv's tile loader:
    fusion_model.py:15-24, bi_directional_fusion_model.py:139-142,201) == prv2_upsample_bilinear into the concat buffer followed by
    prv2_conv2d, BIT FOR BIT; the first c1 channels of the concat buffer are never read (NaN-filled here)"""
    n, (h, w), (H, W), c1, cin, cout = case
    PR = P.L.PREC_NAMES[prec]
    u = P.Feat.from_nchw(rnd(1, n, c1, h, w).to(DEV))
    rest = rnd(2, n, cin - c1, H, W).to(DEV) if cin > c1 else None
    cw = P.pack_conv((rnd(3, cout, cin, 3, 3) / np.sqrt(9 * cin)).to(DEV), rnd(4, cout).to(DEV), pad=1, prec=PR)
    res = P.Feat.from_nchw(rnd(5, n, cout, H, W).to(DEV))

    def concat(fill_nan):
        x = P.Feat.alloc(n, H, W, cin, DEV)
        if rest is not None:
            x.buf[..., c1:cin] = rest.permute(0, 2, 3, 1)
        if fill_nan:
            x.buf[..., :c1] = float("nan")
        return x

    xr = concat(False)
    P.upsample_bilinear(u, H, W, out=xr.slice(0, c1))
    ref = P.conv2d(xr, cw, act=P.ACT_GELU, res=res)
    ref_kernel = P.L.load().prv2_last_kernel().decode()
    x = concat(True) if cin > c1 else P.UpsOnly(u, H, W)
    assert P.conv2d_ups_supported(x, u, cw)
    got = P.conv2d_ups(x, u, cw, act=P.ACT_GELU, res=res)
    assert "ups" in P.L.load().prv2_last_kernel().decode() and "ups" not in ref_kernel
    assert bool(torch.isfinite(got.buf).all())
    assert torch.equal(got.buf, ref.buf), float((got.buf - ref.buf).abs().max())
    # and against plain torch (fp32): the layer itself
    want = F.gelu(F.conv2d(torch.cat([F.interpolate(u.to_nchw().cpu(), (H, W), mode="bilinear", align_corners=True)] +
                                     ([rest.cpu()] if rest is not None else []), 1), rnd(3, cout, cin, 3, 3) / np.sqrt(9 * cin), rnd(4, cout), padding=1)) + res.to_nchw().cpu()
    close(got.to_nchw(), want, 3e-5 if prec == "bf16x3" else 2e-2, "conv2d_ups vs torch")


UPCONV_CASES = [
    # n, (h, w) of u, (H, W) of the output, cin, cout, act, bias
    (2, (24, 32), (48, 64), 64, 128, "none", True),       # output_conv1-like: whole tiles, four passes
    (1, (13, 21), (26, 42), 32, 98, "gelu", False),       # ragged tiles in both directions, cout = 96 + 2 (partial quad), one slab
    (2, (10, 20), (20, 40), 96, 40, "none", False),       # raw partial sum (the addend of conv2d_pre), two passes, second one partial
    (1, (9, 12), (17, 23), 64, 32, "relu", True),         # H = 2 h - 1: the largest source step the contract admits
    (1, (6, 8), (24, 32), 32, 64, "none", True),          # x4: source step 1/4
    (3, (4, 4), (8, 8), 256, 290, "gelu", True),          # tiny images (the low pyramid levels): pass groups spread over workgroups, 8 slabs
    (2, (16, 16), (28, 28), 64, 130, "gelu", False),      # x1.75 (DepthAnything's 256 -> 448): source step 0.556 -> the 14 x 24 tile shape
    (1, (11, 19), (19, 31), 32, 32, "none", True),        # source step 0.556 / 0.6: the contract's upper end, ragged 14 x 24 tiles
]


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", UPCONV_CASES)
def test_upconv3x3_vs_fp64_upsample_then_conv(P, case, prec):
    """prv2_upconv3x3 (csrc/upconv.hip: conv3x3 of a bilinear(align_corners=True) upsample as nine tap GEMMs at the LOW resolution + a
    four-corner gather per tap) against float64 F.interpolate -> F.conv2d (bi_directional_fusion_model.py:139-142,201,
    fusion_model.py:15-24), and within the same tolerance of prv2_conv2d_ups, the kernel that interpolates inside its loader"""
    n, (h, w), (H, W), cin, cout, act, bias = case
    PR = P.L.PREC_NAMES[prec]
    u_t = rnd(1, n, cin, h, w)
    w_t = rnd(2, cout, cin, 3, 3) / np.sqrt(9 * cin)
    b_t = rnd(3, cout) if bias else None
    cw = P.pack_conv(w_t.to(DEV), b_t.to(DEV) if bias else None, pad=1, prec=PR)
    u = P.Feat.from_nchw(u_t.to(DEV))
    assert P.upconv3x3_supported(u, H, W, cw)
    A = dict(none=P.ACT_NONE, gelu=P.ACT_GELU, relu=P.ACT_RELU)[act]
    out = P.Feat.alloc_raw(n, H, W, P.roundup(cout, 4) + 4, DEV)   # (a slice of a wider buffer: nothing behind cout may be touched)
    out.buf.fill_(-7.0)
    got = P.upconv3x3(u, H, W, cw, out=out.slice(0, cout), act=A)
    assert "upconv3x3" in P.L.load().prv2_last_kernel().decode()
    assert bool((out.buf[..., cout:] == -7.0).all()), "channels behind cout were written"
    up = torch.nn.functional.interpolate(u_t.double(), (H, W), mode="bilinear", align_corners=True)
    ref = torch.nn.functional.conv2d(up, w_t.double(), b_t.double() if bias else None, padding=1)
    ref = dict(none=lambda v: v, gelu=torch.nn.functional.gelu, relu=torch.relu)[act](ref).float()
    tol = 2e-5 if prec == "bf16x3" else 6e-3
    close(got.to_nchw(), ref, tol, f"upconv3x3 {case} {prec}")
    if cout > 64 and W >= 24 and H >= 4:  # the loader-interpolating kernel takes this layer too: the two must agree
        other = P.conv2d_ups(P.UpsOnly(u, H, W), u, cw, act=A)
        close(got.to_nchw(), other.to_nchw().cpu(), tol, f"upconv3x3 vs conv2d_ups {case} {prec}")


@pytest.mark.parametrize("case", [(2, (12, 16), 128, 64, 194), (1, (13, 20), 256, 32, 290), (1, (6, 12), 512, 128, 642)])
def test_upconv3x3_addend_splits_a_concat_conv(P, case):
    """UpSample.forward_hardcode's first conv (fusion_model.py:15-24) over cat([interpolate(x1), x2, pred1, pred2]) split by weight
    columns: prv2_conv2d over [x2 | p1 | p2] (raw) as the pre-activation addend of prv2_upconv3x3 over x1 -- IN PLACE (add == out) --
    against the fused-loader kernel on the whole concat and against float64"""
    n, (h, w), c1, c2, cout = case
    H, W, cin = 2 * h, 2 * w, c1 + c2 + 2
    PR = P.L.PREC_NAMES["bf16x3"]
    x1_t, rest_t, w_t = rnd(1, n, c1, h, w), rnd(2, n, cin - c1, H, W), rnd(3, cout, cin, 3, 3) / np.sqrt(9 * cin)
    x1 = P.Feat.from_nchw(x1_t.to(DEV))
    buf = P.Feat.alloc(n, H, W, cin, DEV)
    buf.buf[..., :c1] = float("nan")   # never read on either route
    buf.buf[..., c1:cin] = rest_t.to(DEV).permute(0, 2, 3, 1)
    whole = P.pack_conv(w_t.to(DEV), None, pad=1, prec=PR)
    up_w, rest_w = P.pack_conv(w_t[:, :c1].to(DEV), None, pad=1, prec=PR), P.pack_conv(w_t[:, c1:].to(DEV), None, pad=1, prec=PR)
    ref_k = P.conv2d_ups(buf, x1, whole, act=P.ACT_GELU).to_nchw().cpu()
    t = P.conv2d(buf.slice(c1, cin - c1), rest_w)
    got = P.upconv3x3(x1, H, W, up_w, out=t, act=P.ACT_GELU, add=t, bias=False)
    assert got.buf.data_ptr() == t.buf.data_ptr()
    up = torch.nn.functional.interpolate(x1_t.double(), (H, W), mode="bilinear", align_corners=True)
    ref = torch.nn.functional.gelu(torch.nn.functional.conv2d(torch.cat([up, rest_t.double()], 1), w_t.double(), None, padding=1)).float()
    close(got.to_nchw(), ref, 2e-5, f"split concat conv {case} vs float64")
    close(got.to_nchw(), ref_k, 2e-5, f"split concat conv {case} vs conv2d_ups")


def test_upconv3x3_border_taps_and_batch_independence(P):
    """taps that fall outside the OUTPUT image contribute nothing (zero padding of the conv, not of the source): constant input and
    all-ones weights give 9 / 6 / 4 x cin in the interior / on edges / in corners; an image's result does not depend on its batch"""
    PR = P.L.PREC_NAMES["bf16x3"]
    cin, cout = 32, 32
    cw = P.pack_conv(torch.ones(cout, cin, 3, 3, device=DEV), None, pad=1, prec=PR)
    u = P.Feat.from_nchw(torch.ones(1, cin, 20, 24, device=DEV))
    y = P.upconv3x3(u, 40, 48, cw).to_nchw().cpu()
    exp = torch.nn.functional.conv2d(torch.ones(1, 1, 40, 48), torch.ones(1, 1, 3, 3), padding=1) * cin
    close(y, exp.expand(1, cout, 40, 48), 1e-6, "border taps")
    cw2 = P.pack_conv((rnd(1, 64, 64, 3, 3) / 24).to(DEV), rnd(2, 64).to(DEV), pad=1, prec=PR)
    ub = rnd(3, 3, 64, 11, 17).to(DEV)
    full = P.upconv3x3(P.Feat.from_nchw(ub), 22, 34, cw2).to_nchw()
    one = P.upconv3x3(P.Feat.from_nchw(ub[1:2].contiguous()), 22, 34, cw2).to_nchw()
    assert torch.equal(full[1:2], one)


def test_upconv3x3_rejects_what_it_does_not_cover(P):
    PR = P.L.PREC_NAMES["bf16x3"]
    u = P.Feat.from_nchw(rnd(1, 1, 64, 12, 16).to(DEV))
    ok = P.pack_conv(rnd(2, 98, 64, 3, 3).to(DEV), None, pad=1, prec=PR)
    assert P.upconv3x3_supported(u, 24, 32, ok) and P.upconv3x3_supported(u, 23, 31, ok)
    assert P.upconv3x3_supported(u, 20, 32, ok) and not P.upconv3x3_supported(u, 19, 32, ok)                           # source step 11/19 <= 0.6 < 11/18
    assert not P.upconv3x3_supported(u, 24, 32, P.pack_conv(rnd(2, 98, 64, 3, 3).to(DEV), None, pad=1, prec=P.L.PREC_NAMES["f32"]))
    assert not P.upconv3x3_supported(P.Feat.from_nchw(rnd(1, 1, 48, 12, 16).to(DEV)), 24, 32, P.pack_conv(rnd(2, 98, 48, 3, 3).to(DEV), None, pad=1, prec=PR))
    with pytest.raises(RuntimeError):
        P.upconv3x3(u, 16, 32, ok)


def test_conv2d_ups_rejects_what_it_does_not_cover(P):
    PR = P.L.PREC_NAMES["bf16x3"]
    u = P.Feat.from_nchw(rnd(1, 1, 64, 12, 16).to(DEV))
    x = P.Feat.alloc(1, 24, 32, 98, DEV)
    ok = P.pack_conv(rnd(2, 98, 98, 3, 3).to(DEV), None, pad=1, prec=PR)
    assert P.conv2d_ups_supported(x, u, ok)
    assert not P.conv2d_ups_supported(x, u, P.pack_conv(rnd(2, 32, 98, 3, 3).to(DEV), None, pad=1, prec=PR))             # narrow kernels
    assert not P.conv2d_ups_supported(x, u, P.pack_conv(rnd(2, 98, 98, 3, 3).to(DEV), None, pad=1, prec=P.L.PREC_NAMES["f32"]))
    assert not P.conv2d_ups_supported(x, P.Feat.from_nchw(rnd(1, 1, 48, 12, 16).to(DEV)), ok)                            # channels % 32
    assert not P.conv2d_ups_supported(P.Feat.alloc(1, 24, 16, 98, DEV), P.Feat.from_nchw(rnd(1, 1, 64, 12, 8).to(DEV)), ok)  # width < 24
    with pytest.raises(RuntimeError):
        P.conv2d_ups(P.Feat.alloc(1, 24, 16, 98, DEV), P.Feat.from_nchw(rnd(1, 1, 64, 12, 8).to(DEV)), ok)


@pytest.mark.parametrize("case", [(2, 40, 66, 64, 32, 14, 18), (1, 25, 33, 130, 64, 25, 33), (2, 24, 32, 258, 128, 48, 64), (1, 29, 70, 66, 100, 8, 9)])
def test_conv2d_tail_equals_conv_then_depth_pair_fill(P, case):
    """prv2_conv2d_tail (conv + LayerNorm + GELU that also closes the row with [pred1 | pred2 | 0 | 0], fusion_model.py:91-118) ==
    prv2_conv2d followed by prv2_depth_pair_fill, bit for bit -- as a slice of a wider concat buffer too"""
    n, H, W, cin, cout, ph, pw = case
    PR = P.L.PREC_NAMES["bf16x3"]
    x = P.Feat.from_nchw(rnd(1, n, cin, H, W).to(DEV))
    cw = P.pack_conv((rnd(2, cout, cin, 3, 3) / np.sqrt(9 * cin)).to(DEV), None, pad=1, prec=PR)
    ln = ((torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5).to(DEV), (rnd(4, cout) * 0.1).to(DEV))
    p1, p2 = (P.Feat(rnd(s, n, ph, pw, 1).abs().to(DEV).contiguous()) for s in (5, 6))
    for c_before in (0, 32):   # the conv's slice starts at channel c_before of the buffer (decoder concat: [up(x1) | x2 | p1 p2])
        def run(fused):
            buf = P.Feat.alloc_raw(n, H, W, c_before + cout + 2, DEV)
            buf.buf.fill_(float("nan"))
            dst = buf.slice(c_before, cout)
            if fused:
                assert P.conv2d_tail_supported(x, cw, dst)
                P.conv2d_tail(x, cw, dst, p1, p2, act=P.ACT_GELU, ln=ln)
            else:
                P.conv2d(x, cw, dst, act=P.ACT_GELU, ln=ln)
                P.depth_pair_fill(p1, p2, buf, c_before + cout)
            return buf.buf[..., c_before:]
        a, b = run(True), run(False)
        assert bool(torch.isfinite(a).all()) and torch.equal(a, b), (c_before, float((a - b).abs().max()))
    assert not P.conv2d_tail_supported(x, cw, P.Feat.alloc(n, H, W, cout, DEV))                       # no room behind the slice
    c256 = P.pack_conv(rnd(2, 256, 64, 3, 3).to(DEV), None, pad=1, prec=PR)
    assert not P.conv2d_tail_supported(P.Feat.alloc(1, 24, 32, 64, DEV), c256, P.Feat.alloc_raw(1, 24, 32, 258, DEV).slice(0, 256))  # LayerNorm fused up to 128 columns


@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", [(2, 24, 32, 512), (1, 17, 29, 512), (3, 12, 16, 512)])
def test_x2_presplit_format_gate_unit(P, case, prec):
    """The pre-split "X2" activation format (include/prv2.h PRV2_FMT_*; csrc/conv3x3_gate.hip): a GatedConvUnit's concat buffer
    [out | coarse ROI] written by its producers in the gate kernel's operand format -- prv2_roi_align_x2 and the 256-column conv
    with an X2 output hold exactly hi + lo of what the fp32 versions write, and the fused gate kernel gives the SAME BITS on the X2
    buffer (x and mul) as on the fp32 one (bi_directional_fusion_model.py:56-82)"""
    n, h, w, cin = case
    F_ = cin // 2
    PR = P.L.PREC_NAMES[prec]
    g = torch.Generator().manual_seed(11)
    x = P.Feat.from_nchw(torch.randn(n, F_, h, w, generator=g).to(DEV))                                  # the unit's input
    coarse = P.Feat.from_nchw(torch.randn(1, F_, 2 * h, 2 * w, generator=g).to(DEV))                     # a coarse pyramid level
    boxes = torch.tensor([[1.0 + 3 * i, 2.0 + i, 1.0 + 3 * i + w / 2.0, 2.0 + i + h / 2.0] for i in range(n)], device=DEV)
    cw_c = P.pack_conv((torch.randn(F_, F_, 3, 3, generator=g) / (3 * F_ ** 0.5)).to(DEV), torch.randn(F_, generator=g).to(DEV) * 0.1, pad=1, prec=PR)
    cw_f = P.pack_conv((torch.randn(F_, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV), torch.randn(F_, generator=g).to(DEV) * 0.1, pad=1, prec=PR)
    w3 = (torch.randn(F_, F_, 1, 1, generator=g) / 16).to(DEV)
    gw, gb = P.pack_gate(w3), torch.randn(F_, generator=g).to(DEV) * 0.1
    ln = ((torch.rand(F_, generator=g) + 0.5).to(DEV), (torch.randn(F_, generator=g) * 0.1).to(DEV))
    res = P.Feat.from_nchw(torch.randn(n, F_, h, w, generator=g).to(DEV))

    def unit(x2):
        cat = P.Feat.alloc(n, h, w, cin, DEV)
        cat.x2 = x2
        P.roi_align(coarse, boxes, 1.0, h, w, out=cat.slice(F_, F_))
        out = P.conv2d(x, cw_c, cat.slice(0, F_), relu_in=True, res=x)                                  # GatedConvUnit.conv: conv(relu(x)) + x
        y = P.conv3x3_ln_gate(cat, cw_f, ln, gw, gb, act=P.ACT_RELU, mul=out, res=res)
        return cat, y, P.L.load().prv2_last_kernel().decode()

    cat32, y32, k32 = unit(False)
    catx2, yx2, kx2 = unit(True)
    assert ("x2" in kx2 or "w4" in kx2) and "x2" not in k32 and "w4" not in k32
    assert torch.equal(catx2.x2_to_float().buf, _x2_reconstruct(cat32.buf))
    assert bool(torch.isfinite(yx2.buf).all()) and torch.equal(yx2.buf, y32.buf)
    # the formats are only taken where they are implemented
    with pytest.raises((RuntimeError, AssertionError)):
        P.conv2d(catx2, cw_f)
    narrow = P.pack_conv(torch.randn(64, F_, 3, 3, generator=g).to(DEV), None, pad=1, prec=PR)
    dst = P.Feat.alloc(n, h, w, 64, DEV)
    dst.x2 = True
    with pytest.raises((RuntimeError, AssertionError)):
        P.conv2d(x, narrow, dst)


def _frame_boxes(split, P_hw, k_random, seed):
    """lr-frame ROI boxes [k, 4] (x1, y1, x2, y2) of tiles 1/split of the frame: the four frame corners, then random positions"""
    ph, pw = P_hw
    th, tw = ph / split[0], pw / split[1]
    rng = np.random.default_rng(seed)
    org = [(0.0, 0.0), (pw - tw, 0.0), (0.0, ph - th), (pw - tw, ph - th)]
    org += [(float(rng.uniform(0, pw - tw)), float(rng.uniform(0, ph - th))) for _ in range(k_random)]
    return torch.tensor([[x, y, x + tw, y + th] for x, y in org], dtype=torch.float32)


@pytest.mark.parametrize("case", [((4, 4), 24, 32, 32, 16), ((2, 2), 12, 16, 8, 8), ((4, 4), 13, 18, 16, 12), ((3, 5), 12, 20, 8, 4)])
def test_coarse_tap_gather_is_conv_of_the_roi(P, case):
    """csrc/coarse_taps.hip (prv2_coarse_tap_knots / prv2_coarse_tap_gather): the coarse half of a cat([fine, coarse_roi]) 3x3 conv
    from a per-frame tap table == conv3x3(roi_align(feat, boxes), W, zero padding) (bi_directional_fusion_model.py:70-73 over
    patchrefinerplus.py:263-283), for tiles in the frame's corners (clamped samples) and at random sub-pixel phases; the frame-level
    G is formed in float64 here so that the test isolates the two gather kernels"""
    split, H, W, cin, cout = case
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(1, cin, H, W, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    Pshape = (4 * H, 4 * W)  # patch_process_shape: spatial_scale = H / P_h
    boxes = _frame_boxes(split, Pshape, 6, 3)
    scale = H / Pshape[0]
    roi = o_ops.roi_align(feat, torch.cat([torch.zeros(len(boxes), 1), boxes], 1), (H, W), scale)
    ref = F.conv2d(roi.double(), wt.double(), padding=1).float()                                             # [k, cout, H, W]
    G = torch.einsum("oikl,ihw->hwklo", wt.double(), feat[0].double()).reshape(1, H, W, 9 * cout).float()   # tap-major
    taps = P.CoarseTaps(P.Feat(G.contiguous().to(DEV)), cout, (1.0 / split[0], 1.0 / split[1]))
    got = taps.gather(boxes.to(DEV), scale, H, W)
    close(got.to_nchw(), ref, 3e-6, "coarse_tap_gather")
    # the knot table alone: U on the knots of an interior pixel == the unmasked tap sum there (spot check of the table's layout)
    assert taps.v.buf.shape == (1, 3 * H, 3 * W, cout) and bool(torch.isfinite(taps.v.buf).all())


@pytest.mark.parametrize("prec", ["bf16x3"])
@pytest.mark.parametrize("case", [(5, 24, 32), (6, 17, 29)])
def test_gated_unit_with_coarse_taps_vs_fp32_reference(P, case, prec):
    """GatedConvUnit (bi_directional_fusion_model.py:56-82) restructured: conv3x3 over ``out`` only (K = F) + the per-frame coarse
    half as the gate kernel's pre-LayerNorm addend (prv2_conv3x3_ln_gate_pre) against the reference's order of operations in fp32 --
    roi_align -> cat -> conv3x3(2F -> F) -> LN -> ReLU -> 1x1 -> sigmoid gate (+ res) -- incl. tiles touching the frame border and
    random-phase boxes; and against the unrestructured HIP unit (same tolerance class)"""
    k, h, w = case
    F_ = 256
    PR = P.L.PREC_NAMES[prec]
    g = torch.Generator().manual_seed(17)
    split = (4, 4)
    boxes = _frame_boxes(split, (4 * h, 4 * w), k - 4, 9)
    scale = 0.25
    x_t = torch.randn(k, F_, h, w, generator=g)
    coarse_t = torch.randn(1, F_, h, w, generator=g)
    wc, bc = torch.randn(F_, F_, 3, 3, generator=g) / (3 * F_ ** 0.5), torch.randn(F_, generator=g) * 0.1
    wf, bf = torch.randn(F_, 2 * F_, 3, 3, generator=g) / (3 * (2 * F_) ** 0.5), torch.randn(F_, generator=g) * 0.1
    w3, gb = torch.randn(F_, F_, 1, 1, generator=g) / 16, torch.randn(F_, generator=g) * 0.1
    lnw, lnb = torch.rand(F_, generator=g) + 0.5, torch.randn(F_, generator=g) * 0.1
    res_t = torch.randn(k, F_, h, w, generator=g)
    # fp32 reference in the reference's order
    roi = o_ops.roi_align(coarse_t, torch.cat([torch.zeros(k, 1), boxes], 1), (h, w), scale)
    out_r = F.conv2d(F.relu(x_t), wc, bc, padding=1) + x_t
    fused = F.conv2d(torch.cat([out_r, roi], 1), wf, bf, padding=1)
    mu = fused.mean(1, keepdim=True)
    var = ((fused - mu) ** 2).mean(1, keepdim=True)
    fused = F.relu((fused - mu) / torch.sqrt(var + 1e-6) * lnw[None, :, None, None] + lnb[None, :, None, None])
    ref = out_r * torch.sigmoid(F.conv2d(fused, w3, gb)) + res_t
    # HIP: restructured unit
    x, coarse, res = P.Feat.from_nchw(x_t.to(DEV)), P.Feat.from_nchw(coarse_t.to(DEV)), P.Feat.from_nchw(res_t.to(DEV))
    cw_c = P.pack_conv(wc.to(DEV), bc.to(DEV), pad=1, prec=PR)
    cw_a = P.pack_conv(wf[:, :F_].contiguous().to(DEV), bf.to(DEV), pad=1, prec=PR)
    cw_t = P.pack_conv(P.coarse_tap_weight(wf[:, F_:]).to(DEV), None, prec=PR)
    gw = P.pack_gate(w3.to(DEV))
    ln = (lnw.to(DEV), lnb.to(DEV))
    taps = P.CoarseTaps(P.conv2d(coarse, cw_t), F_, (0.25, 0.25))
    out = P.Feat(torch.empty((k, h, w, F_), device=DEV), x2=True)
    P.conv2d(x, cw_c, out, relu_in=True, res=x)
    pre = taps.gather(boxes.to(DEV), scale, h, w)
    y = P.conv3x3_ln_gate(out, cw_a, ln, gw, gb.to(DEV), act=P.ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_)
    assert "gate_x2" in P.L.load().prv2_last_kernel().decode()
    close(y.to_nchw(), ref, 4e-5, "restructured GatedConvUnit vs fp32 reference")
    # HIP: the unit as the reference orders it (concat buffer + ROI gather), same arithmetic mode
    cat = P.Feat.alloc(k, h, w, 2 * F_, DEV)
    cat.x2 = True
    P.roi_align(coarse, boxes.to(DEV), scale, h, w, out=cat.slice(F_, F_))
    out2 = P.conv2d(x, cw_c, cat.slice(0, F_), relu_in=True, res=x)
    cw_f = P.pack_conv(wf.to(DEV), bf.to(DEV), pad=1, prec=PR)
    y2 = P.conv3x3_ln_gate(cat, cw_f, ln, gw, gb.to(DEV), act=P.ACT_RELU, mul=out2, res=res)
    close(y2.to_nchw(), ref, 4e-5, "concat-order GatedConvUnit vs fp32 reference")
    close(y.to_nchw(), y2.to_nchw().cpu(), 2e-5, "restructured vs concat-order unit")


@pytest.mark.parametrize("case", [(2, 24, 40, 32, 32, True), (2, 24, 32, 256, 64, True), (1, 17, 29, 256, 128, True), (2, 24, 32, 256, 256, True),
                                  (1, 19, 75, 96, 40, False), (2, 30, 70, 64, 130 + 2, False)])
def test_conv2d_pre_addend_in_front_of_the_epilogue(P, case):
    """prv2_conv2d_pre: y = act([LN](conv3x3(x) + pre + bias)) (+ res) on every kernel family that takes it (16x16x32 halo kernels at
    32 / 64 / 128 columns incl. ragged tiles and the remainder strip, the 256-column kernel) == the same conv over the concatenated
    input in fp32 -- fusion_layers_1[l](cat([c, f])) with the coarse half as the addend (bi_directional_fusion_model.py:424-426)"""
    n, h, w, cin, cout, ln = case
    PR = P.L.PREC_NAMES["bf16x3"]
    x_t, pre_t = rnd(1, n, cin, h, w), rnd(2, n, cout, h, w)
    wt, bias = rnd(3, cout, cin, 3, 3) / np.sqrt(9 * cin), rnd(4, cout) * 0.1
    lnw, lnb = torch.rand(cout, generator=torch.Generator().manual_seed(5)) + 0.5, rnd(6, cout) * 0.1
    res_t = None if ln else rnd(7, n, cout, h, w)
    v = F.conv2d(x_t, wt, bias, padding=1) + pre_t
    if ln:
        mu = v.mean(1, keepdim=True)
        v = (v - mu) / torch.sqrt(((v - mu) ** 2).mean(1, keepdim=True) + 1e-6) * lnw[None, :, None, None] + lnb[None, :, None, None]
    ref = F.gelu(v) + (res_t if res_t is not None else 0)
    cw = P.pack_conv(wt.to(DEV), bias.to(DEV), pad=1, prec=PR)
    assert P.conv2d_pre_supported(h, w, cw, ln)
    y = P.conv2d_pre(P.Feat.from_nchw(x_t.to(DEV)), cw, P.Feat.from_nchw(pre_t.to(DEV)), act=P.ACT_GELU, ln=(lnw.to(DEV), lnb.to(DEV)) if ln else None,
                     res=P.Feat.from_nchw(res_t.to(DEV)) if res_t is not None else None)
    close(y.to_nchw(), ref, 3e-5, f"conv2d_pre {case} on {P.L.load().prv2_last_kernel().decode()}")
    if y.ld != cout:  # pad channels behind cout stay zero
        assert float(y.buf[..., cout:].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2, 769, 4), (1, 197, 3), (3, 130, 2)])
def test_attention_score_bias_rows_and_packed_image(P, shape):
    """softmax(q k^T / 8 + bias[head]) v (BEiT relative position bias; midas_beit.py / beit.py:_get_rel_pos_bias) in the bf16x3 kernel:
    the bias as rows [heads, N, ld] and as the prv2_pack_attention_bias image (coalesced loads, log2 e pre-multiplied) give the SAME
    bits -- fp32 and split-swizzled outputs alike -- and agree with the fp32 reference"""
    b, n, heads = shape
    g = torch.Generator().manual_seed(21)
    qkv = torch.randn(b * n, 3 * heads * 64, generator=g)
    ld = (n + 63) // 64 * 64
    bias = torch.zeros(heads, n, ld)
    bias[:, :, :n] = torch.randn(heads, n, n, generator=g) * 2.0
    q, k, v = qkv.view(b, n, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125 + bias[None, :, :, :n], -1) @ v).permute(0, 2, 1, 3).reshape(b * n, heads * 64)
    qd, bd = qkv.to(DEV), bias.to(DEV)
    img = P.pack_attention_bias(bd, n)
    a_rows = P.attention(qd, b, n, heads, P.L.PREC_BF16X3, bias=bd)
    a_img = P.attention(qd, b, n, heads, P.L.PREC_BF16X3, bias=img)
    assert torch.equal(a_rows, a_img)
    assert torch.equal(P.attention(qd, b, n, heads, P.L.PREC_BF16X3, bias=bd, out_ss=True), P.attention(qd, b, n, heads, P.L.PREC_BF16X3, bias=img, out_ss=True))
    close(a_img, ref, 2e-5, "attention + bias image")
    close(P.attention(qd, b, n, heads, P.L.PREC_F32, bias=bd), ref, 2e-6, "attention + bias rows, f32")
    with pytest.raises((RuntimeError, AssertionError)):
        P.attention(qd, b, n, heads, P.L.PREC_F32, bias=img)   # the image is for the bf16 modes
