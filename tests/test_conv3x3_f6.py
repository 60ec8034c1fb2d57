"""prv2_conv3x3_f6 (csrc/conv3x3_f6.hip): the 256-column 3x3 conv in the fp16 + block-scaled-fp6 arithmetic against float64 torch, beside
the bf16x3 kernel on the same inputs.  Tolerances: the scheme's rms error per dot product is 1.2e-5 (profiles/r03_f16f6_study.txt);
rel-L2 of a layer's output <= 4e-5 is asserted, the bf16x3 kernel's number is printed next to it."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from patchrefinerv2_amd import ops
    return ops


def _ref(x, w, b, res, relu_in):
    xi = torch.relu(x.double()) if relu_in else x.double()
    y = torch.nn.functional.conv2d(xi.permute(0, 3, 1, 2), w.double(), b.double() if b is not None else None, padding=1).permute(0, 2, 3, 1)
    return y + (res.double() if res is not None else 0)


def _x2_decode(buf):  # [.., C] float32 holding per 8 channels [8 bf16 hi | 8 bf16 lo] -> hi + lo as float64
    n, h, w, c = buf.shape
    u = buf.contiguous().view(torch.int16).view(n, h, w, c // 8, 2, 8).to(torch.int32) << 16
    f = u.view(torch.float32).double()
    return (f[..., 0, :] + f[..., 1, :]).reshape(n, h, w, c)


# (5 x 64 x 128 = 320 tiles > the 256 persistent workgroups: some walk two tiles -- the next tile's halo staged under the current one, weights wrapping around)
@pytest.mark.parametrize("n,h,w,cin", [(2, 24, 32, 256), (1, 19, 37, 256), (3, 8, 16, 128), (1, 40, 48, 512), (1, 9, 16, 64), (5, 64, 128, 256), (3, 100, 125, 128)])
@pytest.mark.parametrize("relu_in,with_res,with_bias", [(True, True, True), (False, False, False)])
def test_conv3x3_f6_vs_fp64(n, h, w, cin, relu_in, with_res, with_bias):
    P = _ops()
    g = torch.Generator(device=DEV).manual_seed(n * 1000 + h * 10 + cin)
    x = P.Feat(torch.randn(n, h, w, cin, device=DEV, generator=g))
    wt = torch.randn(256, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(256, device=DEV, generator=g) if with_bias else None
    res = P.Feat(torch.randn(n, h, w, 256, device=DEV, generator=g)) if with_res else None
    assert P.conv3x3_f6_supported(x, 256, cin)
    cw = P.pack_conv3x3_f6(wt, b)
    out = P.conv3x3_f6(x, cw, relu_in=relu_in, res=res)
    kernel = P.L.load().prv2_last_kernel().decode()
    assert kernel == "conv3x3_c256_f6_kernel<256,f16f6>", kernel
    ref = _ref(x.buf, wt, b, res.buf if res is not None else None, relu_in)
    err = float((out.buf.double() - ref).norm() / ref.norm())
    cwb = P.pack_conv(wt, b, pad=1, prec=P.L.PREC_NAMES["bf16x3"])
    outb = P.conv2d(x, cwb, relu_in=relu_in, res=res)
    errb = float((outb.buf.double() - ref).norm() / ref.norm())
    print(f"\n{n}x{h}x{w} {cin}->256 relu_in={relu_in}: f16f6 rel-L2 {err:.2e} (bf16x3 {errb:.2e}), {kernel}")
    assert err < 4e-5, err
    seen = torch.tensor([int(cw.range.item())], dtype=torch.int32).view(torch.float32).item()
    xi = torch.relu(x.buf) if relu_in else x.buf
    assert abs(seen - float(xi.abs().max())) <= 1e-6 * seen, (seen, float(xi.abs().max()))


def test_conv3x3_f6_x2_output():
    P = _ops()
    g = torch.Generator(device=DEV).manual_seed(5)
    x = P.Feat(torch.randn(2, 16, 32, 256, device=DEV, generator=g))
    wt = torch.randn(256, 256, 3, 3, device=DEV, generator=g) / 48
    b = torch.randn(256, device=DEV, generator=g)
    cw = P.pack_conv3x3_f6(wt, b)
    o32 = P.conv3x3_f6(x, cw, relu_in=True, res=x)
    o2 = P.Feat(torch.empty(2, 16, 32, 256, device=DEV), x2=True)
    P.conv3x3_f6(x, cw, o2, relu_in=True, res=x)
    dec = _x2_decode(o2.buf)
    # hi + lo of the RNE bf16 split of the fp32 result: 16 mantissa bits
    assert float((dec - o32.buf.double()).abs().max()) <= 2.0 ** -16 * float(o32.buf.abs().max())


def test_conv3x3_f6_range_and_scales():
    """activations far outside fp16's range: with the caller's power-of-two x_scale the result is as accurate as at unit scale; without it
    the fp16 part saturates (finite, never inf / nan) and the range word tells"""
    P = _ops()
    g = torch.Generator(device=DEV).manual_seed(9)
    base = torch.randn(1, 16, 32, 256, device=DEV, generator=g)
    wt = torch.randn(256, 256, 3, 3, device=DEV, generator=g) / 48
    for mag, x_scale in [(1e5, 2.0 ** -14), (1e-7, 2.0 ** 23), (1e5, 1.0)]:
        x = P.Feat(base * mag)
        cw = P.pack_conv3x3_f6(wt * 1e-3, None)  # (small weights: the weight scale is exercised too)
        cw.x_scale = x_scale
        out = P.conv3x3_f6(x, cw, relu_in=True)
        ref = _ref(x.buf, wt * 1e-3, None, None, True)
        err = float((out.buf.double() - ref).norm() / ref.norm())
        seen = torch.tensor([int(cw.range.item())], dtype=torch.int32).view(torch.float32).item()
        print(f"\n|x| ~ {mag:g}, x_scale 2^{torch.log2(torch.tensor(x_scale)).item():.0f}: rel-L2 {err:.2e}, max |x x_scale| seen {seen:.3g}")
        assert torch.isfinite(out.buf).all()
        if x_scale != 1.0:
            assert err < 4e-5, err
        else:
            assert seen > 65504.0  # the monitor reports the overflow; the result is finite but only fp6-grade


def _heavy_tail_scales(c, g, outliers=4):
    """per-channel scales: log-uniform over 1e-3 .. 1e3 INSIDE every 32-channel block (one E8M0 scale per block serves all of them), plus a
    handful of 1e4 channels"""
    s = 10.0 ** (torch.rand(c, generator=g) * 6.0 - 3.0)
    s[torch.randperm(c, generator=g)[:outliers]] = 1e4
    return s


@pytest.mark.parametrize("relu_in", [True, False])
def test_conv3x3_f6_heavy_tailed_channels_vs_fp64(relu_in):
    """VERDICT r05 weak #1b: activations whose channels span seven decades inside one 32-channel block (a real checkpoint's few 1e3-magnitude
    channels beside 1e-2 ones), with a handful of 1e4 outliers -- not the uniform scale of test_conv3x3_f6_range_and_scales.  The scheme keeps
    every channel's fp16 part (11 bits, its own exponent); what the shared block scale costs is the fp6 CORRECTION terms of the small channels,
    i.e. errors of 2^-11 relative to contributions that are themselves 1e-3..1e-7 of the output.  Measured against float64 beside bf16x3; the
    calibrated x_scale (what ops.F6Range sets from the range word) is applied as models.forward would."""
    import math
    P = _ops()
    g = torch.Generator().manual_seed(11)
    n, h, w, cin = 2, 24, 32, 256
    sc = _heavy_tail_scales(cin, g)
    x = P.Feat((torch.randn(n, h, w, cin, generator=g) * sc).to(DEV))
    wt = (torch.randn(256, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    cw = P.pack_conv3x3_f6(wt, b)
    P.conv3x3_f6(x, cw, relu_in=relu_in)                       # a first frame: the range word
    seen = torch.tensor([int(cw.range.item())], dtype=torch.int32).view(torch.float32).item()
    cw.x_scale = 2.0 ** (P.F6Range.TARGET_LOG2 - math.floor(math.log2(seen)))
    out = P.conv3x3_f6(x, cw, relu_in=relu_in)
    ref = _ref(x.buf, wt, b, None, relu_in)
    err = float((out.buf.double() - ref).norm() / ref.norm())
    emax = float((out.buf.double() - ref).abs().max() / ref.abs().max())
    outb = P.conv2d(x, P.pack_conv(wt, b, pad=1, prec=P.L.PREC_NAMES["bf16x3"]), relu_in=relu_in)
    errb = float((outb.buf.double() - ref).norm() / ref.norm())
    # the same input with ONLY its small channels (|scale| < 1): what the block scale does to them when nothing large sits beside them in the sum
    small = (sc < 1.0).to(DEV)
    xs = P.Feat(x.buf * small)
    refs = _ref(xs.buf, wt, None, None, relu_in)
    cw2 = P.pack_conv3x3_f6(wt, None)
    cw2.x_scale = cw.x_scale                                   # (the layer's scale is set by the outliers: the small channels live with it)
    outs = P.conv3x3_f6(xs, cw2, relu_in=relu_in)
    errs = float((outs.buf.double() - refs).norm() / refs.norm())
    print(f"\nheavy-tailed channels (1e-3..1e3 per 32-block + 4 x 1e4), relu_in={relu_in}: f16f6 rel-L2 {err:.2e} max/scale {emax:.2e} (bf16x3 {errb:.2e}); "
          f"small channels alone under the outliers' x_scale 2^{math.log2(cw.x_scale):.0f}: rel-L2 {errs:.2e}")
    assert torch.isfinite(out.buf).all() and err < 4e-5, err
    assert errs < 2e-3, errs  # (fp16 subnormals below 2^-14 / x_scale: 1e-3-scale channels beside 1e4 ones keep ~10 bits -- reported, bounded)


@pytest.mark.parametrize("mag", [1.0, 1e5, 1e-7, "heavy"])
def test_fusion_network_f16f6_range_guard(mag):
    """BiDirectionalFusion with its 256-channel GatedConvUnit convs in the fp16 + fp6 arithmetic against the same network in bf16x3, with the
    refiner features scaled far outside fp16's range: the guard (ops.F6Range: what models.forward runs after every frame) reports the
    layers, moves their power-of-two input scales, and the recomputed result is as close to bf16x3 as at unit scale -- never an inf."""
    import numpy as np
    from oracle.cases import TINY_BIDIR
    from patchrefinerv2_amd import ops, weights as W
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    c = TINY_BIDIR
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"]), seed=c["seed"])
    sizes = [(64, 96), (32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]
    Ph, Pw, K = 64, 96, 3
    rng = np.random.default_rng(4)
    th, tw = Ph / 2, Pw / 2
    org = [(0.0, 0.0), (Pw - tw, Ph - th)] + [(float(rng.uniform(0, Pw - tw)), float(rng.uniform(0, Ph - th))) for _ in range(K - 2)]
    boxes = torch.tensor([[x, y, x + tw, y + th] for x, y in org], dtype=torch.float32)
    rnd = lambda seed, *shape: torch.randn(*shape, generator=torch.Generator().manual_seed(seed))  # noqa: E731
    frame = [rnd(40 + l, 1, ch, *sizes[l]) for l, ch in enumerate(c["coarse_chl"])]
    if mag == "heavy":  # ("heavy": per-channel scales over seven decades inside every 32-channel block + four 1e4 channels per level -- VERDICT r05 weak #1b)
        hg = torch.Generator().manual_seed(77)
        fine = [None] + [rnd(50 + l, K, ch, *sizes[l]) * _heavy_tail_scales(ch, hg).view(1, ch, 1, 1) for l, ch in list(enumerate([32] + c["fine_chl"]))[1:]]
    else:
        fine = [None] + [rnd(50 + l, K, ch, *sizes[l]) * mag for l, ch in list(enumerate([32] + c["fine_chl"]))[1:]]
    pred1 = torch.rand(K, 1, *sizes[0], generator=torch.Generator().manual_seed(7)) * 10

    def run(prec):
        m = BiDirectionalFusion(coarse2fine_type="coarse-gated", coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                                fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec=prec)
        m.load_state_dict(sd)

        def fwd():
            fr = [ops.Feat.from_nchw(t.to(DEV)) for t in frame]
            m.prepare_frame(fr, (0.5, 0.5))
            rois = [ops.RoiSource(f, boxes.to(DEV), f.h / Ph, f.h, f.w) for f in fr]
            ff = [None] + [ops.Feat.from_nchw(t.to(DEV)) for t in fine[1:]]
            return m(rois, ff, pred1.to(DEV), torch.zeros_like(pred1).to(DEV), update_base=pred1.to(DEV), f_sizes=sizes).clone()
        out = fwd()
        redo = []
        if prec == "f16f6":
            n6 = sum("conv_f6" in m._packed["refine"][r][u] for r in range(1, 6) for u in ("u1", "u2"))
            assert n6 == 10, n6
            redo = ops.F6Range.check(DEV)
            if redo:
                assert torch.isfinite(out).all()  # (saturated fp16 parts: finite, imprecise)
                out = fwd()
                assert not ops.F6Range.check(DEV)
        return out, redo

    ref, _ = run("bf16x3")
    got, redo = run("f16f6")
    err = float((got.double() - ref.double()).abs().max()) / max(1.0, float(ref.abs().max()))
    print(f"\n|fine features| x {mag}: {len(redo)} layer(s) recomputed, max |f16f6 - bf16x3| / scale = {err:.2e}")
    if mag != "heavy":
        assert (len(redo) > 0) == (mag != 1.0), [(m_, s_) for _, m_, s_ in redo]
    assert torch.isfinite(got).all() and err <= 3e-5, err


@pytest.mark.parametrize("case", [(2, 24, 32), (1, 19, 37), (3, 8, 16)])
@pytest.mark.parametrize("with_pre,with_res", [(True, True), (False, False)])
def test_gate_unit_tail_f6_vs_fp64_and_bf16x3(case, with_pre, with_res):
    """prv2_conv3x3_ln_gate_f6 (conv3x3_c256_gate_f6_kernel): the GatedConvUnit tail -- conv3x3 over the unit's pre-split ``out`` (+ pre), LayerNorm, ReLU,
    256 x 256 gate, sigmoid, x out (+ res) -- with the 3x3 conv in fp16 + fp6, against float64 and against the bf16x3 kernel on the same X2 buffer"""
    P = _ops()
    n, h, w = case
    F_ = 256
    g = torch.Generator(device=DEV).manual_seed(31 + h)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    o32 = rnd(n, h, w, F_)
    # the unit's ``out`` as its producer writes it: pre-split, i.e. exactly hi + lo of the bf16 split
    out = P.Feat(torch.empty(n, h, w, F_, device=DEV), x2=True)
    ident = torch.zeros(F_, F_, 3, 3, device=DEV)
    ident[torch.arange(F_), torch.arange(F_), 1, 1] = 1.0
    P.conv2d(P.Feat(o32), P.pack_conv(ident, None, pad=1, prec=P.L.PREC_NAMES["bf16x3"]), out)           # (identity conv: the X2 writer of the 256-column kernel)
    ov = _x2_decode(out.buf)                                                                              # float64 view of what the kernels read
    wt = rnd(F_, F_, 3, 3) / 48
    b = rnd(F_) * 0.1
    w3 = rnd(F_, F_, 1, 1) / 16
    gw, gb = P.pack_gate(w3), rnd(F_) * 0.1
    ln = (torch.rand(F_, device=DEV, generator=g) + 0.5, rnd(F_) * 0.1)
    pre = P.Feat(rnd(n, h, w, F_) * 0.5) if with_pre else None
    res = P.Feat(rnd(n, h, w, F_)) if with_res else None
    cw6 = P.pack_conv3x3_f6(wt, b)
    y6 = P.conv3x3_ln_gate_f6(out, cw6, ln, gw, gb, mul=out, res=res, pre=pre, pre_cin=F_)
    assert P.L.load().prv2_last_kernel().decode() == "conv3x3_c256_gate_f6_kernel<256,f16f6>"
    cwb = P.pack_conv(wt, b, pad=1, prec=P.L.PREC_NAMES["bf16x3"])
    yb = P.conv3x3_ln_gate(out, cwb, ln, gw, gb, mul=out, res=res, pre=pre, pre_cin=F_)
    t = torch.nn.functional.conv2d(ov.permute(0, 3, 1, 2), wt.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    if pre is not None:
        t = t + pre.buf.double()
    mu = t.mean(-1, keepdim=True)
    t = (t - mu) / torch.sqrt(((t - mu) ** 2).mean(-1, keepdim=True) + 1e-6) * ln[0].double() + ln[1].double()
    t = torch.relu(t)
    gt = torch.einsum("nhwc,oc->nhwo", t, w3[:, :, 0, 0].double()) + gb.double()
    ref = ov * torch.sigmoid(gt) + (res.buf.double() if res is not None else 0)
    e6 = float((y6.buf.double() - ref).norm() / ref.norm())
    eb = float((yb.buf.double() - ref).norm() / ref.norm())
    print(f"\ngate unit tail {n}x{h}x{w} pre={with_pre} res={with_res}: f16f6 rel-L2 {e6:.2e} (bf16x3 {eb:.2e})")
    assert torch.isfinite(y6.buf).all() and e6 < 2e-5, (e6, eb)


@pytest.mark.parametrize("x2", [False, True])
def test_gate_unit_tail_f6_concat_form(x2):
    """the same tail kernel over the unit's whole [out | coarse ROI] concat (K = 512: the configs whose ROI gather resizes and therefore take no tap
    tables), fp32 or pre-split, mul = the concat's first half -- against float64 and the bf16x3 kernel"""
    P = _ops()
    n, h, w, F_ = 2, 24, 32, 256
    g = torch.Generator(device=DEV).manual_seed(77)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    c32 = rnd(n, h, w, 2 * F_)
    cat = P.Feat(c32.clone())
    if x2:  # the concat as its producers write it in the X2 format: identity convs per half through the 256-column kernel's X2 writer
        cat = P.Feat(torch.empty(n, h, w, 2 * F_, device=DEV), x2=True)
        ident = torch.zeros(F_, F_, 3, 3, device=DEV)
        ident[torch.arange(F_), torch.arange(F_), 1, 1] = 1.0
        cwi = P.pack_conv(ident, None, pad=1, prec=P.L.PREC_NAMES["bf16x3"])
        for half in (0, 1):
            P.conv2d(P.Feat(c32[..., half * F_:(half + 1) * F_].contiguous()), cwi, cat.slice(half * F_, F_))
    out = cat.slice(0, F_)
    hi = c32.bfloat16().float()
    cv = (hi + (c32 - hi).bfloat16().float()).double()            # what the kernels see of an X2 buffer / take of ``mul`` in either format
    xin = cv if x2 else c32.double()
    wt = rnd(F_, 2 * F_, 3, 3) / (3 * (2 * F_) ** 0.5)
    b = rnd(F_) * 0.1
    w3 = rnd(F_, F_, 1, 1) / 16
    gw, gb = P.pack_gate(w3), rnd(F_) * 0.1
    ln = (torch.rand(F_, device=DEV, generator=g) + 0.5, rnd(F_) * 0.1)
    res = P.Feat(rnd(n, h, w, F_))
    assert P.conv3x3_f6_supported(cat, F_, 2 * F_, allow_x2=True)
    y6 = P.conv3x3_ln_gate_f6(cat, P.pack_conv3x3_f6(wt, b), ln, gw, gb, mul=out, res=res)
    assert P.L.load().prv2_last_kernel().decode() == "conv3x3_c256_gate_f6_kernel<256,f16f6>"
    yb = P.conv3x3_ln_gate(cat, P.pack_conv(wt, b, pad=1, prec=P.L.PREC_NAMES["bf16x3"]), ln, gw, gb, mul=out, res=res)
    t = torch.nn.functional.conv2d(xin.permute(0, 3, 1, 2), wt.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    mu = t.mean(-1, keepdim=True)
    t = torch.relu((t - mu) / torch.sqrt(((t - mu) ** 2).mean(-1, keepdim=True) + 1e-6) * ln[0].double() + ln[1].double())
    ref = cv[..., :F_] * torch.sigmoid(torch.einsum("nhwc,oc->nhwo", t, w3[:, :, 0, 0].double()) + gb.double()) + res.buf.double()
    e6, eb = float((y6.buf.double() - ref).norm() / ref.norm()), float((yb.buf.double() - ref).norm() / ref.norm())
    print(f"\ngate unit tail, concat form K = 512, x2 = {x2}: f16f6 rel-L2 {e6:.2e} (bf16x3 {eb:.2e})")
    assert torch.isfinite(y6.buf).all() and e6 < 2e-5, (e6, eb)


def test_conv3x3_f6_matches_the_documented_arithmetic():
    """The kernel against a HOST EMULATION of exactly the arithmetic include/prv2.h documents -- x w = f16(x) f16(w) + q6(x) q6(w - f16 w) + q6(x - f16 x) q6(w), fp16 RNE,
    e2m3 with one power-of-two scale 2^(floor(log2 max) - 2) per 32 channels (tools/studies/split_arith_study.py's quantisers, which the instruction probe
    tools/probes/f16f6_probe.hip matched element for element), products and sums exact in float64: what remains is the kernel's fp32 accumulation order and the
    quantiser's tie rule (the host takes the lower neighbour, the instruction the even one): below 1e-6, several times under the scheme's own error.  A wrong block pairing, a misplaced scale or a swapped correction would show here, not in the 4e-5 float64 tolerance."""
    import importlib.util
    import os
    import numpy as np
    P = _ops()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("split_arith_study", os.path.join(root, "tools", "studies", "split_arith_study.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)
    n, h, w, cin = 1, 9, 17, 128
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g) * torch.exp(torch.randn(n, h, w, 1, device=DEV, generator=g))   # (per-pixel magnitudes over a decade)
    wt = torch.randn(256, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(256, device=DEV, generator=g)
    res = torch.randn(n, h, w, 256, device=DEV, generator=g)
    cw = P.pack_conv3x3_f6(wt, b)
    out = P.conv3x3_f6(P.Feat(x), cw, relu_in=True, res=P.Feat(res)).buf.double().cpu().numpy()
    q6 = lambda v: S.mx_quant(v, S.E2M3, 2).astype(np.float64)  # noqa: E731  (blocks of 32 along the last axis)
    xs = np.maximum(x.cpu().numpy(), 0).astype(np.float32)                                   # relu(x) * x_scale (1.0)
    ws = (wt.cpu().numpy() * np.float32(cw.w_scale)).transpose(0, 2, 3, 1).copy()           # [cout, ky, kx, cin]: channels last
    xh, wh = S.f16(xs), S.f16(ws)
    parts_x = (xh.astype(np.float64), q6(xs), q6(xs - xh))
    parts_w = (wh.astype(np.float64), q6(ws - wh), q6(ws))
    xp = [np.pad(p, ((0, 0), (1, 1), (1, 1), (0, 0))) for p in parts_x]
    y = np.zeros((n, h, w, 256))
    for ky in range(3):
        for kx in range(3):
            for px, pw in zip(xp, parts_w):
                y += px[:, ky:ky + h, kx:kx + w] @ pw[:, ky, kx].T
    y = y / cw.w_scale + b.double().cpu().numpy() + res.double().cpu().numpy()
    err = float(np.sqrt(((out - y) ** 2).mean()) / np.sqrt((y ** 2).mean()))
    exact = torch.nn.functional.conv2d(torch.relu(x.double()).permute(0, 3, 1, 2), wt.double(), b.double(), padding=1).permute(0, 2, 3, 1) + res.double()
    scheme = float(np.sqrt(((y - exact.cpu().numpy()) ** 2).mean()) / np.sqrt((y ** 2).mean()))
    print(f"\nkernel vs host emulation of the documented arithmetic: rms {err:.2e}; the arithmetic itself vs float64: {scheme:.2e}")
    assert err < 2e-6 and err * 3 < scheme, (err, scheme)
