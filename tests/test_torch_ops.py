"""torch.ops.prv2.* -- the PyTorch-ROCm custom-op surface (TORCH_LIBRARY(prv2), patchrefinerv2_amd/csrc/torch_ops.cpp).
CPU: the library loads, every op of the SURVEY.md 8(b) minimum set is registered with its schema, CPU tensors are rejected.
GPU: each op through torch.ops against the oracle (same tolerances as the C-ABI tests)."""
import pytest
import torch

from oracle import ops as o_ops
from oracle.cases import rand_image, randn

torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def T():
    from patchrefinerv2_amd import torch_ops
    return torch_ops.load(), torch_ops


def test_library_loads_and_registers_every_op(T):
    ops, mod = T
    assert ops.abi_version() == mod.ABI_VERSION
    for name in mod.OPS:
        assert hasattr(ops, name), name
        getattr(ops, name).default._schema  # registered with a schema
    s = str(ops.conv2d.default._schema)
    for frag in ("Tensor x", "Tensor w_packed", "Tensor? ln_weight", "Tensor? mul", "int prec", "Tensor(a!)? out"):
        assert frag in s, (frag, s)


def test_cpu_tensors_are_rejected(T):
    ops, _ = T
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.layernorm(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))
    with pytest.raises((NotImplementedError, RuntimeError)):
        ops.upsample_bilinear_ac(torch.zeros(1, 4, 4, 8), 8, 8)


gpu = pytest.mark.gpu
DEV = "cuda"


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def _close(got, ref, tol=2e-5):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert float((got - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max())), float((got - ref).abs().max())


@gpu
@pytest.mark.parametrize("prec", [0, 1])
def test_conv2d_fused_epilogue_and_slices(T, prec):
    """3x3 conv + bias + channels-first LayerNorm + GELU written into a channel slice of a wider buffer (an in-place torch.cat),
    then a 1x1 gate: out * sigmoid(conv1x1(.)) + residual -- the GatedConvUnit pattern (bi_directional_fusion_model.py:56-82)"""
    import torch.nn.functional as F
    ops, mod = T
    x = randn(1, 2, 34, 20, 40)
    w, b = randn(2, 64, 34, 3, 3) / 17.0, randn(3, 64) * 0.1
    lw, lb = 1 + 0.1 * randn(4, 64), 0.1 * randn(5, 64)
    xd = _nhwc(x)
    cat = torch.zeros(2, 20, 40, 96, device=DEV)
    wp = ops.pack_conv_weight(w.to(DEV), None, 0, prec)
    y = ops.conv2d(xd, wp, b.to(DEV), 64, 3, 3, pad=1, act=mod.ACT_GELU, ln_weight=lw.to(DEV), ln_bias=lb.to(DEV), prec=prec,
                   out=cat[..., 32:])
    assert y.data_ptr() == cat[..., 32:].data_ptr()
    c = F.conv2d(x, w, b, padding=1)
    u = c.mean(1, keepdim=True)
    s = (c - u).pow(2).mean(1, keepdim=True)
    ref = F.gelu(lw.view(1, -1, 1, 1) * ((c - u) / torch.sqrt(s + 1e-6)) + lb.view(1, -1, 1, 1))
    _close(_nchw(cat[..., 32:]), ref, 3e-5)
    assert float(cat[..., :32].abs().max()) == 0.0  # the neighbouring slice is untouched
    w1 = randn(6, 64, 64, 1, 1) / 8.0
    g = ops.conv2d(y, ops.pack_conv_weight(w1.to(DEV), None, 0, prec), None, 64, 1, 1, act=mod.ACT_SIGMOID, mul=y, res=y, prec=prec)
    _close(_nchw(g), ref * torch.sigmoid(F.conv2d(ref, w1)) + ref, 5e-5)
    with pytest.raises(RuntimeError):
        ops.conv2d(xd, wp, b.to(DEV), 32, 3, 3, pad=1, prec=prec)  # cout does not match the packed weight: TORCH_CHECK


@gpu
def test_conv3x3_ln_gate_op(T):
    """GatedConvUnit tail as ONE op (bi_directional_fusion_model.py:44-51,70-80): out * sigmoid(conv1x1(relu(LN(conv3x3(cat))))) + xs[0],
    written into a channel slice; == the three-op sequence of the test above at 256 channels"""
    import torch.nn.functional as F
    ops, mod = T
    x = randn(1, 2, 64, 12, 32)
    w, b = randn(2, 256, 64, 3, 3) / 24.0, randn(3, 256) * 0.1
    lw, lb = 1 + 0.1 * randn(4, 256), 0.1 * randn(5, 256)
    w1, b1 = randn(6, 256, 256, 1, 1) / 16.0, randn(7, 256) * 0.1
    mul, res = randn(8, 2, 256, 12, 32), randn(9, 2, 256, 12, 32)
    c = F.conv2d(x, w, b, padding=1)
    u = c.mean(1, keepdim=True)
    sd = (c - u).pow(2).mean(1, keepdim=True)
    fused = F.relu(lw.view(1, -1, 1, 1) * ((c - u) / torch.sqrt(sd + 1e-6)) + lb.view(1, -1, 1, 1))
    ref = mul * torch.sigmoid(F.conv2d(fused, w1, b1)) + res
    buf = torch.zeros(2, 12, 32, 256 + 32, device=DEV)
    wp, gp = ops.pack_conv_weight(w.to(DEV), None, 0, mod.PREC_BF16X3), ops.pack_gate_weight(w1.to(DEV))
    y = ops.conv3x3_ln_gate(_nhwc(x), wp, b.to(DEV), lw.to(DEV), lb.to(DEV), gp, b1.to(DEV), _nhwc(mul), _nhwc(res), act=mod.ACT_RELU,
                            prec=mod.PREC_BF16X3, out=buf[..., 32:])
    assert y.data_ptr() == buf[..., 32:].data_ptr() and float(buf[..., :32].abs().max()) == 0.0
    _close(_nchw(y), ref, 3e-5)
    _close(_nchw(ops.conv3x3_ln_gate(_nhwc(x), wp, b.to(DEV), lw.to(DEV), lb.to(DEV), act=mod.ACT_RELU, prec=mod.PREC_BF16X3)), fused, 3e-5)
    with pytest.raises(RuntimeError, match="not covered"):
        ops.conv3x3_ln_gate(_nhwc(x)[:, :, :12].contiguous(), wp, b.to(DEV), lw.to(DEV), lb.to(DEV))  # narrower than a tile


@gpu
def test_conv_transpose_and_linear(T):
    import torch.nn.functional as F
    ops, _ = T
    x = randn(7, 1, 48, 9, 11)
    w, b = randn(8, 48, 24, 2, 2) / 7.0, randn(9, 24) * 0.1
    y = ops.conv2d(_nhwc(x), ops.pack_conv_weight(w.to(DEV), None, 2, 0), b.to(DEV), 24, 2, 2, convt_k=2)
    _close(_nchw(y), F.conv_transpose2d(x, w, b, stride=2))
    rows, wl = randn(10, 300, 96), randn(11, 160, 96) / 10.0
    z = ops.conv2d(rows.to(DEV).view(1, 300, 1, 96), ops.pack_conv_weight(wl.to(DEV)), None, 160, 1, 1)
    _close(z.view(300, 160), rows @ wl.t())


@gpu
def test_layernorm_and_attention(T):
    import torch.nn.functional as F
    ops, mod = T
    x, w, b = randn(12, 77, 384), 1 + 0.1 * randn(13, 384), 0.1 * randn(14, 384)
    _close(ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-6, mod.ACT_NONE), F.layer_norm(x, (384,), w, b, 1e-6), 1e-5)
    B, N, H = 2, 197, 6
    qkv = randn(15, B * N, 3 * H * 64)
    q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q * 64 ** -0.5) @ k.transpose(-2, -1), dim=-1) @ v).transpose(1, 2).reshape(B * N, H * 64)
    for prec, tol in ((0, 2e-5), (1, 2e-4)):
        _close(ops.attention_fwd(qkv.to(DEV), B, N, H, prec), ref, tol)


@gpu
def test_crop_resize_roi_pyramid_upsample(T):
    ops, _ = T
    img = rand_image(16, 1, 108, 192)[0]
    tiles = torch.tensor([[0, 0], [27, 48], [54, 96]], dtype=torch.int32)
    out = ops.crop_resize_bilinear(img.to(DEV), tiles.to(DEV), 54, 96, 56, 84, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    for i, (h, w) in enumerate(tiles.tolist()):
        ref = (o_ops.resize_da(img[None, :, h:h + 54, w:w + 96], 84, 56, 14) - mean) / std
        _close(_nchw(out[i:i + 1]), ref, 2e-6)
    feats = [randn(17 + i, 1, c, h, w) for i, (c, h, w) in enumerate(((32, 56, 84), (256, 28, 42), (256, 4, 6)))]
    boxes = torch.tensor([[0.0, 0.0, 42.0, 28.0], [21.0, 14.0, 63.0, 42.0]])
    outs = ops.roi_gather_pyramid([_nhwc(f) for f in feats], boxes.to(DEV), 56)
    for f, o in zip(feats, outs):
        bf = torch.cat([torch.arange(2.0).view(2, 1), boxes], dim=1)
        ref = o_ops.roi_align(f.repeat(2, 1, 1, 1), bf, f.shape[-2:], f.shape[-2] / 56, aligned=True)
        _close(_nchw(o), ref, 2e-6)
    x = randn(20, 2, 24, 12, 16)
    _close(_nchw(ops.upsample_bilinear_ac(_nhwc(x), 24, 32)), o_ops.bilinear_ac(x, (24, 32)), 2e-6)


@gpu
def test_blend_ops_sequence(T):
    """blend_init / blend_update / blend_resize == RunningAverageMap (estimator/models/utils.py:22-49), bit for bit"""
    from oracle.tiling import RunningAverageMap
    ops, _ = T
    g = torch.Generator().manual_seed(21)
    ph, pw, H, Wd = 12, 16, 24, 32
    mask = torch.rand(ph, pw, generator=g)
    mask[:2] = 0
    preds = torch.rand(5, ph, pw, generator=g) * 10
    avg, cnt = torch.zeros(H, Wd, device=DEV), torch.zeros(H, Wd, device=DEV)
    t0 = torch.tensor([[0, 0], [0, 16], [12, 0], [12, 16]], dtype=torch.int32)
    ops.blend_init(avg, cnt, preds[:4].to(DEV), mask.to(DEV), t0.to(DEV), ph, pw)
    pd, ct = torch.zeros(H, Wd), torch.zeros(H, Wd)
    for i, (h, w) in enumerate(t0.tolist()):
        pd[h:h + ph, w:w + pw], ct[h:h + ph, w:w + pw] = preds[i], mask
    ram = RunningAverageMap(pd, ct)
    ops.blend_update(avg, cnt, preds[4:].to(DEV), mask.to(DEV), torch.tensor([[6, 8]], dtype=torch.int32).to(DEV), ph, pw)
    pd, ct = torch.zeros(H, Wd), torch.zeros(H, Wd)
    pd[6:18, 8:24], ct[6:18, 8:24] = preds[4], mask
    ram.update(pd, ct)
    assert torch.equal(avg.cpu(), ram.average_map) and torch.equal(cnt.cpu(), ram.count_map)
    a2, c2 = ops.blend_resize(avg, cnt, 36, 48)
    ram.resize((36, 48))
    assert torch.equal(a2.cpu(), ram.average_map)
    _close(c2, ram.count_map, 2e-6)


@gpu
def test_zoe_head_ops_and_layout(T):
    ops, _ = T
    x = randn(22, 2, 20, 6, 8)
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.to(DEV))).cpu(), x)
    attr, bins = torch.rand(1, 6, 8, 16, generator=torch.Generator().manual_seed(23)) * 5, torch.rand(1, 6, 8, 64, generator=torch.Generator().manual_seed(24)) * 5
    out = ops.zoe_attractor(attr.to(DEV), bins.to(DEV), 300.0)
    dx = attr.unsqueeze(-1) - bins.unsqueeze(-2)
    _close(out, bins + (dx / (1 + 300.0 * dx ** 2)).mean(dim=-2), 1e-5)
    pt = torch.rand(1, 6, 8, 4, generator=torch.Generator().manual_seed(25)) + 0.1
    d = ops.zoe_bins_head(pt.to(DEV), bins.to(DEV), 0.0212, 50.0)
    assert tuple(d.shape) == (1, 1, 6, 8) and bool(torch.isfinite(d).all())
    assert float(d.min()) >= float(bins.min()) - 1e-4 and float(d.max()) <= float(bins.max()) + 1e-4  # an expectation over the bins


@gpu
def test_conv3x3_ups_equals_interpolate_then_conv(T):
    """torch.ops.prv2.conv3x3_ups: the bilinear(align_corners=True) upsample of UpSample.forward_hardcode (fusion_model.py:15-24) /
    of output_conv1's input (bi_directional_fusion_model.py:139-142,201) formed inside the conv's tile loader -- against
    F.interpolate + F.conv2d in fp32, with and without pass-through channels, and bit-equal to the unfused ops of this library"""
    import torch.nn.functional as F
    ops, mod = T
    n, h, w, H, W, cu, crest, cout = 2, 12, 17, 24, 33, 64, 34, 96
    u = randn(1, n, h, w, cu).to(DEV)
    rest = randn(2, n, H, W, crest).to(DEV)
    wt = randn(3, cout, cu + crest, 3, 3) / (9 * (cu + crest)) ** 0.5
    bias = randn(4, cout)
    wp = ops.pack_conv_weight(wt.to(DEV), None, 0, mod.PREC_BF16X3)
    xbuf = torch.full((n, H, W, cu + crest + 2), float("nan"), device=DEV)     # (padded row stride; the first cu channels are never read)
    x = xbuf[..., :cu + crest]
    x[..., cu:] = rest
    got = ops.conv3x3_ups(x, u, wp, bias.to(DEV), cout, H, W, act=mod.ACT_GELU, prec=mod.PREC_BF16X3)
    up = F.interpolate(u.permute(0, 3, 1, 2).cpu(), (H, W), mode="bilinear", align_corners=True)
    want = F.gelu(F.conv2d(torch.cat([up, rest.permute(0, 3, 1, 2).cpu()], 1), wt, bias, padding=1))
    err = float((got.permute(0, 3, 1, 2).cpu() - want).abs().max())
    assert err < 3e-5 * max(1.0, float(want.abs().max())), err
    x2 = torch.zeros_like(xbuf)[..., :cu + crest]
    x2[..., cu:] = rest
    ops.upsample_bilinear_ac(u, H, W, out=x2[..., :cu])
    ref = ops.conv2d(x2, wp, bias.to(DEV), cout, 3, 3, pad=1, act=mod.ACT_GELU, prec=mod.PREC_BF16X3)
    assert torch.equal(got, ref)
    # every channel from the upsampled source (x = None)
    wp2 = ops.pack_conv_weight(wt[:, :cu].contiguous().to(DEV), None, 0, mod.PREC_BF16X3)
    got2 = ops.conv3x3_ups(None, u, wp2, None, cout, H, W, prec=mod.PREC_BF16X3)
    want2 = F.conv2d(up, wt[:, :cu], None, padding=1)
    assert float((got2.permute(0, 3, 1, 2).cpu() - want2).abs().max()) < 3e-5 * max(1.0, float(want2.abs().max()))
    with pytest.raises(RuntimeError):
        ops.conv3x3_ups(None, u, ops.pack_conv_weight(wt[:32, :cu].contiguous().to(DEV), None, 0, mod.PREC_BF16X3), None, 32, H, W, prec=mod.PREC_BF16X3)  # cout <= 64


@gpu
def test_upconv3x3_and_the_split_concat_conv(T):
    """torch.ops.prv2.upconv3x3 exactly as INTEGRATION.md section B writes it: output_conv1(interpolate(path_1)) at path_1's resolution
    (bi_directional_fusion_model.py:139-142,201), and UpSample.forward_hardcode's first conv (fusion_model.py:15-24) split by weight
    columns -- conv2d over [x2 | pred1 | pred2] (raw) as the in-place addend of upconv3x3 over x1 -- against F.interpolate + F.conv2d in fp64"""
    import torch.nn.functional as F
    ops, mod = T
    n, h, w, cu, c2, cout = 2, 13, 20, 64, 32, 98
    H, W = 2 * h, 2 * w
    u = randn(1, n, h, w, cu).to(DEV)
    up = F.interpolate(u.permute(0, 3, 1, 2).cpu().double(), (H, W), mode="bilinear", align_corners=True)
    # (a) every input channel interpolated, bias, no activation
    w1, b1 = randn(2, 128, cu, 3, 3) / (9 * cu) ** 0.5, randn(3, 128)
    got = ops.upconv3x3(u, ops.pack_conv_weight(w1.to(DEV), None, 0, mod.PREC_BF16X3), b1.to(DEV), 128, H, W, act=mod.ACT_NONE, prec=mod.PREC_BF16X3)
    want = F.conv2d(up, w1.double(), b1.double(), padding=1).float()
    assert float((got.permute(0, 3, 1, 2).cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    # (b) the concat conv, split
    rest = randn(4, n, H, W, c2 + 2).to(DEV)
    w0 = randn(5, cout, cu + c2 + 2, 3, 3) / (9 * (cu + c2 + 2)) ** 0.5
    cat_buf = torch.full((n, H, W, cu + c2 + 2 + 2), float("nan"), device=DEV)   # [never-written up(x1) slot | x2 | pred1 | pred2 | pad]
    cat_buf[..., cu:cu + c2 + 2] = rest
    cat_buf[..., cu + c2 + 2:] = 0.0
    wa = ops.pack_conv_weight(w0[:, :cu].contiguous().to(DEV), None, 0, mod.PREC_BF16X3)
    wb = ops.pack_conv_weight(w0[:, cu:].contiguous().to(DEV), None, 0, mod.PREC_BF16X3)
    t = ops.conv2d(cat_buf[..., cu:cu + c2 + 2], wb, None, cout, 3, 3, pad=1, prec=mod.PREC_BF16X3)
    y = ops.upconv3x3(u, wa, None, cout, H, W, act=mod.ACT_GELU, prec=mod.PREC_BF16X3, out=t, add=t)
    assert y.data_ptr() == t.data_ptr()
    want = F.gelu(F.conv2d(torch.cat([up, rest.permute(0, 3, 1, 2).cpu().double()], 1), w0.double(), None, padding=1)).float()
    assert float((y.permute(0, 3, 1, 2).cpu() - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    with pytest.raises(RuntimeError):
        ops.upconv3x3(u, wa, None, cout, h + 3, W, prec=mod.PREC_BF16X3)   # source step 12/15 > 3/5


@gpu
def test_every_op_test_passes_on_the_other_route():
    """The host wrappers reach EVERY kernel on two routes: torch.ops.prv2.* (ops.DISPATCH = 'torch', the default) and ctypes straight
    on the C ABI (PRV2_DISPATCH=ctypes).  The whole per-op parity file (tests/test_hip_ops.py: convs with every epilogue, X2 formats,
    gate kernels, coarse taps, gathers, blend, ViT pieces) is run once more on the route that is NOT the default, in a child process
    (the switch is read at import)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    from patchrefinerv2_amd import ops
    env = dict(os.environ, PRV2_DISPATCH="ctypes" if ops.DISPATCH == "torch" else "torch")
    files = [os.path.join(here, f) for f in ("test_hip_ops.py", "test_chain32.py", "test_upconv5.py", "test_conv3x3_f6.py")]   # (round 5: the fused chains, the 5x5 composite, the fp16 + fp6 conv)
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"], env=env,
                       capture_output=True, text=True, timeout=1800, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
