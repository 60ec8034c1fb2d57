"""GPU parity of ops.upconv5x5 (csrc/upconv5.hip; include/prv2.h::prv2_upconv5x5 / _lines / _ring) -- C2FModule's
``output_conv2[0] o output_conv1 o interpolate`` (bi_directional_fusion_model.py:139-146,169-173,201-203, refinenet1.out_conv folded into
output_conv1) as ONE 5x5 conv at the source resolution -- against the float64 evaluation of the reference's layer sequence, and the CPU
check of the decomposition itself (tools/studies/composite5x5_ring.py)."""
import importlib.util
import os

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def test_composite_5x5_with_bias_classes_and_ring_fix_is_exact_in_float64():
    """the whole decomposition (5x5 main term over the zero-padded upsample + 25 bias classes - per-edge 1-D five-tap convs + corner terms)
    against the two 3x3 convs, float64, incl. the folded out_conv bias and non-x2 sizes"""
    spec = importlib.util.spec_from_file_location("c5", os.path.join(ROOT, "tools", "studies", "composite5x5_ring.py"))
    c5 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(c5)
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    for uh, uw, H, W in ((9, 11, 18, 22), (5, 7, 9, 13), (4, 6, 8, 12)):
        err, scale = c5.check(r(2, 8, uh, uw), r(6, 8, 3, 3) / 8, r(6), (r(8, 8) / 3, r(8)), r(4, 6, 3, 3) / 7, r(4), (H, W))
        assert err < 1e-12 * max(scale, 1.0), (uh, uw, H, W, err)


def reference(u, w1, b1, tap_bias, w2, b2, size):
    """float64: relu(conv3x3(conv3x3(up(u); w1) + b1 + [taps of w1 inside] tap_bias; w2) + b2)"""
    d = lambda t: t.double()  # noqa: E731
    U = F.interpolate(d(u), size, mode="bilinear", align_corners=True)
    m = w1.shape[0]
    inside = torch.ones(1, 1, *size, dtype=torch.float64)
    t = F.conv2d(U, d(w1), d(b1), padding=1) + F.conv2d(inside, d(tap_bias).t().reshape(m, 1, 3, 3), padding=1)
    return F.relu(F.conv2d(t, d(w2), d(b2), padding=1))


CASES = [(1, 64, 32, (8, 13), (16, 26)), (2, 256, 128, (12, 16), (24, 32)), (1, 64, 32, (15, 20), (29, 39)), (1, 32, 16, (7, 9), (14, 24)),
         (3, 64, 32, (3, 5), (6, 10)), (1, 128, 64, (20, 31), (40, 62))]


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("bf16x3", 3e-5), ("bf16", 3e-2)])
@pytest.mark.parametrize("case", CASES)
def test_upconv5x5_vs_fp64_two_conv_reference(case, prec, tol):
    from patchrefinerv2_amd import ops as P
    n, ci, m, (uh, uw), (H, W) = case
    u = rnd(1, n, ci, uh, uw)
    w1, b1, tb = rnd(2, m, ci, 3, 3) / (3 * ci ** 0.5), rnd(3, m) * 0.2, rnd(4, 9, m) * 0.1
    w2, b2 = rnd(5, 32, m, 3, 3) / (3 * m ** 0.5), rnd(6, 32) * 0.2
    ref = reference(u, w1, b1, tb, w2, b2, (H, W))
    cw5 = P.compose_upconv5x5(w1, b1, tb, w2, b2, DEV, P.L.PREC_NAMES[prec])
    uf = P.Feat.from_nchw(u.to(DEV))
    assert P.upconv5x5_supported(uf, H, W, cw5)
    out = P.upconv5x5(uf, H, W, cw5, act=P.ACT_RELU)
    torch.cuda.synchronize()
    got = out.to_nchw().cpu().double()
    scale = max(1.0, float(ref.abs().max()))
    err = (got - ref).abs()
    ring = torch.zeros_like(err, dtype=torch.bool)
    ring[..., 0, :] = ring[..., -1, :] = ring[..., :, 0] = ring[..., :, -1] = True
    assert float(err[~ring].max()) <= tol * scale, f"interior max|d| {float(err[~ring].max()):.3e} (scale {scale:.2f})"
    assert float(err[ring].max()) <= tol * scale, f"ring max|d| {float(err[ring].max()):.3e} (scale {scale:.2f})"


@pytest.mark.gpu
def test_upconv5x5_full_tile_batch_independent_and_into_a_slice():
    from patchrefinerv2_amd import ops as P
    n, ci, m, (uh, uw), (H, W) = 2, 256, 128, (48, 64), (96, 128)
    u = rnd(11, n, ci, uh, uw)
    w1, b1, tb = rnd(12, m, ci, 3, 3) / 48, rnd(13, m) * 0.2, rnd(14, 9, m) * 0.1
    w2, b2 = rnd(15, 32, m, 3, 3) / 34, rnd(16, 32) * 0.2
    cw5 = P.compose_upconv5x5(w1, b1, tb, w2, b2, DEV, P.L.PREC_BF16X3)
    uf = P.Feat.from_nchw(u.to(DEV))
    a = P.upconv5x5(uf, H, W, cw5, act=P.ACT_RELU)
    buf = P.Feat(torch.full((n, H, W, 48), 5.0, device=DEV))
    b = P.upconv5x5(uf, H, W, cw5, out=buf.slice(8, 32), act=P.ACT_RELU)
    assert torch.equal(a.to_nchw(), b.to_nchw()) and float(buf.buf[..., :8].min()) == 5.0 and float(buf.buf[..., 40:].max()) == 5.0
    one = P.upconv5x5(uf.batch(1, 2), H, W, cw5, act=P.ACT_RELU)
    assert torch.equal(one.to_nchw(), a.to_nchw()[1:2])
    ref = reference(u, w1, b1, tb, w2, b2, (H, W))
    assert float((a.to_nchw().cpu().double() - ref).abs().max()) <= 3e-5 * max(1.0, float(ref.abs().max()))
