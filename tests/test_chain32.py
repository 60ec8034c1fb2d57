"""GPU parity of the fused 32-channel full-resolution chains (csrc/chain32.hip; include/prv2.h::prv2_chain32_c2f / prv2_chain32_enc)
against the float64 evaluation of the reference's layer sequence (bi_directional_fusion_model.py:56-82,116-146,171-180,203-204 and
:424-431; convs.py:21-29,58-72), and against the unfused HIP kernels they replace."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def P():
    from patchrefinerv2_amd import ops
    ops.L.load()
    return ops


def rnd(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def ln_cf(x, w, b, eps=1e-6):
    """channels-first LayerNorm of convs.py:21-29 on NCHW"""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[None, :, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[None, :, None, None]


def c2f_weights(seed):
    g = lambda i, *s: rnd(seed * 100 + i, *s)  # noqa: E731
    return dict(w1=g(1, 32, 32, 3, 3) / 17, b1=g(2, 32) * 0.1, w2=g(3, 32, 32, 3, 3) / 17, b2=g(4, 32) * 0.1, lnw=1 + 0.2 * g(5, 32), lnb=0.1 * g(6, 32),
                wg=g(7, 32, 32) / 5.6, wo=g(8, 32, 32) / 5.6, bo=g(9, 32) * 0.1, w3=1 + 0.3 * g(10, 32), b3=0.25)


def c2f_reference(x, pre, W):
    """float64: GateresConfUnit2 of a one-input GatedFusionBlock (upscale=False) with the coarse half of fusion_conv.0 given as ``pre``,
    out_conv, output_conv3"""
    d = lambda t: t.double()  # noqa: E731
    x = d(x)
    o = F.conv2d(F.relu(x), d(W["w1"]), d(W["b1"]), padding=1) + x
    f = F.conv2d(o, d(W["w2"]), d(W["b2"]), padding=1) + (d(pre) if pre is not None else 0)
    f = F.relu(ln_cf(f, d(W["lnw"]), d(W["lnb"])))
    y = o * torch.sigmoid(F.conv2d(f, d(W["wg"])[:, :, None, None]))
    last = F.conv2d(y, d(W["wo"])[:, :, None, None], d(W["bo"]))
    depth = F.conv2d(last, d(W["w3"]).view(1, 32, 1, 1)) + W["b3"]
    return last, depth


def pack_c2f(P, W):
    return dict(w1=P.pack_chain32(W["w1"], 0, DEV), w2=P.pack_chain32(W["w2"], 1, DEV), wg=P.pack_chain32(W["wg"], 1, DEV), wo=P.pack_chain32(W["wo"], 1, DEV),
                consts=P.chain32_consts(DEV, b1=W["b1"], ln1w=W["lnw"], ln1b=W["lnb"], b2=W["b2"], bo=W["bo"], w3=W["w3"]), b3=W["b3"])


def close(got, ref, tol, what=""):
    got = got.detach().cpu().double()
    err = float((got - ref).abs().max())
    scale = max(1.0, float(ref.abs().max()))
    assert err <= tol * scale, f"{what}: max|d|={err:.3e} (scale {scale:.2f})"
    return err / scale


C2F_CASES = [(1, 8, 16, True), (2, 24, 32, True), (1, 17, 29, True), (3, 5, 7, False), (1, 40, 100, True), (2, 9, 16, True)]


@pytest.mark.parametrize("case", C2F_CASES)
def test_chain32_c2f_vs_fp64(P, case):
    n, h, w, with_pre = case
    W = c2f_weights(3)
    x = rnd(1, n, 32, h, w)
    pre = rnd(2, n, 32, h, w) * 0.5 if with_pre else None
    ref_last, ref_depth = c2f_reference(x, pre, W)
    cw = pack_c2f(P, W)
    xf = P.Feat.from_nchw(x.to(DEV))
    pf = P.Feat.from_nchw(pre.to(DEV)) if with_pre else None
    last, depth = P.chain32_c2f(xf, cw, pf)
    torch.cuda.synchronize()
    close(last.to_nchw(), ref_last, 3e-5, "last")
    close(depth, ref_depth, 3e-5, "depth")


def test_chain32_c2f_writes_a_channel_slice_and_is_batch_independent(P):
    W = c2f_weights(4)
    x = rnd(5, 3, 32, 19, 37)
    pre = rnd(6, 3, 32, 19, 37)
    cw = pack_c2f(P, W)
    xf, pf = P.Feat.from_nchw(x.to(DEV)), P.Feat.from_nchw(pre.to(DEV))
    last, depth = P.chain32_c2f(xf, cw, pf)
    buf = P.Feat(torch.full((3, 19, 37, 68), 7.0, device=DEV))
    last2, depth2 = P.chain32_c2f(xf, cw, pf, out=buf.slice(32, 32))
    assert torch.equal(last2.to_nchw(), last.to_nchw()) and torch.equal(depth, depth2)
    assert float(buf.buf[..., :32].min()) == 7.0 and float(buf.buf[..., 64:].max()) == 7.0
    one, d1 = P.chain32_c2f(xf.batch(1, 2), cw, pf.batch(1, 2))
    assert torch.equal(one.to_nchw(), last.to_nchw()[1:2]) and torch.equal(d1, depth[1:2])


def test_chain32_c2f_against_the_unfused_kernels(P):
    """the same block through ops.conv2d / conv3x3_ln_gate / conv2d_cout1 (what fusion.py ran before): fp32-grade agreement"""
    n, h, w = 2, 24, 40
    W = c2f_weights(7)
    x, pre = rnd(8, n, 32, h, w), rnd(9, n, 32, h, w) * 0.5
    cw = pack_c2f(P, W)
    xf, pf = P.Feat.from_nchw(x.to(DEV)), P.Feat.from_nchw(pre.to(DEV))
    last, depth = P.chain32_c2f(xf, cw, pf)
    prec = P.L.PREC_BF16X3
    conv = P.pack_conv(W["w1"], W["b1"], device=DEV, prec=prec)
    f0 = P.pack_conv(W["w2"], W["b2"], device=DEV, prec=prec)
    o = P.conv2d(xf, conv, relu_in=True, res=xf)
    y = P.conv3x3_ln_gate(o, f0, (W["lnw"].to(DEV), W["lnb"].to(DEV)), P.pack_gate(W["wg"].to(DEV)), None, act=P.ACT_RELU, mul=o, pre=pf, pre_cin=32)
    l2 = P.conv2d(y, P.pack_conv(W["wo"], W["bo"], device=DEV, prec=prec))
    d2 = P.conv2d_cout1(l2, W["w3"].view(1, 32, 1, 1).to(DEV), torch.tensor([W["b3"]], device=DEV), 1)
    close(last.to_nchw(), l2.to_nchw().cpu().double(), 2e-5, "last vs unfused")
    close(depth, d2.cpu().double(), 2e-5, "depth vs unfused")


def enc_weights(seed):
    g = lambda i, *s: rnd(seed * 100 + i, *s)  # noqa: E731
    return dict(w1=g(1, 32, 32, 3, 3) / 17, b1=g(2, 32) * 0.1, ln1w=1 + 0.2 * g(3, 32), ln1b=0.1 * g(4, 32), w2=g(5, 32, 34, 3, 3) / 17.5, b2=g(6, 32) * 0.1,
                ln2w=1 + 0.2 * g(7, 32), ln2b=0.1 * g(8, 32))


def enc_reference(x, pre, p1, p2, W):
    d = lambda t: t.double()  # noqa: E731
    f = F.gelu(ln_cf(F.conv2d(d(x), d(W["w1"]), d(W["b1"]), padding=1) + d(pre), d(W["ln1w"]), d(W["ln1b"])))
    f = torch.cat([f, d(p1), d(p2)], 1)
    return F.gelu(ln_cf(F.conv2d(f, d(W["w2"]), d(W["b2"]), padding=1), d(W["ln2w"]), d(W["ln2b"])))


def pack_enc(P, W):
    return dict(w1=P.pack_chain32(W["w1"], 0, DEV), w2=P.pack_chain32(W["w2"], 1, DEV), wt=P.pack_chain32(W["w2"], 2, DEV),
                consts=P.chain32_consts(DEV, b1=W["b1"], ln1w=W["ln1w"], ln1b=W["ln1b"], b2=W["b2"], ln2w=W["ln2w"], ln2b=W["ln2b"]))


@pytest.mark.parametrize("case", [(1, 8, 16), (2, 24, 32), (1, 17, 29), (3, 5, 7), (1, 40, 100)])
def test_chain32_enc_vs_fp64(P, case):
    n, h, w = case
    W = enc_weights(5)
    x, pre = rnd(1, n, 32, h, w), rnd(2, n, 32, h, w) * 0.5
    p1, p2 = rnd(3, n, 1, h, w).abs() * 3, rnd(4, n, 1, h, w).abs() * 3
    ref = enc_reference(x, pre, p1, p2, W)
    cw = pack_enc(P, W)
    buf = P.Feat(torch.zeros((n, h, w, 100), device=DEV))  # the decoder's concat buffer [up(x1) 64 | x2 32 | p1 p2 + pad]
    out = P.chain32_enc(P.Feat.from_nchw(x.to(DEV)), cw, P.Feat.from_nchw(pre.to(DEV)), p1.to(DEV).contiguous(), p2.to(DEV).contiguous(), out=buf.slice(64, 32))
    torch.cuda.synchronize()
    close(out.to_nchw(), ref, 3e-5, "enc")
    assert float(buf.buf[..., :64].abs().max()) == 0.0 and float(buf.buf[..., 96:].abs().max()) == 0.0


def test_chain32_full_tile_size_and_determinism(P):
    """one full-resolution tile (384 x 512): every workgroup of the persistent grid is busy, results repeat bit for bit"""
    W = c2f_weights(11)
    x, pre = rnd(12, 2, 32, 384, 512), rnd(13, 2, 32, 384, 512) * 0.5
    cw = pack_c2f(P, W)
    xf, pf = P.Feat.from_nchw(x.to(DEV)), P.Feat.from_nchw(pre.to(DEV))
    last, depth = P.chain32_c2f(xf, cw, pf)
    last_b, depth_b = P.chain32_c2f(xf, cw, pf)
    assert torch.equal(last.buf, last_b.buf) and torch.equal(depth, depth_b)
    ref_last, ref_depth = c2f_reference(x, pre, W)
    close(last.to_nchw(), ref_last, 3e-5, "last")
    close(depth, ref_depth, 3e-5, "depth")


@pytest.mark.gpu
def test_bidir_fusion_with_tap_tables_takes_the_fused_chains_and_matches_the_oracle(P):
    """BiDirectionalFusion (bf16x3) as the frame driver runs it -- the coarse pyramid as per-frame maps + ROI boxes, the coarse half of the
    cat([., c_feat]) convs from the per-frame tap tables (prepare_frame) -- takes ops.chain32_c2f / chain32_enc at level 0 and
    ops.upconv5x5 behind refinenet1; against the fp32 oracle on the ROI-aligned pyramid (bi_directional_fusion_model.py:290-446) and
    against the same network with the round-5 kernels switched off."""
    import numpy as np
    from oracle import fusion as o_fusion, ops as o_ops
    from oracle.cases import TINY_BIDIR
    from patchrefinerv2_amd import ops, weights as W
    from patchrefinerv2_amd.fusion import BiDirectionalFusion
    c = TINY_BIDIR
    sd = W.synth_state_dict(W.bidir_fusion_spec("", c["coarse_chl"], c["fine_chl"], c["fine_chl_after"], c["temp_chl"], c["dec_chl"]), seed=c["seed"])
    sizes = [(64, 96), (32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]       # level sizes (tile == frame size per level: split 2 x 2 zooms x2)
    Ph, Pw = 64, 96                                                        # patch_process_shape
    K = 5
    rng = np.random.default_rng(4)
    th, tw = Ph / 2, Pw / 2
    org = [(0.0, 0.0), (Pw - tw, Ph - th)] + [(float(rng.uniform(0, Pw - tw)), float(rng.uniform(0, Ph - th))) for _ in range(K - 2)]
    boxes = torch.tensor([[x, y, x + tw, y + th] for x, y in org], dtype=torch.float32)
    frame = [rnd(40 + l, 1, ch, *sizes[l]) for l, ch in enumerate(c["coarse_chl"])]
    fine = [None] + [rnd(50 + l, K, ch, *sizes[l]) for l, ch in list(enumerate([32] + c["fine_chl"]))[1:]]
    pred1 = torch.rand(K, 1, *sizes[0], generator=torch.Generator().manual_seed(7)) * 10
    rois_t = [o_ops.roi_align(f, torch.cat([torch.zeros(K, 1), boxes], 1), sizes[l], sizes[l][0] / Ph) for l, f in enumerate(frame)]
    ref = o_fusion.bidirectional_fusion(sd, "", rois_t, [torch.zeros(K, 32, *sizes[0])] + fine[1:], pred1, torch.zeros_like(pred1), update_base=pred1)

    def run(chain, up5):
        ops.CHAIN32, ops.UPCONV5 = chain, up5
        calls = dict(c2f=0, enc=0, up5=0)
        real = (ops.chain32_c2f, ops.chain32_enc, ops.upconv5x5)

        def spy(name, fn):
            def w(*a, **k):
                calls[name] += 1
                return fn(*a, **k)
            return w
        ops.chain32_c2f, ops.chain32_enc, ops.upconv5x5 = spy("c2f", real[0]), spy("enc", real[1]), spy("up5", real[2])
        try:
            m = BiDirectionalFusion(coarse2fine_type="coarse-gated", coarse_chl=c["coarse_chl"], fine_chl=c["fine_chl"],
                                    fine_chl_after_coarse2fine=c["fine_chl_after"], temp_chl=c["temp_chl"], dec_chl=c["dec_chl"], prec="bf16x3")
            m.load_state_dict(sd)
            fr = [P.Feat.from_nchw(t.to(DEV)) for t in frame]
            m.prepare_frame(fr, (0.5, 0.5))
            rois = [ops.RoiSource(f, boxes.to(DEV), f.h / Ph, f.h, f.w) for f in fr]
            ff = [None] + [P.Feat.from_nchw(t.to(DEV)) for t in fine[1:]]
            out = m(rois, ff, pred1.to(DEV), torch.zeros_like(pred1).to(DEV), update_base=pred1.to(DEV), f_sizes=sizes).clone()
        finally:
            ops.chain32_c2f, ops.chain32_enc, ops.upconv5x5 = real
            ops.CHAIN32 = ops.UPCONV5 = True
        return out, calls

    new, calls = run(True, True)
    assert calls == dict(c2f=1, enc=1, up5=1), calls
    old, calls0 = run(False, False)
    assert calls0 == dict(c2f=0, enc=0, up5=0), calls0
    close(new, ref.double(), 3e-5, "fused chains + 5x5 composite vs fp32 oracle")
    close(old, ref.double(), 3e-5, "round-4 kernels vs fp32 oracle")
    close(new, old.cpu().double(), 3e-5, "round-5 vs round-4 kernels")
