"""Float64 check of the COMPLETE decomposition ops.upconv5x5 implements (output_conv2.0 o output_conv1 o interpolate as one 5x5
conv at the source resolution; bi_directional_fusion_model.py:139-146,169-173,201-203 with refinenet1.out_conv folded into output_conv1):

    t = conv3x3(U; W1) + b1 + [taps of W1 inside] . b_oc        U = interpolate(u, x2, bilinear, align_corners=True), zero padding of U
    v = relu(conv3x3(t; W2) + b2)                               zero padding of t

    v = relu( MAIN + BIAS_MAP - RING_FIX ),
    MAIN      = conv5x5(U; Weff) over the zero-padded U                      (csrc/upconv5.hip: tap GEMMs at u's resolution + gather)
    BIAS_MAP  = data independent, 5 x 5 position classes (two border rows / columns on each side + interior)
    RING_FIX  = on the one-pixel border ring only: per edge a 1-D five-tap conv (zero padded) of the border row / column of U -- itself an
                interpolation of u's border row / column, so again tap GEMMs at the source resolution + a 1-D gather -- plus, at the four
                corner pixels, the term of the outside corner position that the two edges count twice.
``check`` returns the maximum deviation from the two-conv float64 reference."""
import torch
import torch.nn.functional as F


def compose(w1, b1, tap_bias, w2, b2):
    """(Weff [o, i, 5, 5], bias_map [5, 5, o], edge weights dict, corner weights dict); w1 [m, i, 3, 3] is the FOLDED output_conv1
    (W1 o out_conv), tap_bias [9, m] = W1[tap] . b_oc, b1 the conv's own bias"""
    o, m = w2.shape[:2]
    i = w1.shape[1]
    weff = torch.zeros(o, i, 5, 5, dtype=w1.dtype)
    for y2 in range(3):
        for x2 in range(3):
            weff[:, :, y2:y2 + 3, x2:x2 + 3] += torch.einsum("om,miyx->oiyx", w2[:, :, y2, x2], w1)
    # bias classes on an 8 x 8 grid: t_bias(q) = b1 + sum of tap_bias over the taps of q that stay inside; v_bias(p) = b2 + sum_{d2 inside} W2[d2] t_bias(p + d2)
    n = 8
    inside = torch.ones(1, 1, n, n, dtype=w1.dtype)
    tb = b1.view(1, m, 1, 1) + F.conv2d(inside, tap_bias.t().reshape(m, 1, 3, 3), padding=1)
    vb = F.conv2d(tb, w2, b2, padding=1)[0]                      # [o, 8, 8]
    cls = [0, 1, 3, 6, 7]                                        # rows / columns standing for classes 0, 1, interior, H - 2, H - 1
    bias_map = vb[:, cls][:, :, cls].permute(1, 2, 0).contiguous()
    # edges: fix(p) = sum_e Wedge[e] a(pos + e), a = the border row / column of U, e = -2 .. 2
    def edge(k2sel, k1sel):
        we = torch.zeros(5, o, i, dtype=w1.dtype)
        for k2 in range(3):
            for k1 in range(3):
                we[k2 + k1] += k2sel(k2) @ k1sel(k1)
        return we
    edges = dict(top=edge(lambda k: w2[:, :, 0, k], lambda k: w1[:, :, 2, k]), bottom=edge(lambda k: w2[:, :, 2, k], lambda k: w1[:, :, 0, k]),
                 left=edge(lambda k: w2[:, :, k, 0], lambda k: w1[:, :, k, 2]), right=edge(lambda k: w2[:, :, k, 2], lambda k: w1[:, :, k, 0]))
    # corners: the outside corner position is counted by both edges: add it back once
    corners = dict(tl=w2[:, :, 0, 0] @ w1[:, :, 2, 2], tr=w2[:, :, 0, 2] @ w1[:, :, 2, 0], bl=w2[:, :, 2, 0] @ w1[:, :, 0, 2], br=w2[:, :, 2, 2] @ w1[:, :, 0, 0])
    return weff, bias_map, edges, corners


def check(u, w1, b1, b_oc, w2, b2, size):
    H, W = size
    U = F.interpolate(u, size, mode="bilinear", align_corners=True)
    tap_bias = torch.einsum("miyx,i->yxm", w1, torch.zeros(w1.shape[1], dtype=u.dtype)) if b_oc is None else None
    if b_oc is not None:   # w1 here is the UNFOLDED output_conv1 over out_conv(u): fold
        woc, boc = b_oc
        tap_bias = torch.einsum("omyx,m->yxo", w1, boc).reshape(9, -1)
        w1 = torch.einsum("omyx,mi->oiyx", w1, woc)
    else:
        tap_bias = tap_bias.reshape(9, -1)
    m = w1.shape[0]
    inside = torch.ones(1, 1, H, W, dtype=u.dtype)
    t = F.conv2d(U, w1, b1, padding=1) + F.conv2d(inside, tap_bias.t().reshape(m, 1, 3, 3), padding=1)
    ref = F.relu(F.conv2d(t, w2, b2, padding=1))
    weff, bias_map, edges, corners = compose(w1, b1, tap_bias, w2, b2)
    main = F.conv2d(U, weff, None, padding=2)
    cy = torch.tensor([0, 1] + [2] * (H - 4) + [3, 4])
    cx = torch.tensor([0, 1] + [2] * (W - 4) + [3, 4])
    v = main + bias_map[cy][:, cx].permute(2, 0, 1)[None]
    def conv1d(a, we):  # a [n, i, L] -> [n, o, L]: zero-padded five-tap conv
        return F.conv1d(a, we.permute(1, 2, 0), padding=2)
    v[:, :, 0, :] -= conv1d(U[:, :, 0, :], edges["top"])
    v[:, :, H - 1, :] -= conv1d(U[:, :, H - 1, :], edges["bottom"])
    v[:, :, :, 0] -= conv1d(U[:, :, :, 0], edges["left"])
    v[:, :, :, W - 1] -= conv1d(U[:, :, :, W - 1], edges["right"])
    for (yy, xx), k in (((0, 0), "tl"), ((0, W - 1), "tr"), ((H - 1, 0), "bl"), ((H - 1, W - 1), "br")):
        v[:, :, yy, xx] += torch.einsum("oi,ni->no", corners[k], U[:, :, yy, xx])
    return float((F.relu(v) - ref).abs().max()), float(ref.abs().max())


if __name__ == "__main__":
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    for (uh, uw, H, W) in ((9, 11, 18, 22), (5, 7, 9, 13), (12, 16, 24, 32)):
        u = r(2, 8, uh, uw)
        print((uh, uw, H, W), check(u, r(6, 8, 3, 3) / 8, r(6), (r(8, 8) / 3, r(8)), r(4, 6, 3, 3) / 7, r(4), (H, W)))
