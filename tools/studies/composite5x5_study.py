"""Groundwork for DESIGN.md section 9 item 0a (not built): C2FModule's output_conv2[0] o output_conv1 o interpolate as ONE 5x5 conv of the upsampled
map (bi_directional_fusion_model.py:139-146,201-203; both convs 3x3 / pad 1, nothing non-linear in between).

    t = conv3x3(U; W1) + b1            U = interpolate(u, x2, bilinear, align_corners=True), zero padding of U
    v = conv3x3(t; W2) + b2            zero padding of t

Interior (every pixel at least 2 pixels from the border): v = conv5x5(U; Weff) + beff with Weff[d] = sum_{d1 + d2 = d} W2[d2] W1[d1] and
beff = b2 + sum_{d2} W2[d2] b1.  Within 2 pixels of the border the OUTER conv's zero padding removes the taps d2 whose t-pixel lies outside the image
-- both from the bias sum (a position-dependent bias map, data independent) and from the composite weights (the 5x5 formula must not see them):
    v(p) = sum_{d2: p + d2 inside} W2[d2] (sum_{d1: p + d2 + d1 inside} W1[d1] U(p + d2 + d1) + b1) + b2.
``composite`` evaluates exactly that as  conv5x5(U; Weff) + bias_map - ring_fix:  the 5x5 conv over the zero-padded U equals the outer conv over the
inner conv's data term on the EXTENDED (H + 2) x (W + 2) grid, so the fix subtracts the outer conv over that grid's outside frame -- non-zero on the
ONE-pixel border ring only (2 (H + W) - 4 pixels: 0.9 % of a 384 x 512 tile) -- and the bias map is b2 + the W2-sum of b1 over the taps that stay inside
(nine distinct vectors).  Everything is float64 here; the
function returns the maximum difference to the two-conv reference and the sizes the three pieces have."""
import torch
import torch.nn.functional as F


def composite(u, w1, b1, w2, b2, size):
    U = F.interpolate(u, size, mode="bilinear", align_corners=True)
    ref = F.conv2d(F.conv2d(U, w1, b1, padding=1), w2, b2, padding=1)
    H, W = size
    # 5x5 composite weights: Weff[o, i, d] = sum over (d1 + d2 = d) of W2[o, m, d2] W1[m, i, d1]
    weff = torch.zeros(w2.shape[0], w1.shape[1], 5, 5, dtype=u.dtype)
    for y2 in range(3):
        for x2 in range(3):
            weff[:, :, y2:y2 + 3, x2:x2 + 3] += torch.einsum("om,miyx->oiyx", w2[:, :, y2, x2], w1)
    main = F.conv2d(U, weff, None, padding=2)                                     # what the upconv-style kernel would compute everywhere
    inside = torch.zeros(1, 1, H + 2, W + 2, dtype=u.dtype)
    inside[..., 1:-1, 1:-1] = 1.0
    # data independent: b2 + sum of W2[d2] b1 over the taps whose t-pixel lies inside -- nine distinct vectors (interior, four edges, four corners)
    bias_map = F.conv2d(b1.view(1, -1, 1, 1).expand(1, -1, H + 2, W + 2) * inside, w2, b2, padding=0)
    # the inner conv's DATA term one pixel beyond the image (zero-padded U), seen through the outer conv: non-zero on the one-pixel ring only
    t_ext_data = F.conv2d(U, w1, None, padding=2)                                 # [H + 2, W + 2]
    ring_fix = F.conv2d(t_ext_data * (1.0 - inside), w2, None, padding=0)
    got = main + bias_map - ring_fix
    ring = ring_fix.abs().amax(1, keepdim=True) > 0
    return dict(err=float((got - ref).abs().max()), scale=float(ref.abs().max()), ring_pixels=int(ring.sum()), pixels=H * W,
                bias_vectors=int(torch.unique(bias_map.flatten(2).round(decimals=9), dim=2).shape[2]),
                err_without_ring_fix=float((main + bias_map - ref).abs().max()),
                macs_two_convs=9 * w1.shape[1] * w1.shape[0] + 9 * w2.shape[1] * w2.shape[0], macs_composite=25 * w1.shape[1] * w2.shape[0])


if __name__ == "__main__":
    g = torch.Generator().manual_seed(1)
    u = torch.randn(1, 8, 9, 11, generator=g, dtype=torch.float64)
    w1 = torch.randn(6, 8, 3, 3, generator=g, dtype=torch.float64) / 8
    b1 = torch.randn(6, generator=g, dtype=torch.float64)
    w2 = torch.randn(4, 6, 3, 3, generator=g, dtype=torch.float64) / 7
    b2 = torch.randn(4, generator=g, dtype=torch.float64)
    print(composite(u, w1, b1, w2, b2, (18, 22)))
