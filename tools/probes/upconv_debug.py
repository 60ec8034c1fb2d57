import sys, numpy as np, torch
sys.path.insert(0, ".")
from patchrefinerv2_amd import ops as P
torch.manual_seed(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
PR = P.L.PREC_NAMES[prec]
n, h, w, H, W, cin, cout = 1, 16, 16, 32, 32, 32, 32
u_t = torch.randn(n, cin, h, w); w_t = torch.randn(cout, cin, 3, 3) / np.sqrt(9 * cin)
cw = P.pack_conv(w_t.cuda(), None, pad=1, prec=PR)
up = torch.nn.functional.interpolate(u_t.double(), (H, W), mode="bilinear", align_corners=True)
ref = torch.nn.functional.conv2d(up, w_t.double(), None, padding=1).float()
for rep in range(3):
    got = P.upconv3x3(P.Feat.from_nchw(u_t.cuda()), H, W, cw).to_nchw().cpu()
    err = (got - ref).abs().amax(dim=(0, 1))
    print("rep", rep, "max", float(err.max()))
    for y in range(H):
        print("".join("#" if e > 0.05 else ("+" if e > 1e-3 else ".") for e in err[y].tolist()))
# per-tap isolation: only one tap non-zero
for tap in range(9):
    wt = torch.zeros_like(w_t); wt[:, :, tap // 3, tap % 3] = w_t[:, :, tap // 3, tap % 3]
    cwt = P.pack_conv(wt.cuda(), None, pad=1, prec=PR)
    got = P.upconv3x3(P.Feat.from_nchw(u_t.cuda()), H, W, cwt).to_nchw().cpu()
    r = torch.nn.functional.conv2d(up, wt.double(), None, padding=1).float()
    e = (got - r).abs().amax(dim=(0, 1))
    bad = torch.nonzero(e > 0.02)
    print("tap", tap, "max err %.3f" % float(e.max()), "bad rows", sorted(set(bad[:, 0].tolist())), "bad cols", sorted(set(bad[:, 1].tolist())))
