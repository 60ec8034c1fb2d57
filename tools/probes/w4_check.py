"""GPU probe: the 4-wave gate kernel (X2 input) against the 8-wave kernel (fp32 input) on the same GatedConvUnit tail.
   python tools/probes/w4_check.py [n h w]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from patchrefinerv2_amd import ops as P

DEV = "cuda"
n, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 24, 32)
F_ = 256
g = torch.Generator(device=DEV).manual_seed(0)
PR = P.L.PREC_NAMES["bf16x3"]
x = P.Feat(torch.randn(n, h, w, F_, device=DEV, generator=g))
coarse = P.Feat(torch.randn(1, 2 * h, 2 * w, F_, device=DEV, generator=g))
boxes = torch.tensor([[1.0 + 3 * i, 2.0 + i, 1.0 + 3 * i + w / 2.0, 2.0 + i + h / 2.0] for i in range(n)], device=DEV)
cw_c = P.pack_conv(torch.randn(F_, F_, 3, 3, device=DEV, generator=g) / 48, torch.randn(F_, device=DEV, generator=g), pad=1, prec=PR)
cw_f = P.pack_conv(torch.randn(F_, 2 * F_, 3, 3, device=DEV, generator=g) / 68, torch.randn(F_, device=DEV, generator=g), pad=1, prec=PR)
gw = P.pack_gate(torch.randn(F_, F_, 1, 1, device=DEV, generator=g) / 16)
gb = torch.randn(F_, device=DEV, generator=g)
ln = (torch.rand(F_, device=DEV, generator=g) + 0.5, torch.randn(F_, device=DEV, generator=g) * 0.1)
res = P.Feat(torch.randn(n, h, w, F_, device=DEV, generator=g))
ys = []
for x2 in (False, True):
    cat = P.Feat.alloc(n, h, w, 2 * F_, DEV)
    cat.x2 = x2
    P.roi_align(coarse, boxes, 1.0, h, w, out=cat.slice(F_, F_))
    out = P.conv2d(x, cw_c, cat.slice(0, F_), relu_in=True, res=x)
    y = P.conv3x3_ln_gate(cat, cw_f, ln, gw, gb, act=P.ACT_RELU, mul=out, res=res)
    print(P.L.load().prv2_last_kernel().decode())
    ys.append(y.buf.clone())
torch.cuda.synchronize()
d = (ys[0] - ys[1]).abs()
print("max|d|", d.max().item(), "scale", ys[0].abs().max().item(), "unequal", int((d > 0).sum()), "of", d.numel(), "finite", bool(torch.isfinite(ys[1]).all()))
if d.max() > 1e-4:
    bad = (d > 1e-4)
    print("bad per image", bad.sum((1, 2, 3)).tolist())
    print("bad rows", bad.sum((0, 2, 3)).tolist())
    print("bad cols", bad.sum((0, 1, 3)).tolist())
    print("bad channels (first 64)", bad.sum((0, 1, 2)).tolist()[:64])
