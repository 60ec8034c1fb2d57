"""see f6_gate_stamps.sh"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
stamps = torch.zeros(8000000 + 100000 * 4, dtype=torch.int64, device="cuda")
os.environ["PRV2_STAMP_PTR"] = hex(stamps.data_ptr())
from patchrefinerv2_amd import lib as L
L.LIB_PATH = os.environ["PRV2_LIB_OVERRIDE"]
from patchrefinerv2_amd import ops as P
NAMES = ["-", "main loop (36 steps) + drain", "C tile (+ pre) -> LDS", "row stats", "normalise + split", "gate GEMM", "gate acc -> LDS", "store loop", "store drain"]
n, h, w, F_ = 14, 192, 256, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)  # noqa: E731
x = P.Feat(rn(n, h, w, F_))
cwc = P.pack_conv3x3_f6(rn(F_, F_, 3, 3) / 48, rn(F_) * .1)
out = P.Feat(torch.empty(n, h, w, F_, device="cuda"), x2=True)
P.conv3x3_f6(x, cwc, out, relu_in=True, res=x)
pre, res = P.Feat(rn(n, h, w, F_) * .5), P.Feat(rn(n, h, w, F_))
y = P.Feat.alloc(n, h, w, F_, "cuda")
cw6 = P.pack_conv3x3_f6(rn(F_, F_, 3, 3) / 48, rn(F_) * .1)
gw, gb = P.pack_gate(rn(F_, F_, 1, 1) / 16), rn(F_) * .1
ln = (torch.rand(F_, device="cuda", generator=g) + 0.5, rn(F_) * 0.1)
for _ in range(3):
    P.conv3x3_ln_gate_f6(out, cw6, ln, gw, gb, y, mul=out, res=res, pre=pre, pre_cin=F_)
torch.cuda.synchronize()
nblk = n * (h // 8) * (w // 16)
sw = stamps[: nblk * 80].view(nblk, 8, 10).cpu().double()
s0 = sw[:, 0]
d = (s0[:, 1:] - s0[:, :-1]) / 1000.0
tot = (s0[:, 9] - s0[:, 0]) / 1000.0
print(f"conv3x3_c256_gate_f6_kernel {n}x{h}x{w} 256(+pre, X2)->256->256: {nblk} workgroups, median {tot.median():.1f} kcycles per workgroup: " +
      "  ".join(f"{nm} {d[:, i].median():.2f}" for i, nm in enumerate(NAMES) if i))
clk = stamps[8000000: 8000000 + nblk * 4].view(nblk, 2, 2).cpu().double()
ok = clk[:, 1, 1] > clk[:, 0, 1]
print(f"in-kernel clock {((clk[ok, 1, 0] - clk[ok, 0, 0]) / (clk[ok, 1, 1] - clk[ok, 0, 1])).median().item() * 0.1:.3f} GHz; workgroup lifetime {((clk[ok, 1, 1] - clk[ok, 0, 1]) / 100).median().item():.1f} us")
