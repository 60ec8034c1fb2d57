"""GPU probe: the GatedConvUnit kernels on fp32 vs pre-split (X2) buffers.   python tools/probes/x2_bench.py [n h w]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from patchrefinerv2_amd import ops as P

DEV = "cuda"


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


n, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (14, 192, 256)
F_ = 256
g = torch.Generator(device=DEV).manual_seed(0)
PR = P.L.PREC_NAMES["bf16x3"]
x = P.Feat(torch.randn(n, h, w, F_, device=DEV, generator=g))
coarse = P.Feat(torch.randn(1, h, w, F_, device=DEV, generator=g))
boxes = torch.tensor([[3.0 * i, 1.0 * i, 3.0 * i + w / 4.0, 1.0 * i + h / 4.0] for i in range(n)], device=DEV)
cw_c = P.pack_conv(torch.randn(F_, F_, 3, 3, device=DEV, generator=g) / 48, torch.randn(F_, device=DEV, generator=g), pad=1, prec=PR)
cw_f = P.pack_conv(torch.randn(F_, 2 * F_, 3, 3, device=DEV, generator=g) / 68, torch.randn(F_, device=DEV, generator=g), pad=1, prec=PR)
gw = P.pack_gate(torch.randn(F_, F_, 1, 1, device=DEV, generator=g) / 16)
gb = torch.randn(F_, device=DEV, generator=g)
ln = (torch.rand(F_, device=DEV, generator=g) + 0.5, torch.randn(F_, device=DEV, generator=g) * 0.1)
res = P.Feat(torch.randn(n, h, w, F_, device=DEV, generator=g))
if os.environ.get("PROBE_ZERO"):  # all-zero operands: the chip holds its full clock -- ranks kernels by CYCLES, not by what they deliver
    for t_ in (x, coarse, res):
        t_.buf.zero_()
    cw_c = P.pack_conv(torch.zeros(F_, F_, 3, 3, device=DEV), torch.zeros(F_, device=DEV), pad=1, prec=PR)
    cw_f = P.pack_conv(torch.zeros(F_, 2 * F_, 3, 3, device=DEV), torch.zeros(F_, device=DEV), pad=1, prec=PR)
    gw = P.pack_gate(torch.zeros(F_, F_, 1, 1, device=DEV))
for rnd in range(2):
    for x2 in (False, True):
        cat = P.Feat.alloc(n, h, w, 2 * F_, DEV)
        cat.x2 = x2
        t_roi = timeit(lambda: P.roi_align(coarse, boxes, 1.0, h, w, out=cat.slice(F_, F_)))
        out = cat.slice(0, F_)
        t_c = timeit(lambda: P.conv2d(x, cw_c, out, relu_in=True, res=x))
        y = P.Feat.alloc(n, h, w, F_, DEV)
        t_g = timeit(lambda: P.conv3x3_ln_gate(cat, cw_f, ln, gw, gb, y, act=P.ACT_RELU, mul=out, res=res))
        print(f"{'X2  ' if x2 else 'fp32'} {n}x{h}x{w}: roi gather {t_roi:.3f} ms | conv 256->256 (+x) {t_c:.3f} ms | gate tail 512->256->256 {t_g:.3f} ms "
              f"({2.0 * n * h * w * F_ * (9 * 2 * F_ + F_) / t_g / 1e9:.0f} TF)", flush=True)
