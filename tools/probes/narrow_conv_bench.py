"""3x3 convs with <= 32 / 64 output channels at full tile resolution (the BN = 32 / 64 instantiations of conv3x3_m16.hip), with a
parity check against the generic kernel.   [PRV2_HALO_N32=0|1] python tools/probes/narrow_conv_bench.py"""
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


g = torch.Generator(device=DEV).manual_seed(0)
PR = P.L.PREC_NAMES["bf16x3"]
for n, h, w, cin, cout in [(14, 384, 512, 64, 32), (14, 384, 512, 128, 32), (14, 384, 512, 98, 32), (14, 384, 512, 34, 32), (14, 384, 512, 32, 32),
                           (14, 192, 256, 512, 64), (14, 192, 256, 194, 64), (14, 192, 256, 66, 64), (3, 50, 70, 98, 20)]:
    x = P.Feat.alloc(n, h, w, cin, DEV)
    x.buf[..., :cin] = torch.randn(n, h, w, cin, device=DEV, generator=g)
    cw = P.pack_conv(torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5), torch.randn(cout, device=DEV, generator=g), pad=1, prec=PR)
    out = P.Feat.alloc(n, h, w, cout, DEV)
    t = timeit(lambda: P.conv2d(x, cw, out, act=P.ACT_GELU))
    ref = P.conv2d(x, cw, act=P.ACT_GELU, force_generic=True)
    err = float((out.view() - ref.view()).abs().max())
    fl = 2.0 * n * h * w * cout * 9 * cin
    print(f"{n}x{h}x{w} {cin}->{cout}: {t:.3f} ms ({fl / t / 1e9:.0f} TF)  {P.L.load().prv2_last_kernel().decode()}  max|d| vs generic {err:.2e}", flush=True)
