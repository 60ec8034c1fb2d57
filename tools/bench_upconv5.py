"""Same-box timing of ops.upconv5x5 (csrc/upconv5.hip) against the two launches it replaces -- upconv3x3 256 -> 128 (+ border bias) and the
128 -> 32 conv -- on one 41-tile batch of the headline workload (192 x 256 -> 384 x 512).   python tools/bench_upconv5.py [n_tiles]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchrefinerv2_amd import ops as P  # noqa: E402

DEV = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 41
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
prec = P.L.PREC_BF16X3
u = P.Feat(torch.randn((n, 192, 256, 256), device=DEV))
w1, b1, tb = r(128, 256, 3, 3) / 48, r(128) * .1, r(9, 128) * .1
w2, b2 = r(32, 128, 3, 3) / 34, r(32) * .1
cw5 = P.compose_upconv5x5(w1, b1, tb, w2, b2, DEV, prec)
c1 = P.pack_conv(w1, b1 + tb.sum(0), pad=1, device=DEV, prec=prec)
c2 = P.pack_conv(w2, b2, device=DEV, prec=prec)
tbd = tb.contiguous().to(DEV)
out = P.Feat(torch.empty((n, 384, 512, 32), device=DEV))


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def old():
    t = P.upconv3x3(u, 384, 512, c1)
    P.conv_border_bias(t, tbd)
    P.conv2d(t, c2, out, act=P.ACT_RELU)


t_new = timeit(lambda: P.upconv5x5(u, 384, 512, cw5, out=out, act=P.ACT_RELU))
a = out.buf.clone()
t_old = timeit(old)
d = (out.buf - a).abs().max().item()
print(f"upconv5x5 {t_new:7.3f} ms   upconv3x3 + border bias + conv 128->32 {t_old:7.3f} ms   max|d| {d:.2e} (scale {a.abs().max().item():.2f})")
