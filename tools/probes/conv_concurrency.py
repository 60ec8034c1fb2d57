"""Does a conv layer give bit-identical results when other kernels run beside it on other streams?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchrefinerv2_amd import ops as P, lib as L
pr = L.PREC_NAMES["bf16x3"]
torch.manual_seed(0)
shapes = [(14, 196, 256, 512, 256, 3), (14, 196, 259, 512, 256, 3), (14, 196, 259, 256, 256, 3), (14, 196, 259, 512, 128, 3), (7, 196, 259, 512, 256, 3), (14, 196, 288, 512, 256, 3)]
layers = []
for (n, h, w, cin, cout, k) in shapes:
    x = P.Feat.alloc(n, h, w, cin, "cuda"); x.buf.normal_()
    cw = P.pack_conv(torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5, torch.randn(cout, device="cuda"), prec=pr)
    layers.append((x, cw, (n, h, w, cout)))
ref = [P.conv2d(x, cw, act=P.ACT_GELU).buf.clone() for x, cw, _ in layers]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
bad = [0] * len(layers)
for it in range(6):
    outs = []
    for i, (x, cw, _) in enumerate(layers):
        with torch.cuda.stream(streams[i % 3]):
            y = P.conv2d(x, cw, act=P.ACT_GELU)
            up = P.upsample_bilinear(y, y.h // 2, y.w // 2)      # some HBM-bound neighbours
            outs.append(y)
    torch.cuda.synchronize()
    for i, y in enumerate(outs):
        if not torch.equal(y.buf, ref[i]):
            bad[i] += 1
            d = (y.buf - ref[i]).abs()
            print(f"iter {it} layer {shapes[i]}: differs, ndiff {int((d > 0).sum())} max {float(d.max()):.3e}")
print("mismatching iterations per layer:", bad)
