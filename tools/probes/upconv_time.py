"""Same-box timing of conv3x3(bilinear x2 upsample) on the two kernels: prv2_conv2d_ups (interpolates inside its halo loader, MFMA work at
the output resolution) vs prv2_upconv3x3 (tap GEMMs at the source resolution + gather).  python tools/probes/upconv_time.py [n]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from patchrefinerv2_amd import ops as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 41
PR = P.L.PREC_NAMES["bf16x3"]
CASES = [(256, 128, 192, 256), (256, 290, 192, 256), (256, 322, 48, 64), (512, 642, 24, 32), (128, 194, 96, 128), (64, 98, 192, 256)]
g = torch.Generator().manual_seed(0)
for cin, cout, h, w in CASES:
    u = P.Feat.from_nchw(torch.randn(n, cin, h, w, generator=g).relu_().cuda())
    cw = P.pack_conv((torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).cuda(), None, pad=1, prec=PR)
    H, W = 2 * h, 2 * w
    out = P.Feat.alloc(n, H, W, cout, "cuda")
    res = {}
    for name, fn in (("conv2d_ups", lambda: P.conv2d_ups(P.UpsOnly(u, H, W), u, cw, out=out)), ("upconv3x3", lambda: P.upconv3x3(u, H, W, cw, out=out))):
        if name == "conv2d_ups" and not P.conv2d_ups_supported(P.UpsOnly(u, H, W), u, cw):
            continue
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        res[name] = (min(ts), out.buf.clone())
    fl = 2.0 * n * H * W * cout * cin * 9
    line = f"{cin:4d}->{cout:4d} {n}x{H}x{W}: " + "  ".join(f"{k} {v[0]:7.3f} ms ({fl / v[0] / 1e9:6.1f} TF of the reference graph)" for k, v in res.items())
    if len(res) == 2:
        a, b = res["conv2d_ups"][1], res["upconv3x3"][1]
        line += f"  max|d| {float((a - b).abs().max()):.2e} (scale {float(a.abs().max()):.2f})"
    print(line, flush=True)
