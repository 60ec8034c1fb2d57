"""Per-shape timing of the conv kernels (HIP events, many iterations).  usage: python tools/bench_conv.py [prec]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from patchrefinerv2_amd import ops, lib as L

prec = L.PREC_NAMES[sys.argv[1] if len(sys.argv) > 1 else "bf16x3"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ONLY = [int(i) for i in sys.argv[3].split(",")] if len(sys.argv) > 3 else None  # shape indices (PMC runs)
SHAPES = [  # (cin, cout, k, h, w)  -- the heavy layers of BiDirectionalFusion, dav2 cfg (SURVEY.md A.1 rescaled to 448^2)
    (512, 256, 3, 224, 224), (256, 256, 3, 224, 224), (256, 128, 3, 448, 448), (512, 256, 3, 112, 112),
    (98, 98, 3, 448, 448), (194, 194, 3, 224, 224), (512, 64, 3, 224, 224), (256, 256, 1, 448, 448),
    (322, 322, 3, 112, 112), (642, 642, 3, 56, 56), (128, 128, 3, 448, 448), (98, 32, 3, 448, 448), (770, 770, 3, 28, 28),
    (256, 32, 3, 448, 448), (34, 32, 3, 448, 448),
    (1024, 4096, 1, 1025, 1), (4096, 1024, 1, 1025, 1),
    (64, 32, 3, 448, 448), (128, 32, 3, 448, 448),  # 17, 18: persistent BN = 32 kernel
    (768, 3072, 1, 24, 31), (3072, 768, 1, 24, 31), (192, 768, 1, 98, 126), (256, 256, 1, 196, 259),  # 19-22: 1x1 GEMMs (b14)
    (1024, 1024, 1, 1037, 1), (1024, 3072, 1, 1037, 1), (4096, 1024, 1, 1037, 1), (384, 1536, 1, 4100, 1),  # 23-26: ViT at few tokens
]
for si, (cin, cout, k, h, w) in enumerate(SHAPES):
    if ONLY is not None and si not in ONLY:
        continue
    b = 1 if w == 1 else (14 if si >= 19 else B)
    x = ops.Feat.alloc(b, h, w, cin, "cuda")
    x.buf.normal_()
    cw = ops.pack_conv(torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5, None, prec=prec)
    y = ops.Feat.alloc(b, h, w, cout, "cuda")
    for force in (False, True):
        if force and not (k == 3 and w >= 24):
            continue
        for _ in range(2):
            ops.conv2d(x, cw, y, force_generic=force)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.conv2d(x, cw, y, force_generic=force)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * b * h * w * cin * cout * k * k
        print(f"{cin:5d}->{cout:4d} k{k} {h:4d}x{w:<4d} b{b} {'generic' if force or not (k == 3 and w >= 24) else 'halo   '} "
              f"{ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
