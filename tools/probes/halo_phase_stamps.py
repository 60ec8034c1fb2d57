"""Diagnostic (needs a library built with the s_memtime stamps patch, variants/lib_stamps.so): per-workgroup phase times
of the 3x3 halo kernel: prologue / main loop / C tile to LDS / store loop incl. drain."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
stamps = torch.zeros(200000 * 6, dtype=torch.int64, device="cuda")
os.environ["PRV2_STAMP_PTR"] = hex(stamps.data_ptr())
from patchrefinerv2_amd import ops as P, lib as L
pr = L.PREC_NAMES["bf16x3"]
for (cin, cout, h, w, b) in ((512, 256, 64, 64, 1), (512, 256, 128, 128, 1), (512, 256, 224, 224, 2), (512, 256, 224, 224, 27), (256, 128, 448, 448, 27), (98, 98, 448, 448, 27), (64, 32, 448, 448, 27)):
    x = P.Feat.alloc(b, h, w, cin, "cuda"); x.buf.normal_()
    cw = P.pack_conv(torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5, None, prec=pr)
    y = P.Feat.alloc(b, h, w, cout, "cuda")
    for _ in range(3):
        P.conv2d(x, cw, y)
    torch.cuda.synchronize()
    nblk = b * (h // 8) * (w // 32) * ((cout + 127) // 128)
    s = stamps[: nblk * 6].view(nblk, 6).cpu().double()
    d = (s[:, 1:] - s[:, :-1]) / 1000.0  # s_memtime = shader-clock cycles -> kcycles
    print(f"{cin}->{cout} {h}x{w}: {nblk} workgroups; kcycles per workgroup (median): prologue {d[:,0].median():.1f}  main loop {d[:,1].median():.1f}  "
          f"C tile -> LDS {d[:,2].median():.1f}  store loop {d[:,3].median():.1f}  store drain {d[:,4].median():.1f}")
