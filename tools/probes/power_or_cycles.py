"""GPU probe: which of the frame's conv layers are POWER-bound (the chip lowers its clock: faster on all-zero operands, cycle savings
return nothing) and which are CYCLE-bound (same time on zeros: issue-side work pays).   python tools/probes/power_or_cycles.py [batch]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from patchrefinerv2_amd import ops as P

DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 14
PR = P.L.PREC_NAMES["bf16x3"]


def timeit(fn, it=8):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


# (h, w, cin, cout, upsampled channels or 0, fused LayerNorm, activation)
LAYERS = [(384, 512, 256, 128, 256, True, "gelu"), (384, 512, 98, 98, 64, True, "gelu"), (192, 256, 194, 194, 128, True, "gelu"), (96, 128, 322, 322, 256, True, "gelu"),
          (48, 64, 642, 642, 512, True, "gelu"), (192, 256, 256, 256, 0, False, "none"), (192, 256, 512, 64, 0, True, "gelu"), (192, 256, 194, 64, 0, True, "gelu"),
          (384, 512, 128, 32, 0, True, "gelu"), (384, 512, 64, 32, 0, True, "gelu"), (384, 512, 98, 32, 0, True, "gelu"), (384, 512, 32, 32, 0, True, "gelu"),
          (96, 128, 322, 128, 0, True, "gelu")]
for h, w, cin, cout, upc, ln, act in LAYERS:
    res = {}
    for zero in (False, True):
        mk = (lambda *s: torch.zeros(*s, device=DEV)) if zero else (lambda *s: torch.randn(*s, device=DEV))
        x = P.Feat.alloc(N, h, w, cin, DEV)
        x.buf[..., :cin] = mk(N, h, w, cin)
        cw = P.pack_conv(mk(cout, cin, 3, 3) / (3 * cin ** 0.5), mk(cout), pad=1, prec=PR)
        lnp = (torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)) if ln and cout <= 128 else None
        out = P.Feat.alloc(N, h, w, cout, DEV)
        a = P.ACT_GELU if act == "gelu" else P.ACT_NONE
        if upc:
            u = P.Feat.alloc(N, h // 2, w // 2, upc, DEV)
            u.buf[..., :upc] = mk(N, h // 2, w // 2, upc)
            fn = lambda: P.conv2d_ups(x, u, cw, out, act=a, ln=lnp)
        else:
            fn = lambda: P.conv2d(x, cw, out, act=a, ln=lnp)
        res[zero] = timeit(fn)
        kname = P.L.load().prv2_last_kernel().decode()
    fl = 2.0 * N * h * w * cout * 9 * cin
    print(f"{N}x{h}x{w} {cin}->{cout}{' (+up %d)' % upc if upc else ''}: random {res[False]:.3f} ms ({fl / res[False] / 1e9:.0f} TF)  zeros {res[True]:.3f} ms "
          f"({fl / res[True] / 1e9:.0f} TF)  zeros/random {res[True] / res[False]:.2f}  {kname}", flush=True)
