import sys, os
sys.path.insert(0, "/root/repo")
import torch
from patchrefinerv2_amd import ops as P
DEV="cuda"; N=int(sys.argv[1]) if len(sys.argv) > 1 else 14; PR=P.L.PREC_NAMES["bf16x3"]
def timeit(fn, it=8):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for h, w, cin, cout, upc in [(384,512,256,128,256),(384,512,98,98,64),(192,256,194,194,128),(96,128,322,322,256),(48,64,642,642,512)][:int(os.environ.get("NL", 5))]:
    x = P.Feat.alloc(N, h, w, cin, DEV); x.buf[..., :cin] = torch.randn(N, h, w, cin, device=DEV)
    cw = P.pack_conv(torch.randn(cout, cin, 3, 3, device=DEV) / (3 * cin ** 0.5), torch.randn(cout, device=DEV), pad=1, prec=PR)
    lnp = (torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)) if cout <= 128 else None
    out = P.Feat.alloc(N, h, w, cout, DEV)
    u = P.Feat.alloc(N, h // 2, w // 2, upc, DEV); u.buf[..., :upc] = torch.randn(N, h // 2, w // 2, upc, device=DEV)
    t_ups = timeit(lambda: P.conv2d_ups(x, u, cw, out, act=P.ACT_GELU, ln=lnp)); k1 = P.L.load().prv2_last_kernel().decode()
    t_plain = timeit(lambda: P.conv2d(x, cw, out, act=P.ACT_GELU, ln=lnp)); k2 = P.L.load().prv2_last_kernel().decode()
    t_up = timeit(lambda: P.upsample_bilinear(u, h, w, out=x.slice(0, upc)))
    fl = 2.0 * N * h * w * cout * 9 * cin
    print(f"{N}x{h}x{w} {cin}->{cout} (+up {upc}): fused {t_ups:.3f} ms ({fl/t_ups/1e9:.0f} TF) {k1} | plain conv {t_plain:.3f} ms ({fl/t_plain/1e9:.0f} TF) {k2} + upsample {t_up:.3f} ms", flush=True)
