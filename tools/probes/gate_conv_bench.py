"""GatedConvUnit tail: fused kernel (conv3x3_gate.hip) vs the unfused sequence conv3x3 -> LayerNorm -> 1x1 gate GEMM.
   python tools/probes/gate_conv_bench.py [n h w cin]"""
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


def main():
    shapes = [(14, 192, 256, 512), (14, 96, 128, 512), (14, 48, 64, 512), (14, 24, 32, 512), (14, 192, 256, 256)]
    if len(sys.argv) == 5:
        shapes = [tuple(int(a) for a in sys.argv[1:5])]
    g = torch.Generator(device=DEV).manual_seed(0)
    PR = P.L.PREC_NAMES["bf16x3"]
    for n, h, w, cin in shapes:
        x = P.Feat(torch.randn(n, h, w, cin, device=DEV, generator=g))
        cw0 = P.pack_conv(torch.randn(256, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5), torch.randn(256, device=DEV, generator=g), pad=1, prec=PR)
        w3 = torch.randn(256, 256, 1, 1, device=DEV, generator=g) / 16
        cw3 = P.pack_conv(w3, torch.randn(256, device=DEV, generator=g), prec=PR)
        gw = P.pack_gate(w3)
        ln = (torch.rand(256, device=DEV, generator=g) + 0.5, torch.randn(256, device=DEV, generator=g) * 0.1)
        mul, res = P.Feat(torch.randn(n, h, w, 256, device=DEV, generator=g)), P.Feat(torch.randn(n, h, w, 256, device=DEV, generator=g))
        out = P.Feat.alloc(n, h, w, 256, DEV)
        tmp = P.Feat.alloc(n, h, w, 256, DEV)
        fl = 2.0 * n * h * w * 256 * (9 * cin)
        t_conv = timeit(lambda: P.conv2d(x, cw0, tmp))
        t_unf = timeit(lambda: P.conv2d(P.conv2d(x, cw0, tmp, act=P.ACT_RELU, ln=ln), cw3, out, act=P.ACT_SIGMOID, mul=mul, res=res))
        ref = out.buf.clone()
        t_ln = timeit(lambda: P.conv3x3_ln_gate(x, cw0, ln, None, None, tmp, act=P.ACT_RELU))
        t_fus = timeit(lambda: P.conv3x3_ln_gate(x, cw0, ln, gw, cw3.bias, out, act=P.ACT_RELU, mul=mul, res=res))
        err = float((out.buf - ref).abs().max())
        t_nomul = timeit(lambda: P.conv3x3_ln_gate(x, cw0, ln, gw, cw3.bias, out, act=P.ACT_RELU))
        print(f"{n}x{h}x{w} {cin}->256: conv<128> alone {t_conv:.3f} ms ({fl / t_conv / 1e9:.0f} TF) | conv+LN+gate unfused {t_unf:.3f} ms | "
              f"BN256 conv+LN {t_ln:.3f} ms ({fl / t_ln / 1e9:.0f} TF) | fused tail {t_fus:.3f} ms (without mul / res {t_nomul:.3f})  (max|fused - unfused| {err:.2e})", flush=True)


main()
