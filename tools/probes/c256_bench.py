"""conv3x3_c256_kernel (ResidualConvUnit conv: y = conv3x3(relu(x)) + bias + res, 256 -> 256) on the frame's shapes.
   python tools/probes/c256_bench.py [mode]      mode: bf16x3 (default) | f16f6 | both (also prints rel-L2 against fp64 on the smallest shape)"""
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
    modes = sys.argv[1:] or ["bf16x3"]
    if modes == ["both"]:
        modes = ["bf16x3", "f16f6"]
    g = torch.Generator(device=DEV).manual_seed(0)
    w = torch.randn(256, 256, 3, 3, device=DEV, generator=g) / 48
    b = torch.randn(256, device=DEV, generator=g)
    for n, h, wd in [(14, 192, 256), (14, 96, 128), (14, 48, 64), (14, 24, 32), (81, 24, 32)]:
        x = P.Feat(torch.randn(n, h, wd, 256, device=DEV, generator=g))
        res = P.Feat(torch.randn(n, h, wd, 256, device=DEV, generator=g))
        out = P.Feat.alloc(n, h, wd, 256, DEV)
        fl = 2.0 * n * h * wd * 256 * 9 * 256
        line = f"{n}x{h}x{wd} 256->256:"
        for m in modes:
            if m == "f16f6":
                cw = P.pack_conv3x3_f6(w, b)
                t = timeit(lambda: P.conv3x3_f6(x, cw, out, relu_in=True, res=res))
            else:
                cw = P.pack_conv(w, b, pad=1, prec=P.L.PREC_NAMES[m])
                t = timeit(lambda: P.conv2d(x, cw, out, relu_in=True, res=res))
            line += f"  {m} {t:.3f} ms ({fl / t / 1e9:.0f} TF, {P.L.load().prv2_last_kernel().decode()})"
            if n * h * wd <= 14 * 24 * 32:
                ref = torch.nn.functional.conv2d(torch.relu(x.buf.double()).permute(0, 3, 1, 2), w.double(), b.double(), padding=1).permute(0, 2, 3, 1) + res.buf.double()
                line += f" rel-L2 {float((out.buf.double() - ref).norm() / ref.norm()):.2e}"
        print(line, flush=True)


main()
