"""The GatedConvUnit as round 4 runs it (coarse half from the per-frame tap table): c256 conv -> X2 ``out``, tap gather -> ``pre``,
gate kernel on K = F with the pre-LayerNorm addend -- timing per stage and the executed TFLOP/s of the gate kernel.
   python tools/probes/gate_taps_bench.py [n h w]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from patchrefinerv2_amd import ops as P

DEV = "cuda"
n, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (14, 192, 256)
F_ = 256
PR = P.L.PREC_NAMES["bf16x3"]
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731


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


x, res, coarse = P.Feat(rn(n, h, w, F_)), P.Feat(rn(n, h, w, F_)), P.Feat(rn(1, h, w, F_))
wf = rn(F_, 2 * F_, 3, 3) / (3 * (2 * F_) ** 0.5)
cw_c = P.pack_conv(rn(F_, F_, 3, 3) / (3 * F_ ** 0.5), rn(F_) * 0.1, pad=1, prec=PR)
cw_a = P.pack_conv(wf[:, :F_].contiguous(), rn(F_) * 0.1, pad=1, prec=PR)
cw_t = P.pack_conv(P.coarse_tap_weight(wf[:, F_:]), None, prec=PR)
w3 = rn(F_, F_, 1, 1) / 16
gw, gb = P.pack_gate(w3), rn(F_) * 0.1
ln = (torch.rand(F_, device=DEV, generator=g) + 0.5, rn(F_) * 0.1)
org = torch.rand(n, 2, device=DEV, generator=g) * torch.tensor([3.0 * w, 3.0 * h], device=DEV)
boxes = torch.cat([org, org + torch.tensor([1.0 * w, 1.0 * h], device=DEV)], 1).contiguous()
taps = P.CoarseTaps(P.conv2d(coarse, cw_t), F_, (0.25, 0.25))
out = P.Feat(torch.empty((n, h, w, F_), device=DEV), x2=True)
pre = P.Feat(torch.empty((n, h, w, F_), device=DEV))
y = P.Feat(torch.empty((n, h, w, F_), device=DEV))
t_conv = timeit(lambda: P.conv2d(x, cw_c, out, relu_in=True, res=x))
cw_c6 = P.pack_conv3x3_f6(rn(F_, F_, 3, 3) / (3 * F_ ** 0.5), cw_c.bias)
t_conv6 = timeit(lambda: P.conv3x3_f6(x, cw_c6, out, relu_in=True, res=x))   # (the same layer on conv3x3_c256_f6_kernel; ``out`` keeps this one's bytes)
t_gather = timeit(lambda: taps.gather(boxes, 0.25, h, w, out=pre))
t_gate = timeit(lambda: P.conv3x3_ln_gate(out, cw_a, ln, gw, gb, y, act=P.ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_))
kern = P.L.load().prv2_last_kernel().decode()
yb = y.buf.clone()
# the same unit tail with its 3x3 conv in fp16 + fp6 (round 5, stage 2: conv3x3_c256_gate_f6_kernel)
cw_a6 = P.pack_conv3x3_f6(wf[:, :F_].contiguous(), cw_a.bias)
t_gate6 = timeit(lambda: P.conv3x3_ln_gate_f6(out, cw_a6, ln, gw, gb, y, act=P.ACT_RELU, mul=out, res=res, pre=pre, pre_cin=F_))
kern6 = P.L.load().prv2_last_kernel().decode()
d6 = float((y.buf - yb).norm() / yb.norm())
t_prep = timeit(lambda: P.CoarseTaps(P.conv2d(coarse, cw_t), F_, (0.25, 0.25)), it=5)
px = n * h * w
print(f"{n}x{h}x{w}: c256 conv -> X2 {t_conv:.3f} ms ({2.0 * px * 9 * F_ * F_ / t_conv / 1e9:.0f} TF), fp16 + fp6 {t_conv6:.3f} ms ({2.0 * px * 9 * F_ * F_ / t_conv6 / 1e9:.0f} TF) | tap gather {t_gather:.3f} ms ({px * F_ * 4 / t_gather / 1e6:.0f} GB/s written) | "
      f"{kern} K={F_} + pre {t_gate:.3f} ms ({2.0 * px * F_ * (9 * F_ + F_) / t_gate / 1e9:.0f} TF executed; the reference's 2F -> F unit: "
      f"{2.0 * px * F_ * (18 * F_ + F_) / t_gate / 1e9:.0f} TF algorithmic) | {kern6} {t_gate6:.3f} ms ({2.0 * px * F_ * (9 * F_ + F_) / t_gate6 / 1e9:.0f} TF executed, "
      f"{t_gate / t_gate6:.2f}x; rel-L2 against the bf16x3 kernel {d6:.1e}) | per-frame table (GEMM + knots, one unit) {t_prep:.3f} ms", flush=True)
