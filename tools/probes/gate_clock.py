"""see gate_clock.sh: per kernel and operand kind -- wall time per launch, in-kernel clock (d s_memtime / d s_memrealtime x 100 MHz,
median over workgroups, after >= 2 s of back-to-back launches) and shader cycles per workgroup."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
stamps = torch.zeros(8000000 + 400000, dtype=torch.int64, device="cuda")
os.environ["PRV2_STAMP_PTR"] = hex(stamps.data_ptr())
from patchrefinerv2_amd import lib as L
L.LIB_PATH = os.environ["PRV2_LIB_OVERRIDE"]
from patchrefinerv2_amd import ops as P
PR = L.PREC_NAMES["bf16x3"]
n, h, w, F_ = 14, 192, 256, 256
nblk = n * (h // 8) * (w // 16)
for zero in (False, True):
    mk = (lambda *s: torch.zeros(*s, device="cuda")) if zero else (lambda *s: torch.randn(*s, device="cuda"))
    x = P.Feat(mk(n, h, w, F_))
    coarse = P.Feat(mk(1, h, w, F_))
    boxes = torch.tensor([[3.0 * i, 1.0 * i, 3.0 * i + w / 4.0, 1.0 * i + h / 4.0] for i in range(n)], device="cuda")
    cw_c = P.pack_conv(mk(F_, F_, 3, 3) / 48, mk(F_), pad=1, prec=PR)
    cw_f = P.pack_conv(mk(F_, 2 * F_, 3, 3) / 68, mk(F_), pad=1, prec=PR)
    gw, gb = P.pack_gate(mk(F_, F_, 1, 1) / 16), mk(F_)
    ln = (torch.rand(F_, device="cuda") + 0.5, mk(F_) * 0.1)
    res = P.Feat(mk(n, h, w, F_))
    for x2 in (False, True):
        cat = P.Feat.alloc(n, h, w, 2 * F_, "cuda")
        cat.x2 = x2
        P.roi_align(coarse, boxes, 1.0, h, w, out=cat.slice(F_, F_))
        out = P.conv2d(x, cw_c, cat.slice(0, F_), relu_in=True, res=x)
        y = P.Feat.alloc(n, h, w, F_, "cuda")
        run = lambda: P.conv3x3_ln_gate(cat, cw_f, ln, gw, gb, y, act=P.ACT_RELU, mul=out, res=res)
        run()
        kname = L.load().prv2_last_kernel().decode()
        torch.cuda.synchronize()
        t0 = time.time()
        while time.time() - t0 < 2.5:   # the chip settles at the clock it can hold
            for _ in range(20):
                run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        st = stamps[8000000: 8000000 + nblk * 4].view(nblk, 2, 2).cpu().double()
        dcyc, dreal = st[:, 1, 0] - st[:, 0, 0], st[:, 1, 1] - st[:, 0, 1]
        clk = (dcyc / dreal * 0.1).median().item()
        print(f"{'zeros ' if zero else 'random'} {kname:44s} {ms:.3f} ms per launch ({2.0 * n * h * w * F_ * (9 * 2 * F_ + F_) / ms / 1e9:.0f} TF)   "
              f"in-kernel clock {clk:.2f} GHz   {dcyc.median().item() / 1000:.1f} kcycles per workgroup", flush=True)
