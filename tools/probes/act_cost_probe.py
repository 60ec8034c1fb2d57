"""Epilogue cost of the activation on the short-K decoder convs (none / ReLU / exact-erf GELU): GELU +4..9 % on 98->98 / 194->194;
ReLU is FASTER than none -- zeros in the output lower the power draw and the chip clocks higher.   python tools/probes/act_cost_probe.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from patchrefinerv2_amd import ops as P
DEV="cuda"
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
g = torch.Generator(device=DEV).manual_seed(0)
PR = P.L.PREC_NAMES["bf16x3"]
for n,h,w,cin,cout in [(14,384,512,98,98),(14,192,256,194,194),(14,384,512,98,32),(14,96,128,322,322)]:
    x = P.Feat.alloc(n,h,w,cin,DEV); x.buf[..., :cin] = torch.randn(n,h,w,cin,device=DEV,generator=g)
    cw = P.pack_conv(torch.randn(cout,cin,3,3,device=DEV,generator=g)/(3*cin**0.5), None, pad=1, prec=PR)
    out = P.Feat.alloc(n,h,w,cout,DEV)
    fl = 2.0*n*h*w*cout*9*cin
    for name, act in (("none",P.ACT_NONE),("relu",P.ACT_RELU),("gelu",P.ACT_GELU)):
        t = timeit(lambda: P.conv2d(x, cw, out, act=act))
        print(f"{cin}->{cout} {h}x{w} act={name}: {t:.3f} ms ({fl/t/1e9:.0f} TF)", flush=True)
