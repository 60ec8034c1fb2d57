"""Same-box timing of the fused 32-channel full-resolution chains (csrc/chain32.hip) against the unfused launches they replace
(one 41-tile batch of the headline workload: 41 x 384 x 512 x 32).  python tools/bench_chain32.py [n_tiles]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from patchrefinerv2_amd import ops as P  # noqa: E402

DEV = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 41
H, W = 384, 512
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
prec = P.L.PREC_BF16X3


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


x = P.Feat(torch.randn((n, H, W, 32), device=DEV).relu_())
pre = P.Feat(torch.randn((n, H, W, 32), device=DEV) * 0.5)
p1 = torch.rand((n, 1, H, W), device=DEV) * 10
p2 = torch.rand((n, 1, H, W), device=DEV) * 10
w1, w2, w34 = r(32, 32, 3, 3) / 17, r(32, 32, 3, 3) / 17, r(32, 34, 3, 3) / 17.5
b1, b2, bo = r(32) * .1, r(32) * .1, r(32) * .1
lnw, lnb = 1 + .2 * r(32), .1 * r(32)
wg, wo, w3 = r(32, 32) / 5.6, r(32, 32) / 5.6, 1 + .3 * r(32)

cw = dict(w1=P.pack_chain32(w1, 0, DEV), w2=P.pack_chain32(w2, 1, DEV), wg=P.pack_chain32(wg, 1, DEV), wo=P.pack_chain32(wo, 1, DEV),
          consts=P.chain32_consts(DEV, b1=b1, ln1w=lnw, ln1b=lnb, b2=b2, bo=bo, w3=w3), b3=0.1)
out = P.Feat(torch.empty((n, H, W, 32), device=DEV))
depth = torch.empty((n, 1, H, W), device=DEV)
stamps = None
if os.environ.get("C32_STAMPS"):
    stamps = torch.zeros((256, 8, 4), dtype=torch.int64, device=DEV)
    os.environ["PRV2_C32_STAMPS"] = str(stamps.data_ptr())
t_c2f = timeit(lambda: P.chain32_c2f(x, cw, pre, out=out, depth=depth))
if stamps is not None:
    torch.cuda.synchronize()
    st = stamps.cpu().double()
    k = st[:, :, 3].clamp(min=1)
    for name, sl in (("stage 1 (waves 0-3)", slice(0, 4)), ("stage 2 (waves 4-7)", slice(4, 8))):
        c, w, b = [(st[:, sl, i] / k[:, sl]).mean().item() for i in range(3)]
        print(f"  c2f {name}: per tile compute {c:8.0f}  window store {w:7.0f}  barrier wait {b:8.0f} cycles (tiles per workgroup {k.mean().item():.0f})")
    stamps.zero_()

conv = P.pack_conv(w1, b1, device=DEV, prec=prec)
f0 = P.pack_conv(w2, b2, device=DEV, prec=prec)
gw = P.pack_gate(wg.to(DEV))
oc = P.pack_conv(wo, bo, device=DEV, prec=prec)
ln = (lnw.to(DEV), lnb.to(DEV))
w3d, b3d = w3.view(1, 32, 1, 1).to(DEV), torch.tensor([0.1], device=DEV)


def unfused_c2f():
    o = P.conv2d(x, conv, relu_in=True, res=x)
    y = P.conv3x3_ln_gate(o, f0, ln, gw, None, act=P.ACT_RELU, mul=o, pre=pre, pre_cin=32)
    l2 = P.conv2d(y, oc)
    P.conv2d_cout1(l2, w3d, b3d, 1)


t_c2f_ref = timeit(unfused_c2f)

ce = dict(w1=P.pack_chain32(w1, 0, DEV), w2=P.pack_chain32(w34, 1, DEV), wt=P.pack_chain32(w34, 2, DEV),
          consts=P.chain32_consts(DEV, b1=b1, ln1w=lnw, ln1b=lnb, b2=b2, ln2w=lnw, ln2b=lnb))
buf = P.Feat(torch.zeros((n, H, W, 100), device=DEV))
t_enc = timeit(lambda: P.chain32_enc(x, ce, pre, p1, p2, out=buf.slice(64, 32)))
if stamps is not None:
    torch.cuda.synchronize()
    st = stamps.cpu().double()
    k = st[:, :, 3].clamp(min=1)
    for name, sl in (("stage 1 (waves 0-3)", slice(0, 4)), ("stage 2 (waves 4-7)", slice(4, 8))):
        c, w, b = [(st[:, sl, i] / k[:, sl]).mean().item() for i in range(3)]
        print(f"  enc {name}: per tile compute {c:8.0f}  window store {w:7.0f}  barrier wait {b:8.0f} cycles")

e1 = P.pack_conv(w1, b1, device=DEV, prec=prec)
e2 = P.pack_conv(w34, b2, device=DEV, prec=prec)
cat2 = P.Feat(torch.zeros((n, H, W, 36), device=DEV), 34)
p1f, p2f = P.Feat(p1.view(n, H, W, 1)), P.Feat(p2.view(n, H, W, 1))


def unfused_enc():
    P.conv2d_pre(x, e1, pre, cat2.slice(0, 32), act=P.ACT_GELU, ln=ln, pre_cin=32)
    P.depth_pair_fill(p1f, p2f, cat2, 32)
    P.conv2d(cat2, e2, buf.slice(64, 32), act=P.ACT_GELU, ln=ln)


t_enc_ref = timeit(unfused_enc)
px = n * H * W
print(f"chain32_c2f  {t_c2f:7.3f} ms  (unfused {t_c2f_ref:7.3f} ms)  {px * (128 + 128 + 132) / t_c2f / 1e9:6.1f} GB/s algorithmic  "
      f"{2.0 * px * (2 * 9 * 1024 + 2 * 1024) / t_c2f / 1e9:6.1f} TF")
print(f"chain32_enc  {t_enc:7.3f} ms  (unfused {t_enc_ref:7.3f} ms)  {px * (128 + 128 + 8 + 128) / t_enc / 1e9:6.1f} GB/s algorithmic  "
      f"{2.0 * px * (9 * 1024 + 9 * 34 * 32) / t_enc / 1e9:6.1f} TF")
