"""Where do the two dispatch routes (ctypes / torch.ops) first differ?  Every ops.* call's output is checksummed in call order.
   python tools/probes/route_probe.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle.cases import rand_image
from patchrefinerv2_amd import ops, dav2, weights as W
from patchrefinerv2_amd.dav2 import DepthAnythingV2
mc = dict(encoder="vits", features=64, out_channels=[48, 96, 192, 384], max_depth=80.0, vit=dict(depth=4, taps=[0, 1, 2, 3]))
m = DepthAnythingV2(**mc, prec="bf16x3")
m.load_state_dict(W.synth_state_dict(W.dav2_spec("", mc), seed=5))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x_img = rand_image(3, 3, 448, 448).cuda()[:B].clone()
LOG = []
NAMES = ["conv2d", "linear", "upsample_bilinear", "conv2d_cout1", "layernorm_rows", "attention", "patchify", "assemble_tokens", "crop_resize", "add", "layernorm_feat"]


def cks(o):
    if isinstance(o, ops.Feat):
        o = o.buf
    if isinstance(o, torch.Tensor):
        return int(o.contiguous().view(torch.int32).sum(dtype=torch.int64))
    return None


def wrap(name):
    f = getattr(ops, name)

    def g(*a, **k):
        r = f(*a, **k)
        tgt = r
        if name == "layernorm_rows":
            tgt = a[8]
        if name == "crop_resize":
            tgt = a[-1] if not k.get("out") else k["out"]
        shp = tuple(tgt.buf.shape) if isinstance(tgt, ops.Feat) else (tuple(tgt.shape) if isinstance(tgt, torch.Tensor) else None)
        LOG.append((name, shp, cks(tgt)))
        return r
    setattr(ops, name, g)


for n in NAMES:
    wrap(n)
runs = {}
for route in ("ctypes", "torch"):
    ops.DISPATCH = route
    LOG.clear()
    m(x_img)
    torch.cuda.synchronize()
    runs[route] = list(LOG)
a, b = runs["ctypes"], runs["torch"]
print("calls", len(a), len(b))
for i, (p, q) in enumerate(zip(a, b)):
    if p != q:
        print("first difference at call", i, p, q)
        for j in range(max(0, i - 3), min(len(a), i + 3)):
            print("   ", j, a[j][:2], "equal" if a[j] == b[j] else "DIFF")
        break
else:
    print("all equal")
