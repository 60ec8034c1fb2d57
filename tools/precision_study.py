"""CPU error-budget study (SURVEY.md 7 'Accuracy vs MFMA rate'): emulate the matrix-kernel
arithmetic modes on the oracle graph and report per-pixel AbsRel of the final depth map vs fp32.
   bf16   : operands rounded to bf16, fp32 accumulate
   bf16x3 : hi/lo bf16 split, hi*hi + hi*lo + lo*hi
   fp16x3 : same with fp16 halves (22-bit effective mantissa; range-limited)
"""
import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import tiling as o_tiling, dav2, fusion, mnv4
from oracle.cases import E2E_V2, E2E_V1, e2e_v2_sd, e2e_v1_sd, rand_image
from patchrefinerv2_amd import weights as W

torch.set_grad_enabled(False)
_conv2d, _linear, _convt = F.conv2d, F.linear, F.conv_transpose2d
MODE = ["f32"]

def split(t, dt):
    hi = t.to(dt).float()
    lo = (t - hi).to(dt).float()
    return hi, lo

def wrap(fn):
    def f(x, w, b=None, *a, **k):
        m = MODE[0]
        if m == "f32" or (k.get("groups", 1) != 1):
            return fn(x, w, b, *a, **k)
        dt = torch.bfloat16 if m.startswith("bf16") else torch.float16
        xh, xl = split(x, dt); wh, wl = split(w, dt)
        y = fn(xh, wh, None, *a, **k)
        if m.endswith("x3"):
            y = y + fn(xh, wl, None, *a, **k) + fn(xl, wh, None, *a, **k)
        if b is not None:
            y = y + (b.view(1, -1, 1, 1) if y.dim() == 4 else b)
        return y
    return f

F.conv2d, F.linear, F.conv_transpose2d = wrap(_conv2d), wrap(_linear), wrap(_convt)
# attention matmuls (q@k^T, p@v)
def attention(sd, p, x, heads):
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * ((C // heads) ** -0.5), qkv[1], qkv[2]
    def mm(a, b):
        m = MODE[0]
        if m == "f32": return a @ b
        dt = torch.bfloat16 if m.startswith("bf16") else torch.float16
        ah, al = split(a, dt); bh, bl = split(b, dt)
        y = ah @ bh
        if m.endswith("x3"): y = y + ah @ bl + al @ bh
        return y
    attn = mm(q, k.transpose(-2, -1)).softmax(dim=-1)
    x = mm(attn, v).transpose(1, 2).reshape(B, N, C)
    return F.linear(x, sd[p + "proj.weight"], sd[p + "proj.bias"])
dav2.attention = attention

def run(kind):
    if kind == "v2":
        c, sd = E2E_V2, e2e_v2_sd()
        cfg = W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]})
        m = o_tiling.OraclePatchRefinerPlus(sd, cfg, patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"])
        mode = "r4"
    else:
        c, sd = E2E_V1, e2e_v1_sd()
        cfg = W.dav2_cfg({**c["da2_cfg"], "max_depth": c["max_depth"]})
        m = o_tiling.OraclePatchRefiner(sd, cfg, cfg, patch_process_shape=c["pps"], image_raw_shape=c["raw"], patch_split_num=c["split"])
        mode = "r8"
    hr = rand_image(c["seed"], 1, *c["raw"]); lr = m.resizer(hr)
    res = {}
    for md in ("f32", "bf16", "bf16x3", "fp16x3"):
        MODE[0] = md
        random.seed(621)
        out, log = m(mode="infer", cai_mode=mode, process_num=4, tile_cfg=dict(image_raw_shape=c["raw"], patch_split_num=c["split"]), image_lr=lr, image_hr=hr)
        res[md] = (out, log["coarse_prediction"])
    ref, refc = res["f32"]
    for md in ("bf16", "bf16x3", "fp16x3"):
        o, oc = res[md]
        ar = float(((o - ref).abs() / ref)[ref > 1e-3].mean()); mx = float(((o - ref).abs() / ref)[ref > 1e-3].max())
        arc = float(((oc - refc).abs() / refc).mean())
        print(f"{kind} {md:7s} final AbsRel {ar:.3e} (max rel {mx:.3e})   coarse AbsRel {arc:.3e}")

for k in sys.argv[1:] or ["v1", "v2"]:
    run(k)
