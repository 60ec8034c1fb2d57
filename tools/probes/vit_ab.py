"""GPU probe: same-process A/B of the ViT-L linears (gemm_ss: one tile per workgroup vs persistent workgroups at each DMA spread) and of the
attention block (fp32 qkv -> qkv_split -> attention vs split-swizzled qkv -> attention) at B crops of 1025 tokens.  Interleaved rounds.
    python tools/probes/vit_ab.py [B=14] [rounds=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchrefinerv2_amd import lib as L, ops as P  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NTOK = int(os.environ.get("NTOK", "1025"))
M = B * NTOK
pr = L.PREC_NAMES["bf16x3"]


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


shapes = [("qkv", 1024, 3072, dict()), ("proj", 1024, 1024, dict(gr=True)), ("fc1", 1024, 4096, dict(act=P.ACT_GELU, out_ss=True)), ("fc2", 4096, 1024, dict(gr=True))]
data = {}
for name, K, N, kw in shapes:
    x = torch.randn(M, K, device="cuda")
    cw = P.pack_conv(torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda") * 0.1, prec=pr)
    data[name] = (P.split_ss(x), cw, 1 + 0.1 * torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda"))
variants = [("one-tile", dict(PRV2_GSS_PERSIST="0"))] + [(f"persist ppb{p}", dict(PRV2_GSS_PERSIST="1", PRV2_GSS_PPB=str(p))) for p in (8, 4, 2, 1)]
if os.environ.get("GSS_ONLY"):  # (tools/probes/gss_ab2.sh: one variant, GEMMs only)
    variants = [v for v in variants if v[0].endswith(os.environ["GSS_ONLY"])]
res = {}
if os.environ.get("ATT_ONLY"):  # (the attention block alone)
    variants = []
for rnd in range(ROUNDS):
    for vname, env in variants:
        os.environ.update(env)
        for name, K, N, kw in shapes:
            xs, cw, gam, r = data[name]
            if kw.get("gr"):
                fn = lambda: P.gemm_ss(xs, cw, gamma=gam, res=r, out=r)  # noqa: E731  (in place, as the blocks call it)
            else:
                fn = lambda: P.gemm_ss(xs, cw, **kw)  # noqa: E731
            res.setdefault((vname, name), []).append(timed(fn))
if variants:
    print(f"gemm_ss at {M} rows (us per launch, {ROUNDS} interleaved rounds: min / median)")
for vname, _ in variants:
    line = f"  {vname:14s}"
    for name, K, N, kw in shapes:
        t = sorted(res[(vname, name)])
        line += f"  {name} {t[0]:7.1f} / {t[len(t) // 2]:7.1f} ({2.0 * M * K * N / t[0] / 1e6:5.0f} TF)"
    print(line, flush=True)

if os.environ.get("GSS_ONLY"):
    sys.exit(0)
# attention block: qkv Linear + attention
os.environ.pop("PRV2_GSS_PPB", None)
os.environ["PRV2_GSS_PERSIST"] = "1"
H = 16
xs, cw, _, _ = data["qkv"]
t_old, t_new, t_oa, t_na = [], [], [], []
for rnd in range(ROUNDS):
    qkv = P.gemm_ss(xs, cw)
    qss = P.gemm_ss_qkv(xs, cw, H)
    t_oa.append(timed(lambda: P.attention(qkv, B, NTOK, H, pr, out_ss=True)))
    t_na.append(timed(lambda: P.attention_qkv_ss(qss, B, NTOK, H)))
    t_old.append(timed(lambda: P.attention(P.gemm_ss(xs, cw), B, NTOK, H, pr, out_ss=True)))
    t_new.append(timed(lambda: P.attention_qkv_ss(P.gemm_ss_qkv(xs, cw, H), B, NTOK, H)))
fl = 4.0 * B * H * NTOK * NTOK * 64
print(f"attention alone (us): pre-pass path {min(t_oa):.1f} ({fl / min(t_oa) / 1e6:.0f} TF)   split-swizzled qkv {min(t_na):.1f} ({fl / min(t_na) / 1e6:.0f} TF)")
print(f"qkv Linear + attention (us): pre-pass path {min(t_old):.1f}   split-swizzled qkv {min(t_new):.1f}")
a = P.attention(P.gemm_ss(xs, cw), B, NTOK, H, pr, out_ss=True)
b = P.attention_qkv_ss(P.gemm_ss_qkv(xs, cw, H), B, NTOK, H)
print("bit-equal:", bool(torch.equal(a, b)))
