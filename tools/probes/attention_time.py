"""Same-box timing of the bf16x3 attention kernel: BEiT-L shape with the packed bias image (769 tokens) and DINOv2 ViT-L (1025 tokens, no bias),
41 crops x 16 heads.  python tools/probes/attention_time.py"""
import sys

import torch

sys.path.insert(0, ".")
from patchrefinerv2_amd import ops as P  # noqa: E402

PR = P.L.PREC_NAMES["bf16x3"]
g = torch.Generator().manual_seed(0)
for name, b, ntok, heads, with_bias in (("BEiT-L 769 tok + bias image", 41, 769, 16, True), ("DINOv2-L 1025 tok", 41, 1025, 16, False)):
    qkv = torch.randn(b * ntok, 3 * heads * 64, generator=g).cuda()
    bias = None
    if with_bias:
        ld = (ntok + 63) // 64 * 64
        bias = P.pack_attention_bias((torch.randn(heads, ntok, ld, generator=g) * 0.5).cuda().contiguous(), ntok)
    for _ in range(3):
        out = P.attention(qkv, b, ntok, heads, PR, bias=bias)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = P.attention(qkv, b, ntok, heads, PR, bias=bias)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    fl = 4.0 * b * heads * ntok * ntok * 64
    print(f"{name}: {min(ts):.3f} ms ({fl / min(ts) / 1e9:.1f} TF incl. the qkv split pre-pass) checksum {float(out.double().abs().sum()):.6e}", flush=True)
