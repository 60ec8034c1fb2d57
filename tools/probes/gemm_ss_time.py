"""GPU probe: time of gemm_ss on the ViT-L linear shapes at 14 crops, no result check (for ablation builds).  [tile code]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchrefinerv2_amd import lib as L, ops as P  # noqa: E402

pr = L.PREC_NAMES["bf16x3"]
if len(sys.argv) > 1:
    os.environ["PRV2_GEMM_SS_TILE"] = sys.argv[1]
M = 14350
for K, N in ((1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024)):
    x = torch.randn(M, K, device="cuda")
    cw = P.pack_conv(torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda") * 0.1, prec=pr)
    xs = P.split_ss(x)
    for _ in range(3):
        P.gemm_ss(xs, cw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        P.gemm_ss(xs, cw)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"M={M} K={K} N={N}: {t:.3f} ms {2.0 * M * K * N / t / 1e9:6.1f} TF", flush=True)
