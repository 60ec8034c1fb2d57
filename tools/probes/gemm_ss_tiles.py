"""GPU probe: gemm_ss tile / stage variants on the ViT-L linear shapes at 14 crops (PRV2_GEMM_SS_TILE codes), interleaved rounds.
   python tools/probes/gemm_ss_tiles.py [codes ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchrefinerv2_amd import lib as L, ops as P  # noqa: E402

pr = L.PREC_NAMES["bf16x3"]
codes = sys.argv[1:] or ["256", "128", "2563", "1283"]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M in (14350, 11074):
    for K, N in ((1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024)):
        x = torch.randn(M, K, device="cuda")
        cw = P.pack_conv(torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda") * 0.1, prec=pr)
        xs = P.split_ss(x)
        y0 = P.linear(x, cw)
        row = f"M={M:6d} K={K:5d} N={N:5d} "
        best = {}
        for rnd in range(2):
            for c in codes:
                os.environ["PRV2_GEMM_SS_TILE"] = c
                y1 = P.gemm_ss(xs, cw)
                assert torch.equal(y0, y1), c
                t = timeit(lambda: P.gemm_ss(xs, cw))
                best[c] = min(best.get(c, 1e9), t)
        os.environ.pop("PRV2_GEMM_SS_TILE")
        print(row + " | ".join(f"{c}: {best[c]:.3f} ms {2.0 * M * K * N / best[c] / 1e9:6.1f} TF" for c in codes), flush=True)
