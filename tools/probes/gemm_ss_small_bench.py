"""gemm_ss on the token counts of a single image (the coarse forwards: 769 BEiT / 1037 DINOv2 tokens): 128 x 128 tiles vs 64 x 64 tiles
with 2 / 4 LDS stages (PRV2_GEMM_SS_SMALL, PRV2_GEMM_SS_DEEP), bit-equality against gemm16 checked on the way.   python tools/probes/gemm_ss_small_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchrefinerv2_amd import lib as L, ops as P  # noqa: E402

pr = L.PREC_NAMES["bf16x3"]


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


for M in (769, 1037, 257):
    for K, N in ((1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024), (384, 1536), (1536, 384)):
        x = torch.randn(M, K, device="cuda")
        cw = P.pack_conv(torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda") * 0.1, prec=pr)
        xs = P.split_ss(x)
        y0 = P.linear(x, cw)
        row = f"M={M:5d} K={K:5d} N={N:5d} |"
        for deep in ("0", "2", "4", "auto"):  # 128 x 128 tiles | 64 x 64 with 2 / 4 LDS stages | the library's rule
            os.environ["PRV2_GEMM_SS_SMALL"] = "0" if deep == "0" else "1"
            os.environ.pop("PRV2_GEMM_SS_DEEP", None)
            if deep in ("2", "4"):
                os.environ["PRV2_GEMM_SS_DEEP"] = deep
            y1 = P.gemm_ss(xs, cw)
            t1 = timeit(lambda: P.gemm_ss(xs, cw))
            row += f" {deep}: {t1 * 1e3:6.1f} us {2.0 * M * K * N / t1 / 1e9:6.1f} TF {'bit-equal' if torch.equal(y0, y1) else 'DIFFERS'} |"
        print(row, flush=True)
