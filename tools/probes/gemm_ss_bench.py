"""GPU probe: ViT-L linear shapes on gemm16_kernel (fp32 operands, split in the kernel) vs gemm_ss_kernel (pre-split operands by
LDS-DMA), both tile sizes; checks bit-equality on the way.   python tools/probes/gemm_ss_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchrefinerv2_amd import lib as L, ops as P  # noqa: E402

pr = L.PREC_NAMES["bf16x3"]


def timeit(fn, n=10):
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


for M in (1037, 4100, 14350):
    for K, N in ((1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024)):
        x = torch.randn(M, K, device="cuda")
        cw = P.pack_conv(torch.randn(N, K, device="cuda") / K ** 0.5, torch.randn(N, device="cuda") * 0.1, prec=pr)
        xs = P.split_ss(x)
        y0 = P.linear(x, cw)
        t0 = timeit(lambda: P.linear(x, cw))
        row = f"M={M:6d} K={K:5d} N={N:5d}  gemm16 {t0:7.3f} ms {2.0 * M * K * N / t0 / 1e9:6.1f} TF |"
        for tile in ("128", "256"):
            for blocked in ("0", "1"):
                os.environ["PRV2_GEMM_SS_TILE"] = tile
                os.environ["PRV2_GEMM_SS_BLOCKED"] = blocked
                y1 = P.gemm_ss(xs, cw)
                t1 = timeit(lambda: P.gemm_ss(xs, cw))
                row += f" ss{tile}{'b' if blocked == '1' else ' '} {t1:7.3f} ms {2.0 * M * K * N / t1 / 1e9:6.1f} TF eq={bool(torch.equal(y0, y1))} |"
        os.environ.pop("PRV2_GEMM_SS_TILE")
        os.environ.pop("PRV2_GEMM_SS_BLOCKED")
        print(row, flush=True)
