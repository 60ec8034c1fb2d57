"""GPU probe: the ViT-L blocks of the per-patch backbone (v1 workloads) as one program -- DepthAnythingV2 ViT-L encoder + head on a
batch of 448 x 448 crops, bf16x3.  Prints the per-kernel HIP-event table; run under rocprofv3 --pmc by tools/pmc_vit.sh.
    python tools/probes/vit_block_bench.py [batch=14] [reps=2] [ss=1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from patchrefinerv2_amd import ops, weights as W  # noqa: E402
from patchrefinerv2_amd.dav2 import DepthAnythingV2  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ops.SS_DISABLED = (len(sys.argv) > 3 and sys.argv[3] == "0")
mc = dict(encoder="vitl", features=256, out_channels=[256, 512, 1024, 1024], max_depth=80.0)
m = DepthAnythingV2(**mc, prec="bf16x3")
m.load_state_dict(W.synth_state_dict(W.dav2_spec("", mc), seed=0))
x = torch.rand(B, 3, 448, 448, generator=torch.Generator().manual_seed(1)).cuda()
m(x)
torch.cuda.synchronize()
ops.PROFILER.start(timed=True)
for _ in range(reps):
    m(x)
torch.cuda.synchronize()
ops.PROFILER.stop()
rows = sorted(ops.PROFILER.summary().items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print(f"ViT-L x {B} crops, pre-split path {'off' if ops.SS_DISABLED else 'on'}: {tot / reps:.2f} ms of matrix kernels per forward")
for k, v in rows:
    print(f"  {k:40s} {v['launches'] // reps:4d} launches {v['ms'] / reps:8.3f} ms {v['flops'] / max(v['ms'], 1e-9) / 1e9:7.1f} TFLOP/s")
