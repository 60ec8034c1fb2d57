"""Standalone reproducer ATTEMPT of the packed-fp32 miscompute (VERDICT r05 #6c): tap_gather_kernel built WITH hipcc 7.2's packed fp32 code
(-DPRV2_TAPS_PK, scratch .so, child process = this one) on TWO streams, an MFMA-heavy conv + the next frame's table kernels on a THIRD, nothing else of
the frame.  Every gather result is compared with the one computed alone on an idle GPU.   python tools/probes/taps_pk_standalone.py [iters=300] [pk|shipped]"""
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
which = sys.argv[2] if len(sys.argv) > 2 else "pk"
if which == "pk":
    cs, tmp = os.path.join(ROOT, "patchrefinerv2_amd", "csrc"), tempfile.mkdtemp(prefix="prv2_pk_")
    obj, so = os.path.join(tmp, "coarse_taps_pk.o"), os.path.join(tmp, "libprv2_hip_pk.so")
    fl = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-gpu-rdc"]
    subprocess.run(["/opt/rocm/bin/hipcc", *fl, "-DPRV2_TAPS_PK", "-c", os.path.join(cs, "coarse_taps.hip"), "-o", obj], check=True, capture_output=True)
    others = [o for o in sorted(glob.glob(os.path.join(cs, "*.o"))) if os.path.basename(o) != "coarse_taps.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *others, obj], check=True, capture_output=True)
    os.environ["PRV2_HIP_LIB"] = so
os.environ["PRV2_DISPATCH"] = "ctypes"
import torch  # noqa: E402
from patchrefinerv2_amd import lib as L, ops as P  # noqa: E402

torch.set_grad_enabled(False)
dev, g = "cuda", torch.Generator().manual_seed(2)
H, W, C, K = (int(v) for v in os.environ.get("SHAPE", "192,256,256,41").split(","))   # default: the headline's level 1 (41 tiles of 192 x 256, 256 channels)
# as the frame has it: ONE table buffer per level, its consumers channel slices of it (ld = 9 * (C + 64): the level's other consumer)
Gbuf = P.Feat((torch.randn(1, H, W, 9 * (C + 64), generator=g) * 0.1).to(dev))
G = Gbuf.slice(9 * 64, 9 * C)
taps = P.CoarseTaps(G, C, (0.25, 0.25))
G2 = P.Feat((torch.randn(1, H, W, 9 * C, generator=g) * 0.1).to(dev))
org = torch.rand(2 * K, 2, generator=g) * torch.tensor([W * 0.75, H * 0.75])
org[0], org[1] = torch.tensor([0.0, 0.0]), torch.tensor([W * 0.75, H * 0.75])   # tiles on the frame border: all five border-pixel classes
boxes = torch.cat([org, org + torch.tensor([W * 0.25, H * 0.25])], 1).float().to(dev)
bA, bB = boxes[:K].contiguous(), boxes[K:].contiguous()
x = P.Feat(torch.randn(K, 96, 128, 256, generator=g).to(dev))
cw = P.pack_conv((torch.randn(256, 256, 3, 3, generator=g) / 48).to(dev), None, pad=1, prec=L.PREC_NAMES["bf16x3"])
refA, refB = taps.gather(bA, 1.0, H, W).buf.clone(), taps.gather(bB, 1.0, H, W).buf.clone()
torch.cuda.synchronize()
sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
for it in range(iters):
    with torch.cuda.stream(sC):                                      # the aggressors: an MFMA-bound conv, then the next frame's knot table
        P.conv2d(x, cw)
        P.CoarseTaps(G2, C, (0.25, 0.25))
    with torch.cuda.stream(sA):
        oA = taps.gather(bA, 1.0, H, W)
    with torch.cuda.stream(sB):
        oB = taps.gather(bB, 1.0, H, W)
    torch.cuda.synchronize()
    for o, r, n in ((oA, refA, "A"), (oB, refB, "B")):
        if not torch.equal(o.buf, r):
            bad += 1
            d = (o.buf != r)
            print(f"iteration {it} stream {n}: {int(d.sum())} values differ, max |d| {float((o.buf - r).abs().max()):.3e}", flush=True)
print(f"RESULT standalone library={which} iterations={iters} mismatching gathers={bad} of {2 * iters}  ({L.LIB_PATH})")
