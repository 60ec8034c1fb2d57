"""Negative control of the multi-stream determinism detector (VERDICT r04 #6a): builds the library with coarse_taps.hip's packed fp32 math ON
(-DPRV2_TAPS_PK: the hipcc 7.2 code that gave intermittently wrong border pixels, csrc/coarse_taps.hip BUILD NOTE) into a scratch .so, loads
THAT in this (child) process through the ctypes route, and runs the bench's frame (41 tiles per batch, 3 streams, next-frame prefetch) N times:
prints how many frames differ from the first.      python tools/probes/taps_pk_negative_control.py [frames] [pk|shipped]"""
import glob
import os
import random
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "pk"
if which == "pk":
    cs = os.path.join(ROOT, "patchrefinerv2_amd", "csrc")
    tmp = tempfile.mkdtemp(prefix="prv2_pk_")
    obj, so = os.path.join(tmp, "coarse_taps_pk.o"), os.path.join(tmp, "libprv2_hip_pk.so")
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-gpu-rdc"]
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-DPRV2_TAPS_PK", "-c", os.path.join(cs, "coarse_taps.hip"), "-o", obj], check=True, capture_output=True)
    others = [o for o in sorted(glob.glob(os.path.join(cs, "*.o"))) if os.path.basename(o) != "coarse_taps.o"]
    assert len(others) >= 10, "the in-tree objects of the other sources are needed (make -C patchrefinerv2_amd/csrc)"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *others, obj], check=True, capture_output=True)
    os.environ["PRV2_HIP_LIB"] = so
os.environ["PRV2_DISPATCH"] = "ctypes"

import torch  # noqa: E402

torch.set_grad_enabled(False)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_headline_parity import _pair  # noqa: E402
from patchrefinerv2_amd import lib as L  # noqa: E402

assert (which == "pk") == ("prv2_pk_" in L.LIB_PATH), L.LIB_PATH
model, _, w = _pair("v2_zoe_4k_r32", max_batch=41, n_streams=3)
hr = [torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(s)).to("cuda") for s in (3, 4)]
lr = [model.resizer(h) for h in hr]
tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])


def run(i, nxt=None):
    random.seed(621)
    return model(mode="infer", cai_mode="r32", process_num=4, tile_cfg=tc, image_lr=lr[i], image_hr=hr[i], next_image_lr=nxt)[0]


ref = [run(0), run(1)]
bad = 0
for f in range(frames):
    i = f & 1
    out = run(i, nxt=lr[1 - i])
    if not torch.equal(out, ref[i]):
        bad += 1
        d = out != ref[i]
        print(f"frame {f}: {int(d.sum())} pixels differ, max |d| {float((out - ref[i]).abs().max()):.3e}", flush=True)
print(f"RESULT library={which} frames={frames} mismatching={bad}")
