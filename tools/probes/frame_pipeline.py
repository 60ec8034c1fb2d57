"""How much of a frame's wall time is the host-side hand-over (D2H of the 33 MB map + the host waiting for it before it enqueues the next frame)?
Same frames (a) as bench.py runs them -- every forward returns a CPU tensor -- and (b) with the map left on the device and ONE synchronisation at the
end: the upper bound of what a deferred hand-over (frame i's D2H collected after frame i + 1 is enqueued) could gain.
   python tools/probes/frame_pipeline.py [workload] [frames]"""
import random
import sys
import time

import torch

sys.path.insert(0, ".")
from patchrefinerv2_amd import models, ops, weights as W  # noqa: E402,F401
from patchrefinerv2_amd.registry import build_model  # noqa: E402
from patchrefinerv2_amd.workloads import DEFAULT_WORKLOAD, WORKLOADS, model_config, state_spec  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_WORKLOAD
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = WORKLOADS[name]
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.set_grad_enabled(False)
mc = model_config(name, prec="bf16x3", max_batch=int(w.get("max_batch", 41)), n_streams=3)
mc["config"]["device"] = str(dev)
model = build_model(mc)
model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
tile_cfg = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
frames = []
for i in range(2):
    hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(i)).to(dev)
    frames.append((hr, model.resizer(hr)))


def run(k, on_device):
    outs = []
    for i in range(k):
        hr, lr = frames[i % 2]
        nxt = None if i == k - 1 else frames[(i + 1) % 2][1]
        random.seed(621)
        d, _ = model(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tile_cfg, image_lr=lr, image_hr=hr, return_device=on_device, next_image_lr=nxt)
        outs.append(d)
    torch.cuda.synchronize()
    return outs


for on_device in (False, True, False, True):
    run(2, on_device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(n, on_device)
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: {'map stays on the device, one sync at the end' if on_device else 'every frame returns a CPU tensor (bench.py)'}: {dt * 1e3:.2f} ms per frame", flush=True)
