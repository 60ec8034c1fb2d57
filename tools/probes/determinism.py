"""Run the default 4K workload several times with identical inputs and report bit differences between runs."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchrefinerv2_amd import models, weights as W  # noqa: F401
from patchrefinerv2_amd.registry import build_model
from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
name = sys.argv[1] if len(sys.argv) > 1 else "v2_zoeda_4k_r32"
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mode = sys.argv[3] if len(sys.argv) > 3 else None
w = WORKLOADS[name]
model = build_model(model_config(name, prec="bf16x3", max_batch=14, n_streams=streams))
model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(3)).cuda()
lr = model.resizer(hr)
tile_cfg = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
outs = []
for i in range(4):
    random.seed(621)
    d, _ = model(mode="infer", cai_mode=mode or w["mode"], process_num=4, tile_cfg=tile_cfg, image_lr=lr, image_hr=hr)
    outs.append(d)
for i in range(1, 4):
    diff = (outs[i] - outs[0]).abs()
    nz = (diff > 0)
    print(f"run {i} vs 0: equal={torch.equal(outs[i], outs[0])} ndiff={int(nz.sum())} max={float(diff.max()):.3e}",
          "rows", nz.any(-1).nonzero()[:, -1].unique()[:6].tolist() if nz.any() else "")
