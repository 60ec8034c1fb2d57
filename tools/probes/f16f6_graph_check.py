"""hip_graph=True with the f16f6 arithmetic at full size: the captured frame (with the fp16 + fp6 kernels and their range words inside) replays bit-identically
to the eager frame, and the range guard still reads the table between replays.   python tools/probes/f16f6_graph_check.py"""
import os, random, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from patchrefinerv2_amd import models, ops, weights as W  # noqa: F401
from patchrefinerv2_amd.registry import build_model
from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec

name = "v2_zoe_4k_r32"
w = WORKLOADS[name]
sd = W.synth_state_dict(state_spec(name), seed=0)
tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(3)).cuda()
outs = {}
for graph in (False, True):
    mc = model_config(name, prec="f16f6", max_batch=41, n_streams=3)
    mc["config"]["hip_graph"] = graph
    m = build_model(mc)
    m.load_state_dict(sd, strict=True)
    lr = m.resizer(hr)
    res = []
    for i in range(4):
        random.seed(621)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        d, _ = m(mode="infer", cai_mode="r32", process_num=4, tile_cfg=tc, image_lr=lr, image_hr=hr)
        res.append((d.clone(), time.perf_counter() - t0))
    outs[graph] = res
    print("hip_graph", graph, "ms per frame:", [round(1e3 * t, 1) for _, t in res], "recalibrations", getattr(m, "f6_recalibrations", 0), flush=True)
    del m
    torch.cuda.empty_cache()
ref = outs[False][0][0]
assert all(torch.equal(ref, d) for d, _ in outs[False]) and all(torch.equal(ref, d) for d, _ in outs[True]), "graph replay differs from the eager frame"
print("f16f6: eager frames and graph replays bit-identical; depth range", float(ref.min()), float(ref.max()))
