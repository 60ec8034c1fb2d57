"""Same frame with / without the next frame's coarse forward riding beside the tiles, several times: where do bits differ?
   python tools/probes/prefetch_determinism.py [workload] [max_batch] [streams]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from patchrefinerv2_amd import models, weights as W  # noqa: F401
from patchrefinerv2_amd.registry import build_model
from patchrefinerv2_amd.workloads import WORKLOADS, model_config, state_spec
name = sys.argv[1] if len(sys.argv) > 1 else "v2_zoe_4k_r32"
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 41
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 3
w = WORKLOADS[name]
model = build_model(model_config(name, prec="bf16x3", max_batch=mb, n_streams=streams))
model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
frames = []
for seed in (3, 4):
    hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(seed)).cuda()
    frames.append((hr, model.resizer(hr)))
tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
if os.environ.get("PROBE_NO_PREFETCH_PREP"):  # the prefetch does the coarse forward only; the tap tables are made at pickup, on the main stream
    _orig = model._prefetch_coarse
    model._prefetch_coarse = lambda nxt, main, tile_cfg=None: _orig(nxt, main, None)
if os.environ.get("PROBE_PREP_ONLY_AGGRESSOR"):  # no coarse forward beside the tiles: only a frame preparation of an already computed pyramid
    _feats = {}

    def _agg(nxt, main, tile_cfg=None):
        if nxt is None:
            return
        if "f" not in _feats:
            _feats["f"] = model.coarse_forward(nxt)[0]
            torch.cuda.synchronize()
        st = model.__dict__.setdefault("_agg_stream", torch.cuda.Stream())
        st.wait_stream(main)
        with torch.cuda.stream(st):
            for f in _feats["f"]:
                f.aux = None
            model._keep = model._prepare_frame(_feats["f"], tile_cfg)
        main.wait_stream(st) if False else None
    model._prefetch_coarse = _agg


def run(i, nxt=None):
    random.seed(621)
    return model(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tc, image_lr=frames[i][1], image_hr=frames[i][0], next_image_lr=nxt)[0]


def cmp(tag, x, y):
    d = (x - y).abs()
    nz = d > 0
    rows = nz.any(-1).nonzero()[:, -1]
    cols = nz.any(-2).nonzero()[:, -1]
    print(f"{tag}: equal={torch.equal(x, y)} ndiff={int(nz.sum())} max={float(d.max()):.3e}",
          (f"rows {int(rows.min())}..{int(rows.max())} cols {int(cols.min())}..{int(cols.max())}" if nz.any() else ""), flush=True)


a = run(0)
cmp("plain vs plain", a, run(0))
b0 = run(0, nxt=frames[1][1])
cmp("plain vs with-prefetch", a, b0)
b1 = run(1)
cmp("prefetched frame1 vs inline frame1", b1, run(1))
cmp("plain vs plain (again)", a, run(0))
b0 = run(0, nxt=frames[1][1])
cmp("plain vs with-prefetch (again)", a, b0)
run(1)
