#!/usr/bin/env python
"""Validates bench.py's ``cpu_baseline`` extrapolation (coarse forward + tiles x a timed 4-tile batch) against a fully timed
oracle frame: the whole m1 frame of the workload (16 tiles, the reference's regular_tile loop in batches of process_num = 4,
paste included) on the same host threads.  CPU only (no GPU needed); ~1 minute on the GPU box's host.

    python tools/cpu_baseline_validate.py [--workload v2_zoe_4k_r32] > profiles/rNN_cpu_baseline_validation.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default=None)
    args = ap.parse_args()
    import bench
    from patchrefinerv2_amd import weights as W
    from patchrefinerv2_amd.workloads import DEFAULT_WORKLOAD, WORKLOADS, state_spec
    name = args.workload or DEFAULT_WORKLOAD
    w = WORKLOADS[name]
    sd = W.synth_state_dict(state_spec(name), seed=0)
    est = bench.cpu_baseline(name, sd, 0)                      # the bench's bounded sample, as it runs inside bench.py
    t_est_frame = 1.0 / est["value"]
    # the same estimate for an m1 frame (16 tiles): coarse + 16 x tile, from the sample's own numbers
    import re
    t_coarse = float(re.search(r"best ([0-9.]+) s", est["sample"]).group(1))
    t_tile = float(re.search(r"median ([0-9.]+) s per tile", est["sample"]).group(1))
    n_m1 = w["split"][0] * w["split"][1]
    est_m1 = t_coarse + n_m1 * t_tile
    # the fully timed frame
    from oracle import dav2 as o_dav2, tiling as o_tiling, zoe as o_zoe
    sd_cpu = {k: v.float() for k, v in sd.items()}
    kw = dict(patch_process_shape=w["pps"], image_raw_shape=w["raw"], patch_split_num=w["split"],
              resizer="zoe" if w.get("zoe_type") == "ZoeDepth" else "da")
    zc = W.zoedepth_cfg(w["zoe"])
    assert w["kind"] == "PatchRefinerPlus" and w.get("zoe"), "validation is wired for the V2 / ZoeDepth workloads"
    m = o_tiling.OraclePatchRefinerPlus(sd_cpu, None, coarse_fn=lambda lr: o_dav2.coarse_features(o_zoe.zoedepth_forward(sd_cpu, "coarse_branch.", lr, zc)), **kw)
    hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(0))
    lr = m.resizer(hr)
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
    with torch.no_grad():
        t0 = time.perf_counter()
        depth, _ = m(mode="infer", cai_mode="m1", process_num=4, tile_cfg=tc, image_lr=lr, image_hr=hr)
        t_m1 = time.perf_counter() - t0
    print(json.dumps(dict(workload=name, cores=est["cores"], bench_estimate_r32_frame_s=round(t_est_frame, 1), sample=est["sample"],
                          m1_frame_tiles=n_m1, m1_frame_estimated_s=round(est_m1, 2), m1_frame_measured_s=round(t_m1, 2),
                          estimate_over_measured=round(est_m1 / t_m1, 3), out_shape=list(depth.shape)), indent=1))


if __name__ == "__main__":
    main()
