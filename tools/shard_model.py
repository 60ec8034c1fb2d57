#!/usr/bin/env python
"""Predicted 1 / 2 / 4 / 8-GPU curve of the patch-sharded mode, from ONE GPU -- a MODEL, not a measurement.

For N in (1, 2, 4, 8) every rank's part of a frame (its tiles of ``shard_layout``; the NEXT frame's coarse forward on the one rank that
owns it -- the owner rotates, modelled as 1 frame in N with rank 0 as owner and N - 1 with another rank) is run on this one GPU and timed; rank 0 is also timed WITH the delivered stacks (receive -> blend -> D2H of the map).  The exchange itself cannot run
here: it is priced from its bytes at a stated per-link xGMI rate plus a per-collective latency.  Frame time(N) = max over ranks
of what the rank does before it is done; rank 0's tail (blend of the last group + D2H) sits behind everybody's last group.

    python tools/shard_model.py [--workload v2_zoe_4k_r32] [--reps 3] [--link-gbs 45] [--coll-us 30] > profiles/rNN_shard_model.json
"""
import argparse
import json
import os
import random
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--link-gbs", type=float, default=45.0, help="assumed achieved GB/s of ONE xGMI link into rank 0 (peak ~153 GB/s per "
                    "direction pair; RCCL gathers of a few MB reach a fraction)")
    ap.add_argument("--coll-us", type=float, default=30.0, help="assumed fixed cost of one RCCL collective")
    ap.add_argument("--no-prefetch", action="store_true")
    ap.add_argument("--no-rotate", action="store_true", help="every rank computes the next frame's coarse pyramid itself (the design until round 5)")
    ap.add_argument("--prec", default="f16f6", choices=["bf16x3", "f16f6"], help="the arithmetic (bench.py's default: f16f6)")
    args = ap.parse_args()
    from patchrefinerv2_amd import models, weights as W  # noqa: F401
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd.workloads import DEFAULT_WORKLOAD, WORKLOADS, model_config, state_spec
    name = args.workload or DEFAULT_WORKLOAD
    w = WORKLOADS[name]
    dev = torch.device("cuda", 0)
    model = build_model(model_config(name, prec=args.prec, max_batch=int(w.get("max_batch", 41)), n_streams=3))
    model.shard_rotate_coarse = not args.no_rotate
    model.load_state_dict(W.synth_state_dict(state_spec(name), seed=0), strict=True)
    frames = []
    for seed in (0, 1):
        hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(seed)).to(dev)
        frames.append((hr, model.resizer(hr)))
    tc = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])

    def run(i, **kw):
        random.seed(621)
        hr, lr = frames[i % 2]
        nxt = None if args.no_prefetch else frames[(i + 1) % 2][1]
        return model(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tc, image_lr=lr, image_hr=hr, next_image_lr=nxt, **kw)[0]

    # the rotating owner's broadcast, emulated: the owner's tensors are kept, a receiving rank copies them (a device copy stands in for the
    # receive; the transfer itself is priced below from its bytes and must fit inside the frame -- it runs on a stream and a communicator of its own)
    store = {}
    cur = dict(rank=0)

    def bcast(tensors, src):
        if cur["rank"] == src:
            store["t"] = list(tensors)
        else:
            for t, o in zip(tensors, store["t"]):
                t.copy_(o)
    model._bcast_hook = bcast

    def timed(fn, reps):
        fn(0)  # warm
        ts = []
        for i in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(i + 1)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        return statistics.median(ts)

    out = dict(kind="MODEL (one-GPU emulation of every rank + priced exchange), not a multi-GPU measurement", workload=name, prec=args.prec,
               assumptions=dict(xgmi_link_gbs=args.link_gbs, collective_us=args.coll_us, prefetch_next_coarse=not args.no_prefetch,
                                rotate_coarse_owner=not args.no_rotate,
                                note="ranks timed one after the other on the same GPU; the exchange is priced, not run; "
                                     "host launch overhead of a rank is inside its time"), curve=[])
    t1 = timed(lambda i: run(i), args.reps)
    out["curve"].append(dict(n_gpus=1, ms_per_frame=t1, maps_per_s=1e3 / t1, speedup=1.0, efficiency=1.0))
    tile_bytes = w["pps"][0] * w["pps"][1] * 4
    for world in (2, 4, 8):
        cases = []
        # the owner of the next frame's coarse pyramid rotates: 1 frame in ``world`` it is the blending rank 0 itself, otherwise another rank
        for own in ((0, 1) if not (args.no_rotate or args.no_prefetch) else (None,)):
            fi = None if own is None else (own - 1) % world
            stacks = {}

            def rec(mine, shard, dst, group=0):
                stacks[(shard[0], group)] = mine.clone()
                return None

            def frame_of(r, i):
                cur["rank"] = r
                return run(i, shard=(r, world), gather_dst=0, frame_index=fi)

            model._exchange = rec
            order = list(range(world)) if own is None else [own] + [r for r in range(world) if r != own]  # (the owner first: its tensors are what the others copy)
            if own is not None:
                frame_of(own, 0); frame_of(own, 1)  # (the first announcement records the recipe, the second rotates)
            t_rank = [0.0] * world
            for r in order:
                t_rank[r] = timed(lambda i, r=r: frame_of(r, i), args.reps)
            groups = model.last_shard_layout
            model._exchange = lambda mine, shard, dst, group=0: torch.cat([stacks[(q, group)] for q in range(world)], dim=0)
            t_rank0_full = timed(lambda i: frame_of(0, i), args.reps)
            model.__dict__.pop("_exchange", None)
            # the exchange of a group: every peer sends per * tile_bytes over its own link to rank 0, in parallel
            t_x = [g["per"] * tile_bytes / (args.link_gbs * 1e9) * 1e3 + args.coll_us * 1e-3 for g in groups]
            plan_bcast = args.coll_us * 1e-3
            tail0 = t_rank0_full - t_rank[0]           # rank 0: what receiving, blending and handing over add to its own tiles
            # rank 0 cannot finish before the slowest rank's last group has arrived; earlier groups' exchanges hide behind compute
            frame = plan_bcast + max(max(t_rank) + t_x[-1] + min(tail0, 4.0), t_rank0_full + t_x[-1])
            bc_bytes = int(getattr(model, "last_coarse_bcast_bytes", 0)) if own is not None else 0
            bc_ms = bc_bytes / (args.link_gbs * 1e9) * 1e3  # owner -> each peer over its own link, in parallel, beside the tiles
            cases.append(dict(coarse_owner=own, ms_per_frame=frame, per_rank_ms=[round(t, 2) for t in t_rank], rank0_with_blend_and_d2h_ms=round(t_rank0_full, 2),
                              tiles_per_rank=[sum(g["share"][r] for g in groups) for r in range(world)], exchange_ms_per_group=[round(t, 3) for t in t_x],
                              coarse_bcast_mb=round(bc_bytes / 1e6, 1), coarse_bcast_ms_at_link_rate=round(bc_ms, 2), bcast_fits_in_frame=bool(bc_ms < frame)))
        if len(cases) == 2:
            frame = (cases[0]["ms_per_frame"] + (world - 1) * cases[1]["ms_per_frame"]) / world
        else:
            frame = cases[0]["ms_per_frame"]
        out["curve"].append(dict(n_gpus=world, ms_per_frame=frame, maps_per_s=1e3 / frame, speedup=t1 / frame, efficiency=t1 / frame / world, cases=cases,
                                 exchange_bytes_into_rank0=sum(g["per"] for g in groups) * tile_bytes * (world - 1)))
    # frame-sharded mode (the reference's data parallelism): independent frames, one RCCL gather of the N maps per step
    map_bytes = w["raw"][0] * w["raw"][1] * 4
    out["frames_mode"] = [dict(n_gpus=n, maps_per_s=n * 1e3 / (t1 + (map_bytes / (args.link_gbs * 1e9) * 1e3 + args.coll_us * 1e-3 if n > 1 else 0.0)))
                          for n in (1, 2, 4, 8)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
