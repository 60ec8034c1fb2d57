#!/usr/bin/env python
"""Headline benchmark: 4K depth maps/sec in cai-mode r32 on 1..8 MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one 4K frame through the whole hot path: coarse forward -> 81 tiles (crop+resize,
ROI pyramid gather, refiner encoder, BiDirectionalFusion) -> overlap blend -> depth map returned
exactly as the reference returns it (CPU tensor).  Inputs (image_hr, image_lr) are resident in HBM
when the timed region starts.  Synthetic frames (torch.rand, seeded) and synthetic weights
(numpy PCG64 keyed by parameter name): the reference ships neither checkpoints nor images.

Multi-GPU (one process per GPU): ``--shard frames`` (default) = the reference's own data
parallelism, frame f -> rank f mod N, no data-path collective, weak scaling;
``--shard patches`` = tiles of ONE frame sharded over the ranks + one RCCL gather of the per-tile
predictions to rank 0, which blends (strong scaling; ``--gather all`` = all-gather, every rank blends).
In ``--shard frames`` mode the per-rank depth maps are gathered to rank 0 over xGMI at the end of every step
(SURVEY.md 8e cfg 5; ``--gather all`` keeps them on their ranks like the reference's data parallelism).

``--gpus N`` with N > 1 started WITHOUT a launcher (no WORLD_SIZE in the environment) launches itself: the parent -- before any
GPU call -- starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...`` as a child process, relays rank 0's
JSON line and exits with the child's code (the reference's launcher contract: docs/user_infer.md:124-129, tools/dist_test.sh).

Rank 0 prints ONE JSON line (contract in the task statement) with ``roofline`` (dominant kernel,
HIP-event timed on its launch stream) and ``cpu_baseline`` (the oracle restatement on host cores,
bounded sample).  ``operating_point``: shader clock / board power sampled while the timed frames run (the frame is power-limited:
its throughput depends on how much the operands toggle -- ``--data zeros|rand`` fixes the two ends of that range).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dense MFMA TFLOP/s (MI355X_MICROARCH.md).  f16f6: 2.5 PF x 2/3 -- per 64 channels two fp16 MFMAs and one fp6 K=128 MFMA of the same
# 16 cycles take the place of six bf16 MFMAs; it prices the kernels tagged <.., f16f6> only, every other kernel of that mode runs bf16x3
PEAK = {"f32": 157.3, "bf16x3": 2500.0 / 3.0, "bf16": 2500.0, "f16f6": 2500.0 * 2.0 / 3.0}


def peak_of(kernel_tag, prec):
    """the roofline a kernel is priced against: its own arithmetic's (the tag's), else the mode's"""
    if kernel_tag.endswith(",f16f6>"):
        return PEAK["f16f6"]
    return PEAK["bf16x3" if prec == "f16f6" else prec]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--prec", default="f16f6", choices=["f32", "bf16x3", "bf16", "f16f6"],
                    help="matrix-kernel arithmetic (DESIGN.md 3): bf16x3 = split-bf16, AbsRel ~5e-6; f32 = exact fp32 MFMA")
    ap.add_argument("--shard", default="frames", choices=["frames", "patches"])
    ap.add_argument("--gather", default="rank0", choices=["rank0", "all"],
                    help="--shard patches: gather the tile predictions to rank 0 (it alone blends and returns the map) or "
                         "all-gather (every rank blends); --shard frames: rank0 = final RCCL gather of the per-rank maps to "
                         "rank 0 (SURVEY.md 8e cfg 5), all = every rank keeps its own map (the reference's data parallelism)")
    ap.add_argument("--max-batch", type=int, default=None, help="patches per launch batch (results are batch independent); default: the "
                    "workload's own (41 unless it names one)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the tile batches are spread over")
    ap.add_argument("--no-prefetch-coarse", dest="prefetch_coarse", action="store_false",
                    help="do not enqueue the next frame's coarse forward beside the current frame's tile batches")
    ap.add_argument("--hip-graph", action="store_true", help="capture the device side of a frame into a hipGraph and replay it per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="--prec f16f6: skip the bf16x3 run reported beside it")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--layer-report", default=None, help="write a per-layer-shape timing table (extra instrumented frame)")
    ap.add_argument("--data", default="rand", choices=["rand", "zeros"],
                    help="synthetic frames: uniform random pixels (default) or all-zero frames -- the frame is power-limited, so its "
                         "rate depends on operand toggling; zeros is the upper end of that range (not a headline number)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)  # gloo + --stub-model: the CPU launcher test
    ap.add_argument("--stub-model", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def self_launch(args) -> int:
    """``--gpus N`` (N > 1) without a launcher: start the N ranks as a FRESH child process tree (torch.distributed.run) -- this parent
    has not touched the GPU (no torch.cuda call, package not imported) and never re-execs; relay the child's stdout (rank 0's JSON
    line) and return its exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:  # relayed line by line (a hung or failing child's output is visible while it runs; stderr is inherited)
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child job failed with exit code {rc}\n")
    return rc


class OperatingPoint(threading.Thread):
    """shader clock (MHz) and board power (W) of one GPU sampled from sysfs (hwmon) while the timed frames run: the frame is
    power-limited, so a throughput number is only comparable together with its operating point.  Values are None when the box
    does not expose the files to this user."""

    def __init__(self, index: int, period: float = 0.05):
        super().__init__(daemon=True)
        self.period, self.samples, self._stop_ev = period, [], threading.Event()
        self.power_f = self.clk_f = None
        import glob
        self.where = None
        try:  # the sysfs node of THIS torch device (the box may expose more cards than the process sees): by PCI address
            pr = torch.cuda.get_device_properties(index)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cands = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:  # noqa: BLE001  (older torch: no PCI fields)
            bdf, cands = None, []
        if not cands:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"), key=lambda q: int(q.split("/card")[1].split("/")[0]))
            cards = [c for c in cards if os.path.exists(os.path.join(c, "power1_average")) or os.path.exists(os.path.join(c, "power1_input"))]
            cands = cards[index:index + 1]
        if cands:
            h = cands[0]
            self.where = bdf or h
            self.power_f = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
            self.clk_f = os.path.join(h, "freq1_input") if os.path.exists(os.path.join(h, "freq1_input")) else None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def run(self):
        while not self._stop_ev.is_set():
            self.samples.append((self._read(self.power_f), self._read(self.clk_f)))
            self._stop_ev.wait(self.period)

    def finish(self):
        self._stop_ev.set()
        self.join(timeout=1.0)
        pw = [p / 1e6 for p, _ in self.samples if p is not None]
        ck = [c / 1e9 for _, c in self.samples if c is not None]
        mean = lambda v: (sum(v) / len(v)) if v else None  # noqa: E731
        return dict(clock_ghz=mean(ck), clock_ghz_min=min(ck) if ck else None, power_w=mean(pw), power_w_max=max(pw) if pw else None,
                    samples=len(self.samples), source=f"sysfs hwmon of {self.where} (freq1_input = shader clock, power1_average)" if (pw or ck) else None)


def cpu_baseline(name, sd, frame_seed):
    """Oracle (fp32 PyTorch restatement of the reference graph) on the host cores, bounded sample:
    one coarse forward + ONE tile through the per-patch path, extrapolated by the exact tile count."""
    from oracle import tiling as o_tiling
    from patchrefinerv2_amd import weights as W
    from patchrefinerv2_amd.workloads import WORKLOADS
    w = WORKLOADS[name]
    # torch's CPU conv peaks at 16-32 threads on the 2x64-core EPYC host (256 threads is 10x slower)
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sd_cpu = {k: v.float() for k, v in sd.items()}
    kw = dict(patch_process_shape=w["pps"], image_raw_shape=w["raw"], patch_split_num=w["split"])
    ccfg = W.dav2_cfg({**w["coarse"], "max_depth": 80.0}) if w.get("coarse") else None
    if w.get("zoe"):
        from oracle import dav2 as o_dav2, zoe as o_zoe
        zc = W.zoedepth_cfg(w["zoe"])
        kw["resizer"] = "zoe" if w.get("zoe_type") == "ZoeDepth" else "da"
        if w["kind"] == "PatchRefinerPlus":
            m = o_tiling.OraclePatchRefinerPlus(
                sd_cpu, None, coarse_fn=lambda lr: o_dav2.coarse_features(o_zoe.zoedepth_forward(sd_cpu, "coarse_branch.", lr, zc)), **kw)
        else:
            fz = W.zoedepth_cfg(w["fine_zoe"])
            m = o_tiling.OraclePatchRefiner(sd_cpu, None, None, coarse_fn=lambda lr: o_zoe.zoedepth_forward(sd_cpu, "coarse_branch.", lr, zc),
                                            fine_fn=lambda x: o_zoe.zoedepth_forward(sd_cpu, "refiner_fine_branch.", x, fz), **kw)
    elif w["kind"] == "PatchRefinerPlus":
        m = o_tiling.OraclePatchRefinerPlus(sd_cpu, ccfg, **kw)
    else:
        m = o_tiling.OraclePatchRefiner(sd_cpu, ccfg, W.dav2_cfg({**w["fine"], "max_depth": 80.0}), **kw)
    image_hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(frame_seed))
    image_lr = m.resizer(image_hr)
    with torch.no_grad():
        tc = []
        for _ in range(2):  # the first coarse forward also warms the thread pool
            t0 = time.perf_counter()
            feats, pred = m.coarse_forward(image_lr)
            tc.append(time.perf_counter() - t0)
        t_coarse = min(tc)
        rh, rw = m.tile_cfg["patch_raw_shape"]
        nb = 4  # one process_num batch of tiles, timed three times (median)
        tp = []
        for rep in range(3):
            t0 = time.perf_counter()
            crops, bboxs = m._crops(image_hr[0], [0, rh // 2, rh, rh + rh // 2], [(rep + 1) * rw // 2], rh, rw)
            bf = o_tiling.bboxs_to_feat(bboxs, w["raw"], w["pps"])
            post = o_tiling.coarse_postprocess_test(pred, feats, bf, w["pps"][0])
            m.infer_forward(crops, post)
            tp.append((time.perf_counter() - t0) / nb)
        t_patch = sorted(tp)[1]
    t_frame = t_coarse + w["patches"] * t_patch
    return dict(value=1.0 / t_frame, unit="depth maps/s", cores=cores, kind="port", estimated=True,
                sample=f"coarse forward x2 (best {t_coarse:.1f} s) + a batch of 4 of the {w['patches']} tiles through crop/ROI/encoder/fusion "
                       f"x3 (median {t_patch:.1f} s per tile), ESTIMATED frame = coarse + {w['patches']} x tile = {t_frame:.0f} s; "
                       f"blend excluded (<1%)")


def pmc_traffic(kernel_tag, workload, prec):
    """HBM bytes per launch of ``kernel_tag`` from the committed rocprofv3 PMC passes of this same command
    (profiles/r<NN>_<prec>_pmc_frame_<workload>.json, newest round first: separate --pmc FETCH_SIZE / WRITE_SIZE runs,
    per-launch averages in KB; tools/profile_round.sh).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 1/2
    of the bytes of wide coalesced reads.  Returns (bytes, source file) -- (None, None) when no pass of this workload is
    committed: the counters cannot be collected inside this process."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{prec}_pmc_frame_{workload}.json")), reverse=True)
    if not cands:
        return None, None
    path = cands[0]
    name, _, targs = kernel_tag.partition("<")
    bn, pr = targs.rstrip(">").split(",")
    # the tag covers every instantiation "prv2::<name><bn, prec[, ...]>" (e.g. the im2col-tail variant): launch-weighted mean
    if pr == "f16f6":  # (a non-template kernel: its symbol carries no arguments)
        rows = [d for k, d in json.load(open(path)).items() if k == f"prv2::{name}" or k.startswith((f"prv2::{name}(", f"prv2::{name}<"))]
        n = sum(d["launches"] for d in rows)
        if not n:
            return None, None
        return sum((2.0 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * d["launches"] for d in rows) / n * 1024.0, os.path.relpath(path, ROOT)
    prefix = f"prv2::{name}<{bn}, {dict(f32=0, bf16x3=1, bf16=2)[pr]}"
    rows = [d for k, d in json.load(open(path)).items()
            if k.startswith(prefix + ">") or k.startswith(prefix + ",")]
    n = sum(d["launches"] for d in rows)
    if not n:
        return None, None
    b = sum((2.0 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * d["launches"] for d in rows) / n * 1024.0
    return b, os.path.relpath(path, ROOT)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))  # (nothing below has run: no GPU call was made in this process)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks -- refusing to report a number under the wrong n_gpus")
    if args.stub_model:
        return stub_main(args, rank, world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(args.backend, device_id=dev)

    from patchrefinerv2_amd import ops, weights as W
    from patchrefinerv2_amd.registry import build_model
    from patchrefinerv2_amd import models  # noqa: F401  (registers the model types)
    from patchrefinerv2_amd.workloads import DEFAULT_WORKLOAD, WORKLOADS, model_config, state_spec

    name = args.workload or DEFAULT_WORKLOAD
    w = WORKLOADS[name]
    if args.max_batch is None:
        args.max_batch = int(WORKLOADS[name].get("max_batch", 41))
    mc = model_config(name, prec=args.prec, max_batch=args.max_batch, n_streams=args.streams)
    mc["config"]["device"] = str(dev)
    mc["config"]["hip_graph"] = bool(args.hip_graph)
    model = build_model(mc)
    sd = W.synth_state_dict(state_spec(name), seed=0)
    model.load_state_dict(sd, strict=True)

    tile_cfg = dict(image_raw_shape=w["raw"], patch_split_num=w["split"])
    n_frames = args.steps + args.warmup
    shard = (rank, world) if (world > 1 and args.shard == "patches") else None

    def frame(i):
        seed = i if shard is not None else rank * 100003 + i
        if args.data == "zeros":
            hr = torch.zeros(1, 3, *w["raw"], device=dev)
        else:
            hr = torch.rand(1, 3, *w["raw"], generator=torch.Generator().manual_seed(seed)).to(dev)
        return hr, model.resizer(hr)

    frames = [frame(i) for i in range(min(n_frames, 2))]  # resident inputs (2 alternating frames)

    gather_dst = 0 if (shard is not None and args.gather == "rank0") else None
    frame_gather = world > 1 and shard is None and args.gather == "rank0"
    if frame_gather:
        import torch.distributed as dist
        maps = [torch.empty((1, 1, *w["raw"]), device=dev) for _ in range(world)] if rank == 0 else None
        host_maps = torch.empty((world, 1, 1, *w["raw"]), pin_memory=True) if rank == 0 else None
    coll = dict(bytes=0, ms=0.0, n=0)

    def step(i, solo=False, timed=False, last=False):
        """one frame; ``solo``: this rank alone, no collective (the instrumented frames that only rank 0 runs)"""
        hr, lr = frames[i % len(frames)]
        # a frame loop knows its next frame: its coarse forward is enqueued beside this frame's tile batches (models.forward)
        nxt = None if (solo or last or not args.prefetch_coarse) else frames[(i + 1) % len(frames)][1]
        random.seed(621)
        sh = None if solo else shard
        fg = frame_gather and not solo
        depth, _ = model(mode="infer", cai_mode=w["mode"], process_num=4, tile_cfg=tile_cfg, image_lr=lr, image_hr=hr,
                         shard=sh, gather_dst=gather_dst if sh is not None else None, return_device=fg, next_image_lr=nxt)
        if fg:
            # cfg-5 "full xGMI gather": the N per-rank maps (33 MB each at 4K) to rank 0, which hands them to the host
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.gather(depth, maps, dst=0)
            e1.record()
            if rank == 0:
                for r in range(world):
                    host_maps[r].copy_(maps[r], non_blocking=True)
                torch.cuda.current_stream().synchronize()
                depth = host_maps[0]
                if timed:
                    coll["bytes"] += depth.numel() * 4 * (world - 1)
                    coll["ms"] += e0.elapsed_time(e1)
                    coll["n"] += 1
            else:
                torch.cuda.current_stream().synchronize()
        return depth

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, last=i == args.warmup - 1)  # (no prefetch across t0: the timed region does exactly K coarse forwards)
    barrier()
    op = OperatingPoint(local) if rank == 0 else None
    if op is not None:
        op.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i, timed=True, last=i == args.steps - 1)
    barrier()
    elapsed = time.perf_counter() - t0
    operating_point = op.finish() if op is not None else None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    frames_done = args.steps * (world if shard is None else 1)

    # --prec f16f6 on a workload without such layers (the v1 FusionUnet models) IS bf16x3, and is labelled so
    f6_on = args.prec == "f16f6" and not args.stub_model and ops.F6Range.active(dev)
    result = dict(
        metric="4K depth maps/sec (cai-mode r32)" if w["mode"] == "r32" else f"depth maps/sec (cai-mode {w['mode']})",
        value=frames_done / elapsed, unit="depth maps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
        ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling="weak" if shard is None else "strong",
        vs_baseline=None, dtype=("bf16x3+f16f6" if f6_on else ("bf16x3" if args.prec == "f16f6" else args.prec)), data="synthetic" if args.data == "rand" else "synthetic (all-zero frames: upper end of the power-limited range, not a headline number)",
        config=dict(workload=name, model=w["kind"], image_raw_shape=w["raw"], patch_split_num=w["split"],
                    patch_process_shape=w["pps"], cai_mode=w["mode"], patches_per_frame=w["patches"],
                    coarse_branch=(w["coarse"]["encoder"] if w.get("coarse") else w.get("zoe_type", "DA-ZoeDepth") + "/" + w["zoe"]["midas_model_type"]), shard=args.shard if world > 1 else "none",
                    max_batch=args.max_batch, streams=args.streams, hip_graph=bool(args.hip_graph), prefetch_next_coarse=bool(args.prefetch_coarse), out_shape=list(out.shape) if out is not None else None,
                    frames=("torch.rand, seeded" if args.data == "rand" else "zeros"), weights="synthetic (numpy PCG64 keyed by parameter name)",
                    parity_note=("refiner encoder MobileNetV4-S: parity unpinned (timm absent; two independent transcriptions agree)"
                                 if w["kind"] == "PatchRefinerPlus" and not w.get("refiner_encoder") else None)))
    result["operating_point"] = operating_point
    if f6_on:
        # the mode's fp16 range guard (ops.F6Range, models.forward): frames recomputed in the timed region because a layer's input left fp16's range (0 on this data)
        result["f16f6_guard"] = dict(recomputed_frames=int(getattr(model, "f6_recalibrations", 0)),
                                     layers=sum(1 for _, r in ops.F6Range._tables[str(dev)]["layers"] if r() is not None))
        result["dtype_note"] = ("bf16x3 (fp32 operands split hi + lo bf16, 3 MFMAs) everywhere except the 3x3 convs of the 256-channel GatedConvUnits -- "
                                "GatedConvUnit.conv and the unit's fusion_conv.0 inside the fused tail kernel (its LayerNorm, gate GEMM and final stage stay "
                                "bf16x3 / fp32) -- which run fp16 + two block-scaled fp6 (e2m3) corrections per product (csrc/conv3x3_f6.hip): fp32-grade, rms "
                                "1.2e-5 per dot product; kernels tagged <..,f16f6> are priced against 2.5 PF x 2/3")
        if world == 1 and not args.no_alt:
            # the same timed loop in the default arithmetic, beside it
            f6_model = model
            mc2 = model_config(name, prec="bf16x3", max_batch=args.max_batch, n_streams=args.streams)
            mc2["config"]["device"] = str(dev)
            mc2["config"]["hip_graph"] = bool(args.hip_graph)
            model = build_model(mc2)
            model.load_state_dict(sd, strict=True)
            for i in range(args.warmup):
                step(i, last=i == args.warmup - 1)
            barrier()
            t1 = time.perf_counter()
            for i in range(args.steps):
                step(args.warmup + i, timed=False, last=i == args.steps - 1)
            barrier()
            e2 = time.perf_counter() - t1
            result["alt"] = dict(dtype="bf16x3", value=args.steps / e2, unit="depth maps/s", ms_per_step=1e3 * e2 / args.steps,
                                 note="the default arithmetic, same process, same frames, measured right after the timed region")
            del model
            torch.cuda.empty_cache()
            model = f6_model

    if world > 1:
        nt = w["patches"]
        if shard is not None:
            groups = model.last_shard_layout  # (models._PatchModel.shard_layout: gather groups, tiles per rank)
            result["multi_gpu"] = dict(mode="patches", gather=args.gather,
                                       tiles_per_rank=[sum(g["share"][r] for g in groups) for r in range(world)],
                                       gather_groups=[dict(tiles=g["n"], per_rank=g["share"], padded_to=g["per"]) for g in groups],
                                       collective=("async gather->rank0 per group" if gather_dst is not None else "async all_gather per group"),
                                       plan_sync="int32 tile tensor, RCCL broadcast from rank 0 per frame",
                                       collective_bytes_per_frame=sum(g["per"] for g in groups) * w["pps"][0] * w["pps"][1] * 4 * (world - 1))
        else:
            result["multi_gpu"] = dict(mode="frames", gather=args.gather, tiles_per_rank=[nt] * world,
                                       collective=("gather->rank0 of the per-rank depth maps" if frame_gather else "none"),
                                       collective_bytes_per_step=(coll["bytes"] // max(coll["n"], 1) if frame_gather else 0),
                                       collective_ms_per_step=(coll["ms"] / max(coll["n"], 1) if frame_gather else 0.0))

    if world > 1:
        # the job's collective part is over: every rank leaves the group here, so that nobody sits in a barrier (under RCCL's watchdog)
        # while rank 0 runs its instrumented single-GPU frames below
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        if rank != 0:
            return
    if rank == 0 and not args.no_roofline:
        # extra, instrumented frames: HIP events on the launch stream around every matrix-kernel launch
        # rank 0 alone runs these frames: UNSHARDED (a sharded forward would wait in a collective nobody else joins)
        model.n_streams = 1  # per-launch durations must not include kernels of other streams
        step(0, solo=True)  # settle into the single-stream regime (allocator, clocks) before timing launches
        torch.cuda.synchronize()
        runs = []
        for i in range(3):  # three instrumented frames; per kernel the median total (the chip's clock wanders by 10-20 %)
            ops.PROFILER.start(timed=True)
            step(i, solo=True)
            torch.cuda.synchronize()
            ops.PROFILER.stop()
            runs.append(ops.PROFILER.summary())
        model.n_streams = args.streams
        summ = {k: dict(runs[0][k], ms=sorted(r[k]["ms"] for r in runs)[1]) for k in runs[0]}
        tot_ms = sum(d["ms"] for d in summ.values())
        dom = max(summ, key=lambda k: summ[k]["ms"])
        d = summ[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        peak = peak_of(dom, args.prec)
        ideal_ms = lambda k, key: summ[k][key] / 1e12 / peak_of(k, args.prec) * 1e3  # noqa: E731  (a kernel's FLOPs at ITS roofline)
        traffic, traffic_src = pmc_traffic(dom, name, args.prec)
        result["roofline"] = dict(
            bound="mfma", kernel=dom, achieved=ach, peak=peak, unit="TFLOP/s", frac=ach / peak,
            traffic=traffic, traffic_source=traffic_src,
            launches_per_frame=d["launches"], avg_launch_ms=d["ms"] / d["launches"],
            executed_gflop_per_launch=d["flops"] / d["launches"] / 1e9,
            # FLOP accounting: ``frame_algorithmic_tflop`` = the REFERENCE graph's 2*MAC per frame (what the PyTorch model
            # spends on conv / linear / attention matmuls); ``executed_tflop`` = what the kernels here multiply -- less, because the coarse
            # half of every cat([fine, coarse_roi]) conv is computed once per frame at coarse resolution (csrc/coarse_taps.hip), the
            # 3x3 convs of bilinearly upsampled tensors run as tap GEMMs at the source resolution (csrc/upconv.hip) and the
            # refinenet1 out_conv is folded into output_conv1's weights.  Per-kernel ``achieved`` / ``frac`` are EXECUTED FLOPs / time
            # (a kernel's frac can never exceed 1); ``whole_frame_frac`` is -- as in rounds 1-3 -- the EXECUTED FLOPs / the timed step,
            # ``whole_frame_frac_reference_flops`` the reference graph's FLOPs / the timed step (what the map rate is worth in the reference's
            # arithmetic: it exceeds the MFMA pipe's real utilisation by the work the algebra removed).
            frame_algorithmic_tflop=sum(x["algo"] for x in summ.values()) / 1e12,
            executed_tflop=sum(x["flops"] for x in summ.values()) / 1e12,
            matrix_kernel_ms_per_frame=tot_ms,
            # the whole frame against the same peak: every algorithmic FLOP of a frame / the TIMED step (all kernels, gathers,
            # blend, D2H and host gaps included) -- what the headline value is worth as a fraction of the MFMA roofline
            whole_frame_frac=sum(ideal_ms(k, "flops") for k in summ) * 1e-3 / (elapsed / args.steps),
            whole_frame_frac_reference_flops=sum(ideal_ms(k, "algo") for k in summ) * 1e-3 / (elapsed / args.steps),
            kernels={k: dict(launches=v["launches"], ms=round(v["ms"], 3),
                             tflops=(v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else None),
                             frac=(round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak_of(k, args.prec), 4) if v["ms"] > 0 else None))
                     for k, v in summ.items()})
    if rank == 0 and args.layer_report:
        model.n_streams = 1
        ops.PROFILER.start(timed=True, by_shape=True)
        step(0, solo=True)
        torch.cuda.synchronize()
        ops.PROFILER.stop()
        rows = sorted(ops.PROFILER.summary().items(), key=lambda kv: -kv[1]["ms"])
        tot = sum(v["ms"] for _, v in rows)
        with open(args.layer_report, "w") as f:
            f.write(f"# per-layer kernel time (matrix kernels with TFLOP/s; gathers / LayerNorm with their byte volume in the tag), one frame, {name}, {args.prec}; total {tot:.1f} ms\n")
            f.write("ms,pct,launches,TFLOP/s,kernel shape\n")
            for k, v in rows:
                f.write(f"{v['ms']:.3f},{100 * v['ms'] / tot:.1f},{v['launches']},{v['flops'] / max(v['ms'], 1e-9) / 1e9:.1f},{k}\n")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(name, sd, 0)
    if rank == 0:
        print(json.dumps(result))


def stub_main(args, rank, world):
    """``--stub-model`` (the CPU launcher test, tests/test_distributed.py): the rank / timing / reporting skeleton of ``main`` with a
    model that is a sleep -- no GPU, gloo; proves that ``--gpus N`` without a launcher really runs N ranks and reports n_gpus = N."""
    import torch.distributed as dist
    if os.environ.get("PRV2_BENCH_STUB_FAIL") and rank == world - 1:
        raise SystemExit("stub rank failing on request (launcher test)")
    if world > 1:
        dist.init_process_group(args.backend)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.01)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01)
    barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks = torch.tensor([1.0])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks)
    elapsed = float(t.item())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(dict(metric="stub frames/sec", value=args.steps * world / elapsed, unit="frames/s", n_gpus=world, steps=args.steps,
                              warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
                              dtype="none", data="stub", config=dict(workload="stub", ranks_seen=int(ranks.item())))))


if __name__ == "__main__":
    main()
