#!/usr/bin/env python
"""User inference CLI with the reference's flags (README.md:51-76, docs/user_infer.md:113-130):

  python tools/test.py CONFIG --ckp-path CKP --cai-mode r32 --cfg-option general_dataloader.dataset.rgb_image_dir=DIR \\
      [--save] --work-dir OUT --test-type general [--gray-scale] --image-raw-shape H W --patch-split-num h w

CONFIG is an MMEngine-style python config (``model=dict(type=..., config=dict(...))``, ``_base_`` supported).
Extras: ``--synthetic-weights`` (the reference has not released checkpoints), ``--prec``, ``--process-num``, ``--max-batch``, ``--streams``.
Multi-GPU: ``sh tools/dist_test.sh CONFIG GPUS [arguments]`` (docs/user_infer.md:113-130): one process per GPU over RCCL;
``--shard frames`` (default, the reference's data parallelism) or ``--shard patches`` (tiles of every frame over the ranks).
"""
import argparse
import ast
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from patchrefinerv2_amd import models, weights as W  # noqa: E402,F401
from patchrefinerv2_amd.registry import DATASETS, Config, build_model  # noqa: E402
from patchrefinerv2_amd.tester import RunnerInfo, Tester  # noqa: E402,F401


def parse_opts(opts):
    out = {}
    for o in opts or []:
        k, v = o.split("=", 1)
        try:
            v = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            pass
        out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--ckp-path", default=None)
    ap.add_argument("--cai-mode", default="m1")
    ap.add_argument("--cfg-option", nargs="+", default=None)
    ap.add_argument("--save", action="store_true")
    ap.add_argument("--work-dir", default="./work_dir/predictions")
    ap.add_argument("--test-type", default="general")
    ap.add_argument("--gray-scale", action="store_true")
    ap.add_argument("--image-raw-shape", nargs=2, type=int, default=[2160, 3840])
    ap.add_argument("--patch-split-num", nargs=2, type=int, default=[4, 4])
    ap.add_argument("--process-num", type=int, default=4)
    ap.add_argument("--prec", default="bf16x3", choices=["f32", "bf16x3", "f16f6"], help="f16f6: bf16x3 with the 256-channel GatedConvUnit convs of the V2 fusion model in fp16 + block-scaled fp6 (what bench.py runs)")
    ap.add_argument("--max-batch", type=int, default=41, help="tiles per launch batch (the result does not depend on it; config key max_batch wins)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the tile batches are spread over (config key n_streams wins)")
    ap.add_argument("--synthetic-weights", action="store_true")
    ap.add_argument("--seed", type=int, default=621)
    ap.add_argument("--launcher", default="none", choices=["none", "pytorch"],
                    help="pytorch: started by torch.distributed.run (tools/dist_test.sh): one process per GPU, RCCL process group")
    ap.add_argument("--shard", default="frames", choices=["frames", "patches"],
                    help="multi-GPU: frames = frame f on rank f mod N (the reference's dist_test.sh), patches = the tiles of every "
                         "frame sharded over the ranks, gathered to rank 0 over RCCL")
    ap.add_argument("--consistency", type=int, default=0, metavar="OVERLAP",
                    help="Tester.run_consistency (estimator/tester/tester.py:211): seam error over crops overlapping by OVERLAP pixels (reference: 270)")
    ap.add_argument("--benchmark", action="store_true", help="Tester.benchmark (estimator/tester/tester.py:325) instead of run")
    ap.add_argument("--repeat-times", type=int, default=10)
    ap.add_argument("--benchmark-iters", nargs=2, type=int, default=[20, 50], metavar=("WARMUP", "TOTAL"))
    args = ap.parse_args()
    if args.test_type != "general":
        raise SystemExit("only --test-type general (folder of images, optional .npy ground truth via dataset gt_dir) is built")

    cfg = Config.fromfile(args.config)
    cfg.merge_from_dict(parse_opts(args.cfg_option))
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
    if world > 1:  # estimator/utils/dist.py:31-33 (init_dist(launcher, backend='nccl')); nccl == RCCL on ROCm
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))))

    mcfg = cfg.model.to_dict()
    # PatchRefiner / PatchRefinerPlus take one ``config`` dict, BaselinePretrain keyword arguments (baseline_pretrain.py:45)
    # (PatchRefinerSemi: the options belong to the student it delegates to, patchrefiner_semi.py:208-210)
    tgt = mcfg["model_cfg_student"] if "model_cfg_student" in mcfg else mcfg
    mopts = tgt["config"] if "config" in tgt else tgt
    mopts["prec"] = args.prec
    mopts.setdefault("max_batch", args.max_batch)  # (the reference's process_num only groups the random tiles of a plan)
    mopts.setdefault("n_streams", args.streams)
    model = build_model(mcfg)
    if args.ckp_path:
        sd = torch.load(args.ckp_path, map_location="cpu")
        print(model.load_dict(sd.get("model_state_dict", sd)))
    elif args.synthetic_weights:
        model.load_state_dict(W.synth_state_dict(model.spec(), seed=0), strict=True)
    else:
        raise SystemExit("give --ckp-path or --synthetic-weights")

    ds_cfg = cfg.general_dataloader.dataset.to_dict()
    ds_cfg["image_resolution"] = args.image_raw_shape
    dataset = DATASETS.build(ds_cfg)
    runner = RunnerInfo(rank=rank, world_size=world, save=args.save, gray_scale=args.gray_scale, work_dir=args.work_dir)
    tester = Tester(cfg, runner, dataset, model)
    if args.consistency:
        for r in tester.run_consistency(image_raw_shape=args.image_raw_shape, patch_split_num=args.patch_split_num, overlap=args.consistency):
            print(f"[rank {rank}] {r['name']}: consistency_error {r['consistency_error']:.6f}")
        print(f"[rank {rank}] consistency_error {tester.last_eval.get('consistency_error', float('nan')):.6f}")
        return
    if args.benchmark:
        b = tester.benchmark(cai_mode=args.cai_mode, process_num=args.process_num, image_raw_shape=args.image_raw_shape,
                             patch_split_num=args.patch_split_num, repeat_times=args.repeat_times,
                             num_warmup=args.benchmark_iters[0], total_iters=args.benchmark_iters[1], seed=args.seed)
        print(f"Average fps of {args.repeat_times} evaluations: {b['average_fps']}")
        print(f"The variance of {args.repeat_times} evaluations: {b['fps_variance']}")
        print(f"Model Flops: {b['flops'] / 1e12:.3f} T  Model Parameters: {b['params'] / 1e6:.1f} M")
        return
    results = tester.run(cai_mode=args.cai_mode, process_num=args.process_num, image_raw_shape=args.image_raw_shape,
                         patch_split_num=args.patch_split_num, seed=args.seed, shard=args.shard)
    if rank == 0 or world == 1:  # (frame-sharded runs: rank 0 holds every rank's results, collected like collect_results_gpu)
        for r in results:
            print(f"[rank {rank}] {r['name']}: depth {r['shape']} mean {r['mean']:.4f}")
        if getattr(tester, "last_eval", None):  # frames that came with ground truth (dataset gt_dir)
            print(f"[rank {rank}] " + ", ".join(f"{k} {v:.4f}" for k, v in tester.last_eval.items()))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
