"""Inference driver + general image dataset (the callers on either side of the hot path).

Tester.run          estimator/tester/tester.py:52-127 (frame loop, model call contract, uint16 PNG x256)
ImageDataset        estimator/datasets/general_dataset.py:161-234 (folder of images -> image_hr / image_lr)
read_image          estimator/datasets/general_dataset.py:22-62 (RGB/255 -> bicubic, align_corners=True)
With ``--save``: <name>.png (colour map, tester.py:72-87), <name>_uint16.png (depth x 256, :89-91), <name>_coarse.png
(coarse prediction resized to the raw shape, :93-96) -- colour maps and metrics in metrics.py, PNGs through a
dependency-free encoder.  Not built: <name>_edge.png (cv2.Canny + kornia blur, both un-vendored, :98-106) and the
dataset-specific ground-truth decoders (general_dataset.py:74-150); ground truth is accepted as metric depth .npy files.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np
import torch
import torch.nn.functional as F

from .registry import DATASETS


def write_png16(path: str, arr_u16: np.ndarray):
    """Minimal PNG encoder: 16-bit grayscale (== PIL's Image.fromarray(uint16).save)."""
    assert arr_u16.dtype == np.uint16 and arr_u16.ndim == 2
    h, w = arr_u16.shape
    raw = np.zeros((h, 1 + 2 * w), dtype=np.uint8)  # filter byte 0 per scanline
    raw[:, 1:] = arr_u16.astype(">u2").view(np.uint8).reshape(h, 2 * w)

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 0, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def write_png8(path: str, arr_u8: np.ndarray):
    """8-bit RGB / RGBA / gray PNG (== cv2.imwrite of the BGR-swapped array the reference builds)."""
    assert arr_u8.dtype == np.uint8 and arr_u8.ndim in (2, 3)
    h, w = arr_u8.shape[:2]
    ch = 1 if arr_u8.ndim == 2 else arr_u8.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    raw = np.zeros((h, 1 + w * ch), dtype=np.uint8)
    raw[:, 1:] = arr_u8.reshape(h, w * ch)

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def read_image_device(path, image_resolution=(2160, 3840), device="cuda") -> torch.Tensor:
    """general_dataset.py:22-62, generic branch, with the resize on the GPU: decode (host), H2D of the SOURCE image (a few MB
    instead of the 99.5 MB 4K frame), RGB/255 + bicubic(align_corners=True) by prv2_bicubic_resize -> [3, H, W] fp32 on device."""
    from . import ops
    if path.endswith(".npy"):
        a = np.load(path)
        img = torch.from_numpy(a.astype(np.float32) / (255.0 if a.max() > 1.5 else 1.0)) if a.dtype != np.uint8 else torch.from_numpy(a)
    else:
        try:
            from PIL import Image
        except ImportError as e:  # pragma: no cover
            raise RuntimeError("PIL is needed to decode image files (or pass .npy arrays)") from e
        img = torch.from_numpy(np.asarray(Image.open(path).convert("RGB")).copy())
    return ops.bicubic_resize(img.to(device), int(image_resolution[0]), int(image_resolution[1]))


def read_image(path, dataset_name="", image_resolution=(2160, 3840)) -> np.ndarray:
    """general_dataset.py:22-62, generic branch on the host (torch CPU): decode, RGB/255, bicubic(align_corners=True)."""
    if path.endswith(".npy"):
        img = np.load(path).astype(np.float32)
        if img.max() > 1.5:
            img = img / 255.0
    else:
        try:
            from PIL import Image
        except ImportError as e:  # pragma: no cover
            raise RuntimeError("PIL is needed to decode image files (or pass .npy arrays)") from e
        img = np.asarray(Image.open(path).convert("RGB")).astype(np.float32) / 255.0
    t = torch.from_numpy(img).unsqueeze(0).permute(0, 3, 1, 2)
    t = F.interpolate(t, tuple(image_resolution), mode="bicubic", align_corners=True)
    return t.squeeze(0).permute(1, 2, 0).numpy()


@DATASETS.register_module()
class ImageDataset:
    def __init__(self, rgb_image_dir, mode="", min_depth=1e-3, max_depth=80, gt_dir=None, image_resolution=(2160, 3840),
                 dataset_name="", network_process_size=(384, 512), resize_mode="zoe"):
        self.rgb_image_dir = rgb_image_dir
        self.files = sorted(os.listdir(rgb_image_dir))
        # ground truth: metric depth as <gt_dir>/<basename>.npy (the reference's per-dataset decoders -- u4k disparity +
        # factor files, gta exr, middlebury pfm ... general_dataset.py:74-150 -- are not built)
        self.gt_dir = gt_dir
        self.min_depth, self.max_depth = min_depth, max_depth
        self.dataset_name = dataset_name
        self.image_resolution = tuple(image_resolution)
        self.network_process_size = tuple(network_process_size)
        self.resize_mode = resize_mode

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        name = self.files[i]
        # image_hr is resized on the device (prv2_bicubic_resize); image_lr is produced there too by model.resizer
        hr = read_image_device(os.path.join(self.rgb_image_dir, name), self.image_resolution)
        item = dict(image_hr=hr, img_file_basename=os.path.splitext(name)[0])
        if self.gt_dir is not None:
            from .metrics import get_boundaries
            gt = np.load(os.path.join(self.gt_dir, item["img_file_basename"] + ".npy")).astype(np.float32)
            item["depth_gt"] = torch.from_numpy(gt)[None, None]
            item["boundary"] = torch.from_numpy(get_boundaries(gt, th=1, dilation=0))
        return item

    def get_metrics(self, depth_gt, result, disp_gt_edges=None, **kw):
        """general_dataset.py:236-245 (a GPU ``result`` is scored where it is: metrics.compute_metrics_device)"""
        from .metrics import compute_metrics, compute_metrics_device
        if isinstance(result, torch.Tensor) and result.is_cuda:
            return compute_metrics_device(depth_gt, result, disp_gt_edges=disp_gt_edges, min_depth_eval=self.min_depth,
                                          max_depth_eval=self.max_depth, garg_crop=False, eigen_crop=False, dataset=self.dataset_name)
        return compute_metrics(depth_gt, result, disp_gt_edges=disp_gt_edges, min_depth_eval=self.min_depth,
                               max_depth_eval=self.max_depth, garg_crop=False, eigen_crop=False, dataset=self.dataset_name)


class RunnerInfo:
    def __init__(self, **kw):
        self.rank, self.save, self.gray_scale, self.work_dir = 0, False, False, "."
        self.__dict__.update(kw)


def collect_results(results, size):
    """mmengine.dist.collect_results_gpu as Tester.run uses it (estimator/tester/tester.py:124-127): the per-rank result lists
    of a frame-sharded run (frame f on rank f mod world) all-gathered as pickled objects and interleaved back into dataset
    order on rank 0 (the other ranks get None); a single process returns its own list."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return results
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, results)
    if dist.get_rank() != 0:
        return None
    ordered = []
    for i in range(max(len(p) for p in parts)):
        ordered.extend(p[i] for p in parts if i < len(p))
    return ordered[:size]


class Tester:
    """``Tester(config, runner_info, dataloader, model).run(cai_mode, process_num, image_raw_shape, patch_split_num)``"""

    def __init__(self, config, runner_info, dataloader, model):
        self.config, self.runner_info, self.dataloader, self.model = config, runner_info, dataloader, model

    @torch.no_grad()
    def run(self, cai_mode="m1", process_num=4, image_raw_shape=(2160, 3840), patch_split_num=(4, 4), seed=None, shard="frames"):
        """``shard='frames'`` (the reference's data parallelism, tester.py:58: frame f on rank f mod world) or ``'patches'`` (every
        rank works on EVERY frame: its tiles are sharded over the ranks and gathered to rank 0, which blends, saves and scores --
        models._PatchModel.forward(shard=...)).  The loop knows its next frame: its low-resolution image is announced to the
        model, which runs that coarse forward beside the current frame's tiles (``next_image_lr``)."""
        import random
        results = []
        rank, world = self.runner_info.rank, getattr(self.runner_info, "world_size", 1)
        patches = shard == "patches" and world > 1
        prefetch = bool(getattr(self.model, "needs_coarse", False))
        todo = list(range(len(self.dataloader))) if patches else list(range(rank, len(self.dataloader), world))

        def load(idx):
            item = self.dataloader[idx]
            hr = item["image_hr"].unsqueeze(0).cuda()
            return item, hr, self.model.resizer(hr)

        nxt = load(todo[0]) if todo else None
        for n, idx in enumerate(todo):
            item, hr, lr = nxt
            nxt = load(todo[n + 1]) if n + 1 < len(todo) else None
            if seed is not None:
                random.seed(seed)
            tile_cfg = dict(image_raw_shape=list(image_raw_shape), patch_split_num=list(patch_split_num))
            # with ground truth the frame is scored on the device (metrics.compute_metrics_device): ask for the device map
            kw = dict(return_device=True) if item.get("depth_gt") is not None and getattr(self.model, "supports_return_device", False) else {}
            if patches:
                kw.update(shard=(rank, world), gather_dst=0)
            if prefetch and nxt is not None:
                kw["next_image_lr"] = nxt[2]
            result, log = self.model(mode="infer", cai_mode=cai_mode, process_num=process_num, tile_cfg=tile_cfg,
                                     image_lr=lr, image_hr=hr, **kw)
            if result is None:  # patch-sharded: only rank 0 holds the map
                continue
            result_dev = result if result.is_cuda else None
            result = result.cpu()  # BaselinePretrain(target='coarse') hands back the device tensor (baseline_pretrain.py:464)
            if self.runner_info.save:
                os.makedirs(self.runner_info.work_dir, exist_ok=True)
                base = os.path.join(self.runner_info.work_dir, item["img_file_basename"])
                # raw depth as 16-bit PNG, multiplier 256 (tester.py:89-91)
                write_png16(base + "_uint16.png", (result.squeeze().numpy() * 256).astype("uint16"))
                from .metrics import colorize
                if getattr(self.runner_info, "gray_scale", False):
                    color = colorize(result, cmap="gray_r")
                else:  # every dataset branch of tester.py:76-84 but cityscapes maps to Spectral, 0..100 percentiles
                    cmap = "magma_r" if getattr(self.dataloader, "dataset_name", "") == "cityscapes" else "Spectral"
                    color = colorize(result, cmap=cmap, vminp=0, vmaxp=100)
                write_png8(base + ".png", np.ascontiguousarray(color[:, :, :3]))
                from .metrics import depth_edges
                write_png8(base + "_edge.png", depth_edges(result).astype(np.uint8) * 255)  # tester.py:99-106
                if log.get("coarse_prediction") is not None:  # absent for BaselinePretrain
                    coarse = F.interpolate(log["coarse_prediction"].cpu(), tuple(image_raw_shape), mode="bilinear")
                    write_png8(base + "_coarse.png", np.ascontiguousarray(colorize(coarse, cmap="Spectral", vminp=0, vmaxp=100)[:, :, :3]))
            entry = dict(name=item["img_file_basename"], shape=tuple(result.shape), mean=float(result.mean()))
            if item.get("depth_gt") is not None:
                entry["metrics"] = self.dataloader.get_metrics(item["depth_gt"], result if result_dev is None else result_dev,
                                                               disp_gt_edges=item.get("boundary"))
            results.append(entry)
        if not patches:
            # collect results from all ranks (tester.py:124-127: collect_results_gpu); rank 0 evaluates the whole dataset
            allr = collect_results(results, len(self.dataloader))
            results = allr if allr is not None else results
        if results and "metrics" in results[0] and rank == 0:
            from .metrics import evaluate
            self.last_eval = evaluate([r["metrics"] for r in results])
        return results

    @torch.no_grad()
    def run_consistency(self, image_raw_shape=(2160, 3840), patch_split_num=(4, 4), overlap=270):
        """Seam-consistency protocol of the reference (estimator/tester/tester.py:211-321 with the U4K / ETH3D consistency
        crops, eth_dataset.py:86-93): the frame's sh x sw crops of patch_raw_shape are shifted towards the centre so that
        neighbours overlap by ``overlap`` pixels, each crop is predicted on its own (the reference drives ``mode='train'`` once
        per crop: coarse forward + that crop's ROI + refiner), resized bilinear(align_corners) to the crop's raw size, and the
        error is the mean |difference| over the strips two adjacent crops share (left and up neighbours, tester.py:250-293).
        Returns one dict per frame with ``consistency_error``; ``self.last_eval`` holds the dataset mean."""
        from . import ops
        sh, sw = patch_split_num
        H, W = image_raw_shape
        rh, rw = H // sh, W // sw
        half = overlap // 2

        def starts(n, size):  # eth_dataset.py:92-93 generalised: shift crop i by ((n - 1) - 2 i) * overlap / 2 towards the centre
            return [int(i * size + ((n - 1) - 2 * i) * overlap / 2) for i in range(n)]

        hs, ws = starts(sh, rh), starts(sw, rw)
        tiles = [(h, w) for h in hs for w in ws]
        tile_cfg = dict(image_raw_shape=list(image_raw_shape), patch_split_num=list(patch_split_num))
        results = []
        rank, world = self.runner_info.rank, getattr(self.runner_info, "world_size", 1)
        for idx in range(rank, len(self.dataloader), world):
            item = self.dataloader[idx]
            hr = item["image_hr"].unsqueeze(0).cuda()
            # the consistency dataset hands the model pre-normalised bboxs (pre_norm_bbox=True in every shipped config)
            pre = bool(getattr(getattr(self.model, "config", None), "get", lambda k, d: d)("pre_norm_bbox", True))
            preds = self.model.predict_tiles(self.model.resizer(hr), hr, tiles, tile_cfg, prenorm_bbox=pre)
            up = torch.empty((len(tiles), rh, rw, 1), device=preds.device)  # dense 1-channel NHWC == [K, rh, rw]
            ops.upsample_bilinear(ops.Feat(preds.view(len(tiles), preds.shape[-2], preds.shape[-1], 1)), rh, rw, out=ops.Feat(up))
            up = up.view(len(tiles), rh, rw)
            errs = []
            for ii in range(sh):
                for jj in range(sw):
                    cur = up[ii * sw + jj]
                    if jj > 0:  # left neighbour: its last ``overlap`` columns == my first ones
                        errs.append((up[ii * sw + jj - 1][:, -overlap:] - cur[:, :overlap]).abs().flatten())
                    if ii > 0:  # upper neighbour
                        errs.append((up[(ii - 1) * sw + jj][-overlap:, :] - cur[:overlap, :]).abs().flatten())
            ce = float(torch.cat(errs).mean()) if errs else 0.0
            entry = dict(name=item["img_file_basename"], consistency_error=ce)
            self.last_crops = up  # [sh * sw, rh, rw] on the device (tests compare them with the reference's)
            if self.runner_info.save:  # the stitched centres (tester.py:243-247) as a colour map
                os.makedirs(self.runner_info.work_dir, exist_ok=True)
                full = torch.zeros((H, W))
                for k, (h, w) in enumerate(tiles):
                    full[h + half:h + rh - half, w + half:w + rw - half] = up[k][half:rh - half, half:rw - half].cpu()
                from .metrics import colorize
                write_png8(os.path.join(self.runner_info.work_dir, entry["name"] + ".png"), np.ascontiguousarray(colorize(full[None, None])[:, :, :3]))
            results.append(entry)
        self.last_eval = dict(consistency_error=float(np.mean([r["consistency_error"] for r in results]))) if results else {}
        return results

    @torch.no_grad()
    def benchmark(self, cai_mode="m1", process_num=4, image_raw_shape=(2160, 3840), patch_split_num=(4, 4), repeat_times=10,
                  log_interval=10, num_warmup=20, total_iters=50, seed=None):
        """The reference's own throughput protocol (estimator/tester/tester.py:325-406): ``repeat_times`` passes over the
        dataloader, each timing frames ``num_warmup`` .. ``total_iters`` one by one (synchronize, perf_counter, model(...),
        synchronize), fps = frames / summed time; reports the mean and variance over the passes plus the model's FLOPs and
        parameter count (mmengine's get_model_complexity_info there; here the per-launch algorithmic 2*MAC accounting of
        ops.PROFILER and the state-dict spec) and writes ``<work_dir>/benchmark.txt``.  The dataset is cycled when it holds
        fewer than ``total_iters`` frames (the reference simply stops early and divides by zero)."""
        import random
        import time
        from . import ops
        n = len(self.dataloader)
        if n == 0:
            raise ValueError("benchmark: empty dataset")
        tile_cfg = dict(image_raw_shape=list(image_raw_shape), patch_split_num=list(patch_split_num))
        items = {}

        def frame(i):
            if i % n not in items:  # decode + device resize once per file: the reference times model(...) only
                hr = self.dataloader[i % n]["image_hr"].unsqueeze(0).cuda()
                items[i % n] = (hr, self.model.resizer(hr))
            return items[i % n]

        bench = dict(unit="img / s")
        fps_list = []
        for rep in range(repeat_times):
            pure = 0.0
            for i in range(total_iters):
                hr, lr = frame(i)
                if seed is not None:
                    random.seed(seed)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                self.model(mode="infer", cai_mode=cai_mode, process_num=process_num, tile_cfg=tile_cfg, image_lr=lr, image_hr=hr)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if i >= num_warmup:
                    pure += dt
                    if (i + 1) % log_interval == 0 and self.runner_info.rank == 0:
                        print(f"Done image [{i + 1:<3}/ {total_iters}], fps: {(i + 1 - num_warmup) / pure:.3f} img / s")
            fps = (total_iters - num_warmup) / pure
            bench[f"overall_fps_{rep + 1}"] = round(fps, 2)
            fps_list.append(fps)
        bench["average_fps"] = round(float(np.mean(fps_list)), 2)
        bench["fps_variance"] = round(float(np.var(fps_list)), 4)
        hr, lr = frame(0)
        ops.PROFILER.start()
        self.model(mode="infer", cai_mode=cai_mode, process_num=process_num, tile_cfg=tile_cfg, image_lr=lr, image_hr=hr)
        torch.cuda.synchronize()
        ops.PROFILER.stop()
        summ = ops.PROFILER.summary()
        bench["flops"] = float(sum(d["flops"] for d in summ.values()))
        bench["params"] = int(sum(int(np.prod(shp)) for shp in self.model.spec().values()))
        if self.runner_info.rank == 0:
            os.makedirs(self.runner_info.work_dir, exist_ok=True)
            with open(os.path.join(self.runner_info.work_dir, "benchmark.txt"), "w") as f:
                f.write("kernel, launches, GFLOP (2*MAC) per frame\n")
                for tag, d in sorted(summ.items(), key=lambda kv: -kv[1]["flops"]):
                    f.write(f"{tag}, {d['launches']}, {d['flops'] / 1e9:.1f}\n")
                f.write(f"\nModel Flops: {bench['flops'] / 1e12:.3f} T per frame\nModel Parameters: {bench['params'] / 1e6:.1f} M\n")
                f.write(f"\n\n Average fps of {repeat_times} evaluations: {bench['average_fps']}")
                f.write(f"\n\n The variance of {repeat_times} evaluations: {bench['fps_variance']}\n")
        return bench
