"""Frame-level drivers: tiling -> per-patch networks -> overlap blend, all on the device.

Host-side mirrors, registered under the reference ``type`` names:
  BaselinePretrain   estimator/models/baseline_pretrain.py:45-464 (tile planner, regular/random tile)
  PatchRefiner       estimator/models/patchrefiner.py:54-404        (V1: DA2 coarse + DA2 per patch + FusionUnet)
  PatchRefinerPlus   estimator/models/patchrefinerplus.py:60-533    (V2: DA2 coarse + LightWeightRefiner + BiDirectionalFusion)
Call contract (tester.py:69, patchrefinerplus.py:367-380,526-530):
  depth, log = model(mode='infer', cai_mode, process_num, tile_cfg, image_lr, image_hr)
  depth: [1,1,H',W'] fp32 CPU tensor; log['coarse_prediction']: [1,1,ph,pw] device tensor.

What differs from the reference on purpose (results identical):
  * the complete tile list is drawn up-front (consuming Python's ``random`` in the reference's
    order) so patches can be batched / sharded freely -- per-patch results do not depend on batching;
  * the blend state stays on the device (one D2H at the end) instead of a CPU map per tile;
  * no ``feat.repeat(K)``: ROI gathers read the single coarse pyramid;
  * no ``torch.cuda.empty_cache()`` per mini-batch.
"""
from __future__ import annotations

import random
from typing import List

import numpy as np
import torch

from . import ops
from .dav2 import DepthAnythingV2, IMAGENET_MEAN, IMAGENET_STD, StateDictModule
from .fusion import BiDirectionalFusion, BiDirectionalFusionHeavy, FusionUnet
from .ops import Feat
from .refiner import LightWeightRefiner
from .registry import MODELS, ConfigDict, build_model

MODELS.register_module(module=FusionUnet)
MODELS.register_module(module=BiDirectionalFusion)
MODELS.register_module(module=BiDirectionalFusionHeavy)
MODELS.register_module(module=LightWeightRefiner)


@MODELS.register_module()
class SILogLoss:  # training-only; constructed by the reference at model init (patchrefinerplus.py:83)
    def __init__(self, **kw):
        pass


@MODELS.register_module()
class GradMatchLoss:
    def __init__(self, **kw):
        pass


# ------------------------------------------------------------------------------------------------
# host-side pieces of the path: blend mask + resize target (tiny, once per frame / model)
# ------------------------------------------------------------------------------------------------
def _gaussian_kernel1d(ksize: int, sigma: float) -> np.ndarray:
    c = (ksize - 1) * 0.5
    i = np.arange(ksize, dtype=np.float64)
    k = np.exp(-((i - c) ** 2) / (2.0 * float(sigma) ** 2))
    return (k / k.sum()).astype(np.float32)


def generatemask(size, border: float = 0.1) -> np.ndarray:
    """estimator/models/utils.py:51-60 with cv2.GaussianBlur restated in numpy (separable Gaussian,
    BORDER_REFLECT_101).  Host logic exactly as in the reference (it runs cv2 on the CPU, once per
    mode per frame); the zero band of the box is wider than the half kernel on every shape of the
    path, so the border mode never matters."""
    h, w = int(size[0]), int(size[1])
    mask = np.zeros((h, w), dtype=np.float32)
    sigma = int(h / 16)
    k_size = int(2 * np.ceil(2 * int(h / 16)) + 1)
    mask[int(border * h): h - int(border * h), int(border * w): w - int(border * w)] = 1
    k = _gaussian_kernel1d(k_size, sigma)
    r = k_size // 2
    pad = np.pad(mask, ((0, 0), (r, r)), mode="reflect")
    tmp = np.zeros_like(mask)
    for i in range(k_size):
        tmp += k[i] * pad[:, i:i + w]
    pad = np.pad(tmp, ((r, r), (0, 0)), mode="reflect")
    out = np.zeros_like(mask)
    for i in range(k_size):
        out += k[i] * pad[i:i + h, :]
    out = (out - out.min()) / (out.max() - out.min())
    return out.astype(np.float32)


_MASK_CACHE = {}


def blend_mask(size, border, add, device) -> torch.Tensor:
    key = (int(size[0]), int(size[1]), float(border), float(add), str(device))
    if key not in _MASK_CACHE:
        _MASK_CACHE[key] = torch.from_numpy(generatemask(size, border) + np.float32(add)).to(device)
    return _MASK_CACHE[key]


class Resizer:
    """``model.resizer`` of the reference: ResizeDA (external/depth_anything/transform.py:6-129,
    keep_aspect_ratio=False, 'minimal', multiple of 14) or ResizeZoe (midas.py:171-174, fixed 384x512),
    executed by the crop+resize HIP kernel."""

    def __init__(self, width: int, height: int, kind: str = "da"):
        if kind == "da":
            self.out_hw = (int(np.round(height / 14) * 14), int(np.round(width / 14) * 14))
        else:
            self.out_hw = (384, 512)
        self.kind = kind

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("Resizer runs on the GPU (HIP crop+resize kernel); move the image to the device")
        B, _, H, W = x.shape
        oh, ow = self.out_hw
        out = Feat.alloc(B, oh, ow, 3, x.device, pad_to=1)
        zero = torch.zeros((1, 2), dtype=torch.int32, device=x.device)
        for b in range(B):
            ops.crop_resize(x[b].contiguous().float(), zero, H, W, oh, ow, None, None, out.batch(b, b + 1))
        return out.to_nchw()


class DeviceRunningAverageMap:
    """RunningAverageMap (estimator/models/utils.py:22-49) resident in HBM."""

    def __init__(self, h, w, device):
        self.avg = torch.zeros((h, w), device=device)
        self.cnt = torch.zeros((h, w), device=device)

    def paste(self, preds, mask, tiles, th, tw):
        ops.blend_paste(self.avg, self.cnt, preds, mask, tiles, th, tw)

    def update(self, preds, mask, tiles, th, tw):
        ops.blend_update(self.avg, self.cnt, preds, mask, tiles, th, tw)

    def resize(self, resolution):
        self.avg, self.cnt = ops.blend_resize(self.avg, self.cnt, int(resolution[0]), int(resolution[1]))


def _cfg(config):
    if isinstance(config, ConfigDict):
        return config
    if hasattr(config, "to_dict"):
        return ConfigDict(config.to_dict())
    return ConfigDict(dict(config))


class _PatchModel(StateDictModule):
    """Tiling + blending driver shared by PatchRefiner / PatchRefinerPlus (they inherit it from
    BaselinePretrain in the reference)."""

    crop_channels = 3
    crop_mean, crop_std = IMAGENET_MEAN, IMAGENET_STD
    STRICT_DA_ZOE = False
    supports_return_device = True  # forward(return_device=True): the depth map stays on the GPU (Tester scores it there)
    blend_border = 0.15      # generatemask(..., border=0.15) (patchrefinerplus.py:485); BaselinePretrain: the default 0.1
    needs_coarse = True      # coarse forward + ROI pyramid per tile (False: BaselinePretrain(target='fine'))

    def random_calls(self, cai_mode: str, process_num: int) -> int:
        """random_tile calls of an r<N> mode: N // process_num (patchrefinerplus.py:517-520, patchrefiner.py:388-391)"""
        return int(cai_mode[1:]) // process_num

    # -- BaselinePretrain.prepare_tile_cfg (baseline_pretrain.py:96-124) ---------------------------
    def prepare_tile_cfg(self, image_raw_shape, patch_split_num):
        ph, pw = self.patch_process_shape
        sh, sw = patch_split_num
        if image_raw_shape[0] % (2 * sh) or image_raw_shape[1] % (2 * sw):
            # the reference only documents this (docs/user_infer.md:16, asserts commented out)
            raise ValueError(f"image_raw_shape {list(image_raw_shape)} must be divisible by 2*patch_split_num "
                             f"{[2 * sh, 2 * sw]}")
        raw = (image_raw_shape[0] // sh, image_raw_shape[1] // sw)
        return dict(patch_split_num=list(patch_split_num), patch_reensemble_shape=(ph * sh, pw * sw),
                    patch_raw_shape=raw, image_raw_shape=list(image_raw_shape),
                    raw_h_split_point=[int(raw[0] * i) for i in range(sh)],
                    raw_w_split_point=[int(raw[1] * i) for i in range(sw)])

    # -- tile plan --------------------------------------------------------------------------------
    def plan_tiles(self, tile_cfg, cai_mode: str, process_num: int):
        """Every tile of the frame, in the reference's execution order.  Returns a list of passes
        ``dict(kind, raw=[(h,w)], proc=[(h,w)])``; kind 'init' | 'grid' | 'random'."""
        H, W = tile_cfg["image_raw_shape"]
        rh, rw = tile_cfg["patch_raw_shape"]
        RH, RW = tile_cfg["patch_reensemble_shape"]
        ph, pw = self.patch_process_shape

        def grid(off, offp):
            assert off[0] >= 0 and off[1] >= 0
            hs = [rh * i + off[0] for i in range((H - off[0]) // rh)]
            ws = [rw * i + off[1] for i in range((W - off[1]) // rw)]
            hps = [ph * i + offp[0] for i in range((RH - offp[0]) // ph)]
            wps = [pw * i + offp[1] for i in range((RW - offp[1]) // pw)]
            return [(h, w) for h in hs for w in ws], [(h, w) for h in hps for w in wps]

        passes = []
        r, p = grid((0, 0), (0, 0))
        passes.append(dict(kind="init", raw=r, proc=p))
        if cai_mode == "m2" or cai_mode[0] == "r":
            for off, offp in (((0, rw // 2), (0, pw // 2)), ((rh // 2, 0), (ph // 2, 0)),
                              ((rh // 2, rw // 2), (ph // 2, pw // 2))):
                r, p = grid(off, offp)
                if not r:
                    # the reference dies here too (torch.stack of an empty crop list, baseline_pretrain.py:280)
                    raise RuntimeError(f"cai_mode {cai_mode!r} needs patch_split_num >= 2 on both axes (got "
                                       f"{tile_cfg['patch_split_num']}): a half-offset pass would be empty")
                passes.append(dict(kind="grid", raw=r, proc=p))
        elif cai_mode != "m1":
            raise ValueError(f"unknown cai_mode {cai_mode!r} (expected m1, m2 or r<N>)")
        if cai_mode[0] == "r":
            tiles = []
            for _ in range(self.random_calls(cai_mode, process_num)):
                # baseline_pretrain.py:160-161: process_num h-starts, then ONE w-start, per call
                hs = [random.randint(0, H - rh - 1) for _ in range(process_num)]
                ws = random.randint(0, W - rw - 1)
                tiles += [(h, ws) for h in hs]
            passes.append(dict(kind="random", raw=tiles, proc=tiles))
        return passes

    # -- coarse pyramid ROI + crops for a set of tiles ---------------------------------------------
    def _boxes(self, tiles, tile_cfg) -> np.ndarray:
        """bboxs * bboxs_feat_factor in float32 (baseline_pretrain.py:289-296)."""
        H, W = tile_cfg["image_raw_shape"]
        rh, rw = tile_cfg["patch_raw_shape"]
        ph, pw = self.patch_process_shape
        bb = np.array([[w, h, w + rw, h + rh] for h, w in tiles], dtype=np.int32).astype(np.float32)
        fac = np.array([1 / W * pw, 1 / H * ph, 1 / W * pw, 1 / H * ph], dtype=np.float32)
        return (bb * fac[None]).astype(np.float32)

    def _prepare_batch(self, image_hr_chw, t_dev, boxes_dev, tile_cfg, coarse_feats: List[Feat], coarse_depth: Feat):
        """t_dev: int32 [k, 2] (h_start, w_start) of the batch's tiles, boxes_dev: fp32 [k, 4] lr-frame ROI boxes, on the device"""
        dev = image_hr_chw.device
        ph, pw = self.patch_process_shape
        rh, rw = tile_cfg["patch_raw_shape"]
        K = t_dev.shape[0]
        crops = Feat.alloc(K, ph, pw, self.crop_channels, dev, pad_to=4)
        ops.crop_resize(image_hr_chw, t_dev, rh, rw, ph, pw, self.crop_mean, self.crop_std, crops)
        if not self.needs_coarse:
            return crops, None, None
        # roi_align(feat, boxes, (h, w), h / ph, aligned=True) per level (patchrefinerplus.py:268-276)
        # (not materialised: each level is gathered straight into the concat buffers that consume it -- ops.RoiSource)
        rois = [ops.RoiSource(f, boxes_dev, f.h / ph, f.h, f.w) for f in coarse_feats]
        depth_roi = ops.roi_align(coarse_depth, boxes_dev, coarse_depth.h / ph, coarse_depth.h, coarse_depth.w,
                                  out=Feat(torch.empty((K, coarse_depth.h, coarse_depth.w, 1), device=dev)))
        return crops, rois, depth_roi

    # -- forward -----------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, mode=None, image_lr=None, image_hr=None, crops_image_hr=None, depth_gt=None, crop_depths=None,
                bboxs=None, tile_cfg=None, cai_mode="m1", process_num=4, select_patch=-1, shard=None,
                return_device=False, gather_dst=None, next_image_lr=None, frame_index=None, **kwargs):
        """``shard=(rank, world)``: this process computes its share of the frame's tiles (``shard_layout``) and the predictions
        are exchanged (RCCL): all-gather when ``gather_dst`` is None (every rank blends and returns the map), gather to rank
        ``gather_dst`` otherwise (only that rank blends; the others return ``depth=None``).
        ``next_image_lr``: the low-resolution image of the frame the caller will submit NEXT (a video / dataset loop knows
        it): its coarse forward -- one image through the backbone, ~10 ms of kernels that cover a fraction of the chip -- is
        enqueued on a stream of its own beside this frame's tile batches and picked up by the next call (same tensor object);
        results are bit-identical with and without.  In the patch-sharded mode ONE rank computes it (the owner rotates with the frame) and
        broadcasts pyramid + tap tables (``_prefetch_coarse_sharded``); ``frame_index``: the frame's number when the caller's loop is not the
        only one driving this model (tools that emulate several ranks with one object) -- by default the sharded frames are counted."""
        if mode != "infer":
            raise NotImplementedError("only mode='infer' is built (training is out of scope, SURVEY.md 2 #12-13)")
        if select_patch != -1:
            raise NotImplementedError("select_patch (feature visualisation hook) is not on the inference path")
        if not (image_lr.is_cuda and image_hr.is_cuda):
            raise RuntimeError("image_lr / image_hr must be on the GPU (tester.py:43-49 moves them); no CPU path")
        # every kernel is enqueued on the current device's stream: make the inputs' device current for the whole frame
        with torch.cuda.device(image_hr.device):
            self._next_lr = next_image_lr
            self._frame_index = frame_index
            try:
                # (the guard follows the fusion model's own flag: PRV2_F16F6=1 switches the layers on under arith 'bf16x3' as well)
                f16f6 = getattr(self, "arith", None) == "f16f6" or bool(getattr(getattr(self, "refiner_fusion_model", None), "f16f6", False))
                guard = f16f6 and ops.F6Range.active(image_hr.device)
                rnd = random.getstate() if guard else None
                out = self._infer(image_lr, image_hr, depth_gt, tile_cfg, cai_mode, process_num, shard, return_device, gather_dst)
                if guard:
                    # fp16 range guard of the fp16 + fp6 layers (ops.F6Range): a frame in which a layer's input left fp16's range is
                    # computed again with that layer's power-of-two input scale moved (the tile plan's random draws are replayed)
                    reduce = None
                    if shard is not None and shard[1] > 1:  # one decision for all ranks of a patch-sharded frame: the tables' maximum (1 KB all-reduce)
                        import torch.distributed as dist
                        if dist.is_available() and dist.is_initialized():  # (not in the one-process emulations of the ranks: tests' ShardEmulation, tools/shard_model.py)
                            reduce = lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX)  # noqa: E731
                    self.f6_guarded_frames = getattr(self, "f6_guarded_frames", 0) + 1
                    for attempt in range(3):
                        redo = ops.F6Range.check(image_hr.device, reduce)
                        if ops.F6Range.moved:  # (also an early move without a redo: a captured frame carries the old scales as kernel arguments)
                            self.__dict__.pop("_graphs", None)
                        if not redo:
                            break
                        if attempt == 2:
                            raise RuntimeError(f"f16f6: input range of {len(redo)} layer(s) not representable after two recalibrations "
                                               f"(largest |x x_scale| seen: {[m for _, m, _ in redo]}); use prec='bf16x3'")
                        self.f6_recalibrations = getattr(self, "f6_recalibrations", 0) + 1
                        self.__dict__.pop("_graphs", None)  # (a captured frame carries the old scales)
                        random.setstate(rnd)
                        out = self._infer(image_lr, image_hr, depth_gt, tile_cfg, cai_mode, process_num, shard, return_device, gather_dst)
                return out
            finally:
                self._next_lr = None
                self._frame_index = None

    __call__ = forward

    def _infer(self, image_lr, image_hr, depth_gt, tile_cfg, cai_mode, process_num, shard, return_device, gather_dst):
        tile_cfg = self.tile_cfg if tile_cfg is None else self.prepare_tile_cfg(tile_cfg["image_raw_shape"],
                                                                              tile_cfg["patch_split_num"])
        assert image_hr.shape[0] == 1
        dev = image_hr.device
        # ---- host: the frame's tile plan (consumes Python's ``random`` in the reference's order) ----------------------
        passes = self.plan_tiles(tile_cfg, cai_mode, process_num)
        self.last_plan = passes
        if shard is not None and shard[1] > 1:
            return self._infer_sharded(image_lr, image_hr, depth_gt, passes, tile_cfg, process_num, shard, return_device, gather_dst)
        flat = [t for p in passes for t in p["raw"]]
        idx = list(range(len(flat)))
        if shard is not None:
            idx = idx[shard[0]::shard[1]]
        mine = [flat[i] for i in idx]
        # the plan's coordinates: [my tiles (h, w) | every tile's blend coordinates (h, w)] and my tiles' ROI boxes
        n_mine, n_all = len(mine), len(flat)
        tiles_i = torch.tensor(mine + [t for p in passes for t in p["proc"]], dtype=torch.int32).view(-1, 2)
        boxes_f = torch.from_numpy(self._boxes(mine, tile_cfg)) if self.needs_coarse else torch.zeros((n_mine, 4))
        plan = dict(kinds=[p["kind"] for p in passes], counts=[len(p["raw"]) for p in passes], n_mine=n_mine, n_all=n_all)

        use_graph = bool(getattr(self, "hip_graph", False)) and shard is None and not ops.PROFILER.enabled
        if use_graph:
            depth, coarse_prediction = self._graph_frame(image_lr, image_hr, tiles_i, boxes_f, plan, tile_cfg, process_num, cai_mode)
        else:
            tiles_dev = tiles_i.to(dev, non_blocking=True)
            boxes_dev = boxes_f.to(dev, non_blocking=True)
            depth, coarse_prediction = self._device_frame(image_lr, image_hr, tiles_dev, boxes_dev, plan, tile_cfg, process_num,
                                                          shard, gather_dst)
            if depth is None:  # gather-to-one: this rank's part of the frame is done
                return None, dict(rgb=image_lr, depth_pred=None, depth_gt=depth_gt, coarse_prediction=coarse_prediction)
        if not return_device:
            depth = self._to_host(depth)
        elif use_graph:
            depth = depth.clone()  # the graph's own output buffer is rewritten by the next replay
        return depth, dict(rgb=image_lr, depth_pred=depth, depth_gt=depth_gt, coarse_prediction=coarse_prediction)

    @staticmethod
    def _to_host(depth):
        """the reference returns a fresh CPU tensor; through torch's caching pinned-memory allocator the 33 MB D2H runs at
        PCIe rate instead of being staged through a pageable buffer (3.4 -> ~1 ms per 4K frame)"""
        hostd = torch.empty(depth.shape, dtype=depth.dtype, pin_memory=True)
        hostd.copy_(depth, non_blocking=True)
        torch.cuda.current_stream(depth.device).synchronize()
        return hostd

    # ==============================================================================================================
    # patch-sharded mode: tiles of ONE frame over the ranks, one process per GPU, RCCL over xGMI
    # ==============================================================================================================
    def _infer_sharded(self, image_lr, image_hr, depth_gt, passes, tile_cfg, process_num, shard, return_device, gather_dst):
        """The frame's plan as ONE int32 tensor [2 * n_all, 2] = (every tile's crop origin | every tile's blend origin), moved to
        the device and overwritten there by rank 0's (an RCCL broadcast enqueued on the stream: no pickling, no host blocking --
        a rank whose ``random`` state differs would otherwise blend the others' predictions at its own coordinates).  Everything
        that depends on coordinates (crops, ROI boxes, blend) reads the device copy; everything the host needs (pass kinds and
        counts, who owns which tile) follows from (cai_mode, process_num, patch_split_num) alone."""
        dev = image_hr.device
        n_all = sum(len(p["raw"]) for p in passes)
        # staged through a pinned buffer (one per plan size, reused once the previous frame's copy out of it has completed): the
        # H2D copy is then really asynchronous -- a pageable source would block the host against the worker streams
        pins = self.__dict__.setdefault("_plan_pins", {})
        ent = pins.get((n_all, str(dev)))
        if ent is None:
            ent = pins[(n_all, str(dev))] = dict(buf=torch.empty((2 * n_all, 2), dtype=torch.int32).pin_memory(), done=None)
        if ent["done"] is not None:
            ent["done"].synchronize()
        ent["buf"].copy_(torch.tensor([t for p in passes for t in p["raw"]] + [t for p in passes for t in p["proc"]], dtype=torch.int32).view(2 * n_all, 2))
        plan_t = ent["buf"].to(dev, non_blocking=True)
        ent["done"] = torch.cuda.Event()
        ent["done"].record(torch.cuda.current_stream(dev))
        plan_t = self._sync_plan_tensor(plan_t)
        plan = dict(kinds=[p["kind"] for p in passes], counts=[len(p["raw"]) for p in passes], n_all=n_all)
        depth, coarse_prediction = self._device_frame_sharded(image_lr, image_hr, plan_t, plan, tile_cfg, shard, gather_dst, process_num)
        if depth is None:  # gather-to-one: this rank's part of the frame is done
            return None, dict(rgb=image_lr, depth_pred=None, depth_gt=depth_gt, coarse_prediction=coarse_prediction)
        if not return_device:
            depth = self._to_host(depth)
        return depth, dict(rgb=image_lr, depth_pred=depth, depth_gt=depth_gt, coarse_prediction=coarse_prediction)

    @staticmethod
    def _sync_plan_tensor(plan_t):
        """rank 0's tile coordinates on every rank (in place; device tensors over RCCL, CPU tensors over gloo in the tests)"""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(plan_t, src=0)
        return plan_t

    def _boxes_dev(self, tiles_dev, tile_cfg):
        """``_boxes`` on the device from int32 tile origins [k, 2] (h, w): int -> float32 is exact and the one multiplication by
        the float32 factor is the same IEEE operation, so the boxes equal the host version bit for bit (tested)."""
        H, W = tile_cfg["image_raw_shape"]
        rh, rw = tile_cfg["patch_raw_shape"]
        ph, pw = self.patch_process_shape
        fac = torch.tensor([1 / W * pw, 1 / H * ph, 1 / W * pw, 1 / H * ph], dtype=torch.float32, device=tiles_dev.device)
        h, w = tiles_dev[:, 0], tiles_dev[:, 1]
        bb = torch.stack([w, h, w + rw, h + rh], dim=1).to(torch.float32)
        return bb * fac[None]

    SHARD_DST_COST = 1.5  # what the blending rank does BEHIND the last group (receive, blend, D2H of the map: ~3-4 ms at 4K),
    #                       in tile-times: tiles of the last group move from it to other ranks while that shortens the modelled
    #                       critical path max(others' tiles, its tiles + cost); config key shard_dst_cost

    SHARD_MERGE_BELOW = 8   # from 8 ranks on, a first gather group with fewer than this many tiles PER RANK is merged with the random tiles:
    #                         6 + 4 tiles per rank in two batches under-fill the chip (2.9 ms per tile against 2.2 at 41: the r04 one-GPU model lost
    #                         36 % at 8 ranks to it); one 10-tile batch per rank and one exchange instead -- rank 0 then blends everything behind the
    #                         last batch (order unchanged: bit-identical).  Config key shard_merge_below (0: never merge).

    SHARD_OWNER_COST = 3.5  # what the rank that computes the NEXT frame's coarse pyramid does beside its tiles (the coarse forward: ~8 of the ~11 ms
    #                         every rank used to spend per frame on the headline; the ~2-3 ms of tap tables stay with every rank), in tile-times;
    #                         config key shard_owner_cost

    def shard_layout(self, kinds, counts, world, dst=None, owner=None):
        """Who computes which tile, and what is exchanged when (pure host arithmetic on the plan's pass structure; cached).
        Passes form GATHER GROUPS -- [init + half-offset grids] and [random tiles] -- so that the receiving rank pastes /
        blends the first group (at the re-ensemble resolution) while every rank still computes the second.  Inside a group the
        tiles are dealt round-robin over the ranks; in the LAST group the blending rank ``dst`` hands tiles to the others while
        that shortens max(others' tiles, its tiles + shard_dst_cost) -- its receive + blend + D2H tail.  ``owner``: the rank that computes the
        next frame's coarse pyramid beside this frame's tiles (``_prefetch_coarse_sharded``): it hands tiles over the same way (shard_owner_cost).
        Returns a list of groups: dict(passes=[pass indices], base, n, per, owner=[rank per tile], mine=[[tile offsets in the
        group] per rank], perm=[position of every tile of the group in the rank-major [world * per] gathered stack])."""
        merge_below = int(getattr(self, "shard_merge_below", self.SHARD_MERGE_BELOW))
        key = (tuple(kinds), tuple(counts), int(world), dst, float(getattr(self, "shard_dst_cost", self.SHARD_DST_COST)), merge_below, owner,
               float(getattr(self, "shard_owner_cost", self.SHARD_OWNER_COST)))
        cache = self.__dict__.setdefault("_shard_layouts", {})
        if key in cache:
            return cache[key]
        cost = key[4]
        extra = [0.0] * world  # tile-times a rank spends beside its tiles
        if dst is not None and world > 1:
            extra[dst] += cost
        if owner is not None and world > 1:
            extra[owner] += key[7]
        fixed = [i for i, k in enumerate(kinds) if k != "random"]
        rnd = [i for i, k in enumerate(kinds) if k == "random" and counts[i] > 0]
        groups = []
        id_groups = [ids for ids in (fixed, rnd) if ids]
        if len(id_groups) == 2 and world >= 8 and sum(counts[i] for i in fixed) < merge_below * world:
            id_groups = [sorted(fixed + rnd)]      # one batch and one exchange per rank (pass order kept: the blend walks g["passes"] in order)
        for ids in id_groups:
            n = sum(counts[i] for i in ids)
            share = [n // world] * world
            order = [r for r in range(world - 1, -1, -1) if r != dst] + ([dst] if dst is not None else [])
            for r in order[:n % world]:         # the remainder: highest ranks first, the blending rank last
                share[r] += 1
            if world > 1 and ids is id_groups[-1] and any(extra):
                while True:                     # the busiest of the ranks with side work hands a tile to the least loaded rank while the critical path shrinks
                    busy = [q for q in range(world) if extra[q] > 0 and share[q] > 0]
                    if not busy:
                        break
                    b = max(busy, key=lambda q: (share[q] + extra[q], -q))
                    r = min((q for q in range(world) if q != b), key=lambda q: (share[q] + extra[q], q))
                    now = max(share[q] + extra[q] for q in range(world))
                    then = max(share[q] + extra[q] + (q == r) - (q == b) for q in range(world))
                    if then >= now:
                        break
                    share[b] -= 1
                    share[r] += 1
            owner, left, r = [], list(share), 0
            for _ in range(n):                  # round-robin deal, skipping ranks that are full
                while left[r % world] == 0:
                    r += 1
                owner.append(r % world)
                left[r % world] -= 1
                r += 1
            mine = [[i for i, o in enumerate(owner) if o == q] for q in range(world)]
            per = max(share)
            slot = {}
            for q in range(world):
                for sl, i in enumerate(mine[q]):
                    slot[i] = q * per + sl
            groups.append(dict(passes=ids, base=sum(counts[:ids[0]]), n=n, per=per, owner=owner, mine=mine, share=share,
                               perm=[slot[i] for i in range(n)]))
        cache[key] = groups
        return groups

    def _device_frame_sharded(self, image_lr, image_hr, plan_t, plan, tile_cfg, shard, gather_dst, process_num=4):
        """A frame on rank ``shard[0]`` of ``shard[1]``: coarse forward (every rank; next frame's beside the tiles when
        announced), this rank's tiles group by group (``shard_layout``), the group's prediction stacks exchanged ASYNCHRONOUSLY
        (RCCL gather to ``gather_dst`` / all-gather on the collective's own stream) while the next group computes, and on the
        receiving rank(s) the overlap blend of a group as soon as its stacks have arrived -- in the reference's order, so the
        map is bit-identical to the unsharded one.  No host synchronisation inside."""
        rank, world = shard
        dev = image_hr.device
        ph, pw = self.patch_process_shape
        rh, rw = tile_cfg["patch_raw_shape"]
        RH, RW = tile_cfg["patch_reensemble_shape"]
        n_all = plan["n_all"]
        # The NEXT frame's coarse pyramid is computed by ONE rank beside this frame's tiles and broadcast (~100 MB): the owner rotates with the
        # frame index, every other rank spends that time on tiles (shard_layout hands the owner's tiles over).  Every rank replicating the coarse
        # forward was the cap of the patch-sharded mode: ~11 ms beside ~18 ms of tiles at 8 ranks.
        fi = self.__dict__.get("_frame_index")
        if fi is None:
            fi = self._shard_frames = getattr(self, "_shard_frames", -1) + 1  # (every rank calls the frames in the same order)
        next_lr = getattr(self, "_next_lr", None)
        rotate = bool(getattr(self, "shard_rotate_coarse", True)) and self.needs_coarse and next_lr is not None and world > 1 and not ops.PROFILER.enabled
        owner = (fi + 1) % world if rotate and self._coarse_recipe_of(next_lr, tile_cfg) is not None else None
        self.last_coarse_owner = owner
        groups = self.shard_layout(plan["kinds"], plan["counts"], world, gather_dst, owner)
        self.last_shard_layout = groups
        if self.needs_coarse:
            coarse_feats, coarse_prediction = self._coarse_of(image_lr, tile_cfg)
            coarse_depth = Feat(coarse_prediction.view(1, coarse_prediction.shape[-2], coarse_prediction.shape[-1], 1))
        else:
            coarse_feats = coarse_prediction = coarse_depth = None
        image_chw = image_hr[0].contiguous().float()
        raw_all, proc_all = plan_t[:n_all], plan_t[n_all:]
        bs = max(1, int(getattr(self, "max_batch", None) or process_num))  # (as the unsharded path: _device_frame)
        n_streams = max(1, int(getattr(self, "n_streams", None) or 1))
        main = torch.cuda.current_stream(dev)
        streams = [main] if n_streams == 1 else self._streams(dev, n_streams)

        def layout_dev(g):
            """this rank's tile indices and the group's permutation as device tensors, made once per layout (the layout is cached:
            no per-frame blocking H2D copies)"""
            ent = g.setdefault("_dev", {}).get((rank, str(dev)))
            if ent is None:
                ent = g["_dev"][(rank, str(dev))] = (torch.tensor([g["base"] + i for i in g["mine"][rank]], dtype=torch.int64, device=dev),
                                                     torch.tensor(g["perm"], dtype=torch.int64, device=dev))
            return ent

        # per group: the (padded) stack this rank sends and its tiles' coordinates, allocated before the worker streams start
        stacks, coords = [], []
        for g in groups:
            # (one send stack per layout and rank, zeroed once: rows behind this rank's share are padding nobody reads; the previous frame's
            #  exchange of it was waited for on the main stream before that frame returned, and the worker streams start behind ``ready``)
            sk = g.setdefault("_stacks", {})
            if (rank, str(dev), ph, pw) not in sk:
                sk[(rank, str(dev), ph, pw)] = torch.zeros((g["per"], 1, ph, pw), device=dev)
            stacks.append(sk[(rank, str(dev), ph, pw)])
            idx = layout_dev(g)[0]
            t = raw_all.index_select(0, idx)
            coords.append((t, self._boxes_dev(t, tile_cfg) if self.needs_coarse else None))
        if n_streams > 1:
            ready = torch.cuda.Event()
            ready.record(main)
            for st in streams:
                st.wait_event(ready)
        receiver = gather_dst is None or rank == gather_dst
        ram = DeviceRunningAverageMap(RH, RW, dev) if receiver else None
        mask = blend_mask((ph, pw), self.blend_border, 0.0, dev) if receiver else None
        self.last_exchange_bytes = 0
        bi = 0

        def launch(gi):
            nonlocal bi
            t, boxes = coords[gi]
            k = t.shape[0]
            used = []
            for s0 in range(0, k, bs):
                e = min(s0 + bs, k)
                st = streams[bi % len(streams)]
                bi += 1
                used.append(st)
                with torch.cuda.stream(st):
                    crops, rois, depth_roi = self._prepare_batch(image_chw, t[s0:e], boxes[s0:e] if boxes is not None else None,
                                                                 tile_cfg, coarse_feats, coarse_depth)
                    self.infer_forward(crops, rois, depth_roi, out=stacks[gi][s0:e])
            return used

        def blend(gi, allp):
            g = groups[gi]
            perm = layout_dev(g)[1]
            preds = allp.view(world * g["per"], ph, pw).index_select(0, perm)
            o = 0
            for pi in g["passes"]:
                kind, k = plan["kinds"][pi], plan["counts"][pi]
                pr = preds[o:o + k]
                tdev = proc_all[g["base"] + o:g["base"] + o + k]
                o += k
                if kind == "init":
                    ram.paste(pr, mask, tdev, ph, pw)
                elif kind == "grid":
                    ram.update(pr, mask, tdev, ph, pw)
                else:
                    ram.resize(tile_cfg["image_raw_shape"])
                    ram.update(pr, blend_mask((rh, rw), self.blend_border, 1e-3, dev), tdev, rh, rw)

        pending = None  # (group index, exchange handle) whose stacks are on their way
        unwaited = []   # handles of exchanges this rank has not waited for (sending-only ranks): EVERY one is waited for before returning
        for gi in range(len(groups)):
            used = launch(gi)
            if gi == 0:
                if owner is None:
                    self._prefetch_coarse(next_lr, main, tile_cfg, record_recipe=rotate)
                else:
                    self._prefetch_coarse_sharded(next_lr, main, tile_cfg, rank, world, owner)
            if pending is not None:
                if receiver:                       # blend the previous group on the main stream while this one computes
                    allp = pending[1]()
                    receiver = allp is not None    # (None on a receiving rank: the tests' recording pass of the shard emulation)
                    if receiver:
                        blend(pending[0], allp)
                else:
                    unwaited.append(pending[1])
            for st in used:
                if st is not main:
                    main.wait_stream(st)
            pending = (gi, self._exchange_begin(stacks[gi], shard, gather_dst, gi))
        if pending is not None:
            if receiver:
                allp = pending[1]()
                receiver = allp is not None
                if receiver:
                    blend(pending[0], allp)
            else:
                unwaited.append(pending[1])
        for h in unwaited:  # the send buffers (``stacks``) must outlive their collectives: the main stream waits for each of them
            h()
        if receiver and "random" in plan["kinds"] and not any(plan["kinds"][i] == "random" for g in groups for i in g["passes"]):
            ram.resize(tile_cfg["image_raw_shape"])  # an r-mode whose N // process_num is 0: the resize still happens
        if not receiver:
            return None, coarse_prediction
        return ram.avg[None, None], coarse_prediction

    def _exchange_begin(self, mine, shard, dst, group=0):
        """start the exchange of equally sized stacks; returns a callable that makes the current stream wait for it and gives
        the rank-major [world * per, ...] result (None on ranks that do not receive).  RCCL: asynchronous on the collective's
        stream.  A patched ``_exchange`` (tests: single-GPU emulation of the ranks) is called synchronously."""
        self.last_exchange_bytes = getattr(self, "last_exchange_bytes", 0) + mine.numel() * 4 * (shard[1] - 1)
        if "_exchange" in self.__dict__:
            res = self._exchange(mine, shard, dst, group)
            return lambda: res
        import torch.distributed as dist
        rank, world = shard
        if dst is None:
            allp = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), device=mine.device)
            work = dist.all_gather_into_tensor(allp, mine, async_op=True)
            parts = None
        else:
            # (one rank-major receive buffer, its per-rank views as the gather list: no torch.cat behind the collective)
            allp = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), device=mine.device) if rank == dst else None
            parts = list(allp.split(mine.shape[0], dim=0)) if allp is not None else None
            work = dist.gather(mine, parts, dst=dst, async_op=True)

        def result():
            work.wait()  # (RCCL: the current stream waits; gloo: the host does)
            return allp
        return result

    def _boxes_prenorm(self, tiles, tile_cfg) -> np.ndarray:
        """the DATASET's pre-normalised bboxs (pre_norm_bbox=True, u4k_dataset.py:171-176: int64 tensor / W * pw in float32) --
        what the reference's ``mode='train'`` forward receives; differs from the infer path's ``bboxs.int() * (1 / W * pw)``
        (``_boxes``) in the last bit"""
        H, W = (np.float32(v) for v in tile_cfg["image_raw_shape"])
        rh, rw = tile_cfg["patch_raw_shape"]
        ph, pw = (np.float32(v) for v in self.patch_process_shape)
        f = np.float32
        return np.array([[f(w) / W * pw, f(h) / H * ph, f(w + rw) / W * pw, f(h + rh) / H * ph] for h, w in tiles], dtype=np.float32)

    @torch.no_grad()
    def predict_tiles(self, image_lr, image_hr, tiles, tile_cfg=None, prenorm_bbox=False):
        """Per-tile predictions [K, 1, ph, pw] (device) for explicit tile origins ``tiles`` = [(h_start, w_start), ...] of
        patch_raw_shape-sized crops, no blending: what the reference's ``mode='train'`` forward computes for given crops /
        bboxs (coarse forward + ROI of the crop + refiner; patchrefinerplus.py:405-467) -- the building block of
        Tester.run_consistency (tester.py:211-321).  ``prenorm_bbox``: ROI boxes in the dataset's arithmetic (``_boxes_prenorm``)
        instead of the infer path's."""
        if not hasattr(self, "infer_forward") or (not self.needs_coarse and getattr(self, "target", "fine") == "coarse"):
            raise NotImplementedError(f"{type(self).__name__}(target='coarse') has no per-tile path (one backbone forward on image_lr, "
                                      "baseline_pretrain.py:409-411): predict_tiles / run_consistency need a tiling model")
        tile_cfg = self.tile_cfg if tile_cfg is None else self.prepare_tile_cfg(tile_cfg["image_raw_shape"], tile_cfg["patch_split_num"])
        dev = image_hr.device
        ph, pw = self.patch_process_shape
        with torch.cuda.device(dev):
            if self.needs_coarse:
                feats, cp = self.coarse_forward(image_lr)
                self._prepare_frame(feats, tile_cfg)
                cd = Feat(cp.view(1, cp.shape[-2], cp.shape[-1], 1))
            else:
                feats = cd = None
            t_dev = torch.tensor(list(tiles), dtype=torch.int32).view(-1, 2).to(dev)
            mk = self._boxes_prenorm if prenorm_bbox else self._boxes
            boxes = torch.from_numpy(mk(list(tiles), tile_cfg)).to(dev) if self.needs_coarse else None
            preds = torch.empty((len(tiles), 1, ph, pw), device=dev)
            bs = max(1, int(getattr(self, "max_batch", None) or 4))
            image_chw = image_hr[0].contiguous().float()
            for s0 in range(0, len(tiles), bs):
                e = min(s0 + bs, len(tiles))
                crops, rois, droi = self._prepare_batch(image_chw, t_dev[s0:e], boxes[s0:e] if boxes is not None else None, tile_cfg, feats, cd)
                self.infer_forward(crops, rois, droi, out=preds[s0:e])
            ops.F6Range.clear(dev)  # (no guarded frame here: what these launches saw must not be judged by the next frame's check)
        return preds

    def _device_frame(self, image_lr, image_hr, tiles_dev, boxes_dev, plan, tile_cfg, process_num, shard=None, gather_dst=None):
        """Everything of a frame that runs on the device, given the plan's coordinates in device memory: coarse forward, the
        per-patch networks over this rank's tiles (batches round-robin over the HIP streams), the exchange (sharded mode), the
        overlap blend.  No host synchronisation inside (captured into a hipGraph by ``_graph_frame``).
        tiles_dev: int32 [n_mine + n_all, 2]: this rank's tile origins, then every tile's blend coordinates."""
        dev = image_hr.device
        ph, pw = self.patch_process_shape
        rh, rw = tile_cfg["patch_raw_shape"]
        RH, RW = tile_cfg["patch_reensemble_shape"]
        n_mine, n_all = plan["n_mine"], plan["n_all"]
        if self.needs_coarse:
            coarse_feats, coarse_prediction = self._coarse_of(image_lr, tile_cfg)
            coarse_depth = Feat(coarse_prediction.view(1, coarse_prediction.shape[-2], coarse_prediction.shape[-1], 1))
        else:
            coarse_feats = coarse_prediction = coarse_depth = None

        # ---- per-patch networks over the tile list (any batching; optional rank sharding) ----
        image_chw = image_hr[0].contiguous().float()
        bs = max(1, int(getattr(self, "max_batch", None) or process_num))
        preds = torch.empty((n_mine, 1, ph, pw), device=dev)  # this rank's predictions, in tile order
        # Tile batches are independent: they are issued round-robin on ``n_streams`` HIP streams so that the
        # HBM-bound kernels of one batch (gathers, LayerNorm, gate 1x1s) run beside the MFMA-bound convs of another.
        n_streams = max(1, int(getattr(self, "n_streams", None) or 1))
        main = torch.cuda.current_stream(dev)
        streams = [main] if n_streams == 1 else self._streams(dev, n_streams)
        if n_streams > 1:
            ready = torch.cuda.Event()
            ready.record(main)
        for bi, s in enumerate(range(0, n_mine, bs)):
            e = min(s + bs, n_mine)
            st = streams[bi % len(streams)]
            with torch.cuda.stream(st):
                if n_streams > 1 and bi < len(streams):
                    st.wait_event(ready)
                crops, rois, depth_roi = self._prepare_batch(image_chw, tiles_dev[s:e], boxes_dev[s:e], tile_cfg, coarse_feats,
                                                             coarse_depth)
                self.infer_forward(crops, rois, depth_roi, out=preds[s:e])
        self._prefetch_coarse(getattr(self, "_next_lr", None), main, tile_cfg)
        if n_streams > 1:
            for st in streams:
                main.wait_stream(st)
        preds = preds.view(n_all, ph, pw)

        # ---- overlap blend, in the reference's order ----------------------------------------------
        mask = blend_mask((ph, pw), self.blend_border, 0.0, dev)
        ram = DeviceRunningAverageMap(RH, RW, dev)
        all_proc = tiles_dev[n_mine:]
        o = 0
        for kind, k in zip(plan["kinds"], plan["counts"]):
            pr = preds[o:o + k]
            tdev = all_proc[o:o + k]
            o += k
            if kind == "init":
                ram.paste(pr, mask, tdev, ph, pw)
            elif kind == "grid":
                ram.update(pr, mask, tdev, ph, pw)
            else:
                mask_r = blend_mask((rh, rw), self.blend_border, 1e-3, dev)  # generatemask(...) + 1e-3 (patchrefinerplus.py:514)
                ram.resize(tile_cfg["image_raw_shape"])
                if k:
                    ram.update(pr, mask_r, tdev, rh, rw)
        return ram.avg[None, None], coarse_prediction

    def _graph_frame(self, image_lr, image_hr, tiles_i, boxes_f, plan, tile_cfg, process_num, cai_mode):
        """``hip_graph=True``: the device side of a frame (``_device_frame``: a few thousand launches on up to ``n_streams``
        streams) is captured once per (mode, geometry, batching) into a hipGraph and replayed per frame; the per-frame inputs
        -- the two images and the plan's coordinates (random tiles change from frame to frame) -- are copied into the graph's
        static buffers first.  The first frame of a key runs eagerly (it fills the constant caches: blend masks, position
        embeddings / relative-position biases), the second is captured.  Results are bit-identical to the eager path."""
        dev = image_hr.device
        key = (cai_mode, process_num, tuple(tile_cfg["image_raw_shape"]), tuple(tile_cfg["patch_split_num"]), tuple(image_lr.shape),
               getattr(self, "max_batch", None), getattr(self, "n_streams", 1), tuple(plan["counts"]), str(dev))
        cache = self.__dict__.setdefault("_graphs", {})
        ent = cache.get(key)
        if ent is None:  # first frame: eager (warm-up of every constant cache)
            cache[key] = "warm"
            return self._device_frame(image_lr, image_hr, tiles_i.to(dev, non_blocking=True), boxes_f.to(dev, non_blocking=True), plan,
                                      tile_cfg, process_num)
        if ent == "warm":
            st = dict(lr=torch.empty_like(image_lr), hr=torch.empty_like(image_hr), tiles=torch.empty(tiles_i.shape, dtype=torch.int32, device=dev),
                      boxes=torch.empty(boxes_f.shape, dtype=torch.float32, device=dev),
                      h_tiles=torch.empty(tiles_i.shape, dtype=torch.int32).pin_memory(), h_boxes=torch.empty(boxes_f.shape, dtype=torch.float32).pin_memory())
            torch.cuda.current_stream(dev).synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["depth"], st["coarse"] = self._device_frame(st["lr"], st["hr"], st["tiles"], st["boxes"], plan, tile_cfg, process_num)
            st["graph"] = g
            cache[key] = ent = st
        if ent.get("staged") is not None:
            ent["staged"].synchronize()  # the previous replay's H2D copies out of the pinned buffers rewritten below
        ent["h_tiles"].copy_(tiles_i)
        ent["h_boxes"].copy_(boxes_f)
        ent["lr"].copy_(image_lr, non_blocking=True)
        ent["hr"].copy_(image_hr, non_blocking=True)
        ent["tiles"].copy_(ent["h_tiles"], non_blocking=True)
        ent["boxes"].copy_(ent["h_boxes"], non_blocking=True)
        ent["staged"] = torch.cuda.Event()
        ent["staged"].record(torch.cuda.current_stream(dev))
        ent["graph"].replay()
        return ent["depth"], (ent["coarse"].clone() if ent["coarse"] is not None else None)

    def _exchange(self, mine, shard, dst, group=0):
        """blocking form of ``_exchange_begin`` (all-gather when dst is None, gather-to-dst otherwise) -> rank-major
        [world * per, ...] or None.  The single-GPU shard emulation of the tests replaces it on the instance."""
        self.__dict__.setdefault("last_exchange_bytes", 0)
        return self._exchange_begin(mine, shard, dst, group)()

    # -- coarse forward of the NEXT frame beside this frame's tile batches (forward(next_image_lr=...)) -----------------------
    def _prepare_frame(self, coarse_feats, tile_cfg):
        """per-frame work of the per-patch networks that depends on the coarse pyramid only: the coarse half of the fusion convs that
        read cat([., coarse_roi]) (fusion._EncDec.prepare_frame: FusionUnet's encoder_layers_1, BiDirectionalFusion's fusion_layers_1
        and GatedConvUnits) -> the device tensors it created"""
        fm = getattr(self, "refiner_fusion_model", None)
        if fm is None or not hasattr(fm, "prepare_frame"):
            return []
        (rh, rw), (H, W) = tile_cfg["patch_raw_shape"], tile_cfg["image_raw_shape"]
        c_feat = coarse_feats[-self.fusion_feat_level:][::-1]
        fm.prepare_frame(c_feat, (rh / H, rw / W))
        return fm.frame_tensors(c_feat)

    def _coarse_of(self, image_lr, tile_cfg=None):
        """this frame's coarse pyramid: the one prefetched by the previous call if it was made for this very tensor OBJECT,
        unmodified since (the prefetch entry holds the tensor itself: that keeps it alive while the side stream reads it, and
        an address the caching allocator hands out again for another image can never match)"""
        pf = self.__dict__.pop("_coarse_prefetched", None)
        if pf is not None and pf["lr"] is image_lr and pf["version"] == image_lr._version:
            torch.cuda.current_stream(image_lr.device).wait_event(pf["done"])
            for t in pf["tensors"]:  # allocated on the prefetch stream, consumed on this frame's streams from here on
                t.record_stream(torch.cuda.current_stream(image_lr.device))
            self._coarse_hold = pf  # (alive until the next frame replaces it: every consumer stream is long done by then)
            if tile_cfg is not None:
                self._prepare_frame(pf["feats"], tile_cfg)  # (a no-op when the prefetch prepared it for this tiling)
            return pf["feats"], pf["pred"]
        feats, pred = self.coarse_forward(image_lr)
        if tile_cfg is not None:
            self._prepare_frame(feats, tile_cfg)
        return feats, pred

    # -- the next frame's coarse pyramid in the patch-sharded mode: one owner computes, everybody else receives ---------------------
    def _coarse_recipe_key(self, lr, tile_cfg):
        return (tuple(lr.shape), str(lr.device), tuple(tile_cfg["image_raw_shape"]), tuple(tile_cfg["patch_raw_shape"])) if tile_cfg is not None else None

    def _coarse_recipe_of(self, lr, tile_cfg):
        return self.__dict__.get("_coarse_recipes", {}).get(self._coarse_recipe_key(lr, tile_cfg))

    def _record_coarse_recipe(self, lr, tile_cfg, feats, pred):
        """what a rank has to allocate to RECEIVE a coarse pyramid of this image size instead of computing it: the pyramid's buffers and the
        prediction.  (NOT the per-level tap tables ``prepare_frame`` derives from them: 3.8 GB per frame on the headline -- 9 x cout columns at coarse
        resolution + the 3H x 3W knot grids -- against ~100 MB of pyramid; every rank derives its own, ~2 ms.)"""
        bufs, index, fl = [], {}, []
        for f in feats:
            k = f.buf.data_ptr()
            if k not in index:
                index[k] = len(bufs)
                bufs.append(tuple(f.buf.shape))
            fl.append(dict(buf=index[k], c=f.c, c0=f.c0, x2=f.x2))
        self.__dict__.setdefault("_coarse_recipes", {})[self._coarse_recipe_key(lr, tile_cfg)] = dict(bufs=bufs, feats=fl, pred=tuple(pred.shape))

    def _bcast(self, tensors, src):
        """the owner's coarse tensors to every rank, on the CURRENT stream, over a process group of its own (on the frame's communicator the
        broadcast -- which waits for the owner's coarse forward -- would hold up the tile stacks' gathers queued behind it)"""
        if "_bcast_hook" in self.__dict__:  # (tests / tools: one-GPU emulation of the ranks)
            return self._bcast_hook(tensors, src)
        import torch.distributed as dist
        grp = self.__dict__.get("_bcast_group")
        if grp is None:
            grp = self._bcast_group = dist.new_group(ranks=list(range(dist.get_world_size())))
        for t in tensors:
            dist.broadcast(t, src=src, group=grp)

    def _prefetch_coarse_sharded(self, next_lr, main, tile_cfg, rank, world, owner):
        """``_prefetch_coarse`` where only ``owner`` runs the coarse forward: the others allocate the recipe's tensors and receive the pyramid; the
        per-level tap tables are derived from it on every rank.  The pyramid is the owner's bits on every rank: identical to what each rank would
        have computed itself (same kernels, same inputs, one device type)."""
        if torch.cuda.is_current_stream_capturing():
            return
        dev = next_lr.device
        st = self.__dict__.setdefault("_coarse_stream", {}).get(str(dev))
        if st is None:
            st = self._coarse_stream[str(dev)] = torch.cuda.Stream(device=dev, priority=0)
        st.wait_stream(main)
        rec = self._coarse_recipe_of(next_lr, tile_cfg)
        with torch.cuda.stream(st):
            if rank == owner:
                feats, pred = self.coarse_forward(next_lr)
            else:
                bufs = [torch.empty(shape, device=dev, dtype=torch.float32) for shape in rec["bufs"]]
                feats = [Feat(bufs[fr["buf"]], fr["c"], fr["c0"], fr["x2"]) for fr in rec["feats"]]
                pred = torch.empty(rec["pred"], device=dev, dtype=torch.float32)
            uniq, seen = [], set()
            for f in feats:
                if f.buf.data_ptr() not in seen:
                    seen.add(f.buf.data_ptr())
                    uniq.append(f.buf)
            uniq.append(pred)
            self.last_coarse_bcast_bytes = sum(t.numel() * 4 for t in uniq)
            self._bcast(uniq, owner)
            extra = self._prepare_frame(feats, tile_cfg)  # (every rank: the tap tables of the pyramid it now holds)
            uniq = uniq + list(extra)
            done = torch.cuda.Event()
            done.record(st)
        self._coarse_prefetched = dict(lr=next_lr, version=next_lr._version, feats=feats, pred=pred, done=done, tensors=uniq)

    def _prefetch_coarse(self, next_lr, main, tile_cfg=None, record_recipe=False):
        if next_lr is None or not self.needs_coarse or ops.PROFILER.enabled or torch.cuda.is_current_stream_capturing():
            return
        dev = next_lr.device
        st = self.__dict__.setdefault("_coarse_stream", {}).get(str(dev))
        if st is None:
            st = self._coarse_stream[str(dev)] = torch.cuda.Stream(device=dev, priority=0)
        st.wait_stream(main)  # (the image may have been produced on the caller's stream)
        with torch.cuda.stream(st):
            feats, pred = self.coarse_forward(next_lr)
            extra = self._prepare_frame(feats, tile_cfg) if tile_cfg is not None else []
            done = torch.cuda.Event()
            done.record(st)
        tensors = [f.buf for f in feats] + [pred] + list(extra)
        self._coarse_prefetched = dict(lr=next_lr, version=next_lr._version, feats=feats, pred=pred, done=done, tensors=tensors)
        if record_recipe and tile_cfg is not None:  # (patch-sharded mode: from the next frame of this size on, one owner computes and the others receive)
            self._record_coarse_recipe(next_lr, tile_cfg, feats, pred)

    def _invalidate_frame_caches(self):
        """whatever was derived from the weights that have just been replaced: the next frame's prefetched coarse pyramid and
        the captured hipGraphs (they hold raw pointers to the packed weights, relative-position biases and blend masks)"""
        for k in ("_coarse_prefetched", "_coarse_hold", "_graphs"):
            self.__dict__.pop(k, None)

    def load_state_dict(self, sd, strict: bool = True):
        res = super().load_state_dict(sd, strict=strict)
        self._invalidate_frame_caches()
        return res

    def _streams(self, dev, n):
        cache = self.__dict__.setdefault("_stream_cache", {})
        if (str(dev), n) not in cache:
            cache[(str(dev), n)] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        return cache[(str(dev), n)]

    # -- checkpoint contract (patchrefinerplus.py:212-216) ------------------------------------------
    def load_dict(self, sd):
        """patchrefinerplus.py:212-213 (``load_state_dict(strict=False)``) made checkpoint-ready: key variants of the two
        un-vendored packages (timm's flattened / unflattened feature-net names, a DDP ``module.`` prefix, BatchNorm bookkeeping
        buffers) are rewritten by ``weights.remap_state_dict``; whatever still does not match is REPORTED, grouped by module
        (``weights.diagnose_state_dict``) -- the reference prints the bare key lists -- and a forward with parameters still
        missing raises 'weights not loaded' instead of running on garbage.  The report is kept in ``self.last_load_report``."""
        import warnings
        from . import weights as W_
        spec = self.spec()
        sd, applied = W_.remap_state_dict(sd, spec)
        diag = W_.diagnose_state_dict(spec, sd)  # BEFORE loading: load_state_dict raises on the first shape mismatch
        if diag["shape_mismatch"]:
            raise RuntimeError("load_dict: checkpoint tensors whose shape differs from the model's -- " + W_.format_diagnosis(diag))
        res = self.load_state_dict(sd, strict=False)
        self.last_load_report = dict(diag, remapped=applied)
        if applied:
            warnings.warn("load_dict: renamed checkpoint keys -- " + ", ".join(f"{k}: {v}" for k, v in applied.items()))
        if res["missing_keys"] or res["unexpected_keys"]:
            warnings.warn("load_dict: " + W_.format_diagnosis(diag))
        return res

    def get_save_dict(self):
        return self.state_dict()

    # -- PyTorchModelHubMixin's on-disk layout (patchrefinerplus.py:39,60): <dir>/config.json + <dir>/model.safetensors.
    #    Local directories only: there is no hub access from this build.
    def save_pretrained(self, save_directory):
        import json
        import os
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.config.to_dict(), f, indent=2, sort_keys=True)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.get_save_dict().items()},
                  os.path.join(save_directory, "model.safetensors"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **config_overrides):
        """``Model.from_pretrained(dir)``: config.json -> ``cls(config)``, model.safetensors -> ``load_dict`` (the mixin
        builds the class with ``config=`` and loads non-strictly, like the reference's own load_dict)."""
        import json
        import os
        from safetensors.torch import load_file
        d = str(pretrained_model_name_or_path)
        if not os.path.isdir(d):
            raise FileNotFoundError(f"from_pretrained: '{d}' is not a local directory (no hub access in this build)")
        with open(os.path.join(d, "config.json")) as f:
            cfg = json.load(f)
        cfg.update(config_overrides)
        m = cls(cfg)
        m.load_dict(load_file(os.path.join(d, "model.safetensors")))
        return m

    def _make_da2(self, branch_cfg, max_depth):
        mc = dict(branch_cfg["model_cfg"])
        m = DepthAnythingV2(**{**mc, "max_depth": max_depth}, device=self.device, prec=self.prec)
        # load_state_dict(torch.load(branch['pretrained'])) -- a bare state dict, strict (patchrefiner.py:94,118)
        self._load_ckpt(m, branch_cfg.get("pretrained"), None, True, "DepthAnythingV2 weights")
        return m

    @staticmethod
    def _load_ckpt(module, path, key, strict, what, keep=None):
        """The constructor-time checkpoint loads of the reference (patchrefiner.py:82-148, patchrefinerplus.py:104-206):
        ``torch.load(path, map_location='cpu')[key]`` into ``module``.  A path that does not exist is skipped with a
        warning (the reference would die in torch.load): the weights then have to arrive through load_state_dict /
        load_dict, and a forward without them raises 'weights not loaded'."""
        if path is None:
            return None
        import os
        import warnings
        if not os.path.exists(str(path)):
            warnings.warn(f"{what}: checkpoint '{path}' does not exist -- skipped; load the weights with load_dict()")
            return None
        sd = torch.load(str(path), map_location="cpu")
        if key is not None:
            sd = sd[key]
        if keep is not None:
            sd = {k: v for k, v in sd.items() if keep(k)}
        widen = getattr(getattr(module, "refiner_fine_branch", None), "coarse_condition", True)  # (:144: only with coarse_condition)
        for k in list(sd):
            # the reference loads a 3-channel timm stem first and then widens it to 4 input channels with a zero
            # plane (stem surgery, patchrefinerplus.py:144-200): same result, done on the checkpoint
            if widen and k.endswith(("refiner_encoder.conv_stem.weight", "refiner_encoder.stem_0.weight")) and sd[k].shape[1] == 3:
                w = torch.zeros((sd[k].shape[0], 4) + tuple(sd[k].shape[2:]), dtype=sd[k].dtype)
                w[:, :3] = sd[k]
                sd[k] = w
        return module.load_state_dict(sd, strict=strict)

    def _common_init(self, config):
        config = _cfg(config)
        self.config = config
        self.min_depth, self.max_depth = config.min_depth, config.max_depth
        self.patch_process_shape = tuple(config.patch_process_shape)
        self.tile_cfg = self.prepare_tile_cfg(config.image_raw_shape, config.patch_split_num)
        self.prec = ops.L.PREC_NAMES[config.get("prec", "f32")]
        self.arith = config.get("prec", "f32")  # the model-level name ("f16f6": bf16x3 + the fp16 / fp6 layers; handed to the fusion model as is)
        self.device = torch.device(config.get("device", "cuda"))
        self.max_batch = config.get("max_batch", None)
        self.n_streams = config.get("n_streams", 1)
        self.hip_graph = bool(config.get("hip_graph", False))  # capture + replay the device side of a frame (_graph_frame)
        self.shard_dst_cost = float(config.get("shard_dst_cost", self.SHARD_DST_COST))  # patch-sharded mode (shard_layout)
        self.shard_owner_cost = float(config.get("shard_owner_cost", self.SHARD_OWNER_COST))
        self.shard_rotate_coarse = bool(config.get("shard_rotate_coarse", True))  # one rotating rank computes the next frame's coarse pyramid
        self.strategy_refiner_target = config.strategy_refiner_target
        self.fusion_feat_level = config.fusion_feat_level
        ctype = config.coarse_branch["type"]
        pcm = config.get("pretrain_coarse_model", None)
        if ctype == "DA2":
            self.coarse_branch = self._make_da2(config.coarse_branch, config.max_depth)
            self.resizer = Resizer(self.patch_process_shape[1], self.patch_process_shape[0], "da")
            self._load_ckpt(self.coarse_branch, pcm, "model_state_dict", True, "pretrain_coarse_model")
        elif ctype in ("DA-ZoeDepth", "ZoeDepth"):
            # ZoeDepth.build(**coarse_branch) (patchrefinerplus.py:102-116): midas_model_type picks the core -- MiDaS
            # DPT_BEiT_L_384 (the default) or a DepthAnything ViT; the config's ``type`` picks the resizer
            from .zoedepth import ZoeDepth
            zc = {k: v for k, v in config.coarse_branch.to_dict().items() if k != "type"}
            self.coarse_branch = ZoeDepth(device=self.device, prec=self.prec, **zc)
            if ctype == "ZoeDepth":
                # ResizeZoe.__call__ is hard-coded to 384 x 512 whatever it is constructed with (midas.py:171-174)
                if tuple(self.patch_process_shape) != (384, 512):
                    raise ValueError(f"coarse_branch type 'ZoeDepth' resizes every image to 384 x 512 (midas.py:171-174): "
                                     f"patch_process_shape {list(self.patch_process_shape)} cannot work (the reference fails too)")
                self.resizer = Resizer(512, 384, "zoe")
            else:
                self.resizer = Resizer(self.patch_process_shape[1], self.patch_process_shape[0], "da")
            # strict only for DA-ZoeDepth under PatchRefinerPlus (patchrefinerplus.py:108,116; patchrefiner.py:82,89)
            self._load_ckpt(self.coarse_branch, pcm, "model_state_dict", ctype == "DA-ZoeDepth" and self.STRICT_DA_ZOE,
                            "pretrain_coarse_model")
        else:
            raise NotImplementedError(f"coarse_branch type {ctype!r}")
        if self.strategy_refiner_target != "offset_coarse":
            raise NotImplementedError("strategy_refiner_target: every shipped config uses 'offset_coarse'")
        return config

    def coarse_forward(self, image_lr):
        """patchrefinerplus.py:218-237: one backbone forward; pyramid low -> high + metric depth."""
        out = self.coarse_branch(image_lr, return_final_centers=True)
        t = out["temp_features"]
        feats = [t["x_d0"], t["x_blocks_feat_0"], t["x_blocks_feat_1"], t["x_blocks_feat_2"], t["x_blocks_feat_3"],
                 t["midas_final_feat"]]
        return feats, out["metric_depth"]


@MODELS.register_module()
class BaselinePretrain(_PatchModel):
    """estimator/models/baseline_pretrain.py:44-93 (constructor: keyword arguments, not one ``config``), 377-464.
    target='coarse': ONE backbone forward on ``image_lr``, no tiling -- returns the DEVICE tensor [1,1,ph,pw] and
    dict(rgb, depth_pred, depth_gt) (:409-411,464).  target='fine': the tiling driver with the bare backbone on every
    tile (``infer_forward`` :144-146), blend mask border 0.1 (generatemask's default, :420,447) and -- unlike the two
    refiners -- N random_tile CALLS for r<N>, i.e. N * process_num random tiles (:449-452); returns (CPU depth, {})."""

    blend_border = 0.1
    needs_coarse = False

    def __init__(self, coarse_branch=None, fine_branch=None, sigloss=None, min_depth=1e-3, max_depth=80,
                 image_raw_shape=(2160, 3840), patch_process_shape=(384, 512), patch_split_num=(4, 4), target="coarse",
                 coarse_branch_zoe=None, device="cuda", prec="f32", max_batch=None, n_streams=1):
        super().__init__()
        self.min_depth, self.max_depth = min_depth, max_depth
        self.patch_process_shape = tuple(patch_process_shape)
        self.tile_cfg = self.prepare_tile_cfg(image_raw_shape, patch_split_num)
        self.prec = ops.L.PREC_NAMES[prec] if isinstance(prec, str) else prec
        self.arith = prec if isinstance(prec, str) else ops.L.PREC_LABEL[prec]
        self.device = torch.device(device)
        self.max_batch, self.n_streams = max_batch, n_streams
        self.target = target
        if target not in ("coarse", "fine"):
            raise NotImplementedError(f"BaselinePretrain target {target!r}")  # baseline_pretrain.py:406,461
        bcfg = _cfg(coarse_branch if target == "coarse" else fine_branch)
        btype = bcfg["type"]
        if btype == "DA2" and target == "coarse":
            branch = self._make_da2(bcfg, max_depth)
        elif btype in ("ZoeDepth", "DA-ZoeDepth"):
            from .zoedepth import ZoeDepth
            branch = ZoeDepth(device=self.device, prec=self.prec, **{k: v for k, v in bcfg.to_dict().items() if k != "type"})
        else:
            raise NotImplementedError(f"BaselinePretrain(target={target!r}) with branch type {btype!r} "
                                      "(baseline_pretrain.py:68-90 builds none either)")
        self.resizer = Resizer(self.patch_process_shape[1], self.patch_process_shape[0], "zoe" if btype == "ZoeDepth" else "da")
        name = "coarse_branch" if target == "coarse" else "fine_branch"
        setattr(self, name, branch)
        self._branch = branch
        self._children = {name: branch}
        self.crop_mean, self.crop_std = getattr(branch, "input_mean", IMAGENET_MEAN), getattr(branch, "input_std", IMAGENET_STD)

    def _pack(self):
        pass

    # baseline_pretrain.py:126-142: checkpoints hold the bare branch
    def load_dict(self, sd):
        self._invalidate_frame_caches()
        return self._branch.load_state_dict(sd, strict=False)

    def get_save_dict(self):
        return self._branch.state_dict()

    def random_calls(self, cai_mode, process_num):
        return int(cai_mode[1:])  # ``for i in range(patch_num)`` (baseline_pretrain.py:449-452)

    @torch.no_grad()
    def forward(self, mode=None, image_lr=None, image_hr=None, depth_gt=None, **kw):
        if mode == "train":
            raise NotImplementedError("only inference is built (training is out of scope, SURVEY.md 2 #12-13)")
        if self.target == "coarse":
            if not image_lr.is_cuda:
                raise RuntimeError("image_lr must be on the GPU (tester.py:43-49 moves it); no CPU path")
            with torch.cuda.device(image_lr.device):
                depth = self.coarse_branch(image_lr)["metric_depth"]
            return depth, dict(rgb=image_lr, depth_pred=depth, depth_gt=depth_gt)
        depth, _ = super().forward(mode="infer", image_lr=image_lr, image_hr=image_hr, depth_gt=depth_gt, **kw)
        return depth, {}

    __call__ = forward

    def infer_forward(self, crops: Feat, rois, depth_roi, out=None):
        """baseline_pretrain.py:144-146: ``self.fine_branch(imgs_crop)['metric_depth']``"""
        d = self.fine_branch.forward_nhwc(crops)["metric_depth"]
        if out is not None:
            out.copy_(d)
            return out
        return d


@MODELS.register_module()
class PatchRefiner(_PatchModel):
    """V1 (estimator/models/patchrefiner.py:54)."""

    def __init__(self, config):
        super().__init__()
        config = self._common_init(config)
        fb = config.refiner.fine_branch
        if fb["type"] == "DA2":
            self.refiner_fine_branch = self._make_da2(fb, config.max_depth)
        elif fb["type"] in ("ZoeDepth", "DA-ZoeDepth"):  # patchrefiner.py:104-115: ZoeDepth.build(**fine_branch)
            from .zoedepth import ZoeDepth
            self.refiner_fine_branch = ZoeDepth(device=self.device, prec=self.prec,
                                                **{k: v for k, v in fb.to_dict().items() if k != "type"})
            self.crop_mean, self.crop_std = self.refiner_fine_branch.input_mean, self.refiner_fine_branch.input_std
        else:
            raise NotImplementedError(f"refiner fine_branch type {fb['type']!r}")
        self._load_ckpt(self.refiner_fine_branch, config.get("pretrain_fine_model", None), "model_state_dict",
                        fb["type"] == "DA2", "pretrain_fine_model")
        self.refiner_fusion_model = build_model({**config.refiner.fusion_model.to_dict(), "device": self.device,
                                                 "prec": self.arith if self.arith == "f16f6" else self.prec})
        self._children = dict(coarse_branch=self.coarse_branch, refiner_fine_branch=self.refiner_fine_branch,
                              refiner_fusion_model=self.refiner_fusion_model)
        # config.pretrained: the refiner part only unless load_whole (patchrefiner.py:125-143)
        whole = bool(config.get("load_whole", False))
        self._load_ckpt(self, config.get("pretrained", None), "model_state_dict", False, "pretrained",
                        keep=None if whole else (lambda k: "coarse_branch" not in k))

    def _pack(self):
        pass

    def get_save_dict(self):  # patchrefiner.py:158-166: the coarse branch is not saved
        return {k: v for k, v in self.state_dict().items() if "coarse_branch." not in k}

    def infer_forward(self, crops: Feat, rois: List[Feat], depth_roi: Feat, out=None):
        """patchrefiner.py:258-283."""
        fine = self.refiner_fine_branch.forward_nhwc(crops)
        t = fine["temp_features"]
        r_feats = [t["x_d0"], t["x_blocks_feat_0"], t["x_blocks_feat_1"], t["x_blocks_feat_2"], t["x_blocks_feat_3"],
                   t["midas_final_feat"]]
        n = self.fusion_feat_level
        base = depth_roi.buf.view(depth_roi.n, 1, depth_roi.h, depth_roi.w)
        return self.refiner_fusion_model(c_feat=rois[-n:][::-1], f_feat=r_feats[-n:][::-1], pred1=base,
                                         pred2=fine["metric_depth"], update_base=base, out=out)


@MODELS.register_module()
class PatchRefinerSemi(StateDictModule):
    """estimator/models/patchrefiner_semi.py:46-210, the INFERENCE side: the configs the real-domain checkpoints ship with
    (patchrefinerv2_dav2/semi_kitti.py, *_cs_semi_*, pr_ssi_*) name this type; at test time ``forward`` hands everything to the student
    (:198-210).  The teacher, the pseudo-label / edge / distillation machinery and ``mode='train'`` are training and are
    not built: only ``model_cfg_student`` is instantiated.  State-dict names carry the ``student_model.`` prefix as in the reference."""

    def __init__(self, model_cfg_student, teacher_pretrain=None, model_cfg_teacher=None, **_training_only):
        super().__init__()
        self.student_model = build_model(model_cfg_student)
        self._children = dict(student_model=self.student_model)
        self.device, self.prec = self.student_model.device, self.student_model.prec

    def _pack(self):
        pass

    def __getattr__(self, name):  # resizer, tile_cfg, patch_process_shape, min / max_depth, predict_tiles ... are the student's (:163)
        if name in ("student_model", "_children", "_spec", "_sd"):
            raise AttributeError(name)
        return getattr(self.student_model, name)

    def load_dict(self, sd):
        """patchrefiner_semi.py:110-116: old checkpoints hold teacher + student (only the ``student_model.`` keys are loaded here: the
        teacher is not built), new ones only the student's own names (strict=False there too)"""
        if "student_model.coarse_branch.core.core.pretrained.model.cls_token" in sd:
            return self.load_state_dict({k: v for k, v in sd.items() if k.startswith("student_model.")}, strict=True)
        return self.student_model.load_state_dict(sd, strict=False)

    def get_save_dict(self):
        return self.student_model.get_save_dict()

    def forward(self, mode=None, image_lr=None, image_hr=None, depth_gt=None, cai_mode="m1", **kw):
        if mode == "train":
            raise NotImplementedError("PatchRefinerSemi: only inference is built (training is out of scope, SURVEY.md 2 #12-14)")
        # (:208-210 forwards cai_mode but neither tile_cfg nor process_num: the student runs with its configured tiling -- unless the
        #  caller passes them, which the reference's wrapper would have dropped)
        extra = {k: v for k, v in kw.items() if k in ("tile_cfg", "process_num", "next_image_lr", "return_device") and v is not None}
        return self.student_model(mode=mode, image_lr=image_lr, image_hr=image_hr, depth_gt=depth_gt, cai_mode=cai_mode, **extra)

    __call__ = forward


@MODELS.register_module()
class PatchRefinerPlus(_PatchModel):
    """V2 (estimator/models/patchrefinerplus.py:60)."""

    crop_channels = 4
    STRICT_DA_ZOE = True

    def __init__(self, config):
        super().__init__()
        config = self._common_init(config)
        if config.get("pretrain_stage", False):
            raise NotImplementedError("pretrain_stage=True is a training configuration")
        self.refiner_fine_branch = build_model({**config.refiner.fine_branch.to_dict(), "device": self.device,
                                                "prec": self.prec})
        self.refiner_fusion_model = build_model({**config.refiner.fusion_model.to_dict(), "device": self.device,
                                                 "prec": self.arith if self.arith == "f16f6" else self.prec})
        self.crop_mean, self.crop_std = self.refiner_fine_branch.mean, self.refiner_fine_branch.std
        self._children = dict(coarse_branch=self.coarse_branch, refiner_fine_branch=self.refiner_fine_branch,
                              refiner_fusion_model=self.refiner_fusion_model)
        # patchrefinerplus.py:132-135 (before the stem surgery: a 3-channel conv_stem checkpoint) and :202-205
        self._load_ckpt(self, config.get("pretrained", None), "model_state_dict", False, "pretrained")
        self._load_ckpt(self, config.get("whole_pretrained", None), "model_state_dict", False, "whole_pretrained")

    def _pack(self):
        pass

    def infer_forward(self, crops: Feat, rois: List[Feat], depth_roi: Feat, out=None):
        """patchrefinerplus.py:330-365: encoder on [norm(rgb), coarse depth roi] then BiDirectionalFusion."""
        ops.upsample_bilinear(depth_roi, crops.h, crops.w, out=crops.slice(3, 1))  # 4th input channel (exact copy)
        f_feat, f_sizes = self.refiner_fine_branch(crops)
        n = self.fusion_feat_level
        base = depth_roi.buf.view(depth_roi.n, 1, depth_roi.h, depth_roi.w)
        zeros = torch.zeros_like(base)  # out_depth = zeros (lightweight_refiner.py:320); replaced by c2f's output
        return self.refiner_fusion_model(c_feat=rois[-n:][::-1], f_feat=f_feat, pred1=base, pred2=zeros,
                                         update_base=base, f_sizes=f_sizes, out=out)
