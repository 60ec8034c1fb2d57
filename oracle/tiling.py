"""Oracle (test infrastructure): tiling, ROI pyramid, blend and the two frame drivers.

prepare_tile_cfg        estimator/models/baseline_pretrain.py:96-124
random_tile             baseline_pretrain.py:149-231
regular_tile            baseline_pretrain.py:235-375
RunningAverageMap       estimator/models/utils.py:22-49
coarse_postprocess_test estimator/models/patchrefinerplus.py:263-296 / patchrefiner.py:199-217
PatchRefinerPlus infer  patchrefinerplus.py:330-365, 470-530
PatchRefiner infer      patchrefiner.py:219-283, 341-401
"""
from __future__ import annotations

import random

import torch

from . import convnext, dav2, effnet, fusion, mnv4
from .ops import bilinear_ac, generatemask, nearest, resize_da, resize_zoe, roi_align


def prepare_tile_cfg(patch_process_shape, image_raw_shape, patch_split_num):
    ph, pw = patch_process_shape
    sh, sw = patch_split_num
    raw = (image_raw_shape[0] // sh, image_raw_shape[1] // sw)
    return dict(
        patch_split_num=list(patch_split_num),
        patch_reensemble_shape=(ph * sh, pw * sw),
        patch_raw_shape=raw,
        image_raw_shape=list(image_raw_shape),
        raw_h_split_point=[int(raw[0] * i) for i in range(sh)],
        raw_w_split_point=[int(raw[1] * i) for i in range(sw)])


class RunningAverageMap:
    def __init__(self, average_map, count_map):
        self.count_map = count_map
        self.average_map_init = average_map
        self.average_map = average_map
        self.update_flag = False

    def update(self, pred_map, ct_map):
        self.update_flag = True
        mask = ct_map > 0
        self.average_map[mask] = (pred_map[mask] * ct_map[mask] + self.count_map[mask] * self.average_map[mask]) / \
            (self.count_map[mask] + ct_map[mask])
        self.count_map[mask] = self.count_map[mask] + ct_map[mask]

    def resize(self, resolution):
        self.average_map = nearest(self.average_map[None, None], resolution).squeeze()
        self.count_map = bilinear_ac(self.count_map[None, None], resolution).squeeze()

    def get_avg_map(self):
        return self.average_map if self.update_flag else self.average_map_init


def bboxs_to_feat(bboxs, image_raw_shape, patch_process_shape):
    """baseline_pretrain.py:289-296 (float32 tensor arithmetic, including the 1/x*y order)."""
    H, W = image_raw_shape
    ph, pw = patch_process_shape
    factor = torch.tensor([1 / W * pw, 1 / H * ph, 1 / W * pw, 1 / H * ph]).unsqueeze(0)
    bf = bboxs.int() * factor
    inds = torch.arange(bboxs.shape[0]).unsqueeze(-1)
    return torch.cat((inds, bf), dim=-1)


def coarse_postprocess_test(coarse_prediction, coarse_features, bboxs_feat, ph):
    """feat.repeat(K) + roi_align per level (patchrefinerplus.py:263-283)."""
    K = bboxs_feat.shape[0]
    rois = []
    for feat in coarse_features:
        h, w = feat.shape[-2:]
        rois.append(roi_align(feat.repeat(K, 1, 1, 1), bboxs_feat, (h, w), h / ph, aligned=True))
    h, w = coarse_prediction.shape[-2:]
    depth_roi = roi_align(coarse_prediction.repeat(K, 1, 1, 1), bboxs_feat, (h, w), h / ph, aligned=True)
    return dict(coarse_depth_roi=depth_roi, coarse_feats_roi=rois)


class OracleRefiner:
    """Shared driver; subclasses provide coarse_forward / infer_forward."""

    border = 0.15        # generatemask(border=0.15): patchrefinerplus.py:485,514; patchrefiner.py:358,386
    needs_coarse = True

    def random_calls(self, cai_mode, process_num):
        return int(cai_mode[1:]) // process_num   # patchrefinerplus.py:517-520

    def __init__(self, sd, patch_process_shape, image_raw_shape, patch_split_num, resizer="da"):
        self.sd = sd
        self.patch_process_shape = tuple(patch_process_shape)
        self.tile_cfg = prepare_tile_cfg(self.patch_process_shape, image_raw_shape, patch_split_num)
        self.resizer_kind = resizer
        self.trace = None  # optional dict for tests: records the tile plan

    def resizer(self, x):
        ph, pw = self.patch_process_shape
        return resize_da(x, pw, ph, 14) if self.resizer_kind == "da" else resize_zoe(x)

    # -- tiles ------------------------------------------------------------------
    def _crops(self, image_hr, h_starts, w_starts, height, width):
        crops, bboxs = [], []
        for hs in h_starts:
            for ws in w_starts:
                crop = image_hr[:, hs:hs + height, ws:ws + width]
                crops.append(self.resizer(crop.unsqueeze(0)).squeeze(0))
                bboxs.append(torch.tensor([ws, hs, ws + width, hs + height]))
        return torch.stack(crops), torch.stack(bboxs)

    def random_tile(self, image_hr, tile_temp, blur_mask, avg, tile_cfg, process_num):
        height, width = tile_cfg["patch_raw_shape"]
        H, W = tile_cfg["image_raw_shape"]
        h_starts = [random.randint(0, H - height - 1) for _ in range(process_num)]
        w_starts = [random.randint(0, W - width - 1)]
        crops, bboxs = self._crops(image_hr, h_starts, w_starts, height, width)
        if tile_temp is None:  # BaselinePretrain(target='fine'): the bare backbone (baseline_pretrain.py:204-206)
            pred = self.infer_forward(crops, None)
        else:
            bf = bboxs_to_feat(bboxs, (H, W), self.patch_process_shape)
            post = coarse_postprocess_test(tile_temp["coarse_prediction"], tile_temp["coarse_features"], bf,
                                           self.patch_process_shape[0])
            pred = self.infer_forward(crops, post)
        pred = nearest(pred, (height, width))
        if self.trace is not None:
            self.trace.setdefault("tiles", []).extend(("r", hs, w_starts[0]) for hs in h_starts)
        i = 0
        for hs in h_starts:
            for ws in w_starts:
                count = torch.zeros((H, W))
                pd = torch.zeros((H, W))
                count[hs:hs + height, ws:ws + width] = blur_mask
                pd[hs:hs + height, ws:ws + width] = pred[i]
                avg.update(pd, count)
                i += 1
        return avg

    def regular_tile(self, offset, offset_process, image_hr, init_flag, tile_temp, blur_mask, avg, tile_cfg,
                     process_num):
        height, width = tile_cfg["patch_raw_shape"]
        H, W = tile_cfg["image_raw_shape"]
        ph, pw = self.patch_process_shape
        RH, RW = tile_cfg["patch_reensemble_shape"]
        h_starts = [height * h + offset[0] for h in range((H - offset[0]) // height)]
        w_starts = [width * w + offset[1] for w in range((W - offset[1]) // width)]
        hp_starts = [ph * h + offset_process[0] for h in range((RH - offset_process[0]) // ph)]
        wp_starts = [pw * w + offset_process[1] for w in range((RW - offset_process[1]) // pw)]
        crops, bboxs = self._crops(image_hr, h_starts, w_starts, height, width)
        if tile_temp is not None:
            bf = bboxs_to_feat(bboxs, (H, W), self.patch_process_shape)
            post = coarse_postprocess_test(tile_temp["coarse_prediction"], tile_temp["coarse_features"], bf, ph)
        preds = []
        for idx, batch in enumerate(torch.split(crops, process_num, dim=0)):
            sl = slice(idx * process_num, (idx + 1) * process_num)
            sub = None if tile_temp is None else dict(coarse_depth_roi=post["coarse_depth_roi"][sl],
                                                      coarse_feats_roi=[f[sl] for f in post["coarse_feats_roi"]])
            preds.append(self.infer_forward(batch, sub))
        preds = torch.cat(preds, dim=0)
        if self.trace is not None:
            self.trace.setdefault("tiles", []).extend(("g", hs, ws) for hs in h_starts for ws in w_starts)
            self.trace.setdefault("preds", []).append(preds.clone())
        count = torch.zeros((RH, RW))
        pd = torch.zeros((RH, RW))
        i = 0
        for hs in hp_starts:
            for ws in wp_starts:
                if init_flag:
                    count[hs:hs + ph, ws:ws + pw] = blur_mask
                    pd[hs:hs + ph, ws:ws + pw] = preds[i]
                else:
                    count = torch.zeros((RH, RW))
                    pd = torch.zeros((RH, RW))
                    count[hs:hs + ph, ws:ws + pw] = blur_mask
                    pd[hs:hs + ph, ws:ws + pw] = preds[i]
                    avg.update(pd, count)
                i += 1
        if init_flag:
            avg = RunningAverageMap(pd, count)
        return avg

    # -- frame driver -------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, mode="infer", cai_mode="m1", process_num=4, tile_cfg=None, image_lr=None, image_hr=None,
                 **_ignored):
        assert mode == "infer"
        if tile_cfg is None:
            tile_cfg = self.tile_cfg
        else:
            tile_cfg = prepare_tile_cfg(self.patch_process_shape, tile_cfg["image_raw_shape"],
                                        tile_cfg["patch_split_num"])
        assert image_hr.shape[0] == 1
        if self.needs_coarse:
            feats, coarse_pred = self.coarse_forward(image_lr)
            tile_temp = dict(coarse_prediction=coarse_pred, coarse_features=feats)
        else:
            tile_temp = coarse_pred = None
        ph, pw = self.patch_process_shape
        rh, rw = tile_cfg["patch_raw_shape"]
        blur = torch.tensor(generatemask((ph, pw), border=self.border))
        kw = dict(image_hr=image_hr[0], tile_temp=tile_temp, tile_cfg=tile_cfg, process_num=process_num)
        avg = self.regular_tile([0, 0], [0, 0], init_flag=True, blur_mask=blur, avg=None, **kw)
        if cai_mode == "m2" or cai_mode[0] == "r":
            avg = self.regular_tile([0, rw // 2], [0, pw // 2], init_flag=False, blur_mask=blur, avg=avg, **kw)
            avg = self.regular_tile([rh // 2, 0], [ph // 2, 0], init_flag=False, blur_mask=blur, avg=avg, **kw)
            avg = self.regular_tile([rh // 2, rw // 2], [ph // 2, pw // 2], init_flag=False, blur_mask=blur,
                                    avg=avg, **kw)
        if cai_mode[0] == "r":
            blur = torch.tensor(generatemask((rh, rw), border=self.border) + 1e-3)
            avg.resize(tile_cfg["image_raw_shape"])
            for _ in range(self.random_calls(cai_mode, process_num)):
                avg = self.random_tile(blur_mask=blur, avg=avg, **kw)
        depth = avg.get_avg_map()[None, None]
        return depth, dict(rgb=image_lr, depth_pred=depth, depth_gt=None, coarse_prediction=coarse_pred)


def consistency_starts(n, size, overlap):
    """u4k_dataset.py:62-65 / eth_dataset.py:92-93 generalised: crop i of n is shifted by ((n - 1) - 2 i) * overlap / 2 towards
    the centre, so that neighbours share ``overlap`` pixels"""
    return [int(i * size + ((n - 1) - 2 * i) * overlap / 2) for i in range(n)]


def prenorm_bbox(ws, hs, w, h, image_raw_shape, patch_process_shape):
    """the dataset's pre_norm_bbox arithmetic (u4k_dataset.py:171-176): int64 tensor / int * int -> float32, which is NOT the
    infer path's ``bboxs.int() * (1 / W * pw)`` (baseline_pretrain.py:289-296) to the last bit"""
    H, W = image_raw_shape
    ph, pw = patch_process_shape
    b = torch.tensor([ws, hs, ws + w, hs + h])
    return torch.tensor([b[0] / W * pw, b[1] / H * ph, b[2] / W * pw, b[3] / H * ph])


@torch.no_grad()
def run_consistency(ora: "OracleRefiner", image_hr, overlap=270):
    """Tester.run_consistency (estimator/tester/tester.py:211-321) over the consistency crops of the U4K / ETH3D datasets:
    each crop through the model's ``mode='train'`` forward (coarse forward on the whole frame + the crop's ROI with the
    dataset's pre-normalised bbox + the per-patch network; patchrefiner.py:300-340), bilinear(align_corners) to the crop's raw
    size, mean |difference| over the strips shared with the left and the upper neighbour.
    Returns (consistency_error, crops [sh*sw, rh, rw])."""
    from .ops import bilinear_ac
    tc = ora.tile_cfg
    H, W = tc["image_raw_shape"]
    sh, sw = tc["patch_split_num"]
    rh, rw = tc["patch_raw_shape"]
    ph = ora.patch_process_shape[0]
    image = image_hr[0]
    feats, cp = ora.coarse_forward(ora.resizer(image_hr))
    crops = []
    for hs in consistency_starts(sh, rh, overlap):
        for ws in consistency_starts(sw, rw, overlap):
            crop = ora.resizer(image[:, hs:hs + rh, ws:ws + rw].unsqueeze(0))
            bf = torch.cat((torch.zeros(1, 1), prenorm_bbox(ws, hs, rw, rh, (H, W), ora.patch_process_shape)[None]), dim=-1)
            # coarse_postprocess_train (patchrefiner.py:187-197): batch 1, no repeat
            rois = [roi_align(f, bf, f.shape[-2:], f.shape[-2] / ph, aligned=True) for f in feats]
            droi = roi_align(cp, bf, cp.shape[-2:], cp.shape[-2] / ph, aligned=True)
            pred = ora.infer_forward(crop, dict(coarse_depth_roi=droi, coarse_feats_roi=rois))
            crops.append(bilinear_ac(pred, (rh, rw)).squeeze())
    errs = []
    for ii in range(sh):
        for jj in range(sw):
            cur = crops[ii * sw + jj]
            if jj > 0:
                errs.append((crops[ii * sw + jj - 1][:, -overlap:] - cur[:, :overlap]).abs().flatten())
            if ii > 0:
                errs.append((crops[(ii - 1) * sw + jj][-overlap:, :] - cur[:overlap, :]).abs().flatten())
    return float(torch.cat(errs).mean()), torch.stack(crops)


class OraclePatchRefiner(OracleRefiner):
    """V1: DA2 coarse + DA2 per-patch + FusionUnet (configs/patchrefiner_dav2/pr_u4k.py)."""

    def __init__(self, sd, coarse_cfg, fine_cfg, coarse_fn=None, fine_fn=None, **kw):
        """coarse_fn / fine_fn: a backbone other than DA2 (ZoeDepth: configs/patchrefiner_zoedepth/pr_u4k.py), image -> the
        backbone's output dict (metric_depth + temp_features)"""
        super().__init__(sd, **kw)
        self.coarse_cfg, self.fine_cfg = coarse_cfg, fine_cfg
        self.coarse_fn, self.fine_fn = coarse_fn, fine_fn

    def coarse_forward(self, image_lr):
        if self.coarse_fn is not None:
            return dav2.coarse_features(self.coarse_fn(image_lr))
        return dav2.coarse_features(dav2.dav2_forward(self.sd, "coarse_branch.", image_lr, self.coarse_cfg))

    def infer_forward(self, imgs_crop, post):
        r_feats, r_depth = dav2.coarse_features(
            self.fine_fn(imgs_crop) if self.fine_fn is not None else
            dav2.dav2_forward(self.sd, "refiner_fine_branch.", imgs_crop, self.fine_cfg))
        return fusion.fusion_unet(self.sd, "refiner_fusion_model.", post["coarse_feats_roi"][::-1], r_feats[::-1],
                                  post["coarse_depth_roi"], r_depth, update_base=post["coarse_depth_roi"])


class OraclePatchRefinerPlus(OracleRefiner):
    """V2: DA2 coarse + LightWeightRefiner(MNv4-S) + BiDirectionalFusion
    (configs/patchrefinerv2_dav2/plus_mobile_u4k_base_coarse_e2e_c2f_pretrain.py)."""

    def __init__(self, sd, coarse_cfg, coarse_fn=None, convnext_arch=None, effnet_arch=None, fusion_kw=None, coarse_condition=True, **kw):
        super().__init__(sd, **kw)
        self.coarse_condition = coarse_condition  # LightWeightRefiner(coarse_condition=...) (MNv4 encoder only)
        self.fusion_kw = fusion_kw or {}    # coarse2fine / coarse2fine_type of the ablation configs (fusion.bidirectional_fusion)
        self.coarse_cfg = coarse_cfg
        self.coarse_fn = coarse_fn
        self.convnext_arch = convnext_arch  # set: the v2_convx_u4k.py variant (ConvNeXt refiner encoder)
        self.effnet_arch = effnet_arch      # set: the v2_eff_u4k.py variant (EfficientNet refiner encoder)

    def coarse_forward(self, image_lr):
        if self.coarse_fn is not None:
            return self.coarse_fn(image_lr)
        return dav2.coarse_features(dav2.dav2_forward(self.sd, "coarse_branch.", image_lr, self.coarse_cfg))

    def infer_forward(self, imgs_crop, post):
        if self.effnet_arch is not None:
            r_feats, r_depth = effnet.lightweight_refiner_effnet(self.sd, "refiner_fine_branch.", imgs_crop,
                                                                 post["coarse_depth_roi"], self.effnet_arch)
        elif self.convnext_arch is not None:
            r_feats, r_depth = convnext.lightweight_refiner_convnext(self.sd, "refiner_fine_branch.", imgs_crop,
                                                                     post["coarse_depth_roi"], self.convnext_arch)
        else:
            r_feats, r_depth = mnv4.lightweight_refiner(self.sd, "refiner_fine_branch.", imgs_crop,
                                                        post["coarse_depth_roi"], self.coarse_condition)
        return fusion.bidirectional_fusion(self.sd, "refiner_fusion_model.", post["coarse_feats_roi"][::-1],
                                           r_feats[::-1], post["coarse_depth_roi"], r_depth,
                                           update_base=post["coarse_depth_roi"], **self.fusion_kw)


class OracleBaselinePretrain(OracleRefiner):
    """BaselinePretrain (estimator/models/baseline_pretrain.py:44-93, 377-464).  ``branch_fn(x)`` = the backbone on a
    [B,3,h,w] image in [0,1] -> metric depth [B,1,h,w].  target='coarse': ``branch_fn(image_lr)`` (:409-411);
    target='fine': tiling with the bare backbone per tile, mask border 0.1 (generatemask's default, :420,447), and N
    random_tile calls for r<N> (:449-452)."""

    border = 0.1
    needs_coarse = False

    def __init__(self, branch_fn, target="coarse", **kw):
        super().__init__(None, **kw)
        self.branch_fn, self.target = branch_fn, target

    def random_calls(self, cai_mode, process_num):
        return int(cai_mode[1:])

    def infer_forward(self, imgs_crop, post):
        return self.branch_fn(imgs_crop)

    @torch.no_grad()
    def __call__(self, mode="infer", image_lr=None, image_hr=None, **kw):
        if self.target == "coarse":
            d = self.branch_fn(image_lr)
            return d, dict(rgb=image_lr, depth_pred=d, depth_gt=None)
        depth, _ = super().__call__(mode="infer", image_lr=image_lr, image_hr=image_hr, **kw)
        return depth, {}
