"""Inference driver + general image dataset (the callers on either side of the hot path).

Tester.run          estimator/tester/tester.py:52-127 (frame loop, model call contract, uint16 PNG x256)
ImageDataset        estimator/datasets/general_dataset.py:161-234 (folder of images -> image_hr / image_lr)
read_image          estimator/datasets/general_dataset.py:22-62 (RGB/255 -> bicubic, align_corners=True)
Metrics, colour maps and edge maps of the reference's Tester are out of scope:
only the 16-bit depth PNG (the on-disk format downstream tools read) is written, with a dependency-free encoder.
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


def read_image(path, dataset_name="", image_resolution=(2160, 3840)) -> np.ndarray:
    """general_dataset.py:22-62, generic branch: decode, RGB/255, bicubic(align_corners=True) to the raw shape."""
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
        if gt_dir is not None:
            raise NotImplementedError("ground-truth metrics are out of scope (SURVEY.md 8f rank 4)")
        self.rgb_image_dir = rgb_image_dir
        self.files = sorted(os.listdir(rgb_image_dir))
        self.dataset_name = dataset_name
        self.image_resolution = tuple(image_resolution)
        self.network_process_size = tuple(network_process_size)
        self.resize_mode = resize_mode

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        name = self.files[i]
        img = read_image(os.path.join(self.rgb_image_dir, name), self.dataset_name, self.image_resolution)
        # image_lr is produced on the device by model.resizer (same bilinear align_corners arithmetic)
        return dict(image_hr=torch.from_numpy(img).permute(2, 0, 1).float(), img_file_basename=os.path.splitext(name)[0])


class RunnerInfo:
    def __init__(self, **kw):
        self.rank, self.save, self.gray_scale, self.work_dir = 0, False, False, "."
        self.__dict__.update(kw)


class Tester:
    """``Tester(config, runner_info, dataloader, model).run(cai_mode, process_num, image_raw_shape, patch_split_num)``"""

    def __init__(self, config, runner_info, dataloader, model):
        self.config, self.runner_info, self.dataloader, self.model = config, runner_info, dataloader, model

    @torch.no_grad()
    def run(self, cai_mode="m1", process_num=4, image_raw_shape=(2160, 3840), patch_split_num=(4, 4), seed=None):
        import random
        results = []
        rank, world = self.runner_info.rank, getattr(self.runner_info, "world_size", 1)
        for idx in range(rank, len(self.dataloader), world):  # frame-sharded data parallelism (tester.py:58)
            item = self.dataloader[idx]
            hr = item["image_hr"].unsqueeze(0).cuda()
            lr = self.model.resizer(hr)
            if seed is not None:
                random.seed(seed)
            tile_cfg = dict(image_raw_shape=list(image_raw_shape), patch_split_num=list(patch_split_num))
            result, log = self.model(mode="infer", cai_mode=cai_mode, process_num=process_num, tile_cfg=tile_cfg,
                                     image_lr=lr, image_hr=hr)
            if self.runner_info.save:
                os.makedirs(self.runner_info.work_dir, exist_ok=True)
                base = os.path.join(self.runner_info.work_dir, item["img_file_basename"])
                # raw depth as 16-bit PNG, multiplier 256 (tester.py:89-91)
                write_png16(base + "_uint16.png", (result.squeeze().numpy() * 256).astype("uint16"))
                coarse = F.interpolate(log["coarse_prediction"].cpu(), tuple(image_raw_shape), mode="bilinear")
                write_png16(base + "_coarse_uint16.png", (coarse.squeeze().numpy() * 256).astype("uint16"))
            results.append((item["img_file_basename"], tuple(result.shape), float(result.mean())))
        return results
