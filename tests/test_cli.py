"""GPU: tools/test.py end to end (reference CLI flags) on a synthetic image folder."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_writes_depth_pngs_and_metrics(tmp_path):
    (tmp_path / "imgs").mkdir()
    (tmp_path / "gts").mkdir()
    np.save(str(tmp_path / "imgs" / "frame0.npy"), np.random.RandomState(1).rand(90, 160, 3).astype(np.float32))
    np.save(str(tmp_path / "gts" / "frame0.npy"), (np.random.RandomState(2).rand(256, 512) * 20 + 0.5).astype(np.float32))
    cfg = tmp_path / "cfg.py"
    cfg.write_text(f"_base_ = ['{os.path.join(ROOT, 'configs', 'v2_dav2_mobile_u4k.py')}']\n"
                   "model = dict(config=dict(patch_process_shape=[112, 224], image_raw_shape=[256, 512], patch_split_num=[2, 2],\n"
                   "    coarse_branch=dict(model_cfg=dict(encoder='vits', features=256, out_channels=[48, 96, 192, 384]))))\n")
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "test.py"), str(cfg), "--synthetic-weights", "--cai-mode", "r4",
                        "--cfg-option", f"general_dataloader.dataset.rgb_image_dir={tmp_path / 'imgs'}",
                        f"general_dataloader.dataset.gt_dir={tmp_path / 'gts'}", "--save", "--work-dir",
                        str(out), "--test-type", "general", "--image-raw-shape", "256", "512", "--patch-split-num", "2", "2"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "frame0: depth (1, 1, 256, 512)" in r.stdout
    from PIL import Image
    d = np.asarray(Image.open(str(out / "frame0_uint16.png")))
    assert d.dtype == np.uint16 and d.shape == (256, 512) and d.max() > 0
    for name in ("frame0.png", "frame0_coarse.png"):   # colour maps (Spectral, 0..100 percentile)
        c = np.asarray(Image.open(str(out / name)))
        assert c.dtype == np.uint8 and c.shape == (256, 512, 3) and c.std() > 0
    e = np.asarray(Image.open(str(out / "frame0_edge.png")))   # Canny of the log depth, dilated (tester.py:99-106)
    assert e.dtype == np.uint8 and e.shape == (256, 512) and set(np.unique(e)) <= {0, 255}
    assert "abs_rel" in r.stdout and "see" in r.stdout


def test_cli_benchmark_protocol(tmp_path):
    """Tester.benchmark: the reference's fps protocol (tester.py:325-406) on a tiny synthetic folder"""
    (tmp_path / "imgs").mkdir()
    for i in range(2):
        np.save(str(tmp_path / "imgs" / f"f{i}.npy"), np.random.RandomState(i).rand(64, 96, 3).astype(np.float32))
    cfg = tmp_path / "cfg.py"
    cfg.write_text(f"_base_ = ['{os.path.join(ROOT, 'configs', 'v2_dav2_mobile_u4k.py')}']\n"
                   "model = dict(config=dict(patch_process_shape=[112, 224], image_raw_shape=[256, 512], patch_split_num=[2, 2],\n"
                   "    coarse_branch=dict(model_cfg=dict(encoder='vits', features=256, out_channels=[48, 96, 192, 384]))))\n")
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "test.py"), str(cfg), "--synthetic-weights", "--cai-mode", "r4",
                        "--cfg-option", f"general_dataloader.dataset.rgb_image_dir={tmp_path / 'imgs'}", "--work-dir", str(out),
                        "--image-raw-shape", "256", "512", "--patch-split-num", "2", "2", "--benchmark", "--repeat-times", "2",
                        "--benchmark-iters", "2", "6"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Average fps of 2 evaluations" in r.stdout and "Model Flops" in r.stdout
    txt = (out / "benchmark.txt").read_text()
    assert "Model Parameters" in txt and "conv3x3_halo" in txt
