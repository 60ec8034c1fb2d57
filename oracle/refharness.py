"""Import the reference Python (zhyever/PatchRefinerV2) on CPU under sys.modules stubs.

BUILD-CONTAINER ONLY: needs /root/reference, which does not exist on the GPU box.  Used by
``oracle/make_golden.py`` to pin the oracle and to emit ``tests/golden`` vectors.  Nothing
here is imported by tests, bench or the product.

Stubbed third-party modules (absent from this image): cv2, torchvision, mmengine, timm,
kornia, wandb ...; ``torchvision.ops.roi_align`` and ``cv2.GaussianBlur`` are routed to the
oracle's restatements (they are un-vendored arithmetic -> parity unpinned).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF = "/root/reference"


class _AnyAttr(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return 0


class AttrDict(dict):
    """Minimal stand-in for mmengine ConfigDict / transformers PretrainedConfig."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, AttrDict) else v) for k, v in self.items()}


def install():
    """Idempotent.  Must run with cwd=/root/reference for the DA-v1 torch.hub relative path."""
    if getattr(install, "_done", False):
        return
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    import torch
    import transformers  # noqa: F401  (before the fake torchvision: its find_spec probe)
    import huggingface_hub  # noqa: F401
    from transformers import PretrainedConfig

    PretrainedConfig.from_dict = classmethod(lambda cls, d, **k: AttrDict(d))

    from oracle import ops as oops

    for p in (REF, os.path.join(REF, "external")):
        if p not in sys.path:
            sys.path.insert(0, p)

    cv2 = _AnyAttr("cv2")
    cv2.GaussianBlur = lambda img, ksize, sigma: oops.gaussian_blur(img, int(ksize[0]), sigma)
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvo = types.ModuleType("torchvision.ops")

    class Normalize:
        def __init__(self, mean, std):
            self.mean = torch.tensor(mean).view(-1, 1, 1)
            self.std = torch.tensor(std).view(-1, 1, 1)

        def __call__(self, x):
            return (x - self.mean) / self.std

    tvt.Normalize = Normalize
    tvt.Compose = lambda fns: fns
    tvt.ToTensor = lambda: None
    tvo.roi_align = lambda feat, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False: \
        oops.roi_align(feat, rois, output_size, spatial_scale, aligned=aligned)
    tv.transforms, tv.ops = tvt, tvo
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.ops": tvo})

    mm = types.ModuleType("mmengine")
    mm.print_log = lambda *a, **k: None
    mmc = types.ModuleType("mmengine.config")
    mmc.ConfigDict = AttrDict
    mm.config = mmc
    sys.modules.update({"mmengine": mm, "mmengine.config": mmc})

    timm = _AnyAttr("timm")
    timml = _AnyAttr("timm.layers")
    timm.layers = timml
    sys.modules.update({"timm": timm, "timm.layers": timml})

    # bare packages (skip estimator/__init__.py which pulls datasets/trainer/wandb)
    registry = {}

    class _Registry:
        def register_module(self, *a, **k):
            def deco(cls):
                registry[cls.__name__] = cls
                return cls
            return deco

        def build(self, cfg):
            cfg = dict(cfg)
            return registry[cfg.pop("type")](**cfg)

    MODELS = _Registry()
    for name, path in (("estimator", "estimator"), ("estimator.models", "estimator/models"),
                       ("estimator.models.blocks", "estimator/models/blocks")):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = m
    reg = types.ModuleType("estimator.registry")
    reg.MODELS = MODELS
    sys.modules["estimator.registry"] = reg
    sys.modules["estimator"].registry = reg
    sys.modules["estimator.models"].build_model = MODELS.build
    blocks = sys.modules["estimator.models.blocks"]
    pe = importlib.import_module("estimator.models.blocks.position_embedding")
    blocks.PositionEmbeddingRandom = pe.PositionEmbeddingRandom

    import torch.nn as nn

    @MODELS.register_module()
    class SILogLoss(nn.Module):
        def forward(self, *a, **k):
            return 0.0

    @MODELS.register_module()
    class GradMatchLoss(nn.Module):
        def forward(self, *a, **k):
            return 0.0

    torch.cuda.empty_cache = lambda: None
    install.registry = registry
    install.MODELS = MODELS
    install._done = True


def ref_module(name: str):
    install()
    return importlib.import_module(name)
