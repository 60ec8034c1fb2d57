"""MMEngine-style registry + config objects (mmengine is not installed on the GPU boxes).

Mirrors estimator/registry/registry.py:7-8 and estimator/models/builder.py:6-8: classes are
registered under their reference ``type`` string and built from ``dict(type=..., **kwargs)``.
If mmengine is importable its Config/ConfigDict objects are accepted as-is (duck typed).
"""
from __future__ import annotations

import importlib.util
import os
from typing import Any, Dict


class ConfigDict(dict):
    """dict with attribute access (the subset of mmengine.config.ConfigDict the hot path uses)."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = _wrap(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = _wrap(v)

    def to_dict(self):
        return {k: _unwrap(v) for k, v in self.items()}


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigDict):
        return ConfigDict(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _unwrap(v):
    if isinstance(v, ConfigDict):
        return v.to_dict()
    if isinstance(v, (list, tuple)):
        return type(v)(_unwrap(x) for x in v)
    return v


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules: Dict[str, Any] = {}

    def register_module(self, name: str = None, module=None, force: bool = False):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        return deco(module) if module is not None else deco

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, **default_args):
        cfg = dict(cfg.to_dict() if hasattr(cfg, "to_dict") else cfg)
        if "type" not in cfg:
            raise KeyError(f"`cfg` must contain the key 'type', got {sorted(cfg)}")
        t = cfg.pop("type")
        cls = self._modules.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f"{t} is not in the {self.name} registry")
        cfg.update(default_args)
        return cls(**cfg)

    def __contains__(self, key):
        return key in self._modules


MODELS = Registry("models")
DATASETS = Registry("datasets")


def build_model(cfg):
    return MODELS.build(cfg)


class Config(ConfigDict):
    """Python-file configs with ``_base_`` inheritance and dotted overrides
    (the behaviour of mmengine.Config the reference configs rely on; README.md:65)."""

    @staticmethod
    def fromfile(path: str) -> "Config":
        return Config(_load_py(os.path.abspath(path)))

    def merge_from_dict(self, options: Dict[str, Any]):
        for k, v in options.items():
            node = self
            keys = k.split(".")
            for kk in keys[:-1]:
                if kk not in node:
                    node[kk] = ConfigDict()
                node = node[kk]
            node[keys[-1]] = _wrap(v)


def _merge(base: dict, new: dict) -> dict:
    out = dict(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get("_delete_", False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != "_delete_"} if isinstance(v, dict) else v
    return out


def _load_py(path: str) -> dict:
    spec = importlib.util.spec_from_file_location("_prv2_cfg_" + str(abs(hash(path))), path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cur = {k: v for k, v in vars(mod).items() if not k.startswith("__") and not callable(v) and
           not isinstance(v, type(os))}
    bases = cur.pop("_base_", [])
    if isinstance(bases, str):
        bases = [bases]
    merged: dict = {}
    for b in bases:
        merged = _merge(merged, _load_py(os.path.normpath(os.path.join(os.path.dirname(path), b))))
    return _merge(merged, cur)
