"""The reference's operator API: `{target: "dotted.path", params: {...}}` dictionaries resolved by
`instantiate_from_config` (sgm/util.py:168-185), plus a PyYAML loader standing in for OmegaConf
(not installed in this image): YAML anchors/aliases are resolved by PyYAML itself, and the returned
mapping supports both `cfg["model"]` and `cfg.model` / `cfg.get(...)` like the OmegaConf objects the
reference's scripts use (scripts/sampling/util.py:38-42)."""
from __future__ import annotations

import importlib
from typing import Any


class Config(dict):
    """dict with attribute access, recursively applied."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o: Any) -> Any:
    if isinstance(o, dict):
        return Config({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def load_config(path: str) -> Config:
    import yaml
    with open(path) as f:
        return _wrap(yaml.safe_load(f))


def get_obj_from_str(string: str, reload: bool = False, invalidate_cache: bool = True):
    """sgm/util.py:178-185"""
    module, cls = string.rsplit(".", 1)
    if invalidate_cache:
        importlib.invalidate_caches()
    m = importlib.import_module(module)
    if reload:
        importlib.reload(m)
    return getattr(m, cls)


def instantiate_from_config(config):
    """sgm/util.py:168-175 — same error behaviour (KeyError without `target`, None for the two sentinels)."""
    if "target" not in config:
        if config == "__is_first_stage__":
            return None
        elif config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))
