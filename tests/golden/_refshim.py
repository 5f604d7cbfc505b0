"""Import shim for the *reference* (RuoyuFeng/CCEdit at /root/reference) — golden generation only.

This file is used ONLY by tests/golden/make_golden.py, in the authoring container, to run the
reference's own Python hot-path modules on CPU/fp32 and record golden vectors.  Nothing in the
product (`ccedit_amd/`), in `bench.py`, in `__graft_entry__.py` or in the `-m gpu` tests imports
it: /root/reference does not exist on the GPU box.

`import sgm` fails as shipped (sgm/__init__.py pulls training data loaders, Lightning, the
un-vendored `src.controlnet11` annotators, `taming`).  The recipe: register empty
namespace shells for the packages so their `__init__` is skipped, and stub four third-party
modules that the hot-path files import at module top but never use at inference.
"""
from __future__ import annotations

import importlib
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _shell(name: str, path: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def install() -> None:
    if "sgm" in sys.modules and getattr(sys.modules["sgm"], "_ccedit_shim", False):
        return
    # --- third-party stubs -------------------------------------------------------------
    ds = types.ModuleType("deepspeed")                      # diffusionmodules/util.py:22
    sys.modules["deepspeed"] = ds

    lora = types.ModuleType("loralib")                      # attention.py:11
    lora.Linear = torch.nn.Linear
    sys.modules["loralib"] = lora

    oc = types.ModuleType("omegaconf")                      # openaimodel.py:1075, sampling.py:9

    class ListConfig(list):
        pass

    class OmegaConf(dict):
        pass

    oc.ListConfig = ListConfig
    oc.OmegaConf = OmegaConf
    lc = types.ModuleType("omegaconf.listconfig")
    lc.ListConfig = ListConfig
    oc.listconfig = lc
    sys.modules["omegaconf"] = oc
    sys.modules["omegaconf.listconfig"] = lc

    pl = types.ModuleType("pytorch_lightning")              # autoencoder.py:7

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl

    # --- namespace shells (skip the packages' own __init__) ------------------------------
    sgm = _shell("sgm", f"{REF_ROOT}/sgm")
    sgm._ccedit_shim = True
    _shell("sgm.modules", f"{REF_ROOT}/sgm/modules")
    _shell("sgm.modules.diffusionmodules", f"{REF_ROOT}/sgm/modules/diffusionmodules")
    _shell("sgm.modules.distributions", f"{REF_ROOT}/sgm/modules/distributions")
    _shell("sgm.modules.autoencoding", f"{REF_ROOT}/sgm/modules/autoencoding")
    _shell("sgm.models", f"{REF_ROOT}/sgm/models")
    sgm.util = importlib.import_module("sgm.util")


def ref(modname: str):
    """Import a reference module, e.g. ref('sgm.modules.diffusionmodules.controlmodel')."""
    install()
    return importlib.import_module(modname)


def ref_class_from_file(relpath: str, classname: str):
    """Evaluate ONE class definition of a reference file whose module cannot be imported (its top-level imports
    are unavailable here), e.g. Img2ImgDiscretizationWrapper in scripts/demo/streamlit_helpers.py (imports
    streamlit).  The class is compiled from the reference file in place, at golden-generation time only."""
    import ast
    src = open(f"{REF_ROOT}/{relpath}").read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == classname)
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[node], type_ignores=[]), f"{REF_ROOT}/{relpath}", "exec"), ns)
    return ns[classname]


def ref_func_from_file(relpath: str, funcname: str, extra_ns=None):
    """Same as ref_class_from_file for ONE top-level function (e.g. convert_load_lora of scripts/sampling/util.py,
    whose module imports omegaconf / decord / cv2 / imageio / torchvision at the top)."""
    import ast
    src = open(f"{REF_ROOT}/{relpath}").read()
    tree = ast.parse(src)
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == funcname)
    ns = {"torch": torch}
    ns.update(extra_ns or {})
    exec(compile(ast.Module(body=[node], type_ignores=[]), f"{REF_ROOT}/{relpath}", "exec"), ns)
    return ns[funcname]
