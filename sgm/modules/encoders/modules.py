"""Conditioner contract (reference: sgm/modules/encoders/modules.py:84-204).

The embedders themselves (HF CLIP ViT-L/14 text encoder, MiDaS / ZoeDepth annotators from the un-vendored
ControlNet-v1-1 repo) run once per clip OUTSIDE the denoising loop and need weights that are not
available offline; they are out of scope for this build.  What the hot
path needs is the dict contract: batch keys `txt` / `control_hint` / `cond_img` -> conditioning keys
`crossattn` (B,77,768) / `control_hint` (B,3,T,H,W) / `cond_feat`.  `GeneralConditioner` keeps that
routing and accepts *precomputed* tensors: batch["crossattn"] may carry the text embedding directly.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn


class _Unavailable(nn.Module):
    def __init__(self, target: str, input_key: str):
        super().__init__()
        self.target, self.input_key = target, input_key

    def forward(self, *a, **k):
        raise NotImplementedError(
            f"embedder {self.target} is outside the hot path of this build and its weights are not available offline; "
            f"pass a precomputed tensor for its output key instead (see GeneralConditioner docstring)")


class GeneralConditioner(nn.Module):
    OUTPUT_KEY = {"txt": "crossattn", "control_hint": "control_hint", "cond_img": "cond_feat"}

    def __init__(self, emb_models: Optional[List[dict]] = None):
        super().__init__()
        self.embedders = nn.ModuleList([_Unavailable(e.get("target", "?"), e.get("input_key", "?")) for e in (emb_models or [])])

    def forward(self, batch: Dict, force_zero_embeddings=None) -> Dict[str, torch.Tensor]:
        out = {}
        for e in self.embedders:
            okey = self.OUTPUT_KEY.get(e.input_key, e.input_key)
            if okey in batch and torch.is_tensor(batch[okey]):
                out[okey] = batch[okey]
            elif e.input_key in batch and torch.is_tensor(batch[e.input_key]):
                out[okey] = batch[e.input_key]
            else:
                e()
        return out

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None) -> Tuple[Dict, Dict]:
        c = self(batch_c)
        uc = self(batch_c if batch_uc is None else batch_uc)
        return c, uc
