"""Parameter containers with the reference's state-dict key names and tensor layouts.

The reference builds its networks from nn.Conv2d / nn.Conv1d / nn.Linear / nn.GroupNorm / nn.LayerNorm
inside nn.Sequential containers, and its checkpoints are keyed by those module paths
(`load_state_dict(sd, strict=False)` with e.g.
`model.diffusion_model.input_blocks.1.0.in_layers_temporal.2.weight (320,320,3)`).
These containers keep exactly those parameter names/shapes (fp32, reference layout) so that
`state_dict()` / `load_state_dict()` are drop-in, and add a `pack()` step that produces the bf16
kernel-layout copy (`.pw`) the HIP kernels consume.  They have no `forward`: arithmetic lives in the
composite modules, which call ccedit_amd.ops (HIP only).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .packing import PackedWeight, pack_weight


class Slot(nn.Module):
    """Parameter-free placeholder keeping nn.Sequential indices aligned with the reference
    (nn.SiLU / nn.Dropout / nn.Identity positions)."""

    def forward(self, *a, **k):   # pragma: no cover - never called
        raise RuntimeError("Slot is a placeholder")


class Conv(nn.Module):
    """nn.Conv2d (dims=2) / nn.Conv1d (dims=1) parameters: weight (O, I, k[, k]), bias (O,)."""

    def __init__(self, cin: int, cout: int, k: int, dims: int = 2, stride: int = 1, bias: bool = True):
        super().__init__()
        shape = (cout, cin, k, k) if dims == 2 else (cout, cin, k)
        self.weight = nn.Parameter(torch.zeros(shape), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False) if bias else None
        self.cin, self.cout, self.k, self.dims, self.stride = cin, cout, k, dims, stride
        self.pw: Optional[PackedWeight] = None

    def pack(self, device, scale: float = 1.0):
        w = self.weight if scale == 1.0 else self.weight * scale
        b = self.bias if (self.bias is None or scale == 1.0) else self.bias * scale
        self.pw = pack_weight(w, b, device=device)


class Linear(nn.Module):
    def __init__(self, cin: int, cout: int, bias: bool = True, geglu: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False) if bias else None
        self.cin, self.cout, self.geglu = cin, cout, geglu
        self.pw: Optional[PackedWeight] = None

    def pack(self, device):
        self.pw = pack_weight(self.weight, self.bias, geglu=self.geglu, device=device)


class Norm(nn.Module):
    """GroupNorm(32, C) / LayerNorm(C) affine parameters."""

    def __init__(self, c: int, eps: float):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.c, self.eps = c, eps
        self.g: Optional[torch.Tensor] = None
        self.b: Optional[torch.Tensor] = None

    def pack(self, device):
        self.g = self.weight.detach().to(device=device, dtype=torch.float32).contiguous()
        self.b = self.bias.detach().to(device=device, dtype=torch.float32).contiguous()


# Bumped by every pack_tree(): anything that bakes the ADDRESSES of packed weights (captured HIP graphs, cached hint-stem outputs:
# network.OpenAIWrapperControlLDM3DTV2V) keys on it, so a re-pack — a second checkpoint, a LoRA merge, new control scales — can never
# be followed by a replay that reads the freed weight tensors of the previous pack.
PACK_GENERATION = [0]


def pack_tree(module: nn.Module, device) -> None:
    """Pack every leaf container under `module` (composite modules may add fused weights on top)."""
    PACK_GENERATION[0] += 1
    for m in module.modules():
        if isinstance(m, (Conv, Linear, Norm)) and not getattr(m, "_packed_by_parent", False):
            m.pack(device)
    for m in reversed(list(module.modules())):          # children before their parents: a parent's post_pack may build on theirs
        post = getattr(m, "post_pack", None)
        if post is not None:
            post(device)
