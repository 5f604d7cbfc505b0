"""Re-pack reference-layout fp32 weights (OIHW / (O,I,K) / (O,I)) into the kernels' bf16 layout.

The state-dict contract keeps the reference's key names and tensor layouts; packing is
a one-time, load-time transformation into what `ccedit_gemm` consumes:

    W_packed[Opad][Kpad]   bf16,  K index = tap * Cin_pad + c   (K contiguous),
    Opad = ceil(N, 256), Cin_pad = ceil(Cin, 8), Kpad = ceil(taps * Cin_pad, 64), zero padded.

Tap order: Conv2d tap = ky * kw + kx (source pixel (oy*s + ky - pad, ox*s + kx - pad));
Conv1d-over-T tap = k (source frame t + k - K//2).  GEGLU projections are row-interleaved in blocks
of 16 ([8 value rows][8 gate rows]) so that a value and its gate land in the same MFMA lane.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b * b


@dataclass
class PackedWeight:
    w: torch.Tensor                 # bf16 [Opad, Kpad]
    bias: Optional[torch.Tensor]    # fp32 [N] (packed row order) or None
    n: int                          # packed rows in use (multiple of 4)
    n_out: int                      # logical output channels (n // 2 for GEGLU)
    cin: int                        # padded channels per tap (multiple of 8)
    taps: int
    kpad: int
    ksize: int = 1
    geglu: bool = False
    flops_per_row: float = 0.0      # algorithmic 2*O*I*taps (unpadded) per output row
    korder: int = 0                 # 0: K = [tap][Cin];  1: K = [Cin/64][tap][64]

    def to(self, device):
        self.w = self.w.to(device)
        if self.bias is not None:
            self.bias = self.bias.to(device)
        return self


def _geglu_perm(two_inner: int) -> torch.Tensor:
    inner = two_inner // 2
    assert inner % 8 == 0, "GEGLU inner dim must be a multiple of 8"
    j = torch.arange(inner // 8)[:, None] * 8 + torch.arange(8)[None, :]      # [blocks][8] value rows
    return torch.cat([j, j + inner], dim=1).reshape(-1)                        # [8 value | 8 gate] per block


def pack_weight(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, geglu: bool = False,
                device: Optional[torch.device] = None) -> PackedWeight:
    """weight: (O, I) | (O, I, K) | (O, I, KH, KW) fp32 in the reference layout."""
    w = weight.detach().to(torch.float32)
    if w.ndim == 2:
        o, i = w.shape
        taps, ksize = 1, 1
        w3 = w[:, None, :]
    elif w.ndim == 3:
        o, i, k = w.shape
        taps, ksize = k, k
        w3 = w.permute(0, 2, 1)
    elif w.ndim == 4:
        o, i, kh, kw = w.shape
        assert kh == kw
        taps, ksize = kh * kw, kh
        w3 = w.permute(0, 2, 3, 1).reshape(o, taps, i)
    else:
        raise ValueError(f"cannot pack weight of shape {tuple(w.shape)}")
    b = None if bias is None else bias.detach().to(torch.float32)
    if geglu:
        perm = _geglu_perm(o).to(w3.device)
        w3 = w3[perm]
        if b is not None:
            b = b[perm]
    cin = _ceil(i, 8)
    n = _ceil(o, 4)
    opad = _ceil(n, 256)            # the widest GEMM block shape covers 256 output channels
    kpad = _ceil(taps * cin, 64)
    dev = device if device is not None else w3.device
    buf = torch.zeros(opad, kpad, dtype=torch.bfloat16, device=dev)
    tmp = torch.zeros(o, taps, cin, dtype=torch.float32, device=w3.device)
    tmp[:, :, :i] = w3
    korder = 1 if (taps > 1 and cin % 64 == 0) else 0
    if korder:      # taps of one 64-channel chunk adjacent in K (see gemm.hip: L2 reuse across the 3x3 taps)
        tmp = tmp.reshape(o, taps, cin // 64, 64).permute(0, 2, 1, 3).contiguous()
    buf[:o, : taps * cin] = tmp.reshape(o, taps * cin).to(torch.bfloat16).to(dev)
    bb = None
    if b is not None:
        bb = torch.zeros(n, dtype=torch.float32, device=dev)
        bb[:o] = b.to(dev)
    return PackedWeight(buf, bb, n, (o // 2) if geglu else o, cin, taps, kpad, ksize, geglu, 2.0 * o * i * taps, korder)


def pack_concat(weights: Sequence[torch.Tensor], biases: Optional[Sequence[Optional[torch.Tensor]]] = None,
                device: Optional[torch.device] = None) -> PackedWeight:
    """Stack several (O_i, I) projections along O (fused q|k|v GEMM)."""
    w = torch.cat([x.detach().to(torch.float32) for x in weights], dim=0)
    b = None
    if biases is not None and any(x is not None for x in biases):
        b = torch.cat([torch.zeros(wi.shape[0]) if bi is None else bi.detach().float() for wi, bi in zip(weights, biases)])
    return pack_weight(w, b, device=device)
