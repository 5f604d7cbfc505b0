"""The first-stage model (KL-VAE) evaluated in fp32 — the arithmetic class the reference decodes in.

Reference: `VideoDiffusionEngine.decode_first_stage` (sgm/models/diffusion.py:151-156) wraps the decoder in
`torch.autocast("cuda", enabled=not self.disable_first_stage_autocast)`; the CCEdit configs set
`disable_first_stage_autocast: True`, so `Decoder.forward` (sgm/modules/diffusionmodules/model.py:728-761) runs on fp32
tensors with fp32 products.  `ccedit_amd/vae.py` evaluates the same graph on the bf16 kernels of the denoising path by
default (a clip's decode is 1.4 % of its FLOPs and the bf16 result is within the tolerance the tests state); this module is
the option that removes the precision difference: every tensor fp32, every contraction on `v_mfma_f32_32x32x2_f32`
(`csrc/f32vae.hip`: `ccedit_gemm_f32`, `ccedit_groupnorm_f32`, `ccedit_softmax_rows_f32`).  Selected by policy `vae_fp32=1`
(or `AutoencoderKL.precision = "fp32"`); `bench.py` reports both decode times on its `clip` object.

HIP only, like the rest of the package: no CPU fallback, nothing here touches `oracle/`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from . import hip, ops

F32 = torch.float32
GN_EPS = 1e-6          # model.py:50-53 (Normalize), as in vae.py


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b * b


@dataclass
class PackedF32:
    """fp32 weights in kernel layout: w [n][taps][cpad] (tap-major K, zero columns beyond cin), bias [n] or None."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    n: int
    cin: int          # source channels rounded up to 4 (the activation tensors carry zero channels there)
    cpad: int
    taps: int


def pack_f32(weight: torch.Tensor, bias: Optional[torch.Tensor], device) -> PackedF32:
    """Reference layout (O, I, 3, 3) / (O, I, 1, 1) / (O, I) fp32 -> PackedF32."""
    w = weight.detach().to(dtype=F32, device="cpu")
    if w.dim() == 4 and w.shape[2] == 3:
        o, i = w.shape[:2]
        taps, w3 = 9, w.permute(0, 2, 3, 1).reshape(o, 9, i)          # tap = 3 ky + kx
    elif w.dim() == 4 and w.shape[2] == 2:                            # a parity window of upsample + conv (pack_f32_parities)
        o, i = w.shape[:2]
        taps, w3 = 4, w.permute(0, 2, 3, 1).reshape(o, 4, i)          # tap = 2 dy + dx
    else:
        o, i = w.shape[:2]
        taps, w3 = 1, w.reshape(o, 1, i)
    cpad = _ceil(i, 16)
    p = torch.zeros((o, taps, cpad), dtype=F32)
    p[:, :, :i] = w3
    b = None if bias is None else bias.detach().to(device=device, dtype=F32).contiguous()
    return PackedF32(p.reshape(o, taps * cpad).contiguous().to(device), b, o, _ceil(i, 4), cpad, taps)


def pack_f32_parities(weight: torch.Tensor, bias: Optional[torch.Tensor], device):
    """conv3x3(nearest_upsample_2x(x)) (model.py:56-71) as four 2 x 2 convolutions on x itself, one per output parity (py, px): output
    pixel (2y + py, 2x + px) reads up-sampled rows 2y + py - 1 + ky, i.e. source rows y - 1, y, y for py = 0 and y, y, y + 1 for py = 1;
    taps that land on the same source pixel are added up (in fp64, one fp32 rounding per merged tap).  4/9 of the multiply-adds of the
    nine-tap gather; the sum of two or four products w_i x becomes (sum w_i) x — equal to fp32 rounding, not bit for bit.  Returns the
    four PackedF32 in CcGemmF32Desc.upsample order p = 2 py + px (the bf16 path's packing.pack_upsample_parities restated for fp32)."""
    w = weight.detach().to(dtype=torch.float64, device="cpu")
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3
    merge = (((0,), (1, 2)), ((0, 1), (2,)))               # parity -> for each of the two window positions, the 3 x 3 taps it collects
    out = []
    for py in range(2):
        for px in range(2):
            w2 = torch.zeros(w.shape[0], w.shape[1], 2, 2, dtype=torch.float64)
            for dy in range(2):
                for dx in range(2):
                    for ky in merge[py][dy]:
                        for kx in merge[px][dx]:
                            w2[:, :, dy, dx] += w[:, :, ky, kx]
            out.append(pack_f32(w2.to(F32), bias, device))
    return out


def _chk(t: torch.Tensor, name: str):
    if t.dtype != F32 or not t.is_cuda or t.stride(-1) != 1:
        raise ValueError(f"{name}: expected a cuda fp32 tensor with contiguous rows, got {t.dtype} {t.device}")


def gemm_f32(a2d: torch.Tensor, pw: PackedF32, *, m: Optional[int] = None, conv=None, res: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, ldc: Optional[int] = None, use_bias: bool = True) -> torch.Tensor:
    """out[m][n] = bias + sum_k W[n][k] src(a2d)[m][k] (+ res).  conv = (hin, win, hout, wout, stride, pad, upsample) selects the
    3x3 gather; otherwise a2d's rows are the GEMM rows."""
    _chk(a2d, "gemm_f32.a")
    d = hip.CcGemmF32Desc()
    m = a2d.shape[0] if m is None else m
    if out is None:
        out = torch.empty((m, ldc or pw.n), dtype=F32, device=a2d.device)
    _chk(out, "gemm_f32.out")
    d.A, d.W, d.out = a2d.data_ptr(), pw.w.data_ptr(), out.data_ptr()
    d.bias = pw.bias.data_ptr() if (use_bias and pw.bias is not None) else None
    d.M, d.N, d.Cin, d.Cpad, d.Kpad = m, pw.n, pw.cin, pw.cpad, pw.taps * pw.cpad
    d.lda, d.ldw, d.ldc = a2d.stride(0), pw.w.stride(0), out.stride(0)
    if res is not None:
        _chk(res, "gemm_f32.res")
        d.res, d.ldr = res.data_ptr(), res.stride(0)
    if conv is not None:
        if pw.taps != (4 if int(conv[6]) >= 2 else 9):
            raise ValueError("gemm_f32: conv geometry given for a 1x1 weight (or a parity geometry for a nine-tap weight)")
        d.mode = 1
        d.Hin, d.Win, d.Hout, d.Wout, d.stride, d.pad, d.upsample = (int(v) for v in conv)
    elif pw.taps != 1:
        raise ValueError("gemm_f32: a 3x3 weight needs the conv geometry")
    hip.check(hip.lib().ccedit_gemm_f32(d, ops._stream()), "ccedit_gemm_f32")
    return out


def conv2d_f32(x: torch.Tensor, pw: PackedF32, stride: int = 1, pad: int = 1, upsample: bool = False, out_hw=None,
               res: Optional[torch.Tensor] = None, ldc: Optional[int] = None) -> torch.Tensor:
    """x (N, H, W, C) fp32 -> (N, Hout, Wout, ldc or Cout) fp32; 3x3 or 1x1 by the packed weight."""
    n, h, w, c = x.shape
    if pw.taps == 1:
        out = gemm_f32(x.reshape(-1, c), pw, res=None if res is None else res.reshape(-1, res.shape[-1]), ldc=ldc)
        return out.view(n, h, w, out.shape[-1])
    hv, wv = (2 * h, 2 * w) if upsample else (h, w)
    hout, wout = out_hw if out_hw is not None else ((hv + 2 * pad - 3) // stride + 1, (wv + 2 * pad - 3) // stride + 1)
    out = gemm_f32(x.reshape(-1, c), pw, m=n * hout * wout, conv=(h, w, hout, wout, stride, pad, int(upsample)),
                   res=None if res is None else res.reshape(-1, res.shape[-1]), ldc=ldc)
    return out.view(n, hout, wout, out.shape[-1])


def parity_form_available() -> bool:
    """upsample + conv 3x3 as four parity convs: host policy `subpix` (shared with the bf16 path) and the six-product kernels (library
    policy `f32_split`, read from the library: tests flip it at run time)."""
    import ctypes
    from . import policy
    v = ctypes.c_int32(0)
    hip.check(hip.lib().ccedit_policy_get(b"f32_split", ctypes.byref(v)), "ccedit_policy_get")
    return policy.on("subpix") and v.value != 0


def upsample_conv2d_f32(x: torch.Tensor, conv) -> torch.Tensor:
    """conv3x3(nearest_upsample_2x(x)), x (N, H, W, C) fp32 -> (N, 2H, 2W, Cout): four parity convs on x (4/9 of the FLOPs) where the
    six-product kernels run, else the nine-tap gather over the virtual up-sampled source."""
    if not parity_form_available():
        return conv2d_f32(x, _pw(conv), upsample=True)
    n, h, w, c = x.shape
    packs = _pw_parities(conv)
    out = torch.empty((n, 2 * h, 2 * w, packs[0].n), dtype=F32, device=x.device)
    for p, pw in enumerate(packs):
        gemm_f32(x.reshape(-1, c), pw, m=n * h * w, conv=(h, w, h, w, 1, 1, 2 + p), out=out.view(-1, pw.n))
    return out


_gn_ws = {}


def groupnorm_f32(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, silu: bool) -> torch.Tensor:
    """GroupNorm(32, C, eps)(x) [+ SiLU], x (N, H, W, C) fp32 contiguous."""
    _chk(x, "groupnorm_f32.x")
    assert x.is_contiguous() and x.ndim == 4
    n, h, w, c = x.shape
    key = (x.device, ops._stream())
    ws = _gn_ws.get(key)
    if ws is None or ws.numel() < n * 64:
        ws = _gn_ws[key] = torch.empty(max(n * 64, 4096), dtype=torch.float64, device=x.device)
    y = torch.empty_like(x)
    hip.check(hip.lib().ccedit_groupnorm_f32(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), n, h * w, c,
                                             eps, int(silu), ops._stream()), "ccedit_groupnorm_f32")
    return y


def softmax_rows_f32_(s: torch.Tensor, cols: int, scale: float) -> torch.Tensor:
    """In place: s[:, :cols] = softmax(scale * s[:, :cols]) row-wise."""
    _chk(s, "softmax_rows_f32.s")
    hip.check(hip.lib().ccedit_softmax_rows_f32(s.data_ptr(), s.shape[0], cols, s.stride(0), scale, ops._stream()), "ccedit_softmax_rows_f32")
    return s


# ------------------------------------------------------------------------------------------
# the module walk (same tree, same parameter names as vae.py; fp32 packs are built on first use and follow re-packs)

def _pw(conv) -> PackedF32:
    from .layers import PACK_GENERATION
    ent = getattr(conv, "_pw32", None)
    if ent is None or ent[0] != PACK_GENERATION[0] or ent[1].w.device != conv.pw.w.device:
        ent = conv._pw32 = (PACK_GENERATION[0], pack_f32(conv.weight, conv.bias, conv.pw.w.device))
    return ent[1]


def _pw_parities(conv):
    from .layers import PACK_GENERATION
    ent = getattr(conv, "_pw32p", None)
    if ent is None or ent[0] != PACK_GENERATION[0] or ent[1][0].w.device != conv.pw.w.device:
        ent = conv._pw32p = (PACK_GENERATION[0], pack_f32_parities(conv.weight, conv.bias, conv.pw.w.device))
    return ent[1]


def resnet_block(blk, x):
    a = groupnorm_f32(x, blk.norm1.g, blk.norm1.b, GN_EPS, True)
    h = conv2d_f32(a, _pw(blk.conv1))
    a = groupnorm_f32(h, blk.norm2.g, blk.norm2.b, GN_EPS, True)
    skip = conv2d_f32(x, _pw(blk.nin_shortcut)) if hasattr(blk, "nin_shortcut") else x
    return conv2d_f32(a, _pw(blk.conv2), res=skip)


def attn_block(blk, x):
    """model.py:161-201 per frame: q k^T (fp32 scores) -> softmax -> p v, all through ccedit_gemm_f32 with the second tensor in the
    role of the weight matrix.  V arrives transposed (V^T = W_v x^T: the GEMM with weight and activation swapped) so that it is the
    K-contiguous operand of p v; its bias is added after the product (rows of p sum to 1)."""
    n, h, w, c = x.shape
    L = h * w
    l4, l16 = _ceil(L, 4), _ceil(L, 16)
    a = groupnorm_f32(x, blk.norm.g, blk.norm.b, GN_EPS, False).view(n * L, c)
    q = gemm_f32(a, _pw(blk.q))
    k = gemm_f32(a, _pw(blk.k))
    wv = _pw(blk.v)
    o = torch.empty((n * L, c), dtype=F32, device=x.device)
    s = torch.zeros((L, l4), dtype=F32, device=x.device)                  # columns [L, l4) stay zero: K padding of p v
    vt = torch.zeros((c, l16), dtype=F32, device=x.device)
    wv_act = wv.w[:, :c] if wv.cpad == c else wv.w[:, :c].contiguous()
    for f in range(n):
        kf = PackedF32(k[f * L:(f + 1) * L], None, L, c, _ceil(c, 16), 1)
        gemm_f32(q[f * L:(f + 1) * L], kf, out=s)
        softmax_rows_f32_(s, L, float(c) ** -0.5)
        xf = PackedF32(a[f * L:(f + 1) * L], None, L, c, _ceil(c, 16), 1)
        gemm_f32(wv_act, xf, out=vt)                                       # V^T [c][L] (no bias)
        gemm_f32(s, PackedF32(vt, wv.bias, c, l4, l16, 1), out=o[f * L:(f + 1) * L])
    y = gemm_f32(o, _pw(blk.proj_out), res=x.view(-1, c))
    return y.view(n, h, w, c)


def decoder(dec, z4: torch.Tensor) -> torch.Tensor:
    """z4 (N, h, w, 4) fp32 (post_quant_conv output) -> (N, 8h, 8w, 4) fp32, 3 real channels (model.py:728-761)."""
    h = conv2d_f32(z4, _pw(dec.conv_in))
    h = resnet_block(dec.mid.block_1, h)
    h = attn_block(dec.mid.attn_1, h)
    h = resnet_block(dec.mid.block_2, h)
    for lvl in reversed(range(len(dec.ch_mult))):
        for i in range(dec.num_res_blocks + 1):
            h = resnet_block(dec.up[lvl].block[i], h)
        if lvl != 0:
            h = upsample_conv2d_f32(h, dec.up[lvl].upsample.conv)
    a = groupnorm_f32(h, dec.norm_out.g, dec.norm_out.b, GN_EPS, True)
    del h
    n, hh, ww, _ = a.shape
    out = torch.zeros((n * hh * ww, 4), dtype=F32, device=a.device)
    pw = _pw(dec.conv_out)
    gemm_f32(a.view(-1, a.shape[-1]), pw, m=n * hh * ww, conv=(hh, ww, hh, ww, 1, 1, 0), out=out)
    return out.view(n, hh, ww, 4)


def encoder(enc, x4: torch.Tensor) -> torch.Tensor:
    """x4 (N, H, W, 4) fp32 frames (3 real channels, the fourth zero) -> (N, H/8, W/8, 2 z) fp32 pre-quant moments
    (model.py:498-614; Downsample :74-93 = pad right / bottom, conv stride 2 pad 0)."""
    h = conv2d_f32(x4, _pw(enc.conv_in))
    for lvl in range(len(enc.ch_mult)):
        for i in range(enc.num_res_blocks):
            h = resnet_block(enc.down[lvl].block[i], h)
        if lvl != len(enc.ch_mult) - 1:
            _, hh, ww, _ = h.shape
            h = conv2d_f32(h, _pw(enc.down[lvl].downsample.conv), stride=2, pad=0, out_hw=((hh + 1 - 3) // 2 + 1, (ww + 1 - 3) // 2 + 1))
    h = resnet_block(enc.mid.block_1, h)
    h = attn_block(enc.mid.attn_1, h)
    h = resnet_block(enc.mid.block_2, h)
    a = groupnorm_f32(h, enc.norm_out.g, enc.norm_out.b, GN_EPS, True)
    return conv2d_f32(a, _pw(enc.conv_out))


def decode_frames(vae, z5: torch.Tensor) -> torch.Tensor:
    """z5 (B, 4, T, h, w) fp32 -> (B*T, 8h, 8w, 4) fp32 channels-last."""
    b, c, t, h, w = z5.shape
    z = z5.permute(0, 2, 3, 4, 1).reshape(b * t, h, w, c).contiguous()
    zq = conv2d_f32(z, _pw(vae.post_quant_conv))
    return decoder(vae.decoder, zq)


def encode_moments(vae, x5: torch.Tensor) -> torch.Tensor:
    """x5 (B, 3, T, H, W) fp32 frames -> quant_conv moments (B*T*h*w, 2 z) fp32 [mean | logvar]."""
    b, c, t, h, w = x5.shape
    x4 = torch.zeros((b * t, h, w, 4), dtype=F32, device=x5.device)
    x4[..., :c] = x5.permute(0, 2, 3, 4, 1).reshape(b * t, h, w, c)
    pre = encoder(vae.encoder, x4)
    return conv2d_f32(pre, _pw(vae.quant_conv))
