"""AutoencoderKL decode (SD-1.5 KL-VAE) on the HIP kernels.

Reference: sgm/models/autoencoder.py:283-343 (AutoencoderKL, AutoencoderKLInferenceWrapper),
sgm/modules/diffusionmodules/model.py:94-151 (ResnetBlock), 161-201 (AttnBlock), 56-71 (Upsample),
617-761 (Decoder).  State-dict keys are the reference's (`decoder.up.3.block.0.norm1.weight`, ...).
The reference runs the VAE in fp32 (autocast disabled); here it runs in bf16 storage with fp32
accumulation / statistics like the rest of the path — the tolerance is stated in the tests.

Scope: `decode` (on the hot path).  `encode` parameters are held so checkpoints load and round-trip,
but running the encoder is a "next" row and raises NotImplementedError.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import ops
from .layers import Conv, Norm, Slot, pack_tree
from .packing import PackedWeight

GN_EPS = 1e-6          # model.py:50-53


def _ceil(a, b):
    return (a + b - 1) // b * b


class ResnetBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.norm1 = Norm(in_channels, GN_EPS)
        self.conv1 = Conv(in_channels, out_channels, 3)
        self.norm2 = Norm(out_channels, GN_EPS)
        self.conv2 = Conv(out_channels, out_channels, 3)
        if in_channels != out_channels:
            self.nin_shortcut = Conv(in_channels, out_channels, 1)

    def run(self, x):
        a = ops.groupnorm_spatial(x, self.norm1.g, self.norm1.b, GN_EPS, True)
        h = ops.conv2d(a, self.conv1.pw)
        a = ops.groupnorm_spatial(h, self.norm2.g, self.norm2.b, GN_EPS, True)
        skip = ops.conv2d(x, self.nin_shortcut.pw) if hasattr(self, "nin_shortcut") else x
        return ops.conv2d(a, self.conv2.pw, res1=skip.view(-1, skip.shape[-1]))


class AttnBlock(nn.Module):
    """Single-head attention with d = C (512): too wide for the flash kernel's register tile, and only
    1.3 TFLOP per clip, so it is evaluated per frame as GEMM(q k^T) -> row softmax -> GEMM(p v) with the
    same MFMA GEMM kernel.  V is produced already transposed (V^T = W_v . x^T, the GEMM with the roles of
    weight and activation swapped) so that it can serve as the K-contiguous operand of the second GEMM;
    its bias is added after p v (rows of p sum to 1)."""

    def __init__(self, c: int):
        super().__init__()
        self.c = c
        self.norm = Norm(c, GN_EPS)
        self.q, self.k, self.v, self.proj_out = Conv(c, c, 1), Conv(c, c, 1), Conv(c, c, 1), Conv(c, c, 1)
        self._wv = None
        self._bv = None

    def post_pack(self, device):
        self._wv = self.v.weight.detach().reshape(self.c, self.c).to(device=device, dtype=torch.bfloat16).contiguous()
        self._bv = self.v.bias.detach().to(device=device, dtype=torch.float32).contiguous()

    def run(self, x):
        n, h, w, c = x.shape
        L = h * w
        lp = _ceil(L, 64)
        a = ops.groupnorm_spatial(x, self.norm.g, self.norm.b, GN_EPS, False)
        a2 = torch.zeros((n * L + 256, c), dtype=torch.bfloat16, device=x.device)      # +256 zero rows: operand padding
        a2[: n * L].copy_(a.view(-1, c))
        q = ops.linear(a2[: n * L], self.q.pw)
        k = torch.zeros((n * L + 256, c), dtype=torch.bfloat16, device=x.device)
        ops.linear(a2[: n * L], self.k.pw, out=k[: n * L])
        o = torch.empty((n * L, c), dtype=torch.bfloat16, device=x.device)
        vt = torch.zeros((_ceil(c, 256), lp), dtype=torch.bfloat16, device=x.device)
        for f in range(n):
            kf = PackedWeight(k[f * L:], None, _ceil(L, 4), L, c, 1, c)
            s = ops.linear(q[f * L:(f + 1) * L], kf, out_f32=True)                       # [L, L] fp32 scores
            p = ops.softmax_rows(s, L, lp, float(c) ** -0.5)                             # [L, lp] bf16
            xf = PackedWeight(a2[f * L:], None, _ceil(L, 4), L, c, 1, c)
            ops.linear(self._wv, xf, out=vt[:c, :_ceil(L, 4)])                           # V^T [c, L]
            vw = PackedWeight(vt, self._bv, c, c, lp, 1, lp)
            ops.linear(p, vw, out=o[f * L:(f + 1) * L])
        y = ops.linear(o, self.proj_out.pw, res1=x.view(-1, c))
        return y.view(n, h, w, c)


class Upsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = Conv(c, c, 3)

    def run(self, x):
        return ops.conv2d(x, self.conv.pw, upsample=True)


class Decoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, z_channels, resolution,
                 in_channels=None, dropout=0.0, double_z=True, attn_type="vanilla", **ignored):
        super().__init__()
        if len(attn_resolutions) != 0:
            raise NotImplementedError("attention inside up-levels is not used by the SD-1.5 VAE config")
        self.ch_mult, self.num_res_blocks = list(ch_mult), num_res_blocks
        nres = len(ch_mult)
        block_in = ch * ch_mult[nres - 1]
        self.conv_in = Conv(z_channels, block_in, 3)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for i_level in reversed(range(nres)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            up = nn.Module()
            up.block = block
            up.attn = nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out = Norm(block_in, GN_EPS)
        self.conv_out = Conv(block_in, out_ch, 3)
        self.out_ch = out_ch

    def run(self, z8):
        h = ops.conv2d(z8, self.conv_in.pw)
        h = self.mid.block_1.run(h)
        h = self.mid.attn_1.run(h)
        h = self.mid.block_2.run(h)
        for lvl in reversed(range(len(self.ch_mult))):
            for i in range(self.num_res_blocks + 1):
                h = self.up[lvl].block[i].run(h)
            if lvl != 0:
                h = self.up[lvl].upsample.run(h)
        a = ops.groupnorm_spatial(h, self.norm_out.g, self.norm_out.b, GN_EPS, True)
        return ops.conv2d(a, self.conv_out.pw, out_f32=True)


class _EncoderParams(nn.Module):
    """Encoder parameters with the reference's keys (model.py:498-614) so checkpoints round-trip."""

    def __init__(self, *, ch, ch_mult, num_res_blocks, in_channels, z_channels, double_z=True, **ignored):
        super().__init__()
        self.conv_in = Conv(in_channels, ch, 3)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(len(ch_mult)):
            block = nn.ModuleList()
            block_in = ch * in_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(block_in, block_out))
                block_in = block_out
            down = nn.Module()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != len(ch_mult) - 1:
                down.downsample = nn.Module()
                down.downsample.conv = Conv(block_in, block_in, 3, stride=2)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(block_in, block_in)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(block_in, block_in)
        self.norm_out = Norm(block_in, GN_EPS)
        self.conv_out = Conv(block_in, 2 * z_channels if double_z else z_channels, 3)


class AutoencoderKL(nn.Module):
    def __init__(self, embed_dim: int, ddconfig=None, lossconfig=None, ckpt_path=None, monitor=None, **ignored):
        super().__init__()
        assert ddconfig["double_z"]
        dd = dict(ddconfig)
        self.encoder = _EncoderParams(**dd)
        self.decoder = Decoder(**dd)
        self.quant_conv = Conv(2 * dd["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = Conv(embed_dim, dd["z_channels"], 1)
        self.embed_dim = embed_dim
        self._packed = False

    def pack(self, device=None):
        device = torch.device("cuda") if device is None else device
        pack_tree(self, device)
        self._packed = True
        return self

    @property
    def device(self):
        return next(self.parameters()).device

    def encode(self, x):
        raise NotImplementedError("VAE encode is a 'next' scope row (SURVEY.md §8f): not on the TV2V hot path")

    def _decode_frames(self, z8):
        """z8: (N, h, w, 8) bf16 latent frames (4 real channels) -> (N, 8h, 8w, 4) fp32 (3 real channels)."""
        n, h, w, _ = z8.shape
        zc = self.post_quant_conv.pw
        zq = torch.zeros((n * h * w, 8), dtype=torch.bfloat16, device=z8.device)
        ops.conv2d(z8, zc, out=zq[:, : zc.n])
        return self.decoder.run(zq.view(n, h, w, 8))


class AutoencoderKLInferenceWrapper(AutoencoderKL):
    """autoencoder.py:322-343: accepts 4-D (N,C,H,W) or 5-D (B,C,T,H,W) latents."""

    def decode(self, z, **decoder_kwargs):
        if not self._packed:
            raise RuntimeError("call .pack() after loading weights")
        is_video = z.dim() == 5
        z5 = z if is_video else z[:, :, None]
        b, c, t, h, w = z5.shape
        z8 = ops.ncthw_to_nhwc(z5.float().contiguous(), 8)
        dec = self._decode_frames(z8)                                   # (b*t, H, W, 4) fp32
        out = ops.nhwc_to_ncthw(dec, b, t, self.decoder.out_ch)
        return out if is_video else out[:, :, 0]
