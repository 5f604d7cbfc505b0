"""CPU oracle: a plain fp32 restatement of CCEdit's denoising hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it, and only as the checker.  `ccedit_amd/` never
imports anything from `oracle/`; the product path has no CPU fallback.

What it restates (reference = RuoyuFeng/CCEdit, paths relative to /root/reference):
  * sgm/modules/diffusionmodules/wrappers.py:156-207   OpenAIWrapperControlLDM3DTV2V.forward
  * sgm/modules/diffusionmodules/controlmodel.py:252-317  ControlNet2D.forward
  * sgm/modules/diffusionmodules/controlmodel.py:471-550  ControlledUNetModel3DTV2V.forward
  * sgm/modules/diffusionmodules/openaimodel.py:129-178 (spatial_temporal_forward), 528-554
    (ResBlock), 730-775 (ResBlock3D), 254-263 / 388-394 (Up/Downsample3D), 1033-1527 (wiring)
  * sgm/modules/attention.py:392-467 (CrossAttention), 695-716, 758-761 (transformer blocks),
    865-889 (SpatialTransformer), 1141-1208 (SpatialTransformer3D), 115-141 (GEGLU/FeedForward)
  * sgm/modules/diffusionmodules/{discretizer,denoiser,denoiser_scaling,guiders,sampling,
    sampling_utils}.py  (sigma schedule, DiscreteDenoiser, VanillaCFGTV2V, DPMPP2SAncestral)
  * sgm/modules/diffusionmodules/model.py:728-761 (VAE Decoder), sgm/models/autoencoder.py:334-343

Style: the reference is an nn.Module tree; this oracle is *functional* — every function takes the
reference-named state dict `sd` (key -> fp32 tensor) and a key prefix, so the weights contract
 is the only coupling.  Arithmetic is torch fp32 on CPU in the reference's own
(N, C, ...) layouts so that ATen's kernels are the same ones the reference's CPU path runs.

Pinning: tests/test_oracle_golden.py checks every function here against vectors recorded from the
reference itself (tests/golden/make_golden.py imports /root/reference in the authoring container).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

GN_EPS_RES = 1e-5    # diffusionmodules/util.py:296-302  normalization() = nn.GroupNorm(32, C)
GN_EPS_ATTN = 1e-6   # attention.py:153-156 Normalize(); model.py:50-53 (VAE)


# ----------------------------------------------------------------------------------------
# configuration / topology
# ----------------------------------------------------------------------------------------
@dataclass
class NetConfig:
    """Hyper-parameters of configs/inference_ccedit/keyframe_no2ndca_depthmidas.yaml:25-56."""
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: int = 768
    hint_channels: int = 3
    control_scales: float = 1.0
    # TVI2V (keyframe_ref_cp_no2ndca_add_cfca_depthzoe.yaml:32-90): anchor cross-frame attention + controlnet_img
    crossframe: bool = False


@dataclass
class BlockSpec:
    kind: str                 # "conv_in" | "res" | "down" | "up-tail"
    cin: int = 0
    cout: int = 0
    attn: bool = False
    up: bool = False


def unet_topology(cfg: NetConfig):
    """Block list exactly as UNetModel.__init__ builds it (openaimodel.py:1230-1500)."""
    mc = cfg.model_channels
    inputs: List[BlockSpec] = [BlockSpec("conv_in", cfg.in_channels, mc)]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            inputs.append(BlockSpec("res", ch, mult * mc, attn=ds in cfg.attention_resolutions))
            ch = mult * mc
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            inputs.append(BlockSpec("down", ch, ch))
            chans.append(ch)
            ds *= 2
    mid_ch = ch
    outputs: List[BlockSpec] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            spec = BlockSpec("res", ch + ich, mc * mult, attn=ds in cfg.attention_resolutions)
            ch = mc * mult
            if level and i == cfg.num_res_blocks:
                spec.up = True
                ds //= 2
            outputs.append(spec)
    return inputs, mid_ch, outputs


# ----------------------------------------------------------------------------------------
# leaf ops
# ----------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[cos, sin] sinusoidal embedding — diffusionmodules/util.py:244-268."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# bf16-emulation mode (test infrastructure, like everything here).  The fp32 restatement is what the reference's golden
# vectors pin; with `with bf16_emulation():` the SAME functions additionally round to bfloat16 wherever the MI355X path
# stores a tensor (its activations live in HBM as bf16, weights are bf16, accumulation and normalisation statistics are
# fp32): conv / linear outputs AFTER their fused bias / row-bias / residual / activation epilogue, GroupNorm(+SiLU) and
# LayerNorm outputs, the softmax probabilities that enter P.V, the GEGLU hidden activation, the skip concatenation.  A
# correct HIP path then differs from this mode only by fp32 summation order (and the rare bf16 tie it flips), so the GPU
# parity tests can hold it to a tolerance several times tighter than the fp32-vs-bf16 noise floor — tight enough to
# expose a wrong low-energy branch.  Outside the context manager nothing changes (same ATen calls, same values).
_EMU = False
# HIP-path choices the emulation has to follow where they move a rounding point
EMU_FF_FUSED_MIN_TOKENS = 1024      # ccedit_amd/network.py: FeedForward.run uses the fused dim-320 kernel from this many tokens on


class bf16_emulation:
    def __enter__(self):
        global _EMU
        self._old, _EMU = _EMU, True
        return self

    def __exit__(self, *a):
        global _EMU
        _EMU = self._old


def _R(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if _EMU else x


def _W(sd: SD, key: str) -> torch.Tensor:
    return _R(sd[key])


def _fin(y, add, act, rnd):
    for a in add:
        if a is not None:
            y = y + a
    if act is not None:
        y = act(y)
    return _R(y) if rnd else y


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float, silu: bool = False) -> torch.Tensor:
    y = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)
    return _R(F.silu(y) if silu else y)


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return _R(F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5))


def _conv2d(sd: SD, p: str, x, stride=1, padding=0, add=(), act=None, rnd=True):
    """add: tensors summed in the epilogue before the (emulated) store; act: activation after them."""
    return _fin(F.conv2d(x, _W(sd, p + ".weight"), sd.get(p + ".bias"), stride=stride, padding=padding), add, act, rnd)


def _conv1d(sd: SD, p: str, x, padding=0, add=(), act=None, rnd=True):
    return _fin(F.conv1d(x, _W(sd, p + ".weight"), sd.get(p + ".bias"), padding=padding), add, act, rnd)


def _linear(sd: SD, p: str, x, add=(), act=None, rnd=True):
    return _fin(F.linear(x, _W(sd, p + ".weight"), sd.get(p + ".bias")), add, act, rnd)


def time_embed(sd: SD, p: str, t: torch.Tensor, model_channels: int) -> torch.Tensor:
    """timestep_embedding -> Linear, SiLU, Linear (openaimodel.py:1216-1223)."""
    e = _R(timestep_embedding(t, model_channels))
    return _linear(sd, p + ".2", _linear(sd, p + ".0", e, act=F.silu))


def _emb_out(sd: SD, p: str, emb):
    """emb_layers = SiLU, Linear (openaimodel.py:470-476); the HIP path keeps this (B, C) row bias in fp32."""
    return _linear(sd, p + ".emb_layers.1", _R(F.silu(emb)), rnd=False)


# ----------------------------------------------------------------------------------------
# 2D+1D factorisation
# ----------------------------------------------------------------------------------------
def _to_frames(x5):                      # (b c t h w) -> (b t) c h w
    b, c, t, h, w = x5.shape
    return x5.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)


def _frames_to_pix(x4, b):               # (b t) c h w -> (b h w) c t
    bt, c, h, w = x4.shape
    t = bt // b
    return x4.reshape(b, t, c, h, w).permute(0, 3, 4, 2, 1).reshape(b * h * w, c, t)


def _pix_to_5d(xp, b, h, w):             # (b h w) c t -> b c t h w
    _, c, t = xp.shape
    return xp.reshape(b, h, w, c, t).permute(0, 3, 4, 1, 2).contiguous()


def _5d_to_pix(x5):                      # b c t h w -> (b h w) c t
    b, c, t, h, w = x5.shape
    return x5.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, t)


def stf(x5, spatial: Callable, temporal: Optional[Callable], extra: Sequence = ()):
    """spatial_temporal_forward (openaimodel.py:129-178): y = temporal(s) + s, s = spatial(x).  `temporal(sp, add)` must sum
    `add` (s itself and, for the callers that fold their next additions in, the `extra` (b c t h w) tensors) into its
    output — on the HIP path these are one GEMM epilogue with a single store."""
    b = x5.shape[0]
    s = spatial(_to_frames(x5))
    h, w = s.shape[-2:]
    sp = _frames_to_pix(s, b)
    add = [sp] + [_5d_to_pix(e.expand(b, s.shape[1], x5.shape[2], h, w)) for e in extra if e is not None]
    if temporal is not None:
        tmp = temporal(sp, add)
    else:
        tmp = sp
        for e in add[1:]:
            tmp = tmp + e
    return _pix_to_5d(tmp, b, h, w)


# ----------------------------------------------------------------------------------------
# residual blocks
# ----------------------------------------------------------------------------------------
def resblock2d(sd: SD, p: str, x, emb):
    """ResBlock._forward, use_scale_shift_norm=False, no up/down (openaimodel.py:528-554)."""
    h = _conv2d(sd, p + ".in_layers.2", _gn(sd, p + ".in_layers.0", x, GN_EPS_RES, silu=True), padding=1,
                add=[_emb_out(sd, p, emb)[:, :, None, None]])
    skip = x if (p + ".skip_connection.weight") not in sd else _conv2d(sd, p + ".skip_connection", x)
    return _conv2d(sd, p + ".out_layers.3", _gn(sd, p + ".out_layers.0", h, GN_EPS_RES, silu=True), padding=1, add=[skip])


def resblock3d(sd: SD, p: str, x5, emb):
    """ResBlock3D._forward (openaimodel.py:730-775); emb is (B, E), broadcast over (t,h,w)."""
    def sp_in(x):
        return _conv2d(sd, p + ".in_layers.2", _gn(sd, p + ".in_layers.0", x, GN_EPS_RES, silu=True), padding=1)

    def tp_in(x, add):
        return _conv1d(sd, p + ".in_layers_temporal.2", _gn(sd, p + ".in_layers_temporal.0", x, GN_EPS_RES, silu=True),
                       padding=1, add=add)

    def sp_out(x):
        return _conv2d(sd, p + ".out_layers.3", _gn(sd, p + ".out_layers.0", x, GN_EPS_RES, silu=True), padding=1)

    def tp_out(x, add):
        return _conv1d(sd, p + ".out_layers_temporal.3", _gn(sd, p + ".out_layers_temporal.0", x, GN_EPS_RES, silu=True),
                       padding=1, add=add)

    # h = stf(x) + emb_out  and  return skip + stf(h): each sum is the epilogue of the temporal conv that precedes it
    h = stf(x5, sp_in, tp_in, extra=[_emb_out(sd, p, emb)[:, :, None, None, None]])
    if (p + ".skip_connection.weight") in sd:
        skip = stf(x5, lambda x: _conv2d(sd, p + ".skip_connection", x),
                   lambda x, add: _conv1d(sd, p + ".skip_connection_temporal", x, add=add))
    else:
        skip = x5            # Identity spatial, temporal None -> zeros + identity
    return stf(h, sp_out, tp_out, extra=[skip])


# ----------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------
def _sdpa(q, k, v):
    if not _EMU:
        return F.scaled_dot_product_attention(q, k, v)
    # HIP kernels: fp32 scores, probabilities rounded to bf16 where they enter the P.V MFMA, the row sum taken from those
    # rounded values (a ones column in the same MFMA), the normalised output stored as bf16
    def one(q, k, v):
        s = (q @ k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
        pb = _R(torch.exp(s - s.amax(dim=-1, keepdim=True)))
        return _R((pb @ v) / pb.sum(dim=-1, keepdim=True))
    # rows are independent: long sequences go one batch entry at a time (a 17-frame 6144 x 6144 score tensor is 41 GB)
    if q.shape[0] > 1 and q.shape[0] * q.shape[1] * q.shape[2] * k.shape[2] > (1 << 29):
        return torch.cat([one(q[i:i + 1], k[i:i + 1], v[i:i + 1]) for i in range(q.shape[0])])
    return one(q, k, v)


def cross_attention(sd: SD, p: str, x, context, heads: int, res=None):
    """CrossAttention.forward (attention.py:392-467): bias-free q/k/v, SDPA scale d^-0.5, out+bias.  `res`: the residual the
    caller adds to the output (folded into to_out's epilogue)."""
    ctx = x if context is None else context
    q, k, v = _linear(sd, p + ".to_q", x), _linear(sd, p + ".to_k", ctx), _linear(sd, p + ".to_v", ctx)
    b, n, c = q.shape
    d = c // heads
    q = q.reshape(b, n, heads, d).transpose(1, 2)
    k = k.reshape(b, -1, heads, d).transpose(1, 2)
    v = v.reshape(b, -1, heads, d).transpose(1, 2)
    o = _sdpa(q, k, v).transpose(1, 2).reshape(b, n, c)
    return _linear(sd, p + ".to_out.0", o, add=[res])


def feed_forward(sd: SD, p: str, pn: str, x):
    """x + FeedForward(LayerNorm(x)) with GEGLU (attention.py:115-141, 695-716): proj -> (a, gate) -> a*gelu_erf(gate) ->
    Linear.  pn = key prefix of the LayerNorm."""
    fused = _EMU and x.shape[-1] == 320 and x.numel() // 320 >= EMU_FF_FUSED_MIN_TOKENS
    if fused:
        # csrc/ff320.hip: the operand is the normalised x WITHOUT the affine part, gamma / beta are folded into the weights
        # (W1' = bf16(W1 diag(gamma)), b1' = b1 + W1 beta in fp32)
        xn = _R(F.layer_norm(x, (320,), None, None, 1e-5))
        w1, g, be = sd[p + ".net.0.proj.weight"], sd[pn + ".weight"], sd[pn + ".bias"]
        pre = F.linear(xn, _R(w1 * g[None, :]), sd[p + ".net.0.proj.bias"] + w1 @ be)
    else:
        pre = _linear(sd, p + ".net.0.proj", _ln(sd, pn, x), rnd=False)
    a, gate = pre.chunk(2, dim=-1)
    return _linear(sd, p + ".net.2", _R(a * F.gelu(gate)), add=[x])


def basic_block(sd: SD, p: str, x, context, heads: int):
    """BasicTransformerBlock._forward (attention.py:695-716)."""
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), None, heads, res=x)
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), context, heads, res=x)
    return feed_forward(sd, p + ".ff", p + ".norm3", x)


def single_block(sd: SD, p: str, x, context, heads: int):
    """BasicTransformerSingleLayerBlock._forward (attention.py:758-761).  Callers pass
    context = the *un-normalised* x (attention.py:1191-1192), so K/V skip the LayerNorm."""
    x = cross_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), context, heads, res=x)
    return feed_forward(sd, p + ".ff", p + ".norm2", x)


def spatial_transformer2d(sd: SD, p: str, x, context, heads: int):
    """SpatialTransformer.forward, use_linear=False, depth 1 (attention.py:865-889)."""
    b, c, h, w = x.shape
    y = _conv2d(sd, p + ".proj_in", _gn(sd, p + ".norm", x, GN_EPS_ATTN))
    tok = y.flatten(2).transpose(1, 2)
    tok = basic_block(sd, p + ".transformer_blocks.0", tok, context, heads)
    y = tok.transpose(1, 2).reshape(b, c, h, w)
    return _conv2d(sd, p + ".proj_out", y, add=[x])


def spatial_transformer3d(sd: SD, p: str, x5, context, heads: int):
    """SpatialTransformer3D.forward with disable_temporal_text_ca=True (attention.py:1141-1208)."""
    b, c, t, h, w = x5.shape
    x = _to_frames(x5)
    ctx = context.repeat_interleave(t, dim=0)        # 'b l c -> (b t) l c'
    x = spatial_transformer2d(sd, p, x, ctx, heads)  # same keys: norm, proj_in, transformer_blocks, proj_out
    xp = _frames_to_pix(x, b)                        # (bhw, c, t)
    y = _conv1d(sd, p + ".proj_in_temporal", _gn(sd, p + ".norm_temporal", xp, GN_EPS_ATTN))
    tok = y.transpose(1, 2)                          # (bhw, t, c)
    tok = single_block(sd, p + ".transformer_blocks_temporal.0", tok, tok, heads)
    return _pix_to_5d(_conv1d(sd, p + ".proj_out_temporal", tok.transpose(1, 2), add=[xp]), b, h, w)


def spatial_transformer2d_selfonly(sd: SD, p: str, x, heads: int):
    """SpatialTransformer with disable_text_ca=True (attention.py:820-838, 880-881): one
    BasicTransformerSingleLayerBlock called as block(x, context=x)."""
    b, c, h, w = x.shape
    y = _conv2d(sd, p + ".proj_in", _gn(sd, p + ".norm", x, GN_EPS_ATTN))
    tok = y.flatten(2).transpose(1, 2)
    tok = single_block(sd, p + ".transformer_blocks.0", tok, tok, heads)
    y = tok.transpose(1, 2).reshape(b, c, h, w)
    return _conv2d(sd, p + ".proj_out", y, add=[x])


def spatial_transformer3dca(sd: SD, p: str, x5, context, heads: int):
    """SpatialTransformer3DCA.forward, ST3DCA_ca_type='center_self' (attention.py:1302-1350): after the 3D
    transformer, every frame attends to [tokens of the centre frame T//2 ; its own tokens] (un-normalised K/V)."""
    x5 = spatial_transformer3d(sd, p, x5, context, heads)
    b, c, t, h, w = x5.shape
    x = _to_frames(x5)
    y = _conv2d(sd, p + ".proj_in_temporal_ca", _gn(sd, p + ".norm_temporal_ca", x, GN_EPS_ATTN))
    tok = y.flatten(2).transpose(1, 2)                                   # (bt, hw, c)
    anchor = tok.reshape(b, t, h * w, c)[:, t // 2].repeat_interleave(t, dim=0)
    ctx = torch.cat([anchor, tok], dim=1)
    tok = single_block(sd, p + ".transformer_blocks_temporal_ca.0", tok, ctx, heads)
    y = _conv2d(sd, p + ".proj_out_temporal_ca", tok.transpose(1, 2).reshape(b * t, c, h, w), add=[x])
    return y.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4).contiguous()


def controlnet2d_img_forward(sd: SD, p: str, cfg: NetConfig, hint4, t, trace=None):
    """ControlNet2D.forward of `controlnet_img` (controlmodel.py:252-317 with no_add_x=True,
    set_input_hint_block_as_identity=True, disable_text_ca=True) on the 4-D reference latent `cond_feat`:
    the noisy x is ignored, the first block's output IS input_blocks[0](hint)."""
    emb = time_embed(sd, p + ".time_embed", t, cfg.model_channels)
    inputs, _, _ = unet_topology(cfg)
    heads = cfg.num_heads
    outs = []
    h = None
    hint4 = _R(hint4)
    for i, spec in enumerate(inputs):
        bp = f"{p}.input_blocks.{i}"
        if spec.kind == "conv_in":
            h = _conv2d(sd, bp + ".0", hint4, padding=1)
        elif spec.kind == "res":
            h = resblock2d(sd, bp + ".0", h, emb)
            if spec.attn:
                h = spatial_transformer2d_selfonly(sd, bp + ".1", h, heads)
        else:
            h = _conv2d(sd, bp + ".0.op", h, stride=2, padding=1)
        outs.append(_conv2d(sd, f"{p}.zero_convs.{i}.0", h) * cfg.control_scales)
        if trace is not None:
            trace[bp] = h
    mp = p + ".middle_block"
    h = resblock2d(sd, mp + ".0", h, emb)
    h = spatial_transformer2d_selfonly(sd, mp + ".1", h, heads)
    h = resblock2d(sd, mp + ".2", h, emb)
    if trace is not None:
        trace[mp] = h
    outs.append(_conv2d(sd, p + ".middle_block_out.0", h) * cfg.control_scales)
    return outs


# ----------------------------------------------------------------------------------------
# ControlNet2D
# ----------------------------------------------------------------------------------------
_HINT_STRIDES = (1, 1, 2, 1, 2, 1, 2, 1)             # controlmodel.py:215-231


def hint_stem(sd: SD, p: str, hint):
    """input_hint_block: 8 conv3x3, SiLU between them (none after the last)."""
    h = hint
    for i, s in enumerate(_HINT_STRIDES):
        h = _conv2d(sd, f"{p}.{2 * i}", h, stride=s, padding=1, act=F.silu if i != len(_HINT_STRIDES) - 1 else None)
    return h


def controlnet2d_forward(sd: SD, p: str, cfg: NetConfig, x5, hint5, t, context, trace=None):
    """ControlNet2D.forward on a 5-D clip (controlmodel.py:252-317) -> 13 residuals (b c t h w)."""
    b, _, nt, _, _ = x5.shape
    emb = time_embed(sd, p + ".time_embed", t, cfg.model_channels).repeat_interleave(nt, dim=0)
    ctx = context.repeat_interleave(nt, dim=0)
    x, hint = _to_frames(x5), _to_frames(hint5)
    guided = hint_stem(sd, p + ".input_hint_block", hint)
    inputs, _, _ = unet_topology(cfg)
    heads = cfg.num_heads
    outs = []
    h = x
    for i, spec in enumerate(inputs):
        bp = f"{p}.input_blocks.{i}"
        if trace is not None:
            trace[bp + ":in"] = h
        if spec.kind == "conv_in":
            h = _conv2d(sd, bp + ".0", h, padding=1, add=[guided])
        elif spec.kind == "res":
            h = resblock2d(sd, bp + ".0", h, emb)
            if spec.attn:
                h = spatial_transformer2d(sd, bp + ".1", h, ctx, heads)
        else:
            h = _conv2d(sd, bp + ".0.op", h, stride=2, padding=1)
        outs.append(_conv2d(sd, f"{p}.zero_convs.{i}.0", h))
        if trace is not None:
            trace[bp] = h
    mp = p + ".middle_block"
    if trace is not None:
        trace[mp + ":in"] = h
    h = resblock2d(sd, mp + ".0", h, emb)
    h = spatial_transformer2d(sd, mp + ".1", h, ctx, heads)
    h = resblock2d(sd, mp + ".2", h, emb)
    if trace is not None:
        trace[mp] = h
    outs.append(_conv2d(sd, p + ".middle_block_out.0", h))
    res = []
    for o in outs:
        o = o * cfg.control_scales
        bt, c, hh, ww = o.shape
        res.append(o.reshape(b, nt, c, hh, ww).permute(0, 2, 1, 3, 4).contiguous())
    return res


# ----------------------------------------------------------------------------------------
# pseudo-3D UNet
# ----------------------------------------------------------------------------------------
def unet3d_forward(sd: SD, p: str, cfg: NetConfig, x5, t, context, control: List[torch.Tensor], trace=None,
                   img_control: Optional[List[torch.Tensor]] = None):
    """ControlledUNetModel3DTV2V.forward (controlmodel.py:471-550).  img_control (TVI2V): 13 (B,C,h,w) residuals
    added IN PLACE to the centre frame after every input block and after the middle block (:529-535)."""
    control = list(control)
    img_control = None if img_control is None else list(img_control)
    st3d = spatial_transformer3dca if cfg.crossframe else spatial_transformer3d

    def add_center(hh):
        if img_control is None:
            return hh
        hh = hh.clone()
        hh[:, :, hh.shape[2] // 2] = _R(hh[:, :, hh.shape[2] // 2] + img_control.pop(0))
        return hh

    def t3(key):
        return lambda x, add: _conv1d(sd, key, x, padding=1, add=add)

    emb = time_embed(sd, p + ".time_embed", t, cfg.model_channels)
    inputs, _, outputs = unet_topology(cfg)
    heads = cfg.num_heads
    hs = []
    h = x5
    for i, spec in enumerate(inputs):
        bp = f"{p}.input_blocks.{i}"
        if trace is not None:
            trace[bp + ":in"] = h
        if spec.kind == "conv_in":
            h = stf(h, lambda x: _conv2d(sd, bp + ".0", x, padding=1), t3(p + ".input_blocks_temporal.0"))
        elif spec.kind == "res":
            h = resblock3d(sd, bp + ".0", h, emb)
            if spec.attn:
                h = st3d(sd, bp + ".1", h, context, heads)
        else:   # Downsample3D (openaimodel.py:388-394)
            h = stf(h, lambda x: _conv2d(sd, bp + ".0.op", x, stride=2, padding=1), t3(bp + ".0.conv_temporal"))
        h = add_center(h)
        hs.append(h)
        if trace is not None:
            trace[bp] = h
    mp = p + ".middle_block"
    if trace is not None:
        trace[mp + ":in"] = h
    h = resblock3d(sd, mp + ".0", h, emb)
    h = st3d(sd, mp + ".1", h, context, heads)
    h = resblock3d(sd, mp + ".2", h, emb)
    if trace is not None:
        trace[mp + ":pre"] = h          # before img_control / control are added
    h = add_center(h)
    h = _R(h + control.pop())
    if trace is not None:
        trace[mp] = h
    for i, spec in enumerate(outputs):
        bp = f"{p}.output_blocks.{i}"
        h = torch.cat([h, _R(hs.pop() + control.pop())], dim=1)
        if trace is not None:
            trace[bp + ":in"] = h
        h = resblock3d(sd, bp + ".0", h, emb)
        j = 1
        if spec.attn:
            h = st3d(sd, f"{bp}.{j}", h, context, heads)
            j += 1
        if spec.up:   # Upsample3D (openaimodel.py:254-263): nearest x(1,2,2) then conv3x3 + conv1d
            up = F.interpolate(h, scale_factor=(1, 2, 2), mode="nearest")
            h = stf(up, lambda x: _conv2d(sd, f"{bp}.{j}.conv", x, padding=1), t3(f"{bp}.{j}.conv_temporal"))
        if trace is not None:
            trace[bp] = h
    # the prediction itself leaves the HIP path as fp32 (out_temporal's epilogue writes floats)
    return stf(h, lambda x: _conv2d(sd, p + ".out.2", _gn(sd, p + ".out.0", x, GN_EPS_RES, silu=True), padding=1),
               lambda x, add: _conv1d(sd, p + ".out_temporal.1", _R(F.silu(x)), padding=1, add=add, rnd=False))


def network_forward(sd: SD, cfg: NetConfig, x5, t, c: Dict[str, torch.Tensor],
                    p: str = "model.diffusion_model", trace=None):
    """OpenAIWrapperControlLDM3DTV2V.forward (wrappers.py:156-207)."""
    hint = _R(1.0 - (c["control_hint"] + 1.0) / 2.0)
    x5, ctx = _R(x5), _R(c["crossattn"])
    control = controlnet2d_forward(sd, p + ".controlnet", cfg, x5, hint, t, ctx, trace)
    img_control = None
    if c.get("cond_feat") is not None:      # TVI2V: controlnet_img on the reference latent (wrappers.py:176-190)
        img_control = controlnet2d_img_forward(sd, p + ".controlnet_img", cfg, c["cond_feat"], t, trace)
    return unet3d_forward(sd, p, cfg, x5, t, ctx, control, trace, img_control)


# ----------------------------------------------------------------------------------------
# sampling numerics
# ----------------------------------------------------------------------------------------
def ddpm_alphas_cumprod(num_timesteps=1000, linear_start=0.00085, linear_end=0.0120) -> np.ndarray:
    """make_beta_schedule('linear') + cumprod, float64 (util.py:24-37, discretizer.py:42-55)."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_timesteps, dtype=torch.float64) ** 2).numpy()
    return np.cumprod(1.0 - betas, axis=0)


def legacy_ddpm_sigmas(n: int, num_timesteps: int = 1000) -> torch.Tensor:
    """LegacyDDPMDiscretization.get_sigmas (discretizer.py:58-69): DEscending f32 (no zero appended)."""
    ac = ddpm_alphas_cumprod(num_timesteps)
    if n < num_timesteps:
        ts = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        ac = ac[ts]
    elif n != num_timesteps:
        raise ValueError
    sig = torch.tensor((1 - ac) / ac, dtype=torch.float32) ** 0.5   # cast to f32 BEFORE the sqrt
    return torch.flip(sig, (0,))


def sampler_sigmas(n: int) -> torch.Tensor:
    """Discretization.__call__(n) with do_append_zero=True (discretizer.py:17-21)."""
    s = legacy_ddpm_sigmas(n)
    return torch.cat([s, s.new_zeros([1])])


def denoiser_sigmas(num_idx: int = 1000) -> torch.Tensor:
    """DiscreteDenoiser buffer: flip=True, no zero -> ascending (denoiser.py:44-59)."""
    return torch.flip(legacy_ddpm_sigmas(num_idx), (0,))


def sigma_to_idx(table: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    """argmin |sigma - table| (denoiser.py:61-63), int64."""
    return (sigma - table[:, None]).abs().argmin(dim=0).view(sigma.shape)


def discrete_denoise(network: Callable, table: torch.Tensor, x, sigma, cond):
    """DiscreteDenoiser.__call__ with EpsScaling (denoiser.py:22-40; denoiser_scaling.py:16-22)."""
    sigma = table[sigma_to_idx(table, sigma)]
    shape = sigma.shape
    s = sigma[(...,) + (None,) * (x.ndim - sigma.ndim)]
    c_in = 1 / (s ** 2 + 1.0) ** 0.5
    idx = sigma_to_idx(table, sigma.reshape(shape))
    return network(x * c_in, idx, cond) * (-s) + x


_CFG_CAT_KEYS = ("vector", "crossattn", "concat", "cond_feat", "control_hint",
                 "interpolate_first", "interpolate_last", "interpolate_first_last")


def cfg_prepare(x, s, c, uc):
    """VanillaCFGTV2V.prepare_inputs (guiders.py:57-67): uc FIRST."""
    out = {}
    for k in c:
        if k in _CFG_CAT_KEYS:
            out[k] = torch.cat((uc[k], c[k]), 0)
        else:
            assert c[k] == uc[k]
            out[k] = c[k]
    return torch.cat([x] * 2), torch.cat([s] * 2), out


def cfg_combine(x, scale: float):
    """VanillaCFG.__call__ + NoDynamicThresholding (guiders.py:25-29; sampling_utils.py:7-9)."""
    u, c = x.chunk(2)
    return u + scale * (c - u)


def ancestral_step_sigmas(sigma_from, sigma_to, eta=1.0):
    """get_ancestral_step (sampling_utils.py:27-36)."""
    up = torch.minimum(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    down = (sigma_to ** 2 - up ** 2) ** 0.5
    return down, up


def _bc(v, x):
    return v[(...,) + (None,) * (x.ndim - v.ndim)]


def dpmpp2s_ancestral_sample(denoiser: Callable, x, cond, uc, num_steps: int, scale: float,
                             noise_fn: Callable[[torch.Tensor], torch.Tensor],
                             eta: float = 1.0, s_noise: float = 1.0, trace: Optional[dict] = None):
    """AncestralSampler.__call__ + DPMPP2SAncestralSampler.sampler_step (sampling.py:44-55,
    190-205, 370-407).  `denoiser(x, sigma, c)` is the closure of sampling_tv2v.py:366-369;
    `noise_fn(x)` stands for torch.randn_like (drawn once per step, including the last)."""
    sigmas = sampler_sigmas(num_steps)
    x = x * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    s_in = x.new_ones([x.shape[0]])

    def denoise(xx, sig):
        d = denoiser(*cfg_prepare(xx, sig, cond, uc))
        return cfg_combine(d, scale)

    for i in range(num_steps):
        sigma, nxt = s_in * sigmas[i], s_in * sigmas[i + 1]
        down, up = ancestral_step_sigmas(sigma, nxt, eta)
        den = denoise(x, sigma)
        x_euler = x + (x - den) / _bc(sigma, x) * _bc(down - sigma, x)
        if torch.sum(down) < 1e-14:
            x = x_euler
        else:
            t, t_next = -sigma.log(), -down.log()
            h = t_next - t
            s = t + 0.5 * h
            m1 = (-s).exp() / (-t).exp()
            m2 = (-0.5 * h).expm1()
            m3 = (-t_next).exp() / (-t).exp()
            m4 = (-h).expm1()
            x2 = _bc(m1, x) * x - _bc(m2, x) * den
            den2 = denoise(x2, (-s).exp())
            x_2s = _bc(m3, x) * x - _bc(m4, x) * den2
            x = torch.where(_bc(down, x) > 0.0, x_2s, x_euler)
        x = torch.where(_bc(nxt, x) > 0.0, x + noise_fn(x) * s_noise * _bc(up, x), x)
        if trace is not None:
            trace.setdefault("x", []).append(x.clone())
    return x


# ----------------------------------------------------------------------------------------
# AutoencoderKL decode
# ----------------------------------------------------------------------------------------
@dataclass
class VAEConfig:
    """ddconfig of keyframe_no2ndca_depthmidas.yaml:83-95."""
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    out_ch: int = 3
    embed_dim: int = 4


def _vae_resblock(sd: SD, p: str, x):
    """model.py ResnetBlock.forward (:131-151), temb=None."""
    h = _conv2d(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, GN_EPS_ATTN)), padding=1)
    h = _conv2d(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, GN_EPS_ATTN)), padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = _conv2d(sd, p + ".nin_shortcut", x)
    return x + h


def _vae_attn(sd: SD, p: str, x):
    """model.py AttnBlock (:161-201): single head, d = C, scale C^-0.5."""
    b, c, h, w = x.shape
    y = _gn(sd, p + ".norm", x, GN_EPS_ATTN)
    q, k, v = (_conv2d(sd, p + "." + n, y).flatten(2).transpose(1, 2) for n in ("q", "k", "v"))
    a = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    return x + _conv2d(sd, p + ".proj_out", a.transpose(1, 2).reshape(b, c, h, w))


def vae_decode(sd: SD, p: str, cfg: VAEConfig, z, scale_factor: float = 0.18215):
    """decode_first_stage (diffusion.py:151-156) -> AutoencoderKLInferenceWrapper.decode
    (autoencoder.py:334-343) -> Decoder.forward (model.py:728-761).  z is 5-D (b c t h w)."""
    b, c, t, h, w = z.shape
    x = _to_frames(z * (1.0 / scale_factor))
    x = _conv2d(sd, p + ".post_quant_conv", x)
    d = p + ".decoder"
    x = _conv2d(sd, d + ".conv_in", x, padding=1)
    x = _vae_resblock(sd, d + ".mid.block_1", x)
    x = _vae_attn(sd, d + ".mid.attn_1", x)
    x = _vae_resblock(sd, d + ".mid.block_2", x)
    for lvl in reversed(range(len(cfg.ch_mult))):
        for i in range(cfg.num_res_blocks + 1):
            x = _vae_resblock(sd, f"{d}.up.{lvl}.block.{i}", x)
        if lvl != 0:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv2d(sd, f"{d}.up.{lvl}.upsample.conv", x, padding=1)
    x = _conv2d(sd, d + ".conv_out", F.silu(_gn(sd, d + ".norm_out", x, GN_EPS_ATTN)), padding=1)
    return x.reshape(b, t, x.shape[1], x.shape[2], x.shape[3]).permute(0, 2, 1, 3, 4).contiguous()


def vae_encode_moments(sd: SD, p: str, cfg: VAEConfig, x4):
    """AutoencoderKL.encode up to the moments (autoencoder.py:306-314): Encoder.forward (model.py:587-614) and
    quant_conv.  x4: (n, 3, H, W) in [-1, 1] -> (n, 2*z_channels, H/8, W/8) = [mean | logvar].
    Downsample (model.py:74-93): zero pad right/bottom by one, conv3x3 stride 2 without padding."""
    e = p + ".encoder"
    h = _conv2d(sd, e + ".conv_in", x4, padding=1)
    nres = len(cfg.ch_mult)
    for lvl in range(nres):
        for i in range(cfg.num_res_blocks):
            h = _vae_resblock(sd, f"{e}.down.{lvl}.block.{i}", h)
        if lvl != nres - 1:
            h = _conv2d(sd, f"{e}.down.{lvl}.downsample.conv", F.pad(h, (0, 1, 0, 1)), stride=2)
    h = _vae_resblock(sd, e + ".mid.block_1", h)
    h = _vae_attn(sd, e + ".mid.attn_1", h)
    h = _vae_resblock(sd, e + ".mid.block_2", h)
    h = _conv2d(sd, e + ".conv_out", F.silu(_gn(sd, e + ".norm_out", h, GN_EPS_ATTN)), padding=1)
    return _conv2d(sd, p + ".quant_conv", h)


def gaussian_sample(moments, noise):
    """DiagonalGaussianDistribution.__init__ / .sample (distributions.py:24-41): logvar clamped to [-30, 20],
    x = mean + exp(0.5 logvar) * noise.  The reference draws `noise = torch.randn(mean.shape)` from the CPU
    global generator; here it is an explicit argument."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def vae_encode(sd: SD, p: str, cfg: VAEConfig, x, noise, scale_factor: float = 0.18215):
    """encode_first_stage (diffusion.py:158-163) -> AutoencoderKLInferenceWrapper.encode (autoencoder.py:323-332):
    4-D (n c h w) or 5-D (b c t h w) frames in, posterior sample * scale_factor out (same rank).  `noise` has
    the shape the reference draws: (n or b*t, z_channels, h/8, w/8)."""
    video = x.dim() == 5
    x4 = _to_frames(x) if video else x
    z = gaussian_sample(vae_encode_moments(sd, p, cfg, x4), noise)
    if video:
        b, t = x.shape[0], x.shape[2]
        z = z.reshape(b, t, *z.shape[1:]).permute(0, 2, 1, 3, 4).contiguous()
    return scale_factor * z


def img2img_sigmas(sigmas: torch.Tensor, strength: float) -> torch.Tensor:
    """Img2ImgDiscretizationWrapper.__call__ (scripts/demo/streamlit_helpers.py:212-233): keep the
    max(int(strength * len), 1) smallest sigmas (the tail of the descending schedule, trailing 0 included)."""
    s = torch.flip(sigmas, (0,))
    s = s[: max(int(strength * len(s)), 1)]
    return torch.flip(s, (0,))


def sdedit_noised_latent(z, noise, sigmas):
    """sampling_tv2v.py:439-448: noised_z = (z + noise * sigma0) / sqrt(1 + sigma0^2) (DDPM-like scaling)."""
    s0 = sigmas[0]
    return (z + noise * s0) / torch.sqrt(1.0 + s0 ** 2.0)


# ------------------------------------------------------------------------------------------
# CLIP text encoder
# ------------------------------------------------------------------------------------------
@dataclass
class CLIPTextConfig:
    """openai/clip-vit-large-patch14 text tower (FrozenCLIPEmbedder default `version`, encoders/modules.py:365)."""
    vocab_size: int = 49408
    hidden: int = 768
    intermediate: int = 3072
    layers: int = 12
    heads: int = 12
    max_len: int = 77
    eps: float = 1e-5


def clip_text_forward(sd: SD, p: str, cfg: CLIPTextConfig, tokens: torch.Tensor) -> torch.Tensor:
    """`FrozenCLIPEmbedder.forward` with layer="last" (encoders/modules.py:393-413) = last_hidden_state of
    HF `CLIPTextModel`.  The algorithm is a third-party dependency absent from /root/reference — transformers==4.19.1
    (requirements.txt:34), `modeling_clip.py: CLIPTextTransformer.forward`: token + learned position embeddings;
    `layers` x pre-LN blocks [LayerNorm -> causal multi-head self-attention (q scaled by d^-0.5) -> residual ->
    LayerNorm -> fc1 -> quick_gelu (x sigmoid(1.702 x)) -> fc2 -> residual]; final LayerNorm.  Pinned against the
    transformers installed in the authoring container (tests/golden/make_golden.py: gen_clip).
    `p` is the prefix of `text_model.` (e.g. "conditioner.embedders.0.transformer.text_model"); tokens int64 (B, L)."""
    b, l = tokens.shape
    x = sd[p + ".embeddings.token_embedding.weight"][tokens] + sd[p + ".embeddings.position_embedding.weight"][:l]
    mask = torch.full((l, l), float("-inf")).triu(1)
    d = cfg.hidden // cfg.heads
    for i in range(cfg.layers):
        q = f"{p}.encoder.layers.{i}"
        h = F.layer_norm(x, (cfg.hidden,), sd[q + ".layer_norm1.weight"], sd[q + ".layer_norm1.bias"], cfg.eps)
        qq, kk, vv = (_linear(sd, f"{q}.self_attn.{n}_proj", h).view(b, l, cfg.heads, d).transpose(1, 2) for n in "qkv")
        a = torch.softmax((qq * d ** -0.5) @ kk.transpose(-1, -2) + mask, dim=-1) @ vv
        x = x + _linear(sd, q + ".self_attn.out_proj", a.transpose(1, 2).reshape(b, l, cfg.hidden))
        h = F.layer_norm(x, (cfg.hidden,), sd[q + ".layer_norm2.weight"], sd[q + ".layer_norm2.bias"], cfg.eps)
        h = _linear(sd, q + ".mlp.fc1", h)
        x = x + _linear(sd, q + ".mlp.fc2", h * torch.sigmoid(1.702 * h))
    return F.layer_norm(x, (cfg.hidden,), sd[p + ".final_layer_norm.weight"], sd[p + ".final_layer_norm.bias"], cfg.eps)
