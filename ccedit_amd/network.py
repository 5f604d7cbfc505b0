"""The CCEdit denoising network (ControlNet2D + pseudo-3D UNet) on the HIP kernels.

Same classes, constructor arguments, attribute names, state-dict keys and forward signatures as the
reference (sgm/modules/diffusionmodules/controlmodel.py, openaimodel.py, attention.py, wrappers.py),
re-designed around ONE activation layout: frames-outermost channels-last, a bf16 (B*T, H, W, C)
tensor == a row-major [pixels][C] matrix.  Consequences:

  * spatial_temporal_forward's three full-tensor transposes per call (openaimodel.py:129-178, 74 calls
    per UNet forward) disappear: spatial kernels index (frame, y, x), temporal kernels step H*W rows.
  * 'b c h w -> b (h w) c' token views are free: proj_in / q,k,v / FF / proj_out are plain GEMMs on
    the same matrix; 1x1 Conv2d, k=1 Conv1d and Linear are the same kernel.
  * residual adds, the timestep-embedding add, GEGLU gating, SiLU of the hint stem, the nearest-2x
    upsample and the skip concat + control add are epilogues / gather modes of the GEMM kernel.

Only what the shipped inference configs use is implemented:
use_spatial_transformer, transformer_depth 1, use_linear_in_transformer False, conv_resample True,
no scale-shift norm, no resblock_updown, num_classes None, disable_temporal_text_ca True.
Anything else raises NotImplementedError at construction.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import ops, policy
from .hip import ACT_SILU
from .layers import PACK_GENERATION, Conv, Linear, Norm, Slot, pack_tree
from .packing import pack_concat

# Per-block output capture for the parity tests (tests/test_network_gpu.py reads the reference's per-block digests):
# set to a dict and every ControlNet / UNet block stores its output under the reference's module path.  None in production.
TRACE: Optional[dict] = None


def _trace(name: str, t: torch.Tensor):
    if TRACE is not None:
        TRACE[name] = t.detach().clone()


GN_EPS_RES = 1e-5     # reference: normalization() = nn.GroupNorm(32, C)     diffusionmodules/util.py:296-302
GN_EPS_ATTN = 1e-6    # reference: Normalize()                               attention.py:153-156


class Geometry:
    """Clip geometry travelling with the activations: B clips x T frames.  With `shard` (parallel.FrameShard) the
    activations hold only this rank's t = shard.t_local keyframes of every clip; temporal ops then exchange halos /
    statistics / K-V rows with the other ranks (see temporal_gn / temporal_conv3 / run_temporal)."""

    def __init__(self, b: int, t: int, shard=None, rows=None):
        self.b, self.t, self.shard = b, t, shard
        self.rows = rows            # parallel.RowShard: every rank holds 1 / N of the ROWS of all frames (sconv3 / sgn / gathered K / V)

    def half(self) -> "Geometry":
        """The geometry of ONE CFG half of the batch (the shared prefix of the two halves is evaluated once, see `twin`)."""
        return Geometry(self.b // 2, self.t, self.shard, self.rows)


_TWIN_PLANS: Dict[tuple, torch.Tensor] = {}


def twin(x: torch.Tensor) -> torch.Tensor:
    """A tensor computed for one CFG half, materialised for both (uc first, then c: the same rows twice).  One HIP copy kernel that
    reads the source once and writes it twice (ccedit_copy_row_blocks with two destination blocks; ATen's cat kernel moved these
    200 MB at 2.3 TB/s: 87 us per level-0 tensor, four of them per step)."""
    if not (x.is_cuda and x.is_contiguous() and x.dim() >= 2 and (x.shape[-1] * x.element_size()) % 16 == 0):
        return torch.cat([x, x])
    rows, width = x.numel() // x.shape[-1], x.shape[-1]
    key = (rows, str(x.device))
    plan = _TWIN_PLANS.get(key)
    if plan is None:
        if torch.cuda.is_current_stream_capturing():      # (plans are made by the eager evaluation that precedes every capture)
            return torch.cat([x, x])
        plan = _TWIN_PLANS[key] = torch.tensor([[0, 0, rows], [0, rows, rows]], dtype=torch.int64, device=x.device)
    out = torch.empty((2 * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    ops.copy_row_blocks(x.view(rows, width), out.view(2 * rows, width), plan, rows)
    return out


def sconv3(x, pw, geo: Geometry, stride: int = 1, **kw):
    """Conv2d 3x3, pad 1 (stride 1 or 2).  Rows sharded (geo.rows): the neighbour ranks' boundary rows are received into two small
    tensors and the kernel reads them in place where a tap leaves the local rows (CcGemmDesc.vpad = 2) — the slab is not copied."""
    rs = geo.rows
    if rs is None:
        return ops.conv2d(x, pw, stride=stride, **kw)
    if stride == 2 and x.shape[1] % 2:
        raise ValueError(f"row-sharded stride-2 convolution on {x.shape[1]} local rows: the local height must be even at every level "
                         f"(RowShard.check_latent)")
    return ops.conv2d(x, pw, stride=stride, halo=rs.halo_exchange(x, below=stride == 1), **kw)


def sgn(x, norm: "Norm", geo: Geometry, silu: bool):
    """Spatial GroupNorm(32) (+SiLU).  Rows sharded: the (frame, group) sums a producer left on `x` (or a statistics pass) are this
    rank's share — all-reduced over the ranks once per tensor, then applied locally."""
    rs = geo.rows
    if rs is not None:
        n, h, w, c = x.shape
        st = ops.gn_stats_of(x, h * w)
        if st is None:
            st = ops.groupnorm_spatial_stats(x)
        if getattr(x, "_gn_global", None) is not st:        # (the tensor these statistics were reduced into, if they were already)
            ops.set_gn_stats(x, rs.gn_stats(st))
            x._gn_global = st
    return ops.groupnorm_spatial(x, norm.g, norm.b, norm.eps, silu)


FUSE_HALO_STATS = policy.on("fuse_halo_stats")      # rows sharded: GroupNorm statistics ride on the following 3x3 conv's halo exchange


def sgn_conv3(x, norm: "Norm", pw, geo: Geometry, silu: bool = True, **kw):
    """conv3x3(SiLU?(GroupNorm(x))) — the in_layers / out_layers pair of every ResBlock (openaimodel.py:441-449, 483-492) and the `out`
    head.  Unsharded: the two launches as before.  Rows sharded (round 6): ONE exchange carries the raw boundary rows to the
    neighbours and the statistics partials to everybody (RowShard.halo_stats_exchange); the slab and the two received rows are then
    normalised locally with the same totals and the conv reads the normalised rows in place — 64 of the 88 statistics all-reduces of a
    step disappear into halo exchanges that were there anyway."""
    rs = geo.rows
    if rs is None or not FUSE_HALO_STATS:
        return sconv3(sgn(x, norm, geo, silu), pw, geo, **kw)
    n, h, w, c = x.shape
    st = ops.gn_stats_of(x, h * w)
    if st is None:
        st = ops.groupnorm_spatial_stats(x)
    if getattr(x, "_gn_global", None) is st:              # (already the frame's: nothing to add — only the halo rows travel)
        a = ops.groupnorm_spatial(x, norm.g, norm.b, norm.eps, silu)
        return sconv3(a, pw, geo, **kw)
    top, bot, total = rs.halo_stats_exchange(x, st)
    ops.set_gn_stats(x, total.mul_(1.0 / rs.world))       # apply divides by the LOCAL element count: (sum / N) / (count / N)
    x._gn_global = ops.gn_stats_of(x, h * w)
    a = ops.groupnorm_spatial(x, norm.g, norm.b, norm.eps, silu)
    halo = []
    for row in (top, bot):
        if row is None:
            halo.append(None)
            continue
        r4 = row.view(n, 1, w, c)
        ops.set_gn_stats(r4, total * (1.0 / h))            # one row of the slab's h: the same mean / variance through the same kernel
        halo.append(ops.groupnorm_spatial(r4, norm.g, norm.b, norm.eps, silu).view(n, w, c))
    return ops.conv2d(a, pw, halo=tuple(halo), **kw)


def gathered_kv(kv2d, frames: int, geo: Geometry):
    """K | V rows of `frames` frames, (frames * pixels, 2C): with sharded rows the other ranks' rows of every frame are all-gathered
    (the spatial self-attention's keys are the whole frame; queries stay local)."""
    if geo is None or geo.rows is None:
        return kv2d
    c2 = kv2d.shape[-1]
    return geo.rows.gather_rows(kv2d.reshape(frames, -1, c2)).reshape(-1, c2)


def temporal_gn(x, norm: "Norm", geo: Geometry, silu: bool, ext: bool = False):
    """GroupNorm over (C/32 x T) per pixel (+SiLU).  ext=True (sharded only): return the halo-extended buffer
    (B*(t+2), H, W, C) with the normalised local frames in the middle, ready for temporal_conv3."""
    sh = geo.shard
    if sh is None:
        return ops.groupnorm_temporal(x, geo.b, geo.t, norm.g, norm.b, norm.eps, silu)
    st = ops.groupnorm_temporal_stats(x, geo.b, geo.t)
    sh.allreduce(st)                       # sum / sumsq over the ranks holding the other keyframes
    if not ext:
        return ops.groupnorm_temporal_apply(x, st, geo.b, geo.t, sh.t_glob, norm.g, norm.b, norm.eps, silu)
    n, h, w, c = x.shape
    buf = torch.empty((geo.b * (geo.t + 2), h, w, c), dtype=x.dtype, device=x.device)
    ops.groupnorm_temporal_apply(x, st, geo.b, geo.t, sh.t_glob, norm.g, norm.b, norm.eps, silu, out=buf,
                                 dst_frames=geo.t + 2, dst_off=1)
    return buf


def _fill_halos(buf, geo: Geometry):
    """Exchange the boundary frames of a halo-extended buffer with the neighbour ranks (in place)."""
    sh = geo.shard
    v = buf.view(geo.b, geo.t + 2, *buf.shape[1:])
    prev, nxt = sh.halo(v[:, 1].contiguous(), v[:, geo.t].contiguous())
    if prev is not None:
        v[:, 0].copy_(prev)
    if nxt is not None:
        v[:, geo.t + 1].copy_(nxt)
    return buf


def _a2a(geo: Geometry) -> bool:
    return geo.shard is not None and geo.shard.mode == "a2a"


def temporal_conv3(a, pw, geo: Geometry, a_is_ext: bool = False, res_self: bool = False, **kw):
    """Conv1d k=3 over T (+ `a` itself when res_self: the stf identity path).
    Sharded, mode "a2a": transpose to (all T, own pixel block), run the unsharded kernel, transpose back.
    Sharded, mode "halo": `a` is (or is copied into) the halo-extended buffer, one frame is exchanged with each
    neighbour rank, and the GEMM gathers through the extended source (zeros outside the clip)."""
    sh = geo.shard
    if sh is None:
        if res_self:
            kw["res1"] = a.view(-1, a.shape[-1])
        return ops.conv_temporal(a, geo.t, pw, **kw)
    if sh.mode == "a2a":
        assert not a_is_ext and "res1" not in kw, "a2a mode: residuals travel as res_self / res2"
        n, h, w, c = a.shape
        ap = sh.to_pixels(a.view(-1, c), geo.b, h * w)
        kw.pop("gn", None)                     # per-frame statistics cannot be produced in the pixel layout
        res2 = kw.pop("res2", None)
        if res_self:
            kw["res1"] = ap
        if "group_rows" in kw:
            kw["group_rows"] = sh.t_glob * sh.hw_local(h * w)
        o = ops.conv_temporal(ap.view(geo.b * sh.t_glob, 1, -1, c), sh.t_glob, pw, **kw)
        return sh.to_frames(o.view(-1, o.shape[-1]), geo.b, h * w, add=res2).view(n, h, w, -1)
    if not a_is_ext:
        n, h, w, c = a.shape
        if res_self:
            kw["res1"] = a.view(-1, c)
        buf = torch.empty((geo.b * (geo.t + 2), h, w, c), dtype=a.dtype, device=a.device)
        buf.view(geo.b, geo.t + 2, h, w, c)[:, 1:geo.t + 1].copy_(a.view(geo.b, geo.t, h, w, c))
        a = buf
    _fill_halos(a, geo)
    return ops.conv_temporal_sharded(a, geo.b, geo.t, sh.t0, sh.t_glob, pw, **kw)


def temporal_gn_conv3(s, norm: "Norm", pw, geo: Geometry, **kw):
    """The temporal half of an stf pair inside a ResBlock3D: s + Conv1d_T(SiLU(GroupNorm_T(s))) (+ epilogue terms)."""
    sh = geo.shard
    n, h, w, c = s.shape
    if _a2a(geo):                              # one transposition carries both the normalisation and the convolution
        sp = sh.to_pixels(s.view(-1, c), geo.b, h * w)
        sp4 = sp.view(geo.b * sh.t_glob, 1, -1, c)
        at = ops.groupnorm_temporal(sp4, geo.b, sh.t_glob, norm.g, norm.b, norm.eps, True)
        kw.pop("gn", None)
        res2 = kw.pop("res2", None)
        if "group_rows" in kw:
            kw["group_rows"] = sh.t_glob * sp4.shape[2]
        o = ops.conv_temporal(at, sh.t_glob, pw, res1=sp, **kw)
        return sh.to_frames(o.view(-1, c), geo.b, h * w, add=res2).view(n, h, w, c)
    at = temporal_gn(s, norm, geo, True, ext=sh is not None)
    return temporal_conv3(at, pw, geo, a_is_ext=sh is not None, res1=s.view(-1, c), **kw)


def _seq(*mods) -> nn.Sequential:
    return nn.Sequential(*mods)


# ------------------------------------------------------------------------------------------
# attention blocks
# ------------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634


class CrossAttention(nn.Module):
    """Parameters of sgm.modules.attention.CrossAttention (attention.py:365-390)."""

    def __init__(self, query_dim: int, context_dim: Optional[int], heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        cdim = query_dim if context_dim is None else context_dim
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(cdim, inner, bias=False)
        self.to_v = Linear(cdim, inner, bias=False)
        self.to_out = _seq(Linear(inner, query_dim), Slot())
        self.self_attn = context_dim is None
        self.qkv = None
        self.kv = None
        self.q_log2 = False

    def wants_q_log2(self) -> bool:
        return self.self_attn and self.dim_head in (40, 80) and ops.ATTN_Q_LOG2       # the head sizes attn_spatial_kernel serves

    def post_pack(self, device):
        if self.self_attn:
            # d = 40 / 80 (the 64x96 and 32x48 levels): softmax scale * log2(e) rides in the packed to_q rows — the attention kernel
            # then exponentiates q.k directly (CCEDIT_ATTN_Q_LOG2), and q is rounded to bf16 once, as in the reference's
            # `q = self.to_q(x)` (attention.py:404), instead of a second time after an in-kernel multiply
            self.q_log2 = self.wants_q_log2()
            qw = self.to_q.weight * (self.dim_head ** -0.5 * LOG2E) if self.q_log2 else self.to_q.weight
            self.qkv = pack_concat([qw, self.to_k.weight, self.to_v.weight], device=device)
        self.kv = pack_concat([self.to_k.weight, self.to_v.weight], device=device)


def ln_linear(tok, norm: "Norm", pw, pw_ln, stats=None):
    """linear(LayerNorm(tok)) without the normalised tensor ever existing in memory, given a folded weight
    (packing.fold_layernorm): K = 320 — the rows are normalised inside the register-resident-weight GEMM (CcGemmDesc.ln_eps);
    K = 640 / 1280 on the persistent GEMM — the GEMM runs on the raw rows and its epilogue applies their (mean, rstd)
    (CcGemmDesc.ln_stats; `stats` = ops.row_stats(tok) when the caller has them already).  Otherwise a LayerNorm pass."""
    if pw_ln is not None and pw_ln.cin == 320 and ops.ln320_applicable(tok.shape[0], pw_ln):
        return ops.linear(tok, pw_ln, ln_eps=norm.eps)
    if ops.lnf_applicable(tok.shape[0], pw_ln):
        sums = ops.ln_sums_of(tok)          # left by the GEMM that wrote tok (linear_ln_producer), else one read-only pass
        if sums is not None:
            return ops.linear(tok, pw_ln, ln_sums=(sums, norm.eps))
        return ops.linear(tok, pw_ln, ln_stats=ops.row_stats(tok, norm.eps) if stats is None else stats)
    return ops.linear(ops.layernorm(tok, norm.g, norm.b, norm.eps), pw)


def linear_ln_producer(x2d, pw, **kw):
    """A Linear whose output is the input of a LayerNorm that ln_linear folds into the next GEMM: its epilogue also accumulates
    the row sums (CcGemmDesc.row_sums) where that kernel runs."""
    return ops.linear(x2d, pw, row_sums=ops.row_sums_applicable(x2d.shape[0], pw, kw.get("act", 0)), **kw)


def _fold_ln(weights, norm: "Norm", device, biases=None, geglu=False, dims=(320, 640, 1280)):
    """Folded projection of LayerNorm(x) (see ln_linear); None for widths no kernel folds."""
    if weights[0].shape[1] not in dims:
        return None
    from .packing import fold_layernorm
    return fold_layernorm(weights, biases, norm.weight, norm.bias, device=device, geglu=geglu)


class FeedForward(nn.Module):
    """FeedForward(glu=True): net.0 = GEGLU(proj), net.1 = Dropout, net.2 = Linear (attention.py:115-141)."""

    def __init__(self, dim: int, mult: int = 4):
        super().__init__()
        inner = dim * mult
        geglu = nn.Module()
        geglu.proj = Linear(dim, inner * 2, geglu=True)
        self.net = _seq(geglu, Slot(), Linear(inner, dim))

        self.fused = None          # PackedFF320 (dim 320): LayerNorm + GEGLU projection + output projection + residual, one kernel

    def pack_fused(self, norm: "Norm", device):
        """Weight stream of the fused dim-320 kernel; `norm` is the LayerNorm the owning block applies in front of this FF."""
        if self.net[2].cout == 320 and self.net[2].cin == 1280 and self.net[0].proj.cin == 320:
            from .packing import pack_ff320
            p, o = self.net[0].proj, self.net[2]
            self.fused = pack_ff320(p.weight, p.bias, o.weight, o.bias, norm.weight, norm.bias, device=device)
            self.fused_eps = norm.eps

        elif self.net[0].proj.cin in (640, 1280):          # wider levels: LayerNorm folded into the GEGLU projection (ln_linear)
            p = self.net[0].proj
            self.proj_ln = _fold_ln([p.weight], norm, device, biases=[p.bias], geglu=True, dims=(640, 1280))

    proj_ln = None
    _tails = None

    def tail_pack(self, to_out: "Linear", proj: Optional["Conv"], device):
        """Weight stream of the block-tail launch (to_out prologue GEMM [+ proj_out epilogue GEMM] around the fused feed-forward),
        built on first use and kept while the packs it was built from live: the entry holds (and is matched by the identity of) this
        feed-forward's plain pack and the two layers' packed weights, all of which a re-pack of THIS network replaces — packing
        another module (a VAE beside the network: bench.py's clip) leaves it alone.  A captured graph holds its address; graphs are
        keyed on PACK_GENERATION and re-captured after any pack, by which time a stale entry has been replaced here."""
        src = (self.fused, to_out.pw, None if proj is None else proj.pw)
        key = (id(to_out), None if proj is None else id(proj))
        if self._tails is None:
            self._tails = {}
        ent = self._tails.get(key)
        if ent is None or any(a is not b for a, b in zip(ent[0], src)):
            from .packing import pack_ff320_tail
            ent = (src, pack_ff320_tail(self.fused, to_out.weight, to_out.bias, None if proj is None else proj.weight,
                                        None if proj is None else proj.bias, device=device))
            self._tails[key] = ent
        return ent[1]

    def run(self, tok, norm: "Norm"):
        """tok + FF(LayerNorm(tok)) (attention.py:695-716 `x = self.ff(self.norm3(x)) + x`)."""
        if self.fused is not None and ops.FF320 and tok.shape[0] >= 1024:
            return ops.ff320(tok, self.fused, eps=self.fused_eps)
        g = ln_linear(tok, norm, self.net[0].proj.pw, self.proj_ln)
        return ops.linear(g, self.net[2].pw, res1=tok)


def transformer_tail(ff: "FeedForward", norm: "Norm", to_out: "Linear", o, tok, proj: "Conv", x2d, gn_rows: int = 0):
    """What every transformer of the networks ends with:  tok = to_out(o) + tok;  tok = FF(LN(tok)) + tok;  y = proj_out(tok) + x
    (attention.py:695-716 / 758-761 and the proj_out of :865-889 / :1141-1208).  At dim 320 (the 64x96 level) ONE launch
    (csrc/ff320.hip, block tail): each of the two projections alone is a read + residual read + write of the whole activation at
    its HBM roofline.  When the consumer wants the GroupNorm statistics of y from the producer's epilogue (gn_rows), only to_out is
    fused and proj_out stays a launch of its own."""
    m = tok.shape[0]
    if (ops.BLOCK_TAIL and ff.fused is not None and m >= 1024 and proj.k == 1 and proj.cin == 320 and proj.cout == 320
            and to_out.cin == 320 and to_out.cout == 320 and tok.is_contiguous() and x2d.is_contiguous() and o.stride(-1) == 1):
        if gn_rows == 0:
            return ops.ff320(None, ff.tail_pack(to_out, proj, tok.device), eps=ff.fused_eps, a=o, res=tok, res2=x2d)
        t2 = ops.ff320(None, ff.tail_pack(to_out, None, tok.device), eps=ff.fused_eps, a=o, res=tok)
        return ops.linear(t2, proj.pw, res1=x2d, gn_rows=gn_rows)
    tok = linear_ln_producer(o, to_out.pw, res1=tok)
    return ops.linear(ff.run(tok, norm), proj.pw, res1=x2d, gn_rows=gn_rows)


class BasicTransformerBlock(nn.Module):
    """attention.py:598-716: x += attn1(LN(x)); x += attn2(LN(x), text); x += FF(LN(x))."""

    def __init__(self, dim, n_heads, d_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head)
        self.norm1, self.norm2, self.norm3 = Norm(dim, 1e-5), Norm(dim, 1e-5), Norm(dim, 1e-5)

    def run(self, tok, frames: int, hw: int, ctx_kv_src, ctx_len: int, frames_per_clip: int, geo: Optional[Geometry] = None, tail=None,
            shared: bool = False):
        """tail = (proj_out Conv, x2d, gn_rows): also apply the owning transformer's proj_out + residual (transformer_tail).
        shared: `tok` holds ONE CFG half (frames / 2 frames) whose twin is identical — the self-attention, which does not see the
        text, is evaluated once; its result is repeated for the text cross-attention, where the halves part."""
        a1, a2 = self.attn1, self.attn2
        c = a1.inner
        full_frames = frames
        if shared:
            frames //= 2
        qkv = ln_linear(tok, self.norm1, a1.qkv, self.qkv_ln)      # (dim 320, 3 slices: folding the norm into lin320 does not pay)
        if geo is not None and geo.rows is not None and geo.rows.heads_ok(a1.heads):
            # rows sharded, head-parallel: all-to-all q, k, v by head -> whole frames of 8 / N heads here -> all-to-all o back
            rs = geo.rows
            cw = c // rs.world
            q_, k_, v_ = rs.to_heads([(qkv, 0), (qkv, c), (qkv, 2 * c)], frames, hw, cw)
            oh = ops.attention(q_, k_, v_, a1.heads // rs.world, a1.dim_head, batches=frames, lq=hw * rs.world, lk=hw * rs.world,
                               q_log2=a1.q_log2)
            o = rs.from_heads(oh, frames, hw, cw)
        elif geo is not None and geo.rows is not None:             # rows sharded: local queries against the gathered frame's keys
            kv = gathered_kv(qkv[:, c:], frames, geo)
            o = ops.attention(qkv[:, :c], kv[:, :c], kv[:, c:], a1.heads, a1.dim_head, batches=frames, lq=hw, lk=hw * geo.rows.world,
                              q_log2=a1.q_log2)
        else:
            o = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], a1.heads, a1.dim_head, batches=frames, lq=hw, lk=hw,
                              q_log2=a1.q_log2)
        tok = linear_ln_producer(o, a1.to_out[0].pw, res1=tok)
        if shared:
            tok, frames = twin(tok), full_frames
        q = ln_linear(tok, self.norm2, a2.to_q.pw, self.q2_ln)
        # [B*L, 2C]: once per clip, shared by its T frames; normally a column slice of the network's batched projection (TextKV)
        kv = ctx_kv_src.of(self) if isinstance(ctx_kv_src, TextKV) else ops.linear(ctx_kv_src, a2.kv)
        o = ops.attention(q, kv[:, :c], kv[:, c:], a2.heads, a2.dim_head, batches=frames, lq=hw, lk=ctx_len,
                          kv_div=frames_per_clip)
        if tail is not None:
            return transformer_tail(self.ff, self.norm3, a2.to_out[0], o, tok, *tail)
        tok = linear_ln_producer(o, a2.to_out[0].pw, res1=tok)
        return self.ff.run(tok, self.norm3)

    q2_ln = None
    qkv_ln = None

    def post_pack(self, device):
        self.ff.pack_fused(self.norm3, device)
        self.q2_ln = _fold_ln([self.attn2.to_q.weight], self.norm2, device)
        a1 = self.attn1
        # the folded-norm pack carries the same softmax scale as a1.qkv (asked of the child directly: pack order does not matter):
        # W' = s W diag(gamma), b' = s W beta — the scale commutes with the fold
        qw = a1.to_q.weight * (a1.dim_head ** -0.5 * LOG2E) if a1.wants_q_log2() else a1.to_q.weight
        self.qkv_ln = _fold_ln([qw, a1.to_k.weight, a1.to_v.weight], self.norm1, device, dims=(640, 1280))


class BasicTransformerSingleLayerBlock(nn.Module):
    """attention.py:719-761, called as block(x, context=x): q from LN(x), k/v from the UN-normalised x."""

    def __init__(self, dim, n_heads, d_head):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.norm1, self.norm2 = Norm(dim, 1e-5), Norm(dim, 1e-5)

    def _finish(self, o, tok, tail):
        if tail is not None:
            return transformer_tail(self.ff, self.norm2, self.attn1.to_out[0], o, tok, *tail)
        tok = linear_ln_producer(o, self.attn1.to_out[0].pw, res1=tok)
        return self.ff.run(tok, self.norm2)

    def run_frames(self, tok, frames: int, hw: int, anchor_t: Optional[int] = None, frames_per_clip: int = 1, shard=None, geo=None,
                   tail=None):
        """Per-frame attention with K/V from the un-normalised tokens.  anchor_t is None: plain self-attention
        (controlnet_img's SpatialTransformer, disable_text_ca).  Otherwise the keys are
        [tokens of frame anchor_t of the same clip ; own tokens] — SpatialTransformer3DCA 'center_self'."""
        a = self.attn1
        c = a.inner
        q = ln_linear(tok, self.norm1, a.to_q.pw, self.q_ln)
        kv = ops.linear(tok, a.kv)
        if geo is not None and geo.rows is not None and geo.rows.heads_ok(a.heads):
            # rows sharded, head-parallel: whole frames of this rank's heads, so the anchor keyframe is addressed as in the unsharded call
            rs = geo.rows
            cw, hwf = c // rs.world, hw * rs.world
            q_, k_, v_ = rs.to_heads([(q, 0), (kv, 0), (kv, c)], frames, hw, cw)
            if anchor_t is None:
                oh = ops.attention(q_, k_, v_, a.heads // rs.world, a.dim_head, batches=frames, lq=hwf, lk=hwf)
            else:
                oh = ops.attention(q_, k_, v_, a.heads // rs.world, a.dim_head, batches=frames, lq=hwf, lk=2 * hwf, kv_outer_rows=hwf,
                                   seg1_len=hwf, seg1_div=frames_per_clip, seg1_mul=frames_per_clip, seg1_add=anchor_t)
            o = rs.from_heads(oh, frames, hw, cw)
        elif geo is not None and geo.rows is not None:             # rows sharded: the keys are the whole frame(s), gathered
            kv = gathered_kv(kv, frames, geo)
            hwk = hw * geo.rows.world
            if anchor_t is None:
                o = ops.attention(q, kv[:, :c], kv[:, c:], a.heads, a.dim_head, batches=frames, lq=hw, lk=hwk)
            else:                                                  # the anchor keyframe's rows are rows of the same gathered tensor
                o = ops.attention(q, kv[:, :c], kv[:, c:], a.heads, a.dim_head, batches=frames, lq=hw, lk=2 * hwk,
                                  kv_outer_rows=hwk, seg1_len=hwk, seg1_div=frames_per_clip, seg1_mul=frames_per_clip,
                                  seg1_add=anchor_t)
        elif anchor_t is None:
            o = ops.attention(q, kv[:, :c], kv[:, c:], a.heads, a.dim_head, batches=frames, lq=hw, lk=hw)
        elif shard is not None:
            # keyframes sharded over ranks: `anchor_t` is a GLOBAL frame index; its K/V rows are broadcast by the owning
            # rank and appended as one extra kv frame per clip (frames [frames, frames + B)), which the kernel's leading
            # segment then addresses: clip b = frame // t_local -> kv frame frames + b
            nb = frames // frames_per_clip
            owner = shard.owner_of(anchor_t)
            kv3 = kv.view(nb, frames_per_clip, hw, 2 * c)
            if shard.rank == owner:
                anchor = kv3[:, anchor_t - shard.t0].contiguous()
            else:
                anchor = torch.empty((nb, hw, 2 * c), dtype=kv.dtype, device=kv.device)
            shard.broadcast(anchor, owner)
            kvx = torch.cat([kv.view(frames, hw, 2 * c), anchor]).view(-1, 2 * c)
            o = ops.attention(q, kvx[:, :c], kvx[:, c:], a.heads, a.dim_head, batches=frames, lq=hw, lk=2 * hw,
                              kv_outer_rows=hw, seg1_len=hw, seg1_div=frames_per_clip, seg1_mul=1, seg1_add=frames)
        else:
            o = ops.attention(q, kv[:, :c], kv[:, c:], a.heads, a.dim_head, batches=frames, lq=hw, lk=2 * hw,
                              kv_outer_rows=hw, seg1_len=hw, seg1_div=frames_per_clip, seg1_mul=frames_per_clip,
                              seg1_add=anchor_t)
        return self._finish(o, tok, tail)

    q_ln = None

    def post_pack(self, device):
        self.ff.pack_fused(self.norm2, device)
        self.q_ln = _fold_ln([self.attn1.to_q.weight], self.norm1, device)

    def run_temporal(self, tok, geo: Geometry, hw: int, tail=None):
        a = self.attn1
        c = a.inner
        q = ln_linear(tok, self.norm1, a.to_q.pw, self.q_ln)
        kv = ops.linear(tok, a.kv)
        t = tk = geo.t
        if geo.shard is not None:      # all-gather the K/V rows of the other ranks' keyframes (RCCL over xGMI)
            kv = geo.shard.gather_frames(kv.view(geo.b * t, hw, 2 * c), geo.b).view(-1, 2 * c)
            tk = geo.shard.t_glob
        o = ops.attention(q, kv[:, :c], kv[:, c:], a.heads, a.dim_head, batches=geo.b * hw, lq=t, lk=tk,
                          q_inner=hw, q_outer_rows=t * hw, q_inner_rows=1, q_seq_rows=hw,
                          kv_inner=hw, kv_outer_rows=tk * hw, kv_inner_rows=1, kv_seq_rows=hw)
        return self._finish(o, tok, tail)


class SpatialTransformer(nn.Module):
    """2-D transformer of the ControlNet (attention.py:764-889), use_linear=False, depth 1."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, context_dim=None, **kw):
        super().__init__()
        if depth != 1 or kw.get("use_linear", False) or kw.get("disable_self_attn", False):
            raise NotImplementedError("SpatialTransformer: only depth=1, conv projections")
        inner = n_heads * d_head
        self.disable_text_ca = bool(kw.get("disable_text_ca", False))
        self.norm = Norm(in_channels, GN_EPS_ATTN)
        self.proj_in = Conv(in_channels, inner, 1)
        if self.disable_text_ca:      # attention.py:820-838: single-layer block, called as block(x, context=x)
            self.transformer_blocks = nn.ModuleList([BasicTransformerSingleLayerBlock(inner, n_heads, d_head)])
        else:
            self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, context_dim)])
        self.proj_out = Conv(inner, in_channels, 1)

    def run_spatial(self, x, ctx2d, ctx_len, frames_per_clip, gn: bool = False, geo: Optional[Geometry] = None, shared: bool = False):
        """shared: x holds ONE of two identical CFG halves; the result is the full batch (the halves part at the text attention)."""
        n, h, w, c = x.shape
        a = sgn(x, self.norm, geo, False) if geo is not None else ops.groupnorm_spatial(x, self.norm.g, self.norm.b, self.norm.eps, False)
        tok = linear_ln_producer(a.view(-1, c), self.proj_in.pw)
        if shared:
            if self.disable_text_ca:
                raise ValueError("a transformer without text attention has no point where the CFG halves part")
            x, n = twin(x), 2 * n
        tail = (self.proj_out, x.view(-1, c), h * w if (gn and self.gn_out) else 0)
        if self.disable_text_ca:
            y = self.transformer_blocks[0].run_frames(tok, n, h * w, geo=geo, tail=tail)
        else:
            y = self.transformer_blocks[0].run(tok, n, h * w, ctx2d, ctx_len, frames_per_clip, geo=geo, tail=tail, shared=shared)
        return ops.carry_gn_stats(y, y.view(n, h, w, c))

    # Does the consumer of this transformer's output read GroupNorm statistics from the producer's epilogue?  True unless the owning
    # network knows better (UNetModel._mark_gn_consumers: a Downsample or the decoder's concatenation follows) — then the statistics
    # are not accumulated, and at dim 320 proj_out can ride in the block-tail launch.
    gn_out = True

    def run(self, x, geo, ctx2d, ctx_len, shared: bool = False):
        return self.run_spatial(x, ctx2d, ctx_len, geo.t, gn=True, geo=geo, shared=shared)      # the next block opens with a GroupNorm


class SpatialTransformer3D(SpatialTransformer):
    """attention.py:1000-1208 with disable_temporal_text_ca=True: spatial block, then temporal
    self-attention over the T keyframes at every pixel."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, context_dim=None, **kw):
        if not kw.pop("disable_temporal_text_ca", False):
            raise NotImplementedError("SpatialTransformer3D: only disable_temporal_text_ca=True (shipped configs)")
        super().__init__(in_channels, n_heads, d_head, depth, context_dim, **kw)
        inner = n_heads * d_head
        self.norm_temporal = Norm(in_channels, GN_EPS_ATTN)
        self.proj_in_temporal = Conv(in_channels, inner, 1, dims=1)
        self.transformer_blocks_temporal = nn.ModuleList([BasicTransformerSingleLayerBlock(inner, n_heads, d_head)])
        self.proj_out_temporal = Conv(inner, in_channels, 1, dims=1)

    def run(self, x, geo, ctx2d, ctx_len, shared: bool = False):
        y = self.run_spatial(x, ctx2d, ctx_len, geo.t, geo=geo, shared=shared)
        n, h, w, c = y.shape
        if _a2a(geo):       # the whole temporal branch (GroupNorm_T, projections, attention over T, FF) is per pixel:
            sh = geo.shard  # it runs on all T frames of this rank's pixel block, between two all-to-alls
            yp = sh.to_pixels(y.view(-1, c), geo.b, h * w)
            hw_me = sh.hw_local(h * w)
            nt = self.norm_temporal
            a = ops.groupnorm_temporal(yp.view(geo.b * sh.t_glob, 1, hw_me, c), geo.b, sh.t_glob, nt.g, nt.b, nt.eps, False)
            tok = ops.linear(a.view(-1, c), self.proj_in_temporal.pw)
            tok = self.transformer_blocks_temporal[0].run_temporal(tok, Geometry(geo.b, sh.t_glob), hw_me)
            z = ops.linear(tok, self.proj_out_temporal.pw, res1=yp)
            return sh.to_frames(z, geo.b, h * w).view(n, h, w, c)
        a = temporal_gn(y, self.norm_temporal, geo, False)
        tok = ops.linear(a.view(-1, c), self.proj_in_temporal.pw)
        z = self.transformer_blocks_temporal[0].run_temporal(tok, geo, h * w, tail=(self.proj_out_temporal, y.view(-1, c),
                                                                                   h * w if self.gn_out_3d() else 0))
        return ops.carry_gn_stats(z, z.view(n, h, w, c))

    def gn_out_3d(self) -> bool:
        return self.gn_out


class SpatialTransformer3DCA(SpatialTransformer3D):
    """attention.py:1211-1350 (TVI2V): after the 3-D transformer, a cross-frame attention in which every frame
    attends to [centre-frame tokens ; own tokens] (ST3DCA_ca_type='center_self').  The anchor keys are rows of the
    same K/V matrix (frame T//2 of the clip): the attention kernel reads them as a leading KV segment, nothing
    is concatenated or repeated in memory."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, context_dim=None, **kw):
        ca_type = kw.pop("ST3DCA_ca_type", "center")
        if ca_type != "center_self":
            raise NotImplementedError(f"ST3DCA_ca_type={ca_type!r}: the shipped TVI2V config uses 'center_self'")
        super().__init__(in_channels, n_heads, d_head, depth, context_dim, **kw)
        inner = n_heads * d_head
        self.norm_temporal_ca = Norm(in_channels, GN_EPS_ATTN)
        self.proj_in_temporal_ca = Conv(in_channels, inner, 1)
        self.transformer_blocks_temporal_ca = nn.ModuleList([BasicTransformerSingleLayerBlock(inner, n_heads, d_head)])
        self.proj_out_temporal_ca = Conv(inner, in_channels, 1)

    def gn_out_3d(self) -> bool:
        return True            # the 3-D transformer's output feeds this class's own GroupNorm (norm_temporal_ca): statistics wanted

    def run(self, x, geo, ctx2d, ctx_len, shared: bool = False):
        y = super().run(x, geo, ctx2d, ctx_len, shared=shared)
        n, h, w, c = y.shape
        nc = self.norm_temporal_ca
        a = sgn(y, nc, geo, False)
        tok = ops.linear(a.view(-1, c), self.proj_in_temporal_ca.pw)
        t_glob = geo.t if geo.shard is None else geo.shard.t_glob
        z = self.transformer_blocks_temporal_ca[0].run_frames(tok, n, h * w, anchor_t=t_glob // 2, frames_per_clip=geo.t,
                                                              shard=geo.shard, geo=geo,
                                                              tail=(self.proj_out_temporal_ca, y.view(-1, c), h * w if self.gn_out else 0))
        return ops.carry_gn_stats(z, z.view(n, h, w, c))


# ------------------------------------------------------------------------------------------
# residual / resampling blocks
# ------------------------------------------------------------------------------------------
class ResBlock(nn.Module):
    """2-D ResBlock of the ControlNet (openaimodel.py:397-554)."""

    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.in_layers = _seq(Norm(channels, GN_EPS_RES), Slot(), Conv(channels, out_channels, 3))
        self.emb_layers = _seq(Slot(), Linear(emb_channels, out_channels))
        self.out_layers = _seq(Norm(out_channels, GN_EPS_RES), Slot(), Slot(), Conv(out_channels, out_channels, 3))
        self.skip_connection = Slot() if out_channels == channels else Conv(channels, out_channels, 1)

    def run(self, x, emb_silu, geo: Geometry):
        n, h, w, _ = x.shape
        e = emb_silu.of(self)                                                      # (B, Cout) fp32 = emb_layers(emb)
        hid = sgn_conv3(x, self.in_layers[0], self.in_layers[2].pw, geo, group_bias=e, group_rows=geo.t * h * w, gn=True)
        skip = x if isinstance(self.skip_connection, Slot) else ops.conv2d(x, self.skip_connection.pw)
        return sgn_conv3(hid, self.out_layers[0], self.out_layers[3].pw, geo, res1=skip.view(-1, skip.shape[-1]), gn=True)


class ResBlock3D(nn.Module):
    """Pseudo-3D ResBlock (openaimodel.py:557-775): every 3x3 conv is followed by a per-pixel
    GroupNorm+SiLU+Conv1d(k3) over T with a residual around it."""

    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        co = out_channels
        self.in_layers = _seq(Norm(channels, GN_EPS_RES), Slot(), Conv(channels, co, 3))
        self.in_layers_temporal = _seq(Norm(co, GN_EPS_RES), Slot(), Conv(co, co, 3, dims=1))
        self.emb_layers = _seq(Slot(), Linear(emb_channels, co))
        self.out_layers = _seq(Norm(co, GN_EPS_RES), Slot(), Slot(), Conv(co, co, 3))
        self.out_layers_temporal = _seq(Norm(co, GN_EPS_RES), Slot(), Slot(), Conv(co, co, 3, dims=1))
        if co == channels:
            self.skip_connection = Slot()
            self.skip_connection_temporal = None
        else:
            self.skip_connection = Conv(channels, co, 1)
            self.skip_connection_temporal = Conv(co, co, 1, dims=1)

    def run(self, x, emb_silu, geo: Geometry):
        n, h, w, _ = x.shape
        s = sgn_conv3(x, self.in_layers[0], self.in_layers[2].pw, geo)
        co = s.shape[-1]
        e = emb_silu.of(self)
        # stf output (s + conv_t) and the `+ emb_out` of openaimodel.py:762 in one epilogue
        hid = temporal_gn_conv3(s, self.in_layers_temporal[0], self.in_layers_temporal[2].pw, geo, group_bias=e,
                                group_rows=geo.t * h * w, gn=True)
        s2 = sgn_conv3(hid, self.out_layers[0], self.out_layers[3].pw, geo)
        if isinstance(self.skip_connection, Slot):
            skip = x
        else:
            k = ops.conv2d(x, self.skip_connection.pw)
            skip = ops.conv_temporal(k, geo.t, self.skip_connection_temporal.pw, res1=k.view(-1, co))
        return temporal_gn_conv3(s2, self.out_layers_temporal[0], self.out_layers_temporal[3].pw, geo,
                                 res2=skip.view(-1, co), gn=True)


class Downsample(nn.Module):
    """openaimodel.py:266-322 (conv_resample): Conv2d 3x3 stride 2."""

    def __init__(self, channels):
        super().__init__()
        self.op = Conv(channels, channels, 3, stride=2)

    def run(self, x, geo):
        return sconv3(x, self.op.pw, geo, stride=2, gn=True)


class Downsample3D(nn.Module):
    """openaimodel.py:325-394: stf(Conv2d s2, Conv1d k3) with identity skip."""

    def __init__(self, channels):
        super().__init__()
        self.op = Conv(channels, channels, 3, stride=2)
        self.conv_temporal = Conv(channels, channels, 3, dims=1)

    def run(self, x, geo):
        s = sconv3(x, self.op.pw, geo, stride=2)
        return temporal_conv3(s, self.conv_temporal.pw, geo, res_self=True, gn=True)


class Upsample3D(nn.Module):
    """openaimodel.py:220-263: nearest x(1,2,2) then stf(Conv2d 3x3, Conv1d k3).  The upsample is a
    gather mode of the conv kernel — the 4x larger tensor is never written."""

    def __init__(self, channels):
        super().__init__()
        self.conv = Conv(channels, channels, 3)
        self.conv_temporal = Conv(channels, channels, 3, dims=1)

    conv_parity = None

    def post_pack(self, device):
        from .packing import pack_upsample_parities
        if ops.SUBPIX and self.conv.cin % 64 == 0:
            self.conv_parity = pack_upsample_parities(self.conv.weight, self.conv.bias, device=device)

    def run(self, x, geo):
        if geo.rows is not None:                # rows sharded: one low-resolution halo row from each neighbour
            if self.conv_parity is None:
                raise NotImplementedError("row-sharded Upsample3D needs the parity form (channels % 64 == 0, CCEDIT_SUBPIX on)")
            s = ops.conv2d_upsampled(x, self.conv_parity, halo=geo.rows.halo_exchange(x))
        elif self.conv_parity is not None:      # four 2 x 2 convolutions on the low-resolution tensor: 4/9 of the multiply-adds
            s = ops.conv2d_upsampled(x, self.conv_parity)
        else:
            s = ops.conv2d(x, self.conv.pw, upsample=True)
        return temporal_conv3(s, self.conv_temporal.pw, geo, res_self=True)


def _splits_at_text(block) -> bool:
    """Does this block end in a transformer whose text cross-attention is where two otherwise identical CFG halves part?"""
    return (len(block) >= 2 and isinstance(block[0], (ResBlock, ResBlock3D)) and isinstance(block[-1], SpatialTransformer)
            and not block[-1].disable_text_ca)


class TimestepEmbedSequential(nn.Sequential):
    """Container with the reference's name (openaimodel.py:85-126); dispatch happens in the nets."""

    def run(self, x, emb_silu, geo, ctx2d, ctx_len, shared: bool = False):
        """shared: x holds ONE of two identical CFG halves (geo is the FULL batch's geometry); the block's transformer returns the
        full batch — the halves part at its text cross-attention.  Only a block that ends in such a transformer can be entered shared."""
        for layer in self:
            if isinstance(layer, (ResBlock, ResBlock3D)):
                x = layer.run(x, emb_silu, geo.half() if shared else geo)
            elif isinstance(layer, SpatialTransformer):
                x = layer.run(x, geo, ctx2d, ctx_len, shared=shared)
                shared = False
            else:
                if shared:
                    raise ValueError("shared evaluation must end at a transformer with text attention")
                x = layer.run(x, geo)
        if shared:
            raise ValueError("shared evaluation must end at a transformer with text attention")
        return x


# ------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------
_UNSUPPORTED_DEFAULTS = dict(dropout=0, conv_resample=True, dims=2, num_classes=None, use_scale_shift_norm=False,
                             resblock_updown=False, use_linear_in_transformer=False, disable_self_attentions=None,
                             num_attention_blocks=None, disable_middle_self_attn=False, adm_in_channels=None,
                             transformer_depth_middle=None, n_embed=None, num_head_channels=-1)


def _check_supported(kw: dict, who: str):
    for k, dflt in _UNSUPPORTED_DEFAULTS.items():
        if k in kw and kw[k] != dflt and not (k == "dims" and kw[k] == 2):
            raise NotImplementedError(f"{who}: option {k}={kw[k]!r} is outside the shipped inference configs")
    if not kw.get("use_spatial_transformer", False):
        raise NotImplementedError(f"{who}: use_spatial_transformer must be True")
    td = kw.get("transformer_depth", 1)
    if (td if isinstance(td, int) else max(td)) != 1:
        raise NotImplementedError(f"{who}: transformer_depth must be 1")


class EmbOut:
    """emb_layers outputs of all ResBlocks of one network evaluation: fp32 (B, sum Cout); `of(block)` = that block's columns.
    The column slice lives ON the block (`_emb_slice`, written by `_pack_emb`): copies of a packed model keep working, and a
    block that was never packed into the batched projection is an error, not a stale lookup."""

    def __init__(self, e_all: torch.Tensor, owner=None):
        self.e_all, self.owner = e_all, owner

    def of(self, block) -> torch.Tensor:
        sl = getattr(block, "_emb_slice", None)
        if sl is None or sl[0] + sl[1] > self.e_all.shape[1]:
            raise RuntimeError("ResBlock is not part of the packed timestep-embedding projection: call pack() after changing modules")
        return self.e_all[:, sl[0]:sl[0] + sl[1]]


class TextKV:
    """Text-conditioning keys / values of one network evaluation.  Every BasicTransformerBlock projects the SAME (B * 77, 768) context
    with its own to_k / to_v (attention.py:392-467; 16 blocks in the UNet, 7 in a ControlNet): they run as ONE GEMM against the
    row-concatenated weights (`UNetModel._pack_text_kv`), `of(block)` is that block's (B * L, 2C) column slice [K | V] — the attention
    kernels take the row stride of the fused buffer.  The projections depend only on the conditioning tensor, which is constant over
    the evaluations of a clip: the wrapper keeps the result per (network, context tensor) while its per-clip caches are on
    (OpenAIWrapperControlLDM3DTV2V._text_kv).  A block outside the packed projection falls back to its own Linear."""

    def __init__(self, ctx2d: torch.Tensor, kv_all: Optional[torch.Tensor] = None):
        self.ctx2d, self.kv_all = ctx2d, kv_all

    def of(self, block) -> torch.Tensor:
        sl = getattr(block, "_tkv_slice", None)
        if self.kv_all is None or sl is None or sl[0] + sl[1] > self.kv_all.shape[1]:
            return ops.linear(self.ctx2d, block.attn2.kv)
        return self.kv_all[:, sl[0]:sl[0] + sl[1]]


TEXT_KV_BATCHED = policy.on("text_kv_batched")      # 0: one K/V projection launch per transformer block (A/B)


class UNetModel(nn.Module):
    """Block wiring of sgm UNetModel.__init__ (openaimodel.py:1033-1527) for the supported options."""

    THREE_D = False

    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 channel_mult=(1, 2, 4, 8), num_heads=-1, context_dim=None, use_checkpoint=False, legacy=True,
                 build_decoder=True, **kw):
        super().__init__()
        _check_supported(dict(kw), type(self).__name__)
        if num_heads == -1:
            raise NotImplementedError("num_head_channels-style head configuration is not used by the shipped configs")
        if isinstance(context_dim, (list, tuple)):
            context_dim = context_dim[0]
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks = len(channel_mult) * [num_res_blocks] if isinstance(num_res_blocks, int) else list(num_res_blocks)
        self.attention_resolutions = list(attention_resolutions)
        self.channel_mult = list(channel_mult)
        self.num_heads, self.context_dim = num_heads, context_dim
        self.num_classes = None
        self.use_checkpoint = use_checkpoint
        res_cls = ResBlock3D if self.THREE_D else ResBlock
        down_cls = Downsample3D if self.THREE_D else Downsample
        if self.THREE_D:
            tkw = dict(disable_temporal_text_ca=kw.get("disable_temporal_text_ca", False))
            st_cls = SpatialTransformer3D
            if kw.get("enable_attention3d_crossframe", False):
                st_cls = SpatialTransformer3DCA
                tkw["ST3DCA_ca_type"] = kw.get("ST3DCA_ca_type", "center")
        else:
            tkw = dict(disable_text_ca=kw.get("disable_text_ca", False))
            st_cls = SpatialTransformer

        def make_st(ch):
            return st_cls(ch, num_heads, ch // num_heads, depth=1, context_dim=context_dim, **tkw)

        ted = model_channels * 4
        self.time_embed = _seq(Linear(model_channels, ted), Slot(), Linear(ted, ted))
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(Conv(in_channels, model_channels, 3))])
        chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks[level]):
                layers = [res_cls(ch, ted, mult * model_channels)]
                ch = mult * model_channels
                if ds in self.attention_resolutions:
                    layers.append(make_st(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                chans.append(ch)
            if level != len(self.channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(down_cls(ch)))
                chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(res_cls(ch, ted, ch), make_st(ch), res_cls(ch, ted, ch))
        self._mid_ch = ch
        if build_decoder:
            self.output_blocks = nn.ModuleList([])
            for level, mult in list(enumerate(self.channel_mult))[::-1]:
                for i in range(self.num_res_blocks[level] + 1):
                    ich = chans.pop()
                    layers = [res_cls(ch + ich, ted, model_channels * mult)]
                    ch = model_channels * mult
                    if ds in self.attention_resolutions:
                        layers.append(make_st(ch))
                    if level and i == self.num_res_blocks[level]:
                        layers.append(Upsample3D(ch))
                        ds //= 2
                    self.output_blocks.append(TimestepEmbedSequential(*layers))
            self.out = _seq(Norm(ch, GN_EPS_RES), Slot(), Conv(model_channels, out_channels, 3))

    # -- shared helpers --
    def _emb_silu(self, timesteps: torch.Tensor) -> "EmbOut":
        """SiLU(time_embed(timestep_embedding(t))): (B, 4*model_channels) bf16.  Every ResBlock applies
        nn.SiLU to emb before its own Linear (openaimodel.py:470-476), so it is hoisted here — and since every one of
        those `emb_layers` Linears (22 in the UNet, 10 in a ControlNet) reads the same (B, 1280) vector, they run as ONE
        GEMM against the row-concatenated weights; a block takes its column slice as the GEMM epilogue's row bias."""
        te = ops.timestep_embedding(timesteps, self.model_channels)
        h = ops.linear(te, self.time_embed[0].pw, act=ACT_SILU)
        e = ops.silu(ops.linear(h, self.time_embed[2].pw))
        return EmbOut(ops.linear(e, self._emb_all, out_f32=True), self)

    def _pack_emb(self, device):
        blocks = [m for m in self.modules() if isinstance(m, (ResBlock, ResBlock3D)) and self._owns(m)]
        off = 0
        for m in blocks:
            c = m.emb_layers[1].cout
            assert c % 8 == 0, "the conv epilogue reads the per-clip bias rows as 16-byte pieces"
            m._emb_slice = (off, c)
            off += c
        self._emb_all = pack_concat([m.emb_layers[1].weight for m in blocks], [m.emb_layers[1].bias for m in blocks], device=device)

    def _owns(self, block) -> bool:
        """ResBlocks of THIS network (a ControlNet nested as `.controlnet` has its own time embedding)."""
        for name, sub in self.named_modules():
            if sub is block:
                return not (name.startswith("controlnet.") or name.startswith("controlnet_img."))
        return False

    def _pack_text_kv(self, device):
        blocks = [m for m in self.modules() if isinstance(m, BasicTransformerBlock) and self._owns(m)]
        off = 0
        for m in blocks:
            m._tkv_slice = (off, 2 * m.attn2.inner)
            off += 2 * m.attn2.inner
        self._text_kv_all = None
        if blocks and TEXT_KV_BATCHED:
            ws = []
            for m in blocks:
                ws += [m.attn2.to_k.weight, m.attn2.to_v.weight]
            self._text_kv_all = pack_concat(ws, device=device)

    def text_kv(self, ctx2d) -> Optional["TextKV"]:
        """One GEMM for the text K / V projections of every transformer block of this network (see TextKV)."""
        if ctx2d is None or isinstance(ctx2d, TextKV):
            return ctx2d
        pw = getattr(self, "_text_kv_all", None)
        return TextKV(ctx2d, None if pw is None else ops.linear(ctx2d, pw))

    def _mark_gn_consumers(self):
        """SpatialTransformer.gn_out: is the transformer's output read next by a GroupNorm that takes its statistics from the
        producer's epilogue?  Encoder: yes iff a ResBlock follows (not a Downsample).  Decoder: the concatenation recomputes them
        (cat_add_gn) and an Upsample is a convolution — only the last block, whose output meets `out.0`, wants them."""
        def first_is_res(seq):
            return isinstance(seq[0], (ResBlock, ResBlock3D))
        blocks = list(self.input_blocks)
        for i, blk in enumerate(blocks):
            if isinstance(blk[-1], SpatialTransformer):
                nxt = blocks[i + 1] if i + 1 < len(blocks) else self.middle_block
                blk[-1].gn_out = first_is_res(nxt)
        outs = list(getattr(self, "output_blocks", []))
        for i, blk in enumerate(outs):
            for j, layer in enumerate(blk):
                if isinstance(layer, SpatialTransformer):
                    layer.gn_out = (i == len(outs) - 1 and j == len(blk) - 1)

    def post_pack(self, device):
        self._pack_emb(device)
        self._pack_text_kv(device)
        self._mark_gn_consumers()

    def pack(self, device=None):
        device = torch.device("cuda") if device is None else device
        pack_tree(self, device)
        return self


class UNetModel3D(UNetModel):
    """openaimodel.py:1581-1639: adds input_blocks_temporal and out_temporal."""

    THREE_D = True

    def __init__(self, *args, temporal_kernel_size=None, **kw):
        if temporal_kernel_size not in (None, 3):
            raise NotImplementedError("temporal_kernel_size != 3")
        kw.pop("unet_type", None)
        super().__init__(*args, **kw)
        mc, oc = self.model_channels, self.out_channels
        self.input_blocks_temporal = TimestepEmbedSequential(Conv(mc, mc, 3, dims=1))
        self.out_temporal = _seq(Slot(), Conv(oc, oc, 3, dims=1))


_HINT_PLAN = ((16, 1), (16, 1), (32, 2), (32, 1), (96, 2), (96, 1), (256, 2))    # controlmodel.py:215-231


class ControlNet2D(UNetModel):
    """Per-frame SD-1.5 encoder copy + hint stem + 13 zero convs (controlmodel.py:195-317)."""

    def __init__(self, hint_channels, control_scales, no_add_x=False, set_input_hint_block_as_identity=False, *args, **kw):
        if no_add_x != set_input_hint_block_as_identity:
            raise NotImplementedError("no_add_x and set_input_hint_block_as_identity are only supported together "
                                      "(the shipped controlnet_img config)")
        kw["out_channels"] = kw["in_channels"]
        super().__init__(*args, build_decoder=False, **kw)
        self.control_scales = float(control_scales)
        self.no_add_x = bool(no_add_x)
        self.set_input_hint_block_as_identity = bool(set_input_hint_block_as_identity)
        mc = self.model_channels
        if self.set_input_hint_block_as_identity:
            self.input_hint_block = TimestepEmbedSequential(Slot())          # nn.Identity(): no parameters
        else:
            mods, cin = [], hint_channels
            for cout, stride in _HINT_PLAN:
                mods += [Conv(cin, cout, 3, stride=stride), Slot()]
                cin = cout
            mods.append(Conv(cin, mc, 3))
            self.input_hint_block = TimestepEmbedSequential(*mods)
        self.zero_convs = nn.ModuleList([TimestepEmbedSequential(Conv(mc, mc, 1))])
        ch = mc
        for level, mult in enumerate(self.channel_mult):
            for _ in range(self.num_res_blocks[level]):
                ch = mult * mc
                self.zero_convs.append(TimestepEmbedSequential(Conv(ch, ch, 1)))
            if level != len(self.channel_mult) - 1:
                self.zero_convs.append(TimestepEmbedSequential(Conv(ch, ch, 1)))
        self.middle_block_out = TimestepEmbedSequential(Conv(ch, ch, 1))

    def post_pack(self, device):
        self._pack_emb(device)
        self._pack_text_kv(device)
        self._mark_gn_consumers()
        if self.control_scales != 1.0:       # `c * scale` (controlmodel.py:311-312) folded into the zero convs
            for zc in list(self.zero_convs) + [self.middle_block_out]:
                zc[0].pack(device, scale=self.control_scales)

    def hint_stem(self, hint_nhwc, rows=None):
        h = hint_nhwc
        geo = Geometry(1, 1, rows=rows)
        convs = [m for m in self.input_hint_block if isinstance(m, Conv)]
        for i, cv in enumerate(convs):
            last = i == len(convs) - 1
            h = sconv3(h, cv.pw, geo, stride=cv.stride, act=0 if last else ACT_SILU)
        return h

    def run(self, x_nhwc, guided, timesteps, ctx2d, ctx_len, geo: Geometry, shared: bool = False) -> List[torch.Tensor]:
        """x_nhwc (B*T, h, w, 8) bf16, guided = hint_stem(remapped hint) (B*T, h, w, C) -> 13 residuals.
        controlnet_img (no_add_x + identity hint block): x_nhwc is ignored and `guided` is the 8-channel-padded
        reference latent; the first block's output is input_blocks[0](guided) (controlmodel.py:283-299).
        shared (see OpenAIWrapperControlLDM3DTV2V._cfg_twins): the two CFG halves of the batch are identical up to the text — x_nhwc
        and guided hold ONE half, and everything up to the first text cross-attention (input_blocks.0, the ResBlock and the
        self-attention of input_blocks.1) is evaluated once; the results are the full batch's."""
        emb_silu = self._emb_silu(timesteps)
        ctx2d = self.text_kv(ctx2d)
        if shared and not _splits_at_text(self.input_blocks[1]):
            raise ValueError("shared CFG prefix: input_blocks.1 must end in a transformer with text attention")
        outs = []
        h = x_nhwc
        for i, (block, zc) in enumerate(zip(self.input_blocks, self.zero_convs)):
            if i == 0 and self.no_add_x:
                h = sconv3(guided, block[0].pw, geo)
            elif i == 0:
                h = sconv3(h, block[0].pw, geo.half() if shared else geo, res1=guided.view(-1, guided.shape[-1]))     # h = conv(x); h += guided_hint
            else:
                h = block.run(h, emb_silu, geo, ctx2d, ctx_len, shared=shared and i == 1)
            if TRACE is not None:          # (controlnet_img — the no_add_x variant — traces under its own module path)
                _trace(f"{'controlnet_img' if self.no_add_x else 'controlnet'}.input_blocks.{i}", h)
                if i == 0 and not self.no_add_x:
                    _trace("controlnet.guided_hint", guided)
            zo = ops.conv2d(h, zc[0].pw)
            outs.append(twin(zo) if (shared and i == 0) else zo)
        h = self.middle_block.run(h, emb_silu, geo, ctx2d, ctx_len)
        _trace(f"{'controlnet_img' if self.no_add_x else 'controlnet'}.middle_block", h)
        outs.append(ops.conv2d(h, self.middle_block_out[0].pw))
        return outs

    def forward(self, x, hint, timesteps=None, context=None, y=None, **kwargs):
        """Reference signature (controlmodel.py:252): 5-D fp32 tensors in, list of 13 (b c t h w) out."""
        assert y is None, "must specify y if and only if the model is class-conditional"
        if x.dim() == 4:       # controlnet_img: (B, C, h, w) reference latent as hint -> 13 x (B, C, h, w)
            if not self.no_add_x:
                raise NotImplementedError("4-D input is only used by the controlnet_img variant")
            b = x.shape[0]
            res = self.run(None, ops.ncthw_to_nhwc(hint.float()[:, :, None].contiguous(), 8), timesteps, None, 0, Geometry(b, 1))
            return [ops.nhwc_to_ncthw(r, b, 1, r.shape[-1])[:, :, 0] for r in res]
        b, _, t, _, _ = x.shape
        geo = Geometry(b, t)
        ctx2d = context.to(torch.bfloat16).reshape(-1, context.shape[-1]).contiguous()
        guided = self.hint_stem(ops.ncthw_to_nhwc(hint.float().contiguous(), 8))
        res = self.run(ops.ncthw_to_nhwc(x.float().contiguous(), 8), guided, timesteps, ctx2d, context.shape[1], geo)
        return [ops.nhwc_to_ncthw(r, b, t, r.shape[-1]) for r in res]


class ControlledUNetModel3DTV2V(UNetModel3D):
    """controlmodel.py:320-553 (TV2V): pseudo-3D UNet that sums ControlNet residuals into its skips."""

    def __init__(self, controlnet_config, *args, **kw):
        if kw.get("crossframe_type") is not None:
            raise NotImplementedError("crossframe_type='reference' (attention hooks) is commented out in the shipped configs")
        controlnet_img_config = kw.pop("controlnet_img_config", None)
        super().__init__(*args, **kw)
        from .config import instantiate_from_config
        self.controlnet = instantiate_from_config(controlnet_config)
        if controlnet_img_config is not None:
            self.controlnet_img = instantiate_from_config(controlnet_img_config)

    def run(self, x_nhwc, timesteps, ctx2d, ctx_len, control: List[torch.Tensor], geo: Geometry,
            img_control: Optional[List[torch.Tensor]] = None, control_ready=None, shared: bool = False):
        """x_nhwc (B*T, h, w, 8) bf16; control = 13 NHWC residuals (consumed); img_control = 13 (B, h, w, C)
        residuals added in place to the centre frame T//2 of every clip (controlmodel.py:529-535)
        -> eps (B*T, h, w, out) fp32."""
        emb_silu = self._emb_silu(timesteps)
        ctx2d = self.text_kv(ctx2d)
        if shared and not _splits_at_text(self.input_blocks[1]):
            raise ValueError("shared CFG prefix: input_blocks.1 must end in a transformer with text attention")

        def add_center(hh):
            if img_control is not None:
                ic = img_control.pop(0)
                centre = geo.t // 2                              # controlmodel.py:529-535: frame T//2 of every clip
                if geo.shard is not None:                        # sharded: only the rank that holds that keyframe adds
                    centre = geo.shard.t_glob // 2 - geo.shard.t0
                    if not 0 <= centre < geo.t:
                        return hh
                for b in range(hh.shape[0] // geo.t):            # (a shared-prefix tensor holds one CFG half; ic then holds one too)
                    fr = hh[b * geo.t + centre]
                    ops.add(fr, ic[b % ic.shape[0]], out=fr)
                if hasattr(hh, "_gn_stats"):
                    del hh._gn_stats                             # modified in place: the producer's statistics are stale
                if hasattr(hh, "_gn_global"):
                    del hh._gn_global
            return hh

        hs = []
        h = x_nhwc
        for i, block in enumerate(self.input_blocks):
            if i == 0:
                g0 = geo.half() if shared else geo
                s = sconv3(h, block[0].pw, g0)
                h = temporal_conv3(s, self.input_blocks_temporal[0].pw, g0, res_self=True)
            else:
                h = block.run(h, emb_silu, geo, ctx2d, ctx_len, shared=shared and i == 1)
            _trace(f"input_blocks.{i}", h)
            h = add_center(h)
            hs.append(twin(h) if (shared and i == 0) else h)     # (the skip connection serves the full batch)
        h = add_center(self.middle_block.run(h, emb_silu, geo, ctx2d, ctx_len))
        if control_ready is not None:      # ControlNet ran on a side stream while the encoder above was running
            torch.cuda.current_stream().wait_event(control_ready)
        h = ops.add(h, control.pop())
        for block in self.output_blocks:
            h = ops.cat_add(h, hs.pop(), control.pop(), gn=True)          # cat([h, hs.pop() + control.pop()], dim=1)
            h = block.run(h, emb_silu, geo, ctx2d, ctx_len)
            _trace(f"output_blocks.{len(self.output_blocks) - len(hs) - 1}", h)
        return self.head(h, geo)

    def head(self, h, geo: Geometry):
        """`out` (GroupNorm + SiLU + conv3x3 -> 4 channels) and `out_temporal` (SiLU + Conv1d k3 over T, residual): the prediction,
        fp32 (controlmodel.py:545-550, openaimodel.py:1627-1632)."""
        n, hh, ww, _ = h.shape
        oc = self.out_channels
        ocp = (oc + 7) // 8 * 8
        s = torch.zeros((n * hh * ww, ocp), dtype=torch.bfloat16, device=h.device)
        sgn_conv3(h, self.out[0], self.out[2].pw, geo, out=s[:, : self.out[2].pw.n])
        if _a2a(geo):       # SiLU + Conv1d_T on this rank's pixel block; the caller all-gathers the pixel blocks
            sh = geo.shard
            sp = sh.to_pixels(s, geo.b, hh * ww)
            at = ops.silu(sp)
            return ops.conv_temporal(at.view(geo.b * sh.t_glob, 1, -1, ocp), sh.t_glob, self.out_temporal[1].pw, res1=sp,
                                     out_f32=True)
        at = ops.silu(s)
        eps = temporal_conv3(at.view(n, hh, ww, ocp), self.out_temporal[1].pw, geo, res1=s, out_f32=True)
        return eps

    def forward(self, x, timesteps=None, context=None, y=None, control=None, img_control=None, only_mid_control=False,
                **kwargs):
        """Reference signature (controlmodel.py:471-481); `control` (5-D fp32 list) is consumed."""
        assert y is None, "must specify y if and only if the model is class-conditional"
        if only_mid_control or control is None:
            raise NotImplementedError("only_mid_control / control=None")
        b, _, t, _, _ = x.shape
        geo = Geometry(b, t)
        ctx2d = context.to(torch.bfloat16).reshape(-1, context.shape[-1]).contiguous()
        ctrl = [ops.ncthw_to_nhwc(c.float().contiguous(), c.shape[1]) for c in control]
        del control[:]
        ictrl = None
        if img_control is not None:      # 13 x (B, C, h, w) -> (B, h, w, C)
            ictrl = [ops.ncthw_to_nhwc(c.float()[:, :, None].contiguous(), c.shape[1]) for c in img_control]
            del img_control[:]
        eps = self.run(ops.ncthw_to_nhwc(x.float().contiguous(), 8), timesteps, ctx2d, context.shape[1], ctrl, geo, ictrl)
        return ops.nhwc_to_ncthw(eps, b, t, self.out_channels)


# ------------------------------------------------------------------------------------------
# wrapper
# ------------------------------------------------------------------------------------------
class IdentityWrapper(nn.Module):
    """wrappers.py:13-25"""

    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


# The two CFG halves (uncond / cond clip of the doubled batch) are independent and CAN be evaluated as two B = 1 passes on two
# HIP streams (each with its own ControlNet side stream), filling each other's launch tails.  Rounds 1-2: -1.5...2.5 % per step
# and the default.  Round 3: the persistent eight-phase GEMM (gemm8p.hip) owns every CU while it runs (128-160 KB of LDS per
# workgroup), so there is little left to overlap, and the batched pass gives it twice the tiles per launch — same-box A/B
# 115.2 (batched) vs 116.3 ms (two streams).  Batched is the default now; CCEDIT_SPLIT_CFG=1 restores the two-stream halves.
_SPLIT_CFG = policy.on("split_cfg")


class OpenAIWrapperControlLDM3DTV2V(IdentityWrapper):
    """wrappers.py:155-207: hint remap -> ControlNet -> UNet.  Stays in the channels-last layout between
    the two networks; only x (4 ch) and the eps output (4 ch) cross the (B, C, T, H, W) boundary."""

    # The hint stem (8 convs at up to 512x768, 0.77 TFLOP) depends only on control_hint, which is constant over
    # the 59 evaluations of a clip; the reference recomputes it every time.  With cache_hint_stem=True the
    # result is kept while the SAME hint tensor (storage, shape, version) is passed again — results are
    # unchanged.  bench.py's per-step metric runs with the cache OFF (every step does the full work).
    cache_hint_stem = True
    _hint_val = None
    _hint_slices = None
    _hint_dup = None
    dedup_hint = policy.on("hint_dedup")      # 0: evaluate the hint stem on both CFG halves (A/B)
    frame_shard = None          # parallel.FrameShard: split the T keyframes of each clip over the ranks (config 4)
    row_shard = None            # parallel.RowShard: split the latent ROWS of every frame over the ranks (config 4, balanced)
    # The ControlNet's residuals are first needed after the UNet's middle block: with overlap_controlnet the ControlNet
    # (16 TFLOP) is launched on a side HIP stream and runs concurrently with the UNet encoder (25 TFLOP) — the two fill
    # each other's launch tails and the small 16x24 / 8x12-level kernels that cannot occupy 256 CUs alone.
    # Round 3: -2.5 ms per step in the batched default (113.0 vs 115.5 ms, same box).  Two streams put waves of different
    # kernels on one SIMD, and that exposed a hazard single-stream runs never meet: a compiler-formed `v_pk_add_f32 ...
    # op_sel:[0,1]` (packed fp32, low lane reading the high half of a register pair — LayerNorm's x - mean of the second
    # row of a pair) read 0 for the swizzled operand in lanes 48-63 of roughly one wave in 10^7 whenever a tap_gemm kernel
    # of the other stream was resident (tools/exp/repro_e4.py: offset == mean * rstd in 16 values of one row, only beside
    # block shapes 1 / 2 / 3 / 6, never when idle, never with the scalar build of the same kernel).  csrc/build.py compiles
    # the two files that had the form without the SLP vectoriser and refuses it in any object (check_isa), which restored
    # run-to-run bit-equality with the side stream on; CCEDIT_OVERLAP_CONTROLNET=0 keeps everything on one stream.
    overlap_controlnet = policy.on("overlap_controlnet")
    _side_stream = None
    _half_stream = None

    @staticmethod
    def _tensor_key(tns: torch.Tensor):
        return (tns.data_ptr(), tuple(tns.shape), tuple(tns.stride()), tns._version, tns.dtype)

    _fail_capture_for_test = False      # tests/test_network_gpu.py: make the next capture raise after its launches were recorded

    def reset_caches(self):
        """Drop the per-clip caches (hint stem, shard slices, captured graphs).  Never needed for correctness — entries pin
        their source storage, see _guided_hint — only to release the previous clip's memory early."""
        self._hint_val = None
        self._hint_slices = None
        self._hint_dup = None
        self._twin_val = None
        self._tkv_val = None
        self._graphs = None

    def _guided_hint(self, hint5d: torch.Tensor, rows=None, half: bool = False):
        """hint_stem(1 - (hint+1)/2), cached per source tensor.  An entry is keyed by (address, shape, strides, in-place
        version) AND keeps a reference to the source tensor: while the entry lives its storage cannot be freed, so no
        later tensor (the next clip's hint) can be handed that address by the caching allocator — a key match always
        means the same bytes."""
        net = self.diffusion_model.controlnet
        key = self._tensor_key(hint5d) + (PACK_GENERATION[0], half)        # (a re-pack replaces the stem's weights)
        if self.cache_hint_stem and isinstance(self._hint_val, dict) and key in self._hint_val:
            return self._hint_val[key][1]
        # The two CFG halves carry the SAME hint (the sampling scripts give `uc` a clone of c's control_hint, sampling_tv2v.py:339-344,
        # and the guider concatenates them): the stem — eight per-frame convolutions at up to 512 x 768 — is evaluated on one half and its
        # output repeated.  Whether the halves are equal is decided by comparing them once per hint tensor (a device compare and one
        # host sync, remembered with the tensor's identity + version like the caches above; never decided while capturing a graph).
        dup = False
        if self.dedup_hint and hint5d.shape[0] % 2 == 0:
            if not isinstance(self._hint_dup, dict) or len(self._hint_dup) >= 8:
                self._hint_dup = {}
            ent = self._hint_dup.get(key[:-1])
            if ent is None and not torch.cuda.is_current_stream_capturing():
                k = hint5d.shape[0] // 2
                ent = self._hint_dup[key[:-1]] = (hint5d, bool(torch.equal(hint5d[:k], hint5d[k:])))
            dup = ent is not None and ent[1]
        dup = dup or half           # (half: the caller — _cfg_twins — has compared the halves itself)
        src = hint5d[: hint5d.shape[0] // 2] if dup else hint5d
        # control_hint in [-1,1] -> 1 - (h+1)/2 (wrappers.py:160-162), fused into the layout change
        hint8 = ops.ncthw_to_nhwc(src.float().contiguous(), 8, scale=-0.5, shift=0.5)
        g = net.hint_stem(hint8, rows=rows)
        if dup and not half:
            g = torch.cat([g, g])
        if self.cache_hint_stem:
            if not isinstance(self._hint_val, dict) or len(self._hint_val) >= 4:     # one entry per CFG half (+ shards)
                self._hint_val = {}
            self._hint_val[key] = (hint5d, g)
        return g

    # The two CFG halves of an evaluation (guiders.py:57-67: x, sigma, control_hint and cond_feat are the SAME tensor twice, only the
    # text differs) are identical up to the first text cross-attention: input_blocks.0, the ResBlock and the 6144-key self-attention
    # of input_blocks.1 — in the UNet and in the ControlNet — and all of controlnet_img.  With `share_cfg_prefix` that prefix is
    # evaluated ONCE (half the rows in every launch) and repeated where the halves part: same results (the halves' shared prefix is
    # then bit-identical, which the batched launches only deliver to summation-order noise), -2...3 ms per step at 17 x 512 x 768.
    # Whether the halves ARE equal is checked on the device (one compare + host sync per new (x, t) tensor pair, remembered like the
    # hint's); anything else — different latents in the two halves, odd batches, sharded evaluation — takes the general path.  bench.py
    # reports the FLOPs executed (`executed_flops_per_step`) next to the algorithmic count the metric is priced with.
    share_cfg_prefix = policy.on("share_cfg_prefix")
    _twin_val = None

    def _cfg_twins(self, x: torch.Tensor, t: torch.Tensor, c: Dict[str, torch.Tensor]) -> bool:
        if not self.share_cfg_prefix or x.shape[0] % 2 or x.shape[0] < 2 or torch.cuda.is_current_stream_capturing():
            return False
        k = x.shape[0] // 2
        # marked by this build's guider / denoiser (sampling.py): no device compare, no host sync
        # (ops.get_mark: a mark is void once its tensor was written in place after marking — then the values are compared below)
        marks = [ops.get_mark(x, "_cfg_twin_halves"), ops.get_mark(t, "_cfg_twin_halves"), ops.get_mark(c["control_hint"], "_halves_equal")]
        if c.get("cond_feat") is not None:
            marks.append(ops.get_mark(c["cond_feat"], "_halves_equal"))
        if marks[0] is True and marks[1] is True and all(m is not None for m in marks[2:]):
            return all(marks)
        key = self._tensor_key(x) + self._tensor_key(t) + tuple(self._tensor_key(c[n]) for n in ("control_hint", "cond_feat") if c.get(n) is not None)
        if not isinstance(self._twin_val, dict) or len(self._twin_val) >= 8:
            self._twin_val = {}
        ent = self._twin_val.get(key)
        if ent is None:
            same = (x[:k] == x[k:]).all() & (t[:k] == t[k:]).all() & (c["control_hint"][:k] == c["control_hint"][k:]).all()
            if c.get("cond_feat") is not None:
                same = same & (c["cond_feat"][:k] == c["cond_feat"][k:]).all()
            ent = self._twin_val[key] = ([x, t] + [c[n] for n in ("control_hint", "cond_feat") if c.get(n) is not None], bool(same.item()))
        return ent[1]

    _tkv_val = None

    def _text_kv(self, net, context: torch.Tensor, ctx2d: torch.Tensor) -> "TextKV":
        """net.text_kv(ctx2d), kept per (network, conditioning tensor) while the per-clip caches are on (`cache_hint_stem`): the text
        K / V projections are constant over the 59 evaluations of a clip (the reference recomputes them every time).  Same
        keying discipline as _guided_hint: identity + version of the source tensor, which the entry pins."""
        if not self.cache_hint_stem:
            return net.text_kv(ctx2d)
        key = (id(net),) + self._tensor_key(context) + (PACK_GENERATION[0],)
        if isinstance(self._tkv_val, dict) and key in self._tkv_val:
            return self._tkv_val[key][1]
        tkv = net.text_kv(ctx2d)
        if not isinstance(self._tkv_val, dict) or len(self._tkv_val) >= 6:
            self._tkv_val = {}
        self._tkv_val[key] = (context, tkv)
        return tkv

    # One network evaluation is ~560 kernel launches issued from Python; where the kernels are short (the 16x24 / 8x12 levels, the
    # norm passes) the GPU outruns the launching thread: 150 gaps of 5-10 us, 1.3 ms per step in the kernel trace
    # (tools/exp/gaps.py).  The 59 evaluations of a clip have identical shapes and conditioning tensors, so the launch sequence is
    # captured once into a HIP graph (torch.cuda.CUDAGraph: hipStreamBeginCapture on the stream our C-ABI launches go to — side
    # streams join through their events) and replayed: first call with a new (shapes, conditioning tensors) key runs eagerly
    # (fills the hint-stem cache, the per-kernel attribute guards, the allocator), the second captures, later ones copy x / t
    # into the static inputs and replay.  Same kernels in the same order: bit-identical to the eager path
    # (tests/test_network_gpu.py).  CCEDIT_GRAPH=0 disables; sharded / profiled / traced evaluations are always eager, and so are
    # the two-stream CFG halves (CCEDIT_SPLIT_CFG=1: capturing four streams that fork and join inside each other crashed the
    # runtime on ROCm 7.2 — not pursued, the batched pass is the default).
    use_graph = policy.on("graph") and not _SPLIT_CFG
    _graphs = None
    _graph_failed = False

    def forward(self, x: torch.Tensor, t: torch.Tensor, c: Dict[str, torch.Tensor], **kwargs) -> torch.Tensor:
        # row-sharded evaluations are captured too when their exchanges are stream operations (RCCL; the host-staged gloo transport of the
        # CPU / one-GPU tests is not): the launch count per rank is the single-GPU one while every kernel is N times shorter
        if (self.use_graph and not OpenAIWrapperControlLDM3DTV2V._graph_failed and not kwargs and x.is_cuda
                and self.frame_shard is None
                and (self.row_shard is None or (not isinstance(self.row_shard, (tuple, list)) and self.row_shard.can_capture()))
                and ops.PROFILE is None and TRACE is None and not torch.cuda.is_current_stream_capturing()):
            return self._forward_graphed(x, t, c)
        return self._forward_eager(x, t, c, **kwargs)

    def _forward_graphed(self, x, t, c):
        # the shared CFG prefix is a property of the VALUES of x and t: decided here, outside any capture, and part of the key — a
        # graph recorded for identical halves is never replayed for different ones
        twins = self._cfg_twins(x, t, c) if (self.frame_shard is None and self.row_shard is None) else False
        # PACK_GENERATION: a captured graph holds the addresses of the packed weights it was recorded with
        key = (tuple(x.shape), x.dtype, tuple(t.shape), t.dtype, self.cache_hint_stem, self.overlap_controlnet, PACK_GENERATION[0], twins,
               None if self.row_shard is None else (id(self.row_shard), self.row_shard.attn),
               tuple(sorted((k, self._tensor_key(v)) if torch.is_tensor(v) else (k, repr(v)) for k, v in c.items())))
        if self._graphs is None:
            self._graphs = {}
        ent = self._graphs.get(key)
        if ent is None:
            while len(self._graphs) >= 2:                       # a clip uses one key; keep the previous clip's until it is replaced
                self._graphs.pop(next(iter(self._graphs)))
            # the conditioning tensors are pinned while the entry lives: a key match always means the same bytes (see _guided_hint)
            self._graphs[key] = dict(pins=[v for v in c.values() if torch.is_tensor(v)])
            return self._forward_eager(x, t, c, _twins=twins)
        if "graph" not in ent:
            ok = True
            try:
                ent["x"], ent["t"] = x.clone(), t.clone()
                g = torch.cuda.CUDAGraph()
                # thread_local: calls from OTHER threads (the process group's watchdog polling its events when torch.distributed is
                # initialised — bench.py --gpus N) must not invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    # scratch arenas used while capturing must live in this graph's pool, and nothing outside the capture may go on
                    # using them: their zero fills are only RECORDED, and after a failed capture never run at all — the eager
                    # fallback would accumulate statistics onto uninitialised memory and wait on garbage split-K counters
                    ops.reset_stream_scratch()
                    try:
                        ent["out"] = self._forward_eager(ent["x"], ent["t"], c, _twins=twins)
                        if self._fail_capture_for_test:
                            raise RuntimeError("capture failure injected by a test")
                    finally:
                        ops.reset_stream_scratch()
                ent["graph"] = g
                ent["pins"].append(dict(self._hint_val) if isinstance(self._hint_val, dict) else None)   # the cached stem output it reads
                ent["pins"].append(dict(self._tkv_val) if isinstance(self._tkv_val, dict) else None)     # ... and the cached text K / V
            except Exception as e:                              # capture is an optimisation: report once, keep evaluating eagerly
                import warnings
                ok = False
                warnings.warn(f"HIP graph capture of the network evaluation failed ({type(e).__name__}: {e}); continuing without graphs")
            # Row-sharded: the ranks must AGREE on graph-or-eager before anyone goes on.  A rank that fell back alone would run
            # the ControlNet's exchanges on the sibling communicator / side stream while its peers replay a single-stream graph: collectives
            # on different communicators in different orders, i.e. a deadlock.  One all-reduced flag, issued outside any capture; every
            # rank reaches it (a rank whose capture succeeded has not replayed yet).
            if self.row_shard is not None:
                ok = self.row_shard.all_agree(ok)
            if not ok:
                OpenAIWrapperControlLDM3DTV2V._graph_failed = True
                self._graphs = None
                ops.reset_stream_scratch()                      # (again: whatever the aborted capture left behind)
                return self._forward_eager(x, t, c, _twins=twins)
        else:
            ent["x"].copy_(x)
            ent["t"].copy_(t)
        ent["graph"].replay()
        return ent["out"].clone()

    def _forward_eager(self, x: torch.Tensor, t: torch.Tensor, c: Dict[str, torch.Tensor], **kwargs) -> torch.Tensor:
        pair = isinstance(self.frame_shard, (tuple, list)) and not kwargs.get("_half")
        # RowShard.cfg_pair (round 6): the two CFG halves are independent evaluations — each on its own stream and communicator, so
        # one half's exchanges (head all-to-alls, halo rows with their statistics) wait under the other half's kernels
        rpair = isinstance(self.row_shard, (tuple, list)) and not kwargs.get("_half")
        if rpair and x.shape[0] != 2:
            raise ValueError("a RowShard.cfg_pair shards the two CFG halves of a batch-2 step")
        if pair and x.shape[0] != 2:
            raise ValueError("a FrameShard.cfg_pair shards the two CFG halves of a batch-2 step")
        if pair or rpair or (_SPLIT_CFG and x.shape[0] == 2 and self.frame_shard is None and ops.PROFILE is None and TRACE is None
                             and not kwargs.get("_half")):
            # the two CFG halves are independent: two streams (and, frame-sharded, two communicators with mirrored
            # partitions — one half's exchanges overlap the other half's kernels)
            shards = self.frame_shard if pair else (None, None)
            rshards = self.row_shard if rpair else (None, None)
            main = torch.cuda.current_stream()
            if OpenAIWrapperControlLDM3DTV2V._half_stream is None:
                OpenAIWrapperControlLDM3DTV2V._half_stream = torch.cuda.Stream()
            hs = OpenAIWrapperControlLDM3DTV2V._half_stream
            hs.wait_stream(main)
            halves = [{k: (v[i:i + 1].contiguous() if torch.is_tensor(v) else v) for k, v in c.items()} for i in range(2)]
            with torch.cuda.stream(hs):
                e1 = self._forward_eager(x[1:2].contiguous(), t[1:2].contiguous(), halves[1], _half=True, _shard=shards[1], _rshard=rshards[1])
            e0 = self._forward_eager(x[0:1].contiguous(), t[0:1].contiguous(), halves[0], _half=True, _shard=shards[0], _rshard=rshards[0])
            main.wait_stream(hs)
            e1.record_stream(main)
            return torch.cat([e0, e1])
        if c.get("concat") is not None and c["concat"].numel():
            raise NotImplementedError("'concat' conditioning is not used by the TV2V configs")
        net = self.diffusion_model
        b, _, nt, lh, lw = x.shape
        if lh % 8 or lw % 8:
            raise ValueError(f"latent {lh}x{lw}: frame sizes must be multiples of 64 pixels (three stride-2 levels whose "
                             f"skips are concatenated with the 2x-upsampled decoder tensors, controlmodel.py:539-543)")
        if tuple(c["control_hint"].shape[-2:]) != (8 * lh, 8 * lw):
            raise ValueError(f"control_hint {tuple(c['control_hint'].shape)} does not match latent {lh}x{lw} (x8)")
        sh = kwargs.get("_shard", None) if kwargs.get("_half") else self.frame_shard
        hint5 = c["control_hint"]
        if sh is not None:             # keep this rank's keyframes of every clip; everything spatial is frame-local
            if sh.t_glob != nt:
                raise ValueError(f"frame shard built for T={sh.t_glob}, got T={nt}")
            x = x[:, :, sh.t0:sh.t1]
            hint5 = hint5[:, :, sh.t0:sh.t1]
            hk = self._tensor_key(c["control_hint"]) + (sh.t0, sh.t1)
            if not isinstance(self._hint_slices, dict) or len(self._hint_slices) >= 4:
                self._hint_slices = {}
            if hk not in self._hint_slices:
                # a stable tensor object so the hint-stem cache can hit; the entry pins the source (see _guided_hint)
                self._hint_slices[hk] = (c["control_hint"], hint5.contiguous())
            hint5 = self._hint_slices[hk][1]
        rs = kwargs.get("_rshard", None) if kwargs.get("_half") else self.row_shard
        cond_feat = c.get("cond_feat", None)
        if rs is not None:             # keep this rank's latent rows of every frame (and the 8x finer hint rows that feed them)
            if sh is not None:
                raise ValueError("frame_shard and row_shard are alternative decompositions of one clip")
            rs.check_latent(lh)
            r0, r1 = rs.rows(lh)
            x = x[:, :, :, r0:r1]
            hk = self._tensor_key(c["control_hint"]) + ("rows", r0, r1)
            if not isinstance(self._hint_slices, dict) or len(self._hint_slices) >= 4:
                self._hint_slices = {}
            if hk not in self._hint_slices:
                self._hint_slices[hk] = (c["control_hint"], hint5[:, :, :, 8 * r0:8 * r1].contiguous())
            hint5 = self._hint_slices[hk][1]
            if cond_feat is not None:
                cond_feat = cond_feat[:, :, r0:r1]
        geo = Geometry(b, x.shape[2], sh, rows=rs)
        context = c["crossattn"]
        ctx2d = context.to(torch.bfloat16).reshape(-1, context.shape[-1]).contiguous()
        x8 = ops.ncthw_to_nhwc(x.float().contiguous(), 8)
        # identical CFG halves: their shared prefix is evaluated once (see _cfg_twins); never for sharded / traced / split evaluations
        twins = kwargs.get("_twins")
        if sh is not None or rs is not None or TRACE is not None or kwargs.get("_half"):
            twins = False
        elif twins is None:
            twins = self._cfg_twins(x, t, c)
        twins = bool(twins) and _splits_at_text(net.input_blocks[1]) and _splits_at_text(net.controlnet.input_blocks[1])
        x8h = x8[: x8.shape[0] // 2] if twins else x8
        control_ready = None
        # Row-sharded: a captured evaluation keeps everything on ONE stream — capturing RCCL collectives issued from two streams that
        # fork and join inside the graph crashes the runtime (ROCm 7.2, one or two communicators alike: tools/exp/rows_rccl_debug.py),
        # and with every kernel N times shorter it is the graph, not the overlap, that matters.  Eager RCCL evaluations (graphs off)
        # do use the side stream, the ControlNet's exchanges on a communicator of their own (RowShard.sibling).
        rows_side_ok = rs is None or (rs.can_capture() and not (self.use_graph and not OpenAIWrapperControlLDM3DTV2V._graph_failed))
        if kwargs.get("_half") and rs is not None:
            rows_side_ok = False                        # a CFG half of a RowShard pair: the other half is what overlaps, one stream each
        if self.overlap_controlnet and sh is None and rows_side_ok and ops.PROFILE is None and TRACE is None:
            main = torch.cuda.current_stream()
            if OpenAIWrapperControlLDM3DTV2V._side_stream is None:
                OpenAIWrapperControlLDM3DTV2V._side_stream = {}
            side = OpenAIWrapperControlLDM3DTV2V._side_stream.setdefault(main.cuda_stream, None)
            if side is None:
                side = OpenAIWrapperControlLDM3DTV2V._side_stream[main.cuda_stream] = torch.cuda.Stream()
            side.wait_stream(main)                      # x8 / ctx2d / t are ready
            with torch.cuda.stream(side):
                # rows sharded: the ControlNet's exchanges go through a communicator of their own (RowShard.sibling)
                rs_side = None if rs is None else rs.sibling()
                geo_side = geo if rs is None else Geometry(b, x.shape[2], sh, rows=rs_side)
                guided = self._guided_hint(hint5, rows=rs_side, half=twins)
                control = net.controlnet.run(x8h, guided, t, self._text_kv(net.controlnet, context, ctx2d), context.shape[1], geo_side,
                                             shared=twins)
                control_ready = torch.cuda.Event()
                control_ready.record(side)
            for tns in (x8, ctx2d):
                tns.record_stream(side)                 # allocated on the main stream, read on the side stream (x8h is a view of x8)
            for tns in control:
                tns.record_stream(main)                 # and vice versa
        else:
            guided = self._guided_hint(hint5, rows=rs, half=twins)
            control = net.controlnet.run(x8h, guided, t, self._text_kv(net.controlnet, context, ctx2d), context.shape[1], geo, shared=twins)
        img_control = None
        if cond_feat is not None and (sh is None or sh.owner_of(nt // 2) == sh.rank):
            # TVI2V (wrappers.py:176-190): controlnet_img on the reference latent; its residuals only touch keyframe T//2,
            # so under frame sharding only the rank holding that keyframe evaluates it
            # (identical CFG halves: controlnet_img has no text input — ALL of it is evaluated once, the UNet indexes its residuals modulo)
            cf = cond_feat[: b // 2] if twins else cond_feat
            tt_ = t[: b // 2] if twins else t
            cf8 = ops.ncthw_to_nhwc(cf.float()[:, :, None].contiguous(), 8)
            img_control = net.controlnet_img.run(None, cf8, tt_, None, 0, Geometry(cf.shape[0], 1, rows=rs))
        eps = net.run(x8h, t, self._text_kv(net, context, ctx2d), context.shape[1], control, geo, img_control, control_ready=control_ready,
                      shared=twins)
        if sh is not None:             # all ranks get the full (B, C, T, h, w) prediction (1.6 MB at 17x64x96)
            eps = sh.gather_pixels(eps.view(-1, eps.shape[-1]), b, lh * lw) if sh.mode == "a2a" else sh.gather_frames(eps, b)
            eps = eps.view(b * nt, lh, lw, -1)
        if rs is not None:             # all ranks get the full prediction: the row blocks of every frame, in rank order
            eps = rs.gather_rows(eps.view(b * nt, -1, eps.shape[-1])).view(b * nt, lh, lw, -1)
        return ops.nhwc_to_ncthw(eps, b, nt, net.out_channels)
