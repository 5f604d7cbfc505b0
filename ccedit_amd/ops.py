"""Python-side operator wrappers over the C ABI (include/ccedit_hip.h).

PyTorch is used for device memory (torch.empty on the caching allocator) and the stream handle only;
every arithmetic result below is produced by a hand-written HIP kernel in libccedit_hip.so.  All
activations are bf16 tensors in the frames-outermost channels-last layout: (N, H, W, C) contiguous,
N = B*T frames, equivalently a row-major [N*H*W][C] matrix.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import hip, policy
from .hip import (ACT_GEGLU, ACT_NONE, ACT_QUICK_GELU, ACT_SILU, CcAttnDesc, CcFf320Desc, CcGemmDesc, GEMM_CONV2D, GEMM_LINEAR,
                  GEMM_TEMPORAL)
from .packing import PackedFF320, PackedWeight

BF16 = torch.bfloat16


class LaunchProfile:
    """Optional per-launch HIP-event timing of the dominant kernel families (used by bench.py only).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = {}          # family -> list of (start_event, end_event, algorithmic_flops, algorithmic_bytes)
        self.executed = 0.0        # sum of executed FLOPs over all recorded launches

    def add(self, family, e0, e1, flops, nbytes, shape=None, kernel=None, exec_flops=None):
        if kernel is None:
            k = hip.lib().ccedit_last_kernel()           # the kernel template the entry point just dispatched to
            kernel = k.decode() if k else "?"
        # FLOPs the matrix pipe really retires for this launch where that differs from the algorithmic count (the parity form of
        # upsample + conv executes 4/9 of the reference convolution's multiply-adds)
        ex = flops if exec_flops is None else exec_flops
        self.records.setdefault(family, []).append((e0, e1, flops, nbytes, shape, kernel, ex))
        self.executed += ex

    def by_kernel(self, families=("tap_gemm", "attention", "memory")):
        """[{kernel, launches, ms, tflops, gbytes_per_s}] over the given families, sorted by time (gbytes_per_s: algorithmic bytes
        over the launch time — the number to hold against HBM for the `memory` family: norms, concatenation, layout passes)."""
        torch.cuda.synchronize()
        acc = {}
        for fam in families:
            for a, b, fl, nb, _, k, ex in self.records.get(fam, []):
                n, ms, f, by, _, fe = acc.get(k, (0, 0.0, 0.0, 0.0, fam, 0.0))
                acc[k] = (n + 1, ms + a.elapsed_time(b), f + fl, by + nb, fam, fe + ex)
        return sorted(({"kernel": k, "family": fam, "launches": n, "ms": ms, "tflops": f / (ms * 1e-3) / 1e12,
                        "exec_tflops": fe / (ms * 1e-3) / 1e12, "gbytes_per_s": by / (ms * 1e-3) / 1e9, "bytes": by}
                       for k, (n, ms, f, by, fam, fe) in acc.items()), key=lambda r: -r["ms"])

    def by_shape(self, family, with_kernel=False):
        """{shape: (launches, total_ms, tflops)} sorted by time (with_kernel: keyed on (shape, kernel label))."""
        torch.cuda.synchronize()
        acc = {}
        for a, b, fl, _, shape, _k, _ex in self.records.get(family, []):
            if with_kernel:
                shape = (shape, _k)
            n, ms, f = acc.get(shape, (0, 0.0, 0.0))
            acc[shape] = (n + 1, ms + a.elapsed_time(b), f + fl)
        return sorted(((k, n, ms, f / (ms * 1e-3) / 1e12) for k, (n, ms, f) in acc.items()), key=lambda r: -r[2])

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            out[fam] = dict(launches=len(recs), total_ms=sum(ms), avg_us=1e3 * sum(ms) / max(len(ms), 1),
                            flops=sum(r[2] for r in recs), bytes=sum(r[3] for r in recs))
        return out


PROFILE: Optional[LaunchProfile] = None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _launch_mem(kernel: str, nbytes: float, fn, shape=None):
    """Run one bandwidth-bound launch; under bench.py's profiled step also time it (HIP events on the launch stream) with its
    algorithmic bytes (every operand read once, the result written once)."""
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    PROFILE.add("memory", e0, e1, 0.0, float(nbytes), (kernel,) + tuple(shape or ()), kernel)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk_act(t: torch.Tensor, name: str):
    if t.dtype != BF16 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous cuda bf16 tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


WFRAG = policy.on("wfrag")      # 0: CcGemmDesc.Wfrag stays null (A/B of the coalesced weight preload)


def gemm(a2d: torch.Tensor, pw: PackedWeight, *, mode: int = GEMM_LINEAR, m: Optional[int] = None,
         hin: int = 0, win: int = 0, hout: int = 0, wout: int = 0, stride: int = 1, pad: int = 0, upsample: bool = False,
         t: int = 0, hw: int = 0, tsrc: int = 0, tsrc_off: int = 0, t0: int = 0, tglob: int = 0,
         a2: Optional[torch.Tensor] = None, act: int = ACT_NONE,
         group_bias: Optional[torch.Tensor] = None, group_rows: int = 0,
         res1: Optional[torch.Tensor] = None, res2: Optional[torch.Tensor] = None,
         out: Optional[torch.Tensor] = None, out_f32: bool = False, use_bias: bool = True, tile: int = 0,
         gn_rows: int = 0, ln_eps: float = 0.0, ln_stats: Optional[torch.Tensor] = None, ln_sums=None,
         row_sums: bool = False, subpix: int = 0, vpad: bool = False, halo=None) -> torch.Tensor:
    """out[m, :] = epilogue(sum_taps W . A[src(m, tap)]).  a2d: [rows, lda] bf16 (last dim contiguous).

    gn_rows > 0 (= H*W of the output frames) asks the epilogue to also accumulate the GroupNorm(32) statistics of
    the output; they travel with the returned tensor (`gn_stats_of`) and `groupnorm_spatial` then skips its own
    statistics pass.  Silently not fused when the shape does not qualify (see include/ccedit_hip.h)."""
    assert a2d.dtype == BF16 and a2d.is_cuda and a2d.stride(-1) == 1
    lda = a2d.stride(0)
    cin1 = a2d.shape[1]
    cin = cin1 + (a2.shape[1] if a2 is not None else 0)
    if cin != pw.cin:
        raise ValueError(f"gemm: source has {cin} channels, packed weight expects {pw.cin}")
    if m is None:
        m = a2d.shape[0]
    n_store = pw.n_out if pw.geglu else pw.n
    if out is None:
        out = torch.empty((m, n_store), dtype=torch.float32 if out_f32 else BF16, device=a2d.device)
    assert out.stride(-1) == 1 and out.shape[0] == (4 * m if subpix else m)
    d = CcGemmDesc()
    d.subpix, d.vpad = subpix, int(vpad)
    if halo is not None:            # (top, bottom) boundary rows of the neighbour ranks, each (frames, Win, C) with A's row stride, or None
        assert not vpad and mode == GEMM_CONV2D
        d.vpad = 2
        for h_ in halo:
            if h_ is not None and not (h_.dtype == BF16 and h_.is_cuda and h_.is_contiguous() and h_.dim() == 3 and h_.shape[1] == win
                                       and h_.shape[2] == lda):
                raise ValueError("gemm: halo rows must be contiguous cuda bf16 (frames, Win, lda) tensors")
        d.halo_top, d.halo_bot = _ptr(halo[0]), _ptr(halo[1])
    d.M, d.N, d.Cin, d.Cin1, d.taps, d.mode = m, pw.n, pw.cin, cin1, pw.taps, mode
    d.Hin, d.Win, d.Hout, d.Wout = hin, win, hout, wout
    d.stride, d.pad, d.ksize, d.upsample = stride, pad, pw.ksize, int(upsample)
    d.T, d.HW = t, hw
    d.Tsrc, d.tsrc_off, d.t0, d.Tglob = tsrc, tsrc_off, t0, tglob
    d.lda, d.lda2, d.ldc, d.Kpad = lda, (a2.stride(0) if a2 is not None else 0), out.stride(0), pw.kpad
    d.act = ACT_GEGLU if pw.geglu else act
    d.out_f32 = int(out.dtype == torch.float32)
    d.group_rows = group_rows
    if group_bias is not None:
        assert group_bias.dtype == torch.float32 and group_bias.stride(-1) == 1 and group_bias.shape[-1] == pw.n
        d.ldgb = group_bias.stride(0) if group_bias.dim() == 2 else pw.n
    d.ldr1 = res1.stride(0) if res1 is not None else 0
    d.ldr2 = res2.stride(0) if res2 is not None else 0
    d.tile = tile
    if ln_eps:          # rows of A normalised on the way in (lin320 only): the weight must come from packing.fold_layernorm
        if not ln320_applicable(m, pw, act, res1, res2, group_bias, out_f32, gn_rows):
            raise ValueError("gemm: ln_eps needs the K = 320 register-resident-weight shape (>= 32768 rows, plain Linear)")
        d.ln_eps, d.tile = ln_eps, 9
    if ln_stats is not None:      # rows of A are un-normalised LayerNorm inputs: (mean, rstd) applied in the epilogue (g8_kernel only)
        if pw.colsum is None or ln_stats.dtype != torch.float32 or tuple(ln_stats.shape) != (m, 2) or not ln_stats.is_contiguous():
            raise ValueError("gemm: ln_stats needs a packing.fold_layernorm weight and a contiguous fp32 [M, 2] statistics tensor")
        d.ln_stats, d.ln_colsum = ln_stats.data_ptr(), pw.colsum.data_ptr()
    if ln_sums is not None:       # ... or their (sum, sum of squares) as the producing GEMM left them (`row_sums=True` there): (tensor, eps)
        sums, eps = ln_sums
        if pw.colsum is None or sums.dtype != torch.float64 or tuple(sums.shape) != (m, 2) or not sums.is_contiguous():
            raise ValueError("gemm: ln_sums needs a packing.fold_layernorm weight and the contiguous fp64 [M, 2] sums of the producer")
        d.ln_sums, d.ln_sums_eps, d.ln_colsum = sums.data_ptr(), eps, pw.colsum.data_ptr()
    if row_sums:                  # LayerNorm statistics of the output, accumulated by the epilogue; they travel with the tensor (ln_sums_of)
        sums = _ZEROS.take(2 * m, out.device).view(m, 2)
        d.row_sums = sums.data_ptr()
        out._ln_sums = sums
    elif hasattr(out, "_ln_sums"):
        del out._ln_sums
    d.korder = pw.korder
    d.A, d.A2, d.W = a2d.data_ptr(), _ptr(a2), pw.w.data_ptr()
    d.Wfrag = _ptr(pw.wfrag) if WFRAG else None
    d.bias = _ptr(pw.bias) if use_bias else None
    d.group_bias = _ptr(group_bias)
    d.res1, d.res2, d.out = _ptr(res1), _ptr(res2), out.data_ptr()
    stats = None
    if (gn_rows > 0 and FUSE_GN_STATS and gn_rows % 128 == 0 and m % gn_rows == 0 and pw.n % 32 == 0 and pw.n >= 256
            and not pw.geglu and out.dtype == BF16 and out.shape[1] == pw.n and out.is_contiguous()):
        stats = zero_stats(m // gn_rows, out.device)
        d.gn_rows, d.gn_stats = gn_rows, stats.data_ptr()
        out._gn_stats = (stats, gn_rows)
    elif hasattr(out, "_gn_stats"):
        del out._gn_stats              # a caller-supplied `out` is being overwritten: statistics left on it are stale
    if m <= 8192 and pw.n * pw.kpad >= 1 << 21 and SPLIT_K:      # few output tiles, long K: lend the split-K scratch (include/ccedit_hip.h)
        need = hip.lib().ccedit_gemm_workspace_bytes(C.byref(d))
        if need:
            ws = _splitk_ws(need, out.device)
            d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(hip.lib().ccedit_gemm(C.byref(d), _stream()), "ccedit_gemm")
        e1.record()
        nres = int(res1 is not None) + int(res2 is not None)
        alg_bytes = (a2d.shape[0] * cin1 * 2 + (a2.shape[0] * a2.shape[1] * 2 if a2 is not None else 0)      # sources, once
                     + m * out.shape[1] * out.element_size() + nres * m * pw.n * 2 + pw.n * pw.kpad * 2)   # out, residuals, W
        PROFILE.add("tap_gemm", e0, e1, pw.flops_per_row * m, float(alg_bytes),
                    (("lin", "conv", "temp")[mode] + ("+up" if upsample else "") + ("+up(parity)" if subpix else ""), m, pw.n, pw.taps * pw.cin, stride,
                     int(res1 is not None) + int(res2 is not None), d.act),
                    exec_flops=pw.flops_per_row * m * (4.0 / 9.0) if subpix else None)
        return out
    hip.check(hip.lib().ccedit_gemm(C.byref(d), _stream()), "ccedit_gemm")
    return out


def linear(x2d, pw, **kw):
    return gemm(x2d, pw, mode=GEMM_LINEAR, **kw)


def set_mark(t: torch.Tensor, name: str, value) -> None:
    """A host-side fact about the VALUES of a tensor (e.g. "its two CFG halves are equal"), carried as a Python attribute together with
    the tensor's in-place version counter: get_mark returns it only while the tensor has not been written since (ADVICE r5 — an
    inpainting blend or per-half noise added in place after marking must not leave a stale mark behind)."""
    setattr(t, name, (value, t._version))


def get_mark(t, name: str):
    """The marked value, or None when the tensor carries no such mark or was modified in place after it was marked."""
    m = getattr(t, name, None) if torch.is_tensor(t) else None
    if not (isinstance(m, tuple) and len(m) == 2):
        return None
    return m[0] if m[1] == t._version else None


LN320 = policy.on("ln320")      # 0: separate LayerNorm pass in front of the K = 320 projections


def ln320_applicable(m, pw, act=ACT_NONE, res1=None, res2=None, group_bias=None, out_f32=False, gn_rows=0) -> bool:
    """Can `linear(layernorm(x), pw)` run as ONE lin320 launch that normalises the rows in LDS?  Only single-slice layers
    (N = 320: to_q): every 320-channel slice of a wider layer is its own workgroup and would normalise the tile again — measured
    114 us against 153 us for LayerNorm + lin320 at N = 320, but 301 against 292 at N = 960 (tools/exp/ln320_time.py)."""
    return (LN320 and m >= 32768 and pw.cin == 320 and pw.taps == 1 and pw.kpad == 320 and pw.n == 320
            and not pw.geglu and act == ACT_NONE and res1 is None and res2 is None and group_bias is None and not out_f32
            and gn_rows == 0)


ATTN_Q_LOG2 = policy.on("attn_q_log2")      # 0: softmax scale applied inside the attention kernels (A/B)
FF320 = policy.on("ff320")      # 0: LayerNorm + two GEMMs instead of the fused dim-320 feed-forward
BLOCK_TAIL = policy.on("block_tail") and FF320      # to_out / proj_out of the dim-320 transformer tails inside the ff320 launch


def ff320(x2d: Optional[torch.Tensor], pk: PackedFF320, eps: float = 1e-5, ln: bool = True, out: Optional[torch.Tensor] = None,
          dbg: Optional[torch.Tensor] = None, a: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
          res2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x + W2 . GEGLU(W1 . LayerNorm(x) + b1) + b2 for dim 320 in one kernel (csrc/ff320.hip).  x2d: [tokens, 320].
    Block tail (pk from packing.pack_ff320_tail): x2d is None and x = W_o . a + b_o + res is computed by the kernel's prologue GEMM;
    with res2 (and an epilogue pack) the result is W_p . (x + FF(LN(x))) + b_p + res2."""
    tail = a is not None
    src = a if tail else x2d
    assert src.dtype == BF16 and src.is_cuda and src.stride(-1) == 1 and src.shape[1] == 320
    m = src.shape[0]
    if out is None:
        out = torch.empty((m, 320), dtype=BF16, device=src.device)
    assert out.dtype == BF16 and out.stride(-1) == 1 and out.shape == (m, 320) and out.data_ptr() != src.data_ptr()
    d = CcFf320Desc()
    d.M, d.dim, d.inner, d.ldo, d.eps, d.ln = m, 320, 1280, out.stride(0), eps, int(ln)
    d.out, d.wstream, d.b2p = out.data_ptr(), pk.stream.data_ptr(), pk.b2p.data_ptr()
    d.dbg = None if dbg is None else dbg.data_ptr()
    nbytes = 2.0 * m * 320 * 2 + pk.stream.numel()
    if tail:
        if pk.bop is None or res is None or x2d is not None:
            raise ValueError("ff320: the block-tail form needs a packing.pack_ff320_tail pack, `a` and `res`, and no x2d")
        if (res2 is not None) != (pk.bpp is not None):
            raise ValueError("ff320: res2 goes with an epilogue pack (pack_ff320_tail(..., wp=...)), and only with one")
        for t_ in (res, res2):
            if t_ is not None:
                assert t_.dtype == BF16 and t_.is_cuda and t_.stride(-1) == 1 and tuple(t_.shape) == (m, 320) and t_.data_ptr() != out.data_ptr()
        d.a, d.lda, d.res, d.ldr, d.bop = a.data_ptr(), a.stride(0), res.data_ptr(), res.stride(0), pk.bop.data_ptr()
        nbytes += 2.0 * m * 320
        if res2 is not None:
            d.res2, d.ldr2, d.bpp = res2.data_ptr(), res2.stride(0), pk.bpp.data_ptr()
            nbytes += 2.0 * m * 320
    else:
        if pk.bop is not None:
            raise ValueError("ff320: a block-tail pack needs `a` / `res` (use pk.tail_of for the plain feed-forward)")
        d.x, d.ldx = x2d.data_ptr(), x2d.stride(0)
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(hip.lib().ccedit_ff320(C.byref(d), _stream()), "ccedit_ff320")
        e1.record()
        PROFILE.add("tap_gemm", e0, e1, pk.flops_per_row * m, float(nbytes),
                    ("ff320" + ("+to_out" if tail else "") + ("+proj_out" if res2 is not None else ""), m, 320, 1280, 1, 1, 2))
        return out
    hip.check(hip.lib().ccedit_ff320(C.byref(d), _stream()), "ccedit_ff320")
    return out


def conv2d(x: torch.Tensor, pw: PackedWeight, stride: int = 1, pad: int = 1, upsample: bool = False,
           x2: Optional[torch.Tensor] = None, gn: bool = False, out_hw: Optional[tuple] = None, vpad: bool = False, halo=None,
           **kw) -> torch.Tensor:
    """x: (N, H, W, C) -> (N, Hout, Wout, Cout); 3x3 (pad 1) or 1x1 (pad 0) by the packed kernel size.
    gn=True: the output feeds a spatial GroupNorm — accumulate its statistics in the epilogue."""
    n, h, w, c = x.shape
    if pw.ksize == 1:
        pad = 0
    hv, wv = (2 * h, 2 * w) if upsample else (h, w)
    hout = (hv + 2 * pad - pw.ksize) // stride + 1
    wout = (wv + 2 * pad - pw.ksize) // stride + 1
    if vpad:                        # the frames carry their own halo rows (RowShard): no vertical padding
        assert pw.ksize == 3 and pad == 1 and not upsample and x2 is None
        hout = (hv - pw.ksize) // stride + 1
        kw["vpad"] = True
    if halo is not None:            # rows sharded over ranks, the neighbours' boundary rows as separate tensors (CcGemmDesc.vpad = 2)
        assert pw.ksize == 3 and pad == 1 and not upsample and x2 is None and not vpad
        kw["halo"] = halo
    if out_hw is not None:          # asymmetric padding (VAE Downsample, model.py:74-93: pad right/bottom only, conv pad 0):
        hout, wout = out_hw         # taps that fall outside the source read zeros, so only the output size changes
    a2 = None if x2 is None else x2.reshape(-1, x2.shape[-1])
    if pw.ksize == 1 and pw.taps == 1 and stride == 1 and not upsample and out_hw is None and not vpad and halo is None and CONV1X1_LINEAR:
        # a 1 x 1 convolution is a Linear over the pixels (zero convs, skip connections): the Linear dispatch reaches the streaming
        # K = 320 / 640 kernels and the persistent GEMM, the convolution modes do not
        out = gemm(x.reshape(-1, c), pw, mode=GEMM_LINEAR, a2=a2, gn_rows=hout * wout if gn else 0, **kw)
        return carry_gn_stats(out, out.view(n, hout, wout, out.shape[-1]))
    out = gemm(x.reshape(-1, c), pw, mode=GEMM_CONV2D, m=n * hout * wout, hin=h, win=w, hout=hout, wout=wout,
               stride=stride, pad=pad, upsample=upsample, a2=a2, gn_rows=hout * wout if gn else 0, **kw)
    return carry_gn_stats(out, out.view(n, hout, wout, out.shape[-1]))


CONV1X1_LINEAR = policy.on("conv1x1_linear")      # 0: 1 x 1 convs through the convolution mode (A/B)
SUBPIX = policy.on("subpix")      # 0: upsample + 3x3 conv through the nine-tap gather (A/B)


def conv2d_upsampled(x: torch.Tensor, pws, vpad: bool = False, halo=None, tile: int = 0) -> torch.Tensor:
    """conv3x3(nearest_upsample_2x(x)) as four 2 x 2 convolutions on x, one per output parity (packing.pack_upsample_parities):
    x (N, H, W, C) -> (N, 2H, 2W, Cout); every launch writes its quarter of the output pixels in place.  vpad: x carries one halo
    row above and below its H rows (RowShard, extended copy); halo = (top, bottom): the same rows as separate tensors."""
    _chk_act(x, "conv2d_upsampled")
    n, hx, w, c = x.shape
    h = hx - 2 if vpad else hx
    out = torch.empty((n * 4 * h * w, pws[0].n), dtype=BF16, device=x.device)
    for p, pw in enumerate(pws):
        gemm(x.reshape(-1, c), pw, mode=GEMM_CONV2D, m=n * h * w, hin=hx, win=w, hout=h, wout=w, stride=1, pad=0, out=out, subpix=p + 1,
             vpad=vpad, halo=halo, tile=tile)
    return out.view(n, 2 * h, 2 * w, pws[0].n)


def conv_temporal(x: torch.Tensor, t: int, pw: PackedWeight, gn: bool = False, **kw) -> torch.Tensor:
    """Conv1d over the T frames of each clip; x: (B*T, H, W, C)."""
    n, h, w, c = x.shape
    if pw.taps == 1:
        out = gemm(x.reshape(-1, c), pw, mode=GEMM_LINEAR, gn_rows=h * w if gn else 0, **kw)
    else:
        out = gemm(x.reshape(-1, c), pw, mode=GEMM_TEMPORAL, t=t, hw=h * w, gn_rows=h * w if gn else 0, **kw)
    return carry_gn_stats(out, out.view(n, h, w, out.shape[-1]))


def conv_temporal_sharded(x_ext: torch.Tensor, b: int, t_local: int, t0: int, t_glob: int, pw: PackedWeight,
                          gn: bool = False, **kw) -> torch.Tensor:
    """Conv1d (k3) over T for a frame shard.  x_ext: (B*(t_local+2), H, W, C) = per clip [halo from the previous
    rank | t_local local frames | halo from the next rank] (halo contents are ignored where they fall outside the
    clip); local frame 0 is global keyframe t0 of t_glob.  Returns (B*t_local, H, W, Cout)."""
    n, h, w, c = x_ext.shape
    assert n == b * (t_local + 2) and pw.taps == 3
    out = gemm(x_ext.reshape(-1, c), pw, mode=GEMM_TEMPORAL, m=b * t_local * h * w, t=t_local, hw=h * w,
               tsrc=t_local + 2, tsrc_off=1, t0=t0, tglob=t_glob, gn_rows=h * w if gn else 0, **kw)
    return carry_gn_stats(out, out.view(b * t_local, h, w, out.shape[-1]))


# ------------------------------------------------------------------------------------------
_ws_cache = {}


SPLIT_K = policy.get("g8_split") != 0
_splitk_cache = {}


def _splitk_ws(nbytes: int, device) -> torch.Tensor:
    """Split-K scratch of the current stream (arrival counters in the first 4096 bytes: zero when handed out, left zero by every
    call; launches of one stream are ordered, other streams get their own buffer)."""
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _splitk_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        _splitk_cache[key] = ws
    return ws


def _stats_ws(frames: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < frames * 64:
        ws = torch.empty(max(frames * 64, 4096), dtype=torch.float64, device=device)
        _ws_cache[key] = ws
    return ws


class _ZeroArena:
    """Zero-initialised fp64 scratch for the fused GroupNorm statistics (double: see include/ccedit_hip.h): one memset per ~256 requests instead of one
    per producer launch (150 per network evaluation).  A slice is handed out once and never recycled — the tensors
    that carry the statistics keep the slab alive — and arenas are per (device, stream)."""
    SLAB = 256 * 34 * 64

    def __init__(self):
        self.slabs = {}

    def take(self, numel: int, device) -> torch.Tensor:
        key = (device, torch.cuda.current_stream().cuda_stream)
        slab, used = self.slabs.get(key, (None, 0))
        if slab is None or used + numel > slab.numel():
            slab, used = torch.zeros(max(self.SLAB, numel), dtype=torch.float64, device=device), 0
        self.slabs[key] = (slab, used + numel)
        return slab[used:used + numel]


_ZEROS = _ZeroArena()


def reset_stream_scratch():
    """Forget every cached scratch buffer (statistics workspace, zeroed arenas, split-K workspace) of EVERY stream.  Called at both
    ends of a HIP-graph capture: what is allocated while capturing belongs to that graph's memory pool and is zeroed by a captured
    fill, so nothing captured later may continue in it — and a capture spans several streams (the ControlNet's side stream keeps
    its own arenas): resetting only the capture stream's left the side stream's arena of an earlier graph in use by the next one,
    whose replays then accumulated GroupNorm statistics onto the previous replay's (round 3: the sampler run after the step
    benchmark produced non-finite frames)."""
    _ws_cache.clear()
    _ZEROS.slabs.clear()
    _splitk_cache.clear()


def zero_stats(frames: int, device) -> torch.Tensor:
    return _ZEROS.take(frames * 64, device).view(frames, 32, 2)


FUSE_GN_STATS = policy.on("fuse_gn_stats")      # 0: always the two-pass GroupNorm


def gn_stats_of(x: torch.Tensor, hw: int):
    """Statistics a producing GEMM left on `x` (None if it did not)."""
    st = getattr(x, "_gn_stats", None)
    return st[0] if st is not None and st[1] == hw else None


def carry_gn_stats(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """`view` is a reshape of `src` (same memory): keep the producer's statistics attached."""
    st = getattr(src, "_gn_stats", None)
    if st is not None:
        view._gn_stats = st
    return view


def groupnorm_spatial(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, silu: bool) -> torch.Tensor:
    _chk_act(x, "groupnorm_spatial")
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    st = gn_stats_of(x, h * w)
    if st is not None:
        assert st.shape[0] == n
        _launch_mem("gn_spatial_apply (statistics from the producer)", 4.0 * x.numel(), lambda: hip.check(
            hip.lib().ccedit_groupnorm_spatial_apply(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     st.data_ptr(), n, h * w, c, eps, int(silu), _stream()),
            "ccedit_groupnorm_spatial_apply"), (n, h, w, c))
        return y
    ws = _stats_ws(n, x.device)
    _launch_mem("gn_spatial stats + apply (two reads)", 6.0 * x.numel(), lambda: hip.check(
        hip.lib().ccedit_groupnorm_spatial(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                           ws.data_ptr(), n, h * w, c, eps, int(silu), _stream()),
        "ccedit_groupnorm_spatial"), (n, h, w, c))
    return y


def groupnorm_spatial_stats(x: torch.Tensor) -> torch.Tensor:
    """(sum, sum of squares) per (frame, group) of x (N, H, W, C): double (N, 32, 2).  The statistics half of `groupnorm_spatial` for
    frames whose rows are sharded over ranks: all-reduce, divide by the number of ranks, attach with `set_gn_stats`."""
    _chk_act(x, "groupnorm_spatial_stats")
    n, h, w, c = x.shape
    st = zero_stats(n, x.device)
    hip.check(hip.lib().ccedit_groupnorm_spatial_stats(x.data_ptr(), st.data_ptr(), n, h * w, c, _stream()), "ccedit_groupnorm_spatial_stats")
    return st


def set_gn_stats(x: torch.Tensor, stats: torch.Tensor):
    x._gn_stats = (stats, x.shape[1] * x.shape[2])
    return x


def groupnorm_temporal(x: torch.Tensor, b: int, t: int, gamma, beta, eps: float, silu: bool) -> torch.Tensor:
    _chk_act(x, "groupnorm_temporal")
    n, h, w, c = x.shape
    assert n == b * t
    y = torch.empty_like(x)
    _launch_mem("gn_temporal", 4.0 * x.numel(), lambda: hip.check(
        hip.lib().ccedit_groupnorm_temporal(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            b, t, h * w, c, eps, int(silu), _stream()), "ccedit_groupnorm_temporal"), (n, h, w, c))
    return y


def groupnorm_temporal_stats(x: torch.Tensor, b: int, t: int) -> torch.Tensor:
    """Partial (sum, sumsq) per (clip, pixel, group) over the local frames: fp32 (b*hw, 32, 2)."""
    _chk_act(x, "groupnorm_temporal_stats")
    n, h, w, c = x.shape
    assert n == b * t
    st = torch.empty((b * h * w, 32, 2), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().ccedit_groupnorm_temporal_stats(x.data_ptr(), st.data_ptr(), b, t, h * w, c, _stream()),
              "ccedit_groupnorm_temporal_stats")
    return st


def groupnorm_temporal_apply(x: torch.Tensor, stats: torch.Tensor, b: int, t: int, t_glob: int, gamma, beta, eps: float,
                             silu: bool, out: Optional[torch.Tensor] = None, dst_frames: int = 0, dst_off: int = 0):
    """Apply with (all-reduced) statistics over t_glob frames; optionally write into a halo-extended buffer."""
    _chk_act(x, "groupnorm_temporal_apply")
    n, h, w, c = x.shape
    y = torch.empty_like(x) if out is None else out
    if dst_frames == 0:
        dst_frames, dst_off = t, 0
    hip.check(hip.lib().ccedit_groupnorm_temporal_apply(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                        stats.data_ptr(), b, t, h * w, c, float((c // 32) * t_glob), eps,
                                                        int(silu), dst_frames, dst_off, _stream()),
              "ccedit_groupnorm_temporal_apply")
    return y


def row_stats(x2d: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """(mean, rstd) of every row: the statistics half of LayerNorm, for a GEMM that folds the normalisation (gemm(ln_stats=...))."""
    _chk_act(x2d, "row_stats")
    assert x2d.is_contiguous()
    st = torch.empty((x2d.shape[0], 2), dtype=torch.float32, device=x2d.device)
    _launch_mem("row_stats", 2.0 * x2d.numel(), lambda: hip.check(
        hip.lib().ccedit_row_stats(x2d.data_ptr(), st.data_ptr(), x2d.shape[0], x2d.shape[1], eps, _stream()), "ccedit_row_stats"))
    return st


LNF = policy.on("lnf")       # 0: LayerNorm passes in front of the 640 / 1280-channel projections (A/B)


def ln_sums_of(x: torch.Tensor):
    """(sum, sum of squares) per row that the producing GEMM left on `x` (None if it did not)."""
    return getattr(x, "_ln_sums", None)


def row_sums_applicable(m: int, pw, act: int = ACT_NONE) -> bool:
    """A Linear whose epilogue can accumulate the row sums of its output (the persistent eight-phase kernel's shapes)."""
    return LNF and LN_SUMS and m >= 4096 and pw.kpad >= 640 and pw.n >= 640 and pw.taps == 1 and not pw.geglu and act == ACT_NONE


LN_SUMS = policy.on("ln_sums")   # 0: statistics by ccedit_row_stats instead of the producer's epilogue (A/B)


def lnf_applicable(m: int, pw) -> bool:
    """ccedit_gemm's conditions for ln_stats (the persistent eight-phase kernel, csrc/gemm.hip)."""
    return LNF and pw is not None and pw.colsum is not None and m >= 4096 and pw.kpad >= 640 and pw.n >= 640 and pw.taps == 1


def layernorm(x2d: torch.Tensor, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    _chk_act(x2d, "layernorm")
    y = torch.empty_like(x2d)
    _launch_mem("layernorm", 4.0 * x2d.numel(), lambda: hip.check(
        hip.lib().ccedit_layernorm(x2d.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                   x2d.shape[0], x2d.shape[1], eps, _stream()), "ccedit_layernorm"), tuple(x2d.shape))
    return y


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int, d: int, *, batches: int, lq: int, lk: int,
              q_inner: int = 1, q_outer_rows: Optional[int] = None, q_inner_rows: int = 0, q_seq_rows: int = 1,
              kv_div: int = 1, kv_inner: int = 1, kv_outer_rows: Optional[int] = None, kv_inner_rows: int = 0,
              kv_seq_rows: int = 1, out: Optional[torch.Tensor] = None,
              seg1_len: int = 0, seg1_div: int = 1, seg1_mul: int = 0, seg1_add: int = 0, causal: bool = False,
              q_log2: bool = False) -> torch.Tensor:
    """q/k/v: 2-D row-major views [rows, >= heads*d] (may be column slices of a fused buffer).  q_log2: q already carries
    d^-0.5 * log2(e) (folded into the to_q weights by the packer)."""
    for tns in (q, k, v):
        assert tns.dtype == BF16 and tns.is_cuda and tns.stride(-1) == 1
    if out is None:
        out = torch.empty((q.shape[0], heads * d), dtype=BF16, device=q.device)
    a = CcAttnDesc()
    a.q, a.k, a.v, a.o = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.heads, a.d, a.batches, a.Lq, a.Lk = heads, d, batches, lq, lk
    a.q_inner, a.q_outer_rows, a.q_inner_rows, a.q_seq_rows = q_inner, (lq if q_outer_rows is None else q_outer_rows), q_inner_rows, q_seq_rows
    a.kv_div, a.kv_inner = kv_div, kv_inner
    a.kv_outer_rows, a.kv_inner_rows, a.kv_seq_rows = (lk if kv_outer_rows is None else kv_outer_rows), kv_inner_rows, kv_seq_rows
    a.scale = float(d) ** -0.5
    a.seg1_len, a.seg1_div, a.seg1_mul, a.seg1_add = seg1_len, seg1_div, seg1_mul, seg1_add
    a.causal = int(causal)
    a.flags = hip.ATTN_Q_LOG2 if q_log2 else 0
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(hip.lib().ccedit_attention(C.byref(a), _stream()), "ccedit_attention")
        e1.record()
        nb = 2.0 * heads * d * (2.0 * batches * lq + 2.0 * (batches // kv_div) * lk)          # q, o, k, v once
        PROFILE.add("attention", e0, e1, 4.0 * batches * heads * lq * lk * d, nb, (batches, heads, d, lq, lk))
        return out
    hip.check(hip.lib().ccedit_attention(C.byref(a), _stream()), "ccedit_attention")
    return out


# ------------------------------------------------------------------------------------------
def ncthw_to_nhwc(x: torch.Tensor, cpad: int, scale_per_b: Optional[torch.Tensor] = None, scale: float = 1.0,
                  shift: float = 0.0) -> torch.Tensor:
    """fp32 (B, C, T, H, W) -> bf16 (B*T, H, W, cpad); y = x*scale(*scale_per_b[b]) + shift."""
    assert x.dtype == torch.float32 and x.is_cuda and x.is_contiguous() and x.ndim == 5
    b, c, t, h, w = x.shape
    y = torch.empty((b * t, h, w, cpad), dtype=BF16, device=x.device)
    hip.check(hip.lib().ccedit_ncthw_to_nhwc(x.data_ptr(), y.data_ptr(), b, c, t, h, w, cpad, _ptr(scale_per_b),
                                             scale, shift, _stream()), "ccedit_ncthw_to_nhwc")
    return y


def nhwc_to_ncthw(x: torch.Tensor, b: int, t: int, c: int) -> torch.Tensor:
    """(B*T, H, W, ld) bf16|fp32 -> fp32 (B, c, T, H, W) taking the first c channels."""
    assert x.is_cuda and x.is_contiguous() and x.ndim == 4
    n, h, w, ld = x.shape
    y = torch.empty((b, c, t, h, w), dtype=torch.float32, device=x.device)
    hip.check(hip.lib().ccedit_nhwc_to_ncthw(x.data_ptr(), int(x.dtype == torch.float32), ld, y.data_ptr(), b, c, t, h, w,
                                             _stream()), "ccedit_nhwc_to_ncthw")
    return y


def cat_add(a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor], gn: bool = False) -> torch.Tensor:
    """(..., C1) ++ ((..., C2) + (..., C2)) along channels.  gn=True (4-D NHWC input): also accumulate the
    GroupNorm(32) statistics of the result for the `groupnorm_spatial` that follows."""
    _chk_act(a, "cat_add.a"), _chk_act(b, "cat_add.b")
    if a.shape[:-1] != b.shape[:-1] or (c is not None and c.shape != b.shape):
        raise ValueError(f"cat_add: mismatched shapes {tuple(a.shape)}, {tuple(b.shape)}"
                         + (f", {tuple(c.shape)}" if c is not None else "")
                         + " (frame sizes must be multiples of 64 pixels so that the UNet's 3 down/up levels agree)")
    c1, c2 = a.shape[-1], b.shape[-1]
    out = torch.empty((*a.shape[:-1], c1 + c2), dtype=BF16, device=a.device)
    if gn and FUSE_GN_STATS and a.dim() == 4 and (c1 + c2) % 32 == 0 and c1 + c2 <= 2560:
        n, hw = a.shape[0], a.shape[1] * a.shape[2]
        stats = zero_stats(n, a.device)
        nb = 2.0 * (a.numel() + b.numel() * (2 if c is not None else 1) + out.numel())
        _launch_mem("cat_add_gn", nb, lambda: hip.check(
            hip.lib().ccedit_cat_add_gn(a.data_ptr(), b.data_ptr(), _ptr(c), out.data_ptr(), stats.data_ptr(), n, hw,
                                        c1, c2, _stream()), "ccedit_cat_add_gn"), (n, hw, c1, c2))
        out._gn_stats = (stats, hw)
        return out
    nb = 2.0 * (a.numel() + b.numel() * (2 if c is not None else 1) + out.numel())
    _launch_mem("cat_add", nb, lambda: hip.check(
        hip.lib().ccedit_cat_add(a.data_ptr(), b.data_ptr(), _ptr(c), out.data_ptr(), a.numel() // c1, c1, c2,
                                 _stream()), "ccedit_cat_add"))
    return out


def copy_row_blocks(src: torch.Tensor, dst: torch.Tensor, blocks: torch.Tensor, max_rows: int, add: Optional[torch.Tensor] = None):
    """dst[blocks[s,1] : +blocks[s,2]] = src[blocks[s,0] : +blocks[s,2]] (+ add at the destination rows) for every block s.
    src / dst / add: 2-D row-major with the same row width; blocks: int64 (n, 3) on the device."""
    assert src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.shape[1] == dst.shape[1]
    assert blocks.dtype == torch.int64 and blocks.is_cuda and blocks.is_contiguous() and blocks.shape[1] == 3
    if add is not None:
        assert add.dtype == BF16 and dst.dtype == BF16 and add.is_contiguous() and add.shape == dst.shape
    hip.check(hip.lib().ccedit_copy_row_blocks(src.data_ptr(), dst.data_ptr(), _ptr(add), blocks.data_ptr(), blocks.shape[0], max_rows,
                                               src.shape[1] * src.element_size(), _stream()), "ccedit_copy_row_blocks")
    return dst


def copy_2d_blocks(src: torch.Tensor, dst: torch.Tensor, blocks: torch.Tensor, rows: int, row_bytes: int):
    """For every block s: dst_bytes[blocks[s,1] + r * dst_pitch : + row_bytes] = src_bytes[blocks[s,0] + r * src_pitch : + row_bytes],
    r < rows.  src / dst: 2-D tensors with contiguous rows (the pitches are their row strides in bytes); blocks: int64 (n, 2) on the
    device, byte offsets from the tensors' first element."""
    assert src.is_cuda and dst.is_cuda and src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1
    assert blocks.dtype == torch.int64 and blocks.is_cuda and blocks.is_contiguous() and blocks.shape[1] == 2
    hip.check(hip.lib().ccedit_copy_2d_blocks(src.data_ptr(), dst.data_ptr(), blocks.data_ptr(), blocks.shape[0], rows, row_bytes,
                                              src.stride(0) * src.element_size(), dst.stride(0) * dst.element_size(), _stream()),
              "ccedit_copy_2d_blocks")
    return dst


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk_act(a, "add.a"), _chk_act(b, "add.b")
    y = torch.empty_like(a) if out is None else out
    hip.check(hip.lib().ccedit_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "ccedit_add")
    return y


def silu(x: torch.Tensor) -> torch.Tensor:
    _chk_act(x, "silu")
    y = torch.empty_like(x)
    hip.check(hip.lib().ccedit_silu(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "ccedit_silu")
    return y


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    assert t.dtype == torch.int64 and t.is_cuda
    out = torch.empty((t.shape[0], dim), dtype=BF16, device=t.device)
    hip.check(hip.lib().ccedit_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, dim, _stream()),
              "ccedit_timestep_embedding")
    return out


def embedding_lookup(ids: torch.Tensor, tok: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """ids int64 (B, L) -> bf16 [B*L, C] = tok[ids] + pos (CLIP text embeddings)."""
    assert ids.dtype == torch.int64 and ids.is_cuda and ids.is_contiguous() and tok.dtype == pos.dtype == torch.float32
    b, l = ids.shape
    assert pos.shape[0] == l and tok.shape[1] == pos.shape[1]
    if int(ids.min()) < 0 or int(ids.max()) >= tok.shape[0]:
        raise ValueError(f"token id outside [0, {tok.shape[0]})")
    out = torch.empty((b * l, tok.shape[1]), dtype=BF16, device=ids.device)
    hip.check(hip.lib().ccedit_embedding_lookup(ids.data_ptr(), tok.data_ptr(), pos.data_ptr(), out.data_ptr(), b * l, l,
                                                tok.shape[1], tok.shape[0], _stream()), "ccedit_embedding_lookup")
    return out


def gaussian_sample(moments: torch.Tensor, noise: torch.Tensor, zc: int, scale: float = 1.0) -> torch.Tensor:
    """moments: fp32 [frames*hw, >=2*zc] channels-last rows [mean | logvar]; noise: fp32 (frames, zc, h, w)."""
    assert moments.dtype == torch.float32 and noise.dtype == torch.float32 and noise.is_contiguous() and moments.stride(-1) == 1
    n, c, h, w = noise.shape
    assert c == zc and moments.shape[0] == n * h * w
    out = torch.empty_like(noise)
    hip.check(hip.lib().ccedit_gaussian_sample(moments.data_ptr(), noise.data_ptr(), out.data_ptr(), n, zc, h * w,
                                               moments.stride(0), scale, _stream()), "ccedit_gaussian_sample")
    return out


def mask_blend(x: torch.Tensor, z: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """x * mask + z * (1 - mask), fp32 tensors of one shape."""
    for tns in (x, z, mask):
        assert tns.dtype == torch.float32 and tns.is_cuda and tns.is_contiguous() and tns.shape == x.shape
    y = torch.empty_like(x)
    hip.check(hip.lib().ccedit_mask_blend(x.data_ptr(), z.data_ptr(), mask.data_ptr(), y.data_ptr(), x.numel(), _stream()),
              "ccedit_mask_blend")
    return y


def cfg_denoise(x: torch.Tensor, eps2: torch.Tensor, sigma: float, scale: float) -> torch.Tensor:
    """x: fp32 latent (n elems); eps2: fp32 [2, n] (uncond first). -> denoised (guided) fp32."""
    assert x.dtype == torch.float32 and eps2.dtype == torch.float32 and x.is_contiguous() and eps2.is_contiguous()
    den = torch.empty_like(x)
    hip.check(hip.lib().ccedit_cfg_denoise(x.data_ptr(), eps2.data_ptr(), den.data_ptr(), x.numel(), sigma, scale, _stream()),
              "ccedit_cfg_denoise")
    return den


def axpby(x: torch.Tensor, z: torch.Tensor, a: float, b: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert x.dtype == torch.float32 and z.dtype == torch.float32 and x.is_contiguous() and z.is_contiguous()
    y = torch.empty_like(x) if out is None else out
    hip.check(hip.lib().ccedit_axpby(x.data_ptr(), z.data_ptr(), y.data_ptr(), x.numel(), a, b, _stream()), "ccedit_axpby")
    return y


def softmax_rows(s: torch.Tensor, cols: int, cols_pad: int, scale: float) -> torch.Tensor:
    """s: fp32 [rows, >= cols] -> bf16 [rows, cols_pad] = softmax(s[:, :cols] * scale), zero pad."""
    assert s.dtype == torch.float32 and s.is_cuda and s.stride(-1) == 1
    p = torch.empty((s.shape[0], cols_pad), dtype=BF16, device=s.device)
    hip.check(hip.lib().ccedit_softmax_rows(s.data_ptr(), p.data_ptr(), s.shape[0], cols, cols_pad, s.stride(0), cols_pad,
                                            scale, _stream()), "ccedit_softmax_rows")
    return p
