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
    colsum: Optional[torch.Tensor] = None   # fp32 [n]: row sums of the bf16 weights as packed (CcGemmDesc.ln_colsum), folded-LayerNorm weights only
    wfrag: Optional[torch.Tensor] = None    # bf16 [Opad * Kpad]: `w` in MFMA A-fragment order (CcGemmDesc.Wfrag), K = 320 / 640 / 960 only

    def to(self, device):
        self.w = self.w.to(device)
        if self.bias is not None:
            self.bias = self.bias.to(device)
        if self.colsum is not None:
            self.colsum = self.colsum.to(device)
        if self.wfrag is not None:
            self.wfrag = self.wfrag.to(device)
        return self


FRAG_KPADS = (320, 640, 960)      # weight widths the register-resident kernels serve (lin320.hip, lin640.hip, temp320.hip)


def fragment_order(w: torch.Tensor) -> torch.Tensor:
    """[rows][Kpad] -> the 16 x 32 blocks of CcGemmDesc.Wfrag: block (t, s) = rows 16 t .. + 15, k 32 s .. + 31 as 64 lanes x 8
    elements, lane (g, c) = W[16 t + c][32 s + 8 g .. + 7] — what one v_mfma_f32_16x16x32_bf16 A operand holds, contiguous."""
    rows, kpad = w.shape
    assert rows % 16 == 0 and kpad % 32 == 0
    return w.view(rows // 16, 16, kpad // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


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
    pw = PackedWeight(buf, bb, n, (o // 2) if geglu else o, cin, taps, kpad, ksize, geglu, 2.0 * o * i * taps, korder)
    if kpad in FRAG_KPADS and taps in (1, 3) and ksize in (1, 3) and w.ndim != 4:
        pw.wfrag = fragment_order(buf)
    return pw


def pack_upsample_parities(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, device: Optional[torch.device] = None):
    """conv3x3(nearest_upsample_2x(x)) (Upsample.forward, openaimodel.py:204-217 / 254-263) as four 2 x 2 convolutions on x itself, one
    per output parity (py, px): an output pixel (2y + py, 2x + px) reads up-sampled rows 2y + py - 1 + ky, i.e. source rows
    y - 1, y, y for py = 0 and y, y, y + 1 for py = 1 — taps that land on the same source pixel are added up in fp32 BEFORE the bf16
    rounding (one rounding per merged tap; the nine-tap path rounds each of the merged taps separately).  4/9 of the multiply-adds.
    Returns [PackedWeight] in CcGemmDesc.subpix order p = 2 py + px; `flops_per_row` stays the ALGORITHMIC 2 * 9 * O * I per output
    pixel of the reference convolution (what bench.py prices), not the 2 * 4 * O * I executed."""
    w = weight.detach().to(torch.float32)
    assert w.ndim == 4 and w.shape[2] == 3 and w.shape[3] == 3
    merge = (((0,), (1, 2)), ((0, 1), (2,)))               # parity -> for each of the two window positions, the 3x3 taps it collects
    out = []
    for py in range(2):
        for px in range(2):
            w2 = torch.zeros(w.shape[0], w.shape[1], 2, 2, dtype=torch.float32, device=w.device)
            for dy in range(2):
                for dx in range(2):
                    for ky in merge[py][dy]:
                        for kx in merge[px][dx]:
                            w2[:, :, dy, dx] += w[:, :, ky, kx]
            pw = pack_weight(w2, bias, device=device)
            pw.flops_per_row = 2.0 * w.shape[0] * w.shape[1] * 9
            out.append(pw)
    return out


def pack_concat(weights: Sequence[torch.Tensor], biases: Optional[Sequence[Optional[torch.Tensor]]] = None,
                device: Optional[torch.device] = None) -> PackedWeight:
    """Stack several (O_i, I) projections along O (fused q|k|v GEMM)."""
    w = torch.cat([x.detach().to(torch.float32) for x in weights], dim=0)
    b = None
    if biases is not None and any(x is not None for x in biases):
        b = torch.cat([torch.zeros(wi.shape[0]) if bi is None else bi.detach().float() for wi, bi in zip(weights, biases)])
    return pack_weight(w, b, device=device)


def fold_layernorm(weights: Sequence[torch.Tensor], biases: Optional[Sequence[Optional[torch.Tensor]]], gamma: torch.Tensor,
                   beta: torch.Tensor, device: Optional[torch.device] = None, geglu: bool = False) -> PackedWeight:
    """Projections of a LayerNorm output, Linear(gamma * xhat + beta) = (W diag(gamma)) xhat + (b + W beta), packed for the
    GEMM that normalises its rows itself (CcGemmDesc.ln_eps, K = 320) or takes the row statistics and applies them in its
    epilogue (CcGemmDesc.ln_stats: rstd (W' x - mean W' 1) + b', for which `colsum` = W' 1 of the bf16 weights as stored):
    the affine half of the LayerNorm lives in the weights."""
    ws = [w.detach().float() for w in weights]
    g, bt = gamma.detach().float().to(ws[0].device), beta.detach().float().to(ws[0].device)
    bs = [None] * len(ws) if biases is None else list(biases)
    wf = torch.cat([w * g[None, :] for w in ws], dim=0)
    bf = torch.cat([(torch.zeros(w.shape[0], device=w.device) if b is None else b.detach().float().to(w.device)) + w @ bt
                    for w, b in zip(ws, bs)])
    pw = pack_weight(wf, bf, geglu=geglu)
    pw.colsum = pw.w[:pw.n].float().sum(dim=1).contiguous()
    return pw if device is None else pw.to(device)


# ------------------------------------------------------------------------------------------
# fused feed-forward (dim 320): the weight stream of csrc/ff320.hip
# ------------------------------------------------------------------------------------------
FF320_DIM, FF320_INNER, FF320_CHUNKS, FF320_CHUNK_BYTES = 320, 1280, 40, 64 * 1024


@dataclass
class PackedFF320:
    stream: torch.Tensor            # uint8 [42 * 65536]: per pipeline iteration 40 GEMM1 + 20 GEMM2 MFMA A-fragments + b1'
    b2p: torch.Tensor               # fp32 [320]: b2 in accumulator order [tile 10][lane half 2][register 16]
    flops_per_row: float = 2.0 * 320 * 2560 + 2.0 * 1280 * 320
    # block tail (pack_ff320_tail): stream = [4 prologue chunks | 42 | 4 epilogue chunks]; bop / bpp = b_o / b_p in accumulator order
    bop: Optional[torch.Tensor] = None
    bpp: Optional[torch.Tensor] = None
    tail_of: Optional["PackedFF320"] = None     # the plain pack these chunks were built around (kept for the fallback launch)


def ff320_acc_row(r: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    """Row of a v_mfma_f32_32x32x16 C/D tile held by accumulator register r (0..15) of a lane in half hi (= lane >> 5)."""
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def ff320_out_channel(t: torch.Tensor, i: torch.Tensor) -> torch.Tensor:
    """Output channel computed by row i (0..31) of GEMM2 accumulator tile t (0..9).  Row i is register r = (i & 3) + 4 (i >> 3)
    of lane half hi = (i >> 2) & 1; registers 0..7 / 8..15 of a lane must be channels 32 t + 8 hi .. + 7 / 32 t + 16 + 8 hi .. + 7 —
    the channels of X fragments 2 t / 2 t + 1 of the same lane (residual initialisation and 16-byte stores without a shuffle)."""
    hi, r = (i >> 2) & 1, (i & 3) + 4 * (i >> 3)
    return 32 * t + torch.where(r < 8, 8 * hi + r, 16 + 8 * hi + r - 8)


def pack_ff320(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, ln_g: Optional[torch.Tensor],
               ln_b: Optional[torch.Tensor], device: Optional[torch.device] = None) -> PackedFF320:
    """w1 (2560, 320) = GEGLU proj [value rows | gate rows] (attention.py:118-126), b1 (2560,), w2 (320, 1280),
    b2 (320,), LayerNorm gamma / beta (320,) or None (no normalisation).  LayerNorm's affine part is folded into the
    weights (W1' = W1 diag(gamma), b1' = b1 + W1 beta); the kernel normalises x itself."""
    D, H, Q = FF320_DIM, FF320_INNER, FF320_CHUNKS
    if tuple(w1.shape) != (2 * H, D) or tuple(w2.shape) != (D, H):
        raise ValueError(f"pack_ff320: expected w1 (2560, 320), w2 (320, 1280); got {tuple(w1.shape)}, {tuple(w2.shape)}")
    w1 = w1.detach().double().cpu()
    w2 = w2.detach().float().cpu()
    b1 = torch.zeros(2 * H, dtype=torch.float64) if b1 is None else b1.detach().double().cpu()
    b2 = torch.zeros(D) if b2 is None else b2.detach().float().cpu()
    gam = torch.ones(D, dtype=torch.float64) if ln_g is None else ln_g.detach().double().cpu()
    bet = torch.zeros(D, dtype=torch.float64) if ln_b is None else ln_b.detach().double().cpu()
    w1g = (w1 * gam[None, :]).float().to(torch.bfloat16)                  # W1 diag(gamma), what the MFMA multiplies
    b1p = (b1 + w1 @ bet).float()                                         # b1 + W1 beta
    w2b = w2.to(torch.bfloat16)

    lane = torch.arange(64)
    i, hi = lane & 31, lane >> 5
    e = torch.arange(8)
    # GEMM1 A fragments in use order [q][k-step s][kind (value | gate)][lane][e]: row = hidden unit 32 q + (lane & 31)
    q = torch.arange(Q)[:, None, None, None, None]
    s = torch.arange(20)[None, :, None, None, None]
    kind = torch.arange(2)[None, None, :, None, None]
    row1 = kind * H + 32 * q + i[None, None, None, :, None]                                       # [q,1,kind,lane,1]
    col1 = 16 * s + 8 * hi[None, None, None, :, None] + e[None, None, None, None, :]             # [1,s,1,lane,e]
    shp = (Q, 20, 2, 64, 8)
    f1 = w1g[row1.expand(shp), col1.expand(shp)]
    # GEMM2 A fragments [q][k-step kappa][out tile t][lane][e] (kappa-major: consecutive MFMAs use different accumulators): K position (kappa, lane half, e) is the hidden unit whose GEGLU
    # result sits in accumulator register 8 kappa + e of that lane half
    kap = torch.arange(2)[None, :, None, None, None]
    t = torch.arange(10)[None, None, :, None, None]
    q2 = torch.arange(Q)[:, None, None, None, None]
    row2 = ff320_out_channel(t, i[None, None, None, :, None])                                     # [1,1,t,lane,1]
    col2 = 32 * q2 + ff320_acc_row(8 * kap + e[None, None, None, None, :], hi[None, None, None, :, None])   # [q,kap,1,lane,e]
    shp2 = (Q, 2, 10, 64, 8)
    f2 = w2b[row2.expand(shp2), col2.expand(shp2)]
    # b1' in accumulator order [q][kind][lane half][register r]
    r16 = torch.arange(16)
    rowa = (torch.arange(2)[None, :, None, None] * H + 32 * torch.arange(Q)[:, None, None, None]
            + ff320_acc_row(r16[None, None, None, :], torch.arange(2)[None, None, :, None]))       # [q,kind,hi,16]
    aux = b1p[rowa].contiguous()
    # Stream chunk c = 0 .. 41 is what iteration c of the kernel's three-stage software pipeline reads: per k-step s the value
    # and gate GEMM1 fragments of hidden chunk c and GEMM2 fragment s = (kappa, out tile) of chunk c - 2, then b1' of chunk c;
    # the missing neighbours of the first / last two iterations stay zero.
    frags = torch.zeros(Q + 2, 20, 3, 64, 8, dtype=torch.bfloat16)
    frags[:Q, :, 0:2] = f1
    frags[2:, :, 2] = f2.reshape(Q, 20, 64, 8)
    stream = torch.zeros(Q + 2, FF320_CHUNK_BYTES, dtype=torch.uint8)
    stream[:, : 60 * 1024] = frags.contiguous().view(torch.uint8).reshape(Q + 2, -1)
    stream[:Q, 60 * 1024: 60 * 1024 + 256] = aux.view(torch.uint8).reshape(Q, -1)
    tt, hh, rr = torch.meshgrid(torch.arange(10), torch.arange(2), torch.arange(16), indexing="ij")
    b2p = b2[32 * tt + torch.where(rr < 8, 8 * hh + rr, 16 + 8 * hh + rr - 8)].reshape(-1).contiguous()
    dev = device if device is not None else torch.device("cpu")
    return PackedFF320(stream.reshape(-1).to(dev), b2p.to(dev))


FF320_PE_FRAGS, FF320_PE_CHUNKS = 50, 4      # prologue / epilogue GEMM: 4 chunks of 5 k-steps x 10 output tiles


def _ff320_acc_order(b: torch.Tensor) -> torch.Tensor:
    """A 320-vector in the accumulator order of the kernel: [tile 10][lane half 2][register 16]."""
    tt, hh, rr = torch.meshgrid(torch.arange(10), torch.arange(2), torch.arange(16), indexing="ij")
    return b[32 * tt + torch.where(rr < 8, 8 * hh + rr, 16 + 8 * hh + rr - 8)].reshape(-1).contiguous()


def _ff320_proj_chunks(w: torch.Tensor) -> torch.Tensor:
    """A 320 x 320 projection as 4 stream chunks of 50 MFMA A-fragments: fragment j of chunk Q is (k-step 5 Q + j // 10, output tile
    j % 10), its row for lane i the output channel ff320_out_channel(tile, i) — the row permutation that makes accumulator registers
    0..7 / 8..15 of a lane the channels of B fragments 2 t / 2 t + 1 of the same lane — and its 8 elements input channels
    16 s + 8 (lane >> 5) .. + 7."""
    wb = w.detach().float().cpu().to(torch.bfloat16)
    assert tuple(wb.shape) == (FF320_DIM, FF320_DIM)
    lane = torch.arange(64)
    i, hi = lane & 31, lane >> 5
    e = torch.arange(8)
    s = torch.arange(20)[:, None, None, None]
    t = torch.arange(10)[None, :, None, None]
    row = ff320_out_channel(t, i[None, None, :, None])                      # [1, t, lane, 1]
    col = 16 * s + 8 * hi[None, None, :, None] + e[None, None, None, :]     # [s, 1, lane, e]
    shp = (20, 10, 64, 8)
    fr = wb[row.expand(shp), col.expand(shp)]                               # [s][t][lane][e] = stream order (s-major, tile-minor)
    stream = torch.zeros(FF320_PE_CHUNKS, FF320_CHUNK_BYTES, dtype=torch.uint8)
    fr = fr.reshape(FF320_PE_CHUNKS, FF320_PE_FRAGS, 64, 8).contiguous()
    stream[:, : FF320_PE_FRAGS * 1024] = fr.view(torch.uint8).reshape(FF320_PE_CHUNKS, -1)
    return stream


def pack_ff320_tail(ff: PackedFF320, wo: torch.Tensor, bo: Optional[torch.Tensor], wp: Optional[torch.Tensor] = None,
                    bp: Optional[torch.Tensor] = None, device: Optional[torch.device] = None) -> PackedFF320:
    """The weight stream of the dim-320 BLOCK TAIL (csrc/ff320.hip, PRO / EPI):  tok = wo a + bo + res;  tok += FF(LN(tok));
    out = wp tok + bp + res2.  wo / wp: (320, 320) reference-layout Linear or 1 x 1 Conv weights (to_out.0 and proj_out of
    attention.py:695-716 / 758-761, 865-889, 1141-1208); wp None = no epilogue GEMM (out = the feed-forward's result)."""
    parts = [_ff320_proj_chunks(wo.reshape(FF320_DIM, FF320_DIM)), ff.stream.detach().cpu().reshape(-1, FF320_CHUNK_BYTES)]
    if wp is not None:
        parts.append(_ff320_proj_chunks(wp.reshape(FF320_DIM, FF320_DIM)))
    dev = device if device is not None else ff.stream.device
    zero = torch.zeros(FF320_DIM)
    bop = _ff320_acc_order(zero if bo is None else bo.detach().float().cpu())
    bpp = None if wp is None else _ff320_acc_order(zero if bp is None else bp.detach().float().cpu())
    extra = 2.0 * FF320_DIM * FF320_DIM * (1 if wp is None else 2)
    return PackedFF320(torch.cat(parts).reshape(-1).to(dev), ff.b2p.to(dev), ff.flops_per_row + extra, bop.to(dev),
                       None if bpp is None else bpp.to(dev), ff)
