"""CPU check of the fused feed-forward's weight stream (ccedit_amd/packing.py:pack_ff320) against a lane-level
emulation of csrc/ff320.hip: every index permutation the kernel relies on (GEMM1 C registers == GEMM2 B operands,
GEMM2 accumulators aligned with the X fragments, LayerNorm's affine part folded into W1' / b1', the 42-iteration
three-stage software pipeline) is exercised with the v_mfma_f32_32x32x16_bf16 register layouts (cdna_hip_programming.md §3):
    A operand: lane l holds A[i = l & 31][k = 8 (l >> 5) + e]      B operand: lane l holds B[k = 8 (l >> 5) + e][j = l & 31]
    C / D:     lane l, register r holds D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31]
The emulation follows the kernel statement by statement on one wave (32 tokens); the expected value is the reference
formula  x + W2 GEGLU(W1 LN(x) + b1) + b2  (attention.py:115-141, 695-716) in float64."""
import math

import numpy as np
import torch

from ccedit_amd.packing import FF320_CHUNK_BYTES, pack_ff320, pack_ff320_tail

L = np.arange(64)
N_, HI_ = L & 31, L >> 5


def mfma32(a_frag, b_frag, c_frag):
    """a_frag, b_frag: [64 lanes][8] (bf16 values); c_frag: [64][16] -> D = A.B + C in the register layouts above."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5): 8 * (l >> 5) + 8] = a_frag[l]
        B[8 * (l >> 5): 8 * (l >> 5) + 8, l & 31] = b_frag[l]
    D = A @ B
    out = c_frag.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


def gelu(v):
    return 0.5 * v * (1.0 + np.vectorize(math.erf)(v / np.sqrt(2.0)))


def to_frags(x32):
    """B-operand fragments of a wave's 32 rows: xf[s][lane][e] = x[lane & 31][16 s + 8 hi + e]."""
    xf = np.zeros((20, 64, 8))
    for s in range(20):
        for l in range(64):
            xf[s, l] = x32[l & 31, 16 * s + 8 * (l >> 5): 16 * s + 8 * (l >> 5) + 8]
    return xf


def acc_rows(acc2):
    """Accumulator tile registers -> [32 tokens][320 channels] (the kernel's store / the fragment layout of the next GEMM)."""
    out = np.zeros((32, 320))
    for t in range(10):
        for l in range(64):
            n, hi = l & 31, l >> 5
            out[n, 32 * t + 8 * hi: 32 * t + 8 * hi + 8] = acc2[t, l, :8]
            out[n, 32 * t + 16 + 8 * hi: 32 * t + 16 + 8 * hi + 8] = acc2[t, l, 8:]
    return out


def emulate_proj(chunks, bias_p, xf, res32):
    """Prologue / epilogue GEMM of the block tail: 4 chunks of 50 fragments (k-step 5 Q + j // 10, out tile j % 10), accumulators
    from the bias in accumulator order, residual rows added after the first chunk.  Returns the accumulators [10][64][16]."""
    b = bias_p.numpy().reshape(10, 2, 16)
    acc = np.zeros((10, 64, 16))
    for t in range(10):
        acc[t] = b[t][HI_]
    rfr = to_frags(res32)
    for q in range(4):
        fr = torch.from_numpy(chunks[q][: 50 * 1024].copy()).view(torch.bfloat16).float().numpy().reshape(50, 64, 8)
        for j in range(50):
            acc[j % 10] = mfma32(fr[j], xf[5 * q + j // 10], acc[j % 10])
        if q == 0:
            for t in range(10):
                acc[t][:, :8] += rfr[2 * t]
                acc[t][:, 8:] += rfr[2 * t + 1]
    return acc


def acc_to_frags(acc):
    xf = np.zeros((20, 64, 8))
    for t in range(10):
        xf[2 * t] = bf16_round(acc[t][:, :8])
        xf[2 * t + 1] = bf16_round(acc[t][:, 8:])
    return xf


def emulate_tail(pk, a32, res32, res2_32, eps):
    """The block-tail launch on one wave, statement by statement (csrc/ff320.hip, PRO [+ EPI])."""
    n_ch = pk.stream.numel() // FF320_CHUNK_BYTES
    stream = pk.stream.numpy().reshape(n_ch, FF320_CHUNK_BYTES)
    assert n_ch == (50 if pk.bpp is not None else 46)
    xf = acc_to_frags(emulate_proj(stream[:4], pk.bop, to_frags(a32), res32))
    acc2 = emulate_ff(stream[4:46], pk.b2p, xf, eps, True)
    if pk.bpp is None:
        return acc_rows(acc2)
    return acc_rows(emulate_proj(stream[46:], pk.bpp, acc_to_frags(acc2), res2_32))


def emulate_wave(stream, b2p, x32, eps, ln=True):
    """x32: [32][320] bf16-representable floats.  Returns [32][320]."""
    return acc_rows(emulate_ff(stream.numpy().reshape(42, FF320_CHUNK_BYTES), b2p, to_frags(x32), eps, ln))


def emulate_ff(stream, b2p, xf, eps, ln):
    """The 42 feed-forward chunks on fragments xf [20][64][8]; returns the GEMM2 accumulators."""
    sm = xf.sum(axis=(0, 2))
    mu = (sm + sm[L ^ 32]) / 320.0
    sq = ((xf - mu[None, :, None]) ** 2).sum(axis=(0, 2))
    rs = 1.0 / np.sqrt((sq + sq[L ^ 32]) / 320.0 + eps) if ln else np.ones(64)
    rm = rs * mu if ln else np.zeros(64)
    b2 = b2p.numpy().reshape(10, 2, 16)
    acc2 = np.zeros((10, 64, 16))
    for t in range(10):
        for l in range(64):
            for r in range(16):
                acc2[t, l, r] = b2[t, l >> 5, r] + xf[2 * t + (r >> 3), l, r & 7]
    xf = bf16_round(rs[None, :, None] * xf - rm[None, :, None])
    # three-stage pipeline: iteration c = GEMM1 of chunk c + GEMM2 of chunk c - 2 (fragments interleaved per k-step) beside the
    # GEGLU of chunk c - 1; register sets alternate with the parity of c
    hf = np.zeros((2, 2, 64, 8))
    acc1 = np.zeros((2, 2, 64, 16))
    for c in range(42):
        par = c & 1
        ch = stream[c]
        fr = torch.from_numpy(ch[: 60 * 1024].copy()).view(torch.bfloat16).float().numpy().reshape(20, 3, 64, 8)
        aux = np.frombuffer(ch[60 * 1024: 60 * 1024 + 256].tobytes(), dtype=np.float32).reshape(2, 2, 16)
        prev = acc1[1 - par].copy()                                                  # chunk c - 1, complete
        for s in range(20):
            for kind in range(2):
                cin = aux[kind][HI_].astype(np.float64) if s == 0 else acc1[par, kind]
                acc1[par, kind] = mfma32(fr[s, kind], xf[s], cin)
            acc2[s % 10] = mfma32(fr[s, 2], hf[par, s // 10], acc2[s % 10])
        hn = bf16_round(prev[0] * gelu(prev[1]))
        hf[1 - par] = np.stack([hn[:, :8], hn[:, 8:]])
    return acc2


def _weights(seed):
    gen = torch.Generator().manual_seed(seed)
    w1 = torch.randn(2560, 320, generator=gen) * 0.05
    b1 = torch.randn(2560, generator=gen) * 0.1
    w2 = torch.randn(320, 1280, generator=gen) * 0.03
    b2 = torch.randn(320, generator=gen) * 0.1
    lg = 1.0 + 0.2 * torch.randn(320, generator=gen)
    lb = 0.1 * torch.randn(320, generator=gen)
    x = (torch.randn(32, 320, generator=gen) * 1.5 + 0.7).to(torch.bfloat16).float()
    return w1, b1, w2, b2, lg, lb, x


def test_weight_stream_matches_kernel_dataflow():
    w1, b1, w2, b2, lg, lb, x = _weights(7)
    pk = pack_ff320(w1, b1, w2, b2, lg, lb)
    assert pk.stream.numel() == 42 * FF320_CHUNK_BYTES and pk.stream.dtype == torch.uint8 and pk.b2p.shape == (320,)
    got = emulate_wave(pk.stream, pk.b2p, x.numpy().astype(np.float64), 1e-5)
    # expected with the SAME bf16 quantities the kernel multiplies, so that only the dataflow / index maps are under test
    xd = x.double()
    xn = torch.from_numpy(bf16_round(((xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5)).numpy()))
    w1g = (w1.double() * lg.double()[None]).float().to(torch.bfloat16)
    pre = xn @ w1g.double().T + (b1.double() + w1.double() @ lb.double()).float().double()
    v, gate = pre.chunk(2, dim=-1)
    h = torch.from_numpy(bf16_round((v * torch.nn.functional.gelu(gate)).numpy()))
    want = xd + h @ w2.to(torch.bfloat16).double().T + b2.double()
    assert np.abs(got - want.numpy()).max() < 1e-5
    # and against the plain reference formula (fp32 weights): only bf16 rounding of weights / operands / hidden apart
    hh = torch.nn.functional.layer_norm(xd, (320,), lg.double(), lb.double(), 1e-5)
    v, gate = (hh @ w1.double().T + b1.double()).chunk(2, dim=-1)
    ref = (xd + (v * torch.nn.functional.gelu(gate)) @ w2.double().T + b2.double()).numpy()
    rel = np.sqrt(((got - ref) ** 2).mean() / ((ref - x.numpy()) ** 2).mean())
    assert rel < 1e-2, rel


def test_weight_stream_without_layernorm():
    w1, b1, w2, b2, _, _, x = _weights(8)
    pk = pack_ff320(w1, b1, w2, b2, None, None)
    got = emulate_wave(pk.stream, pk.b2p, x.numpy().astype(np.float64), 1e-5, ln=False)
    v, gate = (x.double() @ w1.to(torch.bfloat16).double().T + b1.double()).chunk(2, dim=-1)
    h = torch.from_numpy(bf16_round((v * torch.nn.functional.gelu(gate)).numpy()))
    want = x.double() + h @ w2.to(torch.bfloat16).double().T + b2.double()
    assert np.abs(got - want.numpy()).max() < 1e-5


def _ff_ref(xd, w1, b1, w2, b2, lg, lb):
    hh = torch.nn.functional.layer_norm(xd, (320,), lg.double(), lb.double(), 1e-5)
    v, gate = (hh @ w1.double().T + b1.double()).chunk(2, dim=-1)
    return xd + (v * torch.nn.functional.gelu(gate)) @ w2.double().T + b2.double()


def test_block_tail_stream_matches_kernel_dataflow():
    """pack_ff320_tail: prologue chunks (to_out), the 42 feed-forward chunks, epilogue chunks (proj_out) — the row permutation that
    turns a projection's accumulators into the next GEMM's B fragments, the accumulator-order biases and the chunk order, against
    out = W_p (tok + FF(LN(tok))) + b_p + x_in,  tok = W_o a + b_o + res  (attention.py:695-716, 865-889) in float64."""
    w1, b1, w2, b2, lg, lb, a = _weights(9)
    gen = torch.Generator().manual_seed(19)
    wo, wp = torch.randn(320, 320, generator=gen) * 0.05, torch.randn(320, 320, 1, 1, generator=gen) * 0.05
    bo, bp = torch.randn(320, generator=gen) * 0.1, torch.randn(320, generator=gen) * 0.1
    res, res2 = ((torch.randn(32, 320, generator=gen) * 1.2).to(torch.bfloat16).float() for _ in range(2))
    base = pack_ff320(w1, b1, w2, b2, lg, lb)
    for with_epi in (True, False):
        pk = pack_ff320_tail(base, wo, bo, wp if with_epi else None, bp if with_epi else None)
        assert pk.stream.numel() == (50 if with_epi else 46) * FF320_CHUNK_BYTES and pk.bop.shape == (320,) and pk.tail_of is base
        got = emulate_tail(pk, a.numpy().astype(np.float64), res.numpy().astype(np.float64), res2.numpy().astype(np.float64), 1e-5)
        tok = a.double() @ wo.double().T + bo.double() + res.double()
        t2 = _ff_ref(tok, w1, b1, w2, b2, lg, lb)
        ref = (t2 @ wp.reshape(320, 320).double().T + bp.double() + res2.double()) if with_epi else t2
        rel = float(np.sqrt(((got - ref.numpy()) ** 2).mean() / (ref.numpy() ** 2).mean()))
        assert rel < 1e-2, (with_epi, rel)          # bf16 weights / operands / two intermediate roundings apart
        # with the SAME bf16 quantities the kernel multiplies for the projections (FF as emulated): the index maps alone
        tokb = torch.from_numpy(bf16_round((a.double() @ wo.to(torch.bfloat16).double().T + bo.double() + res.double()).numpy()))
        ffb = acc_rows(emulate_ff(pk.stream.numpy().reshape(-1, FF320_CHUNK_BYTES)[4:46], pk.b2p, to_frags(tokb.numpy()), 1e-5, True))
        want = ffb if not with_epi else (torch.from_numpy(bf16_round(ffb)) @ wp.reshape(320, 320).to(torch.bfloat16).double().T
                                         + bp.double() + res2.double()).numpy()
        assert np.abs(got - want).max() < 1e-4, with_epi
