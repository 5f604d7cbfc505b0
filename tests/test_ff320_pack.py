"""CPU check of the fused feed-forward's weight stream (ccedit_amd/packing.py:pack_ff320) against a lane-level
emulation of csrc/ff320.hip: every index permutation the kernel relies on (GEMM1 C tiles == GEMM2 B operand,
GEMM2 accumulators aligned with the X fragments, LayerNorm folded into W1 / s1 / b1') is exercised with the
v_mfma_f32_16x16x32_bf16 register layouts (cdna_hip_programming.md §3):
    A operand: lane l holds A[i = l & 15][k = 8 (l >> 4) + e]      B operand: lane l holds B[k = 8 (l >> 4) + e][j = l & 15]
    C / D:     lane l, register r holds D[i = 4 (l >> 4) + r][j = l & 15]
The emulation follows the kernel statement by statement on one wave (48 tokens); the expected value is the reference
formula  x + W2 GEGLU(W1 LN(x) + b1) + b2  (attention.py:115-141, 695-716) in float64."""
import numpy as np
import torch

from ccedit_amd.packing import FF320_CHUNK_BYTES, pack_ff320

LANE = np.arange(64)
N_, G_ = LANE & 15, LANE >> 4


def mfma16(a_frag, b_frag, c_frag):
    """a_frag, b_frag: [64 lanes][8] float (bf16 values); c_frag: [64][4] -> D = A.B + C in the register layouts above."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4): 8 * (l >> 4) + 8] = a_frag[l]
        B[8 * (l >> 4): 8 * (l >> 4) + 8, l & 15] = b_frag[l]
    D = A @ B
    out = c_frag.copy()
    for l in range(64):
        out[l] += D[4 * (l >> 4): 4 * (l >> 4) + 4, l & 15]
    return out


def bf16_round(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy().astype(np.float64)


def gelu(v):
    import math
    return 0.5 * v * (1.0 + np.vectorize(math.erf)(v / np.sqrt(2.0)))


def emulate_wave(stream, b2p, x48, eps, ln=True, chunks=40):
    """x48: [48][320] bf16-representable floats.  Returns [48][320]."""
    stream = stream.numpy().reshape(41, FF320_CHUNK_BYTES)
    # X fragments: xf[nt][s][lane][e] = x[16 nt + n][32 s + 8 g + e]
    xf = np.zeros((3, 10, 64, 8))
    for nt in range(3):
        for s in range(10):
            for l in range(64):
                xf[nt, s, l] = x48[16 * nt + (l & 15), 32 * s + 8 * (l >> 4): 32 * s + 8 * (l >> 4) + 8]
    mean = np.zeros((3, 64)); rstd = np.ones((3, 64))
    if ln:
        for nt in range(3):
            sm = xf[nt].sum(axis=(0, 2))                       # per lane
            tot = np.array([sm[(l & 15) + 16 * np.arange(4)].sum() for l in range(64)])
            mu = tot / 320.0
            sq = ((xf[nt] - mu[None, :, None]) ** 2).sum(axis=(0, 2))
            tot2 = np.array([sq[(l & 15) + 16 * np.arange(4)].sum() for l in range(64)])
            mean[nt], rstd[nt] = mu, 1.0 / np.sqrt(tot2 / 320.0 + eps)
    acc2 = np.zeros((20, 3, 64, 4))
    b2 = b2p.numpy().reshape(20, 4, 4)
    for t in range(20):
        for nt in range(3):
            acc2[t, nt] = b2[t][G_]
    # the kernel's software pipeline: iteration c does GEMM1 half a of chunk c (+ GEGLU of half b of chunk c - 1 with the
    # s1 / b1' at +256), GEMM2 of chunk c - 1, GEMM1 half b of chunk c (+ GEGLU of half a of chunk c, s1 / b1' at +0)
    hf = np.zeros((3, 64, 8))
    accs = {1: np.zeros((2, 3, 64, 4))}
    for c in range(chunks + 1):
        ch = stream[c]
        f1 = torch.from_numpy(ch[: 40 * 1024].copy()).view(torch.bfloat16).float().numpy().reshape(2, 10, 2, 64, 8)
        f2 = torch.from_numpy(ch[40 * 1024: 60 * 1024].copy()).view(torch.bfloat16).float().numpy().reshape(20, 64, 8)
        aux = np.frombuffer(ch[60 * 1024: 60 * 1024 + 512].tobytes(), dtype=np.float32).reshape(2, 4, 16)
        for phase in (0, 1, 2):
            if phase == 1:
                for t in range(20):
                    for nt in range(3):
                        acc2[t, nt] = mfma16(f2[t], hf[nt], acc2[t, nt])
                continue
            half = 0 if phase == 0 else 1           # GEMM1 half of chunk c computed in this phase
            ge = 1 - half                           # GEGLU half finished in this phase (b of c - 1 in phase 1, a of c in phase 3)
            acc1 = accs[ge]
            for nt in range(3):
                for r in range(4):
                    s1v, s1g, b1v, b1g = (aux[ge, k][4 * G_ + r] for k in range(4))
                    v = rstd[nt] * acc1[0, nt][:, r] - rstd[nt] * mean[nt] * s1v + b1v
                    u = rstd[nt] * acc1[1, nt][:, r] - rstd[nt] * mean[nt] * s1g + b1g
                    hf[nt][:, ge * 4 + r] = bf16_round(v * gelu(u))
            acc1 = np.zeros((2, 3, 64, 4))
            for s in range(10):
                for kind in range(2):
                    for nt in range(3):
                        acc1[kind, nt] = mfma16(f1[half, s, kind], xf[nt, s], acc1[kind, nt])
            accs[half] = acc1
    out = np.zeros((48, 320))
    for nt in range(3):
        for s in range(10):
            for l in range(64):
                n, g = l & 15, l >> 4
                o = np.concatenate([acc2[2 * s, nt, l] + xf[nt, s, l, :4], acc2[2 * s + 1, nt, l] + xf[nt, s, l, 4:]])
                out[16 * nt + n, 32 * s + 8 * g: 32 * s + 8 * g + 8] = o
    return out


def _reference(x, w1, b1, w2, b2, g, b, eps, ln=True):
    x = x.double()
    h = torch.nn.functional.layer_norm(x, (320,), g.double(), b.double(), eps) if ln else x
    p = h @ w1.double().T + b1.double()
    v, gate = p.chunk(2, dim=-1)
    return x + (v * torch.nn.functional.gelu(gate)) @ w2.double().T + b2.double()


def test_weight_stream_matches_kernel_dataflow():
    gen = torch.Generator().manual_seed(7)
    w1 = torch.randn(2560, 320, generator=gen) * 0.05
    b1 = torch.randn(2560, generator=gen) * 0.1
    w2 = torch.randn(320, 1280, generator=gen) * 0.03
    b2 = torch.randn(320, generator=gen) * 0.1
    lg = 1.0 + 0.2 * torch.randn(320, generator=gen)
    lb = 0.1 * torch.randn(320, generator=gen)
    x = (torch.randn(48, 320, generator=gen) * 1.5 + 0.7).to(torch.bfloat16).float()
    pk = pack_ff320(w1, b1, w2, b2, lg, lb)
    assert pk.stream.numel() == 41 * FF320_CHUNK_BYTES and pk.stream.dtype == torch.uint8 and pk.b2p.shape == (320,)
    got = emulate_wave(pk.stream, pk.b2p, x.numpy().astype(np.float64), 1e-5)
    # expected with the SAME bf16 weights the stream carries, so that only the dataflow / index maps are under test
    w1g = (w1.double() * lg.double()[None]).float().to(torch.bfloat16)
    xn = (x.double() - x.double().mean(1, keepdim=True)) / torch.sqrt(x.double().var(1, unbiased=False, keepdim=True) + 1e-5)
    pre = xn @ w1g.double().T + (b1.double() + w1.double() @ lb.double())
    v, gate = pre.chunk(2, dim=-1)
    h = torch.from_numpy(bf16_round((v * torch.nn.functional.gelu(gate)).numpy()))
    want = x.double() + h @ w2.to(torch.bfloat16).double().T + b2.double()
    err = np.abs(got - want.numpy()).max()
    assert err < 1e-6, err
    # and against the plain reference formula (fp32 weights): only bf16 rounding of weights / hidden apart
    ref = _reference(x, w1, b1, w2, b2, lg, lb, 1e-5).numpy()
    rel = np.sqrt(((got - ref) ** 2).mean() / ((ref - x.numpy()) ** 2).mean())
    assert rel < 1e-2, rel


def test_weight_stream_without_layernorm():
    gen = torch.Generator().manual_seed(8)
    w1 = torch.randn(2560, 320, generator=gen) * 0.05
    w2 = torch.randn(320, 1280, generator=gen) * 0.03
    b1 = torch.randn(2560, generator=gen) * 0.1
    b2 = torch.randn(320, generator=gen) * 0.1
    x = torch.randn(48, 320, generator=gen).to(torch.bfloat16).float()
    pk = pack_ff320(w1, b1, w2, b2, None, None)
    got = emulate_wave(pk.stream, pk.b2p, x.numpy().astype(np.float64), 1e-5, ln=False, chunks=40)
    ref = _reference(x, w1.to(torch.bfloat16).float(), b1, w2.to(torch.bfloat16).float(), b2, None, None, 1e-5, ln=False)
    v, gate = (x.double() @ w1.to(torch.bfloat16).double().T + b1.double()).chunk(2, dim=-1)
    h = torch.from_numpy(bf16_round((v * torch.nn.functional.gelu(gate)).numpy()))
    want = x.double() + h @ w2.to(torch.bfloat16).double().T + b2.double()
    assert np.abs(got - want.numpy()).max() < 1e-6
