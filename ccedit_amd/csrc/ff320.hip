// Fused transformer feed-forward for dim = 320 (the 64x96 level) on gfx950:
//
//     out = x + W2 . GEGLU( W1 . LayerNorm(x) + b1 ) + b2          (attention.py:115-141, 695-716 / 758-761)
//
// in ONE kernel that reads x once and writes out once.  The 4C = 1280-wide hidden activation (535 MB per call at
// 34 x 64 x 96 tokens) never exists in memory, LayerNorm is folded into the first GEMM, and the residual is the very
// operand registers the kernel already holds.
//
// Register chaining.  A wave owns 48 tokens (3 MFMA column tiles of 16) for the whole call:
//   * X fragments: the wave's raw x rows as v_mfma_f32_16x16x32_bf16 B operands — lane (n = l & 15, g = l >> 4) holds
//     channels 32 s + 8 g .. + 7 of token n for k-step s: 10 k-steps x 3 tiles = 120 VGPRs, loaded once (16 B per lane).
//   * GEMM1 (hidden rows = A operand from LDS, K = 320): C tile = 16 hidden x 16 tokens, lane (n, g) register r is
//     hidden row 4 g + r of token n.  Two such tiles (hidden 16 a .. and 16 b ..) are, after bias / GEGLU / bf16,
//     EXACTLY one B operand of the second GEMM (k = 8 g + e: e < 4 from tile a, e >= 4 from tile b): the packer orders
//     W2's K axis accordingly, no lane ever exchanges data.
//   * GEMM2 (output channels = A operand, K = 32 hidden per chunk) accumulates out[320 x 48] in 240 accumulator
//     registers across the 40 hidden chunks.  W2's rows are permuted so that accumulator tile pair (2 s, 2 s + 1) of
//     lane (n, g) holds channels 32 s + 8 g .. + 7 — the same channels as X fragment s: the residual add and the final
//     16-byte store need no shuffle either.
//   * LayerNorm: W1' = W1 diag(gamma) is packed, so GEMM1 runs on the RAW x and its accumulators are corrected with the
//     token's statistics: h_pre = rstd (acc - mean s1[row]) + b1'[row], s1 = row sums of bf16(W1'), b1' = b1 + W1 beta.
//     mean / rstd come from the X registers (two-pass, 4-lane reduction).
//
// Weights stream through LDS in 62 KB chunks (one per 32 hidden units: 40 GEMM1 fragments, 20 GEMM2 fragments, 512 B of
// s1 / b1'), pre-packed in fragment order so that every ds_read_b128 is lane-linear (conflict-free) and every
// global_load_lds moves 1 KB contiguous.  Two chunk buffers; chunk j+1 is in flight while chunk j is consumed, one
// barrier per chunk.  One 4-wave workgroup per CU (a wave uses ~450 of the 512 registers of its SIMD); every weight byte
// staged serves 192 tokens: 21 B/clk of L2 -> LDS traffic at the MFMA peak (the per-CU L1-miss path sustains ~20).
#include "common.h"

namespace {

constexpr int kC = 320;                 // model width
constexpr int kKS = kC / 32;            // 10 k-steps of 32
constexpr int kOT = kC / 16;            // 20 output-channel tiles of 16
constexpr int kChunks = 1280 / 32;      // 40 hidden chunks of 32
constexpr int kW2Off = 4 * kKS * 1024;  // 40 KB of GEMM1 fragments, then 20 KB of GEMM2 fragments
constexpr int kAuxOff = kW2Off + kOT * 1024;
constexpr int kChunkBytes = 62 * 1024;  // + 1 KB of s1 / b1' (512 B used) + 1 KB pad
constexpr int kFrags = kChunkBytes / 1024;
constexpr int kNT = 3;                  // 16-token MFMA column tiles per wave
constexpr int kWavePix = 16 * kNT;

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 1) void ff320_kernel(const CcFf320Desc d, int n_rounds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][kChunkBytes]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const char* __restrict__ wstream = (const char*)d.wstream;
    const bf16* __restrict__ xp = (const bf16*)d.x;
    bf16* __restrict__ op = (bf16*)d.out;
    const f32x4* __restrict__ b2p = (const f32x4*)d.b2p;                 // [kOT][4 lane groups] x 4 floats

    // DMA share of this wave: fragments f = wave, wave + 4, ... of every chunk
    auto issue_chunk = [&](int q, int buf) {
        const char* src = wstream + (size_t)q * kChunkBytes + lane * 16;
        char* dst = smem + buf * kChunkBytes;
#pragma unroll
        for (int f = 0; f < (kFrags + 3) / 4; ++f) {
            const int fi = f * 4 + wave;
            if (fi < kFrags) glds16(src + fi * 1024, dst + fi * 1024);
        }
    };

    int cc = 0;                          // running chunk counter: buffer = cc & 1
    issue_chunk(0, 0);

    for (int round = blockIdx.x; round < n_rounds; round += gridDim.x) {
        const int64_t pix0 = ((int64_t)round * 4 + wave) * kWavePix;
        // ---- this wave's tokens: raw x as B-operand fragments (rows past M repeat the last row; never stored) ----
        bf16x8 xf[kNT][kKS];
        int64_t prow[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const int64_t p = pix0 + nt * 16 + n;
            prow[nt] = p;
            const bf16* row = xp + (p < d.M ? p : d.M - 1) * d.ldx + g * 8;
#pragma unroll
            for (int s = 0; s < kKS; ++s) xf[nt][s] = *(const bf16x8*)(row + s * 32);
        }
        // ---- LayerNorm statistics of each token: two passes over the registers, 4 lanes share a token ----
        float mean[kNT], rstd[kNT];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            float sm = 0.f;
#pragma unroll
            for (int s = 0; s < kKS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += bf2f(xf[nt][s][e]);
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            const float mu = sm * (1.0f / kC);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < kKS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dv = bf2f(xf[nt][s][e]) - mu;
                    sq = fmaf(dv, dv, sq);
                }
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
            mean[nt] = d.ln ? mu : 0.f;
            rstd[nt] = d.ln ? rsqrtf(sq * (1.0f / kC) + d.eps) : 1.f;
        }
        // ---- out accumulators start from b2 (already in accumulator order) ----
        f32x4 acc2[kOT][kNT];
#pragma unroll
        for (int t = 0; t < kOT; ++t) {
            const f32x4 bv = b2p[t * 4 + g];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt) acc2[t][nt] = bv;
        }

        for (int q = 0; q < kChunks; ++q, ++cc) {
            // chunk q (issued one chunk ago) has landed for this wave; after the barrier: for every wave, and every wave
            // has finished reading the other buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool last = (q == kChunks - 1) && (round + (int)gridDim.x >= n_rounds);
            if (!last) issue_chunk(q + 1 == kChunks ? 0 : q + 1, (cc + 1) & 1);
            const char* buf = smem + (cc & 1) * kChunkBytes;
            const char* fa = buf + lane * 16;

            // ---- GEMM1: 4 hidden tiles (value a, gate a, value b, gate b) x 3 token tiles, K = 320 ----
            f32x4 acc1[4][kNT];
#pragma unroll
            for (int tk = 0; tk < 4; ++tk)
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) acc1[tk][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < kKS; ++s) {
                bf16x8 a[4];
#pragma unroll
                for (int tk = 0; tk < 4; ++tk) a[tk] = *(const bf16x8*)(fa + (s * 4 + tk) * 1024);
#pragma unroll
                for (int tk = 0; tk < 4; ++tk)
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) acc1[tk][nt] = mfma16(a[tk], xf[nt][s], acc1[tk][nt]);
            }
            // ---- LayerNorm correction, bias, GEGLU -> the hidden chunk as GEMM2 B operands ----
            f32x4 s1v[4], b1v[4];
#pragma unroll
            for (int tk = 0; tk < 4; ++tk) {
                s1v[tk] = *(const f32x4*)(buf + kAuxOff + (tk * 16 + g * 4) * 4);
                b1v[tk] = *(const f32x4*)(buf + kAuxOff + 256 + (tk * 16 + g * 4) * 4);
            }
            bf16x8 hf[kNT];
#pragma unroll
            for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = fmaf(rstd[nt], acc1[2 * half][nt][r] - mean[nt] * s1v[2 * half][r], b1v[2 * half][r]);
                        const float u = fmaf(rstd[nt], acc1[2 * half + 1][nt][r] - mean[nt] * s1v[2 * half + 1][r],
                                             b1v[2 * half + 1][r]);
                        hf[nt][half * 4 + r] = f2bf(v * gelu_erf_f(u));
                    }
            // ---- GEMM2: out[320 x 48] += W2[:, chunk] . h ----
#pragma unroll
            for (int t = 0; t < kOT; ++t) {
                const bf16x8 a = *(const bf16x8*)(fa + kW2Off + t * 1024);
#pragma unroll
                for (int nt = 0; nt < kNT; ++nt) acc2[t][nt] = mfma16(a, hf[nt], acc2[t][nt]);
            }
        }

        // ---- residual + store: accumulator tiles (2 s, 2 s + 1) line up with X fragment s ----
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            if (prow[nt] < d.M) {
                bf16* row = op + prow[nt] * d.ldo + g * 8;
#pragma unroll
                for (int s = 0; s < kKS; ++s) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = f2bf(acc2[2 * s][nt][e] + bf2f(xf[nt][s][e]));
                        o[4 + e] = f2bf(acc2[2 * s + 1][nt][e] + bf2f(xf[nt][s][4 + e]));
                    }
                    *(bf16x8*)(row + s * 32) = o;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" int ccedit_ff320(const CcFf320Desc* desc, void* stream) {
    CC_CHECK_ARG(desc != nullptr, "ccedit_ff320: null descriptor");
    const CcFf320Desc d = *desc;
    CC_CHECK_ARG(d.x && d.out && d.wstream && d.b2p, "ccedit_ff320: null x/out/wstream/b2p");
    CC_CHECK_ARG(d.M > 0, "ccedit_ff320: M=%lld", (long long)d.M);
    CC_UNSUPPORTED(d.dim != kC || d.inner != 1280, "ccedit_ff320: only dim 320 / inner 1280 (got %d / %d)", d.dim, d.inner);
    CC_UNSUPPORTED(d.ldx % 8 != 0 || d.ldo % 8 != 0 || d.ldx < kC || d.ldo < kC, "ccedit_ff320: ldx=%d / ldo=%d", d.ldx, d.ldo);
    const int64_t rounds = (d.M + 4 * kWavePix - 1) / (4 * kWavePix);
    CC_UNSUPPORTED(rounds > 2147483647LL, "ccedit_ff320: M too large");
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)ff320_kernel, 2 * kChunkBytes, &attr_done, "ff320")) return rc;
    int cus = 256;
    {
        int dev = 0;
        static int cu_cache[64] = {0};
        if (hipGetDevice(&dev) == hipSuccess && dev < 64) {
            if (!cu_cache[dev]) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cu_cache[dev] = v;
            }
            if (cu_cache[dev]) cus = cu_cache[dev];
        }
    }
    const int grid = (int)(rounds < cus ? rounds : cus);
    hipLaunchKernelGGL(ff320_kernel, dim3(grid), dim3(256), 2 * kChunkBytes, (hipStream_t)stream, d, (int)rounds);
    return cc_launch_status("ff320_kernel");
}
