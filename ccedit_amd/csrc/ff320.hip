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
// Weights stream through LDS in 64 KB chunks (one per 32 hidden units: 40 GEMM1 fragments, 20 GEMM2 fragments, 512 B of
// s1 / b1'), pre-packed in fragment order so that every ds_read_b128 is lane-linear (conflict-free) and every
// global_load_lds moves 1 KB contiguous.  Two chunk buffers; chunk j+1 is in flight while chunk j is consumed, one
// barrier per chunk.  One 4-wave workgroup per CU (a wave uses ~450 of the 512 registers of its SIMD); every weight byte
// staged serves 192 tokens: 21 B/clk of L2 -> LDS traffic at the MFMA peak (the per-CU L1-miss path sustains ~20).
#include "common.h"
#include <stdlib.h>
#include <utility>

namespace {

constexpr int kC = 320;                 // model width
constexpr int kKS = kC / 32;            // 10 k-steps of 32
constexpr int kOT = kC / 16;            // 20 output-channel tiles of 16
constexpr int kIters = 1280 / 32 + 1;   // 40 hidden chunks of 32, software-pipelined over 41 iterations
constexpr int kW2Off = 4 * kKS * 1024;  // 40 KB of GEMM1 fragments, then 20 KB of GEMM2 fragments
constexpr int kAuxOff = kW2Off + kOT * 1024;
constexpr int kChunkBytes = 64 * 1024;  // + s1 / b1' (512 B used of the last 4 KB: every wave issues exactly 16 DMA instructions)
constexpr int kNT = 3;                  // 16-token MFMA column tiles per wave
constexpr int kWavePix = 16 * kNT;
constexpr int kD = 6;                   // fragment reads in flight ahead of their use

typedef __attribute__((ext_vector_type(2))) float f32x2;

template <class F, int... J>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, J...>) {
    (f(std::integral_constant<int, J>{}), ...);
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LDS reads the compiler cannot move: one wave per SIMD means nothing but the wave's own later instructions hides a
// ds_read's latency, and under this kernel's register pressure hipcc sinks every read next to its consumer (measured:
// 2 reads, lgkmcnt(0), 6 MFMAs, ... = 34 - 61 cycles per MFMA).  The reads are issued kD fragments ahead through these
// statements and retired with counted waits (LDS returns in order); the wait names the register, so no consumer can be
// scheduled above it.
#define FF_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define FF_WAIT(reg, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "i"(cnt))
// Fragment used at step j of an iteration: GEMM1 half a (0..19), GEMM2 (40..59), GEMM1 half b (20..39).
__device__ __forceinline__ constexpr int frag_of(int j) { return j < 20 ? j : (j < 40 ? j + 20 : j - 20); }

// Two GEGLU elements (rows r, r + 1 of one token tile): value x exact-erf GELU(gate) (common.h) -> bf16 pair; the biases
// arrived as the C operand of the first MFMA.  Scalar fp32 on purpose, and this file is built with -fno-slp-vectorize:
// beside MFMAs a v_pk_fma_f32 / v_pk_mul_f32 costs several times a v_fma_f32 (MI355X_MICROARCH.md, "price of one filler
// beside MFMAs"; measured here: the packed form of this function made the kernel slower).
__device__ __forceinline__ bf16x2 geglu2(f32x2 v, f32x2 u, bool plain) {
    bf16x2 o;
#pragma unroll
    for (int i = 0; i < 2; ++i) o[i] = f2bf(plain ? v[i] + u[i] : v[i] * gelu_erf_f(u[i]));
    return o;
}

// ABL (tuning only, CCEDIT_FF320_ABL, bit mask): 1 = no weight stream after the first chunk; 2 = GEGLU replaced by an add; 4 = no
// barrier; 8 = no LDS fragment reads; 16 = no GEGLU at all.
//
// Software pipeline over the 40 hidden chunks (iteration c = 0 .. 40 works on stream chunk c = [W1(c) | W2(c-1) | s1,b1'(c) a-half | (c-1) b-half]):
//     phase 1   GEMM1 half a of chunk c       with the GEGLU of half b of chunk c-1 between its MFMA steps  -> completes h(c-1)
//     phase 2   GEMM2 of chunk c-1            out += W2(c-1) . h(c-1)
//     phase 3   GEMM1 half b of chunk c       with the GEGLU of half a of chunk c between its MFMA steps
// Chunk -1 / 40 do not exist: the packer supplies zero fragments, so iteration 0's GEMM2 and iteration 40's GEMM1 add zeros.
template <int ABL>
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
    const int lane16 = lane * 16;

    // DMA share of this wave: fragments wave, wave + 4, ... (16 per chunk); uniform base + lane offset
    auto issue_frag = [&](int c, int buf, int k) {
        const char* src = wstream + ((size_t)c * kChunkBytes + (size_t)(k * 4 + wave) * 1024);     // wave-uniform
        glds16(src + lane16, smem + buf * kChunkBytes + (k * 4 + wave) * 1024);
    };

    const int gdim = (int)gridDim.x;
    int cc = 0;                          // running chunk counter: buffer = cc & 1
#pragma unroll
    for (int k = 0; k < 16; ++k) issue_frag(0, 0, k);

    f32x4 accB[2][kNT];                  // GEMM1 half b of the previous iteration (zero at c = 0: iteration 40 leaves zeros)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) accB[k][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 hf[kNT];                      // the hidden chunk as GEMM2 B operands: elements 0..3 half a, 4..7 half b
#pragma unroll
    for (int nt = 0; nt < kNT; ++nt)
#pragma unroll
        for (int e = 0; e < 8; ++e) hf[nt][e] = f2bf(0.f);

    for (int round = blockIdx.x; round < n_rounds; round += gdim) {
        const int64_t pix0 = ((int64_t)round * 4 + wave) * kWavePix;
        // ---- this wave's tokens: raw x as B-operand fragments (rows past M repeat the last row; never stored) ----
        bf16x8 xf[kNT][kKS];
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const int64_t p = pix0 + nt * 16 + n;
            const bf16* row = xp + (p < d.M ? p : d.M - 1) * d.ldx + g * 8;
#pragma unroll
            for (int s = 0; s < kKS; ++s) xf[nt][s] = *(const bf16x8*)(row + s * 32);
        }
        // ---- LayerNorm statistics of each token (two passes over the registers, 4 lanes share a token); the accumulators
        //      of the second GEMM start from b2 + x (the residual, exact in fp32: accumulator tiles (2 s, 2 s + 1) hold the
        //      channels of X fragment s); then the fragments are normalised IN PLACE: xf <- bf16((x - mean) rstd), gamma and
        //      beta live in W1' / b1' ----
        f32x4 acc2[kOT][kNT];
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
            const float rs = d.ln ? rsqrtf(sq * (1.0f / kC) + d.eps) : 1.f;
            const float rm = d.ln ? rs * mu : 0.f;
#pragma unroll
            for (int s = 0; s < kKS; ++s) {
                const f32x4 b0 = b2p[(2 * s) * 4 + g], b1 = b2p[(2 * s + 1) * 4 + g];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc2[2 * s][nt][e] = b0[e] + bf2f(xf[nt][s][e]);
                    acc2[2 * s + 1][nt][e] = b1[e] + bf2f(xf[nt][s][4 + e]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[nt][s][e] = f2bf(fmaf(rs, bf2f(xf[nt][s][e]), -rm));
            }
        }

        for (int c = 0; c < kIters; ++c, ++cc) {
            // chunk c (issued during the previous iteration) has landed for this wave; after the barrier: for every wave, and
            // every wave has finished reading the other buffer
            if constexpr (!(ABL & 4)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            const bool last = (c == kIters - 1) && (round + gdim >= n_rounds);
            const bool issue_next = !last && !((ABL & 1) && cc >= 1);
            const int cn = c + 1 == kIters ? 0 : c + 1;
            const int nb = (cc + 1) & 1;
            const char* buf = smem + (cc & 1) * kChunkBytes;
            const unsigned la = (unsigned)(uintptr_t)(LDS_AS const char*)(buf + lane16);                  // fragment base
            const unsigned lx = (unsigned)(uintptr_t)(LDS_AS const char*)(buf + kAuxOff + g * 16);       // s1 / b1' of this lane group

            // LDS read sequence of the iteration: b1' of this chunk's half a, then the 60 fragments in use order (GEMM1 a =
            // 0..19, GEMM2 = 40..59, GEMM1 b = 20..39) kD ahead of their use; b1' of half b is slipped in behind step 30.
            f32x4 biasA[2], biasB[2];    // b1' [value, gate], rows 4 g .. 4 g + 3 of the half: the C operand of the first k-step
#pragma unroll
            for (int k = 0; k < 2; ++k) FF_DS_READ(biasA[k], lx, k * 64);
            bf16x8 ring[kD];
#pragma unroll
            for (int j = 0; j < kD; ++j) FF_DS_READ(ring[j], la, frag_of(j) * 1024);
            f32x4 accA[2][kNT];

            auto step = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int f = frag_of(j);
                constexpr int left = 59 - j < kD - 1 ? 59 - j : kD - 1;      // younger fragment reads that may stay in flight
                if constexpr (!(ABL & 8)) FF_WAIT(ring[j % kD], (j > 30 && j <= 30 + kD ? left + 2 : left));
                if constexpr (j == 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) FF_WAIT(biasA[k], kD - 1);  // older than fragment 0: already complete
                }
                if constexpr (j == 40) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) FF_WAIT(biasB[k], kD - 1);  // older than fragment 31 + kD
                }
                if constexpr (f < 20) {          // phase 1: GEMM1 half a of chunk c
                    constexpr int s = f >> 1, kind = f & 1;
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) {
                        accA[kind][nt] = mfma16(ring[j % kD], xf[nt][s], s == 0 ? biasA[kind] : accA[kind][nt]);
                    }
                } else if constexpr (f >= 40) {  // phase 2: GEMM2 of chunk c - 1
                    constexpr int t = f - 40;
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) acc2[t][nt] = mfma16(ring[j % kD], hf[nt], acc2[t][nt]);
                } else {                         // phase 3: GEMM1 half b of chunk c
                    constexpr int s = (f - 20) >> 1, kind = f & 1;
#pragma unroll
                    for (int nt = 0; nt < kNT; ++nt) {
                        accB[kind][nt] = mfma16(ring[j % kD], xf[nt][s], s == 0 ? biasB[kind] : accB[kind][nt]);
                    }
                }
                if constexpr (j + kD < 60 && !(ABL & 8)) FF_DS_READ(ring[j % kD], la, frag_of(j + kD) * 1024);
                if constexpr (j == 30) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) FF_DS_READ(biasB[k], lx, 128 + k * 64);
                }
                if constexpr (j < 16) {
                    if (issue_next) issue_frag(cn, nb, j);                   // the next chunk's DMA, spread over the MFMA steps
                }
                // GEGLU pairs between the MFMA steps: half b of chunk c - 1 in phase 1 (steps 1, 4, .. 16), half a of chunk c
                // in phase 3 (steps 41, 44, .. 56); pair p = (token tile p / 2, rows 2 (p % 2), + 1)
                if constexpr (j < 20 && j % 3 == 1 && j / 3 < 2 * kNT && !(ABL & 16)) {
                    constexpr int p = j / 3, nt = p >> 1, r = 2 * (p & 1);
                    const bf16x2 h = geglu2(f32x2{accB[0][nt][r], accB[0][nt][r + 1]}, f32x2{accB[1][nt][r], accB[1][nt][r + 1]}, (ABL & 2) != 0);
                    hf[nt][4 + r] = h[0];
                    hf[nt][4 + r + 1] = h[1];
                }
                if constexpr (j >= 40 && (j - 40) % 3 == 1 && (j - 40) / 3 < 2 * kNT && !(ABL & 16)) {
                    constexpr int p = (j - 40) / 3, nt = p >> 1, r = 2 * (p & 1);
                    const bf16x2 h = geglu2(f32x2{accA[0][nt][r], accA[0][nt][r + 1]}, f32x2{accA[1][nt][r], accA[1][nt][r + 1]}, (ABL & 2) != 0);
                    hf[nt][r] = h[0];
                    hf[nt][r + 1] = h[1];
                }
            };
            for_seq(step, std::make_integer_sequence<int, 60>{});
        }

        // ---- store: accumulator tiles (2 s, 2 s + 1) of lane (n, g) are channels 32 s + 8 g .. + 7 of token n ----
#pragma unroll
        for (int nt = 0; nt < kNT; ++nt) {
            const int64_t p = pix0 + nt * 16 + n;
            if (p < d.M) {
                bf16* row = op + p * d.ldo + g * 8;
#pragma unroll
                for (int s = 0; s < kKS; ++s) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o[e] = f2bf(acc2[2 * s][nt][e]);
                        o[4 + e] = f2bf(acc2[2 * s + 1][nt][e]);
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
    static const int abl = getenv("CCEDIT_FF320_ABL") ? atoi(getenv("CCEDIT_FF320_ABL")) : 0;
    void (*kern)(const CcFf320Desc, int) = ff320_kernel<0>;
    switch (abl) {        // tuning only (bit mask, see the kernel); results are wrong for abl != 0
        case 1: kern = ff320_kernel<1>; break;
        case 2: kern = ff320_kernel<2>; break;
        case 4: kern = ff320_kernel<4>; break;
        case 8: kern = ff320_kernel<8>; break;
        case 16: kern = ff320_kernel<16>; break;
        case 24: kern = ff320_kernel<24>; break;
        case 29: kern = ff320_kernel<29>; break;
        default: break;
    }
    static unsigned long long attr_done[32] = {0};
    if (int rc = cc_max_dynamic_lds((const void*)kern, 2 * kChunkBytes, &attr_done[abl & 31], "ff320")) return rc;
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
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 2 * kChunkBytes, (hipStream_t)stream, d, (int)rounds);
    return cc_launch_status("ff320_kernel");
}
