// fp32 kernels of the first-stage model (KL-VAE) — the precision the reference decodes in: `decode_first_stage` runs the
// decoder with autocast DISABLED (sgm/models/diffusion.py:151-156, `disable_first_stage_autocast`), i.e. fp32 operands, fp32
// products, fp32 tensors.  The bf16 kernels of the rest of the path serve the VAE by default (a clip's decode is 1.4 % of its
// FLOPs); these three entry points are the option that reproduces the reference's arithmetic class: fp32 in, fp32 MFMA
// (`v_mfma_f32_32x32x2_f32`, true fp32 products and sums), fp32 out.
//
//   ccedit_gemm_f32           out[m][n] = bias[n] + sum_k W[n][k] * gather(A)[m][k] (+ res[m][n])
//                             Linear / Conv 1x1 (mode 0) and Conv2d 3x3 with stride, padding, an explicit output size (the encoder's
//                             asymmetric pad) and an optional fused nearest-2x upsample of the source (mode 1)
//                             (model.py:56-71 Upsample, 74-93 Downsample, 94-151 ResnetBlock, 161-201 AttnBlock)
//   ccedit_groupnorm_f32      GroupNorm(32, C, eps) [+ SiLU] over (H W) x C/32 of a frame (model.py:45-53), statistics in fp64
//   ccedit_softmax_rows_f32   in-place softmax(scale * s) of the single-head attention scores (model.py:180-195)
//
// GEMM kernel: 128 channels x 128 pixels per workgroup, four waves of 64 x 64 (2 x 2 MFMA tiles, 64 accumulator registers),
// K tiles of 16 through a two-buffer LDS ring filled from registers (the next tile's global loads are issued before the MFMAs of
// the current one).  Weights are the A operand, pixels the B operand, as everywhere in this library: a lane owns one pixel and
// runs of four consecutive channels, stored as 16-byte pieces.  The K index inside a tile is permuted (lanes 0-31 take k 0..7,
// lanes 32-63 k 8..15: MFMA j multiplies k = {j, 8 + j}) so that a lane's eight operands are two ds_read_b128; rows are 80 bytes
// apart in LDS (conflict-free for that pattern).  The matrix pipe runs fp32 at 256 FLOP / clk / CU (157 TFLOP/s): one MFMA is 64
// cycles against two 16-byte LDS reads, so this simple loop is MFMA-bound; held to 128 registers: four workgroups per CU (4 x 40 KB of LDS = the CU's 160 KB).
#include "common.h"

namespace {

constexpr int F_BM = 128, F_BN = 128, F_BK = 16, F_LD = 20;           // LDS row = 16 floats + 4 pad
constexpr int F_TILE = (F_BM + F_BN) * F_LD;                           // floats per buffer

struct F32Pix {            // geometry of one pixel row of the tile (conv mode)
    int64_t base;          // element offset of its frame in A
    int oy, ox;            // output coordinates times stride, minus pad (top-left tap position in the virtual source)
};

// Epilogue shared by the three GEMM kernels: a wave's 2 x 2 accumulator tiles (64 channels from n0, 64 pixels from m0) to rows of
// `out` (+ bias, + residual).  Lane = pixel (column) l31 of tile tj; accumulator r = channel 8 (r / 4) + 4 hi + r % 4 of tile ti.
// out_row(m): the row of GEMM row m (m itself, or its place in the up-sampled frame for a parity conv).
template <typename OutRow>
__device__ __forceinline__ void f32_store_tiles(const CcGemmF32Desc& d, const f32x16 (&acc)[2][2], int64_t m0, int n0, int l31, int hi, OutRow out_row) {
    const bool vec = (d.ldc & 3) == 0 && (!d.res || (d.ldr & 3) == 0);
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int64_t m = m0 + tj * 32 + l31;
        if (m >= d.M) continue;
        float* orow = d.out + out_row(m) * d.ldc;
        const float* rrow = d.res ? d.res + (size_t)m * d.ldr : nullptr;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + ti * 32 + 8 * q + 4 * hi;
                if (n >= d.N) continue;
                f32x4 v = {acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1], acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]};
                if (vec && n + 3 < d.N) {
                    if (d.bias) v += *(const f32x4*)(d.bias + n);
                    if (rrow) v += *(const f32x4*)(rrow + n);
                    *(f32x4*)(orow + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < d.N) {
                            float o = v[e];
                            if (d.bias) o += d.bias[n + e];
                            if (rrow) o += rrow[n + e];
                            orow[n + e] = o;
                        }
                }
            }
    }
}

template <bool CONV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void f32_gemm_kernel(const CcGemmF32Desc d) {
    __shared__ __attribute__((aligned(16))) float smem[2 * F_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    // channel tiles fastest: the workgroups that run together share the pixel rows they gather
    const int ct_n = (d.N + F_BN - 1) / F_BN;
    const int64_t pt = blockIdx.x / ct_n;
    const int ct = blockIdx.x - (int)pt * ct_n;
    const int64_t m0 = pt * F_BM;
    const int n0 = ct * F_BN;
    const int nk = d.Kpad / F_BK;

    // staging: thread -> rows (tid >> 2) and (tid >> 2) + 64 of both operands, floats 4 (tid & 3) .. + 3 of the K tile
    const int srow = tid >> 2, sg = tid & 3;
    const float* wsrc[2];
    F32Pix px[2];
    const float* asrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = min(n0 + srow + 64 * i, d.N - 1);
        wsrc[i] = d.W + (size_t)n * d.ldw + sg * 4;
        const int64_t m = min(m0 + srow + 64 * i, d.M - 1);
        if constexpr (CONV) {
            const int hw = d.Hout * d.Wout;
            const int64_t f = m / hw;
            const int r = (int)(m - f * hw);
            const int y = r / d.Wout, x = r - y * d.Wout;
            px[i].base = f * (int64_t)d.Hin * d.Win * d.lda;
            px[i].oy = y * d.stride - d.pad;
            px[i].ox = x * d.stride - d.pad;
        } else {
            asrc[i] = d.A + (size_t)m * d.lda + sg * 4;
        }
    }
    const int Hv = d.upsample ? 2 * d.Hin : d.Hin, Wv = d.upsample ? 2 * d.Win : d.Win;      // the (virtual) source the taps walk over
    const int us = d.upsample ? 1 : 0;

    f32x4 ra[2], rb[2];
    auto fetch = [&](int kt) {
        const int k0 = kt * F_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *(const f32x4*)(wsrc[i] + k0);
        if constexpr (CONV) {
            const int tap = k0 / d.Cpad, c = k0 - tap * d.Cpad + sg * 4;           // a K tile never straddles taps (Cpad % 16 == 0)
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int iy = px[i].oy + ky, ix = px[i].ox + kx;
                const bool ok = iy >= 0 && iy < Hv && ix >= 0 && ix < Wv && c < d.Cin;
                const float* p = d.A + px[i].base + ((int64_t)(iy >> us) * d.Win + (ix >> us)) * d.lda + c;
                rb[i] = ok ? *(const f32x4*)p : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            const bool ok = k0 + sg * 4 < d.Cin;
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[i] = ok ? *(const f32x4*)(asrc[i] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        float* s = smem + buf * F_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(f32x4*)(s + (srow + 64 * i) * F_LD + sg * 4) = ra[i];
            *(f32x4*)(s + (F_BM + srow + 64 * i) * F_LD + sg * 4) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fetch(kt + 1);
        const float* s = smem + (kt & 1) * F_TILE;
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* pa = s + (wr * 64 + t * 32 + l31) * F_LD + hi * 8;
            const float* pb = s + (F_BM + wc * 64 + t * 32 + l31) * F_LD + hi * 8;
            fa[t][0] = *(const f32x4*)pa;
            fa[t][1] = *(const f32x4*)(pa + 4);
            fb[t][0] = *(const f32x4*)pb;
            fb[t][1] = *(const f32x4*)(pb + 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ti][j >> 2][j & 3], fb[tj][j >> 2][j & 3], acc[ti][tj], 0, 0, 0);
        if (kt + 1 < nk) stash((kt + 1) & 1);
        __syncthreads();
    }

    f32_store_tiles(d, acc, m0 + wc * 64, n0 + wr * 64, l31, hi, [](int64_t m) { return (size_t)m; });
}

// ---- the same contraction on the bf16 matrix pipe: exact three-way operand split (policy f32_split, the default) ----
// An fp32 number is EXACTLY the sum of three bf16 numbers: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (round-to-nearest-even
// each; x - h and x - h - m are exact in fp32, the last remainder has <= 8 significant bits, and bf16 has fp32's exponent range).
// Then w * a = sum of nine bf16 x bf16 products, each EXACT in fp32 (8 x 8 significant bits).  The kernel keeps the six with
// weight >= 2^-16 (hh, hm, mh, hl, lh, mm) and drops ml, lm, ll: <= 2 * 2^-9 * 2^-17 + 2^-34 = 2^-25 of |w a|, signs random — below
// the 2^-24 |acc| rounding that EVERY fp32 accumulation step of v_mfma_f32_32x32x2_f32 makes as well.  Sums are fp32 inside the matrix
// pipe exactly as before; per output and 16 k there are six accumulations where the fp32 instruction makes eight.  The result is an
// fp32 contraction in every measurable sense (tests/test_vae_f32_gpu.py holds BOTH arms to the same 3e-6 against fp64 and the
// whole decoder to the reference's fp32 goldens) at six v_mfma_f32_32x32x16_bf16 (8 passes each: 48) where the fp32 form
// needs eight of 16 passes (128): 2.67 x the matrix rate.
//
// 128 channels x 256 pixels per workgroup, eight waves of 64 x 64; K tiles of 16; operands arrive as fp32 in registers (the
// gather of f32_gemm_kernel, unchanged), are split there and stored as three bf16 planes of 32-byte rows (granule g of row r at
// slot g ^ ((r >> 3) & 1): ds_read_b128's four 16-lane groups each touch sixteen distinct 16-byte slots); two buffers of 36 KB:
// two workgroups per CU.  Six fragment sets per k-tile, loaded just before their first use (three sets live at a time).
#ifndef F32S_ABL
#define F32S_ABL 0          // probe builds only (tools/exp/f32s_abl.sh): 1 no split arithmetic, 2 no loads in the loop, 3 no LDS stores in the loop, 4 no MFMAs, 5 (f32p) no workgroup barriers in the loop
#endif
constexpr int S_BM = 128, S_BN = 256, S_BK = 16;
constexpr int S_ROWS = S_BM + S_BN;
constexpr int S_PLANE = S_ROWS * 32;           // bytes of one plane of one buffer
constexpr int S_BUF = 3 * S_PLANE;

// The split, two elements at a time (x0, x1 -> one packed bf16 pair per plane): eleven instructions.  Inline asm for the conversion
// and the subtraction: left to itself hipcc converts the low element a second time to get its value back and forms packed-fp32
// subtractions (csrc/build.py: EXTRA_FLAGS, what those cost beside MFMAs).
__device__ __forceinline__ uint32_t f32s_cvt_pk(float a, float b) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float f32s_sub(float a, uint32_t bits) {
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(bits));
    return r;
}
struct F32Split {          // four consecutive k of one row, three planes
    u32x2 p[3];
};
__device__ __forceinline__ F32Split f32_split3(f32x4 v) {
    F32Split s;
#if F32S_ABL == 1          // (probe builds: no split arithmetic)
    s.p[0] = u32x2{f32s_cvt_pk(v[0], v[1]), f32s_cvt_pk(v[2], v[3])};
    s.p[1] = s.p[0];
    s.p[2] = s.p[0];
    return s;
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float x0 = v[2 * h], x1 = v[2 * h + 1];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint32_t b = f32s_cvt_pk(x0, x1);
            s.p[p][h] = b;
            if (p < 2) {
                x0 = f32s_sub(x0, b << 16);
                x1 = f32s_sub(x1, b & 0xffff0000u);
            }
        }
    }
    return s;
}

// MODE 0: Linear / Conv 1x1; 1: Conv2d 3x3; 2: Conv2d 3x3 on the nearest-2x up-sampled source (stride 1, pad 1).
// Gather addressing: a pixel row keeps ONE pointer — its top-left tap in the source, this thread's k piece included — and a 9-bit
// mask of the taps that fall inside the frame; a tap adds a wave-uniform element offset (mode 2: (dy Win + dx) lda with
// dy = (parity_y + ky) >> 1, the source row of the virtual row, per lane).
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void f32s_gemm_kernel(const CcGemmF32Desc d) {
    constexpr bool CONV = MODE != 0, UPS = MODE == 2;
    __shared__ __attribute__((aligned(16))) char smem[2 * S_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int ct_n = (d.N + S_BM - 1) / S_BM;
    const int64_t pt = blockIdx.x / ct_n;
    const int ct = blockIdx.x - (int)pt * ct_n;
    const int64_t m0 = pt * S_BN;
    const int n0 = ct * S_BM;
    const int nk = d.Kpad / S_BK;

    // staging: thread -> weight row tid >> 2 and pixel rows (tid >> 2), (tid >> 2) + 128; floats 4 (tid & 3) .. + 3 of the K tile
    const int srow = tid >> 2, sg = tid & 3;
    const float* wsrc = d.W + (size_t)min(n0 + srow, d.N - 1) * d.ldw + sg * 4;
    const float* asrc[2];
    uint32_t tapmask[2];          // CONV: bit t = tap t of this pixel lies inside the frame
    int par[2];                   // UPS: parity of the top-left tap's virtual (y, x): bit 1 = y, bit 0 = x
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = min(m0 + srow + 128 * i, d.M - 1);
        if constexpr (CONV) {
            const int hw = d.Hout * d.Wout;
            const int64_t f = m / hw;
            const int r = (int)(m - f * hw);
            const int y = r / d.Wout, x = r - y * d.Wout;
            const int oy = y * d.stride - d.pad, ox = x * d.stride - d.pad;        // top-left tap in the (virtual) source
            const int Hv = UPS ? 2 * d.Hin : d.Hin, Wv = UPS ? 2 * d.Win : d.Win;
            uint32_t mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy + t / 3, ix = ox + t % 3;
                mk |= (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) ? 1u << t : 0u;
            }
            tapmask[i] = mk;
            par[i] = UPS ? ((oy & 1) << 1 | (ox & 1)) : 0;
            const int sy = UPS ? oy >> 1 : oy, sx = UPS ? ox >> 1 : ox;            // (arithmetic shift: -1 >> 1 = -1)
            asrc[i] = d.A + (f * (int64_t)d.Hin * d.Win + (int64_t)sy * d.Win + sx) * d.lda + sg * 4;
        } else {
            asrc[i] = d.A + (size_t)m * d.lda + sg * 4;
        }
    }
    const int cmax = d.Cin - sg * 4;           // this thread's k piece holds real channels while the tile's first channel is < cmax

    f32x4 ra, rb[2];
    auto fetch = [&](int kt) {
        const int k0 = kt * S_BK;
        ra = *(const f32x4*)(wsrc + k0);
        if constexpr (CONV) {
            const int tap = k0 / d.Cpad, c0 = k0 - tap * d.Cpad;           // a K tile never straddles taps (Cpad % 16 == 0)
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int uoff = (ky * d.Win + kx) * d.lda + c0;               // (not UPS) wave-uniform
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int off = uoff;
                if constexpr (UPS) off = ((((par[i] >> 1) + ky) >> 1) * d.Win + (((par[i] & 1) + kx) >> 1)) * d.lda + c0;
                const bool ok = ((tapmask[i] >> tap) & 1) && c0 < cmax;
                rb[i] = ok ? *(const f32x4*)(asrc[i] + off) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            const bool ok = k0 < cmax;
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[i] = ok ? *(const f32x4*)(asrc[i] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // byte offset of (row r, k piece sg) inside a plane
    auto slot = [&](int r) { return r * 32 + ((((sg >> 1) ^ (r >> 3)) & 1) << 4) + ((sg & 1) << 3); };
    const int so_a = slot(srow), so_b0 = slot(S_BM + srow), so_b1 = slot(S_BM + srow + 128);
    auto stash = [&](int buf) {
        char* s = smem + buf * S_BUF;
        const F32Split sa = f32_split3(ra), sb0 = f32_split3(rb[0]), sb1 = f32_split3(rb[1]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *(u32x2*)(s + p * S_PLANE + so_a) = sa.p[p];
            *(u32x2*)(s + p * S_PLANE + so_b0) = sb0.p[p];
            *(u32x2*)(s + p * S_PLANE + so_b1) = sb1.p[p];
        }
    };
    // fragment addresses: row l31 of a 32-row tile, k granule hi
    const int fo = l31 * 32 + ((hi ^ (l31 >> 3)) & 1) * 16;
    const int fa0 = (wr * 64) * 32 + fo, fb0 = (S_BM + wc * 64) * 32 + fo;

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
#if F32S_ABL != 2
        if (kt + 1 < nk) fetch(kt + 1);
#endif
        const char* s = smem + (kt & 1) * S_BUF;
        auto fragA = [&](int p, bf16x8 (&f)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) f[t] = *(const bf16x8*)(s + p * S_PLANE + fa0 + t * 1024);
        };
        auto fragB = [&](int p, bf16x8 (&f)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) f[t] = *(const bf16x8*)(s + p * S_PLANE + fb0 + t * 1024);
        };
        auto mm = [&](const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
#if F32S_ABL == 4
                    acc[ti][tj][0] += (float)a[ti][0] * (float)b[tj][0];
#else
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
#endif
                }
        };
        bf16x8 a_l[2], a_m[2], a_h[2], b_h[2], b_m[2], b_l[2];
        fragA(2, a_l);
        fragB(0, b_h);
        fragA(1, a_m);
        mm(a_l, b_h);
        fragB(1, b_m);
        mm(a_m, b_h);
        fragA(0, a_h);
        mm(a_m, b_m);
        fragB(2, b_l);
        mm(a_h, b_m);
        mm(a_h, b_l);
        mm(a_h, b_h);
#if F32S_ABL != 3
        if (kt + 1 < nk) stash((kt + 1) & 1);
#endif
        __syncthreads();
    }

    f32_store_tiles(d, acc, m0 + wc * 64, n0 + wr * 64, l31, hi, [](int64_t m) { return (size_t)m; });
}

// ---- the pipelined form of the same kernel (large M, Cout > 128): ONE 16-wave workgroup per CU, tiles two K steps ahead ----
// The ablation of f32s_gemm_kernel (tools/exp/f32s_abl.sh, 17 x 512 x 768, 128 -> 128: 10.35 ms; without the loop's global loads 8.3;
// without its MFMAs 6.0; without the split arithmetic 10.1) says an iteration is a load round trip (~1.5 us) that two workgroups per
// CU overlap badly — not VALU, not LDS.  Here a workgroup is 1024 threads on 256 ch x 256 pix (4 x 4 waves of 64 x 64: the same 64
// accumulator registers and four waves per SIMD, half the staged bytes per FLOP, TWO 16-byte pieces per thread and K tile instead of
// three): tile kt + 2 is REQUESTED while tile kt is multiplied and tile kt + 1 — requested one iteration earlier — is split and
// stored behind the first product groups (two register sets, two LDS buffers of 48 KB).
constexpr int P_BM = 256, P_BN = 256;
constexpr int P_PLANE = (P_BM + P_BN) * 32, P_BUF = 3 * P_PLANE;

template <int MODE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void f32p_gemm_kernel(const CcGemmF32Desc d) {
    // MODE 3: one output parity (py, px) = d.upsample - 2 of `conv3x3(nearest_upsample_2x(x))` as a 2 x 2 convolution on x itself
    // (model.py:56-71; the merged taps are the packer's, vae_f32.pack_f32_parities): K = [4][Cpad], the window of low-resolution
    // pixel (y, x) starts at (y - 1 + py, x - 1 + px), its output row is pixel (2 y + py, 2 x + px) of the up-sampled frame.
    constexpr bool CONV = MODE != 0, UPS = MODE == 2, PAR = MODE == 3;
    extern __shared__ __attribute__((aligned(16))) char psmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int ct_n = (d.N + P_BM - 1) / P_BM;
    const int64_t pt = blockIdx.x / ct_n;
    const int ct = blockIdx.x - (int)pt * ct_n;
    const int64_t m0 = pt * P_BN;
    const int n0 = ct * P_BM;
    const int nk = d.Kpad / S_BK;

    // staging: thread -> weight row tid >> 2 and pixel row tid >> 2; floats 4 (tid & 3) .. + 3 of the K tile
    const int srow = tid >> 2, sg = tid & 3;
    const float* wsrc = d.W + (size_t)min(n0 + srow, d.N - 1) * d.ldw + sg * 4;
    const float* asrc;
    uint32_t geo = 0;             // CONV: bits 0-8 = taps inside the frame; UPS: bit 9 / 10 = parity of the top-left tap's virtual x / y
    {
        const int64_t m = min(m0 + srow, d.M - 1);
        if constexpr (CONV) {
            const int hw = d.Hout * d.Wout;
            const int64_t f = m / hw;
            const int r = (int)(m - f * hw);
            const int y = r / d.Wout, x = r - y * d.Wout;
            const int oy = PAR ? y - 1 + ((d.upsample - 2) >> 1) : y * d.stride - d.pad;
            const int ox = PAR ? x - 1 + ((d.upsample - 2) & 1) : x * d.stride - d.pad;
            const int Hv = UPS ? 2 * d.Hin : d.Hin, Wv = UPS ? 2 * d.Win : d.Win;
#pragma unroll
            for (int t = 0; t < (PAR ? 4 : 9); ++t) {
                const int iy = oy + (PAR ? t >> 1 : t / 3), ix = ox + (PAR ? t & 1 : t % 3);
                geo |= (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) ? 1u << t : 0u;
            }
            if constexpr (UPS) geo |= (uint32_t)(ox & 1) << 9 | (uint32_t)(oy & 1) << 10;
            const int sy = UPS ? oy >> 1 : oy, sx = UPS ? ox >> 1 : ox;
            asrc = d.A + (f * (int64_t)d.Hin * d.Win + (int64_t)sy * d.Win + sx) * d.lda + sg * 4;
        } else {
            asrc = d.A + (size_t)m * d.lda + sg * 4;
        }
    }
    const int cmax = d.Cin - sg * 4;

    // tile kt >= nk reads as zeros (an odd tile count is rounded up: the loop below is two steps per trip, no early exit — a second
    // exit made hipcc carry the accumulators through scratch)
    auto fetch = [&](int kt, f32x4& ra, f32x4& rb) {
        const int k0 = kt * S_BK;
        const bool live = kt < nk;
        ra = live ? *(const f32x4*)(wsrc + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (CONV) {
            const int tap = k0 / d.Cpad, c0 = k0 - tap * d.Cpad;
            const int ky = PAR ? tap >> 1 : tap / 3, kx = PAR ? tap & 1 : tap - 3 * ky;
            int off = (ky * d.Win + kx) * d.lda + c0;
            if constexpr (UPS) off = (((((geo >> 10) & 1) + ky) >> 1) * d.Win + ((((geo >> 9) & 1) + kx) >> 1)) * d.lda + c0;
            const bool ok = live && ((geo >> tap) & 1) && c0 < cmax;
            rb = ok ? *(const f32x4*)(asrc + off) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            rb = live && k0 < cmax ? *(const f32x4*)(asrc + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // byte offset of (row srow, k piece sg) inside the weight half of a plane; the pixel rows follow P_BM rows later
    const int so = srow * 32 + ((((sg >> 1) ^ (srow >> 3)) & 1) << 4) + ((sg & 1) << 3);
    auto stash = [&](char* s, f32x4 ra, f32x4 rb) {
        const F32Split sa = f32_split3(ra), sb = f32_split3(rb);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *(u32x2*)(s + p * P_PLANE + so) = sa.p[p];
            *(u32x2*)(s + p * P_PLANE + P_BM * 32 + so) = sb.p[p];
        }
    };
    const int fo = l31 * 32 + ((hi ^ (l31 >> 3)) & 1) * 16;
    const int fa0 = (wr * 64) * 32 + fo, fb0 = (P_BM + wc * 64) * 32 + fo;

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    // one K tile: six product groups on buffer `buf`; the NEXT tile (already in ra / rb) is split and stored into the other buffer
    // behind the second group (after the last tile: zeros into a buffer nobody reads)
    auto step = [&](int buf, f32x4 ra, f32x4 rb) {
        const char* s = psmem + buf * P_BUF;
        auto fragA = [&](int p, bf16x8 (&f)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) f[t] = *(const bf16x8*)(s + p * P_PLANE + fa0 + t * 1024);
        };
        auto fragB = [&](int p, bf16x8 (&f)[2]) {
#pragma unroll
            for (int t = 0; t < 2; ++t) f[t] = *(const bf16x8*)(s + p * P_PLANE + fb0 + t * 1024);
        };
        auto mm = [&](const bf16x8 (&a)[2], const bf16x8 (&b)[2]) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
#if F32S_ABL == 4
                    acc[ti][tj][0] += (float)a[ti][0] * (float)b[tj][0];
#else
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
#endif
                }
        };
        bf16x8 a_l[2], a_m[2], a_h[2], b_h[2], b_m[2], b_l[2];
        __builtin_amdgcn_sched_barrier(0);          // (the scheduler otherwise carries MFMAs across the workgroup barrier and spills accumulators)
        fragA(2, a_l);
        fragB(0, b_h);
        fragA(1, a_m);
        mm(a_l, b_h);
        fragB(1, b_m);
        mm(a_m, b_h);
        __builtin_amdgcn_sched_barrier(0);
#if F32S_ABL != 3
        stash(psmem + (buf ^ 1) * P_BUF, ra, rb);
#endif
        __builtin_amdgcn_sched_barrier(0);
        fragA(0, a_h);
        mm(a_m, b_m);
        fragB(2, b_l);
        mm(a_h, b_m);
        mm(a_h, b_l);
        mm(a_h, b_h);
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4 ra0, rb0, ra1, rb1;
    fetch(0, ra0, rb0);
    fetch(1, ra1, rb1);
    stash(psmem, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
        // LDS buffer 0 holds tile kt, register set 1 tile kt + 1; set 0 is free
#if F32S_ABL != 2
        fetch(kt + 2, ra0, rb0);
#endif
        step(0, ra1, rb1);
#if F32S_ABL != 5
        __syncthreads();
#endif
#if F32S_ABL != 2
        fetch(kt + 3, ra1, rb1);
#endif
        step(1, ra0, rb0);
#if F32S_ABL != 5
        __syncthreads();
#endif
    }

    f32_store_tiles(d, acc, m0 + wc * 64, n0 + wr * 64, l31, hi, [&](int64_t m) {
        if constexpr (PAR) {
            const int hw = d.Hin * d.Win;
            const int64_t f = m / hw;
            const int r = (int)(m - f * hw);
            const int y = r / d.Win, x = r - y * d.Win;
            return ((size_t)f * 2 * d.Hin + 2 * y + ((d.upsample - 2) >> 1)) * (2 * d.Win) + 2 * x + ((d.upsample - 2) & 1);
        } else {
            return (size_t)m;
        }
    });
}

template <int MODE>
static int f32p_launch(const CcGemmF32Desc& d, hipStream_t s) {
    constexpr int LDS = 2 * P_BUF;
    static unsigned long long done = 0;
    const int rc = cc_max_dynamic_lds((const void*)f32p_gemm_kernel<MODE>, LDS, &done, "f32p_gemm_kernel");
    if (rc != CCEDIT_OK) return rc;
    const int64_t blocks = ((d.M + P_BN - 1) / P_BN) * ((d.N + P_BM - 1) / P_BM);
    cc_note_kernel("f32p_gemm_kernel 256ch x 256pix%s, six bf16 products",
                   MODE == 3 ? ", upsample parity taps" : MODE == 2 ? ", 3x3 taps on the 2x up-sampled source" : MODE == 1 ? ", 3x3 taps" : "");
    hipLaunchKernelGGL((f32p_gemm_kernel<MODE>), dim3((unsigned)blocks), dim3(1024), LDS, s, d);
    return cc_launch_status("f32p_gemm_kernel");
}

// ---- GroupNorm(32) over fp32 frames ----
// VEC = 4 (C a multiple of 128: a 16-byte granule never straddles groups) or 1 (C = 32 / 64: the reduced-width test models).
template <int VEC>
struct GnVec {
    float v[VEC];
    __device__ __forceinline__ void load(const float* p) {
        if constexpr (VEC == 4) {
            const f32x4 t = *(const f32x4*)p;
            v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
        } else {
            v[0] = *p;
        }
    }
    __device__ __forceinline__ void store(float* p) const {
        if constexpr (VEC == 4)
            *(f32x4*)p = f32x4{v[0], v[1], v[2], v[3]};
        else
            *p = v[0];
    }
};

// statistics: a block takes a slice of a frame's rows; thread = (granule of VEC channels, row lane); fp64 partial sums, folded
// per group and added to the frame's fp64 slots
template <int VEC>
__global__ __launch_bounds__(256) void gn32_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int hw, int C, int rows_per_block) {
    __shared__ double red[256][2];
    const int frame = blockIdx.y;
    const int gpr = C / VEC;                                 // granules per row
    const int g = threadIdx.x % gpr, rl = threadIdx.x / gpr, nrl = 256 / gpr;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, hw);
    const float* base = x + ((size_t)frame * hw) * C + g * VEC;
    double s = 0.0, q = 0.0;
    int r = r0 + rl;
    for (; r + 3 * nrl < r1; r += 4 * nrl) {                 // four rows in flight
        GnVec<VEC> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u].load(base + (size_t)(r + u * nrl) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const double t = (double)v[u].v[e];
                s += t;
                q += t * t;
            }
    }
    for (; r < r1; r += nrl) {
        GnVec<VEC> v;
        v.load(base + (size_t)r * C);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const double t = (double)v.v[e];
            s += t;
            q += t * t;
        }
    }
    red[threadIdx.x][0] = s;
    red[threadIdx.x][1] = q;
    __syncthreads();
    if (threadIdx.x < 32) {                                   // one thread per group: fixed summation order inside the block
        const int gpg = gpr / 32;                             // granules per group
        double ts = 0.0, tq = 0.0;
        for (int l = 0; l < nrl; ++l)
            for (int k = 0; k < gpg; ++k) {
                ts += red[l * gpr + threadIdx.x * gpg + k][0];
                tq += red[l * gpr + threadIdx.x * gpg + k][1];
            }
        atomicAdd(stats + ((size_t)frame * 32 + threadIdx.x) * 2, ts);
        atomicAdd(stats + ((size_t)frame * 32 + threadIdx.x) * 2 + 1, tq);
    }
}

template <int VEC, bool SILU>
__global__ __launch_bounds__(256) void gn32_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const double* __restrict__ stats,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int hw, int C, float eps,
                                                         int rows_per_block) {
    const int frame = blockIdx.y;
    const int gpr = C / VEC;
    const int g = threadIdx.x % gpr, rl = threadIdx.x / gpr, nrl = 256 / gpr;
    const int grp = (g * VEC) / (C / 32);
    const double cnt = (double)hw * (C / 32);
    const double mean_d = stats[((size_t)frame * 32 + grp) * 2] / cnt;
    double var = stats[((size_t)frame * 32 + grp) * 2 + 1] / cnt - mean_d * mean_d;
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
    GnVec<VEC> ga, be;
    ga.load(gamma + g * VEC);
    be.load(beta + g * VEC);
    const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, hw);
    const size_t off = ((size_t)frame * hw) * C + g * VEC;
    auto norm = [&](const GnVec<VEC>& v) {
        GnVec<VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float t = (v.v[e] - mean) * rstd * ga.v[e] + be.v[e];
            if (SILU) t = t / (1.0f + expf(-t));
            o.v[e] = t;
        }
        return o;
    };
    int r = r0 + rl;
    for (; r + 3 * nrl < r1; r += 4 * nrl) {
        GnVec<VEC> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u].load(x + off + (size_t)(r + u * nrl) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) norm(v[u]).store(y + off + (size_t)(r + u * nrl) * C);
    }
    for (; r < r1; r += nrl) {
        GnVec<VEC> v;
        v.load(x + off + (size_t)r * C);
        norm(v).store(y + off + (size_t)r * C);
    }
}

// ---- in-place row softmax of fp32 scores: one block per row, the row held in registers ----
template <int PER>
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(float* __restrict__ s, int cols, int64_t ld, float scale) {
    __shared__ float red[4];
    float* row = s + (size_t)blockIdx.x * ld;
    float v[PER];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = threadIdx.x + 256 * i;
        v[i] = c < cols ? row[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        v[i] = expf(v[i] - mx);
        sum += v[i];
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < cols) row[c] = v[i] * inv;
    }
}

// Rows longer than 8192 columns (latents beyond 96 x 85: e.g. 768 x 768 frames = 9216 tokens; the reference has no such limit,
// model.py:180-195): the same three steps — maximum, sum of exponentials, normalise — with the row re-read from memory instead of
// held in registers; same per-thread column assignment and reduction order as the register kernel.
__global__ __launch_bounds__(256) void softmax_rows_f32_long_kernel(float* __restrict__ s, int cols, int64_t ld, float scale) {
    __shared__ float red[4];
    float* row = s + (size_t)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, row[c] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float e = expf(row[c] * scale - mx);
        row[c] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int c = threadIdx.x; c < cols; c += 256) row[c] *= inv;
}

}  // namespace

extern "C" int ccedit_gemm_f32(const CcGemmF32Desc* desc, void* stream) {
    CC_CHECK_ARG(desc != nullptr, "ccedit_gemm_f32: null descriptor");
    const CcGemmF32Desc& d = *desc;
    CC_CHECK_ARG(d.A && d.W && d.out, "ccedit_gemm_f32: null operand");
    CC_CHECK_ARG(d.M >= 0 && d.N > 0 && d.Cin > 0, "ccedit_gemm_f32: bad sizes (M=%lld N=%d Cin=%d)", (long long)d.M, d.N, d.Cin);
    CC_CHECK_ARG(d.mode == 0 || d.mode == 1, "ccedit_gemm_f32: mode %d (0 = Linear / Conv 1x1, 1 = Conv2d 3x3)", d.mode);
    CC_CHECK_ARG(d.Cin % 4 == 0 && d.lda % 4 == 0 && d.lda >= d.Cin, "ccedit_gemm_f32: Cin (%d) and lda (%d) must be multiples of 4, lda >= Cin", d.Cin, d.lda);
    CC_CHECK_ARG(d.Cpad % 16 == 0 && d.Cpad >= d.Cin, "ccedit_gemm_f32: Cpad (%d) must be a multiple of 16 and >= Cin", d.Cpad);
    const bool parity = d.mode == 1 && d.upsample >= 2;          // one output parity of upsample + conv 3x3: four merged taps
    CC_CHECK_ARG(d.upsample >= 0 && d.upsample <= 5, "ccedit_gemm_f32: upsample %d (0 none, 1 fused nearest-2x source, 2 + 2 py + px = one output parity)", d.upsample);
    CC_CHECK_ARG(d.Kpad == (d.mode == 1 ? (parity ? 4 : 9) : 1) * d.Cpad && d.ldw >= d.Kpad && d.ldw % 4 == 0,
                 "ccedit_gemm_f32: Kpad (%d) must be taps x Cpad, ldw (%d) >= Kpad and a multiple of 4", d.Kpad, d.ldw);
    CC_CHECK_ARG(d.ldc >= d.N && (!d.res || d.ldr >= d.N), "ccedit_gemm_f32: ldc / ldr smaller than N");
    CC_CHECK_ARG(((uintptr_t)d.A | (uintptr_t)d.W) % 16 == 0, "ccedit_gemm_f32: A and W must be 16-byte aligned");
    CC_CHECK_ARG((d.ldc % 4 != 0 || (uintptr_t)d.out % 16 == 0) && (!d.bias || (uintptr_t)d.bias % 16 == 0) &&
                     (!d.res || d.ldr % 4 != 0 || (uintptr_t)d.res % 16 == 0),
                 "ccedit_gemm_f32: out / bias / res must be 16-byte aligned when their row stride is a multiple of 4");
    if (d.mode == 1) {
        CC_CHECK_ARG(d.Hin > 0 && d.Win > 0 && d.Hout > 0 && d.Wout > 0 && d.stride >= 1 && d.pad >= 0, "ccedit_gemm_f32: bad convolution geometry");
        CC_CHECK_ARG(!parity || (d.Hout == d.Hin && d.Wout == d.Win && d.stride == 1 && d.pad == 1 && !d.res),
                     "ccedit_gemm_f32: a parity conv runs over the SOURCE pixels (Hout x Wout = Hin x Win, stride 1, pad 1, no residual); out holds the 2 Hin x 2 Win frames");
        CC_UNSUPPORTED(parity && cc_policy().f32_split == 0, "ccedit_gemm_f32: the parity form of upsample + conv exists on the six-product kernels only (policy f32_split)");
        CC_CHECK_ARG(d.M % ((int64_t)d.Hout * d.Wout) == 0, "ccedit_gemm_f32: M (%lld) is not a whole number of %d x %d output frames", (long long)d.M, d.Hout, d.Wout);
    }
    if (d.M == 0) return CCEDIT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (cc_policy().f32_split) {
        const int64_t sblocks = ((d.M + S_BN - 1) / S_BN) * ((d.N + S_BM - 1) / S_BM);
        CC_CHECK_ARG(sblocks < (1LL << 31), "ccedit_gemm_f32: too many tiles");
        const int ups = d.mode == 1 && d.upsample;
        CC_UNSUPPORTED(ups && !(d.stride == 1 && d.pad == 1), "ccedit_gemm_f32: the fused up-sampling gather takes stride 1, pad 1 (got %d, %d)", d.stride, d.pad);
        if (parity) return f32p_launch<3>(d, s);
        if (cc_policy().f32_split == 1 && d.M >= 4096 && d.N > 128) {          // (2: always the two-workgroups-per-CU kernel)
            const int mode = ups ? 2 : d.mode;
            return mode == 2 ? f32p_launch<2>(d, s) : mode == 1 ? f32p_launch<1>(d, s) : f32p_launch<0>(d, s);
        }
        if (d.mode == 1 && d.upsample && d.stride == 1 && d.pad == 1) {
            cc_note_kernel("f32s_gemm_kernel 128ch x 256pix, 3x3 taps on the 2x up-sampled source, six bf16 products");
            hipLaunchKernelGGL((f32s_gemm_kernel<2>), dim3((unsigned)sblocks), dim3(512), 0, s, d);
        } else if (d.mode == 1) {
            cc_note_kernel("f32s_gemm_kernel 128ch x 256pix, 3x3 taps, six bf16 products");
            hipLaunchKernelGGL((f32s_gemm_kernel<1>), dim3((unsigned)sblocks), dim3(512), 0, s, d);
        } else {
            cc_note_kernel("f32s_gemm_kernel 128ch x 256pix, six bf16 products");
            hipLaunchKernelGGL((f32s_gemm_kernel<0>), dim3((unsigned)sblocks), dim3(512), 0, s, d);
        }
        return cc_launch_status("f32s_gemm_kernel");
    }
    const int64_t blocks = ((d.M + F_BM - 1) / F_BM) * ((d.N + F_BN - 1) / F_BN);
    CC_CHECK_ARG(blocks < (1LL << 31), "ccedit_gemm_f32: too many tiles");
    if (d.mode == 1) {
        cc_note_kernel("f32_gemm_kernel 128ch x 128pix, 3x3 taps");
        hipLaunchKernelGGL((f32_gemm_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    } else {
        cc_note_kernel("f32_gemm_kernel 128ch x 128pix");
        hipLaunchKernelGGL((f32_gemm_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    }
    return cc_launch_status("f32_gemm_kernel");
}

extern "C" int ccedit_groupnorm_f32(const float* x, float* y, const float* gamma, const float* beta, double* stats, int32_t frames, int32_t hw,
                                    int32_t C, float eps, int32_t silu, void* stream) {
    CC_CHECK_ARG(x && y && gamma && beta && stats, "ccedit_groupnorm_f32: null pointer");
    CC_CHECK_ARG(frames >= 0 && hw > 0, "ccedit_groupnorm_f32: bad sizes");
    const int vec = C % 128 == 0 ? 4 : 1;
    CC_CHECK_ARG(C % 32 == 0 && C / vec <= 256 && 256 % (C / vec) == 0,
                 "ccedit_groupnorm_f32: C (%d) must be 32, 64 or 128, 256, 512, 1024 (32 groups, whole granules per thread column)", C);
    CC_CHECK_ARG(((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) % 16 == 0, "ccedit_groupnorm_f32: 16-byte alignment");
    if (frames == 0) return CCEDIT_OK;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(stats, 0, (size_t)frames * 64 * sizeof(double), s);
    if (e != hipSuccess) {
        cc_set_error("ccedit_groupnorm_f32: hipMemsetAsync: %s", hipGetErrorString(e));
        return (int)e;
    }
    // ~2048 blocks over the frames, at least 64 rows each
    int per_frame = (2048 + frames - 1) / frames;
    int rows = (hw + per_frame - 1) / per_frame;
    rows = rows < 64 ? 64 : rows;
    const int bx = (hw + rows - 1) / rows;
#define GN32_GO(V)                                                                                                                  \
    do {                                                                                                                            \
        hipLaunchKernelGGL((gn32_stats_kernel<V>), dim3(bx, frames), dim3(256), 0, s, x, stats, hw, C, rows);                       \
        if (silu)                                                                                                                   \
            hipLaunchKernelGGL((gn32_apply_kernel<V, true>), dim3(bx, frames), dim3(256), 0, s, x, y, stats, gamma, beta, hw, C, eps, rows);  \
        else                                                                                                                        \
            hipLaunchKernelGGL((gn32_apply_kernel<V, false>), dim3(bx, frames), dim3(256), 0, s, x, y, stats, gamma, beta, hw, C, eps, rows); \
    } while (0)
    if (vec == 4)
        GN32_GO(4);
    else
        GN32_GO(1);
#undef GN32_GO
    return cc_launch_status("gn32 kernels");
}

extern "C" int ccedit_softmax_rows_f32(float* s, int64_t rows, int32_t cols, int64_t ld, float scale, void* stream) {
    CC_CHECK_ARG(s != nullptr && rows >= 0 && cols > 0 && ld >= cols, "ccedit_softmax_rows_f32: bad arguments");
    CC_CHECK_ARG(rows < (1LL << 31), "ccedit_softmax_rows_f32: too many rows");
    if (rows == 0) return CCEDIT_OK;
    hipStream_t st = (hipStream_t)stream;
    if (cols <= 256 * 8)
        hipLaunchKernelGGL((softmax_rows_f32_kernel<8>), dim3((unsigned)rows), dim3(256), 0, st, s, cols, ld, scale);
    else if (cols <= 256 * 32)
        hipLaunchKernelGGL((softmax_rows_f32_kernel<32>), dim3((unsigned)rows), dim3(256), 0, st, s, cols, ld, scale);
    else          // (a thread reads back only columns it wrote itself: no barrier between the passes is needed for the data)
        hipLaunchKernelGGL(softmax_rows_f32_long_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, cols, ld, scale);
    return cc_launch_status("softmax_rows_f32_kernel");
}
