// GroupNorm(32) (+SiLU) in its spatial and temporal forms and LayerNorm, for the frames-outermost
// channels-last layout [frames][pixels][C] (bf16 in/out, fp32 statistics).  All HBM-bound.
//
// Thread mapping shared by every kernel here: one wave reads one pixel row at a time; lane l owns the
// 16-byte granules (8 channels) l, l+64, l+128, ... of the row, so a wave's load of one pixel is a
// single contiguous, fully coalesced C*2-byte burst and per-channel affine parameters live in registers.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kMaxCols = 5;    // C <= 2560

// ------------------------------------------------------------------------------------------
// spatial GroupNorm: statistics over (C/32 channels x H*W pixels) of one frame
// ------------------------------------------------------------------------------------------
constexpr int kGnWaves = 4;
constexpr int kGnPixPerBlock = 64;

// pixels per 4-wave block: 64 for large frames, down to 4 (one row per wave) so that the small latent levels
// (8x12 pixels x 34 frames) still put a few thousand workgroups on the 256 CUs
inline int gn_pix_per_block(int hw, int frames, int lo, int64_t want_blocks) {
    int ppb = kGnPixPerBlock;
    while (ppb > lo && (int64_t)((hw + ppb - 1) / ppb) * frames < want_blocks) ppb >>= 1;
    return ppb;
}

// Run-to-run reproducibility of the statistics: a wave accumulates into its OWN LDS slots (same-wave LDS atomics
// retire in program / lane order), the waves' slots are added in a fixed order, and only the cross-workgroup sum uses
// global atomics — in double, so that the arrival order moves the total by ~1e-16 relative, far below the fp32 rounding
// of the mean / rstd derived from it.  (With fp32 atomics the order noise flipped bf16 roundings downstream and two runs
// of the same clip differed by the full bf16 noise floor, ~2 % rel. RMS at 17x512x768.)
__global__ __launch_bounds__(256) void gn_spatial_stats_kernel(const bf16* __restrict__ x, double* __restrict__ stats,
                                                               int hw, int C, int pix_per_block) {
    __shared__ float s_sum[kGnWaves][32], s_sq[kGnWaves][32];
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G8 = C >> 3, cpg = C >> 5;
    if (threadIdx.x < kGnWaves * 32) {
        (&s_sum[0][0])[threadIdx.x] = 0.f;
        (&s_sq[0][0])[threadIdx.x] = 0.f;
    }
    __syncthreads();
    float sum[kMaxCols][8], sq[kMaxCols][8];
#pragma unroll
    for (int k = 0; k < kMaxCols; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[k][e] = sq[k][e] = 0.f;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, hw);
    const bf16* xf = x + (size_t)frame * hw * C;
    for (int pix = p0 + wave; pix < p1; pix += kGnWaves) {
        const bf16* row = xf + (size_t)pix * C;
#pragma unroll
        for (int k = 0; k < kMaxCols; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                const bf16x8 v = *(const bf16x8*)(row + gc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf2f(v[e]);
                    sum[k][e] += f;
                    sq[k][e] += f * f;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxCols; ++k) {
        const int gc = lane + 64 * k;
        if (gc < G8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = (gc * 8 + e) / cpg;
                atomicAdd(&s_sum[wave][g], sum[k][e]);
                atomicAdd(&s_sq[wave][g], sq[k][e]);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double ts = 0.0, tq = 0.0;
#pragma unroll
        for (int w = 0; w < kGnWaves; ++w) {
            ts += (double)s_sum[w][threadIdx.x];
            tq += (double)s_sq[w][threadIdx.x];
        }
        unsafeAtomicAdd(&stats[(frame * 32 + threadIdx.x) * 2 + 0], ts);
        unsafeAtomicAdd(&stats[(frame * 32 + threadIdx.x) * 2 + 1], tq);
    }
}

template <int COLS>      // 16-byte granules per lane: ceil(C / 512); fewer live coefficient registers -> more resident waves
__global__ __launch_bounds__(256) void gn_spatial_apply_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                               const double* __restrict__ stats,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int hw, int C, float eps,
                                                               int silu, int pix_per_block) {
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G8 = C >> 3, cpg = C >> 5;
    const float inv_n = 1.0f / ((float)cpg * (float)hw);
    // per-lane affine coefficients of its 8 * COLS channels.  A granule of 8 aligned channels spans at most two groups
    // (C/32 >= 8): two statistics loads and four 16-byte gamma/beta loads per granule — this set-up is paid once per
    // wave and would otherwise dominate at the small latent levels, where a wave normalises only a few pixel rows.
    float a[COLS][8], b[COLS][8];
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const int gc = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[k][e] = b[k][e] = 0.f;
        if (gc < G8 && cpg < 8) {                          // narrow test configurations: a granule spans several groups
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = gc * 8 + e;
                const int g = c / cpg;
                const float mean = (float)stats[(frame * 32 + g) * 2] * inv_n;
                const float rstd = rsqrtf(fmaxf((float)stats[(frame * 32 + g) * 2 + 1] * inv_n - mean * mean, 0.f) + eps);
                a[k][e] = rstd * gamma[c];
                b[k][e] = beta[c] - mean * a[k][e];
            }
        } else if (gc < G8) {
            const int c0 = gc * 8;
            const int g0 = c0 / cpg, g1 = min(g0 + 1, 31);
            const int split = (g0 + 1) * cpg - c0;           // channels of this granule that belong to group g0
            const double* sp0 = stats + (frame * 32 + g0) * 2;
            const double* sp1 = stats + (frame * 32 + g1) * 2;
            const f32x2 st0 = {(float)sp0[0], (float)sp0[1]}, st1 = {(float)sp1[0], (float)sp1[1]};
            const f32x4 ga0 = *(const f32x4*)(gamma + c0), ga1 = *(const f32x4*)(gamma + c0 + 4);
            const f32x4 be0 = *(const f32x4*)(beta + c0), be1 = *(const f32x4*)(beta + c0 + 4);
            const float m0 = st0[0] * inv_n, m1 = st1[0] * inv_n;
            const float r0 = rsqrtf(fmaxf(st0[1] * inv_n - m0 * m0, 0.f) + eps), r1 = rsqrtf(fmaxf(st1[1] * inv_n - m1 * m1, 0.f) + eps);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ga = e < 4 ? ga0[e & 3] : ga1[e & 3], be = e < 4 ? be0[e & 3] : be1[e & 3];
                const bool first = e < split;
                a[k][e] = (first ? r0 : r1) * ga;
                b[k][e] = be - (first ? m0 : m1) * a[k][e];
            }
        }
    }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, hw);
    const bf16* xf = x + (size_t)frame * hw * C;
    bf16* yf = y + (size_t)frame * hw * C;
    for (int pix = p0 + wave; pix < p1; pix += kGnWaves) {
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                const bf16x8 v = *(const bf16x8*)(xf + (size_t)pix * C + gc * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = bf2f(v[e]) * a[k][e] + b[k][e];
                    if (silu) f = silu_f(f);
                    o[e] = f2bf(f);
                }
                *(bf16x8*)(yf + (size_t)pix * C + gc * 8) = o;
            }
        }
    }
}

// One-pass form for small frames (the 8x12 level: 96 pixels x 40 channels per group = 7.7 KB): a workgroup owns one (frame, group),
// keeps its hw x (C/32) values in registers between the statistics and the normalisation — one read, one write, one launch, where
// the general path is a zero-fill, a statistics kernel and an apply kernel over a tensor the caches hold anyway (38 us per site
// against 8; 21 sites per step have no producer statistics at that level).  C % 256 == 0 (a group is whole 16-byte granules),
// at most 8 granules per thread.  Fixed summation order: lanes -> wave (butterfly), waves in index order.
constexpr int kGnOneUnits = 8;
__global__ __launch_bounds__(256) void gn_spatial_onepass_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 int hw, int C, float eps, int silu) {
    __shared__ float s_part[4][2];
    const int frame = blockIdx.y, grp = blockIdx.x;
    const int cpg = C >> 5, gpg = cpg >> 3;                 // channels / granules per group
    const int units = hw * gpg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16* xf = x + (size_t)frame * hw * C + grp * cpg;
    bf16* yf = y + (size_t)frame * hw * C + grp * cpg;
    bf16x8 v[kGnOneUnits];
    int off[kGnOneUnits];
#pragma unroll
    for (int k = 0; k < kGnOneUnits; ++k) {
        const int u = k * 256 + threadIdx.x;
        const int pix = u / gpg, gr = u - pix * gpg;
        off[k] = u < units ? pix * C + gr * 8 : -1;
        if (off[k] >= 0) v[k] = *(const bf16x8*)(xf + off[k]);
    }
    // Two passes over the REGISTER-resident values (no extra memory traffic): the mean first, then the sum of squared deviations from
    // it — no E[x^2] - mean^2 cancellation when |mean| >> std (the other spatial paths accumulate raw moments in double; this one keeps
    // fp32 lanes but never subtracts two large numbers).  Wave partials meet in double, in wave-index order.
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kGnOneUnits; ++k)
        if (off[k] >= 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s += bf2f(v[k][e]);
        }
    s = wave_sum(s);
    if (lane == 0) s_part[wave][0] = s;
    __syncthreads();
    const double inv_n = 1.0 / ((double)cpg * (double)hw);
    const float mean = (float)((((double)s_part[0][0] + (double)s_part[1][0]) + ((double)s_part[2][0] + (double)s_part[3][0])) * inv_n);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < kGnOneUnits; ++k)
        if (off[k] >= 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dv = bf2f(v[k][e]) - mean;
                q = fmaf(dv, dv, q);
            }
        }
    q = wave_sum(q);
    if (lane == 0) s_part[wave][1] = q;
    __syncthreads();
    const double tq = ((double)s_part[0][1] + (double)s_part[1][1]) + ((double)s_part[2][1] + (double)s_part[3][1]);
    const float rstd = rsqrtf((float)(tq * inv_n) + eps);
#pragma unroll
    for (int k = 0; k < kGnOneUnits; ++k)
        if (off[k] >= 0) {
            const int c0 = grp * cpg + (off[k] % C);                       // (off % C = granule offset inside the group slice)
            const f32x4 ga0 = *(const f32x4*)(gamma + c0), ga1 = *(const f32x4*)(gamma + c0 + 4);
            const f32x4 be0 = *(const f32x4*)(beta + c0), be1 = *(const f32x4*)(beta + c0 + 4);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = rstd * (e < 4 ? ga0[e & 3] : ga1[e & 3]);
                float f = (bf2f(v[k][e]) - mean) * a + (e < 4 ? be0[e & 3] : be1[e & 3]);
                if (silu) f = silu_f(f);
                o[e] = f2bf(f);
            }
            *(bf16x8*)(yf + off[k]) = o;
        }
}

// ------------------------------------------------------------------------------------------
// temporal GroupNorm: statistics over (C/32 channels x T frames) at one pixel of one clip.
// One wave per (clip b, pixel); two sweeps over the T rows (the second one hits L2).
// ------------------------------------------------------------------------------------------
constexpr int kGtCols = 3;     // temporal norms only see C in {320, 640, 1280} (<= 1536)

__global__ __launch_bounds__(256) void gn_temporal_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int B, int T, int hw, int C,
                                                          float eps, int silu) {
    __shared__ float s_sum[4][32], s_sq[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wp = (int64_t)blockIdx.x * 4 + wave;     // (b, pixel) index
    const bool active = wp < (int64_t)B * hw;
    const int b = active ? (int)(wp / hw) : 0;
    const int pix = active ? (int)(wp - (int64_t)b * hw) : 0;
    const int G8 = C >> 3, cpg = C >> 5;
    if (lane < 32) {
        s_sum[wave][lane] = 0.f;
        s_sq[wave][lane] = 0.f;
    }
    __syncthreads();
    const size_t fstride = (size_t)hw * C;
    const bf16* xb = x + ((size_t)b * T * hw + pix) * C;
    bf16* yb = y + ((size_t)b * T * hw + pix) * C;
    float sum[kGtCols][8], sq[kGtCols][8];
#pragma unroll
    for (int k = 0; k < kGtCols; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[k][e] = sq[k][e] = 0.f;
    if (active) {
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int k = 0; k < kGtCols; ++k) {
                const int gc = lane + 64 * k;
                if (gc < G8) {
                    const bf16x8 v = *(const bf16x8*)(xb + t * fstride + gc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = bf2f(v[e]);
                        sum[k][e] += f;
                        sq[k][e] += f * f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kGtCols; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int g = (gc * 8 + e) / cpg;
                    atomicAdd(&s_sum[wave][g], sum[k][e]);
                    atomicAdd(&s_sq[wave][g], sq[k][e]);
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const float inv_n = 1.0f / ((float)cpg * (float)T);
    float a[kGtCols][8], bb[kGtCols][8];
#pragma unroll
    for (int k = 0; k < kGtCols; ++k) {
        const int gc = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[k][e] = 0.f;
            bb[k][e] = 0.f;
            if (gc < G8) {
                const int c = gc * 8 + e;
                const int g = c / cpg;
                const float mean = s_sum[wave][g] * inv_n;
                const float var = fmaxf(s_sq[wave][g] * inv_n - mean * mean, 0.f);
                const float rstd = rsqrtf(var + eps);
                a[k][e] = rstd * gamma[c];
                bb[k][e] = beta[c] - mean * a[k][e];
            }
        }
    }
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int k = 0; k < kGtCols; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                const bf16x8 v = *(const bf16x8*)(xb + t * fstride + gc * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = bf2f(v[e]) * a[k][e] + bb[k][e];
                    if (silu) f = silu_f(f);
                    o[e] = f2bf(f);
                }
                *(bf16x8*)(yb + t * fstride + gc * 8) = o;
            }
        }
    }
}

// Single-sweep form for short clips (T <= kGtCacheT): a wave owns one (clip, pixel, channel slice) — a slice is a
// whole number of groups, at most 512 channels — keeps its T rows in registers between the statistics and the
// normalisation, and issues all T loads back to back.  The launch has only B*H*W*slices waves, each touching T
// rows one frame apart, so it is bound by load latency: memory-level parallelism per wave and the number of
// waves are what count (the 16x24 / 8x12 levels have 768 / 192 pixels).
constexpr int kGtCacheT = 20;

__global__ __launch_bounds__(256) void gn_temporal_cached_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int B, int T, int hw,
                                                                 int C, int nsl, float eps, int silu) {
    __shared__ float s_sum[4][32], s_sq[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;     // ((b, pixel), slice)
    const bool active = wid < (int64_t)B * hw * nsl;
    const int64_t wp = active ? wid / nsl : 0;
    const int sl = active ? (int)(wid - wp * nsl) : 0;
    const int b = (int)(wp / hw);
    const int pix = (int)(wp - (int64_t)b * hw);
    const int cpg = C >> 5, slg = (C >> 3) / nsl;            // granules per slice (<= 64)
    const int gc = sl * slg + lane;                          // this lane's granule of the row
    const bool lane_on = active && lane < slg;
    if (lane < 32) {
        s_sum[wave][lane] = 0.f;
        s_sq[wave][lane] = 0.f;
    }
    __syncthreads();
    const size_t fstride = (size_t)hw * C;
    const bf16* xb = x + ((size_t)b * T * hw + pix) * C + gc * 8;
    bf16* yb = y + ((size_t)b * T * hw + pix) * C + gc * 8;
    bf16x8 cache[kGtCacheT];
    if (lane_on) {
#pragma unroll
        for (int t = 0; t < kGtCacheT; ++t)
            if (t < T) cache[t] = *(const bf16x8*)(xb + t * fstride);
        float sum[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
#pragma unroll
        for (int t = 0; t < kGtCacheT; ++t)
            if (t < T) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf2f(cache[t][e]);
                    sum[e] += f;
                    sq[e] += f * f;
                }
            }
        if (cpg >= 8) {           // a granule spans at most two groups: four LDS atomics per lane instead of sixteen
            const int g0 = (gc * 8) / cpg, split = (g0 + 1) * cpg - gc * 8;
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool first = e < split;
                s0 += first ? sum[e] : 0.f;
                q0 += first ? sq[e] : 0.f;
                s1 += first ? 0.f : sum[e];
                q1 += first ? 0.f : sq[e];
            }
            atomicAdd(&s_sum[wave][g0], s0);
            atomicAdd(&s_sq[wave][g0], q0);
            if (split < 8) {
                atomicAdd(&s_sum[wave][g0 + 1], s1);
                atomicAdd(&s_sq[wave][g0 + 1], q1);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int g = (gc * 8 + e) / cpg;
                atomicAdd(&s_sum[wave][g], sum[e]);
                atomicAdd(&s_sq[wave][g], sq[e]);
            }
        }
    }
    __syncthreads();
    if (!lane_on) return;
    const float inv_n = 1.0f / ((float)cpg * (float)T);
    float a[8], bb[8];
    {
        const f32x4 ga0 = *(const f32x4*)(gamma + gc * 8), ga1 = *(const f32x4*)(gamma + gc * 8 + 4);
        const f32x4 be0 = *(const f32x4*)(beta + gc * 8), be1 = *(const f32x4*)(beta + gc * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = (gc * 8 + e) / cpg;
            const float mean = s_sum[wave][g] * inv_n;
            const float var = fmaxf(s_sq[wave][g] * inv_n - mean * mean, 0.f);
            a[e] = rsqrtf(var + eps) * (e < 4 ? ga0[e & 3] : ga1[e & 3]);
            bb[e] = (e < 4 ? be0[e & 3] : be1[e & 3]) - mean * a[e];
        }
    }
#pragma unroll
    for (int t = 0; t < kGtCacheT; ++t)
        if (t < T) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(cache[t][e]) * a[e] + bb[e];
                if (silu) f = silu_f(f);
                o[e] = f2bf(f);
            }
            *(bf16x8*)(yb + t * fstride) = o;
        }
}

// The same single sweep with every lane busy: a THREAD owns one 16-byte granule of one pixel (all T rows of it in registers),
// consecutive threads walk the granules of a pixel and then the next pixel — [pixels][C] is contiguous, so a wave's load is one
// 1 KB burst whatever C is (the wave-per-(pixel, slice) mapping above runs 40 of 64 lanes at C = 320 / 640 / 1280: 2.7 - 3.9 TB/s).
// 320 threads = 8 / 4 / 2 pixels of 320 / 640 / 1280 channels.  Statistics: every thread writes the sums of the (at most two) groups
// its granule touches into its OWN LDS slot; one thread per (pixel, group) adds the two or three granules of that group in
// ascending order — no atomics, same bits every run.  C % 320 == 0, C / 32 >= 8.
constexpr int kGtFlatThreads = 320;
template <int CT>            // rows cached per thread (>= T): 17 for the 17-keyframe clips — 12 registers fewer than the general 20
__global__ __launch_bounds__(kGtFlatThreads) void gn_temporal_flat_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                          int64_t npix, int T, int hw, int C, float eps, int silu) {
    __shared__ f32x4 s_part[kGtFlatThreads];                 // per granule: (sum, sq) of its first group part, (sum, sq) of its second
    __shared__ f32x2 s_stat[8 * 32];                         // per (pixel of the block, group): mean, rstd
    const int gpp = C >> 3, cpg = C >> 5;
    const int ppb = kGtFlatThreads / gpp;                    // pixels per block
    const int pl = threadIdx.x / gpp, gc = threadIdx.x - pl * gpp;
    const int64_t gp = (int64_t)blockIdx.x * ppb + pl;       // (clip, pixel) index
    const bool on = gp < npix;
    const int64_t b = on ? gp / hw : 0;
    const int64_t pix = on ? gp - b * hw : 0;
    const size_t fstride = (size_t)hw * C;
    const bf16* xb = x + ((size_t)b * T * hw + pix) * C + gc * 8;
    bf16* yb = y + ((size_t)b * T * hw + pix) * C + gc * 8;
    bf16x8 cache[CT];
    const int g0 = (gc * 8) / cpg, split = (g0 + 1) * cpg - gc * 8;       // channels [0, split) of the granule belong to group g0
    if (on) {
#pragma unroll
        for (int t = 0; t < CT; ++t)
            if (t < T) cache[t] = *(const bf16x8*)(xb + t * fstride);
        float sum[8], sq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t)
            if (t < T) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = bf2f(cache[t][e]);
                    sum[e] += f;
                    sq[e] += f * f;
                }
            }
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool first = e < split;
            s0 += first ? sum[e] : 0.f;
            q0 += first ? sq[e] : 0.f;
            s1 += first ? 0.f : sum[e];
            q1 += first ? 0.f : sq[e];
        }
        s_part[threadIdx.x] = f32x4{s0, q0, s1, q1};
    }
    __syncthreads();
    if (threadIdx.x < ppb * 32) {
        const int p = threadIdx.x >> 5, g = threadIdx.x & 31;
        const int gr0 = (g * cpg) >> 3, gr1 = ((g + 1) * cpg - 1) >> 3;    // granules that hold channels of group g
        float ts = 0.f, tq = 0.f;
        for (int gr = gr0; gr <= gr1; ++gr) {
            const f32x4 pt = s_part[p * gpp + gr];
            const bool first = (gr * 8) / cpg == g;                         // is g the granule's first group?
            ts += first ? pt[0] : pt[2];
            tq += first ? pt[1] : pt[3];
        }
        const float inv_n = 1.0f / ((float)cpg * (float)T);
        const float mean = ts * inv_n;
        s_stat[p * 32 + g] = f32x2{mean, rsqrtf(fmaxf(tq * inv_n - mean * mean, 0.f) + eps)};
    }
    __syncthreads();
    if (!on) return;
    float a[8], bb[8];
    {
        const f32x4 ga0 = *(const f32x4*)(gamma + gc * 8), ga1 = *(const f32x4*)(gamma + gc * 8 + 4);
        const f32x4 be0 = *(const f32x4*)(beta + gc * 8), be1 = *(const f32x4*)(beta + gc * 8 + 4);
        const f32x2 st0 = s_stat[pl * 32 + g0], st1 = s_stat[pl * 32 + min(g0 + 1, 31)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool first = e < split;
            a[e] = (first ? st0[1] : st1[1]) * (e < 4 ? ga0[e & 3] : ga1[e & 3]);
            bb[e] = (e < 4 ? be0[e & 3] : be1[e & 3]) - (first ? st0[0] : st1[0]) * a[e];
        }
    }
#pragma unroll
    for (int t = 0; t < CT; ++t)
        if (t < T) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = bf2f(cache[t][e]) * a[e] + bb[e];
                if (silu) f = silu_f(f);
                o[e] = f2bf(f);
            }
            *(bf16x8*)(yb + t * fstride) = o;
        }
}

// Two-phase form for frame-sharded clips: local partial statistics, (all-reduce by the caller), apply.
__global__ __launch_bounds__(256) void gn_temporal_stats_kernel(const bf16* __restrict__ x, float* __restrict__ stats, int B,
                                                                int T, int hw, int C) {
    __shared__ float s_sum[4][32], s_sq[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wp = (int64_t)blockIdx.x * 4 + wave;
    const bool active = wp < (int64_t)B * hw;
    const int b = active ? (int)(wp / hw) : 0;
    const int pix = active ? (int)(wp - (int64_t)b * hw) : 0;
    const int G8 = C >> 3, cpg = C >> 5;
    if (lane < 32) {
        s_sum[wave][lane] = 0.f;
        s_sq[wave][lane] = 0.f;
    }
    __syncthreads();
    if (active) {
        const size_t fstride = (size_t)hw * C;
        const bf16* xb = x + ((size_t)b * T * hw + pix) * C;
        float sum[kGtCols][8], sq[kGtCols][8];
#pragma unroll
        for (int k = 0; k < kGtCols; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum[k][e] = sq[k][e] = 0.f;
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int k = 0; k < kGtCols; ++k) {
                const int gc = lane + 64 * k;
                if (gc < G8) {
                    const bf16x8 v = *(const bf16x8*)(xb + t * fstride + gc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = bf2f(v[e]);
                        sum[k][e] += f;
                        sq[k][e] += f * f;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < kGtCols; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int g = (gc * 8 + e) / cpg;
                    atomicAdd(&s_sum[wave][g], sum[k][e]);
                    atomicAdd(&s_sq[wave][g], sq[k][e]);
                }
            }
        }
    }
    __syncthreads();
    if (active && lane < 32) {
        stats[(wp * 32 + lane) * 2 + 0] = s_sum[wave][lane];
        stats[(wp * 32 + lane) * 2 + 1] = s_sq[wave][lane];
    }
}

__global__ __launch_bounds__(256) void gn_temporal_apply_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ stats, int B, int T, int hw, int C,
                                                                float count, float eps, int silu, int dst_frames, int dst_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wp = (int64_t)blockIdx.x * 4 + wave;
    if (wp >= (int64_t)B * hw) return;
    const int b = (int)(wp / hw);
    const int pix = (int)(wp - (int64_t)b * hw);
    const int G8 = C >> 3, cpg = C >> 5;
    const float inv_n = 1.0f / count;
    float a[kGtCols][8], bb[kGtCols][8];
#pragma unroll
    for (int k = 0; k < kGtCols; ++k) {
        const int gc = lane + 64 * k;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[k][e] = 0.f;
            bb[k][e] = 0.f;
            if (gc < G8) {
                const int c = gc * 8 + e;
                const int g = c / cpg;
                const float mean = stats[(wp * 32 + g) * 2] * inv_n;
                const float var = fmaxf(stats[(wp * 32 + g) * 2 + 1] * inv_n - mean * mean, 0.f);
                const float rstd = rsqrtf(var + eps);
                a[k][e] = rstd * gamma[c];
                bb[k][e] = beta[c] - mean * a[k][e];
            }
        }
    }
    const size_t fstride = (size_t)hw * C;
    const bf16* xb = x + ((size_t)b * T * hw + pix) * C;
    bf16* yb = y + (((size_t)b * dst_frames + dst_off) * hw + pix) * C;
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int k = 0; k < kGtCols; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                const bf16x8 v = *(const bf16x8*)(xb + t * fstride + gc * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = bf2f(v[e]) * a[k][e] + bb[k][e];
                    if (silu) f = silu_f(f);
                    o[e] = f2bf(f);
                }
                *(bf16x8*)(yb + t * fstride + gc * 8) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm over C: one wave per row, two-pass in registers
// ------------------------------------------------------------------------------------------
constexpr int kLnCols = 3;     // C <= 1536

// One wave normalises kLnRowsPerWave consecutive rows, two at a time (two independent 16-byte-per-lane row loads in
// flight, the two reductions interleaved); gamma / beta live in registers for all of them.
constexpr int kLnRowsPerWave = 8;   // upper bound; the launch picks 8 / 4 / 2 / 1 so that small inputs still fill the chip

template <int COLS>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t rows, int C, float eps, int rpw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * rpw;
    if (row0 >= rows) return;
    const int G8 = C >> 3;
    const float inv_c = 1.0f / (float)C;
    float g[COLS][8], bta[COLS][8];
#pragma unroll
    for (int k = 0; k < COLS; ++k) {
        const int gc = lane + 64 * k;
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
        if (gc < G8) {
            g0 = *(const f32x4*)(gamma + gc * 8);
            g1 = *(const f32x4*)(gamma + gc * 8 + 4);
            b0 = *(const f32x4*)(beta + gc * 8);
            b1 = *(const f32x4*)(beta + gc * 8 + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[k][e] = g0[e];
            g[k][4 + e] = g1[e];
            bta[k][e] = b0[e];
            bta[k][4 + e] = b1[e];
        }
    }
    const int nr = (int)min((int64_t)rpw, rows - row0);
    for (int r = 0; r < nr; r += 2) {
        const bool two = r + 1 < nr;
        const bf16* x0 = x + (size_t)(row0 + r) * C;
        const bf16* x1 = x0 + (two ? C : 0);
        bf16x8 t0[COLS], t1[COLS];
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                t0[k] = *(const bf16x8*)(x0 + gc * 8);
                t1[k] = *(const bf16x8*)(x1 + gc * 8);
            }
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < COLS; ++k)
            if (lane + 64 * k < G8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s0 += bf2f(t0[k][e]);
                    s1 += bf2f(t1[k][e]);
                }
            }
        const float m0 = wave_sum(s0) * inv_c, m1 = wave_sum(s1) * inv_c;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k < COLS; ++k)
            if (lane + 64 * k < G8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d0 = bf2f(t0[k][e]) - m0, d1 = bf2f(t1[k][e]) - m1;
                    q0 += d0 * d0;
                    q1 += d1 * d1;
                }
            }
        const float r0 = rsqrtf(wave_sum(q0) * inv_c + eps), r1 = rsqrtf(wave_sum(q1) * inv_c + eps);
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                bf16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o0[e] = f2bf((bf2f(t0[k][e]) - m0) * r0 * g[k][e] + bta[k][e]);
                    o1[e] = f2bf((bf2f(t1[k][e]) - m1) * r1 * g[k][e] + bta[k][e]);
                }
                *(bf16x8*)(y + (size_t)(row0 + r) * C + gc * 8) = o0;
                if (two) *(bf16x8*)(y + (size_t)(row0 + r + 1) * C + gc * 8) = o1;
            }
        }
    }
}

// The apply pass with every lane busy and several rows in flight per lane (the cat_add_gn recipe).  A THREAD owns one 16-byte granule
// column (its 8 channels' coefficients in 16 registers, set up once) and walks the pixel rows RS apart, FOUR rows requested before the
// first is used; gt * RS threads = C / 8 columns x RS row lanes (320 threads at C = 320 / 640 / 1280 / 2560).  The wave-per-row kernel
// above runs 40 of 64 lanes at C = 320 with one row in flight per wave: 20 KB outstanding per CU where the HBM latency x bandwidth
// product asks for ~60 KB (3.9 - 4.0 TB/s; LayerNorm, which keeps two rows in flight, reaches 5 - 5.9).
__global__ __launch_bounds__(320) void gn_spatial_apply_flat_kernel(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                                    const double* __restrict__ stats, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int hw, int C, float eps, int silu,
                                                                    int rows_per_block, int RS) {
    const int frame = blockIdx.y;
    const int gt = C >> 3, cpg = C >> 5;
    const int rs = threadIdx.x / gt, gc = threadIdx.x - rs * gt;
    if (rs >= RS) return;
    const float inv_n = 1.0f / ((float)cpg * (float)hw);
    float a[8], b[8];
    {
        const int c0 = gc * 8;
        const int g0 = c0 / cpg, g1 = min(g0 + 1, 31);
        const int split = (g0 + 1) * cpg - c0;               // channels of this granule that belong to group g0 (cpg >= 8: at most two groups)
        const double* sp0 = stats + (frame * 32 + g0) * 2;
        const double* sp1 = stats + (frame * 32 + g1) * 2;
        const float s00 = (float)sp0[0], s01 = (float)sp0[1], s10 = (float)sp1[0], s11 = (float)sp1[1];
        const f32x4 ga0 = *(const f32x4*)(gamma + c0), ga1 = *(const f32x4*)(gamma + c0 + 4);
        const f32x4 be0 = *(const f32x4*)(beta + c0), be1 = *(const f32x4*)(beta + c0 + 4);
        const float m0 = s00 * inv_n, m1 = s10 * inv_n;
        const float r0 = rsqrtf(fmaxf(s01 * inv_n - m0 * m0, 0.f) + eps), r1 = rsqrtf(fmaxf(s11 * inv_n - m1 * m1, 0.f) + eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ga = e < 4 ? ga0[e & 3] : ga1[e & 3], be = e < 4 ? be0[e & 3] : be1[e & 3];
            const bool first = e < split;
            a[e] = (first ? r0 : r1) * ga;
            b[e] = be - (first ? m0 : m1) * a[e];
        }
    }
    const int p0 = blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, hw);
    const bf16* xf = x + (size_t)frame * hw * C + gc * 8;
    bf16* yf = y + (size_t)frame * hw * C + gc * 8;
    auto emit = [&](int pix, bf16x8 v) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = fmaf(bf2f(v[e]), a[e], b[e]);
            if (silu) f = silu_f(f);
            o[e] = f2bf(f);
        }
        *(bf16x8*)(yf + (size_t)pix * C) = o;
    };
    int pix = p0 + rs;
    for (; pix + 3 * RS < p1; pix += 4 * RS) {
        bf16x8 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *(const bf16x8*)(xf + (size_t)(pix + k * RS) * C);
#pragma unroll
        for (int k = 0; k < 4; ++k) emit(pix + k * RS, v[k]);
    }
    for (; pix < p1; pix += RS) emit(pix, *(const bf16x8*)(xf + (size_t)pix * C));
}

void launch_gn_apply(dim3 grid, hipStream_t s, const bf16* x, bf16* y, const double* stats, const float* gamma,
                     const float* beta, int hw, int C, float eps, int silu, int apb) {
    const int gt = C >> 3;
    if (cc_policy().gn_apply_flat && (C >> 5) >= 8 && gt <= 320 && (int64_t)hw * grid.y >= 4096) {
        int RS = 320 / gt;
        RS = RS < 1 ? 1 : (RS > 8 ? 8 : RS);
        int rpb = 256;                                       // rows per workgroup: as many as still leave ~2000 workgroups
        while (rpb > 8 * RS && (int64_t)((hw + rpb - 1) / rpb) * grid.y < 2048) rpb >>= 1;
        hipLaunchKernelGGL(gn_spatial_apply_flat_kernel, dim3((hw + rpb - 1) / rpb, grid.y), dim3(gt * RS), 0, s, x, y, stats, gamma, beta,
                           hw, C, eps, silu, rpb, RS);
        return;
    }
    const int cols = (C / 8 + 63) / 64;
#define CC_GA(N) hipLaunchKernelGGL(gn_spatial_apply_kernel<N>, grid, dim3(256), 0, s, x, y, stats, gamma, beta, hw, C, eps, silu, apb)
    if (cols == 1) CC_GA(1); else if (cols == 2) CC_GA(2); else if (cols == 3) CC_GA(3); else if (cols == 4) CC_GA(4); else CC_GA(kMaxCols);
#undef CC_GA
}

__global__ void zero_f64_kernel(double* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

}  // namespace

extern "C" int ccedit_groupnorm_spatial(const void* x, void* y, const float* gamma, const float* beta, double* stats_ws,
                                        int32_t frames, int32_t hw, int32_t C, float eps, int32_t silu, void* stream) {
    CC_CHECK_ARG(x && y && gamma && beta && stats_ws, "ccedit_groupnorm_spatial: null pointer");
    CC_CHECK_ARG(frames > 0 && hw > 0 && C > 0, "ccedit_groupnorm_spatial: bad sizes");
    CC_UNSUPPORTED(C % 32 != 0 || C > kMaxCols * 512, "ccedit_groupnorm_spatial: C=%d (need C%%32==0, C<=%d)", C,
                   kMaxCols * 512);
    hipStream_t s = (hipStream_t)stream;
    if (C % 256 == 0 && (int64_t)hw * (C / 256) <= 256 * kGnOneUnits) {       // small frames: one pass, one launch
        hipLaunchKernelGGL(gn_spatial_onepass_kernel, dim3(32, frames), dim3(256), 0, s, (const bf16*)x, (bf16*)y, gamma, beta, hw, C, eps, silu);
        return cc_launch_status("groupnorm_spatial (one pass)");
    }
    // (a kernel, not hipMemsetAsync: the runtime's fill of these 17 KB takes ~23 us of device time per call, 0.5 ms per step)
    hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)((64 * frames + 255) / 256)), dim3(256), 0, s, stats_ws, 64 * frames);
    const int apb = gn_pix_per_block(hw, frames, 4, 2048);
    dim3 grid((hw + apb - 1) / apb, frames);
    // the per-block reduction tail (LDS + global atomics) costs about as much as reading 16 pixel rows per wave:
    // give a stats block 256 pixels when the frame is large enough to still fill the chip
    int spb = 256;
    while (spb > 16 && (int64_t)((hw + spb - 1) / spb) * frames < 512) spb >>= 1;
    dim3 sgrid((hw + spb - 1) / spb, frames);
    hipLaunchKernelGGL(gn_spatial_stats_kernel, sgrid, dim3(256), 0, s, (const bf16*)x, stats_ws, hw, C, spb);
    launch_gn_apply(grid, s, (const bf16*)x, (bf16*)y, stats_ws, gamma, beta, hw, C, eps, silu, apb);
    return cc_launch_status("groupnorm_spatial");
}

extern "C" int ccedit_groupnorm_spatial_stats(const void* x, double* stats, int32_t frames, int32_t hw, int32_t C, void* stream) {
    CC_CHECK_ARG(x && stats && frames > 0 && hw > 0 && C > 0, "ccedit_groupnorm_spatial_stats: bad args");
    CC_UNSUPPORTED(C % 32 != 0 || C > kMaxCols * 512, "ccedit_groupnorm_spatial_stats: C=%d (need C%%32==0, C<=%d)", C, kMaxCols * 512);
    int spb = 256;
    while (spb > 16 && (int64_t)((hw + spb - 1) / spb) * frames < 512) spb >>= 1;
    hipLaunchKernelGGL(gn_spatial_stats_kernel, dim3((hw + spb - 1) / spb, frames), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, stats, hw, C, spb);
    return cc_launch_status("groupnorm_spatial_stats");
}

extern "C" int ccedit_groupnorm_spatial_apply(const void* x, void* y, const float* gamma, const float* beta,
                                              const double* stats, int32_t frames, int32_t hw, int32_t C, float eps,
                                              int32_t silu, void* stream) {
    CC_CHECK_ARG(x && y && gamma && beta && stats, "ccedit_groupnorm_spatial_apply: null pointer");
    CC_CHECK_ARG(frames > 0 && hw > 0 && C > 0, "ccedit_groupnorm_spatial_apply: bad sizes");
    CC_UNSUPPORTED(C % 32 != 0 || C > kMaxCols * 512, "ccedit_groupnorm_spatial_apply: C=%d (need C%%32==0, C<=%d)", C,
                   kMaxCols * 512);
    const int apb = gn_pix_per_block(hw, frames, 4, 2048);
    dim3 grid((hw + apb - 1) / apb, frames);
    launch_gn_apply(grid, (hipStream_t)stream, (const bf16*)x, (bf16*)y, stats, gamma, beta, hw, C, eps, silu, apb);
    return cc_launch_status("groupnorm_spatial_apply");
}

extern "C" int ccedit_groupnorm_temporal(const void* x, void* y, const float* gamma, const float* beta, int32_t B,
                                         int32_t T, int32_t hw, int32_t C, float eps, int32_t silu, void* stream) {
    CC_CHECK_ARG(x && y && gamma && beta, "ccedit_groupnorm_temporal: null pointer");
    CC_CHECK_ARG(B > 0 && T > 0 && hw > 0 && C > 0, "ccedit_groupnorm_temporal: bad sizes");
    CC_UNSUPPORTED(C % 32 != 0 || C > kGtCols * 512, "ccedit_groupnorm_temporal: C=%d (need C%%32==0, C<=%d)", C,
                   kGtCols * 512);
    const int64_t waves = (int64_t)B * hw;
    dim3 grid((unsigned)((waves + 3) / 4));
    int nsl = 1;                                   // channel slices: whole groups, at most 512 channels each
    while (nsl < 32 && C / nsl > 512) nsl <<= 1;
    const int flat_env = cc_policy().gn_flat;      // 0: A/B against the wave-per-slice mapping
    // (measured, 2 x 17 frames: 70 / 38 us against 73 / 40 at 64x96 / 32x48; the small levels — a few hundred workgroups — are
    //  1 - 2 us better with the wave-per-slice mapping's larger grid)
    if (flat_env && T <= kGtCacheT && C % 320 == 0 && C <= 1280 && waves * (C >> 3) >= 700 * kGtFlatThreads) {
        const int ppb = kGtFlatThreads / (C >> 3);
        if (T <= 17)
            hipLaunchKernelGGL(gn_temporal_flat_kernel<17>, dim3((unsigned)((waves + ppb - 1) / ppb)), dim3(kGtFlatThreads), 0, (hipStream_t)stream,
                               (const bf16*)x, (bf16*)y, gamma, beta, waves, T, hw, C, eps, silu);
        else
            hipLaunchKernelGGL(gn_temporal_flat_kernel<kGtCacheT>, dim3((unsigned)((waves + ppb - 1) / ppb)), dim3(kGtFlatThreads), 0, (hipStream_t)stream,
                               (const bf16*)x, (bf16*)y, gamma, beta, waves, T, hw, C, eps, silu);
    } else if (T <= kGtCacheT && (C >> 3) % nsl == 0) {
        const int64_t nw = waves * nsl;
        hipLaunchKernelGGL(gn_temporal_cached_kernel, dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16*)x, (bf16*)y, gamma, beta, B, T, hw, C, nsl, eps, silu);
    } else {
        hipLaunchKernelGGL(gn_temporal_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, gamma,
                           beta, B, T, hw, C, eps, silu);
    }
    return cc_launch_status("groupnorm_temporal");
}

extern "C" int ccedit_groupnorm_temporal_stats(const void* x, float* stats, int32_t B, int32_t T, int32_t hw, int32_t C,
                                               void* stream) {
    CC_CHECK_ARG(x && stats && B > 0 && T > 0 && hw > 0 && C > 0, "ccedit_groupnorm_temporal_stats: bad args");
    CC_UNSUPPORTED(C % 32 != 0 || C > kGtCols * 512, "ccedit_groupnorm_temporal_stats: C=%d", C);
    const int64_t waves = (int64_t)B * hw;
    hipLaunchKernelGGL(gn_temporal_stats_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, stats, B, T, hw, C);
    return cc_launch_status("groupnorm_temporal_stats");
}

extern "C" int ccedit_groupnorm_temporal_apply(const void* x, void* y, const float* gamma, const float* beta,
                                               const float* stats, int32_t B, int32_t T, int32_t hw, int32_t C, float count,
                                               float eps, int32_t silu, int32_t dst_frames, int32_t dst_off, void* stream) {
    CC_CHECK_ARG(x && y && gamma && beta && stats && B > 0 && T > 0 && hw > 0 && C > 0 && count > 0,
                 "ccedit_groupnorm_temporal_apply: bad args");
    CC_CHECK_ARG(dst_frames >= T + dst_off && dst_off >= 0, "ccedit_groupnorm_temporal_apply: bad destination geometry");
    CC_UNSUPPORTED(C % 32 != 0 || C > kGtCols * 512, "ccedit_groupnorm_temporal_apply: C=%d", C);
    const int64_t waves = (int64_t)B * hw;
    hipLaunchKernelGGL(gn_temporal_apply_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, (bf16*)y, gamma, beta, stats, B, T, hw, C, count, eps, silu, dst_frames, dst_off);
    return cc_launch_status("groupnorm_temporal_apply");
}

// The statistics half of layernorm_kernel: (mean, rstd) per row, same summation order (a consumer that folds the LayerNorm into
// its GEMM — CcGemmDesc.ln_stats — sees the numbers the normalising kernel would have used).
template <int COLS>
__global__ __launch_bounds__(256) void row_stats_kernel(const bf16* __restrict__ x, float* __restrict__ st, int64_t rows, int C, float eps, int rpw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * rpw;
    if (row0 >= rows) return;
    const int G8 = C >> 3;
    const float inv_c = 1.0f / (float)C;
    const int nr = (int)min((int64_t)rpw, rows - row0);
    for (int r = 0; r < nr; r += 2) {
        const bool two = r + 1 < nr;
        const bf16* x0 = x + (size_t)(row0 + r) * C;
        const bf16* x1 = x0 + (two ? C : 0);
        bf16x8 t0[COLS], t1[COLS];
#pragma unroll
        for (int k = 0; k < COLS; ++k) {
            const int gc = lane + 64 * k;
            if (gc < G8) {
                t0[k] = *(const bf16x8*)(x0 + gc * 8);
                t1[k] = *(const bf16x8*)(x1 + gc * 8);
            }
        }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int k = 0; k < COLS; ++k)
            if (lane + 64 * k < G8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s0 += bf2f(t0[k][e]);
                    s1 += bf2f(t1[k][e]);
                }
            }
        const float m0 = wave_sum(s0) * inv_c, m1 = wave_sum(s1) * inv_c;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k < COLS; ++k)
            if (lane + 64 * k < G8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d0 = bf2f(t0[k][e]) - m0, d1 = bf2f(t1[k][e]) - m1;
                    q0 += d0 * d0;
                    q1 += d1 * d1;
                }
            }
        const float r0 = rsqrtf(wave_sum(q0) * inv_c + eps), r1 = rsqrtf(wave_sum(q1) * inv_c + eps);
        if (lane == 0) {
            *(f32x2*)(st + 2 * (row0 + r)) = f32x2{m0, r0};
            if (two) *(f32x2*)(st + 2 * (row0 + r + 1)) = f32x2{m1, r1};
        }
    }
}

extern "C" int ccedit_row_stats(const void* x, float* stats, int64_t rows, int32_t C, float eps, void* stream) {
    CC_CHECK_ARG(x && stats, "ccedit_row_stats: null pointer");
    CC_CHECK_ARG(rows > 0 && C > 0, "ccedit_row_stats: bad sizes");
    CC_UNSUPPORTED(C % 8 != 0 || C > kLnCols * 512, "ccedit_row_stats: C=%d (need C%%8==0, C<=%d)", C, kLnCols * 512);
    int rpw = kLnRowsPerWave;
    while (rpw > 1 && (rows + 4 * rpw - 1) / (4 * rpw) < 4096) rpw >>= 1;
    dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
    const int cols = (C / 8 + 63) / 64;
#define CC_RS(N) hipLaunchKernelGGL(row_stats_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, stats, rows, C, eps, rpw)
    if (cols == 1) CC_RS(1); else if (cols == 2) CC_RS(2); else CC_RS(kLnCols);
#undef CC_RS
    return cc_launch_status("row_stats");
}

extern "C" int ccedit_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows, int32_t C,
                                float eps, void* stream) {
    CC_CHECK_ARG(x && y && gamma && beta, "ccedit_layernorm: null pointer");
    CC_CHECK_ARG(rows > 0 && C > 0, "ccedit_layernorm: bad sizes");
    CC_UNSUPPORTED(C % 8 != 0 || C > kLnCols * 512, "ccedit_layernorm: C=%d (need C%%8==0, C<=%d)", C, kLnCols * 512);
    int rpw = kLnRowsPerWave;
    while (rpw > 1 && (rows + 4 * rpw - 1) / (4 * rpw) < 4096) rpw >>= 1;
    dim3 grid((unsigned)((rows + 4 * rpw - 1) / (4 * rpw)));
    const int cols = (C / 8 + 63) / 64;
#define CC_LN(N) hipLaunchKernelGGL(layernorm_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, gamma, beta, rows, C, eps, rpw)
    if (cols == 1) CC_LN(1); else if (cols == 2) CC_LN(2); else CC_LN(kLnCols);
#undef CC_LN
    return cc_launch_status("layernorm");
}
