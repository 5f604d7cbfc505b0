// Layout conversions and small elementwise kernels (all HBM-bound or negligible).
#include "common.h"

namespace {

// fp32 (B, C, T, H, W) -> bf16 [B*T][H*W][Cpad]; y = x * scale(b) + shift, pad channels = 0.
// Reads are coalesced along W for each channel; each thread assembles the Cpad channels of one pixel
// (Cpad <= 8 on this path: latent C=4, hint C=3) and writes one 16-byte granule.
__global__ void ncthw_to_nhwc_kernel(const float* __restrict__ x, bf16* __restrict__ y, int B, int C, int T, int H, int W,
                                     int Cpad, const float* __restrict__ scale_per_b, float scale, float shift) {
    const int64_t hw = (int64_t)H * W;
    const int64_t total = (int64_t)B * T * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t frame = i / hw;
        const int64_t pix = i - frame * hw;
        const int b = (int)(frame / T), t = (int)(frame - (int64_t)b * T);
        const float sc = scale_per_b ? scale_per_b[b] * scale : scale;
        bf16* out = y + i * Cpad;
        for (int c = 0; c < Cpad; ++c) {
            float v = 0.f;
            if (c < C) v = x[(((int64_t)b * C + c) * T + t) * hw + pix] * sc + shift;
            out[c] = f2bf(v);
        }
    }
}

template <bool F32>
__global__ void nhwc_to_ncthw_kernel(const void* __restrict__ x, int ld, float* __restrict__ y, int B, int C, int T, int H,
                                     int W) {
    const int64_t hw = (int64_t)H * W;
    const int64_t total = (int64_t)B * C * T * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % hw;
        int64_t r = i / hw;
        const int t = (int)(r % T);
        r /= T;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const int64_t row = ((int64_t)b * T + t) * hw + pix;
        float v;
        if (F32) v = ((const float*)x)[row * ld + c];
        else v = bf2f(((const bf16*)x)[row * ld + c]);
        y[i] = v;
    }
}

// out[:, :C1] = a; out[:, C1:] = b + c   (16-byte granules)
__global__ void cat_add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, const bf16* __restrict__ c,
                               bf16* __restrict__ out, int64_t rows, int C1, int C2) {
    const int g1 = C1 >> 3, g2 = C2 >> 3, gt = g1 + g2;
    const int64_t total = rows * gt;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / gt;
        const int g = (int)(i - row * gt);
        bf16x8 v;
        if (g < g1) {
            v = *(const bf16x8*)(a + row * C1 + g * 8);
        } else {
            const bf16x8 u = *(const bf16x8*)(b + row * C2 + (g - g1) * 8);
            if (c) {
                const bf16x8 w = *(const bf16x8*)(c + row * C2 + (g - g1) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(u[e]) + bf2f(w[e]));
            } else {
                v = u;
            }
        }
        *(bf16x8*)(out + i * 8) = v;
    }
}

// The same concatenation, also accumulating the GroupNorm(32, C1+C2) statistics of the tensor it writes (the decoder's in_layers.0
// reads them instead of re-reading the tensor).  A thread owns ONE 16-byte granule column of the output row (its source — a, or
// b + c — is fixed, so there is no divergence inside a wave except at the seam) and walks rows RS apart, four rows in flight:
// 16 statistics registers per thread instead of 80, eight waves per SIMD, every lane busy at every width (round 3's kernel gave a
// wave one row and left 40 of 64 lanes idle at 640 channels: 0.6 - 3.4 TB/s).  Per-channel sums meet in LDS slots that each have ONE
// writer, are added per group in a fixed order, and only the cross-workgroup sum uses (double) atomics: same bits every run.
constexpr int kCatMaxC = 2560;
__global__ __launch_bounds__(320) void cat_add_gn_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                         const bf16* __restrict__ c, bf16* __restrict__ out,
                                                         double* __restrict__ stats, int hw, int C1, int C2,
                                                         int rows_per_block, int RS) {
    extern __shared__ __attribute__((aligned(16))) char cat_smem[];
    float* const csum = (float*)cat_smem;                 // [RS][C] sums, then [RS][C] sums of squares
    const int C = C1 + C2, gt = C >> 3, g1 = C1 >> 3, cpg = C >> 5;
    float* const csq = csum + RS * C;
    const int frame = blockIdx.y;
    const int rs = threadIdx.x / gt, g = threadIdx.x - rs * gt;          // row lane, granule column
    const int p0 = blockIdx.x * rows_per_block, p1 = min(p0 + rows_per_block, hw);
    const bool from_a = g < g1;
    const bf16* src0 = from_a ? a + g * 8 : b + (g - g1) * 8;
    const bf16* src1 = (!from_a && c) ? c + (g - g1) * 8 : nullptr;
    const int ld = from_a ? C1 : C2;
    float sum[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) sum[e] = sq[e] = 0.f;
    auto emit = [&](int64_t row, bf16x8 u, bf16x8 w) {
        bf16x8 v = u;
        if (src1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(u[e]) + bf2f(w[e]));
        }
        *(bf16x8*)(out + row * C + g * 8) = v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = bf2f(v[e]);
            sum[e] += f;
            sq[e] += f * f;
        }
    };
    if (rs < RS) {
        int pix = p0 + rs;
        for (; pix + 3 * RS < p1; pix += 4 * RS) {              // four rows of this column in flight
            bf16x8 u[4], w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t row = (int64_t)frame * hw + pix + k * RS;
                u[k] = *(const bf16x8*)(src0 + row * ld);
                if (src1) w[k] = *(const bf16x8*)(src1 + row * ld);
                else w[k] = u[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) emit((int64_t)frame * hw + pix + k * RS, u[k], w[k]);
        }
        for (; pix < p1; pix += RS) {
            const int64_t row = (int64_t)frame * hw + pix;
            const bf16x8 u = *(const bf16x8*)(src0 + row * ld);
            emit(row, u, src1 ? *(const bf16x8*)(src1 + row * ld) : u);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            csum[rs * C + g * 8 + e] = sum[e];
            csq[rs * C + g * 8 + e] = sq[e];
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {                                      // thread (group, sum | sum of squares): fixed summation order
        const int grp = threadIdx.x >> 1;
        const float* src = (threadIdx.x & 1) ? csq : csum;
        double t = 0.0;
        for (int r = 0; r < RS; ++r) {
            float part = 0.f;
            for (int ch = grp * cpg; ch < (grp + 1) * cpg; ++ch) part += src[r * C + ch];
            t += (double)part;
        }
        unsafeAtomicAdd(&stats[frame * 64 + threadIdx.x], t);
    }
}

__global__ void add_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 u = *(const bf16x8*)(a + i * 8), w = *(const bf16x8*)(b + i * 8);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(u[e]) + bf2f(w[e]));
        *(bf16x8*)(y + i * 8) = v;
    }
}

__global__ void silu_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = f2bf(silu_f(bf2f(x[i])));
}

// [cos(t f_k) | sin(t f_k)], f_k = exp(-ln(10000) k / half)   (diffusionmodules/util.py:244-268)
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, bf16* __restrict__ out, int n, int dim, int ld) {
    const int half = dim / 2;
    const int i = blockIdx.x;
    for (int k = threadIdx.x; k < half; k += blockDim.x) {
        const float f = expf(-9.210340371976184f * (float)k / (float)half);
        const float a = (float)t[i] * f;
        out[(size_t)i * ld + k] = f2bf(cosf(a));
        out[(size_t)i * ld + half + k] = f2bf(sinf(a));
    }
}

// out[row] = tok[ids[row]] + pos[row % L], 8 channels per thread (fp32 tables -> bf16 activations)
__global__ void embedding_lookup_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                        const float* __restrict__ pos, bf16* __restrict__ out, int64_t rows, int L, int C,
                                        int vocab) {
    const int g8 = C >> 3;
    const int64_t total = rows * g8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / g8;
        const int g = (int)(i - row * g8);
        int64_t id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float* t = tok + id * C + g * 8;
        const float* p = pos + (row % L) * C + g * 8;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(t[e] + p[e]);
        *(bf16x8*)(out + row * C + g * 8) = o;
    }
}

// DiagonalGaussianDistribution.sample (distributions.py:24-41) on the channels-last moments of the VAE encoder:
// out[n][c][p] = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise[n][c][p]),  moments row = [mean(zc) | logvar(zc)]
__global__ void gaussian_sample_kernel(const float* __restrict__ mom, const float* __restrict__ noise,
                                       float* __restrict__ out, int64_t total, int zc, int hw, int ldm, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t nc = i / hw;                   // (frame, channel) of the NCHW output element
        const int p = (int)(i - nc * hw);
        const int64_t n = nc / zc;
        const int c = (int)(nc - n * zc);
        const float* row = mom + (n * hw + p) * ldm;
        const float logvar = fminf(fmaxf(row[zc + c], -30.0f), 20.0f);
        out[i] = scale * (row[c] + expf(0.5f * logvar) * noise[i]);
    }
}

// y = x * m + z * (1 - m): the known-region re-injection of sample_inpainting (sampling.py:150-153, 213-216)
__global__ void mask_blend_kernel(const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ m,
                                  float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i] * m[i] + z[i] * (1.0f - m[i]);
}

__global__ void cfg_denoise_kernel(const float* __restrict__ x, const float* __restrict__ eps2, float* __restrict__ den,
                                   int64_t n, float sigma, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        // DiscreteDenoiser: d = net * c_out + x * c_skip with c_out = -sigma, c_skip = 1, for each CFG half;
        // VanillaCFG: d_u + scale * (d_c - d_u).  Evaluated in the reference's order of operations.
        const float du = eps2[i] * (-sigma) + x[i];
        const float dc = eps2[n + i] * (-sigma) + x[i];
        den[i] = du + scale * (dc - du);
    }
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ y, int64_t n,
                             float a, float b) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a * x[i] + b * z[i];
}

// p[r][c] = softmax_c(s[r][c] * scale) for c < cols, 0 for cols <= c < cols_pad.  One block per row,
// the row lives in registers (cols <= 256*32).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16* __restrict__ p, int cols,
                                                           int cols_pad, int64_t lds_, int64_t ldp, float scale) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const float* sr = s + row * lds_;
    bf16* pr = p + row * ldp;
    float v[32];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = threadIdx.x + k * 256;
        v[k] = (c < cols) ? sr[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        v[k] = __expf(v[k] - mx);
        sum += v[k];
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int c = threadIdx.x + k * 256;
        if (c < cols_pad) pr[c] = f2bf(c < cols ? v[k] * inv : 0.f);
    }
}

// Row-block copy: block s moves `rows` whole rows (row_bytes each) from row src_row of `src` to row dst_row of `dst`, optionally
// adding the rows of `add` that sit where the DESTINATION rows sit (bf16: fp32 add, one rounding — x.add_(skip)).  The pack /
// unpack halves of the frame <-> pixel layout transposition of a frame-sharded clip (parallel.FrameShard): for every (peer rank,
// clip, keyframe) one block, 16 bytes per lane, rows walked contiguously.
__global__ __launch_bounds__(256) void copy_row_blocks_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                              const bf16* __restrict__ add, const int64_t* __restrict__ blocks,
                                                              int row_gran) {
    const int64_t src_row = blocks[3 * blockIdx.y], dst_row = blocks[3 * blockIdx.y + 1], rows = blocks[3 * blockIdx.y + 2];
    const int64_t total = rows * row_gran;                       // 16-byte granules of this block (rows are contiguous on both sides)
    const u32x4* s = (const u32x4*)(src + src_row * row_gran * 16);
    u32x4* o = (u32x4*)(dst + dst_row * row_gran * 16);
    const bf16x8* a = add ? (const bf16x8*)((const char*)add + dst_row * row_gran * 16) : nullptr;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (!a) {                                                     // plain copy: four granules requested before the first is stored
        for (; i + 3 * stride < total; i += 4 * stride) {
            const u32x4 v0 = s[i], v1 = s[i + stride], v2 = s[i + 2 * stride], v3 = s[i + 3 * stride];
            o[i] = v0;
            o[i + stride] = v1;
            o[i + 2 * stride] = v2;
            o[i + 3 * stride] = v3;
        }
    }
    for (; i < total; i += stride) {
        u32x4 v = s[i];
        if (a) {
            bf16x8 x = *(bf16x8*)&v;
            const bf16x8 y = a[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
            v = *(u32x4*)&x;
        }
        o[i] = v;
    }
}

// n_blocks strided 2-D copies of ONE geometry (rows x row_gran 16-byte granules, a row pitch on each side) that differ in their byte
// offsets: the column-slice <-> contiguous-buffer halves of the head-parallel attention exchange of a row-sharded clip
// (parallel.RowShard.to_heads / from_heads): block (peer rank, q | k | v) gathers that peer's head columns out of the [rows][3C]
// projection into its contiguous send buffer, and the way back scatters [rows][C / N] buffers into the columns of [rows][C].
__global__ __launch_bounds__(256) void copy_2d_blocks_kernel(const char* __restrict__ src, char* __restrict__ dst,
                                                             const int64_t* __restrict__ blocks, int64_t rows, int row_gran,
                                                             int64_t src_pitch, int64_t dst_pitch) {
    const char* s = src + blocks[2 * blockIdx.y];
    char* o = dst + blocks[2 * blockIdx.y + 1];
    const int64_t total = rows * row_gran;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / row_gran;
        const int g = (int)(i - r * row_gran);
        *(u32x4*)(o + r * dst_pitch + g * 16) = *(const u32x4*)(s + r * src_pitch + g * 16);
    }
}

inline unsigned grid_for(int64_t n, int threads) {
    int64_t g = (n + threads - 1) / threads;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int ccedit_ncthw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                                    int32_t Cpad, const float* scale_per_b, float scale, float shift, void* stream) {
    CC_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && Cpad >= C, "ccedit_ncthw_to_nhwc: bad args");
    const int64_t total = (int64_t)B * T * H * W;
    hipLaunchKernelGGL(ncthw_to_nhwc_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, B,
                       C, T, H, W, Cpad, scale_per_b, scale, shift);
    return cc_launch_status("ncthw_to_nhwc");
}

extern "C" int ccedit_nhwc_to_ncthw(const void* x, int32_t x_is_f32, int32_t ld, float* y, int32_t B, int32_t C, int32_t T,
                                    int32_t H, int32_t W, void* stream) {
    CC_CHECK_ARG(x && y && B > 0 && C > 0 && T > 0 && H > 0 && W > 0 && ld >= C, "ccedit_nhwc_to_ncthw: bad args");
    const int64_t total = (int64_t)B * C * T * H * W;
    if (x_is_f32)
        hipLaunchKernelGGL(nhwc_to_ncthw_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ld,
                           y, B, C, T, H, W);
    else
        hipLaunchKernelGGL(nhwc_to_ncthw_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                           ld, y, B, C, T, H, W);
    return cc_launch_status("nhwc_to_ncthw");
}

extern "C" int ccedit_copy_row_blocks(const void* src, void* dst, const void* add, const int64_t* blocks, int32_t n_blocks,
                                      int64_t max_rows, int32_t row_bytes, void* stream) {
    CC_CHECK_ARG(src && dst && blocks && n_blocks > 0 && max_rows > 0 && row_bytes > 0, "ccedit_copy_row_blocks: bad args");
    CC_UNSUPPORTED(row_bytes % 16 != 0 || n_blocks > 65535, "ccedit_copy_row_blocks: row_bytes=%d must be a multiple of 16 (n_blocks=%d <= 65535)",
                   row_bytes, n_blocks);
    const int row_gran = row_bytes / 16;
    int64_t gx = (max_rows * row_gran + 256 * 8 - 1) / (256 * 8);       // ~8 granules per thread in the longest block
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(copy_row_blocks_kernel, dim3((unsigned)gx, (unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, (const bf16*)add, blocks, row_gran);
    return cc_launch_status("copy_row_blocks");
}

extern "C" int ccedit_copy_2d_blocks(const void* src, void* dst, const int64_t* blocks, int32_t n_blocks, int64_t rows,
                                     int32_t row_bytes, int64_t src_pitch, int64_t dst_pitch, void* stream) {
    CC_CHECK_ARG(src && dst && blocks && n_blocks > 0 && rows > 0 && row_bytes > 0, "ccedit_copy_2d_blocks: bad args");
    CC_UNSUPPORTED(row_bytes % 16 || src_pitch % 16 || dst_pitch % 16 || src_pitch < row_bytes || dst_pitch < row_bytes,
                   "ccedit_copy_2d_blocks: row_bytes=%d, pitches %lld / %lld (multiples of 16, pitches >= row_bytes)", row_bytes,
                   (long long)src_pitch, (long long)dst_pitch);
    CC_UNSUPPORTED(n_blocks > 65535, "ccedit_copy_2d_blocks: more than 65535 blocks");
    const int row_gran = row_bytes / 16;
    int64_t gx = (rows * row_gran + 256 * 8 - 1) / (256 * 8);           // ~8 granules per thread
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    hipLaunchKernelGGL(copy_2d_blocks_kernel, dim3((unsigned)gx, (unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const char*)src, (char*)dst, blocks, rows, row_gran, src_pitch, dst_pitch);
    return cc_launch_status("copy_2d_blocks");
}

extern "C" int ccedit_cat_add(const void* a, const void* b, const void* c, void* out, int64_t rows, int32_t C1, int32_t C2,
                              void* stream) {
    CC_CHECK_ARG(a && b && out && rows > 0, "ccedit_cat_add: bad args");
    CC_UNSUPPORTED(C1 % 8 || C2 % 8, "ccedit_cat_add: C1=%d C2=%d must be multiples of 8", C1, C2);
    const int64_t total = rows * ((C1 + C2) / 8);
    hipLaunchKernelGGL(cat_add_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                       (const bf16*)b, (const bf16*)c, (bf16*)out, rows, C1, C2);
    return cc_launch_status("cat_add");
}

extern "C" int ccedit_cat_add_gn(const void* a, const void* b, const void* c, void* out, double* stats, int32_t frames,
                                 int32_t hw, int32_t C1, int32_t C2, void* stream) {
    CC_CHECK_ARG(a && b && out && stats && frames > 0 && hw > 0, "ccedit_cat_add_gn: bad args");
    CC_UNSUPPORTED(C1 % 8 || C2 % 8 || (C1 + C2) % 32 || C1 + C2 > kCatMaxC || C1 + C2 < 64,
                   "ccedit_cat_add_gn: C1=%d C2=%d (multiples of 8, sum a multiple of 32, 64 <= sum <= %d)", C1, C2, kCatMaxC);
    const int C = C1 + C2, gt = C / 8;
    int RS = 320 / gt;                       // rows a workgroup works on side by side
    RS = RS < 1 ? 1 : (RS > 4 ? 4 : RS);
    int ppb = 256;                           // rows per workgroup: as many as still leave ~2000 workgroups
    while (ppb > 8 * RS && (int64_t)((hw + ppb - 1) / ppb) * frames < 2048) ppb >>= 1;
    // at least 64 threads: the final per-(group, sum | sum of squares) reduction is done by threads 0..63 (C = 64 / 96 give only
    // 32 / 48 column threads; the body is guarded by rs < RS, the extra threads only take part in the reduction)
    const int threads = gt * RS < 64 ? 64 : gt * RS;
    hipLaunchKernelGGL(cat_add_gn_kernel, dim3((hw + ppb - 1) / ppb, frames), dim3(threads), 2 * RS * C * sizeof(float),
                       (hipStream_t)stream, (const bf16*)a, (const bf16*)b, (const bf16*)c, (bf16*)out, stats, hw, C1, C2, ppb, RS);
    return cc_launch_status("cat_add_gn");
}

extern "C" int ccedit_add(const void* a, const void* b, void* y, int64_t n, void* stream) {
    CC_CHECK_ARG(a && b && y && n > 0, "ccedit_add: bad args");
    CC_UNSUPPORTED(n % 8, "ccedit_add: n must be a multiple of 8");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                       (const bf16*)b, (bf16*)y, n / 8);
    return cc_launch_status("add");
}

extern "C" int ccedit_silu(const void* x, void* y, int64_t n, void* stream) {
    CC_CHECK_ARG(x && y && n > 0, "ccedit_silu: bad args");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, n);
    return cc_launch_status("silu");
}

extern "C" int ccedit_timestep_embedding(const int64_t* t, void* out, int32_t n, int32_t dim, int32_t ld, void* stream) {
    CC_CHECK_ARG(t && out && n > 0 && dim > 0 && ld >= dim, "ccedit_timestep_embedding: bad args");
    CC_UNSUPPORTED(dim % 2, "ccedit_timestep_embedding: odd dim");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, t, (bf16*)out, n, dim, ld);
    return cc_launch_status("timestep_embedding");
}

extern "C" int ccedit_embedding_lookup(const int64_t* ids, const float* tok, const float* pos, void* out, int64_t rows,
                                       int32_t L, int32_t C, int32_t vocab, void* stream) {
    CC_CHECK_ARG(ids && tok && pos && out && rows > 0 && L > 0 && vocab > 0, "ccedit_embedding_lookup: bad args");
    CC_UNSUPPORTED(C % 8 != 0, "ccedit_embedding_lookup: C=%d must be a multiple of 8", C);
    hipLaunchKernelGGL(embedding_lookup_kernel, dim3(grid_for(rows * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, ids,
                       tok, pos, (bf16*)out, rows, L, C, vocab);
    return cc_launch_status("embedding_lookup");
}

extern "C" int ccedit_gaussian_sample(const float* moments, const float* noise, float* out, int64_t frames, int32_t zc,
                                      int32_t hw, int32_t ldm, float scale, void* stream) {
    CC_CHECK_ARG(moments && noise && out && frames > 0 && zc > 0 && hw > 0 && ldm >= 2 * zc,
                 "ccedit_gaussian_sample: bad args");
    const int64_t total = frames * zc * hw;
    hipLaunchKernelGGL(gaussian_sample_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, moments, noise,
                       out, total, zc, hw, ldm, scale);
    return cc_launch_status("gaussian_sample");
}

extern "C" int ccedit_mask_blend(const float* x, const float* z, const float* mask, float* y, int64_t n, void* stream) {
    CC_CHECK_ARG(x && z && mask && y && n > 0, "ccedit_mask_blend: bad args");
    hipLaunchKernelGGL(mask_blend_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, z, mask, y, n);
    return cc_launch_status("mask_blend");
}

extern "C" int ccedit_cfg_denoise(const float* x, const float* eps2, float* den, int64_t n, float sigma, float scale,
                                  void* stream) {
    CC_CHECK_ARG(x && eps2 && den && n > 0, "ccedit_cfg_denoise: bad args");
    hipLaunchKernelGGL(cfg_denoise_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, eps2, den, n, sigma,
                       scale);
    return cc_launch_status("cfg_denoise");
}

extern "C" int ccedit_axpby(const float* x, const float* z, float* y, int64_t n, float a, float b, void* stream) {
    CC_CHECK_ARG(x && z && y && n > 0, "ccedit_axpby: bad args");
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, z, y, n, a, b);
    return cc_launch_status("axpby");
}

extern "C" int ccedit_softmax_rows(const float* s, void* p, int64_t rows, int32_t cols, int32_t cols_pad, int64_t lds,
                                   int64_t ldp, float scale, void* stream) {
    CC_CHECK_ARG(s && p && rows > 0 && cols > 0 && cols_pad >= cols, "ccedit_softmax_rows: bad args");
    CC_UNSUPPORTED(cols_pad > 256 * 32, "ccedit_softmax_rows: cols_pad=%d > 8192", cols_pad);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16*)p, cols,
                       cols_pad, lds, ldp, scale);
    return cc_launch_status("softmax_rows");
}
