// Tap-gather GEMM on bf16 MFMA (v_mfma_f32_32x32x16_bf16) for gfx950.
//
//   out[m][n] = epilogue( sum_{tap, c} W[n][tap][c] * A[src(m, tap)][c] )
//
// One kernel family for every contraction of the CCEdit hot path in the frames-outermost
// channels-last layout: Linear / Conv 1x1 / Conv1d k1 (1 tap, identity), Conv2d 3x3 s1/s2 (+ fused
// nearest-2x upsample of the source), Conv1d k3 over the T keyframes (3 taps, rows H*W apart).
// See include/ccedit_hip.h for the reference call sites.
//
// Mapping to the hardware
//   * MFMA A operand (32 rows i) = weights (output channels), B operand (32 cols j) = pixels, so each
//     lane of the accumulator holds 4 consecutive CHANNELS of one pixel per register quad: the
//     epilogue stores 8-byte channel runs into the channels-last output (lanes l, l+32 adjacent).
//   * block = 256 threads = 4 waves, each wave a 64ch x 64pix sub-tile (2x2 MFMA tiles, 64 fp32
//     accumulators per lane).  Two block shapes: 128ch x 128pix (waves 2x2) and 64ch x 256pix (1x4,
//     for Cout = 320 / 960 which are multiples of 64 but not 128).
//   * K tile = 64 bf16 = one 128-byte LDS row per tile row; both operands are staged with
//     global_load_lds_dwordx4 (no VGPR round trip), double buffered, one barrier per K tile.
//   * LDS is written lane-linearly by the DMA, so the bank swizzle is applied on the SOURCE granule
//     index and again on the fragment read: 16-byte granule g of row r lives at slot g ^ ((r>>1)&7).
//     With the 32x32x16 fragment pattern (lane -> row l&31, granule 2*ks + (l>>5)) every ds_read_b128
//     lane group touches 16 distinct 16-byte slots of the 256-byte bank row: conflict-free.
//   * the implicit-GEMM gather (3x3 halo, stride 2, upsample, temporal neighbours, concat of two
//     sources, K/M tails) only changes the per-lane source ADDRESS of the DMA; out-of-range granules
//     read a 64-byte zero page.
#include "common.h"

namespace {

__device__ __attribute__((aligned(64))) char g_zero_page[64];

constexpr int kRowBytes = 128;   // 64 bf16 per tile row

template <int WM, int WN>
__global__ __launch_bounds__(256) void tap_gemm_kernel(const CcGemmDesc d) {
    constexpr int BMC = WM * 64;             // channels per block
    constexpr int BNP = WN * 64;             // pixels per block
    constexpr int A_ISSUES = BMC / 32;
    constexpr int B_ISSUES = BNP / 32;
    constexpr int A_BYTES = BMC * kRowBytes;
    constexpr int B_BYTES = BNP * kRowBytes;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;                    // [2][A_BYTES]
    char* const sB = smem + 2 * A_BYTES;      // [2][B_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int64_t pix0 = (int64_t)blockIdx.x * BNP;
    const int ch0 = blockIdx.y * BMC;

    // ---- staging coordinates: thread -> (row rsub + 32*i, LDS slot p) ----
    const int p = tid & 7;
    const int rsub = tid >> 3;
    const int gcol = p ^ ((rsub >> 1) & 7);    // source granule held in slot p of this row
    const int gpt = d.Cin >> 3;                // 16-byte granules per tap
    const int gtot = d.taps * gpt;
    const int nk = d.Kpad / 64;

    const bf16* __restrict__ Ap = (const bf16*)d.A;
    const bf16* __restrict__ A2p = (const bf16*)d.A2;
    const bf16* __restrict__ Wp = (const bf16*)d.W;
    const bf16* zp = (const bf16*)g_zero_page;

    // per staged pixel row: (base, a, b), meaning depends on the mode
    int64_t rbase[B_ISSUES];
    int ra[B_ISSUES], rb[B_ISSUES];
#pragma unroll
    for (int i = 0; i < B_ISSUES; ++i) {
        const int64_t m = pix0 + i * 32 + rsub;
        const bool ok = m < d.M;
        if (d.mode == CCEDIT_GEMM_CONV2D) {
            const int hwout = d.Hout * d.Wout;
            const int64_t n = m / hwout;
            const int rem = (int)(m - n * hwout);
            const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
            rbase[i] = n * (int64_t)(d.Hin * d.Win);
            ra[i] = ok ? oy * d.stride - d.pad : -100000;
            rb[i] = ox * d.stride - d.pad;
        } else if (d.mode == CCEDIT_GEMM_TEMPORAL) {
            const int64_t frame = m / d.HW;
            rbase[i] = m;
            ra[i] = ok ? (int)(frame % d.T) : -100000;
            rb[i] = 0;
        } else {
            rbase[i] = m;
            ra[i] = ok ? 0 : -100000;
            rb[i] = 0;
        }
    }

    // running (tap, granule-in-tap) of this thread's source granule for the next tile to stage
    int s_tap = gcol / gpt;
    int s_cg = gcol - s_tap * gpt;

    auto stage = [&](int kt, int buf) {
        // weights: always in range (rows and K are zero-padded by the packer)
#pragma unroll
        for (int i = 0; i < A_ISSUES; ++i) {
            const bf16* src = Wp + (size_t)(ch0 + i * 32 + rsub) * d.Kpad + kt * 64 + gcol * 8;
            glds16(src, sA + buf * A_BYTES + i * 4096 + wave * 1024);
        }
        // activations: gather
        const bool kvalid = (kt * 8 + gcol) < gtot;
        const int c0 = s_cg * 8;
        const bool second = c0 >= d.Cin1;
        const bf16* sp = second ? A2p : Ap;
        const int ld = second ? d.lda2 : d.lda;
        const int cc = second ? c0 - d.Cin1 : c0;
        int dy = 0, dx = 0;
        if (d.mode == CCEDIT_GEMM_CONV2D) {
            dy = s_tap / d.ksize;
            dx = s_tap - dy * d.ksize;
        } else if (d.mode == CCEDIT_GEMM_TEMPORAL) {
            dy = s_tap - (d.taps >> 1);
        }
#pragma unroll
        for (int i = 0; i < B_ISSUES; ++i) {
            const bf16* src = zp;
            if (d.mode == CCEDIT_GEMM_CONV2D) {
                int iy = ra[i] + dy, ix = rb[i] + dx;
                bool v;
                if (d.upsample) {
                    v = (iy >= 0) & (iy < 2 * d.Hin) & (ix >= 0) & (ix < 2 * d.Win);
                    iy >>= 1;
                    ix >>= 1;
                } else {
                    v = (iy >= 0) & (iy < d.Hin) & (ix >= 0) & (ix < d.Win);
                }
                if (v & kvalid) src = sp + (size_t)(rbase[i] + iy * d.Win + ix) * ld + cc;
            } else if (d.mode == CCEDIT_GEMM_TEMPORAL) {
                const int t = ra[i] + dy;
                if ((t >= 0) & (t < d.T) & kvalid) src = sp + (size_t)(rbase[i] + (int64_t)dy * d.HW) * ld + cc;
            } else {
                if ((ra[i] >= 0) & kvalid) src = sp + (size_t)rbase[i] * ld + cc;
            }
            glds16(src, sB + buf * B_BYTES + i * 4096 + wave * 1024);
        }
        // advance to the next K tile (+8 granules)
        s_cg += 8;
        while (s_cg >= gpt) {
            s_cg -= gpt;
            ++s_tap;
        }
    };

    // ---- fragment read coordinates ----
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int sw = (l31 >> 1) & 7;
    const char* fa = sA + (wm * 64 + l31) * kRowBytes;
    const char* fb = sB + (wn * 64 + l31) * kRowBytes;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const char* pa = fa + buf * A_BYTES;
        const char* pb = fb + buf * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((ks * 2 + hi) ^ sw) << 4;
            bf16x8 a0 = *(const bf16x8*)(pa + off);
            bf16x8 a1 = *(const bf16x8*)(pa + 32 * kRowBytes + off);
            bf16x8 b0 = *(const bf16x8*)(pb + off);
            bf16x8 b1 = *(const bf16x8*)(pb + 32 * kRowBytes + off);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
    };

    // ---- main loop: stage(t+1) || compute(t), one barrier per tile ----
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk - 1; ++kt) {
        stage(kt + 1, cur ^ 1);
        compute(cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---- epilogue ----
    const float* __restrict__ bias = d.bias;
    const float* __restrict__ gbias = d.group_bias;
    const bf16* __restrict__ r1 = (const bf16*)d.res1;
    const bf16* __restrict__ r2 = (const bf16*)d.res2;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int64_t m = pix0 + wn * 64 + tj * 32 + l31;
        if (m >= d.M) continue;
        const float* gb = gbias ? gbias + (size_t)(m / d.group_rows) * d.N : nullptr;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            if (d.act == CCEDIT_ACT_GEGLU) {
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    const int rx = ch0 + wm * 64 + ti * 32 + q * 8 + hi * 4;   // packed rows of x
                    if (rx >= d.N) continue;
                    const int oc = ((ch0 + wm * 64 + ti * 32 + q * 8) >> 1) + hi * 4;
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float xv = acc[ti][tj][q * 4 + e], gv = acc[ti][tj][(q + 1) * 4 + e];
                        if (bias) {
                            xv += bias[rx + e];
                            gv += bias[rx + 8 + e];
                        }
                        o[e] = f2bf(xv * gelu_erf_f(gv));
                    }
                    *(bf16x4*)((bf16*)d.out + (size_t)m * d.ldc + oc) = o;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cb = ch0 + wm * 64 + ti * 32 + q * 8 + hi * 4;
                    if (cb >= d.N) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[ti][tj][q * 4 + e];
                        if (bias) v[e] += bias[cb + e];
                        if (gb) v[e] += gb[cb + e];
                        if (d.act == CCEDIT_ACT_SILU) v[e] = silu_f(v[e]);
                    }
                    if (r1) {
                        const bf16x4 rv = *(const bf16x4*)(r1 + (size_t)m * d.ldr1 + cb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
                    }
                    if (r2) {
                        const bf16x4 rv = *(const bf16x4*)(r2 + (size_t)m * d.ldr2 + cb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
                    }
                    if (d.out_f32) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *(f32x4*)((float*)d.out + (size_t)m * d.ldc + cb) = o;
                    } else {
                        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                        *(bf16x4*)((bf16*)d.out + (size_t)m * d.ldc + cb) = o;
                    }
                }
            }
        }
    }
}

template <int WM, int WN>
int launch(const CcGemmDesc& d, hipStream_t s) {
    constexpr int BMC = WM * 64, BNP = WN * 64;
    constexpr int lds = 2 * (BMC + BNP) * kRowBytes;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)tap_gemm_kernel<WM, WN>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            cc_set_error("hipFuncSetAttribute(tap_gemm): %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    dim3 grid((unsigned)((d.M + BNP - 1) / BNP), (unsigned)((d.N + BMC - 1) / BMC));
    hipLaunchKernelGGL((tap_gemm_kernel<WM, WN>), grid, dim3(256), lds, s, d);
    return cc_launch_status("tap_gemm_kernel");
}

}  // namespace

extern "C" int ccedit_gemm(const CcGemmDesc* desc, void* stream) {
    CC_CHECK_ARG(desc != nullptr, "ccedit_gemm: null descriptor");
    CcGemmDesc d = *desc;
    CC_CHECK_ARG(d.A && d.W && d.out, "ccedit_gemm: null A/W/out");
    CC_CHECK_ARG(d.M > 0 && d.N > 0 && d.Cin > 0 && d.taps > 0, "ccedit_gemm: bad sizes M=%lld N=%d Cin=%d taps=%d",
                 (long long)d.M, d.N, d.Cin, d.taps);
    CC_UNSUPPORTED(d.Cin % 8 != 0, "ccedit_gemm: Cin=%d must be a multiple of 8", d.Cin);
    CC_UNSUPPORTED(d.N % 4 != 0, "ccedit_gemm: N=%d must be a multiple of 4", d.N);
    CC_UNSUPPORTED(d.Kpad % 64 != 0 || d.Kpad < d.taps * d.Cin, "ccedit_gemm: Kpad=%d invalid for taps*Cin=%d", d.Kpad,
                   d.taps * d.Cin);
    CC_UNSUPPORTED(d.lda % 8 != 0 || d.ldc % 4 != 0, "ccedit_gemm: lda=%d / ldc=%d alignment", d.lda, d.ldc);
    if (d.A2 == nullptr) d.Cin1 = d.Cin;
    CC_UNSUPPORTED(d.A2 && (d.Cin1 % 8 != 0 || d.lda2 % 8 != 0 || d.Cin1 <= 0 || d.Cin1 >= d.Cin),
                   "ccedit_gemm: bad concat split Cin1=%d lda2=%d", d.Cin1, d.lda2);
    if (d.mode == CCEDIT_GEMM_CONV2D) {
        CC_CHECK_ARG(d.ksize * d.ksize == d.taps && d.Hin > 0 && d.Win > 0 && d.Hout > 0 && d.Wout > 0 && d.stride > 0,
                     "ccedit_gemm: bad conv2d geometry");
        CC_CHECK_ARG(d.M % ((int64_t)d.Hout * d.Wout) == 0, "ccedit_gemm: M not a whole number of frames");
    } else if (d.mode == CCEDIT_GEMM_TEMPORAL) {
        CC_CHECK_ARG(d.T > 0 && d.HW > 0 && d.M % ((int64_t)d.T * d.HW) == 0, "ccedit_gemm: bad temporal geometry");
        CC_CHECK_ARG(d.taps % 2 == 1, "ccedit_gemm: temporal taps must be odd");
    } else {
        CC_CHECK_ARG(d.mode == CCEDIT_GEMM_LINEAR && d.taps == 1, "ccedit_gemm: bad mode/taps");
    }
    if (d.act == CCEDIT_ACT_GEGLU) {
        CC_UNSUPPORTED(d.N % 16 != 0 || d.res1 || d.res2 || d.group_bias || d.out_f32,
                       "ccedit_gemm: GEGLU epilogue needs N%%16==0 and no residual/group bias/f32 out");
    }
    if (d.group_bias) CC_CHECK_ARG(d.group_rows > 0, "ccedit_gemm: group_bias without group_rows");
    hipStream_t s = (hipStream_t)stream;
    int tile = d.tile;
    if (tile == 0) {
        const int w128 = (d.N + 127) / 128 * 128, w64 = (d.N + 63) / 64 * 64;
        tile = (w64 < w128) ? 2 : 1;
    }
    if (tile == 2) return launch<1, 4>(d, s);
    return launch<2, 2>(d, s);
}
