// Attention over SHORT sequences (Lq, Lk <= 32) for gfx950: the temporal self-attention of SpatialTransformer3D
// (attention.py:1172-1204) — every pixel attends over the T = 17 keyframes, 12288 / 3072 / 768 sequences of 17 rows
// per launch, rows H*W apart in the frames-outermost layout.
//
// This case is HBM-bound (0.5 GB in and out at the 64x96 level, 4.5 GFLOP), so the kernel is organised around the
// memory access, not the MFMA: one workgroup owns 320 consecutive channels (8 / 4 / 2 heads of d = 40 / 80 / 160) of
// ONE pixel, reads its Lq + 2 Lk rows as whole 640-byte segments (5 full cache lines each), keeps them in LDS, runs
// one 32x32 score tile per head on the MFMA (S^T = K Q^T with the key rows permuted so that the exponentiated scores
// are directly the B operand of O^T = V^T P^T, as in attention.hip), and writes the output rows back as whole 640-byte
// segments through an LDS staging tile.  The general flash kernel (attention.hip, one 64-thread workgroup per
// (pixel, head), 80-byte row pieces) reaches 2.2 / 1.9 / 0.9 TB/s on these launches.
#include "common.h"

namespace {

constexpr int GW = 320;                   // channels per workgroup
constexpr int GR = GW / 8;                // 16-byte granules per row
constexpr int RS = GW * 2 + 16;           // LDS row stride: 164 dwords, rows land 36 banks apart

template <int D>
__global__ __launch_bounds__(256) void attn_short_kernel(const CcAttnDesc a) {
    constexpr int HG = GW / D;            // heads per workgroup
    constexpr int KS = (D + 15) / 16;     // k-steps of the score MFMA
    constexpr int DT = (D + 31) / 32;     // 32-channel tiles of the output
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Lq = a.Lq, Lk = a.Lk;
    char* const sQ = smem;                // [Lq][RS]
    char* const sK = sQ + Lq * RS;        // [Lk][RS]
    char* const sV = sK + Lk * RS;        // [Lk][RS]
    char* const sO = sV + Lk * RS;        // [Lq][RS] output staging
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int groups = a.heads * D / GW;
    const int batch = blockIdx.x / groups;
    const int c0 = (blockIdx.x - batch * groups) * GW;
    const int64_t qbase = (int64_t)(batch / a.q_inner) * a.q_outer_rows + (int64_t)(batch % a.q_inner) * a.q_inner_rows;
    const int kvb = batch / a.kv_div;
    const int64_t kvbase = (int64_t)(kvb / a.kv_inner) * a.kv_outer_rows + (int64_t)(kvb % a.kv_inner) * a.kv_inner_rows;
    const bf16* __restrict__ Q = (const bf16*)a.q + c0;
    const bf16* __restrict__ K = (const bf16*)a.k + c0;
    const bf16* __restrict__ V = (const bf16*)a.v + c0;

    // ---- global -> LDS: (Lq + 2 Lk) rows of 40 granules, 8 loads in flight per thread ----
    const int nq = Lq * GR, nk = Lk * GR, total = nq + 2 * nk;
    for (int base = 0; base < total; base += 8 * 256) {
        bf16x8 buf[8];
        char* dst[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = base + it * 256 + tid;
            dst[it] = nullptr;
            if (idx < total) {
                const bf16* src;
                if (idx < nq) {
                    const int r = idx / GR, g = idx - r * GR;
                    src = Q + (size_t)(qbase + (int64_t)r * a.q_seq_rows) * a.ldq + g * 8;
                    dst[it] = sQ + r * RS + g * 16;
                } else if (idx < nq + nk) {
                    const int i2 = idx - nq, r = i2 / GR, g = i2 - r * GR;
                    src = K + (size_t)(kvbase + (int64_t)r * a.kv_seq_rows) * a.ldk + g * 8;
                    dst[it] = sK + r * RS + g * 16;
                } else {
                    const int i2 = idx - nq - nk, r = i2 / GR, g = i2 - r * GR;
                    src = V + (size_t)(kvbase + (int64_t)r * a.kv_seq_rows) * a.ldv + g * 8;
                    dst[it] = sV + r * RS + g * 16;
                }
                buf[it] = *(const bf16x8*)src;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it)
            if (dst[it]) *(bf16x8*)dst[it] = buf[it];
    }
    __syncthreads();

    // MFMA row m of the score tile holds key perm(m) = m with bits 2 and 3 swapped: accumulator register j of lane
    // (query n, half hi) is then key (j & 3) + 4 ((j >> 2) & 1) + 8 hi + 16 (j >> 3), i.e. registers 8 kk .. 8 kk + 7
    // are the keys 16 kk + 8 hi .. + 7 that the B operand of the second MFMA wants from this lane.
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const float sl2 = a.scale * 1.44269504088896340736f;
    const int KK = (Lk + 15) >> 4;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int h = wave; h < HG; h += 4) {
        const int hc = h * D;
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dofs = ks * 16 + hi * 8;
            const bool dv = dofs < D;
            const bf16x8 kf = (dv && krow < Lk) ? *(const bf16x8*)(sK + krow * RS + (hc + dofs) * 2) : zero8;
            const bf16x8 qf = (dv && l31 < Lq) ? *(const bf16x8*)(sQ + l31 * RS + (hc + dofs) * 2) : zero8;
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, sacc, 0, 0, 0);
        }
        // softmax over the keys of query l31: 16 keys in this lane, the other 16 in lane ^ 32
        float p[16];
        float mx = -3.0e38f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = (j & 3) + 4 * ((j >> 2) & 1) + 8 * hi + 16 * (j >> 3);
            p[j] = key < Lk ? sacc[j] * sl2 : -3.0e38f;
            mx = fmaxf(mx, p[j]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int key = (j & 3) + 4 * ((j >> 2) & 1) + 8 * hi + 16 * (j >> 3);
            p[j] = key < Lk ? __builtin_amdgcn_exp2f(p[j] - mx) : 0.f;
        }
        bf16x8 pf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pf[kk][e] = f2bf(p[kk * 8 + e]);
                sum += bf2f(pf[kk][e]);              // the denominator of what is actually multiplied into V
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;

        // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; A operand: 8 consecutive keys of one channel
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x16 oacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
            const int dch = dt * 32 + l31;
            const bool cv = dch < D;
            for (int kk = 0; kk < KK; ++kk) {
                union {
                    bf16x8 v;
                    unsigned short u[8];
                } vf;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int key = kk * 16 + hi * 8 + e;
                    const int kc = key < Lk ? key : Lk - 1;
                    const unsigned short raw = *(const unsigned short*)(sV + kc * RS + (hc + (cv ? dch : 0)) * 2);
                    vf.u[e] = (cv && key < Lk) ? raw : (unsigned short)0;
                }
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf[kk], oacc, 0, 0, 0);
            }
            // accumulator register j of lane (query l31, hi) = channel dt*32 + (j & 3) + 8 (j >> 2) + 4 hi
            if (l31 < Lq) {
#pragma unroll
                for (int jg = 0; jg < 4; ++jg) {
                    const int dd = dt * 32 + 8 * jg + 4 * hi;
                    if (dd < D) {
                        bf16x4 o = {f2bf(oacc[jg * 4 + 0] * inv), f2bf(oacc[jg * 4 + 1] * inv), f2bf(oacc[jg * 4 + 2] * inv),
                                    f2bf(oacc[jg * 4 + 3] * inv)};
                        *(bf16x4*)(sO + l31 * RS + (hc + dd) * 2) = o;
                    }
                }
            }
        }
    }
    __syncthreads();

    // ---- LDS -> global: Lq rows of 40 granules ----
    bf16* __restrict__ O = (bf16*)a.o + c0;
    for (int idx = tid; idx < nq; idx += 256) {
        const int r = idx / GR, g = idx - r * GR;
        *(bf16x8*)(O + (size_t)(qbase + (int64_t)r * a.q_seq_rows) * a.ldo + g * 8) = *(const bf16x8*)(sO + r * RS + g * 16);
    }
}

template <int D>
int launch_short(const CcAttnDesc& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)attn_short_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * RS);
        if (e != hipSuccess) {
            cc_set_error("hipFuncSetAttribute(attn_short): %s", hipGetErrorString(e));
            return (int)e;
        }
        attr_set = true;
    }
    const int64_t nblk = (int64_t)a.batches * (a.heads * D / GW);
    if (nblk > 2147483647LL) {
        cc_set_error("ccedit_attention: grid too large");
        return CCEDIT_EUNSUPPORTED;
    }
    hipLaunchKernelGGL((attn_short_kernel<D>), dim3((unsigned)nblk), dim3(256), (2 * a.Lq + 2 * a.Lk) * RS, s, a);
    return cc_launch_status("attn_short_kernel");
}

}  // namespace

// Short sequences whose heads tile 320-channel groups, with 16-byte aligned rows.
bool cc_attn_short_applicable(const CcAttnDesc& a) {
    return (a.d == 40 || a.d == 80 || a.d == 160) && (a.heads * a.d) % GW == 0 && a.Lq <= 32 && a.Lk <= 32 &&
           a.seg1_len == 0 && !a.causal && a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 8 == 0 &&
           ((uintptr_t)a.q % 16 == 0) && ((uintptr_t)a.k % 16 == 0) && ((uintptr_t)a.v % 16 == 0) && ((uintptr_t)a.o % 16 == 0);
}

int cc_attn_short_launch(const CcAttnDesc& a, hipStream_t s) {
    switch (a.d) {
        case 40: return launch_short<40>(a, s);
        case 80: return launch_short<80>(a, s);
        default: return launch_short<160>(a, s);
    }
}
