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

// LDS-only workgroup barrier: __syncthreads() would also drain vmcnt, i.e. wait for the prefetch of the next pixel
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int D>
__global__ __launch_bounds__(256) void attn_short_kernel(const CcAttnDesc a, int nblk) {
    constexpr int HG = GW / D;            // heads per workgroup
    constexpr int HW = (HG + 3) / 4;      // heads per wave
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
    const int nq = Lq * GR, total = (Lq + 2 * Lk) * GR;      // <= 8 * 256 granules (launcher)

    // Per-thread slots of the row list [Q rows | K rows | V rows] (sQ, sK, sV are contiguous, same stride): the row,
    // granule and LDS address of a slot do not depend on the pixel, only the base row does.
    int64_t srow[8];                       // row offset relative to the pixel's base row, times the row stride
    int soff[8];                           // LDS byte offset, or -1
    unsigned char skind[8];                // 0 = q, 1 = k, 2 = v
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 256 + tid;
        const int ic = min(idx, total - 1);
        const int r = ic / GR, g = ic - r * GR;
        const bool isq = r < Lq, isk = r < Lq + Lk;
        const int rr = isq ? r : (isk ? r - Lq : r - Lq - Lk);
        const int ld = isq ? a.ldq : (isk ? a.ldk : a.ldv);
        srow[it] = (int64_t)rr * (isq ? a.q_seq_rows : a.kv_seq_rows) * ld + g * 8;
        soff[it] = idx < total ? r * RS + g * 16 : -1;
        skind[it] = isq ? 0 : (isk ? 1 : 2);
    }
    auto bases = [&](int blk, int64_t& qbase, int64_t& kvbase, int& c0) {
        const int batch = blk / groups;
        c0 = (blk - batch * groups) * GW;
        qbase = (int64_t)(batch / a.q_inner) * a.q_outer_rows + (int64_t)(batch % a.q_inner) * a.q_inner_rows;
        const int kvb = batch / a.kv_div;
        kvbase = (int64_t)(kvb / a.kv_inner) * a.kv_outer_rows + (int64_t)(kvb % a.kv_inner) * a.kv_inner_rows;
    };
    bf16x8 buf[8];
    auto prefetch = [&](int blk) {
        int64_t qbase, kvbase;
        int c0;
        bases(blk, qbase, kvbase, c0);
        const bf16* qb = (const bf16*)a.q + qbase * a.ldq + c0;
        const bf16* kb = (const bf16*)a.k + kvbase * a.ldk + c0;
        const bf16* vb = (const bf16*)a.v + kvbase * a.ldv + c0;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const bf16* bp = skind[it] == 0 ? qb : (skind[it] == 1 ? kb : vb);
            buf[it] = *(const bf16x8*)(bp + srow[it]);
        }
    };

    // MFMA row m of the score tile holds key perm(m) = m with bits 2 and 3 swapped: accumulator register j of lane
    // (query n, half hi) is then key (j & 3) + 4 ((j >> 2) & 1) + 8 hi + 16 (j >> 3), i.e. registers 8 kk .. 8 kk + 7
    // are the keys 16 kk + 8 hi .. + 7 that the B operand of the second MFMA wants from this lane.
    const int krow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const float sl2 = a.scale * 1.44269504088896340736f;
    const int KK = (Lk + 15) >> 4;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    // Persistent loop, software pipelined across pixels: while pixel i is computed, the rows of pixel i+1 are in flight
    // to registers and the output rows of pixel i-1 drain from the staging tile.
    int blk = blockIdx.x;
    int64_t o_qbase = 0;
    int o_c0 = 0;
    bool have_out = false;
    prefetch(blk);
    for (; blk < nblk; blk += gridDim.x) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
            if (soff[it] >= 0) *(bf16x8*)(smem + soff[it]) = buf[it];
        lds_barrier();                                   // (1) this pixel's rows are in LDS
        if (have_out) {                                  // pixel i-1: staging tile -> global, whole 640-byte rows
            bf16* __restrict__ O = (bf16*)a.o + o_c0;
            for (int idx = tid; idx < nq; idx += 256) {
                const int r = idx / GR, g = idx - r * GR;
                *(bf16x8*)(O + (size_t)(o_qbase + (int64_t)r * a.q_seq_rows) * a.ldo + g * 8) = *(const bf16x8*)(sO + r * RS + g * 16);
            }
        }
        if (blk + (int)gridDim.x < nblk) prefetch(blk + gridDim.x);
        {
            int64_t kvb_unused;
            bases(blk, o_qbase, kvb_unused, o_c0);
            have_out = true;
        }

        bf16x8 pf[HW][2];
        float inv[HW];
#pragma unroll
        for (int hh = 0; hh < HW; ++hh) {
            const int h = wave + 4 * hh;
            if (h >= HG) break;
            const int hc = h * D;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int dofs = ks * 16 + hi * 8;
                const bool dv = dofs < D;
                // rows past Lk / Lq read the neighbouring tile (finite values): those keys are masked below, those
                // queries never stored.  Channels past D belong to the next head: zeroed in both operands.
                const int cofs = dv ? hc + dofs : 0;
                bf16x8 kf = *(const bf16x8*)(sK + krow * RS + cofs * 2);
                bf16x8 qf = *(const bf16x8*)(sQ + l31 * RS + cofs * 2);
                kf = dv ? kf : zero8;
                qf = dv ? qf : zero8;
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
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    pf[hh][kk][e] = f2bf(p[kk * 8 + e]);
                    sum += bf2f(pf[hh][kk][e]);          // the denominator of what is actually multiplied into V
                }
            sum += __shfl_xor(sum, 32, 64);
            inv[hh] = 1.0f / sum;
        }
        lds_barrier();                                   // (2) the staging tile of pixel i-1 has been read out

        // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; A operand: 8 consecutive keys of one channel
#pragma unroll
        for (int hh = 0; hh < HW; ++hh) {
            const int h = wave + 4 * hh;
            if (h >= HG) break;
            const int hc = h * D;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                f32x16 oacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
                // channels past D (rows of O^T that are never stored) read a valid neighbouring channel; keys past Lk
                // are clamped to the last row: their probabilities are exactly 0
                const int dch = min(dt * 32 + l31, D - 1);
                const char* vcol = sV + (hc + dch) * 2;
                for (int kk = 0; kk < KK; ++kk) {
                    union {
                        bf16x8 v;
                        unsigned short u[8];
                    } vf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int kc = min(kk * 16 + hi * 8 + e, Lk - 1);
                        vf.u[e] = *(const unsigned short*)(vcol + kc * RS);
                    }
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, kk == 0 ? pf[hh][0] : pf[hh][1], oacc, 0, 0, 0);
                }
                // accumulator register j of lane (query l31, hi) = channel dt*32 + (j & 3) + 8 (j >> 2) + 4 hi
                if (l31 < Lq) {
#pragma unroll
                    for (int jg = 0; jg < 4; ++jg) {
                        const int dd = dt * 32 + 8 * jg + 4 * hi;
                        if (dd < D) {
                            const float iv = inv[hh];
                            bf16x4 o = {f2bf(oacc[jg * 4 + 0] * iv), f2bf(oacc[jg * 4 + 1] * iv), f2bf(oacc[jg * 4 + 2] * iv),
                                        f2bf(oacc[jg * 4 + 3] * iv)};
                            *(bf16x4*)(sO + l31 * RS + (hc + dd) * 2) = o;
                        }
                    }
                }
            }
        }
        lds_barrier();                                   // (3) rows of pixel i are no longer read; its output is staged
    }
    if (have_out) {
        bf16* __restrict__ O = (bf16*)a.o + o_c0;
        for (int idx = tid; idx < nq; idx += 256) {
            const int r = idx / GR, g = idx - r * GR;
            *(bf16x8*)(O + (size_t)(o_qbase + (int64_t)r * a.q_seq_rows) * a.ldo + g * 8) = *(const bf16x8*)(sO + r * RS + g * 16);
        }
    }
}

template <int D>
int launch_short(const CcAttnDesc& a, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)attn_short_kernel<D>, 128 * RS, &attr_done, "attn_short")) return rc;
    const int64_t nblk = (int64_t)a.batches * (a.heads * D / GW);
    if (nblk > 2147483647LL) {
        cc_set_error("ccedit_attention: grid too large");
        return CCEDIT_EUNSUPPORTED;
    }
    const int lds = (2 * a.Lq + 2 * a.Lk) * RS;
    const int per_cu = 160 * 1024 / lds < 4 ? 160 * 1024 / lds : 4;          // resident workgroups per CU (LDS / 16 waves)
    const int64_t grid = nblk < 256 * per_cu ? nblk : 256 * per_cu;
    cc_note_kernel("attn_short_kernel d=%d", D);
    hipLaunchKernelGGL((attn_short_kernel<D>), dim3((unsigned)grid), dim3(256), lds, s, a, (int)nblk);
    return cc_launch_status("attn_short_kernel");
}

}  // namespace

// Short sequences whose heads tile 320-channel groups, with 16-byte aligned rows.
bool cc_attn_short_applicable(const CcAttnDesc& a) {
    return (a.d == 40 || a.d == 80 || a.d == 160) && (a.heads * a.d) % GW == 0 && a.Lq <= 32 && a.Lk <= 32 &&
           (a.Lq + 2 * a.Lk) * GR <= 8 * 256 &&          // one 8-deep register prefetch per thread covers a pixel: T <= 17
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
