// Flash-style attention on bf16 MFMA (v_mfma_f32_32x32x16_bf16) for gfx950.
//
// Serves the three attention shapes of the CCEdit hot path (reference: CrossAttention.forward,
// sgm/modules/attention.py:392-467, which calls F.scaled_dot_product_attention):
//   spatial self-attention (Lq = Lk = h*w per frame, d = 40/80/160), text cross-attention (Lk = 77,
//   K/V shared by the frames of a clip) and temporal self-attention (Lq = Lk = T per pixel).
// q/k/v/o are row-major [rows][ld] with the heads side by side — exactly what the q/k/v GEMMs write in
// the channels-last layout, so there is no 'b n (h d) -> b h n d' transpose anywhere.
//
// Structure (per workgroup: NW waves x 32 query rows, KV tiles of 64 rows):
//   * "swapped" QK^T: S^T[kv][q] = K . Q^T with K as the MFMA A operand and Q (held in registers for
//     the whole kernel) as B, so that every accumulator register of a lane belongs to ONE query
//     (q = lane&31): the online-softmax max/sum are lane-local plus one lane^32 exchange.
//   * the K rows feeding A-row i are permuted (bits 2,3 of i swapped) so that a lane's 8 consecutive
//     S^T registers are 8 consecutive kv positions: exp'd and packed to bf16 they ARE the B operand of
//     the PV product, no cross-lane shuffle.
//   * O^T[dv][q] = V^T . P^T: V stays row-major [kv][d] in LDS and its transposed A-fragments come from
//     ds_read_b64_tr_b16 (two per 16-kv step).  O^T again has q = lane&31 on every register, so the
//     rescale by exp2(m_old - m_new) is a lane-scalar multiply.
//   * K/V tiles are staged by global_load_lds_dwordx4 into a 2-deep LDS ring (tile j+1 in flight while
//     tile j is consumed), one barrier per tile.  Pad columns / rows past Lk come from a zero page.
#include "common.h"
#include <stdlib.h>

bool cc_attn_short_applicable(const CcAttnDesc& a);      // attnshort.hip
int cc_attn_short_launch(const CcAttnDesc& a, hipStream_t s);
bool cc_attn_text_applicable(const CcAttnDesc& a);       // attntext.hip
int cc_attn_text_launch(const CcAttnDesc& a, hipStream_t s);
bool cc_attn_spatial_applicable(const CcAttnDesc& a);    // attnspatial.hip
int cc_attn_spatial_launch(const CcAttnDesc& a, hipStream_t s);

namespace {

__device__ __attribute__((aligned(64))) char g_attn_zero_page[64];

__device__ __forceinline__ bf16x8 tr_pair(const char* p, int second_off) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + second_off));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

// amdgpu_waves_per_eu(2, 8): with a 1-wave lower bound hipcc parks part of the S^T / O^T accumulators in AGPRs and
// pays ~150 v_accvgpr_read/write per KV tile to run the softmax on them (the kernel is VALU-bound: 20 VALU per
// MFMA measured); with >= 2 waves/EU it keeps everything in arch VGPRs (0 moves).
template <int D, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 8))) void attn_kernel(const CcAttnDesc a) {
    constexpr int KS = (D + 15) / 16;         // QK^T k-steps
    constexpr int NT = (D + 31) / 32;         // O^T row tiles
    // LDS row widths.  For d <= 128 rows are padded to a power of two so that an XOR swizzle of the 16-byte
    // granule index makes the K fragment reads (ds_read_b128, rows 96/160 B apart otherwise: 2-way) and the V
    // transpose reads (ds_read_b64_tr_b16: rows r and r+2 on the same banks) conflict-free; the pad granules
    // are DMA'd from the zero page.  d = 160 keeps the compact, unswizzled image (LDS capacity).
    constexpr bool SWZ = (D <= 128);
    constexpr int DK = SWZ ? (D <= 64 ? 64 : 128) : KS * 16;     // K tile width (elements)
    constexpr int DV = SWZ ? (D <= 64 ? 64 : 128) : NT * 32;     // V tile width
    constexpr int GK = DK / 8, GV = DV / 8;   // 16-byte granules per tile row
    constexpr int KB = 64 * DK * 2, VB = 64 * DV * 2;
    constexpr int NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // One KV tile (Lk <= 64: the temporal attention over T = 17 keyframes, one wave per (pixel, head)) needs no ring:
    // half the LDS per workgroup, twice the resident waves on what is a latency-bound launch.
    const int nbuf = a.Lk <= 64 ? 1 : 2;
    char* const sK = smem;                // [nbuf][KB]
    char* const sV = smem + nbuf * KB;    // [nbuf][VB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware 1-D block order (speed only): workgroup b runs on XCD b % 8, so all query tiles of one
    // (batch, head) group are given to ONE XCD back to back — its K/V (0.98 MB at 6144 x 40) then stays in that
    // XCD's 4 MB L2 instead of being fetched from HBM through all 8 L2s (measured: 9.6 GB fetched per launch).
    // Few keys (text cross-attention, Lk = 77: K / V are a few KB, the launch is bound by streaming Q in and O out): the other way
    // round — an XCD takes every eighth (batch, QUERY tile) unit and runs its heads back to back, so the heads of a query
    // row (D-element pieces of one 128-byte line; with the head = XCD mapping above every line of Q is fetched and every line of
    // O written through up to eight L2s) meet in one L2.
    const int qtiles = (a.Lq + NW * 32 - 1) / (NW * 32);
    const int xcd = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    int batch, head, qt;
    if (a.Lk <= 128 && a.seg1_len == 0) {
        const int u = (local / a.heads) * 8 + xcd;         // (batch, query tile) unit: every eighth one is this XCD's, heads back to back
        if (u >= a.batches * qtiles) return;
        head = local % a.heads;
        batch = u / qtiles;
        qt = u - batch * qtiles;
    } else {
        const int grp = (local / qtiles) * 8 + xcd;        // (batch, head) group
        qt = local % qtiles;
        if (grp >= a.batches * a.heads) return;
        batch = grp / a.heads;
        head = grp - batch * a.heads;
    }
    const int q0 = qt * (NW * 32) + wave * 32;

    const bf16* zp = (const bf16*)g_attn_zero_page;
    const int64_t qbase = (int64_t)(batch / a.q_inner) * a.q_outer_rows + (int64_t)(batch % a.q_inner) * a.q_inner_rows;
    const int kvb = batch / a.kv_div;
    const int64_t kvbase = (int64_t)(kvb / a.kv_inner) * a.kv_outer_rows + (int64_t)(kvb % a.kv_inner) * a.kv_inner_rows;
    const bf16* __restrict__ Q = (const bf16*)a.q + head * D;
    const bf16* __restrict__ K = (const bf16*)a.k + head * D;
    const bf16* __restrict__ V = (const bf16*)a.v + head * D;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + l31][16 ks + 8 hi .. +8] ----
    bf16x8 qf[KS];
    {
        const int qi = q0 + l31;
        const bf16* qrow = Q + (size_t)(qbase + (int64_t)qi * a.q_seq_rows) * a.ldq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dofs = ks * 16 + hi * 8;
            const bf16* src = (qi < a.Lq && dofs < D) ? qrow + dofs : zp;
            qf[ks] = *(const bf16x8*)src;
        }
    }

    // ---- DMA plan: every thread owns fixed (row, granule) slots of the K and V tiles; only the tile index moves,
    //      so the source pointers are computed once and advanced by a constant per KV tile ----
    constexpr int ITK = (64 * GK + NTHR - 1) / NTHR, ITV = (64 * GV + NTHR - 1) / NTHR;
    const bf16* kp[ITK];
    const bf16* vp[ITV];
    int krw[ITK], vrw[ITV];            // tile row of the slot, or -1 when the slot is a pad granule / out of range
#pragma unroll
    for (int it = 0; it < ITK; ++it) {
        const int idx = it * NTHR + tid;
        const int row = idx / GK;
        int g = idx - row * GK;
        if (SWZ) g ^= (GK == 8) ? ((row >> 1) & 7) : (row & 15);      // LDS slot -> source granule
        const bool use = (idx < 64 * GK) && (g * 8 < D);               // pad granules are never DMA'd
        krw[it] = use ? row : -1;
        kp[it] = K + (size_t)(kvbase + (int64_t)row * a.kv_seq_rows) * a.ldk + g * 8;
    }
#pragma unroll
    for (int it = 0; it < ITV; ++it) {
        const int idx = it * NTHR + tid;
        const int row = idx / GV;
        int g = idx - row * GV;
        if (SWZ) g ^= ((row >> 1) & 1) << 2;                           // rows r, r+2 -> different bank halves
        const bool use = (idx < 64 * GV) && (g * 8 < D);
        vrw[it] = use ? row : -1;
        vp[it] = V + (size_t)(kvbase + (int64_t)row * a.kv_seq_rows) * a.ldv + g * 8;
    }
    const int64_t kstep = 64 * a.kv_seq_rows * (int64_t)a.ldk, vstep = 64 * a.kv_seq_rows * (int64_t)a.ldv;
    // optional leading segment (anchor-frame keys): its rows are addressed from another kv batch
    int64_t seg1base = 0;
    if (a.seg1_len > 0) {
        const int sb = (batch / a.seg1_div) * a.seg1_mul + a.seg1_add;
        seg1base = (int64_t)(sb / a.kv_inner) * a.kv_outer_rows + (int64_t)(sb % a.kv_inner) * a.kv_inner_rows;
    }

    auto stage = [&](int j, int buf) {
        const int rows_left = a.Lk - j * 64;        // rows >= rows_left of this tile come from the zero page
        if (a.seg1_len > 0) {
            // two-segment keys: recompute the row address per slot (the running pointers assume one segment)
#pragma unroll
            for (int it = 0; it < ITK; ++it) {
                if (krw[it] >= 0) {
                    const int kv = j * 64 + krw[it];
                    const int64_t r = (kv < a.seg1_len) ? seg1base + (int64_t)kv * a.kv_seq_rows
                                                        : kvbase + (int64_t)(kv - a.seg1_len) * a.kv_seq_rows;
                    const bf16* src = (kv < a.Lk) ? kp[it] + (r - kvbase - (int64_t)krw[it] * a.kv_seq_rows) * a.ldk : zp;
                    glds16(src, sK + buf * KB + (it * NTHR + wave * 64) * 16);
                }
            }
#pragma unroll
            for (int it = 0; it < ITV; ++it) {
                if (vrw[it] >= 0) {
                    const int kv = j * 64 + vrw[it];
                    const int64_t r = (kv < a.seg1_len) ? seg1base + (int64_t)kv * a.kv_seq_rows
                                                        : kvbase + (int64_t)(kv - a.seg1_len) * a.kv_seq_rows;
                    const bf16* src = (kv < a.Lk) ? vp[it] + (r - kvbase - (int64_t)vrw[it] * a.kv_seq_rows) * a.ldv : zp;
                    glds16(src, sV + buf * VB + (it * NTHR + wave * 64) * 16);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < ITK; ++it) {
            if (krw[it] >= 0) {
                const bf16* src = (krw[it] < rows_left) ? kp[it] : zp;
                glds16(src, sK + buf * KB + (it * NTHR + wave * 64) * 16);
            }
            kp[it] += kstep;
        }
#pragma unroll
        for (int it = 0; it < ITV; ++it) {
            if (vrw[it] >= 0) {
                const bf16* src = (vrw[it] < rows_left) ? vp[it] : zp;
                glds16(src, sV + buf * VB + (it * NTHR + wave * 64) * 16);
            }
            vp[it] += vstep;
        }
    };

    f32x16 o[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    // fold log2(e): softmax via exp2 (CCEDIT_ATTN_Q_LOG2: q already carries scale * log2 e)
    const float sc = (a.flags & CCEDIT_ATTN_Q_LOG2) ? 1.0f : a.scale * 1.4426950408889634f;

    // A-row i of an S^T tile reads K row swap23(i): i = c | hi2<<2 | b<<3 | a4<<4  ->  c | b<<2 | hi2<<3 | a4<<4
    const int krow_l = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int i16 = lane & 15, dvhalf = (lane >> 4) & 1;
    const int ntiles = (a.Lk + 63) / 64;

    // zero, once, the K pad granules that the last QK^T k-step reads (columns [D, 16*KS)) in both ring slots
    if constexpr (KS * 2 > (D + 7) / 8) {
        constexpr int NPAD = KS * 2 - (D + 7) / 8;
        for (int idx = tid; idx < nbuf * 64 * NPAD; idx += NTHR) {
            const int b = idx / (64 * NPAD), rem = idx - b * 64 * NPAD;
            const int row = rem / NPAD, g = (D + 7) / 8 + (rem - row * NPAD);
            const int slot = SWZ ? (g ^ ((GK == 8) ? ((row >> 1) & 7) : (row & 15))) : g;
            *(u32x4*)(sK + b * KB + row * (DK * 2) + slot * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // When d is not a multiple of 32 the last O^T tile has unused rows: put a column of ones at V[:, D] (written
    // once, never overwritten by the DMA) and the PV MFMA delivers the softmax denominator sum_kv p in O^T row D
    // for free — no per-element row-sum adds in the (VALU-bound) softmax.
    constexpr bool MFMA_ROWSUM = (D % 32 != 0);
    if constexpr (MFMA_ROWSUM) {
        for (int idx = tid; idx < nbuf * 64; idx += NTHR) {
            const int b = idx >> 6, row = idx & 63;
            const int g = D / 8;
            const int slot = SWZ ? (g ^ (((row >> 1) & 1) << 2)) : g;
            *(u32x4*)(sV + b * VB + row * (DV * 2) + slot * 16) = u32x4{0x00003F80u, 0u, 0u, 0u};   // bf16 {1,0,0,...}
        }
    }
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int j = 0; j < ntiles; ++j) {
        if (j + 1 < ntiles) stage(j + 1, buf ^ 1);
        const char* kb = sK + buf * KB;
        const char* vb = sV + buf * VB;

        // ---- S^T = K Q^T for the 64 kv rows of this tile ----
        f32x16 s[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
            const int krow = t2 * 32 + krow_l;
            const char* kr = kb + krow * (DK * 2);
            const int ksw = SWZ ? ((GK == 8) ? ((krow >> 1) & 7) : (krow & 15)) : 0;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(kr + (((ks * 2 + hi) ^ ksw) << 4));
                s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t2], 0, 0, 0);
            }
        }
        // ---- online softmax (this lane: query q0 + l31, kv = 64 j + 32 t2 + 16 (r>>3) + 8 hi + (r&7)) ----
        // The running max is kept in raw-score units; p = exp2(s*sc - m*sc) is one FMA + one v_exp_f32 per
        // element (raw hardware exp2: arguments are <= 0, flush-to-zero of tiny results is what we want).
        if (j * 64 + 64 > a.Lk) {          // wave-uniform: only the last KV tile has masked columns
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = j * 64 + 32 * t2 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (kv >= a.Lk) s[t2][r] = -INFINITY;
                }
        }
        if (a.causal && j * 64 + 63 > q0) {   // wave-uniform: the tile reaches past this wave's first query
            const int qi_c = q0 + l31;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = j * 64 + 32 * t2 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (kv > qi_c) s[t2][r] = -INFINITY;
                }
        }
        float mt = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        // rescale only when some row's max actually moved (exact: alpha == 1 otherwise); wave-uniform branch
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);
            l_run *= alpha;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
            m_run = m_new;
        }
        const float msc = -m_run * sc;
        float psum = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 e = {s[t2][r], s[t2][r + 1]};
                e = __builtin_elementwise_fma(e, f32x2{sc, sc}, f32x2{msc, msc});      // v_pk_fma_f32
                const float p0 = __builtin_amdgcn_exp2f(e[0]), p1 = __builtin_amdgcn_exp2f(e[1]);
                s[t2][r] = p0;
                s[t2][r + 1] = p1;
                if constexpr (!MFMA_ROWSUM) psum += p0 + p1;
            }
        if constexpr (!MFMA_ROWSUM) l_run += psum;

        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = f2bf(s[sp >> 1][8 * (sp & 1) + e]);
            const int vrow = 16 * sp + 8 * hi + (i16 >> 2);                 // rows vrow and vrow + 4: same swizzle bit
            const char* vr = vb + vrow * (DV * 2) + (dvhalf * 16 + (i16 & 3) * 4) * 2;
            const int vsw = SWZ ? (((vrow >> 1) & 1) << 6) : 0;              // granule bit 2 == byte bit 6
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 vf = tr_pair(vr + ((n * 64) ^ vsw), 4 * DV * 2);
                o[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[n], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }

    // ---- normalise and store: lane holds O^T[dv = 32 n + (r&3) + 8 (r>>2) + 4 hi][q = l31] ----
    float l_tot;
    if constexpr (MFMA_ROWSUM) {
        // O^T row D lives in register (D%32/8)*4 of tile D/32 on the hi = 0 lanes
        l_tot = __shfl(o[D / 32][((D % 32) / 8) * 4], l31, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = 1.0f / l_tot;
    const int qi = q0 + l31;
    if (qi < a.Lq) {
        bf16* orow = (bf16*)a.o + (size_t)(qbase + (int64_t)qi * a.q_seq_rows) * a.ldo + head * D;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int dv = 32 * n + 8 * qd + 4 * hi;
                if (dv < D) {
                    bf16x4 w = {f2bf(o[n][qd * 4 + 0] * inv), f2bf(o[n][qd * 4 + 1] * inv), f2bf(o[n][qd * 4 + 2] * inv),
                                f2bf(o[n][qd * 4 + 3] * inv)};
                    *(bf16x4*)(orow + dv) = w;
                }
            }
    }
}

template <int D, int NW>
int launch_attn(const CcAttnDesc& a, hipStream_t s) {
    constexpr bool SWZ = (D <= 128);
    constexpr int DK = SWZ ? (D <= 64 ? 64 : 128) : (D + 15) / 16 * 16;
    constexpr int DV = SWZ ? (D <= 64 ? 64 : 128) : (D + 31) / 32 * 32;
    constexpr int lds = 2 * 64 * DK * 2 + 2 * 64 * DV * 2;
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)attn_kernel<D, NW>, lds, &attr_done, "attn")) return rc;
    const int64_t qtiles = (a.Lq + NW * 32 - 1) / (NW * 32);
    const int64_t groups = ((int64_t)a.batches * a.heads + 7) / 8 * 8;
    const bool by_qtile = a.Lk <= 128 && a.seg1_len == 0;          // block order of the kernel: see there
    dim3 grid((unsigned)(by_qtile ? (qtiles * a.batches + 7) / 8 * 8 * (int64_t)a.heads : qtiles * groups));
    cc_note_kernel("attn_kernel d=%d", D);
    hipLaunchKernelGGL((attn_kernel<D, NW>), grid, dim3(NW * 64), a.Lk <= 64 ? lds / 2 : lds, s, a);
    return cc_launch_status("attn_kernel");
}

template <int D>
int dispatch_nw(const CcAttnDesc& a, hipStream_t s) {
    if (a.Lq <= 32) return launch_attn<D, 1>(a, s);
    if constexpr (D <= 80) {
        if (a.Lq >= 1024) return launch_attn<D, 8>(a, s);     // 256 query rows per K/V tile load: -2.5 % attention time
    }
    return launch_attn<D, 4>(a, s);
}

}  // namespace

extern "C" int ccedit_attention(const CcAttnDesc* desc, void* stream) {
    CC_CHECK_ARG(desc != nullptr, "ccedit_attention: null descriptor");
    const CcAttnDesc& a = *desc;
    CC_CHECK_ARG(a.q && a.k && a.v && a.o, "ccedit_attention: null q/k/v/o");
    CC_CHECK_ARG(a.heads > 0 && a.batches > 0 && a.Lq > 0 && a.Lk > 0 && a.q_inner > 0 && a.kv_inner > 0 && a.kv_div > 0,
                 "ccedit_attention: bad sizes");
    CC_UNSUPPORTED(a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 4, "ccedit_attention: row strides must be multiples of 8");
    CC_CHECK_ARG(a.seg1_len >= 0 && a.seg1_len <= a.Lk && (a.seg1_len == 0 || a.seg1_div > 0), "ccedit_attention: bad leading segment");
    CC_CHECK_ARG(!a.causal || (a.Lq == a.Lk && a.seg1_len == 0), "ccedit_attention: causal needs Lq == Lk and no leading segment");
    CC_UNSUPPORTED(((int64_t)a.batches * a.heads + 8) * ((a.Lq + 31) / 32) > 2147483647LL, "ccedit_attention: grid too large");
    hipStream_t s = (hipStream_t)stream;
    // temporal self-attention (T <= 32 keyframes per pixel): HBM-bound, own kernel organised around whole-row loads
    const bool plain_q = !(a.flags & CCEDIT_ATTN_Q_LOG2);       // the two kernels below apply `scale` themselves
    if (cc_policy().attn_short && plain_q && cc_attn_short_applicable(a)) return cc_attn_short_launch(a, s);
    // text cross-attention (<= 96 keys shared by the frames of a clip): bound by streaming the query rows, own kernel (attntext.hip)
    if (cc_policy().attn_text && plain_q && cc_attn_text_applicable(a)) return cc_attn_text_launch(a, s);
    // long self-attention at d = 40 (the 64x96 level): the softmax arithmetic is the bound, own kernel (attnspatial.hip)
    if (cc_policy().attn_spatial && cc_attn_spatial_applicable(a)) return cc_attn_spatial_launch(a, s);
    switch (a.d) {
        case 8: return dispatch_nw<8>(a, s);
        case 16: return dispatch_nw<16>(a, s);
        case 32: return dispatch_nw<32>(a, s);
        case 40: return dispatch_nw<40>(a, s);
        case 64: return dispatch_nw<64>(a, s);
        case 80: return dispatch_nw<80>(a, s);
        case 128: return dispatch_nw<128>(a, s);
        case 160: return dispatch_nw<160>(a, s);
        default: break;
    }
    cc_set_error("ccedit_attention: head dim %d not instantiated (have 8,16,32,40,64,80,128,160)", a.d);
    return CCEDIT_EUNSUPPORTED;
}
