// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 2; numbers in DESIGN.md): bit-correct (it passed every attention test
// while it was dispatched) but SLOWER than attn_kernel — 2.92 ms (plain), 3.70 ms (fragments preloaded in the vector
// segment) against 2.58 ms for the 34 x 8 x 6144^2, d = 40 launch.  PMC: the kernel is bound by VALU ISSUE, not by a lack of
// overlap: 121 VALU instructions per wave and KV tile (161 here) at ~5 cycles each = 670 cycles against 448 cycles of MFMA;
// with two waves per SIMD and two barriers per tile the waves of this version wait 41 % of their cycles.
//
// Long-sequence attention (spatial self-attention 6144 / 1536 tokens, TVI2V anchor + self 2 x 6144 keys) with the two
// halves of an 8-wave workgroup running in counter-phase ("ping-pong") — gfx950.
//
// Same data layout, MFMA operand tricks and numerics as attn_kernel (attention.hip: swapped QK^T with Q in registers,
// permuted K rows so that exp'd scores are the PV B operand, ds_read_b64_tr_b16 for V^T, denominator from a ones column);
// what changes is WHEN a wave does what.  Per KV tile a wave has a matrix segment (QK^T of tile t + P.V of tile t-1: 14
// MFMAs at d = 40, ~225 ns of matrix pipe) and a vector segment (online softmax of tile t: 32 v_exp_f32 + max / fma /
// convert, ~230 ns of VALU), and inside one wave they are strictly dependent.  In attn_kernel all eight waves of a
// workgroup pass the per-tile barrier together, so the two waves that share a SIMD sit in the SAME segment at the same
// time: first they contend for the matrix pipe while the VALU idles, then the other way round — the kernel ran at the SUM
// of the two (PMC: VALU busy 67 % + MFMA busy 45 %; tools/exp/coexec.hip shows that an MFMA stream and a v_exp stream of
// two different waves on one SIMD take 1.08x the longer of the two, not the sum).  Here waves 4-7 run one segment behind
// waves 0-3 (a workgroup's waves w and w + 4 land on the same SIMD): in every barrier interval one wave of each SIMD is in
// its matrix segment and the other in its vector segment.
//
//   interval i (one s_barrier each):   group g = wave / 4 works on phase p = i - g:
//       p = 2 t      matrix segment t : S_t = K_t Q^T  (t < ntiles),  O += V_{t-1}^T P_{t-1}^T  (t >= 1)
//       p = 2 t + 1  vector segment t : P_t = exp2(S_t sc - m sc), running max / rescale of O
//   (every wave runs the same loop body [matrix, barrier, vector, barrier]; group 1 passes one extra barrier first)
//   The operand fragments of matrix segment t + 1 are read from LDS at the end of vector segment t.
//   DMA (all waves, start of odd intervals 2 u + 1): K_{u+2} and V_{u+1} into the slots whose last fragment reads were in
//   interval 2 u; waited for (vmcnt(0), then the barrier) at the end of interval 2 u + 2.  Two K slots + two V slots: the
//   same 32 KB (d = 40) as attn_kernel.
#include "common.h"

namespace {

__device__ __attribute__((aligned(64))) char g_attnpp_zero_page[64];

__device__ __forceinline__ bf16x8 tr_pair_pp(const char* p, int second_off) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(p + second_off));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, r);
}

template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 8))) void attn_pp_kernel(const CcAttnDesc a) {
    constexpr int NW = 8;
    constexpr int KS = (D + 15) / 16;         // QK^T k-steps
    constexpr int NT = (D + 31) / 32;         // O^T row tiles
    constexpr int DK = D <= 64 ? 64 : 128;    // LDS row widths (elements), power of two for the XOR swizzles
    constexpr int DV = DK;
    constexpr int GK = DK / 8, GV = DV / 8;
    constexpr int KB = 64 * DK * 2, VB = 64 * DV * 2;
    constexpr int NTHR = NW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sK = smem;                // [2][KB]
    char* const sV = smem + 2 * KB;       // [2][VB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp2 = wave >> 2;           // 0: leads, 1: one segment behind
    const int l31 = lane & 31, hi = lane >> 5;
    const int qtiles = (a.Lq + NW * 32 - 1) / (NW * 32);
    const int xcd = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    const int grp = (local / qtiles) * 8 + xcd;            // (batch, head) group, all its query tiles on one XCD
    if (grp >= a.batches * a.heads) return;
    const int batch = grp / a.heads;
    const int head = grp - batch * a.heads;
    const int q0 = (local % qtiles) * (NW * 32) + wave * 32;

    const bf16* zp = (const bf16*)g_attnpp_zero_page;
    const int64_t qbase = (int64_t)(batch / a.q_inner) * a.q_outer_rows + (int64_t)(batch % a.q_inner) * a.q_inner_rows;
    const int kvb = batch / a.kv_div;
    const int64_t kvbase = (int64_t)(kvb / a.kv_inner) * a.kv_outer_rows + (int64_t)(kvb % a.kv_inner) * a.kv_inner_rows;
    const bf16* __restrict__ Q = (const bf16*)a.q + head * D;
    const bf16* __restrict__ K = (const bf16*)a.k + head * D;
    const bf16* __restrict__ V = (const bf16*)a.v + head * D;

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

    // ---- DMA plan (as attn_kernel): fixed (row, granule) slots per thread, K and V advance independently ----
    constexpr int ITK = (64 * GK + NTHR - 1) / NTHR, ITV = (64 * GV + NTHR - 1) / NTHR;
    int krw[ITK], vrw[ITV], kg[ITK], vg[ITV];
#pragma unroll
    for (int it = 0; it < ITK; ++it) {
        const int idx = it * NTHR + tid;
        const int row = idx / GK;
        int g = idx - row * GK;
        g ^= (GK == 8) ? ((row >> 1) & 7) : (row & 15);               // LDS slot -> source granule
        krw[it] = ((idx < 64 * GK) && (g * 8 < D)) ? row : -1;        // pad granules are never DMA'd
        kg[it] = g * 8;
    }
#pragma unroll
    for (int it = 0; it < ITV; ++it) {
        const int idx = it * NTHR + tid;
        const int row = idx / GV;
        int g = idx - row * GV;
        g ^= ((row >> 1) & 1) << 2;                                    // rows r, r+2 -> different bank halves
        vrw[it] = ((idx < 64 * GV) && (g * 8 < D)) ? row : -1;
        vg[it] = g * 8;
    }
    int64_t seg1base = 0;
    if (a.seg1_len > 0) {
        const int sb = (batch / a.seg1_div) * a.seg1_mul + a.seg1_add;
        seg1base = (int64_t)(sb / a.kv_inner) * a.kv_outer_rows + (int64_t)(sb % a.kv_inner) * a.kv_inner_rows;
    }
    // source row (in units of rows of the k / v matrices) of key `kv`; keys [0, seg1_len) come from the leading segment
    auto kv_row = [&](int kv) -> int64_t {
        return (kv < a.seg1_len) ? seg1base + (int64_t)kv * a.kv_seq_rows : kvbase + (int64_t)(kv - a.seg1_len) * a.kv_seq_rows;
    };
    auto stageK = [&](int j) {
#pragma unroll
        for (int it = 0; it < ITK; ++it)
            if (krw[it] >= 0) {
                const int kv = j * 64 + krw[it];
                const bf16* src = (kv < a.Lk) ? K + (size_t)kv_row(kv) * a.ldk + kg[it] : zp;
                glds16(src, sK + (j & 1) * KB + (it * NTHR + wave * 64) * 16);
            }
    };
    auto stageV = [&](int j) {
#pragma unroll
        for (int it = 0; it < ITV; ++it)
            if (vrw[it] >= 0) {
                const int kv = j * 64 + vrw[it];
                const bf16* src = (kv < a.Lk) ? V + (size_t)kv_row(kv) * a.ldv + vg[it] : zp;
                glds16(src, sV + (j & 1) * VB + (it * NTHR + wave * 64) * 16);
            }
    };

    f32x16 o[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sc = a.scale * 1.4426950408889634f;   // fold log2(e): softmax via exp2
    const int krow_l = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int i16 = lane & 15, dvhalf = (lane >> 4) & 1;
    const int ntiles = (a.Lk + 63) / 64;

    // zero, once, the K pad granules the last QK^T k-step reads; put the ones column at V[:, D] (softmax denominator from
    // the PV MFMA) — both slots of both rings; the DMA never writes these granules
    if constexpr (KS * 2 > (D + 7) / 8) {
        constexpr int NPAD = KS * 2 - (D + 7) / 8;
        for (int idx = tid; idx < 2 * 64 * NPAD; idx += NTHR) {
            const int b = idx / (64 * NPAD), rem = idx - b * 64 * NPAD;
            const int row = rem / NPAD, g = (D + 7) / 8 + (rem - row * NPAD);
            const int slot = g ^ ((GK == 8) ? ((row >> 1) & 7) : (row & 15));
            *(u32x4*)(sK + b * KB + row * (DK * 2) + slot * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    constexpr bool MFMA_ROWSUM = (D % 32 != 0);
    if constexpr (MFMA_ROWSUM) {
        for (int idx = tid; idx < 2 * 64; idx += NTHR) {
            const int b = idx >> 6, row = idx & 63;
            const int g = D / 8;
            const int slot = g ^ (((row >> 1) & 1) << 2);
            *(u32x4*)(sV + b * VB + row * (DV * 2) + slot * 16) = u32x4{0x00003F80u, 0u, 0u, 0u};   // bf16 {1,0,0,...}
        }
    }
    stageK(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x16 s[2];                 // S^T of the tile in flight; after the vector segment: the probabilities
    bf16x8 pf[4];                // P of the previous tile, packed for the PV MFMAs

    // Every wave runs the same straight-line loop body (matrix segment, barrier, vector segment, barrier) — no divergent
    // arms to merge the accumulators across; group 1 is shifted by one barrier at the start (group 0 pays it back at the
    // end).  The operand fragments of matrix segment t + 1 (K_{t+1} rows, V_t transposed) are read from LDS at the END of
    // vector segment t, while the partner wave owns the matrix pipe: a matrix segment is then 14 back-to-back MFMAs with no
    // LDS latency in it.  DMA: the tiles (K_{u+2}, V_{u+1}) are issued by all waves where interval 2 u + 1 begins for them
    // and waited for at the end of interval 2 u + 2; their slots were last read in interval 2 u.
    bf16x8 kfr[2][KS], vfr[4][NT];
    auto issue = [&](int u) {                 // interval 2 u + 1: K_{u+2} and V_{u+1}
        if (u + 2 < ntiles) stageK(u + 2);
        if (u + 1 < ntiles) stageV(u + 1);
    };
    auto load_frags = [&](int t) {            // operands of matrix segment t: K_t (t < ntiles) and V_{t-1} (t >= 1)
        if (t < ntiles) {
            const char* kb = sK + (t & 1) * KB;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const int krow = t2 * 32 + krow_l;
                const char* kr = kb + krow * (DK * 2);
                const int ksw = (GK == 8) ? ((krow >> 1) & 7) : (krow & 15);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kfr[t2][ks] = *(const bf16x8*)(kr + (((ks * 2 + hi) ^ ksw) << 4));
            }
        }
        if (t >= 1) {
            const char* vb = sV + ((t - 1) & 1) * VB;
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const int vrow = 16 * sp + 8 * hi + (i16 >> 2);
                const char* vr = vb + vrow * (DV * 2) + (dvhalf * 16 + (i16 & 3) * 4) * 2;
                const int vsw = ((vrow >> 1) & 1) << 6;
#pragma unroll
                for (int n = 0; n < NT; ++n) vfr[sp][n] = tr_pair_pp(vr + ((n * 64) ^ vsw), 4 * DV * 2);
            }
        }
    };
    // prologue ("interval -1"): K_0 has landed; K_1 and V_0 in flight (waited for at the end of interval 0)
    if (1 < ntiles) stageK(1);
    stageV(0);
    load_frags(0);
    if (grp2 == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // interval 0 belongs to group 0's matrix segment 0
    }
    for (int t = 0; t <= ntiles; ++t) {
        if (grp2 == 1) issue(t);              // group 1: its matrix segment t is interval 2 t + 1
        // ================= matrix segment t: S_t = K_t Q^T, O^T += V_{t-1}^T P_{t-1}^T =================
        if (t < ntiles) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[t2][ks], qf[ks], s[t2], 0, 0, 0);
            }
        }
        if (t >= 1) {
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
#pragma unroll
                for (int n = 0; n < NT; ++n) o[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[sp][n], pf[sp], o[n], 0, 0, 0);
        }
        if (grp2 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // end of an even interval: K_{t+1}, V_t have landed
        __builtin_amdgcn_s_barrier();
        if (grp2 == 0) issue(t);              // group 0: its vector segment t is interval 2 t + 1
        // ================= vector segment t: online softmax of S_t -> P_t =================
        if (t < ntiles) {
            if (t * 64 + 64 > a.Lk) {          // wave-uniform: only the last KV tile has masked columns
                asm volatile("; masked tile" ::: "memory");      // (keeps this a branch: if-converted it costs ~110 VALU per tile)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = t * 64 + 32 * t2 + 16 * (r >> 3) + 8 * hi + (r & 7);
                        if (kv >= a.Lk) s[t2][r] = -INFINITY;
                    }
            }
            float mt = s[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s[1][r]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {      // rescale only when some row's max moved
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
                    e = __builtin_elementwise_fma(e, f32x2{sc, sc}, f32x2{msc, msc});
                    const float p0 = __builtin_amdgcn_exp2f(e[0]), p1 = __builtin_amdgcn_exp2f(e[1]);
                    s[t2][r] = p0;
                    s[t2][r + 1] = p1;
                    if constexpr (!MFMA_ROWSUM) psum += p0 + p1;
                }
            if constexpr (!MFMA_ROWSUM) l_run += psum;
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[sp][e] = f2bf(s[sp >> 1][8 * (sp & 1) + e]);
            load_frags(t + 1);                // K_{t+1}, V_t: landed one barrier ago for this wave
        }
        if (grp2 == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // end of an even interval (group 1's vector segment)
        __builtin_amdgcn_s_barrier();
    }
    if (grp2 == 0) __builtin_amdgcn_s_barrier();      // group 1's last matrix segment

    // ---- normalise and store: lane holds O^T[dv = 32 n + (r&3) + 8 (r>>2) + 4 hi][q = l31] ----
    float l_tot;
    if constexpr (MFMA_ROWSUM) {
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

template <int D>
int launch_pp(const CcAttnDesc& a, hipStream_t s) {
    constexpr int DK = D <= 64 ? 64 : 128;
    constexpr int lds = 4 * 64 * DK * 2;
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)attn_pp_kernel<D>, lds, &attr_done, "attn_pp")) return rc;
    const int64_t qtiles = (a.Lq + 255) / 256;
    const int64_t groups = ((int64_t)a.batches * a.heads + 7) / 8 * 8;
    hipLaunchKernelGGL((attn_pp_kernel<D>), dim3((unsigned)(qtiles * groups)), dim3(512), lds, s, a);
    return cc_launch_status("attn_pp_kernel");
}

}  // namespace

// long query AND key sequences, d = 40 / 80 (the 64x96 and 32x48 levels), not causal
bool cc_attn_pp_applicable(const CcAttnDesc& a) {
    return (a.d == 40 || a.d == 80) && a.Lq >= 1024 && a.Lk >= 256 && !a.causal;
}

int cc_attn_pp_launch(const CcAttnDesc& a, hipStream_t s) {
    return a.d == 40 ? launch_pp<40>(a, s) : launch_pp<80>(a, s);
}
