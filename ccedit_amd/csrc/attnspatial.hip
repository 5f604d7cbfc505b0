// Spatial self-attention with a small head dimension (d = 40: the 6144-token attention of the 64x96 level, 8 heads x 34 frames;
// reference: CrossAttention.forward, sgm/modules/attention.py:392-467, F.scaled_dot_product_attention at :446) — gfx950.
//
// The general kernel (attention.hip) is VALU-bound at d = 40: per 64-key tile and wave 14 MFMAs (448 matrix-pipe cycles) against
// 121 VALU instructions (667 cycles) — one v_pk_fma_f32 per score pair to apply scale and running max, a rescale of the 32 output
// accumulators on most tiles (ANY of a wave's 32 rows moving its max triggers it: ~70 % of the tiles of a 6144-key row), ~30
// instructions of pointer / zero-page selects around the four DMA requests.  This kernel removes instructions instead of
// re-arranging them:
//   * softmax scale * log2(e) arrives folded into the to_q weights (CcAttnDesc.flags: CCEDIT_ATTN_Q_LOG2 — what the network does),
//     so a score leaves the MFMA in log2 units; without the flag one multiply per score remains (see QLOG2 below);
//   * the running reference m~ of a query row enters THROUGH THE MFMA: d = 40 pads to three 16-deep k-steps, column 40 of the
//     staged K tile is a constant 1 and element 40 of the lane's query fragment holds -m~ (a bf16-representable value — softmax is
//     invariant to the reference as long as numerator and denominator use the same one).  S^T = K Q^T then IS s - m~ with the
//     accumulators started from the inline constant 0: no per-score subtract, p = v_exp_f32(s) directly;
//   * the reference moves only when it must: a tile takes the update path when some score exceeds the reference by more than
//     2^16 (or on the first tile); otherwise nothing is rescaled.  bf16 P and fp32 accumulators have eight exponent bits, so a
//     common factor <= 2^16 on a row's P changes no relative precision; the denominator comes from the same bf16 P through a ones
//     column of V (row 40 of O^T), so numerator and denominator stay consistent whatever the reference is;
//   * K / V tiles stream through a THREE-slot LDS ring: the requests of tile j+2 are issued at the top of tile j and waited for
//     with a counted vmcnt two tiles later; one workgroup barrier per tile; base addresses advance in SGPRs, LDS slots are
//     immediates of a loop unrolled by three — no address arithmetic in the loop;
//   * the tile with masked keys (Lk % 64 != 0) is a separate instantiation: the main loop has no selects.
// Same tile geometry, LDS image, K-row permutation and V transpose reads as attention.hip (see there).
//
// d = 80 (the 1536-token attention of the 32x48 level; round 5): the same body on 256-byte rows.  80 is a multiple of 16, so the
// reference takes a k-step of its own whose K fragment is a register constant (column 80 = 1: no LDS read); three O^T row tiles
// (80 value rows + the denominator row) in 32x32x16 tiles; ring of three 32 KB slots = one workgroup per CU, 157 registers, two
// waves per SIMD.  Per wave and 64-key tile: 24 MFMAs (768 matrix-pipe cycles), ~24 KB of LDS fragment reads (768 cycles of the
// CU's 128 B / clk shared by eight waves) and the same 84 VALU instructions as d = 40 — three co-critical resources; measured
// 308 us against 387 us on the general kernel for 34 x 8 x 1536^2 (667 against 531 TFLOP/s).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __attribute__((aligned(64))) char g_as_zero_page[64];

constexpr float kThr = 16.0f;          // log2 units: the reference is raised when a score exceeds it by more than this
// OPTIMISTIC (round 6, policy attn_opt): only the FIRST key tile looks at its scores.  The reference of a query row is set to that
// tile's row maximum + kOptMargin and never moves; the 16 v_max3 + compare + ballot + branch of every later tile are gone (19 of 84
// VALU instructions per tile on a kernel whose VALU port is the critical resource).  bf16 P and the fp32 accumulators have eight
// exponent bits: a row whose later scores exceed the first tile's maximum by up to ~2^(127 - margin) keeps every relative precision
// (numerator and denominator carry the same factor), and terms 2^-126 below the reference are 2^-94 of the row's largest term.  A
// row outside that window produces inf / NaN in its accumulators: every workgroup checks its accumulators after the key loop and,
// if any lane saw a non-finite value (or an empty denominator), REPEATS its key loop with the exact per-tile tracking above — the
// result is then bit-identical to policy attn_opt = 0.  tests/test_ops_gpu.py drives both outcomes (spikes of 2^9 log2 units).
constexpr float kOptMargin = 32.0f;

template <int N>
__device__ __forceinline__ void as_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void as_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// V^T fragments by ds_read_b64_tr_b16 in inline asm.  Through the builtin, hipcc (ROCm 7.2) treats the transpose read as a load that
// may alias every LDS-DMA in flight and puts `s_waitcnt vmcnt(0)` in front of the first one of each tile — the K / V requests of the
// tiles ahead were drained in the middle of every tile (attention.hip has that wait).  In asm the reads are invisible to the
// compiler's counters: `as_tr_wait` is the wait (LDS returns in order: "at most N outstanding" retires everything older) and ties the
// destination registers to it, so no consumer can be scheduled above it.
struct VFrag {
    u32x2 lo, hi;
};
template <int OFF, int ROW4 = 512>          // ROW4: bytes between tile rows r and r + 4 (the second half of the fragment)
__device__ __forceinline__ void as_tr_issue(VFrag& f, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
                 : "=&v"(f.lo), "=&v"(f.hi)
                 : "v"(addr), "n"(OFF), "n"(OFF + ROW4));
}
template <int N>
__device__ __forceinline__ void as_tr_wait(VFrag& a, VFrag& b) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi) : "n"(N));
}
template <int N>
__device__ __forceinline__ void as_tr_wait3(VFrag& a, VFrag& b, VFrag& c) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi), "+v"(c.lo), "+v"(c.hi) : "n"(N));
}
__device__ __forceinline__ bf16x8 as_frag(const VFrag& f) {
    return __builtin_bit_cast(bf16x8, u32x4{f.lo[0], f.lo[1], f.hi[0], f.hi[1]});
}

// smallest bf16-representable value >= x (x finite)
__device__ __forceinline__ float bf16_ceil(float x) {
    const uint32_t u = __float_as_uint(x);
    const uint32_t r = (x >= 0.f) ? ((u + 0xFFFFu) & 0xFFFF0000u) : (u & 0xFFFF0000u);
    return __uint_as_float(r);
}

// hipcc forms v_max3_f32 from this itself — and pads the MFMA-result -> VALU-read wait states, which it does NOT do for an operand
// of an inline-asm v_max3_f32 (first version of this kernel: the maxima were read while the MFMA was still writing its tile)
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

constexpr int kNW = 8, kNT = kNW * 64;
constexpr int kSlots = 3;
// LDS image per head dimension: d = 40 in rows of 8 granules (128 B: 40 channels + the pad granule), d = 80 in rows of 16 (256 B)
template <int D>
struct AsGeo {
    static constexpr bool WIDE = D > 56;
    static constexpr int RB = WIDE ? 256 : 128;                  // bytes per tile row
    static constexpr int KB = 64 * RB, SLOT = 2 * KB;            // K tile, ring slot (K + V): 16 KB / 32 KB
    static constexpr int LDS = kSlots * SLOT;
};

// QLOG2: q arrives in log2 units (CCEDIT_ATTN_Q_LOG2) — scores leave the MFMA ready for v_exp_f32.  Otherwise q is used as it is
// (scaling the bf16 fragments in here would round q a second time: at logits of +-70 that alone is several per cent of a
// probability), reference and threshold are kept in raw q.k units and every score is multiplied by scale * log2(e) on its way into
// the exponential: 32 more VALU instructions per tile, the price of not packing the scale into the weights.
template <int D, bool QLOG2, bool PV16, bool OPTIMISTIC>
__device__ __forceinline__ void attn_spatial_body(const CcAttnDesc& a) {
    static_assert(D % 8 == 0 && D % 32 != 0 && D <= 112, "the reference rides in a pad column of the last k-step; the denominator in a spare O^T row");
    // d = 40: the last k-step holds 8 channels + the reference column (element 0 of the HI lanes' fragment) + 7 zeros.
    // d = 80 (a multiple of 16): one more k-step that holds nothing but the reference column (element 0 of the LO lanes' fragment):
    // 12 MFMAs instead of 10 for S^T, which still costs less than 32 subtractions on the VALU this kernel is bound by.
    constexpr bool REF_HI = D % 16 == 8;
    constexpr int KS = D / 16 + 1;            // QK^T k-steps
    constexpr int NT = (D + 1 + 31) / 32;     // O^T row tiles (D value rows + the denominator row D)
    constexpr int PADG = D / 8;               // granule of columns [D, D+8): K: {1, 0...}, V: {1, 0...}
    using G = AsGeo<D>;
    constexpr bool WIDE = G::WIDE;
    constexpr int RB = G::RB, kKB = G::KB, kSlot = G::SLOT;
    static_assert(!(WIDE && PV16), "the 16x16x32 PV image is laid out for 128-byte rows");
    // ring slot s: narrow rows [K s | V s] back to back; wide rows [K 0 | K 1 | K 2 | V 0 | V 1 | V 2] — the transpose reads take their
    // slot as a 16-bit instruction offset from ONE base register, and 2 x 32 KB + a k-step would not fit it
    constexpr int VBASE = WIDE ? kSlots * kKB : kKB;              // first V tile
    constexpr int SSTEP = WIDE ? kKB : kSlot;                     // bytes from a slot's K (V) tile to the next slot's
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // block order: all query tiles of one (batch, head) on ONE XCD back to back (its K / V stay in that L2) — attention.hip
    const int qtiles = (a.Lq + kNW * 32 - 1) / (kNW * 32);
    const int xcd = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    const int grp = (local / qtiles) * 8 + xcd;
    const int qt = local % qtiles;
    if (grp >= a.batches * a.heads) return;
    const int batch = grp / a.heads;
    const int head = grp - batch * a.heads;
    const int q0 = qt * (kNW * 32) + wave * 32;

    const bf16* zp = (const bf16*)g_as_zero_page;
    const int64_t qbase = (int64_t)(batch / a.q_inner) * a.q_outer_rows + (int64_t)(batch % a.q_inner) * a.q_inner_rows;
    const int kvb = batch / a.kv_div;
    const int64_t kvbase = (int64_t)(kvb / a.kv_inner) * a.kv_outer_rows + (int64_t)(kvb % a.kv_inner) * a.kv_inner_rows;
    const bf16* __restrict__ Q = (const bf16*)a.q + head * D;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0 + l31][16 ks + 8 hi .. +8], scaled into log2 units ----
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
    const float sc = QLOG2 ? 1.0f : a.scale * 1.4426950408889634f;      // score units -> log2 units
    const float thr = QLOG2 ? kThr : kThr / sc;                          // the threshold in score units
    const float margin = QLOG2 ? kOptMargin : kOptMargin / sc;          // OPTIMISTIC: head-room above the first tile's maximum, score units
    // the reference column: element 0 of the last k-step's fragment on the hi lanes is column D
    auto set_ref = [&](float ref) {
        u32x4 w = __builtin_bit_cast(u32x4, qf[KS - 1]);
        if ((hi != 0) == REF_HI) w[0] = __float_as_uint(-ref) >> 16;
        qf[KS - 1] = __builtin_bit_cast(bf16x8, w);
    };

    // ---- DMA plan: thread -> (row, granule) of the 64-row K and V tiles (lane-linear LDS image: piece it * 512 + tid) ----
    // narrow rows (8 granules): one piece per thread, wave w fills rows 8w..8w+7; wide rows (16 granules): two, rows 4w..4w+3 and 32 more
    constexpr int IT = WIDE ? 2 : 1;
    const int drow = WIDE ? tid >> 4 : tid >> 3;
    const int dslot = WIDE ? tid & 15 : tid & 7;
    const int gk = dslot ^ (WIDE ? (drow & 15) : ((drow >> 1) & 7));     // LDS slot -> source granule (XOR swizzle, attention.hip)
    // PV16: V rows are swizzled in 32-byte chunks (one 16-channel MFMA row tile each): chunk c of row r at c ^ vsw16(r)
    auto vsw16 = [](int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 1); };
    const int gv = PV16 ? (((((tid & 7) >> 1) ^ vsw16(drow)) << 1) | (tid & 1)) : (dslot ^ (((drow >> 1) & 1) << 2));
    const bool use_k = gk * 8 < D, use_v = gv * 8 < D;         // pad granules are never written by the DMA
    uint32_t koff[IT], voff[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {                            // (row + 32 has the same swizzle bits)
        koff[it] = (uint32_t)((int64_t)(drow + 32 * it) * a.kv_seq_rows * a.ldk + gk * 8) * 2u;
        voff[it] = (uint32_t)((int64_t)(drow + 32 * it) * a.kv_seq_rows * a.ldv + gv * 8) * 2u;
    }
    const int64_t kstep = 64 * a.kv_seq_rows * (int64_t)a.ldk * 2, vstep = 64 * a.kv_seq_rows * (int64_t)a.ldv * 2;   // bytes per tile
    const char* const kreg = (const char*)((const bf16*)a.k + head * D + kvbase * a.ldk);
    const char* const vreg = (const char*)((const bf16*)a.v + head * D + kvbase * a.ldv);
    // optional leading segment (anchor-frame keys, a whole number of tiles): its tiles come from another kv batch
    const int seg_tiles = a.seg1_len >> 6;
    const char* kseg = kreg;
    const char* vseg = vreg;
    if (a.seg1_len > 0) {
        const int sb = (batch / a.seg1_div) * a.seg1_mul + a.seg1_add;
        const int64_t sbase = (int64_t)(sb / a.kv_inner) * a.kv_outer_rows + (int64_t)(sb % a.kv_inner) * a.kv_inner_rows;
        kseg = (const char*)((const bf16*)a.k + head * D + sbase * a.ldk);
        vseg = (const char*)((const bf16*)a.v + head * D + sbase * a.ldv);
    }
    const int ntiles = (a.Lk + 63) / 64;
    char* const lds_wave = smem + wave * 1024;

    const char* kcur = seg_tiles > 0 ? kseg : kreg;            // tiles are requested in order: running (scalar) pointers
    const char* vcur = seg_tiles > 0 ? vseg : vreg;
    auto stage = [&](int t, int slot) {
        if (t == seg_tiles && t > 0) {
            kcur = kreg;
            vcur = vreg;
        }
        const char* kb = kcur;
        const char* vb = vcur;
        kcur += kstep;
        vcur += vstep;
        if ((t + 1) * 64 <= a.Lk) {                             // wave-uniform
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                if (use_k) glds16(kb + koff[it], lds_wave + slot * SSTEP + it * 8192);
                if (use_v) glds16(vb + voff[it], lds_wave + VBASE + slot * SSTEP + it * 8192);
            }
        } else {                                                // the tile that reaches past Lk: those rows come from the zero page
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const bool valid = t * 64 + drow + 32 * it < a.Lk;
                if (use_k) glds16(valid ? (const void*)(kb + koff[it]) : (const void*)zp, lds_wave + slot * SSTEP + it * 8192);
                if (use_v) glds16(valid ? (const void*)(vb + voff[it]) : (const void*)zp, lds_wave + VBASE + slot * SSTEP + it * 8192);
            }
        }
    };

    // ---- per-lane LDS read addresses (slot and tile-row offsets are immediates) ----
    // A-row i of an S^T tile reads K row swap23(i): a lane's 8 consecutive S^T registers are 8 consecutive keys
    // (PV16: A-row i reads K row 16 (i>>2 & 1) + (i & 3) + 4 (i >> 3): register r of lane (q, hi) is key 32 t2 + 16 hi + r — sixteen
    //  consecutive keys per lane, which two v_permlane16_swap per register pair turn into the 16x16x32 B operand)
    const int krow_l = PV16 ? (((l31 >> 2) & 1) << 4) + (l31 & 3) + ((l31 >> 3) << 2) : ((l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1));
    const int ksw = WIDE ? (krow_l & 15) : ((krow_l >> 1) & 7);
    const char* kaddr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kaddr[ks] = smem + krow_l * RB + (((ks * 2 + hi) ^ ksw) << 4);
    const int i16 = lane & 15, dvhalf = (lane >> 4) & 1;
    const int vsw = ((i16 >> 3) & 1) << 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    uint32_t vaddr[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) vaddr[n] = lds0 + VBASE + (8 * hi + (i16 >> 2)) * RB + (dvhalf * 16 + (i16 & 3) * 4) * 2 + ((n * 64) ^ vsw);
    // PV16: A operand of the 16x16x32 product = V^T[dv = 16 m + (lane & 15)][kv = 32 ks + 8 g + e]: lane group g transposes the 4-key x
    // 16-channel blocks at rows 8 g + (0..3) and 8 g + 4 + (0..3) of channel chunk m
    constexpr int NM = (D + 1 + 15) / 16;     // 16-channel row tiles of O^T (D value rows + the denominator row D)
    const int g4 = lane >> 4;
    uint32_t vaddr16[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m)
        vaddr16[m] = lds0 + kKB + (8 * g4 + (i16 >> 2)) * 128 + (((m ^ (((i16 >> 3) & 1) | ((g4 & 1) << 1))) << 5) + (i16 & 3) * 8);

    // ---- constant pad granules of all three slots: K[:, D] = 1 (the reference column), V[:, D] = 1 (the denominator row) ----
    for (int idx = tid; idx < kSlots * 64; idx += kNT) {
        const int sl = idx >> 6, row = idx & 63;
        const int rsw = WIDE ? (row & 15) : ((row >> 1) & 7);
        if (REF_HI) *(u32x4*)(smem + sl * SSTEP + row * RB + ((PADG ^ rsw) << 4)) = u32x4{0x00003F80u, 0u, 0u, 0u};      // (d = 80: the reference k-step's K fragment is a register constant)
        const int vslot = PV16 ? ((((PADG >> 1) ^ vsw16(row)) << 1) | (PADG & 1)) : (PADG ^ (((row >> 1) & 1) << 2));
        *(u32x4*)(smem + VBASE + sl * SSTEP + row * RB + (vslot << 4)) = u32x4{0x00003F80u, 0u, 0u, 0u};
    }
    // K columns (D, 16 KS) beyond the reference column are zero in that same granule (and the one written above); granules above
    // are never read.

    f32x16 o[PV16 ? 1 : NT];
    f32x4 o16[PV16 ? NM : 1][2];              // PV16: O^T[dv = 16 m + 4 g + reg][q = 16 t + (lane & 15)]
    float mref = 0.f;
    // (re)start of a key loop: zero accumulators, reference 0, the first two tiles requested
    auto begin = [&]() {
#pragma unroll
        for (int n = 0; n < (PV16 ? 1 : NT); ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[n][r] = 0.f;
#pragma unroll
        for (int m = 0; m < (PV16 ? NM : 1); ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) o16[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        mref = 0.f;
        set_ref(0.f);
        kcur = seg_tiles > 0 ? kseg : kreg;
        vcur = seg_tiles > 0 ? vseg : vreg;
        stage(0, 0);
        if (ntiles > 1) {
            stage(1, 1);
            as_vmcnt<2>();
        } else {
            as_vmcnt<0>();
        }
        as_barrier();
    };

    const bool tail_masked = (a.Lk & 63) != 0;
    auto tile = [&](auto SLOTC, auto OPTC, int j) {
        constexpr int SLOT = decltype(SLOTC)::value;
        constexpr bool OPT = decltype(OPTC)::value;
        constexpr int SB = SLOT * SSTEP;

        // ---- S^T = K Q^T - m~ for the 64 keys of this tile, log2 units ----
        f32x16 s[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 kf;
                if (!REF_HI && ks == KS - 1)      // the k-step that holds only the reference column: K[:, D] = 1, a constant — no LDS read
                    kf = __builtin_bit_cast(bf16x8, u32x4{hi ? 0u : 0x00003F80u, 0u, 0u, 0u});
                else
                    kf = *(const bf16x8*)(kaddr[ks] + SB + t2 * (32 * RB));
                s[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[t2], 0, 0, 0);
            }
        }
        if (tail_masked && j == ntiles - 1) {          // wave-uniform: the tile that reaches past Lk
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = PV16 ? j * 64 + 32 * t2 + 16 * hi + r : j * 64 + 32 * t2 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (kv >= a.Lk) s[t2][r] = -INFINITY;
                }
        }
        // ---- does the reference have to move?  (this lane: query q0 + l31, keys 64 j + 32 t2 + 16 (r>>3) + 8 hi + (r&7)) ----
        // OPT: asked on the first tile only (wave-uniform scalar branch); its answer stands for the whole row
        auto track = [&]() {
            float m0 = max3f(s[0][0], s[0][1], s[0][2]), m1 = max3f(s[0][3], s[0][4], s[0][5]);
            float m2 = max3f(s[1][0], s[1][1], s[1][2]), m3 = max3f(s[1][3], s[1][4], s[1][5]);
#pragma unroll
            for (int r = 6; r < 16; r += 4) {
                m0 = max3f(m0, s[0][r], s[0][r + 1]);
                m2 = max3f(m2, s[1][r], s[1][r + 1]);
                if (r + 2 < 16) {
                    m1 = max3f(m1, s[0][r + 2], s[0][r + 3]);
                    m3 = max3f(m3, s[1][r + 2], s[1][r + 3]);
                }
            }
            float mt = fmaxf(max3f(m0, m1, m2), m3);
            if (OPT || j == 0 || __builtin_amdgcn_ballot_w64(mt > thr) != 0) {          // wave-uniform
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));        // row maximum (both lane halves of a query)
                float nref = bf16_ceil(OPT ? mref + mt + margin : mref + mt);
                if (j != 0) nref = fmaxf(nref, mref);          // later tiles only raise it
                const float delta = nref - mref;
                if (!OPT && j != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(QLOG2 ? -delta : -delta * sc);
                    if constexpr (PV16) {
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float at = __shfl(alpha, 16 * t + i16, 64);      // the factor of query 16 t + (lane & 15)
#pragma unroll
                            for (int m = 0; m < NM; ++m) o16[m][t] *= at;
                        }
                    } else {
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[n][r] *= alpha;
                    }
                }
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[t2][r] -= delta;
                mref = nref;
                set_ref(mref);
            }
        };
        if constexpr (OPT) {
            if (j == 0) track();
        } else {
            track();
        }
        // ---- p = 2^s;  O^T += V^T P^T (row D of O^T: the denominator, from the ones column of V) ----
        // k-step sp covers keys 16 sp .. 16 sp + 15; its V^T fragments are requested two k-steps ahead of their MFMAs
        if constexpr (PV16) {
            // 16x16x32 tiles: k-step t2 covers keys 32 t2 .. + 31; 3 channel tiles x 2 query tiles = 6 MFMAs of 16 cycles per k-step
            // (192 cycles per tile against the 256 of four 32x32x16 k-steps over a half-empty second channel tile)
            static_assert(NM == 3, "fragment bookkeeping below");
            VFrag vf[2][NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) as_tr_issue<SB>(vf[0][m], vaddr16[m]);
#pragma unroll
            for (int m = 0; m < NM; ++m) as_tr_issue<SB + 4096>(vf[1][m], vaddr16[m]);
            auto pexp16 = [&](int t2, bf16x8& b0, bf16x8& b1) {
                uint32_t pk[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float x0 = s[t2][2 * jj], x1 = s[t2][2 * jj + 1];
                    const bf16x2 pr = {f2bf(__builtin_amdgcn_exp2f(QLOG2 ? x0 : x0 * sc)), f2bf(__builtin_amdgcn_exp2f(QLOG2 ? x1 : x1 * sc))};
                    pk[jj] = __builtin_bit_cast(uint32_t, pr);
                }
                u32x4 w0, w1;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    // rows (16 lanes) of the wave: [q 0-15 hi 0 | q 16-31 hi 0 | q 0-15 hi 1 | q 16-31 hi 1]; the swap exchanges the
                    // odd rows of keys 2jj.. with the even rows of keys 8 + 2jj..: first result = query tile 0, second = query tile 1,
                    // each with keys 8 g + 2 jj, + 1 in lane group g — the B operand's k order
                    const auto r = __builtin_amdgcn_permlane16_swap(pk[jj], pk[4 + jj], false, false);
                    w0[jj] = r[0];
                    w1[jj] = r[1];
                }
                b0 = __builtin_bit_cast(bf16x8, w0);
                b1 = __builtin_bit_cast(bf16x8, w1);
            };
            auto pv16 = [&](int ks, const bf16x8& b0, const bf16x8& b1) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    o16[m][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[ks][m]), b0, o16[m][0], 0, 0, 0);
                    o16[m][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(vf[ks][m]), b1, o16[m][1], 0, 0, 0);
                }
            };
            bf16x8 b0, b1;
            pexp16(0, b0, b1);
            as_tr_wait3<6>(vf[0][0], vf[0][1], vf[0][2]);
            pv16(0, b0, b1);
            pexp16(1, b0, b1);
            as_tr_wait3<0>(vf[1][0], vf[1][1], vf[1][2]);
            pv16(1, b0, b1);
            return;
        }
        static_assert(NT == 2 || NT == 3, "fragment bookkeeping below");
        VFrag vf[4][NT];
        auto issue = [&](auto SPC) {
            constexpr int SP = decltype(SPC)::value;
#pragma unroll
            for (int n = 0; n < NT; ++n) as_tr_issue<SB + SP * (16 * RB), 4 * RB>(vf[SP][n], vaddr[n]);
        };
        auto wait = [&](auto SPC, auto PENDING) {          // the fragments of k-step SP are there when at most PENDING later reads are outstanding
            constexpr int SP = decltype(SPC)::value;
            constexpr int N = decltype(PENDING)::value * 2 * NT;
            if constexpr (NT == 2) as_tr_wait<N>(vf[SP][0], vf[SP][1]);
            else as_tr_wait3<N>(vf[SP][0], vf[SP][1], vf[SP][2]);
        };
        auto pexp = [&](int sp) {
            bf16x8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = s[sp >> 1][8 * (sp & 1) + e];
                pf[e] = f2bf(__builtin_amdgcn_exp2f(QLOG2 ? x : x * sc));
            }
            return pf;
        };
        auto pv = [&](int sp, bf16x8 pf) {
#pragma unroll
            for (int n = 0; n < NT; ++n) o[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(vf[sp][n]), pf, o[n], 0, 0, 0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        issue(I0{});
        issue(I1{});
        bf16x8 pf = pexp(0);
        wait(I0{}, I1{});
        pv(0, pf);
        issue(I2{});
        pf = pexp(1);
        wait(I1{}, I1{});
        pv(1, pf);
        issue(I3{});
        pf = pexp(2);
        wait(I2{}, I1{});
        pv(2, pf);
        pf = pexp(3);
        wait(I3{}, I0{});
        pv(3, pf);
    };
    // tile j: request tile j+2 (its slot was read in tile j-1, before the last barrier), compute, wait for tile j+1 (counted: the
    // requests just issued stay in flight across the barrier)
    auto step = [&](auto SLOTC, auto OPTC, int j) {
        constexpr int SL = decltype(SLOTC)::value;
        if (j + 2 < ntiles) stage(j + 2, (SL + 2) % kSlots);
        tile(SLOTC, OPTC, j);
        if (j + 2 < ntiles) as_vmcnt<2>();
        else as_vmcnt<0>();
        as_barrier();
    };
    auto keys = [&](auto OPTC) {
        begin();
        for (int j = 0; j < ntiles; j += 3) {
            step(std::integral_constant<int, 0>{}, OPTC, j);
            if (j + 1 < ntiles) step(std::integral_constant<int, 1>{}, OPTC, j + 1);
            if (j + 2 < ntiles) step(std::integral_constant<int, 2>{}, OPTC, j + 2);
        }
    };
    if constexpr (OPTIMISTIC) {
        keys(std::true_type{});
        // every accumulator finite, every denominator positive?  x * 0 is NaN for x = +-inf / NaN (no fast-math: not folded)
        float chk = 0.f;
        bool bad = false;
        if constexpr (PV16) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (16 * m < D + 8) chk = fmaf(o16[m][t][e], 0.f, chk);          // (see below: rows fed by unwritten V columns are skipped)
            if (g4 == (D % 16) / 4) bad = !(o16[D / 16][0][0] > 0.f) || !(o16[D / 16][1][0] > 0.f);
        } else {
            // (only rows the kernel owns: value rows, the denominator row and the zero rows of its pad granule — V columns beyond
            //  D + 8 are never written in LDS, and what the last O^T tile accumulates from them is never stored either)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * n + 8 * (r >> 2) < D + 8) chk = fmaf(o[n][r], 0.f, chk);
            if (hi == ((D % 32) / 4) % 2) bad = !(o[D / 32][((D % 32) / 8) * 4] > 0.f);
        }
        bad = bad || !(chk == 0.f);
        if (__syncthreads_or(bad ? 1 : 0)) keys(std::false_type{});          // workgroup-uniform: the rings and barriers are shared
    } else {
        keys(std::false_type{});
    }

    if constexpr (PV16) {
        // ---- lane holds O^T[dv = 16 m + 4 g + reg][q = 16 t + (lane & 15)]; row D = 40 (m 2, g 2, reg 0) is the denominator ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float l_t = __shfl(o16[D / 16][t][0], 16 * ((D % 16) / 4) + i16, 64);
            const float inv_t = 1.0f / l_t;
            const int qi = q0 + 16 * t + i16;
            if (qi < a.Lq) {
                bf16* orow = (bf16*)a.o + (size_t)(qbase + (int64_t)qi * a.q_seq_rows) * a.ldo + head * D;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int dv = 16 * m + 4 * g4;
                    if (dv < D) {
                        bf16x4 w = {f2bf(o16[m][t][0] * inv_t), f2bf(o16[m][t][1] * inv_t), f2bf(o16[m][t][2] * inv_t), f2bf(o16[m][t][3] * inv_t)};
                        *(bf16x4*)(orow + dv) = w;
                    }
                }
            }
        }
        return;
    }
    // ---- normalise and store: lane holds O^T[dv = 32 n + (r&3) + 8 (r>>2) + 4 hi][q = l31]; row D is the denominator ----
    const float l_tot = __shfl(o[D / 32][((D % 32) / 8) * 4], l31, 64);
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

// four waves per SIMD (two workgroups per CU): 2.04 ms against 2.30 ms with three on the 34 x 8 x 6144^2 launch
template <int D, bool QLOG2, bool PV16, bool OPTIMISTIC>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_spatial_kernel(const CcAttnDesc a) {
    attn_spatial_body<D, QLOG2, PV16, OPTIMISTIC>(a);
}
// d = 80: 96 KB of ring (one workgroup per CU) and 48 more accumulator registers: two waves per SIMD
template <int D, bool QLOG2, bool OPTIMISTIC>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_spatial_wide_kernel(const CcAttnDesc a) {
    attn_spatial_body<D, QLOG2, false, OPTIMISTIC>(a);
}

template <int D, bool QLOG2, bool PV16, bool OPT>
int launch_spatial(const CcAttnDesc& a, hipStream_t s) {
    constexpr int LDS = AsGeo<D>::LDS;
    const void* fn;
    if constexpr (AsGeo<D>::WIDE) fn = (const void*)attn_spatial_wide_kernel<D, QLOG2, OPT>;
    else fn = (const void*)attn_spatial_kernel<D, QLOG2, PV16, OPT>;
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds(fn, LDS, &attr_done, "attn_spatial")) return rc;
    const int64_t qtiles = (a.Lq + kNW * 32 - 1) / (kNW * 32);
    const int64_t groups = ((int64_t)a.batches * a.heads + 7) / 8 * 8;
    cc_note_kernel("attn_spatial_kernel d=%d", D);
    if constexpr (AsGeo<D>::WIDE)
        hipLaunchKernelGGL((attn_spatial_wide_kernel<D, QLOG2, OPT>), dim3((unsigned)(qtiles * groups)), dim3(kNT), LDS, s, a);
    else
        hipLaunchKernelGGL((attn_spatial_kernel<D, QLOG2, PV16, OPT>), dim3((unsigned)(qtiles * groups)), dim3(kNT), LDS, s, a);
    return cc_launch_status("attn_spatial_kernel");
}

}  // namespace

// policy attn_spatial: 1 = d 40 and d 80, 2 = d 40 only (the d = 80 A/B arm: 1536-key attention on the general flash kernel), 0 = off
bool cc_attn_spatial_applicable(const CcAttnDesc& a) {
    // per-lane byte offsets of a 64-row tile must fit 32 bits
    const int64_t span = 64 * a.kv_seq_rows * (int64_t)(a.ldk > a.ldv ? a.ldk : a.ldv) * 2;
    return (a.d == 40 || (a.d == 80 && cc_policy().attn_spatial == 1)) && a.Lq >= 1024 && a.Lk >= 192 && !a.causal && (a.seg1_len & 63) == 0 &&
           span < (1ll << 31);
}

int cc_attn_spatial_launch(const CcAttnDesc& a, hipStream_t s) {
    // policy attn_pv16 = 0: the PV product in 32x32x16 tiles (A/B; same sums in a different order: results differ in the last bit)
    // policy attn_opt = 0: the reference of every row tracked on every key tile (no optimistic pass; the A/B and test arm)
    const int pv16 = cc_policy().attn_pv16;
    const bool ql = (a.flags & CCEDIT_ATTN_Q_LOG2) != 0;
    if (cc_policy().attn_opt) {
        if (a.d == 80) return ql ? launch_spatial<80, true, false, true>(a, s) : launch_spatial<80, false, false, true>(a, s);
        if (pv16) return ql ? launch_spatial<40, true, true, true>(a, s) : launch_spatial<40, false, true, true>(a, s);
        return ql ? launch_spatial<40, true, false, true>(a, s) : launch_spatial<40, false, false, true>(a, s);
    }
    if (a.d == 80) return ql ? launch_spatial<80, true, false, false>(a, s) : launch_spatial<80, false, false, false>(a, s);
    if (pv16) return ql ? launch_spatial<40, true, true, false>(a, s) : launch_spatial<40, false, true, false>(a, s);
    return ql ? launch_spatial<40, true, false, false>(a, s) : launch_spatial<40, false, false, false>(a, s);
}
