// Cross-attention onto a SHORT key sequence (the 77 CLIP text tokens: attention.py:436-441 called from BasicTransformerBlock.attn2,
// attention.py:695-716) — gfx950.
//
//     O[row, h] = softmax(Q[row, h] K[clip, h]^T * scale) V[clip, h]        Lk <= 96, K / V shared by all query rows of a clip
//
// 20 GFLOP against 268 MB of Q in / O out at the 64x96 level: the launch is bound by streaming the query rows, not by arithmetic.
// The general kernel (attention.hip) gives one (frame, head) to a workgroup: every workgroup reads an 80-byte piece of each
// 640-byte query row and writes an 80-byte piece of each output row — every 128-byte line is touched by several workgroups, the
// K / V tile is re-staged per workgroup and the online-softmax loop runs for two masked key tiles (125 us, 2.1 TB/s).
// Here a wave owns 32 query ROWS and walks the heads of a 320-channel group (8 x 40, 4 x 80 or 2 x 160) itself:
//   * K (row-major, padded rows) and V^T (channel-major) of the group stay in LDS for all rows of the clip: 136 KB, staged once
//     per workgroup and clip;
//   * S^T = K Q^T with the query fragments loaded straight from global memory in MFMA B layout (16 bytes per lane and k-step; the
//     heads of a row are consecutive, so the lines a head leaves partly used are finished by the next one out of the L1);
//   * all keys at once: one max, one exp2 per score, no running rescale.  The rows of a K tile are read in the permuted order that
//     makes the score accumulators of a lane — taken in register order — the B operand of the second product (8 consecutive keys
//     per k-step), so P never leaves the registers;
//   * O^T = V^T P; the accumulators of a lane are 4-channel runs of ONE query row: a lane-half exchange (v_permlane32_swap) makes
//     them 8-channel runs, stored as 16 bytes.
#include "common.h"
#include <stdlib.h>

#ifndef AT_PROBE
#define AT_PROBE 0          // tuning builds only: 1 = no output stores, 2 = no query loads, 3 = no K / V staging, 4 = no softmax arithmetic
#endif

namespace {

constexpr int kG = 320;                 // channels per group
constexpr int kKeys = 96;               // padded key count: three 32-row MFMA tiles
constexpr int kKRow = (kG + 8) * 2;     // bytes per K row in LDS: 656 = 164 dwords (rows 36 banks apart), 8 zero pad channels
constexpr int kVRows = 352;             // V^T rows: 320 channels + the rows a last head's padded 32-row tiles reach into (zeros)
constexpr int kVRow = (kKeys + 8) * 2;  // bytes per V^T row: 208
constexpr int kLds = kKeys * kKRow + kVRows * kVRow;      // 62,976 + 73,216 = 136,192 B
constexpr int kNT = 512;

__device__ __forceinline__ void at_swap(float& x, float& y) {      // see g8_swap (gemm8p.hip)
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}

template <int D>
__global__ __launch_bounds__(kNT, 1) void attn_text_kernel(const CcAttnDesc a, int ngroups, int nkvb) {
    constexpr int HPG = kG / D;                     // heads per group
    constexpr int KS = (D + 15) / 16;               // k-steps of Q K^T
    constexpr int DT = (D + 31) / 32;               // 32-row tiles of O^T
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sK = smem;
    char* const sV = smem + kKeys * kKRow;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;

    // workgroup b runs on XCD b % 8; the 32 workgroups of an XCD = (32 / ngroups) row lanes x ngroups channel groups, so the groups
    // of a query row run beside each other on one L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int group = j % ngroups, qlane = j / ngroups;
    const int nlanes = (int)(gridDim.x >> 3) / ngroups;
    if (qlane >= nlanes) return;
    const int stride = nlanes * 8 * 8;              // 32-row tiles between two tiles of one wave
    const int first = (qlane * 8 + wave) * 8 + xcd;
    const int ch0 = group * kG;
    const float c2 = a.scale * 1.4426950408889634f;
    const int64_t rows_per_kvb = (int64_t)a.kv_div * a.Lq;

    // logical key of MFMA row m of a key tile: m = 8 a + 4 h + i  ->  16 (a >> 1) + 8 h + 4 (a & 1) + i
    const int krow = 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);

    // keys >= Lk: the third key tile starts from -1e30 there instead of 0 (its first MFMA takes this as its C operand) — the zero K
    // rows of the padding add nothing, exp2 turns it into 0.  Register r of tile kt is key 32 kt + 16 (r >> 3) + 8 hi + (r & 7).
    f32x16 mask2;
#pragma unroll
    for (int r = 0; r < 16; ++r) mask2[r] = (64 + 16 * (r >> 3) + 8 * hi + (r & 7)) < a.Lk ? 0.f : -1e30f;
    const bf16x8 ones = {1, 1, 1, 1, 1, 1, 1, 1};

    for (int kvb = 0; kvb < nkvb; ++kvb) {
        // ---- stage K (zero-padded rows / columns) and V^T of (kvb, group) ----
        __syncthreads();
        // (requesting all sixteen 16-byte loads of a thread before the first store was slower: 128 / 58 / 59 us against 120 / 50 / 38)
        for (int t = tid; t < (AT_PROBE == 3 ? 0 : kKeys * (kG / 8 + 1)); t += kNT) {       // K: consecutive threads = consecutive channels
            const int key = t / (kG / 8 + 1), g8 = t - key * (kG / 8 + 1);
            bf16x8 kv = {0, 0, 0, 0, 0, 0, 0, 0};
            if (key < a.Lk && g8 < kG / 8) kv = *(const bf16x8*)((const bf16*)a.k + (size_t)((int64_t)kvb * a.kv_outer_rows + key) * a.ldk + ch0 + g8 * 8);
            *(bf16x8*)(sK + key * kKRow + g8 * 16) = kv;
        }
        for (int t = tid; t < (AT_PROBE == 3 ? 0 : kKeys * (kG / 8)); t += kNT) {           // V^T: consecutive threads = consecutive KEYS, so the
            const int g8 = t / kKeys, key = t - g8 * kKeys;                                 // 2-byte transposing writes of a wave fall into one row
            bf16x8 vv = {0, 0, 0, 0, 0, 0, 0, 0};                                          // (channel-fastest they were a 32-way bank conflict: 11 us)
            if (key < a.Lk) vv = *(const bf16x8*)((const bf16*)a.v + (size_t)((int64_t)kvb * a.kv_outer_rows + key) * a.ldv + ch0 + g8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) *(bf16*)(sV + (g8 * 8 + e) * kVRow + key * 2) = vv[e];
        }
        for (int t = tid; t < (kVRows - kG) * (kVRow / 16); t += kNT)          // the V^T rows past the last channel
            *(bf16x8*)(sV + kG * kVRow + t * 16) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = tid; t < kG; t += kNT)                                     // ... and the pad keys 96..103 of every row (never read)
            *(bf16x8*)(sV + t * kVRow + kKeys * 2) = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        __syncthreads();

        const int64_t row0 = (int64_t)kvb * rows_per_kvb;
        const int ntiles = (int)((rows_per_kvb + 31) / 32);
        // Q fragments of head h of tile t (B operand): channels 16 ks + 8 hi .. + 7 of this lane's row
        auto load_q = [&](int t, int h, bf16x8(&q)[KS]) {
            const int64_t rloc = (int64_t)t * 32 + l31;
            const bf16* const qr = (const bf16*)a.q + (size_t)(row0 + (rloc < rows_per_kvb ? rloc : 0)) * a.ldq + ch0 + h * D;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // always a GLOBAL load from inside the tensor (rows past the end read row 0 of the clip and are never stored; a
                // pointer select against a zero page would make these flat loads, whose lgkmcnt ties every LDS wait to them)
                const int dofs = 16 * ks + 8 * hi;
                if (AT_PROBE == 2) q[ks] = bf16x8{1, 1, 1, 1, 1, 1, 1, 1};
                else q[ks] = *(const bf16x8*)(qr + (dofs < D ? dofs : D - 8));
                if (16 * ks + 8 >= D && dofs >= D) q[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};        // (D = 40: the upper half of the third k-step)
            }
        };
        // All heads' fragments of a tile are requested a whole tile ahead: head h's registers are re-requested for the wave's NEXT tile as
        // soon as this tile's head h has used them, so every load has eight head iterations (~8 us) to land.  (One head ahead was
        // not enough: an iteration is ~1 us of work, a loaded-HBM round trip several — the kernel ran at the memory latency.)
        bf16x8 qa[HPG][KS];
        if (first < ntiles) {
#pragma unroll
            for (int h = 0; h < HPG; ++h) load_q(first, h, qa[h]);
        }
        for (int t = first; t < ntiles; t += stride) {
            const int64_t rloc = (int64_t)t * 32 + l31;
            const bool rok = rloc < rows_per_kvb;
            // row of (batch, i): contiguous token rows (q_outer_rows == Lq is checked by the launcher)
            bf16* const orow = (bf16*)a.o + (size_t)(row0 + (rok ? rloc : 0)) * a.ldo + ch0;
#pragma unroll
            for (int h = 0; h < HPG; ++h) {
                __builtin_amdgcn_sched_barrier(0);       // one head at a time: the unrolled heads must not be interleaved (register budget)
                bf16x8 qf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) qf[ks] = qa[h][ks];
                // ---- S^T = K Q^T, three key tiles; rows read in permuted order (krow); the fragments of tile kt + 1 are requested
                //      before the MFMAs of tile kt are issued (left alone, hipcc reads, waits, multiplies, reads again) ----
                f32x16 sc[3];
                bf16x8 kf[2][KS];
                auto read_k = [&](int kt, bf16x8(&f)[KS]) {
                    const char* const kp = sK + (32 * kt + krow) * kKRow + (h * D + 8 * hi) * 2;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) f[ks] = *(const bf16x8*)(kp + ks * 32);
                };
                read_k(0, kf[0]);
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    if (kt + 1 < 3) read_k(kt + 1, kf[(kt + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[kt][r] = kt == 2 ? mask2[r] : 0.f;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt & 1][ks], qf[ks], sc[kt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (t + stride < ntiles) load_q(t + stride, h, qa[h]);
                __builtin_amdgcn_sched_barrier(0);
                // ---- softmax over the keys (all of them at once: one max, one exp2 per score) ----
                float mx = -1e30f;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float mc = mx * c2;
                bf16x8 pf[3][2];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float p = AT_PROBE == 4 ? sc[kt][r] : __builtin_amdgcn_exp2f(sc[kt][r] * c2 - mc);
                        pf[kt][r >> 3][r & 7] = f2bf(p);
                    }
                // the row sums on the matrix pipe: 1^T P (every row of the product is the sum over the keys of a query column) — six
                // MFMAs with a constant A operand instead of 48 dependent VALU adds, and of the bf16 values the second product uses
                f32x16 ssum;
#pragma unroll
                for (int r = 0; r < 16; ++r) ssum[r] = 0.f;
#pragma unroll
                for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) ssum = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[kt][s2], ssum, 0, 0, 0);
                const float inv = 1.0f / ssum[0];
                // ---- O^T = V^T P ----
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    f32x16 o;
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = 0.f;
                    const char* const vp = sV + (h * D + 32 * dt + l31) * kVRow + 8 * hi * 2;
                    bf16x8 vf[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) vf[i] = *(const bf16x8*)(vp + 16 * i * 2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 6; ++i) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[i], pf[i >> 1][i & 1], o, 0, 0, 0);
                    // register r: channel 32 dt + (r & 3) + 8 (r >> 2) + 4 hi of this lane's row -> 8-channel runs by the half exchange
#pragma unroll
                    for (int tq = 0; tq < 2; ++tq) {
                        float x[4], y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            x[e] = o[8 * tq + e] * inv;
                            y[e] = o[8 * tq + 4 + e] * inv;
                            at_swap(x[e], y[e]);
                        }
                        const int dofs = 32 * dt + 16 * tq + 8 * hi;
                        if (rok && dofs < D && (AT_PROBE != 1 || x[0] == 1234.5f)) {
                            const bf16x8 w = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3]), f2bf(y[0]), f2bf(y[1]), f2bf(y[2]), f2bf(y[3])};
                            *(bf16x8*)(orow + h * D + dofs) = w;
                        }
                    }
                }
            }
        }
    }
}

template <int D>
int launch_text(const CcAttnDesc& a, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)attn_text_kernel<D>, kLds, &attr_done, "attn_text")) return rc;
    const int ngroups = a.heads * a.d / kG;
    const int nkvb = a.batches / a.kv_div;
    cc_note_kernel("attn_text_kernel d=%d", D);
    hipLaunchKernelGGL((attn_text_kernel<D>), dim3(256), dim3(kNT), kLds, s, a, ngroups, nkvb);
    return cc_launch_status("attn_text_kernel");
}

}  // namespace

// text cross-attention geometry: few keys shared by all frames of a clip, contiguous token rows, whole 320-channel groups.
// Measured against the general kernel (34 frames, 77 keys, cold): d = 40 at 6144 rows per frame 120 / 127 us, d = 80 at 1536 50 / 75,
// d = 160 at 384 38 / 34 — the staging of two clips' K / V per workgroup is a fifth of that launch: d = 160 stays on the general kernel.
bool cc_attn_text_applicable(const CcAttnDesc& a) {
    return (a.d == 40 || a.d == 80) && a.Lk <= kKeys && a.Lk >= 64 && a.seg1_len == 0 && !a.causal &&
           (a.heads * a.d) % kG == 0 && 32 % (a.heads * a.d / kG) == 0 && a.q_inner == 1 && a.q_seq_rows == 1 && a.q_outer_rows == a.Lq &&
           a.kv_inner == 1 && a.kv_seq_rows == 1 && a.kv_outer_rows >= a.Lk && a.batches % a.kv_div == 0 && a.ldo % 8 == 0 &&
           (int64_t)a.kv_div * a.Lq >= 2048;
}

int cc_attn_text_launch(const CcAttnDesc& a, hipStream_t s) {
    switch (a.d) {
        case 40: return launch_text<40>(a, s);
        case 80: return launch_text<80>(a, s);
        default: return launch_text<160>(a, s);
    }
}
