// Linear layers with K = 640 over many pixels (the 32x48 level: proj_in / proj_out, to_q, to_out, the fused q,k,v projection) with
// the WEIGHTS HELD IN REGISTERS — lin320s_kernel's scheme (lin320.hip) at twice the K — gfx950.
//
// The persistent LDS-tiled GEMM (gemm8p.hip) runs these at 480-650 TF/s: 640 channels are 2.5 of its 256-channel tiles, a tile's
// prologue and epilogue are not covered by anything (one workgroup per CU), and both operands go through LDS again for every
// tile.  Here a workgroup keeps a 256-channel slice of W (320 KB) as MFMA A-operand fragments for its whole life — wave w holds
// channels [32 w, 32 w + 32) x all 640 k (2 x 20 fragments, 160 VGPRs) — and streams 16-pixel activation tiles past it: global ->
// LDS by DMA into a ring of five 21 KB buffers (82 KB in flight per CU, rows XOR-swizzled: conflict-free fragment reads), 40 v_mfma_f32_16x16x32_bf16 per wave and tile, every
// activation fragment read from LDS feeding two of them.  (One 16-channel tile per wave — 128-channel slices — needs a fragment
// read per MFMA: 256 B/clk, the whole LDS bandwidth of the CU; measured 2 us per 32-pixel tile whatever the memory system did,
// slower than gemm8p.)  The slices of a layer (3 for 640 channels — the third half empty —, 8 for the 1920 of q,k,v) are different
// workgroups of one XCD walking the SAME pixel tiles, so the activation rows come out of that XCD's L2 for all but the first.
//
// Waves 0..3 issue the DMA, waves 4..7 the stores and atomics: a requesting wave's vector-memory queue then holds loads only and
// the counted wait is exact (a store in the same queue has to complete as well before the count can fall to the number of younger
// loads).  No register-returning load inside the loop (the compiler's wait for it would drain the DMA queue).
//
// Epilogue in registers: the C layout gives lane (pixel, 4 channels) — the accumulators start from the bias (+ the residual
// cell: the residual slice arrives by DMA like the activations), leave as bf16 into an LDS tile in row layout (for the residual
// variant the tile the residual came in, every lane reading and later writing its own 8 bytes) and go to memory one iteration
// later as whole 512-byte row pieces, 16 bytes per lane.  One barrier per tile.
//   LNF  (CcGemmDesc.ln_stats / ln_sums + ln_colsum): the rows are LayerNorm inputs; the tile's 16 (sum, sum of squares) or
//        (mean, rstd) pairs ride in spare lanes of the tile's last DMA instruction; accumulators start at
//        b' / rstd - mean colsum(W') and are multiplied by rstd at the end (the formula of gemm8p.hip).
//   row_sums: every lane sums its eight bf16-rounded outputs (and their squares), two cross-lane steps make that 32 channels, the
//        eight waves meet in DOUBLE in LDS (ds_add_f64: exact for these fp32 partials, so order-independent) and 32 threads add
//        the slice's totals to row_sums[M][2] with double atomics one iteration later — as gemm8p.hip does across channel tiles.
// Requires M % 16 == 0 (whole tiles: the counted wait relies on the number of loads per iteration), K = Kpad = 640, N % 128 == 0.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kK = 640, kKS = kK / 32;          // 20 MFMA k-steps
constexpr int kP = 16;                          // pixels per tile
// Activation tile in LDS: 16 rows of 80 sixteen-byte granules, NO padding; granule j of row r sits at position j ^ (r & 15) of its
// row (the DMA picks the source granule per destination slot).  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md): lane (c, g) of a B fragment reads granule 4 ks + g of row c, so a group holds all
// sixteen c — half of them with g, half with g + 1 — and both halves are closed under c ^ 1 and c ^ 2: the positions
// ((4 ks + g) ^ c) cover the sixteen 16-byte bank groups exactly once.  (Rows padded to 1296 B, the usual recipe, put two lanes of
// every group on the same banks: SQ_LDS_BANK_CONFLICT = 20 % of the kernel's cycles.)
constexpr int kRS = kK * 2;                     // 1280 B = 5 x 256: a row's position in the banks depends on the swizzle alone
constexpr int kGPR = kRS / 16;                  // 80
constexpr int kXI = kP * kGPR / 64;             // 20 wave-wide DMA instructions per tile (+ one 16-lane instruction for the statistics, LNF)
constexpr int kStatOff = kP * kRS;              // 20,480: 16 x 16 B — (sum, sumsq) doubles or (mean, rstd) floats of the tile's rows
constexpr int kXBuf = kStatOff + 1024;          // 21,504 B
constexpr int kSlice = 256;                     // channels per workgroup
constexpr int kORS = kSlice * 2;                // output / residual tile row: 512 B = 32 granules, granule g at (g & 16) | ((g & 15) ^ row)
constexpr int kOBuf = kP * kORS;                // 8,192 B = 8 DMA instructions, two per requesting wave
constexpr int kRX = 5;                          // activation ring
constexpr int kROres = 6, kROplain = 2;         // residual / output ring (residual tiles travel as far ahead as the activations)
constexpr int kNT = 512;
constexpr int kLds = kRX * kXBuf + kROres * kOBuf + 2 * kP * 16;      // + two [16][2] double accumulators of the row sums: 157,184 B
constexpr int kXDmax = 6;                       // activation fragments in flight per wave (LDS read -> MFMA distance, in k-steps; 5 with the LayerNorm's extra registers)

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void l6_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// (these loops are bound by instruction issue — temp320.hip has the measurement: ring slots are wrap-around counters, tile addresses
//  running byte offsets, every lane of every DMA instruction has a valid source (no selects, no exec masks), and every requesting wave
//  issues the same number of loads per tile, so the counted waits are compile-time constants)

template <bool RES, bool LNF>
__global__ __launch_bounds__(kNT, 1) void lin640s_kernel(const CcGemmDesc d, int nslice, int pt_n) {
    static_assert(!(RES && LNF), "the normalised projections have no residual");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RO = RES ? kROres : kROplain;
    char* const sXr = smem;
    char* const sOr = smem + kRX * kXBuf;
    double* const sSum = (double*)(smem + kRX * kXBuf + kROres * kOBuf);      // [2][16][2]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;

    // workgroup b runs on XCD b % 8: its 32 workgroups take (32 / nslice) pixel lanes x nslice channel slices
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int lanes = 32 / nslice;
    const int slice = j % nslice, plane = j / nslice;
    if (plane >= lanes) return;
    const int per_xcd = (pt_n + 7) >> 3;
    const int pt_lo = xcd * per_xcd, pt_hi = min(pt_lo + per_xcd, pt_n);
    const int pt0 = pt_lo + plane;
    if (pt0 >= pt_hi) return;
    const int ntile = (pt_hi - pt0 + lanes - 1) / lanes;        // tiles of this workgroup: pt0 + i * lanes
    const int ch0 = slice * kSlice;
    const bool rsum = d.row_sums != nullptr;                     // (uniform)
    const bool live = ch0 + 32 * wave < d.N;                     // N % 256 == 128: waves 4..7 of the last slice have no channels

    // ---- DMA plan (waves 0..3; every lane of every instruction has a valid source).  Activations: instruction q -> wave q % 4, slot n of
    //      the tile = row n / 80, position n % 80 <- source granule position ^ (row & 15).  Residual slice: two instructions per wave,
    //      slot (r, p) <- granule (p & 16) | ((p & 15) ^ r) (channels beyond N clamped).  LNF: four lanes per wave fetch the statistics. ----
    const bool dma_wave = wave < 4;
    constexpr int kXQ = kXI / 4;                                 // 5 per requesting wave
    constexpr int kPerTile = kXQ + (RES ? 2 : 0) + (LNF ? 1 : 0);      // loads of one tile in a requesting wave's queue
    int soffx[kXQ], soffr[2] = {0, 0};
#pragma unroll
    for (int i = 0; i < kXQ; ++i) {
        const int n = (i * 4 + (wave & 3)) * 64 + lane;
        const int r = n / kGPR, p = n - r * kGPR;
        soffx[i] = (r * d.lda + (p ^ (r & 15)) * 8) * 2;         // bytes
    }
    if constexpr (RES) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = (k * 4 + (wave & 3)) * 64 + lane;
            const int r = n >> 5, p = n & 31;
            const int g = (p & 16) | ((p & 15) ^ r);
            soffr[k] = (r * d.ldr1 + min(ch0 + g * 8, d.N - 8)) * 2;
        }
    }
    const bool sums_in = LNF && d.ln_sums != nullptr;            // else d.ln_stats (floats)
    const char* const Ab = (const char*)d.A;
    const char* const Rb = (const char*)d.res1;
    const char* const Sb = sums_in ? (const char*)d.ln_sums : (const char*)d.ln_stats;
    const int64_t tile_rows = (int64_t)lanes * kP;               // consecutive tiles of this workgroup are `lanes` tiles apart
    const int64_t x_step = tile_rows * d.lda * 2, r_step = tile_rows * d.ldr1 * 2, o_step = tile_rows * d.ldc * 2;
    const int s_step = (int)tile_rows * (sums_in ? 16 : 8);
    int64_t st_x = (int64_t)pt0 * kP * d.lda * 2, st_r = RES ? (int64_t)pt0 * kP * d.ldr1 * 2 : 0, st_s = (int64_t)pt0 * kP * (sums_in ? 16 : 8);
    // statistics: 16 rows x 16 B (sums) or 8 B (mean, rstd): wave w fetches lanes 0..3 -> bytes [64 w, 64 w + 64) of the 256 (128: waves 0, 1)
    const int soffs = min((wave & 3) * 64 + (lane & 3) * 16, (sums_in ? 256 : 128) - 16);
    int st_xs = 0, st_os = 0;                                    // ring slots of the tile being requested
    auto stage_next = [&]() {
        if (dma_wave) {
            char* const xd = sXr + st_xs * kXBuf;
#pragma unroll
            for (int q = 0; q < kXQ; ++q) glds16(Ab + st_x + soffx[q], xd + (q * 4 + wave) * 1024);
            if constexpr (LNF) {
                if (lane < 4) glds16(Sb + st_s + soffs, xd + kStatOff + wave * 64);
            }
            if constexpr (RES) {
#pragma unroll
                for (int k = 0; k < 2; ++k) glds16(Rb + st_r + soffr[k], sOr + st_os * kOBuf + (k * 4 + wave) * 1024);
            }
        }
        st_x += x_step;
        st_r += r_step;
        st_s += s_step;
        st_xs = st_xs == kRX - 1 ? 0 : st_xs + 1;
        st_os = st_os == (RES ? kROres : kROplain) - 1 ? 0 : st_os + 1;
    };
    auto wait_later = [&](int later) {                            // at most the loads of `later` (<= kRX - 2) newer tiles may be outstanding
        if (later >= 3) l6_vmcnt<3 * kPerTile>();
        else if (later == 2) l6_vmcnt<2 * kPerTile>();
        else if (later == 1) l6_vmcnt<kPerTile>();
        else l6_vmcnt<0>();
    };
    static_assert(kRX - 2 == 3, "wait_later covers three tiles in flight behind the awaited one");

    // ---- the weight rows of this wave: A-operand fragments, resident for the whole kernel; bias (and colsum) of the lane's 2 x 4 channels ----
    bf16x8 wf[2][kKS];
    f32x4 bq[2], cq[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        if (d.Wfrag) {      // fragment-ordered copy (CcGemmDesc.Wfrag): a fragment is one contiguous kilobyte per wave (rows are padded to 256: dead waves read zeros)
            const bf16* __restrict__ blk = (const bf16*)d.Wfrag + ((size_t)((ch0 + 32 * wave + 16 * ti) >> 4) * (kK / 32) * 64 + lane) * 8;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) wf[ti][ks] = *(const bf16x8*)(blk + ks * 512);
        } else {
            const int cr = min(ch0 + 32 * wave + 16 * ti + c16, d.N - 1);      // (dead waves: any valid row)
            const bf16* __restrict__ row = (const bf16*)d.W + (size_t)cr * d.Kpad + g4 * 8;
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) wf[ti][ks] = *(const bf16x8*)(row + ks * 32);
        }
        const int cb = min(ch0 + 32 * wave + 16 * ti + 4 * g4, d.N - 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bq[ti][e] = d.bias ? d.bias[cb + e] : 0.f;
            cq[ti][e] = LNF ? d.ln_colsum[cb + e] : 0.f;
        }
    }
    if (tid < 2 * kP * 2) sSum[tid] = 0.0;
    // the first tiles are requested BEHIND the weight loads, in the same breath (round 6: one latency instead of two at the start of
    // every launch); one wait below covers both — from then on a requesting wave's queue holds DMA only
    int staged = 0;
    for (; staged < kRX - 1 && staged < ntile; ++staged) stage_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    int xlane[4];                                               // B fragment of k-step ks: xlane[ks & 3] + (ks >> 2) * 256
#pragma unroll
    for (int m = 0; m < 4; ++m) xlane[m] = c16 * kRS + (((4 * m + g4) ^ c16) << 4);
    // C cell (pixel c16, channels 32 w + 16 ti + 4 g4 .. + 3) = half (g4 & 1) of granule 4 w + 2 ti + (g4 >> 1), at its swizzled position
    int olane[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int gr = 4 * wave + 2 * ti + (g4 >> 1);
        olane[ti] = c16 * kORS + (((gr & 16) | ((gr & 15) ^ c16)) << 4) + (g4 & 1) * 8;
    }
    // output pass (waves 4..7): thread t - 256 takes 16-byte granule t % 32 of rows t / 32 and t / 32 + 8 of the tile (16 rows x 32 granules)
    const int ot0 = tid & 255;
    const int ogr = ot0 & 31, orow = ot0 >> 5;
    const int olds0 = orow * kORS + (((ogr & 16) | ((ogr & 15) ^ orow)) << 4);
    const int olds1 = (orow + 8) * kORS + (((ogr & 16) | ((ogr & 15) ^ (orow + 8))) << 4);
    const int oout = (orow * d.ldc + ch0 + ogr * 8) * 2;        // bytes; second row: + 8 ldc
    const int oout8 = 8 * d.ldc * 2;
    const bool ost = !dma_wave && ch0 + ogr * 8 < d.N;
    char* const Ob = (char*)d.out;
    int64_t so_off = (int64_t)pt0 * kP * d.ldc * 2;              // output cursor: the tile stored at step i is i - 1
    int64_t so_rows = (int64_t)pt0 * kP;                         // ... its first row (row sums)

    int xb = 0, ob = 0, obp = 0;                                 // slots of tile i / of tile i - 1's output
    const double inv_k = 1.0 / kK;
    for (int i = 0; i < ntile; ++i) {
        lds_barrier();      // tile i is in LDS for every wave; output tile i-1 and its row sums are complete; the X slot of tile i-1 and the O slot of tile i-2 are free
        if (staged < ntile) {
            stage_next();
            ++staged;
        }
        const char* const xt = sXr + xb * kXBuf;
        char* const ot = sOr + ob * kOBuf;
        // output pass of tile i - 1, first half: its row pieces out of LDS; its row sums to memory
        bf16x8 ov[2];
        if (i > 0 && !dma_wave) {
            ov[0] = *(const bf16x8*)(sOr + obp * kOBuf + olds0);
            ov[1] = *(const bf16x8*)(sOr + obp * kOBuf + olds1);
            if (rsum && wave == 5 && lane < 2 * kP) {
                double* const acc2 = sSum + ((i - 1) & 1) * 2 * kP + lane;
                const double v = *acc2;
                *acc2 = 0.0;
                unsafeAtomicAdd(d.row_sums + 2 * so_rows + lane, v);
            }
            // stored right away, while the requesting waves issue their DMA (inside the MFMA loop the store would wait for `ov`
            // behind the fragment prefetch in the LDS queue: temp320.hip measured the storing waves 350 cycles per step behind)
            if (ost) {
                char* const op = Ob + so_off + oout;
                *(bf16x8*)op = ov[0];
                *(bf16x8*)(op + oout8) = ov[1];
            }
        }
        // accumulators start from the bias (+ the residual cell) — or b' / rstd - mean colsum for the folded LayerNorm
        f32x4 acc[2];
        float rstd = 1.f;
        if constexpr (LNF) {
            float mu;
            const char* sp = xt + kStatOff;
            if (sums_in) {
                const double s = *(const double*)(sp + c16 * 16), q = *(const double*)(sp + c16 * 16 + 8);
                const double m = s * inv_k;
                const double var = fmax(q * inv_k - m * m, 0.0);
                mu = (float)m;
                rstd = rsqrtf((float)var + d.ln_sums_eps);
            } else {
                const f32x2 st = *(const f32x2*)(sp + c16 * 8);
                mu = st[0];
                rstd = st[1];
            }
            const float ir = 1.0f / rstd;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[ti][e] = bq[ti][e] * ir - mu * cq[ti][e];
        } else {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                acc[ti] = bq[ti];
                if constexpr (RES) {
                    const bf16x4 r = *(const bf16x4*)(ot + olane[ti]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[ti][e] += bf2f(r[e]);
                }
            }
        }
        if (live) {
            constexpr int kXD = LNF ? kXDmax - 1 : kXDmax;
            bf16x8 xq[kXD];
#pragma unroll
            for (int ks = 0; ks < kXD - 1; ++ks) xq[ks] = *(const bf16x8*)(xt + xlane[ks & 3] + (ks >> 2) * 256);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if (ks + kXD - 1 < kKS) xq[(ks + kXD - 1) % kXD] = *(const bf16x8*)(xt + xlane[(ks + kXD - 1) & 3] + ((ks + kXD - 1) >> 2) * 256);
                __builtin_amdgcn_sched_barrier(0);
                // the compiler counts the reads still in flight behind this fragment (all issued above, in order)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) acc[ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ti][ks], xq[ks % kXD], acc[ti], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // bf16 cells of this wave's 32 channels into the output tile; row sums of the rounded values
        if (live) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = f2bf(LNF ? acc[ti][e] * rstd : acc[ti][e]);
                    const float v = bf2f(o[e]);
                    s += v;
                    q += v * v;
                }
                *(bf16x4*)(ot + olane[ti]) = o;
            }
            if (rsum) {
                s += __shfl_xor(s, 16);
                q += __shfl_xor(q, 16);
                s += __shfl_xor(s, 32);
                q += __shfl_xor(q, 32);
                if (g4 == 0) {
                    // (asm: for an LDS atomic the compiler first drains the vector-memory queue — the DMA writes LDS too)
                    const uint32_t a2 = (uint32_t)(uintptr_t)(LDS_AS char*)(sSum + (i & 1) * 2 * kP + 2 * c16);
                    asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(a2), "v"((double)s), "v"((double)q) : "memory");
                }
            }
        }
        // tile i + 1 must have landed (this wave's part) before the next barrier: at most the loads of the tiles after it may be outstanding
        if (i + 1 < ntile && dma_wave) wait_later(staged - (i + 2));
        if (i > 0) {                                             // the output cursor follows one tile behind
            so_off += o_step;
            so_rows += tile_rows;
        }
        obp = ob;
        xb = xb == kRX - 1 ? 0 : xb + 1;
        ob = ob == RO - 1 ? 0 : ob + 1;
    }
    lds_barrier();
    {
        if (ost) {
            char* const op = Ob + so_off + oout;
            *(bf16x8*)op = *(const bf16x8*)(sOr + obp * kOBuf + olds0);
            *(bf16x8*)(op + oout8) = *(const bf16x8*)(sOr + obp * kOBuf + olds1);
        }
        if (rsum && wave == 5 && lane < 2 * kP) unsafeAtomicAdd(d.row_sums + 2 * so_rows + lane, sSum[((ntile - 1) & 1) * 2 * kP + lane]);
    }
}

}  // namespace

// plain Linear with K = 640 over whole 16-pixel tiles: bias, one residual, row_sums, or the folded LayerNorm (no residual)
bool cc_lin640_applicable(const CcGemmDesc& d) {
    const bool lnf = d.ln_stats || d.ln_sums;
    return d.mode == CCEDIT_GEMM_LINEAR && d.taps == 1 && d.A2 == nullptr && d.Cin == kK && d.Kpad == kK && d.N % 128 == 0 &&
           (d.N + kSlice - 1) / kSlice <= 32 && d.M % kP == 0 && d.gn_stats == nullptr && d.res2 == nullptr && d.group_bias == nullptr && !d.out_f32 &&
           d.act == CCEDIT_ACT_NONE && d.ldc % 8 == 0 && (d.res1 == nullptr || d.ldr1 % 8 == 0) && d.ln_eps == 0.f &&
           (!lnf || (d.ln_colsum && !(d.ln_stats && d.ln_sums) && !d.res1 && !d.row_sums)) && (lnf || !d.ln_colsum);
}

int cc_lin640_launch(const CcGemmDesc& d, hipStream_t s) {
    static unsigned long long attr_done[3] = {0, 0, 0};
    if (int rc = cc_max_dynamic_lds((const void*)lin640s_kernel<false, false>, kLds, &attr_done[0], "lin640s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)lin640s_kernel<true, false>, kLds, &attr_done[1], "lin640s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)lin640s_kernel<false, true>, kLds, &attr_done[2], "lin640s")) return rc;
    const int64_t pt_n = d.M / kP;
    if (pt_n > 2147483647LL) {
        cc_set_error("ccedit_gemm: grid too large");
        return CCEDIT_EUNSUPPORTED;
    }
    cc_note_kernel("lin640s_kernel");
    const int nslice = (d.N + kSlice - 1) / kSlice;
    if (d.res1) hipLaunchKernelGGL((lin640s_kernel<true, false>), dim3(256), dim3(kNT), kLds, s, d, nslice, (int)pt_n);
    else if (d.ln_stats || d.ln_sums) hipLaunchKernelGGL((lin640s_kernel<false, true>), dim3(256), dim3(kNT), kLds, s, d, nslice, (int)pt_n);
    else hipLaunchKernelGGL((lin640s_kernel<false, false>), dim3(256), dim3(kNT), kLds, s, d, nslice, (int)pt_n);
    return cc_launch_status("lin640s_kernel");
}
