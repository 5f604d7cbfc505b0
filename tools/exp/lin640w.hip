// Linear layers with K = 640 and N % 320 == 0 over many pixels (the 32x48 level: proj_in / proj_out, to_q, to_out, the fused
// q,k,v projection) — the register-resident-weight scheme of lin640.hip with WIDE slices: a workgroup of FOUR waves owns 320 output
// channels, each wave 80 of them x all 640 k as MFMA A-operand fragments (5 x 20 fragments, 400 registers, one wave per SIMD).
//
// Why a second K = 640 kernel (round 6, tools/exp/lin640_slices.py): lin640s_kernel (256-channel slices, eight waves of 32 channels)
// costs  17 us + 1.25 us per 16-pixel tile and workgroup  whatever the width — 33 / 49 / 65 / 68 us at N = 256 / 512 / 640 / 768 over
// 52224 rows: at N = 640 the third, half-empty slice is a full pass, and a tile costs 2500 cycles against 1280 matrix-pipe cycles
// per SIMD because every one of the eight waves reads the WHOLE 20 KB activation tile from LDS for 40 MFMAs (160 KB of ds_read_b128
// per tile and CU at <= 128 B/clk: the loop is LDS-bandwidth-bound at half the matrix rate).  Here an activation fragment read from
// LDS feeds FIVE v_mfma_f32_16x16x32_bf16 (80 KB of fragment reads per tile and CU for 100 MFMAs per wave = 1600 matrix-pipe
// cycles), 640 channels are exactly two slices, and all 32 workgroups of an XCD have work.
//
// One barrier per tile, in the MIDDLE of the tile's K loop: by then this wave's share of tile i + 1's DMA has been waited for, so
// after the barrier tile i + 1 is in LDS for everybody, the output tile i - 1 (written into LDS at the end of tile i - 1 by all four
// waves) can go to memory as whole row pieces, and the slots tile i + LEAD will land in are free.  The tile boundary itself has no
// barrier: a wave runs from its last MFMAs of tile i through the epilogue into the first fragment reads of tile i + 1 alone.
// All four waves issue the same vector-memory instructions per tile (DMA, then stores), so one counted wait — the number of
// instructions issued after tile i + 1's DMA, kept at run time — is exact for every wave (loads and stores retire in order on this
// queue; lin640.hip explains why it keeps them in different waves: here there are no spare waves).
//
// Epilogues as in lin640.hip: bias (+ residual tile, which arrives by DMA in the slot the output leaves from), the folded
// LayerNorm (ln_stats / ln_sums + ln_colsum: accumulators start at b' / rstd - mean colsum(W'), times rstd at the end), row_sums
// (per-row sums of the bf16-rounded outputs, the four waves meeting in double in LDS, double atomics to memory).
// Requires M % 16 == 0, K = Kpad = 640, N % 320 == 0.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kK = 640, kKS = kK / 32;          // 20 MFMA k-steps
constexpr int kP = 16;                          // pixels per tile
constexpr int kRS = kK * 2;                     // activation row: 1280 B = 80 granules, granule j of row r at position j ^ (r & 15)
constexpr int kGPR = kRS / 16;
constexpr int kXQ = 5;                          // DMA instructions per wave and activation tile (20 KB / 4 waves / 1 KB)
constexpr int kStatOff = kP * kRS;              // 16 x 16 B: (sum, sumsq) doubles or (mean, rstd) floats of the tile's rows
constexpr int kXBuf = kStatOff + 256;
constexpr int kSlice = 320, kCW = 80, kTI = 5;  // channels per workgroup / per wave; 16-channel MFMA tiles per wave
constexpr int kTA = 3;                          // ... of which this many keep their weight fragments in AGPRs
constexpr int kORS = kSlice * 2;                // output / residual tile row: 640 B = 40 granules, granule g at (g & ~7) | ((g & 7) ^ (row & 7))
constexpr int kOGR = kORS / 16;
constexpr int kOBuf = kP * kORS;                // 10,240 B = ten wave-wide DMA instructions
constexpr int kRQ = 3;                          // residual DMA instructions per wave (12 for 10: the last two repeat the wave's previous one)
constexpr int kSQ = 3;                          // output stores per thread (768 for 640 granules: threads 128.. repeat their second)
constexpr int kNT = 256;
constexpr int kXD = 5;                          // activation fragments in flight (LDS read -> MFMA distance in k-steps)
constexpr int kMid = 9;                         // the k-step in front of which the tile's barrier sits

template <int RES>
struct Ring {
    static constexpr int LEAD = RES ? 3 : 5;    // tiles requested ahead
    static constexpr int RX = LEAD + 1;         // activation slots: tile i + LEAD lands in the slot tile i - 1 left
    static constexpr int RO = RES ? LEAD + 2 : 2;      // output slots (the residual tile arrives in the slot its output leaves from)
    static constexpr int kLds = RX * kXBuf + RO * kOBuf + 2 * kSlice * 4 + 2 * kP * 2 * 8;
};

__device__ __forceinline__ void w_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt vmcnt(n) for a run-time n (wave-uniform): the immediate has six bits
__device__ __forceinline__ void wait_vm_rt(int n) {
#define W_CASE(k) \
    case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        W_CASE(0) W_CASE(1) W_CASE(2) W_CASE(3) W_CASE(4) W_CASE(5) W_CASE(6) W_CASE(7) W_CASE(8) W_CASE(9)
        W_CASE(10) W_CASE(11) W_CASE(12) W_CASE(13) W_CASE(14) W_CASE(15) W_CASE(16) W_CASE(17) W_CASE(18) W_CASE(19)
        W_CASE(20) W_CASE(21) W_CASE(22) W_CASE(23) W_CASE(24) W_CASE(25) W_CASE(26) W_CASE(27) W_CASE(28) W_CASE(29)
        W_CASE(30) W_CASE(31) W_CASE(32) W_CASE(33) W_CASE(34) W_CASE(35) W_CASE(36) W_CASE(37) W_CASE(38) W_CASE(39)
        W_CASE(40) W_CASE(41) W_CASE(42) W_CASE(43) W_CASE(44) W_CASE(45) W_CASE(46) W_CASE(47) W_CASE(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef W_CASE
}

template <int RES, int LNF, int RSUM>
__global__ __launch_bounds__(kNT, 1) void lin640w_kernel(const CcGemmDesc d, int nslice, int pt_n) {
    static_assert(!(RES && LNF), "the normalised projections have no residual");
    using R = Ring<RES>;
    constexpr int LEAD = R::LEAD, RX = R::RX, RO = R::RO;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sXr = smem;
    char* const sOr = smem + RX * kXBuf;
    float* const sBias = (float*)(sOr + RO * kOBuf);        // [320]
    float* const sCol = sBias + kSlice;                      // [320]
    double* const sSum = (double*)(sCol + kSlice);           // [2][16][2]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;

    // workgroup b runs on XCD b % 8: its 32 workgroups take (32 / nslice) pixel lanes x nslice channel slices; the slices of a pixel
    // lane walk the same tiles at the same time, so all but the first find the activation rows in that XCD's L2
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

    // ---- DMA plan.  Every wave-wide piece is 1 KB of LDS; the same number of pieces in every wave; one address register per stream.
    //  activation tile: piece (rg, pc) = rows 4 rg .. 4 rg + 3 x positions 16 pc .. 16 pc + 15, lane (rl, pl) = (lane >> 4, lane & 15);
    //      position p of row r holds source granule p ^ (r & 15).  Wave w requests rg = w, pc = 0..4: ONE lane offset + pc * 256 B.
    //  residual / output tile: piece (rg, pc) = rows 8 rg .. 8 rg + 7 x positions 8 pc .. 8 pc + 7, lane (rl, pl) = (lane >> 3, lane & 7);
    //      position p of row r holds granule (p & ~7) | ((p & 7) ^ (r & 7)).  Wave w requests rg = w & 1 and pc = 0, 1, 2 (waves 0, 1) or
    //      3, 4, 4 (waves 2, 3: the last piece twice — same bytes to the same place). ----
    constexpr int kPerTile = kXQ + (RES ? kRQ : 0) + (LNF ? 1 : 0);
    constexpr int kPerStore = kSQ + (RSUM ? 1 : 0);
    constexpr int kSteady = (LEAD - 2) * (kPerTile + kPerStore) + kPerStore;      // instructions younger than tile i + 1's DMA in the middle of tile i
    const int xoff = ((4 * wave + (lane >> 4)) * d.lda + (((lane & 15) ^ (4 * wave + (lane >> 4))) & 15) * 8) * 2;
    const int rg_w = wave & 1, pc_w = wave < 2 ? 0 : 3;
    const int roff = RES ? ((8 * rg_w + (lane >> 3)) * d.ldr1 + ch0 + ((lane & 7) ^ (lane >> 3)) * 8 + pc_w * 64) * 2 : 0;
    const bool sums_in = LNF && d.ln_sums != nullptr;            // else d.ln_stats (floats)
    const char* const Ab = (const char*)d.A;
    const char* const Rb = (const char*)d.res1;
    const char* const Sb = sums_in ? (const char*)d.ln_sums : (const char*)d.ln_stats;
    const int64_t tile_rows = (int64_t)lanes * kP;               // consecutive tiles of this workgroup are `lanes` tiles apart
    const int64_t x_step = tile_rows * d.lda * 2, r_step = RES ? tile_rows * d.ldr1 * 2 : 0, o_step = tile_rows * d.ldc * 2;
    const int s_step = (int)tile_rows * (sums_in ? 16 : 8);
    int64_t st_x = (int64_t)pt0 * kP * d.lda * 2, st_r = RES ? (int64_t)pt0 * kP * d.ldr1 * 2 : 0, st_s = (int64_t)pt0 * kP * (sums_in ? 16 : 8);
    // statistics: 16 rows x 16 B (sums) or 8 B (mean, rstd): wave w fetches lanes 0..3 -> bytes [64 w, 64 w + 64) of the 256 (128: waves 0, 1)
    const int soffs = min(wave * 64 + (lane & 3) * 16, (sums_in ? 256 : 128) - 16);
    int st_xs = 0, st_os = 0;                                    // ring slots of the tile being requested
    auto stage_next = [&]() {
        char* const xd = sXr + st_xs * kXBuf + wave * (5 * 1024);
        const char* const xs = Ab + st_x + xoff;
#pragma unroll
        for (int pc = 0; pc < kXQ; ++pc) glds16(xs + pc * 256, xd + pc * 1024);
        if constexpr (LNF) {
            if (lane < 4) glds16(Sb + st_s + soffs, sXr + st_xs * kXBuf + kStatOff + wave * 64);
        }
        if constexpr (RES) {
            char* const od = sOr + st_os * kOBuf + (rg_w * 5 + pc_w) * 1024;
            const char* const rs = Rb + st_r + roff;
            glds16(rs, od);
            glds16(rs + 128, od + 1024);
            if (wave < 2) glds16(rs + 256, od + 2048);
            else glds16(rs + 128, od + 1024);
        }
        st_x += x_step;
        st_r += r_step;
        st_s += s_step;
        st_xs = st_xs == RX - 1 ? 0 : st_xs + 1;
        st_os = st_os == RO - 1 ? 0 : st_os + 1;
    };

    // ---- the weight rows of this wave: A-operand fragments, resident for the whole kernel.  400 registers of a 512-register wave:
    // tiles 0..kTA-1 (240 registers) live in ACCUMULATION registers and enter the MFMA from there (srcA may be an AGPR on this ISA;
    // the compiler itself only ever parks values there and copies them back — four v_accvgpr_read per MFMA, measured 2.1 us per tile —
    // hence the asm form below), tiles kTA.. in ordinary registers. ----
    bf16x8 wf[kTI][kKS];
#pragma unroll
    for (int ti = 0; ti < kTI; ++ti) {
#ifdef LW_COALESCED_W          // timing probe only (wrong values): the same bytes as whole 1 KB pieces per wave instruction
        const bf16* __restrict__ row = (const bf16*)d.W + (size_t)(ch0 + kCW * wave + 16 * ti) * d.Kpad + lane * 8;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) wf[ti][ks] = *(const bf16x8*)(row + ks * 512);
#else
        const bf16* __restrict__ row = (const bf16*)d.W + (size_t)(ch0 + kCW * wave + 16 * ti + c16) * d.Kpad + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) wf[ti][ks] = *(const bf16x8*)(row + ks * 32);
#endif
    }
    for (int t = tid; t < kSlice; t += kNT) {
        sBias[t] = d.bias ? d.bias[ch0 + t] : 0.f;
        sCol[t] = LNF ? d.ln_colsum[ch0 + t] : 0.f;
    }
    if (tid < 2 * kP * 2) sSum[tid] = 0.0;
    int staged = 0;
    for (; staged < LEAD && staged < ntile; ++staged) stage_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the weights and the first LEAD tiles are in (this wave's part)
    w_barrier();

    // B fragment of k-step ks (lane (c16, g4): row c16, granule 4 ks + g4): xlane[ks & 3] + (ks >> 2) * 1024
    int xlane[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xlane[m] = (c16 >> 2) * 5120 + (c16 & 3) * 256 + ((((4 * m + g4) ^ c16) & 15) << 4);
    // C cell (pixel c16, channels 80 w + 16 ti + 4 g4 .. + 3) = half (g4 & 1) of granule 10 w + 2 ti + (g4 >> 1) of row c16
    int olane[kTI];
#pragma unroll
    for (int ti = 0; ti < kTI; ++ti) {
        const int gr = 10 * wave + 2 * ti + (g4 >> 1);
        olane[ti] = (((c16 >> 3) * 5 + (gr >> 3)) * 64 + (c16 & 7) * 8 + ((gr & 7) ^ (c16 & 7))) * 16 + (g4 & 1) * 8;
    }
    const int bl0 = (kCW * wave + 4 * g4) * 4;                   // byte offset of the lane's first bias / colsum cell (tile ti: + 64 ti)
    // output pass: wave w takes pieces w, 4 + w and 8 + w (waves 2, 3: 4 + w again) of the tile's ten
    const int oout = ((lane >> 3) * d.ldc + ch0 + ((lane & 7) ^ (lane >> 3)) * 8) * 2;      // + (8 rg ldc + 64 pc) * 2 of the piece
    int opiece[kSQ], oscal[kSQ];
#pragma unroll
    for (int q = 0; q < kSQ; ++q) {
        int iq = q * 4 + wave;
        if (iq >= 10) iq -= 4;
        opiece[q] = iq * 1024;
        oscal[q] = (8 * (iq / 5) * d.ldc + 64 * (iq % 5)) * 2;
    }
    char* const Ob = (char*)d.out;
    int64_t so_off = (int64_t)pt0 * kP * d.ldc * 2;              // output cursor: the tile stored in the middle of tile i is i - 1
    int64_t so_rows = (int64_t)pt0 * kP;                         // ... its first row (row sums)

    // vector-memory bookkeeping of this wave: `issued` instructions since the queue was empty; mk[k] = its value right after the DMA
    // of tile i + 1 + k was issued (the prologue's tiles: 0 — they have landed)
    int issued = 0;
    int mk[LEAD];
#pragma unroll
    for (int k = 0; k < LEAD; ++k) mk[k] = 0;

    auto store_tile = [&](int slot, int par) {                   // output tile in `slot` -> memory; its row sums (parity par)
        bf16x8 ov[kSQ];
#pragma unroll
        for (int q = 0; q < kSQ; ++q) ov[q] = *(const bf16x8*)(sOr + slot * kOBuf + opiece[q] + lane * 16);
        if constexpr (RSUM) {
            if (lane < 8) {
                double* const a2 = sSum + par * 2 * kP + 8 * wave + lane;
                const double v = *a2;
                *a2 = 0.0;
                unsafeAtomicAdd(d.row_sums + 2 * so_rows + 8 * wave + lane, v);
            }
        }
#pragma unroll
        for (int q = 0; q < kSQ; ++q) *(bf16x8*)(Ob + so_off + oscal[q] + oout) = ov[q];
    };

    // The accumulators of a tile start from `nxt`: bias (+ the residual cell), or b' / rstd - mean colsum for the folded LayerNorm —
    // prepared, one 16-channel tile per k-step, in the second half of the PREVIOUS tile's K loop (its inputs landed with that tile's
    // barrier), so that a tile's first MFMAs wait for nothing.
    f32x4 nxt[kTI];
    float rstd = 1.f, rstd_n = 1.f, mu_n = 0.f, ir_n = 1.f;
    f32x4 tb, tc;
    bf16x4 tr;
    const double inv_k = 1.0 / kK;
    auto prep_stats = [&](const char* xn) {                      // LNF: (mean, rstd) of the lane's pixel in the tile at xn
        if constexpr (LNF) {
            const char* sp = xn + kStatOff;
            if (sums_in) {
                const double sm = *(const double*)(sp + c16 * 16), q = *(const double*)(sp + c16 * 16 + 8);
                const double m = sm * inv_k;
                const double var = fmax(q * inv_k - m * m, 0.0);
                mu_n = (float)m;
                rstd_n = rsqrtf((float)var + d.ln_sums_eps);
            } else {
                const f32x2 st = *(const f32x2*)(sp + c16 * 8);
                mu_n = st[0];
                rstd_n = st[1];
            }
            ir_n = 1.0f / rstd_n;
        }
    };
    auto prep_read = [&](int ti, const char* on) {
        tb = *(const f32x4*)((const char*)sBias + bl0 + 64 * ti);
        if constexpr (LNF) tc = *(const f32x4*)((const char*)sCol + bl0 + 64 * ti);
        if constexpr (RES) tr = *(const bf16x4*)(on + olane[ti]);
    };
    auto prep_make = [&](int ti) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (LNF) nxt[ti][e] = tb[e] * ir_n - mu_n * tc[e];
            else if constexpr (RES) nxt[ti][e] = tb[e] + bf2f(tr[e]);
            else nxt[ti][e] = tb[e];
        }
    };

    int xb = 0, ob = 0, obp = 0;                                 // slots of tile i / of tile i - 1's output
    // tile 0: accumulator start values and the first fragments
    prep_stats(sXr);
#pragma unroll
    for (int ti = 0; ti < kTI; ++ti) {
        prep_read(ti, sOr);
        prep_make(ti);
    }
    bf16x8 xq[kXD];
#pragma unroll
    for (int ks = 0; ks < kXD - 1; ++ks) xq[ks] = *(const bf16x8*)(sXr + xlane[ks & 3] + (ks >> 2) * 1024);

    for (int i = 0; i < ntile; ++i) {
        const char* const xt = sXr + xb * kXBuf;
        char* const ot = sOr + ob * kOBuf;
        const int xbn = xb == RX - 1 ? 0 : xb + 1, obn = ob == RO - 1 ? 0 : ob + 1;
        const char* const xn = i + 1 < ntile ? sXr + xbn * kXBuf : xt;       // (after the last tile: harmless reads of the same slot)
        const char* const on = i + 1 < ntile ? sOr + obn * kOBuf : ot;
        rstd = rstd_n;
        f32x4 acc[kTI];
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            // fragment kXD - 1 k-steps ahead: of this tile, or (the last four steps) the first ones of the next tile
            if (ks + kXD - 1 < kKS) xq[(ks + kXD - 1) % kXD] = *(const bf16x8*)(xt + xlane[(ks + kXD - 1) & 3] + ((ks + kXD - 1) >> 2) * 1024);
            else xq[(ks + kXD - 1) % kXD] = *(const bf16x8*)(xn + xlane[(ks + kXD - 1 - kKS) & 3] + ((ks + kXD - 1 - kKS) >> 2) * 1024);
            if (ks == kMid) {
                // ---- the tile's barrier: tile i + 1 has landed (this wave's part), then for everybody ----
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < ntile) {
                    const int n = issued - mk[0];
                    if (n == kSteady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kSteady) : "memory");
                    else wait_vm_rt(n);
                }
                w_barrier();
                // tile i + LEAD into the slots tile i - 1 (activations) / tile i - 2 (output) left
#ifdef LW_NO_DMA
                if (false) {
#else
                if (staged < ntile) {
#endif
                    stage_next();
                    ++staged;
                    issued += kPerTile;
                }
                mk[LEAD - 1] = issued;                           // mark of tile i + LEAD; the marks move up by one tile
#pragma unroll
                for (int k = 0; k + 1 < LEAD; ++k) mk[k] = mk[k + 1];
#ifdef LW_NO_STORE
                if (false) {
#else
                if (i > 0) {                                     // output tile i - 1 to memory
#endif
                    store_tile(obp, (i - 1) & 1);
                    issued += kPerStore;
                    so_off += o_step;
                    so_rows += tile_rows;
                }
            }
            // start values of tile i + 1, one channel tile per step: read at step kMid + 1 + ti, made at the next
            if (ks == kMid + 1) prep_stats(xn);
            if (ks >= kMid + 2 && ks <= kMid + 1 + kTI) prep_make(ks - kMid - 2);
            if (ks >= kMid + 1 && ks <= kMid + kTI) prep_read(ks - kMid - 1, on);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ti = 0; ti < kTI; ++ti) {
#ifdef LW_NO_MFMA
                if (ks == 0) acc[ti] = nxt[ti];
                if (ks == kKS - 1) acc[ti][0] += bf2f(xq[ks % kXD][0]);
                continue;
#endif
                if (ks == 0) {
                    // (nxt is rewritten only from step kMid + 2 on: the early-clobber output keeps acc and nxt apart)
                    if (ti < kTA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc[ti]) : "a"(wf[ti][ks]), "v"(xq[ks % kXD]), "v"(nxt[ti]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc[ti]) : "v"(wf[ti][ks]), "v"(xq[ks % kXD]), "v"(nxt[ti]));
                } else {
#if defined(LW_ALL_V)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ti]) : "v"(wf[3 + (ti & 1)][ks]), "v"(xq[ks % kXD]));
#elif defined(LW_ALL_A)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ti]) : "a"(wf[ti % 3][ks]), "v"(xq[ks % kXD]));
#else
                    if (ti < kTA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ti]) : "a"(wf[ti][ks]), "v"(xq[ks % kXD]));
                    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[ti]) : "v"(wf[ti][ks]), "v"(xq[ks % kXD]));
#endif
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (the MFMAs are asm: the compiler does not know that the accumulators come out of the matrix pipe — the wait states a
        //  VALU read of a four-pass MFMA result needs are spelled out here)
        asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));
        // bf16 cells of this wave's 80 channels into the output tile; row sums of the rounded values
        {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int ti = 0; ti < kTI; ++ti) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = f2bf(LNF ? acc[ti][e] * rstd : acc[ti][e]);
                    if constexpr (RSUM) {
                        const float v = bf2f(o[e]);
                        s += v;
                        q += v * v;
                    }
                }
                *(bf16x4*)(ot + olane[ti]) = o;
            }
            if constexpr (RSUM) {
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
        obp = ob;
        xb = xbn;
        ob = obn;
    }
    w_barrier();
    store_tile(obp, (ntile - 1) & 1);
}

}  // namespace

// plain Linear with K = 640 onto whole 320-channel slices over whole 16-pixel tiles: bias, one residual, row_sums, or the folded LayerNorm
bool cc_lin640w_applicable(const CcGemmDesc& d) {
    const bool lnf = d.ln_stats || d.ln_sums;
    return d.mode == CCEDIT_GEMM_LINEAR && d.taps == 1 && d.A2 == nullptr && d.Cin == kK && d.Kpad == kK && d.N % kSlice == 0 &&
           d.N / kSlice <= 32 && d.M % kP == 0 && d.gn_stats == nullptr && d.res2 == nullptr && d.group_bias == nullptr && !d.out_f32 &&
           d.act == CCEDIT_ACT_NONE && d.ldc % 8 == 0 && d.lda % 8 == 0 && (d.res1 == nullptr || d.ldr1 % 8 == 0) && d.ln_eps == 0.f &&
           (!lnf || (d.ln_colsum && !(d.ln_stats && d.ln_sums) && !d.res1 && !d.row_sums)) && (lnf || !d.ln_colsum);
}

template <int RES, int LNF, int RSUM>
static int launch_w(const CcGemmDesc& d, hipStream_t s, int nslice, int pt_n) {
    static unsigned long long attr_done = 0;
    constexpr int lds = Ring<RES>::kLds;
    if (int rc = cc_max_dynamic_lds((const void*)lin640w_kernel<RES, LNF, RSUM>, lds, &attr_done, "lin640w")) return rc;
    hipLaunchKernelGGL((lin640w_kernel<RES, LNF, RSUM>), dim3(256), dim3(kNT), lds, s, d, nslice, pt_n);
    return cc_launch_status("lin640w_kernel");
}

int cc_lin640w_launch(const CcGemmDesc& d, hipStream_t s) {
    const int64_t pt_n = d.M / kP;
    if (pt_n > 2147483647LL) {
        cc_set_error("ccedit_gemm: grid too large");
        return CCEDIT_EUNSUPPORTED;
    }
    cc_note_kernel("lin640w_kernel");
    const int nslice = d.N / kSlice;
    const bool rs = d.row_sums != nullptr;
    if (d.res1) return rs ? launch_w<1, 0, 1>(d, s, nslice, (int)pt_n) : launch_w<1, 0, 0>(d, s, nslice, (int)pt_n);
    if (d.ln_stats || d.ln_sums) return launch_w<0, 1, 0>(d, s, nslice, (int)pt_n);
    return rs ? launch_w<0, 0, 1>(d, s, nslice, (int)pt_n) : launch_w<0, 0, 0>(d, s, nslice, (int)pt_n);
}
