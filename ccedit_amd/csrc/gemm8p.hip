// Persistent 256ch x 256pix bf16 GEMM for the plain Linear layers of the path (tile 11): eight waves, K tiles of 64,
// a two-buffer LDS ring of sixteen-KB half tiles filled by global_load_lds, and an eight-phase K loop in which the two
// halves of the workgroup run one barrier apart — while waves 0-3 issue the MFMAs of a phase, waves 4-7 read the
// fragments of theirs and request the next half tile, then the roles swap.
//
//   out[m][n] = epilogue( sum_k W[n][k] * A[m][k] )        (Linear / Conv 1x1 / Conv1d k1; attention.py:115-141, 377-383)
//
// Why another kernel next to tap_gemm_kernel: the FF / GEGLU projections of the 32x48 and 16x24 levels and the C -> C
// Linears are plain long GEMMs (27 ms of a 120 ms step); on those the two-barrier 128x128 loop of tap_gemm tops out at
// 650-900 TF/s while a library GEMM reaches 1030-1250 on the same box.  What this kernel changes:
//   * 256 x 256 tile: 128 FLOP per byte staged (64 for 128 x 128), so the ~20 B/clk a CU can pull through L1 misses
//     stops being the ceiling;
//   * operand half tiles stay in flight ACROSS barriers: the only vmcnt wait of a K tile is `vmcnt(6)` in its last
//     phase (three half tiles = 48 KB still travelling); nothing in the loop drains the queue;
//   * every half tile is read in exactly ONE phase by all eight waves (a wave's 128 x 64 output is two 64-row bands,
//     one from each A half, times two 32-pixel bands, one from each B half), so a half-tile slot is free one or two
//     phases after it was read and is re-filled four to six phases before it is read again;
//   * the wave halves are offset by one barrier (ping-pong): LDS fragment reads and DMA issue of one half run under the
//     MFMAs of the other, `s_setprio` keeps the matrix pipe with the half that is in its MFMA block;
//   * persistent workgroups (one per CU): the first K tile of the NEXT output tile is requested (buffer 0) before the epilogue of
//     the current one, so the cold-operand latency (~1.5 us in the network) hides under the stores; the second after it;
//   * bias (+ the per-clip embedding row) is the accumulators' initial value; the epilogue turns each wave's 128-channel x 32-pixel
//     blocks around in 8 KB of its OWN inside buffer 1 (a wave's LDS accesses execute in order: no workgroup barrier) and stores
//     16 bytes per lane along a pixel's channels — whole 128-byte lines; GEGLU before the staging, residuals after it, GroupNorm /
//     LayerNorm statistics from the bf16 values written.  (The first version went registers -> global through v_permlane32_swap:
//     8.2 us per tile against 2.9)
//
// LDS: [buffer 0 | buffer 1] x [A half 0 | A half 1 | B half 0 | B half 1], a half = 128 rows x 128 B (64 k).  Rows are
// written lane-linearly by the DMA; 16-byte granule g of row r sits at slot g ^ ((r >> 1) & 7) (source-side swizzle,
// same involution on the fragment read: conflict-free for the 32x32x16 fragment pattern, see gemm.hip).
//
// Hazards, in phase numbers j (K tile t = j / 4; a phase = fragment reads + one half-tile request, barrier, 8 MFMAs,
// barrier; the second wave half runs every step one barrier later):
// (X = B, Y = A for the 256 x 256 shape; the 128 x 512 shape swaps the roles)
//   reads   phase 4t: X half 0 then Y half 0;  4t+1: X half 1;  4t+2: Y half 1;  4t+3: none (X half 0 is still in registers)
//   refills phase 4t: Y1 of tile t+1;  4t+1: X0 of t+2;  4t+2: Y0 of t+2;  4t+3: X1 of t+2, then vmcnt(6)
//   RAW: tile t+1 is complete when every wave has passed the vmcnt(6) of phase 4t+3 (it leaves only the three requests
//        of phases 4t+1..4t+3 in flight) and a barrier; its first read is in phase 4t+4.
//   WAR: X0 is re-requested one phase after its reads, which `lgkmcnt(8)` retires BEFORE the reading phase's first
//        barrier (they are issued first); every other slot is re-requested two phases after its reads, whose results the
//        MFMAs of the reading phase consumed before that phase's second barrier.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

template <int N>
__device__ __forceinline__ void g8_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void g8_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
// Split-K slots cross workgroups (possibly XCDs, i.e. L2s): written and read with agent-scope accesses (sc1: write-through /
// read past the non-coherent levels) instead of fences — a release fence is a write-back of the whole 4 MB L2 per wave, which made
// the first version of the hand-over cost more than the K loop it shortened.  The loads are invisible to the compiler's counters:
// the caller waits (g8_vmcnt<0>) before using the values.
__device__ __forceinline__ void g8_store_agent(f32x4* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 g8_load_agent(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ void g8_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}


struct G8Order {          // 32-bit on purpose: the tile walk runs once per output tile in every wave (M < 2^31 rows)
    int pt_n, ct_n, Q, total, per;
};

// Tile walk.  All pt_n x ct_n tiles in ONE linear order — channel tiles in groups of Q; inside a group pixel-major, channel-minor
// — cut into eight equal contiguous ranges, one per XCD (workgroup b runs on XCD b % 8).  The workgroups of an XCD take the
// tiles of its range round-robin, so the ~32 that run together cover (32 / Q pixel tiles) x (Q channel tiles): the weight rows of
// a group (Q x 256 x K, chosen by the launcher to fit the XCD's L2 with room to spare) are fetched once per XCD and group,
// every activation tile once per group.  Q = ct_n when the whole weight matrix fits.
template <int BM>
__device__ __forceinline__ G8Order g8_order(const CcGemmDesc& d, int bn) {
    G8Order o;
    o.ct_n = (d.N + BM - 1) / BM;
    o.pt_n = (int)((d.M + bn - 1) / bn);
    o.Q = (d.cgroup & 0xFFFF) > 0 ? (d.cgroup & 0xFFFF) : o.ct_n;
    o.total = o.pt_n * o.ct_n;
    o.per = (o.total + 7) >> 3;
    return o;
}

__device__ __forceinline__ void g8_decode(const G8Order& o, int g, int& pt, int& ct) {
    const int gfull = o.pt_n * o.Q;
    const int cg = g / gfull;
    const int r = g - cg * gfull;
    const int qn = min(o.Q, o.ct_n - cg * o.Q);
    pt = r / qn;
    ct = cg * o.Q + (r - pt * qn);
}

// TIH / TJH: 32-row MFMA tiles per wave and operand HALF.  A half = 2 wave rows x TIH x 32 channels, B half = 4 wave columns x
// TJH x 32 pixels; the block computes (TIH x 128) channels x (TJH x 256) pixels, a wave 2 TIH x 2 TJH accumulator tiles:
//   <2, 1>  256ch x 256pix, 128 KB LDS           — Cout a multiple of 256
//   <1, 2>  128ch x 512pix, 160 KB LDS           — Cout = 640 (5 tiles instead of 2.5), 102 instead of 128 FLOP per staged byte
enum { G8_PLAIN = 0, G8_RES = 1, G8_GEGLU = 2 };      // epilogue variants (compiled separately: one register budget each)

__device__ __attribute__((aligned(64))) const char g8_zero_page[64] = {0};

// GroupNorm(32, N) statistics of the tensor being written (CcGemmDesc.gn_stats): gs / gq = this lane's sums / sums of squares of
// the bf16-rounded outputs of channels cb .. cb + 7 over its rows.  Lanes with the same channels sit `G` apart: butterfly over
// those, then the first G lanes add their (at most two) groups' shares to the frame's double-precision slots — the arrival
// order of double adds cannot move the fp32 mean / rstd taken from them (same argument as gemm_epilogue.h).
template <int G>
__device__ __forceinline__ void g8_flush_stats(float (&gs)[8], float (&gq)[8], int lane, int cb, int N, double* slots) {
#pragma unroll
    for (int off = G; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gs[e] += __shfl_xor(gs[e], off);
            gq[e] += __shfl_xor(gq[e], off);
        }
    if (lane < G && cb < N) {
        const int cpg = N >> 5, g0 = cb / cpg;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool first = cb + e < (g0 + 1) * cpg;
            s0 += first ? gs[e] : 0.f;
            q0 += first ? gq[e] : 0.f;
            s1 += first ? 0.f : gs[e];
            q1 += first ? 0.f : gq[e];
        }
        unsafeAtomicAdd(slots + 2 * g0, (double)s0);
        unsafeAtomicAdd(slots + 2 * g0 + 1, (double)q0);
        if (g0 < 31 && (cb + 7) / cpg > g0) {
            unsafeAtomicAdd(slots + 2 * g0 + 2, (double)s1);
            unsafeAtomicAdd(slots + 2 * g0 + 3, (double)q1);
        }
    }
}

// LayerNorm statistics of the tensor being written (CcGemmDesc.row_sums): rs / rq = this lane's sums / sums of squares over its
// channels of pixel rows j * step + lane / G (j = 0 .. NR - 1) of the 32-pixel tile starting at row m0.  The G lanes of a pixel
// row are neighbours: butterfly over them, then the first of each adds the wave's share of the row to the row's double-precision
// (sum, sum of squares).  The channel tiles of a row meet in DOUBLE atomics: arrival order moves a total by ~1e-16 relative, which
// reaches the fp32 (mean, rstd) derived from it only where that double value sits within 1e-16 of an fp32 rounding boundary —
// not excluded, never observed (tests/test_fullsize_gpu.py compares whole evaluations bit for bit across processes), and unlike
// split-K (fixed summation order) not guaranteed.  The variance is q / K - mean^2 in double from fp32 lane partials of bf16 values:
// for |mean| >> std it is as accurate as the partials (test_layernorm_folded_into_persistent_gemm runs rows with mean = 30 std).
template <int G, int NR>
__device__ __forceinline__ void g8_flush_rows(float (&rs)[NR], float (&rq)[NR], int lane, int64_t m0, int step, int64_t M, double* sums) {
#pragma unroll
    for (int off = 1; off < G; off <<= 1)
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            rs[j] += __shfl_xor(rs[j], off);
            rq[j] += __shfl_xor(rq[j], off);
        }
    if ((lane & (G - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const int64_t m = m0 + j * step + lane / G;
            if (m < M) {
                unsafeAtomicAdd(sums + 2 * m, (double)rs[j]);
                unsafeAtomicAdd(sums + 2 * m + 1, (double)rq[j]);
            }
        }
    }
}

// GATHER: implicit-GEMM convolutions, K = [Cin / 64][taps][64] (weight K order 1).  K tile kt reads channel chunk kt / taps of the
// pixel's row shifted by the tap: Conv1d k3 over the T keyframes of a clip (openaimodel.py:617-629, 674-687; rows HW apart,
// zeros outside the clip) or Conv2d 3x3 stride 1 pad 1 (openaimodel.py:445-449, 483-492; rows dy * W + dx apart, zeros outside the
// frame).  Only the activation request changes: a uniform row offset per tap and one select (row or zero page) per 16-byte piece;
// which taps a thread's rows have is a 9-bit mask per row, computed once per output tile.
// G8_SUBPIX (round 6): one output parity of `conv3x3(nearest_upsample_2x(x))` (CcGemmDesc.subpix, Upsample.forward,
// openaimodel.py:204-217 / 254-263) — four taps of a 2 x 2 window on the LOW-resolution source starting at (oy - 1 + py, ox - 1 + px),
// K = [Cin / 64][4][64]; the output row of source pixel m is (2 oy + py, 2 ox + px) of the up-sampled frame (cc_out_row).
enum { G8_LINEAR = 0, G8_TEMPORAL = 1, G8_CONV3 = 2, G8_SUBPIX = 3 };
__device__ __forceinline__ size_t g8_out_row(const CcGemmDesc& d, int64_t m) {       // (cc_out_row of gemm_epilogue.h)
    const int hw = d.Hout * d.Wout;
    const int n = (int)(m / hw), rem = (int)(m - (int64_t)n * hw);
    const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
    return ((size_t)n * 2 * d.Hout + 2 * oy + ((d.subpix - 1) >> 1)) * (2 * d.Wout) + 2 * ox + ((d.subpix - 1) & 1);
}

// SPLIT = 1: split-K for outputs with far fewer tiles than CUs (the 8x12 level: 3264 pixels x 1280 channels = 65 tiles, K loops of
// 60-360 K tiles).  A work item is (tile, split); the d.split_k items of a tile are consecutive in the walk (same XCD), each
// runs a contiguous share of the K tiles from zero accumulators and writes them, in register order, to its slot of d.workspace;
// an arrival counter per tile elects the LAST arriver, which starts from the bias, adds the slots in split order 0, 1, 2, ...
// (a fixed summation order whoever arrives last: results do not depend on timing) and runs the ordinary epilogue.
// LNF = 1 (plain Linear, no residual): the rows of A are LayerNorm inputs that were NOT normalised.  With gamma folded into the
// weights (W' = W diag(gamma), b' = b + W beta) and the row statistics (mean, rstd) in d.ln_stats,
//     W' xhat + b' = rstd (W' x - mean W' 1) + b'
// so the accumulators start at b' / rstd - mean * colsum(W') and the epilogue multiplies by rstd: `to_q(norm(x))`, the GEGLU
// projection of `ff(norm(x))` (attention.py:695-716) without the normalised tensor ever being written or read.
template <int TIH, int TJH, int EPI, int GATHER, int SPLIT = 0, int LNF = 0>
__global__ __launch_bounds__(512) void g8_kernel(const CcGemmDesc d) {
    static_assert(TIH * TJH == 2, "eight MFMAs per phase");
    constexpr int BM = TIH * 128, BN = TJH * 256;
    constexpr int AH = TIH * 8192, BH = TJH * 16384;          // bytes of an A / B half tile (128-byte rows)
    constexpr int BUF = 2 * AH + 2 * BH;                      // [A0 | A1 | B0 | B1]
    constexpr int AI = TIH, BI = 2 * TJH;                     // 64-row DMA issues per half tile
    constexpr bool XA = TIH < TJH;                            // which operand's half 0 stays in registers through a K tile (see ktile)
    constexpr int KEEP = XA ? 2 * AI + BI : 2 * BI + AI;      // DMA instructions of the three half tiles that stay in flight
    constexpr int NI = 2 * TIH, NJ = 2 * TJH;                 // accumulator tiles per wave
    constexpr int CW = NI * 32;                               // consecutive output channels per wave row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nk_all = d.Kpad >> 6;
    const int flags = d.cgroup >> 24;           // tuning only: 1 = no output stores, 2 = next tile requested AFTER the epilogue
    const int S = SPLIT ? d.split_k : 1;
    int nk = nk_all, kbase = 0, sp_cur = 0;     // this work item's K tiles: [kbase, kbase + nk), its split index

    const G8Order ord = g8_order<BM>(d, BN);
    const int xcd = blockIdx.x & 7;
    const int nw = gridDim.x >> 3;              // workgroups per XCD
    const int items = ord.total * S, per = SPLIT ? (items + 7) >> 3 : ord.per;
    int g = xcd * per + (blockIdx.x >> 3);
    const int gend = min((xcd + 1) * per, items);
    if (g >= gend) return;
    auto decode = [&](int g_, int& pt_, int& ct_) {
        if constexpr (SPLIT) {
            const int tile = g_ / S, sp = g_ - tile * S;
            g8_decode(ord, tile, pt_, ct_);
            sp_cur = sp;
            kbase = (int)((int64_t)sp * nk_all / S);
            nk = (int)((int64_t)(sp + 1) * nk_all / S) - kbase;
        } else {
            g8_decode(ord, g_, pt_, ct_);
        }
    };

    // ---- staging: thread -> (row rsub of a 64-row issue, LDS slot p), source granule p ^ ((rsub >> 1) & 7) ----
    const int p = tid & 7, rsub = tid >> 3;
    const int gcol = p ^ ((rsub >> 1) & 7);
    // LDS row rho = i * 64 + rsub of A half h belongs to wave row rho / (TIH * 32); it holds weight row (= output channel)
    //   wave_row * CW + h * (TIH * 32) + rho % (TIH * 32),   CW = 2 * TIH * 32 channels per wave row,
    // so that the 2 TIH row tiles a wave accumulates are CW CONSECUTIVE channels (whole 128-byte lines in the epilogue).
    const int a_row = (rsub / (TIH * 32)) * CW + rsub % (TIH * 32);
    const uint32_t a_lane = ((uint32_t)a_row * (uint32_t)d.Kpad + gcol * 8) * 2;          // per-lane byte offset into W (issue 0, half 0)
    const char* const Wp = (const char*)d.W;
    const char* const Ap = (const char*)d.A;
    char* const lds_wave = smem + wave * 1024;

    // ---- fragment read addresses: row l31 of a 32-row MFMA tile, granule (2 ks + hi) ^ ((l31 >> 1) & 7) ----
    const int sw = (l31 >> 1) & 7;
    const char* fa[4];
    const char* fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int off = l31 * 128 + (((2 * ks + hi) ^ sw) << 4);
        fa[ks] = smem + wr * (TIH * 4096) + off;                 // A half h at + h * AH, row tile ti at + ti * 4096
        fb[ks] = smem + 2 * AH + wc * (TJH * 4096) + off;        // B half h at + h * BH, pixel tile tj at + tj * 4096
    }

    int pt, ct;
    decode(g, pt, ct);

    // per-tile source bases (uniform).  Pixel rows past M are clamped to M - 1 when the B tile is requested (computed from
    // valid memory, never stored): rmax = M - 1 - pix0 is the last valid row of the tile, >= BN - 1 except in the last one.
    const char* wt;
    const char* at;
    int rmax;
    const uint32_t ldab = (uint32_t)d.lda * 2, gcol16 = gcol * 16;
    constexpr int NTAP = GATHER == G8_CONV3 ? 9 : (GATHER == G8_SUBPIX ? 4 : 3);
    uint32_t tap_ok[(2 * BI + 2) / 3] = {};      // GATHER: 9 bits per row j * 64 + rsub of the tile (3 rows per word): bit = that tap exists
    const int64_t row_bytes = (int64_t)d.lda * 2;
    auto set_tile = [&](int pt_, int ct_) {
        wt = Wp + (size_t)ct_ * BM * d.Kpad * 2;
        const int64_t pix0 = (int64_t)pt_ * BN;
        at = Ap + (size_t)pix0 * d.lda * 2;
        const int64_t left = d.M - 1 - pix0;
        rmax = left < BN ? (int)left : BN;
        if constexpr (GATHER != G8_LINEAR) {
#pragma unroll
            for (int w = 0; w < (2 * BI + 2) / 3; ++w) tap_ok[w] = 0;
#pragma unroll
            for (int j = 0; j < 2 * BI; ++j) {
                const int64_t m = pix0 + min(j * 64 + rsub, rmax);
                uint32_t ok;
                if constexpr (GATHER == G8_TEMPORAL) {
                    const int fr = (int)(m / d.HW) % d.T;                                   // keyframe index inside the clip
                    ok = (uint32_t)(fr > 0) | 2u | ((uint32_t)(fr < d.T - 1) << 2);
                } else if constexpr (GATHER == G8_SUBPIX) {
                    const int rem = (int)(m % ((int64_t)d.Hin * d.Win));
                    const int y = rem / d.Win, x = rem - y * d.Win;
                    const int sy = y - 1 + ((d.subpix - 1) >> 1), sx = x - 1 + ((d.subpix - 1) & 1);   // window origin; tap = 2 dy + dx
                    const uint32_t xm = (uint32_t)(sx >= 0) | ((uint32_t)(sx + 1 < d.Win) << 1);
                    ok = (sy >= 0 ? xm : 0u) | (sy + 1 < d.Hin ? xm << 2 : 0u);
                } else {
                    const int rem = (int)(m % ((int64_t)d.Hin * d.Win));
                    const int y = rem / d.Win, x = rem - y * d.Win;
                    const uint32_t xm = (uint32_t)(x > 0) | 2u | ((uint32_t)(x < d.Win - 1) << 2);     // dx = -1, 0, +1
                    ok = (y > 0 ? xm : 0u) | (xm << 3) | (y < d.Hin - 1 ? xm << 6 : 0u);                 // tap = 3 (dy + 1) + (dx + 1)
                }
                tap_ok[j / 3] |= ok << (9 * (j % 3));
            }
        }
    };
    auto stage_a = [&](int h, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            // uniform part in SGPRs (readfirstlane keeps hipcc from turning it into per-lane 64-bit induction variables),
            // per-lane part a 32-bit VGPR offset: global_load_lds_dwordx4 v, s[..]
            const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)((i * (64 / (TIH * 32)) * CW + h * (TIH * 32)) * d.Kpad * 2 + (kt + kbase) * 128));
            glds16(wt + u + a_lane, lds_wave + buf * BUF + h * AH + i * 8192);
        }
    };
    auto stage_b = [&](int h, int kt_, int buf) {
        const int kt = kt_ + kbase;
        if constexpr (GATHER != G8_LINEAR) {
            const int chunk = kt / NTAP, tap = kt - NTAP * chunk;                  // uniform
            const int shift = GATHER == G8_TEMPORAL ? (tap - 1) * d.HW
                              : (GATHER == G8_SUBPIX ? ((tap >> 1) - 1 + ((d.subpix - 1) >> 1)) * d.Win + ((tap & 1) - 1 + ((d.subpix - 1) & 1))
                                                     : (tap / 3 - 1) * d.Win + (tap % 3 - 1));       // rows
            const char* const base = at + (int64_t)shift * row_bytes + chunk * 128;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int j = h * BI + i;
                const int r = min(j * 64 + rsub, rmax);
                const char* src = base + ((uint32_t)r * ldab + gcol16);
                src = ((tap_ok[j / 3] >> (9 * (j % 3) + tap)) & 1) ? src : g8_zero_page;
                glds16(src, lds_wave + buf * BUF + 2 * AH + h * BH + i * 8192);
            }
        } else {
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)(kt * 128));
                const int r = min((h * BI + i) * 64 + rsub, rmax);
                glds16(at + u + ((uint32_t)r * ldab + gcol16), lds_wave + buf * BUF + 2 * AH + h * BH + i * 8192);
            }
        }
    };
    auto prologue_a = [&]() {          // K tile 0 -> buffer 0
        stage_a(0, 0, 0);
        stage_a(1, 0, 0);
        stage_b(0, 0, 0);
        stage_b(1, 0, 0);
    };
    auto prologue_b = [&]() {          // the three half tiles of K tile 1 that are in flight when the loop starts -> buffer 1
        if constexpr (XA) {
            stage_a(0, 1, 1);
            stage_b(0, 1, 1);
            stage_a(1, 1, 1);
        } else {
            stage_b(0, 1, 1);
            stage_a(0, 1, 1);
            stage_b(1, 1, 1);
        }
    };
#ifdef G8_PROBE
    unsigned long long* const probe = (unsigned long long*)d.gn_stats + (size_t)blockIdx.x * 64;
    int probe_n = 0;
#define G8_STAMP()                                                            \
    do {                                                                      \
        if (probe && tid == 0 && probe_n < 64) probe[probe_n++] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define G8_STAMP() \
    do {           \
    } while (0)
#endif

    // The bias is the INITIAL VALUE of the accumulators: register 4 q + e of row tile tf holds channel 32 tf + 8 q + 4 hi + e of
    // this wave row, the same for every pixel tile — 16 loads of 16 bytes per lane and output tile straight into the accumulator
    // registers, requested before the operand requests whose counted wait also covers them.  The epilogue never touches it.
    // LNF: (mean, rstd) of row m — as ccedit_row_stats left them (ln_stats), or from the (sum, sum of squares) a producing GEMM's
    // epilogue accumulated in double (ln_sums, see row_sums below): mean = s / K, var = q / K - mean^2 in double
    auto ln_row = [&](int64_t m) -> f32x2 {
        m = min(m, d.M - 1);
        if (d.ln_sums) {
            const double inv_k = 1.0 / (double)d.Cin;
            const double mu = d.ln_sums[2 * m] * inv_k;
            const double var = fmax(d.ln_sums[2 * m + 1] * inv_k - mu * mu, 0.0);
            return f32x2{(float)mu, rsqrtf((float)var + d.ln_sums_eps)};
        }
        return *(const f32x2*)(d.ln_stats + 2 * m);
    };
    f32x16 acc[NI][NJ];
    auto pixbase = [&](int tjf) { return (tjf / TJH) * (TJH * 128) + wc * (TJH * 32) + (tjf % TJH) * 32; };
    auto init_acc = [&](int pt_, int ct_, bool with_bias = !SPLIT) {
        if (!with_bias) {           // split-K partial: the bias is the reducer's starting value
#pragma unroll
            for (int tf = 0; tf < NI; ++tf)
#pragma unroll
                for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tf][tj][e] = 0.f;
            return;
        }
        if constexpr (LNF) {
            float mean[NJ], invr[NJ];
#pragma unroll
            for (int tj = 0; tj < NJ; ++tj) {
                const f32x2 st = ln_row((int64_t)pt_ * BN + pixbase(tj) + l31);
                mean[tj] = st[0];
                invr[tj] = 1.0f / st[1];
            }
#pragma unroll
            for (int tf = 0; tf < NI; ++tf)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = min(ct_ * BM + wr * CW + 32 * tf + 8 * q + 4 * hi, d.N - 4);
                    f32x4 b = {0.f, 0.f, 0.f, 0.f};
                    if (d.bias) b = *(const f32x4*)(d.bias + c);
                    const f32x4 cs = *(const f32x4*)(d.ln_colsum + c);
#pragma unroll
                    for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tf][tj][4 * q + e] = b[e] * invr[tj] - mean[tj] * cs[e];
                }
            return;
        }
        const float* const bias = d.bias;
        const float* const gbias = d.group_bias;       // + per-clip row bias (ResBlock: h + emb_out): a 32-pixel tile lies in ONE clip
        const int ldgb = d.ldgb ? d.ldgb : d.N;
#pragma unroll
        for (int tf = 0; tf < NI; ++tf)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = min(ct_ * BM + wr * CW + 32 * tf + 8 * q + 4 * hi, d.N - 4);          // (past N: never stored)
                f32x4 b = {0.f, 0.f, 0.f, 0.f};
                if (bias) b = *(const f32x4*)(bias + c);
                if (gbias) {
#pragma unroll
                    for (int tj = 0; tj < NJ; ++tj) {
                        const int64_t m = min((int64_t)pt_ * BN + pixbase(tj), d.M - 1);
                        const f32x4 gb = *(const f32x4*)(gbias + (size_t)(m / d.group_rows) * ldgb + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tf][tj][4 * q + e] = b[e] + gb[e];
                    }
                } else {
#pragma unroll
                    for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tf][tj][4 * q + e] = b[e];
                }
            }
    };

    G8_STAMP();
    set_tile(pt, ct);
    prologue_a();
    prologue_b();
    init_acc(pt, ct);
    g8_vmcnt<KEEP>();                            // K tile 0 has landed (this wave's part)

    for (;;) {
        const int64_t pix0 = (int64_t)pt * BN;
        const int ch0 = ct * BM;

        g8_barrier();                            // ... and everybody else's
        G8_STAMP();
        if (wr == 1) g8_barrier();               // second wave half: one barrier behind from here on

        // ---- K loop: one K tile = four phases on LDS buffer CUR.  X = the operand with the smaller per-wave fragment set (B for
        // 256 x 256, A for 128 x 512): its half 0 stays in registers for phases 0..3, half 1 for phases 1..2; the other operand
        // (Y) is read half 0 in phase 0, half 1 in phase 2.  64 fragment VGPRs at the peak.
        auto ktile = [&](auto CURC, int t) {
            constexpr int CUR = decltype(CURC)::value;
            constexpr int BASE = CUR * BUF;
            constexpr int NX = XA ? TIH : TJH, NY = XA ? TJH : TIH;
            constexpr int XH = XA ? AH : BH, YH = XA ? BH : AH;
            bf16x8 x0[NX][4], x1[NX][4], y[NY][4];
            auto read_x = [&](int h, bf16x8(&f)[NX][4]) {
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) f[i][ks] = *(const bf16x8*)((XA ? fa[ks] : fb[ks]) + BASE + h * XH + i * 4096);
            };
            auto read_y = [&](int h) {
#pragma unroll
                for (int i = 0; i < NY; ++i)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) y[i][ks] = *(const bf16x8*)((XA ? fb[ks] : fa[ks]) + BASE + h * YH + i * 4096);
            };
            auto stage_x = [&](int h, int kt, int buf) {
                if constexpr (XA) stage_a(h, kt, buf);
                else stage_b(h, kt, buf);
            };
            auto stage_y = [&](int h, int kt, int buf) {
                if constexpr (XA) stage_b(h, kt, buf);
                else stage_a(h, kt, buf);
            };
            auto mma = [&](const bf16x8(&xf)[NX][4], int xh, int yh) {
                const int ah = XA ? xh : yh, bh = XA ? yh : xh;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int ti = 0; ti < TIH; ++ti)
#pragma unroll
                        for (int tj = 0; tj < TJH; ++tj)
                            acc[ah * TIH + ti][bh * TJH + tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                XA ? xf[ti][ks] : y[ti][ks], XA ? y[tj][ks] : xf[tj][ks], acc[ah * TIH + ti][bh * TJH + tj], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            };
            // phase 0: X half 0 (first: retired by the lgkmcnt before the barrier), Y half 0; request Y1 of tile t + 1
            read_x(0, x0);
            __builtin_amdgcn_sched_barrier(0);
            read_y(0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nk) stage_y(1, t + 1, CUR ^ 1);
            g8_lgkmcnt<NY * 4>();
            g8_barrier();
            mma(x0, 0, 0);
            g8_barrier();
            // phase 1: X half 1; request X0 of tile t + 2
            read_x(1, x1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) stage_x(0, t + 2, CUR);
            g8_barrier();
            mma(x1, 1, 0);
            g8_barrier();
            // phase 2: Y half 1; request Y0 of tile t + 2
            read_y(1);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) stage_y(0, t + 2, CUR);
            g8_barrier();
            mma(x1, 1, 1);
            g8_barrier();
            // phase 3: no reads; request X1 of tile t + 2; tile t + 1 must have landed before the next phase reads it
            if (t + 2 < nk) {
                stage_x(1, t + 2, CUR);
                g8_vmcnt<KEEP>();
            } else {
                g8_vmcnt<0>();
            }
            g8_barrier();
            mma(x0, 0, 1);
            g8_barrier();
        };
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            ktile(std::integral_constant<int, 0>{}, t);
            ktile(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < nk) ktile(std::integral_constant<int, 0>{}, t);
        if (wr == 0) g8_barrier();               // both halves level again: every fragment read of this tile is done
        G8_STAMP();

        // ---- residual epilogue: the first residual's rows of the first two 64-channel x 32-pixel blocks are requested NOW —
        // ahead of the next tile's operand requests in the memory queue; the other blocks two blocks ahead of their use ----
        const bf16* __restrict__ r1 = (const bf16*)d.res1;
        const bf16* __restrict__ r2 = (const bf16*)d.res2;
        constexpr int NSB = NJ * (CW / 64);                  // blocks: sb = tjf * (CW / 64) + cs
        bf16x8 rv1[EPI == G8_RES ? NSB : 1][4];
        auto load_r1 = [&](int sb) {          // unconditional (rows / channels clamped into the tensor): a conditional load would keep the
            const int tjf = sb / (CW / 64), cs = sb % (CW / 64);      // registers alive around the whole persistent loop.  r1 != null here
#pragma unroll                                                        // (the launcher moves a lone second residual into the first slot)
            for (int j = 0; j < 4; ++j) {
                const int64_t m = min(pix0 + pixbase(tjf) + 8 * j + (lane >> 3), d.M - 1);
                const int cb = min(ch0 + wr * CW + cs * 64 + 8 * (lane & 7), d.N - 8);
                rv1[sb][j] = *(const bf16x8*)(r1 + (size_t)m * d.ldr1 + cb);
            }
        };
        if constexpr (EPI == G8_RES) {
            load_r1(0);
            load_r1(1);
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- next tile: request its first K tile now (buffer 0), it lands under the epilogue; buffer 1 is the epilogue's ----
        g += nw;
        const bool more = g < gend;
        int npt = 0, nct = 0;
        const int sp = sp_cur;
        if (more) {
            decode(g, npt, nct);
            set_tile(npt, nct);
            if (!(flags & 2)) prologue_a();
        }
        bool do_epi = true;
        if constexpr (SPLIT) {
            // ---- split-K hand-over.  Slot of (tile, split): 8 waves x NI*NJ*4 quads x 64 lanes of f32x4, register order ----
            const int tile = ct * ord.pt_n + pt;                      // any bijection of (pt, ct) onto [0, tiles)
            int* const cnt = (int*)d.workspace;
            f32x4* const part = (f32x4*)((char*)d.workspace + 4096) + (size_t)tile * S * (8 * NI * NJ * 4 * 64);
            f32x4* const mine = part + ((size_t)sp * 8 + wave) * (NI * NJ * 4 * 64) + lane;
#pragma unroll
            for (int tf = 0; tf < NI; ++tf)
#pragma unroll
                for (int tj = 0; tj < NJ; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[tf][tj][4 * q + e];
                        g8_store_agent(mine + ((tf * NJ + tj) * 4 + q) * 64, v);
                    }
            g8_vmcnt<0>();                                            // this wave's slot has left the CU (agent-scope stores) ...
            __syncthreads();                                          // ... and everybody else's, before the count
            int* const flag = (int*)(smem + 2 * BUF);
            if (tid == 0) *flag = atomicAdd(cnt + tile, 1);
            __syncthreads();
            do_epi = *flag == S - 1;
            if (do_epi) {
                if (tid == 0) cnt[tile] = 0;                          // ready for the next launch on this stream
                init_acc(pt, ct, true);
                for (int s2 = 0; s2 < S; ++s2) {
                    const f32x4* const src = part + ((size_t)s2 * 8 + wave) * (NI * NJ * 4 * 64) + lane;
#pragma unroll
                    for (int tf = 0; tf < NI; ++tf)
#pragma unroll
                        for (int tj = 0; tj < NJ; ++tj) {
                            f32x4 v[4];                               // agent-scope loads: the slots as the other workgroups wrote them
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = g8_load_agent(src + ((tf * NJ + tj) * 4 + q) * 64);
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");   // (ties the uses below to the wait)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[tf][tj][4 * q + e] += v[q][e];
                        }
                }
            }
        }

        // ---- epilogue: accumulators -> wave-private LDS tile -> global, whole 128-byte lines per row.
        // In the 32x32 MFMA layout a lane owns 4 channels of ONE pixel, so a direct store instruction touches 64 different
        // 16-byte pieces (measured: 8 us per 256 x 256 tile, the request rate of the vector memory path).  Each wave turns its
        // CW-channel x 32-pixel blocks around in 8 KB of its own (no workgroup barrier: LDS executes a wave's accesses in order)
        // and then stores 16 bytes per lane with consecutive lanes walking a pixel's channels.
        bf16* __restrict__ outp = (bf16*)d.out;
        const bool st = !(flags & 1);
        char* const stg = smem + BUF + wave * 8192;
        const int chw = ch0 + wr * CW;                                   // this wave's first channel (packed row)
        // GroupNorm statistics: a wave's 32-pixel tile tjf lies inside one 128-pixel block of the output tile, and frames are whole
        // numbers of such blocks (gn_rows % 128 == 0, e.g. 384 at the 16x24 level): the frame is taken per tjf and flushed per tjf
#ifdef G8_PROBE
        const bool gn = false;                       // (probe builds borrow the gn_stats pointer for their stamps)
#else
        const bool gn = d.gn_stats != nullptr;
#endif
        auto gn_slots = [&](int tjf) { return d.gn_stats + (size_t)(min(pix0 + pixbase(tjf), d.M - 1) / d.gn_rows) * 64; };
        float ln_rstd[LNF ? NJ : 1];                 // LNF: rstd of this lane's pixel of every pixel tile (the accumulators hold W' x / 1 - ...)
        if constexpr (LNF) {
#pragma unroll
            for (int tj = 0; tj < NJ; ++tj) ln_rstd[tj] = ln_row(pix0 + pixbase(tj) + l31)[1];
        }
        // x * rstd as an explicit scalar v_mul_f32: left to the compiler, the SLP vectoriser pairs neighbouring accumulators into
        // v_pk_mul_f32 with rstd broadcast by op_sel — from the HIGH half of whatever register pair it landed in, which is the one
        // packed-fp32 form csrc/build.py refuses (not safe beside another stream's GEMM)
        auto ln_mul = [&](int tjf, float x) {
            if constexpr (LNF) {
                float y;
                asm("v_mul_f32_e32 %0, %1, %2" : "=v"(y) : "v"(ln_rstd[LNF ? tjf : 0]), "v"(x));
                return y;
            } else {
                return x;
            }
        };
        if (!do_epi) {
            // another split of this tile arrives later and writes the output
        } else if constexpr (EPI == G8_GEGLU) {
            // packed rows 16 g + [0, 8) are values, + [8, 16) their gates: a 32-row tile yields 16 output channels
            constexpr int RB = CW, G = RB / 16;                          // staged row: CW / 2 bf16 outputs of one pixel
#pragma unroll
            for (int tjf = 0; tjf < NJ; ++tjf) {
#pragma unroll
                for (int tf = 0; tf < NI; ++tf) {
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        bf16x4 o;
#pragma unroll
                        // (scalar fp32 GELU: the packed-fp32 form — v_pk_fma_f32 / v_pk_mul_f32, two values per issue slot — measured
                        //  the same within noise in the same-box A/B, 852 / 885 vs 836 / 900 TF/s on 52224 x 5120 <- 640)
                        for (int e = 0; e < 4; ++e) o[e] = f2bf(ln_mul(tjf, acc[tf][tjf][8 * gq + e]) * gelu_erf_f(ln_mul(tjf, acc[tf][tjf][8 * gq + 4 + e])));
                        *(bf16x4*)(stg + l31 * RB + (((2 * tf + gq) ^ (l31 & (G - 1))) << 4) + hi * 8) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < G / 2; ++j) {
                    const int row = j * (64 / G) + lane / G, c = lane % G;
                    const bf16x8 v = *(const bf16x8*)(stg + row * RB + ((c ^ (row & (G - 1))) << 4));
                    const int64_t m = pix0 + pixbase(tjf) + row;
                    if (st && m < d.M && chw + 16 * c < d.N) *(bf16x8*)(outp + (size_t)m * d.ldc + (chw >> 1) + 8 * c) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (EPI == G8_RES) {
            // fp32 staging, 64 channels x 32 pixels at a time: the residuals are added before the one rounding to bf16
#pragma unroll
            for (int tjf = 0; tjf < NJ; ++tjf) {
                float gs[CW / 64][8], gq[CW / 64][8];
                float rs[4] = {0.f, 0.f, 0.f, 0.f}, rq[4] = {0.f, 0.f, 0.f, 0.f};      // row_sums: this lane's share of its 4 pixel rows
#pragma unroll
                for (int cs = 0; cs < CW / 64; ++cs)
#pragma unroll
                    for (int e = 0; e < 8; ++e) gs[cs][e] = gq[cs][e] = 0.f;
#pragma unroll
                for (int cs = 0; cs < CW / 64; ++cs) {
                    const int cb64 = chw + cs * 64;
                    // this lane's part of the block: pixel rows 8 j + lane / 8, channels cb64 + 8 (lane % 8) .. + 7
                    const int c = lane & 7, cb = cb64 + 8 * c;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[2 * cs + u][tjf][4 * q + e];
                            *(f32x4*)(stg + l31 * 256 + (((8 * u + 2 * q + hi) ^ (l31 & 15)) << 4)) = v;
                        }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int row = 8 * j + (lane >> 3);
                        const f32x4 f0 = *(const f32x4*)(stg + row * 256 + (((2 * c) ^ (row & 15)) << 4));
                        const f32x4 f1 = *(const f32x4*)(stg + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
                        float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
                        const int64_t m = pix0 + pixbase(tjf) + row;
                        if (m < d.M && cb < d.N) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += bf2f(rv1[tjf * (CW / 64) + cs][j][e]);
                            if (r2) {
                                const bf16x8 rv2 = *(const bf16x8*)(r2 + (size_t)m * d.ldr2 + cb);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += bf2f(rv2[e]);
                            }
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                            if (st) *(bf16x8*)(outp + (size_t)m * d.ldc + cb) = o;
                            if constexpr (GATHER == G8_LINEAR) {
                                if (d.row_sums) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) {
                                        const float f = bf2f(o[e]);      // of what the consumer will read
                                        rs[j] += f;
                                        rq[j] += f * f;
                                    }
                                }
                            }
                            if (gn) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    const float f = bf2f(o[e]);      // statistics of what the consumer will read
                                    gs[cs][e] += f;
                                    gq[cs][e] += f * f;
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (tjf * (CW / 64) + cs + 2 < NSB) load_r1(tjf * (CW / 64) + cs + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (gn) {
#pragma unroll
                    for (int cs = 0; cs < CW / 64; ++cs)
                        g8_flush_stats<8>(gs[cs], gq[cs], lane, chw + cs * 64 + 8 * (lane & 7), d.N, gn_slots(tjf));
                }
                if constexpr (GATHER == G8_LINEAR) {
                    if (d.row_sums) g8_flush_rows<8>(rs, rq, lane, pix0 + pixbase(tjf), 8, d.M, d.row_sums);
                }
            }
        } else {
            // bf16 staging, all CW channels x 32 pixels at a time
            constexpr int RB = CW * 2, G = RB / 16;
#pragma unroll
            for (int tjf = 0; tjf < NJ; ++tjf) {
                float gs[8], gq[8];
                float rs[G / 2], rq[G / 2];                  // row_sums: this lane's share of its G / 2 pixel rows
#pragma unroll
                for (int j = 0; j < G / 2; ++j) rs[j] = rq[j] = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
#pragma unroll
                for (int tf = 0; tf < NI; ++tf)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = f2bf(ln_mul(tjf, acc[tf][tjf][4 * q + e]));
                        *(bf16x4*)(stg + l31 * RB + (((4 * tf + q) ^ (l31 & (G - 1))) << 4) + hi * 8) = o;
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < G / 2; ++j) {
                    const int row = j * (64 / G) + lane / G, c = lane % G;
                    const bf16x8 v = *(const bf16x8*)(stg + row * RB + ((c ^ (row & (G - 1))) << 4));
                    const int64_t m = pix0 + pixbase(tjf) + row;
                    if (st && m < d.M && chw + 8 * c < d.N) *(bf16x8*)(outp + (GATHER == G8_SUBPIX ? g8_out_row(d, m) : (size_t)m) * d.ldc + chw + 8 * c) = v;
                    if constexpr (GATHER == G8_LINEAR && !LNF) {
                        if (d.row_sums && chw + 8 * c < d.N) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float f = bf2f(v[e]);
                                rs[j] += f;
                                rq[j] += f * f;
                            }
                        }
                    }
                    if (gn && m < d.M) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float f = bf2f(v[e]);
                            gs[e] += f;
                            gq[e] += f * f;
                        }
                    }
                }
                if (gn) g8_flush_stats<G>(gs, gq, lane, chw + 8 * (lane % G), d.N, gn_slots(tjf));
                if constexpr (GATHER == G8_LINEAR && !LNF) {
                    if (d.row_sums) g8_flush_rows<G>(rs, rq, lane, pix0 + pixbase(tjf), 64 / G, d.M, d.row_sums);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        G8_STAMP();
        if (!more) return;
        pt = npt;
        ct = nct;
        init_acc(pt, ct);
        g8_barrier();                            // every wave is done with its staging block in buffer 1
        if (flags & 2) prologue_a();
        prologue_b();
        // K tile 0 of the next tile has landed.  Loads complete in order among themselves (stores are counted by vmcnt too, but
        // may be acknowledged in any order relative to loads).  K tile 0 was requested BEFORE the epilogue; younger loads are the
        // 16 bias loads of init_acc and the KEEP requests just issued.  If any K-tile-0 request were still outstanding, all of
        // those would be too, i.e. more than KEEP + 16 operations — so a count of at most KEEP + 12 (margin: the compiler may merge
        // bias loads) proves it has landed WITHOUT waiting for the epilogue's stores to be acknowledged (measured: ~1.5 us per tile).
        // Without a bias there is no such padding and the count is KEEP.
        if (!SPLIT && (d.bias || d.group_bias)) g8_vmcnt<KEEP + 12>();          // (split-K partials start from zero: no bias loads)
        else g8_vmcnt<KEEP>();
    }
}

template <int TIH, int TJH, int EPI, int GATHER, int SPLIT = 0, int LNF = 0>
int g8_launch_shape(const CcGemmDesc& d, hipStream_t s, int n_cu, int split_k = 1) {
    constexpr int BM = TIH * 128, BN = TJH * 256;
    constexpr int LDS = 2 * (2 * TIH * 8192 + 2 * TJH * 16384) + (SPLIT ? 16 : 0);
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)g8_kernel<TIH, TJH, EPI, GATHER, SPLIT, LNF>, LDS, &attr_done, "g8_kernel")) return rc;
    const int64_t pt_n = (d.M + BN - 1) / BN, ct_n = (d.N + BM - 1) / BM;
    CcGemmDesc dd = d;
    dd.cgroup = 0;
    if (!dd.res1 && dd.res2) {          // the residual epilogue always has a FIRST residual
        dd.res1 = dd.res2;
        dd.ldr1 = dd.ldr2;
        dd.res2 = nullptr;
    }
    // channel-tile groups when the weight matrix does not fit an XCD's 4 MB L2 beside the activation tiles in flight: the largest
    // group of at most ~2 MB of weight rows, evened out over the groups
    const double tile_bytes = (double)BM * d.Kpad * 2.0;
    if (ct_n * tile_bytes > 3.0 * 1024 * 1024) {
        int q = (int)(2.0 * 1024 * 1024 / tile_bytes);
        q = q < 2 ? 2 : q;
        if (q < ct_n) {
            const int ng = (int)((ct_n + q - 1) / q);
            dd.cgroup = (int)((ct_n + ng - 1) / ng);
        }
    }
#ifdef CCEDIT_TUNING      // probe builds only (-DCCEDIT_TUNING): the product library never reads a switch that changes results
    static const int flag_env = getenv("CCEDIT_G8_FLAGS") ? atoi(getenv("CCEDIT_G8_FLAGS")) : 0;   // see `flags` in the kernel
#else
    constexpr int flag_env = 0;
#endif
    dd.cgroup |= flag_env << 24;
    int wgs = n_cu - n_cu % 8;
    dd.split_k = SPLIT ? split_k : 1;
    const int64_t tiles = pt_n * ct_n * dd.split_k;
    if (tiles < wgs) wgs = (int)((tiles + 7) / 8 * 8);
    if (LNF)
        cc_note_kernel("g8_kernel %dch x %dpix, LayerNorm folded", BM, BN);
    else if (SPLIT)
        cc_note_kernel(GATHER == G8_TEMPORAL ? "g8_kernel %dch x %dpix, temporal taps, split-K" : (GATHER == G8_CONV3 ? "g8_kernel %dch x %dpix, 3x3 taps, split-K" : (GATHER == G8_SUBPIX ? "g8_kernel %dch x %dpix, upsample parity taps, split-K" : "g8_kernel %dch x %dpix, split-K")), BM, BN);
    else
        cc_note_kernel(GATHER == G8_TEMPORAL ? "g8_kernel %dch x %dpix, temporal taps" : (GATHER == G8_CONV3 ? "g8_kernel %dch x %dpix, 3x3 taps" : (GATHER == G8_SUBPIX ? "g8_kernel %dch x %dpix, upsample parity taps" : "g8_kernel %dch x %dpix")), BM, BN);
    hipLaunchKernelGGL((g8_kernel<TIH, TJH, EPI, GATHER, SPLIT, LNF>), dim3((unsigned)wgs), dim3(512), LDS, s, dd);
    return cc_launch_status("g8_kernel");
}

}  // namespace

// Pixels per tile of the block shape cc_g8_launch picks for this Cout (shape 0): 128ch x 512pix when Cout leaves half a 256-channel
// tile (640 = 2.5 tiles), else 256ch x 256pix
static int g8_auto_shape(const CcGemmDesc& d) {
    const int rem = d.N % 256;
    return (rem > 0 && rem <= 128 && d.N <= 1024) ? 2 : 1;
}

// shape: 0 = by Cout, 1 = 256ch x 256pix, 2 = 128ch x 512pix
bool cc_g8_applicable(const CcGemmDesc& d, int shape) {
    if (shape == 0) shape = g8_auto_shape(d);
    bool geo;
    if (d.mode == CCEDIT_GEMM_LINEAR)
        geo = d.taps == 1 && d.Kpad == d.Cin;
    else if (d.mode == CCEDIT_GEMM_TEMPORAL)
        geo = d.taps == 3 && d.korder == 1 && d.Kpad == 3 * d.Cin && d.T > 0 && d.HW > 0 &&
              (d.Tsrc == 0 || (d.Tsrc == d.T && d.tsrc_off == 0 && d.t0 == 0 && d.Tglob == d.T)) &&      // unsharded clips only
              d.act == CCEDIT_ACT_NONE;
    else if (d.subpix)      // one parity of upsample + conv 3x3 on the low-resolution source: 2 x 2 window, plain epilogue only
        geo = d.taps == 4 && d.ksize == 2 && d.korder == 1 && d.Kpad == 4 * d.Cin && d.stride == 1 && !d.upsample && !d.vpad &&
              d.Hin == d.Hout && d.Win == d.Wout && d.Hin > 1 && d.Win > 1 && d.act == CCEDIT_ACT_NONE && !d.res1 && !d.res2 &&
              !d.group_bias && !d.gn_stats && 4 * d.M < (1LL << 31);
    else          // Conv2d 3x3, stride 1, pad 1, same-size output, no fused upsample
        geo = !d.vpad && d.taps == 9 && d.ksize == 3 && d.korder == 1 && d.Kpad == 9 * d.Cin && d.stride == 1 && d.pad == 1 && !d.upsample &&
              d.Hin == d.Hout && d.Win == d.Wout && d.Hin > 1 && d.Win > 1 && d.act == CCEDIT_ACT_NONE;
    return geo && d.A2 == nullptr && d.Cin % 64 == 0 && d.Kpad >= 128 && d.N % 16 == 0 &&
           (d.act == CCEDIT_ACT_NONE || d.act == CCEDIT_ACT_GEGLU) && !d.out_f32 && d.ln_eps == 0.f &&
           ((!d.ln_stats && !d.ln_sums) || (d.ln_colsum && !(d.ln_stats && d.ln_sums) && d.mode == CCEDIT_GEMM_LINEAR && !d.res1 && !d.res2 &&
                                            !d.group_bias && !d.gn_stats && !d.row_sums)) &&
           (!d.row_sums || (d.mode == CCEDIT_GEMM_LINEAR && d.act == CCEDIT_ACT_NONE)) &&
           (!d.group_bias || (d.act == CCEDIT_ACT_NONE && d.group_rows > 0 && d.group_rows % 32 == 0 && (d.ldgb == 0 || d.ldgb % 4 == 0))) &&
#ifndef G8_PROBE
           (!d.gn_stats || (d.act == CCEDIT_ACT_NONE && d.gn_rows > 0 && d.gn_rows % 128 == 0 && d.N % 32 == 0 && d.N >= 256)) &&
#endif
           d.lda % 8 == 0 && d.ldc % 8 == 0 && (!d.res1 || d.ldr1 % 8 == 0) && (!d.res2 || d.ldr2 % 8 == 0) &&
           (int64_t)512 * d.lda * 2 < (1LL << 31) && d.M * ((d.N + 127) / 128) < (1LL << 37);
}

// Split-K factor for the 256ch x 256pix shape (1 = none): outputs whose tiles fill less than half of the chip and whose K loop is
// long enough to share — as many splits as fit one round of the chip, at least 8 K tiles each, at most 8.  The caller's workspace
// decides: without one (or with one too small) there is no split.  n_cu = 0: the 256 CUs of an MI355X (size queries without a GPU).
int cc_g8_split(const CcGemmDesc& d, int n_cu) {
    const int env = cc_policy().g8_split;       // -1 auto, 0 off, n fixed
    if (env == 0 || d.act == CCEDIT_ACT_GEGLU || !cc_g8_applicable(d, 1)) return 1;
    const int wgs = (n_cu > 0 ? n_cu : 256) / 8 * 8;
    const int64_t tiles = ((d.M + 255) / 256) * ((d.N + 255) / 256);
    const int nk = d.Kpad >> 6;
    if (tiles > 1024 || tiles * 2 > wgs || nk < 48) return 1;
    int s = env > 0 ? env : (int)(wgs / tiles);
    s = s > 8 ? 8 : s;
    while (s > 1 && nk / s < 8) --s;
    return s;
}

int64_t cc_g8_workspace_bytes(const CcGemmDesc& d, int n_cu) {
    const int s = cc_g8_split(d, n_cu);
    if (s <= 1) return 0;
    return 4096 + ((d.M + 255) / 256) * ((d.N + 255) / 256) * s * (int64_t)(256 * 256 * 4);
}

int cc_g8_launch(const CcGemmDesc& d, hipStream_t s, int shape) {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            cc_set_error("g8: cannot query the device");
            return CCEDIT_EINVAL;
        }
        n_cu = prop.multiProcessorCount;
    }
    if (shape == 0) shape = g8_auto_shape(d);
    if (d.ln_stats || d.ln_sums) {           // LayerNorm folded into a plain Linear / GEGLU projection (cc_g8_applicable checked the rest)
        const bool geglu = d.act == CCEDIT_ACT_GEGLU;
        if (shape == 2) return geglu ? g8_launch_shape<1, 2, G8_GEGLU, G8_LINEAR, 0, 1>(d, s, n_cu) : g8_launch_shape<1, 2, G8_PLAIN, G8_LINEAR, 0, 1>(d, s, n_cu);
        return geglu ? g8_launch_shape<2, 1, G8_GEGLU, G8_LINEAR, 0, 1>(d, s, n_cu) : g8_launch_shape<2, 1, G8_PLAIN, G8_LINEAR, 0, 1>(d, s, n_cu);
    }
    if (shape == 1 && d.workspace) {
        const int sk = cc_g8_split(d, n_cu);
        if (sk > 1 && d.workspace_bytes >= cc_g8_workspace_bytes(d, n_cu)) {
            const bool res = d.res1 || d.res2;
            if (d.mode == CCEDIT_GEMM_TEMPORAL)
                return res ? g8_launch_shape<2, 1, G8_RES, G8_TEMPORAL, 1>(d, s, n_cu, sk) : g8_launch_shape<2, 1, G8_PLAIN, G8_TEMPORAL, 1>(d, s, n_cu, sk);
            if (d.mode == CCEDIT_GEMM_CONV2D && d.subpix) return g8_launch_shape<2, 1, G8_PLAIN, G8_SUBPIX, 1>(d, s, n_cu, sk);
            if (d.mode == CCEDIT_GEMM_CONV2D)
                return res ? g8_launch_shape<2, 1, G8_RES, G8_CONV3, 1>(d, s, n_cu, sk) : g8_launch_shape<2, 1, G8_PLAIN, G8_CONV3, 1>(d, s, n_cu, sk);
            return res ? g8_launch_shape<2, 1, G8_RES, G8_LINEAR, 1>(d, s, n_cu, sk) : g8_launch_shape<2, 1, G8_PLAIN, G8_LINEAR, 1>(d, s, n_cu, sk);
        }
    }
    const int epi = d.act == CCEDIT_ACT_GEGLU ? G8_GEGLU : ((d.res1 || d.res2) ? G8_RES : G8_PLAIN);
    const int gm = d.mode == CCEDIT_GEMM_TEMPORAL ? G8_TEMPORAL : (d.mode == CCEDIT_GEMM_CONV2D ? G8_CONV3 : G8_LINEAR);
    if (d.mode == CCEDIT_GEMM_CONV2D && d.subpix)
        return shape == 2 ? g8_launch_shape<1, 2, G8_PLAIN, G8_SUBPIX>(d, s, n_cu) : g8_launch_shape<2, 1, G8_PLAIN, G8_SUBPIX>(d, s, n_cu);
#define G8_GO(TI, TJ, EP)                                                                            \
    (gm == G8_TEMPORAL ? g8_launch_shape<TI, TJ, EP, G8_TEMPORAL>(d, s, n_cu)                        \
                       : (gm == G8_CONV3 ? g8_launch_shape<TI, TJ, EP, G8_CONV3>(d, s, n_cu) : g8_launch_shape<TI, TJ, EP, G8_LINEAR>(d, s, n_cu)))
    if (shape == 2) {
        if (epi == G8_GEGLU) return g8_launch_shape<1, 2, G8_GEGLU, G8_LINEAR>(d, s, n_cu);
        return epi == G8_RES ? G8_GO(1, 2, G8_RES) : G8_GO(1, 2, G8_PLAIN);
    }
    if (epi == G8_GEGLU) return g8_launch_shape<2, 1, G8_GEGLU, G8_LINEAR>(d, s, n_cu);
    return epi == G8_RES ? G8_GO(2, 1, G8_RES) : G8_GO(2, 1, G8_PLAIN);
#undef G8_GO
}
