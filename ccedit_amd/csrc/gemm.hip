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
//   * MFMA A operand (32 rows i) = weights (output channels), B operand (32 cols j) = pixels; every wave
//     owns a 64ch x 64pix sub-tile (2x2 MFMA tiles, 64 fp32 accumulators per lane).
//   * block shapes (template WM x WN waves, STAGES-deep LDS ring):
//       2x2 waves, 128ch x 128pix, 2 stages, 2 workgroups/CU   (small problems)
//       1x4 waves,  64ch x 256pix, 2 stages                    (Cout multiple of 64 but not 128)
//       2x4 waves, 128ch x 256pix, 3 stages, 1 workgroup/CU    (large problems: 85 FLOP per byte staged,
//                  two K tiles in flight behind a counted s_waitcnt vmcnt and a bare s_barrier)
//   * K tile = 64 bf16 = one 128-byte LDS row per tile row; both operands are staged with
//     global_load_lds_dwordx4 (no VGPR round trip).
//   * LDS is written lane-linearly by the DMA, so the bank swizzle is applied on the SOURCE granule
//     index and again on the fragment read: 16-byte granule g of row r lives at slot g ^ ((r>>1)&7).
//     With the 32x32x16 fragment pattern (lane -> row l&31, granule 2*ks + (l>>5)) every ds_read_b128
//     lane group touches 16 distinct 16-byte slots of the 256-byte bank row: SQ_LDS_BANK_CONFLICT = 0.
//   * the implicit-GEMM gather (3x3 halo, stride 2, upsample, temporal neighbours, concat of two
//     sources, K/M tails) only changes the per-lane source ADDRESS of the DMA (branch-free: a select
//     between the computed address and a 64-byte zero page).
//   * XCD-aware 1-D block order and a chunk-major K order keep the activation tile and its 3x3 halo in
//     one XCD's L2 (measured: TCC_MISS 23.6M -> 3.3M per launch on the 320->320 conv at 34x64x96).
//   * epilogue: accumulators -> LDS (fp32) -> 16-byte-per-lane row-contiguous stores with bias,
//     timestep-embedding row bias, SiLU / GEGLU and up to two residual reads fused.
#include "common.h"
#include "gemm_epilogue.h"
#include <math.h>
#include <stdlib.h>

bool cc_conv_halo_applicable(const CcGemmDesc& d);        // convhalo.hip
int cc_conv_halo_launch(const CcGemmDesc& d, hipStream_t s);
bool cc_lin320_applicable(const CcGemmDesc& d);           // lin320.hip
int cc_lin320_launch(const CcGemmDesc& d, hipStream_t s);
bool cc_lin640_applicable(const CcGemmDesc& d);           // lin640.hip
bool cc_temp320_applicable(const CcGemmDesc& d);          // temp320.hip
int cc_temp320_launch(const CcGemmDesc& d, hipStream_t s);
int cc_lin640_launch(const CcGemmDesc& d, hipStream_t s);
bool cc_small_conv_applicable(const CcGemmDesc& d);       // smallconv.hip
int cc_small_conv_launch(const CcGemmDesc& d, hipStream_t s);
bool cc_g8_applicable(const CcGemmDesc& d, int shape);    // gemm8p.hip
int cc_g8_launch(const CcGemmDesc& d, hipStream_t s, int shape);
int cc_g8_split(const CcGemmDesc& d, int n_cu);
int64_t cc_g8_workspace_bytes(const CcGemmDesc& d, int n_cu);

namespace {

__device__ __attribute__((aligned(64))) char g_zero_page[64];


enum { M_LINEAR = 0, M_CONV = 1, M_TEMPORAL = 2, M_CONV_UP = 3 };

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// HALO3 (M_CONV, CcGemmDesc.vpad == 2): the frames' rows are sharded over ranks — taps that fall one row above / below the local rows read
// the neighbour ranks' boundary rows from d.halo_top / d.halo_bot ([frames][Win][lda]; null = the frame ends there: zeros).
template <int WM, int WN, int TI, int TJ, int STAGES, int BKE, int MODE, bool HALO3 = false>
__global__ __launch_bounds__(WM* WN * 64) void tap_gemm_kernel(const CcGemmDesc d) {
    static_assert(BKE == 64 || BKE == 32, "K tile");
    constexpr int kRowBytes = BKE * 2;       // one K tile row of one tile row in LDS (128 or 64 bytes)
    constexpr int GPR = BKE / 8;             // 16-byte granules per tile row (8 or 4)
    constexpr int KSTEPS = BKE / 16;         // MFMA k-steps per K tile
    constexpr int NT = WM * WN * 64;         // threads per block
    constexpr int RPI = NT / GPR;            // tile rows staged per DMA issue (GPR lanes x 16 B per row)
    constexpr int BMC = WM * TI * 32;        // channels per block (each wave: TI x TJ MFMA tiles of 32 x 32)
    constexpr int BNP = WN * TJ * 32;        // pixels per block
    constexpr int A_ISSUES = BMC / RPI;
    constexpr int B_ISSUES = BNP / RPI;
    constexpr int LPS = A_ISSUES + B_ISSUES; // DMA instructions per thread per stage
    constexpr int A_BYTES = BMC * kRowBytes;
    constexpr int B_BYTES = BNP * kRowBytes;
    static_assert(RPI % 32 == 0 && BMC % RPI == 0 && BNP % RPI == 0, "tile / thread-count mismatch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;                         // [STAGES][A_BYTES]
    char* const sB = smem + STAGES * A_BYTES;      // [STAGES][B_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware block order.  Workgroup b runs on XCD b % 8 (observed; used for speed only): give every XCD a
    // contiguous range of pixel tiles and walk the channel tiles of one pixel tile back to back, so the
    // activation tile (and the 3x3 halo rows it shares with its neighbour) is fetched into that XCD's L2 once
    // and re-used by all channel tiles instead of being re-read from HBM/MALL through 8 different L2s.
    const int ct_n = (d.N + BMC - 1) / BMC;
    const int64_t pt_n = (d.M + BNP - 1) / BNP;
    const int64_t pt_per_xcd = (pt_n + 7) / 8;
    const int64_t bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const int64_t local = bid >> 3;
    // Within an XCD the channel tiles are walked in groups of Q = d.cgroup (host-chosen, see launch()): the ~64
    // workgroups resident on the XCD's 32 CUs then cover (64/Q pixel tiles) x (Q channel tiles), so a weight tile
    // that does not fit in the 4 MB L2 is shared by 64/Q concurrently running workgroups instead of being
    // re-fetched by every one of them.  Q = ct_n (weights L2-resident) degenerates to "all channel tiles of one
    // pixel tile back to back".
    const int Q = (d.cgroup & 0xFFFF) > 0 ? (d.cgroup & 0xFFFF) : ct_n;
    const int64_t gsz = pt_per_xcd * Q;
    int cg = (int)(local / gsz);
    int64_t rr = local - cg * gsz;
    int qn = min(Q, ct_n - cg * Q);
    int64_t pl = rr / qn;
    int64_t pt = xcd * pt_per_xcd + pl;
    if ((d.cgroup >> 18) & 1) {
        // No channel grouping: the same walk (an XCD owns a contiguous range of pixel tiles, channel tiles back to back), but
        // the ranges are cut at WORKGROUP granularity, so the XCDs differ by at most one workgroup.  Cutting at pixel-tile
        // granularity leaves e.g. 51 pixel tiles x 5 channel tiles as 35,35,..,10 workgroups on 32-CU XCDs: two rounds where
        // 255 workgroups fit the chip in one.
        const int64_t total = pt_n * ct_n;
        const int64_t per = (total + 7) / 8, w = xcd * per + local;
        if (local >= per || w >= total) return;
        pt = w / ct_n;
        cg = 0;
        qn = ct_n;
        pl = 0;
        rr = w - pt * ct_n;
    } else if (pt >= pt_n) return;
    if constexpr (MODE == M_TEMPORAL) {
        // Conv1d over T: the three taps of pixel tile (frame t, pixel block b) read frames t-1, t, t+1 of the SAME pixel
        // block.  Walk the frames of one pixel block back to back (frame-minor order), so that the tiles sharing input
        // rows are resident on one XCD together and the 2 re-reads hit its L2 instead of going back to the fabric.
        const int tpf = d.HW / BNP;                       // pixel tiles per frame (the launcher checks divisibility)
        if ((d.cgroup >> 17) & 1) {
            const int64_t nfr = pt_n / tpf;
            const int64_t pb = pt / nfr;
            pt = (pt - pb * nfr) * tpf + pb;
        }
    }
    const int64_t pix0 = pt * BNP;
    const int ch0 = (cg * Q + (int)(rr - pl * qn)) * BMC;

    // ---- staging coordinates: thread -> (row rsub + RPI*i, LDS slot p) ----
    // LDS slot p of row r holds source granule p ^ f(r): f(r) = (r>>1)&7 for 128-byte rows (2 rows per 256-byte
    // bank row), (r>>2)&3 for 64-byte rows (4 rows per bank row) — both make the 16-lane ds_read_b128 groups of
    // the 32x32x16 fragment pattern hit 16 distinct 16-byte slots.
    const int p = tid & (GPR - 1);
    const int rsub = tid / GPR;
    const int gcol = p ^ (GPR == 8 ? ((rsub >> 1) & 7) : ((rsub >> 2) & 3));
    const int gpt = d.Cin >> 3;                // 16-byte granules per tap
    const int gtot = d.taps * gpt;
    const int nk = d.Kpad / BKE;

    const bf16* __restrict__ Ap = (const bf16*)d.A;
    const bf16* __restrict__ A2p = (const bf16*)d.A2;
    const bf16* __restrict__ Wp = (const bf16*)d.W;
    const bf16* zp = (const bf16*)g_zero_page;

    // per staged pixel row: source row index of tap (0,0) and the coordinates the bounds checks need
    int rrow[B_ISSUES], ra[B_ISSUES], rb[B_ISSUES];
    int rn[HALO3 ? B_ISSUES : 1];              // HALO3: frame of the staged pixel row x Win (the halo rows' base)
#pragma unroll
    for (int i = 0; i < B_ISSUES; ++i) {
        const int64_t m = pix0 + i * RPI + rsub;
        const bool ok = m < d.M;
        if constexpr (MODE == M_CONV || MODE == M_CONV_UP) {
            const int hwout = d.Hout * d.Wout;
            const int n = (int)(m / hwout);
            const int rem = (int)(m - (int64_t)n * hwout);
            const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
            // subpix: parity (py, px) of an upsample + 3x3 conv on the low-resolution source — a 2 x 2 window starting at (oy - 1 + py, ox - 1 + px)
            // vpad: the source frames carry their own halo rows (RowShard) — one row less of vertical padding
            // (vpad 2: the halo rows are separate tensors — the padding geometry of the whole frame)
            const int pad_y = (d.subpix ? 1 - ((d.subpix - 1) >> 1) : d.pad) - (d.vpad == 1 ? 1 : 0), pad_x = d.subpix ? 1 - ((d.subpix - 1) & 1) : d.pad;
            ra[i] = ok ? oy * d.stride - pad_y : -100000;
            rb[i] = ox * d.stride - pad_x;
            rrow[i] = (MODE == M_CONV) ? n * d.Hin * d.Win + ra[i] * d.Win + rb[i] : n * d.Hin * d.Win;
            if constexpr (HALO3) rn[i] = n * d.Win;
        } else if constexpr (MODE == M_TEMPORAL) {
            const int frame = (int)(m / d.HW);
            const int b = frame / d.T, tl = frame - b * d.T;
            rrow[i] = (b * d.Tsrc + tl + d.tsrc_off) * d.HW + (int)(m - (int64_t)frame * d.HW);   // row in the (halo-extended) source
            ra[i] = ok ? d.t0 + tl : -100000;                                                      // global keyframe index
            rb[i] = 0;
        } else {
            rrow[i] = (int)m;
            ra[i] = ok ? 0 : -100000;
            rb[i] = 0;
        }
    }

    // running (tap, granule-in-tap) of this thread's source granule for the next tile to stage.
    // korder 0: K = [tap][Cin] (any Cin % 8 == 0).  korder 1 (Cin % 64 == 0): K = [Cin/64][tap][64] — the taps
    // of one 64-channel chunk are consecutive K tiles, so the 9 shifted reads of a 3x3 conv hit lines that
    // were fetched one K tile earlier (L2 resident) instead of lines last touched Cin/64 tiles ago.
    int s_tap = gcol / gpt;
    int s_cg = gcol - s_tap * gpt;
    int s_half = 0;                            // BKE == 32: which half of the 64-channel chunk (korder 1)
    if (d.korder) {
        s_tap = 0;
        s_cg = gcol;
    }

    // LINEAR: workgroups that share an operand tile (the channel tiles of one pixel tile share its activation rows, the
    // pixel tiles of one channel tile share its weight rows) start their K loops at different K tiles, so that at any
    // moment they pull different lines through L2 instead of queueing on the same ones (tools/exp/readpat.hip pat3 ->
    // pat9: 5 workgroups re-reading one tile, 53.6 -> 39.6 us).  The sum over K is order-independent up to fp32 rounding.
    const int krot = (MODE == M_LINEAR && (d.cgroup >> 16)) ? (int)((pt + ch0 / BMC) % nk) : 0;
    auto stage = [&](int kt, int buf) {
        if constexpr (MODE == M_LINEAR) {
            kt += krot;
            if (kt >= nk) kt -= nk;
            s_cg = kt * GPR + gcol;                   // taps == 1: the granule index is the K tile position
        }
        // weights: always in range (rows and K are zero-padded by the packer)
#pragma unroll
        for (int i = 0; i < A_ISSUES; ++i) {
            const bf16* src = Wp + (size_t)(ch0 + i * RPI + rsub) * d.Kpad + kt * BKE + gcol * 8;
            glds16(src, sA + buf * A_BYTES + i * (RPI * kRowBytes) + wave * 1024);
        }
        // activations: gather (branch-free address select)
        const bool kvalid = d.korder ? true : (kt * GPR + gcol) < gtot;
        const int c0 = s_cg * 8;
        const bool second = c0 >= d.Cin1;
        const bf16* sp = second ? A2p : Ap;
        const int ld = second ? d.lda2 : d.lda;
        const int cc = second ? c0 - d.Cin1 : c0;
        int dy = 0, dx = 0, delta = 0;
        if constexpr (MODE == M_CONV || MODE == M_CONV_UP) {
            dy = s_tap / d.ksize;
            dx = s_tap - dy * d.ksize;
            delta = dy * d.Win + dx;
        } else if constexpr (MODE == M_TEMPORAL) {
            dy = s_tap - (d.taps >> 1);
            delta = dy * d.HW;
        }
#pragma unroll
        for (int i = 0; i < B_ISSUES; ++i) {
            bool v;
            int row;
            if constexpr (MODE == M_CONV) {
                const int iy = ra[i] + dy, ix = rb[i] + dx;
                v = ((unsigned)iy < (unsigned)d.Hin) & ((unsigned)ix < (unsigned)d.Win);
                row = rrow[i] + delta;
            } else if constexpr (MODE == M_CONV_UP) {
                const int iy = ra[i] + dy, ix = rb[i] + dx;
                v = ((unsigned)iy < (unsigned)(2 * d.Hin)) & ((unsigned)ix < (unsigned)(2 * d.Win));
                row = rrow[i] + (iy >> 1) * d.Win + (ix >> 1);
            } else if constexpr (MODE == M_TEMPORAL) {
                v = (unsigned)(ra[i] + dy) < (unsigned)d.Tglob;
                row = rrow[i] + delta;
            } else {
                v = ra[i] >= 0;
                row = rrow[i];
            }
            const bf16* src = sp + ((int64_t)row * ld + cc);
            src = (v & kvalid) ? src : zp;
            if constexpr (HALO3) {
                const int iy = ra[i] + dy, ix = rb[i] + dx;
                const bool edge = ((iy == -1) | (iy == d.Hin)) & ((unsigned)ix < (unsigned)d.Win) & kvalid;
                const bf16* hb = (const bf16*)(iy < 0 ? d.halo_top : d.halo_bot);
                if (edge && hb) src = hb + ((int64_t)(rn[i] + ix) * ld + cc);
            }
            glds16(src, sB + buf * B_BYTES + i * (RPI * kRowBytes) + wave * 1024);
        }
        // advance to the next K tile (+GPR granules)
        if (d.korder) {
            if (GPR == 4 && s_half == 0) {         // second 32-channel half of the same (chunk, tap)
                s_half = 1;
                s_cg += 4;
            } else {
                if (GPR == 4) {
                    s_half = 0;
                    s_cg -= 4;
                }
                if (++s_tap == d.taps) {
                    s_tap = 0;
                    s_cg += 8;
                }
            }
        } else {
            s_cg += GPR;
            while (s_cg >= gpt) {
                s_cg -= gpt;
                ++s_tap;
            }
        }
    };

    // ---- fragment read coordinates ----
    const int l31 = lane & 31;
    const int hi = lane >> 5;
    const int sw = GPR == 8 ? ((l31 >> 1) & 7) : ((l31 >> 2) & 3);
    const char* fa = sA + (wm * TI * 32 + l31) * kRowBytes;
    const char* fb = sB + (wn * TJ * 32 + l31) * kRowBytes;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const char* pa = fa + buf * A_BYTES;
        const char* pb = fb + buf * B_BYTES;
        // Fragments are read per k-step and the compiler interleaves the next step's ds_read_b128 with the MFMAs.
        // (Measured alternative: issuing all 16 reads of the K tile up front behind a sched_barrier lifts the
        // DMA-free ceiling 877 -> 922 TF/s but costs 46 VGPRs and is 5 % SLOWER in the network: 160 vs 152 ms/step.)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int off = ((ks * 2 + hi) ^ sw) << 4;
            bf16x8 af[TI], bfr[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8*)(pa + i * 32 * kRowBytes + off);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[j] = *(const bf16x8*)(pb + j * 32 * kRowBytes + off);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop ----
    if constexpr (STAGES == 2) {
        // stage(t+1) || compute(t), one barrier per tile; a second resident workgroup hides the wait
        stage(0, 0);
        wait_vmcnt<0>();
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk - 1; ++kt) {
            stage(kt + 1, cur ^ 1);
            compute(cur);
            wait_vmcnt<0>();
            __syncthreads();
            cur ^= 1;
        }
        compute(cur);
    } else {
        // STAGES-deep ring, one workgroup per CU: tiles t+1 .. t+STAGES-1 are in flight while tile t is consumed.
        // The DMA queue is drained only down to the newer stages (counted vmcnt) and the barrier is a bare
        // s_barrier, so the loads stay in flight across it.
        static_assert(STAGES == 3 || STAGES == 4, "ring depth");
#pragma unroll
        for (int st = 0; st < STAGES - 1; ++st)
            if (st < nk) stage(st, st);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int newer = nk - 1 - kt;         // stages issued after tile kt that may stay in flight
            if (newer >= STAGES - 2) wait_vmcnt<LPS*(STAGES - 2)>();
            else if (newer == 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // tile kt landed for every wave; the oldest buffer is free again
            if (kt + STAGES - 1 < nk) {
                int nb = cur + STAGES - 1;
                if (nb >= STAGES) nb -= STAGES;
                stage(kt + STAGES - 1, nb);
            }
            compute(cur);
            if (++cur == STAGES) cur = 0;
        }
    }

    // ---- epilogue (gemm_epilogue.h) ----
    gemm_epilogue<WM, WN, TI, TJ, STAGES * (A_BYTES + B_BYTES)>(
        d, acc, smem, ch0, [&](int p) -> int64_t { const int64_t m = pix0 + p; return m < d.M ? m : -1; },
        d.gn_stats ? pix0 / d.gn_rows : 0);
}

template <int WM, int WN, int TI, int TJ, int STAGES, int BKE, int MODE, bool HALO3 = false>
int launch(const CcGemmDesc& d, hipStream_t s) {
    constexpr int BMC = WM * TI * 32, BNP = WN * TJ * 32;
    constexpr int lds = epi_lds_total(BMC, BNP, TJ, STAGES * (BMC + BNP) * BKE * 2);
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)tap_gemm_kernel<WM, WN, TI, TJ, STAGES, BKE, MODE, HALO3>, lds, &attr_done, "tap_gemm"))
        return rc;
    const int64_t pt_n = (d.M + BNP - 1) / BNP, ct_n = (d.N + BMC - 1) / BMC;
    int64_t nblk = 8 * ((pt_n + 7) / 8) * ct_n;
    if (nblk > 2147483647LL) {
        cc_set_error("ccedit_gemm: grid too large");
        return CCEDIT_EUNSUPPORTED;
    }
    // Channel-tile group width (see the block-order comment in the kernel).  Weights that fit in an XCD's L2 are
    // fetched once whatever the order; otherwise, with C workgroups resident per XCD as (C/Q pixel tiles) x (Q
    // channel tiles), the fabric traffic per workgroup is  wtile * Q/C + atile / Q, minimal at Q = sqrt(C*atile/wtile)
    // (atile shrinks by the tap re-use of a 3x3 / temporal gather).
    CcGemmDesc dd = d;
    dd.cgroup = 0;
    const double wbytes = (double)ct_n * BMC * d.Kpad * 2.0;
    if (ct_n > 4 && wbytes > 3.0 * 1024 * 1024) {
        const double resident = (lds <= 80 * 1024 ? 2.0 : 1.0) * 32.0;
        const double a_over_w = (double)BNP / ((double)BMC * d.taps);
        int q = (int)(sqrt(resident * a_over_w) + 0.5);
        q = q < 2 ? 2 : q;
        if (q < ct_n) {
            const int ng = (int)((ct_n + q - 1) / q);
            dd.cgroup = (int)((ct_n + ng - 1) / ng);
        }
    }
    if (d.Kpad <= 640) dd.cgroup |= 1 << 16;            // K rotation, short K only: measured neutral or -3 % at K = 1280
    if (MODE == M_TEMPORAL && d.HW % BNP == 0 && d.M % d.HW == 0) dd.cgroup |= 1 << 17;      // frame-minor tile order
    if ((dd.cgroup & 0xFFFF) == 0 && !((dd.cgroup >> 17) & 1)) {       // XCD ranges cut at workgroup granularity
        dd.cgroup |= 1 << 18;
        nblk = 8 * ((pt_n * ct_n + 7) / 8);
    }
    dim3 grid((unsigned)nblk);
    cc_note_kernel("tap_gemm_kernel %dch x %dpix, %d stages of K=%d%s", BMC, BNP, STAGES, BKE, HALO3 ? ", neighbour halo rows" : "");
    hipLaunchKernelGGL((tap_gemm_kernel<WM, WN, TI, TJ, STAGES, BKE, MODE, HALO3>), grid, dim3(WM * WN * 64), lds, s, dd);
    return cc_launch_status("tap_gemm_kernel");
}

template <int MODE>
int launch_tile(const CcGemmDesc& d, int tile, hipStream_t s) {
    if constexpr (MODE == M_CONV) {
        if (d.vpad == 2) {          // rows sharded over ranks, halo rows as separate tensors: the generic block shapes
            if (tile == 3) return launch<2, 4, 2, 2, 3, 64, MODE, true>(d, s);
            if (tile == 2) return launch<1, 4, 2, 2, 2, 64, MODE, true>(d, s);
            return launch<2, 2, 2, 2, 2, 64, MODE, true>(d, s);
        }
    }
    if constexpr (MODE == M_LINEAR) {
        if (tile == 7) return launch<2, 2, 4, 4, 4, 32, MODE>(d, s);   // 256ch x 256pix, 4 waves of 128ch x 128pix, 4 stages of K=32
        if (tile == 10) return launch<2, 2, 5, 4, 4, 32, MODE>(d, s);  // 320ch x 256pix, 4 waves of 160ch x 128pix, 4 stages of K=32
    }
    if (tile == 6) return launch<2, 2, 5, 2, 2, 32, MODE>(d, s);  // 320ch x 128pix, 4 waves of 160ch x 64pix, 2 stages of K=32, 2 WG/CU
    if (tile == 5) return launch<2, 4, 2, 4, 4, 32, MODE>(d, s);  // 128ch x 512pix, 8 waves of 64ch x 128pix, 4 stages of K=32
    if (tile == 4) return launch<2, 4, 4, 2, 4, 32, MODE>(d, s);  // 256ch x 256pix, 8 waves of 128ch x 64pix, 4 stages of K=32
    if (tile == 3) return launch<2, 4, 2, 2, 3, 64, MODE>(d, s);
    if (tile == 2) return launch<1, 4, 2, 2, 2, 64, MODE>(d, s);
    return launch<2, 2, 2, 2, 2, 64, MODE>(d, s);
}

}  // namespace

extern "C" int64_t ccedit_gemm_workspace_bytes(const CcGemmDesc* desc) {
    if (!desc || desc->M <= 0 || desc->N <= 0 || desc->Kpad <= 0) return 0;
    if (!(desc->tile == 0 || desc->tile == 11 || desc->tile == 12)) return 0;
    return cc_policy().g8 ? cc_g8_workspace_bytes(*desc, 0) : 0;
}

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
    CC_UNSUPPORTED(d.M >= (1LL << 31), "ccedit_gemm: M too large");
    if (d.A2 == nullptr) d.Cin1 = d.Cin;
    CC_UNSUPPORTED(d.korder && (d.Cin % 64 != 0 || d.Cin1 % 64 != 0 || d.Kpad != d.taps * d.Cin),
                   "ccedit_gemm: korder=1 needs Cin (and the concat split) to be multiples of 64");
    CC_UNSUPPORTED(d.N > 4 && ((!d.out_f32 && d.ldc % 8 != 0) || (d.res1 && d.ldr1 % 8 != 0) || (d.res2 && d.ldr2 % 8 != 0)),
                   "ccedit_gemm: N > 4 needs ldc / residual strides that are multiples of 8");
    CC_UNSUPPORTED(d.A2 && (d.Cin1 % 8 != 0 || d.lda2 % 8 != 0 || d.Cin1 <= 0 || d.Cin1 >= d.Cin),
                   "ccedit_gemm: bad concat split Cin1=%d lda2=%d", d.Cin1, d.lda2);
    if (d.mode == CCEDIT_GEMM_CONV2D) {
        CC_CHECK_ARG(d.ksize * d.ksize == d.taps && d.Hin > 0 && d.Win > 0 && d.Hout > 0 && d.Wout > 0 && d.stride > 0,
                     "ccedit_gemm: bad conv2d geometry");
        CC_CHECK_ARG(d.M % ((int64_t)d.Hout * d.Wout) == 0, "ccedit_gemm: M not a whole number of frames");
        CC_CHECK_ARG(d.subpix >= 0 && d.subpix <= 4, "ccedit_gemm: subpix must be 0..4");
        CC_CHECK_ARG(d.vpad >= 0 && d.vpad <= 2, "ccedit_gemm: vpad must be 0, 1 or 2");
        CC_CHECK_ARG(d.vpad == 2 || (!d.halo_top && !d.halo_bot), "ccedit_gemm: halo_top / halo_bot go with vpad 2");
        CC_UNSUPPORTED(d.vpad && (d.upsample || d.tile > 3 || d.A2), "ccedit_gemm: vpad runs on the generic tap-gather block shapes only, without the fused upsample / a second source");
        CC_UNSUPPORTED(d.subpix && (d.ksize != 2 || d.stride != 1 || d.upsample || d.Hin != d.Hout + (d.vpad == 1 ? 2 : 0) || d.Win != d.Wout || d.gn_stats || d.A2 ||
                                    4 * d.M >= (1LL << 31)),
                       "ccedit_gemm: subpix needs ksize 2, stride 1, a same-size low-resolution frame, no gn_stats / second source");
    } else if (d.subpix || d.vpad || d.halo_top || d.halo_bot) {
        CC_CHECK_ARG(false, "ccedit_gemm: subpix / vpad / halo rows are CONV2D modes");
    } else if (d.mode == CCEDIT_GEMM_TEMPORAL) {
        CC_CHECK_ARG(d.T > 0 && d.HW > 0 && d.M % ((int64_t)d.T * d.HW) == 0, "ccedit_gemm: bad temporal geometry");
        CC_CHECK_ARG(d.taps % 2 == 1, "ccedit_gemm: temporal taps must be odd");
        if (d.Tsrc == 0) {          // unsharded clip
            d.Tsrc = d.T;
            d.tsrc_off = 0;
            d.t0 = 0;
            d.Tglob = d.T;
        }
        CC_CHECK_ARG(d.Tsrc >= d.T + d.tsrc_off && d.tsrc_off >= 0 && d.t0 >= 0 && d.t0 + d.T <= d.Tglob,
                     "ccedit_gemm: bad frame-shard geometry");
        CC_UNSUPPORTED((int64_t)(d.M / ((int64_t)d.T * d.HW)) * d.Tsrc * d.HW >= (1LL << 31), "ccedit_gemm: source too large");
    } else {
        CC_CHECK_ARG(d.mode == CCEDIT_GEMM_LINEAR && d.taps == 1, "ccedit_gemm: bad mode/taps");
    }
    if (d.act == CCEDIT_ACT_GEGLU) {
        CC_UNSUPPORTED(d.N % 16 != 0 || d.res1 || d.res2 || d.group_bias || d.out_f32,
                       "ccedit_gemm: GEGLU epilogue needs N%%16==0 and no residual/group bias/f32 out");
    }
    if (d.group_bias) CC_CHECK_ARG(d.group_rows > 0 && (d.ldgb == 0 || d.ldgb >= d.N), "ccedit_gemm: group_bias without group_rows / ldgb < N");
    if (d.gn_stats && !(d.tile >= 11 && d.tile <= 13)) {      // (the persistent Linear shapes refuse gn_stats themselves)
        CC_CHECK_ARG(d.gn_rows > 0 && d.gn_rows % 128 == 0 && d.M % d.gn_rows == 0,
                     "ccedit_gemm: gn_stats needs gn_rows %% 128 == 0 dividing M (gn_rows=%d)", d.gn_rows);
        CC_CHECK_ARG(d.gn_rows % 256 == 0 || d.tile <= 1 || d.tile == 8 || d.tile == 14,
                     "ccedit_gemm: gn_rows=%d is not a multiple of 256: only the 128-pixel block shape (tile 1) applies", d.gn_rows);
        CC_UNSUPPORTED(d.tile == 6, "ccedit_gemm: gn_stats is not available with the 320-channel block shape (tile 6)");
        CC_UNSUPPORTED(d.N % 32 != 0 || d.N < 256 || d.out_f32 || d.act == CCEDIT_ACT_GEGLU,
                       "ccedit_gemm: gn_stats needs N%%32==0, N>=256, bf16 output, no GEGLU (N=%d)", d.N);
    }
    hipStream_t s = (hipStream_t)stream;
    // ln_eps: only lin320_kernel normalises its rows; every other kernel would multiply the gamma / beta-folded weights with
    // un-normalised rows.  Whatever block shape the caller asked for: that kernel, or a refusal — never a silently wrong product.
    if (d.ln_eps != 0.f) {
        CC_UNSUPPORTED(!(d.tile == 0 || d.tile == 9) || !cc_lin320_applicable(d),
                       "ccedit_gemm: ln_eps needs the register-resident K = 320 kernel (block shape 0 / 9, plain Linear with K = 320 "
                       "and N %% 320 == 0, no activation / residual): not applicable to this descriptor");
        return cc_lin320_launch(d, s);
    }
    // Conv1d k3 over T at 320 input channels over many pixels (the 64x96 level): all three taps' weights in registers, pixel columns
    // streamed frame by frame (temp320.hip).  Policy temp320 = 0: A/B against the tiled kernels.
    const CcPolicy& pol = cc_policy();
    if ((d.tile == 0 && pol.temp320 && d.M >= 100000) || d.tile == 14) {
        if (cc_temp320_applicable(d)) return cc_temp320_launch(d, s);
        CC_UNSUPPORTED(d.tile == 14, "ccedit_gemm: tile 14 (streaming Conv1d k3, 320 input channels) does not apply to this descriptor");
    }
    // K = 640 Linear over many whole 16-pixel tiles (the 32x48 level): weights resident in registers, activations streamed once
    // (lin640.hip) — bias, one residual, row_sums, or the folded LayerNorm (ln_stats / ln_sums).  Policy lin640 = 0: A/B against gemm8p.
    // (at 52224 rows, lin640s / gemm8p: 1920 channels 147 / 220 us, 1280: 105 / 140; at 640 the three slices — the third half empty —
    //  fetch the activations three times: plain 63 / 68, residual 66 / 75, with row_sums 74 / 82, but with the folded LayerNorm
    //  67 / 65 — that one stays on gemm8p below 1024 channels)
    if ((d.tile == 0 && pol.lin640 && d.M >= 16384 && (d.N >= 1024 || (!d.ln_stats && !d.ln_sums))) || d.tile == 10) {
        if (cc_lin640_applicable(d)) return cc_lin640_launch(d, s);
        CC_UNSUPPORTED(d.tile == 10, "ccedit_gemm: tile 10 (register-resident weights, K = 640) does not apply to this descriptor");
    }
    // ln_stats: the epilogue applies the rows' LayerNorm statistics — the persistent eight-phase kernel and lin640s implement it, nothing else
    if (d.ln_stats || d.ln_sums || d.ln_colsum) {
        const int shape = (d.tile >= 11 && d.tile <= 13) ? d.tile - 11 : 0;
        CC_UNSUPPORTED((!d.ln_stats && !d.ln_sums) || !d.ln_colsum || !(d.tile == 0 || (d.tile >= 11 && d.tile <= 13)) || !cc_g8_applicable(d, shape),
                       "ccedit_gemm: ln_stats needs ln_colsum and the persistent eight-phase kernel (block shape 0 / 11-13: plain Linear with "
                       "Cin %% 64 == 0, N %% 16 == 0, no residual / row bias / gn_stats): not applicable to this descriptor");
        return cc_g8_launch(d, s, shape);
    }
    if (d.row_sums) {           // the epilogue accumulates the LayerNorm statistics of what it writes: same kernel only
        const int shape = (d.tile >= 11 && d.tile <= 13) ? d.tile - 11 : 0;
        CC_UNSUPPORTED(!(d.tile == 0 || (d.tile >= 11 && d.tile <= 13)) || !cc_g8_applicable(d, shape),
                       "ccedit_gemm: row_sums needs the persistent eight-phase kernel (block shape 0 / 11-13: Linear with Cin %% 64 == 0, "
                       "N %% 16 == 0, no activation): not applicable to this descriptor");
        return cc_g8_launch(d, s, shape);
    }
    if (d.tile == 0 && !d.vpad && cc_small_conv_applicable(d)) return cc_small_conv_launch(d, s);     // few-channel 3x3 (hint stem top)
    // 3x3 convs onto >= 1024 channels (the 16x24 level): 29-59 MB of weights do not fit an XCD's L2 and the 128 x 128 tiles of the LDS-halo
    // kernel re-stream them per pixel tile (450 MB fetched for 63 MB of operands, round 2).  The persistent 256 x 256 tap-gather loop
    // halves the weight bytes per FLOP: cold sweep 1280->1280 1162 / 843, 2560->1280 1220 / 895, 640->1280 1067 / 729 TF/s.  At 640
    // channels and below the halo re-use wins (32x48 640->640: 890 / 925) and those stay.  Policy g8_conv = 0 for the A/B.
    if (d.tile == 0 && pol.g8_conv && pol.g8 && !d.vpad && d.mode == CCEDIT_GEMM_CONV2D && d.N >= 1024 && d.M >= 12000 && cc_g8_applicable(d, 1))
        return cc_g8_launch(d, s, 1);
    // Few tiles, long K (the 8x12 level: 3264 pixels x 1280 channels = 65 tiles of 256 x 256, K loops of 60-360 K tiles — a quarter of the
    // chip busy for the whole loop on any block shape): split-K in the persistent kernel when the caller lent a workspace.
    if (d.tile == 0 && pol.g8 && (!d.subpix || pol.g8_conv) && !d.vpad && d.workspace && cc_g8_split(d, 0) > 1 && d.workspace_bytes >= cc_g8_workspace_bytes(d, 0))
        return cc_g8_launch(d, s, 1);
    // The four parity convs of an upsample + conv 3x3 (K = 4 Cin on the low-resolution source): the same persistent gather loop with a
    // 2 x 2 window (round 6).  16x24 -> 32x48 (13056 source pixels x 1280 channels, K = 5120: 255 tiles, one round of the chip) is taken
    // by the clause above, the 8x12 source by split-K; this one takes the 32x48 source (640 channels: 128ch x 512pix).
    if (d.tile == 0 && pol.g8_conv && pol.g8 && d.subpix && d.M >= 12000 && d.N >= 640 && cc_g8_applicable(d, 0)) return cc_g8_launch(d, s, 0);
    if ((d.tile == 0 && pol.conv_halo && !d.vpad) || d.tile == 8) {
        if (!d.vpad && cc_conv_halo_applicable(d)) return cc_conv_halo_launch(d, s);
        CC_UNSUPPORTED(d.tile == 8, "ccedit_gemm: tile 8 (LDS-halo 3x3 conv) does not apply to this descriptor");
    }
    // K = 320 Linear over many pixels: weights resident in registers, activations streamed once (lin320.hip)
    if ((d.tile == 0 && pol.lin320 && d.M >= 32768) || d.tile == 9) {
        if (cc_lin320_applicable(d)) return cc_lin320_launch(d, s);
        CC_UNSUPPORTED(d.tile == 9, "ccedit_gemm: tile 9 (register-resident weights, K = 320) does not apply to this descriptor");
    }
    // plain long Linear: persistent 256ch x 256pix eight-phase kernel (gemm8p.hip)
    if (d.tile >= 11 && d.tile <= 13) {     // 11: block shape by Cout, 12: 256ch x 256pix, 13: 128ch x 512pix
        CC_UNSUPPORTED(!cc_g8_applicable(d, d.tile - 11), "ccedit_gemm: tile 11-13 (persistent eight-phase GEMM) does not apply to this descriptor");
        return cc_g8_launch(d, s, d.tile - 11);
    }
    // Long plain Linears (FF / GEGLU projections and the C -> C projections of the 32x48 and 16x24 levels): the persistent
    // eight-phase kernel (gemm8p.hip).  Cold-operand sweep (tools/exp/g8_check.py), TF/s against the best older block shape:
    //   52224 x 5120 <- 640 GEGLU 1030 / 713    13056 x 10240 <- 1280 GEGLU 1176 / 818    52224 x 640 <- 2560 + res 858 / 752
    //   13056 x 1280 <- 5120 + res 1171 / 904   13056 x 1280 <- 1280 + res 814 / 553      52224 x 640 <- 640 + res 582 / 525
    // and the single CFG halves (two-stream execution): 26112 x 5120 <- 640 GEGLU 945 / 588, 6528 x 10240 <- 1280 GEGLU 964 / 677.
    // Not below 4096 rows (fewer than ~100 tiles: most CUs idle) and not for K = 320 (lin320 / ff320 own those).
    if (d.tile == 0 && pol.g8 && !d.subpix && d.M >= 4096 && d.Kpad >= 640 && d.N >= 640 && cc_g8_applicable(d, 0)) {
        // Conv1d k3 over T: the same K loop with the activation rows gathered HW rows away (cold sweep, + residual, against the
        // best older shape: 32x48 640->640 746 / 703, 1280->1280 1034 / 883; 16x24 1280->1280 981 / 917; 64x96 640->640 801 / 692 —
        // and 507 / 519 at 320->320, where three 128-channel tiles re-read every activation row: those stay on tap_gemm, N >= 640 above).
        // Below ~200 tiles the persistent grid is mostly idle: the 8x12 level (3264 rows) and single CFG halves of 16x24 stay too.
        if (d.mode == CCEDIT_GEMM_LINEAR || (d.mode == CCEDIT_GEMM_TEMPORAL && pol.g8_temporal && d.M >= 12000)) return cc_g8_launch(d, s, 0);
    }
    CC_UNSUPPORTED(d.tile == 6 && d.N % 320 != 0, "ccedit_gemm: tile 6 (320-channel block shape) needs N %% 320 == 0 (N=%d)", d.N);
    int tile = d.tile;
    if (tile == 0) {
        // Chosen from IN-NETWORK timings (bench.py --breakdown), where operands arrive cold from HBM; the isolated
        // sweep (tools/tile_sweep.py, operands hot in the 256 MB Infinity Cache) over-rates the wide 8-wave shapes
        // (t4 = 256ch x 256pix, t5 = 128ch x 512pix, 8 waves with 8 MFMA tiles each, 4-stage K=32 rings — also tried
        // as 2-stage K=64 rings and as 4-wave / 2-workgroup-per-CU 128ch x 256pix shapes: up to 1064 TF/s isolated,
        // but every policy that routes network layers to them measured 1-8 % slower per step than this one):
        //   Cout multiple of 64 but not 128 (320, 960): 64ch x 256pix (t2) for gathers and short K, else 128x128 (t1)
        //   very large M with Cout % 128 == 0 and long K: 128ch x 256pix 3-stage (t3)
        //   everything else: 128ch x 128pix, two workgroups per CU (t1)
        const int w128 = (d.N + 127) / 128 * 128, w64 = (d.N + 63) / 64 * 64;
        if (w64 < w128) tile = (d.taps > 1 || d.Kpad <= 512) ? 2 : 1;
        else if (d.M >= 150000 && d.Kpad >= 2048) tile = 3;
        else if (d.act == CCEDIT_ACT_GEGLU && d.Kpad <= 640) tile = 2;    // with the K rotation: +9 / +5 % over t1 at K = 320 / 640
        else tile = 1;
        //   long K with Cout % 320 == 0 (every UNet width): 320ch x 128pix, K tiles of 32, still two workgroups per CU
        //   (t6: 91 FLOP per staged byte against 64 for t1; cold sweep +4 % at 1280 -> 1280 / 3840 / 10240 and +8 % at
        //   1280 -> 320 on 209k pixels, +22 % on the 1280-channel temporal conv; slower for K <= 640 and below 8k pixels;
        //   in the network +0.6 % per step)
        if (d.N % 320 == 0 && !d.gn_stats && d.M >= 8192 &&
            ((d.mode == CCEDIT_GEMM_LINEAR && d.Kpad >= 1280) || (d.mode == CCEDIT_GEMM_TEMPORAL && d.Kpad >= 3840)))
            tile = 6;
        //   16x24-level Linears onto 1280 channels: 256ch x 256pix (t4).  With the workgroup-granular XCD cut (cgroup bit 18)
        //   13056 pixels x 1280 channels is 255 workgroups = one round of the chip: 706 / 906 / 954 TF/s at K = 1280 / 5120 /
        //   11520 against 644 / 817 / 846 for t6 (tools/exp/gemm_vs_vendor.py; the same shapes cut at pixel-tile granularity
        //   ran 374 / 540 / 591).
        //   Only with BOTH CFG halves in the launch: at 6528 pixels (one half, the default two-stream execution) it is 130
        //   workgroups and loses to t1 (383 / 544 against 454 / 577 TF/s at K = 1280 / 5120; -0.6 ms per step in the A/B).
        if (d.mode == CCEDIT_GEMM_LINEAR && d.N % 256 == 0 && d.N <= 2560 && d.act != CCEDIT_ACT_GEGLU && !d.gn_stats &&
            d.Kpad >= 1280 && d.M >= 12000 && d.M <= 16384)
            tile = 4;
        if (d.gn_stats && d.gn_rows % 256 != 0) tile = 1;     // a block must not straddle two frames
    }
    const int mode = d.mode == CCEDIT_GEMM_CONV2D ? (d.upsample ? M_CONV_UP : M_CONV)
                                                  : (d.mode == CCEDIT_GEMM_TEMPORAL ? M_TEMPORAL : M_LINEAR);
    switch (mode) {
        case M_CONV: return launch_tile<M_CONV>(d, tile, s);
        case M_CONV_UP: return launch_tile<M_CONV_UP>(d, tile, s);
        case M_TEMPORAL: return launch_tile<M_TEMPORAL>(d, tile, s);
        default: return launch_tile<M_LINEAR>(d, tile, s);
    }
}
