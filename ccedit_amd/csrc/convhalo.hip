// 3x3 stride-1 convolution with the input tile staged ONCE per 64-channel chunk (halo included) and the nine taps read
// from LDS at shifted rows.
//
// Why a second conv kernel: tap_gemm_kernel gathers every tap's activation tile from global memory again (9 x per
// chunk).  PMC on the 64x96-level 320->320 conv: 124 M L1 accesses, 46 % of them missing to L2 (3.6 GB through the
// TCP->TCC path per launch).  A CU sustains only ~20 B/clk of L1-miss traffic (outstanding-miss queue x L2 latency,
// measured with tools/exp/readpat.hip), so that traffic — not MFMA, not HBM — bounded the convs at 30-38 % of peak.
// Here a workgroup owns a TH x TW pixel rectangle (128 pixels) of one frame and 128 output channels:
//   per chunk  : (TH+2) x (TW+2) halo rows x 128 B  -> LDS once           (23 KB instead of 9 x 16 KB)
//   per tap    : only the 128 x 64 weight tile streams (2-deep ring), the B fragments are ds_read_b128 at
//                halo row (ty+dy)*(TW+2) + tx+dx with the same XOR swizzle as everywhere else
// K order of the packed weights is [Cin/64][tap][64] (korder 1), i.e. k-tile c*9 + t.  Epilogue: gemm_epilogue.h with
// a 2-D row map.  Shapes that do not qualify (stride 2, fused upsample, Cin % 64 != 0, frames not divisible into
// 8x16 / 16x8 rectangles) stay on tap_gemm_kernel.
#include "common.h"
#include "gemm_epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace {

__device__ __attribute__((aligned(64))) char g_zero_page_h[64];     // source of out-of-image halo rows

#ifdef CONV_PROBE          // probe builds only (tools/exp/conv_probe.py): per-workgroup time stamps, never in the product library
constexpr int kProbeWgs = 8192, kProbeStamps = 8;
__device__ unsigned long long g_conv_probe[kProbeWgs * kProbeStamps];
#define CONV_STAMP(i)                                                                                                     \
    do {                                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < kProbeWgs) g_conv_probe[blockIdx.x * kProbeStamps + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CONV_STAMP(i) \
    do {              \
    } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_h() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int kHaloRows = 180;                 // (8+2) x (16+2) = (16+2) x (8+2)
constexpr int kHaloBytes = 184 * 128;          // rounded to whole 8-row DMA groups
constexpr int kHaloIssues = 6;                 // ceil(184 rows / 32 rows per 256-thread issue)

template <int WM, int WN, int TI, int TJ>
__global__ __launch_bounds__(WM* WN * 64) void conv_halo_kernel(const CcGemmDesc d, int tw_log2_flags) {
    const int tw_log2 = tw_log2_flags & 0xFF;      // bit 8: narrow last channel tile allowed
    constexpr int NT = WM * WN * 64;
    constexpr int BMC = WM * TI * 32, BNP = WN * TJ * 32;
    static_assert(NT == 256 && BNP == 128, "tile geometry");
    constexpr int RPI = NT / 8;                    // rows per DMA issue (8 granules of 16 B per 128-byte row)
    constexpr int W_ISSUES = BMC / RPI;
    constexpr int W_BYTES = BMC * 128;
    constexpr int LDS_MAIN = 2 * W_BYTES + 2 * kHaloBytes;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sW = smem;                         // [2][W_BYTES]
    char* const sH = smem + 2 * W_BYTES;           // [2][kHaloBytes]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int TW = 1 << tw_log2, TH = BNP >> tw_log2, HW_ = TW + 2;
    const int tiles_x = (d.Wout + TW - 1) >> tw_log2, tiles_y = d.Hout / TH;      // the last column of rectangles may be ragged
    const int tpf = tiles_x * tiles_y;

    // XCD-aware block order, same scheme as tap_gemm_kernel (pixel tiles contiguous per XCD, channel tiles in groups)
    const int ct_n = (d.N + BMC - 1) / BMC;
    const int64_t pt_n = (int64_t)(d.M / (d.Hout * d.Wout)) * tpf;
    const int64_t pt_per_xcd = (pt_n + 7) / 8;
    const int64_t bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const int64_t local = bid >> 3;
    const int Q = d.cgroup > 0 ? d.cgroup : ct_n;
    const int64_t gsz = pt_per_xcd * Q;
    const int cg = (int)(local / gsz);
    const int64_t rr = local - cg * gsz;
    const int qn = min(Q, ct_n - cg * Q);
    const int64_t pl = rr / qn;
    const int64_t pt = xcd * pt_per_xcd + pl;
    if (pt >= pt_n) return;
    const int ch0 = (cg * Q + (int)(rr - pl * qn)) * BMC;
    const int frame = (int)(pt / tpf);
    const int tr = (int)(pt - (int64_t)frame * tpf);
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) << tw_log2;

    const bf16* __restrict__ Ap = (const bf16*)d.A;
    const bf16* __restrict__ Wp = (const bf16*)d.W;
    const bf16* zp = (const bf16*)g_zero_page_h;
    const int nc = d.Cin >> 6, nk = nc * 9;
    // Last channel tile of a Cout that is not a multiple of 128 (320 = 128 + 128 + 64): only WM x 32 channels are real.
    // The block then stages half the weight tile and every wave keeps ONE of its two MFMA row tiles (wave row wm takes
    // channels 32 wm .. 32 wm + 31) — half the matrix work instead of multiplying 64 rows of padding (17 % of the MFMA
    // energy of a 320-channel conv on a part that runs these kernels at its power limit).
    const bool narrow = (TI == 2) && ((tw_log2_flags >> 8) & 1) && (d.N - ch0 <= WM * 32);

    // ---- staging coordinates ----
    // Halo swizzle: LDS slot s of halo entry (hy, hx) holds source granule s ^ f(hy, hx),
    //   f = (hx >> 1) & 7                     for 8 x 16 rectangles,
    //   f = ((hx >> 1) + 4 * (hy & 1)) & 7    for 16 x 8 rectangles.
    // ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (MI355X_MICROARCH.md); with
    // the halo pitch TW+2 a swizzle keyed on the linear row index (as in tap_gemm_kernel) is 2-way conflicted for every
    // tap, this one puts the 16 lanes of a group on 16 distinct 16-byte slots of the 256-byte bank row for all nine
    // shifts (checked exhaustively; PMC: SQ_LDS_BANK_CONFLICT 28.2 M -> ~0 per launch).
    const int f_hy = (tw_log2 == 3) ? 1 : 0;
    const int p = tid & 7, rsub = tid >> 3;
    const int gcol_w = p ^ ((rsub >> 1) & 7);              // weight rows rsub + 32 i: (row >> 1) & 7 is i-independent
    // halo: issue i stages halo rows i*32 + rsub; source element offset (without the chunk) or -1
    int64_t hoff[kHaloIssues];
#pragma unroll
    for (int i = 0; i < kHaloIssues; ++i) {
        const int hrow = i * 32 + rsub;
        const int hy = hrow / HW_, hx = hrow - hy * HW_;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool v = hrow < kHaloRows && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        const int gsrc = p ^ (((hx >> 1) + ((hy & 1) << 2) * f_hy) & 7);     // halo swizzle, see compute()
        hoff[i] = v ? (((int64_t)frame * d.Hin + iy) * d.Win + ix) * d.lda + gsrc * 8 : (hrow < 184 ? -1 : -2);
    }

    auto stageW = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < W_ISSUES; ++i)
            if (!narrow || i * RPI < WM * 32)
                glds16(Wp + (size_t)(ch0 + i * RPI + rsub) * d.Kpad + kt * 64 + gcol_w * 8, sW + buf * W_BYTES + i * (RPI * 128) + wave * 1024);
    };
    auto stageH = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < kHaloIssues; ++i) {
            if (hoff[i] != -2) {                            // rows 184..191 of the last issue do not exist
                const bf16* src = hoff[i] >= 0 ? Ap + hoff[i] + c * 64 : zp;
                glds16(src, sH + buf * kHaloBytes + i * (32 * 128) + wave * 1024);
            }
        }
    };

    // ---- fragment coordinates ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw_w = (l31 >> 1) & 7;
    const char* fa = sW + ((narrow ? wm * 32 : wm * TI * 32) + l31) * 128;
    int hb[TJ], pty[TJ], ptx[TJ];                           // halo row of this lane's pixel for tap (0, 0); its (ty, tx)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int px = (wn * TJ + j) * 32 + l31;
        pty[j] = px >> tw_log2;
        ptx[j] = px & (TW - 1);
        hb[j] = pty[j] * HW_ + ptx[j];
    }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int wbuf, int hbuf, int tap) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const int shift = dy * HW_ + dx;
        const char* pa = fa + wbuf * W_BYTES;
        const char* ph = sH + hbuf * kHaloBytes;
        int hr[TJ], hsw[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            hr[j] = hb[j] + shift;
            hsw[j] = (((ptx[j] + dx) >> 1) + (((pty[j] + dy) & 1) << 2) * f_hy) & 7;
        }
        // The fragments of k-step ks + 1 are requested before the MFMAs of k-step ks are issued (hipcc on its own reads, waits,
        // multiplies, then reads again: the LDS latency of every k-step sat between two MFMA groups): -1.2 ... -2.6 % per conv.
        bf16x8 af[2][TI], bfr[2][TJ];
        auto load = [&](int ks, int b) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
                if (i == 0 || !narrow) af[b][i] = *(const bf16x8*)(pa + i * 32 * 128 + (((ks * 2 + hi) ^ sw_w) << 4));
#pragma unroll
            for (int j = 0; j < TJ; ++j) bfr[b][j] = *(const bf16x8*)(ph + hr[j] * 128 + (((ks * 2 + hi) ^ hsw[j]) << 4));
        };
        load(0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) load(ks + 1, (ks + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TI; ++i)
                if (i == 0 || !narrow) {
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- main loop: weights 2-deep ring per k-tile, halo 2-deep ring per chunk ----
    CONV_STAMP(0);
    stageW(0, 0);
    stageH(0, 0);
    wait_vmcnt_h<0>();
    __syncthreads();
    CONV_STAMP(1);
    int kt = 0;
    for (int c = 0; c < nc; ++c) {
        for (int t = 0; t < 9; ++t, ++kt) {
            if (kt + 1 < nk) stageW(kt + 1, (kt + 1) & 1);
            const bool pre = (t == 0) && (c + 1 < nc);
            if (pre) stageH(c + 1, (c + 1) & 1);            // issued AFTER the weight tile: loads complete in order
            compute(kt & 1, c & 1, t);
            // the next weight tile must have landed; the next halo (needed 8 k-tiles from now) may stay in flight.
            // A wave issues 5 or 6 halo loads (the last issue covers rows 160..183 only): count conservatively.
            if (pre) wait_vmcnt_h<kHaloIssues - 1>(); else wait_vmcnt_h<0>();
            __syncthreads();
        }
    }

    // ---- epilogue ----
    CONV_STAMP(2);
    const int64_t row_base = ((int64_t)frame * d.Hout + y0) * d.Wout + x0;
    gemm_epilogue<WM, WN, TI, TJ, LDS_MAIN>(
        d, acc, smem, ch0,
        [&](int px) -> int64_t {
            const int tx = px & (TW - 1);
            return x0 + tx < d.Wout ? row_base + (int64_t)(px >> tw_log2) * d.Wout + tx : -1;
        },
        (int64_t)frame, narrow);
    CONV_STAMP(3);
#ifdef CONV_PROBE
    if (threadIdx.x == 0 && blockIdx.x < kProbeWgs) {
        unsigned hw_id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_conv_probe[blockIdx.x * kProbeStamps + 4] = ((unsigned long long)xcc << 32) | hw_id;
        g_conv_probe[blockIdx.x * kProbeStamps + 5] = (unsigned long long)ch0;
    }
#endif
}


// ------------------------------------------------------------------------------------------------------------------------------
// Round 6: the same tile with the WEIGHT operand in a four-slot ring of K = 32 half tiles and counted waits — the default (policy
// conv_halo = 1); conv_halo = 2 keeps the two-slot kernel above as the A/B arm (bit-identical results: same k-step order).
// Measured, same process, cold operands (tools/exp/conv_ring_ab.py): 34 x 64 x 96 320 -> 320 424.6 -> 412.4 us, 640 -> 320 787.9 -> 758.9,
// 34 x 32 x 48 640 -> 640 377.2 -> 366.0, 960 -> 640 547.9 -> 523.9, 1920 -> 640 1033.9 -> 1024.1: -1 ... -4.4 %, 14.81 -> 14.39 ms over the
// 29 launches of a step.
//
// What the per-workgroup time stamps of the kernel above say (tools/exp/conv_probe.py, 34 x 64 x 96, 320 -> 320): a full 128-channel
// tile spends 40.1 us in its K loop = 0.89 us per k-tile, a NARROW tile — half the MFMAs, half the weight bytes — 35.5 us = 0.79 us:
// 0.69 us of a k-tile do not depend on the matrix work at all.  That is the round trip of the weight tile: it is requested at the top
// of tap t and `vmcnt(0)` + barrier at the bottom of the same tap wait for it (prefetch distance ONE: what the two workgroups of a CU
// can hide is one tap of the partner's MFMAs, 0.24 us, against 0.6-0.8 us of L2 -> LDS latency under load).  Here the weight tile
// of a tap is two half tiles of K = 32 (128 rows x 64 B = 8 KB), four slots = the same 32 KB, and half tile s + 3 is requested at
// the top of phase s: three phases (1.5 taps of this workgroup, 3 of the CU) of latency cover; the wait at the bottom of a phase is
// COUNTED — it retires half tile s + 1 and leaves s + 2, s + 3 (and, for the three phases after a chunk boundary, the next chunk's
// halo rectangle, which comes from HBM) in flight.  Everything else is the kernel above: tile geometry, halo image and its
// swizzle, narrow last channel tile, block order, epilogue.
//   W half-tile rows are 64 bytes: granule g of row r sits at slot g ^ ((r >> 2) & 3) — the 16 lanes of a ds_read_b128 group hold
//   rows {0-3, 12-15, 20-27} (+ k): every residue r % 4 four times with four different (r >> 2) & 3, i.e. 16 distinct 16-byte bank
//   groups (MI355X_MICROARCH.md, LDS); the DMA writes a wave's 1 KB = rows 16 w .. 16 w + 15 lane-linearly.
constexpr int kHaloBytes4 = 192 * 128;         // whole 32-row DMA issues: every wave issues exactly kHaloIssues loads (exact counted waits)

template <int WM, int WN, int TI, int TJ>
__global__ __launch_bounds__(WM* WN * 64) void conv_halo4_kernel(const CcGemmDesc d, int tw_log2_flags) {
    const int tw_log2 = tw_log2_flags & 0xFF;
    constexpr int NT = WM * WN * 64;
    constexpr int BMC = WM * TI * 32, BNP = WN * TJ * 32;
    static_assert(NT == 256 && BNP == 128 && BMC == 128 && TI == 2, "tile geometry");
    constexpr int NS = 4;                          // weight ring slots
    constexpr int W_SUB = BMC * 64;                // 8 KB: 128 rows x 32 k
    constexpr int LDS_MAIN = NS * W_SUB + 2 * kHaloBytes4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sW = smem;                         // [NS][W_SUB]
    char* const sH = smem + NS * W_SUB;            // [2][kHaloBytes4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int TW = 1 << tw_log2, TH = BNP >> tw_log2, HW_ = TW + 2;
    const int tiles_x = (d.Wout + TW - 1) >> tw_log2, tiles_y = d.Hout / TH;
    const int tpf = tiles_x * tiles_y;

    const int ct_n = (d.N + BMC - 1) / BMC;
    const int64_t pt_n = (int64_t)(d.M / (d.Hout * d.Wout)) * tpf;
    const int64_t pt_per_xcd = (pt_n + 7) / 8;
    const int64_t bid = blockIdx.x;
    const int xcd = (int)(bid & 7);
    const int64_t local = bid >> 3;
    const int Q = d.cgroup > 0 ? d.cgroup : ct_n;
    const int64_t gsz = pt_per_xcd * Q;
    const int cg = (int)(local / gsz);
    const int64_t rr = local - cg * gsz;
    const int qn = min(Q, ct_n - cg * Q);
    const int64_t pl = rr / qn;
    const int64_t pt = xcd * pt_per_xcd + pl;
    if (pt >= pt_n) return;
    const int ch0 = (cg * Q + (int)(rr - pl * qn)) * BMC;
    const int frame = (int)(pt / tpf);
    const int tr = (int)(pt - (int64_t)frame * tpf);
    const int y0 = (tr / tiles_x) * TH, x0 = (tr % tiles_x) << tw_log2;

    const bf16* __restrict__ Ap = (const bf16*)d.A;
    const bf16* __restrict__ Wp = (const bf16*)d.W;
    const bf16* zp = (const bf16*)g_zero_page_h;
    const int nc = d.Cin >> 6, nsub = nc * 18;
    const bool narrow = ((tw_log2_flags >> 8) & 1) && (d.N - ch0 <= WM * 32);

    // ---- staging coordinates ----
    const int f_hy = (tw_log2 == 3) ? 1 : 0;
    const int p = tid & 7, rsub = tid >> 3;        // halo: 8 granules per 128-byte row
    int64_t hoff[kHaloIssues];                     // halo issue i stages rows i * 32 + rsub: element offset without the chunk, or -1 (zero page)
#pragma unroll
    for (int i = 0; i < kHaloIssues; ++i) {
        const int hrow = i * 32 + rsub;
        const int hy = hrow / HW_, hx = hrow - hy * HW_;
        const int iy = y0 + hy - 1, ix = x0 + hx - 1;
        const bool v = hrow < kHaloRows && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        const int gsrc = p ^ (((hx >> 1) + ((hy & 1) << 2) * f_hy) & 7);
        hoff[i] = v ? (((int64_t)frame * d.Hin + iy) * d.Win + ix) * d.lda + gsrc * 8 : -1;      // rows 180 .. 191: zero page (never read)
    }
    const int p4 = tid & 3, r4 = tid >> 2;         // weights: 4 granules per 64-byte row, rows r4 and r4 + 64
    const int gsrc_w = p4 ^ ((r4 >> 2) & 3);
    const bf16* const wsrc = Wp + (size_t)(ch0 + r4) * d.Kpad + gsrc_w * 8;
    const size_t wsrc64 = (size_t)64 * d.Kpad;

    auto stageH = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < kHaloIssues; ++i) {
            const bf16* src = hoff[i] >= 0 ? Ap + hoff[i] + c * 64 : zp;
            glds16(src, sH + buf * kHaloBytes4 + i * (32 * 128) + wave * 1024);
        }
    };

    // ---- fragment coordinates ----
    const int l31 = lane & 31, hi = lane >> 5;
    const int sw_w = (l31 >> 2) & 3;
    int hb[TJ], pty[TJ], ptx[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int px = (wn * TJ + j) * 32 + l31;
        pty[j] = px >> tw_log2;
        ptx[j] = px & (TW - 1);
        hb[j] = pty[j] * HW_ + ptx[j];
    }

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    CONV_STAMP(0);
    // The K loop, compiled once per tile kind (NARROW: one MFMA row tile per wave and one weight issue per half tile — no selects
    // or branches around the MFMAs).
    auto kloop = [&](auto NARROWC) {
        constexpr bool NARROW = decltype(NARROWC)::value;
        constexpr int WI = NARROW ? 1 : 2;         // DMA instructions per thread and weight half tile
        constexpr int NTI = NARROW ? 1 : TI;
        const char* const fa = sW + ((NARROW ? wm * 32 : wm * TI * 32) + l31) * 64;
        auto stageW = [&](int sidx) {
            const bf16* src = wsrc + sidx * 32;
            char* dst = sW + (sidx & (NS - 1)) * W_SUB + wave * 1024;
#pragma unroll
            for (int i = 0; i < WI; ++i) glds16(src + i * wsrc64, dst + i * 4096);
        };
        auto wait_for = [&](int w_tiles, bool halo) {          // at most (w_tiles weight half tiles [+ the halo rectangle]) still in flight
            if (halo) {
                if (w_tiles >= 2) wait_vmcnt_h<2 * WI + kHaloIssues>();
                else if (w_tiles == 1) wait_vmcnt_h<WI + kHaloIssues>();
                else wait_vmcnt_h<kHaloIssues>();
            } else {
                if (w_tiles >= 2) wait_vmcnt_h<2 * WI>();
                else if (w_tiles == 1) wait_vmcnt_h<WI>();
                else wait_vmcnt_h<0>();
            }
        };
        // one phase = one weight half tile (two k-steps of 16) against the halo rows of tap (dy, dx)
        auto phase = [&](int sidx, int hbuf, int half, const int (&hr)[TJ], const int (&hsw)[TJ]) {
            const char* pa = fa + (sidx & (NS - 1)) * W_SUB;
            const char* ph = sH + hbuf * kHaloBytes4;
            bf16x8 af[2][NTI], bfr[2][TJ];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                for (int i = 0; i < NTI; ++i) af[k2][i] = *(const bf16x8*)(pa + i * 32 * 64 + (((k2 * 2 + hi) ^ sw_w) << 4));
#pragma unroll
                for (int j = 0; j < TJ; ++j) bfr[k2][j] = *(const bf16x8*)(ph + hr[j] * 128 + ((((half * 2 + k2) * 2 + hi) ^ hsw[j]) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int i = 0; i < NTI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k2][i], bfr[k2][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };

        // prologue: half tiles 0, 1, 2 and the first halo rectangle; 0 and the halo must have landed
        stageW(0);
        stageH(0, 0);
        stageW(1);
        stageW(2);
        wait_vmcnt_h<2 * WI>();
        __syncthreads();
        CONV_STAMP(1);
        int sidx = 0;
        for (int c = 0; c < nc; ++c) {
            const bool more_chunks = c + 1 < nc;
            for (int t = 0; t < 9; ++t) {
                const int dy = t / 3, dx = t - dy * 3;
                const int shift = dy * HW_ + dx;
                int hr[TJ], hsw[TJ];
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    hr[j] = hb[j] + shift;
                    hsw[j] = (((ptx[j] + dx) >> 1) + (((pty[j] + dy) & 1) << 2) * f_hy) & 7;
                }
#pragma unroll
                for (int half = 0; half < 2; ++half, ++sidx) {
                    if (sidx + 3 < nsub) stageW(sidx + 3);
                    const bool first = (t == 0 && half == 0);
                    if (first && more_chunks) stageH(c + 1, (c + 1) & 1);       // AFTER the weight request: the youngest loads of the queue
                    phase(sidx, c & 1, half, hr, hsw);
                    // half tile sidx + 1 must have landed; sidx + 2, sidx + 3 may stay in flight, and so may the next chunk's halo while
                    // it is younger than the half tile waited for (the three phases after it was requested)
                    const int left = nsub - 2 - sidx;                               // half tiles beyond sidx + 1
                    wait_for(left < 0 ? 0 : left, more_chunks && t == 0 || (more_chunks && t == 1 && half == 0));
                    __syncthreads();
                }
            }
        }
    };
    if (narrow) kloop(std::true_type{});
    else kloop(std::false_type{});

    // ---- epilogue ----
    CONV_STAMP(2);
    const int64_t row_base = ((int64_t)frame * d.Hout + y0) * d.Wout + x0;
    gemm_epilogue<WM, WN, TI, TJ, LDS_MAIN>(
        d, acc, smem, ch0,
        [&](int px) -> int64_t {
            const int tx = px & (TW - 1);
            return x0 + tx < d.Wout ? row_base + (int64_t)(px >> tw_log2) * d.Wout + tx : -1;
        },
        (int64_t)frame, narrow);
    CONV_STAMP(3);
#ifdef CONV_PROBE
    if (threadIdx.x == 0 && blockIdx.x < kProbeWgs) {
        g_conv_probe[blockIdx.x * kProbeStamps + 4] = 0;
        g_conv_probe[blockIdx.x * kProbeStamps + 5] = (unsigned long long)ch0;
    }
#endif
}

}  // namespace

#ifdef CONV_PROBE
extern "C" int ccedit_conv_probe_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_conv_probe), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost);
}
#endif

bool cc_conv_halo_applicable(const CcGemmDesc& d) {
    if (!(d.mode == CCEDIT_GEMM_CONV2D && d.taps == 9 && d.ksize == 3 && d.stride == 1 && d.pad == 1 && !d.upsample &&
          !d.A2 && d.korder == 1 && d.Cin % 64 == 0 && d.Cin1 == d.Cin && d.Hout == d.Hin && d.Wout == d.Win && d.N >= 64 &&
          d.act != CCEDIT_ACT_GEGLU))
        return false;
    if (d.M % ((int64_t)d.Hout * d.Wout) != 0) return false;
    if (d.gn_stats && d.gn_rows != d.Hout * d.Wout) return false;
    // whole rectangles vertically; a ragged last column (8x12 frames: 12 of 16 columns used) is masked
    return d.Hout % 8 == 0 || (d.Wout % 8 == 0 && d.Hout % 16 == 0);
}

int cc_conv_halo_launch(const CcGemmDesc& d, hipStream_t s) {
    constexpr int WM = 2, WN = 2, TI = 2, TJ = 2;
    constexpr int BMC = WM * TI * 32, BNP = WN * TJ * 32;
    constexpr int lds = epi_lds_total(BMC, BNP, TJ, 2 * BMC * 128 + 2 * kHaloBytes);
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)conv_halo_kernel<WM, WN, TI, TJ>, lds, &attr_done, "conv_halo")) return rc;
    // orientation with the least padding: 8 x 16 needs Hout % 8 == 0, 16 x 8 needs Hout % 16 == 0
    const int pad16 = (d.Hout % 8 == 0) ? (d.Wout + 15) / 16 * 16 : 1 << 30;
    const int pad8 = (d.Hout % 16 == 0) ? (d.Wout + 7) / 8 * 8 : 1 << 30;
    const int tw_log2 = pad16 <= pad8 ? 4 : 3;
    const int TWh = 1 << tw_log2, THh = BNP >> tw_log2;
    const int64_t frames = d.M / ((int64_t)d.Hout * d.Wout);
    const int64_t pt_n = frames * ((d.Wout + TWh - 1) / TWh) * (d.Hout / THh), ct_n = (d.N + BMC - 1) / BMC;
    const int64_t nblk = 8 * ((pt_n + 7) / 8) * ct_n;
    if (nblk > 2147483647LL) {
        cc_set_error("ccedit_gemm: grid too large");
        return CCEDIT_EUNSUPPORTED;
    }
    CcGemmDesc dd = d;
    dd.cgroup = 0;                                // weights of these convs exceed L2: share a weight tile among the
    if (ct_n > 3) {                               // resident workgroups (see the block-order note in gemm.hip)
        const int q = 3;
        const int ng = (int)((ct_n + q - 1) / q);
        dd.cgroup = (int)((ct_n + ng - 1) / ng);
    }
    if (cc_policy().conv_halo != 2) {            // the four-slot weight ring (round 6; policy conv_halo = 2: the two-slot kernel of rounds 2-5, the A/B arm)
        constexpr int lds4 = epi_lds_total(BMC, BNP, TJ, 4 * BMC * 64 + 2 * kHaloBytes4);
        static unsigned long long attr_done4 = 0;
        if (int rc = cc_max_dynamic_lds((const void*)conv_halo4_kernel<WM, WN, TI, TJ>, lds4, &attr_done4, "conv_halo4")) return rc;
        cc_note_kernel("conv_halo_kernel");
        hipLaunchKernelGGL((conv_halo4_kernel<WM, WN, TI, TJ>), dim3((unsigned)nblk), dim3(WM * WN * 64), lds4, s, dd, tw_log2 | (1 << 8));
        return cc_launch_status("conv_halo4_kernel");
    }
    cc_note_kernel("conv_halo_kernel");
    hipLaunchKernelGGL((conv_halo_kernel<WM, WN, TI, TJ>), dim3((unsigned)nblk), dim3(WM * WN * 64), lds, s, dd,
                       tw_log2 | (1 << 8));
    return cc_launch_status("conv_halo_kernel");
}
