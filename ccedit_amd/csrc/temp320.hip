// Conv1d k3 over the T keyframes of a clip at 320 input channels (ResBlock3D in / out layers, Down / Upsample3D of the 64x96 level:
// openaimodel.py:617-629, 674-687) as a STREAMING kernel with the weights held in registers — gfx950.
//
// Every block shape of the tiled GEMMs lands at 230-290 us for this layer (208896 rows, 128 GFLOP, 0.4-0.5 GB of operands:
// a tile sweep, round 4): with 320 output channels a pixel tile meets 3-5 channel tiles, and each of them pulls the rows of
// three frames through the L2 -> LDS path again.  Here (the scheme of lin320s / lin640s, lin640.hip) a workgroup keeps the
// weights of ALL THREE TAPS for a 128-channel slice in registers — wave w: 16 channels x 3 taps x 320 k = 30 A fragments, 120 VGPRs —
// and walks pixel COLUMNS (clip, 16 pixels) frame by frame: the 16 x 320 activation tile of frame t arrives once by DMA and every
// fragment read from LDS feeds three MFMAs — tap +1 of output t-1, tap 0 of output t, tap -1 of output t+1 (three rolling
// accumulators, zero padding at the clip ends is simply the missing MFMA).  Output t-1 is complete after step t: bias and the
// per-clip row bias were its initial value, the residual cells come from tiles that travelled with the activations, it leaves as
// bf16 through the first residual's tile (in place) and goes to memory one step later as whole 256-byte row pieces.
//
// DMA by waves 0-3 only, stores / atomics by waves 4-7 (the counted vmcnt wait is then exact), ring of six 12 KB activation tiles
// (rows of 768 B, granules XOR-swizzled by the row: conflict-free ds_read_b128 in the 16x16x32 B layout), one barrier per step.
// GroupNorm(32) statistics of the stored values (gn_stats): per lane the sums of its 4 channels split at the group boundary, a
// butterfly over the 16 pixels, exact double adds in LDS across waves, double atomics to the frame's slots one step later.
// Requires Cin = 320 (Kpad = 960), N % 64 == 0, HW % 16 == 0, unsharded frames.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int kC = 320, kKS = kC / 32;          // 10 MFMA k-steps per tap
constexpr int kP = 16;                          // pixels per tile
constexpr int kRS = 768;                        // activation row in LDS: 48 granules, 40 used; granule j at (j & 48) | ((j & 15) ^ (row & 15))
constexpr int kXBuf = kP * kRS;                 // 12,288 B = 12 DMA instructions, three per requesting wave
constexpr int kSlice = 128;
constexpr int kOBuf = kP * kSlice * 2;          // residual / output tile: 16 rows of 256 B = 4 DMA instructions, granule g of row r at g ^ r
constexpr int kGB = 512;                        // row-bias slice of a tile's clip: 128 floats, 8 lanes per requesting wave
constexpr int kRX = 6;                          // activation ring: tile s + 5 is requested at step s
constexpr int kRR = 9;                          // residual rings: staged 5 ahead, cells read at step + 2, the output (in R1) stored at step + 3
constexpr int kNT = 512;
constexpr int kLds = kRX * kXBuf + 2 * kRR * kOBuf + kRX * kGB + 2 * 16 * 16;      // 151,040 B (incl. two [16][2] double statistics arrays)
constexpr int kXD = 5;                          // activation fragments in flight per wave

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void t3_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// These loops are bound by INSTRUCTION ISSUE, not by a pipe: a wave issues at most one instruction per four cycles, a step has 30 MFMAs
// (480 cycles) per wave, and the first version of this kernel spent ~860 instructions per step and wave — 500 of them scalar (ring
// wrap-arounds, 64-bit address products, exec-mask juggling around lane-dependent DMA selects, a 45-way switch for the counted
// wait, SGPR spills) — i.e. 1.5 us per step whatever the memory system or the matrix pipe did (ablation: all of MFMA, DMA and
// stores switched off left 0.56 us per step).  Hence: ring slots as wrap-around counters (three scalar instructions each), running byte
// offsets instead of row x stride products, DMA sources valid for every lane (padding slots copy a neighbouring granule: no
// selects, no exec masks), the same number of loads per tile in every requesting wave (compile-time wait counts).
template <int NRES, bool GB>
__global__ __launch_bounds__(kNT, 1) void temp320s_kernel(const CcGemmDesc d, int nslice, int ncol_clip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sX = smem;
    char* const sR1 = smem + kRX * kXBuf;                         // NRES == 0: used as the output ring
    char* const sR2 = sR1 + kRR * kOBuf;
    char* const sGB = sR2 + kRR * kOBuf;
    double* const sSt = (double*)(sGB + kRX * kGB);              // [2][16 groups][2]
    constexpr int kPerTile = 3 + NRES + (GB ? 1 : 0);            // loads of one tile in a requesting wave's queue
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, g4 = lane >> 4;

    // workgroup b runs on XCD b % 8: its 32 workgroups take (32 / nslice) column lanes x nslice channel slices
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int lanes = 32 / nslice;
    const int slice = j % nslice, plane = j / nslice;
    if (plane >= lanes) return;
    const int T = d.T, HW = d.HW;
    const int ncol = (int)(d.M / ((int64_t)T * HW)) * ncol_clip; // columns: (clip, 16-pixel block)
    const int per_xcd = (ncol + 7) >> 3;
    const int c_lo = xcd * per_xcd, c_hi = min(c_lo + per_xcd, ncol);
    const int col0 = c_lo + plane;
    if (col0 >= c_hi) return;
    const int mycols = (c_hi - col0 + lanes - 1) / lanes;
    const int nstep = mycols * T;                                // step s = (column s / T of this workgroup, frame s % T)
    const int ch0 = slice * kSlice;
    const bool live = ch0 + 16 * wave < d.N;                     // N = 320: waves 4..7 of the third slice have no channels
    const bool gn = d.gn_stats != nullptr;
    const bool dma_wave = wave < 4;

    // ---- DMA plan (waves 0..3, every lane of every instruction has a valid source) ----
    // activations: destination slot (r, p) <- source granule jg = (p & 48) | ((p & 15) ^ (r & 15)); the 8 unused positions of a row
    // (jg >= 40) copy granule jg - 8.  Residual tiles: slot (r, p) <- granule p ^ r of the slice's 16 (channels beyond N: clamped).
    int soffx[3], soff1 = 0, soff2 = 0, soffg = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int n = (i * 4 + (wave & 3)) * 64 + lane;
        const int r = n / 48, p = n - r * 48;
        int jg = (p & 48) | ((p & 15) ^ (r & 15));
        jg = jg < kC / 8 ? jg : jg - 8;
        soffx[i] = (r * d.lda + jg * 8) * 2;                     // bytes
    }
    {
        const int n = (wave & 3) * 64 + lane, r = n >> 4, g = (n & 15) ^ r;
        const int cg = min(ch0 + g * 8, d.N - 8);
        if constexpr (NRES >= 1) soff1 = (r * d.ldr1 + cg) * 2;
        if constexpr (NRES >= 2) soff2 = (r * d.ldr2 + cg) * 2;
        soffg = min(ch0 + ((wave & 3) * 8 + (lane & 7)) * 4, d.N - 4) * 4;      // row bias: lanes 0..7 of wave w fetch floats [32 w, 32 w + 32)
    }
    const char* const Ab = (const char*)d.A;
    const char* const R1b = (const char*)d.res1;
    const char* const R2b = (const char*)d.res2;
    const char* const Gb = (const char*)d.group_bias;
    const int64_t x_step = (int64_t)HW * d.lda * 2, r1_step = (int64_t)HW * d.ldr1 * 2, r2_step = (int64_t)HW * d.ldr2 * 2, o_step = (int64_t)HW * d.ldc * 2;
    const int gb_ld = (d.ldgb ? d.ldgb : d.N) * 4;

    // ---- the weights of this wave: A fragments [tap][k-step] of channels ch0 + 16 w + (lane & 15); bias of the lane's 4 channels ----
    bf16x8 wf[3][kKS];
    f32x4 bq;
    {
        const int cr = min(ch0 + 16 * wave + c16, d.N - 1);
        const bf16* __restrict__ row = (const bf16*)d.W + (size_t)cr * d.Kpad;
        // fragment-ordered copy (CcGemmDesc.Wfrag): one contiguous kilobyte per fragment and wave (rows are padded to 256: a wave
        // beyond N reads zeros); the 32-element k block of (tap, ks) is the one its g4 = 0 lanes start
        const bf16* __restrict__ blk = (const bf16*)d.Wfrag + ((size_t)((ch0 >> 4) + wave) * (d.Kpad / 32) * 64 + lane) * 8;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                const int c0 = 32 * ks;
                const int off0 = d.korder ? (c0 >> 6) * 192 + tap * 64 + (c0 & 63) : tap * kC + c0;
                if (d.Wfrag) wf[tap][ks] = *(const bf16x8*)(blk + (off0 >> 5) * 512);
                else wf[tap][ks] = *(const bf16x8*)(row + off0 + 8 * g4);
            }
        const int cb = min(ch0 + 16 * wave + 4 * g4, d.N - 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[e] = d.bias ? d.bias[cb + e] : 0.f;
    }
    if (tid < 64) sSt[tid] = 0.0;

    int xlane[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xlane[m] = c16 * kRS + (((4 * m + g4) ^ c16) << 4);
    // C cell (pixel c16, channels 16 w + 4 g4 .. + 3) = half (g4 & 1) of granule 2 w + (g4 >> 1), at position granule ^ pixel
    const int olane = c16 * 256 + (((2 * wave + (g4 >> 1)) ^ c16) << 4) + (g4 & 1) * 8;
    const int ot0 = tid & 255;                                   // output pass (waves 4..7): granule ot0 % 16 of row ot0 / 16
    const int olds = (ot0 >> 4) * 256 + (((ot0 & 15) ^ (ot0 >> 4)) << 4);
    const int oout = ((ot0 >> 4) * d.ldc + ch0 + (ot0 & 15) * 8) * 2;          // bytes
    const bool ost = !dma_wave && ch0 + (ot0 & 15) * 8 < d.N;
    // GroupNorm groups of the lane's 4 channels: cpg channels per group; the quad may straddle one boundary
    const int cpg = d.N >> 5;
    const int cbq = ch0 + 16 * wave + 4 * g4;
    const int gq0 = cbq / cpg;
    int nlo = (gq0 + 1) * cpg - cbq;                             // channels of the quad in group gq0
    nlo = nlo > 4 ? 4 : nlo;
    const int g_first = ch0 / cpg;                               // first group this slice touches
    const int ngrp = (min(ch0 + kSlice, d.N) - 1) / cpg - g_first + 1;      // <= 14

    // ---- cursors: running byte offsets of the tile being requested and of the output being stored; (column, frame) counters ----
    auto col_row0 = [&](int ci, int& clip) -> int64_t {          // first row of column ci of this workgroup (frame 0 of its clip)
        const int col = col0 + ci * lanes;
        clip = col / ncol_clip;
        return (int64_t)clip * T * HW + (int64_t)(col - clip * ncol_clip) * kP;
    };
    int st_t = 0, st_col = 0, st_clip;
    int64_t st_row0 = col_row0(0, st_clip);
    int64_t st_x = st_row0 * d.lda * 2, st_r1 = NRES >= 1 ? st_row0 * d.ldr1 * 2 : 0, st_r2 = NRES >= 2 ? st_row0 * d.ldr2 * 2 : 0;
    int st_xs = 0, st_rs = 0;                                    // ring slots of the tile being requested
    auto stage_next = [&]() {
        if (dma_wave) {
            char* const xd = sX + st_xs * kXBuf;
#pragma unroll
            for (int q = 0; q < 3; ++q) glds16(Ab + st_x + soffx[q], xd + (q * 4 + wave) * 1024);
            if constexpr (NRES >= 1) glds16(R1b + st_r1 + soff1, sR1 + st_rs * kOBuf + wave * 1024);
            if constexpr (NRES >= 2) glds16(R2b + st_r2 + soff2, sR2 + st_rs * kOBuf + wave * 1024);
            if constexpr (GB) {
                if (lane < 8) glds16(Gb + (int64_t)st_clip * gb_ld + soffg, sGB + st_xs * kGB + wave * 128);
            }
        }
        st_xs = st_xs == kRX - 1 ? 0 : st_xs + 1;
        st_rs = st_rs == kRR - 1 ? 0 : st_rs + 1;
        if (++st_t == T) {
            st_t = 0;
            if (++st_col < mycols) {
                st_row0 = col_row0(st_col, st_clip);
                st_x = st_row0 * d.lda * 2;
                if constexpr (NRES >= 1) st_r1 = st_row0 * d.ldr1 * 2;
                if constexpr (NRES >= 2) st_r2 = st_row0 * d.ldr2 * 2;
            }
        } else {
            st_x += x_step;
            if constexpr (NRES >= 1) st_r1 += r1_step;
            if constexpr (NRES >= 2) st_r2 += r2_step;
        }
    };
    auto wait_later = [&](int later) {                            // at most the loads of `later` (<= kRX - 2) newer tiles may be outstanding
        if (later >= 4) t3_vmcnt<4 * kPerTile>();
        else if (later == 3) t3_vmcnt<3 * kPerTile>();
        else if (later == 2) t3_vmcnt<2 * kPerTile>();
        else if (later == 1) t3_vmcnt<kPerTile>();
        else t3_vmcnt<0>();
    };
    static_assert(kRX - 2 == 4, "wait_later covers four tiles in flight behind the awaited one");
    int staged = 0;
    for (; staged < kRX - 1 && staged < nstep; ++staged) stage_next();
    // (round 6) the first tiles are requested BEHIND the weight loads, in the same breath — one latency instead of two at the start of
    // every launch; this wait covers both: from here on a requesting wave's queue holds DMA only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int cx = 0, fr = kRR - 2, sr = kRR - 3;                      // slots: activations of tile s; cells of output s - 2; output s - 3
    // output cursor: the output stored at step s is s - 3
    int so_t = 0, so_col = 0, so_clip;
    int64_t so_off = col_row0(0, so_clip) * d.ldc * 2;
    int so_frame = so_clip * T;

    f32x4 accp = {0.f, 0.f, 0.f, 0.f}, accc = accp, accn = accp, accd = accp;      // accd: the complete output s - 2 waiting for its epilogue
    int tcur = 0;

    // row sum over the 16 pixels of a lane row: inclusive scan by row_shr 1, 2, 4, 8 — lane 15 of the row holds the total
    auto rowsum = [&](float v) {
        auto shr = [&](float x, auto n_) {
            constexpr int N = decltype(n_)::value;
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x110 + N, 0xf, 0xf, true));
        };
        v += shr(v, std::integral_constant<int, 1>{});
        v += shr(v, std::integral_constant<int, 2>{});
        v += shr(v, std::integral_constant<int, 4>{});
        v += shr(v, std::integral_constant<int, 8>{});
        return v;
    };
    // epilogue of a complete output: acc (+ residual cells) -> bf16 cell in LDS (in place over the first residual); its GroupNorm sums
    auto finish = [&](int parity, const f32x4& acc, const bf16x4& rc1, const bf16x4& rc2, int oslot) {
        f32x4 v = acc;
        if constexpr (NRES >= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(rc1[e]);
        }
        if constexpr (NRES >= 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(rc2[e]);
        }
        bf16x4 ob;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ob[e] = f2bf(v[e]);
            const float f = bf2f(ob[e]);
            if (e < nlo) { s0 += f; q0 += f * f; } else { s1 += f; q1 += f * f; }
        }
        *(bf16x4*)(sR1 + oslot * kOBuf + olane) = ob;
        if (gn) {
            s0 = rowsum(s0);
            q0 = rowsum(q0);
            s1 = rowsum(s1);
            q1 = rowsum(q1);
            if (c16 == 15) {
                const uint32_t a0 = (uint32_t)(uintptr_t)(LDS_AS char*)(sSt + parity * 32 + 2 * (gq0 - g_first));
                asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:8" ::"v"(a0), "v"((double)s0), "v"((double)q0) : "memory");
                if (nlo < 4) asm volatile("ds_add_f64 %0, %1 offset:16\n\tds_add_f64 %0, %2 offset:24" ::"v"(a0), "v"((double)s1), "v"((double)q1) : "memory");
            }
        }
    };
    auto flush_stats = [&](int parity, int frame) {
        if (gn && wave == 5 && lane < 2 * ngrp) {
            double* const a = sSt + parity * 32 + lane;
            const double v = *a;
            *a = 0.0;
            unsafeAtomicAdd(d.gn_stats + (int64_t)frame * 64 + 2 * g_first + lane, v);
        }
    };

    // Step s: [barrier] request tile s + 3 | stores of output s - 3 | MFMAs of tile s with the epilogue of output s - 2 in their shadow |
    // output s - 1 becomes complete.  Tile k's activations are in slot k % 6, its residual cells (later its output) in slot k % 9.
    for (int s = 0; s < nstep + 2; ++s) {
        lds_barrier();
        const bool work = s < nstep;                             // the last two iterations only drain the epilogue / store pipeline
        if (staged < nstep) {
            stage_next();
            ++staged;
        }
        // stores of output s - 3 (its cells were written during step s - 1), first half: out of LDS; its statistics to memory
        bf16x8 ov;
        const bool have_out = s >= 3;
        if (have_out && !dma_wave) {
            // (stored right away, while the requesting waves issue their DMA: a store inside the MFMA loop waits for `ov` with the
            //  fragment prefetch already in the LDS queue — the storing waves were 350 cycles per step behind the others)
            ov = *(const bf16x8*)(sR1 + sr * kOBuf + olds);
            flush_stats((s - 3) & 1, so_frame);
            if (ost) *(bf16x8*)((char*)d.out + so_off + oout) = ov;
        }
        // epilogue of output s - 2, first half: its residual cells
        const bool have_fin = s >= 2 && live;
        const int o_fin = fr;
        bf16x4 rc1, rc2;
        if (have_fin) {
            if constexpr (NRES >= 1) rc1 = *(const bf16x4*)(sR1 + o_fin * kOBuf + olane);
            if constexpr (NRES >= 2) rc2 = *(const bf16x4*)(sR2 + o_fin * kOBuf + olane);
        }
        const char* const xt = sX + cx * kXBuf;
        f32x4 init = bq;
        if constexpr (GB) {
            if (work) {
                const f32x4 gbv = *(const f32x4*)(sGB + cx * kGB + (16 * wave + 4 * g4) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) init[e] += gbv[e];
            }
        }
        if (tcur == 0) accc = init;
        accn = init;
        const bool has_prev = tcur > 0, has_next = tcur + 1 < T;
        auto taps = [&](auto hp_, auto hn_) {
            constexpr bool HP = decltype(hp_)::value, HN = decltype(hn_)::value;
            bf16x8 xq[kXD];
#pragma unroll
            for (int ks = 0; ks < kXD - 1; ++ks) xq[ks] = *(const bf16x8*)(xt + xlane[ks & 3] + (ks >> 2) * 256);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if (ks + kXD - 1 < kKS) xq[(ks + kXD - 1) % kXD] = *(const bf16x8*)(xt + xlane[(ks + kXD - 1) & 3] + ((ks + kXD - 1) >> 2) * 256);
                __builtin_amdgcn_sched_barrier(0);
                accc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][ks], xq[ks % kXD], accc, 0, 0, 0);
                if constexpr (HP) accp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][ks], xq[ks % kXD], accp, 0, 0, 0);      // x[t] is the t + 1 neighbour of output t - 1
                if constexpr (HN) accn = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][ks], xq[ks % kXD], accn, 0, 0, 0);      // ... the t - 1 neighbour of output t + 1
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 5 && have_fin) {
                    finish(s & 1, accd, rc1, rc2, o_fin);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (live && work) {
            using Y = std::true_type;
            using N_ = std::false_type;
            if (has_prev && has_next) taps(Y{}, Y{});
            else if (has_next) taps(N_{}, Y{});
            else taps(Y{}, N_{});                                 // (T >= 2: the last frame of a column has a predecessor)
        } else {
            if (have_fin) finish(s & 1, accd, rc1, rc2, o_fin);
        }
        // tile s + 1 must have landed (this wave's part): at most the loads of the tiles requested after it may be outstanding
        if (dma_wave && s + 1 < nstep) wait_later(staged - (s + 2));
        cx = cx == kRX - 1 ? 0 : cx + 1;
        fr = fr == kRR - 1 ? 0 : fr + 1;
        sr = sr == kRR - 1 ? 0 : sr + 1;
        // rotate: output s - 1 is complete now (its last tap was this step's; at a column start it was complete already)
        accd = accp;
        accp = accc;
        accc = accn;
        if (++tcur == T) tcur = 0;
        if (have_out) {                                          // the output cursor follows three steps behind
            if (++so_t == T) {
                so_t = 0;
                if (++so_col < mycols) {
                    so_off = col_row0(so_col, so_clip) * d.ldc * 2;
                    so_frame = so_clip * T;
                }
            } else {
                so_off += o_step;
                ++so_frame;
            }
        }
    }
    // the last output's cells (written in iteration nstep + 1) and statistics
    lds_barrier();
    if (!dma_wave) {
        const bf16x8 ov = *(const bf16x8*)(sR1 + sr * kOBuf + olds);
        flush_stats((nstep - 1) & 1, so_frame);
        if (ost) *(bf16x8*)((char*)d.out + so_off + oout) = ov;
    }
}

}  // namespace

bool cc_temp320_applicable(const CcGemmDesc& d) {
    return d.mode == CCEDIT_GEMM_TEMPORAL && d.taps == 3 && d.A2 == nullptr && d.Cin == kC && d.Kpad == 3 * kC && d.N % 64 == 0 && d.N >= 64 &&
           d.N <= 32 * kSlice && d.T >= 2 && d.HW % kP == 0 && d.M % ((int64_t)d.T * d.HW) == 0 && (d.Tsrc == 0 || (d.Tsrc == d.T && d.tsrc_off == 0 && d.t0 == 0 && d.Tglob == d.T)) && !d.out_f32 &&
           d.act == CCEDIT_ACT_NONE && d.ldc % 8 == 0 && (d.res1 == nullptr || d.ldr1 % 8 == 0) && (d.res2 == nullptr || (d.res1 && d.ldr2 % 8 == 0)) &&
           d.ln_eps == 0.f && !d.ln_stats && !d.ln_sums && !d.row_sums && !d.subpix && !d.vpad &&
           (d.group_bias == nullptr || (d.group_rows == d.T * d.HW && (d.ldgb == 0 || d.ldgb % 4 == 0) && d.N % 4 == 0)) &&
           (d.gn_stats == nullptr || (d.gn_rows == d.HW && d.N % 32 == 0 && (d.N >> 5) >= 10));      // (a slice then touches <= 14 groups)
}

int cc_temp320_launch(const CcGemmDesc& d, hipStream_t s) {
    static unsigned long long attr_done[6] = {0, 0, 0, 0, 0, 0};
    if (int rc = cc_max_dynamic_lds((const void*)temp320s_kernel<0, false>, kLds, &attr_done[0], "temp320s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)temp320s_kernel<1, false>, kLds, &attr_done[1], "temp320s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)temp320s_kernel<2, false>, kLds, &attr_done[2], "temp320s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)temp320s_kernel<0, true>, kLds, &attr_done[3], "temp320s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)temp320s_kernel<1, true>, kLds, &attr_done[4], "temp320s")) return rc;
    if (int rc = cc_max_dynamic_lds((const void*)temp320s_kernel<2, true>, kLds, &attr_done[5], "temp320s")) return rc;
    cc_note_kernel("temp320s_kernel");
    const int nslice = (d.N + kSlice - 1) / kSlice;
    const int ncol_clip = d.HW / kP;
    const int nres = d.res2 ? 2 : d.res1 ? 1 : 0;
    const dim3 grid(256), block(kNT);
    if (d.group_bias) {
        if (nres == 2) hipLaunchKernelGGL((temp320s_kernel<2, true>), grid, block, kLds, s, d, nslice, ncol_clip);
        else if (nres == 1) hipLaunchKernelGGL((temp320s_kernel<1, true>), grid, block, kLds, s, d, nslice, ncol_clip);
        else hipLaunchKernelGGL((temp320s_kernel<0, true>), grid, block, kLds, s, d, nslice, ncol_clip);
    } else {
        if (nres == 2) hipLaunchKernelGGL((temp320s_kernel<2, false>), grid, block, kLds, s, d, nslice, ncol_clip);
        else if (nres == 1) hipLaunchKernelGGL((temp320s_kernel<1, false>), grid, block, kLds, s, d, nslice, ncol_clip);
        else hipLaunchKernelGGL((temp320s_kernel<0, false>), grid, block, kLds, s, d, nslice, ncol_clip);
    }
    return cc_launch_status("temp320s_kernel");
}
