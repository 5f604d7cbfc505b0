// Shared epilogue of the MFMA GEMM kernels (tap_gemm_kernel in gemm.hip, conv_halo_kernel in convhalo.hip):
// accumulators -> LDS (fp32, [pixel][channel]) -> row-wise, fully coalesced stores with bias, timestep-embedding row
// bias, SiLU / quick-GELU / GEGLU, two residuals and the optional GroupNorm(32) statistics of the tensor written.
#pragma once
#include "common.h"
// Output row of GEMM row m: m itself, or — CcGemmDesc.subpix — the pixel (2 oy + py, 2 ox + px) of the 2x-upsampled frame
__device__ __forceinline__ size_t cc_out_row(const CcGemmDesc& d, int64_t m) {
    if (!d.subpix) return (size_t)m;
    const int hw = d.Hout * d.Wout;
    const int n = (int)(m / hw), rem = (int)(m - (int64_t)n * hw);
    const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
    return ((size_t)n * 2 * d.Hout + 2 * oy + ((d.subpix - 1) >> 1)) * (2 * d.Wout) + 2 * ox + ((d.subpix - 1) & 1);
}
#include "common.h"

// Dynamic LDS a kernel needs so that the epilogue can stage at least one wave column (host and device agree on it).
constexpr int epi_lds_total(int bmc, int bnp, int tj, int lds_main) {
    const int erow = bmc * 4 + 16, full = bnp * erow, one_col = tj * 32 * erow, one_tile = 32 * erow;
    if (full > lds_main && full <= 70 * 1024) return full;
    if (lds_main >= one_col) return lds_main;
    return lds_main >= one_tile ? lds_main : one_col;      // wide channel tiles: stage 32 pixels at a time
}

// Largest whole number of `unit`-pixel groups that fits `cap` staged rows and divides the block's pixel count.
constexpr int epi_chunk_pixels(int cap, int unit, int bnp) {
    int best = 0;
    for (int e = unit; e <= bnp && e <= cap; e += unit)
        if (bnp % e == 0) best = e;
    return best;
}

// In the MFMA layout a lane owns 4 channels of 32 different pixels, i.e. 8-byte pieces of 32 different output rows
// per store.  Staging the tile through LDS turns that into 16-byte-per-lane accesses that walk each pixel row
// contiguously (bias / timestep-embedding / activation / residual reads use the same coalesced pattern).
//   rowmap(p): output row (pixel index into out / residuals / group_bias) of tile pixel p in [0, BNP), or -1 when the
//              tile pixel lies outside the tensor; stat_frame: frame index for the GroupNorm statistics.
//   narrow: the block computed only its first WM x 32 channels, one MFMA row tile per wave row (conv_halo_kernel's last
//           channel tile when fewer than BMC / 2 channels are left): accumulator row tile 0 of wave row wm is channels
//           wm * 32 .. + 31 and the other row tiles are not staged.
template <int WM, int WN, int TI, int TJ, int LDS_MAIN, class RowMap>
__device__ __forceinline__ void gemm_epilogue(const CcGemmDesc& d, f32x16 (&acc)[TI][TJ], char* smem, int ch0,
                                              RowMap rowmap, int64_t stat_frame, bool narrow = false) {
    constexpr int NT = WM * WN * 64;
    constexpr int BMC = WM * TI * 32, BNP = WN * TJ * 32;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int EROW = BMC * 4 + 16;        // +16 B: the ds_write_b128 of 8 consecutive lanes cover all banks
    // (the 128x128 shape allocates 3.5 KB more than its operand ring so that the whole tile is staged at once)
    constexpr int WPIX = TJ * 32;             // pixels per wave column
    constexpr int LDS_TOTAL = epi_lds_total(BMC, BNP, TJ, LDS_MAIN);
    // pixels staged per chunk: whole wave columns, or single 32-pixel MFMA tile columns when a wave column does not fit
    constexpr int EGR = LDS_TOTAL / EROW >= WPIX ? WPIX : 32;
    constexpr int ECH = epi_chunk_pixels(LDS_TOTAL / EROW, EGR, BNP);
    static_assert(ECH >= EGR && BNP % ECH == 0, "epilogue chunking");
    char* const sE = smem;
    const float* __restrict__ bias = d.bias;
    const float* __restrict__ gbias = d.group_bias;
    const bf16* __restrict__ r1 = (const bf16*)d.res1;
    const bf16* __restrict__ r2 = (const bf16*)d.res2;
    float gs[8], gq[8];                       // GroupNorm statistics of this thread's 8 output channels (gn_stats)
#pragma unroll
    for (int e = 0; e < 8; ++e) gs[e] = gq[e] = 0.f;
  for (int ec = 0; ec < BNP / ECH; ++ec) {
    __syncthreads();                          // operand tiles (or the previous chunk) are no longer being read
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        const int pb = wn * WPIX + tj * 32;       // first tile pixel of this wave's MFMA tile column tj
        if (pb >= ec * ECH && pb < (ec + 1) * ECH) {
#pragma unroll
            for (int ti = 0; ti < TI; ++ti) {
                if (narrow && ti > 0) break;
                const int row0 = narrow ? wm * 32 : wm * TI * 32 + ti * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v = {acc[ti][tj][q * 4 + 0], acc[ti][tj][q * 4 + 1], acc[ti][tj][q * 4 + 2], acc[ti][tj][q * 4 + 3]};
                    *(f32x4*)(sE + (pb - ec * ECH + l31) * EROW + (row0 + q * 8 + hi * 4) * 4) = v;
                }
            }
        }
    }
    __syncthreads();

    if (d.act == CCEDIT_ACT_GEGLU) {
        constexpr int CPR = BMC / 16;                    // 16 packed rows = 8 value + 8 gate channels
        const int g = tid % CPR, r0 = tid / CPR;          // (threads past (NT / CPR) * CPR idle when CPR does not divide NT)
        const int rx = ch0 + g * 16;                      // packed row of the 8 values; gates at rx + 8
        if (rx < d.N && r0 < NT / CPR) {
            float bx[8], bg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bx[e] = bias ? bias[rx + e] : 0.f;
                bg[e] = bias ? bias[rx + 8 + e] : 0.f;
            }
            for (int row = r0; row < ECH; row += NT / CPR) {
                const int64_t m = rowmap(ec * ECH + row);
                if (m < 0) continue;
                const char* src = sE + row * EROW + g * 64;
                const f32x4 x0 = *(const f32x4*)(src), x1 = *(const f32x4*)(src + 16);
                const f32x4 g0 = *(const f32x4*)(src + 32), g1 = *(const f32x4*)(src + 48);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = f2bf((x0[e] + bx[e]) * gelu_erf_f(g0[e] + bg[e]));
                    o[4 + e] = f2bf((x1[e] + bx[4 + e]) * gelu_erf_f(g1[e] + bg[4 + e]));
                }
                *(bf16x8*)((bf16*)d.out + cc_out_row(d, m) * d.ldc + (rx >> 1)) = o;
            }
        }
    } else {
        constexpr int CPR = BMC / 8;
        const int g = tid % CPR, r0 = tid / CPR;
        const int cb = ch0 + g * 8;
        if (cb < d.N && r0 < NT / CPR) {
            const bool full = (cb + 8 <= d.N);            // otherwise exactly 4 valid channels (N % 4 == 0)
            float bv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = (bias && (full || e < 4)) ? bias[cb + e] : 0.f;
            // Rows of this thread in the chunk: r0, r0 + NT / CPR, ...  Their residual and per-clip-bias rows are REQUESTED FIRST, for
            // all rows at once (the accumulators are in LDS by now, registers are free): left inside the row loop, every row
            // paid one global-memory round trip before its store (hipcc: load, s_waitcnt vmcnt(0), add, store, next row) — the
            // "store phase" of the short-K Linears and temporal convs was a chain of 4-8 such round trips per workgroup.
            constexpr int RSTEP = NT / CPR;
            constexpr int RPT = (ECH + RSTEP - 1) / RSTEP;           // rows per thread and chunk (compile time)
            constexpr bool PREFETCH = (ECH == BNP) && RPT <= 8;      // whole tile staged: the accumulators are dead, registers are free
                                                                   // (the chunked shapes t3 / t4 / t6 still hold theirs: +96 VGPRs cost them 30-120 %)
            bf16x8 pr1[PREFETCH ? RPT : 1], pr2[PREFETCH ? RPT : 1];
            f32x4 pg0[PREFETCH ? RPT : 1], pg1[PREFETCH ? RPT : 1];
            if constexpr (PREFETCH) {
                if (full && (r1 || r2 || gbias)) {
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const int row = r0 + k * RSTEP;
                        const int64_t m = row < ECH ? rowmap(ec * ECH + row) : -1;
                        if (m >= 0) {
                            if (r1) pr1[k] = *(const bf16x8*)(r1 + (size_t)m * d.ldr1 + cb);
                            if (r2) pr2[k] = *(const bf16x8*)(r2 + (size_t)m * d.ldr2 + cb);
                            if (gbias) {
                                const float* gb = gbias + (size_t)(m / d.group_rows) * (d.ldgb ? d.ldgb : d.N) + cb;
                                pg0[k] = *(const f32x4*)gb;
                                pg1[k] = *(const f32x4*)(gb + 4);
                            }
                        }
                    }
                }
            }
            int kk = -1;
            for (int row = r0; row < ECH; row += NT / CPR) {
                ++kk;
                const int64_t m = rowmap(ec * ECH + row);
                if (m < 0) continue;
                const char* src = sE + row * EROW + g * 32;
                const f32x4 a0 = *(const f32x4*)(src), a1 = *(const f32x4*)(src + 16);
                float v[8] = {a0[0] + bv[0], a0[1] + bv[1], a0[2] + bv[2], a0[3] + bv[3],
                              a1[0] + bv[4], a1[1] + bv[5], a1[2] + bv[6], a1[3] + bv[7]};
                if (gbias) {
                    if (PREFETCH && full) {
#pragma unroll
                        for (int k = 0; k < RPT; ++k)
                            if (k == kk) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    v[e] += pg0[k][e];
                                    v[4 + e] += pg1[k][e];
                                }
                            }
                    } else {
                        const float* gb = gbias + (size_t)(m / d.group_rows) * (d.ldgb ? d.ldgb : d.N) + cb;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (full || e < 4) v[e] += gb[e];
                    }
                }
                if (d.act == CCEDIT_ACT_SILU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                } else if (d.act == CCEDIT_ACT_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
                }
                if (full) {
                    if (r1) {
                        bf16x8 rv;
                        if constexpr (PREFETCH) {
#pragma unroll
                            for (int k = 0; k < RPT; ++k)
                                if (k == kk) rv = pr1[k];
                        } else {
                            rv = *(const bf16x8*)(r1 + (size_t)m * d.ldr1 + cb);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
                    }
                    if (r2) {
                        bf16x8 rv;
                        if constexpr (PREFETCH) {
#pragma unroll
                            for (int k = 0; k < RPT; ++k)
                                if (k == kk) rv = pr2[k];
                        } else {
                            rv = *(const bf16x8*)(r2 + (size_t)m * d.ldr2 + cb);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bf2f(rv[e]);
                    }
                    if (d.out_f32) {
                        float* op = (float*)d.out + cc_out_row(d, m) * d.ldc + cb;
                        *(f32x4*)op = f32x4{v[0], v[1], v[2], v[3]};
                        *(f32x4*)(op + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    } else {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                        *(bf16x8*)((bf16*)d.out + cc_out_row(d, m) * d.ldc + cb) = o;
                        if (d.gn_stats) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float f = bf2f(o[e]);      // statistics of what the consumer will read
                                gs[e] += f;
                                gq[e] += f * f;
                            }
                        }
                    }
                } else {
                    if (r1) {
                        const bf16x4 rv = *(const bf16x4*)(r1 + (size_t)m * d.ldr1 + cb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
                    }
                    if (r2) {
                        const bf16x4 rv = *(const bf16x4*)(r2 + (size_t)m * d.ldr2 + cb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bf2f(rv[e]);
                    }
                    if (d.out_f32) {
                        *(f32x4*)((float*)d.out + cc_out_row(d, m) * d.ldc + cb) = f32x4{v[0], v[1], v[2], v[3]};
                    } else {
                        bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                        *(bf16x4*)((bf16*)d.out + cc_out_row(d, m) * d.ldc + cb) = o;
                    }
                }
            }
        }
    }
  }   // epilogue chunks

    // ---- fused GroupNorm(32) statistics: thread -> lanes sharing the channel granule -> LDS -> global atomics ----
    if (d.gn_stats) {
        constexpr int CPR = BMC / 8;              // (power of two: the launcher keeps gn_stats off the 320-channel shape)
        // [wave][32 groups][sum, sumsq]: a wave adds into its own slots and the slots are summed in a fixed order, the
        // workgroups meet in double-precision global atomics — reproducible from run to run (see norm.hip)
        float* const sS = (float*)smem;
        __syncthreads();                          // the last chunk's staging area has been consumed
        sS[tid] = 0.f;                            // NT = 64 * waves entries
        __syncthreads();
        const int cb = ch0 + (tid % CPR) * 8;
        const int cpg = d.N >> 5;                 // >= 8: eight aligned channels span at most two groups
        const int g0 = cb / cpg;
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool first = (cb + e) < (g0 + 1) * cpg;
            s0 += first ? gs[e] : 0.f;
            q0 += first ? gq[e] : 0.f;
            s1 += first ? 0.f : gs[e];
            q1 += first ? 0.f : gq[e];
        }
#pragma unroll
        for (int off = CPR; off < 64; off <<= 1) {
            s0 += __shfl_xor(s0, off);
            q0 += __shfl_xor(q0, off);
            s1 += __shfl_xor(s1, off);
            q1 += __shfl_xor(q1, off);
        }
        if ((tid & 63) < CPR && cb < d.N) {
            float* const mine = sS + wave * 64;
            atomicAdd(&mine[g0 * 2], s0);
            atomicAdd(&mine[g0 * 2 + 1], q0);
            if (g0 < 31) {
                atomicAdd(&mine[g0 * 2 + 2], s1);
                atomicAdd(&mine[g0 * 2 + 3], q1);
            }
        }
        __syncthreads();
        if (tid < 64) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < WM * WN; ++w) v += (double)sS[w * 64 + tid];
            if (v != 0.0) unsafeAtomicAdd(d.gn_stats + (size_t)stat_frame * 64 + tid, v);
        }
    }
}
