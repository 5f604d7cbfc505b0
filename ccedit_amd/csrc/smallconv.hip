// 3x3 convolution for the few-channel, many-pixel layers at the top of the ControlNet hint stem
// (controlmodel.py:215-231: 3->16, 16->16 at 512x768, 16->32 stride 2, 32->32 at 256x384; 34 frames): Cin <= 32 and
// Cout <= 32 on up to 13.4 M pixels.  These are HBM-bound (61 GFLOP against 0.9 GB for the 16->16 layer); the tiled
// LDS-staged GEMM needs >= 64 output channels per block and spent 1.1-1.2 ms per layer on them.
//
// No LDS: one wave owns 32 consecutive output pixels and all output channels.
//   v_mfma_f32_32x32x16_bf16, A = weights [32 channel rows][K] (rows >= Cout are zero in the packed operand),
//   B = pixels [32][K], K = [tap][Cin]: one MFMA k-step covers 16 K elements = (16/Cin) taps of Cin channels.
//   B fragment of lane (pixel p = lane & 31, half = lane >> 5): the 8 channels [8*half', ...) of tap t at pixel p,
//   a 16-byte load straight from the channels-last source — consecutive lanes read consecutive pixels, so one wave
//   load is a contiguous (stride 1) run of 32 * Cin * 2 bytes; the 9 shifted reads re-hit L1/L2.
//   The weight fragments of all k-steps stay in registers while the wave walks over its pixel groups.
// Epilogue in registers: bias + SiLU, 8-byte stores of 4 consecutive channels.
#include "common.h"

namespace {

template <int CIN>      // channels per tap as stored (8, 16 or 32)
__global__ __launch_bounds__(256) void small_conv3x3_kernel(const CcGemmDesc d, int groups_per_wave) {
    constexpr int KSTEPS = (9 * CIN + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const bf16* __restrict__ W = (const bf16*)d.W;
    const bf16* __restrict__ X = (const bf16*)d.A;
    const bf16* zp = nullptr;

    bf16x8 wf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) wf[ks] = *(const bf16x8*)(W + (size_t)l31 * d.Kpad + ks * 16 + hi * 8);
    // bias of this lane's channels q*8 + hi*4 + e
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = q * 8 + hi * 4;
        bq[q] = (d.bias && c < d.N) ? *(const f32x4*)(d.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int hwout = d.Hout * d.Wout;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave;
    for (int gi = 0; gi < groups_per_wave; ++gi) {
        const int64_t m0 = (wave_id * groups_per_wave + gi) * 32;
        if (m0 >= d.M) break;
        const int64_t m = m0 + l31;
        const bool ok = m < d.M;
        const int n = (int)(m / hwout);
        const int rem = (int)(m - (int64_t)n * hwout);
        const int oy = rem / d.Wout, ox = rem - oy * d.Wout;
        const int iy0 = oy * d.stride - d.pad, ix0 = ox * d.stride - d.pad;
        const bf16* xn = X + (size_t)n * d.Hin * d.Win * d.lda;

        bf16x8 xf[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            // K elements [ks*16 + hi*8, +8): tap and channel offset inside the tap
            const int kk = ks * 16 + hi * 8;
            const int tap = kk / CIN, c0 = kk - tap * CIN;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int iy = iy0 + dy, ix = ix0 + dx;
            const bool v = ok && tap < 9 && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            bf16x8 t = {0, 0, 0, 0, 0, 0, 0, 0};
            if (v) t = *(const bf16x8*)(xn + ((size_t)iy * d.Win + ix) * d.lda + c0);
            xf[ks] = t;
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], xf[ks], acc, 0, 0, 0);
        if (ok) {
            bf16* orow = (bf16*)d.out + (size_t)m * d.ldc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 8 + hi * 4;
                if (c < d.N) {
                    float v0 = acc[q * 4 + 0] + bq[q][0], v1 = acc[q * 4 + 1] + bq[q][1];
                    float v2 = acc[q * 4 + 2] + bq[q][2], v3 = acc[q * 4 + 3] + bq[q][3];
                    if (d.act == CCEDIT_ACT_SILU) {
                        v0 = silu_f(v0);
                        v1 = silu_f(v1);
                        v2 = silu_f(v2);
                        v3 = silu_f(v3);
                    }
                    *(bf16x4*)(orow + c) = bf16x4{f2bf(v0), f2bf(v1), f2bf(v2), f2bf(v3)};
                }
            }
        }
    }
    (void)zp;
}

}  // namespace

// Called from ccedit_gemm for CONV2D descriptors that qualify (see cc_small_conv_applicable).
bool cc_small_conv_applicable(const CcGemmDesc& d) {
    return d.mode == CCEDIT_GEMM_CONV2D && d.taps == 9 && d.ksize == 3 && !d.upsample && !d.A2 && d.korder == 0 &&
           (d.Cin == 8 || d.Cin == 16 || d.Cin == 32) && d.Cin1 == d.Cin && d.N <= 32 && d.N % 4 == 0 && !d.res1 && !d.res2 &&
           !d.group_bias && !d.out_f32 && !d.gn_stats && (d.act == CCEDIT_ACT_NONE || d.act == CCEDIT_ACT_SILU) &&
           d.M >= (1 << 16) && d.lda % 8 == 0 && d.ldc % 4 == 0;
}

int cc_small_conv_launch(const CcGemmDesc& d, hipStream_t s) {
    const int64_t groups = (d.M + 31) / 32;
    int gpw = 8;                                         // pixel groups per wave: amortises the weight-fragment loads
    while (gpw > 1 && groups / (4 * gpw) < 4096) gpw >>= 1;
    const int64_t blocks = (groups + 4 * gpw - 1) / (4 * gpw);
    dim3 grid((unsigned)blocks);
    cc_note_kernel("small_conv3x3_kernel");
    if (d.Cin == 8) hipLaunchKernelGGL(small_conv3x3_kernel<8>, grid, dim3(256), 0, s, d, gpw);
    else if (d.Cin == 16) hipLaunchKernelGGL(small_conv3x3_kernel<16>, grid, dim3(256), 0, s, d, gpw);
    else hipLaunchKernelGGL(small_conv3x3_kernel<32>, grid, dim3(256), 0, s, d, gpw);
    return cc_launch_status("small_conv3x3_kernel");
}
