// How fast can a workgroup feed v_mfma_f32_32x32x16_bf16 from LDS alone?  No global traffic in the loop: the A tile (BM rows x 64 B)
// and the B tile (BN rows x 64 B) sit in LDS, every k-step a wave reads its TI + TJ fragments (ds_read_b128, conflict-free
// swizzle) and issues TI x TJ MFMAs, with a workgroup barrier every `steps_per_barrier` k-steps (as the operand rings need).
// Prints TF/s for the whole chip: block shapes as in ccedit_gemm / conv_halo (2x2 per wave), t6 (5x2), t4 (4x2), and 2x4 / 4x4.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int WM, int WN, int TI, int TJ, int WPS>
__global__ __launch_bounds__(WM* WN * 64, WPS * 4 / (WM * WN) > 0 ? WPS * 4 / (WM * WN) : 1) void k(float* out, int iters, int spb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = WM * TI * 32, BN = WN * TJ * 32;
    char* sA = smem;                 // [2][BM][64 B]
    char* sB = smem + 2 * BM * 64;   // [2][BN][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (2 * BM + 2 * BN) * 4; i += WM * WN * 64) ((float4*)smem)[i] = float4{1.f + (i & 7) * 1e-3f, 0.5f, 0.25f, 2.f};
    __syncthreads();
    const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 2) & 3;
    const char* fa = sA + (wm * TI * 32 + l31) * 64;
    const char* fb = sB + (wn * TJ * 32 + l31) * 64;
    f32x16 acc[TI][TJ];
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int buf = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[TI], bf[TJ];
            const int off = ((ks * 2 + hi) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const bf16x8*)(fa + buf * BM * 64 + i * 32 * 64 + off);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = *(const bf16x8*)(fb + buf * BN * 64 + j * 32 * 64 + off);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (spb > 0 && (it % spb) == spb - 1) {
            __syncthreads();
            buf ^= 1;
        }
    }
    float s = 0;
    for (int i = 0; i < TI; ++i) for (int j = 0; j < TJ; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int WM, int WN, int TI, int TJ, int WPS>
void run(float* d, const char* name, int spb) {
    constexpr int BM = WM * TI * 32, BN = WN * TJ * 32, NT = WM * WN * 64;
    const int lds = (2 * BM + 2 * BN) * 64;
    const int wgs = 256 * (WPS * 4 / (WM * WN));          // WPS waves per SIMD
    (void)hipFuncSetAttribute((const void*)k<WM, WN, TI, TJ, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<WM, WN, TI, TJ, WPS>), dim3(wgs), dim3(NT), lds, 0, d, iters, spb);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 2.0 * BM * BN * 32.0 * iters * wgs;
    printf("%-34s waves/SIMD %d, barrier every %d x 32 k: %7.1f TF/s   (%.2f LDS KB per MFMA)\n", name, WPS, spb, flops / ms / 1e9,
           (TI + TJ) / (double)(TI * TJ));
}

int main() {
    float* d; if (hipMalloc(&d, (1 << 24) * 4) != hipSuccess) return 1;
    for (int spb : {0, 2, 1}) {
        run<2, 2, 2, 2, 2>(d, "128x128, 64x64 per wave (t1, conv)", spb);
        run<2, 2, 2, 2, 1>(d, "128x128, 64x64 per wave", spb);
        run<2, 2, 5, 2, 2>(d, "320x128, 160x64 per wave (t6)", spb);
        run<2, 4, 4, 2, 2>(d, "256x256, 8 waves 128x64 (t4)", spb);
        run<2, 2, 2, 4, 2>(d, "128x256, 64x128 per wave", spb);
        run<2, 2, 4, 4, 1>(d, "256x256, 128x128 per wave", spb);
    }
    return 0;
}
