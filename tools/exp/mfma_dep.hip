// v_mfma_f32_32x32x16_bf16 / 16x16x32: time per MFMA of one wave per SIMD with NACC accumulators used round-robin (dependent distance NACC).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.01f); }
    float s = 0;
    if (SHAPE == 32) {
        f32x16 acc[NACC];
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[q], 0, 0, 0);
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    } else {
        f32x4 acc[NACC];
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[q], 0, 0, 0);
        for (int q = 0; q < NACC; ++q) for (int r = 0; r < 4; ++r) s += acc[q][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int SHAPE> void run(float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, SHAPE>), dim3(256), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mfma %dx%d, %d accumulators round-robin: %.2f ns per MFMA\n", SHAPE, SHAPE, NACC, ms * 1e6 / ((double)iters * 24));
}
int main() {
    float* d; if (hipMalloc(&d, (1 << 22) * 4) != hipSuccess) return 1;
    run<1, 32>(d); run<2, 32>(d); run<3, 32>(d); run<4, 32>(d); run<6, 32>(d);
    run<1, 16>(d); run<2, 16>(d); run<3, 16>(d); run<4, 16>(d); run<6, 16>(d);
    return 0;
}
