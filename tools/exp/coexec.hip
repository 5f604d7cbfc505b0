// Do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap?  8-wave workgroups (2 waves per SIMD): waves 0-3 run
// role A, waves 4-7 role B.  Roles: 0 = idle, 1 = MFMA 32x32x16 stream, 2 = v_exp_f32 stream, 3 = v_fma_f32 stream, 4 = mixed
// (one wave alternating 1 MFMA : 8 exp).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void k(float* out, int iters, int roleA, int roleB) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    f32x16 acc[4];
    float a[8];
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 0.001f + i); y[i] = (__bf16)(i * 0.01f); a[i] = threadIdx.x * 1e-3f + i; }
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    if (role == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[q], 0, 0, 0);
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 32; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
    } else if (role == 4) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[q], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                }
    }
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    float* d; if (hipMalloc(&d, (1 << 22) * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 1000;
    const int cfg[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {1, 3}, {4, 0}, {4, 4}};
    for (auto& c : cfg) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, iters, c[0], c[1]);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("roles (%d, %d): %.3f ms   [per role stream: 32 MFMA or 256 VALU per iteration, %d iterations]\n", c[0], c[1], ms, iters);
    }
    return 0;
}
