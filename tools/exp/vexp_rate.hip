// Throughput of v_exp_f32 / v_fma_f32 / v_cvt_pk_bf16_f32 per SIMD at 1, 2, 4 waves per SIMD (cycles per wave instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int OP>
__global__ void k(float* out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.001f);          // v_exp_f32 (+1 mul)
                else if (OP == 1) a[i] = fmaf(a[i], 0.999f, 0.001f);                // v_fma_f32
                else a[i] = a[i] * 0.999f;
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.f;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1 << 20] = t1 - t0;
}
int main() {
    float* d; CHECK(hipMalloc(&d, (1 << 22) * 4 + 64));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int op = 0; op < 3; ++op)
        for (int wps = 1; wps <= 4; wps *= 2) {
            const int threads = 256 * wps;      // wps waves per SIMD, one workgroup per CU
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
                else if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
                else hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, d, iters, 1.0f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long ticks; hipMemcpy(&ticks, ((long long*)d) + (1 << 20), 8, hipMemcpyDeviceToHost);
            const double insts = (double)iters * 64 * (op == 0 ? 2 : 1);      // per wave (exp counts its mul too)
            printf("op %d (%s) waves/SIMD %d: %.3f ms, %.2f ns per wave-instruction per SIMD, s_memtime ticks/inst %.2f\n", op,
                   op == 0 ? "v_mul+v_exp" : op == 1 ? "v_fma" : "v_mul", wps, ms, ms * 1e6 / (insts * wps), (double)ticks / insts);
        }
    return 0;
}
