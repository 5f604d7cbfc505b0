// Throughput per SIMD (ns per wave instruction at 2 waves per SIMD) of the VALU ops a softmax could be built from.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
template <int OP>
__global__ void k(float* out, int iters, float seed) {
    float a[8]; f2 p[8]; _Float16 h[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1}; h[i] = (_Float16)(a[i] * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
                else if (OP == 1) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
                else if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
                else if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
                else if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                else if (OP == 5) asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(a[i]));
                else if (OP == 6) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
                else if (OP == 7) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(a[i]));
                else if (OP == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(a[i]));
                else if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 1, %0" : "+v"(a[i]));
                else if (OP == 10) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                else if (OP == 11) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
            }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)h[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(const char* name, float* d) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k<OP>, dim3(256), dim3(256 * wps), 0, 0, d, iters, 1.0f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-20s waves/SIMD %d: %.2f ns per wave-instruction per SIMD\n", name, wps, ms * 1e6 / ((double)iters * 64 * wps));
    }
}
int main() {
    float* d; if (hipMalloc(&d, (1 << 22) * 4) != hipSuccess) return 1;
    run<0>("v_exp_f32", d); run<1>("v_exp_f16", d); run<10>("v_rcp_f32", d); run<4>("v_fma_f32", d); run<2>("v_pk_fma_f32", d); run<3>("v_pk_mul_f32", d);
    run<11>("v_pk_add_f32", d); run<5>("v_ldexp_f32", d); run<6>("v_floor_f32", d); run<7>("v_max3_f32", d); run<8>("v_cvt_pk_bf16_f32", d); run<9>("v_lshl_add_u32", d);
    return 0;
}
