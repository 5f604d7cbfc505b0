// Issue-rate microbenchmark for one SIMD of gfx950: v_exp_f32, v_pk_fma_f32, v_mfma_f32_32x32x16_bf16 alone and interleaved in ONE wave,
// with 1 / 2 / 4 waves per SIMD.  Answers: is the transcendental pipe shared with the VALU issue, and does a wave's own MFMA run under
// its VALU work?   hipcc --offload-arch=gfx950 -O3 pipes.hip -o pipes && ./pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NEXP, int NFMA, int NMFMA>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    float e[8];
    f32x2 p[8];
    f32x16 acc[2];
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = seed * (threadIdx.x + i) * 1e-6f; p[i] = f32x2{seed + i, seed - i}; a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    const f32x2 c1 = {1.0001f, 0.9999f}, c2 = {1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NEXP; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i & 7]));
#pragma unroll
            for (int i = 0; i < NFMA; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(c1), "v"(c2));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += e[i] + p[i][0] + p[i][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NEXP, int NFMA, int NMFMA>
void run(const char* name, float* out, int waves_per_simd) {
    const int iters = 20000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NEXP, NFMA, NMFMA>), dim3(256), dim3(threads), 0, 0, out, 100, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NEXP, NFMA, NMFMA>), dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_round = ms * 1e6 / (iters * 4.0);          // one round = NMFMA mfma + NEXP exp + NFMA pk_fma per wave
    printf("%-34s waves/SIMD %d: %7.1f ns per round per SIMD (%5.1f ns per wave-round)", name, waves_per_simd, ns_per_round, ns_per_round / waves_per_simd);
    const double per_wave = ns_per_round / waves_per_simd;
    if (NEXP && !NFMA && !NMFMA) printf("  -> %.2f ns per v_exp_f32 wave-op", per_wave / NEXP);
    if (!NEXP && NFMA && !NMFMA) printf("  -> %.2f ns per v_pk_fma_f32 wave-op", per_wave / NFMA);
    if (!NEXP && !NFMA && NMFMA) printf("  -> %.2f ns per mfma 32x32x16 (%.0f TF/s chip)", per_wave / NMFMA, 32768.0 * 1024 / (per_wave / NMFMA) / 1e3);
    printf("\n");
}

int main() {
    float* out; hipMalloc(&out, 4096 * 4);
    for (int w : {1, 2, 4}) {
        run<32, 0, 0>("32 exp", out, w);
        run<0, 32, 0>("32 pk_fma", out, w);
        run<0, 0, 8>("8 mfma", out, w);
        run<32, 32, 0>("32 exp + 32 pk_fma", out, w);
        run<32, 0, 8>("32 exp + 8 mfma", out, w);
        run<0, 32, 8>("32 pk_fma + 8 mfma", out, w);
        run<32, 32, 8>("32 exp + 32 pk_fma + 8 mfma", out, w);
    }
    return 0;
}
