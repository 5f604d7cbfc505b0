// lds_mfma.hip plus a DMA stream: per 64-k tile every wave also issues NDMA global_load_lds_dwordx4 (1 KB each) from an
// L2-resident buffer into a spare LDS area and the workgroup waits for them (vmcnt(0) + barrier) at the end of the tile, exactly
// like a two-slot operand ring — but the loaded bytes are never read, so the MFMA loop's own data do not change.
// Separates "fragment reads + MFMAs" from "the same with operand delivery running beside it".
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int NDMA, int WAITMODE, int COLD, int BURST>
__global__ __launch_bounds__(256) void k(float* out, const char* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN = 128;
    char* sA = smem;                 // [BM][128 B]
    char* sB = smem + BM * 128;      // [BN][128 B]
    char* sD = smem + (BM + BN) * 128;       // DMA landing area: 2 x 4 waves x (NDMA + 6) KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < (BM + BN) * 8; i += 256) ((float4*)smem)[i] = float4{1.f + (i & 7) * 1e-3f, 0.5f, 0.25f, 2.f};
    __syncthreads();
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5, sw = (l31 >> 1) & 7;
    const char* fa = sA + (wm * 64 + l31) * 128;
    const char* fb = sB + (wn * 64 + l31) * 128;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* gsrc = src + ((size_t)(blockIdx.x & 63) * 4 + wave) * 65536 + lane * 16;
    const char* gcold = src + (size_t)64 * 4 * 65536 + ((size_t)blockIdx.x * 4 + wave) * (size_t)(COLD ? 512 * 1024 : 0) + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NDMA; ++n)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gsrc + ((it * NDMA + n) & 63) * 1024),
                                             (LDS_AS void*)(sD + wave * 1024), 16, 0, 0);
        if (BURST && it % 9 == 0) {          // the next chunk's halo: 6 KB per wave, from memory nobody has touched
#pragma unroll
            for (int n = 0; n < 6; ++n)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(gcold + (size_t)(((it / 9) * 6 + n) & 511) * 1024),
                                                 (LDS_AS void*)(sD + wave * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bf[2];
            const int off = ((ks * 2 + hi) ^ sw) << 4;
            for (int i = 0; i < 2; ++i) af[i] = *(const bf16x8*)(fa + i * 32 * 128 + off);
            for (int j = 0; j < 2; ++j) bf[j] = *(const bf16x8*)(fb + j * 32 * 128 + off);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (WAITMODE == 2) __syncthreads();
        else {
            if (WAITMODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s + sD[tid];
}

template <int NDMA, int WAITMODE, int COLD = 0, int BURST = 0>
void run(float* d, const char* src) {
    const int lds = 72 * 1024;                 // as conv_halo_kernel: two workgroups per CU (every DMA lands in the wave's one spare KB)
    (void)hipFuncSetAttribute((const void*)k<NDMA, WAITMODE, COLD, BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, wgs = 512;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NDMA, WAITMODE, COLD, BURST>), dim3(wgs), dim3(256), lds, 0, d, src, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = 2.0 * 128 * 128 * 64.0 * iters * wgs;
    printf("128x128 tile, %d KB of DMA per wave and 64-k tile (%4.1f B/clk/CU at this rate), %s: %7.1f TF/s, %.3f us per tile\n", NDMA,
           2.0 * 4 * NDMA * 1024 / (ms * 1e-3 / iters * 2.4e9), WAITMODE == 0 ? (BURST ? (COLD ? "vmcnt(0)+barrier, +6 KB COLD burst every 9th tile" : "vmcnt(0)+barrier, +6 KB hot burst every 9th tile ") : "vmcnt(0) + barrier per tile") : WAITMODE == 2 ? (BURST ? "__syncthreads, +6 KB COLD burst every 9th tile  " : "__syncthreads per tile") : "never waited for      ",
           flops / ms / 1e9, ms * 1e3 / iters);
}

int main() {
    float* d; if (hipMalloc(&d, (1 << 22) * 4) != hipSuccess) return 1;
    const size_t bytes = (size_t)64 * 4 * 65536 + (size_t)512 * 4 * 512 * 1024 + 65536;      // hot part + 1 GB streamed part
    char* src; if (hipMalloc(&src, bytes) != hipSuccess) return 1;
    (void)hipMemset(src, 0, bytes);
    run<0, 0>(d, src); run<4, 0>(d, src); run<5, 0>(d, src); run<8, 0>(d, src);
    run<4, 2>(d, src);
    run<4, 0, 0, 1>(d, src); run<4, 0, 1, 1>(d, src); run<4, 2, 1, 1>(d, src);
    return 0;
}
