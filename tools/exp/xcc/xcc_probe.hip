// Which XCD does workgroup b of a 1-D grid run on?  Prints XCC_ID per block index (HW_REG_XCC_ID, gfx942+).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    int cu;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(cu));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = cu; }
    __builtin_amdgcn_s_sleep(127);
}
int main() {
    int n = 256, *d, h[512];
    hipMalloc(&d, sizeof(h));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(512), 160 * 1024, 0, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int b = 0; b < n; ++b) bad += ((h[2 * b] & 15) != (b & 7));
        printf("launch %d: blocks whose XCC_ID != b %% 8: %d of %d; first 16 XCC ids:", rep, bad, n);
        for (int b = 0; b < 16; ++b) printf(" %d", h[2 * b] & 15);
        printf("\n");
    }
    return 0;
}
