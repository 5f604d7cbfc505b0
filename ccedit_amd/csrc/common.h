// Shared device helpers for the gfx950 kernels of libccedit_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ccedit_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// 16-byte asynchronous global -> LDS copy (global_load_lds_dwordx4).  The LDS destination is the
// wave-uniform `lds_wave_base` + lane*16; the global source address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// SiLU, v * sigmoid(v) (nn.SiLU of openaimodel.py:441-444 etc.).  The reciprocal is v_rcp_f32 (1 ulp), not an IEEE division: hipcc
// expands `v / (1 + exp(-v))` into the ten-instruction v_div_scale / v_div_fmas / v_div_fixup sequence (2984 of them in norm.o), which
// made the GroupNorm + SiLU passes VALU-bound (temporal GroupNorm at 34 x 64 x 96 x 320: 70 us with SiLU, 56 us without, 45 us for a
// copy of the same bytes).  The result is rounded to bf16 (8 bits) right after.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// GELU, 0.5 v (1 + erf(v / sqrt 2)) = v Phi(v) (GEGLU, attention.py:115-126: F.gelu, the exact-erf form), as v * sigmoid(u(v)) with
// u = v (a + b v^2 + c v^4) fitted to Phi (minimax over |v| <= 9; the argument is clamped there — beyond, Phi is 0 or 1 to 1e-11 and
// the quartic would turn around at |v| = 11).  |error| <= 2.6e-5 ABSOLUTE over all v (tools/exp/gelu_fit.py: fp32 evaluation against
// fp64 erf) — 300 times below the bf16 rounding of the result at |v| ~ 3, where the maximum sits, and the result is rounded to bf16
// right after.  Six plain VALU instructions + v_exp_f32 + v_rcp_f32; rounds 1-5 used Abramowitz & Stegun 7.1.28
// (1 - (1 + a1 x + ... + a6 x^6)^-16: 3e-7, sixteen VALU instructions) — the GELU is the VALU share of ff320 and of the GEGLU
// epilogues (13-16 % of a GEGLU GEMM at the 64x96 and 32x48 levels with the cheaper of the two before), libdevice's erff twice that.
// The coefficients carry -log2(e): v_exp_f32 is 2^x.
#ifdef CCEDIT_GELU_AS71_28          // probe builds only (tools/exp/build_gelu_variant.sh): the formula of rounds 1-5, for the same-box A/B
__device__ __forceinline__ float gelu_erf_f(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    float p = fmaf(x, 0.0000430638f, 0.0002765672f);
    p = fmaf(x, p, 0.0001520143f);
    p = fmaf(x, p, 0.0092705272f);
    p = fmaf(x, p, 0.0422820123f);
    p = fmaf(x, p, 0.0705230784f);
    p = fmaf(x, p, 1.0f);
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;
    const float e = 1.0f - __builtin_amdgcn_rcpf(p);          // erf(|v| / sqrt 2)
    return 0.5f * v + 0.5f * fabsf(v) * e;                     // 0.5 v (1 + sign(v) e)
}
#else
__device__ __forceinline__ float gelu_erf_f(float v) {
    const float vc = __builtin_amdgcn_fmed3f(v, -9.0f, 9.0f);
    const float t = vc * vc;
    float p = fmaf(t, 0.00101426306f, -0.106775724f);       // -log2(e) * (c, b, a) = -log2(e) * (-7.03033577e-4, 7.40112921e-2, 1.59501577)
    p = fmaf(t, p, -2.30112134f);
    const float e = __builtin_amdgcn_exp2f(p * vc);          // exp(-u)
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The library's dispatch policy: ONE table of named integer switches (core.cpp), all defaulting to the fast path.  Nothing in the
// library reads the environment: the host sets entries through ccedit_policy_set (ccedit_amd/policy.py reads CCEDIT_POLICY once and
// pushes it).  Every switch selects between kernels that compute the same fp32 sums in a different order — A/B and test arms
// ("specialised kernels reproduce the generic ones", tests/test_fullsize_gpu.py), never a change of arithmetic.
struct CcPolicy {
    int conv_halo = 1;      // 3x3 stride-1 convs on the LDS-halo kernel with the four-slot weight ring (2: the two-slot kernel of rounds 2-5; 0: tap-gather)
    int g8 = 1;             // persistent eight-phase GEMM for long Linears (0: the tap_gemm block shapes)
    int g8_conv = 1;        // ... its tap-gather mode for 3x3 convs onto >= 1024 channels
    int g8_temporal = 1;    // ... and for Conv1d k3 over T at >= 640 channels
    int g8_split = -1;      // split-K at the 8x12 level: -1 auto, 0 off, n fixed
    int lin320 = 1;         // register-resident-weight K = 320 Linears (0: tap_gemm)
    int lin320s = 1;        // ... the streaming deep-ring variant (0: the K-split kernel)
    int lin640 = 1;         // streaming K = 640 Linears (0: g8)
    int temp320 = 1;        // streaming Conv1d k3 at 320 channels (0: tap_gemm)
    int attn_short = 1;     // temporal attention kernel (0: the general flash kernel)
    int attn_text = 1;      // text cross-attention kernel
    int attn_spatial = 1;   // long self-attention kernel with the reference in the MFMA: 1 = d 40 and d 80, 2 = d 40 only, 0 = off
    int attn_pv16 = 1;      // ... its PV product in 16x16x32 tiles (0: 32x32x16)
    int attn_opt = 1;       // ... softmax reference fixed by the first key tile, exact re-run of a workgroup that overflowed (0: tracked on every tile)
    int gn_flat = 1;        // flat thread mapping of the temporal GroupNorm at the two large levels
    int gn_apply_flat = 1;  // column-per-thread, four-rows-in-flight mapping of the spatial GroupNorm apply pass (0: a wave per pixel row)
    int f32_split = 1;      // fp32 first-stage contractions as six exact bf16 products per fp32 product on the bf16 matrix pipe (0: v_mfma_f32_32x32x2_f32)
};
const CcPolicy& cc_policy();

void cc_set_error(const char* fmt, ...);
void cc_note_kernel(const char* fmt, ...);      // which kernel template the entry point dispatched to (ccedit_last_kernel)

#define CC_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            cc_set_error(__VA_ARGS__);   \
            return CCEDIT_EINVAL;        \
        }                                \
    } while (0)

#define CC_UNSUPPORTED(cond, ...)        \
    do {                                 \
        if (cond) {                      \
            cc_set_error(__VA_ARGS__);   \
            return CCEDIT_EUNSUPPORTED;  \
        }                                \
    } while (0)

// Per-device once-guard for hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute is a property of the
// (function, device) pair, so a process that drives a second GPU must set it there too.  `done` is a per-call-site
// bitmask over device ordinals (devices >= 64 simply set it every time).
static inline int cc_max_dynamic_lds(const void* fn, int bytes, unsigned long long* done, const char* what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess && dev < 64 && ((*done >> dev) & 1ull)) return CCEDIT_OK;
    if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        cc_set_error("hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
        return (int)e;
    }
    if (dev < 64) *done |= 1ull << dev;
    return CCEDIT_OK;
}

static inline int cc_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        cc_set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return CCEDIT_OK;
}
