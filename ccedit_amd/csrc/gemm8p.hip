// Persistent 256ch x 256pix bf16 GEMM for the plain Linear layers of the path (tile 11): eight waves, K tiles of 64,
// a two-buffer LDS ring of sixteen-KB half tiles filled by global_load_lds, and an eight-phase K loop in which the two
// halves of the workgroup run one barrier apart — while waves 0-3 issue the MFMAs of a phase, waves 4-7 read the
// fragments of theirs and request the next half tile, then the roles swap.
//
//   out[m][n] = epilogue( sum_k W[n][k] * A[m][k] )        (Linear / Conv 1x1 / Conv1d k1; attention.py:115-141, 377-383)
//
// Why another kernel next to tap_gemm_kernel: the FF / GEGLU projections of the 32x48 and 16x24 levels and the C -> C
// Linears are plain long GEMMs (27 ms of a 120 ms step); on those the two-barrier 128x128 loop of tap_gemm tops out at
// 650-900 TF/s while a library GEMM reaches 1030-1250 on the same box.  What this kernel changes:
//   * 256 x 256 tile: 128 FLOP per byte staged (64 for 128 x 128), so the ~20 B/clk a CU can pull through L1 misses
//     stops being the ceiling;
//   * operand half tiles stay in flight ACROSS barriers: the only vmcnt wait of a K tile is `vmcnt(6)` in its last
//     phase (three half tiles = 48 KB still travelling); nothing in the loop drains the queue;
//   * every half tile is read in exactly ONE phase by all eight waves (a wave's 128 x 64 output is two 64-row bands,
//     one from each A half, times two 32-pixel bands, one from each B half), so a half-tile slot is free one or two
//     phases after it was read and is re-filled four to six phases before it is read again;
//   * the wave halves are offset by one barrier (ping-pong): LDS fragment reads and DMA issue of one half run under the
//     MFMAs of the other, `s_setprio` keeps the matrix pipe with the half that is in its MFMA block;
//   * persistent workgroups (one per CU): the first seven half tiles of the NEXT output tile are requested before the
//     epilogue of the current one, so the cold-operand latency (~1.5 us in the network) hides under the stores;
//   * the epilogue goes from the accumulators straight to global memory: v_permlane32_swap turns the 32x32 MFMA layout
//     (a lane owns 4 channels of a pixel) into 8 consecutive channels per lane = 16-byte stores, bias / SiLU / GEGLU /
//     residuals applied in registers — no LDS staging, no barriers, so the prefetch above may use the whole ring.
//
// LDS: [buffer 0 | buffer 1] x [A half 0 | A half 1 | B half 0 | B half 1], a half = 128 rows x 128 B (64 k).  Rows are
// written lane-linearly by the DMA; 16-byte granule g of row r sits at slot g ^ ((r >> 1) & 7) (source-side swizzle,
// same involution on the fragment read: conflict-free for the 32x32x16 fragment pattern, see gemm.hip).
//
// Hazards, in phase numbers j (K tile t = j / 4; a phase = fragment reads + one half-tile request, barrier, 8 MFMAs,
// barrier; the second wave half runs every step one barrier later):
//   reads   phase 4t: B half 0 then A half 0;  4t+1: B half 1;  4t+2: A half 1;  4t+3: none (B half 0 is still in registers)
//   refills phase 4t: A1 of tile t+1;  4t+1: B0 of t+2;  4t+2: A0 of t+2;  4t+3: B1 of t+2, then vmcnt(6)
//   RAW: tile t+1 is complete when every wave has passed the vmcnt(6) of phase 4t+3 (it leaves only the three requests
//        of phases 4t+1..4t+3 in flight) and a barrier; its first read is in phase 4t+4.
//   WAR: B0 is re-requested one phase after its reads, which `lgkmcnt(8)` retires BEFORE the reading phase's first
//        barrier (they are issued first); every other slot is re-requested two phases after its reads, whose results the
//        MFMAs of the reading phase consumed before that phase's second barrier.
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int G8_HALF = 128 * 128;          // bytes of a half tile
constexpr int G8_BUF = 4 * G8_HALF;         // A0 A1 B0 B1
constexpr int G8_LDS = 2 * G8_BUF;          // 128 KB

template <int N>
__device__ __forceinline__ void g8_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void g8_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void g8_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// Exchange between the two lane halves: afterwards lanes 0-31 hold (x, y) = (own x, partner's x), lanes 32-63
// (partner's y, own y) — with x / y the 4-channel groups q / q + 1 of a 32x32 accumulator tile that makes 8 consecutive
// channels per lane: 8 q + (0..7) in the low half, 8 (q + 1) + (0..7) in the high half.
__device__ __forceinline__ void g8_swap(float& x, float& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
}

struct G8Order {          // 32-bit on purpose: the tile walk runs once per output tile in every wave (M < 2^31 rows)
    int pt_n, pt_per_xcd, nlocal;
    int ct_n, Q, balanced;
};

__device__ __forceinline__ G8Order g8_order(const CcGemmDesc& d) {
    G8Order o;
    o.ct_n = (d.N + 255) >> 8;
    o.pt_n = (int)((d.M + 255) >> 8);
    o.pt_per_xcd = (o.pt_n + 7) / 8;
    o.Q = (d.cgroup & 0xFFFF) > 0 ? (d.cgroup & 0xFFFF) : o.ct_n;
    o.balanced = (d.cgroup >> 18) & 1;
    o.nlocal = o.balanced ? (o.pt_n * o.ct_n + 7) / 8 : o.pt_per_xcd * o.ct_n;
    return o;
}

// Tile `local` of XCD `xcd` (same walk as tap_gemm_kernel: an XCD owns a contiguous range of pixel tiles; channel tiles in
// groups of Q, inside a group channel-minor — so the workgroups running together on an XCD cover (32 / Q pixel tiles) x
// (Q channel tiles) and an over-L2 weight matrix is shared by 32 / Q of them).
__device__ __forceinline__ bool g8_decode(const G8Order& o, int local, int xcd, int& pt, int& ct) {
    if (o.balanced) {
        const int w = xcd * o.nlocal + local;
        if (w >= o.pt_n * o.ct_n) return false;
        pt = w / o.ct_n;
        ct = w - pt * o.ct_n;
        return true;
    }
    const int gsz = o.pt_per_xcd * o.Q;
    const int cg = local / gsz;
    const int rr = local - cg * gsz;
    const int qn = min(o.Q, o.ct_n - cg * o.Q);
    const int pl = rr / qn;
    pt = xcd * o.pt_per_xcd + pl;
    ct = cg * o.Q + (rr - pl * qn);
    return pt < o.pt_n;
}

__global__ __launch_bounds__(512) void g8_kernel(const CcGemmDesc d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l31 = lane & 31, hi = lane >> 5;
    const int nk = d.Kpad >> 6;

    const G8Order ord = g8_order(d);
    const int xcd = blockIdx.x & 7;
    const int nw = gridDim.x >> 3;              // workgroups per XCD
    int local = blockIdx.x >> 3;

    // ---- staging: thread -> (row rsub of a 64-row issue, LDS slot p), source granule p ^ ((rsub >> 1) & 7) ----
    const int p = tid & 7, rsub = tid >> 3;
    const int gcol = p ^ ((rsub >> 1) & 7);
    const uint32_t a_lane = ((uint32_t)rsub * (uint32_t)d.Kpad + gcol * 8) * 2;          // byte offset inside a 64-row issue of W
    const char* const Wp = (const char*)d.W;
    const char* const Ap = (const char*)d.A;
    char* const lds_wave = smem + wave * 1024;

    // ---- fragment read addresses: row l31 of a 32-row MFMA tile, granule (2 ks + hi) ^ ((l31 >> 1) & 7) ----
    const int sw = (l31 >> 1) & 7;
    const char* fa[4];
    const char* fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int off = l31 * 128 + (((2 * ks + hi) ^ sw) << 4);
        fa[ks] = smem + wr * 8192 + off;                       // A half h at + h * G8_HALF, row tile ti' at + ti' * 4096
        fb[ks] = smem + 2 * G8_HALF + wc * 4096 + off;         // B half h at + h * G8_HALF
    }

    int pt, ct;
    // first tile of this workgroup
    for (;; local += nw) {
        if (local >= ord.nlocal) return;
        if (g8_decode(ord, local, xcd, pt, ct)) break;
    }

    // per-tile source bases (uniform) and the pixel-row offsets of the four 64-row issues of the B tile (clamped at M - 1:
    // rows past the end are computed from valid memory and never stored)
    const char* wt;
    const char* at;
    uint32_t b_lane[4];
    auto set_tile = [&](int pt_, int ct_) {
        wt = Wp + (size_t)ct_ * 256 * d.Kpad * 2;
        const int64_t pix0 = (int64_t)pt_ * 256;
        at = Ap + (size_t)pix0 * d.lda * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t r = pix0 + i * 64 + rsub;
            r = r < d.M ? r : d.M - 1;
            b_lane[i] = (uint32_t)((r - pix0) * d.lda + gcol * 8) * 2;
        }
    };
    // one half tile = two 64-row issues
    auto stage_a = [&](int h, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // uniform part in SGPRs (readfirstlane keeps hipcc from turning it into eight per-lane 64-bit induction variables),
            // per-lane part a 32-bit VGPR offset: global_load_lds_dwordx4 v, s[..]
            const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)((h * 128 + i * 64) * d.Kpad * 2 + kt * 128));
            glds16(wt + u + a_lane, lds_wave + buf * G8_BUF + h * G8_HALF + i * 8192);
        }
    };
    auto stage_b = [&](int h, int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t u = __builtin_amdgcn_readfirstlane((uint32_t)(kt * 128));
            glds16(at + u + b_lane[h * 2 + i], lds_wave + buf * G8_BUF + (2 + h) * G8_HALF + i * 8192);
        }
    };
    auto prologue = [&]() {
        stage_a(0, 0, 0);
        stage_a(1, 0, 0);
        stage_b(0, 0, 0);
        stage_b(1, 0, 0);
        stage_b(0, 1, 1);
        stage_a(0, 1, 1);
        stage_b(1, 1, 1);
    };

    set_tile(pt, ct);
    prologue();
    g8_vmcnt<6>();                               // K tile 0 has landed (this wave's part)

    for (;;) {
        const int64_t pix0 = (int64_t)pt * 256;
        const int ch0 = ct * 256;
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        g8_barrier();                            // ... and everybody else's
        if (wr == 1) g8_barrier();               // second wave half: one barrier behind from here on

        // ---- K loop: one K tile = four phases on LDS buffer CUR ----
        auto ktile = [&](auto CURC, int t) {
            constexpr int CUR = decltype(CURC)::value;
            constexpr int BASE = CUR * G8_BUF;
            bf16x8 a[2][4], b0[4], b1[4];
            // phase 0: B half 0 (first: retired by lgkmcnt(8) before the barrier), A half 0; request A1 of tile t + 1
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) b0[ks] = *(const bf16x8*)(fb[ks] + BASE);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a[ti][ks] = *(const bf16x8*)(fa[ks] + BASE + ti * 4096);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < nk) stage_a(1, t + 1, CUR ^ 1);
            g8_lgkmcnt<8>();
            g8_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    acc[ti][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][ks], b0[ks], acc[ti][0], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            g8_barrier();
            // phase 1: B half 1; request B0 of tile t + 2
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) b1[ks] = *(const bf16x8*)(fb[ks] + BASE + G8_HALF);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) stage_b(0, t + 2, CUR);
            g8_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    acc[ti][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][ks], b1[ks], acc[ti][1], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            g8_barrier();
            // phase 2: A half 1; request A0 of tile t + 2
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a[ti][ks] = *(const bf16x8*)(fa[ks] + BASE + G8_HALF + ti * 4096);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 2 < nk) stage_a(0, t + 2, CUR);
            g8_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    acc[2 + ti][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][ks], b1[ks], acc[2 + ti][1], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            g8_barrier();
            // phase 3: no reads; request B1 of tile t + 2; tile t + 1 must have landed before the next phase reads it
            if (t + 2 < nk) {
                stage_b(1, t + 2, CUR);
                g8_vmcnt<6>();
            } else {
                g8_vmcnt<0>();
            }
            g8_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
                    acc[2 + ti][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ti][ks], b0[ks], acc[2 + ti][0], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            g8_barrier();
        };
        int t = 0;
        for (; t + 1 < nk; t += 2) {
            ktile(std::integral_constant<int, 0>{}, t);
            ktile(std::integral_constant<int, 1>{}, t + 1);
        }
        if (t < nk) ktile(std::integral_constant<int, 0>{}, t);
        if (wr == 0) g8_barrier();               // both halves level again: every fragment read of this tile is done

        // ---- next tile: request its first seven half tiles now, they land under the epilogue ----
        int npt = 0, nct = 0;
        bool more = false;
        for (local += nw; local < ord.nlocal; local += nw)
            if (g8_decode(ord, local, xcd, npt, nct)) {
                more = true;
                break;
            }
        if (more) {
            set_tile(npt, nct);
            prologue();
        }

        // ---- epilogue: accumulators -> global ----
        const float* __restrict__ bias = d.bias;
        const bf16* __restrict__ r1 = (const bf16*)d.res1;
        const bf16* __restrict__ r2 = (const bf16*)d.res2;
        bf16* __restrict__ outp = (bf16*)d.out;
        if (d.act == CCEDIT_ACT_GEGLU) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int rowb = ch0 + (ti >> 1) * 128 + wr * 64 + (ti & 1) * 32;     // first packed row of this MFMA tile
                // packed rows rowb + 16 g + [0, 8) are values, + [8, 16) their gates; this lane: 4 hi + (0..3) of each
                f32x4 bx[2], bg[2];
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int rx = rowb + 16 * g + 4 * hi;
                    const bool ok = bias && rx < d.N;
                    bx[g] = ok ? *(const f32x4*)(bias + rx) : f32x4{0.f, 0.f, 0.f, 0.f};
                    bg[g] = ok ? *(const f32x4*)(bias + rx + 8) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    const int64_t m = pix0 + tj * 128 + wc * 32 + l31;
                    float o0[4], o1[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = (acc[ti][tj][e] + bx[0][e]) * gelu_erf_f(acc[ti][tj][4 + e] + bg[0][e]);
                        o1[e] = (acc[ti][tj][8 + e] + bx[1][e]) * gelu_erf_f(acc[ti][tj][12 + e] + bg[1][e]);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) g8_swap(o0[e], o1[e]);
                    const int oc = (rowb >> 1) + 8 * hi;                               // 8 consecutive output channels
                    if (m < d.M && rowb + 16 * hi < d.N) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = f2bf(o0[e]);
                            o[4 + e] = f2bf(o1[e]);
                        }
                        *(bf16x8*)(outp + (size_t)m * d.ldc + oc) = o;
                    }
                }
            }
        } else {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                const int rowb = ch0 + (ti >> 1) * 128 + wr * 64 + (ti & 1) * 32;
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const int cb = rowb + 16 * qp + 8 * hi;                            // this lane's 8 channels after the swap
                    const bool cok = cb < d.N;
                    f32x4 bv0 = {0.f, 0.f, 0.f, 0.f}, bv1 = {0.f, 0.f, 0.f, 0.f};
                    if (bias && cok) {
                        bv0 = *(const f32x4*)(bias + cb);
                        bv1 = *(const f32x4*)(bias + cb + 4);
                    }
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) {
                        const int64_t m = pix0 + tj * 128 + wc * 32 + l31;
                        const bool ok = cok && m < d.M;
                        bf16x8 rv1, rv2;
                        if (r1 && ok) rv1 = *(const bf16x8*)(r1 + (size_t)m * d.ldr1 + cb);
                        if (r2 && ok) rv2 = *(const bf16x8*)(r2 + (size_t)m * d.ldr2 + cb);
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc[ti][tj][8 * qp + e];
                            v[4 + e] = acc[ti][tj][8 * qp + 4 + e];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) g8_swap(v[e], v[4 + e]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] += bv0[e];
                            v[4 + e] += bv1[e];
                        }
                        if (d.act == CCEDIT_ACT_SILU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                        } else if (d.act == CCEDIT_ACT_QUICK_GELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
                        }
                        if (ok) {
                            if (r1) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += bf2f(rv1[e]);
                            }
                            if (r2) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += bf2f(rv2[e]);
                            }
                            bf16x8 o;
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
                            *(bf16x8*)(outp + (size_t)m * d.ldc + cb) = o;
                        }
                    }
                }
            }
        }
        if (!more) return;
        pt = npt;
        ct = nct;
        g8_vmcnt<0>();                           // the next tile's first K tiles (and this tile's stores) are through
    }
}

}  // namespace

bool cc_g8_applicable(const CcGemmDesc& d) {
    return d.mode == CCEDIT_GEMM_LINEAR && d.taps == 1 && d.A2 == nullptr && d.Cin % 64 == 0 && d.Kpad == d.Cin && d.Kpad >= 128 &&
           d.N % 8 == 0 && (d.act != CCEDIT_ACT_GEGLU || d.N % 16 == 0) && !d.out_f32 && !d.gn_stats && !d.group_bias &&
           d.ln_eps == 0.f && d.lda % 8 == 0 && d.ldc % 8 == 0 && (!d.res1 || d.ldr1 % 8 == 0) && (!d.res2 || d.ldr2 % 8 == 0) &&
           (int64_t)256 * d.lda * 2 < (1LL << 31);
}

int cc_g8_launch(const CcGemmDesc& d, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (int rc = cc_max_dynamic_lds((const void*)g8_kernel, G8_LDS, &attr_done, "g8_kernel")) return rc;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            cc_set_error("g8: cannot query the device");
            return CCEDIT_EINVAL;
        }
        n_cu = prop.multiProcessorCount;
    }
    const int64_t pt_n = (d.M + 255) / 256, ct_n = (d.N + 255) / 256;
    CcGemmDesc dd = d;
    dd.cgroup = 0;
    // channel-tile groups when the weight matrix does not fit an XCD's L2 (see tap_gemm's launch()): 32 workgroups per XCD as
    // (32 / Q pixel tiles) x (Q channel tiles); equal tile footprints, so Q = sqrt(32)
    static const int cg_env = getenv("CCEDIT_CGROUP") ? atoi(getenv("CCEDIT_CGROUP")) : -1;
    const double wbytes = (double)ct_n * 256 * d.Kpad * 2.0;
    if (cg_env != 0 && ct_n > 6 && wbytes > 3.0 * 1024 * 1024) {
        const int q = cg_env > 0 ? cg_env : 6;
        if (q < ct_n) {
            const int ng = (int)((ct_n + q - 1) / q);
            dd.cgroup = (int)((ct_n + ng - 1) / ng);
        }
    }
    if ((dd.cgroup & 0xFFFF) == 0) dd.cgroup |= 1 << 18;        // no groups: cut the XCD ranges at tile granularity
    int wgs = n_cu - n_cu % 8;
    const int64_t tiles = pt_n * ct_n;
    if (tiles < wgs) wgs = (int)((tiles + 7) / 8 * 8);
    hipLaunchKernelGGL(g8_kernel, dim3((unsigned)wgs), dim3(512), G8_LDS, s, dd);
    return cc_launch_status("g8_kernel");
}
