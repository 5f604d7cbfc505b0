// Fused transformer feed-forward for dim = 320 (the 64x96 level) on gfx950:
//
//     out = x + W2 . GEGLU( W1 . LayerNorm(x) + b1 ) + b2          (attention.py:115-141, 695-716 / 758-761)
//
// in ONE kernel that reads x once and writes out once (and, round 5, the two projections around it: "Block tail" below).  The 4C = 1280-wide hidden activation (535 MB per call at
// 34 x 64 x 96 tokens) never exists in memory, LayerNorm is applied to the operand registers, and the residual is folded
// into the accumulator initialisation.
//
// Register chaining (v_mfma_f32_32x32x16_bf16; A = weights from LDS, B = tokens, C/D = [rows][tokens]).  A wave owns 32
// tokens for the whole call:
//   * X fragments: lane (token n = l & 31, hi = l >> 5) holds channels 16 s + 8 hi .. + 7 for k-step s: 20 k-steps = 80
//     VGPRs, loaded once, normalised in place (gamma / beta live in the packed W1' / b1').
//   * GEMM1 (K = 320): value tile and gate tile of 32 hidden units each; in the C layout lane (n, hi) register r is hidden
//     row (r & 3) + 8 (r >> 2) + 4 hi of token n.  After GEGLU the 16 results of a lane, taken in register order, ARE the
//     two B operands of the second GEMM (k-step kappa = registers 8 kappa .. 8 kappa + 7): the packer orders W2's K axis
//     accordingly, no lane ever exchanges data.
//   * GEMM2 (K = 32 hidden per chunk) accumulates out[320 x 32] in 160 accumulator registers across the 40 hidden
//     chunks.  W2's rows are permuted so that registers 0..7 / 8..15 of accumulator tile t of lane (n, hi) are channels
//     32 t + 8 hi .. + 7 / 32 t + 16 + 8 hi .. + 7 — the channels of X fragments 2 t / 2 t + 1 of the same lane: the
//     accumulators start from b2 + x (the residual, exact in fp32) and are stored with 16-byte writes, no shuffle.
//
// One wave per SIMD (a wave needs ~340 registers): only the wave's own instruction stream hides latencies, and every
// instruction costs the wave ~4 cycles of issue time (PMC: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.1 cycles), so the
// kernel is an instruction-count problem: per 32 hidden units 60 MFMAs (1920 cycles of matrix pipe) against ~400 VALU
// (the exact-erf GELU is 16 per element), 68 LDS reads and 16 DMA instructions.  Hence the 32x32x16 shape (the 16x16x32
// version of this kernel, three 16-token tiles per wave, was at its register limit and ran at the SUM of its MFMA, VALU
// and DMA issue times: 750 us against 620-640 us for this one and 940 us for LayerNorm + two GEMMs at 34 x 64 x 96
// tokens), the asm-pinned LDS read ring, scalar fp32 GEGLU, and the software pipeline below.  Measured on this version:
// MFMA pipe busy 36 % of the kernel; waves issue 53 %, wait 23 %, stall 24 % of their cycles.
//
// Weights stream through LDS in 64 KB chunks, pre-packed in fragment order so that every ds_read_b128 is lane-linear
// (conflict-free) and every global_load_lds moves 1 KB contiguous; two chunk buffers, one barrier per chunk.  Every
// weight byte staged serves 128 tokens: 64 KB per ~2800 cycles = the ~23 B/clk a CU's L2 -> LDS path sustains, i.e. the
// weight stream and the MFMA + exposed VALU time are co-limiting.
//
// Block tail (round 5).  The transformer blocks of this level end  tok = to_out(attn) + tok;  tok = FF(LN(tok)) + tok;
// out = proj_out(tok) + x_in  (attention.py:695-716 / 758-761, 865-889, 1141-1208): two K = 320 Linears around the feed-forward, each
// of which is a 134 MB read + 134 MB residual read + 134 MB write that runs AT its HBM roofline as a launch of its own.  With
// PRO / EPI the kernel owns the whole tail on the row tile it already holds:
//   PRO   x is not read: the wave loads its 32 rows of the attention output as B fragments, 4 stream chunks of 50 A-fragments
//         (5 k-steps x 10 output tiles of W_o, rows permuted like W2's) accumulate  b_o + W_o . attn  in the 160 accumulator registers,
//         the residual rows (requested at the top of the round, so their latency hides behind the 200 MFMAs) are added in fp32 and the
//         bf16 rounding of the sum IS the X fragment set of the feed-forward — same register mapping as GEMM2's output, no shuffle;
//   EPI   the feed-forward's result is rounded to bf16 into the same fragment registers, 4 more chunks accumulate
//         b_p + W_p . tok, the x_in rows (requested when the epilogue starts) are added and the sum is stored.
// The chunk machinery (64 KB chunks, two LDS buffers, barrier A / B, the fragment read ring) is the feed-forward's; a round is
// [P0..P3] F0..F41 [E0..E3].  Ring depth 5 divides both chunk lengths (50 and 60 fragments), so the ring runs across all of them.
#include "common.h"
#ifndef KD_VALUE
#define KD_VALUE 5
#endif
// block-tail schedule knobs (probe builds: tools/exp/build_variant.sh -DTAIL_...=0/1; the defaults are the measured best)
#ifndef TAIL_RF_LATE
#define TAIL_RF_LATE 1          // residual rows added after the projection's LAST chunk (0: after its first)
#endif
#ifndef TAIL_DMA_EARLY
#define TAIL_DMA_EARLY 1        // prologue / epilogue chunks issue the next chunk's DMA in their first steps (0: every other step)
#endif
#ifndef TAIL_PREFETCH_A
#define TAIL_PREFETCH_A 0       // next round's attention-output rows requested during the epilogue GEMM
#endif
#include <stdlib.h>
#include <utility>

namespace {

constexpr int kC = 320;                 // model width
constexpr int kKS = kC / 16;            // 20 k-steps of 16
constexpr int kOT = kC / 32;            // 10 output-channel tiles of 32
constexpr int kFF = 1280 / 32 + 2;      // 40 hidden chunks of 32, software-pipelined (three stages) over 42 iterations
constexpr int kPE = 50;                 // fragments of a prologue / epilogue chunk: 5 k-steps x 10 output tiles
constexpr int kPEChunks = 4;            // 4 x 5 = the 20 k-steps of a 320 x 320 projection
constexpr int kAuxOff = 60 * 1024;      // 40 GEMM1 + 20 GEMM2 fragments of 1 KB, then b1'
constexpr int kChunkBytes = 64 * 1024;  // every wave issues at most 16 DMA instructions per chunk
constexpr int kWavePix = 32;
constexpr int kD = KD_VALUE;            // slots of the fragment read ring (divides 60 and 50: the ring runs across chunks)
static_assert(60 % kD == 0 && kPE % kD == 0, "ring depth");
// Which slot a step refills.  FF_RING_LAG 0: the slot its own MFMA has just been issued with (kD fragments ahead) — the ds_read then
// writes registers that MFMA is still reading as its A operand, and the hardware holds the read's issue until the MFMA has fetched
// them.  1: the slot of the PREVIOUS step's MFMA (kD - 1 fragments ahead), which was issued a whole step earlier.  Measured neutral
// (block tail 760.2 / 762.1 us, feed-forward alone 640.8 / 632.8 us for 0 / 1): the stall is not what bounds the chunks; 0 stays.
#ifndef FF_RING_LAG
#define FF_RING_LAG 0
#endif
constexpr int kLA = kD - FF_RING_LAG;   // fragments in flight ahead of their use

typedef __attribute__((ext_vector_type(2))) float f32x2;

template <class F, int... J>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, J...>) {
    (f(std::integral_constant<int, J>{}), ...);
}

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// LDS reads the compiler cannot move: one wave per SIMD means nothing but the wave's own later instructions hides a
// ds_read's latency, and hipcc sinks reads next to their consumer under register pressure (measured on the first version
// of this kernel: 2 reads, lgkmcnt(0), 6 MFMAs, ... = 34 - 61 cycles per MFMA).  The reads are issued kD fragments ahead
// through these statements and retired with counted waits (LDS returns in order); the wait names the register, so no
// consumer can be scheduled above it.
#define FF_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))
#define FF_WAIT(reg, cnt) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(reg) : "i"(cnt))

// GEGLU as a software pipeline across the MFMA steps.  A volatile asm statement ends a scheduling region, so the VALU work
// placed between two of them stays between them.  One step gets ONE stage of the exact-erf GELU (common.h: gelu_erf_f,
// A&S 7.1.28) for EIGHT accumulator registers at once: eight independent instructions, each reading what the previous step
// produced — the VALU pipe runs at issue rate instead of at the latency of a 17-deep dependent chain (which is what one
// element per step cost: tools/exp/coexec.hip shows a wave interleaving MFMAs with 8 independent VALU chains overlaps them
// almost completely, and the first version of this schedule spent as long in the GEGLU as in the MFMAs).
struct GeluPipe {
    float u[8], hv[8], ax[8], x[8], p[8];
    // stage 0 loads (value, gate) of 8 accumulator registers; stage kStages - 1 leaves v * gelu(u) in p[]
    static constexpr int kStages = 16;
    // hipcc moves plain VALU instructions across volatile asm statements; an empty asm that takes the stage's live values
    // as read-write operands pins the stage between the statements of its step
    __device__ __forceinline__ void pin() {
        asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]));
    }
    template <int ST>
    __device__ __forceinline__ void stage(const f32x16& accv, const f32x16& accg, int r0, bool plain) {
        if constexpr (ST > 3) pin();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (ST == 0) { u[i] = accg[r0 + i]; hv[i] = 0.5f * accv[r0 + i]; }
            else if constexpr (ST == 1) ax[i] = fabsf(u[i]);
            else if constexpr (ST == 2) x[i] = ax[i] * 0.70710678118654752440f;
            else if constexpr (ST == 3) p[i] = fmaf(x[i], 0.0000430638f, 0.0002765672f);
            else if constexpr (ST == 4) p[i] = fmaf(x[i], p[i], 0.0001520143f);
            else if constexpr (ST == 5) p[i] = fmaf(x[i], p[i], 0.0092705272f);
            else if constexpr (ST == 6) p[i] = fmaf(x[i], p[i], 0.0422820123f);
            else if constexpr (ST == 7) p[i] = fmaf(x[i], p[i], 0.0705230784f);
            else if constexpr (ST == 8) p[i] = fmaf(x[i], p[i], 1.0f);
            else if constexpr (ST >= 9 && ST <= 12) p[i] = p[i] * p[i];
            else if constexpr (ST == 13) p[i] = 1.0f - __builtin_amdgcn_rcpf(p[i]);                 // erf(|u| / sqrt 2)
            else if constexpr (ST == 14) p[i] = fmaf(ax[i], p[i], u[i]);                             // u + |u| erf = 2 gelu(u)
            else if constexpr (ST == 15) p[i] = plain ? 2.0f * hv[i] + u[i] : hv[i] * p[i];          // v gelu(u)
        }
        if constexpr (ST == 0) asm volatile("" : "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]), "+v"(u[4]), "+v"(u[5]), "+v"(u[6]), "+v"(u[7]));
        if constexpr (ST == 1) asm volatile("" : "+v"(ax[0]), "+v"(ax[1]), "+v"(ax[2]), "+v"(ax[3]), "+v"(ax[4]), "+v"(ax[5]), "+v"(ax[6]), "+v"(ax[7]));
        if constexpr (ST == 2) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
        if constexpr (ST >= 3) pin();
    }
};

// ABL (tuning only, CCEDIT_FF320_ABL, bit mask; PRO = EPI = false only): 1 = no weight stream after the first chunk; 2 = GEGLU replaced
// by an add; 32 = phase cycle counters into d.dbg.  Results are wrong for ABL & 3.
//
// Software pipeline over the 40 hidden chunks, three stages deep: iteration c = 0 .. 41 works on stream chunk
// c = [W1'(c) | W2(c-2) | b1'(c)], its 60 fragments interleaved per k-step s as (value_s, gate_s, W2 fragment s):
//     MFMA   GEMM1 of chunk c  (value / gate accumulators: a dependent MFMA every third issue = 96 cycles apart)
//            GEMM2 of chunk c-2 (twenty different accumulators)
//     VALU   GEGLU of chunk c-1, one accumulator register (one hidden row of the lane's token) per k-step
// Chunks -2, -1, 40, 41 do not exist: the packer supplies zero fragments / biases, so those MFMAs add zeros.
// The fragment ring runs across iterations: barrier B (step NF - 2 - kLA, after the wait for the next chunk's DMA) lets the last
// steps prefetch the next chunk; barrier A (iteration top) lets the DMA overwrite the buffer every wave has left.
//
// One chunk of the round, as a compile-time description: KIND (0 prologue GEMM, 1 feed-forward, 2 epilogue GEMM), Q (index of a
// prologue / epilogue chunk: k-steps 5 Q .. 5 Q + 4), PAR (LDS buffer = stream index & 1), HAS_BQ (the chunk before this one read
// b1' rows ahead: they are retired — and, in a feed-forward chunk, used — at step 0), NEXT_BQ (read the next chunk's b1' ahead),
// NISS (DMA instructions per wave that move the next chunk: 16 for a feed-forward chunk with its b1' rows, 13 for 50 fragments).
template <int KIND_, int Q_, int PAR_, bool HAS_BQ_, bool NEXT_BQ_, int NISS_>
struct Chunk {
    static constexpr int KIND = KIND_, Q = Q_, PAR = PAR_, NISS = NISS_;
    static constexpr bool HAS_BQ = HAS_BQ_, NEXT_BQ = NEXT_BQ_;
    static constexpr int NF = KIND_ == 1 ? 60 : kPE;
};

template <int ABL, bool PRO, bool EPI>
__global__ __launch_bounds__(256, 1) void ff320_kernel(const CcFf320Desc d, int n_rounds) {
    static_assert(ABL == 0 || (!PRO && !EPI), "ablations exist for the plain feed-forward only");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][kChunkBytes]
    constexpr int cF = PRO ? kPEChunks : 0;                              // stream index of the first feed-forward chunk
    constexpr int cE = cF + kFF;                                         // ... of the first epilogue chunk
    constexpr int NCH = cE + (EPI ? kPEChunks : 0);                      // chunks of one round
    static_assert(NCH % 2 == 0 && cF % 2 == 0 && cE % 2 == 0, "a chunk's buffer is its stream index & 1 in every round");
    constexpr bool FIRST_BQ = !PRO;                                      // the first chunk of a round is a feed-forward chunk
    constexpr bool FIRST_HAS_BQ = PRO ? !EPI : true;                     // ... or retires the b1' rows the last chunk of a round reads ahead
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hi = lane >> 5;
    const char* __restrict__ wstream = (const char*)d.wstream;
    const bf16* __restrict__ xp = (const bf16*)d.x;
    bf16* __restrict__ op = (bf16*)d.out;
    const f32x4* __restrict__ b2p = (const f32x4*)d.b2p;                 // [kOT][2 lane halves][16 registers]
    const int lane16 = lane * 16;

    // DMA share of this wave: fragments wave, wave + 4, ... (at most 16 per chunk); uniform base + lane offset
    auto issue_frag = [&](int c, int buf, int k) {
        const char* src = wstream + ((size_t)c * kChunkBytes + (size_t)(k * 4 + wave) * 1024);     // wave-uniform
        glds16(src + lane16, smem + buf * kChunkBytes + (k * 4 + wave) * 1024);
    };
    const unsigned la0 = (unsigned)(uintptr_t)(LDS_AS const char*)(smem + lane16);                  // fragment base, buffer 0
    const unsigned lx0 = (unsigned)(uintptr_t)(LDS_AS const char*)(smem + kAuxOff + hi * 64);      // b1' of this lane half, buffer 0

    const int gdim = (int)gridDim.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) issue_frag(0, 0, k);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the read stream starts: the first kLA fragments of chunk 0 (and its b1' when it is a feed-forward chunk)
    f32x4 bq[2][4];                      // b1' (value, gate) of the chunk GEMM1 starts next, in accumulator order
    bf16x8 ring[kD];
    FF_DS_READ(ring[0], la0, 0);          // same order as in the steady state: fragment 0, b1', fragments 1 .. kLA - 1
    if constexpr (FIRST_HAS_BQ) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) FF_DS_READ(bq[k][q4], lx0, k * 128 + q4 * 16);
    }
#pragma unroll
    for (int j = 1; j < kLA; ++j) FF_DS_READ(ring[j], la0, j * 1024);

    f32x16 acc1[2][2];                   // [chunk parity][value, gate]
    bf16x8 hf[2][2];                     // [chunk parity][kappa]: GEGLU outputs = GEMM2 B operands
    // What the software pipeline reads before writing it — the GEGLU input of "chunk -1" and the GEMM2 operand of "chunk -2" (the
    // fragments they meet are zeros).  Plain feed-forward: set once, the values that cross a round boundary are as harmless.  Block
    // tail: set every round, so that nothing of the pipeline is live across the prologue / epilogue GEMMs.
    auto pipeline_reset = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hf[0][q][e] = f2bf(0.f);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[1][q][r] = 0.f;
        }
    };
    if constexpr (!PRO) pipeline_reset();

    // Every workgroup starts its round at the same moment and asks HBM for its 80 KB of rows at once: ~7k cycles per round when the
    // wave has to wait for them (MI355X_MICROARCH.md, "prologue HBM burst").  The block tail reads three such row sets per round; two
    // are requested four chunks before they are used (residual rows: `rf`), and with the epilogue GEMM present the NEXT round's
    // attention-output rows are requested while the epilogue runs (`xnext`), so only the very first round waits for its input.
    constexpr bool PREFETCH_A = PRO && EPI && TAIL_PREFETCH_A;
    bf16x8 xnext[kKS];
    auto load_rows = [&](bf16x8* dst, const bf16* base, int ld, int rnd) __attribute__((always_inline)) {
        const int64_t q = ((int64_t)rnd * 4 + wave) * kWavePix + n;
        const bf16* row = base + (q < d.M ? q : d.M - 1) * ld + hi * 8;                 // rows past M repeat the last row; never stored
#pragma unroll
        for (int s = 0; s < kKS; ++s) dst[s] = *(const bf16x8*)(row + s * 16);
    };
    if constexpr (PREFETCH_A) load_rows(xnext, (const bf16*)d.a, d.lda, (int)blockIdx.x);          // (grid <= rounds)

    for (int round = blockIdx.x; round < n_rounds; round += gdim) {
        const int64_t p = ((int64_t)round * 4 + wave) * kWavePix + n;
        const int64_t pr = p < d.M ? p : d.M - 1;                         // rows past M repeat the last row; never stored
        const bool last_round = round + gdim >= n_rounds;
        // ---- this wave's tokens as B-operand fragments: x (plain) or the attention output (PRO); with PRO also the residual rows,
        //      which are only needed after the 200 prologue MFMAs ----
        bf16x8 xf[kKS];
        bf16x8 rf[kKS];                  // PRO: residual rows of the prologue; EPI: x_in rows of the epilogue (see add_rows)
        {
            if constexpr (PREFETCH_A) {
#pragma unroll
                for (int s = 0; s < kKS; ++s) xf[s] = xnext[s];
            } else {
                const bf16* row = (PRO ? (const bf16*)d.a + pr * d.lda : xp + pr * d.ldx) + hi * 8;
#pragma unroll
                for (int s = 0; s < kKS; ++s) xf[s] = *(const bf16x8*)(row + s * 16);
            }
            if constexpr (PRO) {
                const bf16* rrow = (const bf16*)d.res + pr * d.ldr + hi * 8;
#pragma unroll
                for (int s = 0; s < kKS; ++s) rf[s] = *(const bf16x8*)(rrow + s * 16);
            }
        }
        f32x16 acc2[kOT];                // the 320 x 32 accumulator tile of whichever GEMM is running (prologue, GEMM2, epilogue)
        auto init_acc = [&](const f32x4* __restrict__ bias, bool add_x) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < kOT; ++t) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const f32x4 bv = bias[(t * 2 + hi) * 4 + q4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = q4 * 4 + e;
                        acc2[t][r] = add_x ? bv[e] + bf2f(xf[2 * t + (r >> 3)][r & 7]) : bv[e];
                    }
                }
            }
        };
        // registers 0..7 / 8..15 of accumulator tile t are channels 32 t + 8 hi .. / 32 t + 16 + 8 hi .. of token n = the channels of
        // fragments 2 t / 2 t + 1 of the same lane: accumulator -> bf16 -> the next GEMM's B operand
        auto acc_to_frags = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < kOT; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xf[2 * t][e] = f2bf(acc2[t][e]);
                    xf[2 * t + 1][e] = f2bf(acc2[t][8 + e]);
                }
        };

        unsigned long long tacc[2] = {0, 0};            // ABL & 32: cycles in [barrier A, the 60 steps]
        // one chunk; `c` = its stream index (run time for the feed-forward chunks), CH = Chunk<...>
        auto iteration = [&](auto chunk, int c, bool final_chunk) __attribute__((always_inline)) {
            using CH = decltype(chunk);
            constexpr int PAR = CH::PAR, NF = CH::NF, KIND = CH::KIND;
            unsigned long long tm0 = 0, tm1 = 0;
            if constexpr (ABL & 32) tm0 = __builtin_amdgcn_s_memtime();
            // barrier A: every wave has left the other buffer (chunk c - 1) -> the DMA of chunk c + 1 may overwrite it
            __builtin_amdgcn_s_barrier();
            if constexpr (ABL & 32) tm1 = __builtin_amdgcn_s_memtime();
            const bool last = final_chunk && last_round;
            const bool issue_next = !last && !((ABL & 1) && (c > 0 || round != (int)blockIdx.x));
            int cn = final_chunk ? 0 : c + 1;
            // (opaque to the optimiser: with the compile-time stream indices of the prologue / epilogue chunks hipcc precomputes the 13
            //  DMA source addresses of every one of them before the round loop — 200 registers — and spills them)
            asm volatile("" : "+s"(cn));
            const unsigned la = la0 + PAR * kChunkBytes, lan = la0 + (1 - PAR) * kChunkBytes;
            const unsigned lxn = lx0 + (1 - PAR) * kChunkBytes;

            GeluPipe gp;
            auto step = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                // younger reads that may stay in flight: the kLA - 1 fragments behind this one, + the 8 b1' reads slipped in
                // behind fragment NF (= fragment 0 of the next chunk, read at step NF - kLA) while that one is among them
                constexpr int young8 = kLA - 1 + 8 > 15 ? 15 : kLA - 1 + 8;                            // (lgkmcnt is a 4-bit counter)
                FF_WAIT(ring[j % kD], (CH::NEXT_BQ && j >= NF + 1 - kLA ? young8 : kLA - 1));
                if constexpr (j == 0 && CH::HAS_BQ) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) FF_WAIT(bq[k][q4], kLA - 1);     // older than fragment 1: retired with fragment 0
                }
                if constexpr (KIND == 1) {
                    constexpr int s = j / 3, role = j % 3;                        // k-step, (value, gate, GEMM2) fragment
                    if constexpr (role < 2) {        // GEMM1 of chunk c: value / gate tile, C operand of the first k-step = b1'
                        if constexpr (s == 0) {
                            f32x16 c0;
#pragma unroll
                            for (int r = 0; r < 16; ++r) c0[r] = bq[role][r >> 2][r & 3];
                            acc1[PAR][role] = mfma32(ring[j % kD], xf[s], c0);
                        } else {
                            acc1[PAR][role] = mfma32(ring[j % kD], xf[s], acc1[PAR][role]);
                        }
                    } else {                         // GEMM2 of chunk c - 2: fragment s = (kappa = s / 10, out tile s % 10)
                        acc2[s % kOT] = mfma32(ring[j % kD], hf[PAR][s / kOT], acc2[s % kOT]);
                    }
                } else {                             // prologue / epilogue GEMM: fragment j = (k-step 5 Q + j / 10, out tile j % 10)
                    acc2[j % kOT] = mfma32(ring[j % kD], xf[CH::Q * (kPE / kOT) + j / kOT], acc2[j % kOT]);
                }
                // the ring continues into the next chunk's buffer behind barrier B
                if constexpr (j + kLA < NF) FF_DS_READ(ring[(j + kLA) % kD], la, (j + kLA) * 1024);
                else FF_DS_READ(ring[(j + kLA) % kD], lan, (j + kLA - NF) * 1024);
                if constexpr (CH::NEXT_BQ && j == NF - kLA) {
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) FF_DS_READ(bq[k][q4], lxn, k * 128 + q4 * 16);
                }
                // the next chunk's DMA: spread over the MFMA steps of a feed-forward chunk (16 of 60 steps); a prologue / epilogue chunk is
                // 50 bare MFMAs (1600 cycles) against ~2300 cycles of DMA, so there the requests go out first
                constexpr int dma_k = (KIND != 1 && TAIL_DMA_EARLY) ? j : ((j & 1) == 0 ? (j >> 1) : 99);
                if constexpr (dma_k < CH::NISS && j < 32) {
                    if (issue_next) issue_frag(cn, 1 - PAR, dma_k);
                }
                if constexpr (j == NF - 2 - kLA) {   // barrier B: chunk c + 1 has landed for every wave
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                if constexpr (KIND == 1) {
                    // GEGLU of chunk c - 1: registers 0..7 in steps 2..17 + 18 (stages 0..15, then the bf16 pack), registers 8..15 in
                    // steps 20..35 + 36 — one stage of eight independent VALU instructions beside each step's MFMA
                    if constexpr (j >= 2 && j < 2 + GeluPipe::kStages) gp.template stage<j - 2>(acc1[1 - PAR][0], acc1[1 - PAR][1], 0, (ABL & 2) != 0);
                    if constexpr (j >= 20 && j < 20 + GeluPipe::kStages) gp.template stage<j - 20>(acc1[1 - PAR][0], acc1[1 - PAR][1], 8, (ABL & 2) != 0);
                    if constexpr (j == 18 || j == 36) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) hf[1 - PAR][j == 18 ? 0 : 1][i] = f2bf(gp.p[i]);
                    }
                }
            };
            for_seq(step, std::make_integer_sequence<int, NF>{});
            if constexpr (ABL & 32) {
                const unsigned long long tm2 = __builtin_amdgcn_s_memtime();
                tacc[0] += tm1 - tm0;
                tacc[1] += tm2 - tm1;
            }
        };

        // the residual rows of a projection, requested when its first chunk starts and added (fp32) after its last: their latency
        // hides behind the projection's 200 MFMAs and four chunk DMAs
        auto add_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < kOT; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc2[t][e] += bf2f(rf[2 * t][e]);
                    acc2[t][8 + e] += bf2f(rf[2 * t + 1][e]);
                }
        };
        // ---- prologue GEMM: tok = W_o . attn + b_o + residual, left in xf as the feed-forward's input ----
        if constexpr (PRO) {
            init_acc((const f32x4*)d.bop, false);
            iteration(Chunk<0, 0, 0, !EPI, false, 13>{}, 0, false);      // (the round before ended with a feed-forward chunk iff !EPI)
            if constexpr (!TAIL_RF_LATE) add_rows();
            iteration(Chunk<0, 1, 1, false, false, 13>{}, 1, false);
            iteration(Chunk<0, 2, 0, false, false, 13>{}, 2, false);
            iteration(Chunk<0, 3, 1, false, true, 16>{}, 3, false);
            if constexpr (TAIL_RF_LATE) add_rows();
            acc_to_frags();
        }
        // ---- LayerNorm statistics (two passes over the registers, the two lanes of a token meet once); the accumulators of the
        //      second GEMM start from b2 + x; then the fragments are normalised in place ----
        {
            float sm = 0.f;
#pragma unroll
            for (int s = 0; s < kKS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) sm += bf2f(xf[s][e]);
            sm += __shfl_xor(sm, 32);
            const float mu = sm * (1.0f / kC);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < kKS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dv = bf2f(xf[s][e]) - mu;
                    sq = fmaf(dv, dv, sq);
                }
            sq += __shfl_xor(sq, 32);
            const float rs = d.ln ? rsqrtf(sq * (1.0f / kC) + d.eps) : 1.f;
            const float rm = d.ln ? rs * mu : 0.f;
            init_acc(b2p, true);
#pragma unroll
            for (int s = 0; s < kKS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[s][e] = f2bf(fmaf(rs, bf2f(xf[s][e]), -rm));
        }

        // ---- feed-forward: 42 chunks, unrolled by two (PAR selects the register sets with static indices) ----
        if constexpr (PRO) pipeline_reset();
        static_assert(kFF % 2 == 0, "the iteration loop is unrolled by two");
        for (int c = 0; c < kFF; c += 2) {
            iteration(Chunk<1, 0, 0, true, true, 16>{}, cF + c, false);
            iteration(Chunk<1, 0, 1, true, true, 16>{}, cF + c + 1, !EPI && c + 2 == kFF);
        }
        if constexpr (ABL & 32) {
            if (round == (int)blockIdx.x && lane == 0 && d.dbg) {
                unsigned long long* o = (unsigned long long*)d.dbg + ((size_t)blockIdx.x * 4 + wave) * 4;
                o[0] = tacc[0];
                o[1] = tacc[1];
                o[2] = o[3] = 0;
            }
        }

        // ---- epilogue GEMM: out = W_p . tok + b_p + x_in ----
        if constexpr (EPI) {
            acc_to_frags();
            {
                const bf16* rrow = (const bf16*)d.res2 + pr * d.ldr2 + hi * 8;
#pragma unroll
                for (int s = 0; s < kKS; ++s) rf[s] = *(const bf16x8*)(rrow + s * 16);
            }
            init_acc((const f32x4*)d.bpp, false);
            iteration(Chunk<2, 0, 0, true, false, 13>{}, cE, false);     // (retires the b1' rows the last feed-forward chunk read ahead)
            if constexpr (!TAIL_RF_LATE) add_rows();
            if constexpr (PREFETCH_A) {
                // unconditional (the last round re-reads its own rows): a conditional load would keep xnext's old value — and its 80
                // registers — alive through the whole feed-forward
                load_rows(xnext, (const bf16*)d.a, d.lda, last_round ? round : round + gdim);
            }
            iteration(Chunk<2, 1, 1, false, false, 13>{}, cE + 1, false);
            iteration(Chunk<2, 2, 0, false, false, 13>{}, cE + 2, false);
            iteration(Chunk<2, 3, 1, false, FIRST_BQ, FIRST_BQ ? 16 : 13>{}, cE + 3, true);
            if constexpr (TAIL_RF_LATE) add_rows();
        }
        if (last_round) {    // the reads issued ahead for a chunk that does not exist: retire them before their registers are reused
#pragma unroll
            for (int j = 0; j < kD; ++j) FF_WAIT(ring[j], 0);
        }

        // ---- store: registers 0..7 / 8..15 of accumulator tile t are channels 32 t + 8 hi .. / 32 t + 16 + 8 hi .. of token n ----
        if (p < d.M) {
            bf16* row = op + p * d.ldo + hi * 8;
#pragma unroll
            for (int t = 0; t < kOT; ++t) {
                bf16x8 o0, o1;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o0[e] = f2bf(acc2[t][e]);
                    o1[e] = f2bf(acc2[t][8 + e]);
                }
                *(bf16x8*)(row + t * 32) = o0;
                *(bf16x8*)(row + t * 32 + 16) = o1;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace

extern "C" int ccedit_ff320(const CcFf320Desc* desc, void* stream) {
    CC_CHECK_ARG(desc != nullptr, "ccedit_ff320: null descriptor");
    const CcFf320Desc d = *desc;
    CC_CHECK_ARG(d.out && d.wstream && d.b2p, "ccedit_ff320: null out/wstream/b2p");
    CC_CHECK_ARG(d.M > 0, "ccedit_ff320: M=%lld", (long long)d.M);
    CC_UNSUPPORTED(d.dim != kC || d.inner != 1280, "ccedit_ff320: only dim 320 / inner 1280 (got %d / %d)", d.dim, d.inner);
    const bool pro = d.a != nullptr, epi = d.res2 != nullptr;
    if (pro) {
        CC_CHECK_ARG(d.res && d.bop, "ccedit_ff320: the prologue GEMM (a) needs its residual (res) and bias (bop)");
        CC_UNSUPPORTED(d.lda % 8 != 0 || d.ldr % 8 != 0 || d.lda < kC || d.ldr < kC, "ccedit_ff320: lda=%d / ldr=%d", d.lda, d.ldr);
    } else {
        CC_CHECK_ARG(d.x != nullptr && !d.res && !d.bop, "ccedit_ff320: null x (or res / bop without a)");
        CC_UNSUPPORTED(d.ldx % 8 != 0 || d.ldx < kC, "ccedit_ff320: ldx=%d", d.ldx);
    }
    if (epi) {
        CC_CHECK_ARG(d.bpp != nullptr, "ccedit_ff320: the epilogue GEMM (res2) needs its bias (bpp)");
        CC_UNSUPPORTED(d.ldr2 % 8 != 0 || d.ldr2 < kC, "ccedit_ff320: ldr2=%d", d.ldr2);
    } else {
        CC_CHECK_ARG(!d.bpp, "ccedit_ff320: bpp without res2");
    }
    CC_UNSUPPORTED(epi && !pro, "ccedit_ff320: the epilogue GEMM is built together with the prologue GEMM only");
    CC_UNSUPPORTED(d.ldo % 8 != 0 || d.ldo < kC, "ccedit_ff320: ldo=%d", d.ldo);
    const int64_t rounds = (d.M + 4 * kWavePix - 1) / (4 * kWavePix);
    CC_UNSUPPORTED(rounds > 2147483647LL, "ccedit_ff320: M too large");
#ifdef CCEDIT_TUNING      // probe builds only (-DCCEDIT_TUNING): the product library never reads a switch that changes results
    static const int abl = getenv("CCEDIT_FF320_ABL") ? atoi(getenv("CCEDIT_FF320_ABL")) : 0;
#else
    constexpr int abl = 0;
#endif
    void (*kern)(const CcFf320Desc, int) = ff320_kernel<0, false, false>;
    int slot = 0;
    if (pro && epi) kern = ff320_kernel<0, true, true>, slot = 1;
    else if (pro) kern = ff320_kernel<0, true, false>, slot = 2;
#ifdef CCEDIT_TUNING
    if (!pro) switch (abl) {        // tuning only (bit mask, see the kernel); results are wrong for abl != 0
        case 1: kern = ff320_kernel<1, false, false>; slot = 3; break;
        case 2: kern = ff320_kernel<2, false, false>; slot = 4; break;
        case 3: kern = ff320_kernel<3, false, false>; slot = 5; break;
        case 32: kern = ff320_kernel<32, false, false>; slot = 6; break;
        default: break;
    }
#endif
    static unsigned long long attr_done[8] = {0};
    if (int rc = cc_max_dynamic_lds((const void*)kern, 2 * kChunkBytes, &attr_done[slot], "ff320")) return rc;
    int cus = 256;
    {
        int dev = 0;
        static int cu_cache[64] = {0};
        if (hipGetDevice(&dev) == hipSuccess && dev < 64) {
            if (!cu_cache[dev]) {
                int v = 0;
                if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cu_cache[dev] = v;
            }
            if (cu_cache[dev]) cus = cu_cache[dev];
        }
    }
    const int grid = (int)(rounds < cus ? rounds : cus);
    cc_note_kernel(pro && epi ? "ff320_kernel (block tail: to_out + FF + proj_out)" : pro ? "ff320_kernel (to_out + FF)" : "ff320_kernel");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 2 * kChunkBytes, (hipStream_t)stream, d, (int)rounds);
    return cc_launch_status("ff320_kernel");
}
