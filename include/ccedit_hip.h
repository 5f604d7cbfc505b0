/*
 * ccedit_hip.h — C ABI of libccedit_hip.so: the MI355X (gfx950) kernels behind CCEdit's denoising
 * hot path (sgm pseudo-3D UNet + ControlNet2D + DPMPP2SAncestral + AutoencoderKL decode).
 *
 * The reference (RuoyuFeng/CCEdit) is pure Python on ATen and has NO FFI of its own: its operator
 * API is `instantiate_from_config` (sgm/util.py:168-185).  This header is the boundary the build adds
 * underneath that API.  Each entry point names the reference call sites whose vendor kernels
 * (cuDNN / cuBLAS / SDPA / ATen elementwise) it replaces.  INTEGRATION.md shows the Python-side
 * binding (ctypes) a reference maintainer would add.
 *
 * Conventions
 *   - plain C: device pointers as void*, sizes as int32/int64, scalars by value, one POD descriptor
 *     struct per fused op; no torch types.
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns immediately;
 *     no host synchronisation, no allocation, no ownership transfer.  Workspaces are caller-owned.
 *   - return value: 0 = ok; <0 = CCEDIT_E* (invalid argument / unsupported shape); >0 = hipError_t.
 *     `ccedit_last_error()` returns a thread-local message for the last non-zero return.
 *   - activations: bf16, "frames-outermost channels-last": logical (B*T, C, H, W) stored with
 *     strides (H*W*C, 1, W*C, C), i.e. a row-major [B*T*H*W pixels][C] matrix.  Weights: bf16,
 *     packed [Cout_pad][taps][Cin_pad] (K contiguous) by ccedit_amd/packing.py from the reference's
 *     OIHW / (O,I,K) / (O,I) fp32 tensors.  Biases, norm affine parameters, statistics: fp32.
 */
#ifndef CCEDIT_HIP_H
#define CCEDIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCEDIT_ABI_VERSION 12

#define CCEDIT_OK 0
#define CCEDIT_EINVAL (-1)       /* null pointer / bad size */
#define CCEDIT_EUNSUPPORTED (-2) /* shape or option outside what the kernels implement */

int ccedit_abi_version(void);
const char* ccedit_last_error(void);
/* name of the kernel template the last ccedit_gemm / ccedit_ff320 / ccedit_attention call of this thread dispatched to
 * ("g8_kernel 256x256", "conv_halo_kernel", "tap_gemm_kernel 128x128 s2", ...): bench.py's per-kernel roofline table */
const char* ccedit_last_kernel(void);
/* fills name with hipDeviceProp_t.gcnArchName of the current device; returns CU count or <0 */
int ccedit_device_info(char* name, int name_len);

/* Dispatch policy: ONE table of named integer switches that choose between kernels computing the same fp32 sums in a different order
 * (A/B arms and the "specialised kernels reproduce the generic ones" tests).  All default to the fast path.  The library never reads
 * the environment; the host sets entries before launching (process-wide, not thread-safe against concurrent launches).  Names:
 *   conv_halo g8 g8_conv g8_temporal g8_split lin320 lin320s lin640 temp320 attn_short attn_text attn_spatial attn_pv16 attn_opt gn_flat
 *   gn_apply_flat f32_split
 *   (ccedit_policy_names() returns them comma-separated; semantics in csrc/common.h: CcPolicy)
 * Unknown name: CCEDIT_EINVAL. */
int ccedit_policy_set(const char* name, int32_t value);
int ccedit_policy_get(const char* name, int32_t* value);
const char* ccedit_policy_names(void);

/* ------------------------------------------------------------------------------------------
 * Tap-gather GEMM on MFMA: out[m][n] = epilogue( sum_{tap,c} W[n][tap][c] * A[src(m,tap)][c] )
 *
 * One kernel family covers every contraction of the path (reference call sites):
 *   mode LINEAR   nn.Linear (attention.py:377-383 to_q/to_k/to_v/to_out, :118 GEGLU proj, :136 FF out;
 *                 openaimodel.py:1216-1223 time_embed, :470-476 emb_layers), Conv2d 1x1
 *                 (attention.py:818-820 proj_in/out, openaimodel.py:508 skip_connection,
 *                 controlmodel.py:249-250 zero convs, model.py:169-180 VAE q/k/v/proj), Conv1d k=1 over T
 *                 (attention.py:1088-1130 proj_in/out_temporal, openaimodel.py:712-715) — all the same
 *                 [pixels][C] x [Cout][C]^T product in the channels-last layout.
 *   mode CONV2D   Conv2d 3x3 stride 1/2 pad 1 (openaimodel.py:445-449, 483-492, 369-376 Downsample op,
 *                 controlmodel.py:215-231 hint stem, model.py:100-110 VAE), optional fused nearest-2x
 *                 upsample of the source (openaimodel.py:254-263 Upsample3D, model.py:65-71).
 *   mode TEMPORAL Conv1d k=3 pad 1 over the T keyframes of one clip (openaimodel.py:617-629, 674-687,
 *                 250-252, 377-386, 1611-1621, 1627-1632): rows of frame t gather frames t-1, t, t+1
 *                 at the same pixel (row distance H*W), zero outside [0, T).
 * A second source (A2) supplies channels [Cin1, Cin) — the torch.cat([h, hs.pop()+control.pop()])
 * of controlmodel.py:539-543 without materialising it.
 * ------------------------------------------------------------------------------------------ */
enum { CCEDIT_GEMM_LINEAR = 0, CCEDIT_GEMM_CONV2D = 1, CCEDIT_GEMM_TEMPORAL = 2 };
enum { CCEDIT_ACT_NONE = 0, CCEDIT_ACT_SILU = 1, CCEDIT_ACT_GEGLU = 2,
       CCEDIT_ACT_QUICK_GELU = 3 /* x * sigmoid(1.702 x): the CLIP text MLP (FrozenCLIPEmbedder, encoders/modules.py:358-420) */ };

typedef struct CcGemmDesc {
    int64_t M;            /* output rows: pixels (B*T*Hout*Wout) or tokens */
    int32_t N;            /* packed output rows of W actually used (Cout; 2*inner for GEGLU) */
    int32_t Cin;          /* channels per tap (multiple of 8) */
    int32_t Cin1;         /* channels taken from A (== Cin when A2 is null) */
    int32_t taps;         /* 1, 3 (temporal) or 9 (3x3) */
    int32_t mode;         /* CCEDIT_GEMM_* */
    int32_t Hin, Win;     /* CONV2D: stored source frame size */
    int32_t Hout, Wout;   /* CONV2D: output frame size */
    int32_t stride, pad, ksize;
    int32_t upsample;     /* CONV2D: 1 = source is nearest-2x upsampled on the fly */
    int32_t T, HW;        /* TEMPORAL: frames per clip, pixels per frame */
    int32_t lda, lda2;    /* source row strides (elements) */
    int32_t ldc;          /* output row stride (elements) */
    int32_t Kpad;         /* weight row stride (elements), multiple of 64, >= taps*Cin */
    int32_t act;          /* CCEDIT_ACT_* (applied to acc + bias + group_bias, before residuals) */
    int32_t out_f32;      /* 1: store fp32 instead of bf16 */
    int32_t group_rows;   /* rows per group_bias row (T*H*W for the timestep-embedding add); 0 = unused */
    int32_t ldr1, ldr2;   /* residual row strides */
    int32_t tile;         /* 0 = auto; forcing a block shape is for tests / tuning: 1 = 128ch x 128pix, 2 = 64ch x 256pix,
                           * 3-5 = wider 8-wave shapes, 6 = 320ch x 128pix (N % 320 == 0, no gn_stats), 8 = LDS-halo 3x3 conv,
                           * 9 = register-resident weights (Linear, K = 320, N % 320 == 0, bias + one residual or GEGLU epilogue),
                           * 10 = the same at K = 640 (Linear, N % 128 == 0, M % 16 == 0; bias, one residual, row_sums, ln_stats / ln_sums),
                           * 11-13 = persistent eight-phase GEMM (block shape by Cout / 256ch x 256pix / 128ch x 512pix),
                           * 14 = streaming Conv1d k3 over T with 320 input channels (TEMPORAL, N % 64 == 0, HW % 16 == 0, unsharded frames;
                           *      bias, per-clip row bias, up to two residuals, gn_stats) */
    int32_t korder;       /* weight K order: 0 = [tap][Cin]; 1 = [Cin/64][tap][64] (needs Cin % 64 == 0) */
    int32_t gn_rows;      /* with gn_stats: output rows per frame (H*W), a multiple of 128 dividing M (not a
                           * multiple of 256: block shape 1 is used); else 0 */
    /* TEMPORAL with the T keyframes of a clip sharded over ranks (0 = unsharded): the source holds Tsrc frames
     * per clip — tsrc_off halo frames received from the previous rank, the T local frames, then the next rank's —
     * and local frame 0 is global keyframe t0 of Tglob; taps outside [0, Tglob) read zeros (Conv1d padding). */
    int32_t Tsrc, tsrc_off, t0, Tglob;
    int32_t cgroup;       /* internal (set by the library): block-order parameters; pass 0 */
    int32_t ldgb;         /* row stride of group_bias in elements; 0 = N (rows of a wider matrix: all ResBlocks' emb_layers
                           * projections of one network come out of ONE GEMM, each layer reads its column slice) */
    float ln_eps;         /* 0 = off.  > 0 (Linear, K = 320, N % 320 == 0, no activation / residual; tile 0 or 9 — the call always runs the
                           * register-resident-weights kernel and is REFUSED with CCEDIT_EUNSUPPORTED otherwise): every row of
                           * A is LayerNorm-normalised WITHOUT affine, (x - mean) * rsqrt(var + ln_eps) rounded to bf16, before the
                           * product — `to_q(norm(x))` of attention.py:695-716 with gamma / beta folded into W / bias by the caller
                           * (W diag(gamma), b + W beta: ccedit_amd/packing.py:fold_layernorm) */
    const void* A;        /* bf16 [rows][lda] */
    const void* A2;       /* optional second source */
    const void* W;        /* bf16 [ceil(N,256)][Kpad] (rows zero-padded to the widest block shape) */
    const float* bias;    /* [N] in packed row order, or null */
    const float* group_bias; /* [M/group_rows][N] fp32 or null  (ResBlock: h + emb_out, openaimodel.py:762) */
    const void* res1;     /* bf16 residuals added after the activation, or null */
    const void* res2;
    void* out;            /* bf16 (or fp32) [M][ldc]; GEGLU writes N/2 columns */
    /* optional: GroupNorm(32, N) statistics of the tensor being written, accumulated in the epilogue so that the
     * following normalisation (openaimodel.py:441-444, 479-482; attention.py:153-156) does not re-read it:
     * float[M/gn_rows][32][2] (sum, sum of squares of the bf16-rounded outputs), ZEROED BY THE CALLER, added to
     * atomically.  Needs N % 32 == 0, N >= 256, bf16 output, no GEGLU.  Consumed by ccedit_groupnorm_spatial_apply. */
    double* gn_stats;
    /* optional scratch for split-K (outputs with far fewer 256 x 256 tiles than the chip has CUs and a long K loop — the 8x12 level
     * of the networks): ccedit_gemm_workspace_bytes(desc) bytes, 16-byte aligned, whose FIRST 4096 BYTES ARE ZERO before the
     * first call that uses the buffer (the arrival counters; every call leaves them zero again).  One buffer may serve all calls
     * of one stream; calls on different streams need different buffers.  null / too small: no split (same results up to the
     * fp32 summation order of the K loop, which split-K fixes per split count). */
    void* workspace;
    int64_t workspace_bytes;
    int32_t split_k;      /* internal (set by the library); pass 0 */
    /* 0 = off.  p + 1 (p = 0..3; ABI 9, was reserved): one output PARITY of `conv3x3(nearest_upsample_2x(x))` (Upsample.forward,
     * openaimodel.py:204-217 / 254-263) evaluated on the LOW-resolution source.  An output pixel (2 oy + py, 2 ox + px), py = p >> 1,
     * px = p & 1, sees only a 2 x 2 neighbourhood of the source: its nine taps collapse onto four with summed weights
     * (ccedit_amd/packing.py:pack_upsample_parities), 4/9 of the multiply-adds.  CONV2D with ksize 2, stride 1, Hout x Wout = Hin x Win =
     * the source frame, M = source pixels; tap (dy, dx) reads source pixel (oy - 1 + py + dy, ox - 1 + px + dx), zeros outside the
     * frame; the result row of source pixel (n, oy, ox) is stored at row (n 2H + 2 oy + py) 2W + 2 ox + px of `out`, which holds
     * 4 M rows.  `pad` / `upsample` are ignored; no gn_stats. */
    int32_t subpix;
    /* optional: LayerNorm folded into a plain Linear / GEGLU projection whose K is not 320 (`to_q(norm(x))`, `ff.net[0].proj(norm(x))`
     * of attention.py:695-716 at 640 / 1280 channels).  A holds the UN-normalised rows, W / bias the gamma / beta-folded weights
     * (as for ln_eps), ln_colsum[N] the row sums of the bf16 weights as packed, ln_stats[M][2] = (mean, rstd) of every row of A
     * (ccedit_row_stats).  The call computes rstd (W' x - mean colsum) + b'.  Persistent eight-phase kernel only (tile 0 / 11-13,
     * Linear, no residual / row bias / gn_stats); refused with CCEDIT_EUNSUPPORTED elsewhere. */
    const float* ln_colsum;
    const float* ln_stats;
    /* ... or ln_sums[M][2] = (sum, sum of squares) of every row of A in double, as a producing call accumulated them through
     * row_sums; then mean = s / Cin, rstd = rsqrt(q / Cin - mean^2 + ln_sums_eps).  Exactly one of ln_stats / ln_sums. */
    const double* ln_sums;
    /* optional, producer side (Linear, no activation, persistent eight-phase kernel; refused elsewhere): the epilogue adds every
     * output row's (sum, sum of squares) of the bf16-rounded values it stores to row_sums[M][2] (double, ZEROED BY THE CALLER) —
     * the LayerNorm statistics of the tensor being written, for the call that consumes it through ln_sums. */
    double* row_sums;
    float ln_sums_eps;
    /* CONV2D, 0 = off (ABI 9, was reserved).  1: the source frames carry their own vertical halo — Hin = the rows the taps may touch
     * (one row above and, for stride 1, one below the rows that produce output; zeros where the frame ends), Hout = output rows — so
     * the vertical padding is one less than `pad` (resp. than the parity's, with subpix; then Hin == Hout + 2).  How a frame whose ROWS
     * are sharded over ranks is convolved after the neighbours' boundary rows were received (ccedit_amd/parallel.py: RowShard;
     * BASELINE.json config 4).  Generic tap-gather kernel only.
     * 2 (ABI 10): the same sharded frame WITHOUT an extended copy — A holds the local rows only (Hin = local rows, padding geometry of
     * the whole frame), and taps one row above / below them read halo_top / halo_bot, bf16 [frames][Win][lda-strided rows] as received
     * from the neighbour ranks; a null pointer means the frame ends there (zeros). */
    int32_t vpad;
    const void* halo_top;
    const void* halo_bot;
    /* optional (ABI 12): the same weights as W once more in MFMA A-FRAGMENT order, for the kernels that keep a weight slice in
     * registers for their whole life (tile 9 / 10 / 14: K = 320 / 640 Linears, 320-channel Conv1d k3).  Block (t, s) — output rows
     * 16 t .. 16 t + 15, k 32 s .. 32 s + 31 of the [rows][Kpad] matrix W — is 1 KB at element offset (t * (Kpad / 32) + s) * 512:
     * 64 lanes x 8 bf16, lane (g = lane >> 4, c = lane & 15) holding W[16 t + c][32 s + 8 g .. + 7]; rows padded to a multiple of 16.
     * A wave then loads a fragment as ONE contiguous kilobyte instead of sixteen 64-byte row pieces: the per-launch weight preload
     * (200-400 KB per workgroup through the CU's L1-miss path) is 12 -> 4 us at 400 KB (tools/exp/lin640w.hip, round 6).
     * null: the kernels read W (same values). */
    const void* Wfrag;
} CcGemmDesc;

int ccedit_gemm(const CcGemmDesc* desc, void* stream);
/* Bytes of CcGemmDesc.workspace this call could use on an MI355X (0: the shape is never split).  No GPU needed. */
int64_t ccedit_gemm_workspace_bytes(const CcGemmDesc* desc);

/* ------------------------------------------------------------------------------------------
 * Fused transformer feed-forward, dim 320 (the 64x96 level of the UNet / ControlNet):
 *     out = x + W2 . GEGLU(W1 . LayerNorm(x) + b1) + b2
 * = `x = self.ff(self.norm3(x)) + x` of BasicTransformerBlock._forward (attention.py:695-716) and
 * `x = self.ff(self.norm2(x)) + x` of BasicTransformerSingleLayerBlock._forward (:758-761), with FeedForward /
 * GEGLU as in attention.py:115-141.  One kernel: x is read once, out written once, the 1280-wide hidden activation
 * stays in registers (ff320.hip).  Replaces ccedit_layernorm + two ccedit_gemm calls.
 *   wstream: the weights as ccedit_amd/packing.py:pack_ff320 lays them out — 42 chunks of 64 KB, one per iteration of the
 *            kernel's three-stage software pipeline over the 40 groups of 32 hidden units: per k-step the two GEMM1
 *            A-fragments (value, gate rows of W1 diag(gamma), group c; bf16, v_mfma_f32_32x32x16 lane order) and one GEMM2
 *            A-fragment of W2 (group c - 2), then float b1' = b1 + W1 beta of group c in accumulator order.
 *   b2p:     b2 in accumulator order, float[10][2][16].
 *   ln = 0:  no normalisation (mean 0, rstd 1; the packer must then be given gamma = 1, beta = 0).
 * ------------------------------------------------------------------------------------------ */
typedef struct CcFf320Desc {
    int64_t M;            /* tokens (rows of x / out) */
    int32_t dim, inner;   /* 320, 1280 */
    int32_t ldx, ldo;     /* row strides in elements (multiples of 8) */
    float eps;            /* LayerNorm eps (1e-5) */
    int32_t ln;           /* 1: LayerNorm folded in (statistics computed in the kernel) */
    const void* x;        /* bf16 [M][ldx] */
    void* out;            /* bf16 [M][ldo]; may NOT alias x (other workgroups' reads are not ordered against the stores) */
    const void* wstream;  /* packed weights, 42 * 65536 bytes (block tail: 46 or 50 chunks, see below) */
    const float* b2p;     /* float[320] in accumulator order */
    void* dbg;            /* null (tuning builds only: per-wave phase cycle counters) */
    /* Block tail (ABI 10; all null / 0 = the plain feed-forward above).  With `a` the kernel computes the whole tail of a dim-320
     * transformer block on the row tile it holds (attention.py:695-716 / 758-761 and the proj_out of :865-889 / :1141-1208):
     *     tok = W_o a + b_o + res;   tok = tok + W2 GEGLU(W1 LayerNorm(tok) + b1) + b2;   out = W_p tok + b_p + res2   (with res2)
     * x is not read.  wstream then is [4 prologue chunks | the 42 feed-forward chunks | 4 epilogue chunks] of 65536 bytes
     * (packing.pack_ff320_tail: 50 A-fragments of W_o / W_p per chunk, rows permuted like W2's); bop / bpp: b_o / b_p in accumulator
     * order like b2p.  res2 requires a.  tok and the feed-forward output are rounded to bf16 where the separate launches store them. */
    const void* a;        /* bf16 [M][lda]: input of the prologue projection (the attention output) */
    const void* res;      /* bf16 [M][ldr]: residual added to the prologue projection */
    const void* res2;     /* bf16 [M][ldr2]: residual added to the epilogue projection (null: no epilogue GEMM, out = the feed-forward's) */
    const float* bop;     /* float[320] */
    const float* bpp;     /* float[320] */
    int32_t lda, ldr, ldr2, pad_;
} CcFf320Desc;

int ccedit_ff320(const CcFf320Desc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation (fp32 statistics, bf16 in/out)
 * ------------------------------------------------------------------------------------------ */
/* GroupNorm(32, C) over (C/32 x H x W) per frame, optional fused SiLU.
 * Replaces nn.GroupNorm + nn.SiLU of openaimodel.py:441-444, 479-482 (eps 1e-5), attention.py:153-156,
 * model.py:50-53 (eps 1e-6).  x: [frames][hw][C]; stats workspace: double[frames*32*2] (zeroed here). */
int ccedit_groupnorm_spatial(const void* x, void* y, const float* gamma, const float* beta,
                             double* stats_ws, int32_t frames, int32_t hw, int32_t C, float eps,
                             int32_t silu, void* stream);
/* The apply half alone, for statistics that a producer already accumulated (CcGemmDesc.gn_stats):
 * stats: double[frames][32][2] = (sum, sum of squares) over the C/32 x hw elements of each group (double: the
 * producers' workgroups meet in global atomics, and in double their arrival order cannot move the fp32 mean / rstd —
 * results are reproducible from run to run). */
int ccedit_groupnorm_spatial_apply(const void* x, void* y, const float* gamma, const float* beta,
                                   const double* stats, int32_t frames, int32_t hw, int32_t C, float eps,
                                   int32_t silu, void* stream);
/* The statistics half alone: adds (sum, sum of squares) of every (frame, group) to stats = double[frames][32][2], ZEROED BY THE CALLER —
 * for frames whose rows are sharded over ranks (RowShard): local sums, all-reduced by the caller, then ccedit_groupnorm_spatial_apply
 * with the sums divided by the number of ranks (its element count is the LOCAL hw). */
int ccedit_groupnorm_spatial_stats(const void* x, double* stats, int32_t frames, int32_t hw, int32_t C, void* stream);
/* GroupNorm(32, C) over (C/32 x T) per pixel — the normalization() / norm_temporal applied to the
 * '(b h w) c t' view (openaimodel.py:617-619, 674-676; attention.py:1085, 1176).
 * x: [B*T][hw][C]; one statistics group = T frames x C/32 channels at one pixel. */
int ccedit_groupnorm_temporal(const void* x, void* y, const float* gamma, const float* beta,
                              int32_t B, int32_t T, int32_t hw, int32_t C, float eps, int32_t silu,
                              void* stream);
/* The same normalisation in two phases for frame-sharded clips: per-(clip, pixel, group) partial sum / sum-of-
 * squares over the LOCAL frames (stats: float[B*hw*32*2], fully overwritten), all-reduced by the caller across the
 * ranks that hold the other frames, then applied with count = (C/32) * T_global.  The output may be written into
 * a halo-extended buffer: frame t of clip b goes to frame index b*dst_frames + t + dst_off. */
int ccedit_groupnorm_temporal_stats(const void* x, float* stats, int32_t B, int32_t T, int32_t hw, int32_t C,
                                    void* stream);
int ccedit_groupnorm_temporal_apply(const void* x, void* y, const float* gamma, const float* beta,
                                    const float* stats, int32_t B, int32_t T, int32_t hw, int32_t C,
                                    float count, float eps, int32_t silu, int32_t dst_frames, int32_t dst_off,
                                    void* stream);
/* LayerNorm over C (eps 1e-5) — attention.py:667-669, 755-756. x,y: [rows][C] */
/* (mean, rstd = rsqrt(var + eps)) of every row of x[rows][C] (bf16, contiguous), fp32 pairs: the statistics half of
 * nn.LayerNorm (attention.py:611-613), consumed by CcGemmDesc.ln_stats.  Same arithmetic as ccedit_layernorm. */
int ccedit_row_stats(const void* x, float* stats, int64_t rows, int32_t C, float eps, void* stream);
int ccedit_layernorm(const void* x, void* y, const float* gamma, const float* beta, int64_t rows,
                     int32_t C, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention: softmax(q k^T * scale) v, bf16 MFMA, online softmax, fp32 accumulate.
 * Replaces F.scaled_dot_product_attention in CrossAttention.forward (attention.py:392-467) for
 *   - spatial self-attention      (Lq = Lk = h*w, one batch per frame)
 *   - text cross-attention        (Lk = 77, K/V shared by the T frames of a clip: kv_batch = batch / kv_div)
 *   - temporal self-attention     (Lq = Lk = T per pixel: rows of one sequence are H*W apart)
 * q/k/v/o are [rows][ld] matrices with heads side by side: head h occupies columns [h*d, (h+1)*d).
 * Row of (batch, i) = (batch / inner) * outer_rows + (batch % inner) * inner_rows + i * seq_rows.
 * ------------------------------------------------------------------------------------------ */
typedef struct CcAttnDesc {
    const void* q; const void* k; const void* v; void* o;
    int32_t ldq, ldk, ldv, ldo;       /* row strides in elements */
    int32_t heads, d;                 /* d in {40, 80, 160} or any multiple of 8 up to 160 */
    int32_t batches;                  /* number of (sequence) batches */
    int32_t Lq, Lk;
    int32_t q_inner; int64_t q_outer_rows, q_inner_rows, q_seq_rows;
    int32_t kv_div;                   /* kv batch = batch / kv_div (text K/V shared across frames) */
    int32_t kv_inner; int64_t kv_outer_rows, kv_inner_rows, kv_seq_rows;
    float scale;                      /* d^-0.5 */
    /* optional leading KV segment (anchor-frame keys of SpatialTransformer3DCA, attention.py:1324-1336:
     * context = cat([anchor tokens, x tokens])): keys [0, seg1_len) come from the kv batch
     * (batch / seg1_div) * seg1_mul + seg1_add, keys [seg1_len, Lk) from the regular kv batch.  0 = unused. */
    int32_t seg1_len, seg1_div, seg1_mul, seg1_add;
    int32_t causal;       /* 1: key j is visible to query i only if j <= i (CLIP text encoder); Lq == Lk, no segment */
    int32_t flags;        /* CCEDIT_ATTN_*; 0 = none (ABI 9; was reserved) */
} CcAttnDesc;

/* q already carries scale * log2(e) (folded into the to_q weights by the packer, one rounding instead of two): kernels then take
 * p = 2^(q.k - reference) without multiplying, and `scale` is ignored */
#define CCEDIT_ATTN_Q_LOG2 1

int ccedit_attention(const CcAttnDesc* desc, void* stream);

/* ------------------------------------------------------------------------------------------
 * Layout / elementwise helpers (all HBM-bound)
 * ------------------------------------------------------------------------------------------ */
/* fp32 (B, C, T, H, W) -> bf16 [B*T][H*W][Cpad] (zero-filled pad channels), y = x*scale + shift.
 * Used for the latent (`x * c_in`, denoiser.py:40) and for the hint remap 1-(h+1)/2 (wrappers.py:160-162). */
int ccedit_ncthw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W,
                         int32_t Cpad, const float* scale_per_b, float scale, float shift, void* stream);
/* bf16/fp32 [B*T][H*W][ld] (first C columns) -> fp32 (B, C, T, H, W) */
int ccedit_nhwc_to_ncthw(const void* x, int32_t x_is_f32, int32_t ld, float* y, int32_t B, int32_t C,
                         int32_t T, int32_t H, int32_t W, void* stream);
/* out[:, :C1] = a ; out[:, C1:C1+C2] = b + c   (cat([h, hs.pop() + control.pop()]), controlmodel.py:543) */
int ccedit_cat_add(const void* a, const void* b, const void* c, void* out, int64_t rows, int32_t C1,
                   int32_t C2, void* stream);
/* The same, additionally accumulating the GroupNorm(32, C1+C2) statistics of `out` ([frames][hw][C1+C2]) into
 * stats = double[frames][32][2] (sum, sum of squares; ZEROED BY THE CALLER) for ccedit_groupnorm_spatial_apply —
 * the decoder ResBlock's in_layers.0 (openaimodel.py:441-444) then does not re-read the concatenation. */
int ccedit_cat_add_gn(const void* a, const void* b, const void* c, void* out, double* stats, int32_t frames,
                      int32_t hw, int32_t C1, int32_t C2, void* stream);
/* Row-block copy for the layout transposition of a frame-sharded clip (ccedit_amd/parallel.py: FrameShard.to_pixels / to_frames;
 * BASELINE.json config 4 — the reference has no multi-GPU path, scripts/sampling/sampling_tv2v.py:106 is a single .to("cuda")):
 * for s < n_blocks, rows [blocks[3s], + blocks[3s+2]) of src go to rows [blocks[3s+1], ...) of dst; with `add` (bf16 rows laid out
 * like dst) dst = src + add in fp32 with one rounding — the unpack of an all-to-all with the ResBlock skip folded in.
 * blocks: int64 [n_blocks][3] in DEVICE memory; max_rows = the largest block (grid sizing); row_bytes % 16 == 0. */
int ccedit_copy_row_blocks(const void* src, void* dst, const void* add, const int64_t* blocks, int32_t n_blocks,
                           int64_t max_rows, int32_t row_bytes, void* stream);
/* n_blocks strided 2-D copies sharing one geometry (rows x row_bytes, byte pitches) and differing in their byte offsets:
 * blocks = int64 [n_blocks][2] (src offset, dst offset) in DEVICE memory.  The column-slice halves of the head-parallel attention
 * exchange of a row-sharded clip (ccedit_amd/parallel.py: RowShard.to_heads / from_heads; BASELINE.json config 4).
 * row_bytes, pitches and offsets are multiples of 16. */
int ccedit_copy_2d_blocks(const void* src, void* dst, const int64_t* blocks, int32_t n_blocks, int64_t rows, int32_t row_bytes,
                          int64_t src_pitch, int64_t dst_pitch, void* stream);
/* y = a + b (bf16), n elements — `h = h + control.pop()` (controlmodel.py:537), `h += guided_hint` (:300) */
int ccedit_add(const void* a, const void* b, void* y, int64_t n, void* stream);
/* y = silu(x), bf16 elementwise (out_temporal's leading nn.SiLU, openaimodel.py:1627-1632) */
int ccedit_silu(const void* x, void* y, int64_t n, void* stream);
/* timestep_embedding (diffusionmodules/util.py:244-268): out bf16 [n][ld], [cos | sin], dim even */
int ccedit_timestep_embedding(const int64_t* t, void* out, int32_t n, int32_t dim, int32_t ld, void* stream);

/* p = softmax(s * scale) row-wise, fp32 in -> bf16 out, columns [cols, cols_pad) zero-filled.
 * The single-head d=512 attention of the VAE mid block (model.py:180-195) is evaluated as
 * GEMM (q k^T) -> this -> GEMM (p v); cols_pad <= 8192. */
int ccedit_softmax_rows(const float* s, void* p, int64_t rows, int32_t cols, int32_t cols_pad, int64_t lds,
                        int64_t ldp, float scale, void* stream);

/* Token + position embedding lookup of the CLIP text encoder (transformers CLIPTextEmbeddings, called from
 * FrozenCLIPEmbedder.forward, encoders/modules.py:393-413): out[b*L + i] = tok[ids[b*L + i]] + pos[i]  (bf16 out).
 * ids: int64 [rows]; tok: fp32 [vocab][C]; pos: fp32 [L][C].  ids are clamped to [0, vocab): validate on the host. */
int ccedit_embedding_lookup(const int64_t* ids, const float* tok, const float* pos, void* out, int64_t rows, int32_t L,
                            int32_t C, int32_t vocab, void* stream);

/* Posterior sample of the KL-VAE encoder (DiagonalGaussianDistribution.sample, distributions.py:24-41, called from
 * AutoencoderKLInferenceWrapper.encode, autoencoder.py:323-332): moments = fp32 [frames*hw][ldm] channels-last rows
 * [mean(zc) | logvar(zc)] (quant_conv output); noise, out = fp32 (frames, zc, h, w);
 * out = scale * (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise).  The caller supplies the noise (the reference
 * draws it with torch.randn on the CPU global generator). */
int ccedit_gaussian_sample(const float* moments, const float* noise, float* out, int64_t frames, int32_t zc,
                           int32_t hw, int32_t ldm, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp32 first-stage model (ABI 11).  The reference decodes with autocast disabled (sgm/models/diffusion.py:151-156): fp32 operands,
 * products and tensors.  These three entry points evaluate the KL-VAE in that arithmetic class: fp32 tensors in and out, fp32 sums.
 * ccedit_gemm_f32 has two realisations of the same fp32 contraction (policy `f32_split`): 1 (default, round 6) — every fp32 operand is
 * split EXACTLY into three bf16 numbers and an fp32 product is formed as six exact bf16 x bf16 products on v_mfma_f32_32x32x16_bf16 (the
 * dropped terms are <= 2^-25 of the product, below the rounding of any fp32 accumulation step); 0 — v_mfma_f32_32x32x2_f32; 2 — as 1 but only
 * the eight-wave kernel.  Same descriptor, same results to fp32 rounding (tests/test_vae_f32_gpu.py holds both to the same tolerances).
 * The engine selects the fp32 first stage from the yaml (ccedit_amd/vae.py, policy `vae_fp32`).
 *
 * ccedit_gemm_f32: out[m][n] = bias[n] + sum_k W[n][k] * src(A)[m][k] (+ res[m][n]); everything fp32, channels-last rows.
 *   mode 0  Linear / Conv 1x1: A [M][lda], k = channel < Cin (Cin % 4 == 0); W [N][ldw] with Kpad = Cpad = Cin rounded up to 16,
 *           columns [Cin, Kpad) ZERO (model.py:161-201 q/k/v/proj_out, nin_shortcut, quant / post_quant convs; the attention's
 *           q k^T and p v products with the other tensor in the role of W)
 *   mode 1  Conv2d 3x3: A [frames][Hin][Win][lda]; W [N][9][Cpad] (tap-major, tap = 3 ky + kx, columns [Cin, Cpad) zero);
 *           output pixel (y, x) reads source (y stride - pad + ky, x stride - pad + kx), zeros outside; Hout / Wout are given (the
 *           encoder's Downsample pads right / bottom only, model.py:74-93); upsample = 1: the taps walk over the nearest-2x
 *           upsampled source (2 Hin x 2 Win) without materialising it (model.py:56-71).  M = frames * Hout * Wout.
 *           upsample = 2 + 2 py + px (round 6, policy f32_split != 0 only): ONE OUTPUT PARITY of the same `conv3x3(nearest_upsample_2x(x))`
 *           as a 2 x 2 convolution on x itself — W [N][4][Cpad] holds the merged taps (ccedit_amd/vae_f32.py: pack_f32_parities),
 *           Hout x Wout = Hin x Win, stride 1, pad 1, no residual, M = frames * Hin * Win; the row of source pixel (y, x) is written to
 *           pixel (2 y + py, 2 x + px) of `out` = the 2 Hin x 2 Win frames.  Four calls fill the tensor at 4/9 of the multiply-adds.
 * A and W 16-byte aligned, lda % 4 == 0, ldw % 4 == 0; out / res any row stride >= N (16-byte accesses when a multiple of 4).
 * Fixed summation order: repeated calls are bit-identical. */
typedef struct CcGemmF32Desc {
    const float* A;
    const float* W;
    const float* bias; /* [N] or NULL */
    const float* res;  /* [M][ldr] or NULL */
    float* out;        /* [M][ldc] */
    int64_t M;
    int32_t N, Cin, Cpad, Kpad;
    int32_t lda, ldw, ldc, ldr;
    int32_t mode;
    int32_t Hin, Win, Hout, Wout, stride, pad, upsample;
} CcGemmF32Desc;
int ccedit_gemm_f32(const CcGemmF32Desc* desc, void* stream);
/* y = GroupNorm(32, C, eps)(x) [then SiLU when silu != 0] per frame over (hw x C / 32); x, y fp32 [frames][hw][C] (y may alias x),
 * C in {32, 64, 128, 256, 512, 1024}; stats = fp64 scratch [frames][32][2] (sum, sum of squares; zeroed here).  model.py:45-53. */
int ccedit_groupnorm_f32(const float* x, float* y, const float* gamma, const float* beta, double* stats, int32_t frames, int32_t hw,
                         int32_t C, float eps, int32_t silu, void* stream);
/* s[r][0..cols) = softmax(scale * s[r][0..cols)) in place, fp32 rows of stride ld; columns beyond cols are left alone
 * (rows of up to 8192 columns are held in registers, longer ones are re-read: three passes) */
int ccedit_softmax_rows_f32(float* s, int64_t rows, int32_t cols, int64_t ld, float scale, void* stream);

/* Sampler / guider / denoiser elementwise math on the fp32 latent (417,792 elements at 17x64x96):
 *   ccedit_cfg_denoise : den = x + (-sigma) * (eps_u + scale*(eps_c - eps_u))        [denoiser.py:40 with
 *                        c_skip=1, c_out=-sigma; guiders.py:25-29]   eps = fp32 [2][n]
 *   ccedit_axpby       : y = a*x + b*z                                                [sampling.py:388, 398-402]
 *   ccedit_mask_blend  : y = x*m + z*(1-m), all fp32 [n]                              [sampling.py:150-153, 213-216]
 */
int ccedit_mask_blend(const float* x, const float* z, const float* mask, float* y, int64_t n, void* stream);
int ccedit_cfg_denoise(const float* x, const float* eps2, float* den, int64_t n, float sigma, float scale,
                       void* stream);
int ccedit_axpby(const float* x, const float* z, float* y, int64_t n, float a, float b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CCEDIT_HIP_H */
