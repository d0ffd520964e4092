/*
 * photoverse_hip.h - C-ABI of libphotoverse_hip.so, the MI355X (gfx950) kernels behind the
 * PhotoVerse denoising hot path.
 *
 * The reference (idonahum/photoVerse) is pure Python on top of diffusers and has no FFI of
 * its own.  Its "operators" for this path are the vendor kernels the
 * eager PyTorch graph dispatches to.  Each entry point below replaces one family of those
 * dispatches; the call site in the reference that reaches it is cited per function.  The
 * Python host layer (photoverse_amd/) binds these with ctypes and mirrors the reference's
 * Python interfaces (attention-processor protocol, UNet call, run_inference).
 *
 * Conventions
 *   - plain C, raw device pointers, caller-owned outputs and workspace;
 *   - no allocation, no host synchronisation, safe under HIP stream capture;
 *   - every launcher returns a hipError_t value as int (0 = success) and enqueues on `stream`
 *     (a hipStream_t passed as void*);
 *   - activations are NHWC / token-major fp16: a (B,C,H,W) tensor of the reference is stored
 *     as rows [B*H*W][C] with an explicit row stride `ld*` in elements;
 *   - GEMM weights are fp16 [N][K] (torch Linear layout); 3x3 conv weights are repacked to
 *     [Cout][ky][kx][Cin];  biases / norm affine parameters are fp32.
 */
#ifndef PHOTOVERSE_HIP_H
#define PHOTOVERSE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_ABI_VERSION 17

enum pv_act { PV_ACT_NONE = 0, PV_ACT_SILU = 1, PV_ACT_QUICK_GELU = 2, PV_ACT_LEAKY_RELU = 3, PV_ACT_GELU = 4 };

int pv_abi_version(void);
/* number of HIP devices visible, or -1 on runtime error; never initialises a context */
int pv_device_count(void);

/* ------------------------------------------------------------------------------------------
 * pv_gemm_conv: out[M][N] = epilogue( A[M][K] * W[N][K]^T ), MFMA f16 -> f32 accumulate.
 *   taps == 1 : Linear / 1x1 conv.  Replaces nn.Linear / Conv2d(k=1) dispatches of
 *               attention_processor.py:297,304-305,392-393,423 (to_q/to_k/to_v/to_k_ip/
 *               to_v_ip/to_out), adapters.py:14-28, and the [EXT] diffusers proj_in/proj_out/
 *               GEGLU/FF/time-embedding/conv_shortcut layers reached from infer.py:103-114.
 *   taps == 9 : implicit-GEMM 3x3 conv, zero padding, stride 1|2, optional nearest x2 upsample
 *               folded into the gather (ResnetBlock2D conv1/conv2, Downsample2D, Upsample2D).
 *   A may come from two tensors concatenated along channels (skip connections): channels
 *   [0,c0) from a0 and [c0,c0+c1) from a1; c0, c1 multiples of 64.
 *   epilogue: (+bias[n]) (+rowadd[img][n]) act (+residual[m][n]) -> fp16 (or fp32) store.
 *   Requirements: N % 128 == 0 or N % 160 == 0; (c0+c1) % 64 == 0; pointers 16-byte aligned; each operand < 2 GiB.
 */
typedef struct pv_gemm_params {
    const void* a0;        /* fp16 */
    const void* a1;        /* fp16 or NULL */
    int32_t c0, c1;        /* channels taken from a0 / a1 */
    int32_t lda0, lda1;    /* row strides in elements */
    const void* w;         /* fp16 [N][taps*(c0+c1)] */
    const float* bias;     /* [N] or NULL */
    const float* rowadd;   /* fp32 [images][rowadd_ld] added per image (time embedding) or NULL */
    int32_t rowadd_ld;     /* 0 => one row shared by all images */
    const void* residual;  /* fp16 [M][ldr] or NULL */
    int32_t ldr;
    void* out;             /* fp16 [M][ldc] (fp32 when out_f32) */
    int32_t ldc;
    int32_t M, N;
    int32_t taps;          /* 1 or 9 */
    int32_t batch, hin, win, hout, wout; /* taps==9 geometry; taps==1: hout*wout = rows per image */
    int32_t stride;        /* 1 or 2 */
    int32_t upsample;      /* 1 => logical input is the x2 nearest upsample of (hin,win) */
    int32_t pad;           /* taps==9: zero padding in front (top / left): 1 = Conv2d(padding=1) (every UNet / decoder conv);
                              0 = the VAE encoder's Downsample2D, F.pad(x,(0,1,0,1)) + Conv2d(stride=2,padding=0) (stride 2
                              only).  Reads past the bottom / right edge are zeros in both cases (hout, wout say how far). */
    int32_t act;           /* enum pv_act */
    int32_t out_f32;
    int32_t geglu;         /* 1 => W rows are tile-interleaved (value|gate); out[M][N/2] = value*gelu(gate) */
    int32_t splitk;        /* > 1: split the K loop over this many workgroups per tile (small-M layers); needs splitk_ws */
    float* splitk_ws;      /* fp32 workspace [splitk][M][N] for the partial slabs, reduced in fixed order */
    float* colstats;       /* optional fp32 [ceil(M/64)][2][N]: per 64-row block and output column, the sum and the sum of
                              squares of the (fp16-rounded) outputs - GroupNorm statistics of the tensor being written,
                              consumed by pv_groupnorm_stats_from_colstats.  NULL = off.  fp16 output, no geglu (with splitk > 1 the
                              reduce launch produces them). */
    const float* ln_rowsum; /* optional (ABI 12) fp32 [N]: sum_k w[n][k] of the fp16 weights.  Non-NULL = the launch computes
                              epilogue( LayerNorm_noaffine(a0) . w^T + bias ): BasicTransformerBlock.norm1 / norm3 [EXT] folded into the Linear that
                              follows it - the GEMM runs on the RAW rows and the epilogue applies rstd * (acc - mean * ln_rowsum[n]); the caller
                              folds the affine part (gamma scales the columns of w, w . beta joins the bias).  Row statistics come from the
                              fragments the MFMAs consume.  Only the 256-row-tile Linear path takes it (taps == 1, K >= 640, c1 == 0, no split-K,
                              N % 320 == 0 or geglu with N % 256 == 0, >= 256 tiles): anything else returns hipErrorInvalidValue. */
    float ln_eps;
    int32_t big_tile_min;  /* (ABI 12) 256-row-tile path (pv_convbig.hip): minimum number of 256-row tiles (x K slices) the launch must have to take it.
                              0 = library default (256 = one workgroup per CU; env PV_CONV_BIG overrides), < 0 = never.  A caller that runs two such
                              launches side by side on two streams (the uncond / cond forwards of a CFG step) passes 128: each fills half the chip. */
    const float* a_norm;   /* (ABI 15) optional fp32 [batch][2][c0+c1]: per image and INPUT channel (a0's channels, then a1's) the scale and the shift of a
                              GroupNorm over the conv's input, from pv_groupnorm_scale_shift.  Non-NULL = the launch computes conv( act_a( a * scale + shift ) ):
                              [EXT] ResnetBlock2D.norm1 / norm2 + SiLU folded into conv1 / conv2 - the conv reads the RAW tensor(s), each input pixel is
                              normalised once, in LDS, where the LDS-resident input patch of the 256-row tile holds it (zero padding stays zero), the same
                              fp32 arithmetic and fp16 rounding as pv_groupnorm_apply: results equal the two-launch path bit for bit.  Only that path takes it
                              (taps == 9, stride 1, no upsampling, 64- or 32-pixel image rows, no split-K): anything else returns hipErrorInvalidValue. */
    int32_t a_norm_act;    /* PV_ACT_NONE or PV_ACT_SILU, applied after the affine part */
} pv_gemm_params;
int pv_gemm_conv(const pv_gemm_params* p, void* stream);
/* (ABI 17) Which kernel pv_gemm_conv would launch for this block: the symbol as rocprofv3 prints it (without the namespace and the argument
 * list, e.g. "big_tile_kernel<true, false, 8, 3, false>") and the workgroup count (split-K slices included).  Same validation and the same dispatch
 * code as pv_gemm_conv with the launchers in describe-only mode: no HIP call, no GPU needed.  The host side tags its recorded launches with it
 * (bench.py's per-symbol roofline reads the tags); nothing in the reference corresponds (measurement scaffolding of this build). */
int pv_gemm_conv_kernel_info(const pv_gemm_params* p, char* name, int32_t name_len, int64_t* workgroups);

/* ------------------------------------------------------------------------------------------
 * GroupNorm(32 groups) over NHWC fp16, optionally over a channel concat of two tensors.
 * Replaces F.group_norm (+ SiLU) dispatches of [EXT] ResnetBlock2D.norm1/norm2,
 * Transformer2DModel.norm, conv_norm_out.
 *   pv_groupnorm_stats : partial (sum, sumsq) per (image, split, group) -> partial[B][S][G][2], then reduced in a
 *                        fixed order to (mean, rstd) stored at partial[b][0][g][0..1] (deterministic, no atomics)
 *   pv_groupnorm_apply : y = act(gamma*(x-mean)*rstd+beta), contiguous fp16 [B*HW][C]; must follow _stats
 */
typedef struct pv_groupnorm_params {
    const void* x0; const void* x1;  /* fp16 sources; x1 may be NULL */
    int32_t c0, c1, ld0, ld1;
    int32_t batch, hw, groups;
    int32_t splits;                  /* S: pixel splits per image */
    float* partial;                  /* [B][S][G][2] fp32 workspace */
    const float* gamma; const float* beta;
    float eps;
    int32_t act;                     /* PV_ACT_NONE or PV_ACT_SILU */
    void* y;                         /* fp16 [B*HW][c0+c1] */
    const float* colstats0;          /* pv_gemm_params.colstats of the launches that produced x0 / x1 (ld == c), or NULL */
    const float* colstats1;
} pv_groupnorm_params;
int pv_groupnorm_stats(const pv_groupnorm_params* p, void* stream);
/* same result as pv_groupnorm_stats, from the column statistics the producing GEMM epilogues left behind (no pass over x);
 * needs hw % 64 == 0, colstats0 (and colstats1 when c1 > 0); fixed reduction order */
int pv_groupnorm_stats_from_colstats(const pv_groupnorm_params* p, void* stream);
/* (ABI 15) the same statistics turned into the per-(image, channel) affine form a consumer can apply by itself:
 * table[b][0][c] = gamma[c] * rstd(b, group(c)), table[b][1][c] = beta[c] - mean(b, group(c)) * table[b][0][c]   (fp32 [batch][2][c0+c1]),
 * so that pv_groupnorm_apply's y = act(x * table[b][0][c] + table[b][1][c]) can be folded into the conv that follows (pv_gemm_params.a_norm).
 * Needs what pv_groupnorm_stats_from_colstats needs; also leaves (mean, rstd) in partial like it.  Fixed reduction order. */
int pv_groupnorm_scale_shift(const pv_groupnorm_params* p, float* table, void* stream);
int pv_groupnorm_apply(const pv_groupnorm_params* p, void* stream);

/* LayerNorm over the last dim of fp16 rows (BasicTransformerBlock.norm1-3, CLIP layer norms,
 * adapters.py:15,18 with LeakyReLU fused).  y = act(gamma*(x-mean)*rstd+beta) */
typedef struct pv_layernorm_params {
    const void* x; int32_t ldx;
    void* y; int32_t ldy;
    const float* gamma; const float* beta;
    int32_t rows, cols;              /* cols % 8 == 0, cols <= 4096 */
    float eps;
    int32_t act;
} pv_layernorm_params;
int pv_layernorm(const pv_layernorm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * Flash-style self attention (stock AttnProcessor2_0 on attn1, unet.py:20-24; CLIP encoder
 * self-attention).  q,k,v are column slices of one fp16 buffer (fused QKV GEMM output):
 * head h uses columns [h*d,(h+1)*d) of each.  softmax scale = 1/sqrt(d).  d in {40,64,80,160}.
 */
typedef struct pv_attn_params {
    const void* q; const void* k; const void* v;
    int32_t ldq, ldk, ldv;
    void* out; int32_t ldo;
    int32_t batch, heads, nq, nk, d;
    int32_t causal;
    float* lse;                      /* optional fp32 [B][heads][nq]: log-sum-exp of the scaled scores in log2 units, kept for
                                        pv_attention_backward; NULL = not written */
} pv_attn_params;
int pv_attention(const pv_attn_params* p, void* stream);
/* (ABI 17) Which kernel pv_attention would launch for this block (pv_gemm_conv_kernel_info's contract): "attn8_kernel<497>", "attn_kernel<80, 2, false>", ... */
int pv_attention_kernel_info(const pv_attn_params* p, char* name, int32_t name_len, int64_t* workgroups);

/* Backward of pv_attention (the [EXT] AttnProcessor2_0 SDPA of attn1, models/unet.py:20-24, and the CLIP text layers the gradient
 * of text_adapter crosses, train.py:498-500).  Probabilities are recomputed from lse; delta: fp32 workspace [B][heads][nq]; qs: fp16 workspace like q.
 * dq / dk / dv: fp16 rows (may be column slices of one [dq | dk | dv] buffer).  d in {40, 64, 80, 160}.  Deterministic. */
typedef struct pv_attn_bwd_params {
    const void* q; const void* k; const void* v;
    int32_t ldq, ldk, ldv;
    const void* out; int32_t ldo;    /* the forward output O */
    const void* dout; int32_t lddo;
    const float* lse;
    float* delta;
    void* qs; int32_t ldqs;          /* fp16 workspace [B*nq][heads*d]: the scaled queries, written by the call */
    void* dq; void* dk; void* dv;
    int32_t lddq, lddk, lddv;
    int32_t batch, heads, nq, nk, d;
    int32_t causal;
    void* ws; int64_t ws_bytes;      /* optional workspace: with >= 2 * batch * heads * nq * 96 bytes (d = 40; * 224 at d = 80) the unmasked launches
                                      * with nq, nk % 512 == 0 (d = 40) / % 256 == 0 (d = 80) take the 8-wave staggered kernels (pv_attnbwd.hip);
                                      * NULL = the 4-wave kernels */
} pv_attn_bwd_params;
int pv_attention_backward(const pv_attn_bwd_params* p, void* stream);

/* GroupNorm (+ SiLU) backward over NHWC fp16 ([EXT] ResnetBlock2D.norm1/2, Transformer2DModel.norm, conv_norm_out): data gradient
 * only (the affine parameters are frozen in train.py).  stats: the forward's pv_groupnorm_params.partial after the statistics launch
 * ((mean, rstd) at stats[b * stats_stride + g * 2]); partial: fp32 [B][splits][2][C]; sums: fp32 [B][groups][2].
 * dx0 / dx1 receive the gradient of x0 / x1 (+ add0 / add1 when given - gradient accumulation of a tensor with several consumers);
 * a NULL dx is skipped. */
typedef struct pv_groupnorm_bwd_params {
    const void* x0; const void* x1;
    int32_t c0, c1, ld0, ld1;
    int32_t batch, hw, groups, splits;
    const float* stats; int32_t stats_stride;
    const float* gamma; const float* beta;
    int32_t act;
    const void* dy; int32_t ld_dy;
    float* partial; float* sums;
    void* dx0; int32_t ld_dx0; const void* add0; int32_t ld_add0;
    void* dx1; int32_t ld_dx1; const void* add1; int32_t ld_add1;
} pv_groupnorm_bwd_params;
int pv_groupnorm_backward(const pv_groupnorm_bwd_params* p, void* stream);
/* GEGLU backward: h = [value | gate] fp16 [rows][2n] (pre-activation of GEGLU.proj), dy [rows][n] -> dh [rows][2n] */
int pv_geglu_backward(const void* h, int32_t ldh, const void* dy, int32_t lddy, void* dh, int32_t lddh, int32_t rows, int32_t n, void* stream);
/* dx = dy * act'(x) for a saved pre-activation x (quick-GELU of the CLIP MLP, SiLU, LeakyReLU, GELU) */
int pv_act_backward(const void* x, int32_t ldx, const void* dy, int32_t lddy, void* dx, int32_t lddx, int32_t rows, int32_t cols, int32_t act,
                    void* stream);
/* y = act(x) as its own pass (the training forward keeps the pre-activation) */
int pv_act_forward(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t rows, int32_t cols, int32_t act, void* stream);
/* Dropout on the input of a LoRA branch (peft LoRA layer: result += B(A(dropout(x))) * scaling; train.py:265-270).  Counter-based
 * (Philox4x32-10): the keep mask is a function of (rng key, site, rng[2] = iteration, copy, element), recomputed by the backward.
 * forward: out[r][k*cols + c] = x[r][c] * keep_k / (1-p), k < copies (several independently masked copies side by side);
 * backward: out[r][c] = sum_k x[r][k*cols + c] * keep_k / (1-p) (+ add).  rng: the {key lo, key hi, counter} block of pv_fusion_draw. */
int pv_dropout_f16(const void* x, int32_t ldx, void* out, int32_t ldo, const void* add, int32_t ldadd, int32_t rows, int32_t cols, int32_t copies,
                   float p, const int32_t* rng, int32_t site, int32_t backward, void* stream);
/* ---- the ArcFace identity loss (models/loss.py, models/arcface_resnet.py; train.py:521-535) and the VAE-decoder backward under it ---- */
/* y = x * scale[c] + shift[c] over fp16 rows: an eval-mode BatchNorm in FRONT of a zero-padded conv (IRBlock.bn0, bn4, arcface_resnet.py:18,82);
 * shift == NULL: its backward */
int pv_col_affine_f16(const void* x, int32_t ldx, const float* scale, const float* shift, void* y, int32_t ldy, int32_t rows, int32_t cols, void* stream);
/* nn.PReLU() with one slope (device scalar): dy == NULL: out = prelu(x); else out = dy * prelu'(x) */
int pv_prelu_f16(const void* x, int32_t ldx, const void* dy, int32_t lddy, const float* slope, void* out, int32_t ldo, int32_t rows, int32_t cols,
                 void* stream);
/* nn.MaxPool2d(2, 2) over NHWC fp16 (B,h,w,c): dy == NULL: out (B,h/2,w/2,c); else out (B,h,w,c) = dy routed to each window's first maximum */
int pv_maxpool2x2(const void* x, const void* dy, void* out, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* FaceLoss.preprocess (loss.py:26-62): y (B,1,size,size) = bilinear(align_corners=False) of gray(x) * mul + add; x fp32 (B,3,h,w) with the given
 * image stride (elements).  _backward: dx (B,3,h,w) contiguous from dy (B,1,size,size) with its image stride; deterministic gather */
int pv_gray_resize(const float* x, int64_t x_image_stride, float* y, int32_t batch, int32_t h, int32_t w, int32_t size, float mul, float add, void* stream);
int pv_gray_resize_backward(const float* dy, int64_t dy_image_stride, float* dx, int32_t batch, int32_t h, int32_t w, int32_t size, float mul, void* stream);
/* nn.CosineEmbeddingLoss (margin 0) per sample: target > 0: 1 - cos(e1, e2), else max(0, cos); e1 / e2 fp16 [batch][dim];
 * de2 (optional, fp16) = gscale / batch * d(loss_b)/d(e2_b) - the gradient of the batch MEAN times gscale */
int pv_cosine_embedding_loss(const void* e1, const void* e2, int32_t batch, int32_t dim, float target, float gscale, float* per_sample, void* de2,
                             void* stream);
/* in-place backward of pv_softmax_rows: dp <- scale * p * (dp - sum_j p_j dp_j) per row */
/* Weight gradient of a Linear layer (loss.backward() reaching a trainable weight, train.py:536): partial slabs
 * out[s][n][k] (fp32, s < nsplit) = sum over rows m in [s * rows_per_split, min(m, (s + 1) * rows_per_split)) of dy[m][n] * x[m][k] -
 * both operands fp16 row matrices as the forward / data-gradient chain leaves them (no transposed copies; the MFMA operands are read with the
 * transposing LDS load).  n, k, lddy, ldx multiples of 8; rows_per_split a multiple of 64.  Sum the slabs in s order with pv_reduce_blocks
 * (nblk = nsplit, inner = n * k): deterministic. */
int pv_wgrad_tn(const void* dy, int32_t lddy, const void* x, int32_t ldx, int32_t m, int32_t n, int32_t k, float* out, int32_t nsplit,
                int32_t rows_per_split, void* stream);
int pv_softmax_rows_backward(const void* p, int32_t ldp, void* dp, int32_t lddp, int32_t rows, int32_t cols, float scale, void* stream);
/* out = dy where lo < y < hi else 0: gradient of images.clamp(-1, 1) (infer.py:122) */
int pv_clamp_mask_f32(const float* y, const float* dy, float lo, float hi, float* out, int64_t n, void* stream);
/* out = a + b over fp16 rows (gradient accumulation) */
int pv_add_rows_f16(const void* a, int32_t lda, const void* b, int32_t ldb, void* out, int32_t ldo, int32_t rows, int32_t cols, void* stream);
/* z (B,2h,2w,c) = x (B,h,w,c) at the even positions, 0 elsewhere: input of the data gradient of a stride-2 3x3 conv (Downsample2D) */
int pv_dilate2x(const void* x, int32_t ldx, void* z, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* out (B,h,w,c) = 2x2 block sums of g (B,2h,2w,c) (+ add): gradient of the x2 nearest upsample (Upsample2D) */
int pv_pool2x_sum(const void* g, const void* add, int32_t ldadd, void* out, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* out = coef * sign(x): gradient of |x|.mean() (train.py:509, coef = weight / n) */
int pv_sign_f32(const float* x, float coef, float* out, int64_t n, void* stream);
/* out[b][e][:] = scale * x[b*seq + idx[b] + e][:] (fp16 rows -> fp32): gradient of the rows _inject_concept_embeddings wrote (clip.py:17-24) */
int pv_gather_rows_f32(const void* x, int32_t ldx, const int32_t* idx, float* out, int32_t batch, int32_t seq, int32_t n_e, int32_t dim, float scale,
                       void* stream);

/* Fused dual-branch cross attention = the SDPA part of PhotoVerseAttnProcessor2_0.__call__
 * (attention_processor.py:307-322 text branch, :392-420 image-token branch and fusion):
 *   out = w_text * softmax(q Kt^T / sqrt(d)) Vt  +  w_ip * softmax(q Kip^T / sqrt(d)) Vip
 * with two INDEPENDENT softmaxes.  (w_text,w_ip) = (1,1) under no_grad (:411-412); the
 * grad-mode fusion rule (:413-420) selects (2,0), (0,2) or (1,1).
 * Also emits to_v_ip_norm[b][h][p] = ||Vip[b,p,h,:]||_2 (:397) when vnorm != NULL.
 * nt <= 77+..., nt + nip <= 96.
 */
typedef struct pv_xattn_params {
    const void* q; int32_t ldq;
    const void* kt; const void* vt; int32_t ldkt, ldvt;   /* text K/V rows [B*nt] */
    const void* kip; const void* vip; int32_t ldkip, ldvip; /* image-token K/V rows [B*nip] */
    void* out; int32_t ldo;
    float* vnorm;                                          /* [B][H][nip] or NULL */
    int32_t batch, heads, nq, nt, nip, d;
    float w_text, w_ip;
    const float* fusion;                                   /* optional DEVICE pair overriding (w_text, w_ip), see pv_fusion_draw */
} pv_xattn_params;
int pv_cross_attention(const pv_xattn_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * The whole attn2 branch of a BasicTransformerBlock as ONE launch (heads = 8; C = heads*d = 320 / d = 40 or C = 640 / d = 80):
 *   out = hs + to_out( w_text*softmax(q Kt^T/sqrt(d)) Vt + w_ip*softmax(q Kip^T/sqrt(d)) Vip ) + bias_o,
 *   q = to_q(LayerNorm(hs))
 * = BasicTransformerBlock.norm2 [EXT diffusers] -> PhotoVerseAttnProcessor2_0.__call__
 * (attention_processor.py:297 to_q, :307-322 text SDPA, :392-420 image-token SDPA + fusion rule,
 * :423 to_out[0]) -> the block's residual add.  Replaces pv_layernorm + pv_gemm_conv +
 * pv_cross_attention + pv_gemm_conv.  nq % 128 == 0, 64 < nt <= 80, nip <= 16.
 *
 * pv_xattn_pack_kv (once per conditioning) turns the projected text / image-token K,V rows into
 * the kernel's K / V images (kimg: batch*heads*96*64 halfs at d = 40, *128 at d = 80; vimg: batch*(C/80)*96*80 halfs) and
 * emits to_v_ip_norm (:397).  wo is to_out[0].weight with its COLUMNS reordered: column slot s of
 * the packed matrix holds natural column pv_xattn_fused_wo_slot(s).
 */
typedef struct pv_xattn_fused_params {
    const void* hs; int32_t ld_hs;             /* fp16 [batch*nq][C]: block input (pre-norm2) = residual */
    int32_t ln; float ln_eps;                  /* ln != 0: q = to_q((hs - mean) * rstd) - norm2 WITHOUT its affine part, which the */
                                               /* caller folds into wq (gamma scales its columns) and q_bias (= to_q.weight . beta) */
    const void* wq;                            /* fp16 [C][C] to_q.weight (x diag(gamma) when ln) */
    const float* q_bias;                       /* fp32 [C] added to q, or NULL */
    const void* wo;                            /* fp16 [C][C] to_out[0].weight, columns in slot order */
    const float* bias_o;                       /* fp32 [C] or NULL */
    const void* kimg; const void* vimg;        /* from pv_xattn_pack_kv */
    void* out; int32_t ld_out;                 /* fp16 [batch*nq][C] */
    int32_t batch, nq, heads, d, nt, nip;
    float w_text, w_ip;                        /* branch weights: (1,1) no_grad; (2,0) / (0,2) / (1,1) grad mode */
    const float* fusion;                       /* optional DEVICE pair overriding (w_text, w_ip): graph-safe grad-mode fusion */
    int32_t rows_per_workgroup;                /* C = 640 only; 0 = the launch decides (64-row workgroups while 128-row ones would not fill the chip: */
                                               /* < 384), 64 / 128 = the caller's choice: a caller that runs two such launches side by side on two */
                                               /* streams (the uncond / cond forwards, infer.py:103-114) asks for 128.  Same results either way (ABI 13) */
} pv_xattn_fused_params;
int pv_cross_attention_fused(const pv_xattn_fused_params* p, void* stream);
int pv_xattn_pack_kv(const void* kt, const void* vt, int32_t ldkt, int32_t ldvt, const void* kip, const void* vip,
                     int32_t ldkip, int32_t ldvip, void* kimg, void* vimg, float* vnorm, int32_t batch,
                     int32_t heads, int32_t d, int32_t nt, int32_t nip, void* stream);
int pv_xattn_fused_wo_slot(int32_t slot);

/* ------------------------------------------------------------------------------------------
 * The C = 1280 / d = 160 attn2 layers (16x16 and 8x8 levels): norm2 -> to_q -> dual-branch SDPA + fusion in ONE head-parallel
 * launch (one workgroup = 128 query rows of ONE head); attn.to_out[0] + bias + the block's residual add follow as one pv_gemm_conv.
 * Replaces pv_layernorm + pv_gemm_conv (to_q, attention_processor.py:297) + pv_cross_attention (:307-322, :392-420) - two launches
 * per layer instead of four (the one-launch pv_cross_attention_fused exists for C = 320 / 640).
 *   ctx[b, m, h*d:(h+1)*d] = w_text*softmax(q Kt^T/sqrt(d)) Vt + w_ip*softmax(q Kip^T/sqrt(d)) Vip,  q = to_q(LayerNorm(hs))[:, head h]
 * norm2 is folded algebraically so that the GEMM reads the raw rows: to_q(LN(x)) = rstd * (wq . x - mean * wq_rowsum) + q_bias with
 * wq = to_q.weight x diag(gamma) (fp16), wq_rowsum[n] = sum_k wq[n][k] (of the fp16 values), q_bias = to_q.weight . beta.
 * d in {160, 80} (80: two heads per 160-feature block); every row extent (batch*nq*ld_hs, ...) below 2 GiB; nt <= 80, nip <= 16; K / V rows as for pv_cross_attention.  (ABI 11)
 */
typedef struct pv_xattn_lnq_params {
    const void* hs; int32_t ld_hs;             /* fp16 [batch*nq][heads*d]: block input (pre-norm2) */
    int32_t ln; float ln_eps;                  /* ln != 0: LayerNorm statistics over the row, affine part folded by the caller */
    const void* wq;                            /* fp16 [C][C] */
    const float* q_bias;                       /* fp32 [C] or NULL */
    const float* wq_rowsum;                    /* fp32 [C]; required when ln */
    const void* kt; const void* vt; int32_t ldkt, ldvt;       /* text K / V rows [batch*nt] */
    const void* kip; const void* vip; int32_t ldkip, ldvip;   /* image-token K / V rows [batch*nip] */
    void* out; int32_t ldo;                    /* fp16 ctx [batch*nq][C] */
    float* vnorm;                              /* [batch][heads][nip] to_v_ip_norm (:397) or NULL */
    int32_t batch, nq, heads, d, nt, nip;
    float w_text, w_ip;
    const float* fusion;                       /* optional DEVICE pair overriding (w_text, w_ip), see pv_fusion_draw */
} pv_xattn_lnq_params;
int pv_cross_attention_lnq(const pv_xattn_lnq_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * pv_row_gemm: LayerNorm + Linear (+ GEGLU gate) for the K = 320 layers of the 64x64-level transformer blocks as ONE row-owning
 * launch: BasicTransformerBlock.norm1 -> [to_q; to_k; to_v] of attn1 (stock AttnProcessor2_0, /root/reference/models/unet.py:20-24) and
 * norm3 -> ff.net[0] (GEGLU) [EXT diffusers transformer block, driven by /root/reference/models/infer.py:103-114].
 *   out[M][N'] = epi( ((x - mean) * rstd)[M][320] . w[N][320]^T + bias ),  N' = N, or N / 2 with geglu (value * gelu_erf(gate))
 * ln != 0: rows are normalised WITHOUT the affine part - the caller folds gamma into the columns of w and w . beta into bias.
 * geglu: w / bias rows packed per 160-row chunk c as 10 fragments of 16 rows, fragment 2q = value rows of output columns
 * 80 c + 16 q .. + 15, fragment 2q + 1 = their gate rows.  N % 320 == 0; K must be 320; M arbitrary (tails through the descriptors).
 */
typedef struct pv_row_gemm_params {
    const void* x; int32_t ld_x;               /* fp16 [M][K] rows */
    int32_t M, K, N;
    const void* w;                             /* fp16 [N][K] */
    const float* bias;                         /* fp32 [N] or NULL */
    int32_t ln; float ln_eps;
    int32_t geglu;
    void* out; int32_t ld_out;                 /* fp16 [M][N or N/2] */
    const float* x_norm;                       /* (ABI 15) optional fp32 [images][2][K]: a GroupNorm of the rows as a per-(image, channel) scale / shift
                                                  (pv_groupnorm_scale_shift): the launch computes epi( (x * scale + shift) . w^T + bias ) - [EXT]
                                                  Transformer2DModel.norm folded into proj_in on the raw tensor, the normalised rows rounded to fp16 exactly as
                                                  pv_groupnorm_apply writes them.  Needs rows_per_image % 128 == 0; not together with ln */
    int32_t rows_per_image;
} pv_row_gemm_params;
int pv_row_gemm(const pv_row_gemm_params* p, void* stream);

/* ------------------------------------------------------------------------------------------
 * BACKWARD of PhotoVerse's own trainable modules (the backward of the stock SD-v1.5 / CLIP blocks the gradient crosses is
 * declared further up: pv_attention_backward, pv_groupnorm_backward, ...).
 *
 * pv_cross_attention_backward: gradient of the dual-branch SDPA (attention_processor.py:317-322,
 * :392-420) given dout = dL/d(attention output before to_out):
 *   dq   fp16 [B*nq][C]         dkt, dvt  fp32 [B*nt][C]        dkip, dvip  fp32 [B*nip][C]
 * every output is multiplied by out_scale (un-scaling of a loss-scaled dout); dvip additionally gets
 * (vnorm_coef + vnorm_grad[b][h][p]) * v / ||v||_head (gradient through to_v_ip_norm, :397 / train.py:512-513)
 * before that scaling.  partial: fp32 workspace batch*heads*ceil(nq/512)*2*96*d.  MFMA (pv_train.hip).  Deterministic.
 */
typedef struct pv_xattn_bwd_params {
    const void* q; int32_t ldq;
    const void* kt; const void* vt; int32_t ldkt, ldvt;
    const void* kip; const void* vip; int32_t ldkip, ldvip;
    const void* dout; int32_t lddo;
    void* dq; int32_t lddq;
    float* partial;
    float* stats;                      /* fp32 workspace [B][heads][nq][4]: per-query (lse_text, lse_ip, delta_text, delta_ip) */
    float* dkt; float* dvt; float* dkip; float* dvip;
    int32_t ld_dt, ld_di;              /* row strides (floats) of dkt / dvt and of dkip / dvip (e.g. 2C for a [dK | dV] buffer) */
    int32_t batch, heads, nq, nt, nip, d;
    float w_text, w_ip;
    const float* fusion;               /* optional device pair overriding (w_text, w_ip) */
    float out_scale, vnorm_coef;
    const float* vnorm_grad;           /* optional dL/d(to_v_ip_norm) fp32 [B][heads][nip], added to vnorm_coef per (b, h, p) */
} pv_xattn_bwd_params;
int pv_cross_attention_backward(const pv_xattn_bwd_params* p, void* stream);
/* out[c][r] = x[r][c] (fp16), rows zero-padded to rows_pad: operand layout of dW = dY^T . X on pv_gemm_conv */
int pv_transpose_f16(const void* x, int32_t ldx, int32_t rows, int32_t cols, void* out, int32_t ldo, int32_t rows_pad, void* stream);
/* LayerNorm (+ LeakyReLU) backward (adapters.py:15-19): dx fp16; dgb_partial (optional) fp32 [ceil(rows / (4 * rows_per_wave))][2][cols] =
 * per-row-block (dgamma, dbeta) terms, to be summed with pv_reduce_blocks */
typedef struct pv_layernorm_bwd_params {
    const void* x; int32_t ldx;
    const void* dy; int32_t lddy;
    void* dx; int32_t lddx;
    const float* gamma; const float* beta;
    float* dgb_partial;
    int32_t rows, cols;
    float eps;
    int32_t act;
    int32_t dy_group, dy_skip;         /* dy_group > 1: row r reads dy row r / dy_group; the first dy_skip rows of a group get 0 */
    float dy_scale;                    /* multiplies dy (1 / count of a mean; 1.0 otherwise) */
    int32_t rows_per_wave;             /* rows each wave walks (0 = 1): dgb_partial has ceil(rows / (4 * rows_per_wave)) blocks */
    const void* add; int32_t ldadd;    /* optional fp16 [rows][cols]: dx = LayerNorm gradient + add (the gradient x already holds from its other */
                                       /* consumers - the residual stream: one launch instead of the gradient + an add pass) (ABI 14) */
} pv_layernorm_bwd_params;
int pv_layernorm_backward(const pv_layernorm_bwd_params* p, void* stream);
/* out[i] = scale * sum_b x[b][i], b in order (deterministic) */
int pv_reduce_blocks(const float* x, int32_t nblk, int64_t inner, float scale, float* out, void* stream);
/* out[0] = scale * sum(a^2), deterministic (gradient norms of clip_grad_norm_) */
int pv_reduce_sumsq(const float* a, int64_t n, float scale, float* partial, int32_t n_partial, float* out, void* stream);
/* Multi-tensor forms of the three optimizer launches: ONE launch each over every parameter tensor.  entries: int64 [T][6] =
 * {param, grad, exp_avg, exp_avg_sq, gscale pointer (or 0), n}; workgroup b handles elements [blk_chunk[b] * chunk, + chunk) of tensor
 * blk_tensor[b].  pv_sumsq_multi: partial[b] = sum of grad^2 of that range; pv_clip_coef_groups: group g owns partial[group_start[g] ..
 * group_start[g+1]): out[2g] = base * min(1, max_norm / (base * sqrt(sum) + 1e-6)), out[2g+1] = base * sqrt(sum) (base = 1 / loss scale);
 * pv_adamw_multi: the AdamW update (torch.optim.AdamW, train.py:372-377, :545) of every range, reading its tensor's gscale.
 * Overflow guard (fp16 gradients under a static loss scale): when any group norm is not finite, pv_clip_coef_groups writes -1 into every
 * out[2g] and adds 1 to counters[1]; otherwise it adds 1 to counters[0] (applied steps).  pv_adamw_multi skips tensors whose gscale is
 * negative and, given counters, takes its bias-correction step from counters[0] instead of `step` - the behaviour of torch's GradScaler,
 * without a host synchronisation.  counters may be NULL (no guard bookkeeping; `step` is used).  `out` holds groups + 1 rows of two floats: the
 * extra row [groups] = {base, 0}, or {-1, 0} on overflow, is the gscale of tensors in NO clip group (skipped with the rest); with counters[0] == 0
 * (nothing applied yet) pv_adamw_multi leaves everything untouched (ABI 10). */
int pv_sumsq_multi(const int64_t* entries, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t n_blocks, int32_t chunk, float* partial, void* stream);
int pv_clip_coef_groups(const float* partial, const int32_t* group_start, int32_t groups, float max_norm, float base, float* out, int32_t* counters,
                        void* stream);
int pv_adamw_multi(const int64_t* entries, const int32_t* blk_tensor, const int32_t* blk_chunk, int32_t n_blocks, int32_t chunk, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int32_t step, const int32_t* counters, void* stream);
/* The fp16 working copies of the trainable weights, re-made from their fp32 masters after every optimizer step (the reference's autocast does
 * the same cast inside every Linear, train.py:464-470) - ONE launch over all of them.  entries: int64 [E][9] = {src (fp32), ld_src, rows, cols,
 * scale (float bits in the low word), dst (fp16), ld_dst, dstT (fp16 or 0), ld_dstT}: dst[r][c] = fp16(scale * src[r][c]) and, when given,
 * dstT[c][r] = the same value (the transposed operand of the data-gradient GEMM).  Workgroup b handles the 32 x 32 tile blk_tile[b] (row-major
 * over ceil(rows/32) x ceil(cols/32)) of entry blk_entry[b].  A destination may be a block of a larger zero-padded matrix (block-diagonal LoRA
 * factors, stacked k / v projections). */
int pv_pack_weights(const int64_t* entries, const int32_t* blk_entry, const int32_t* blk_tile, int32_t n_blocks, void* stream);
/* out[c] = sum_r x[r][c] over fp16 rows (bias gradients); partial: nblk*cols floats */
int pv_colsum_f16(const void* x, int32_t ldx, int32_t rows, int32_t cols, float* partial, int32_t nblk, float* out, void* stream);

/* GEGLU gate (diffusers GEGLU, exact-erf GELU): out[m][j] = x[m][j] * gelu(x[m][n+j]) */
int pv_geglu(const void* x, int32_t ldx, void* out, int32_t ldo, int32_t rows, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Step-state driven pieces of the denoising loop (infer.py:98-119).  `state` is a small device
 * block of int32: step index at state[0], number of table rows at state[1] (0 = unknown); the
 * tables are indexed with min(state[0], state[1]-1) so that ONE captured HIP graph can be
 * replayed for every step and a step past the end of the schedule never reads beyond them.
 */
/* sinusoidal timestep embedding (flip_sin_to_cos, shift 0) -> fp16 [rows][dim];
 * t taken from timesteps[state ? *state : 0 ... ] : rows>1 => per-row timesteps[row] */
int pv_timestep_embedding(const float* timesteps, const int32_t* state, int32_t rows, int32_t dim,
                          void* out, void* stream);
/* conv_out: NHWC fp16 (B,H,W,cin) -> NCHW fp32 (B,cout,H,W), 3x3 pad 1; w fp16 [cout][3][3][cin]; cout 4 (UNet) or 3 (VAE);
 * cin in {64, 128, 256, 320, 512} (512: the data gradient of the VAE decoder's conv_in) */
int pv_conv_out(const void* x, const void* w, const float* bias, float* out, int32_t batch, int32_t cin,
                int32_t h, int32_t wd, int32_t cout, void* stream);
/* CFG combine (infer.py:116) + DPM-Solver++(2M) update (infer.py:119) on fp32 NCHW latents.
 * coef[step][4] = {c_x, c_eps, c_x0, c_x0prev}:  x0 = inv_alpha*x - sig_over_alpha*eps ... see
 * photoverse_amd/scheduler.py; advances nothing (pv_step_advance does). */
int pv_cfg_dpm_step(const float* eps_uncond, const float* eps_cond, float* latents, float* x0_prev,
                    const float* coef, const int32_t* state, float guidance, int64_t n, void* stream);
int pv_step_advance(int32_t* state, void* stream);

/* Grad-mode branch fusion of PhotoVerseAttnProcessor2_0 (attention_processor.py:413-420) WITHOUT the reference's per-layer
 * host sync (`torch.rand(1).item()`): one tiny launch draws u ~ U(0,1) per cross-attention layer on the device
 * (Philox4x32-10, key = rng[0..1], counter = {rng[2] = launch count, layer}) and writes out[layer] = (w_text, w_ip):
 * u < rule1 -> (scale, 0); u > rule2 -> (0, scale); else (1, 1).  rng[2] advances by one per launch, so a captured graph draws
 * fresh numbers on every replay.  forced (optional, [n_layers]): entries >= 0 replace the drawn u (tests).
 * only_last_step != 0 with state != NULL: the rule applies only when state[0] == state[1] - 1 (run_inference's
 * training_mode, infer.py:99: grad is enabled on the last denoising step only), otherwise every layer gets (1, 1). */
int pv_fusion_draw(const int32_t* state, uint32_t* rng, const float* forced, float* out, int32_t n_layers, float rule1,
                   float rule2, float scale, int32_t only_last_step, void* stream);

/* small helpers */
int pv_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream);
int pv_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream);
/* mean over `count` consecutive rows: x fp16 [groups*count][cols] -> y fp16 [groups][cols]
 * (adapters.py:36 mean over the 256 patch tokens); group g starts at row g*group_rows (group_rows >= count);
 * y may be accumulated (+=) when accumulate!=0 */
int pv_rows_mean(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t groups, int32_t count, int32_t group_rows,
                 int32_t cols, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pre-loop conditioning front ends (infer.py:76-96).
 */
/* ------------------------------------------------------------------------------------------
 * VAE decode helpers (infer.py:121-123; [EXT] diffusers AutoencoderKL.decode).  The decoder's convolutions, GroupNorms
 * and Linear layers run on pv_gemm_conv / pv_groupnorm_*; its single-head attention over H*W tokens (head dim 512) is
 * GEMM -> pv_softmax_rows -> GEMM.
 */
/* in place: x[r][c] = softmax_c(scale * x[r][c]) over fp16 rows (fp32 math), cols % 8 == 0 */
int pv_softmax_rows(void* x, int32_t ld, int32_t rows, int32_t cols, float scale, void* stream);
/* 1x1 conv over NCHW fp32 with few channels (post_quant_conv 4->4): w [cout][cin], bias [cout] or NULL */
int pv_pointwise_nchw(const float* x, const float* w, const float* bias, float* out, int32_t batch, int32_t cin, int32_t cout,
                      int32_t hw, void* stream);
/* in place clamp of fp32 values (images.clamp(-1, 1), infer.py:122) */
int pv_clamp_f32(float* x, float lo, float hi, int64_t n, void* stream);
/* out[b][i] = ca[b]*x[b][i] (+ cb[b]*y[b][i]): scheduler.add_noise (infer.py:65, train.py:484: ca = sqrt(acp[t_b]),
 * cb = sqrt(1 - acp[t_b])) and the 1/scaling_factor latent scaling (infer.py:121) */
int pv_affine_rows_f32(const float* x, const float* y, const float* ca, const float* cb, float* out, int64_t per_sample,
                       int32_t batch, void* stream);
/* vae.encode(x).latent_dist.sample() (infer.py:63, train.py:473): moments fp32 [B][2c][hw] (mean | logvar, NCHW),
 * out = mean + exp(0.5*clamp(logvar,-30,20)) * eps */
int pv_posterior_sample(const float* moments, const float* eps, float* out, int32_t batch, int64_t chw, void* stream);
/* deterministic mean reductions of the training losses (train.py:509-516): mode 0 mean(a), 1 mean|a| (concept_text_loss),
 * 2 mean((a-b)^2) (F.mse_loss); a / b fp32 or fp16 (is_f16); partial: workspace of n_partial floats; out[0] = result */
int pv_reduce_mean(const void* a, const void* b, int32_t mode, int32_t is_f16, int64_t n, float* partial, int32_t n_partial,
                   float* out, void* stream);

/* im2col of a 3x3 / pad-1 conv over NCHW fp32 with few channels (UNet conv_in): fp16 rows [B*H*W][kpad], column
 * k = ci*9 + ky*3 + kx, zero padded (kpad % 64 == 0), so conv_in runs on pv_gemm_conv with w.reshape(cout, cin*9). */
int pv_im2col3x3(const float* x, void* out, int32_t batch, int32_t cin, int32_t h, int32_t wd, int32_t kpad, void* stream);
/* CLIP ViT patch embedding input: NCHW fp32 pixels -> fp16 rows [B*(img/patch)^2][kpad], one patch per row in
 * (channel, py, px) order = the flattened Conv2d(3,dim,patch,stride=patch) weight order, zero padded to kpad
 * (kpad % 64 == 0) so the patch embedding is a pv_gemm_conv call.  [EXT transformers CLIPVisionEmbeddings] */
int pv_patchify(const float* x, void* out, int32_t batch, int32_t ch, int32_t img, int32_t patch, int32_t kpad, void* stream);
/* out[b][0] = cls + pos[0]; out[b][1+i] = patches[b][i] + pos[1+i]  (fp16 rows [B*ntok][dim]) */
int pv_clip_vision_embed(const void* patches, const float* cls, const float* pos, void* out, int32_t batch, int32_t ntok,
                         int32_t dim, void* stream);
/* Token + position embedding of the CLIP text encoder with PhotoVerse's concept injection fused
 * (models/clip.py:17-24 _inject_concept_embeddings, :57-63): per row the placeholder position
 * placeholder_idx[b] is overwritten by n_concept concept embeddings and the tail is shifted right by
 * n_concept-1 (truncated).  n_concept == 0: stock embedding.  ids/placeholder_idx int64, tok/pos fp32, out fp16. */
int pv_clip_text_embed(const int64_t* ids, const float* tok, const float* pos, const void* concept, const int64_t* placeholder_idx,
                       int32_t n_concept, void* out, int32_t batch, int32_t seq, int32_t dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PHOTOVERSE_HIP_H */
