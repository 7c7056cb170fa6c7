/*
 * p3hip — C-ABI of the MI355X-native (gfx950) Pix2Poly / FFL encoder-fusion-decoder path.
 *
 * The reference (raphaelsulzer/PixelsPointsPolygons) has NO FFI / plugin boundary for this path:
 * it is a Python nn.Module API whose arithmetic runs inside third-party binaries (ATen/cuDNN/
 * cuBLAS kernels behind timm + torch.nn, and Open3D's compiled `voxelize` / `ragged_to_dense`
 * ops).  This header is the boundary this project introduces one level below those nn.Modules:
 * each entry point names the reference call it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller (PyTorch);
 *     the library allocates nothing persistent and keeps no pointer after return.
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on that stream,
 *     never synchronise, and are safe to capture into a hipGraph.
 *   - return 0 on success; negative P3_E* for argument errors (p3_last_error_string() has text);
 *     positive values are hipError_t codes from a failed launch.
 *   - dtype codes: P3_F32 = 0 (exact fp32 path, fp32 MFMA), P3_BF16 = 1 (bf16 storage, fp32 accumulate), P3_F32X3 = 2 (fp32 storage, products as
 *     bf16 x 3 on the bf16 MFMA: accepted wherever a descriptor's dtype names the operands of a PRODUCT - p3_gemm_desc.dtype_in, p3_gemm_tn's dtype,
 *     p3_attn_desc.dtype, p3_pillar_desc.dtype - and chosen per call, so two models of different precision share one process).
 *   - matrices are row-major with explicit element strides.
 */
#ifndef P3HIP_H
#define P3HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3_OK 0
#define P3_EINVAL (-1)
#define P3_ESHAPE (-2)
#define P3_EALIGN (-3)
#define P3_EUNSUP (-4)

#define P3_F32 0
#define P3_BF16 1
#define P3_F32X3 2 /* operands stored as fp32 exactly like P3_F32; every value is split into hi = bf16(x) and lo = bf16(x - hi) while it is staged and the product
                    * accumulates a_lo b_hi + a_hi b_lo + a_hi b_hi in fp32 on the bf16 MFMA (2^-17 relative per product instead of the exact fp32 MFMA's 2^-24;
                    * outputs, epilogues and every non-product kernel stay fp32).  The "fp32x3" precision of the Python host; tests hold it to the north star's 1e-3. */

#define P3_ACT_NONE 0
#define P3_ACT_GELU 1 /* exact erf GELU (timm Mlp act_layer=nn.GELU) */
#define P3_ACT_RELU 2
#define P3_ACT_MUL 3  /* p3_gemm_desc.bwd_act only: the saved tensor already holds act'(pre) (aux_mode = 1): plain multiply */
#define P3_ACT_BN_RELU 4 /* p3_gemm_desc.bwd_act only: bwd_saved = H, the input of a train-mode BatchNorm + ReLU whose OUTPUT this GEMM's product is the gradient
                          * of: C = [H*bn[0] + bn[1] > 0] * (A'W^T) * bn[0] + bn[2] + bn[3] * H  with bwd_bn = four rows of N floats (scale, shift, and the a / b
                          * of p3_bn_bwd_coeffs) - the whole BatchNorm backward of the layer in front, no [M,N] gradient stored in between */

#define P3_A_PLAIN 0
#define P3_A_CONV3X3 1 /* A is an NHWC map [B,H,W,lda]; K = 9*C, zero padding 1 (implicit GEMM) */
#define P3_A_AFFINE_RELU 2 /* A'[m,k] = relu(A[m,k]*a_scale[k] + a_shift[k])  (BN+ReLU folded into the load) */
#define P3_A_PAIR_AFFINE_RELU 3 /* A'[(b,i,j),k] = relu((U[b,i,k]+V[b,j,k])*a_scale[k]+a_shift[k]) ScoreNet conv1 */
#define P3_A_AFFINE_MASK2 5 /* p3_gemm_tn_ex only: B'[m, k] = [y > 0] for k < K and [y > 0] * B[m, k - K] for K <= k < 2K, y = B[m, k]*b_scale[k]+b_shift[k]:
                              * ONE pass over (A, B) gives G = A^T [y > 0] and G2 = A^T ([y > 0] B) - a BatchNorm/ReLU layer's weight gradient AND the sums
                              * of its input gradient's BatchNorm backward (p3_bn_sums_from_g); C is [N, 2K] */
#define P3_A_CONV3X3_AFFINE_RELU 4 /* CONV3X3 over relu(A*a_scale[c]+a_shift[c]) (a_scale/a_shift indexed by input channel): FFL heads */

int p3_version(void);
const char* p3_last_error_string(void);
/* Measurement hook: after p3_trace_kernels(1), p3_last_kernel() names the device kernel the calling thread's last p3_gemm launched, spelled as
 * rocprofv3 --kernel-trace prints it after tools/kstats.py's bf16 rewrite ("gemm_dma_kernel<bf16, 32, 2>"); "" when tracing is off. */
void p3_trace_kernels(int on);
const char* p3_last_kernel(void);

/* Deterministic reductions.  The reference seeds everything and sets cudnn.deterministic (misc/shared_utils.py:120-126, trainer.py:214);
 * here the kernels that would finish in fp32 atomicAdd's over workgroup partials (BatchNorm sums of the ScoreNet, split-M weight
 * gradients, bias-gradient column sums) store the partials into `scratch` instead and add them in workgroup order in float64: the same
 * bits every run.  scratch: device memory owned by the caller (16-byte aligned, >= 8 MB for the shapes of this path; a launch whose
 * partials do not fit falls back to atomics), used launch by launch in stream order on ONE stream; NULL / 0 switches the mode off.
 * all_dtypes = 0: only fp32 (parity-mode) launches take the path; 1: bf16 launches too.  p3_get_deterministic: 0 off, 1 fp32, 2 all. */
int p3_set_deterministic(void* scratch, int64_t bytes, int all_dtypes);
int p3_get_deterministic(void);
/* More than one launch stream (r04: the independent branches of model_pix2poly.py:256-264 - scorenet1 || scorenet2 - the weight-gradient
 * GEMMs and the pillar stem beside the patch embedding, early_fusion_vit.py:99-100, may be enqueued on side streams): the scratch is cut
 * into `n` equal regions (p3_scratch_regions, before or after p3_set_deterministic).  Region 0 serves every stream that is not registered
 * (default stream, hipGraph capture streams); p3_scratch_side_stream registers a side stream (returns its region 1 .. 15) and
 * p3_scratch_stream names the stream of the launches that follow (returns the region, -1 = a side stream beyond the region count: those
 * launches take their atomics path).  Single-threaded like the rest of the library. */
int p3_scratch_regions(int n);
int p3_scratch_side_stream(void* stream);
/* Deferred parameter-gradient reduces (r04).  p3_reduce_defer(arena, floats): from now on a launch whose workgroup partials only feed parameter gradients (the
 * LayerNorm backward's dgamma / dbeta: 42 launches per Pix2Poly train step) parks them in `arena` while p3_reduce_defer_enable(1) is in force (device memory, caller-owned; NULL switches the mode off) instead
 * of launching its own reduce; p3_reduce_flush(stream) adds every parked set to its outputs in ONE launch, in the same fixed float64 order (bit-identical
 * gradients) - call it after the backward pass, before anything reads the gradients.  p3_reduce_pending(): sets parked since the last flush.  A full arena or
 * table (48 sets) falls back to the immediate reduce. */
int p3_reduce_defer(float* arena, int64_t floats);
int p3_reduce_defer_enable(int on);    /* parking happens only between enable(1) and enable(0): the host brackets exactly the launches whose outputs are
                                         * accumulation targets that stay valid until the flush (gradient arena views), returns the previous setting */
int p3_reduce_flush(void* stream);
int p3_reduce_pending(void);
int p3_reduce_drop(void);              /* forget the parked sets without adding them (after a backward pass that raised); returns how many */
int p3_scratch_stream(void* stream);
/* Deferred weight-gradient reduces (r05), the same idea for the split-M weight-gradient GEMMs (p3_gemm_tn / p3_gemm_tn_ex / p3_gemm_tn_x3) in the deterministic mode:
 * between p3_tn_defer_enable(1) and (0) a launch that would store its partial tiles in the caller's `slabs` and add them to C with a reduce launch of its own
 * (109 launches of ~16 us per fp32x3 train step) parks them in `arena` instead (p3_tn_defer: device memory, caller-owned; a full arena / table falls back to the
 * immediate reduce); p3_tn_flush(stream) adds every parked set to its C - float64, split order: bit-identical gradients - in ceil(sets / 96) launches.  Only for
 * C that stays valid and unread until the flush (views of the optimizer's gradient arena); call the flush after the backward pass. */
int p3_tn_defer(float* arena, int64_t floats);
int p3_tn_defer_enable(int on);
int p3_tn_flush(void* stream);
int p3_tn_pending(void);
int p3_tn_drop(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[M,N] = act(A'[M,K] * W[N,K]^T + bias) + residual
 * Replaces every nn.Linear / 1x1 / kxk(stride k) conv on the path:
 *   timm PatchEmbed.proj            models/fusion_layers/early_fusion_vit.py:69-70,99  (im2col'd by p3_patchify)
 *   timm Attention.qkv/.proj, Mlp   models/vision_transformer/vit.py:48 (timm Block x12)
 *   fusion Conv3x3                  models/fusion_layers/early_fusion_vit.py:75-79     (P3_A_CONV3X3)
 *   nn.MultiheadAttention in/out proj, linear1/2, output   models/pix2poly/model_pix2poly.py:138-139,185
 *   ScoreNet conv1..3               models/pix2poly/model_pix2poly.py:74-80            (P3_A_PAIR_AFFINE_RELU / P3_A_AFFINE_RELU)
 * K must be a multiple of 64 (bf16) / 16 (f32); M, N arbitrary.
 * ------------------------------------------------------------------------------------------ */
/* Counter-based dropout (nn.Dropout / the attention-probability dropout of nn.MultiheadAttention in training mode, reference
 * Decoder: model_pix2poly.py:136,139,143): element (row, col) of site `site` is kept iff 16 bits of hash(seed, site, row, col) are
 * >= round(p * 65536), and scaled by 1/(1-p); the mask is never stored, backward kernels regenerate it.  `seed` points to a DEVICE counter (advance it once per
 * training step with p3_rng_advance, inside the captured graph).  seed == NULL disables dropout. */
typedef struct {
    const unsigned long long* seed;
    unsigned int site;
    float p;
} p3_dropout;
int p3_rng_advance(unsigned long long* seed, void* stream);
/* Batched 2-D bf16 transposes inside one arena (the W^T copies the dX GEMMs read, refreshed after the optimizer step in one launch).
 * table: device array of n_entries records {int64 src_off, int64 dst_off, int32 rows, int32 cols, int32 first_tile, int32 tiles_c}
 * (element offsets; 32x32 tiles, tiles_c = ceil(cols / 32), first_tile = running sum of tile counts); grid = total_tiles blocks. */
int p3_transpose_many(const void* src, void* dst, const void* table, int n_entries, int total_tiles, void* stream);
/* out[i] = keep(i / ncols, i % ncols) ? in[i] / (1-p) : 0   (elementwise dropout forward, and its backward applied to the gradient) */
int p3_dropout_apply(const void* in, int dtype_in, void* out, int dtype_out, int64_t n, int64_t ncols, const p3_dropout* drop, void* stream);

typedef struct {
    int M, N, K;
    int lda, ldb, ldc;
    int dtype_in;  /* dtype of A and W */
    int dtype_out; /* dtype of C and aux */
    int act;
    int a_mode;
    const float* bias;    /* [N] or NULL */
    const void* residual; /* [M,N] (ldr) or NULL; added after the activation */
    int ldr;
    int dtype_res;        /* dtype of residual (may differ from dtype_out: bf16 stream + fp32 pre-LayerNorm sum) */
    void* aux;            /* optional [M,N] (ldc): pre-activation values (for backward) */
    int conv_H, conv_W, conv_C; /* P3_A_CONV3X3 */
    const float* a_scale; /* [K] for the AFFINE modes */
    const float* a_shift; /* [K] */
    const void* pair_V;   /* P3_A_PAIR_AFFINE_RELU: V [B*n, K] (A is U [B*n, K]); M = B*n*n */
    int pair_n;
    float* colsum;        /* optional [N]: += sum over rows of (A'W^T + bias)   (train-mode BatchNorm statistics) */
    float* colsumsq;      /* optional [N]: += sum of squares */
    p3_dropout drop;      /* dropout of act(A'W^T + bias) before the residual add; element = (row, col) */
    /* backward of an upstream activation fused into this (dX) GEMM's epilogue: C = (A'W^T) * act'(bwd_saved) * bwd_scale, with
     * bwd_saved [M,N] (ldc, dtype_out) = the pre-activation (P3_ACT_GELU) or the activation output (P3_ACT_RELU) saved by forward */
    const void* bwd_saved;
    int bwd_act;
    float bwd_scale;
    int aux_mode;         /* what `aux` receives: 0 = the pre-activation, 1 = act'(pre) (GELU: cdf + x*pdf, computed with the
                           * activation from the same erf / exp), so that backward is a multiply with no transcendental */
    int conv_pad;         /* P3_A_CONV3X3: 1 = A is a zero-bordered image [B, conv_H+2, conv_W+2, lda] (p3_pad_nhwc): taps are read
                           * without bounds checks; conv_H / conv_W stay the OUTPUT size */
    const float* bwd_bn;  /* bwd_act = P3_ACT_BN_RELU: [4, N] floats (scale | shift | a | b) */
    const void* w_lo;     /* dtype_in = P3_F32X3 with a plain / 3x3-gathered A only: non-NULL = the weight comes as PLANES - W addresses hi = bf16(w) [N, K], w_lo the
                           * matching lo = bf16(w - hi), both with row stride ldb (bf16 elements, % 8 == 0, 16-byte aligned) - what the optimizer keeps fresh for every
                           * 2-D master (training.FlatAdamW).  The kernel then stages W with plain 16-byte copies instead of splitting it in every tile (r06: MFMA and VALU
                           * work do not overlap on a SIMD - tools/probe/coissue_probe.hip - so the split of an operand that never changes was pure added time);
                           * the bits of the result are those of the fp32 weight's split */
} p3_gemm_desc;
int p3_gemm(const void* A, const void* W, void* C, const p3_gemm_desc* d, void* stream);
/* The LDS-DMA kernels behind p3_gemm for the plain bf16 products (csrc/gemm_dma.hip), callable directly for A/B measurements and parity tests: same
 * descriptor, P3_EUNSUP when the problem is not eligible (plain bf16 A, K % 64 == 0, N % 8 == 0, 16-byte aligned rows, no column sums).  variant 4: 128 x 128
 * tile, 64-deep slices, two in LDS; 6: 32-deep, two = 4 workgroups / CU; 9: 128 x 384 tile with 8 waves.  p3_gemm itself picks them from M = 2048 on (N = 384
 * and K >= 1024: 9; other K >= 1024: 4; K <= 512 and N >= 1024: 6; P3_GEMM_DMA=0 switches that off).  All of them add the same 16-deep MFMA blocks in
 * ascending k order as the register-staged kernel: bit-identical outputs. */
int p3_gemm_dma(const void* A, const void* W, void* C, const p3_gemm_desc* d, int variant, void* stream);

/* ------------------------------------------------------------------------------------------
 * "Planes" - the operand format of the fp32x3 ViT block (r05).  A value x of an fp32 tensor travels between the kernels of one timm Block
 * (models/vision_transformer/vit.py:48) as TWO bf16 numbers, hi = bf16(x) and lo = bf16(x - hi) (16 significant bits, the same 4 bytes as the fp32 value), stored
 * as two row-major bf16 matrices with one leading dimension (typically the two halves [:, :K] and [:, K:] of one [M, 2K] buffer).  The PRODUCER writes the
 * split (LayerNorm output, GELU output, attention output, the gradients between the backward GEMMs), so that the consuming GEMM stages all four operand
 * images global -> LDS by LDS-DMA with no conversion pass and multiplies a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA - the arithmetic of P3_F32X3
 * (2^-17 per product) at the memory path of the bf16 kernels.  Weights: the optimizer keeps hi / lo bf16 arenas beside the fp32 master (p3_adamw, shadow_lo).
 *
 * p3_gemm_x3:  C = epilogue(A W^T),  A = a_hi + a_lo [M, K], W = w_hi + w_lo [N, K]
 *   v = A W^T + bias;  act == P3_ACT_GELU: aux <- GELU'(v) (fp32 [M, N], ldaux), v <- GELU(v);  mul != NULL: v *= mul[m, n] (fp32, ldmul: the backward of
 *   that GELU);  residual != NULL: v += residual[m, n] (fp32, ldr);  then C <- v as fp32 (c, ldc; c_lo == NULL) or as planes (c = hi, c_lo = lo, ldc).
 *   Fused LayerNorm of the OUTPUT row (ln_gamma != NULL; needs N == 384 == the tile width, fp32 C): besides C the kernel writes LN(C) as planes (ln_hi, ln_lo,
 *   ldln) and the row statistics (ln_mean, ln_rstd) - the norm2 / next block's norm1 of timm's Block, whose separate pass re-read the 77 MB stream.
 * Shapes: K % 32 == 0, N % 8 == 0, 16-byte aligned rows.  Kernels: K = 256 / 384 with 1024 <= N <= 4096, N % 32 == 0 - the A-stationary persistent kernel
 * (gemm_x3_as.hip: the A rows of a wave live in registers, the weights stream through LDS, one workgroup per CU walks (row block, column block) units);
 * otherwise tiles of 128 x 128 (4 waves) or, for N == 384 and K >= 1024, 128 x 384 (8 waves).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int M, N, K;
    const void* a_hi; const void* a_lo; int lda;
    const void* w_hi; const void* w_lo; int ldb;
    void* c; void* c_lo; int ldc;
    const float* bias;
    const float* residual; int ldr;
    int act;
    float* aux; int ldaux;
    const float* mul; int ldmul;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    void* ln_hi; void* ln_lo; int ldln;
    float* ln_mean; float* ln_rstd;
    int tile;             /* 0: the library picks the kernel (measured rule); 1: the 128 x 128 tile kernel, 2: the 128 x 384 tile kernel, 3: the A-stationary kernel -
                           * for THIS call (the host splits a launch with a ragged last round of 128 x 384 tiles into a head on tile 2 and a remainder on tile 1);
                           * the process-wide p3_gemm_x3_tile hook is for measurement tools only and loses against this field */
} p3_gemm_x3_desc;
int p3_gemm_x3(const p3_gemm_x3_desc* d, void* stream);
/* measurement hook (tools/mb_x3.py, tools/mb_as.py): 0 = the library's rule, 1 = every product on the 128 x 128 tile, 2 = on the 128 x 384 tile, 3 = on the
 * A-stationary persistent kernel (csrc/gemm_x3_as.hip; P3_EUNSUP unless K = 256 / 384, N % 32 == 0, N <= 4096); returns the previous mode */
int p3_gemm_x3_tile(int mode);
/* measurement hook: device buffer [256][8][8] of 64-bit cycle sums written by the instrumented variants of the A-stationary kernel (environment P3_AS_VAR & 64);
 * NULL switches it off */
int p3_gemm_x3_as_debug(void* buf);
/* weight gradient from planes:  C[N, K] (+)= (a_hi + a_lo)[M, N]^T (b_hi + b_lo)[M, K]  (fp32 C, split over M: fp32 atomics, or - `slabs` given - partial
 * tiles + a fixed-order reduce like p3_gemm_tn_ex); colsum (optional, [N]) += column sums of A (the bias gradient).  M % 64 == 0, N % 128 == 0, K % 128 == 0. */
int p3_gemm_tn_x3(const void* a_hi, const void* a_lo, int lda, const void* b_hi, const void* b_lo, int ldb, float* C, int ldc, int M, int N, int K,
                  float* colsum, float* slabs, int max_slabs, void* stream);
/* measurement hook: 0 keeps every weight gradient on the 128 x 128 tile (the 128 x 384 tile is the default where K % 384 == 0 and it has >= 8 tiles); returns the previous setting */
int p3_gemm_tn_x3_wide(int on);
/* fp32 [rows, cols] (ld_src) -> planes (hi, lo; ld_dst), and back (x = hi + lo): the seams of the planes region (attention outputs, tests) */
int p3_to_planes(const float* src, int ld_src, void* hi, void* lo, int ld_dst, int64_t rows, int cols, void* stream);
int p3_from_planes(const void* hi, const void* lo, int ld_src, float* dst, int ld_dst, int64_t rows, int cols, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim:  y = (x - mean) / sqrt(var + eps) * gamma + beta
 * timm Block.norm1/norm2/VisionTransformer.norm (eps 1e-6); nn.TransformerDecoderLayer.norm1..3
 * (eps 1e-5, models/pix2poly/model_pix2poly.py:138).  x dtype_in, y dtype_out; optional save of
 * mean / rstd (float[rows]) for the backward pass.
 * ------------------------------------------------------------------------------------------ */
int p3_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int cols, int ldx, int ldy,
                 float eps, int dtype_in, int dtype_out, float* save_mean, float* save_rstd, void* stream);
/* dres (optional, dtype_dx, [rows, cols]): gradient arriving at x through a residual connection; added to dx in the same pass */
int p3_layernorm_bwd_res(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                         void* dx, float* dgamma, float* dbeta, int64_t rows, int cols, int dtype_dy, int dtype_x, int dtype_dx,
                         void* stream);
/* dx_lo (optional, bf16 [rows, cols], needs cols % 128 == 0 and an fp32 dx): a bf16 copy of dx written in the same pass - what the
 * GEMMs of the sublayer below read, instead of a separate cast pass over the fp32 residual-gradient stream */
int p3_layernorm_bwd_lo(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                        void* dx, void* dx_lo, float* dgamma, float* dbeta, int64_t rows, int cols, int dtype_dy, int dtype_x, int dtype_dx,
                        void* stream);
/* lo_drop (optional; needs dx_lo, no dres, bf16 dy, fp32 x / dx, cols 256 / 384 / 768): dx_lo = mask(dx) / (1-p) with the mask of
 * dropout site lo_drop at element (row, col) - the gradient the nn.Dropout in front of this LayerNorm's residual add
 * (nn.TransformerDecoderLayer.dropout1..3, model_pix2poly.py:136) passes down, so that sublayer's backward needs no p3_dropout_apply pass.
 * dx (the residual path) is not masked. */
int p3_layernorm_bwd_lo_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                             void* dx, void* dx_lo, const p3_dropout* lo_drop, float* dgamma, float* dbeta, int64_t rows, int cols,
                             int dtype_dy, int dtype_x, int dtype_dx, void* stream);
/* planes forms (fp32x3 ViT block, see p3_gemm_x3): LayerNorm of the fp32 stream written as planes (y_hi, y_lo, row stride ldy) - the qkv / fc1 operand;
 * and the backward whose fp32 dx (the residual-gradient stream, + dres) is ALSO written as planes from the same registers (dx_hi, dx_lo, ld_planes): the
 * operand of the dX / dW GEMMs of the sublayer below. */
int p3_layernorm_planes(const float* x, const float* gamma, const float* beta, void* y_hi, void* y_lo, int64_t rows, int cols, int ldx, int ldy, float eps,
                        float* save_mean, float* save_rstd, void* stream);
int p3_layernorm_bwd_planes(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                            void* dx_hi, void* dx_lo, int ld_planes, float* dgamma, float* dbeta, int64_t rows, int cols, void* stream);
int p3_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                     float* dgamma, float* dbeta, int64_t rows, int cols, int dtype_dy, int dtype_x, int dtype_dx,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention (flash style, MFMA):  O = softmax(Q K^T * scale + bias) V
 * Replaces F.scaled_dot_product_attention in timm Attention (785x785, 6 heads x 64) and the three
 * attention products of nn.TransformerDecoderLayer (models/pix2poly/model_pix2poly.py:177-182):
 *   causal != 0   : tgt_mask of create_mask (model_pix2poly.py:12-19), keys j > i masked to -inf
 *   key_bias      : float [B, Lk] ADDED to the scores - the reference passes (tgt == PAD).float() as
 *                   tgt_key_padding_mask, which torch treats as an additive +1.0 bias (model_pix2poly.py:28-29)
 * Q/K/V/O are addressed as  ptr + b*batch_stride + t*row_stride + h*head_dim  (elements).
 * lse (optional, float [B,H,Lq]) receives log-sum-exp rows for the backward pass.
 * head_dim in {32, 64}.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int B, H, Lq, Lk, head_dim;
    int64_t q_bs, k_bs, v_bs, o_bs; /* batch strides */
    int q_rs, k_rs, v_rs, o_rs;     /* row (token) strides */
    float scale;
    int causal;
    const float* key_bias; /* [B, Lk] or NULL */
    int dtype;             /* Q,K,V,O dtype */
    float* lse;            /* [B,H,Lq] or NULL */
    p3_dropout drop;       /* dropout of the attention probabilities; element = (row (b*H + h)*Lq + q, col k) */
    unsigned int* drop_rows; /* optional keep-bit words, B*H*ceil(Lk/32)*Lq of them laid out [B*H][ceil(Lk/32)][Lq] (bit k%32 of word k/32
                              * of query q = element kept; q innermost so that a wave's 32 queries touch 128 contiguous bytes): WRITTEN
                              * by p3_attention when given, READ by p3_attention_bwd instead of re-hashing (NULL: hash again) */
    int grad_planes;        /* p3_attention_bwd with dtype P3_F32X3 only: 1 = dQ / dK / dV are written as PLANES (see p3_gemm_x3): the pointers address the bf16 hi
                             * plane with batch stride g_bs and row stride g_rs (bf16 elements), the lo plane lies g_lo elements behind - the packed qkv gradient
                             * leaves as the operand of the projection's dX / dW GEMMs, no fp32 tensor and no conversion pass.  0: dtype / strides of Q, K, V */
    int g_rs;
    int64_t g_bs, g_lo;
    void* o_planes;         /* p3_attention with dtype P3_F32X3 only, optional: the output ALSO as planes - the bf16 hi plane of O at o_planes + b*op_bs + t*op_rs + h*head_dim
                             * (bf16 elements), the lo plane op_lo elements behind it: the operand of the output projection's planes GEMM, written from the registers that
                             * hold the fp32 row (r06: replaces a p3_to_planes pass over O per ViT block).  O itself is written as before (the backward reads it) */
    int op_rs;
    int64_t op_bs, op_lo;
} p3_attn_desc;
int p3_attention(const void* Q, const void* K, const void* V, void* O, const p3_attn_desc* d, void* stream);



/* ------------------------------------------------------------------------------------------
 * LiDAR pillar stem = PointPillarsEncoder.forward (models/pointpillars/pointpillars_o3d.py:85-107):
 * Open3D-ML PointPillars.voxelize (hard voxelisation: inclusive range test, <= max_points lowest point
 * indices per pillar, <= max_voxels pillars per sample in ascending hash order) -> PillarFeatureNet
 * (decorate to 8 ch; Linear(8,32,no bias)+BatchNorm1d(eps 1e-3, momentum 0.01)+ReLU+max+concat;
 * Linear(64,C)+BN+ReLU+max; padded slots are zeros and take part in BN statistics and in the max) ->
 * PointPillarsScatter.  Input is the jagged layout of datasets/collate_funcs.py:108:
 * values [sumN,3] f32 + offsets [B+1] int64.  Output: token-major canvas out[b, y*nx+x, col_off + c]
 * (row stride out_ld) = NHWC of the reference's [B,C,ny,nx]; empty pillars are exact zeros.
 * training != 0: BatchNorm batch statistics + running-stat update, else running statistics.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int B;
    int64_t total_points; /* offsets[B] (rows of `values`) */
    int nx, ny;           /* pillar grid (patch_feature_width/height) */
    float vx, vy, vz;     /* in_voxel_size */
    float zmax;           /* point_cloud_range z max (range min is 0) */
    int max_points;       /* max_num_points_per_voxel (1..4096; > 64 = the lidar_density_ablation{128,256,512} configs) */
    int max_voxels;       /* max_num_voxels.{train,test} */
    int C;                /* patch_feature_dim */
    int training;
    float bn_eps, bn_momentum;
    int dtype;            /* dtype of w2, of the internal X2/H2 matrices and of `out` */
    int out_ld, out_col_off;
    int no_backward;      /* != 0: no p3_pillar_stem_bwd will follow on this workspace - the layer-1 activation matrix need not be kept (C = 128 / 384, bf16 / P3_F32X3: never written) */
} p3_pillar_desc;
int64_t p3_pillar_stem_workspace_bytes(const p3_pillar_desc* d);
int p3_pillar_stem(const float* values, const int64_t* offsets, const float* w1, const float* bn1_gamma,
                   const float* bn1_beta, float* bn1_rmean, float* bn1_rvar, const void* w2, const float* bn2_gamma,
                   const float* bn2_beta, float* bn2_rmean, float* bn2_rvar, void* out, void* workspace,
                   const p3_pillar_desc* d, void* stream);
/* byte offsets (17 int64) of the workspace sections: sorted, vox_xy, vox_start, vox_cnt, vox_row, nvox, X2, H2, hmax, hmin, F8 (decorated
 * point features per X2 row), row_vox (pillar slot per row, -1 unused), row_w, then the statistics SyncBatchNorm all-reduces between the
 * phases: totals (int32: kept pillars of the batch), sums1 (fp32 [2*32]), sums2 (fp32 [2*C]), acc1 (fp32 [584]: layer-0 backward sums,
 * dbeta1 at [520:552], dgamma1 at [552:584]) */
int p3_pillar_stem_layout(const p3_pillar_desc* d, int64_t* offsets);
/* p3_pillar_stem split at its two BatchNorm statistics (nn.SyncBatchNorm.convert_sync_batchnorm, model_pix2poly.py:326): phases is a
 * bit mask, 1 = pillarize + layer-0 sums, 2 = layer-0 apply + layer-1 GEMM + sums, 4 = finalize + scatter; all-reduce totals / sums1
 * after phase 1 and sums2 after phase 2.  p3_pillar_stem == phases 7. */
int p3_pillar_stem_phased(const float* values, const int64_t* offsets, const float* w1, const float* bn1_gamma,
                          const float* bn1_beta, float* bn1_rmean, float* bn1_rvar, const void* w2, const float* bn2_gamma,
                          const float* bn2_beta, float* bn2_rmean, float* bn2_rvar, void* out, void* workspace,
                          const p3_pillar_desc* d, int phases, void* stream);
/* Backward of p3_pillar_stem w.r.t. the PillarFeatureNet parameters (what autograd produces for
 * PointPillarsEncoder.forward, pointpillars_o3d.py:85-107: voxel_encoder.pfn_layers.{0,1}.{linear.weight, norm.weight, norm.bias};
 * the point coordinates carry no gradient).  dcanvas: gradient of the token-major canvas the forward wrote (same dtype, row stride
 * dcanvas_ld; columns d->out_col_off .. +C are read).  w2t = pfn_layers.1.linear.weight transposed, [64, C] in the compute dtype.
 * `workspace` must be the buffer the matching forward call filled (pillar tables, layer inputs / outputs, BatchNorm statistics) and
 * is consumed.  Outputs (fp32, overwritten): dw1 [32,8], dg1/db1 [32], dw2 [C,64], dg2/db2 [C]. */
int p3_pillar_stem_bwd(const void* dcanvas, int dcanvas_ld, const float* w1, const float* bn1_gamma, const void* w2t,
                       const float* bn2_gamma, void* workspace, const p3_pillar_desc* d, float* dw1, float* dg1, float* db1,
                       float* dw2, float* dg2, float* db2, void* stream);
/* The same in phases (1 = layer-1 arg-max gradients + local dg2/db2, 2 = dH2 rows + GEMMs + layer-0 sums, 4 = layer-0 finalize) for
 * SyncBatchNorm: stat2 = all-reduced [db2 (C) | dg2 (C)], stat1 = all-reduced [dbeta1 (32) | dgamma1 (32)] (acc1[520:584]) feed the
 * input-gradient terms; NULL = this rank's own sums.  The parameter gradients written to dg*, db* stay per-rank, as in torch. */
int p3_pillar_stem_bwd_phased(const void* dcanvas, int dcanvas_ld, const float* w1, const float* bn1_gamma, const void* w2t,
                              const float* bn2_gamma, void* workspace, const p3_pillar_desc* d, float* dw1, float* dg1, float* db1,
                              float* dw2, float* dg2, float* db2, const float* stat2, const float* stat1, int phases, void* stream);

/* ------------------------------------------------------------------------------------------
 * HBM-bound glue of the encoders / decoder (each replaces a chain of ATen elementwise kernels)
 * ------------------------------------------------------------------------------------------ */
/* im2col of timm PatchEmbed.proj (Conv2d k=P, s=P): img NCHW f32 -> rows [B*(H/P)*(W/P), Cin*P*P], k = c*P*P + py*P + px */
int p3_patchify(const float* img, void* out, int B, int Cin, int H, int W, int P, int dtype_out, void* stream);
/* timm VisionTransformer._pos_embed (+ fusion BN2d+ReLU, early_fusion_vit.py:75-79,123 when scale/shift given):
 *   x[b,0,:] = cls + pos[0];  x[b,1+p,:] = f(src[b,p,:]) + pos[1+p];  x is the fp32 residual stream [B, np+1, D] */
int p3_tokens_assemble(const void* src, int src_ld, int dtype_src, const float* scale, const float* shift, const float* cls,
                       const float* pos, float* x, int B, int np, int D, void* stream);
/* drop CLS + nn.AdaptiveAvgPool1d(Dout) over channels (vit.py:41,49; early_fusion_vit.py:94,125) (+ Decoder's
 * `encoder_out + encoder_pos_embed`, model_pix2poly.py:171-173, when pos != NULL).  y: [B, np+1, Din] */
int p3_pool_pos(const void* y, int dtype_in, const float* pos, void* out, void* out_nopos, int dtype_out, int B, int np, int Din,
                int Dout, void* stream);
/* Decoder.forward head (model_pix2poly.py:164-168 + create_mask :21-31): x = embedding[tgt] + decoder_pos_embed,
 * key_bias = (tgt == pad).float() */
int p3_embed_tokens(const int64_t* tokens, const float* emb, const float* pos, void* x, float* key_bias, int B, int L, int D,
                    int pad_idx, int dtype_out, void* stream);
/* ScoreNet.forward (model_pix2poly.py:86-112), never materialising the [B,512,N,N] pair tensor:
 *   p3_pair_mean   feats[:,1:] -> mean of token pairs [B,N,D]
 *   p3_gemm x2     U = F W1[:, :D]^T + b1, V = F W1[:, D:]^T              (conv1 is separable over (i, j))
 *   p3_pair_stats  closed-form BatchNorm2d batch statistics of U_i + V_j  (sums[0:C] = sum, sums[C:2C] = sum of squares)
 *   p3_bn_finalize scale/shift (+ running stats update, torch momentum semantics)
 *   p3_gemm        conv2 with P3_A_PAIR_AFFINE_RELU, conv3 with P3_A_AFFINE_RELU (colsum -> next BN)
 *   p3_score_out   BN3+ReLU+conv4 -> scores [B,N,N]; transpose_accumulate adds s2^T (perm = s1 + s2^T, :257-259) */
int p3_pair_mean(const void* feats, void* out, int B, int L, int N, int D, int dtype, void* stream);
int p3_pair_stats(const void* U, const void* V, int B, int N, int C, int dtype, float* sums, void* stream);
int p3_bn_finalize(const float* sums, int C, float count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float eps, float momentum, int training, float* scale, float* shift, float* save_mean,
                   float* save_rstd, void* stream);
int p3_score_out(const void* H3, int dtype, const float* scale, const float* shift, const float* w4, const float* b4, float* out,
                 int B, int N, int C, int transpose_accumulate, void* stream);
/* log_optimal_transport (model_pix2poly.py:44-66) + [:, :m, :n] + softmax(-1) (:261-264) in one launch.
 * perm [B,m,n] (may be NULL), z_full [B,m+1,n+1] = log_optimal_transport's return value (may be NULL),
 * uv_hist [B,iters,(m+1)+(n+1)] optional dual iterates for the backward pass. */
int p3_sinkhorn(const float* scores, const float* alpha, int B, int m, int n, int iters, float* perm, float* z_full,
                float* uv_hist, void* stream);
/* scores_to_permutations (predictor_pix2poly.py:307-319): per tile scipy.optimize.linear_sum_assignment(-scores[b]) (maximize = 1),
 * the float64 shortest-augmenting-path solver with scipy's tie rule, one wave per tile.  scores [B,N,N] fp32;
 * col4row [B,N] (column assigned to row r; -1 if the tile failed); perm [B,N,N] 0/1 fp32 (may be NULL);
 * status [B]: 0 ok, 1 NaN / -inf cost (scipy: ValueError "invalid numeric entries"), 2 infeasible. */
int p3_assignment(const float* scores, int B, int N, int maximize, int32_t* col4row, float* perm, int32_t* status, void* stream);
/* greedy decode step (predictor_pix2poly.py:165,196-197): argmax over the last dim, first maximum wins */
int p3_argmax(const float* x, int64_t* out, int rows, int cols, int ld, void* stream);

/* One nn.TransformerDecoderLayer (post-norm, ReLU, eval mode) applied to ONE new position per sample with key/value caches: the body
 * of Decoder.predict's loop (model_pix2poly.py:187-219, layer built at :138-141) as a single launch.  bf16 activations / weights, fp32
 * biases and LayerNorm parameters; D = 256, 8 heads, FF = 2048.  Weights are row-major [out, in] like nn.Linear.weight. */
typedef struct {
    int B, t, steps, Lmem;       /* batch, index of the new position, cache length (positions), memory tokens */
    int D, H, FF;
    const void* x_in;  long long x_in_stride;    /* [B] rows of D bf16 */
    void* x_out;       long long x_out_stride;   /* [B] rows of D bf16: norm3 output */
    void* kv_self;               /* [B, steps, 3D] bf16: row t is WRITTEN (q|k|v of the new position), rows < t are read */
    const void* kv_mem;          /* [B, Lmem, 2D] bf16: k|v of the memory tokens (multihead_attn.in_proj rows D..3D applied once) */
    const float* key_bias; long long key_bias_stride;   /* [B] rows of >= t+1 floats added to the self-attention scores, or NULL */
    const void* w_in;  const float* b_in;        /* self_attn.in_proj   [3D, D] */
    const void* w_so;  const float* b_so;        /* self_attn.out_proj  [D, D] */
    const void* w_q;   const float* b_q;         /* multihead_attn.in_proj rows 0..D (query)  [D, D] */
    const void* w_co;  const float* b_co;        /* multihead_attn.out_proj [D, D] */
    const void* w1;    const float* b1;          /* linear1 [FF, D] */
    const void* w2;    const float* b2;          /* linear2 [D, FF] */
    const float *g1, *be1, *g2, *be2, *g3, *be3; /* norm1..3 weight / bias */
    float eps, scale;            /* LayerNorm eps; 1/sqrt(D/H) */
    int cluster;                 /* workgroups per sample: 1, or 4 (heads / hidden units split over the cluster; needs the three below) */
    float* exch;                 /* [B, 3, cluster, D] fp32 scratch: the partial projections the cluster members exchange */
    unsigned int* sync;          /* [B, 2] {arrival count, generation}: zero before the FIRST launch, never touched by the host afterwards */
    int* err;                    /* optional: set to 1 if a cluster barrier gave up (spin limit) - the outputs are then invalid */
    int fp32;                    /* 0 (a zeroed descriptor, the r02 layout's meaning): x_in / x_out / kv_self / kv_mem and the six weight matrices are
                                  * bf16 as the comments above say; 1 (r03): all of them fp32 - the parity mode's decode step as one launch per
                                  * layer too, nothing rounded between the stages */
} p3_decode_layer_desc;
int p3_decode_layer(const p3_decode_layer_desc* d, void* stream);
int p3_cast(const void* a, int dtype_a, void* b, int dtype_b, int64_t n, void* stream);
/* out[b,t,:] = x[b,t,:] + pos[t,:]  (Decoder: encoder_out + encoder_pos_embed, model_pix2poly.py:171-173) */
int p3_add_pos(const void* x, const float* pos, void* out, int B, int L, int D, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step (a-11: trainer_pix2poly.py:284-351 = forward, CE + 10*BCE, backward, AdamW)
 * ------------------------------------------------------------------------------------------ */
/* attention backward (recompute, no atomics): delta_ws float [B,H,Lq] scratch; dQ/dK/dV use the q/k/v strides, dO the o strides */
int p3_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse, void* dQ,
                     void* dK, void* dV, float* delta_ws, const p3_attn_desc* d, void* stream);
/* weight gradient: C[N,K] += A[M,N]^T B[M,K] (fp32 C, split over M with fp32 atomics) ; column sums (bias gradients) */
int p3_gemm_tn(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int dtype, void* stream);
int p3_colsum(const void* x, float* out, int64_t M, int N, int ld, int dtype, void* stream);
/* dpre = dy * act'(.)  (GELU: saved = pre-activation; ReLU: saved = output) */
/* scale: extra factor on dy (1/(1-p) when a dropout followed the activation: the saved ReLU output is already masked) */
int p3_act_bwd(const void* dy, int dtype_dy, const void* saved, int dtype_saved, void* out, int dtype_out, int64_t n, int act, float scale,
               void* stream);
/* dpos may be NULL (then take it as the column sums of dx viewed as [B, L*D]: no atomics) */
int p3_embed_tokens_bwd(const void* dx, int dtype, const int64_t* tokens, float* demb, float* dpos, int B, int L, int D, void* stream);
/* the embedding part alone for a vocabulary of V rows (nn.Embedding backward of model_pix2poly.py:135,164): tokens outside [0, V) are ignored;
 * V * 256 bytes of LDS per workgroup (falls back to the atomics form above beyond 64 KiB) */
int p3_embed_tokens_bwd_v(const void* dx, int dtype, const int64_t* tokens, float* demb, int B, int L, int D, int V, void* stream);
/* dscale is the centred sum  sum dz*(src - mean)  when mean != NULL (feeds p3_bn_bwd_coeffs) */
int p3_tokens_assemble_bwd(const float* dx, const void* src, int src_ld, int dtype_src, const float* scale, const float* shift, const float* mean,
                           void* dsrc, float* dscale, float* dshift, int B, int np, int D, void* stream);
int p3_pool_pos_bwd(const void* dout, int dtype_dout, void* dy, int dtype_dy, int B, int np, int Din, int Dout, void* stream);
int p3_pair_mean_bwd(const float* dF, void* dfeats, int dtype, int B, int L, int N, int D, int accumulate, void* stream);
/* reverse-mode through all Sinkhorn iterations + slice + softmax in one launch (uv_hist from p3_sinkhorn) */
/* workspace: p3_sinkhorn_bwd_workspace_bytes(B, m, n, iters) bytes of device scratch (per-tile path flags + the per-iteration
 * row / column factor vectors of the linear-domain sweep) */
int p3_sinkhorn_bwd(const float* scores, const float* alpha, int B, int m, int n, int iters, const float* perm,
                    const float* uv_hist, const float* dperm, float* dscores, float* dalpha, void* workspace, void* stream);
int64_t p3_sinkhorn_bwd_workspace_bytes(int B, int m, int n, int iters);
/* nn.CrossEntropyLoss(ignore_index) / nn.BCELoss (trainer_pix2poly.py:91-93): acc[0] = loss sum, acc[1] = #valid rows;
 * backward kernels read the upstream gradient x loss weight from the device scalar `gscale` (no host sync) */
int p3_ce_loss_fwd(const float* logits, int ld, const int64_t* targets, int R, int V, int ignore_index, float* row_lse, float* acc, void* stream);
int p3_ce_loss_bwd(const float* logits, int ld, const int64_t* targets, int R, int V, int ignore_index, const float* row_lse,
                   const float* acc, const float* gscale, void* dlogits, int dtype_out, int ld_out, int Vpad, void* stream);
int p3_bce_loss_fwd(const float* p, const float* y, int64_t n, float* acc, void* stream);
int p3_bce_loss_bwd(const float* p, const float* y, int64_t n, const float* gscale, float* dp, void* stream);
/* Device-side step counter + learning-rate schedule of the optimizer (torch.optim.AdamW bias corrections, and the reference's
 * transformers.get_linear_schedule_with_warmup, train/trainer_pix2poly.py:62-77): reads step[0] = s (optimizer steps done), writes
 * hyper = {base_lr * lambda(s), 1 - beta1^(s+1), 1 - beta2^(s+1)} and step[0] = s + 1.  kind 0 = constant, 1 = linear warm-up / decay.
 * Launch it right before p3_adamw on the same stream (inside the captured step graph). */
int p3_adamw_schedule(long long* step, float* hyper, float base_lr, int kind, int warmup_steps, int total_steps, float beta1, float beta2,
                      void* stream);
/* the same with sched = device float[4] {base_lr, kind, warm-up steps, total steps} (all < 2^24: exact in fp32) read by the kernel: a captured
 * step follows a learning rate / schedule the host rewrites after the capture */
int p3_adamw_schedule_dev(long long* step, float* hyper, const float* sched, float beta1, float beta2, void* stream);
/* torch.optim.AdamW step over a flat parameter arena; hyper = {lr, 1-beta1^t, 1-beta2^t} on the device; optional bf16 shadow */
int p3_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper, float beta1,
             float beta2, float eps, float weight_decay, float grad_scale, void* bf16_shadow, void* stream);

/* ScoreNet backward (model_pix2poly.py:69-112) without the [B,512,N,N] pair tensor; see csrc/scorenet_bwd.hip.
 * p3_gemm_tn_ex: weight gradient with a generated B operand  B' = relu((B (+V)) * b_scale + b_shift)  (b_mode = P3_A_*). */
int p3_gemm_tn_ex(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int dtype, int b_mode,
                  const float* b_scale, const float* b_shift, const void* pair_V, int pair_n, float* colsum /* optional [N]: += column sums of A */,
                  float* slabs /* optional scratch [max_slabs][N][K] fp32: split-M partials stored + reduced instead of fp32 atomics */,
                  int max_slabs, void* stream);
/* tail (dS != NULL, C = 64): dS [B,N,N] (read transposed if transpose) -> dHd = dz*scale, acc = [dscale(C) | dshift(C) | dw4(C) | db4]
 * matrix (dA != NULL, C = 128): dA [R,C] -> dHd = dA*(z>0)*scale (may alias dA), acc = [dscale(C) | dshift(C)] */
int p3_row_affine_bwd(const void* dA, const float* dS, const void* H, const float* scale, const float* shift, const float* mean, const float* w4,
                      void* dHd, float* acc, int64_t R, int C, int N, int transpose, int dtype, void* stream);
/* Two-pass form for train-mode BatchNorm: call once with dHd = NULL (sums only, nothing stored), derive (a, b) with
 * p3_bn_bwd_coeffs, call again with acc = NULL and fix_a / fix_b to write the final gradient dz*scale + a + b*H directly. */
int p3_row_affine_bwd2(const void* dA, const float* dS, const void* H, const float* scale, const float* shift, const float* mean,
                       const float* w4, void* dHd, float* acc, const float* fix_a, const float* fix_b, int64_t R, int C, int N,
                       int transpose, int dtype, void* stream);
/* BatchNorm backward through scale/shift and (train mode) the batch statistics: d(pre) = direct + a[c] + b[c]*pre.
 * `dscale` is the CENTRED sum  sum dz*(pre - mean)  as accumulated by p3_row_affine_bwd / p3_pair_bwd (acc[0:C]).
 * training: bit 0 = batch statistics (a, b non-zero), bit 1 = dgamma / dbeta are accumulated (+=) instead of stored */
int p3_bn_bwd_coeffs(const float* dscale, const float* dshift, const float* gamma, const float* mean, const float* rstd, float count,
                     int training, int C, float* dgamma, float* dbeta, float* a, float* b, void* stream);
int p3_affine_fix(void* dH, const void* H, const float* a, const float* b, int64_t R, int C, int dtype, void* stream);
/* same with a row stride on H (dH stays dense [R, C]) */
int p3_affine_fix_ld(void* dH, const void* H, int ldh, const float* a, const float* b, int64_t R, int C, int dtype, void* stream);
/* pair grid: dA [B*N*N, C] -> dU (written), dV (+=, zero-filled by the caller) [B*N, C] fp32, acc = [dscale(C) | dshift(C)] */
int p3_pair_bwd(const void* dA, const void* U, const void* V, const float* scale, const float* shift, const float* mean, float* dU, float* dV, float* acc,
                int B, int N, int C, int dtype, void* stream);
/* same with a scratch of p3_pair_bwd_workspace_bytes(B, N, C) bytes: the per-block dV partial rows are stored there and summed by a second
 * kernel instead of being added with global fp32 atomics (bf16 path; other dtypes ignore the workspace) */
int64_t p3_pair_bwd_workspace_bytes(int B, int N, int C);
int64_t p3_pair_bwd_workspace_bytes_dt(int B, int N, int C, int dtype);   /* per dtype: the fp32 form takes smaller row chunks (more slabs) */
int p3_pair_bwd_ws(const void* dA, const void* U, const void* V, const float* scale, const float* shift, const float* mean, float* dU, float* dV,
                   float* acc, int B, int N, int C, int dtype, void* workspace, void* stream);
/* p3_gemm (dA2 = dH2 . W2, the input gradient of ScoreNet.conv2, model_pix2poly.py:76) and p3_pair_bwd in ONE launch (bf16; csrc/pair_bwd_mma.hip): a
 * workgroup owns (tile, 8 rows i, all j), the 128 x 256 product tiles of dA2 stay in its MFMA accumulators and are masked / summed there - the
 * [B N^2, 256] gradient (1.2 GB per net at B = 64, N = 192) is neither written nor read.  dH2 [B N^2, 128] bf16 (dense), W2t = conv2.weight transposed
 * [256, 128] bf16, U / V [B N, 256] bf16, scale / shift / mean [256] fp32; outputs as p3_pair_bwd: dU (=), dV (+=, zero it first), acc[0:256] += centred
 * scale sums, acc[256:512] += shift sums.  workspace: p3_pair_bwd_fused_workspace_bytes(B, N) bytes (dV partial rows of the N / 8 row blocks). */
/* From G [N, 2K] = p3_gemm_tn_ex(dH [M, N], H [M, K], P3_A_AFFINE_MASK2) (G | G2 side by side, row stride ldg) and W [N, K] fp32 (the weight of the layer
 * dH belongs to, e.g. ScoreNet.conv3.weight, model_pix2poly.py:78): dW [N, K] += scale G2 + shift G (that layer's weight gradient over relu(bn(H))), and
 * acc[0:K] += sum dz (H - mean), acc[K:2K] += sum dz for dz = [bn(H) > 0] (dH W) - the BatchNorm backward sums of the layer below, WITHOUT a pass over
 * the [M, K] input gradient (r01 - r03: p3_row_affine_bwd pass 1, 290 us per ScoreNet at B = 64). */
int p3_bn_sums_from_g(const float* G, int ldg, const float* W, const float* scale, const float* shift, const float* mean, float* dW, float* acc,
                      int N, int K, void* stream);
int64_t p3_pair_bwd_fused_workspace_bytes(int B, int N);
int p3_pair_bwd_fused(const void* dH2, const void* W2t, const void* U, const void* V, const float* scale, const float* shift, const float* mean,
                      float* dU, float* dV, float* acc, int B, int N, void* workspace, void* stream);
int p3_pair_stats_bwd(const void* U, const void* V, const float* a, const float* b, float* dU, float* dV, int B, int N, int C, int dtype,
                      void* stream);
/* the same launch for P3_F32X3 (csrc/pair_bwd_x3.hip): everything fp32 in memory (dH2 [B N^2, 128], W2t [256, 128], U / V [B N, 256]), the products as
 * three bf16 MFMA terms with fp32 accumulation; W2t's hi / lo split lives in registers.  Same outputs, same workspace size. */
int p3_pair_bwd_fused_x3(const float* dH2, const float* W2t, const float* U, const float* V, const float* scale, const float* shift, const float* mean,
                         float* dU, float* dV, float* acc, int B, int N, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * FFL / *CNN encoder tails (models/fusion_layers/early_fusion_vit_cnn.py:87-104, models/vision_transformer/vit_cnn.py:45-57,
 * models/ffl/model_ffl.py:53-96).  The 3x3 convolutions are p3_gemm with P3_A_CONV3X3 / P3_A_CONV3X3_AFFINE_RELU.
 * ------------------------------------------------------------------------------------------ */
/* nn.Upsample(size=(H,W), mode='bilinear', align_corners=False) of the token map: src tokens [B, src_tok_per_img, C] starting at token
 * src_tok_off (1 = CLS dropped) viewed as [h, w] -> NHWC dst [B, H, W, ld] */
int p3_upsample_bilinear(const void* src, int dtype_src, void* dst, int dtype_dst, int B, int h, int w, int C, int H, int W, int ld,
                         int src_tok_off, int src_tok_per_img, void* stream);
/* Conv1x1(256 -> n_out) on relu(x*scale+shift) + Sigmoid (act 0) | post_mul*Tanh (act 1); out NCHW fp32 [B, n_out, H*W]; optionally the
 * first output is also written to copy_dst[r*copy_ld] (the seg channel concatenated to the features, model_ffl.py:87-89) */
int p3_head1x1(const void* X, int ld, int dtype, const float* scale, const float* shift, const float* W, const float* bias, int n_out,
               int act, float post_mul, float* out_nchw, void* copy_dst, int copy_ld, int64_t R, int64_t HW, void* stream);
int p3_nhwc_to_nchw(const void* X, int ld, int dtype, const float* scale, const float* shift, float* out, int B, int C, int64_t HW, void* stream);

/* ---- backward of the FFL / *CNN tail (autograd of models/ffl/model_ffl.py:71-96 and of the *CNN encoders' `proj`) ----
 * p3_head1x1_bwd: through Conv1x1 + Sigmoid | post_mul*Tanh and the BatchNorm + ReLU in front of it.  out_nchw / dout_nchw are the
 *   forward output and its gradient [B, n_out, HW] fp32.  dHd [R,256] = dy*scale (add the batch-statistics term with p3_affine_fix).
 *   acc (zeroed by the caller) += [dscale centred (256) | dshift (256) | dW (n_out*256) | db (n_out)].
 * p3_affine_relu_bwd256: dA = gradient w.r.t. relu(bn(H)) -> dHd = dA*[bn(H) > 0]*scale, acc += [dscale centred | dshift].
 * p3_pad_nhwc: [B*H*W, ld_src] -> zero-bordered [B, H+2, W+2, Cp] image (channels < c_aff through relu(x*scale+shift) when scale is
 *   given) that the shifted-row weight-gradient GEMMs (p3_gemm_tn) read.
 * p3_upsample_bilinear_bwd: adjoint of p3_upsample_bilinear (separable, gather form, deterministic); tmp = fp32 [B, H, w, C]. */
int p3_head1x1_bwd(const void* H, int dtype, const float* scale, const float* shift, const float* mean, const float* W, int n_out,
                   const float* out_nchw, const float* dout_nchw, int act, float post_mul, void* dHd, float* acc, int64_t R, int64_t HW,
                   void* stream);
int p3_affine_relu_bwd256(const void* dA, const void* H, int ldh, int dtype, const float* scale, const float* shift, const float* mean,
                          void* dHd, float* acc, int64_t R, void* stream);
int p3_pad_nhwc(const void* src, int ld_src, int dtype, const float* scale, const float* shift, int c_aff, int C, int Cp, void* dst, int B, int H,
                int W, void* stream);

/* ---- HiSup head set (models/hisup/model_hisup.py:38-64 `ECA`, :122-226 `EncoderDecoder.forward_common` after the encoder).
 * The 3x3 / 1x1 convolutions, BatchNorm statistics and zero-bordered images are p3_gemm (P3_A_CONV3X3 + conv_pad, column sums), p3_bn_finalize
 * and p3_pad_nhwc; these three entries are the rest.  Maps are token-major [B*H*W, ld] in `dtype`, a producer's BatchNorm + ReLU travels as
 * per-channel (scale, shift) and is applied where the map is read.
 * p3_nchw_to_nhwc:    the encoder's NCHW fp32 feature map (what the reference hands to the heads, model_hisup.py:205) -> token-major.
 * p3_eca_gate:        pooled[b,c] = mean_hw(relu(bn(a1)) + relu(bn(a2)))  (ECA.avg_pool(x1 + x2)), gate = sigmoid(conv1d_k(pooled) over c)
 *                     (Conv1d(1, 1, k, padding k/2, no bias), model_hisup.py:47,59-61); k odd.
 * p3_affine_relu_mix: out = f(a) * gate[b, c] + g(b)  with f / g = relu(x*scale + shift) when scale is given, identity otherwise; gate and the
 *                     second source are optional: x2 * y (ECA.forward, :63), feature + attention feature (:217-218), the channel halves of
 *                     torch.cat((features, afm_conv)) (:222). */
int p3_nchw_to_nhwc(const float* X, void* out, int ld, int dtype, int B, int C, int64_t HW, void* stream);
int p3_eca_gate(const void* a1, int ld1, const float* scale1, const float* shift1, const void* a2, int ld2, const float* scale2,
                const float* shift2, const float* conv_w, int k, float* pooled, float* gate, int B, int64_t HW, int C, int dtype, void* stream);
int p3_affine_relu_mix(void* out, int ld_out, const void* a, int ld_a, const float* scale_a, const float* shift_a, const float* gate,
                       const void* b, int ld_b, const float* scale_b, const float* shift_b, int64_t R, int C, int64_t HW, int dtype,
                       void* stream);
int p3_upsample_bilinear_bwd(const void* dUp, int dtype, float* tmp, void* dtok, int B, int h, int w, int C, int H, int W, int tok_off,
                             int tok_per_img, void* stream);


/* ------------------------------------------------------------------------------------------
 * Input pipeline on the device.  The reference augments per sample on CPU workers: albumentations
 * D4(p=1) + Normalize + ToTensorV2 (datasets/build_datasets.py:53-75) and apply_d4_augmentations_to_lidar (datasets/p3_coco.py:115-164).
 * group[b] in 0..7 = albumentations' D4 element of tile b: e, r90, r180, r270, v, hvt, h, t.
 * ------------------------------------------------------------------------------------------ */
/* src u8 [B,H,W,C] (HWC as rasterio/albumentations hold it) -> dst f32 [B,C,H,W] = ToTensorV2(Normalize(D4_g(img))):
 * dst = ((float)px - sub[c]) * mul[c] with the HOST-side constants sub = mean*max_pixel_value, mul = 1/(std*max_pixel_value) (fp32).
 * group may be NULL (no augmentation: validation / prediction). */
int p3_image_prepare(const uint8_t* src, const int32_t* group, float* dst, int B, int H, int W, int C, const float* sub, const float* mul,
                     void* stream);
/* in-place D4 of the jagged point list values [total,3] (x, y, z) / offsets [B+1] around the centre (cx, cy) = (in_width // 2,
 * in_height // 2): subtract centre, swap / negate as p3_coco.py:135-158, add centre - fp32, same operation order */
int p3_points_d4(float* values, const int64_t* offsets, const int32_t* group, int B, int64_t total, float cx, float cy, void* stream);
/* FFL ground truth of a batch (datasets/p3_coco.py:254-296 + apply_augmentations_to_ffl_crossfield_angle :166-205), every input optional:
 * gt_polygons_u8 [B,H,W,3] -> out_gt f32 [B,3,H,W] = clamp(u8 / 255, 0, 1); crossfield_angle_u8 [B,H,W] -> out_angle f32 [B,1,H,W] =
 * ((u8 * pi / 255 + pi/2) % pi) rotated / mirrored with the tile; distances / sizes f32 [B,H,W] -> [B,1,H,W] permuted only.
 * group[b] = the tile's D4 element (NULL: none). */
int p3_ffl_targets_prepare(const uint8_t* gt_polygons_u8, const uint8_t* crossfield_angle_u8, const float* distances, const float* sizes,
                           const int32_t* group, int B, int H, int W, float* out_gt, float* out_angle, float* out_distances, float* out_sizes,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * FFL frame-field loss, forward + gradients: build_combined_loss / MultiLoss of models/ffl/losses.py:84-141,237-316 with
 * the shipped config/model/ffl.yaml (seg = interior channel only, crossfield on, no freq / dist / size weights, seg.type "bool"):
 *   0 seg (bce_coef * BCE(seg, gt0 > 0.98) + dice_coef * dice(seg, gt0), :318-365)   1 crossfield_align (:368-386)
 *   2 crossfield_align90 (:389-406)   3 crossfield_smooth (:409-419)   4 seg_interior_crossfield (:220-235, :422-444)
 * seg [B,1,H,W], crossfield [B,4,H,W], gt_polygons_image [B,3,H,W] (interior, edge, vertex), gt_crossfield_angle [B,1,H,W]: fp32 NCHW.
 * seg_weights [B,1,H,W] or NULL: the per-pixel BCE weights of compute_seg_loss_weigths (:150-205; all ones in the shipped config).
 * coef[5] (host) = weight_i(epoch) / norm_i.  losses[6] (device) = the five RAW losses and total = sum coef_i * loss_i.
 * dseg / dcrossfield (both or neither): d total / d input.  workspace: p3_ffl_loss_workspace_bytes(B, H, W) bytes of device scratch.
 * ------------------------------------------------------------------------------------------ */
int p3_ffl_loss(const float* seg, const float* crossfield, const float* gt_polygons_image, const float* gt_crossfield_angle,
                const float* seg_weights, int B, int H, int W, const float* coef, float bce_coef, float dice_coef, float* losses, float* dseg,
                float* dcrossfield, void* workspace, void* stream);
int64_t p3_ffl_loss_workspace_bytes(int B, int H, int W);

/* ------------------------------------------------------------------------------------------
 * HiSup attraction field map (the reference's only native kernel: models/hisup/afm_module/afm_op/cuda/afm.cu:29-112,
 * bound as afm(lines, shape_info, height, width) in models/hisup/model_hisup.py:95).
 * lines [L,4] fp32 (x1, y1, x2, y2 in source pixels), shape_info [B,4] int32 = (first line, one past the last line, source height,
 * source width) per tile -> afmap [B,2,height,width] fp32 = -sgn(a) log(|a| / size + 1e-6) of the offset a to the closest segment
 * point, aflabel [B,1,height,width] int32 = index of that segment inside the tile (first of equally close ones); tiles without
 * segments get zeros.
 * ------------------------------------------------------------------------------------------ */
int p3_afm(const float* lines, const int32_t* shape_info, int B, int height, int width, float* afmap, int32_t* aflabel, void* stream);

#ifdef __cplusplus
}
#endif
#endif
