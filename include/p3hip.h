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
 *   - dtype codes: P3_F32 = 0 (exact fp32 path, fp32 MFMA), P3_BF16 = 1 (bf16 storage, fp32 accumulate).
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

#define P3_ACT_NONE 0
#define P3_ACT_GELU 1 /* exact erf GELU (timm Mlp act_layer=nn.GELU) */
#define P3_ACT_RELU 2

#define P3_A_PLAIN 0
#define P3_A_CONV3X3 1 /* A is an NHWC map [B,H,W,lda]; K = 9*C, zero padding 1 (implicit GEMM) */
#define P3_A_AFFINE_RELU 2 /* A'[m,k] = relu(A[m,k]*a_scale[k] + a_shift[k])  (BN+ReLU folded into the load) */
#define P3_A_PAIR_AFFINE_RELU 3 /* A'[(b,i,j),k] = relu((U[b,i,k]+V[b,j,k])*a_scale[k]+a_shift[k]) ScoreNet conv1 */

int p3_version(void);
const char* p3_last_error_string(void);

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
typedef struct {
    int M, N, K;
    int lda, ldb, ldc;
    int dtype_in;  /* dtype of A and W */
    int dtype_out; /* dtype of C, aux and residual */
    int act;
    int a_mode;
    const float* bias;    /* [N] or NULL */
    const void* residual; /* [M,N] (ldr) or NULL; added after the activation */
    int ldr;
    void* aux;            /* optional [M,N] (ldc): pre-activation values (for backward) */
    int conv_H, conv_W, conv_C; /* P3_A_CONV3X3 */
    const float* a_scale; /* [K] for the AFFINE modes */
    const float* a_shift; /* [K] */
    const void* pair_V;   /* P3_A_PAIR_AFFINE_RELU: V [B*n, K] (A is U [B*n, K]); M = B*n*n */
    int pair_n;
    float* colsum;        /* optional [N]: += sum over rows of (A'W^T + bias)   (train-mode BatchNorm statistics) */
    float* colsumsq;      /* optional [N]: += sum of squares */
} p3_gemm_desc;
int p3_gemm(const void* A, const void* W, void* C, const p3_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim:  y = (x - mean) / sqrt(var + eps) * gamma + beta
 * timm Block.norm1/norm2/VisionTransformer.norm (eps 1e-6); nn.TransformerDecoderLayer.norm1..3
 * (eps 1e-5, models/pix2poly/model_pix2poly.py:138).  x dtype_in, y dtype_out; optional save of
 * mean / rstd (float[rows]) for the backward pass.
 * ------------------------------------------------------------------------------------------ */
int p3_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int cols, int ldx, int ldy,
                 float eps, int dtype_in, int dtype_out, float* save_mean, float* save_rstd, void* stream);
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
} p3_attn_desc;
int p3_attention(const void* Q, const void* K, const void* V, void* O, const p3_attn_desc* d, void* stream);
/* backward: dQ,dK,dV given dO, O, lse (same addressing as forward; dq/dk/dv strides = q/k/v strides) */
int p3_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, void* dQ, void* dK,
                     void* dV, float* delta_ws, const p3_attn_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
