// p3hip — common device/host helpers (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/p3hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even like torch .to(bfloat16); native casts so that hipcc emits gfx950's v_cvt_pk_bf16_f32 (one VALU op per PAIR;
// the integer RNE sequence this replaces cost ~5 ops per element and sat in every epilogue and in the attention P / dS packing)
typedef float p3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const p3_f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, p3_bf16x2));
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- counter-based dropout (p3_dropout) ---------------------------------------------------------------------------------
// Element (row, col) of a site: bits = hash32(rowkey(row) ^ colkey(col >> 1)), 16 bits per element (low half: even col, high half:
// odd col); kept iff bits16 >= p * 65536.  A lane that owns one row (attention fwd / dQ, GEMM epilogue) pays one 2-multiply hash
// per TWO elements; v_mul_lo_u32 is quarter rate on CDNA, so the multiply count is what matters.
__device__ __forceinline__ uint32_t p3_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
struct DropKey { uint32_t k0, k1, thresh; float inv_keep; bool on; };
__device__ __forceinline__ DropKey drop_key(const p3_dropout& dr) {
    DropKey k; k.on = dr.seed != nullptr && dr.p > 0.f; k.k0 = 0; k.k1 = 0; k.thresh = 0; k.inv_keep = 1.f;
    if (k.on) {
        const unsigned long long s = *dr.seed;
        k.k0 = p3_hash32((uint32_t)s ^ (dr.site * 0x9E3779B9u));
        k.k1 = p3_hash32((uint32_t)(s >> 32) + dr.site * 0x85EBCA6Bu + 0x1234567u);
        k.thresh = (uint32_t)(dr.p * 65536.f + 0.5f);
        if (k.thresh > 65535u) k.thresh = 65535u;
        k.inv_keep = 65536.f / (float)(65536u - k.thresh);
    }
    return k;
}
__device__ __forceinline__ uint32_t drop_rowkey(const DropKey& k, uint64_t row) {
    return (uint32_t)row * 0x9E3779B1u + (uint32_t)(row >> 32) * 0x7FEB352Du + k.k0;
}
__device__ __forceinline__ uint32_t drop_colkey(const DropKey& k, uint32_t col) { return (col >> 1) * 0x85EBCA77u + k.k1; }
__device__ __forceinline__ uint32_t drop_bits(uint32_t rowkey, uint32_t colkey) { return p3_hash32(rowkey ^ colkey); }
__device__ __forceinline__ bool drop_keep_lo(const DropKey& k, uint32_t bits) { return (bits & 0xffffu) >= k.thresh; }
__device__ __forceinline__ bool drop_keep_hi(const DropKey& k, uint32_t bits) { return (bits >> 16) >= k.thresh; }
__device__ __forceinline__ bool drop_keep(const DropKey& k, uint64_t row, uint32_t col) {
    const uint32_t bits = drop_bits(drop_rowkey(k, row), drop_colkey(k, col));
    return (col & 1u) ? drop_keep_hi(k, bits) : drop_keep_lo(k, bits);
}

// XCD-aware bijective block remap: hardware hands consecutive workgroup ids round-robin to the 8 XCDs (each with its own 4 MiB L2);
// this gives XCD x the CONSECUTIVE logical ids [x*n/8, (x+1)*n/8), so blocks that share operands (same attention head, same
// M-split of a weight gradient, same A row panel) run on one XCD and hit its L2 instead of fetching the operand 8 times.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Attention launches: logical block id -> (128-row block, (batch, head) pair).  L = 785 = 6 x 128 + 17 (ViT) and 385 = 3 x 128 + 1 (decoder)
// leave a nearly empty last block per pair that still walks the whole key / query sequence: one wave live, a slot held for most of a full
// block's time, and a grid of 7/6 (4/3) x the full blocks - ViT forward: 2304 full blocks = exactly 3 rounds of the 768 resident slots, the
// 384 tail blocks mixed in made it 3.5.  mode 1: inside each XCD's consecutive id range (xcd_remap) the TAIL blocks come first - short, they
// finish while the first round of full blocks starts, and the launch ends on whole rounds of full blocks; the blocks of one pair stay on one
// XCD (K / V in one L2).  Needs pairs % 8 == 0 and a ragged last block; anything else keeps the pair-major order (mode 0).
__device__ __forceinline__ void attn_block_of(int lid, int nblk, int L, int pairs, int mode, int& blk, int& pair) {
    if (mode == 1 && nblk > 1 && (L & 127) != 0 && (pairs & 7) == 0) {
        const int p8 = pairs >> 3, per_xcd = p8 * nblk;
        const int x = lid / per_xcd, local = lid - x * per_xcd;
        if (local < p8) { pair = x * p8 + local; blk = nblk - 1; }
        else { const int l2 = local - p8; pair = x * p8 + l2 / (nblk - 1); blk = l2 % (nblk - 1); }
    } else { blk = lid % nblk; pair = lid / nblk; }
}
int p3_attn_order(void);      // host: 1 unless P3_ATTN_TAIL_FIRST=0 (A/B switch)

// row index of accumulator register r of a 32x32 MFMA C/D fragment (col = lane & 31)
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// exact (erf) GELU and its derivative from ONE exponential: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 parity
// tolerances by four orders), exp(-z^2) with z = x/sqrt(2) is also the Gaussian pdf of GELU'.  ~20 VALU ops for both values; the
// libm erff + expf pair this replaces made the fc1 epilogue cost as much as its whole K = 384 main loop (r01: 146 vs 97 us).
__device__ __forceinline__ void gelu_and_grad(float x, float& h, float& g) {
    const float z = x * 0.70710678118654752440f;
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    const float e = __expf(-z * z);
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float erf_abs = fmaf(-poly, e, 1.0f);
    const float cdf = 0.5f * (1.0f + copysignf(erf_abs, z));
    h = x * cdf;
    g = fmaf(x * 0.39894228040143267794f, e, cdf);
}
__device__ __forceinline__ float gelu_erf(float x) { float h, g; gelu_and_grad(x, h, g); return h; }

// a workgroup's partial of value idx (n values per workgroup): into `slab` [gridDim.x][n] - deterministic mode, summed in workgroup order in float64 by
// p3_det_reduce right behind the launch - or, without a slab, an fp32 atomic onto out[idx].  EVERY workgroup of the grid must commit every idx (zeros included).
__device__ __forceinline__ void p3_commit(float* out, float* slab, int n, int idx, float v) {
    if (slab) slab[(int64_t)blockIdx.x * n + idx] = v;
    else atomicAdd(out + idx, v);
}

static inline int p3_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void p3_set_error(const char* msg);
int p3_tracing(void);                     // p3_trace_kernels(1): launch sites record the kernel they picked (p3_last_kernel)
void p3_note_kernel(const char* name);
// deterministic reductions (det_reduce.hip): scratch for workgroup partials (NULL: atomics path) and the fixed-order float64 reduce
float* p3_det_scratch(int64_t floats, int dtype);
int p3_det_reduce(const float* parts, int nparts, int64_t stride, float* out, int nvals, int accumulate, hipStream_t s);
float* p3_reduce_scratch(int64_t floats);     // the registered scratch for any dtype (per-tile partials), NULL if none / too small
float* p3_reduce_park(int64_t floats, int nparts, int nvals, int split, float* out, float* out2);
float* p3_colsum_parts(int splits, int N, float* colsum, int dtype, int* parked);   // bias-gradient partial column sums: a parked slot (*parked = 1) or the deterministic scratch
   // deferred parameter-gradient reduce (det_reduce.hip): slot or NULL
int p3_det_reduce2(const float* parts, int nparts, int64_t stride, float* tmp, float* out, float* out2, int split, int nvals, int accumulate, hipStream_t s);
#define P3_CHECK(cond, code, msg) \
    do {                          \
        if (!(cond)) {            \
            p3_set_error(msg);    \
            return (code);        \
        }                         \
    } while (0)
#define P3_LAUNCH_CHECK()                              \
    do {                                               \
        hipError_t e_ = hipGetLastError();             \
        if (e_ != hipSuccess) {                        \
            p3_set_error(hipGetErrorString(e_));       \
            return (int)e_;                            \
        }                                              \
    } while (0)
