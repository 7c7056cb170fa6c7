// p3hip - the ScoreNet's conv3 forward (model_pix2poly.py:78, 128 -> 64 channels over R = B*N*N pair rows) for P3_F32X3: fp32 rows in memory, products as
// bf16 x 3 - the weight-stationary streaming form of rows_gemm.hip (which is the bf16 kernel of this launch).
//
//   Y = relu(X * a_scale + a_shift) W^T + bias,  per-workgroup column sums / sums of squares of Y (train-mode BatchNorm-3 statistics: fixed-order partials ->
//   p3_det_reduce2, bit-reproducible)
//
// Until r05 this ran on p3_gemm's tile kernel (P3_A_AFFINE_RELU): 713 us per net for 1.8 GB of traffic (2.5 TB/s) - four k-steps per tile between a prologue
// and an epilogue.  Here a WAVE owns 32 of the 64 output channels (its slice of W split into hi / lo once: 64 registers) and streams 32-row groups on its own; the
// two waves of a pair take the same rows (the second read is an L1 / L2 hit).  No barrier in the loop; a group's rows arrive in two 64-deep halves, the next
// half always in flight (32 registers each).  The generated operand is computed and split in registers in the MFMA layout (lane: row = lane % 32, k = 8 (lane /
// 32) .. + 8); in the accumulator layout a lane holds one channel and 16 rows: 4-byte stores, 32 lanes = 128 contiguous bytes of an output row - no LDS image.
#include "p3_common.h"

#define P3_ROWS_SKIP 0x7fffffff

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

struct RXArgs {
    const float* X;         // [R, 128]
    const float* W;         // [64, 128]
    float* Y;               // [R, 64]
    const float* bias;      // [64] or NULL
    const float* a_scale;   // [128]
    const float* a_shift;   // [128]
    float* stats;           // [gridDim.x][128] or NULL
    int64_t groups;         // R / 32
};

__device__ __forceinline__ void rx_split8(const float (&v)[8], u32x4_t& h, u32x4_t& l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
        h[k] = hw;
        l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
    }
}

__global__ __launch_bounds__(256, 2) void rows_x3_fwd_kernel(RXArgs g) {
    constexpr int K = 128, N = 64;
    __shared__ float tab[2 * K];
    __shared__ float red[2][2 * N];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l32 = lane & 31, hi = lane >> 5;
    const int nb = w & 1, stream = w >> 1;                              // channel half, row stream of the workgroup
    for (int i = tid; i < K; i += 256) { tab[i] = g.a_scale[i]; tab[K + i] = g.a_shift[i]; }
    __syncthreads();
    const int ch = nb * 32 + l32;
    bf16x8_t wh[8], wl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float* wp = g.W + (int64_t)ch * K + 16 * s + 8 * hi;
        const float4 x0 = *reinterpret_cast<const float4*>(wp), x1 = *reinterpret_cast<const float4*>(wp + 4);
        const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        u32x4_t h, l;
        rx_split8(v, h, l);
        wh[s] = __builtin_bit_cast(bf16x8_t, h); wl[s] = __builtin_bit_cast(bf16x8_t, l);
    }
    // scale / shift of this lane's k positions: k = 16 s + 8 hi + e
    const float bias = g.bias ? g.bias[ch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const int64_t nw = (int64_t)gridDim.x * 2;
    int64_t gi = (int64_t)blockIdx.x * 2 + stream;
    float4 xa[4][2], xb[4][2];
    auto fetch = [&](float4 (&x)[4][2], int64_t grp, int half) __attribute__((always_inline)) {
        const float* xp = g.X + (grp * 32 + l32) * K + half * 64 + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) { x[s][0] = *reinterpret_cast<const float4*>(xp + 16 * s); x[s][1] = *reinterpret_cast<const float4*>(xp + 16 * s + 4); }
    };
    f32x16 acc;
    auto mma_half = [&](const float4 (&x)[4][2], int half) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k0 = half * 64 + 16 * s + 8 * hi;
            const float4 c0 = *reinterpret_cast<const float4*>(tab + k0), c1 = *reinterpret_cast<const float4*>(tab + k0 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(tab + K + k0), h1 = *reinterpret_cast<const float4*>(tab + K + k0 + 4);
            const float a[8] = {fmaxf(fmaf(x[s][0].x, c0.x, h0.x), 0.f), fmaxf(fmaf(x[s][0].y, c0.y, h0.y), 0.f), fmaxf(fmaf(x[s][0].z, c0.z, h0.z), 0.f),
                                fmaxf(fmaf(x[s][0].w, c0.w, h0.w), 0.f), fmaxf(fmaf(x[s][1].x, c1.x, h1.x), 0.f), fmaxf(fmaf(x[s][1].y, c1.y, h1.y), 0.f),
                                fmaxf(fmaf(x[s][1].z, c1.z, h1.z), 0.f), fmaxf(fmaf(x[s][1].w, c1.w, h1.w), 0.f)};
            u32x4_t ah_, al_;
            rx_split8(a, ah_, al_);
            const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, ah_), al = __builtin_bit_cast(bf16x8_t, al_);
            const int ks = half * 4 + s;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh[ks], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                          // one k-step's operand at a time (the scheduler otherwise builds all four first: spills)
        }
    };
    if (gi < g.groups) fetch(xa, gi, 0);
    for (; gi < g.groups; gi += nw) {
        fetch(xb, gi, 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        mma_half(xa, 0);
        if (gi + nw < g.groups) fetch(xa, gi + nw, 0);
        mma_half(xb, 1);
        float* yp = g.Y + (gi * 32 + 4 * hi) * N + ch;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + bias;
            s1 += v; s2 = fmaf(v, v, s2);
            yp[((r & 3) + 8 * (r >> 2)) * N] = v;
        }
    }
    if (g.stats) {
        // lanes l and l + 32 hold the two row halves of one channel; then the workgroup's two row streams in order (fixed order: bit-reproducible)
        const float a1 = s1 + __shfl_xor(s1, 32, 64), a2 = s2 + __shfl_xor(s2, 32, 64);
        if (hi == 0) { red[stream][ch] = a1; red[stream][N + ch] = a2; }
        __syncthreads();
        if (tid < 2 * N) g.stats[(int64_t)blockIdx.x * 2 * N + tid] = red[0][tid] + red[1][tid];
    }
}

}  // namespace

// p3_gemm's hook for P3_F32X3: P3_ROWS_SKIP when the problem is not the conv3-forward shape (the caller goes on with its tile kernel), else the launch status
int p3_rows_x3_try(const void* A, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s) {
    if (d->dtype_in != P3_F32 || d->dtype_out != P3_F32 || d->M % 32 != 0 || d->M < 4096) return P3_ROWS_SKIP;
    if (d->a_mode != P3_A_AFFINE_RELU || d->K != 128 || d->N != 64 || d->bwd_saved) return P3_ROWS_SKIP;
    if (d->lda != d->K || d->ldb != d->K || d->ldc != d->N) return P3_ROWS_SKIP;
    if (d->act != P3_ACT_NONE || d->residual || d->aux || (d->drop.seed && d->drop.p > 0.f)) return P3_ROWS_SKIP;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) % 16 != 0) return P3_ROWS_SKIP;
    RXArgs g;
    g.X = (const float*)A; g.W = (const float*)W; g.Y = (float*)C; g.bias = d->bias; g.a_scale = d->a_scale; g.a_shift = d->a_shift; g.stats = nullptr;
    g.groups = d->M / 32;
    int64_t blocks = (g.groups + 1) / 2;
    if (blocks > 512) blocks = 512;                      // 2 workgroups / CU (180 registers): every wave pair walks 72 groups at the bench size
    float* scratch = nullptr;
    if (d->colsum) {
        const int nch = (int)((blocks + 127) / 128);
        scratch = p3_reduce_scratch(blocks * 128 + (int64_t)nch * 128);
        if (!scratch) return P3_ROWS_SKIP;
        g.stats = scratch;
    }
    if (p3_tracing()) p3_note_kernel("rows_x3_fwd_kernel");
    hipLaunchKernelGGL(rows_x3_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (scratch) return p3_det_reduce2(scratch, (int)blocks, 128, scratch + blocks * 128, d->colsum, d->colsumsq, 64, 128, 1, s);
    return P3_OK;
}
