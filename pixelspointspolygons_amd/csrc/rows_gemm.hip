// p3hip - weight-stationary "rows" products for the ScoreNet's thin 1x1 convolutions (model_pix2poly.py:69-112: conv3 128 -> 64 and its input
// gradient 64 -> 128 over R = B*N*N = 2.36 M pair rows at the bench size).
//
// These products are HBM streams with a little MFMA work per row (K, N <= 128: 16 KB of weights): the tiled GEMM (gemm.hip) spends its time in
// workgroup barriers and tile prologues / epilogues, not in its 2..4 k-steps (r04 profile: conv3 forward 393 us for 906 MB = 2.3 TB/s; the input
// gradient with the BatchNorm-2 epilogue ~490 us for 1.5 GB).  Here every WAVE keeps the whole weight matrix as MFMA B fragments in registers
// (64 VGPRs) and streams 32-row groups on its own: no workgroup barrier in the loop, the next group's rows are in flight while this one is
// multiplied, and the epilogue goes through a wave-private fp32 image in LDS so that every global access is a 16-byte piece of a full row.
//
//   MODE 0 (conv3 forward):   Y = relu(X * a_scale + a_shift) W^T + bias,  per-workgroup column sums / sums of squares of the fp32 Y (train-mode
//                             BatchNorm statistics: fixed-order partials -> p3_det_reduce2, bit-reproducible)
//   MODE 1 (conv3 backward):  Y = [H s + h > 0] (X W^T) s + a + b H  (P3_ACT_BN_RELU: the BatchNorm-2 / ReLU backward as the epilogue)
//
// MFMA 32x32x16 bf16: A = the 32 data rows (lane: row = lane % 32, k = 8 * (lane / 32) .. + 8: one 16-byte load), B = weights (lane: out channel
// = lane % 32), D: lane = out channel, 16 rows -> the column sums are plain per-lane adds.
#include "p3_common.h"

#define P3_ROWS_SKIP 0x7fffffff     // "not one of my shapes" (no P3_E* / hipError_t value)

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

struct RGArgs {
    const bf16_t* X;        // [R, K]
    const bf16_t* W;        // [N, K]
    bf16_t* Y;              // [R, N]
    const float* bias;      // [N] or NULL            (MODE 0)
    const float* a_scale;   // [K]                    (MODE 0)
    const float* a_shift;   // [K]
    float* stats;           // [gridDim.x][2 N] or NULL (MODE 0)
    const bf16_t* H;        // [R, N]                 (MODE 1)
    const float* bn;        // [4, N]                 (MODE 1)
    int64_t groups;         // R / 32
};

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int K, int N, int MODE>
__global__ __launch_bounds__(256, 2) void rows_gemm_kernel(RGArgs g) {
    constexpr int KS = K / 16, NB = N / 32, P = N + 4, CH = N / 8;     // k-steps, 32-channel blocks, image pitch (floats), 8-channel chunks per row
    constexpr int TASKS = 32 * CH / 64;                                 // (row, chunk) tasks per lane in the row-major pass
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l32 = lane & 31, hi = lane >> 5;
    float* img = lds + w * 32 * P;
    float* tab = lds + 4 * 32 * P;                                      // MODE 0: a_scale | a_shift
    if constexpr (MODE == 0) {
        for (int i = tid; i < K; i += 256) { tab[i] = g.a_scale[i]; tab[K + i] = g.a_shift[i]; }
        __syncthreads();
    }
    u32x4_t wf[NB][KS];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s = 0; s < KS; ++s) wf[nb][s] = *reinterpret_cast<const u32x4_t*>(g.W + (int64_t)(nb * 32 + l32) * K + 16 * s + 8 * hi);
    float bias[NB], s1[NB], s2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { bias[nb] = (MODE == 0 && g.bias) ? g.bias[nb * 32 + l32] : 0.f; s1[nb] = 0.f; s2[nb] = 0.f; }
    // row-major pass geometry: lane -> fixed 8-channel chunk, rows rsel + (64 / CH) * t
    const int c8 = (lane % CH) * 8, rsel = lane / CH;
    constexpr int RSTEP = 64 / CH;
    float bsc[8], bsh[8], ba[8], bb[8];
    if constexpr (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { bsc[k] = g.bn[c8 + k]; bsh[k] = g.bn[N + c8 + k]; ba[k] = g.bn[2 * N + c8 + k]; bb[k] = g.bn[3 * N + c8 + k]; }
    }
    const int64_t nw = (int64_t)gridDim.x * 4;
    int64_t gi = (int64_t)blockIdx.x * 4 + w;
    u32x4_t xa[KS], xn[KS];
    if (gi < g.groups) {
        const bf16_t* xp = g.X + (gi * 32 + l32) * K + 8 * hi;
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = *reinterpret_cast<const u32x4_t*>(xp + 16 * s);
    }
    for (; gi < g.groups; gi += nw) {
        const int64_t gn = gi + nw;
        if (gn < g.groups) {
            const bf16_t* xp = g.X + (gn * 32 + l32) * K + 8 * hi;
#pragma unroll
            for (int s = 0; s < KS; ++s) xn[s] = *reinterpret_cast<const u32x4_t*>(xp + 16 * s);
        }
        u32x4_t hv[TASKS];
        if constexpr (MODE == 1) {
#pragma unroll
            for (int t = 0; t < TASKS; ++t) hv[t] = *reinterpret_cast<const u32x4_t*>(g.H + (gi * 32 + rsel + RSTEP * t) * N + c8);
        }
        f32x16 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4_t a = xa[s];
            if constexpr (MODE == 0) {
                const float4 sc0 = *reinterpret_cast<const float4*>(tab + 16 * s + 8 * hi), sc1 = *reinterpret_cast<const float4*>(tab + 16 * s + 8 * hi + 4);
                const float4 sh0 = *reinterpret_cast<const float4*>(tab + K + 16 * s + 8 * hi), sh1 = *reinterpret_cast<const float4*>(tab + K + 16 * s + 8 * hi + 4);
                const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
                const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = fmaxf(fmaf(__uint_as_float(a[q] << 16), sc[2 * q], sh[2 * q]), 0.f);
                    const float hi_ = fmaxf(fmaf(__uint_as_float(a[q] & 0xffff0000u), sc[2 * q + 1], sh[2 * q + 1]), 0.f);
                    a[q] = pack_bf2(lo, hi_);
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, wf[nb][s]), acc[nb], 0, 0, 0);
        }
        // fp32 image of the 32 x N tile (wave-private: LDS operations of one wave execute in order)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[nb][r] + bias[nb];
                if constexpr (MODE == 0) { s1[nb] += v; s2[nb] += v * v; }
                img[crow32(r, hi) * P + nb * 32 + l32] = v;
            }
        wave_lds_fence();
#pragma unroll
        for (int t = 0; t < TASKS; ++t) {
            const int row = rsel + RSTEP * t;
            const float4 v0 = *reinterpret_cast<const float4*>(img + row * P + c8), v1 = *reinterpret_cast<const float4*>(img + row * P + c8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float h0 = __uint_as_float(hv[t][q] << 16), h1 = __uint_as_float(hv[t][q] & 0xffff0000u);
                    v[2 * q] = (h0 * bsc[2 * q] + bsh[2 * q] > 0.f ? v[2 * q] * bsc[2 * q] : 0.f) + ba[2 * q] + bb[2 * q] * h0;
                    v[2 * q + 1] = (h1 * bsc[2 * q + 1] + bsh[2 * q + 1] > 0.f ? v[2 * q + 1] * bsc[2 * q + 1] : 0.f) + ba[2 * q + 1] + bb[2 * q + 1] * h1;
                }
            }
            u32x4_t o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = pack_bf2(v[2 * q], v[2 * q + 1]);
            *reinterpret_cast<u32x4_t*>(g.Y + (gi * 32 + row) * N + c8) = o;
        }
        wave_lds_fence();       // the image is read before the next group overwrites it
#pragma unroll
        for (int s = 0; s < KS; ++s) xa[s] = xn[s];
    }
    if constexpr (MODE == 0) {
        if (g.stats) {
            // lanes l and l + 32 hold the two row halves of one channel; then the four waves in wave order (fixed order: bit-reproducible)
            __syncthreads();
            float* red = lds;          // [4][2 N]
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const float a1 = s1[nb] + __shfl_xor(s1[nb], 32, 64), a2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
                if (hi == 0) { red[w * 2 * N + nb * 32 + l32] = a1; red[w * 2 * N + N + nb * 32 + l32] = a2; }
            }
            __syncthreads();
            for (int i = tid; i < 2 * N; i += 256)
                g.stats[(int64_t)blockIdx.x * 2 * N + i] = ((red[i] + red[2 * N + i]) + red[4 * N + i]) + red[6 * N + i];
        }
    }
}

template <int K, int N, int MODE>
int rows_launch(RGArgs& g, const char* name, float* colsum, float* colsumsq, hipStream_t s) {
    constexpr int P = N + 4;
    const size_t lds_bytes = (size_t)(4 * 32 * P + (MODE == 0 ? 2 * K : 0)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rows_gemm_kernel<K, N, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    int64_t blocks = (g.groups + 3) / 4;
    if (blocks > 512) blocks = 512;                      // 2 workgroups / CU: every wave walks ~36 groups at the bench size
    float* scratch = nullptr;
    if (MODE == 0 && colsum) {
        const int nch = (int)((blocks + 127) / 128);
        scratch = p3_reduce_scratch(blocks * 2 * N + (int64_t)nch * 2 * N);
        if (!scratch) return P3_ROWS_SKIP;                         // no scratch registered: the caller's tiled path (atomics / persistent sums)
        g.stats = scratch;
    }
    if (p3_tracing()) p3_note_kernel(name);
    hipLaunchKernelGGL((rows_gemm_kernel<K, N, MODE>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (scratch) return p3_det_reduce2(scratch, (int)blocks, 2 * (int64_t)N, scratch + blocks * 2 * N, colsum, colsumsq, N, 2 * N, 1, s);
    return P3_OK;
}

}  // namespace

// p3_gemm's hook: returns P3_ROWS_SKIP when the problem is not one of the two shapes (the caller goes on with its tiled kernels), else the launch status.
int p3_rows_gemm_try(const void* A, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s) {
    if (d->dtype_in != P3_BF16 || d->dtype_out != P3_BF16 || d->M % 32 != 0 || d->M < 4096) return P3_ROWS_SKIP;
    if (d->lda != d->K || d->ldb != d->K || d->ldc != d->N) return P3_ROWS_SKIP;
    if (d->act != P3_ACT_NONE || d->residual || d->aux || (d->drop.seed && d->drop.p > 0.f)) return P3_ROWS_SKIP;
    if (((uintptr_t)A | (uintptr_t)W | (uintptr_t)C) % 16 != 0) return P3_ROWS_SKIP;
    RGArgs g;
    memset(&g, 0, sizeof(g));
    g.X = (const bf16_t*)A; g.W = (const bf16_t*)W; g.Y = (bf16_t*)C; g.groups = d->M / 32;
    if (d->a_mode == P3_A_AFFINE_RELU && d->K == 128 && d->N == 64 && !d->bwd_saved) {
        g.bias = d->bias; g.a_scale = d->a_scale; g.a_shift = d->a_shift;
        return rows_launch<128, 64, 0>(g, "rows_gemm_kernel<128, 64, 0>", d->colsum, d->colsumsq, s);
    }
    if (d->a_mode == P3_A_PLAIN && d->K == 64 && d->N == 128 && d->bwd_saved && d->bwd_act == P3_ACT_BN_RELU && !d->bias && !d->colsum &&
        (uintptr_t)d->bwd_saved % 16 == 0) {
        g.H = (const bf16_t*)d->bwd_saved; g.bn = d->bwd_bn;
        return rows_launch<64, 128, 1>(g, "rows_gemm_kernel<64, 128, 1>", nullptr, nullptr, s);
    }
    return P3_ROWS_SKIP;
}
