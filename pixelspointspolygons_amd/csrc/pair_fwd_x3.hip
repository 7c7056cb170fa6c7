// p3hip ScoreNet forward, fp32x3: conv2 over the pair grid with conv1's BatchNorm / ReLU generated on the fly, fp32 storage, products as bf16 x 3.
//
// Reference: ScoreNet.forward (models/pix2poly/model_pix2poly.py:86-112), as csrc/pair_fwd_mma.hip (the bf16 form).  Until r05 the fp32x3 mode ran this
// layer on p3_gemm's P3_A_PAIR_AFFINE_RELU tile kernel: 1058 us per net for 1.2 GB of output and 464 GF of issued products (131 TF effective).
//
// A workgroup (8 waves: 2 row groups x 4 channel groups of 32) owns tile b and walks groups of 8 rows i; a step is 16 columns j = 128 pair rows ordered
// (j, i), in four 64-deep stages of K = 256:
//   * a wave's slice of W2 (32 channels x 256 k), split into hi / lo ONCE, lives in 128 registers for the workgroup's life (hi + lo of the whole W2 would be
//     128 KB of LDS);
//   * the generated operand relu(fma(V_j, scale, U_i scale + shift)) is computed in fp32, split, and written as hi / lo bf16 images (2 x 16 KB, 128 rows x 64 k,
//     chunk slot c ^ (r & 7) ^ ((r >> 3) & 1)) for stage g + 1 WHILE stage g is multiplied out of the other pair of images - one split per element;
//   * (U_i scale + shift) of the group's 8 rows: an LDS table built once per group; the V rows of a step (16 KB) are fetched one step ahead through registers
//     into a double-buffered LDS copy; scale: an LDS table;
//   * ONE LDS-only barrier per stage (s_waitcnt lgkmcnt(0); s_barrier), as in the bf16 kernel; the epilogue needs no LDS: a lane holds one channel and 16 rows ->
//     4-byte stores, 32 lanes = 128 contiguous bytes of one output row; BatchNorm-2 column sums are per-lane adds.
// Per stage 24 MFMA 32x32x16 per wave (two waves per SIMD: 1536 cycles) against ~224 KB of LDS traffic (1792 cycles): LDS-bound by design.
#include <stdlib.h>

#include "p3_common.h"

#define P3_PAIR_FWD_SKIP 0x7fffffff

namespace {

constexpr int QX_IB = 8, QX_JT = 16, QX_KS = 64;
constexpr int QX_IMG = 128 * 128;                         // one bf16 image: 128 rows x 64 k
constexpr int QX_A_BYTES = 2 * QX_IMG;                    // hi | lo
constexpr int QX_V_BYTES = QX_JT * 1024;
constexpr int QX_US_STRIDE = 260;                         // floats per row of the (U scale + shift) table: 256 + 4 (bank spread of the 8 rows)
constexpr int QX_OFF_V = 2 * QX_A_BYTES, QX_OFF_US = QX_OFF_V + 2 * QX_V_BYTES, QX_OFF_SC = QX_OFF_US + 8 * QX_US_STRIDE * 4;
constexpr int QX_LDS = QX_OFF_SC + 256 * 4;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct QxFrag { u32x4_t ah[2], al[2]; };

struct QxArgs {
    const float* U; const float* V; const float* W2; float* Y;
    const float* bias; const float* sc; const float* sh;
    float* stats;          // [gridDim.y * gridDim.x][256] (sum | sum of squares) or NULL
    int B, N, ngroups;
};

__device__ __forceinline__ void qx_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void qx_split8(const float (&v)[8], u32x4_t& h, u32x4_t& l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
        h[k] = hw;
        l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
    }
}

__global__ __launch_bounds__(512, 2) void pair_fwd_x3_kernel(QxArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;                  // 2 x 4 waves: 64 rows x 32 channels each
    const int ch = wc * 32 + l31;
    float* ust = reinterpret_cast<float*>(lds + QX_OFF_US);
    float* sct = reinterpret_cast<float*>(lds + QX_OFF_SC);
    // ---- W2 row ch, k = 16 kk + 8 hi .. + 8: B fragments of the 16 k-blocks, hi and lo
    bf16x8_t wh[16], wl[16];
    {
        const float* wrow = g.W2 + (int64_t)ch * 256 + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 x0 = *reinterpret_cast<const float4*>(wrow + kk * 16), x1 = *reinterpret_cast<const float4*>(wrow + kk * 16 + 4);
            const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            u32x4_t h, l;
            qx_split8(v, h, l);
            wh[kk] = __builtin_bit_cast(bf16x8_t, h); wl[kk] = __builtin_bit_cast(bf16x8_t, l);
        }
    }
    if (tid < 256) sct[tid] = g.sc[tid];
    // generation geometry: thread = (8-k chunk gc of a stage, row group rg): rows r = rg + 64 q -> i = rg & 7 fixed, jj = (rg >> 3) + 8 q
    const int gc = tid & 7, rg = tid >> 3, gii = rg & 7, gjb = rg >> 3;
    const float* Vb = g.V + (int64_t)b * N * 256;
    const float bias_ = g.bias ? g.bias[ch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    const int nsteps = (N + QX_JT - 1) / QX_JT;
    const uint32_t arow = (uint32_t)((wr * 64 + l31) * 128);
    const int sxa = (l31 & 7) ^ ((l31 >> 3) & 1);            // chunk swizzle of this lane's fragment rows: rows r and r + 8 of a 16-lane pass in different slots

    for (int grp = blockIdx.x; grp < g.ngroups; grp += gridDim.x) {
        const int i0 = grp * QX_IB;
        float4 vpre[2];
        auto fetch_v = [&](int st) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = tid + 512 * q, row = x >> 6, c4 = x & 63;
                vpre[q] = *reinterpret_cast<const float4*>(Vb + (int64_t)min(st * QX_JT + row, N - 1) * 256 + c4 * 4);
            }
        };
        auto put_v = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int x = tid + 512 * q, row = x >> 6, c4 = x & 63;
                *reinterpret_cast<float4*>(lds + QX_OFF_V + buf * QX_V_BYTES + row * 1024 + c4 * 16) = vpre[q];
            }
        };
        auto gen = [&](int s, unsigned char* img, int vbuf, int q) __attribute__((always_inline)) {
            const int k0 = s * QX_KS + gc * 8;
            const float4 c0 = *reinterpret_cast<const float4*>(sct + k0), c1 = *reinterpret_cast<const float4*>(sct + k0 + 4);
            const float4 u0 = *reinterpret_cast<const float4*>(ust + gii * QX_US_STRIDE + k0), u1 = *reinterpret_cast<const float4*>(ust + gii * QX_US_STRIDE + k0 + 4);
            const float scv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            const float usv[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            {
                const int r = rg + 64 * q, jj = gjb + 8 * q;
                const float* vp = reinterpret_cast<const float*>(lds + QX_OFF_V + vbuf * QX_V_BYTES + jj * 1024) + k0;
                const float4 v0 = *reinterpret_cast<const float4*>(vp), v1 = *reinterpret_cast<const float4*>(vp + 4);
                const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                float a[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = fmaxf(fmaf(vv[e], scv[e], usv[e]), 0.f);
                u32x4_t h, l;
                qx_split8(a, h, l);
                const uint32_t off = (uint32_t)(r * 128 + ((gc ^ ((r & 7) ^ ((r >> 3) & 1))) * 16));
                *reinterpret_cast<u32x4_t*>(img + off) = h;
                *reinterpret_cast<u32x4_t*>(img + QX_IMG + off) = l;
            }
        };
        f32x16 acc[2];
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ib][r] = 0.f;
        };
        auto rd = [&](const unsigned char* img, int kk, QxFrag& f) __attribute__((always_inline)) {
            const uint32_t co = (uint32_t)(((2 * kk + hi) ^ sxa) * 16);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                f.ah[ib] = *reinterpret_cast<const u32x4_t*>(img + arow + ib * 32 * 128 + co);
                f.al[ib] = *reinterpret_cast<const u32x4_t*>(img + QX_IMG + arow + ib * 32 * 128 + co);
            }
        };
        auto mma1 = [&](int ks, const QxFrag& f) __attribute__((always_inline)) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.al[ib]), wh[ks], acc[ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.ah[ib]), wl[ks], acc[ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.ah[ib]), wh[ks], acc[ib], 0, 0, 0);
        };
        // one 64-deep stage: the products of stage s out of `img`, with the next stage's operand (two row halves) generated into `gimg` in the gaps; fragments
        // double-buffered per 16-deep block (the next block's four reads are in flight during the six MFMAs of this one)
        auto stage = [&](int s, const unsigned char* img, bool dogen, int gs_, unsigned char* gimg, int gvb) __attribute__((always_inline)) {
            QxFrag f0, f1;
            rd(img, 0, f0);
            if (dogen) gen(gs_, gimg, gvb, 0);
            __builtin_amdgcn_sched_barrier(0);
            rd(img, 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mma1(s * 4 + 0, f0);
            __builtin_amdgcn_sched_barrier(0);
            if (dogen) gen(gs_, gimg, gvb, 1);
            __builtin_amdgcn_sched_barrier(0);
            rd(img, 2, f0);
            __builtin_amdgcn_sched_barrier(0);
            mma1(s * 4 + 1, f1);
            __builtin_amdgcn_sched_barrier(0);
            rd(img, 3, f1);
            __builtin_amdgcn_sched_barrier(0);
            mma1(s * 4 + 2, f0);
            __builtin_amdgcn_sched_barrier(0);
            mma1(s * 4 + 3, f1);
            __builtin_amdgcn_sched_barrier(0);
        };
        unsigned char* A0 = lds;
        unsigned char* A1 = lds + QX_A_BYTES;
        // ---- prologue of the group: (U_i scale + shift) table, V rows of step 0, image of stage 0
        qx_barrier();                                         // the previous group's last reads (images, V copies, table) are done
        {
            const int i = tid >> 6, k4 = (tid & 63) * 4;
            const float4 u = *reinterpret_cast<const float4*>(g.U + ((int64_t)b * N + min(i0 + i, N - 1)) * 256 + k4);
            const float4 s4 = *reinterpret_cast<const float4*>(g.sc + k4), h4 = *reinterpret_cast<const float4*>(g.sh + k4);
            *reinterpret_cast<float4*>(ust + i * QX_US_STRIDE + k4) = make_float4(fmaf(u.x, s4.x, h4.x), fmaf(u.y, s4.y, h4.y), fmaf(u.z, s4.z, h4.z), fmaf(u.w, s4.w, h4.w));
        }
        fetch_v(0);
        put_v(0);
        qx_barrier();
        gen(0, A0, 0, 0); gen(0, A0, 0, 1);
        zero_acc();
        for (int st = 0; st < nsteps; ++st) {
            const bool more = st + 1 < nsteps;
            const int vb = st & 1;
            // stage 0: multiply A0, generate stage 1 into A1; the next step's V rows start their way here
            qx_barrier();
            if (more) fetch_v(st + 1);
            stage(0, A0, true, 1, A1, vb);
            // stage 1: the next step's V rows land in the other copy (last read three barriers ago)
            qx_barrier();
            if (more) put_v(vb ^ 1);
            stage(1, A1, true, 2, A0, vb);
            // stage 2
            qx_barrier();
            stage(2, A0, true, 3, A1, vb);
            // stage 3: generate the next step's stage 0, multiply, epilogue
            qx_barrier();
            stage(3, A1, more, 0, A0, vb ^ 1);
            {
                // N % 8 == 0 (host check): every row i of the group exists; a column j beyond N (ragged last step) is a wave-uniform skip
                const int j0 = st * QX_JT;
                float* Yb = g.Y + ((int64_t)b * N + i0) * (int64_t)N * 128;
                const uint32_t off0 = (uint32_t)((4 * hi * N + j0 + wr * 8) * 128 + ch);
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        if (j0 + wr * 8 + ib * 4 + q4 >= N) continue;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float v = acc[ib][q4 * 4 + q] + bias_;
                            s1 += v; s2 = fmaf(v, v, s2);
                            (Yb + (int64_t)q * N * 128)[off0 + (uint32_t)((ib * 4 + q4) * 128)] = v;
                        }
                    }
            }
            zero_acc();
        }
    }
    // ---- BatchNorm-2 column sums of this workgroup: half-waves, then the two wave rows, in a fixed order
    if (g.stats) {
        qx_barrier();
        float* red = reinterpret_cast<float*>(lds);               // [2][256]
        const float a1 = s1 + __shfl_xor(s1, 32, 64), a2 = s2 + __shfl_xor(s2, 32, 64);
        if (hi == 0) { red[wr * 256 + ch] = a1; red[wr * 256 + 128 + ch] = a2; }
        qx_barrier();
        if (tid < 256) g.stats[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = red[tid] + red[256 + tid];
    }
}

}  // namespace

// p3_gemm's hook for P3_A_PAIR_AFFINE_RELU with P3_F32X3 operands: P3_PAIR_FWD_SKIP when the problem is not the ScoreNet conv2 shape
int p3_pair_fwd_x3_try(const void* U, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s) {
    if (d->a_mode != P3_A_PAIR_AFFINE_RELU || d->dtype_in != P3_F32 || d->dtype_out != P3_F32 || d->K != 256 || d->N != 128) return P3_PAIR_FWD_SKIP;
    if (d->lda != 256 || d->ldb != 256 || d->ldc != 128 || d->pair_n < 8 || d->pair_n % QX_IB != 0) return P3_PAIR_FWD_SKIP;
    if (d->act != P3_ACT_NONE || d->residual || d->aux || d->bwd_saved || (d->drop.seed && d->drop.p > 0.f)) return P3_PAIR_FWD_SKIP;
    if ((((uintptr_t)U | (uintptr_t)W | (uintptr_t)C | (uintptr_t)d->pair_V | (uintptr_t)d->a_scale | (uintptr_t)d->a_shift) % 16) != 0) return P3_PAIR_FWD_SKIP;
    const int N = d->pair_n, B = (int)((int64_t)d->M / ((int64_t)N * N));
    if ((int64_t)B * N * N != d->M || B > 65535 || (int64_t)QX_IB * N * 128 + (int64_t)N * 128 >= (1ll << 31)) return P3_PAIR_FWD_SKIP;
    QxArgs g;
    g.U = (const float*)U; g.V = (const float*)d->pair_V; g.W2 = (const float*)W; g.Y = (float*)C;
    g.bias = d->bias; g.sc = d->a_scale; g.sh = d->a_shift; g.stats = nullptr;
    g.B = B; g.N = N; g.ngroups = (N + QX_IB - 1) / QX_IB;
    // walkers per tile: 512 workgroups = two resident rounds at the bench size (64 tiles x 8 walkers x 3 groups of 8 rows)
    int gx = (512 + B - 1) / B;
    if (gx > g.ngroups) gx = g.ngroups;
    if (gx < 1) gx = 1;
    const int64_t nblocks = (int64_t)gx * B;
    float* scratch = nullptr;
    if (d->colsum) {
        const int nch = (int)((nblocks + 127) / 128);
        scratch = p3_reduce_scratch(nblocks * 256 + (int64_t)nch * 256);
        if (!scratch) return P3_PAIR_FWD_SKIP;
        g.stats = scratch;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_fwd_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QX_LDS);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("pair_fwd_x3_kernel");
    hipLaunchKernelGGL(pair_fwd_x3_kernel, dim3(gx, B), dim3(512), QX_LDS, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (scratch) return p3_det_reduce2(scratch, (int)nblocks, 256, scratch + nblocks * 256, d->colsum, d->colsumsq, 128, 256, 1, s);
    return P3_OK;
}
