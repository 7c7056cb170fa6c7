// p3hip ScoreNet forward: conv2 over the pair grid with conv1's BatchNorm / ReLU generated on the fly - a dedicated kernel.
//
// Reference: ScoreNet.forward (models/pix2poly/model_pix2poly.py:86-112): conv1 (1 x 1 over cat(X_i, X_j), separable: U_i + V_j) -> bn1 -> relu -> conv2
// (256 -> 128).  r01 - r03 ran it as p3_gemm's P3_A_PAIR_AFFINE_RELU mode: the general tile kernel with three staging register sets for the generated
// operand needs more than its 168 registers (42 spilled; 210 at two workgroups / CU: no spill, same time) - r04 PMC: 1384 MB written for a 604 MB
// output and 349 MB fetched for 6 MB of operands (scratch traffic), 415 us per launch.
//
// Here a workgroup (8 waves) owns tile b and walks groups of 8 rows i; a step is 32 columns j = 256 pair rows ordered (j, i), in four 64-deep stages
// of K = 256:
//   * W2 (128 x 256 bf16, 64 KB) stays in LDS for the workgroup's life (chunk c of row r at slot c ^ (r & 15): conflict-free ds_read_b128 fragments);
//   * the generated operand relu(fma(V_j, scale, U_i scale + shift)) of stage g + 1 is written into one of two 32 KB LDS images (256 rows x 64 k, chunk
//     slot c ^ (r & 7)) WHILE stage g is multiplied out of the other: a thread keeps (U_i scale + shift) of its row i in registers (32 values: its 8-k chunk of
//     each stage), reads scale from an LDS table and the V chunk from a register set loaded two or more stages earlier (12 + 4 loads per step, L2-resident);
//   * ONE barrier per stage, and it waits for LDS only (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads would also drain the V prefetch and the output
//     stores.  The epilogue needs no LDS: in the 32 x 32 accumulator layout a lane holds one channel and 16 rows -> 2-byte stores, 32 lanes = 64
//     contiguous bytes of one output row, and the BatchNorm-2 column sums are plain per-lane adds;
//   * vmcnt counts loads and stores and retires them out of order against each other: a wait for V after the epilogue's 64 stores is a wait for the
//     stores.  The V loads of the NEXT step's stages 0..2 are therefore issued (and waited for) before the epilogue; the first wait after it is 2.5
//     stages later.
// MFMA 256 x 128 x 64 per stage (1024 cycles / SIMD) against 128 KB of fragment reads + 32 KB of image writes (1280 LDS cycles) and 112 VALU operations
// per thread: the launch is LDS-bound by design, not spill- or barrier-bound.
#include <stdlib.h>

#include <type_traits>

#include "p3_common.h"

#define P3_PAIR_FWD_SKIP 0x7fffffff

namespace {

constexpr int QF_IB = 8, QF_JT = 32, QF_KS = 64;
constexpr int QF_W_BYTES = 128 * 512, QF_A_BYTES = 256 * 128;
constexpr int QF_LDS = QF_W_BYTES + 2 * QF_A_BYTES + 256 * 4;       // + the scale table

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

struct QfArgs {
    const bf16_t* U; const bf16_t* V; const bf16_t* W2; bf16_t* Y;
    const float* bias; const float* sc; const float* sh;
    float* stats;          // [gridDim.y * gridDim.x][256] (sum | sum of squares) or NULL
    int B, N, ngroups;
};

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 1) void pair_fwd_mma_kernel(QfArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;                  // 4 x 2 waves: 64 rows x 64 channels each
    float* sct = reinterpret_cast<float*>(lds + QF_W_BYTES + 2 * QF_A_BYTES);
    // ---- W2 -> LDS once (8 x 16 bytes per thread), scale table
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int x = tid + 512 * q, row = x >> 5, c = x & 31;
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(g.W2 + row * 256 + c * 8);
        *reinterpret_cast<u32x4_t*>(lds + row * 512 + ((c ^ (row & 15)) * 16)) = w;
    }
    if (tid < 256) sct[tid] = g.sc[tid];
    // generation geometry: thread = (8-k chunk c of a stage, row group): rows r = rg + 64 q  ->  i = r & 7 fixed, j = (rg >> 3) + 8 q
    const int gc = tid & 7, rg = tid >> 3, gii = rg & 7, gjb = rg >> 3;
    const bf16_t* Vb = g.V + (int64_t)b * N * 256;
    float bias_[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) bias_[cb] = g.bias ? g.bias[wc * 64 + cb * 32 + l31] : 0.f;
    const int nsteps = (N + QF_JT - 1) / QF_JT;
    const uint32_t arow = (uint32_t)((wr * 64 + l31) * 128), brow = (uint32_t)((wc * 64 + l31) * 512);
    const int sxa = l31 & 7, sxb = l31 & 15;

    for (int grp = blockIdx.x; grp < g.ngroups; grp += gridDim.x) {
        const int i0 = grp * QF_IB;
        // (U_i scale + shift) of this thread's row i: its chunk of every stage
        float us[4][8];
        {
            const int i = min(i0 + gii, N - 1);
            const bf16_t* up = g.U + ((int64_t)b * N + i) * 256 + gc * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4_t uu = *reinterpret_cast<const u32x4_t*>(up + s * QF_KS);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = s * QF_KS + gc * 8 + 2 * e;
                    us[s][2 * e] = fmaf(__uint_as_float(uu[e] << 16), g.sc[k], g.sh[k]);
                    us[s][2 * e + 1] = fmaf(__uint_as_float(uu[e] & 0xffff0000u), g.sc[k + 1], g.sh[k + 1]);
                }
            }
        }
        u32x4_t vq[4][4];                                     // V chunks: [stage slot][row q]
        auto load_v = [&](int st, int s) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = min(st * QF_JT + gjb + 8 * q, N - 1);
                vq[s][q] = *reinterpret_cast<const u32x4_t*>(Vb + (int64_t)j * 256 + s * QF_KS + gc * 8);
            }
        };
        auto gen = [&](int s, unsigned char* img) __attribute__((always_inline)) {
            const float4 c0 = *reinterpret_cast<const float4*>(sct + s * QF_KS + gc * 8), c1 = *reinterpret_cast<const float4*>(sct + s * QF_KS + gc * 8 + 4);
            const float scv[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rg + 64 * q;
                u32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = fmaxf(fmaf(__uint_as_float(vq[s][q][e] << 16), scv[2 * e], us[s][2 * e]), 0.f);
                    const float hi_ = fmaxf(fmaf(__uint_as_float(vq[s][q][e] & 0xffff0000u), scv[2 * e + 1], us[s][2 * e + 1]), 0.f);
                    o[e] = pack_bf2(lo, hi_);
                }
                *reinterpret_cast<u32x4_t*>(img + r * 128 + ((gc ^ (r & 7)) * 16)) = o;
            }
        };
        f32x16 acc[2][2];
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ib][cb][r] = 0.f;
        };
        auto mma = [&](int s, const unsigned char* img) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                u32x4_t af[2], bf[2];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) af[ib] = *reinterpret_cast<const u32x4_t*>(img + arow + ib * 32 * 128 + (((2 * kk + hi) ^ sxa) * 16));
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) bf[cb] = *reinterpret_cast<const u32x4_t*>(lds + brow + cb * 32 * 512 + (((s * 8 + 2 * kk + hi) ^ sxb) * 16));
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
                        acc[ib][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[ib]), __builtin_bit_cast(bf16x8_t, bf[cb]), acc[ib][cb], 0, 0, 0);
            }
        };
        unsigned char* A0 = lds + QF_W_BYTES;
        unsigned char* A1 = A0 + QF_A_BYTES;
        // ---- prologue of the group: V of step 0 (all four stages), image of stage 0
        load_v(0, 0); load_v(0, 1); load_v(0, 2); load_v(0, 3);
        lds_only_barrier();                                   // W2 / table stored (first group); the previous group's last reads of A0 are done
        gen(0, A0);
        zero_acc();
        for (int st = 0; st < nsteps; ++st) {
            const bool more = st + 1 < nsteps;
            // stage 0: multiply A0, generate stage 1 into A1; V of stage 3 was loaded one step (or the prologue) ago
            lds_only_barrier();
            gen(1, A1);
            mma(0, A0);
            // stage 1
            lds_only_barrier();
            gen(2, A0);
            mma(1, A1);
            // stage 2: generate stage 3, then fetch the next step's stages 0..2 (their slots are free now)
            lds_only_barrier();
            gen(3, A1);
            if (more) { load_v(st + 1, 0); load_v(st + 1, 1); load_v(st + 1, 2); }
            mma(2, A0);
            // stage 3: generate the next step's stage 0 (waits for the loads above - no store is pending yet), multiply, epilogue, THEN the stage-3 V loads
            lds_only_barrier();
            if (more) gen(0, A0);
            mma(3, A1);
            {
                // 2-byte stores: a uniform base per row i + ONE per-lane element offset + immediates.  N % 8 == 0 (host check): every row i of the
                // group exists; a column j beyond N (ragged last step) is a wave-uniform skip
                const int j0 = st * QF_JT;
                bf16_t* Yb = g.Y + ((int64_t)b * N + i0) * (int64_t)N * 128;
                const uint32_t off0 = (uint32_t)((4 * hi * N + j0 + wr * 8) * 128 + wc * 64 + l31);
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        if (j0 + wr * 8 + ib * 4 + q4 >= N) continue;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int cb = 0; cb < 2; ++cb) {
                                const float v = acc[ib][cb][q4 * 4 + q] + bias_[cb];
                                s1[cb] += v; s2[cb] = fmaf(v, v, s2[cb]);
                                (Yb + (int64_t)q * N * 128)[off0 + (uint32_t)((ib * 4 + q4) * 128 + cb * 32)] = f2bf(v);
                            }
                    }
            }
            zero_acc();
            if (more) load_v(st + 1, 3);
        }
    }
    // ---- BatchNorm-2 column sums of this workgroup: half-waves, then the four wave rows, in a fixed order
    if (g.stats) {
        lds_only_barrier();
        float* red = reinterpret_cast<float*>(lds + QF_W_BYTES);          // [4][256]
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const float a1 = s1[cb] + __shfl_xor(s1[cb], 32, 64), a2 = s2[cb] + __shfl_xor(s2[cb], 32, 64);
            if (hi == 0) { red[wr * 256 + wc * 64 + cb * 32 + l31] = a1; red[wr * 256 + 128 + wc * 64 + cb * 32 + l31] = a2; }
        }
        lds_only_barrier();
        if (tid < 256) g.stats[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 256 + tid] = ((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid];
    }
}

}  // namespace

// p3_gemm's hook for P3_A_PAIR_AFFINE_RELU: P3_PAIR_FWD_SKIP when the problem is not the ScoreNet conv2 shape (the caller goes on with its tile kernel)
int p3_pair_fwd_try(const void* U, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s) {
    if (d->a_mode != P3_A_PAIR_AFFINE_RELU || d->dtype_in != P3_BF16 || d->dtype_out != P3_BF16 || d->K != 256 || d->N != 128) return P3_PAIR_FWD_SKIP;
    if (d->lda != 256 || d->ldb != 256 || d->ldc != 128 || d->pair_n < 8 || d->pair_n % QF_IB != 0) return P3_PAIR_FWD_SKIP;
    if (d->act != P3_ACT_NONE || d->residual || d->aux || d->bwd_saved || (d->drop.seed && d->drop.p > 0.f)) return P3_PAIR_FWD_SKIP;
    if ((((uintptr_t)U | (uintptr_t)W | (uintptr_t)C | (uintptr_t)d->pair_V) % 16) != 0) return P3_PAIR_FWD_SKIP;
    const int N = d->pair_n, B = (int)((int64_t)d->M / ((int64_t)N * N));
    if ((int64_t)B * N * N != d->M || B > 65535) return P3_PAIR_FWD_SKIP;
    QfArgs g;
    g.U = (const bf16_t*)U; g.V = (const bf16_t*)d->pair_V; g.W2 = (const bf16_t*)W; g.Y = (bf16_t*)C;
    g.bias = d->bias; g.sc = d->a_scale; g.sh = d->a_shift; g.stats = nullptr;
    g.B = B; g.N = N; g.ngroups = (N + QF_IB - 1) / QF_IB;
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
        hipError_t e = hipFuncSetAttribute((const void*)pair_fwd_mma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, QF_LDS);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("pair_fwd_mma_kernel");
    hipLaunchKernelGGL(pair_fwd_mma_kernel, dim3(gx, B), dim3(512), QF_LDS, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (scratch) return p3_det_reduce2(scratch, (int)nblocks, 256, scratch + nblocks * 256, d->colsum, d->colsumsq, 128, 256, 1, s);
    return P3_OK;
}
