// p3hip ScoreNet backward, fp32x3: conv2's weight gradient over the pair grid, fp32 storage, products as bf16 x 3.
//
//   dW2[n, c] += sum over pair rows (b, i, j) of dH2[(b, i, j), n] * relu(bn1(U[b, i, c] + V[b, j, c]))        (128 x 256, 2.36 M rows at the bench size)
//
// Reference: autograd of ScoreNet.conv2 (models/pix2poly/model_pix2poly.py:88-93); csrc/pair_dw_mma.hip is the bf16 kernel of this launch.  Until r05 the fp32x3
// mode ran it on gemm_tn.hip's generated-operand tile kernel (739 us per net).  Here, in the geometry of pair_dw_mma.hip (a workgroup walks (tile b, 8 rows i)
// units, a step = 16 columns j = 128 pair rows ordered (j, i), the 128 x 256 accumulator lives across all units of the workgroup):
//   * the fp32 dH2 tile of step s + 1 is loaded into 32 registers per thread during step s and split ONCE per element into hi / lo bf16 images in LDS (the
//     staging of pair_bwd_x3.hip: 256-byte rows, chunk slot ^ (row & 15)); dH2^T fragments come from the images by transposing reads (ds_read_b64_tr_b16);
//   * the generated operand is built in fp32 IN REGISTERS in the MFMA layout (the 8 rows a lane feeds per 16-row block are the 8 rows i of one column j:
//     relu(fma(V[j, c], scale[c], us[i][c])), us = U_i scale + shift in 8 registers), then split: three MFMAs per (32-channel block, 16 rows);
//   * wave w owns columns c = 32 w .. + 31 for ALL 128 channels n (4 blocks): the generated fragment of a column is built by exactly one wave.
// Per step and wave: 96 MFMA 32x32x16 (3072 cycles; two waves per SIMD) against ~500 VALU operations (generation + split) and 128 transposing reads.
// Shapes: N % 16 == 0 (no ragged group of rows / step of columns); everything else stays on gemm_tn.hip.
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int DX_IB = 8, DX_JT = 16;
constexpr int DX_PLANE = 128 * 256;
constexpr int DX_LDS = 4 * DX_PLANE;                          // (hi, lo) x two buffers

struct DxArgs {
    const float* dH; const float* U; const float* V;
    const float* sc; const float* sh;
    float* C; int ldc;
    float* slabs;           // [gridDim.x][128][256] or NULL (atomics)
    int B, N, nblk, units;
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dx_split8(const float (&v)[8], u32x4_t& h, u32x4_t& l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
        h[k] = hw;
        l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
    }
}

__global__ __launch_bounds__(512, 2) void pair_dw_x3_kernel(DxArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int c = wave * 32 + l31;                                // this lane's column of dW2
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const int nsteps = N / DX_JT;
    const int my_units = (g.units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * nsteps;                       // steps of this workgroup, over all its units
    float4 pre[4][2];
    float v[8], vn[8];
    // global step gs -> (unit, step): dH2 tile rows (jj, ii) -> tile row jj * 8 + ii; unit x = q * 512 + tid covers row x >> 4, k = (x & 15) * 8 .. + 8
    auto fetch = [&](int gs, float (&vv)[8]) __attribute__((always_inline)) {
        const int un = (int)blockIdx.x + (gs / nsteps) * (int)gridDim.x, st = gs % nsteps;
        const int b = un / g.nblk, i0 = (un % g.nblk) * DX_IB, j0 = st * DX_JT;
        const float* dHb = g.dH + ((int64_t)b * N + i0) * (int64_t)N * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = q * 32 + (tid >> 4), ck = tid & 15;
            const float* src = dHb + ((int64_t)(r & 7) * N + j0 + (r >> 3)) * 128 + ck * 8;
            pre[q][0] = *reinterpret_cast<const float4*>(src);
            pre[q][1] = *reinterpret_cast<const float4*>(src + 4);
        }
        const float* Vb = g.V + ((int64_t)b * N + j0 + hi) * 256 + c;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) vv[kk] = Vb[(int64_t)(2 * kk) * 256];       // column j = j0 + 2 kk + hi
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = lds + buf * 2 * DX_PLANE;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = q * 32 + (tid >> 4), ck = tid & 15;
            const float x[8] = {pre[q][0].x, pre[q][0].y, pre[q][0].z, pre[q][0].w, pre[q][1].x, pre[q][1].y, pre[q][1].z, pre[q][1].w};
            u32x4_t h, l;
            dx_split8(x, h, l);
            const uint32_t off = (uint32_t)(r * 256 + ((ck ^ (r & 15)) * 16));
            *reinterpret_cast<u32x4_t*>(base + off) = h;
            *reinterpret_cast<u32x4_t*>(base + DX_PLANE + off) = l;
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ib][r] = 0.f;
    // transposing-read geometry (pair_dw_mma.hip): lane -> (8-row half g4 >> 1, row li >> 2 of a 4-row piece, 16-channel half g4 & 1, 4 channels (li & 3) * 4)
    const int g4 = lane >> 4, li = lane & 15;
    uint32_t troff[4][2];                                      // [channel block ib][rows +0..3 | +4..7] byte offset inside an image, without the 16-row block
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row = (g4 >> 1) * 8 + hh * 4 + (li >> 2);                      // row & 15 of every 16-row block
            const int slot = ib * 4 + (g4 & 1) * 2 + ((li & 3) >> 1);                 // 16-byte slot of channels 32 ib + 16 (g4 & 1) + 4 (li & 3)
            troff[ib][hh] = (uint32_t)(row * 256 + ((slot ^ row) * 16) + ((li & 3) & 1) * 8);
        }
    float s_ = 0.f, us[8];
    if (total > 0) { fetch(0, v); commit(0); }
    for (int gs = 0; gs < total; ++gs) {
        const int st = gs % nsteps;
        __syncthreads();                                        // images of step gs complete; reads of step gs - 1 (whose buffer step gs + 1 takes) are done
        if (gs + 1 < total) fetch(gs + 1, vn);
        if (st == 0) {                                          // a new unit: (U_i scale + shift) of its 8 rows i at this lane's column c
            const int un = (int)blockIdx.x + (gs / nsteps) * (int)gridDim.x;
            const int b = un / g.nblk, i0 = (un % g.nblk) * DX_IB;
            s_ = g.sc[c];
            const float hh = g.sh[c];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) us[ii] = fmaf(g.U[((int64_t)b * N + i0 + ii) * 256 + c], s_, hh);
        }
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t ab = lds_addr + (uint32_t)((gs & 1) * 2 * DX_PLANE);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            u32x2_t fh[4][2], fl[4][2];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fh[ib][hh]) : "v"(ab + (uint32_t)(kk * 16 * 256) + troff[ib][hh]));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fl[ib][hh]) : "v"(ab + (uint32_t)(DX_PLANE + kk * 16 * 256) + troff[ib][hh]));
                }
            // generated operand: rows (j = 2 kk + hi, i = 0..7) of column c, fp32 -> hi / lo
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(fmaf(v[kk], s_, us[e]), 0.f);
            u32x4_t bh_, bl_;
            dx_split8(a, bh_, bl_);
            const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, bh_), bl = __builtin_bit_cast(bf16x8_t, bl_);
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fh[0][0]), "+v"(fh[0][1]), "+v"(fh[1][0]), "+v"(fh[1][1]), "+v"(fh[2][0]), "+v"(fh[2][1]), "+v"(fh[3][0]), "+v"(fh[3][1]),
                           "+v"(fl[0][0]), "+v"(fl[0][1]), "+v"(fl[1][0]), "+v"(fl[1][1]), "+v"(fl[2][0]), "+v"(fl[2][1]), "+v"(fl[3][0]), "+v"(fl[3][1]));
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, u32x4_t{fh[ib][0].x, fh[ib][0].y, fh[ib][1].x, fh[ib][1].y});
                const bf16x8_t al = __builtin_bit_cast(bf16x8_t, u32x4_t{fl[ib][0].x, fl[ib][0].y, fl[ib][1].x, fl[ib][1].y});
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[ib], 0, 0, 0);
                acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[ib], 0, 0, 0);
            }
        }
        if (gs + 1 < total) {
            commit((gs + 1) & 1);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) v[kk] = vn[kk];
        }
    }
    // ---- the workgroup's partial tile
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = ib * 32 + crow32(r, hi);
            if (g.slabs) g.slabs[((int64_t)blockIdx.x * 128 + n) * 256 + c] = acc[ib][r];
            else atomicAdd(g.C + (int64_t)n * g.ldc + c, acc[ib][r]);
        }
}

}  // namespace

void p3_tn_reduce_launch(const float* slabs, float* C, int N, int K, int ldc, int splits, hipStream_t s);      // gemm_tn.hip

// p3_gemm_tn_ex's hook for the pair mode with P3_F32X3 operands: 1 when the shape is not this kernel's (the caller goes on with gemm_tn.hip), else the launch status
int p3_pair_dw_x3_try(const void* A, const void* U, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* scale, const float* shift,
                      const void* pair_V, int pair_n, float* slabs, int max_slabs, hipStream_t s) {
    if (N != 128 || K != 256 || lda != 128 || ldb != 256 || pair_n < DX_JT || pair_n % DX_JT != 0) return 1;
    if ((((uintptr_t)A | (uintptr_t)U | (uintptr_t)pair_V) % 16) != 0) return 1;
    const int n = pair_n;
    const int64_t B = (int64_t)M / ((int64_t)n * n);
    if (B * n * n != M || B < 1) return 1;
    DxArgs g;
    g.dH = (const float*)A; g.U = (const float*)U; g.V = (const float*)pair_V; g.sc = scale; g.sh = shift; g.C = C; g.ldc = ldc;
    g.B = (int)B; g.N = n; g.nblk = n / DX_IB; g.units = (int)B * g.nblk;
    int grid = g.units < 256 ? g.units : 256;                // one workgroup per CU, ~6 units each at the bench size
    g.slabs = (slabs && grid <= max_slabs) ? slabs : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_dw_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DX_LDS);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("pair_dw_x3_kernel");
    hipLaunchKernelGGL(pair_dw_x3_kernel, dim3(grid), dim3(512), DX_LDS, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (g.slabs) p3_tn_reduce_launch(g.slabs, C, 128, 256, ldc, grid, s);
    return P3_OK;
}
