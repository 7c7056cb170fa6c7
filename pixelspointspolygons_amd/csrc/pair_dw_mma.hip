// p3hip ScoreNet backward: conv2's weight gradient over the pair grid - a dedicated kernel.
//
//   dW2[n, c] += sum over pair rows (b, i, j) of dH2[(b, i, j), n] * relu(bn1(U[b, i, c] + V[b, j, c]))        (128 x 256, 2.36 M rows at the bench size)
//
// Reference: autograd of ScoreNet.conv2 (models/pix2poly/model_pix2poly.py:88-93) - the [B, 256, N, N] activation it multiplies with is never stored here.
// r01 - r03: p3_gemm_tn's P3_A_PAIR_AFFINE_RELU mode (gemm_tn.hip): the generated operand goes registers -> ds_write -> transposing ds_read per 64-row step
// with three staging sets (359 us per launch = 432 TF, r04 profile).  Here, in the geometry of pair_bwd_mma.hip (a workgroup walks (tile b, 8 rows i) units,
// a step = 16 columns j = 128 pair rows ordered (j, i), the dH2 tile and the 16 V rows double-buffered by LDS-DMA):
//   * the reduction index of the MFMA is the pair row.  In the (j, i) order the 8 consecutive rows a lane feeds per 16-row block are the 8 rows i of ONE column
//     j: the generated operand's fragment is relu(fma(V[j, c], scale[c], us[i][c])) for i = 0 .. 7 with us = U_i scale + shift held in 16 registers - it is
//     built IN REGISTERS in the MFMA layout (8 fma + 8 max + 4 packs per fragment, one 2-byte LDS read for V): no LDS image, no staging set, no spill;
//   * dH2^T fragments come from the row-major tile by transposing reads (ds_read_b64_tr_b16: 16 lanes read 4 rows x 32 bytes and receive 4 rows of their own
//     column) at the tile's 16-byte-slot swizzle, applied on the DMA source address like in pair_bwd_mma.hip - but slot ^ ((row & 3) << 2), not that kernel's
//     slot ^ (row & 15): the 16 lanes of a transposing read take 32 bytes of each of 4 consecutive rows, and the row-fragment swizzle puts rows r, r + 1 into the same
//     32-byte bank range (r05, SQ_LDS_BANK_CONFLICT of the fp32x3 twin pair_dw_x3.hip: 23 % of the wave cycles with the old swizzle, 0 with this one);
//   * the 128 x 256 accumulator (64 registers per lane, 8 waves as 2 x 4) lives across ALL units of the workgroup (one workgroup per CU): 256 partial tiles
//     per launch -> fp32 atomics (8.4 M) or, with slabs, stores + the float64 reduce of the weight-gradient path.
// Shapes: bf16, N % 16 == 0 (no ragged group of rows / step of columns); everything else stays on gemm_tn.hip.
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int PD_IB = 8, PD_JT = 16;
constexpr int PD_A_BYTES = 128 * 256, PD_V_BYTES = PD_JT * 512;
constexpr int PD_LDS = 2 * PD_A_BYTES + 2 * PD_V_BYTES;

struct PdArgs {
    const bf16_t* dH; const bf16_t* U; const bf16_t* V;
    const float* sc; const float* sh;
    float* C; int ldc;
    float* slabs;           // [gridDim.x][128][256] or NULL (atomics)
    int B, N, nblk, units;
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 1) void pair_dw_mma_kernel(PdArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;                  // output block: channels n of dH2 [64 wr, +64) x columns c [64 wc, +64)
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto dma1 = [&](const void* base, uint32_t dst, uint32_t voff) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };
    const int nsteps = N / PD_JT;
    const int my_units = (g.units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * nsteps;                       // steps of this workgroup, over all its units
    // staging of global step gs: dH2 tile rows (jj, ii) -> tile row jj * 8 + ii (4 pieces per wave), V rows of the 16 j (2 rows per wave)
    auto stage = [&](int gs) __attribute__((always_inline)) {
        const int un = (int)blockIdx.x + (gs / nsteps) * (int)gridDim.x, st = gs % nsteps;
        const int b = un / g.nblk, i0 = (un % g.nblk) * PD_IB, j0 = st * PD_JT, buf = gs & 1;
        const bf16_t* dHb = g.dH + ((int64_t)b * N + i0) * (int64_t)N * 128;
        const bf16_t* Vb = g.V + (int64_t)b * N * 256;
        const uint32_t da = lds_addr + (uint32_t)(buf * PD_A_BYTES), dv = lds_addr + (uint32_t)(2 * PD_A_BYTES + buf * PD_V_BYTES);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave * 4 + q, r = p * 4 + (lane >> 4), slot = lane & 15;      // tile row r = jj * 8 + ii
            dma1(dHb, da + (uint32_t)(p * 1024), (uint32_t)((((int64_t)(r & 7) * N + j0 + (r >> 3)) * 128 + ((slot ^ ((r & 3) << 2)) * 8)) * 2));
        }
        dma1(Vb, dv + (uint32_t)(wave * 1024), (uint32_t)(((j0 + wave * 2 + (lane >> 5)) * 256 + (lane & 31) * 8) * 2));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ib][jb][r] = 0.f;
    // transposing-read geometry: lane -> (8-row half g4 >> 1, row li >> 2 of a 4-row piece, 16-channel half g4 & 1, 4 channels (li & 3) * 4)
    const int g4 = lane >> 4, li = lane & 15;
    uint32_t troff[2][2];                                      // [channel block ib][rows +0..3 | +4..7] byte offset inside a tile, without the 16-row block
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int row = (g4 >> 1) * 8 + hh * 4 + (li >> 2);                      // row & 15 of every 16-row block
            const int slot = (wr * 2 + ib) * 4 + (g4 & 1) * 2 + ((li & 3) >> 1);       // 16-byte slot of channels (64 wr + 32 ib) + 16 (g4 & 1) + 4 (li & 3)
            troff[ib][hh] = (uint32_t)(row * 256 + ((slot ^ ((row & 3) << 2)) * 16) + ((li & 3) & 1) * 8);
        }
    float s_[2], us[2][8];
    if (total > 0) stage(0);
    for (int gs = 0; gs < total; ++gs) {
        const int st = gs % nsteps;
        if (st == 0) {                                          // a new unit: (U_i scale + shift) of its 8 rows i at this lane's two columns c
            const int un = (int)blockIdx.x + (gs / nsteps) * (int)gridDim.x;
            const int b = un / g.nblk, i0 = (un % g.nblk) * PD_IB;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const int c = wc * 64 + jb * 32 + l31;
                s_[jb] = g.sc[c];
                const float hh = g.sh[c];
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) us[jb][ii] = fmaf(bf2f(g.U[((int64_t)b * N + i0 + ii) * 256 + c]), s_[jb], hh);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of step gs (the U loads above included)
        __builtin_amdgcn_s_barrier();                           // every wave's pieces; all reads of step gs - 1 (whose buffers step gs + 1 takes) are done
        if (gs + 1 < total) stage(gs + 1);
        const uint32_t ab = lds_addr + (uint32_t)((gs & 1) * PD_A_BYTES);
        const unsigned char* Vt = lds + 2 * PD_A_BYTES + (gs & 1) * PD_V_BYTES;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            u32x2_t fa[2][2];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[ib][hh]) : "v"(ab + (uint32_t)(kk * 16 * 256) + troff[ib][hh]));
            // generated operand: rows (j = 2 kk + hi, i = 0..7) of columns c
            bf16x8_t bfr[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                const float v = bf2f(*reinterpret_cast<const bf16_t*>(Vt + (2 * kk + hi) * 512 + (wc * 64 + jb * 32 + l31) * 2));
                u32x4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf2(fmaxf(fmaf(v, s_[jb], us[jb][2 * e]), 0.f), fmaxf(fmaf(v, s_[jb], us[jb][2 * e + 1]), 0.f));
                bfr[jb] = __builtin_bit_cast(bf16x8_t, o);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]));
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, u32x4_t{fa[ib][0].x, fa[ib][0].y, fa[ib][1].x, fa[ib][1].y});
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[jb], acc[ib][jb], 0, 0, 0);
            }
        }
    }
    // ---- the workgroup's partial tile
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int c = wc * 64 + jb * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = wr * 64 + ib * 32 + crow32(r, hi);
                if (g.slabs) g.slabs[((int64_t)blockIdx.x * 128 + n) * 256 + c] = acc[ib][jb][r];
                else atomicAdd(g.C + (int64_t)n * g.ldc + c, acc[ib][jb][r]);
            }
        }
}

}  // namespace

void p3_tn_reduce_launch(const float* slabs, float* C, int N, int K, int ldc, int splits, hipStream_t s);      // gemm_tn.hip

// p3_gemm_tn_ex's hook for the pair mode: 1 when the shape is not this kernel's (the caller goes on with gemm_tn.hip), else the launch status
int p3_pair_dw_try(const void* A, const void* U, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* scale, const float* shift,
                   const void* pair_V, int pair_n, float* slabs, int max_slabs, hipStream_t s) {
    if (N != 128 || K != 256 || lda != 128 || ldb != 256 || pair_n < PD_JT || pair_n % PD_JT != 0) return 1;
    if ((((uintptr_t)A | (uintptr_t)U | (uintptr_t)pair_V) % 16) != 0) return 1;
    const int n = pair_n;
    const int64_t B = (int64_t)M / ((int64_t)n * n);
    if (B * n * n != M || B < 1) return 1;
    if ((int64_t)PD_IB * n * 256 >= (1ll << 31) || (int64_t)n * 512 >= (1ll << 31)) return 1;       // 32-bit DMA offsets inside a unit
    PdArgs g;
    g.dH = (const bf16_t*)A; g.U = (const bf16_t*)U; g.V = (const bf16_t*)pair_V; g.sc = scale; g.sh = shift; g.C = C; g.ldc = ldc;
    g.B = (int)B; g.N = n; g.nblk = n / PD_IB; g.units = (int)B * g.nblk;
    int grid = g.units < 256 ? g.units : 256;                // one workgroup per CU, ~6 units each at the bench size
    g.slabs = (slabs && grid <= max_slabs) ? slabs : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_dw_mma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PD_LDS);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("pair_dw_mma_kernel");
    hipLaunchKernelGGL(pair_dw_mma_kernel, dim3(grid), dim3(512), PD_LDS, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (g.slabs) p3_tn_reduce_launch(g.slabs, C, 128, 256, ldc, grid, s);
    return P3_OK;
}
