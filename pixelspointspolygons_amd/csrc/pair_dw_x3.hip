// p3hip ScoreNet backward, fp32x3: conv2's weight gradient over the pair grid, fp32 storage, products as bf16 x 3.
//
//   dW2[n, c] += sum over pair rows (b, i, j) of dH2[(b, i, j), n] * relu(bn1(U[b, i, c] + V[b, j, c]))        (128 x 256, 2.36 M rows at the bench size)
//
// Reference: autograd of ScoreNet.conv2 (models/pix2poly/model_pix2poly.py:88-93); csrc/pair_dw_mma.hip is the bf16 kernel of this launch.  Until r05 the fp32x3
// mode ran it on gemm_tn.hip's generated-operand tile kernel (739 us per net).  Here, in the geometry of pair_dw_mma.hip (a workgroup walks (tile b, 8 rows i)
// units, a step = 16 columns j = 128 pair rows ordered (j, i), the 128 x 256 accumulator lives across all units of the workgroup):
//   * the fp32 dH2 tile of step s + 1 is parked in LDS by LDS-DMA during step s and turned in place into hi / lo bf16 images at its end - the staging of
//     pair_bwd_x3.hip (4-row groups of 1 KB hi | 1 KB lo; every lane converts what its own DMA wrote); dH2^T fragments come from the images by transposing
//     reads (ds_read_b64_tr_b16, through the builtin: the compiler tracks them), double-buffered per 16-row block.  The 16-byte chunk c of row r sits at slot
//     c ^ ((r & 3) << 2): a transposing read's 16 lanes take 32 bytes of each of 4 consecutive rows - with the row-fragment swizzle c ^ (r & 15) of the other
//     kernels rows r, r + 1 land in the same 32-byte bank range (measured: SQ_LDS_BANK_CONFLICT = 23 % of the wave cycles, 68 % of the LDS-active ones); this
//     one spreads the 4 rows x 2 chunk pairs of a 32-lane pass over all 8 ranges;
//   * the generated operand is built in fp32 IN REGISTERS in the MFMA layout (the 8 rows a lane feeds per 16-row block are the 8 rows i of one column j:
//     relu(fma(V[j, c], scale[c], us[i][c])), us = U_i scale + shift in 8 registers), then split: three MFMAs per (32-channel block, 16 rows);
//   * wave w owns columns c = 32 w .. + 31 for ALL 128 channels n (4 blocks): the generated fragment of a column is built by exactly one wave.
// Per step and wave: 96 MFMA 32x32x16 (3072 cycles; two waves per SIMD) against ~500 VALU operations (generation + split) and 128 transposing reads.
// Shapes: N % 16 == 0 (no ragged group of rows / step of columns); everything else stays on gemm_tn.hip.
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int DX_IB = 8, DX_JT = 16;
constexpr int DX_TILE = 128 * 512;                            // one tile: 64 KB as parked fp32, then as hi / lo images
constexpr int DX_LDS = 2 * DX_TILE;

struct DxArgs {
    const float* dH; const float* U; const float* V;
    const float* sc; const float* sh;
    float* C; int ldc;
    float* slabs;           // [gridDim.x][128][256] or NULL (atomics)
    int B, N, nblk, units;
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ void dx_split8(const float (&v)[8], u32x4_t& h, u32x4_t& l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
        h[k] = hw;
        l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
    }
}

struct DxSet { s16x4_t h[4][2], l[4][2]; };                  // dH2^T fragments of one 16-row block: [channel block][rows +0..3 | +4..7], hi and lo

__global__ __launch_bounds__(512, 2) void pair_dw_x3_kernel(DxArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int c = wave * 32 + l31;                                // this lane's column of dW2
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto dma1 = [&](const void* base, uint32_t dst, uint32_t voff) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };
    const int nsteps = N / DX_JT;
    const int my_units = (g.units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_units * nsteps;                       // steps of this workgroup, over all its units
    float v[8], vn[8];
    // global step gs -> (unit, step).  Parking: group p = wave * 4 + q = tile rows 4 p .. 4 p + 3 (tile row r = jj * 8 + ii); lane -> (row 4 p + (lane >> 4), slot
    // lane & 15) holds chunk slot ^ ((r & 3) << 2): floats 0..3 to the group's first 1 KB block, 4..7 to the second
    auto park = [&](int gs, float (&vv)[8]) __attribute__((always_inline)) {
        const int un = (int)blockIdx.x + (gs / nsteps) * (int)gridDim.x, st = gs % nsteps;
        const int b = un / g.nblk, i0 = (un % g.nblk) * DX_IB, j0 = st * DX_JT;
        const float* dHb = g.dH + ((int64_t)b * N + i0) * (int64_t)N * 128;
        const uint32_t dst = lds_addr + (uint32_t)((gs & 1) * DX_TILE);
        // the V loads go FIRST: vmcnt retires in order, a wait for them must not be a wait for the DMA pieces behind them
        const float* Vb = g.V + ((int64_t)b * N + j0 + hi) * 256 + c;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) vv[kk] = Vb[(int64_t)(2 * kk) * 256];       // column j = j0 + 2 kk + hi
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave * 4 + q, r = p * 4 + (lane >> 4), ck = (lane & 15) ^ ((r & 3) << 2);
            const uint32_t voff = (uint32_t)((((int64_t)(r & 7) * N + j0 + (r >> 3)) * 128 + ck * 8) * 4);
            dma1(dHb, dst + (uint32_t)(p * 2048), voff);
            dma1(dHb, dst + (uint32_t)(p * 2048 + 1024), voff + 16u);
        }
    };
    auto convert = [&](int gs) __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): this lane's own pieces and everything older (the builtin, not asm: the compiler's
        asm volatile("" ::: "memory");                                // counter model then knows the V loads are complete too and inserts no waits of its own for them)
        unsigned char* base = lds + (gs & 1) * DX_TILE + wave * 4 * 2048 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x0 = *reinterpret_cast<const float4*>(base + q * 2048), x1 = *reinterpret_cast<const float4*>(base + q * 2048 + 1024);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            u32x4_t h, l;
            dx_split8(x, h, l);
            *reinterpret_cast<u32x4_t*>(base + q * 2048) = h;
            *reinterpret_cast<u32x4_t*>(base + q * 2048 + 1024) = l;
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
            troff[ib][hh] = (uint32_t)((row >> 2) * 2048 + (row & 3) * 256 + ((slot ^ ((row & 3) << 2)) * 16) + ((li & 3) & 1) * 8);
        }
    float s_ = 0.f, us[8];
    if (total > 0) { park(0, v); convert(0); }
    // one step; (vc, vx) = the V values of this step / the registers the next step's are loaded into - the caller alternates them (no copies: a copy would wait
    // for the loads at the top of the step instead of at the top of the next one)
    auto step = [&](int gs, float (&vc)[8], float (&vx)[8]) __attribute__((always_inline)) {
        const int st = gs % nsteps;
        __syncthreads();                                        // images of step gs complete; reads of step gs - 1 (whose buffer step gs + 1 takes) are done
        if (st == 0) {                                          // a new unit: (U_i scale + shift) of its 8 rows i at this lane's column c (loads ahead of the DMA)
            const int un = (int)blockIdx.x + (gs / nsteps) * (int)gridDim.x;
            const int b = un / g.nblk, i0 = (un % g.nblk) * DX_IB;
            s_ = g.sc[c];
            const float hh = g.sh[c];
            float ur[8];
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) ur[ii] = g.U[((int64_t)b * N + i0 + ii) * 256 + c];
            __builtin_amdgcn_sched_barrier(0);
            if (gs + 1 < total) park(gs + 1, vx);
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) us[ii] = fmaf(ur[ii], s_, hh);
        } else if (gs + 1 < total) {
            park(gs + 1, vx);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* ab = lds + (gs & 1) * DX_TILE;
        auto rdset = [&](int kk, DxSet& f) __attribute__((always_inline)) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const unsigned char* p = ab + kk * 8192 + troff[ib][hh];
                    f.h[ib][hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
                    f.l[ib][hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p + 1024));
                }
        };
        auto block = [&](int kk, const DxSet& f) __attribute__((always_inline)) {
            // generated operand: rows (j = 2 kk + hi, i = 0..7) of column c, fp32 -> hi / lo
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(fmaf(vc[kk], s_, us[e]), 0.f);
            u32x4_t bh_, bl_;
            dx_split8(a, bh_, bl_);
            const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, bh_), bl = __builtin_bit_cast(bf16x8_t, bl_);
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t ah[4], al[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                ah[ib] = __builtin_bit_cast(bf16x8_t, s16x8_t{f.h[ib][0][0], f.h[ib][0][1], f.h[ib][0][2], f.h[ib][0][3], f.h[ib][1][0], f.h[ib][1][1], f.h[ib][1][2], f.h[ib][1][3]});
                al[ib] = __builtin_bit_cast(bf16x8_t, s16x8_t{f.l[ib][0][0], f.l[ib][0][1], f.l[ib][0][2], f.l[ib][0][3], f.l[ib][1][0], f.l[ib][1][1], f.l[ib][1][2], f.l[ib][1][3]});
            }
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ib], bh, acc[ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ib], bl, acc[ib], 0, 0, 0);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ib], bh, acc[ib], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        DxSet f0, f1;
        rdset(0, f0);
#pragma unroll
        for (int kk = 0; kk < 8; kk += 2) {
            rdset(kk + 1, f1);
            block(kk, f0);
            if (kk + 2 < 8) rdset(kk + 2, f0);
            block(kk + 1, f1);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0), unconditionally: on a path without it the compiler's counter model keeps the V loads
        if (gs + 1 < total) convert(gs + 1);                    // pending and waits for them inside the next step's products - i.e. for the DMA pieces queued behind
    };
    for (int gs = 0; gs < total; gs += 2) {
        step(gs, v, vn);
        if (gs + 1 < total) step(gs + 1, vn, v);
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
