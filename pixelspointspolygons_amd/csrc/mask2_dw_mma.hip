// p3hip ScoreNet backward: the dual-operand weight-gradient product of conv3 - a dedicated streaming kernel.
//
//   G [n, c]       += sum over rows r of dH3[r, n] * [y(r, c) > 0]                 (n < 64, c < 128)
//   G2[n, c]       += sum over rows r of dH3[r, n] * [y(r, c) > 0] * H2[r, c]       y = H2 * scale + shift  (BatchNorm-2 in front of the ReLU)
// written side by side as one [64, 256] matrix (p3_gemm_tn_ex, b_mode = P3_A_AFFINE_MASK2): conv3's weight gradient AND the BatchNorm-2 backward sums of its input
// gradient follow from G / G2 (p3_bn_sums_from_g; reference: autograd of ScoreNet.conv3 / bn2, models/pix2poly/model_pix2poly.py:88-93).
// r04 so far: gemm_tn.hip's 128 x 128 tile - half of it idle at N = 64, the generated operand through registers -> ds_write -> transposing read: 267 us for 906 MB.
// Here, in the scheme of pair_dw_mma.hip: 128-row steps of dH3 (16 KB) and H2 (32 KB) double-buffered by LDS-DMA, one workgroup per CU walking every 256th step,
// 8 waves = 2 channel blocks of dH3 x 4 column blocks of H2; both fragments come from transposing reads (ds_read_b64_tr_b16) and the TWO generated operands are
// built in registers from the one H2 fragment (a lane owns ONE column c: scale / shift are two registers): mask ? 1 : 0 and mask ? h : 0 - 2 MFMAs per 16 rows.
// The accumulators (2 x 16 registers) live across all steps of the workgroup; 256 partial tiles per launch -> atomics or slabs + the float64 reduce.
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int MD_A_BYTES = 128 * 128, MD_B_BYTES = 128 * 256, MD_STEP = MD_A_BYTES + MD_B_BYTES;
constexpr int MD_LDS = 2 * MD_STEP;

struct MdArgs {
    const bf16_t* dH; const bf16_t* H2; const float* sc; const float* sh;
    float* C; int ldc; float* slabs; int64_t steps;
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 1) void mask2_dw_mma_kernel(MdArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31;
    const int mi = wave & 1, nq = wave >> 1;                     // dH3 channels [32 mi, +32) x H2 columns [32 nq, +32)
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto dma1 = [&](const void* base, uint32_t dst, uint32_t voff) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };
    const int64_t my_steps = (g.steps - (int64_t)blockIdx.x + (int64_t)gridDim.x - 1) / (int64_t)gridDim.x;
    // step q of this workgroup = rows [128 (blockIdx.x + q gridDim.x), +128): dH3 rows are 128 bytes (8 slots, slot ^ (((row >> 1) & 1) << 2)), H2 rows 256 bytes (16 slots, ^ ((row & 3) << 2))
    auto stage = [&](int64_t q) __attribute__((always_inline)) {
        const int64_t row0 = ((int64_t)blockIdx.x + q * (int64_t)gridDim.x) * 128;
        const int buf = (int)(q & 1);
        const bf16_t* ab = g.dH + row0 * 64;
        const bf16_t* bb = g.H2 + row0 * 128;
        const uint32_t da = lds_addr + (uint32_t)(buf * MD_STEP), db = da + MD_A_BYTES;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            const int p = wave * 2 + p2, r = p * 8 + (lane >> 3), slot = lane & 7;
            dma1(ab, da + (uint32_t)(p * 1024), (uint32_t)((r * 64 + ((slot ^ (((r >> 1) & 1) << 2)) * 8)) * 2));
        }
#pragma unroll
        for (int p4 = 0; p4 < 4; ++p4) {
            const int p = wave * 4 + p4, r = p * 4 + (lane >> 4), slot = lane & 15;
            dma1(bb, db + (uint32_t)(p * 1024), (uint32_t)((r * 128 + ((slot ^ ((r & 3) << 2)) * 8)) * 2));
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    // transposing-read geometry (see pair_dw_mma.hip): lane -> (8-row half g4 >> 1, row li >> 2 of a 4-row piece, 16-column half g4 & 1, 4 columns (li & 3) * 4)
    const int g4 = lane >> 4, li = lane & 15;
    uint32_t offa[2], offb[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int row = (g4 >> 1) * 8 + hh * 4 + (li >> 2);
        const int sa = mi * 4 + (g4 & 1) * 2 + ((li & 3) >> 1), sb = nq * 4 + (g4 & 1) * 2 + ((li & 3) >> 1);
        // slot swizzles for transposing reads (r05; see pair_dw_mma.hip / mask2_dw_x3.hip): 256-byte rows chunk ^ ((row & 3) << 2), 128-byte rows chunk ^ (((row >> 1) & 1) << 2)
        offa[hh] = (uint32_t)(row * 128 + ((sa ^ (((row >> 1) & 1) << 2)) * 16) + ((li & 3) & 1) * 8);
        offb[hh] = (uint32_t)(MD_A_BYTES + row * 256 + ((sb ^ ((row & 3) << 2)) * 16) + ((li & 3) & 1) * 8);
    }
    const float s_ = g.sc[nq * 32 + l31], h_ = g.sh[nq * 32 + l31];
    if (my_steps > 0) stage(0);
    for (int64_t q = 0; q < my_steps; ++q) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (q + 1 < my_steps) stage(q + 1);
        const uint32_t base = lds_addr + (uint32_t)((q & 1) * MD_STEP);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            u32x2_t fa[2], fb[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[hh]) : "v"(base + (uint32_t)(kk * 16 * 128) + offa[hh]));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[hh]) : "v"(base + (uint32_t)(kk * 16 * 256) + offb[hh]));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
            const uint32_t hw[4] = {fb[0].x, fb[0].y, fb[1].x, fb[1].y};
            u32x4_t m1, m2;                                      // [y > 0] and [y > 0] H2 as bf16 pairs
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = __uint_as_float(hw[e] << 16), x1 = __uint_as_float(hw[e] & 0xffff0000u);
                const bool on0 = fmaf(x0, s_, h_) > 0.f, on1 = fmaf(x1, s_, h_) > 0.f;
                m1[e] = (on0 ? 0x3f80u : 0u) | (on1 ? 0x3f800000u : 0u);
                m2[e] = (on0 ? (hw[e] & 0xffffu) : 0u) | (on1 ? (hw[e] & 0xffff0000u) : 0u);
            }
            const bf16x8_t af = __builtin_bit_cast(bf16x8_t, u32x4_t{fa[0].x, fa[0].y, fa[1].x, fa[1].y});
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, m1), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8_t, m2), acc[1], 0, 0, 0);
        }
    }
    const int hi = lane >> 5;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = k * 128 + nq * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = mi * 32 + crow32(r, hi);
            if (g.slabs) g.slabs[((int64_t)blockIdx.x * 64 + n) * 256 + c] = acc[k][r];
            else atomicAdd(g.C + (int64_t)n * g.ldc + c, acc[k][r]);
        }
    }
}

}  // namespace

void p3_tn_reduce_launch(const float* slabs, float* C, int N, int K, int ldc, int splits, hipStream_t s);      // gemm_tn.hip

// p3_gemm_tn_ex's hook for P3_A_AFFINE_MASK2: 1 when the shape is not this kernel's (the caller goes on with gemm_tn.hip), else the launch status
int p3_mask2_dw_try(const void* A, const void* B, float* C, int M, int N, int Kb, int lda, int ldb, int ldc, const float* scale, const float* shift,
                    float* slabs, int max_slabs, hipStream_t s) {
    if (N != 64 || Kb != 128 || lda != 64 || ldb != 128 || M % 128 != 0 || M < 128 * 256) return 1;
    if ((((uintptr_t)A | (uintptr_t)B) % 16) != 0) return 1;
    MdArgs g;
    g.dH = (const bf16_t*)A; g.H2 = (const bf16_t*)B; g.sc = scale; g.sh = shift; g.C = C; g.ldc = ldc; g.steps = M / 128;
    const int grid = 256;
    g.slabs = (slabs && grid <= max_slabs) ? slabs : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)mask2_dw_mma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MD_LDS);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("mask2_dw_mma_kernel");
    hipLaunchKernelGGL(mask2_dw_mma_kernel, dim3(grid), dim3(512), MD_LDS, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (g.slabs) p3_tn_reduce_launch(g.slabs, C, 64, 256, ldc, grid, s);
    return P3_OK;
}
