// p3hip ScoreNet backward, fp32x3: the dual-operand weight-gradient product of conv3, fp32 storage, products as bf16 x 3.
//
//   G [n, c]       += sum over rows r of dH3[r, n] * [y(r, c) > 0]                 (n < 64, c < 128)
//   G2[n, c]       += sum over rows r of dH3[r, n] * [y(r, c) > 0] * H2[r, c]       y = H2 * scale + shift  (BatchNorm-2 in front of the ReLU)
// side by side as one [64, 256] matrix (p3_gemm_tn_ex, b_mode = P3_A_AFFINE_MASK2; reference: autograd of ScoreNet.conv3 / bn2, models/pix2poly/model_pix2poly.py:
// 88-93); csrc/mask2_dw_mma.hip is the bf16 kernel of this launch.  Until r05 the fp32x3 mode ran it on gemm_tn.hip's tile kernel: 613 us per net for 1.8 GB.
// Here: 64-row steps, one workgroup per CU walking every 256th step.  The fp32 rows of step s + 1 are loaded into registers during step s (24 per thread); at the
// end of the step every element is turned ONCE into the MFMA operands, as bf16 images in LDS: dH3 -> hi / lo; H2 -> the mask [fma(h, scale, shift) > 0] (exact in
// bf16) and mask * h -> hi / lo - the decision is taken on the fp32 value, as the forward took it.  The product loop is transposing reads + MFMAs only: per 16 rows
// G takes 2 terms (the mask has no lo part), G2 three.  8 waves = 2 channel blocks of dH3 x 4 column blocks of H2, accumulators (2 x 16 registers) across all steps
// of the workgroup; 256 partial tiles per launch -> slabs + the float64 reduce, or atomics.
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int MX_ROWS = 64;
constexpr int MX_A = MX_ROWS * 128, MX_B = MX_ROWS * 256;              // one bf16 image of dH3 (64 channels) / of H2 (128 columns)
constexpr int MX_OFF_AL = MX_A, MX_OFF_M1 = 2 * MX_A, MX_OFF_M2H = MX_OFF_M1 + MX_B, MX_OFF_M2L = MX_OFF_M2H + MX_B;
constexpr int MX_STEP = MX_OFF_M2L + MX_B;                             // 64 KB
constexpr int MX_LDS = 2 * MX_STEP;

struct MxArgs {
    const float* dH; const float* H2; const float* sc; const float* sh;
    float* C; int ldc; float* slabs; int64_t steps;
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mx_split8(const float (&v)[8], u32x4_t& h, u32x4_t& l) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
        h[k] = hw;
        l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
    }
}

__global__ __launch_bounds__(512, 2) void mask2_dw_x3_kernel(MxArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31;
    const int mi = wave & 1, nq = wave >> 1;                     // dH3 channels [32 mi, +32) x H2 columns [32 nq, +32)
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    const int64_t my_steps = (g.steps - (int64_t)blockIdx.x + (int64_t)gridDim.x - 1) / (int64_t)gridDim.x;
    // staging geometry: dH3 (row tid >> 3, 8-channel chunk tid & 7); H2 (row q * 32 + (tid >> 4), 8-column chunk tid & 15), q = 0, 1
    const int ar = tid >> 3, ack = tid & 7, bck = tid & 15;
    float scv[8], shv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { scv[e] = g.sc[bck * 8 + e]; shv[e] = g.sh[bck * 8 + e]; }
    float4 pa[2], pb[2][2];
    auto fetch = [&](int64_t q) __attribute__((always_inline)) {
        const int64_t row0 = ((int64_t)blockIdx.x + q * (int64_t)gridDim.x) * MX_ROWS;
        const float* ap = g.dH + (row0 + ar) * 64 + ack * 8;
        pa[0] = *reinterpret_cast<const float4*>(ap); pa[1] = *reinterpret_cast<const float4*>(ap + 4);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float* bp = g.H2 + (row0 + k * 32 + (tid >> 4)) * 128 + bck * 8;
            pb[k][0] = *reinterpret_cast<const float4*>(bp); pb[k][1] = *reinterpret_cast<const float4*>(bp + 4);
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        unsigned char* base = lds + buf * MX_STEP;
        {
            const float x[8] = {pa[0].x, pa[0].y, pa[0].z, pa[0].w, pa[1].x, pa[1].y, pa[1].z, pa[1].w};
            u32x4_t h, l;
            mx_split8(x, h, l);
            const uint32_t off = (uint32_t)(ar * 128 + ((ack ^ (((ar >> 1) & 1) << 2)) * 16));
            *reinterpret_cast<u32x4_t*>(base + off) = h;
            *reinterpret_cast<u32x4_t*>(base + MX_OFF_AL + off) = l;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = k * 32 + (tid >> 4);
            const float x[8] = {pb[k][0].x, pb[k][0].y, pb[k][0].z, pb[k][0].w, pb[k][1].x, pb[k][1].y, pb[k][1].z, pb[k][1].w};
            float m[8];
            u32x4_t m1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool on0 = fmaf(x[2 * e], scv[2 * e], shv[2 * e]) > 0.f, on1 = fmaf(x[2 * e + 1], scv[2 * e + 1], shv[2 * e + 1]) > 0.f;
                m1[e] = (on0 ? 0x3f80u : 0u) | (on1 ? 0x3f800000u : 0u);
                m[2 * e] = on0 ? x[2 * e] : 0.f; m[2 * e + 1] = on1 ? x[2 * e + 1] : 0.f;
            }
            u32x4_t h, l;
            mx_split8(m, h, l);
            const uint32_t off = (uint32_t)(r * 256 + ((bck ^ ((r & 3) << 2)) * 16));
            *reinterpret_cast<u32x4_t*>(base + MX_OFF_M1 + off) = m1;
            *reinterpret_cast<u32x4_t*>(base + MX_OFF_M2H + off) = h;
            *reinterpret_cast<u32x4_t*>(base + MX_OFF_M2L + off) = l;
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
        // slot swizzles for TRANSPOSING reads (16 lanes take 32 bytes of each of 4 consecutive rows; see pair_dw_x3.hip): 256-byte rows - chunk ^ ((row & 3) << 2);
        // 128-byte rows - rows r, r + 1 already sit in different bank halves, chunk ^ (((row >> 1) & 1) << 2) separates the pairs
        offa[hh] = (uint32_t)(row * 128 + ((sa ^ (((row >> 1) & 1) << 2)) * 16) + ((li & 3) & 1) * 8);
        offb[hh] = (uint32_t)(row * 256 + ((sb ^ ((row & 3) << 2)) * 16) + ((li & 3) & 1) * 8);
    }
    if (my_steps > 0) { fetch(0); commit(0); }
    for (int64_t q = 0; q < my_steps; ++q) {
        __syncthreads();                                          // images of step q complete; reads of step q - 1 (whose buffer step q + 1 takes) are done
        if (q + 1 < my_steps) fetch(q + 1);
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t base = lds_addr + (uint32_t)((q & 1) * MX_STEP);
#pragma unroll
        for (int kk = 0; kk < MX_ROWS / 16; ++kk) {
            u32x2_t fah[2], fal[2], f1[2], f2h[2], f2l[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const uint32_t oa = base + (uint32_t)(kk * 16 * 128) + offa[hh], ob = base + (uint32_t)(kk * 16 * 256) + offb[hh];
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fah[hh]) : "v"(oa));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fal[hh]) : "v"(oa + (uint32_t)MX_OFF_AL));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f1[hh]) : "v"(ob + (uint32_t)MX_OFF_M1));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f2h[hh]) : "v"(ob + (uint32_t)MX_OFF_M2H));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f2l[hh]) : "v"(ob + (uint32_t)MX_OFF_M2L));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fah[0]), "+v"(fah[1]), "+v"(fal[0]), "+v"(fal[1]), "+v"(f1[0]), "+v"(f1[1]), "+v"(f2h[0]), "+v"(f2h[1]),
                                                  "+v"(f2l[0]), "+v"(f2l[1]));
            auto frag = [](const u32x2_t (&f)[2]) __attribute__((always_inline)) { return __builtin_bit_cast(bf16x8_t, u32x4_t{f[0].x, f[0].y, f[1].x, f[1].y}); };
            const bf16x8_t ah = frag(fah), al = frag(fal), m1 = frag(f1), m2h = frag(f2h), m2l = frag(f2l);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, m1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, m1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, m2h, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, m2l, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, m2h, acc[1], 0, 0, 0);
        }
        if (q + 1 < my_steps) commit((int)((q + 1) & 1));
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

// p3_gemm_tn_ex's hook for P3_A_AFFINE_MASK2 with P3_F32X3 operands: 1 when the shape is not this kernel's (the caller goes on with gemm_tn.hip), else the launch status
int p3_mask2_dw_x3_try(const void* A, const void* B, float* C, int M, int N, int Kb, int lda, int ldb, int ldc, const float* scale, const float* shift,
                       float* slabs, int max_slabs, hipStream_t s) {
    if (N != 64 || Kb != 128 || lda != 64 || ldb != 128 || M % MX_ROWS != 0 || M < MX_ROWS * 256) return 1;
    if ((((uintptr_t)A | (uintptr_t)B) % 16) != 0) return 1;
    MxArgs g;
    g.dH = (const float*)A; g.H2 = (const float*)B; g.sc = scale; g.sh = shift; g.C = C; g.ldc = ldc; g.steps = M / MX_ROWS;
    const int grid = 256;
    g.slabs = (slabs && grid <= max_slabs) ? slabs : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)mask2_dw_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("mask2_dw_x3_kernel");
    hipLaunchKernelGGL(mask2_dw_x3_kernel, dim3(grid), dim3(512), MX_LDS, s, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    if (g.slabs) p3_tn_reduce_launch(g.slabs, C, 64, 256, ldc, grid, s);
    return P3_OK;
}
