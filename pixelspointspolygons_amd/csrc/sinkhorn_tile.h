// p3hip - register-resident tiling of the (m+1) x (n+1) coupling matrix shared by the linear-domain Sinkhorn forward (sinkhorn.hip) and
// backward (train_kernels.hip) kernels.
//
// r01-r03 kept E = exp(Z - rowmax) in LDS and ran every iteration's two matrix-vector products as 2 ds_read_b32 per FMA: 1568
// wave-level LDS reads per half iteration = the 128 B/clk/CU ds_read_b32 rate for ~3100 cycles - the loop was LDS-bandwidth bound (forward
// 509 us, backward 730 us for 64 tiles of 193 x 193 x 100 iterations, on 64 of the 256 CUs).  Here the 1024 threads of a tile's workgroup form
// a 64 x 16 grid: thread (ty = tid / 16, tx = tid % 16) owns rows {ty + 64 a} and columns {tx + 16 b} of E in REGISTERS (RA x CB values,
// 4 x 13 for the reference's 193 x 193).
//   row products    s_i = sum_j E_ij x_j : the 16 threads that share a row are the 16 lanes of one DPP row -> four DPP adds, every lane ends
//                   with the same bits (quad_perm / row_half_mirror / row_mirror are symmetric exchanges), no LDS, no barrier;
//   column products t_j = sum_i E_ij y_i : per-thread partials -> LDS slab P[ty][.] -> 4 threads per column add 16 partials each and fold
//                   through two quad_perm adds.  Two barriers per iteration, ~40 LDS instructions per wave.
#pragma once
#include "p3_common.h"

namespace sk {

template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row, the same bits in all 16 lanes (every step adds a value to its symmetric partner's)
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_f<0xB1>(x);     // quad_perm [1,0,3,2]
    x += dpp_f<0x4E>(x);     // quad_perm [2,3,0,1]
    x += dpp_f<0x141>(x);    // row_half_mirror
    x += dpp_f<0x140>(x);    // row_mirror
    return x;
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, dpp_f<0xB1>(x));
    x = fmaxf(x, dpp_f<0x4E>(x));
    x = fmaxf(x, dpp_f<0x141>(x));
    x = fmaxf(x, dpp_f<0x140>(x));
    return x;
}
__device__ __forceinline__ float row16_min(float x) {
    x = fminf(x, dpp_f<0xB1>(x));
    x = fminf(x, dpp_f<0x4E>(x));
    x = fminf(x, dpp_f<0x141>(x));
    x = fminf(x, dpp_f<0x140>(x));
    return x;
}
__device__ __forceinline__ float quad_sum(float x) {
    x += dpp_f<0xB1>(x);
    x += dpp_f<0x4E>(x);
    return x;
}

// base (wave-uniform -> SGPR pair) + 32-bit BYTE offset: the one-VGPR "saddr" addressing form.  With element offsets hipcc builds a 64-bit
// address pair per access and, for the ~100 strided accesses of a thread's tile, spills them.
__device__ __forceinline__ float ld_off(const float* base, uint32_t byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void st_off(float* base, uint32_t byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access (s_waitcnt vmcnt(0)): with the
// per-iteration dual / vector stores of these loops in flight that is a ~1 us store round trip per barrier (r04: 2.5 us per iteration, two
// barriers each).  The stores are consumed by a later launch (or behind the final __syncthreads of the kernel), never inside the loop.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// row pitch of the column-partial slab in floats: 16 CB + 8 keeps the four parts of a column (rows 4k + part) on banks 8 apart
template <int CB> struct Slab { static constexpr int LD = 16 * CB + 8, FLOATS = 64 * LD; };

// column products: every thread hands in its CB partial sums (columns tx + 16 b); returns, in the 4 threads (tid >> 2 == column) of each
// column, the full sum (same bits in the four).  One barrier inside; the caller separates successive calls by another barrier (it has one
// anyway: publishing what it derives from the result).
template <int CB>
__device__ __forceinline__ float col_reduce(float* __restrict__ P, const float (&q)[CB], int tid, int ncols) {
    constexpr int LD = Slab<CB>::LD;
    const int tx = tid & 15, ty = tid >> 4;
#pragma unroll
    for (int b = 0; b < CB; ++b) P[ty * LD + tx + 16 * b] = q[b];
    lds_barrier();
    const int rc = tid >> 2, rp = tid & 3;
    float s = 0.f;
    if (rc < ncols) {
#pragma unroll
        for (int k = 0; k < 16; ++k) s += P[(4 * k + rp) * LD + rc];
    }
    return quad_sum(s);
}

}  // namespace sk
