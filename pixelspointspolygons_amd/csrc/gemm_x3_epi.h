// p3hip planes GEMMs: the epilogue arithmetic every p3_gemm_x3 kernel shares (gemm_x3.hip, gemm_x3_as.hip)
#pragma once
#include "p3_common.h"

namespace {

__device__ __forceinline__ void x3_split8(const float (&v)[8], uint4& h, uint4& l) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        hw[k] = pack_bf2(v[2 * k], v[2 * k + 1]);
        const float r0 = v[2 * k] - __uint_as_float(hw[k] << 16), r1 = v[2 * k + 1] - __uint_as_float(hw[k] & 0xffff0000u);
        lw[k] = pack_bf2(r0, r1);
    }
    h = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    l = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}

// one row chunk of 8 columns: v = product + bias -> GELU (+ aux <- GELU') -> * mul -> + residual -> C (fp32 or planes); 16-byte accesses
template <bool PLANES>
__device__ __forceinline__ void x3_epi8(const p3_gemm_x3_desc& d, int row, int col, float (&v)[8]) {
    if (d.act == P3_ACT_GELU) {
        float gd[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gelu_and_grad(v[k], v[k], gd[k]);
        if (d.aux) {
            float* a = d.aux + (int64_t)row * d.ldaux + col;
            *reinterpret_cast<float4*>(a) = make_float4(gd[0], gd[1], gd[2], gd[3]);
            *reinterpret_cast<float4*>(a + 4) = make_float4(gd[4], gd[5], gd[6], gd[7]);
        }
    }
    if (d.mul) {
        const float* m = d.mul + (int64_t)row * d.ldmul + col;
        const float4 m0 = *reinterpret_cast<const float4*>(m), m1 = *reinterpret_cast<const float4*>(m + 4);
        v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w; v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
    }
    if (d.residual) {
        const float* r = d.residual + (int64_t)row * d.ldr + col;
        const float4 r0 = *reinterpret_cast<const float4*>(r), r1 = *reinterpret_cast<const float4*>(r + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    const int64_t co = (int64_t)row * d.ldc + col;
    if constexpr (PLANES) {
        uint4 h, l;
        x3_split8(v, h, l);
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.c) + co) = h;
        *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.c_lo) + co) = l;
    } else {
        float* c = reinterpret_cast<float*>(d.c) + co;
        *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace
