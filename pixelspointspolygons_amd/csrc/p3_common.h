// p3hip — common device/host helpers (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/p3hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (matches torch .to(bfloat16))
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

template <typename T> struct Cvt;
template <> struct Cvt<float> {
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};
template <> struct Cvt<bf16_t> {
    static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// row index of accumulator register r of a 32x32 MFMA C/D fragment (col = lane & 31)
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

static inline int p3_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

void p3_set_error(const char* msg);
#define P3_CHECK(cond, code, msg) \
    do {                          \
        if (!(cond)) {            \
            p3_set_error(msg);    \
            return (code);        \
        }                         \
    } while (0)
#define P3_LAUNCH_CHECK()                              \
    do {                                               \
        hipError_t e_ = hipGetLastError();             \
        if (e_ != hipSuccess) {                        \
            p3_set_error(hipGetErrorString(e_));       \
            return (int)e_;                            \
        }                                              \
    } while (0)
