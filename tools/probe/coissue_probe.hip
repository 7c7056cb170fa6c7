// r06 probe: do MFMA and VALU instructions overlap on a SIMD?  256 workgroups; every wave runs ITER rounds of NM dependent-free MFMA 32x32x16 (two accumulators) and / or
// NV independent v_fma_f32 (eight chains), in one wave (interleaved by the compiler inside one basic block) or split over the waves of a SIMD (even waves multiply, odd
// waves do the vector work).  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/probe/coissue_probe.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE 0: MFMA only; 1: VALU only; 2: both in every wave; 3: waves 0..3 MFMA, waves 4..7 VALU (one of each per SIMD); 4: both in every wave, exp instead of fma
template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(const float* __restrict__ src, float* __restrict__ out, int iters) {
    const int tid = threadIdx.x, wave = tid >> 6;
    f32x16 c0, c1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)src[(tid + i) & 255]; b[i] = (__bf16)src[(tid + 8 + i) & 255]; }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[(tid + 16 + i) & 255];
    const float m = src[3], ad = src[5];
    const bool do_m = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && wave < 4);
    const bool do_v = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_m && do_v) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = MODE == 4 ? __builtin_amdgcn_exp2f(v[i]) : fmaf(v[i], m, ad);
            }
        } else if (do_m) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
            }
        } else if (do_v) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], m, ad);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int MODE>
void run(const char* name, int nw, const float* src, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * nw), 0, 0, src, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(64 * nw), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per round and wave: 16 MFMA 32x32x16 (512 matrix-pipe cycles) and / or 128 v_fma_f32 (512 VALU cycles at 4 per wave64 instruction)
    printf("%-78s %d waves/CU: %8.1f us/launch  %6.1f ns per round\n", name, nw, ms / 5 * 1e3, ms / 5 * 1e6 / iters);
}

int main() {
    float* src; float* out;
    hipMalloc(&src, 256 * 4); hipMalloc(&out, 256 * 512 * 4);
    float h[256];
    for (int i = 0; i < 256; ++i) h[i] = 0.5f + 0.001f * i;
    h[3] = 0.999f; h[5] = 0.001f;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const int IT = 2000;
    run<0>("MFMA only: 16 x 32x32x16 per round, every wave", 4, src, out, IT);
    run<1>("VALU only: 128 v_fma_f32 per round, every wave", 4, src, out, IT);
    run<2>("both in every wave (one basic block)", 4, src, out, IT);
    run<4>("both in every wave, v_exp_f32 instead of v_fma_f32", 4, src, out, IT);
    run<0>("MFMA only, every wave", 8, src, out, IT);
    run<1>("VALU only, every wave", 8, src, out, IT);
    run<2>("both in every wave", 8, src, out, IT);
    run<3>("waves 0..3 MFMA only, waves 4..7 VALU only (one of each per SIMD)", 8, src, out, IT);
    return 0;
}
