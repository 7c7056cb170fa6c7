// Probe (gfx950): do MFMA and VALU instructions overlap on one SIMD - inside one wave, and between the two waves of a SIMD - and does it matter whether the MFMA
// accumulators live in VGPRs or in AccVGPRs?  One workgroup per CU (256 x 512 threads = 2 waves per SIMD, or 256 threads = 1 wave per SIMD), a loop of
//   M  : 8 independent v_mfma_f32_32x32x16_bf16 per iteration (4 accumulators, operands fixed)
//   V  : 64 independent v_fma_f32 per iteration
//   MV : both in the same wave, source-interleaved (1 MFMA, 8 FMAs)
//   M|V: 2 waves per SIMD, even hardware wave slot runs M, odd slot runs V
// Times are per launch; the counts are chosen so that M alone and V alone take about the same time.  Build: hipcc --offload-arch=gfx950 -O3 coissue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool AGPR>
__device__ __forceinline__ void mfma(f32x16& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// mode 0: M, 1: V, 2: MV interleaved in one wave, 3: role by hardware wave slot parity (even: M, odd: V)
template <bool AGPR>
__global__ __launch_bounds__(512, 2) void probe(int mode, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(float)(lane + k); b[k] = (__bf16)(float)(lane - k); }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    float x[16];
    for (int k = 0; k < 16; ++k) x[k] = (float)(lane + k);
    const float m = 1.0001f, d = 0.5f;
    const bool even = (__builtin_amdgcn_s_getreg(0x1804) & 1) == 0;
    int role = mode;
    if (mode == 3) role = even ? 0 : 1;
    if (role == 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) mfma<AGPR>(c[u & 3], a, b);
        }
    } else if (role == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(m), "v"(d));
        }
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                mfma<AGPR>(c[u & 3], a, b);
#pragma unroll
                for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(u & 1) * 8 + k]) : "v"(m), "v"(d));
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r];
    for (int k = 0; k < 16; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool AGPR>
static float run(int mode, int threads, int iters, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<AGPR>, dim3(256), dim3(threads), 0, 0, mode, iters, d);      // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<AGPR>, dim3(256), dim3(threads), 0, 0, mode, iters, d);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;          // per wave: 160 k MFMAs (32 cycles each = 5.1 M cycles) or 1.28 M FMAs (4 cycles each = 5.1 M cycles)
    for (int ag = 0; ag < 2; ++ag) {
        printf("accumulators in %s\n", ag ? "AccVGPRs" : "VGPRs");
        auto R = [&](int mode, int threads) { return ag ? run<true>(mode, threads, iters, d) : run<false>(mode, threads, iters, d); };
        printf("  1 wave / SIMD : M %.0f us   V %.0f us   MV (one wave, interleaved) %.0f us\n", R(0, 256), R(1, 256), R(2, 256));
        printf("  2 waves / SIMD: M %.0f us   V %.0f us   MV %.0f us   M|V (one wave each) %.0f us\n", R(0, 512), R(1, 512), R(2, 512), R(3, 512));
    }
    return 0;
}
