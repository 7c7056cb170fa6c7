// r06 probe: what bounds the inner loop of csrc/gemm_x3_as.hip?  One workgroup of NW waves per CU; every wave keeps a 192-register "A slice" and runs
// TICKS x STEPS x (READS fragment reads from LDS + 12 MFMA 16x16x32 on four accumulators | 6 MFMA 32x32x16 on one or two), with or without a barrier per tick.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/as_probe tools/probe/as_probe.hip && /tmp/as_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x32, 4 accumulators; 1: 32x32x16, ONE accumulator (36 dependent per 6 steps); 2: 32x32x16, two accumulators alternating
// READS: LDS fragment reads per step (0 = operands stay in registers);  BAR: s_barrier per tick;  NSLICE: A-slice registers / 4 (48 = the K = 384 kernel)
template <int MODE, bool READS, bool BAR, int NSLICE>
__global__ __launch_bounds__(512, 2) void probe(const u32x4* __restrict__ src, float* __restrict__ out, int ticks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 96 * 1024 / 16; i += blockDim.x) reinterpret_cast<u32x4*>(lds)[i] = src[i & 1023];
    u32x4 A[NSLICE];
#pragma unroll
    for (int i = 0; i < NSLICE; ++i) A[i] = src[(tid + 64 * i) & 1023];
    __syncthreads();
    f32x4 c4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 c16a, c16b;
#pragma unroll
    for (int i = 0; i < 16; ++i) { c16a[i] = 0.f; c16b[i] = 0.f; }
    const uint32_t foff = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) * 16));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < ticks; ++t) {
        if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
        const unsigned char* sb = lds + (t % 4) * 24576;
        u32x4 wf[2][4];
        auto rd = [&](int j) __attribute__((always_inline)) {
            if (READS) {
                uint32_t fo;
                asm volatile("v_xor_b32 %0, %2, %1" : "=v"(fo) : "v"(foff), "n"((j & 1) * 64));
                const unsigned char* gp = sb + (j >> 1) * 4096 + fo;
#pragma unroll
                for (int k = 0; k < 4; ++k) wf[j & 1][k] = *reinterpret_cast<const u32x4*>(gp + (k >> 1) * 12288 + (k & 1) * 2048);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) wf[j & 1][k] = A[(j + k) % NSLICE];
            }
        };
        rd(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (j + 1 < 6) rd(j + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int f = j & 1;
            if (MODE == 0) {
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x4 av = A[(8 * j + 2 * term + (k & 1)) % NSLICE];
                        c4[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[f][(term == 0 ? 2 : 0) + (k >> 1)]), __builtin_bit_cast(bf16x8_t, av), c4[k], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const u32x4 av = A[(8 * j + m) % NSLICE];
                    if (MODE == 1 || (m & 1) == 0) c16a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[f][m & 3]), __builtin_bit_cast(bf16x8_t, av), c16a, 0, 0, 0);
                    else c16b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[f][m & 3]), __builtin_bit_cast(bf16x8_t, av), c16b, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += c4[k][0] + c4[k][1] + c4[k][2] + c4[k][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c16a[i] + c16b[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, bool READS, bool BAR, int NSLICE>
void run(const char* name, int nw, const u32x4* src, float* out, unsigned long long* cyc, int ticks) {
    auto k = probe<MODE, READS, BAR, NSLICE>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(64 * nw), 96 * 1024 + 1024, 0, src, out, ticks, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(64 * nw), 96 * 1024 + 1024, 0, src, out, ticks, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256;
    // matrix-pipe cycles per tick and SIMD: waves per SIMD x (72 MFMA 16x16x32 x 16 | 36 MFMA 32x32x16 x 32) = waves per SIMD x 1152
    const double pipe = (nw / 4.0) * 1152.0;
    printf("%-64s %2d waves: %8.1f us/launch  %7.0f cycles/tick (s_memtime)  pipe %5.1f %%  clock %.2f GHz\n", name, nw, ms / 5 * 1e3, mean / ticks, 100.0 * pipe / (mean / ticks),
           mean / (ms / 5 * 1e-3) / 1e9);
}

int main() {
    u32x4* src; float* out; unsigned long long* cyc;
    hipMalloc(&src, 1024 * 16); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    std::vector<uint16_t> h(1024 * 8);
    uint32_t x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00 + ((x >> 16) & 0x3ff)) ^ (uint16_t)((x >> 3) & 0x8000); }     // random bf16 around +-1
    hipMemcpy(src, h.data(), 1024 * 16, hipMemcpyHostToDevice);
    const int T = 60;
    for (int nw : {8, 4}) {
        run<0, false, false, 48>("16x16x32 x4 acc, operands in registers, no barrier", nw, src, out, cyc, T);
        run<0, true, false, 48>("16x16x32 x4 acc, LDS fragment reads, no barrier", nw, src, out, cyc, T);
        run<0, true, true, 48>("16x16x32 x4 acc, LDS fragment reads, barrier per tick", nw, src, out, cyc, T);
        run<0, true, true, 24>("16x16x32 x4 acc, LDS reads, barrier, 96-register slice", nw, src, out, cyc, T);
        run<1, false, false, 48>("32x32x16 ONE acc, operands in registers, no barrier", nw, src, out, cyc, T);
        run<1, true, true, 48>("32x32x16 ONE acc, LDS fragment reads, barrier per tick", nw, src, out, cyc, T);
        run<2, false, false, 48>("32x32x16 two acc, operands in registers, no barrier", nw, src, out, cyc, T);
        run<2, true, true, 48>("32x32x16 two acc, LDS fragment reads, barrier per tick", nw, src, out, cyc, T);
    }
    return 0;
}
