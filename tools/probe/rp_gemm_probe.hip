// Probe: "row-panel" GEMM for short K (K <= 512) on gfx950.
//   C[M,N] = X[M,K] * W[N,K]^T (+bias), bf16 in, fp32 accumulate.
// A workgroup (4 waves) owns 128 token rows; every wave keeps the MFMA fragments of ITS 32 rows for the whole K in registers
// (K/16 x 16 B per lane) and walks all N in 128-column panels.  W is streamed through a 3-slot LDS ring by LDS-DMA
// (global_load_lds_dwordx4, swizzled on the source side), one 128 x 64 slice per step, the same cyclic slice sequence for every row
// tile.  The product is formed transposed (W fragment = MFMA A operand, X fragment = B operand), so a lane owns one token row of
// the result and the epilogue is row-per-lane with 16-byte stores after a v_permlane32_swap pairing.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 rp_gemm_probe.hip -o rp_gemm_probe
//
// Round-2 findings on MI355X (M = 50240, K = 384, N = 1152; the production 128x128 kernel takes 72 us on this shape), kept here because
// they decide what a successor has to do differently:
//   * correct and 69-73 us as written; no variant beat the production kernel, so it is NOT wired into the library;
//   * pieces in isolation: LDS-DMA stream + X load 23 us, MFMA / LDS loop alone 44-46 us (SQ_LDS_BANK_CONFLICT = 0 with the source-side
//     swizzle; hand-scheduled asm reads two MFMA groups ahead did not move it: the loop is bounded by per-step barrier + first-read
//     latency with both workgroups of a CU phase-locked, and by the epilogue's bias loads), row-per-lane 16-byte stores + 19 us
//     (32 rows x 32 B per instruction = 4x the L1 transactions of a coalesced store: the epilogue must be staged through LDS);
//   * counting epilogue stores in s_waitcnt vmcnt(N) RACES (VAR & 8): loads and stores retire out of order with respect to each other
//     on gfx950; only loads issued after the awaited DMA may be counted, stores have to be issued before the next DMA instead;
//   * a later form with every fix the list above asks for (64-column panels x 128-k slices, LDS-staged 128-B-per-row stores issued one step
//     late, bias from LDS, DMA pieces spread over the step) reached 68 us with compiler-scheduled MFMAs - parity with the production
//     kernel, not a win: the 2-workgroups-per-CU row-panel structure itself is bounded near 45-50 us by barrier / first-read bubbles;
//   * hipcc's own global_load_lds builtin makes it wait vmcnt(0) before the next ds_read (LDS-DMA alias tracking): the DMA has to be
//     inline asm; fragments carried ACROSS asm statements while their ds_read is in flight are moved / spilled by the compiler.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef uint16_t bf16_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}
__device__ __forceinline__ int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }


template <int K, int RING, int VAR>
__global__ __launch_bounds__(256, 2) void rp_gemm(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, bf16_t* __restrict__ C,
                                                  const float* __restrict__ bias, int M, int N, int lda, int ldb, int ldc) {
    constexpr int KS = K / 64;
    __shared__ __attribute__((aligned(16))) uint4 ring[RING * 1024];   // 3 x 16 KB: [128 n-rows][8 chunks of 16 B], chunk slot = c ^ ((row >> 1) & 7)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int row0 = blockIdx.x * 128 + wave * 32;
    const int grow = min(row0 + l31, M - 1);
    // ---- resident X fragments
    uint4 xf[K / 16];
    {
        const bf16_t* xp = X + (int64_t)grow * lda + 8 * hi;
#pragma unroll
        for (int s = 0; s < K / 16; ++s) xf[s] = *reinterpret_cast<const uint4*>(xp + 16 * s);
    }
    const int NP = N / 128, NS = NP * KS;
    // ---- LDS-DMA source offsets of this lane's four chunks of a slice (elements, relative to the slice origin)
    int64_t soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = wave * 256 + i * 64 + lane, row = q >> 3, slot = q & 7, c = slot ^ ((row >> 1) & 7);
        soff[i] = (int64_t)row * ldb + c * 8;
    }
    const uint32_t ring_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(&ring[0]));
    uint32_t voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) voff[i] = (uint32_t)(soff[i] * 2);
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int p = t / KS, ks = t - p * KS;
        const bf16_t* base = W + (int64_t)p * 128 * ldb + ks * 64;
        const uint32_t dst = ring_addr + ((t % RING) * 1024 + wave * 256) * 16;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(base), "s"(dst) : "memory");
    };
    if (!(VAR & 2)) {
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) if (i < NS) issue(i);
    }
    const int sw = (l31 >> 1) & 7;
    for (int p = 0; p < NP; ++p) {
        f32x16 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int t = p * KS + ks;
            {
                // outstanding VMEM ops younger than slice t's DMA: the DMAs of slices t+1 .. t+RING-2 (4 each) and, on the first step of a
                // panel, the 8 row stores of the previous panel's epilogue (VAR & 8: counted, gfx9 returns VMEM in issue order)
                const int later = min(RING - 2, NS - 1 - t);
                const bool st8 = (VAR & 8) && ks == 0 && p > 0 && !(VAR & 4);
                if (st8) {
                    if (later >= 3) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                    else if (later == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else if (later == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    if (later >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (later == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!(VAR & 2) && t + RING - 1 < NS) issue(t + RING - 1);
            const uint4* sb = ring + (t % RING) * 1024;
            if (VAR & 1) continue;
            uint4 bfr[2][4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bfr[0][nb] = sb[(nb * 32 + l31) * 8 + ((hi) ^ sw)];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < 3) {
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) bfr[(kk + 1) & 1][nb] = sb[(nb * 32 + l31) * 8 + ((2 * (kk + 1) + hi) ^ sw)];
                }
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bfr[kk & 1][nb]), __builtin_bit_cast(bf16x8, xf[ks * 4 + kk]), acc[nb], 0, 0, 0);
            }
        }
        // ---- epilogue: lane = token row, acc[nb][r] = column nb*32 + crow32(r, hi)
        const bool rok = row0 + l31 < M;
        bf16_t* crow = C + (int64_t)(row0 + l31) * ldc + p * 128;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            uint32_t pk[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int colb = p * 128 + nb * 32 + 8 * j + 4 * hi;
                const float4 b4 = *reinterpret_cast<const float4*>(bias + colb);
                pk[2 * j] = pack_bf2(acc[nb][4 * j] + b4.x, acc[nb][4 * j + 1] + b4.y);
                pk[2 * j + 1] = pack_bf2(acc[nb][4 * j + 2] + b4.z, acc[nb][4 * j + 3] + b4.w);
            }
            // pk[2j..2j+1] = 4 columns 8j + 4hi ..; pair (j = 0, 1) and (j = 2, 3): lo lanes end with columns 8j .. 8j+7 of the even j, hi lanes of the odd j
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    auto r2 = __builtin_amdgcn_permlane32_swap(pk[4 * jp + w], pk[4 * jp + 2 + w], false, false);
                    pk[4 * jp + w] = r2[0]; pk[4 * jp + 2 + w] = r2[1];
                }
            }
            // now lo lane: pk[0,1] = cols 0..3 (own), pk[2,3] = cols 4..7 (from hi); hi lane: pk[0,1] = cols 8..11 (from lo), pk[2,3] = cols 12..15 (own)
            if (rok && (!(VAR & 4) || M == -1)) {
#pragma unroll
                for (int jp = 0; jp < 2; ++jp)
                    *reinterpret_cast<uint4*>(crow + nb * 32 + 16 * jp + 8 * hi) = make_uint4(pk[4 * jp], pk[4 * jp + 1], pk[4 * jp + 2], pk[4 * jp + 3]);
            }
        }
    }
}


static float bf2f_h(bf16_t v);
template <int K, int RING, int VAR>
void run(bf16_t* dx, bf16_t* dw, bf16_t* dc, float* db, int M, int N, std::vector<bf16_t>& hx, std::vector<bf16_t>& hw, std::vector<float>& hb, bool check) {
    dim3 grid((M + 127) / 128), block(256);
    HIPCHECK(hipMemset(dc, 0xff, (size_t)M * N * 2));
    hipLaunchKernelGGL((rp_gemm<K, RING, VAR>), grid, block, 0, 0, dx, dw, dc, db, M, N, K, K, N);
    HIPCHECK(hipDeviceSynchronize());
    if (check) {
        std::vector<bf16_t> hc((size_t)M * N);
        HIPCHECK(hipMemcpy(hc.data(), dc, hc.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0; int bad = 0;
        for (int it = 0; it < 20000; ++it) {
            const int r = it < 2000 ? (M - 1 - (it % 300)) : rand() % M, c = rand() % N;
            double s = hb[c];
            for (int k = 0; k < K; ++k) s += (double)bf2f_h(hx[(size_t)r * K + k]) * bf2f_h(hw[(size_t)c * K + k]);
            const double e = fabs(s - bf2f_h(hc[(size_t)r * N + c]));
            if (e > maxerr) maxerr = e;
            if (e > 0.02 + 0.01 * fabs(s)) { if (bad < 5) printf("bad at (%d,%d): ref %f got %f\n", r, c, s, bf2f_h(hc[(size_t)r * N + c])); ++bad; }
        }
        printf("M=%d N=%d K=%d RING=%d: max abs err %.4g, bad %d\n", M, N, K, RING, maxerr, bad);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        const int iters = 50;
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rp_gemm<K, RING, VAR>), grid, block, 0, 0, dx, dw, dc, db, M, N, K, K, N);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters;
        if (us < best) best = us;
    }
    printf("  RING=%d VAR=%d: %.1f us  %.0f TF  %.2f TB/s (A + C bytes)\n", RING, VAR, best, 2.0 * M * N * K / best * 1e-6, ((double)M * K * 2 + (double)M * N * 2) / best * 1e-6);
}
static float bf2f_h(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf_h(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (bf16_t)(u >> 16); }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 50240, N = argc > 2 ? atoi(argv[2]) : 1152;
    constexpr int K = 384;
    std::vector<bf16_t> hx((size_t)M * K), hw((size_t)N * K);
    std::vector<float> hb(N);
    srand(1);
    for (auto& v : hx) v = f2bf_h((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto& v : hw) v = f2bf_h((rand() / (float)RAND_MAX) * 0.2f - 0.1f);
    for (auto& v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    bf16_t *dx, *dw, *dc; float* db;
    HIPCHECK(hipMalloc(&dx, hx.size() * 2)); HIPCHECK(hipMalloc(&dw, hw.size() * 2)); HIPCHECK(hipMalloc(&dc, (size_t)M * N * 2)); HIPCHECK(hipMalloc(&db, N * 4));
    HIPCHECK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(dc, 0xff, (size_t)M * N * 2));
    run<K, 3, 0>(dx, dw, dc, db, M, N, hx, hw, hb, true);
    run<K, 3, 8>(dx, dw, dc, db, M, N, hx, hw, hb, true);
    run<K, 4, 8>(dx, dw, dc, db, M, N, hx, hw, hb, true);
    run<K, 3, 1>(dx, dw, dc, db, M, N, hx, hw, hb, false);
    run<K, 3, 5>(dx, dw, dc, db, M, N, hx, hw, hb, false);
    run<K, 5, 5>(dx, dw, dc, db, M, N, hx, hw, hb, false);
    run<K, 3, 9>(dx, dw, dc, db, M, N, hx, hw, hb, false);
    run<K, 3, 6>(dx, dw, dc, db, M, N, hx, hw, hb, false);
    return 0;
}
