// p3hip ScoreNet backward: conv2's input gradient and the BatchNorm-1 / ReLU / pair-sum backward of conv1 in ONE kernel.
//
// Reference: ScoreNet.forward (models/pix2poly/model_pix2poly.py:86-112) builds the pair tensor cat(X_i, X_j), conv1 (1 x 1, separable: U_i + V_j)
// -> bn1 -> relu -> conv2 ...; autograd walks it back through a [B, 256, N, N] gradient.  r01 - r03 here: dA2 = dH2 . W2 (p3_gemm, [B N^2, 256] bf16 =
// 1.2 GB per net WRITTEN, 442 us) followed by p3_pair_bwd (1.2 GB READ, 356 us): mask with relu'(bn1(U_i + V_j)), sum over j into dU, over i into dV,
// BatchNorm sums.  Here a workgroup owns (tile b, 8 rows i, all j) and the 128 x 256 product tiles of dA2 never leave its registers:
//   * rows of a tile are ordered (j, i): 16 columns j x 8 rows i - in the 32 x 32 MFMA accumulator layout a lane then holds, per 32 x 32 block, 4 values
//     of j x 4 of its i's (i = (reg & 3) + 4 hi, j = (reg >> 2) + 4 block + 8 wave row): the sum over the 8 i of one j is 4 adds + one exchange with
//     lane ^ 32 (-> dV partial of this workgroup: stored to a slab, pair_dv_reduce_kernel adds the N / 8 slabs), the sum over j accumulates in 8
//     registers across the whole walk (-> dU: complete inside the workgroup, no atomics);
//   * W2^T (256 x 128 bf16, 64 KB) stays in LDS for the workgroup's life; the dH2 tile (128 rows gathered at stride N, 32 KB) and the V rows of the 16 j
//     (8 KB) are double-buffered by LDS-DMA; U rows / BatchNorm constants sit in registers;
//   * 256-byte operand rows: chunk c of row r sits at slot c ^ (r & 15) (applied on the DMA's source address) - ds_read_b128 fragment reads conflict free;
//   * vmcnt counts loads AND stores and retires them out of order against each other on gfx950, so the wait at the top of a step is vmcnt(0); the dV
//     partial stores of step s are therefore issued at the START of step s + 1 (right after its DMA): they have a whole step to retire.
// 8 waves, acc 64 registers; MFMA 128 x 256 x 128 per step (2061 CU cycles) against 64 masked elements per lane of epilogue VALU (5 operations each).
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int PF_IB = 8;            // rows i per workgroup
constexpr int PF_JT = 16;           // columns j per step: 128 pair rows
constexpr int PF_W_BYTES = 256 * 256, PF_A_BYTES = 128 * 256, PF_V_BYTES = PF_JT * 512;
constexpr int PF_LDS = PF_W_BYTES + 2 * PF_A_BYTES + 2 * PF_V_BYTES;

struct PfArgs {
    const bf16_t* dH; const bf16_t* W2t; const bf16_t* U; const bf16_t* V;
    const float* sc; const float* sh; const float* mean;
    float* dU; float* dv_slab; float* acc; float* acc_slab;
    int B, N, nblk;
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void pair_bwd_mma_kernel(PfArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N;
    const int b = blockIdx.y, blk = blockIdx.x, i0 = blk * PF_IB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto dma1 = [&](const void* base, uint32_t dst, uint32_t voff) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };
    // ---- W2^T -> LDS once: 64 pieces of 4 rows x 256 B, 8 per wave; source chunk = slot ^ (row & 15)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int p = wave * 8 + q, row = p * 4 + (lane >> 4), slot = lane & 15;
        dma1(g.W2t, lds_addr + (uint32_t)(p * 1024), (uint32_t)((row * 128 + ((slot ^ (row & 15)) * 8)) * 2));
    }
    // ---- per-lane constants: the two 32-column blocks of this wave's 64 columns.  us = U_i * scale + shift, so that the ReLU decision of an element
    // is fma(V_j, scale, us) > 0; a row i beyond N gets us = -inf (never on).  The BatchNorm sums are rebuilt from the dU / dV partial sums at the end:
    //   sum dz (p - mean) = sum_i (U_i - mean) sum_j dz  +  sum_j V_j sum_i dz        (p = U_i + V_j)
    // - five VALU operations per element (fma, compare, select, two adds) instead of twelve.
    float s_[2], mu_[2], u_[2][4], us_[2][4];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const int c = wc * 64 + jb * 32 + l31;
        s_[jb] = g.sc[c]; mu_[jb] = g.mean[c];
        const float hh = g.sh[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q + 4 * hi;
            u_[jb][q] = bf2f(g.U[((int64_t)b * N + min(i, N - 1)) * 256 + c]);
            us_[jb][q] = i < N ? fmaf(u_[jb][q], s_[jb], hh) : -INFINITY;
        }
    }
    // ---- staging of a step: dH2 tile rows (jj, ii) -> tile row jj * 8 + ii (4 pieces per wave), V rows of the 16 j (1 piece = 2 rows per wave)
    const bf16_t* dHb = g.dH + ((int64_t)b * N + i0) * (int64_t)N * 128;          // pair row (b, i0, 0)
    const bf16_t* Vb = g.V + (int64_t)b * N * 256;
    auto stage = [&](int st) __attribute__((always_inline)) {
        const int j0 = st * PF_JT, buf = st & 1;
        const uint32_t da = lds_addr + (uint32_t)(PF_W_BYTES + buf * PF_A_BYTES), dv = lds_addr + (uint32_t)(PF_W_BYTES + 2 * PF_A_BYTES + buf * PF_V_BYTES);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave * 4 + q, r = p * 4 + (lane >> 4), slot = lane & 15;      // tile row r = jj * 8 + ii
            const int ii = min(r & 7, N - 1 - i0), jj = min(j0 + (r >> 3), N - 1);      // clamped: masked in the epilogue
            dma1(dHb, da + (uint32_t)(p * 1024), (uint32_t)((((int64_t)ii * N + jj) * 128 + ((slot ^ (r & 15)) * 8)) * 2));
        }
        const int jv = min(j0 + wave * 2 + (lane >> 5), N - 1);
        dma1(Vb, dv + (uint32_t)(wave * 1024), (uint32_t)((jv * 256 + (lane & 31) * 8) * 2));
    };
    const int nsteps = (N + PF_JT - 1) / PF_JT;
    float du[2][4], a_v[2], dvp[2][2][4];         // du: RAW sums of dz over j (scaled at the end); a_v = sum_j V_j sum_i dz
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        a_v[jb] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { du[jb][q] = 0.f; dvp[jb][0][q] = 0.f; dvp[jb][1][q] = 0.f; }
    }
    float* slab = g.dv_slab + ((int64_t)b * g.nblk + blk) * (int64_t)N * 256;
    // dV partials of step `st`: values of lanes hi == 0 (already folded with lane ^ 32), 128-byte coalesced rows
    auto store_dv = [&](int st) __attribute__((always_inline)) {
        if (hi != 0) return;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = st * PF_JT + wr * 8 + ib * 4 + q;
                if (j < N) {
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) slab[(int64_t)j * 256 + wc * 64 + jb * 32 + l31] = dvp[jb][ib][q];
                }
            }
    };
    const int sx = l31 & 15;                                        // swizzle of the fragment rows this lane reads (row & 15)
    const uint32_t arow = (uint32_t)((wr * 64 + l31) * 256), brow = (uint32_t)((wc * 64 + l31) * 256);
    if (nsteps > 0) stage(0);
    for (int st = 0; st < nsteps; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of step st (and W2^T); the stores of step st - 2 are long done
        __builtin_amdgcn_s_barrier();                               // ... every wave's pieces; all reads of step st - 1 (whose buffers step st + 1 takes) are done
        if (st + 1 < nsteps) stage(st + 1);
        if (st > 0) store_dv(st - 1);
        const unsigned char* Ab = lds + PF_W_BYTES + (st & 1) * PF_A_BYTES;
        const unsigned char* Vt = lds + PF_W_BYTES + 2 * PF_A_BYTES + (st & 1) * PF_V_BYTES;
        f32x16 acc[2][2];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ib][jb][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const uint32_t co = (uint32_t)(((2 * kk + hi) ^ sx) * 16);
            u32x4_t af[2], bf[2];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) af[ib] = *reinterpret_cast<const u32x4_t*>(Ab + arow + ib * 32 * 256 + co);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) bf[jb] = *reinterpret_cast<const u32x4_t*>(lds + brow + jb * 32 * 256 + co);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
                    acc[ib][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[ib]), __builtin_bit_cast(bf16x8_t, bf[jb]), acc[ib][jb], 0, 0, 0);
        }
        // ---- epilogue: mask with relu'(bn1(U_i + V_j)); dz summed over j (registers, per i) and over the 8 i (4 here + the other half-wave, per j)
        const int j0 = st * PF_JT;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int jj = wr * 8 + ib * 4 + q4;
                if (j0 + jj < N) {                                    // wave-uniform (ragged last step only)
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb) {
                        const float v = bf2f(*reinterpret_cast<const bf16_t*>(Vt + jj * 512 + (wc * 64 + jb * 32 + l31) * 2));
                        float dvs = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dz = fmaf(v, s_[jb], us_[jb][q]) > 0.f ? acc[ib][jb][q4 * 4 + q] : 0.f;
                            du[jb][q] += dz; dvs += dz;
                        }
                        dvs += __shfl_xor(dvs, 32, 64);
                        a_v[jb] = fmaf(v, dvs, a_v[jb]);              // the folded sum: both half-waves hold the same a_v
                        dvp[jb][ib][q4] = dvs * s_[jb];
                    }
                } else {
                    dvp[0][ib][q4] = 0.f; dvp[1][ib][q4] = 0.f;
                }
            }
    }
    if (nsteps > 0) store_dv(nsteps - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // operand images are dead: their space takes the folds
    // ---- dU[b, i0 + ii, c] = sum over the two wave rows; BatchNorm sums over both half-waves and wave rows
    float* red = reinterpret_cast<float*>(lds);                     // [2][8][256] dU, then [2][2][256] sums
    float* red2 = red + 2 * 8 * 256;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        const int c = wc * 64 + jb * 32 + l31;
        float t_sh = 0.f, t_sc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[(wr * 8 + q + 4 * hi) * 256 + c] = du[jb][q] * s_[jb];
            t_sh += du[jb][q];
            t_sc = fmaf(u_[jb][q] - mu_[jb], du[jb][q], t_sc);
        }
        t_sh += __shfl_xor(t_sh, 32, 64);
        t_sc += __shfl_xor(t_sc, 32, 64);
        t_sc += a_v[jb];                                              // sum_j V_j sum_i dz: computed from the folded sums, the same in both half-waves
        if (hi == 0) { red2[(wr * 2 + 0) * 256 + c] = t_sc; red2[(wr * 2 + 1) * 256 + c] = t_sh; }
    }
    __syncthreads();
    for (int x = tid; x < 8 * 256; x += 512) {
        const int ii = x >> 8, c = x & 255;
        if (i0 + ii < N) g.dU[((int64_t)b * N + i0 + ii) * 256 + c] = red[ii * 256 + c] + red[(8 + ii) * 256 + c];
    }
    {
        const int which = tid >> 8, c = tid & 255;                  // 0: centred scale sums, 1: shift sums
        const float t = red2[which * 256 + c] + red2[(2 + which) * 256 + c];
        if (g.acc_slab) g.acc_slab[((int64_t)b * g.nblk + blk) * 512 + which * 256 + c] = t;
        else atomicAdd(g.acc + which * 256 + c, t);
    }
}

}  // namespace

void p3_pair_dv_reduce_launch(const float* slab, float* dV, int nblk, int B, int N, int C, hipStream_t s);      // scorenet_bwd.hip

extern "C" int64_t p3_pair_bwd_fused_workspace_bytes(int B, int N) { return (int64_t)B * ((N + PF_IB - 1) / PF_IB) * N * 256 * 4; }

extern "C" int p3_pair_bwd_fused(const void* dH2, const void* W2t, const void* U, const void* V, const float* scale, const float* shift, const float* mean,
                                 float* dU, float* dV, float* acc, int B, int N, void* workspace, void* stream) {
    P3_CHECK(dH2 && W2t && U && V && scale && shift && mean && dU && dV && acc && workspace && B > 0 && N > 0, P3_EINVAL, "p3_pair_bwd_fused: bad arguments");
    P3_CHECK(((uintptr_t)dH2 % 16) == 0 && ((uintptr_t)W2t % 16) == 0 && ((uintptr_t)V % 16) == 0, P3_EALIGN, "p3_pair_bwd_fused: 16-byte base alignment");
    P3_CHECK((int64_t)PF_IB * N * 256 < (1ll << 31) && (int64_t)N * 512 < (1ll << 31), P3_EUNSUP, "p3_pair_bwd_fused: N too large for 32-bit DMA offsets");
    hipStream_t s = (hipStream_t)stream;
    PfArgs g;
    g.dH = (const bf16_t*)dH2; g.W2t = (const bf16_t*)W2t; g.U = (const bf16_t*)U; g.V = (const bf16_t*)V;
    g.sc = scale; g.sh = shift; g.mean = mean; g.dU = dU; g.dv_slab = (float*)workspace; g.acc = acc;
    g.B = B; g.N = N; g.nblk = (N + PF_IB - 1) / PF_IB;
    g.acc_slab = p3_det_scratch((int64_t)B * g.nblk * 512, P3_BF16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_bwd_mma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(pair_bwd_mma_kernel, dim3(g.nblk, B), dim3(512), PF_LDS, s, g);
    P3_LAUNCH_CHECK();
    p3_pair_dv_reduce_launch(g.dv_slab, dV, g.nblk, B, N, 256, s);
    P3_LAUNCH_CHECK();
    if (g.acc_slab) return p3_det_reduce(g.acc_slab, B * g.nblk, 512, acc, 512, 1, s);
    return P3_OK;
}
