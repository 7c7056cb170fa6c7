// p3hip ScoreNet backward, fp32x3: conv2's input gradient and the BatchNorm-1 / ReLU / pair-sum backward of conv1 in ONE kernel, fp32 storage.
//
// Reference: ScoreNet.forward (models/pix2poly/model_pix2poly.py:86-112), as csrc/pair_bwd_mma.hip (the bf16 form of this launch).  Until r05 the fp32x3
// mode formed dA2 = dH2 . W2 with p3_gemm ([B N^2, 256] fp32 = 2.4 GB per net WRITTEN, 962 us) and read it back in p3_pair_bwd (651 us).  Here the 128 x 256
// product tiles stay in the MFMA accumulators, as in the bf16 kernel, with the three bf16 products of the split (a_hi w_hi + a_lo w_hi + a_hi w_lo):
//   * a workgroup owns (tile b, 8 rows i, all j) and walks j in steps of 16: tile row r = jj * 8 + ii (see pair_bwd_mma.hip for why (j, i) order: in the
//     32 x 32 accumulator layout a lane then holds 4 values of j x 4 of its i's and both sums are register adds);
//   * wave w owns output channels 32 w .. 32 w + 31 for ALL 128 rows of a step: its slice of W2^T, split into hi / lo ONCE, lives in 128 registers for the
//     workgroup's life (the bf16 kernel keeps W2^T in LDS: hi + lo would take 128 KB there) - and dU / the BatchNorm sums of a channel are complete inside
//     one wave: no fold through LDS at the end;
//   * the fp32 dH2 tile of step s + 1 (128 rows gathered at stride N, 64 KB) is PARKED in LDS by LDS-DMA while step s computes - no registers - and turned
//     in place into the hi / lo bf16 images at the end of step s: a thread's two 16-byte DMA pieces (floats 0..3 and 4..7 of one 8-k chunk) land in two 1 KB
//     blocks, which are exactly where the chunk's hi and lo image slots live, so every lane converts what its OWN DMA wrote: one s_waitcnt vmcnt(0), no
//     barrier before the conversion, one after it.  Image geometry: 4-row groups of 2 KB (1 KB hi | 1 KB lo), 256-byte rows inside, chunk c of row r at slot
//     c ^ (r & 15): conflict-free ds_read_b128 fragments;
//   * a step runs as two halves of 64 rows (2 accumulators = 32 registers each): products, then the mask / sums epilogue of that half - with fragments
//     double-buffered per 16-deep block (the next block's four reads are in flight during the six MFMAs of this one);
//   * the epilogue has no cross-lane operation: a lane keeps its half-wave's partial of the sum over i (per j) and of sum_j V_j sum_i dz; the partials of a
//     step are folded at the top of the next one, two columns per v_permlane32_swap (+ one add), and stored by all 64 lanes.
// The waits are the compiler's except around the DMA; barriers: one per step.
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int PX_IB = 8;            // rows i per workgroup
constexpr int PX_JT = 16;           // columns j per step: 128 pair rows
constexpr int PX_TILE = 128 * 512;  // one tile: 64 KB as parked fp32, then as hi / lo images
constexpr int PX_LDS = 2 * PX_TILE;

struct PxArgs {
    const float* dH; const float* W2t; const float* U; const float* V;
    const float* sc; const float* sh; const float* mean;
    float* dU; float* dv_slab; float* acc; float* acc_slab;
    int B, N, nblk;
};

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void px_split8(const float4& a, const float4& b, u32x4_t& h, u32x4_t& l) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
        h[k] = hw;
        l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
    }
}

struct PxFrag { u32x4_t ah[2], al[2]; };

__global__ __launch_bounds__(512, 2) void pair_bwd_x3_kernel(PxArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int N = g.N;
    const int b = blockIdx.y, blk = blockIdx.x, i0 = blk * PX_IB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int c = wave * 32 + l31;                                   // this lane's output channel
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds);
    auto dma1 = [&](const void* base, uint32_t dst, uint32_t voff) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
    };
    // ---- W2^T row c, k = 16 kk + 8 hi .. + 8: B fragments of the 8 k-blocks, hi and lo
    bf16x8_t wh[8], wl[8];
    {
        const float* wrow = g.W2t + (int64_t)c * 128 + hi * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const float4 x0 = *reinterpret_cast<const float4*>(wrow + kk * 16), x1 = *reinterpret_cast<const float4*>(wrow + kk * 16 + 4);
            u32x4_t h, l;
            px_split8(x0, x1, h, l);
            wh[kk] = __builtin_bit_cast(bf16x8_t, h); wl[kk] = __builtin_bit_cast(bf16x8_t, l);
        }
    }
    // ---- per-lane constants (see pair_bwd_mma.hip): us = U_i * scale + shift; the ReLU decision of an element is fma(V_j, scale, us) > 0
    const float s_ = g.sc[c];
    float us_[4];
    {
        const float hh = g.sh[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q + 4 * hi;
            us_[q] = i < N ? fmaf(g.U[((int64_t)b * N + min(i, N - 1)) * 256 + c], s_, hh) : -INFINITY;
        }
    }
    // ---- parking: group p = wave * 4 + q = tile rows 4 p .. 4 p + 3 (tile row r = jj * 8 + ii); lane -> (row 4 p + (lane >> 4), slot lane & 15) holds chunk
    // slot ^ (r & 15); its floats 0..3 go to the group's first 1 KB block, floats 4..7 to the second - where the conversion puts the chunk's hi / lo slots
    const float* dHb = g.dH + ((int64_t)b * N + i0) * (int64_t)N * 128;          // pair row (b, i0, 0)
    const float* Vb = g.V + (int64_t)b * N * 256 + c;
    auto park = [&](int st) __attribute__((always_inline)) {
        const int j0 = st * PX_JT;
        const uint32_t dst = lds_addr + (uint32_t)((st & 1) * PX_TILE);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int p = wave * 4 + q, r = p * 4 + (lane >> 4), ck = (lane & 15) ^ (r & 15);
            const int ii = min(r & 7, N - 1 - i0), jj = min(j0 + (r >> 3), N - 1);      // clamped: masked in the epilogue
            const uint32_t voff = (uint32_t)((((int64_t)ii * N + jj) * 128 + ck * 8) * 4);
            dma1(dHb, dst + (uint32_t)(p * 2048), voff);
            dma1(dHb, dst + (uint32_t)(p * 2048 + 1024), voff + 16u);
        }
    };
    auto convert = [&](int st) __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): this lane's own pieces and everything older (the builtin, not asm: the compiler's
        asm volatile("" ::: "memory");                                // counter model then knows the V loads are complete too and inserts no waits of its own for them)
        unsigned char* base = lds + (st & 1) * PX_TILE + wave * 4 * 2048 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x0 = *reinterpret_cast<const float4*>(base + q * 2048), x1 = *reinterpret_cast<const float4*>(base + q * 2048 + 1024);
            u32x4_t h, l;
            px_split8(x0, x1, h, l);
            *reinterpret_cast<u32x4_t*>(base + q * 2048) = h;
            *reinterpret_cast<u32x4_t*>(base + q * 2048 + 1024) = l;
        }
    };
    const int nsteps = (N + PX_JT - 1) / PX_JT;
    float du[4] = {0.f, 0.f, 0.f, 0.f}, a_v = 0.f, dvp[16];      // du: RAW sums of dz over j; a_v, dvp: THIS HALF-WAVE's partials (folded at the store / the end)
    float* slab = g.dv_slab + ((int64_t)b * g.nblk + blk) * (int64_t)N * 256 + c;
    auto store_dv = [&](int st) __attribute__((always_inline)) {    // column j = 16 st + 2 e + hi: lanes 0..31 fold the even column of a pair, 32..63 the odd one
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(dvp[2 * e]), __float_as_uint(dvp[2 * e + 1]), false, false);
            const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            const int j = st * PX_JT + 2 * e + hi;
            if (j < N) slab[(int64_t)j * 256] = tot * s_;
        }
    };
    const int sx = l31 & 15;
    const uint32_t arow = (uint32_t)((l31 >> 2) * 2048 + (l31 & 3) * 256);
    if (nsteps > 0) { park(0); convert(0); }
    for (int st = 0; st < nsteps; ++st) {
        __syncthreads();                                            // images of step st complete; reads of step st - 1 (buffer (st + 1) & 1) done
        const int j0 = st * PX_JT;
        // V loads FIRST: vmcnt retires in order - the epilogue's wait for them must not be a wait for the DMA pieces (HBM) issued behind them
        float v[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) v[jj] = Vb[(int64_t)min(j0 + jj, N - 1) * 256];
        if (st > 0) store_dv(st - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < nsteps) park(st + 1);
        __builtin_amdgcn_sched_barrier(0);                          // the loads above stay above the products
        const unsigned char* Ah = lds + (st & 1) * PX_TILE + arow;
        auto rd = [&](int half, int kk, PxFrag& f) __attribute__((always_inline)) {
            const uint32_t co = (uint32_t)(((2 * kk + hi) ^ sx) * 16);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                const unsigned char* p = Ah + (half * 2 + ib) * 16384 + co;
                f.ah[ib] = *reinterpret_cast<const u32x4_t*>(p);
                f.al[ib] = *reinterpret_cast<const u32x4_t*>(p + 1024);
            }
        };
        PxFrag f0, f1;
        rd(0, 0, f0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 acc[2];
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ib][r] = 0.f;
            auto mma = [&](int kk, const PxFrag& f) __attribute__((always_inline)) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.al[ib]), wh[kk], acc[ib], 0, 0, 0);
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.ah[ib]), wl[kk], acc[ib], 0, 0, 0);
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) acc[ib] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.ah[ib]), wh[kk], acc[ib], 0, 0, 0);
            };
#pragma unroll
            for (int kk = 0; kk < 8; kk += 2) {
                rd(half, kk + 1, f1);
                __builtin_amdgcn_sched_barrier(0);                  // keep the order: four reads, then the six MFMAs that cover their latency
                mma(kk, f0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 2 < 8) rd(half, kk + 2, f0);
                else if (half == 0) rd(1, 0, f0);                   // the second half's first block: in flight during this half's epilogue
                __builtin_amdgcn_sched_barrier(0);
                mma(kk + 1, f1);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue of the half: mask with relu'(bn1(U_i + V_j)); dz summed over j (per i) and over this half-wave's 4 i (per j)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int jj = (half * 2 + ib) * 4 + q4;
                    if (j0 + jj < N) {                                // workgroup-uniform (ragged last step only)
                        float dvs = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dz = fmaf(v[jj], s_, us_[q]) > 0.f ? acc[ib][q4 * 4 + q] : 0.f;
                            du[q] += dz; dvs += dz;
                        }
                        a_v = fmaf(v[jj], dvs, a_v);
                        dvp[jj] = dvs;
                    } else {
                        dvp[jj] = 0.f;
                    }
                }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0) on every path (see convert)
        if (st + 1 < nsteps) convert(st + 1);
    }
    if (nsteps > 0) {
        __builtin_amdgcn_sched_barrier(0);
        store_dv(nsteps - 1);
    }
    // ---- dU[b, i0 + ii, c]; BatchNorm sums over both half-waves:  sum dz (p - mean) = sum_i (U_i - mean) sum_j dz + sum_j V_j sum_i dz   (p = U_i + V_j)
    const float mu_ = g.mean[c];
    float t_sh = 0.f, t_sc = a_v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + q + 4 * hi;
        if (i < N) {
            g.dU[((int64_t)b * N + i) * 256 + c] = du[q] * s_;
            t_sc = fmaf(g.U[((int64_t)b * N + i) * 256 + c] - mu_, du[q], t_sc);      // du of a row beyond N is 0 (never on)
        }
        t_sh += du[q];
    }
    t_sh += __shfl_xor(t_sh, 32, 64);
    t_sc += __shfl_xor(t_sc, 32, 64);
    if (hi == 0) {
        if (g.acc_slab) {
            float* o = g.acc_slab + ((int64_t)b * g.nblk + blk) * 512;
            o[c] = t_sc; o[256 + c] = t_sh;
        } else {
            atomicAdd(g.acc + c, t_sc); atomicAdd(g.acc + 256 + c, t_sh);
        }
    }
}

}  // namespace

void p3_pair_dv_reduce_launch(const float* slab, float* dV, int nblk, int B, int N, int C, hipStream_t s);      // scorenet_bwd.hip

extern "C" int p3_pair_bwd_fused_x3(const float* dH2, const float* W2t, const float* U, const float* V, const float* scale, const float* shift,
                                    const float* mean, float* dU, float* dV, float* acc, int B, int N, void* workspace, void* stream) {
    P3_CHECK(dH2 && W2t && U && V && scale && shift && mean && dU && dV && acc && workspace && B > 0 && N > 0, P3_EINVAL, "p3_pair_bwd_fused_x3: bad arguments");
    P3_CHECK(((uintptr_t)dH2 % 16) == 0 && ((uintptr_t)W2t % 16) == 0, P3_EALIGN, "p3_pair_bwd_fused_x3: 16-byte base alignment");
    hipStream_t s = (hipStream_t)stream;
    PxArgs g;
    g.dH = dH2; g.W2t = W2t; g.U = U; g.V = V;
    g.sc = scale; g.sh = shift; g.mean = mean; g.dU = dU; g.dv_slab = (float*)workspace; g.acc = acc;
    g.B = B; g.N = N; g.nblk = (N + PX_IB - 1) / PX_IB;
    g.acc_slab = p3_det_scratch((int64_t)B * g.nblk * 512, P3_F32);      // the fp32 family reduces in fixed order
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_bwd_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    if (p3_tracing()) p3_note_kernel("pair_bwd_x3_kernel");
    hipLaunchKernelGGL(pair_bwd_x3_kernel, dim3(g.nblk, B), dim3(512), PX_LDS, s, g);
    P3_LAUNCH_CHECK();
    p3_pair_dv_reduce_launch(g.dv_slab, dV, g.nblk, B, N, 256, s);
    P3_LAUNCH_CHECK();
    if (g.acc_slab) return p3_det_reduce(g.acc_slab, B * g.nblk, 512, acc, 512, 1, s);
    return P3_OK;
}
