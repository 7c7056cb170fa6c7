// r03 probe, moved out of the product library in r04 (default-off since it lost the step-level A/B: 42.55 ms without, 43.35 ms with it): the 256 x 256-tile,
// 8-wave LDS-DMA GEMM with a phased K loop.  It includes csrc/p3_common.h and links against libp3hip objects; numbers: profiles/r03_mb_gemm8.txt, r03_g8_probe.txt.
// p3hip GEMM, 256 x 256 tile, 8 waves, LDS-DMA staging, phased K loop (bf16 in, fp32 accumulate) for the wide plain GEMMs of the path:
//   C[M,N] = epilogue(A[M,K] * W[N,K]^T)            (timm Block: qkv / fc1, their dX products; decoder linear1 / linear2 ...)
//
// Why a second GEMM kernel: the 128 x 128 kernel of gemm.hip (register-staged, one barrier per 32-deep slice, 3 workgroups / CU) tops out
// near 850 TF at any K and at 400 - 650 TF on the K = 384 shapes of the ViT.  This one follows the structure the
// CDNA4 guide measures at 1.3 - 1.5 PF (cdna_hip_programming.md, "The 256^2 8-phase template"):
//   * one workgroup of 8 waves (2 x 4) per CU owns a 256 x 256 tile; a wave owns 128 x 64 = 4 x 2 MFMA 32x32 accumulators (128 registers);
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, inline asm: no staging registers, no ds_write pass), in half-tiles
//     of 128 rows x 64 k = 16 KB (2 instructions per thread), two K-tile buffers = 8 slots = 128 KB;
//   * the LDS image is [row][8 chunks of 16 B] with chunk slot = chunk ^ ((row >> 1) & 7), applied on the SOURCE address of the DMA
//     (the destination of an LDS-DMA is lane-linear): every ds_read_b128 of an MFMA fragment is bank-conflict free;
//   * a K-tile is four phases (one 64 x 32 quadrant of the wave's output each: 8 MFMAs); every phase issues ONE half-tile of DMA for a later
//     K-tile into the slot that became free one phase earlier, so three half-tiles are always in flight and the only wait of the loop is
//     one counted s_waitcnt vmcnt(6) per K-tile (never 0 in the steady state);
//   * epilogue: every wave stages its own 32 x 64 blocks through a private 9 KB piece of the (now idle) operand LDS - no workgroup barrier -
//     and applies bias / GELU (+ aux) / saved-activation factor / dropout / residual on whole 8-column row chunks, 16-byte accesses.
// Hazards: RAW - a half-tile is read only after a barrier that every wave reached AFTER the counted wait that retires its share of the
// DMA (phase 4's wait precedes phase 4's first barrier, the reads are in phase 1 of the next K-tile, two barriers later - still one barrier
// late enough when the second wave group runs one barrier behind); WAR - a slot is re-staged only after a barrier that follows the
// COMPLETION (s_waitcnt lgkmcnt(0) before the phase's first barrier) of its last reads in every wave, lagging group included.
#include <stdio.h>
#include <stdlib.h>

#include "p3_common.h"

namespace {

struct G8Args {
    const bf16_t* A; const bf16_t* W; void* C;
    p3_gemm_desc d;
    int tiles_m, tiles_n;
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int G8_BM = 256, G8_BN = 256, G8_BK = 64;
constexpr int SLOT_U4 = 1024;                         // one half-tile: 128 rows x 8 chunks of 16 B

__device__ __forceinline__ float g8_act_grad(float x, int act) {
    if (act == P3_ACT_MUL) return x;
    if (act == P3_ACT_GELU) { float h, g; gelu_and_grad(x, h, g); return g; }
    return x > 0.f ? 1.f : 0.f;
}

// Loop structures (tools/probe/g8_probe.hip measured them side by side, profiles/r03_g8_probe.txt):
//   STRUCT 0  four phases per K-tile, two barriers each, one half-tile of DMA per phase (the description above);
//   STRUCT 1  the same with the waves of the second row half (wr = 1) one barrier behind the first, so that one group's LDS reads / DMA
//             issue overlap the other group's MFMAs (the guide's `if (wr == 1) s_barrier`): best from ~16 K-tiles on (K = 1536: 84 vs 104 us);
//   STRUCT 2  ONE barrier per K-tile: wait for this wave's pieces of K-tile kt, barrier (K-tile kt readable by all, buffer of kt - 1 free),
//             issue all of K-tile kt + 1, then the four quadrants without any barrier between them: best on the 6-K-tile ViT shapes
//             (qkv 65 vs 73 us, fc1 84 vs 94).
template <typename TO, int STRUCT>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(G8Args g) {
    __shared__ __attribute__((aligned(1024))) uint4 lds[8 * SLOT_U4];       // [buffer 0/1][A_lo, A_hi, B_lo, B_hi][128][8]
    const p3_gemm_desc& d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles);     // consecutive tiles = one A row panel = one XCD's L2
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int nk = d.K / G8_BK;

    // ---- LDS-DMA source offsets (bytes, 32 bit) of this lane's two pieces of each half-tile: piece q of wave w covers rows (2w + q) * 8 ..
    // + 7 of the half-tile, lane -> (row = lane >> 3, slot = lane & 7), source chunk = slot ^ ((row >> 1) & 7).  Rows beyond M / N are clamped
    // to the last row (their products are never stored).
    uint32_t voffA[2][2], voffB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rr = (wave * 2 + q) * 8 + (lane >> 3), slot = lane & 7, c = slot ^ ((rr >> 1) & 7);
            const int ra = min(tm * G8_BM + h * 128 + rr, d.M - 1), rb = min(tn * G8_BN + h * 128 + rr, d.N - 1);
            voffA[h][q] = (uint32_t)(((int64_t)ra * d.lda + c * 8) * 2);
            voffB[h][q] = (uint32_t)(((int64_t)rb * d.ldb + c * 8) * 2);
        }
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(&lds[0]));
    // stage half-tile `which` (0 A_lo, 1 A_hi, 2 B_lo, 3 B_hi) of K-tile kt into buffer kt & 1
    auto stage = [&](int kt, int which) __attribute__((always_inline)) {
        if (kt >= nk) return;
        const bf16_t* base = (which < 2 ? g.A : g.W) + (int64_t)kt * G8_BK;
        const uint32_t dst = lds_addr + (uint32_t)((((kt & 1) * 4 + which) * SLOT_U4 + wave * 128) * 16);
        const uint32_t v0 = which == 0 ? voffA[0][0] : which == 1 ? voffA[1][0] : which == 2 ? voffB[0][0] : voffB[1][0];
        const uint32_t v1 = which == 0 ? voffA[0][1] : which == 1 ? voffA[1][1] : which == 2 ? voffB[0][1] : voffB[1][1];
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(dst) : "memory");
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l31 >> 1) & 7;
    uint4 af[2][4], bfr[2][4];
    const int arow = l31 * 8, brow = ((wc & 1) * 64 + l31) * 8;      // uint4 index of the fragment row inside its half-tile (+ 32-row blocks)
    constexpr bool STAGGER = STRUCT == 1;
    if constexpr (STRUCT == 2) {
        // RAW: vmcnt(0) retires this wave's pieces of K-tile kt (the only DMA in flight), the barrier extends that to every wave's pieces.
        // WAR: a wave reaches the barrier of iteration kt after the MFMAs that consumed its reads of K-tile kt - 1, so after the barrier the
        // buffer of kt - 1 belongs to the DMA of kt + 1.
        stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage(kt + 1, 0); stage(kt + 1, 1); stage(kt + 1, 2); stage(kt + 1, 3);
            const uint4* abuf = lds + ((kt & 1) * 4 + wr) * SLOT_U4;
            const uint4* bbuf = lds + ((kt & 1) * 4 + 2 + (wc >> 1)) * SLOT_U4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[j][kk] = bbuf[brow + j * 256 + ((2 * kk + hi) ^ sw)];
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i][kk] = abuf[arow + i * 256 + ((2 * kk + hi) ^ sw)];
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < 2; ++i) af[i][kk] = abuf[arow + (2 + i) * 256 + ((2 * kk + hi) ^ sw)];
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = half == 0 ? jj : 1 - jj;
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            acc[2 * half + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][kk]), __builtin_bit_cast(bf16x8_t, bfr[j][kk]),
                                                                                          acc[2 * half + i][j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // every wave is done with the operands: the epilogue may overwrite them
    } else {
    // ---- prologue: K-tile 0 whole, then the three half-tiles of K-tile 1 the steady state would have issued in phases 2..4 of "K-tile -1"
    stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
    stage(1, 2); stage(1, 3); stage(1, 0);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (STAGGER) { if (wr == 1) __builtin_amdgcn_s_barrier(); }
    for (int kt = 0; kt < nk; ++kt) {
        const uint4* abuf = lds + ((kt & 1) * 4 + wr) * SLOT_U4;
        const uint4* bbuf = lds + ((kt & 1) * 4 + 2 + (wc >> 1)) * SLOT_U4;
        // ---------------- phase 1: quadrant (rows 0..63, cols 0..31); reads A rows 0..63 and both B blocks; DMA A_hi(kt + 1)
        stage(kt + 1, 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i][kk] = abuf[arow + i * 256 + ((2 * kk + hi) ^ sw)];
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[j][kk] = bbuf[brow + j * 256 + ((2 * kk + hi) ^ sw)];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads COMPLETE before the barrier: the slot may be re-staged right after it
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][kk]), __builtin_bit_cast(bf16x8_t, bfr[0][kk]), acc[i][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 2: quadrant (rows 0..63, cols 32..63); no reads; DMA B_lo(kt + 2) into this K-tile's B slot, whose reads every
        // wave completed before the barrier above (lgkmcnt(0) precedes the MFMAs of phase 1)
        stage(kt + 2, 2);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][kk]), __builtin_bit_cast(bf16x8_t, bfr[1][kk]), acc[i][1], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 3: quadrant (rows 64..127, cols 32..63); reads A rows 64..127; DMA B_hi(kt + 2)
        stage(kt + 2, 3);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i][kk] = abuf[arow + (2 + i) * 256 + ((2 * kk + hi) ^ sw)];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // reads COMPLETE before the barrier: the slot may be re-staged right after it
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][kk]), __builtin_bit_cast(bf16x8_t, bfr[1][kk]), acc[2 + i][1], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 4: quadrant (rows 64..127, cols 0..31); no reads; DMA A_lo(kt + 2) into this K-tile's A slot (last read in phase 3,
        // completed before the barrier above); the counted wait: everything but the three newest half-tiles (B_lo, B_hi, A_lo of kt + 2) has
        // landed, i.e. all of K-tile kt + 1 (its A_hi was issued in phase 1) - read from phase 1 of the next iteration, two barriers later
        stage(kt + 2, 0);
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                acc[2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][kk]), __builtin_bit_cast(bf16x8_t, bfr[0][kk]), acc[2 + i][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (STAGGER) { if (wr == 0) __builtin_amdgcn_s_barrier(); }      // the lagging group catches up: every wave is done with the operands
    }

    // ---- epilogue: per wave, four 32 x 64 blocks through a private fp32 image [32][72] (9 KB of the wave's 16 KB slice of the operand LDS)
    constexpr int EP = 72;
    float* st = reinterpret_cast<float*>(lds) + wave * 4096;
    TO* C = reinterpret_cast<TO*>(g.C);
    TO* aux = reinterpret_cast<TO*>(d.aux);
    const TO* bwd_saved = reinterpret_cast<const TO*>(d.bwd_saved);
    const bool has_res = d.residual != nullptr, res_bf = d.dtype_res == P3_BF16, aux_grad = d.aux_mode == 1;
    const int act = d.act;
    const DropKey dk = drop_key(d.drop);
    const int c8 = (lane & 7) * 8, rl0 = lane >> 3;
    const int col = tn * G8_BN + wc * 64 + c8;
    float bias[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias[k] = (d.bias && col + k < d.N) ? d.bias[col + k] : 0.f;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[crow32(r, hi) * EP + j * 32 + l31] = acc[ib][j][r];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rl = pass * 8 + rl0;
            const int row = tm * G8_BM + wr * 128 + ib * 32 + rl;
            const float4 v0 = *reinterpret_cast<const float4*>(st + rl * EP + c8);
            const float4 v1 = *reinterpret_cast<const float4*>(st + rl * EP + c8 + 4);
            if (row >= d.M || col >= d.N) continue;
            float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3], v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
            const int64_t co = (int64_t)row * d.ldc + col;
            if (act == P3_ACT_GELU) {
                float gd[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float x = v[k]; gelu_and_grad(x, v[k], gd[k]); if (!aux_grad) gd[k] = x; }
                if (aux) {
                    if constexpr (sizeof(TO) == 2) *reinterpret_cast<uint4*>(aux + co) = make_uint4(pack_bf2(gd[0], gd[1]), pack_bf2(gd[2], gd[3]), pack_bf2(gd[4], gd[5]), pack_bf2(gd[6], gd[7]));
                    else { *reinterpret_cast<float4*>(aux + co) = make_float4(gd[0], gd[1], gd[2], gd[3]); *reinterpret_cast<float4*>(aux + co + 4) = make_float4(gd[4], gd[5], gd[6], gd[7]); }
                }
            } else {
                if (aux) {
                    if constexpr (sizeof(TO) == 2) *reinterpret_cast<uint4*>(aux + co) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
                    else { *reinterpret_cast<float4*>(aux + co) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(aux + co + 4) = make_float4(v[4], v[5], v[6], v[7]); }
                }
                if (act == P3_ACT_RELU) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
            }
            if (dk.on) {
                const uint32_t rk = drop_rowkey(dk, (uint64_t)row);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    const uint32_t bits = drop_bits(rk, drop_colkey(dk, (uint32_t)(col + k)));
                    v[k] = drop_keep_lo(dk, bits) ? v[k] * dk.inv_keep : 0.f;
                    v[k + 1] = drop_keep_hi(dk, bits) ? v[k + 1] * dk.inv_keep : 0.f;
                }
            }
            if (bwd_saved) {
                float sv[8];
                if constexpr (sizeof(TO) == 2) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(bwd_saved + co);
                    const uint32_t w[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { sv[2 * k] = __uint_as_float(w[k] << 16); sv[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
                } else {
                    const float4 r0 = *reinterpret_cast<const float4*>(bwd_saved + co);
                    const float4 r1 = *reinterpret_cast<const float4*>(bwd_saved + co + 4);
                    sv[0] = r0.x; sv[1] = r0.y; sv[2] = r0.z; sv[3] = r0.w; sv[4] = r1.x; sv[5] = r1.y; sv[6] = r1.z; sv[7] = r1.w;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] *= g8_act_grad(sv[k], d.bwd_act) * d.bwd_scale;
            }
            if (has_res) {
                const int64_t ro = (int64_t)row * d.ldr + col;
                if (res_bf) {
                    const uint4 rr = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(d.residual) + ro);
                    const uint32_t w[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[2 * k] += __uint_as_float(w[k] << 16); v[2 * k + 1] += __uint_as_float(w[k] & 0xffff0000u); }
                } else {
                    const float4 r0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d.residual) + ro);
                    const float4 r1 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d.residual) + ro + 4);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                }
            }
            if constexpr (sizeof(TO) == 2) *reinterpret_cast<uint4*>(C + co) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            else { *reinterpret_cast<float4*>(C + co) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(C + co + 4) = make_float4(v[4], v[5], v[6], v[7]); }
        }
    }
}

}  // namespace

// 1 when p3_gemm may hand this problem to the 256^2 kernel (plain bf16 A, whole 16-byte row chunks everywhere, 32-bit byte offsets)
int p3_gemm8_eligible(const p3_gemm_desc* d, const void* A, const void* W, const void* C) {
    if (d->dtype_in != P3_BF16 || d->a_mode != P3_A_PLAIN || d->colsum) return 0;
    if (d->K % G8_BK != 0 || d->N % 8 != 0 || d->lda % 8 != 0 || d->ldb % 8 != 0) return 0;
    const int vo = d->dtype_out == P3_BF16 ? 8 : 4;
    if (d->ldc % vo != 0 || ((uintptr_t)C % 16) != 0 || ((uintptr_t)A % 16) != 0 || ((uintptr_t)W % 16) != 0) return 0;
    if (d->aux && (uintptr_t)d->aux % 16 != 0) return 0;
    if (d->bwd_saved && (uintptr_t)d->bwd_saved % 16 != 0) return 0;
    if (d->residual) { const int vr = d->dtype_res == P3_BF16 ? 8 : 4; if (d->ldr % vr != 0 || (uintptr_t)d->residual % 16 != 0) return 0; }
    if ((int64_t)d->M * d->lda * 2 >= (1ll << 31) || (int64_t)d->N * d->ldb * 2 >= (1ll << 31)) return 0;
    return 1;
}

// structure: 0 / 1 / 2 as above, < 0 = by K (one barrier per K-tile for short K, staggered phases from 16 K-tiles on)
int p3_gemm8_launch(const void* A, const void* W, void* C, const p3_gemm_desc* d, int structure, hipStream_t s) {
    G8Args g;
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = C; g.d = *d;
    g.tiles_m = p3_ceil_div(d->M, G8_BM);
    g.tiles_n = p3_ceil_div(d->N, G8_BN);
    dim3 grid(g.tiles_m * g.tiles_n), block(512);
    if (structure < 0) structure = d->K >= 1024 ? 1 : 2;
    if (p3_tracing()) { char nm[96]; snprintf(nm, sizeof(nm), "gemm8_kernel<%s, %d>", d->dtype_out == P3_BF16 ? "bf16" : "float", structure); p3_note_kernel(nm); }
    if (d->dtype_out == P3_BF16) {
        if (structure == 1) hipLaunchKernelGGL((gemm8_kernel<bf16_t, 1>), grid, block, 0, s, g);
        else if (structure == 2) hipLaunchKernelGGL((gemm8_kernel<bf16_t, 2>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm8_kernel<bf16_t, 0>), grid, block, 0, s, g);
    } else {
        if (structure == 1) hipLaunchKernelGGL((gemm8_kernel<float, 1>), grid, block, 0, s, g);
        else if (structure == 2) hipLaunchKernelGGL((gemm8_kernel<float, 2>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm8_kernel<float, 0>), grid, block, 0, s, g);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}
