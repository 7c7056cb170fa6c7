// p3hip GEMM, 128 x 128 tile, 4 waves, LDS-DMA staging (bf16 in, fp32 accumulate): the plain bf16 products of the path
//   C[M,N] = epilogue(A[M,K] * W[N,K]^T)
//
// Between gemm.hip (128 x 128, register-staged: global -> VGPR -> ds_write_b128 -> LDS, 3 workgroups / CU) and gemm8.hip (256 x 256, LDS-DMA,
// 1 workgroup / CU).  The register-staged kernel pays, per 32-deep slice and workgroup, 16 KB of ds_write_b128 at the LDS write rate
// (64 - 85 B/clk: ~ 220 cycles) next to the 128 cycles of ds_read_b128 and the 256 MFMA cycles per SIMD - with three workgroups per CU the
// LDS pipe, not the MFMA pipe, is the busiest unit (880 TF at 8192^3).  Here the operands go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, inline asm): no ds_write pass, no staging registers, and the tile keeps the 2 - 3 workgroups per CU that hide
// the short K loops of the path (K = 256 / 384: the 256 x 256 kernel's single workgroup per CU cannot overlap its epilogue with anything).
//   * LDS image per operand slice: [128 rows][BK / 8 chunks of 16 B], chunk slot = chunk ^ swz(row) applied on the SOURCE address (the
//     destination of an LDS-DMA is lane-linear); swz = (row >> 2) & 3 for 64-byte rows (BK = 32), (row >> 1) & 7 for 128-byte rows
//     (BK = 64): every ds_read_b128 lane group touches 16 distinct 16-byte bank groups;
//   * NBUF slices in LDS, NBUF - 1 in flight: iteration kt waits (counted vmcnt) for this wave's pieces of slice kt, one barrier makes the
//     whole slice readable and proves that slice kt - 1 has been consumed by every wave, then slice kt + NBUF - 1 is issued into that buffer;
//   * epilogue = gemm8.hip's: wave-private fp32 staging of 32 x 64 blocks, whole 8-column chunks, 16-byte accesses.
// Shipped instantiations: BK = 64, NBUF = 2 (64 KB -> 2 workgroups / CU) for K >= 1024; BK = 32, NBUF = 2 (36 KB -> 4 workgroups / CU) for wide outputs with
// K <= 512.  (Measured and dropped, r03: three / four 32-deep slices, a persistent wave-specialised form - tools/probe/gemm_ws_kernel.hip.txt - and the
// 256 x 256 tile of tools/probe/gemm8_probe.hip.)
#include <stdio.h>
#include <stdlib.h>

#include "p3_common.h"

namespace {

struct GDArgs {
    const bf16_t* A; const bf16_t* W; void* C;
    p3_gemm_desc d;
    int tiles_m, tiles_n;
    long long* timeline;        // diagnostic (-DP3_GD_DIAG build, P3_GD_TIMELINE=<device address>, tools/mb_gemm_dma_timeline.py): per workgroup {start, first slice readable, loop end, end} at 100 MHz + HW_ID + XCC_ID
    int ablate;                 // diagnostic (-DP3_GD_DIAG build, P3_GD_ABLATE): 1 no C / aux stores, 2 no MFMAs, 4 only the first slices are loaded
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float gd_act_grad(float x, int act) {
    if (act == P3_ACT_MUL) return x;
    if (act == P3_ACT_GELU) { float h, g; gelu_and_grad(x, h, g); return g; }
    return x > 0.f ? 1.f : 0.f;
}

// one row chunk of 8 columns: v = product + bias -> activation (+ aux) -> dropout -> saved-activation factor -> residual -> C (16-byte accesses)
template <typename TO>
__device__ __forceinline__ void gd_epi8(const p3_gemm_desc& d, const DropKey& dk, TO* C, TO* aux, const TO* bwd_saved, bool has_res, bool res_bf, bool aux_grad,
                                        int act, int row, int col, float (&v)[8]) {
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
        for (int k = 0; k < 8; ++k) v[k] *= gd_act_grad(sv[k], d.bwd_act) * d.bwd_scale;
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

template <int N> __device__ __forceinline__ void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

template <typename TO, int BK, int NBUF>
__global__ __launch_bounds__(256, BK == 32 ? (NBUF == 2 ? 4 : 3) : 2) void gemm_dma_kernel(GDArgs g) {
    constexpr int CPR = BK / 8;                 // 16-byte chunks per row
    constexpr int RPP = 64 / CPR;               // rows per 1 KB DMA piece (one instruction of one wave)
    constexpr int PW = 128 / RPP / 4;           // pieces per wave and operand (2 or 4)
    constexpr int KK = BK / 16;                 // MFMA k-substeps per slice
    constexpr int TILE_U4 = 128 * CPR;          // uint4 per operand slice
    constexpr int LA = NBUF - 1;                // slices in flight
    constexpr int PER_SLICE = 2 * PW;           // DMA instructions per wave and slice
    constexpr int OPER_U4 = NBUF * 2 * TILE_U4, EPI_U4 = 4 * 32 * 72 * 4 / 16;
    __shared__ __attribute__((aligned(1024))) uint4 lds[OPER_U4 > EPI_U4 ? OPER_U4 : EPI_U4];
    const p3_gemm_desc& d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles);     // consecutive tiles = one A row panel = one XCD's L2
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int nk = d.K / BK;
#define GD_T(k) do { if (g.timeline && tid == 0) g.timeline[(int64_t)blockIdx.x * 6 + (k)] = wall_clock64(); } while (0)
    GD_T(0);
    if (g.timeline && tid == 0) { g.timeline[(int64_t)blockIdx.x * 6 + 4] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); g.timeline[(int64_t)blockIdx.x * 6 + 5] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); }

    // LDS-DMA source offsets (bytes, 32 bit): piece q of wave w = rows (w * PW + q) * RPP .. of the slice, lane -> (row = lane / CPR,
    // slot = lane % CPR), source chunk = slot ^ swz(row).  Rows beyond M / N are clamped (their products are never stored).
    uint32_t voffA[PW], voffB[PW];
#pragma unroll
    for (int q = 0; q < PW; ++q) {
        const int rr = (wave * PW + q) * RPP + lane / CPR, slot = lane % CPR;
        const int c = slot ^ (BK == 32 ? ((rr >> 2) & 3) : ((rr >> 1) & 7));
        const int ra = min(tm * 128 + rr, d.M - 1), rb = min(tn * 128 + rr, d.N - 1);
        voffA[q] = (uint32_t)(((int64_t)ra * d.lda + c * 8) * 2);
        voffB[q] = (uint32_t)(((int64_t)rb * d.ldb + c * 8) * 2);
    }
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(&lds[0]));
    auto dma2 = [&](const bf16_t* base, uint32_t dst, uint32_t v0, uint32_t v1) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(dst) : "memory");
    };
    auto stage = [&](int kt) __attribute__((always_inline)) {        // slice kt -> buffer kt % NBUF (caller: kt < nk)
        const int buf = kt % NBUF;
        const bf16_t* ab = g.A + (int64_t)kt * BK;
        const bf16_t* wb = g.W + (int64_t)kt * BK;
        const uint32_t da = lds_addr + (uint32_t)(((buf * 2 + 0) * TILE_U4 + wave * PW * 64) * 16);
        const uint32_t db = lds_addr + (uint32_t)(((buf * 2 + 1) * TILE_U4 + wave * PW * 64) * 16);
        dma2(ab, da, voffA[0], voffA[1]);
        if constexpr (PW == 4) dma2(ab, da + 0x800, voffA[2], voffA[3]);
        dma2(wb, db, voffB[0], voffB[1]);
        if constexpr (PW == 4) dma2(wb, db + 0x800, voffB[2], voffB[3]);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = BK == 32 ? ((l31 >> 2) & 3) : ((l31 >> 1) & 7);
    const int arow = (wr * 64 + l31) * CPR, brow = (wc * 64 + l31) * CPR;        // uint4 index of the fragment row (+ 32 * CPR per block)
#pragma unroll
    for (int p = 0; p < LA; ++p)
        if (p < nk) stage(p);
    for (int kt = 0; kt < nk; ++kt) {
        // RAW: slices kt + 1 .. kt + LA - 1 may stay in flight (they were issued after slice kt: loads retire in order, no store is
        // outstanding in this loop); the barrier extends this wave's wait to every wave's pieces.
        // WAR: a wave reaches this barrier after the MFMAs that consumed its reads of slice kt - 1, whose buffer slice kt + LA takes.
        const int ahead = (g.ablate & 4) ? 0 : min(LA - 1, nk - 1 - kt);
        if (ahead <= 0) wait_vm<0>();
        else if (LA >= 2 && ahead == 1) wait_vm<PER_SLICE>();
        else if (LA >= 3 && ahead == 2) wait_vm<2 * PER_SLICE>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt == 0) GD_T(1);
        if (kt + LA < nk && !(g.ablate & 4)) stage(kt + LA);
        const uint4* abuf = lds + ((kt % NBUF) * 2 + 0) * TILE_U4;
        const uint4* bbuf = lds + ((kt % NBUF) * 2 + 1) * TILE_U4;
        uint4 af[2][KK], bfr[2][KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i][kk] = abuf[arow + i * 32 * CPR + ((2 * kk + hi) ^ sw)];
                bfr[i][kk] = bbuf[brow + i * 32 * CPR + ((2 * kk + hi) ^ sw)];
            }
        }
        if (g.ablate & 2) { acc[0][0][0] += __uint_as_float(af[0][0].x ^ bfr[0][0].y ^ af[1][KK - 1].z ^ bfr[1][KK - 1].w); continue; }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i][kk]), __builtin_bit_cast(bf16x8_t, bfr[j][kk]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // every wave is done with the operands: the epilogue may overwrite them
    GD_T(2);

    // ---- epilogue: per wave, two 32 x 64 blocks through a private fp32 image [32][72] (9 KB)
    constexpr int EP = 72;
    float* st = reinterpret_cast<float*>(lds) + wave * (32 * EP);
    TO* C = reinterpret_cast<TO*>(g.C);
    TO* aux = reinterpret_cast<TO*>(d.aux);
    const TO* bwd_saved = reinterpret_cast<const TO*>(d.bwd_saved);
    const bool has_res = d.residual != nullptr, res_bf = d.dtype_res == P3_BF16, aux_grad = d.aux_mode == 1;
    const int act = d.act;
    const DropKey dk = drop_key(d.drop);
    const int c8 = (lane & 7) * 8, rl0 = lane >> 3;
    const int col = tn * 128 + wc * 64 + c8;
    float bias[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias[k] = (d.bias && col + k < d.N) ? d.bias[col + k] : 0.f;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[crow32(r, hi) * EP + j * 32 + l31] = acc[ib][j][r];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rl = pass * 8 + rl0;
            const int row = tm * 128 + wr * 64 + ib * 32 + rl;
            const float4 v0 = *reinterpret_cast<const float4*>(st + rl * EP + c8);
            const float4 v1 = *reinterpret_cast<const float4*>(st + rl * EP + c8 + 4);
            if (row >= d.M || col >= d.N || ((g.ablate & 1) && v0.x != 12345.678f)) continue;
            float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3], v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
            gd_epi8<TO>(d, dk, C, aux, bwd_saved, has_res, res_bf, aux_grad, act, row, col, v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GD_T(3);
}

// ---- 128 x 384 tile, 8 waves (variant 9): the 384-column outputs of the ViT (proj, fc2, dX of fc1 / qkv) ----------------------------------------
// M = 50240 makes 393 x 3 = 1179 tiles of 128 x 128: 1.5 resident waves of workgroups at 3 per CU, 2.3 at 2 per CU - the last round runs a
// third to a half empty.  One workgroup per 128-row panel (393 workgroups: ONE round at 2 per CU) takes all 384 columns: 8 waves as 2 x 4, a wave
// owns 64 x 96 = 2 x 3 MFMA blocks; the A panel is fetched once instead of three times.  32-deep slices, two in LDS (A 8 KB + W 24 KB each).
template <typename TO>
__global__ __launch_bounds__(512, 4) void gemm_dma_n384_kernel(GDArgs g) {
    constexpr int BK = 32, CPR = 4, A_U4 = 128 * CPR, B_U4 = 384 * CPR, BUF_U4 = A_U4 + B_U4, KK = 2;
    __shared__ __attribute__((aligned(1024))) uint4 lds[2 * BUF_U4];
    const p3_gemm_desc& d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles);
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int nk = d.K / BK;
    uint32_t voffA, voffB[3];
    {
        const int slot = lane & 3;
        const int ra_l = wave * 16 + (lane >> 2);
        voffA = (uint32_t)(((int64_t)min(tm * 128 + ra_l, d.M - 1) * d.lda + (slot ^ ((ra_l >> 2) & 3)) * 8) * 2);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int rb_l = (wave * 3 + q) * 16 + (lane >> 2);
            voffB[q] = (uint32_t)(((int64_t)min(tn * 384 + rb_l, d.N - 1) * d.ldb + (slot ^ ((rb_l >> 2) & 3)) * 8) * 2);
        }
    }
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(&lds[0]));
    auto stage = [&](int kt) __attribute__((always_inline)) {
        const int buf = kt & 1;
        const bf16_t* ab = g.A + (int64_t)kt * BK;
        const bf16_t* wb = g.W + (int64_t)kt * BK;
        const uint32_t da = lds_addr + (uint32_t)((buf * BUF_U4 + wave * 64) * 16);
        const uint32_t db = lds_addr + (uint32_t)((buf * BUF_U4 + A_U4 + wave * 3 * 64) * 16);
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %6\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %6\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %6\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(voffA), "v"(voffB[0]), "v"(voffB[1]), "v"(voffB[2]), "s"(ab), "s"(wb), "s"(da), "s"(db) : "memory");
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sw = (l31 >> 2) & 3;
    const int arow = (wr * 64 + l31) * CPR, brow = (wc * 96 + l31) * CPR;
    stage(0);
    for (int kt = 0; kt < nk; ++kt) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();                  // slice kt readable; slice kt - 1 consumed by every wave: its buffer takes slice kt + 1
        if (kt + 1 < nk) stage(kt + 1);
        const uint4* abuf = lds + (kt & 1) * BUF_U4;
        const uint4* bbuf = abuf + A_U4;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {              // one 16-deep block at a time: 96 accumulator + 20 fragment registers fit 128 (4 waves / SIMD = 2 workgroups / CU)
            uint4 af[2], bfr[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = abuf[arow + i * 32 * CPR + ((2 * kk + hi) ^ sw)];
#pragma unroll
            for (int j = 0; j < 3; ++j) bfr[j] = bbuf[brow + j * 32 * CPR + ((2 * kk + hi) ^ sw)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, af[i]), __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // epilogue: 32 x 32 blocks through a private fp32 image [32][36]
    constexpr int EP = 36;
    float* st = reinterpret_cast<float*>(lds) + wave * (32 * EP);
    TO* C = reinterpret_cast<TO*>(g.C);
    TO* aux = reinterpret_cast<TO*>(d.aux);
    const TO* bwd_saved = reinterpret_cast<const TO*>(d.bwd_saved);
    const bool has_res = d.residual != nullptr, res_bf = d.dtype_res == P3_BF16, aux_grad = d.aux_mode == 1;
    const int act = d.act;
    const DropKey dk = drop_key(d.drop);
    const int c8 = (lane & 3) * 8, rl0 = lane >> 2;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = tn * 384 + wc * 96 + j * 32 + c8;
        float bias[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) bias[k] = (d.bias && col + k < d.N) ? d.bias[col + k] : 0.f;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[crow32(r, hi) * EP + l31] = acc[ib][j][r];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int rl = pass * 16 + rl0;
                const int row = tm * 128 + wr * 64 + ib * 32 + rl;
                const float4 v0 = *reinterpret_cast<const float4*>(st + rl * EP + c8);
                const float4 v1 = *reinterpret_cast<const float4*>(st + rl * EP + c8 + 4);
                if (row >= d.M || col >= d.N) continue;
                float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3], v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
                gd_epi8<TO>(d, dk, C, aux, bwd_saved, has_res, res_bf, aux_grad, act, row, col, v);
            }
        }
    }
}

}  // namespace

int p3_gemm_dma_eligible(const p3_gemm_desc* d, const void* A, const void* W, const void* C) {
    if (d->dtype_in != P3_BF16 || d->a_mode != P3_A_PLAIN || d->colsum) return 0;
    if (d->K % 64 != 0 || d->N % 8 != 0 || d->lda % 8 != 0 || d->ldb % 8 != 0) return 0;
    const int vo = d->dtype_out == P3_BF16 ? 8 : 4;
    if (d->ldc % vo != 0 || ((uintptr_t)C % 16) != 0 || ((uintptr_t)A % 16) != 0 || ((uintptr_t)W % 16) != 0) return 0;
    if (d->aux && (uintptr_t)d->aux % 16 != 0) return 0;
    if (d->bwd_saved && ((uintptr_t)d->bwd_saved % 16 != 0 || d->bwd_act == P3_ACT_BN_RELU)) return 0;
    if (d->residual) { const int vr = d->dtype_res == P3_BF16 ? 8 : 4; if (d->ldr % vr != 0 || (uintptr_t)d->residual % 16 != 0) return 0; }
    if ((int64_t)d->M * d->lda * 2 >= (1ll << 31) || (int64_t)d->N * d->ldb * 2 >= (1ll << 31)) return 0;      // 32-bit DMA source offsets
    return 1;
}

// variant 4: 128 x 128 tile, BK = 64, two slices in LDS (2 workgroups / CU); 6: BK = 32, two slices (36 KB = the epilogue image: 4 / CU); 9: 128 x 384 tile
int p3_gemm_dma_launch(const void* A, const void* W, void* C, const p3_gemm_desc* d, int variant, hipStream_t s) {
    GDArgs g;
    g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.C = C; g.d = *d;
    g.tiles_m = p3_ceil_div(d->M, 128);
    g.tiles_n = p3_ceil_div(d->N, 128);
    g.ablate = 0; g.timeline = nullptr;
#ifdef P3_GD_DIAG
    { const char* e = getenv("P3_GD_ABLATE"); g.ablate = e ? atoi(e) : 0; }
    { const char* e = getenv("P3_GD_TIMELINE"); g.timeline = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    const bool bf = d->dtype_out == P3_BF16;
    if (variant == 9) {
        g.tiles_n = p3_ceil_div(d->N, 384);
        if (p3_tracing()) p3_note_kernel(bf ? "gemm_dma_n384_kernel<bf16>" : "gemm_dma_n384_kernel<float>");
        dim3 ngrid(g.tiles_m * g.tiles_n), nblock(512);
        if (bf) hipLaunchKernelGGL((gemm_dma_n384_kernel<bf16_t>), ngrid, nblock, 0, s, g);
        else hipLaunchKernelGGL((gemm_dma_n384_kernel<float>), ngrid, nblock, 0, s, g);
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    dim3 grid(g.tiles_m * g.tiles_n), block(256);
    if (p3_tracing()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "gemm_dma_kernel<%s, %d, 2>", bf ? "bf16" : "float", variant == 4 ? 64 : 32);
        p3_note_kernel(nm);
    }
    if (variant == 4) {
        if (bf) hipLaunchKernelGGL((gemm_dma_kernel<bf16_t, 64, 2>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_dma_kernel<float, 64, 2>), grid, block, 0, s, g);
    } else {
        if (bf) hipLaunchKernelGGL((gemm_dma_kernel<bf16_t, 32, 2>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_dma_kernel<float, 32, 2>), grid, block, 0, s, g);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}
