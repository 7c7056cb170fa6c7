// p3hip GEMM on PLANES (include/p3hip.h, p3_gemm_x3):  C = epilogue((a_hi + a_lo) (w_hi + w_lo)^T),  three bf16 MFMAs per product
//
// The fp32x3 precision (P3_F32X3) multiplies fp32 operands as a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA.  gemm.hip's form of it keeps the operands
// fp32 in HBM and splits every value while its slice is staged global -> VGPR -> LDS (4 VALU ops + two ds_write_b64 per 4 values, 24 staging registers): at the
// ViT's shapes that kernel ran 173 us (qkv) ... 302 us (fc1) per launch, 27.5 ms of a 76.7 ms step (profiles/r05_g01_shapes_fp32x3.txt).  Here the PRODUCER of
// an operand has already written the split (two bf16 matrices, 4 bytes per value like the fp32 tensor they replace), so that
//   * all four operand images of a 32-deep slice go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no VALU, no ds_write;
//   * a wave reads EIGHT fragments (a_hi, a_lo x 2 row blocks; w_hi, w_lo x 2 column blocks) per 16-deep step and issues TWELVE MFMAs on them - 2/3 of a
//     fragment read per MFMA against the bf16 kernel's 1: the matrix pipe, not the LDS pipe or the L2 -> LDS path, is the busy unit of the main loop
//     (98 MFMA-flop per staged byte against the bf16 kernel's 64);
//   * the epilogue writes fp32, or planes again (GELU output, the hidden gradient): the next GEMM's operand, produced in the registers that hold it.
// LDS image of an operand slice: [rows][4 chunks of 16 B], chunk slot = chunk ^ ((row >> 2) & 3) applied on the SOURCE address (gemm_dma.hip's layout:
// every ds_read_b128 lane group touches 16 distinct 16-byte bank groups).  Two slices in LDS, one in flight.
//   128 x 128 tile, 4 waves (2 x 2), 64 KB of operands -> 2 workgroups / CU: the wide outputs (qkv 1152, fc1 / dX of fc2 1536);
//   128 x 384 tile, 8 waves (2 x 4), 128 KB -> 1 workgroup / CU: the 384-column outputs (proj, fc2, dX of fc1 / qkv / proj) - the A panel is staged once,
//     and the whole output row lies in one workgroup (the fused LayerNorm of the output row: ln_* of the descriptor).
#include <stdio.h>
#include <stdlib.h>

#include "p3_common.h"
#include "gemm_x3_epi.h"

namespace {

struct X3Args {
    p3_gemm_x3_desc d;
    int tiles_m, tiles_n;
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void x3_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// two 1 KB LDS-DMA pieces (consecutive in LDS) from one base pointer
__device__ __forceinline__ void x3_dma2(const bf16_t* base, uint32_t dst, uint32_t v0, uint32_t v1) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(dst) : "memory");
}
__device__ __forceinline__ void x3_dma1(const bf16_t* base, uint32_t dst, uint32_t v0) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(v0), "s"(base), "s"(dst) : "memory");
}

// ---- 128 x 128 tile, 4 waves, 64 KB of LDS -> TWO workgroups per CU ---------------------------------------------------------------------------
// Same software-pipelined loop as the 128 x 384 kernel below (fragments double-buffered per 16-deep block, the next slice's DMA pieces and the next block's
// fragment reads issued in the gaps between the MFMAs, one barrier in the middle of the iteration).  What the smaller tile buys: two INDEPENDENT workgroups
// share a CU - their barriers are not coupled, so while one sits in its epilogue (for the K = 384 products with 1152 / 1536-wide outputs the epilogue moves
// more bytes than the main loop stages) or in a barrier the other one's MFMAs fill the pipe; and 393 x N / 128 tiles quantise better over 256 CUs than 393 x
// N / 384 (N = 384: 1179 tiles = 4.6 rounds against 393 = 1.54 rounds of the big tile).  What it costs: 98 MFMA-flop per staged byte against 147.
struct X3Frag2 { uint4 ah[2], al[2], bh[2], bl[2]; };

template <bool PLANES>
__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(X3Args g) {
    constexpr int CPR = 4, TILE_U4 = 128 * CPR, STAGE_U4 = 4 * TILE_U4;      // a_hi | a_lo | w_hi | w_lo
    constexpr int OPER_U4 = 2 * STAGE_U4, EPI_U4 = 4 * 32 * 72 * 4 / 16;
    __shared__ __attribute__((aligned(1024))) uint4 lds[OPER_U4 > EPI_U4 ? OPER_U4 : EPI_U4];
    const p3_gemm_x3_desc& d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int ntiles = g.tiles_m * g.tiles_n;
    const int bid = xcd_remap(blockIdx.x, ntiles);        // consecutive tiles = one A row panel = one XCD's L2
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int nk = d.K / 32;
    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(d.a_hi);
    const bf16_t* Al = reinterpret_cast<const bf16_t*>(d.a_lo);
    const bf16_t* Wh = reinterpret_cast<const bf16_t*>(d.w_hi);
    const bf16_t* Wl = reinterpret_cast<const bf16_t*>(d.w_lo);

    // LDS-DMA source offsets: piece q of wave w = rows (w * 2 + q) * 16 .. + 15 of the slice, lane -> (row = lane / 4, slot = lane % 4), source chunk = slot ^ swz(row)
    uint32_t voffA[2], voffB[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rr = (wave * 2 + q) * 16 + (lane >> 2), slot = lane & 3;
        const int c = slot ^ ((rr >> 2) & 3);
        const int ra = min(tm * 128 + rr, d.M - 1), rb = min(tn * 128 + rr, d.N - 1);
        voffA[q] = (uint32_t)(((int64_t)ra * d.lda + c * 8) * 2);
        voffB[q] = (uint32_t)(((int64_t)rb * d.ldb + c * 8) * 2);
    }
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(&lds[0]));
    auto dma_part = [&](int kt, int n) __attribute__((always_inline)) {      // n = 0 a_hi, 1 a_lo, 2 w_hi, 3 w_lo: two pieces each
        const uint32_t base = lds_addr + (uint32_t)(((kt & 1) * STAGE_U4 + n * TILE_U4 + wave * 2 * 64) * 16);
        const int64_t ko = (int64_t)kt * 32;
        if (n == 0) x3_dma2(Ah + ko, base, voffA[0], voffA[1]);
        else if (n == 1) x3_dma2(Al + ko, base, voffA[0], voffA[1]);
        else if (n == 2) x3_dma2(Wh + ko, base, voffB[0], voffB[1]);
        else x3_dma2(Wl + ko, base, voffB[0], voffB[1]);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sw = (l31 >> 2) & 3;
    const int arow = (wr * 64 + l31) * CPR, brow = 2 * TILE_U4 + (wc * 64 + l31) * CPR;
    auto frag_read = [&](X3Frag2& f, const uint4* sb, int kk, int n) __attribute__((always_inline)) {       // n = 0-1 a_hi, 2-3 a_lo, 4-5 w_hi, 6-7 w_lo
        const int ch = (2 * kk + hi) ^ sw;
        if (n < 2) f.ah[n] = sb[arow + n * 32 * CPR + ch];
        else if (n < 4) f.al[n - 2] = sb[TILE_U4 + arow + (n - 2) * 32 * CPR + ch];
        else if (n < 6) f.bh[n - 4] = sb[brow + (n - 4) * 32 * CPR + ch];
        else f.bl[n - 6] = sb[TILE_U4 + brow + (n - 6) * 32 * CPR + ch];
    };
    auto mfma_n = [&](const X3Frag2& f, int n) __attribute__((always_inline)) {      // n = 0 .. 11: small terms first, term-major
        const int t = n / 4, i = (n % 4) / 2, j = n % 2;
        const uint4& av = t == 0 ? f.al[i] : f.ah[i];
        const uint4& bv = t == 1 ? f.bl[j] : f.bh[j];
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv), acc[i][j], 0, 0, 0);
    };
    X3Frag2 F0, F1;
#pragma unroll
    for (int n = 0; n < 4; ++n) dma_part(0, n);
    if (nk > 1) {
#pragma unroll
        for (int n = 0; n < 4; ++n) dma_part(1, n);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // slice 0 landed (this wave's pieces); slice 1 stays in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int n = 0; n < 8; ++n) frag_read(F0, lds, 0, n);
    for (int kt = 0; kt < nk; ++kt) {
        const uint4* sb = lds + (kt & 1) * STAGE_U4;
        const uint4* sn = lds + ((kt + 1) & 1) * STAGE_U4;
        const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int n = 0; n < 12; ++n) {                       // block (kt, 0) on F0; the reads of block (kt, 1) fill F1 in the gaps
            mfma_n(F0, n);
            if (n % 3 == 2 && n < 11) {
                __builtin_amdgcn_sched_barrier(0);
                frag_read(F1, sb, 1, (n / 3) * 3); frag_read(F1, sb, 1, (n / 3) * 3 + 1);
                if ((n / 3) * 3 + 2 < 8) frag_read(F1, sb, 1, (n / 3) * 3 + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // slice kt + 1 landed (this wave's pieces); this wave's reads of slice kt done
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int n = 0; n < 12; ++n) {                       // block (kt, 1) on F1; DMA of slice kt + 2 and the reads of block (kt + 1, 0) in the gaps
            mfma_n(F1, n);
            if (n % 3 == 2 && n < 11) {
                __builtin_amdgcn_sched_barrier(0);
                const int sl = n / 3;                      // 0 .. 2
                if (more2) { dma_part(kt + 2, sl); if (sl == 2) dma_part(kt + 2, 3); }
                if (more) { frag_read(F0, sn, 0, sl * 3); frag_read(F0, sn, 0, sl * 3 + 1); if (sl * 3 + 2 < 8) frag_read(F0, sn, 0, sl * 3 + 2); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // every wave is done with the operands: the epilogue may overwrite them

    // ---- epilogue: per wave, two 32 x 64 blocks through a private fp32 image [32][72]
    constexpr int EP = 72;
    float* st = reinterpret_cast<float*>(lds) + wave * (32 * EP);
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
            if (row >= d.M || col >= d.N) continue;
            float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3], v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
            x3_epi8<PLANES>(d, row, col, v);
        }
    }
}

// ---- 128 x 384 tile, 8 waves, software-pipelined main loop ------------------------------------------------------------------------------------
// r05 measurements that shaped it (profiles/r05_x3_kernel_notes.txt): the plain form - wait, barrier, issue the next slice's 8 LDS-DMA pieces, read 20 fragments,
// 36 MFMAs - kept the matrix pipe 31 - 42 % busy, and a four-slice ring with three slices in flight behind a counted vmcnt changed NOTHING (fc2 230 -> 250 us): the
// waves were not waiting for data.  Both waves of a SIMD leave the barrier together, both spend the next ~1000 cycles ISSUING (a DMA piece costs 60 - 185
// issue cycles, MI355X_MICROARCH.md) and waiting for their fragment reads, and only then does either of them have an MFMA to issue.  So here the issue work rides
// in the gaps BETWEEN the MFMAs of the same wave (an MFMA occupies the pipe for 32 cycles, the wave's issue port for ~4):
//   * fragments are double-buffered per 16-deep block: while the 18 MFMAs of block (kt, 0) run on set F0, the reads of block (kt, 1) fill F1, two per gap;
//   * while the 18 MFMAs of block (kt, 1) run on F1, the wave issues its 8 DMA pieces of slice kt + 2 and reads block (kt + 1, 0) into F0;
//   * one barrier per 32-deep slice, in the MIDDLE of the iteration (slice kt + 1 readable / slice kt's LDS image dead), vmcnt(0) there: the only DMA in
//     flight is slice kt + 1, issued a whole iteration (36 MFMAs x 2 waves = 2304 pipe cycles) earlier.
// 64-byte rows (slot = chunk ^ ((row >> 2) & 3) on the source address), two slices of 64 KB in LDS.  The tile serves every ViT product: N = 384 with the whole
// output row in one workgroup (LN = true: LayerNorm of that row in the epilogue), N = 1152 / 1536 as 3 / 4 column tiles (consecutive workgroups = one A panel).
struct X3Frag { uint4 ah[2], al[2], bh[3], bl[3]; };

template <bool LN>
__global__ __launch_bounds__(512, 2) void gemm_x3_n384_kernel(X3Args g) {
    constexpr int CPR = 4, A_U4 = 128 * CPR, B_U4 = 384 * CPR, STAGE_U4 = 2 * A_U4 + 2 * B_U4;      // a_hi | a_lo | w_hi | w_lo = 64 KB
    extern __shared__ __attribute__((aligned(1024))) uint4 lds[];
    const p3_gemm_x3_desc& d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2, wc = wave & 3;
    const int bid = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int nk = d.K / 32;
    const bf16_t* Ah = reinterpret_cast<const bf16_t*>(d.a_hi);
    const bf16_t* Al = reinterpret_cast<const bf16_t*>(d.a_lo);
    const bf16_t* Wh = reinterpret_cast<const bf16_t*>(d.w_hi);
    const bf16_t* Wl = reinterpret_cast<const bf16_t*>(d.w_lo);
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
    // the 8 DMA pieces of a slice, individually placeable: n = 0 a_hi, 1 a_lo, 2 w_hi (two pieces), 3 w_hi (third), 4 w_lo (two), 5 w_lo (third)
    auto dma_part = [&](int kt, int n) __attribute__((always_inline)) {
        const uint32_t sbase = lds_addr + (uint32_t)((kt & 1) * STAGE_U4 * 16);
        const int64_t ko = (int64_t)kt * 32;
        const uint32_t da = sbase + (uint32_t)(wave * 64 * 16);
        const uint32_t db = sbase + (uint32_t)((2 * A_U4 + wave * 3 * 64) * 16);
        if (n == 0) x3_dma1(Ah + ko, da, voffA);
        else if (n == 1) x3_dma1(Al + ko, da + A_U4 * 16, voffA);
        else if (n == 2) x3_dma2(Wh + ko, db, voffB[0], voffB[1]);
        else if (n == 3) x3_dma1(Wh + ko, db + 0x800, voffB[2]);
        else if (n == 4) x3_dma2(Wl + ko, db + B_U4 * 16, voffB[0], voffB[1]);
        else x3_dma1(Wl + ko, db + B_U4 * 16 + 0x800, voffB[2]);
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sw = (l31 >> 2) & 3;
    const int arow = (wr * 64 + l31) * CPR, brow = 2 * A_U4 + (wc * 96 + l31) * CPR;
    // fragment read n (0 .. 9) of 16-deep block kk of the slice at sb: 0-1 a_hi, 2-3 a_lo, 4-6 w_hi, 7-9 w_lo
    auto frag_read = [&](X3Frag& f, const uint4* sb, int kk, int n) __attribute__((always_inline)) {
        const int ch = (2 * kk + hi) ^ sw;
        if (n < 2) f.ah[n] = sb[arow + n * 32 * CPR + ch];
        else if (n < 4) f.al[n - 2] = sb[A_U4 + arow + (n - 2) * 32 * CPR + ch];
        else if (n < 7) f.bh[n - 4] = sb[brow + (n - 4) * 32 * CPR + ch];
        else f.bl[n - 7] = sb[B_U4 + brow + (n - 7) * 32 * CPR + ch];
    };
    // MFMA n (0 .. 17) of a block, small terms first, term-major: six independent accumulators between two MFMAs on the same one
    auto mfma_n = [&](const X3Frag& f, int n) __attribute__((always_inline)) {
        const int t = n / 6, i = (n % 6) / 3, j = n % 3;
        const uint4& av = t == 0 ? f.al[i] : f.ah[i];
        const uint4& bv = t == 1 ? f.bl[j] : f.bh[j];
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv), acc[i][j], 0, 0, 0);
    };
    X3Frag F0, F1;
#pragma unroll
    for (int n = 0; n < 6; ++n) dma_part(0, n);
    if (nk > 1) {
#pragma unroll
        for (int n = 0; n < 6; ++n) dma_part(1, n);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // slice 0 landed (this wave's pieces); slice 1 stays in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int n = 0; n < 10; ++n) frag_read(F0, lds, 0, n);
    for (int kt = 0; kt < nk; ++kt) {
        const uint4* sb = lds + (kt & 1) * STAGE_U4;
        const uint4* sn = lds + ((kt + 1) & 1) * STAGE_U4;
        const bool more = kt + 1 < nk, more2 = kt + 2 < nk;
        // ---- block (kt, 0) on F0; the reads of block (kt, 1) fill F1 in the gaps
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int n = 0; n < 18; ++n) {
            mfma_n(F0, n);
            if (n % 3 == 2 && n < 15) {
                __builtin_amdgcn_sched_barrier(0);
                frag_read(F1, sb, 1, (n / 3) * 2);
                frag_read(F1, sb, 1, (n / 3) * 2 + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // ---- middle of the iteration: slice kt + 1 readable for everyone, slice kt's image dead (every wave has read its block (kt, 1) fragments - the
        // s_waitcnt lgkmcnt(0) keeps them out of the next slice's DMA writes)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- block (kt, 1) on F1; DMA of slice kt + 2 (into slice kt's buffer) and the reads of block (kt + 1, 0) into F0 in the gaps
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int n = 0; n < 18; ++n) {
            mfma_n(F1, n);
            if (n % 3 == 2 && n < 15) {
                __builtin_amdgcn_sched_barrier(0);
                const int sl = n / 3;                      // 0 .. 4
                if (more2) { dma_part(kt + 2, sl == 0 ? 0 : sl + 1); if (sl == 0) dma_part(kt + 2, 1); }
                if (more) { frag_read(F0, sn, 0, sl * 2); frag_read(F0, sn, 0, sl * 2 + 1); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- epilogue: 32 x 32 blocks through a private fp32 image [32][36] (8 waves x 4.5 KB); LN: + row statistics [128 rows][4 column waves] behind the images
    constexpr int EP = 36;
    float* st = reinterpret_cast<float*>(lds) + wave * (32 * EP);
    float* rstat = reinterpret_cast<float*>(lds) + 8 * 32 * EP;            // [128][4]
    const int c8 = (lane & 3) * 8, rl0 = lane >> 2;
    const bool planes = d.c_lo != nullptr;

    // value of (row block ib, column block j, pass) in row-chunk layout: acc -> image -> + bias -> GELU / mul / residual (the x3_epi8 arithmetic without its stores)
    auto load_block = [&](int ib, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[crow32(r, hi) * EP + l31] = acc[ib][j][r];
    };
    if constexpr (!LN) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = tn * 384 + wc * 96 + j * 32 + c8;
            float bias[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) bias[k] = (d.bias && col + k < d.N) ? d.bias[col + k] : 0.f;
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                load_block(ib, j);
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int rl = pass * 16 + rl0;
                    const int row = tm * 128 + wr * 64 + ib * 32 + rl;
                    const float4 v0 = *reinterpret_cast<const float4*>(st + rl * EP + c8);
                    const float4 v1 = *reinterpret_cast<const float4*>(st + rl * EP + c8 + 4);
                    if (row >= d.M || col >= d.N) continue;
                    float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3], v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
                    if (planes) x3_epi8<true>(d, row, col, v); else x3_epi8<false>(d, row, col, v);
                }
            }
        }
    } else {
        // LN needs the full 384-wide row (host checks N == 384, fp32 C, no GELU / mul).  v = product + bias + residual.
        // pass 0: store C, row sums -> mean; pass 1: sum of squared deviations -> rstd; pass 2: (v - mean) * rstd * gamma + beta as planes.
        float mean_[2][2], rstd_[2][2];                // this lane's rows: [ib][pass]
#pragma unroll
        for (int phase = 0; phase < 3; ++phase) {
            float part[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int col = wc * 96 + j * 32 + c8;
                float bias[8], gam[8], bet[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) bias[k] = d.bias ? d.bias[col + k] : 0.f;
                if (phase == 2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { gam[k] = d.ln_gamma[col + k]; bet[k] = d.ln_beta[col + k]; }
                }
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    load_block(ib, j);
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const int rl = pass * 16 + rl0;
                        const int row = tm * 128 + wr * 64 + ib * 32 + rl;
                        const float4 v0 = *reinterpret_cast<const float4*>(st + rl * EP + c8);
                        const float4 v1 = *reinterpret_cast<const float4*>(st + rl * EP + c8 + 4);
                        if (row >= d.M) continue;
                        float v[8] = {v0.x + bias[0], v0.y + bias[1], v0.z + bias[2], v0.w + bias[3], v1.x + bias[4], v1.y + bias[5], v1.z + bias[6], v1.w + bias[7]};
                        if (d.residual) {
                            const float* r = d.residual + (int64_t)row * d.ldr + col;
                            const float4 r0 = *reinterpret_cast<const float4*>(r), r1 = *reinterpret_cast<const float4*>(r + 4);
                            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                        }
                        if (phase == 0) {
                            float* c = reinterpret_cast<float*>(d.c) + (int64_t)row * d.ldc + col;
                            *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                            part[ib][pass] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                        } else if (phase == 1) {
                            float s = 0.f;
#pragma unroll
                            for (int k = 0; k < 8; ++k) { const float dv = v[k] - mean_[ib][pass]; s = fmaf(dv, dv, s); }
                            part[ib][pass] += s;
                        } else {
                            float y[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) y[k] = fmaf((v[k] - mean_[ib][pass]) * rstd_[ib][pass], gam[k], bet[k]);
                            uint4 h, l;
                            x3_split8(y, h, l);
                            const int64_t lo_ = (int64_t)row * d.ldln + col;
                            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.ln_hi) + lo_) = h;
                            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(d.ln_lo) + lo_) = l;
                        }
                    }
                }
            }
            if (phase < 2) {
                // fold the 4 lanes of a row (lane & 3), then the 4 column waves through LDS
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        float s = part[ib][pass];
                        s += __shfl_xor(s, 1, 64);
                        s += __shfl_xor(s, 2, 64);
                        if ((lane & 3) == 0) rstat[(wr * 64 + ib * 32 + pass * 16 + rl0) * 4 + wc] = s;
                    }
                __syncthreads();
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const float4 q = *reinterpret_cast<const float4*>(rstat + (wr * 64 + ib * 32 + pass * 16 + rl0) * 4);
                        const float tot = (q.x + q.y) + (q.z + q.w);
                        if (phase == 0) mean_[ib][pass] = tot * (1.0f / 384.0f);
                        else rstd_[ib][pass] = rsqrtf(tot * (1.0f / 384.0f) + d.ln_eps);
                    }
                __syncthreads();                       // rstat is rewritten by the next phase
            }
        }
        if (wc == 0 && (lane & 3) == 0) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int row = tm * 128 + wr * 64 + ib * 32 + pass * 16 + rl0;
                    if (row < d.M) {
                        if (d.ln_mean) d.ln_mean[row] = mean_[ib][pass];
                        if (d.ln_rstd) d.ln_rstd[row] = rstd_[ib][pass];
                    }
                }
        }
    }
}

// ---- fp32 <-> planes ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ src, int ld_src, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int ld_dst, int64_t rows, int c8n) {
    const int64_t total = rows * c8n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c8n;
        const int c = (int)(i - r * c8n) * 8;
        const float* s = src + r * ld_src + c;
        const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint4 h, l;
        x3_split8(v, h, l);
        *reinterpret_cast<uint4*>(hi + r * ld_dst + c) = h;
        *reinterpret_cast<uint4*>(lo + r * ld_dst + c) = l;
    }
}
__global__ __launch_bounds__(256) void from_planes_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int ld_src, float* __restrict__ dst, int ld_dst, int64_t rows, int c8n) {
    const int64_t total = rows * c8n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c8n;
        const int c = (int)(i - r * c8n) * 8;
        const uint4 h = *reinterpret_cast<const uint4*>(hi + r * ld_src + c), l = *reinterpret_cast<const uint4*>(lo + r * ld_src + c);
        const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(hw[k] << 16) + __uint_as_float(lw[k] << 16);
            v[2 * k + 1] = __uint_as_float(hw[k] & 0xffff0000u) + __uint_as_float(lw[k] & 0xffff0000u);
        }
        float* o = dst + r * ld_dst + c;
        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

}  // namespace

// gemm_x3_as.hip: the A-stationary persistent kernel for K = 256 / 384
bool p3_gemm_x3_as_ok(const p3_gemm_x3_desc* d);
bool p3_gemm_x3_as_default(const p3_gemm_x3_desc* d);
int p3_gemm_x3_as(const p3_gemm_x3_desc* d, hipStream_t s);

static int g_x3_tile = 0;
extern "C" int p3_gemm_x3_tile(int mode) { const int was = g_x3_tile; g_x3_tile = mode; return was; }
// the 128 x 384 tile (one workgroup per CU, 147 MFMA-flop per staged byte) pays where its quantisation over the CUs is not worse than the small tile's and
// the epilogue is short against the main loop; see the table in DESIGN.md section 5
static bool x3_big_tile(const p3_gemm_x3_desc* d) {
    if (d->N <= 256) return false;
    return d->K >= 1024;
}

extern "C" int p3_gemm_x3(const p3_gemm_x3_desc* d, void* stream) {
    P3_CHECK(d && d->a_hi && d->a_lo && d->w_hi && d->w_lo && d->c, P3_EINVAL, "p3_gemm_x3: null pointer");
    P3_CHECK(d->M > 0 && d->N > 0 && d->K > 0, P3_ESHAPE, "p3_gemm_x3: empty problem");
    P3_CHECK(d->N % 8 == 0, P3_ESHAPE, "p3_gemm_x3: N % 8 == 0");
    P3_CHECK(d->lda % 8 == 0 && d->ldb % 8 == 0, P3_EALIGN, "p3_gemm_x3: lda / ldb must keep 16-byte rows");
    const bool planes = d->c_lo != nullptr;
    P3_CHECK(d->ldc % (planes ? 8 : 4) == 0, P3_EALIGN, "p3_gemm_x3: ldc must keep 16-byte rows");
    P3_CHECK(((uintptr_t)d->a_hi | (uintptr_t)d->a_lo | (uintptr_t)d->w_hi | (uintptr_t)d->w_lo | (uintptr_t)d->c | (uintptr_t)d->c_lo) % 16 == 0, P3_EALIGN,
             "p3_gemm_x3: 16-byte base alignment");
    P3_CHECK(!d->residual || (d->ldr % 4 == 0 && (uintptr_t)d->residual % 16 == 0), P3_EALIGN, "p3_gemm_x3: residual alignment");
    P3_CHECK(!d->aux || (d->ldaux % 4 == 0 && (uintptr_t)d->aux % 16 == 0), P3_EALIGN, "p3_gemm_x3: aux alignment");
    P3_CHECK(!d->mul || (d->ldmul % 4 == 0 && (uintptr_t)d->mul % 16 == 0), P3_EALIGN, "p3_gemm_x3: mul alignment");
    P3_CHECK(d->act == P3_ACT_NONE || d->act == P3_ACT_GELU, P3_EUNSUP, "p3_gemm_x3: act NONE or GELU");
    P3_CHECK((int64_t)d->M * d->lda * 2 < (1ll << 31) && (int64_t)d->N * d->ldb * 2 < (1ll << 31), P3_EUNSUP, "p3_gemm_x3: operand larger than the 32-bit DMA offsets");
    const bool ln = d->ln_gamma != nullptr;
    if (ln) {
        P3_CHECK(d->N == 384 && !planes && d->act == P3_ACT_NONE && !d->mul, P3_EUNSUP, "p3_gemm_x3: fused LayerNorm needs N == 384, fp32 C, no activation");
        P3_CHECK(d->ln_beta && d->ln_hi && d->ln_lo && d->ldln % 8 == 0 && ((uintptr_t)d->ln_hi | (uintptr_t)d->ln_lo) % 16 == 0, P3_EINVAL, "p3_gemm_x3: ln_* arguments");
    }
    X3Args g;
    g.d = *d;
    g.tiles_m = p3_ceil_div(d->M, 128);
    hipStream_t s = (hipStream_t)stream;
    P3_CHECK(d->K % 32 == 0, P3_ESHAPE, "p3_gemm_x3: K % 32 == 0");
    // K = 256 / 384 with a 32-column-block output: the A-stationary persistent kernel (p3_gemm_x3_tile(3) asks for it explicitly, (1) / (2) keep the tile kernels)
    // default rule (tools/mb_as.py, profiles/r06_mb_as.txt): the wide outputs - qkv 1152, fc1 / dX of fc2 1536, the decoder's linear1 2048 - where a
    // workgroup walks >= 28 units per A slice; the 384- / 256- / 768-wide ones stay on the tile kernels (their 9 - 12 units per workgroup do not pay the slice load)
    static int as_on = -1;                            // P3_X3_AS=0: same-box A/B of the step without the A-stationary kernel (bench.py --lean)
    if (as_on < 0) { const char* e = getenv("P3_X3_AS"); as_on = (e && atoi(e) == 0) ? 0 : 1; }
    // the kernel of THIS call: p3_gemm_x3_desc.tile (1 / 2 / 3; the host's ragged-round split uses it) before the process-wide measurement hook p3_gemm_x3_tile before
    // the measured rule
    P3_CHECK(d->tile >= 0 && d->tile <= 3, P3_EINVAL, "p3_gemm_x3: tile must be 0 (the library's choice), 1 (128 x 128), 2 (128 x 384) or 3 (A-stationary)");
    const int tile = d->tile ? d->tile : g_x3_tile;
    if (((tile == 0 && as_on && p3_gemm_x3_as_default(d)) || tile == 3) && p3_gemm_x3_as_ok(d)) return p3_gemm_x3_as(d, s);
    P3_CHECK(tile != 3, P3_EUNSUP, "p3_gemm_x3: the A-stationary kernel needs K = 256 / 384, N % 32 == 0, N <= 4096, an epilogue it builds, no fused LayerNorm");
    // tile choice: 1 forces the 128 x 128 kernel, 2 the 128 x 384 kernel, 0 the measured rule (tools/mb_x3.py, profiles/r05_mb_x3.txt)
    const bool big = tile == 2 ? d->N > 128 : (tile == 1 ? false : x3_big_tile(d));
    if (ln || big) {
        g.tiles_n = p3_ceil_div(d->N, 384);
        constexpr size_t LDS = 2 * (2 * 128 * 4 + 2 * 384 * 4) * 16;          // 2 slices x 64 KB
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_x3_n384_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_x3_n384_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
        if (p3_tracing()) p3_note_kernel(ln ? "gemm_x3_n384_kernel<true>" : "gemm_x3_n384_kernel<false>");
        if (ln) hipLaunchKernelGGL(gemm_x3_n384_kernel<true>, dim3(g.tiles_m * g.tiles_n), dim3(512), LDS, s, g);
        else hipLaunchKernelGGL(gemm_x3_n384_kernel<false>, dim3(g.tiles_m * g.tiles_n), dim3(512), LDS, s, g);
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    P3_CHECK(!ln, P3_EUNSUP, "p3_gemm_x3: fused LayerNorm needs N == 384");
    g.tiles_n = p3_ceil_div(d->N, 128);
    if (p3_tracing()) p3_note_kernel(planes ? "gemm_x3_kernel<true>" : "gemm_x3_kernel<false>");
    dim3 grid(g.tiles_m * g.tiles_n), block(256);
    if (planes) hipLaunchKernelGGL(gemm_x3_kernel<true>, grid, block, 0, s, g);
    else hipLaunchKernelGGL(gemm_x3_kernel<false>, grid, block, 0, s, g);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_to_planes(const float* src, int ld_src, void* hi, void* lo, int ld_dst, int64_t rows, int cols, void* stream) {
    P3_CHECK(src && hi && lo && rows >= 0 && cols > 0, P3_EINVAL, "p3_to_planes: bad arguments");
    P3_CHECK(cols % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src | (uintptr_t)hi | (uintptr_t)lo) % 16 == 0, P3_EALIGN, "p3_to_planes: cols % 8, 16-byte rows");
    if (rows == 0) return P3_OK;
    const int64_t total = rows * (cols / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(to_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, ld_src, (bf16_t*)hi, (bf16_t*)lo, ld_dst, rows, cols / 8);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_from_planes(const void* hi, const void* lo, int ld_src, float* dst, int ld_dst, int64_t rows, int cols, void* stream) {
    P3_CHECK(dst && hi && lo && rows >= 0 && cols > 0, P3_EINVAL, "p3_from_planes: bad arguments");
    P3_CHECK(cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 4 == 0 && ((uintptr_t)dst | (uintptr_t)hi | (uintptr_t)lo) % 16 == 0, P3_EALIGN, "p3_from_planes: cols % 8, 16-byte rows");
    if (rows == 0) return P3_OK;
    const int64_t total = rows * (cols / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(from_planes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)hi, (const bf16_t*)lo, ld_src, dst, ld_dst, rows, cols / 8);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
