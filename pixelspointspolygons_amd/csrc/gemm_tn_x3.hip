// p3hip weight-gradient GEMM on PLANES (include/p3hip.h, p3_gemm_tn_x3):  C[N,K] += (a_hi + a_lo)[M,N]^T (b_hi + b_lo)[M,K], three bf16 MFMAs per product
//
// gemm_tn_dma.hip's kernel with FOUR operand images per step instead of two: both operands arrive split (the producers wrote hi = bf16(x), lo = bf16(x - hi)),
// every image goes global -> LDS by LDS-DMA as it lies in memory ([64 rows of m][128 columns] bf16, 256-byte rows, 64-byte granules XOR-swizzled with (row & 3)
// on the source address), a wave reads the four fragment sets with ds_read_b64_tr_b16 and accumulates a_lo b_hi + a_hi b_lo + a_hi b_hi: the P3_F32X3 arithmetic
// (gemm_tn.hip SPLIT: fp32 operands split while they are staged through registers, 248 - 250 us per fc1 / fc2 weight gradient) at the bf16 kernel's memory path.
// 512 threads = two groups of four waves on the SAME output tile (group g multiplies rows 16 g .. 16 g + 15 of every 32-row step, the groups fold through LDS
// at the end); FOUR steps of 32 KB in LDS, three in flight behind a counted vmcnt (r05 PMC of the two-step 64-row form: waves parked 37 % of their cycles
// in s_waitcnt / s_barrier, the matrix pipe 40 % busy); one workgroup per CU, all (n, k) tiles of one M split on one XCD.
#include <stdlib.h>

#include <type_traits>

#include "p3_common.h"

namespace {

constexpr int TD_BM = 32;                       // rows of m per step
constexpr int TD_IMG = TD_BM * 256;              // one operand image of a step
constexpr int TD_STEP_BYTES = 4 * TD_IMG;      // a_hi | a_lo | b_hi | b_lo

struct TdArgs {
    const bf16_t* A; const bf16_t* Al; const bf16_t* B; const bf16_t* Bl; float* C;
    int M, N, K, lda, ldb, ldc, rows_per_split, tiles_k, splits;
    float* slabs;       // optional [splits][N][K]: partial tiles stored instead of atomics (deterministic mode)
    float* colsum;      // optional [N]: += column sums of A (bias gradient) from the tk == 0 tiles
    float* cs_slab;     // deterministic mode: [splits][N]
};

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int N> __device__ __forceinline__ void td_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

template <int NBUF>
__global__ __launch_bounds__(512, 2) void gemm_tn_x3_kernel(TdArgs g) {
    constexpr int LA = NBUF - 1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1, l31 = lane & 31, hi = lane >> 5;
    // all (n, k) tiles of one M split run on ONE XCD (they read the same rows of A and B)
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_all = gridDim.x / g.splits;
    const int tile = lid % tiles_all, split = lid / tiles_all;
    const int tn = tile / g.tiles_k, tk = tile - tn * g.tiles_k;
    const int m_beg = split * g.rows_per_split;
    const int m_end = min(g.M, m_beg + g.rows_per_split);
    const int nsteps = (m_end - m_beg) / TD_BM;               // rows_per_split and M are multiples of 64

    // ---- LDS-DMA: a step = 4 images x 8 pieces of 1 KB (4 rows x 256 B); wave w issues the four pieces (w & 1) * 4 .. + 3 of image w >> 1 (a_hi, a_lo, b_hi, b_lo).
    // lane -> (row = lane / 16, slot = lane % 16) of a piece, source chunk = slot ^ ((row & 3) << 2): 64-byte granule swizzle
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_raw);
    const int img = wave >> 1;
    const bf16_t* src = img == 0 ? g.A : (img == 1 ? g.Al : (img == 2 ? g.B : g.Bl));
    const int sld = img < 2 ? g.lda : g.ldb, scol = img < 2 ? tn * 128 : tk * 128;
    uint32_t voff[4];
    {
        const int prow = lane >> 4, slot = lane & 15, chunk = slot ^ (prow << 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = ((wave & 1) * 4 + q) * 4 + prow;          // row inside the step
            voff[q] = (uint32_t)(((int64_t)r * sld + scol + chunk * 8) * 2);
        }
    }
    auto dma2 = [&](const bf16_t* base, uint32_t dst, uint32_t v0, uint32_t v1) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(dst) : "memory");
    };
    auto stage = [&](int st) __attribute__((always_inline)) {      // step st -> buffer st % NBUF (caller: st < nsteps)
        const int buf = st % NBUF;
        const int64_t m0 = (int64_t)m_beg + (int64_t)st * TD_BM;
        const uint32_t da = lds_addr + (uint32_t)(buf * TD_STEP_BYTES + wave * 4096);
        dma2(src + m0 * sld, da, voff[0], voff[1]);
        dma2(src + m0 * sld, da + 2048, voff[2], voff[3]);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- transposing fragment reads: lane -> (row block g4 >> 1, row li >> 2, 32-byte half g4 & 1, 8-byte piece li & 3); 32-column block i of
    // the wave's 64 columns = 64-byte granule (w * 2 + i), swizzled with the row's low bits (rows advance by multiples of 4 between reads)
    const int g4 = lane >> 4, li = lane & 15;
    const uint32_t lrow = (uint32_t)((g4 >> 1) * 8 + (li >> 2));
    const uint32_t lin = (uint32_t)((g4 & 1) * 32 + (li & 3) * 8);
    uint32_t offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        offA[i] = lds_addr + (uint32_t)((grp * 16 + lrow) * 256 + (((wm * 2 + i) ^ (li >> 2)) * 64) + lin);
        offB[i] = lds_addr + (uint32_t)(2 * TD_IMG + (grp * 16 + lrow) * 256 + (((wn * 2 + i) ^ (li >> 2)) * 64) + lin);
    }
    // bias gradient: column sums of A, by the tk == 0 tiles, from the LDS image (thread -> chunk tid % 16 of rows tid / 16 and tid / 16 + 32)
    const bool do_cs = g.colsum != nullptr && tk == 0;
    float csum[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) csum[q] = 0.f;
    const int cs_row = tid >> 4, cs_chunk = tid & 15;

#pragma unroll
    for (int p = 0; p < LA; ++p)
        if (p < nsteps) stage(p);
    for (int st = 0; st < nsteps; ++st) {
        // RAW: this wave's pieces of step st have landed once at most `ahead` younger steps stay in flight (loads retire in order; nothing else is outstanding);
        // the barrier extends that to every wave's pieces.  WAR: a wave reaches the barrier after its reads of step st - 1, whose buffer step st + LA takes.
        const int ahead = min(LA - 1, nsteps - 1 - st);
        if (ahead >= 2) td_wait_vm<8>();
        else if (ahead == 1) td_wait_vm<4>();
        else td_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + LA < nsteps) stage(st + LA);
        const uint32_t bo = (uint32_t)((st % NBUF) * TD_STEP_BYTES);
        if (do_cs) {
            const unsigned char* ab = lds_raw + bo;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const int r = cs_row;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ab + im * TD_IMG + r * 256 + ((cs_chunk ^ ((r & 3) << 2)) * 16));
#pragma unroll
                for (int q = 0; q < 4; ++q) { csum[2 * q] += __uint_as_float(v[q] << 16); csum[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u); }
            }
        }
        {
            constexpr int kk = 0;                                          // a group's 16 rows of the step = one 16-deep MFMA block
            u32x2_t fah[2][2], fal[2][2], fbh[2][2], fbl[2][2];            // [32-column block][rows +0..3 | +4..7]
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t ro = bo + (uint32_t)((kk * 16 + hh * 4) * 256);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fah[i][hh]) : "v"(offA[i] + ro));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fal[i][hh]) : "v"(offA[i] + ro + TD_IMG));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fbh[i][hh]) : "v"(offB[i] + ro));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fbl[i][hh]) : "v"(offB[i] + ro + TD_IMG));
                }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fah[0][0]), "+v"(fah[0][1]), "+v"(fah[1][0]), "+v"(fah[1][1]), "+v"(fal[0][0]), "+v"(fal[0][1]), "+v"(fal[1][0]), "+v"(fal[1][1]),
                           "+v"(fbh[0][0]), "+v"(fbh[0][1]), "+v"(fbh[1][0]), "+v"(fbh[1][1]), "+v"(fbl[0][0]), "+v"(fbl[0][1]), "+v"(fbl[1][0]), "+v"(fbl[1][1]));
            __builtin_amdgcn_sched_barrier(0);
            bf16x8_t ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fah[i][0].x, fah[i][0].y, fah[i][1].x, fah[i][1].y});
                al[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fal[i][0].x, fal[i][0].y, fal[i][1].x, fal[i][1].y});
                bh[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fbh[i][0].x, fbh[i][0].y, fbh[i][1].x, fbh[i][1].y});
                bl[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fbl[i][0].x, fbl[i][0].y, fbl[i][1].x, fbl[i][1].y});
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // every wave is done with the operand images: the epilogue reuses the LDS

    // ---- bias gradient: fold the 32 row lanes of a chunk through LDS, one value per column
    float* red = reinterpret_cast<float*>(lds_raw);
    if (do_cs) {                                     // wave-uniform (tk is)
#pragma unroll
        for (int q = 0; q < 8; ++q) red[cs_row * 128 + cs_chunk * 8 + q] = csum[q];
        __syncthreads();
        if (tid < 128) {
            float a = 0.f;
            for (int r = 0; r < 32; ++r) a += red[r * 128 + tid];
            if (g.cs_slab) g.cs_slab[(int64_t)split * g.N + tn * 128 + tid] = a;
            else atomicAdd(g.colsum + tn * 128 + tid, a);
        }
        __syncthreads();
    }
    // ---- fold the two groups: group 0 keeps the row block i = 0 of its waves' tiles, group 1 the row block i = 1; each hands the other block
    // over through LDS ([wave pair][j][r][lane] fp32: conflict-free ds_write_b32 / ds_read_b32)
    auto fold = [&](auto KEEP) __attribute__((always_inline)) {       // static register indices (a run-time index would put acc into scratch)
        constexpr int keep = decltype(KEEP)::value, give = 1 - keep;
        float* x = red + (grp * 4 + w4) * (2 * 16 * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[(j * 16 + r) * 64 + lane] = acc[give][j][r];
        __syncthreads();
        const float* y = red + ((grp ^ 1) * 4 + w4) * (2 * 16 * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tk * 128 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[keep][j][r] + y[(j * 16 + r) * 64 + lane];
                const int row = tn * 128 + wm * 64 + keep * 32 + crow32(r, hi);
                if (g.slabs) g.slabs[((int64_t)split * g.N + row) * g.K + col] = v;
                else atomicAdd(g.C + (int64_t)row * g.ldc + col, v);
            }
        }
    };
    if (grp == 0) fold(std::integral_constant<int, 0>{});          // group g finishes row block i = g of its waves' tiles
    else fold(std::integral_constant<int, 1>{});
}

// ---- the WIDE tile: 128 (n) x 384 (k) per workgroup, every wave on all rows ---------------------------------------------------------------------------------
// What bounds the 128 x 128 kernel above is the CU's address path: a 32-row step is 32 LDS-DMA pieces (1 KB each, ~40 cycles of that path per piece) for 96 MFMAs
// - 1280 cycles of staging against 768 of matrix pipe per SIMD (r05 PMC: pipe 39 % busy).  Here the two wave groups no longer split the ROWS of one tile (which
// staged every byte for half the MFMAs): the 8 waves tile a 128 x 384 output as 2 x 4 sub-tiles of 64 x 96, a step is 16 rows = one 16-deep MFMA block of
// EIGHT images (a_hi, a_lo, three 128-column blocks of b_hi and of b_lo) = 32 pieces for 8 x 18 MFMAs: 1280 cycles of staging against 1152 of pipe.  No fold
// between groups at the end.  Shapes: N % 128 == 0, K % 384 == 0 - fc1 (1536 x 384), fc2 (384 x 1536), qkv (1152 x 384) of the timm Block.
constexpr int TW_BM = 16;                       // rows of m per step
constexpr int TW_IMG = TW_BM * 256;             // one image of a step: 16 rows x 128 columns
constexpr int TW_STEP_BYTES = 8 * TW_IMG;       // 32 KB

template <int NBUF>
__global__ __launch_bounds__(512, 2) void gemm_tn_x3_wide_kernel(TdArgs g) {
    constexpr int LA = NBUF - 1;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, hi = lane >> 5;          // sub-tile: rows wm * 64 .. + 63 of n, columns wn * 96 .. + 95 of k
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_all = gridDim.x / g.splits;
    const int tile = lid % tiles_all, split = lid / tiles_all;
    const int tn = tile / g.tiles_k, tk = tile - tn * g.tiles_k;
    const int m_beg = split * g.rows_per_split;
    const int m_end = min(g.M, m_beg + g.rows_per_split);
    const int nsteps = (m_end - m_beg) / TW_BM;

    // ---- LDS-DMA: wave w stages image w of every step (0: a_hi, 1: a_lo, 2 + 2 c: b_hi block c, 3 + 2 c: b_lo block c), four pieces of 4 rows x 256 B;
    // lane -> (row = lane / 16, slot = lane % 16) of a piece, source chunk = slot ^ ((row & 3) << 2)
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_raw);
    const bf16_t* src = wave == 0 ? g.A : (wave == 1 ? g.Al : ((wave & 1) ? g.Bl : g.B));
    const int sld = wave < 2 ? g.lda : g.ldb, scol = wave < 2 ? tn * 128 : tk * 384 + ((wave - 2) >> 1) * 128;
    uint32_t voff[4];
    {
        const int prow = lane >> 4, slot = lane & 15, chunk = slot ^ (prow << 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) voff[q] = (uint32_t)(((int64_t)(q * 4 + prow) * sld + scol + chunk * 8) * 2);
    }
    auto dma2 = [&](const bf16_t* base, uint32_t dst, uint32_t v0, uint32_t v1) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
            "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(v0), "v"(v1), "s"(base), "s"(dst) : "memory");
    };
    auto stage = [&](int st) __attribute__((always_inline)) {
        const int64_t m0 = (int64_t)m_beg + (int64_t)st * TW_BM;
        const uint32_t da = lds_addr + (uint32_t)((st % NBUF) * TW_STEP_BYTES + wave * TW_IMG);
        dma2(src + m0 * sld, da, voff[0], voff[1]);
        dma2(src + m0 * sld, da + 2048, voff[2], voff[3]);
    };

    f32x16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- transposing fragment reads (as above): lane -> (row (g4 >> 1) * 8 + (li >> 2), 32-byte half g4 & 1, 8-byte piece li & 3) of a 16-row image;
    // the wave's A blocks: granules wm * 2 + i of image 0 / 1; its B blocks: column wn * 96 + j * 32 -> image 2 + 2 (c / 128), granule (c % 128) / 32
    const int g4 = lane >> 4, li = lane & 15;
    const uint32_t lrow = (uint32_t)((g4 >> 1) * 8 + (li >> 2));
    const uint32_t lin = (uint32_t)((g4 & 1) * 32 + (li & 3) * 8);
    uint32_t offA[2], offB[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) offA[i] = lds_addr + (uint32_t)(lrow * 256 + (((wm * 2 + i) ^ (li >> 2)) * 64) + lin);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = wn * 96 + j * 32;
        offB[j] = lds_addr + (uint32_t)((2 + 2 * (c >> 7)) * TW_IMG + lrow * 256 + ((((c & 127) >> 5) ^ (li >> 2)) * 64) + lin);
    }
    // bias gradient: column sums of A by the tk == 0 tiles, from the LDS images (thread -> chunk tid % 16 of row tid / 16: threads 0..255 cover the 16 rows)
    const bool do_cs = g.colsum != nullptr && tk == 0;
    float csum[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) csum[q] = 0.f;
    const int cs_row = (tid >> 4) & 15, cs_chunk = tid & 15;

#pragma unroll
    for (int p = 0; p < LA; ++p)
        if (p < nsteps) stage(p);
    for (int st = 0; st < nsteps; ++st) {
        const int ahead = min(LA - 1, nsteps - 1 - st);
        if (ahead >= 2) td_wait_vm<8>();
        else if (ahead == 1) td_wait_vm<4>();
        else td_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + LA < nsteps) stage(st + LA);
        const uint32_t bo = (uint32_t)((st % NBUF) * TW_STEP_BYTES);
        if (do_cs && tid < 256) {
            const unsigned char* ab = lds_raw + bo;
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ab + im * TW_IMG + cs_row * 256 + ((cs_chunk ^ ((cs_row & 3) << 2)) * 16));
#pragma unroll
                for (int q = 0; q < 4; ++q) { csum[2 * q] += __uint_as_float(v[q] << 16); csum[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u); }
            }
        }
        u32x2_t fah[2][2], fal[2][2], fbh[3][2], fbl[3][2];            // [32-column block][rows +0..3 | +4..7]
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const uint32_t ro = bo + (uint32_t)(hh * 4 * 256);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fah[i][hh]) : "v"(offA[i] + ro));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fal[i][hh]) : "v"(offA[i] + ro + TW_IMG));
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fbh[j][hh]) : "v"(offB[j] + ro));
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fbl[j][hh]) : "v"(offB[j] + ro + TW_IMG));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fah[0][0]), "+v"(fah[0][1]), "+v"(fah[1][0]), "+v"(fah[1][1]), "+v"(fal[0][0]), "+v"(fal[0][1]), "+v"(fal[1][0]), "+v"(fal[1][1]),
                       "+v"(fbh[0][0]), "+v"(fbh[0][1]), "+v"(fbh[1][0]), "+v"(fbh[1][1]), "+v"(fbh[2][0]), "+v"(fbh[2][1]),
                       "+v"(fbl[0][0]), "+v"(fbl[0][1]), "+v"(fbl[1][0]), "+v"(fbl[1][1]), "+v"(fbl[2][0]), "+v"(fbl[2][1]));
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t ah[2], al[2], bh[3], bl[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fah[i][0].x, fah[i][0].y, fah[i][1].x, fah[i][1].y});
            al[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fal[i][0].x, fal[i][0].y, fal[i][1].x, fal[i][1].y});
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bh[j] = __builtin_bit_cast(bf16x8_t, u32x4_t{fbh[j][0].x, fbh[j][0].y, fbh[j][1].x, fbh[j][1].y});
            bl[j] = __builtin_bit_cast(bf16x8_t, u32x4_t{fbl[j][0].x, fbl[j][0].y, fbl[j][1].x, fbl[j][1].y});
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // every wave is done with the operand images: the bias fold reuses the LDS

    if (do_cs) {                                     // wave-uniform (tk is)
        float* red = reinterpret_cast<float*>(lds_raw);
        if (tid < 256) {
#pragma unroll
            for (int q = 0; q < 8; ++q) red[cs_row * 128 + cs_chunk * 8 + q] = csum[q];
        }
        __syncthreads();
        if (tid < 128) {
            float a = 0.f;
            for (int r = 0; r < 16; ++r) a += red[r * 128 + tid];
            if (g.cs_slab) g.cs_slab[(int64_t)split * g.N + tn * 128 + tid] = a;
            else atomicAdd(g.colsum + tn * 128 + tid, a);
        }
    }
    // ---- the wave's 64 x 96 sub-tile: partial tile into the split's slab, or fp32 atomics
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int col = tk * 384 + wn * 96 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tn * 128 + wm * 64 + i * 32 + crow32(r, hi);
                if (g.slabs) g.slabs[((int64_t)split * g.N + row) * g.K + col] = acc[i][j][r];
                else atomicAdd(g.C + (int64_t)row * g.ldc + col, acc[i][j][r]);
            }
        }
}

}  // namespace

// C[n,k] += sum_s slabs[s][n][k] (float64, split order) - gemm_tn.hip
void p3_tn_reduce_launch(const float* slabs, float* C, int N, int K, int ldc, int splits, hipStream_t s);
float* p3_tn_park(float* C, int N, int K, int ldc, int splits);      // gemm_tn.hip: deferred reduce slot or NULL

static int g_tn_wide = 1;
extern "C" int p3_gemm_tn_x3_wide(int on) { const int was = g_tn_wide; g_tn_wide = on; return was; }

extern "C" int p3_gemm_tn_x3(const void* a_hi, const void* a_lo, int lda, const void* b_hi, const void* b_lo, int ldb, float* C, int ldc, int M, int N, int K,
                             float* colsum, float* slabs, int max_slabs, void* stream) {
    P3_CHECK(a_hi && a_lo && b_hi && b_lo && C && M > 0 && N > 0 && K > 0, P3_EINVAL, "p3_gemm_tn_x3: bad arguments");
    P3_CHECK(M % TD_BM == 0 && N % 128 == 0 && K % 128 == 0, P3_ESHAPE, "p3_gemm_tn_x3: M % 32 == 0, N % 128 == 0, K % 128 == 0");
    P3_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)b_hi | (uintptr_t)b_lo) % 16 == 0, P3_EALIGN, "p3_gemm_tn_x3: 16-byte rows");
    P3_CHECK((int64_t)TD_BM * lda * 2 + 256 < (1ll << 31) && (int64_t)TD_BM * ldb * 2 + 256 < (1ll << 31), P3_EUNSUP, "p3_gemm_tn_x3: row stride beyond the 32-bit DMA offsets");
    hipStream_t s = (hipStream_t)stream;
    TdArgs g;
    g.A = (const bf16_t*)a_hi; g.Al = (const bf16_t*)a_lo; g.B = (const bf16_t*)b_hi; g.Bl = (const bf16_t*)b_lo; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.colsum = colsum;
    const int tiles_n = N / 128;
    // the wide tile (128 x 384, every wave on all rows) where K is a multiple of 384 and there are enough tiles for its split count to stay moderate
    // (p3_gemm_tn_x3_wide(0) switches it off: same-box A/B, tools/mb_x3.py)
    static int env_wide = -1;                        // P3_TN_WIDE=0: same-box A/B of the step (bench.py --lean)
    if (env_wide < 0) { const char* e = getenv("P3_TN_WIDE"); env_wide = (e && atoi(e) == 0) ? 0 : 1; }
    const bool wide = g_tn_wide && env_wide && K % 384 == 0 && tiles_n * (K / 384) >= 8;
    g.tiles_k = wide ? K / 384 : K / 128;
    const int step_rows = wide ? TW_BM : TD_BM;
    const int tiles = tiles_n * g.tiles_k;
    P3_CHECK(tiles <= 256, P3_EUNSUP, "p3_gemm_tn_x3: more than 256 output tiles");
    int splits = 256 / tiles;                        // one workgroup per CU, ONE resident round
    if (splits < 1) splits = 1;
    const int max_splits = M / (2 * TD_BM) > 0 ? M / (2 * TD_BM) : 1;
    if (splits > max_splits) splits = max_splits;
    if (slabs && splits > max_slabs) splits = max_slabs;
    g.rows_per_split = p3_ceil_div(p3_ceil_div(M, splits), TD_BM) * TD_BM;
    splits = p3_ceil_div(M, g.rows_per_split);
    g.splits = splits;
    g.slabs = (slabs && splits > 1) ? slabs : nullptr;
    bool parked = false;                             // p3_tn_defer: the partial tiles wait in the caller's arena for p3_tn_flush instead of a reduce launch of their own
    if (g.slabs) { float* slot = p3_tn_park(C, N, K, ldc, splits); if (slot) { g.slabs = slot; parked = true; } }
    (void)step_rows;
    int cs_parked = 0;
    g.cs_slab = colsum ? p3_colsum_parts(splits, N, colsum, P3_F32, &cs_parked) : nullptr;
    constexpr int NBUF = 4;
    const size_t lds = (size_t)NBUF * TD_STEP_BYTES;       // 4 x 32 KB (>= the 64 KB the fold needs); the wide kernel's steps are 32 KB too
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_x3_kernel<NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn_x3_wide_kernel<NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    dim3 grid(tiles * splits), block(512);
    if (wide) hipLaunchKernelGGL(gemm_tn_x3_wide_kernel<NBUF>, grid, block, lds, s, g);
    else hipLaunchKernelGGL(gemm_tn_x3_kernel<NBUF>, grid, block, lds, s, g);
    if (p3_tracing()) p3_note_kernel(wide ? "gemm_tn_x3_wide_kernel<4>" : "gemm_tn_x3_kernel<4>");
    if (g.slabs && !parked) p3_tn_reduce_launch(g.slabs, C, N, K, ldc, splits, s);
    P3_LAUNCH_CHECK();
    if (g.cs_slab && !cs_parked) return p3_det_reduce(g.cs_slab, splits, N, colsum, N, 1, s);
    return P3_OK;
}
