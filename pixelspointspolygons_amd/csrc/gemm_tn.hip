// p3hip weight-gradient GEMM:  C[N,K] += A[M,N]^T * B[M,K]   (reduction over the M rows; fp32 output, split over M)
//
// dW = dY^T X of every Linear / conv on the path.  Both operands are "k-strided" for MFMA (the reduction index is the
// row index), so for bf16 the staging pass transposes while writing to LDS: each thread loads two consecutive rows and
// packs the row pair of every column into one dword ([col][m, m+1]) -> fragments are then plain ds_read_b128 like in
// the NT kernel.  f32 (v_mfma_f32_32x32x2_f32 takes one element per lane) needs no transpose.
// Grid = tiles_n x tiles_k x splits; every split adds its partial tile with fp32 atomics (C must be zero-filled or hold
// the gradient being accumulated).  Rows >= M contribute zeros.
#include <stdlib.h>

#include <type_traits>

#include "p3_common.h"

namespace {

constexpr int TN = 128, TK = 128;

struct TnArgs {
    const void* A; const void* B; float* C;
    int M, N, K, lda, ldb, ldc, rows_per_split, tiles_k, splits;
    // optional generated B operand (ScoreNet backward): B'[m,k] = relu((B[m',k] (+ V[m'',k])) * b_scale[k] + b_shift[k])
    int Kb;                     // columns of B in memory (== K but for P3_A_AFFINE_MASK2: K = 2 Kb output columns)
    int b_mode;                 // 0 plain, P3_A_AFFINE_RELU, P3_A_PAIR_AFFINE_RELU (m = (b,i,j): B row b*n+i, V row b*n+j), P3_A_AFFINE_MASK2
    const float* b_scale; const float* b_shift; const void* pair_V; int pair_n;
    float* slabs;               // optional [splits][N][K] fp32: partial tiles are STORED here (coalesced) and summed by tn_reduce_kernel
                                // instead of splits x N x K fp32 atomics on C (measured: the atomics, not the MFMAs, bounded this kernel)
    float* colsum;              // optional [N]: += column sums of A (bias gradient), accumulated by the tk == 0 tiles from the staged registers
    float* cs_slab;             // deterministic mode: [splits][N] partial column sums (det_reduce.hip) instead of atomics on colsum
};

template <typename T> struct TTr;
// bf16: the LDS image is the operand tile AS IT LIES IN MEMORY, [m][col] with 16-byte stores (no transposition, no packing VALU);
// the k-contiguous MFMA fragments (8 consecutive m for one column) come from gfx950's transposing LDS read ds_read_b64_tr_b16:
// per 16-lane group, lanes 4j..4j+3 point at row j's 16 elements and lane i receives column i of those 4 rows (semantics pinned by
// tools/probe/tr_probe.hip).  Row pitch 160 elements = 80 dwords: the 4 rows of a read land 16 banks apart -> conflict free.
// (Round-1 history: transposing ds_write_b32 stores -> XOR swizzle -> 8-byte stores; this form removes the transposing stores.)
template <> struct TTr<bf16_t> { static constexpr int BM = 64, PITCH = 160, ELEMS = 64 * 160; };
template <> struct TTr<float> { static constexpr int BM = 16, PITCH = 132, ELEMS = 16 * 132; };  // [m][col]

// GENB = false: B is a plain matrix (all weight gradients but two): no mode branches in the step, loads are UNCONDITIONAL on clamped
// addresses and zeroed by selects (a conditional 16-byte load compiles to a branch per load; with the branches in the loop hipcc's
// s_waitcnt pass drains vmcnt at every merge - ISA of r02: 8 branches and several vmcnt(0) per step).  BMODE = P3_A_AFFINE_RELU /
// P3_A_PAIR_AFFINE_RELU: generated B operand - the raw rows (and, pair mode, the V rows) are LOADED in load_step like any operand and
// turned into relu(scale * (x [+ y]) + shift) when they are stored to LDS (the transform at load time consumed every load at once:
// vmcnt(0) inside the step, no prefetch at all; 519 us per ScoreNet weight gradient in r02).
// SPLIT (T = float, p3_set_gemm_split): the fp32 operands are split into bf16 hi / lo images when a step is stored to LDS ([16 rows][160] bf16 each, the bf16
// path's pitch) and multiplied as a_lo b_hi + a_hi b_lo + a_hi b_hi from transposing reads - see gemm.hip BK_SPLIT.
constexpr int TS_PITCH = 160, TS_IMG_B = 16 * TS_PITCH * 2, TS_ELEMS = 2 * TS_IMG_B / 4;
template <typename T, int BMODE, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnArgs g) {
    static_assert(!SPLIT || sizeof(T) == 4, "the split form is the fp32 operands' path");
    constexpr int BM = TTr<T>::BM, PITCH = TTr<T>::PITCH, ELEMS = SPLIT ? TS_ELEMS : TTr<T>::ELEMS;
    constexpr bool BF = sizeof(T) == 2;
    constexpr bool GENB = BMODE != 0, PAIR = BMODE == P3_A_PAIR_AFFINE_RELU, GMASK2 = BMODE == P3_A_AFFINE_MASK2;
    __shared__ __attribute__((aligned(16))) T lds[4 * ELEMS];
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;          // byte offset of the tile buffers inside the LDS aperture (inline-asm reads)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    // 1-D grid, XCD-aware: all (n, k) tiles of one M-split (they read the same rows of A and B) run on ONE XCD, so those rows
    // come from HBM once instead of once per XCD (PMC r01: 395 MB fetched per launch on average, ~3x the operand bytes)
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_all = gridDim.x / g.splits;
    const int tile = lid % tiles_all, split = lid / tiles_all;
    const int tn = tile / g.tiles_k, tk = tile - tn * g.tiles_k;
    const int m_beg = split * g.rows_per_split;
    const int m_end = min(g.M, m_beg + g.rows_per_split);
    const T* A = reinterpret_cast<const T*>(g.A);
    const T* B = reinterpret_cast<const T*>(g.B);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging geometry
    constexpr int VEC = BF ? 8 : 4;                 // elements per 16-byte load
    constexpr int TPR = TN / VEC;                   // threads per row: 16 (bf16) / 32 (f32)
    const int cv = (tid % TPR) * VEC;               // column offset inside the tile
    const int rt = tid / TPR;                       // bf16: row-pair index 0..15 ; f32: row 0..7
    constexpr int NLOAD = BF ? 4 : 2;               // 16-byte loads per operand per step
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // two staging register sets: the loads of step t+2 are issued while step t is multiplied (a global load gets two MFMA phases to land;
    // r02 SQ counters: 46 % of the wave time parked on waits with the one-deep pipeline)
    u32x4 rsa[2][NLOAD], rsb[2][NLOAD];
    u32x4 rsy[PAIR ? 2 : 1][PAIR ? NLOAD : 1];    // pair mode: the V rows
    uint32_t rok[2];                              // generated B: bit i = row i of the set lies inside [m_beg, m_end)
    const int coln = tn * TN + cv, colk = tk * TK + cv;
    const bool okn = coln < g.N, okk = colk < g.K;  // N, K are multiples of VEC (checked on the host)
    // GMASK2: output columns [Kb, 2 Kb) are a second generated operand over the SAME columns of B (block-uniform: Kb is a multiple of VEC)
    const bool second = GMASK2 && colk >= g.Kb;
    const int colb = second ? colk - g.Kb : colk;
    float bsc[GENB ? VEC : 1], bsh[GENB ? VEC : 1];
    if constexpr (GENB) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) { bsc[q] = okk ? g.b_scale[colb + q] : 0.f; bsh[q] = okk ? g.b_shift[colb + q] : 0.f; }
    }

    // vector part of the operand addresses (thread row rt, column chunk); the step / pass part (m0 + 16 i) * ld is wave-uniform
    const int64_t abase = (int64_t)rt * g.lda + (okn ? coln : 0), bbase = (int64_t)rt * g.ldb + (okk ? colb : 0);
    auto load_step = [&](auto SET, auto INRANGE, int m0) __attribute__((always_inline)) {
        constexpr int ss = decltype(SET)::value;
        u32x4 (&ra)[NLOAD] = rsa[ss];
        u32x4 (&rb)[NLOAD] = rsb[ss];
        if constexpr (decltype(INRANGE)::value && !GENB) {
            // every row of the step lies inside the split (all steps but the last one): no clamp, no per-lane 64-bit multiply - the
            // r02 form spent two v_mad_i64_i32 + a v_min per 16-byte load on the clamped row address
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                const int64_t ro = (int64_t)(m0 + (BF ? 16 : 8) * i);
                const u32x4 va = *reinterpret_cast<const u32x4*>(A + abase + ro * g.lda);
                const u32x4 vb = *reinterpret_cast<const u32x4*>(B + bbase + ro * g.ldb);
                ra[i] = okn ? va : u32x4{0, 0, 0, 0};
                rb[i] = okk ? vb : u32x4{0, 0, 0, 0};
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            int row;
            if constexpr (BF) row = m0 + rt + 16 * i;         // thread = (row rt + 16 i, 8 columns at cv): 16-byte row-major stores
            else row = m0 + rt + 8 * i;
            const bool okr = row < m_end;
            const int rowc = okr ? row : m_end - 1;            // m_end > m_beg whenever a step runs
            const u32x4 va = *reinterpret_cast<const u32x4*>(A + (int64_t)rowc * g.lda + (okn ? coln : 0));
            ra[i] = (okr && okn) ? va : u32x4{0, 0, 0, 0};
            if constexpr (!GENB) {
                const u32x4 vb = *reinterpret_cast<const u32x4*>(B + (int64_t)rowc * g.ldb + (okk ? colk : 0));
                rb[i] = (okr && okk) ? vb : u32x4{0, 0, 0, 0};
            } else {
                int64_t r1 = rowc, r2 = 0;
                if constexpr (PAIR) {
                    const int n = g.pair_n, nn = n * n;
                    const int bb = rowc / nn, p = rowc - bb * nn, ii = p / n, jj = p - ii * n;
                    r1 = (int64_t)bb * n + ii; r2 = (int64_t)bb * n + jj;
                }
                rb[i] = *reinterpret_cast<const u32x4*>(B + r1 * g.ldb + (okk ? colb : 0));
                if constexpr (PAIR) rsy[ss][i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(g.pair_V) + r2 * g.ldb + (okk ? colb : 0));
                if (i == 0) rok[ss] = 0;
                rok[ss] |= (uint32_t)(okr && okk) << i;
            }
        }
    };
    float csum[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) csum[q] = 0.f;
    const bool do_cs = g.colsum != nullptr && tk == 0;
    auto store_step = [&](auto SET, int buf) __attribute__((always_inline)) {
        constexpr int ss = decltype(SET)::value;
        u32x4 (&ra)[NLOAD] = rsa[ss];
        u32x4 (&rb)[NLOAD] = rsb[ss];
        if (do_cs) {
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                if constexpr (BF) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { csum[2 * q] += __uint_as_float(ra[i][q] << 16); csum[2 * q + 1] += __uint_as_float(ra[i][q] & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) csum[q] += __uint_as_float(ra[i][q]);
                }
            }
        }
        if constexpr (GENB) {       // relu(scale * (x [+ y]) + shift), zero rows beyond the split
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                float v[VEC];
                const u32x4 x = rb[i];
                if constexpr (BF) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(x[q] << 16); v[2 * q + 1] = __uint_as_float(x[q] & 0xffff0000u); }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = __uint_as_float(x[q]);
                }
                if constexpr (PAIR) {
                    const u32x4 y = rsy[ss][i];
                    if constexpr (BF) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { v[2 * q] += __uint_as_float(y[q] << 16); v[2 * q + 1] += __uint_as_float(y[q] & 0xffff0000u); }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += __uint_as_float(y[q]);
                    }
                }
                const bool ok = (rok[ss] >> i) & 1u;
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    if constexpr (GMASK2) { const bool on = ok && (v[q] * bsc[q] + bsh[q] > 0.f); v[q] = on ? (second ? v[q] : 1.f) : 0.f; }
                    else v[q] = ok ? fmaxf(v[q] * bsc[q] + bsh[q], 0.f) : 0.f;
                }
                u32x4 r;
                if constexpr (BF) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = pack_bf2(v[2 * q], v[2 * q + 1]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) r[q] = __float_as_uint(v[q]);
                }
                rb[i] = r;
            }
        }
        T* as = lds + buf * ELEMS;
        T* bs = lds + (2 + buf) * ELEMS;
        if constexpr (BF) {
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                *reinterpret_cast<u32x4*>(as + (rt + 16 * i) * PITCH + cv) = ra[i];
                *reinterpret_cast<u32x4*>(bs + (rt + 16 * i) * PITCH + cv) = rb[i];
            }
        } else if constexpr (SPLIT) {
            auto put = [&](T* img, int row, const u32x4& v) __attribute__((always_inline)) {
                const float x0 = __uint_as_float(v[0]), x1 = __uint_as_float(v[1]), x2 = __uint_as_float(v[2]), x3 = __uint_as_float(v[3]);
                const uint32_t h0 = pack_bf2(x0, x1), h1 = pack_bf2(x2, x3);              // 3 VALU ops per value (gemm.hip lds_put_split)
                const uint32_t l0 = pack_bf2(x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = pack_bf2(x2 - __uint_as_float(h1 << 16), x3 - __uint_as_float(h1 & 0xffff0000u));
                unsigned char* b = reinterpret_cast<unsigned char*>(img) + (row * TS_PITCH + cv) * 2;
                *reinterpret_cast<u32x2*>(b) = u32x2{h0, h1};
                *reinterpret_cast<u32x2*>(b + TS_IMG_B) = u32x2{l0, l1};
            };
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) { put(as, rt + 8 * i, ra[i]); put(bs, rt + 8 * i, rb[i]); }
        } else {
#pragma unroll
            for (int i = 0; i < NLOAD; ++i) {
                *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(as) + (rt + 8 * i) * PITCH + cv) = ra[i];
                *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(bs) + (rt + 8 * i) * PITCH + cv) = rb[i];
            }
        }
    };

    const int nsteps = (m_end - m_beg + BM - 1) / BM;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (nsteps > 0) { load_step(S0{}, std::false_type{}, m_beg); if (nsteps > 1) load_step(S1{}, std::false_type{}, m_beg + BM); store_step(S0{}, 0); }
    __syncthreads();
    // FULLT = steady state (steps t+1 and t+2 exist): the loads / LDS stores carry no condition.  With the conditions inside the loop hipcc's
    // s_waitcnt pass drains vmcnt before re-issuing loads into a set (same finding as in gemm.hip); the last steps run the conditional form.
    auto body = [&](auto SET, auto FULLT, int t) __attribute__((always_inline)) {
        // SET = register set that held step t (free now: step t is in LDS); step t+1 sits in the other set
        constexpr int s0 = decltype(SET)::value;
        constexpr bool FULL = decltype(FULLT)::value;
        const int cur = t & 1;
        if (FULL || t + 2 < nsteps) load_step(std::integral_constant<int, s0>{}, FULLT, m_beg + (t + 2) * BM);
        const T* as = lds + cur * ELEMS;
        const T* bs = lds + (2 + cur) * ELEMS;
        if constexpr (BF) {
            // lane -> (4-row block, 16-column block, row in block, 4-column chunk) of the transposing read
            const int g4 = lane >> 4, li = lane & 15;
            const uint32_t lane_off = (uint32_t)((((g4 >> 1) * 8 + (li >> 2)) * PITCH + (g4 & 1) * 16 + (li & 3) * 4) * 2);
            const uint32_t abase = lds_base + (uint32_t)(cur * ELEMS * 2) + lane_off + (uint32_t)(wm * 64 * 2);
            const uint32_t bbase = lds_base + (uint32_t)((2 + cur) * ELEMS * 2) + lane_off + (uint32_t)(wn * 64 * 2);
            // software pipeline by hand (the compiler neither schedules nor counts inline-asm LDS reads): the reads of k-step kk+1 are
            // in flight while the MFMAs of kk run; each wait is tied to the registers it guards
            u32x2 fa[2][2][2], fb[2][2][2];          // [pipeline slot][32-column block][rows +0..3 | +4..7]
            auto issue = [&](int slot, int kk) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t off = (uint32_t)(((kk * 16 + hh * 4) * PITCH + i * 32) * 2);
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[slot][i][hh]) : "v"(abase + off));
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[slot][i][hh]) : "v"(bbase + off));
                    }
            };
            auto wait = [&](int slot) __attribute__((always_inline)) {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[slot][0][0]), "+v"(fa[slot][0][1]), "+v"(fa[slot][1][0]), "+v"(fa[slot][1][1]),
                               "+v"(fb[slot][0][0]), "+v"(fb[slot][0][1]), "+v"(fb[slot][1][0]), "+v"(fb[slot][1][1]));
            };
            issue(0, 0);
            wait(0);
#pragma unroll
            for (int kk = 0; kk < BM / 16; ++kk) {
                const int sl = kk & 1;
                if (kk + 1 < BM / 16) issue(sl ^ 1, kk + 1);
                s16x8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = __builtin_bit_cast(s16x8, u32x4{fa[sl][i][0].x, fa[sl][i][0].y, fa[sl][i][1].x, fa[sl][i][1].y});
                    bf[i] = __builtin_bit_cast(s16x8, u32x4{fb[sl][i][0].x, fb[sl][i][0].y, fb[sl][i][1].x, fb[sl][i][1].y});
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), af[i]),
                            __builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), bf[j]), acc[i][j], 0, 0, 0);
                if (kk + 1 < BM / 16) wait(sl ^ 1);
            }
        } else if constexpr (SPLIT) {
            const int g4 = lane >> 4, li = lane & 15;
            const uint32_t lane_off = (uint32_t)((((g4 >> 1) * 8 + (li >> 2)) * TS_PITCH + (g4 & 1) * 16 + (li & 3) * 4) * 2);
            const uint32_t abase = lds_base + (uint32_t)(cur * ELEMS * 4) + lane_off + (uint32_t)(wm * 64 * 2);
            const uint32_t bbase = lds_base + (uint32_t)((2 + cur) * ELEMS * 4) + lane_off + (uint32_t)(wn * 64 * 2);
            u32x2 fa[2][2][2], fb[2][2][2];          // [hi | lo image][32-column block][rows +0..3 | +4..7]
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const uint32_t off = (uint32_t)(im * TS_IMG_B + (hh * 4 * TS_PITCH + i * 32) * 2);
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[im][i][hh]) : "v"(abase + off));
                        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[im][i][hh]) : "v"(bbase + off));
                    }
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]),
                           "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]));
            typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
            bfx8 af[2][2], bf[2][2];                 // [image][block]
#pragma unroll
            for (int im = 0; im < 2; ++im)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[im][i] = __builtin_bit_cast(bfx8, u32x4{fa[im][i][0].x, fa[im][i][0].y, fa[im][i][1].x, fa[im][i][1].y});
                    bf[im][i] = __builtin_bit_cast(bfx8, u32x4{fb[im][i][0].x, fb[im][i][0].y, fb[im][i][1].x, fb[im][i][1].y});
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][j], acc[i][j], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < BM / 2; ++kk) {
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = reinterpret_cast<const float*>(as)[(kk * 2 + hi) * PITCH + wm * 64 + i * 32 + l31];
                    bf[i] = reinterpret_cast<const float*>(bs)[(kk * 2 + hi) * PITCH + wn * 64 + i * 32 + l31];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (FULL || t + 1 < nsteps) store_step(std::integral_constant<int, s0 ^ 1>{}, cur ^ 1);
        __syncthreads();
    };
    {
        int t = 0;
        for (; t + 4 < nsteps; t += 2) { body(S0{}, std::true_type{}, t); body(S1{}, std::true_type{}, t + 1); }   // loads steps <= nsteps - 2: full rows
        for (; t < nsteps; t += 2) {
            body(S0{}, std::false_type{}, t);
            if (t + 1 < nsteps) body(S1{}, std::false_type{}, t + 1);
        }
    }
    if (do_cs) {   // fold the per-thread column partials (threads with equal cv) through the idle LDS, one atomic per column
        float* red = reinterpret_cast<float*>(lds);
        constexpr int NRT = 256 / TPR;
#pragma unroll
        for (int q = 0; q < VEC; ++q) red[rt * TN + cv + q] = csum[q];
        __syncthreads();
        if (tid < TN && tn * TN + tid < g.N) {
            float a = 0.f;
            for (int r = 0; r < NRT; ++r) a += red[r * TN + tid];
            if (g.cs_slab) g.cs_slab[(int64_t)split * g.N + tn * TN + tid] = a;
            else atomicAdd(g.colsum + tn * TN + tid, a);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = tk * TK + wn * 64 + j * 32 + l31;
        if (col >= g.K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = tn * TN + wm * 64 + i * 32 + crow32(r, hi);
                if (row < g.N) {
                    if (g.slabs) g.slabs[((int64_t)split * g.N + row) * g.K + col] = acc[i][j][r];
                    else atomicAdd(g.C + (int64_t)row * g.ldc + col, acc[i][j][r]);
                }
            }
    }
}

// C[n,k] += sum_s slabs[s][n][k]
__global__ void tn_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ C, int N, int K, int ldc, int splits) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0;                        // split order, float64: bit-reproducible and nothing lost between the partial tiles
        for (int s = 0; s < splits; ++s) a += (double)slabs[(int64_t)s * total + i];
        const int n = (int)(i / K), k = (int)(i - (int64_t)n * K);
        C[(int64_t)n * ldc + k] = (float)((double)C[(int64_t)n * ldc + k] + a);
    }
}

// column sums of a [M, N] matrix (bias / positional-embedding gradients): out[c] += sum_m x[m, c]
// block = ncq column-quads x (256 / ncq) row lanes (ncq = 64 for N >= 256, else N/4 rounded up to a power of two, so narrow
// matrices such as the ScoreNet's [B*N*N, 64] still use every lane); 8/16-byte coalesced loads, 4 rows in flight per thread,
// register accumulation, LDS fold, one atomic per column per block with ~2048 blocks (r01: one atomic per 128 rows = 2.4 M
// atomics on 128 addresses made this kernel run at 1.5 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t M, int N, int ld, int rows_per_block,
                                                     int ncq, float* __restrict__ slab) {
    __shared__ float red[1024];
    const int q = threadIdx.x % ncq, ry = threadIdx.x / ncq, nrl = 256 / ncq;
    const int c0 = (blockIdx.y * 64 + q) * 4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    auto add = [&](const T* p) {
        if constexpr (sizeof(T) == 2) {
            const uint2 raw = *reinterpret_cast<const uint2*>(p);
            a[0] += __uint_as_float(raw.x << 16); a[1] += __uint_as_float(raw.x & 0xffff0000u);
            a[2] += __uint_as_float(raw.y << 16); a[3] += __uint_as_float(raw.y & 0xffff0000u);
        } else {
            const float4 raw = *reinterpret_cast<const float4*>(p);
            a[0] += raw.x; a[1] += raw.y; a[2] += raw.z; a[3] += raw.w;
        }
    };
    if (c0 < N) {
        int64_t r = r0 + ry;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {
            const T* p = x + r * ld + c0;
            add(p); add(p + (int64_t)nrl * ld); add(p + (int64_t)2 * nrl * ld); add(p + (int64_t)3 * nrl * ld);
        }
        for (; r < r1; r += nrl) add(x + r * ld + c0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[ry * (ncq * 4) + q * 4 + k] = a[k];
    __syncthreads();
    if ((int)threadIdx.x < ncq * 4) {
        float s = 0.f;
        for (int y = 0; y < nrl; ++y) s += red[y * (ncq * 4) + threadIdx.x];
        const int c = blockIdx.y * 256 + threadIdx.x;
        if (c < N) {
            if (slab) slab[(int64_t)blockIdx.x * N + c] = s;      // deterministic mode: row-block partials, summed in block order
            else atomicAdd(out + c, s);
        }
    }
}

// scalar fallback for N / ld not a multiple of 4
template <typename T>
__global__ __launch_bounds__(256) void colsum_scalar_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t M, int N, int ld, int rows_per_block,
                                                            float* __restrict__ slab) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    for (int c = threadIdx.x + blockIdx.y * 256; c < N; c += 256 * gridDim.y) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s += Cvt<T>::to_f(x[r * ld + c]);
        if (slab) slab[(int64_t)blockIdx.x * N + c] = s;
        else atomicAdd(out + c, s);
    }
}

}  // namespace

void p3_tn_reduce_launch(const float* slabs, float* C, int N, int K, int ldc, int splits, hipStream_t s) {
    int64_t gr = ((int64_t)N * K + 255) / 256; if (gr > 4096) gr = 4096;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((int)gr), dim3(256), 0, s, slabs, C, N, K, ldc, splits);
}

// ---- deferred weight-gradient reduces (r05) ------------------------------------------------------------------------------------------------------
// In the deterministic mode every split-M weight gradient stores its partial tiles and a tn_reduce launch adds them to C at once: 109 launches of ~16 us per
// fp32x3 train step.  When C is an accumulation target that stays valid and unread until the end of the backward pass (a view of the optimizer's gradient
// arena), the partial tiles may instead be PARKED in a caller-provided arena (p3_tn_defer) and added by p3_tn_flush: the same float64 sum in split order per
// element - bit-identical gradients - in ceil(n / 96) launches.  The host brackets exactly those launches with p3_tn_defer_enable(1 / 0).
namespace {
constexpr int TNP_MAX = 96;                 // entries per flush launch (kernel-argument table: 96 x 40 B)
struct TnpEntry { const float* slab; float* C; int N, K, ldc, splits, first_block, pad; };
struct TnpTable { TnpEntry e[TNP_MAX]; int n; };
constexpr int TNP_CAP = 4 * TNP_MAX;
TnpEntry g_tnp[TNP_CAP];
int g_tnp_n = 0, g_tnp_on = 0;
float* g_tnp_arena = nullptr;
int64_t g_tnp_cap = 0, g_tnp_used = 0;

__global__ __launch_bounds__(256) void tn_flush_kernel(TnpTable t) {
    // block -> entry (the table is small: a linear walk), then 256 x 4 consecutive elements of that entry's tile
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.e[e + 1].first_block) ++e;
    const TnpEntry en = t.e[e];
    const int64_t total = (int64_t)en.N * en.K;
    const int64_t i0 = ((int64_t)(blockIdx.x - en.first_block) * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    // split order, float64 (tn_reduce_kernel's sum); eight loads in flight per thread (r06: one at a time the launch ran at ~3 TB/s over the 2 - 4 GB of parked tiles)
    int s = 0;
    for (; s + 8 <= en.splits; s += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(en.slab + (int64_t)(s + u) * total + i0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w; }
    }
    for (; s < en.splits; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(en.slab + (int64_t)s * total + i0);
        a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
    }
    const int n = (int)(i0 / en.K), k = (int)(i0 - (int64_t)n * en.K);       // K % 4 == 0: the four elements share a row
    float* c = en.C + (int64_t)n * en.ldc + k;
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = (float)((double)c[q] + a[q]);
}
}  // namespace

extern "C" int p3_tn_defer(float* arena, int64_t floats) { g_tnp_arena = arena; g_tnp_cap = arena ? floats : 0; g_tnp_used = 0; g_tnp_n = 0; return P3_OK; }
extern "C" int p3_tn_defer_enable(int on) { const int was = g_tnp_on; g_tnp_on = on ? 1 : 0; return was; }
extern "C" int p3_tn_pending(void) { return g_tnp_n; }
extern "C" int p3_tn_drop(void) { const int n = g_tnp_n; g_tnp_n = 0; g_tnp_used = 0; return n; }

// a slot of splits x N x K floats for this launch's partial tiles, with the reduce left to p3_tn_flush; NULL: the caller reduces at once
float* p3_tn_park(float* C, int N, int K, int ldc, int splits) {
    const int64_t floats = (int64_t)splits * N * K;
    if (!g_tnp_on || !g_tnp_arena || g_tnp_n >= TNP_CAP || g_tnp_used + floats > g_tnp_cap || K % 4 != 0 || ldc % 4 != 0 || ((uintptr_t)C % 16) != 0) return nullptr;
    // a target that overlaps an already parked one (a shared weight, a second gradient into the same rows) is reduced at once: two entries of one flush would read-modify-write
    // the same addresses from different workgroups (the parked one keeps its place; both sums end up in C, in launch order)
    {
        const uintptr_t lo = (uintptr_t)C, hi = (uintptr_t)(C + (int64_t)(N - 1) * ldc + K);
        for (int i = 0; i < g_tnp_n; ++i) {
            const TnpEntry& o = g_tnp[i];
            const uintptr_t olo = (uintptr_t)o.C, ohi = (uintptr_t)(o.C + (int64_t)(o.N - 1) * o.ldc + o.K);
            if (lo < ohi && olo < hi) {
                // address ranges interleave - but two COLUMN SLICES of one matrix (the nine shifted products of a 3 x 3 convolution's weight gradient, each into its
                // own [Co, Ci] slice of the [Co, 9 Ci] weight) share no element: same row stride, column intervals disjoint inside a row
                if (o.ldc == ldc) {
                    const bool fwd = C >= o.C;
                    const int64_t dist = fwd ? C - o.C : o.C - C;
                    const int64_t cs = dist % ldc;                           // the later slice's first column, counted from the earlier one's
                    const int k_first = fwd ? o.K : K, k_second = fwd ? K : o.K;
                    if (cs >= k_first && cs + k_second <= ldc) continue;
                }
                return nullptr;
            }
        }
    }
    float* slot = g_tnp_arena + g_tnp_used;
    g_tnp_used += (floats + 63) / 64 * 64;
    TnpEntry& en = g_tnp[g_tnp_n++];
    en.slab = slot; en.C = C; en.N = N; en.K = K; en.ldc = ldc; en.splits = splits; en.first_block = 0; en.pad = 0;
    return slot;
}

extern "C" int p3_tn_flush(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < g_tnp_n; base += TNP_MAX) {
        TnpTable t;
        t.n = g_tnp_n - base < TNP_MAX ? g_tnp_n - base : TNP_MAX;
        int blocks = 0;
        for (int i = 0; i < t.n; ++i) {
            t.e[i] = g_tnp[base + i];
            t.e[i].first_block = blocks;
            blocks += p3_ceil_div((int64_t)t.e[i].N * t.e[i].K, 1024);
        }
        hipLaunchKernelGGL(tn_flush_kernel, dim3(blocks), dim3(256), 0, s, t);
    }
    g_tnp_n = 0; g_tnp_used = 0;
    P3_LAUNCH_CHECK();
    return P3_OK;
}
int p3_gemm_tn_dma_try(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, float* colsum, float* slabs, int max_slabs,
                       hipStream_t s);      // gemm_tn_dma.hip

int p3_pair_dw_try(const void* A, const void* U, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* scale, const float* shift,
                   const void* pair_V, int pair_n, float* slabs, int max_slabs, hipStream_t s);      // pair_dw_mma.hip

int p3_mask2_dw_try(const void* A, const void* B, float* C, int M, int N, int Kb, int lda, int ldb, int ldc, const float* scale, const float* shift,
                    float* slabs, int max_slabs, hipStream_t s);      // mask2_dw_mma.hip
int p3_mask2_dw_x3_try(const void* A, const void* B, float* C, int M, int N, int Kb, int lda, int ldb, int ldc, const float* scale, const float* shift,
                       float* slabs, int max_slabs, hipStream_t s);      // mask2_dw_x3.hip: the P3_F32X3 form
int p3_pair_dw_x3_try(const void* A, const void* U, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* scale, const float* shift,
                      const void* pair_V, int pair_n, float* slabs, int max_slabs, hipStream_t s);   // pair_dw_x3.hip: the P3_F32X3 form

extern "C" int p3_gemm_tn_ex(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int dtype_in, int b_mode,
                             const float* b_scale, const float* b_shift, const void* pair_V, int pair_n, float* colsum, float* slabs, int max_slabs, void* stream) {
    const bool split = dtype_in == P3_F32X3;       // fp32 operands, products as bf16 x 3
    const int dtype = split ? P3_F32 : dtype_in;
    P3_CHECK(A && B && C && M > 0 && N > 0 && K > 0, P3_EINVAL, "p3_gemm_tn: bad arguments");
    P3_CHECK(b_mode == 0 || b_mode == P3_A_AFFINE_RELU || b_mode == P3_A_PAIR_AFFINE_RELU || b_mode == P3_A_AFFINE_MASK2, P3_EINVAL, "p3_gemm_tn: b_mode");
    P3_CHECK(b_mode != P3_A_AFFINE_MASK2 || ldc >= 2 * K, P3_ESHAPE, "p3_gemm_tn: P3_A_AFFINE_MASK2 writes [N, 2K]");
    P3_CHECK(b_mode != P3_A_AFFINE_MASK2 || K % 8 == 0, P3_ESHAPE, "p3_gemm_tn: P3_A_AFFINE_MASK2 needs K % 8 == 0 (a staged vector must not straddle the two generated operands)");
    P3_CHECK(b_mode == 0 || (b_scale && b_shift), P3_EINVAL, "p3_gemm_tn: generated B operand needs b_scale / b_shift");
    P3_CHECK(b_mode != P3_A_PAIR_AFFINE_RELU || (pair_V && pair_n > 0 && M % (pair_n * pair_n) == 0), P3_ESHAPE, "p3_gemm_tn: pair mode needs V and M == B*n*n");
    P3_CHECK(dtype == P3_F32 || dtype == P3_BF16, P3_EUNSUP, "p3_gemm_tn: dtype");
    const int vec = dtype == P3_BF16 ? 8 : 4;
    P3_CHECK(N % vec == 0 && K % vec == 0 && lda % vec == 0 && ldb % vec == 0, P3_EALIGN, "p3_gemm_tn: N, K, lda, ldb must be multiples of 8 (bf16) / 4 (f32)");
    P3_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, P3_EALIGN, "p3_gemm_tn: 16-byte base alignment");
    if (dtype == P3_BF16 && b_mode == 0) {          // the plain bf16 weight gradients at 64 / 128-aligned shapes: LDS-DMA kernel (gemm_tn_dma.hip)
        const int rc = p3_gemm_tn_dma_try(A, B, C, M, N, K, lda, ldb, ldc, colsum, slabs, max_slabs, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    if (dtype == P3_BF16 && b_mode == P3_A_PAIR_AFFINE_RELU && !colsum) {      // the ScoreNet's conv2 weight gradient: pair_dw_mma.hip
        const int rc = p3_pair_dw_try(A, B, C, M, N, K, lda, ldb, ldc, b_scale, b_shift, pair_V, pair_n, slabs, max_slabs, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    if (split && b_mode == P3_A_PAIR_AFFINE_RELU && !colsum) {                     // the same launch with fp32 operands, products as bf16 x 3: pair_dw_x3.hip
        const int rc = p3_pair_dw_x3_try(A, B, C, M, N, K, lda, ldb, ldc, b_scale, b_shift, pair_V, pair_n, slabs, max_slabs, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    if (dtype == P3_BF16 && b_mode == P3_A_AFFINE_MASK2 && !colsum) {             // conv3's dual-operand weight gradient: mask2_dw_mma.hip
        const int rc = p3_mask2_dw_try(A, B, C, M, N, K, lda, ldb, ldc, b_scale, b_shift, slabs, max_slabs, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    if (split && b_mode == P3_A_AFFINE_MASK2 && !colsum) {                        // fp32 operands, products as bf16 x 3: mask2_dw_x3.hip
        const int rc = p3_mask2_dw_x3_try(A, B, C, M, N, K, lda, ldb, ldc, b_scale, b_shift, slabs, max_slabs, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    TnArgs g; g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.Kb = K;
    if (b_mode == P3_A_AFFINE_MASK2) { K *= 2; g.K = K; }       // two generated operands side by side
    g.b_mode = b_mode; g.b_scale = b_scale; g.b_shift = b_shift; g.pair_V = pair_V; g.pair_n = pair_n; g.colsum = colsum;
    const int tiles_n = p3_ceil_div(N, TN);
    g.tiles_k = p3_ceil_div(K, TK);
    const int tiles = tiles_n * g.tiles_k;
    const int bm = dtype == P3_BF16 ? 64 : 16;
    // M-split count (measured r01, tools/mb_tn.py sweep 512..2048): ~768 workgroups (1.5 x the 512 that fit the chip at 2 per CU)
    // balances per-block prologue / atomic-epilogue overhead against tail imbalance: +8..13 % over 1024; exactly one full wave of
    // 512 blocks is better still when the tile count divides it (decoder linear1: 64 vs 82 us).  P3_TN_BLOCKS overrides for sweeps.
    // same-box sweep r01: 600 -> 60.1 ms, 700 -> 59.6, 768 -> 59.6, 850 -> 59.3, 950 -> 59.3; r03 (two-deep prefetch since r02, every split costs
    // N x K fp32 atomics whose lines migrate between the XCDs' L2s - 15 us of the 70 us mean launch, P3_DETERMINISTIC=2 A/B): 512 -> 39.59 ms,
    // 576 -> 39.47, 640 -> 39.05, 704 -> 39.00, 768 -> 39.39, 896 -> 39.34, 1024 -> 39.91
    // r03, second sweep (the first never went below 512): the best grids are the ones that fit ONE resident wave of workgroups (2 per CU = 512) -
    // 320 -> 40.48 ms, 352 -> 40.11, 384 -> 39.89, 416 -> 39.59, 448 -> 39.41, 480 -> 39.55, 704 -> 40.33 (same box); per shape (tools/probe/mb_tn_sweep_r03.py)
    // qkv 82 us at 405 workgroups, 105 at 513 (one past the wave), 93 at 704
    const int tgt = 448;
    // (measured and dropped: outputs of <= 8 tiles - the decoder's 256 x 256 projections, the head - run 18.5 instead of 27.7 us ALONE with ~192
    // instead of 512 workgroups, tools/probe/mb_tn_sweep_r03.py, but the captured step got 0.2 ms slower with that rule: 39.93 vs 39.72 ms, same box)
    int splits = (512 % tiles == 0) ? 512 / tiles : p3_ceil_div(tgt, tiles);
    if (tgt <= 512 && splits > 1 && splits * tiles > 512) --splits;              // never one workgroup past the resident wave
    if (slabs && splits > max_slabs) splits = max_slabs;
    int max_splits = p3_ceil_div(M, 4 * bm);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    g.rows_per_split = p3_ceil_div(p3_ceil_div(M, splits), bm) * bm;
    splits = p3_ceil_div(M, g.rows_per_split);
    g.slabs = (slabs && splits > 1) ? slabs : nullptr;
    bool parked = false;
    if (g.slabs) { float* slot = p3_tn_park(C, N, K, ldc, splits); if (slot) { g.slabs = slot; parked = true; } }
    g.splits = splits;
    int cs_parked = 0;
    g.cs_slab = colsum ? p3_colsum_parts(splits, N, colsum, dtype, &cs_parked) : nullptr;
    dim3 grid(tiles * splits), block(256);
    hipStream_t s = (hipStream_t)stream;
#define P3_TN_LAUNCH(MODE)                                                                            \
    do {                                                                                              \
        if (dtype == P3_BF16) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, MODE>), grid, block, 0, s, g); \
        else if (split) hipLaunchKernelGGL((gemm_tn_kernel<float, MODE, true>), grid, block, 0, s, g); \
        else hipLaunchKernelGGL((gemm_tn_kernel<float, MODE>), grid, block, 0, s, g);                  \
    } while (0)
    if (g.b_mode == 0) P3_TN_LAUNCH(0);
    else if (g.b_mode == P3_A_AFFINE_RELU) P3_TN_LAUNCH(P3_A_AFFINE_RELU);
    else if (g.b_mode == P3_A_AFFINE_MASK2) P3_TN_LAUNCH(P3_A_AFFINE_MASK2);
    else P3_TN_LAUNCH(P3_A_PAIR_AFFINE_RELU);
#undef P3_TN_LAUNCH
    if (g.slabs && !parked) {
        int64_t gr = ((int64_t)N * K + 255) / 256; if (gr > 4096) gr = 4096;
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((int)gr), dim3(256), 0, s, g.slabs, C, N, K, ldc, splits);
    }
    P3_LAUNCH_CHECK();
    if (g.cs_slab && !cs_parked) return p3_det_reduce(g.cs_slab, splits, N, colsum, N, 1, s);
    return P3_OK;
}

extern "C" int p3_gemm_tn(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int dtype, void* stream) {
    return p3_gemm_tn_ex(A, B, C, M, N, K, lda, ldb, ldc, dtype, 0, nullptr, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
}

extern "C" int p3_colsum(const void* x, float* out, int64_t M, int N, int ld, int dtype, void* stream) {
    P3_CHECK(x && out && M >= 0 && N > 0, P3_EINVAL, "p3_colsum: bad arguments");
    P3_CHECK(dtype == P3_BF16 || dtype == P3_F32, P3_EUNSUP, "p3_colsum: dtype");
    if (M == 0) return P3_OK;
    hipStream_t s = (hipStream_t)stream;
    const int es = dtype == P3_BF16 ? 2 : 4;
    const bool vec = (N % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)x % (4 * es) == 0);
    int rpb = M > 16384 ? 128 : (M > 2048 ? 32 : 8);
    if (vec) {
        int ncq = 64;
        if (N < 256) { const int nq = (N + 3) / 4; ncq = 1; while (ncq < nq) ncq <<= 1; }
        const int cols_blocks = p3_ceil_div(N, 256);
        const int64_t want = 2048 / cols_blocks > 0 ? 2048 / cols_blocks : 1;        // ~2048 blocks in total
        const int64_t r = (M + want - 1) / want;
        if (r > rpb) rpb = (int)r;
        dim3 grid(p3_ceil_div(M, rpb), cols_blocks), block(256);
        float* slab = p3_det_scratch((int64_t)grid.x * N, dtype);
        if (dtype == P3_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, out, M, N, ld, rpb, ncq, slab);
        else hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, s, (const float*)x, out, M, N, ld, rpb, ncq, slab);
        P3_LAUNCH_CHECK();
        if (slab) return p3_det_reduce(slab, (int)grid.x, N, out, N, 1, s);
        return P3_OK;
    }
    dim3 grid(p3_ceil_div(M, rpb), p3_ceil_div(N, 256)), block(256);
    float* slab = p3_det_scratch((int64_t)grid.x * N, dtype);
    {
        if (dtype == P3_BF16) hipLaunchKernelGGL((colsum_scalar_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, out, M, N, ld, rpb, slab);
        else hipLaunchKernelGGL((colsum_scalar_kernel<float>), grid, block, 0, s, (const float*)x, out, M, N, ld, rpb, slab);
    }
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce(slab, (int)grid.x, N, out, N, 1, s);
    return P3_OK;
}
