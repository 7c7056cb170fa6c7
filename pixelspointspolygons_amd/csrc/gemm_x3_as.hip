// p3hip GEMM on PLANES, A-STATIONARY form (include/p3hip.h, p3_gemm_x3) for the short-K products of the timm Block (K = 384: qkv, proj, fc1, dX of fc2 / proj;
// vit.py:48, early_fusion_vit.py:124) and of nn.TransformerDecoderLayer (K = 256; model_pix2poly.py:138).
//
// What bounds the tile kernels of gemm_x3.hip at these shapes is the CU's memory path: a 128 x 128 tile stages 393 KB global -> LDS for 1152 MFMAs and then
// stores 64 .. 128 KB, every LDS-DMA GEMM of this library tops out at 8 - 10 TB/s of staging, and at K = 384 the whole A row of a wave
// is only 192 registers.  So here the ACTIVATION rows stay in registers and only the weights move:
//   * a workgroup = 8 waves x 32 rows of A; a wave loads its 32 x K slice of a_hi / a_lo ONCE per row block, fragment-shaped, straight into registers
//     (K / 16 x 2 x 4 VGPRs: 192 at K = 384) - A is read from HBM exactly once per launch and never touches LDS;
//   * the weights stream through LDS in blocks of 32 output columns x K (hi + lo: 48 KB at K = 384), by LDS-DMA in full 128-byte lines (8 rows x 64 k per
//     piece, chunk slot = chunk ^ ((row >> 1) & 7) applied on the SOURCE address), shared by all 8 waves: 48 KB staged per 576 MFMAs - a quarter of the
//     128 x 128 tile's bytes per MFMA.  W (<= 2.4 MB as planes) stays in every XCD's L2;
//   * the product is computed TRANSPOSED, D[n][m] = sum_k W[n][k] A[m][k] (the weight fragment is the MFMA's A operand), so a lane owns one output ROW m and
//     16 of the block's 32 columns; one v_permlane32_swap per register pair turns that into two chunks of 8 consecutive columns per lane: the epilogue
//     (bias, GELU + GELU', x mul, + residual, fp32 or planes out - the arithmetic of x3_epi8) runs out of the accumulators with 16-byte accesses, no LDS;
//   * a unit of work = (row block, column block); the launch is PERSISTENT: one workgroup per CU takes a contiguous range of the row-block-major unit list
//     (197 x 48 units for fc1 = 36.94 per CU: no tile-quantisation rounds), reloading its A slice when the row block changes (once or twice per launch);
//   * time is cut into TICKS of half a column block (K / 2 deep: 36 MFMAs per wave at K = 384), one barrier per tick.  Waves 4..7 run ONE TICK BEHIND waves
//     0..3, so on every SIMD the wave that has just finished a block runs its epilogue (loads, VALU, stores) while its partner is in the middle of a block and
//     keeps the matrix pipe busy - the overlap the tile kernels could only get from a second workgroup per CU.
// r06 measurements that shaped the tick (s_memtime sums per wave and stage, profiles/r06_as_stage_cycles.txt): the first form - wait vmcnt(0), barrier, epilogue,
// DMA issue, 36 MFMAs - ran 5977 cycles per tick on qkv for 2304 cycles of matrix-pipe work: an epilogue cost 3256 cycles (each chunk's loads waited for the
// previous chunk's STORES: vmcnt counts both), the 3 DMA pieces + their address arithmetic (a division per tick) 260 .. 500 cycles in front of the first MFMA.  Now:
//   * the epilogue issues ALL its loads first and its stores last, and nothing waits for a store before the next tick's barrier;
//   * the DMA pieces of half block t + 4 are issued BEHIND the MFMAs of tick t (ring of six half blocks), by waves 0..3 only, and stay in flight across the
//     barrier: the tick opens with a counted `s_waitcnt vmcnt`;
//   * what an epilogue will load (multiplier / residual) and the next row block's A slice are pulled into L2 ahead of time by register-less LDS-DMA touches;
//   * unit -> (row block, column block) is kept incrementally (no division in the loop).
// MFMA order per 16-deep step: w_lo a_hi, w_hi a_lo, w_hi a_hi (small terms first), ascending k - one accumulator per wave, the partner's MFMAs in between.
#include <stdio.h>
#include <stdlib.h>

#include "p3_common.h"
#include "gemm_x3_epi.h"

namespace {

struct AsArgs {
    p3_gemm_x3_desc d;
    int CB;              // column blocks of 32
    int units;           // row blocks x CB
    unsigned long long* dbg;   // DBG: [workgroup][wave][8] cycle sums (barrier wait, epilogue, DMA issue, A reload, half 0, half 1, total, ticks)
};

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef uint32_t as_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void as_dma1(const bf16_t* base, uint32_t dst, uint32_t v0) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(v0), "s"(base), "s"(dst) : "memory");
}

// 16- / 8-byte stores at wave-uniform base + 32-bit byte offset; SC1: write-through (the line is not kept in this XCD's L2, where the weight stream lives)
typedef float as_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t as_u32x2 __attribute__((ext_vector_type(2)));
template <bool SC1> __device__ __forceinline__ void as_store16(void* base, uint32_t off, as_f32x4 v) {
    if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(off), "v"(v), "s"(base) : "memory");
    else *reinterpret_cast<as_f32x4*>(reinterpret_cast<char*>(base) + (size_t)off) = v;
}
template <bool SC1> __device__ __forceinline__ void as_store8(void* base, uint32_t off, as_u32x2 v) {
    if constexpr (SC1) asm volatile("global_store_dwordx2 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(off), "v"(v), "s"(base) : "memory");
    else *reinterpret_cast<as_u32x2*>(reinterpret_cast<char*>(base) + (size_t)off) = v;
}

// The lane id again, opaque to the optimiser: what the epilogue and the A reload derive from it (row, column chunk, pointers) is computed where it is used
// instead of living in ~16 VGPRs across the MFMA loop - at K = 384 the A slice (192) + accumulator (16) + two fragment sets (16) leave 32 registers in all.
__device__ __forceinline__ int as_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

constexpr int AS_PLANES = 1, AS_GELU = 2, AS_MUL = 4, AS_RES = 8;      // EPI bits (bias and aux stay run-time switches: wave-uniform, cheap)
constexpr int AS_NOEPI = 16, AS_NODMA = 32, AS_FOLD = 64;             // measurement: no epilogue / no weight stream / every row block's epilogue traffic folded
                                                                      // onto rows 0..255 (cache-resident) - tools/mb_as.py; wrong results, timing only
constexpr int AS_SC1 = 128;                                           // write-through (sc1) epilogue stores: the output lines do not stay in the XCD's L2, where the weight
                                                                      // stream lives (r06 same-box A/B, profiles/r06_mb_as.txt: qkv 161 -> 151 us, fc1 279 -> 271); var 5 = plain
constexpr int AS_NOPRIO = 256;                                        // measurement: the finisher's MFMAs at normal priority
constexpr int AS_SLOTS = 5;                                           // half blocks in the LDS ring (120 KB at K = 384; + bias, junk, 18 KB of epilogue images)

// KS = K / 16 (24: K = 384, 16: K = 256); EPI: which epilogue streams exist (compile-time: the register budget has no room for the union of their operands)
template <int KS, int EPI, bool DBG>
__global__ __launch_bounds__(512, 2) void gemm_x3_as_kernel(AsArgs g) {
    constexpr int HS = KS / 2;                  // 16-deep steps per half block
    constexpr int GH = HS / 4;                  // 64-deep groups per half block
    constexpr int HALF_B = GH * 2 * 4096;       // bytes of a half block in LDS: [image][group][32 rows][8 slots of 16 B]
    constexpr bool PLANES = (EPI & AS_PLANES) != 0;
    static_assert(HS % 4 == 0, "K must be a multiple of 128");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const p3_gemm_x3_desc& d = g.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int grp = wave >> 2;                  // 0: waves 0..3, 1: waves 4..7 (one tick behind)
    const int CB = g.CB;
    // this workgroup's contiguous unit range
    const int nwg = gridDim.x, q = g.units / nwg, r = g.units % nwg, b = blockIdx.x;
    const int u0 = b * q + min(b, r), nU = q + (b < r ? 1 : 0);
    if (nU == 0) return;
    const int rb0 = u0 / CB, cb0 = u0 - rb0 * CB;
    const bf16_t* Ah_ = reinterpret_cast<const bf16_t*>(d.a_hi);
    const bf16_t* Al_ = reinterpret_cast<const bf16_t*>(d.a_lo);
    const bf16_t* Wh_ = reinterpret_cast<const bf16_t*>(d.w_hi);
    const bf16_t* Wl_ = reinterpret_cast<const bf16_t*>(d.w_lo);

    // ---- LDS-DMA: issued by waves 0..3 only (they lose no time: the older wave of a SIMD wins the matrix pipe and would sit at the barrier anyway; a piece costs
    // 120 - 230 issue cycles, and on waves 4..7 those cycles were the tick's critical path).  Wave w fills row group rg = w (8 rows x 128 B) of both images,
    // groups 0 .. GH - 1 of the half block: 2 GH pieces
    const int rg = wave & 3;
    uint32_t voffW;
    {
        const int row = rg * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
        voffW = (uint32_t)((row * d.ldb + c * 8) * 2);
    }
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(&lds[0]));
    const uint32_t dst_w = lds_addr + (uint32_t)(rg * 1024);
    // the DMA stream walks half blocks hb = 0, 1, 2, ...: (column block, half) kept incrementally
    int d_hb = 0, d_cb = cb0, d_slot = 0;
    // piece x of the next half block: x < GH: w_hi, 64-deep group x; else w_lo, group x - GH
    auto dma_piece = [&](int x) __attribute__((always_inline)) {
        const int64_t so = (int64_t)d_cb * 32 * d.ldb + (d_hb & 1) * (GH * 64) + (x % GH) * 64;
        as_dma1((x < GH ? Wh_ : Wl_) + so, dst_w + (uint32_t)(d_slot * HALF_B + x * 4096), voffW);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        if (d_hb & 1) { if (++d_cb == CB) d_cb = 0; }
        ++d_hb;
        if (++d_slot == AS_SLOTS) d_slot = 0;
    };
    auto dma_next = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < 2 * GH; ++x) dma_piece(x);
        dma_advance();
    };
    // Prefetch without registers: a 4-byte LDS-DMA per lane into a junk corner of LDS pulls the line the lane points at into this XCD's L2 -
    //   * the multiplier / residual row segment of a unit (32 columns x 4 B = one 128-byte line per lane), issued when the unit's first half starts: two ticks
    //     later the epilogue's loads are L2 hits instead of 2 - 3 us of HBM latency in front of the partner-covered window;
    //   * the next row block's A slice (32 rows x K x 2 B per image = K / 64 lines per row), issued when the last unit of the current row block starts.
    const uint32_t junk = lds_addr + (uint32_t)(AS_SLOTS * HALF_B + d.N * 4 + wave * 256);
    auto touch = [&](const void* base, uint32_t byte_off) __attribute__((always_inline)) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep) : "v"(byte_off), "s"(base), "s"(junk) : "memory");
    };
    auto prefetch_x = [&](int rb, int cb) __attribute__((always_inline)) {
        if constexpr ((EPI & (AS_MUL | AS_RES)) != 0) {
            const float* xs = (EPI & AS_MUL) ? d.mul : d.residual;
            const uint32_t ldx = (uint32_t)((EPI & AS_MUL) ? d.ldmul : d.ldr);
            const uint32_t row = (uint32_t)min(rb * 256 + wave * 32 + (as_lane() & 31), d.M - 1);
            touch(xs, (row * ldx + (uint32_t)(cb * 32)) * 4u);
        }
    };
    auto prefetch_a = [&](int rb) __attribute__((always_inline)) {
        constexpr int LPR = KS / 4;                       // 128-byte lines per row and image
        const int ln = as_lane();
#pragma unroll
        for (int j = 0; j < LPR / 2; ++j) {               // 32 rows x LPR lines = LPR / 2 instructions of 64 lanes
            const int i = ln + 64 * j, rr = i / LPR, li = i - rr * LPR;
            const uint32_t off = (uint32_t)(min(rb * 256 + wave * 32 + rr, d.M - 1) * d.lda * 2 + li * 128);
            touch(Ah_, off);
            touch(Al_, off);
        }
    };

    // ---- the bias vector (zeros without one) behind the ring: a unit's accumulators start from it - LDS reads in the register order of the MFMA result, on
    // lgkmcnt: a global load here would sit behind the epilogue's stores in the vmcnt order, and the epilogue has no registers for it
    float* bias_lds = reinterpret_cast<float*>(lds + AS_SLOTS * HALF_B);
    for (int i = tid; i < d.N / 4; i += 512)
        reinterpret_cast<float4*>(bias_lds)[i] = d.bias ? reinterpret_cast<const float4*>(d.bias)[i] : make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- MFMA: v_mfma_f32_16x16x32_bf16.  A unit's 32 x 32 block is FOUR 16 x 16 accumulators (column tile ct x row tile rt, 4 registers each - the 16 of one
    // 32 x 32 accumulator), i.e. four INDEPENDENT dependency chains: issued round-robin, a wave's next MFMA on the same accumulator is 4 issues (64 cycles)
    // away and one wave alone keeps the matrix pipe busy.  (r06, profiles/r06_as_pmc_single_chain.txt: the first form accumulated into ONE 32 x 32 register
    // block - 36 dependent MFMAs per tick; SQ counters: matrix pipe 44 - 54 % busy, the waves stalled at MFMA issue 43 - 49 % of their cycles.  A dependent
    // chain issues one MFMA per result latency, about half the pipe rate, and the staggered ticks make a wave run alone most of the time.)
    //   W fragment (the MFMA's A operand): lane (n = l % 16, q = l / 16) reads the 16 bytes k = 32 j + 8 q .. + 8 of row 16 ct + n: chunk 4 (j & 1) + q of the 64-deep
    //   group j >> 1, slot = chunk ^ ((n >> 1) & 7) (the 16 ct of the row index does not reach the swizzle bits) -> foff0 ^ (64 (j & 1)) + constants;
    //   A fragment (B operand): row 16 rt + (l % 16) of the wave's 32 rows, the same k chunk - K / 32 steps x 2 row tiles x 2 images x 4 registers (192 at K = 384).
    //   D[n][m]: lane (m = l % 16, q) holds columns n = 16 ct + 4 q + r, r = 0..3, of rows 16 rt + m.
    constexpr int NJ = KS / 2, NJH = NJ / 2;    // 32-deep steps per unit / per half block
    const int l15 = lane & 15, q4 = lane >> 4;
    const uint32_t foff0 = (uint32_t)(l15 * 128 + ((q4 ^ ((l15 >> 1) & 7)) * 16));

    as_u32x4 Ah[2][NJ], Al[2][NJ];
    f32x4 acc[2][2];                            // [ct][rt]
    auto load_a = [&](int rb) __attribute__((always_inline)) {
        const int ln = as_lane();
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = min(rb * 256 + wave * 32 + 16 * rt + (ln & 15), d.M - 1);
            const bf16_t* ph = Ah_ + (int64_t)row * d.lda + (ln >> 4) * 8;
            const bf16_t* pl = Al_ + (int64_t)row * d.lda + (ln >> 4) * 8;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                Ah[rt][j] = *reinterpret_cast<const as_u32x4*>(ph + j * 32);
                Al[rt][j] = *reinterpret_cast<const as_u32x4*>(pl + j * 32);
            }
        }
        // the slice is COMPLETE before this (rare) branch rejoins the tick: left to itself hipcc parks its `s_waitcnt vmcnt(1) / (0)` for the last fragments at
        // the join, in front of every unit's first MFMA - where, on the common path, it waits for the DMA pieces and prefetch touches in flight instead
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(Ah[rt][j]), "+v"(Al[rt][j]));
    };
    // a unit's accumulators start from the bias: columns 16 ct + 4 q .. + 3, the same for both row tiles
    auto init_acc = [&](int cb) __attribute__((always_inline)) {
        const int ln = as_lane();
        const float* bp = bias_lds + cb * 32 + 4 * (ln >> 4);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float4 bv = *reinterpret_cast<const float4*>(bp + 16 * ct);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) { acc[ct][rt][0] = bv.x; acc[ct][rt][1] = bv.y; acc[ct][rt][2] = bv.z; acc[ct][rt][3] = bv.w; }
        }
        // four separate register tuples from here on (hipcc would keep ONE copy of the bias per column tile as the srcC of both row tiles' first MFMA, write
        // their results elsewhere - and spill an A fragment for the eight extra registers)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
    };
    // the MFMAs of a half block: per 32-deep step four fragment reads (w_hi, w_lo x two column tiles) and twelve MFMAs - w_hi a_lo, w_lo a_hi (the small terms
    // first), w_hi a_hi; the order is pinned: left alone under this register pressure hipcc reads one fragment, waits, multiplies
    // `dma`: this wave also issues the 2 GH pieces of half block tt + 3, behind the MFMAs (queued, they keep the pipe busy while the pieces issue)
    auto half_mma = [&](int half, uint32_t sbase, bool dma) __attribute__((always_inline)) {        // `half` is a compile-time constant at every call site
        const unsigned char* sb = lds + sbase;
        // w_hi feeds two of the three terms: two fragment sets, the reads of step j + 1 in flight under the MFMAs of step j.  w_lo feeds one term (the middle
        // four MFMAs): ONE set, re-read for step j + 1 right behind them - eight MFMAs ahead of its use.  24 fragment registers instead of 32.
        as_u32x4 whf[2][2], wlf[2];             // [set][ct], [ct]
        auto rd = [&](int j, int im) __attribute__((always_inline)) {
            uint32_t fo;
            asm volatile("v_xor_b32 %0, %2, %1" : "=v"(fo) : "v"(foff0), "n"((j & 1) * 64));      // volatile: not hoisted into live registers
            const unsigned char* gp = sb + (j >> 1) * 4096 + fo + im * (GH * 4096);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const as_u32x4 t = *reinterpret_cast<const as_u32x4*>(gp + ct * 2048);
                if (im == 0) whf[j & 1][ct] = t; else wlf[ct] = t;
            }
        };
        auto mma4 = [&](const as_u32x4 (&w)[2], bool alo, int kj) __attribute__((always_inline)) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const as_u32x4 av = alo ? Al[rt][kj] : Ah[rt][kj];
                    acc[ct][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[ct]), __builtin_bit_cast(bf16x8_t, av), acc[ct][rt], 0, 0, 0);
                }
        };
        rd(0, 0);
        rd(0, 1);
#pragma unroll
        for (int j = 0; j < NJH; ++j) {
            if (j + 1 < NJH) rd(j + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int kj = half * NJH + j;
            mma4(whf[j & 1], true, kj);            // w_hi a_lo
            mma4(wlf, false, kj);                  // w_lo a_hi
            __builtin_amdgcn_sched_barrier(0);
            if (j + 1 < NJH) rd(j + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma4(whf[j & 1], false, kj);           // w_hi a_hi
            __builtin_amdgcn_sched_barrier(0);
        }
        // the pieces of half block tt + 3 ride behind this half's MFMAs (interleaving one piece per step measured the same time and cost ~12 registers)
        if (dma) dma_next();
    };

    // ---- epilogue of unit (rb, cb), IN TWO HALVES (chunk pr = row tile pr of the wave: rows 16 pr + m): acc[ct][rt] register r = column 16 ct + 4 q + r of
    // row 16 rt + m.  v_permlane16_swap of (acc[0][rt], acc[1][rt]) trades the column-tile-1 registers of the even 16-lane rows for the column-tile-0 registers
    // of the odd ones: a lane then holds 8 CONSECUTIVE columns (q = 0, 1, 2, 3 -> columns 0, 16, 8, 24 .. + 7) of its rows m (chunk 0) and 16 + m (chunk 1).
    //   epi_half(0): at the END of the finisher's tick - row tile 0's swap and epilogue;
    //   epi_half(1): at the START of the next tick (the wave is the starter then: the partner has the matrix pipe anyway) - row tile 1, whose accumulators
    //   simply stay where they are until then: init_acc comes after it, no parking registers.
    // So every wave runs HALF an epilogue per tick, always beside the other wave's MFMAs: starter [epilogue half | MFMAs], finisher [MFMAs | epilogue half].
    // (One whole epilogue per unit - 3700 cycles with GELU + GELU' - was longer than the partner's MFMA phase: r06, fc1 272 us against 150 without epilogue.)
    // The bias is already in (the accumulators start from it).  The multiplier / residual loads of a half (2 x 16 bytes, L2 hits after prefetch_x) are issued
    // before its stores, and no store is waited for.  Addresses are wave-uniform base + 32-bit byte offset: one VGPR per stream instead of a pointer pair.
    // The half goes through a private LDS image of the wave ([16 rows][36 floats]: written as the lane holds it, read back ROW-MAJOR - lane -> (row l / 8 + 8
    // pass, columns 4 (l % 8) .. + 3)), because of what the memory pipeline does with a store (or load) instruction: it merges ADJACENT lanes into 64-byte requests.
    // Out of the registers adjacent lanes are different ROWS (6 KB apart): 64 requests of 16 bytes per instruction, and the store path of a CU saturates at
    // ~7 - 10 B / clk in that form (r06: every epilogue cost its bytes at ~5 TB/s ON TOP of the MFMA time, whatever was overlapped with what: fc1 150 -> 277 us).
    // Row-major, 8 lanes cover 128 contiguous bytes of a row.
    float* stage = reinterpret_cast<float*>(lds + AS_SLOTS * HALF_B + d.N * 4 + 8 * 256) + wave * (16 * 36);
    auto epi_half = [&](int pr, int rb, int cb) __attribute__((always_inline)) {      // pr is a compile-time constant at both call sites
        if constexpr ((EPI & AS_NOEPI) != 0) {
            asm volatile("" :: "v"(acc[0][pr]), "v"(acc[1][pr]));
            return;
        }
        const int ln = as_lane();
        {
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][pr][i]), __float_as_uint(acc[1][pr][i]), false, false);
                v[i] = __uint_as_float(sw[0]); v[4 + i] = __uint_as_float(sw[1]);
            }
            const int eq = ln >> 4;
            float* wp = stage + (ln & 15) * 36 + ((eq & 1) ? 12 + 4 * eq : 4 * eq);
            *reinterpret_cast<float4*>(wp) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(wp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        const uint32_t col = (uint32_t)(cb * 32 + (ln & 7) * 4);
        const uint32_t urow0 = (uint32_t)(((EPI & AS_FOLD) ? 0 : rb * 256) + wave * 32 + 16 * pr + (ln >> 3));       // pass p: row urow0 + 8 p
        // rows beyond M take no part at all: a load issued for them and never consumed would leave hipcc a pending register at the join, i.e. an
        // `s_waitcnt vmcnt(0)` in front of the next LDS read - which waits for every store and DMA piece this wave has in flight
        float4 xq[2];
        if constexpr ((EPI & (AS_MUL | AS_RES)) != 0) {
            const float* xs = (EPI & AS_MUL) ? d.mul : d.residual;
            const uint32_t ldx = (uint32_t)((EPI & AS_MUL) ? d.ldmul : d.ldr);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const uint32_t rr = min(urow0 + 8u * p, (uint32_t)d.M - 1u);
                xq[p] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xs) + (size_t)((rr * ldx + col) * 4u));
            }
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float4 w = *reinterpret_cast<const float4*>(stage + ((ln >> 3) + 8 * p) * 36 + (ln & 7) * 4);
            float v[4] = {w.x, w.y, w.z, w.w};
            const uint32_t urow = urow0 + 8u * p;
            if constexpr ((EPI & AS_MUL) != 0) { v[0] *= xq[p].x; v[1] *= xq[p].y; v[2] *= xq[p].z; v[3] *= xq[p].w; }
            if constexpr ((EPI & AS_RES) != 0) { v[0] += xq[p].x; v[1] += xq[p].y; v[2] += xq[p].z; v[3] += xq[p].w; }
            const bool live = urow < (uint32_t)d.M;
            if constexpr ((EPI & AS_GELU) != 0) {
                float gd[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) gelu_and_grad(v[k], v[k], gd[k]);
                if (d.aux && live) as_store16<(EPI & AS_SC1) != 0>(d.aux, (urow * (uint32_t)d.ldaux + col) * 4u, as_f32x4{gd[0], gd[1], gd[2], gd[3]});
            }
            if (live) {
                if constexpr (PLANES) {
                    const uint32_t h0 = pack_bf2(v[0], v[1]), h1 = pack_bf2(v[2], v[3]);
                    const uint32_t l0 = pack_bf2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                    const uint32_t l1 = pack_bf2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                    const uint32_t co = (urow * (uint32_t)d.ldc + col) * 2u;
                    as_store8<(EPI & AS_SC1) != 0>(d.c, co, as_u32x2{h0, h1});
                    as_store8<(EPI & AS_SC1) != 0>(d.c_lo, co, as_u32x2{l0, l1});
                } else {
                    as_store16<(EPI & AS_SC1) != 0>(d.c, (urow * (uint32_t)d.ldc + col) * 4u, as_f32x4{v[0], v[1], v[2], v[3]});
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto now = [&]() __attribute__((always_inline)) -> unsigned long long {
        if constexpr (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return __builtin_amdgcn_s_memtime(); }
        return 0ull;
    };
    const unsigned long long t_start = now();

    // ---- prologue: the first four half blocks on their way
    const int nhb = 2 * nU;
    if (grp == 0 && (EPI & AS_NODMA) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < nhb) dma_next();
    }
    int c_rb = rb0, c_cb = cb0, c_left = nU;       // the unit this wave multiplies; units left including it
    int e_rb = rb0, e_cb = cb0;                    // the unit whose second epilogue half is still in the accumulators
    int cur_rb = -1;
    // vmcnt bookkeeping of waves 0..3: S1 / S2 / S3 = store instructions this wave issued one / two / three ticks back, F1.. = it was the finisher then (a
    // finisher's stores FOLLOW its DMA batch, a starter's precede it).  Behind the batch of tick tt - 3 came: S3 if F3, S2 + a batch, S1 + a batch.
    int S1 = 0, S2 = 0, S3 = 0;
    bool F1 = false, F2 = false, F3 = false;
    // store instructions of one epilogue half of a wave whose 16 rows are ALL live (counted only then: an under-count merely waits a little longer)
    const int s_half = (EPI & AS_NOEPI) ? 0 : (PLANES ? 4 : 2) + (((EPI & AS_GELU) != 0 && d.aux) ? 2 : 0);
    const int nticks = nhb + 2;                    // group 1 runs one tick behind group 0; a wave's last epilogue half is the tick after its last MFMAs
    for (int tt = 0; tt < nticks; ++tt) {
        const unsigned long long t0 = now();
        // Opening of tick tt.  Waves 0..3 (the DMA issuers): half block tt's pieces were issued three ticks ago and two batches (4 GH pieces) since - with at
        // most that many operations in flight they have landed, while the newest pieces and the epilogue's stores stay in flight across the barrier; once the
        // stream has run out (tt + 2 >= nhb) everything is waited for.  Waves 4..7 issue no pieces: nothing of theirs has to land before the barrier.  After the
        // barrier half block tt is readable by group 0, tt - 1 by group 1, and the slot of half block tt - 2 (last read in tick tt - 1) is free for tt + 3.
        if (grp == 0) {
            if (tt + 2 < nhb) {
                // in flight at most: the two newest batches (4 GH pieces) + every store issued behind the batch of tick tt - 3 (vmcnt retires in order: counting
                // the stores in keeps them in flight for ~3 ticks instead of ~1.5 - the HBM write stream needs that window; r06: fc1 150 us without epilogue,
                // 210 with its stores folded onto cache-resident rows, 279 real)
                const int extra = S1 + S2 + (F3 ? S3 : 0);
                switch (extra >> 1) {
                    case 0: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
                    case 1: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
                    case 2: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
                    case 3: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory"); break;
                    case 4: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory"); break;
                    case 5: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory"); break;
                    case 6: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory"); break;
                    case 7: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(26) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(22) lgkmcnt(0)" ::: "memory"); break;
                    case 8: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(28) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory"); break;
                    default: if constexpr (GH == 3) asm volatile("s_waitcnt vmcnt(30) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(26) lgkmcnt(0)" ::: "memory"); break;
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        S3 = S2; F3 = F2; S2 = S1; F2 = F1; S1 = 0; F1 = false;          // tick tt's own counts are filled in below
        __builtin_amdgcn_s_barrier();
        const unsigned long long t1 = now();
        tsum[0] += t1 - t0; tsum[7] += 1;
        const int lt = tt - grp;                   // this wave's own half-block counter
        // (one home for the accumulators per tick: without this pin hipcc splits their live ranges over the two role branches - MFMAs with vdst != srcC, copies,
        // and a spilled A fragment to pay for them)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        if (lt < 0 || lt > nhb) continue;
        const bool dma = (EPI & AS_NODMA) == 0 && grp == 0 && tt + 3 < nhb;
        if ((lt & 1) == 0) {
            // STARTER of this tick: the parked epilogue half of the unit it finished last tick FIRST (the partner wave of the SIMD, at raised priority, has the
            // matrix pipe), then the first half of the next unit
            if (lt > 0) {
                epi_half(1, e_rb, e_cb);
                if (e_rb * 256 + wave * 32 + 32 <= d.M) S1 += s_half;
            }
            const unsigned long long t2 = now();
            tsum[1] += t2 - t1;
            if (lt == nhb) continue;               // that was this wave's last unit
            prefetch_x(c_rb, c_cb);
            if (c_cb == CB - 1 && c_left > 1) prefetch_a(c_rb + 1);
            if (c_rb != cur_rb) { load_a(c_rb); cur_rb = c_rb; }
            init_acc(c_cb);
            const unsigned long long t4 = now();
            half_mma(0, (uint32_t)((lt % AS_SLOTS) * HALF_B), dma);
            const unsigned long long t5 = now();
            tsum[3] += t4 - t2; tsum[4] += t5 - t4;
        } else {
            // FINISHER: second half at RAISED priority (its MFMAs go first), then the unit's first epilogue half under the rest of the partner's MFMAs; the
            // second half waits in the accumulators for the next tick.  The stores stay in flight across the barrier.
            if constexpr ((EPI & AS_NOPRIO) == 0) __builtin_amdgcn_s_setprio(1);
            half_mma(1, (uint32_t)((lt % AS_SLOTS) * HALF_B), dma);
            if constexpr ((EPI & AS_NOPRIO) == 0) __builtin_amdgcn_s_setprio(0);
            const unsigned long long t5 = now();
            epi_half(0, c_rb, c_cb);
            if (c_rb * 256 + wave * 32 + 16 <= d.M) S1 += s_half;
            F1 = true;
            e_rb = c_rb; e_cb = c_cb;
            if (++c_cb == CB) { c_cb = 0; ++c_rb; }
            --c_left;
            tsum[5] += t5 - t1;
            if constexpr (DBG) tsum[2] += now() - t5;
        }
    }
    if constexpr (DBG) {
        tsum[6] = now() - t_start;
        if (g.dbg && lane == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) g.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + k] = tsum[k];
        }
    }
}

template <int KS, int EPI, bool DBG>
int as_launch1(const AsArgs& g, int nwg, hipStream_t s) {
    const size_t LDS = AS_SLOTS * ((KS / 8) * 2 * 4096) + (size_t)g.d.N * 4 + 8 * 256 + 8 * 16 * 36 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_x3_as_kernel<KS, EPI, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_x3_as_kernel<KS, EPI, DBG>), dim3(nwg), dim3(512), LDS, s, g);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

// var: 0 the kernel, 1 its instrumented twin (s_memtime sums), 2 without epilogue, 3 without epilogue and weight stream (measurement)
template <int KS, int EPI>
int as_launch(const AsArgs& g, int var, int nwg, hipStream_t s) {
    if (var == 1) return as_launch1<KS, EPI | AS_SC1, true>(g, nwg, s);
    if (var == 2) return as_launch1<KS, EPI | AS_NOEPI, false>(g, nwg, s);
    if (var == 3) return as_launch1<KS, EPI | AS_NOEPI | AS_NODMA, false>(g, nwg, s);
    if (var == 4) return as_launch1<KS, EPI | AS_FOLD, false>(g, nwg, s);
    if (var == 5) return as_launch1<KS, EPI, false>(g, nwg, s);
    if (var == 6) return as_launch1<KS, EPI | AS_SC1 | AS_NOPRIO, false>(g, nwg, s);
    return as_launch1<KS, EPI | AS_SC1, false>(g, nwg, s);
}

int as_epi_of(const p3_gemm_x3_desc* d) {
    return (d->c_lo ? AS_PLANES : 0) | (d->act == P3_ACT_GELU ? AS_GELU : 0) | (d->mul ? AS_MUL : 0) | (d->residual ? AS_RES : 0);
}
bool as_epi_built(int epi) {
    return epi == 0 || epi == AS_RES || epi == AS_PLANES || epi == (AS_PLANES | AS_GELU) || epi == (AS_PLANES | AS_MUL) || epi == AS_MUL || epi == AS_GELU;
}

template <int KS>
int as_dispatch(const AsArgs& g, int epi, int var, int nwg, hipStream_t s) {
    switch (epi) {
        case 0: return as_launch<KS, 0>(g, var, nwg, s);
        case AS_RES: return as_launch<KS, AS_RES>(g, var, nwg, s);
        case AS_MUL: return as_launch<KS, AS_MUL>(g, var, nwg, s);
        case AS_GELU: return as_launch<KS, AS_GELU>(g, var, nwg, s);
        case AS_PLANES: return as_launch<KS, AS_PLANES>(g, var, nwg, s);
        case AS_PLANES | AS_GELU: return as_launch<KS, AS_PLANES | AS_GELU>(g, var, nwg, s);
        case AS_PLANES | AS_MUL: return as_launch<KS, AS_PLANES | AS_MUL>(g, var, nwg, s);
        default: break;
    }
    p3_set_error("p3_gemm_x3: epilogue combination not built for the A-stationary kernel");
    return P3_EUNSUP;
}

}  // namespace

// eligibility of the A-stationary kernel (the caller, p3_gemm_x3, has done the alignment checks).  No condition on M: the kernel choice must not depend on the
// batch (a tile run alone gives the bits it gives inside a batch of 64 - tests/test_model_gpu.py::test_full_bench_batch_is_batch_independent...)
// the default rule leaves the planes + multiplier epilogue (dX of fc2) on the tile kernels: its epilogue LOADS 309 MB, every load wait drains the wave's stores
// (vmcnt retires in order), and in the step it measured 291 us against the tile kernel's 281 (profiles/r06_fp32x3_step_summary_mid.txt)
bool p3_gemm_x3_as_default(const p3_gemm_x3_desc* d) { return d->N >= 1024 && !(d->mul && d->c_lo); }

bool p3_gemm_x3_as_ok(const p3_gemm_x3_desc* d) {
    return (d->K == 384 || d->K == 256) && d->N % 32 == 0 && d->N <= 4096 && !d->ln_gamma && as_epi_built(as_epi_of(d)) &&
           (int64_t)d->N * d->ldb * 2 < (1ll << 31) && (!d->bias || (uintptr_t)d->bias % 16 == 0) &&
           (int64_t)d->M * d->ldc * 4 < (1ll << 32) && (!d->aux || (int64_t)d->M * d->ldaux * 4 < (1ll << 32)) &&
           (!d->mul || (int64_t)d->M * d->ldmul * 4 < (1ll << 32)) && (!d->residual || (int64_t)d->M * d->ldr * 4 < (1ll << 32));
}

static unsigned long long* g_as_dbg = nullptr;
// measurement hook (tools/mb_as.py): device buffer [256 workgroups][8 waves][8] for the cycle sums of the instrumented twin (P3_AS_VAR=1)
extern "C" int p3_gemm_x3_as_debug(void* buf) { g_as_dbg = (unsigned long long*)buf; return P3_OK; }

int p3_gemm_x3_as(const p3_gemm_x3_desc* d, hipStream_t s) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { p3_set_error("p3_gemm_x3: device query failed"); return P3_EINVAL; }
        n_cu = prop.multiProcessorCount;
    }
    AsArgs g;
    g.d = *d;
    g.CB = d->N / 32;
    g.units = p3_ceil_div(d->M, 256) * g.CB;
    g.dbg = g_as_dbg;
    const int nwg = g.units < n_cu ? g.units : n_cu;
    const int epi = as_epi_of(d);
    static int var = -1;
    if (var < 0) { const char* e = getenv("P3_AS_VAR"); var = e ? atoi(e) : 0; }
    if (p3_tracing()) {                               // the instantiation as rocprofv3 spells it (bench.py matches its PMC table by this name)
        char nm[64];
        snprintf(nm, sizeof(nm), "gemm_x3_as_kernel<%d, %d, false>", d->K / 16, epi | (var == 0 ? AS_SC1 : 0));
        p3_note_kernel(nm);
    }
    return d->K == 384 ? as_dispatch<24>(g, epi, var, nwg, s) : as_dispatch<16>(g, epi, var, nwg, s);
}
