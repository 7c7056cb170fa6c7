// p3hip weight-gradient GEMM, LDS-DMA form:  C[N,K] += A[M,N]^T * B[M,K]  (bf16 operands, fp32 accumulate, split over M)
//
// The register-staged kernel of gemm_tn.hip (128 x 128 tile, 4 waves, two workgroups per CU) pays per 64-row step and workgroup 32 KB of
// ds_write_b128 at the LDS store rate (~13 cycles per wave-instruction: ~830 cycles per CU and step pair) next to 512 cycles of transposing
// reads - the LDS pipe, not the MFMA pipe (1024 cycles), is its busiest unit - and ends in one fp32 atomic per accumulator of EVERY resident
// workgroup (448 x 16 K = 7.3 M per launch whose lines migrate between the XCDs' L2s: 15 - 20 us of a 60 - 80 us launch, r03).  Here:
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  The LDS image of a step is the
//     operand tile as it lies in memory, [64 rows of m][128 columns] bf16 = 256-byte rows, UNPADDED; the 64-byte granule of a row is XOR-swizzled
//     with (row & 3) on the SOURCE address (the destination of an LDS-DMA is lane-linear), so that the four rows a transposing read
//     (ds_read_b64_tr_b16: per 16 lanes, 4 rows x 32 bytes) touches lie on four different quarters of the 64 banks;
//   * ONE 512-thread workgroup per CU = two groups of four waves that work on the SAME output tile: group g multiplies rows 32 g .. 32 g + 31 of
//     every step; at the end the groups exchange halves of their accumulators through LDS, so a CU issues 16 K atomics instead of 32 K at the
//     same number of resident waves (the M split count halves);
//   * NBUF steps in LDS (32 KB each), NBUF - 1 in flight, one barrier and one counted vmcnt per step.
// Shapes: M % 64 == 0, N % 128 == 0, K % 128 == 0 (every plain-operand weight gradient of the path at its bench batch); anything else, the fp32
// parity mode and the generated-B forms stay on gemm_tn.hip.  The arithmetic per output element is the same fp32 MFMA accumulation in ascending m
// inside a split; the split boundaries differ from the register-staged kernel's, so results agree to fp32 summation order, not bit for bit.
#include <stdlib.h>

#include <type_traits>

#include "p3_common.h"

namespace {

constexpr int TD_BM = 64;                       // rows of m per step
constexpr int TD_STEP_BYTES = 2 * TD_BM * 256;  // A image + B image

struct TdArgs {
    const bf16_t* A; const bf16_t* B; float* C;
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
__global__ __launch_bounds__(512, 2) void gemm_tn_dma_kernel(TdArgs g) {
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

    // ---- LDS-DMA: a step = 16 + 16 pieces of 1 KB (4 rows x 256 B); wave w issues pieces 2w, 2w + 1 of A and of B.
    // lane -> (row = lane / 16, slot = lane % 16) of a piece, source chunk = slot ^ ((row & 3) << 2): 64-byte granule swizzle
    const uint32_t lds_addr = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_raw);
    uint32_t voffA[2], voffB[2];
    {
        const int prow = lane >> 4, slot = lane & 15, chunk = slot ^ (prow << 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wave * 2 + q) * 4 + prow;          // row inside the step
            voffA[q] = (uint32_t)(((int64_t)r * g.lda + tn * 128 + chunk * 8) * 2);
            voffB[q] = (uint32_t)(((int64_t)r * g.ldb + tk * 128 + chunk * 8) * 2);
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
        const uint32_t da = lds_addr + (uint32_t)(buf * TD_STEP_BYTES + wave * 2048);
        dma2(g.A + m0 * g.lda, da, voffA[0], voffA[1]);
        dma2(g.B + m0 * g.ldb, da + TD_BM * 256, voffB[0], voffB[1]);
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
        offA[i] = lds_addr + (uint32_t)((grp * 32 + lrow) * 256 + (((wm * 2 + i) ^ (li >> 2)) * 64) + lin);
        offB[i] = lds_addr + (uint32_t)(TD_BM * 256 + (grp * 32 + lrow) * 256 + (((wn * 2 + i) ^ (li >> 2)) * 64) + lin);
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
    // (r04, measured and dropped: a software pipeline over the steps - the fragments of step s + 1 read from LDS while the MFMAs of step s run
    // from a second fragment register set, 240 VGPRs - ran 8 - 11 % SLOWER on every shape: fc1 94 vs 85 us, qkv 71 vs 64)
    for (int st = 0; st < nsteps; ++st) {
        // RAW: this wave's pieces of step st have landed once at most (LA - 1) younger steps stay in flight (loads retire in order; nothing else
        // is outstanding); the barrier extends that to every wave's pieces.  WAR: a wave reaches the barrier after its reads of step st - 1, whose
        // buffer step st + LA takes.
        const int ahead = min(LA - 1, nsteps - 1 - st);
        if (ahead <= 0) td_wait_vm<0>();
        else if (ahead == 1) td_wait_vm<4>();
        else if (ahead == 2) td_wait_vm<8>();
        else td_wait_vm<12>();
        __builtin_amdgcn_s_barrier();
        if (st + LA < nsteps) stage(st + LA);
        const uint32_t bo = (uint32_t)((st % NBUF) * TD_STEP_BYTES);
        u32x2_t fa[2][2][2], fb[2][2][2];            // [kk][32-column block][rows +0..3 | +4..7]
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t ro = bo + (uint32_t)((kk * 16 + hh * 4) * 256);
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[kk][i][hh]) : "v"(offA[i] + ro));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[kk][i][hh]) : "v"(offB[i] + ro));
                }
        if (do_cs) {
            const unsigned char* ab = lds_raw + bo;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int r = cs_row + 32 * h2;
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(ab + r * 256 + ((cs_chunk ^ ((r & 3) << 2)) * 16));
#pragma unroll
                for (int q = 0; q < 4; ++q) { csum[2 * q] += __uint_as_float(v[q] << 16); csum[2 * q + 1] += __uint_as_float(v[q] & 0xffff0000u); }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]),
                       "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1]));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fa[kk][i][0].x, fa[kk][i][0].y, fa[kk][i][1].x, fa[kk][i][1].y});
                bf[i] = __builtin_bit_cast(bf16x8_t, u32x4_t{fb[kk][i][0].x, fb[kk][i][0].y, fb[kk][i][1].x, fb[kk][i][1].y});
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
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

}  // namespace

// C[n,k] += sum_s slabs[s][n][k] (float64, split order) - gemm_tn.hip
void p3_tn_reduce_launch(const float* slabs, float* C, int N, int K, int ldc, int splits, hipStream_t s);
float* p3_tn_park(float* C, int N, int K, int ldc, int splits);      // gemm_tn.hip: deferred reduce slot or NULL

// returns P3_OK when the launch was taken, 1 when the shape / mode is not this kernel's (the caller then runs gemm_tn.hip's kernel)
int p3_gemm_tn_dma_try(const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, float* colsum, float* slabs, int max_slabs,
                       hipStream_t s) {
    static int on = -1;                              // P3_TN_DMA=0: every weight gradient on gemm_tn.hip's register-staged kernel (A/B: profiles/r04_mb_tn.txt)
    if (on < 0) { const char* e = getenv("P3_TN_DMA"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || M % TD_BM != 0 || N % 128 != 0 || K % 128 != 0 || lda % 8 != 0 || ldb % 8 != 0) return 1;
    if (((uintptr_t)A % 16) != 0 || ((uintptr_t)B % 16) != 0) return 1;
    if ((int64_t)TD_BM * lda * 2 + 256 >= (1ll << 31) || (int64_t)TD_BM * ldb * 2 + 256 >= (1ll << 31)) return 1;     // 32-bit DMA offsets inside a step
    TdArgs g;
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.colsum = colsum;
    const int tiles_n = N / 128;
    g.tiles_k = K / 128;
    const int tiles = tiles_n * g.tiles_k;
    if (tiles > 256) return 1;
    // one workgroup per CU: splits = 256 / tiles (the r03 finding for the register-staged kernel - the best grids fill ONE resident round -
    // holds here by construction); never fewer than 2 steps per split
    int splits = 256 / tiles;
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
    int cs_parked = 0;
    g.cs_slab = colsum ? p3_colsum_parts(splits, N, colsum, P3_BF16, &cs_parked) : nullptr;
    constexpr int NBUF = 4;                          // three steps in flight (NBUF = 3 measured 2 % slower, profiles/r04_mb_tn.txt)
    const size_t lds = (size_t)NBUF * TD_STEP_BYTES;       // >= the 64 KB the fold needs
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    dim3 grid(tiles * splits), block(512);
    hipLaunchKernelGGL(gemm_tn_dma_kernel<NBUF>, grid, block, lds, s, g);
    if (p3_tracing()) p3_note_kernel("gemm_tn_dma_kernel<4>");
    if (g.slabs && !parked) p3_tn_reduce_launch(g.slabs, C, N, K, ldc, splits, s);
    P3_LAUNCH_CHECK();
    if (g.cs_slab && !cs_parked) return p3_det_reduce(g.cs_slab, splits, N, colsum, N, 1, s);
    return P3_OK;
}
