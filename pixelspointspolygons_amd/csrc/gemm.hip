// p3hip GEMM:  C[M,N] = act(A'[M,K] * W[N,K]^T + bias) + residual     (gfx950 MFMA, LDS-tiled)
//
// One 256-thread workgroup (4 waves, 2x2) computes a 128x128 tile; every wave owns 64x64 = 2x2
// MFMA 32x32 accumulators (64 AGPR/VGPR per lane).  Operands are staged global -> VGPR -> LDS with
// the next K-slice's global loads issued before the MFMAs of the current slice (register-staged
// double buffering, guide T14 form), two LDS buffers, one barrier per K-slice.
//   bf16: v_mfma_f32_32x32x16_bf16, BK = 64, LDS rows padded to 72 elements (ds_read_b128, conflict free)
//   f32 : v_mfma_f32_32x32x2_f32 (exact fp32, = fmaf chain), BK = 16, LDS k-major [16][132]
// The A operand can be generated on the fly (implicit 3x3 conv gather, BN+ReLU fold, ScoreNet pair sum).
#include <stdlib.h>

#include <type_traits>

#include <stdio.h>

#include "p3_common.h"

namespace {

struct GemmArgs {
    const void* A; const void* W; void* C;
    p3_gemm_desc d;
    int tiles_m, tiles_n;
    int vec_epi;   // 16-byte epilogue accesses are legal (strides / base pointers aligned)
    float* stat_slab;   // deterministic mode: [gridDim.x / tiles_n][2 row halves][2][N] partials of (colsum, colsumsq) instead of atomics
    float* tile_stats;  // non-STATS launch that still owes BatchNorm column sums: [tiles_m][2 row halves][2][N] per-tile partials (see launch_bk)
    int split;          // the caller's dtype_in was P3_F32X3: fp32 operands, products as bf16 x 3
};

constexpr int BM = 128, BN = 128;

// Two bf16 K-slice depths.  BK = 32 (rows padded to 40 elements = 80 B, conflict free for ds_read_b128; 40 KB operand buffers +
// two-pass 34 KB epilogue staging -> more workgroups per CU) is the default: the GEMMs of this path are short-K and HBM / latency
// bound.  BK = 64 (74 KB, 2 workgroups / CU, half the barriers per FLOP, one-pass epilogue) takes over from K = 2048 (decoder linear2,
// the 3x3 convolutions).  Same-box A/B of the whole train step (tools/ab.sh, r01): all-64 61.4 ms, all-32 60.1, 64 above K = 512
// 61.1, 64 from K = 2048 60.0.
template <typename T, int BKSEL> struct Tr;
template <int BKSEL> struct Tr<bf16_t, BKSEL> { static constexpr int BK = BKSEL, PITCH = BKSEL + 8, VEC = 8, LDS_ELEMS = BM * (BKSEL + 8); };
template <int BKSEL> struct Tr<float, BKSEL> { static constexpr int BK = 16, PITCH = 132, VEC = 4, LDS_ELEMS = 16 * 132; };
// BKSEL == BK_SPLIT (T = float): fp32 operands multiplied as bf16 x 3 (r04, p3_set_gemm_split): a value x is split into hi = bf16(x) and lo = bf16(x - hi)
// when its slice is stored to LDS (two bf16 images [128 rows][32 k], 80-byte rows: conflict-free 16-byte fragment reads) and the product is
// a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA (2^-17 relative per product; the dropped a_lo b_lo term is 2^-18) - 16 x the fp32 MFMA rate for 3 x the issues.
constexpr int BK_SPLIT = 17, SPLIT_BK = 32, SPLIT_PITCH_B = (SPLIT_BK + 8) * 2, SPLIT_IMG_B = 128 * SPLIT_PITCH_B;      // 80-byte rows: 8 consecutive rows cover all 32 banks
template <> struct Tr<float, BK_SPLIT> { static constexpr int BK = SPLIT_BK, PITCH = SPLIT_BK + 8, VEC = 4, LDS_ELEMS = 2 * SPLIT_IMG_B / 4; };
// BKSEL == BK_SPLIT_WPL: the split form with the WEIGHT given as planes (p3_gemm_desc.w_lo): the W slice's hi / lo images are copied as they are (16-byte loads,
// ds_write_b128; see ld_w / put_w) - half of the loop's split arithmetic (3 VALU ops per staged value) gone
constexpr int BK_SPLIT_WPL = 18;
template <> struct Tr<float, BK_SPLIT_WPL> : Tr<float, BK_SPLIT> {};
template <typename T> struct VecOf { static constexpr int VEC = 16 / (int)sizeof(T); };

// ---- per-thread A-row descriptor (fixed over the K loop) -------------------------------------
struct RowSrc {
    int64_t off;   // element offset of the row start (plain / affine) or U row (pair)
    int64_t off2;  // pair: V row offset
    int y, x;      // conv: pixel coordinates
    int64_t img;   // conv: element offset of the image (b*H*W*lda)
};

template <int AMODE>
__device__ __forceinline__ RowSrc make_row(const p3_gemm_desc& d, int gm) {
    RowSrc r; r.off = 0; r.off2 = 0; r.y = 0; r.x = 0; r.img = 0;
    if (gm >= d.M) gm = d.M - 1;
    if (AMODE == P3_A_CONV3X3 || AMODE == P3_A_CONV3X3_AFFINE_RELU) {
        int hw = d.conv_H * d.conv_W;
        int b = gm / hw, p = gm - b * hw;
        r.y = p / d.conv_W; r.x = p - r.y * d.conv_W;
        // zero-bordered source: image stride (H+2)(W+2) rows, origin moved to the first interior pixel
        r.img = d.conv_pad ? ((int64_t)b * (d.conv_H + 2) * (d.conv_W + 2) + (d.conv_W + 2) + 1) * d.lda : (int64_t)b * hw * d.lda;
    } else if (AMODE == P3_A_PAIR_AFFINE_RELU) {
        int n = d.pair_n, nn = n * n;
        int b = gm / nn, p = gm - b * nn;
        int i = p / n, j = p - i * n;
        r.off = (int64_t)(b * n + i) * d.lda;
        r.off2 = (int64_t)(b * n + j) * d.lda;
    } else {
        r.off = (int64_t)gm * d.lda;
    }
    return r;
}

// ---- 16-byte staged vectors: 8 bf16 or 4 f32 -------------------------------------------------
template <typename T>
__device__ __forceinline__ void unpack(const uint4& raw, float (&v)[VecOf<T>::VEC]) {
    if constexpr (sizeof(T) == 2) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    } else {
        v[0] = __uint_as_float(raw.x); v[1] = __uint_as_float(raw.y); v[2] = __uint_as_float(raw.z); v[3] = __uint_as_float(raw.w);
    }
}
template <typename T>
__device__ __forceinline__ uint4 repack(const float (&v)[VecOf<T>::VEC]) {
    uint4 p;
    if constexpr (sizeof(T) == 2) {
        p.x = pack_bf2(v[0], v[1]); p.y = pack_bf2(v[2], v[3]); p.z = pack_bf2(v[4], v[5]); p.w = pack_bf2(v[6], v[7]);
    } else {
        p.x = __float_as_uint(v[0]); p.y = __float_as_uint(v[1]); p.z = __float_as_uint(v[2]); p.w = __float_as_uint(v[3]);
    }
    return p;
}

// load VEC consecutive k-elements of one A row starting at global k index `k`
template <typename T, int AMODE>
__device__ __forceinline__ uint4 load_a(const p3_gemm_desc& d, const T* A, const RowSrc& r, int k) {
    constexpr int VEC = VecOf<T>::VEC;
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (AMODE == P3_A_CONV3X3 || AMODE == P3_A_CONV3X3_AFFINE_RELU) {
        int tap = k / d.conv_C, c = k - tap * d.conv_C;
        int yy = r.y + tap / 3 - 1, xx = r.x + tap % 3 - 1;
        if (d.conv_pad) return *reinterpret_cast<const uint4*>(A + r.img + (int64_t)(yy * (d.conv_W + 2) + xx) * d.lda + c);
        if ((yy >= 0) && (yy < d.conv_H) && (xx >= 0) && (xx < d.conv_W)) {
            raw = *reinterpret_cast<const uint4*>(A + r.img + (int64_t)(yy * d.conv_W + xx) * d.lda + c);
            if (AMODE == P3_A_CONV3X3_AFFINE_RELU) {   // BN + ReLU of the producer folded into the gather; zero padding stays zero
                float v[VEC];
                unpack<T>(raw, v);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[i] = fmaxf(v[i] * d.a_scale[c + i] + d.a_shift[c + i], 0.f);
                raw = repack<T>(v);
            }
        }
        return raw;
    }
    return *reinterpret_cast<const uint4*>(A + r.off + k);     // AFFINE_RELU / PAIR_AFFINE_RELU: raw row; xform_a() runs at LDS-store time
}

// generated A operand (ScoreNet conv2 / conv3: BN + ReLU of the producer [over the pair grid] folded into the operand): the raw U row
// (and V row) are loaded like any operand; relu(scale * (u [+ v]) + shift) is applied when the slice is stored to LDS, with scale / shift
// read from LDS.  (r02 applied it at load time: every load was consumed at once - s_waitcnt vmcnt(0) inside the step, no prefetch.)
template <typename T, int AMODE>
__device__ __forceinline__ uint4 xform_a(const uint4& raw, const uint4& raw2, const float* sc, const float* sh, int k) {
    constexpr int VEC = VecOf<T>::VEC;
    float v[VEC];
    unpack<T>(raw, v);
    if (AMODE == P3_A_PAIR_AFFINE_RELU) {
        float v2[VEC];
        unpack<T>(raw2, v2);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] += v2[i];
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = fmaxf(v[i] * sc[k + i] + sh[k + i], 0.f);
    return repack<T>(v);
}

// act'(.) for the fused activation backward: GELU' of the saved pre-activation, ReLU' from the saved output
__device__ __forceinline__ float act_grad(float x, int act) {
    if (act == P3_ACT_MUL) return x;                 // the forward already stored act'(pre)
    if (act == P3_ACT_GELU) { float h, g; gelu_and_grad(x, h, g); return g; }
    return x > 0.f ? 1.f : 0.f;
}

template <typename T>
__device__ __forceinline__ uint4 load_w(const T* W, int64_t rowoff, int k) {
    return *reinterpret_cast<const uint4*>(W + rowoff + k);
}

// store one thread's staged vector into LDS.  bf16: row-major [row][PITCH]; f32: k-major [k][PITCH]
template <typename T, int PITCH>
__device__ __forceinline__ void lds_put(T* buf, int row, int kq, const uint4& v) {
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(buf + row * PITCH + kq) = v;
    } else {
        buf[(kq + 0) * PITCH + row] = __uint_as_float(v.x);
        buf[(kq + 1) * PITCH + row] = __uint_as_float(v.y);
        buf[(kq + 2) * PITCH + row] = __uint_as_float(v.z);
        buf[(kq + 3) * PITCH + row] = __uint_as_float(v.w);
    }
}

// split form: 4 consecutive k of one row -> 8 bytes into the hi image and 8 into the lo image
__device__ __forceinline__ void lds_put_split(float* buf, int row, int kq, const uint4& v) {
    const float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
    // hi by one v_cvt_pk_bf16_f32 per PAIR, its halves back as fp32 by a shift / a mask, lo = pack(x - hi): 3 VALU ops per value (r06; the form that rounded every
    // value on its own and packed the rounded floats again cost 4 - the same bits)
    const uint32_t h0 = pack_bf2(x[0], x[1]), h1 = pack_bf2(x[2], x[3]);
    const uint32_t l0 = pack_bf2(x[0] - __uint_as_float(h0 << 16), x[1] - __uint_as_float(h0 & 0xffff0000u));
    const uint32_t l1 = pack_bf2(x[2] - __uint_as_float(h1 << 16), x[3] - __uint_as_float(h1 & 0xffff0000u));
    unsigned char* b = reinterpret_cast<unsigned char*>(buf) + row * SPLIT_PITCH_B + kq * 2;
    *reinterpret_cast<uint2*>(b) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(b + SPLIT_IMG_B) = make_uint2(l0, l1);
}

// STATS = true (BatchNorm column sums requested): persistent workgroups walk the tiles vb, vb + gridDim.x, ... of ONE tile column
// and keep the per-column sum / sum-of-squares in registers across them, so each column gets gridDim.x / tiles_n atomics instead of
// one per 128-row tile: the tall-skinny ScoreNet / FFL GEMMs have 18 000 - 25 000 row tiles, and 37 000 same-address atomics are a
// 0.4 ms serial chain (r01: conv3 of the ScoreNet 906 us for 39 GFLOP).
template <typename T, typename TO, int AMODE, int BKSEL, bool STATS>
__global__ __launch_bounds__(256, (BKSEL == 32 && !STATS && AMODE != P3_A_PAIR_AFFINE_RELU ? 3 : 2)) void gemm_kernel(GemmArgs g) {
    using TR = Tr<T, BKSEL>;
    constexpr int BK = TR::BK, PITCH = TR::PITCH, VEC = TR::VEC, LDSE = TR::LDS_ELEMS;
    constexpr int EPI_PASSES = BKSEL == 32 ? 2 : 1;  // epilogue staged through LDS in 1 pass of 128 rows or 2 passes of 64
    constexpr int EPI_ROWS = BM / EPI_PASSES;
    // rows handled per thread per operand per stage
    constexpr int ROWS_PER_PASS = 256 / (BK / VEC);  // bf16: 32, f32: 64
    constexpr int NPASS = BM / ROWS_PER_PASS;        // bf16: 4,  f32: 2
    constexpr int LDS_BYTES = (4 * LDSE * (int)sizeof(T) > EPI_ROWS * 132 * 4) ? 4 * LDSE * (int)sizeof(T) : EPI_ROWS * 132 * 4;
    __shared__ __attribute__((aligned(16))) T lds[LDS_BYTES / sizeof(T)];

    const p3_gemm_desc& d = g.d;
    const T* A = reinterpret_cast<const T*>(g.A);
    const T* W = reinterpret_cast<const T*>(g.W);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, hi = lane >> 5;
    const int kq = (tid % (BK / VEC)) * VEC;
    const int r0 = tid / (BK / VEC);
    const int ntiles = g.tiles_m * g.tiles_n;
    float cs1[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};           // STATS: column sums of this workgroup's tiles (one tile column)
    int tn_stats = 0;
    // -DP3_GEMM_TIMING (diagnostic build, tools/mb_gemm_stages.py): workgroup 0 stores 100 MHz timestamps of a plain tile's stages through
    // the otherwise unused pair_V pointer
#ifdef P3_GEMM_TIMING
#define GT(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && g.d.pair_V) reinterpret_cast<long long*>(const_cast<void*>(g.d.pair_V))[k] = wall_clock64(); } while (0)
#else
#define GT(k) do { } while (0)
#endif
    GT(0);
  int vb = blockIdx.x;
  do {
    // plain mode: one tile per workgroup, XCD-aware order (consecutive logical tiles = same A row-panel = one XCD's L2);
    // STATS mode: gridDim.x is a multiple of tiles_n, so tn stays fixed along a workgroup's walk
    const int bid = STATS ? vb : xcd_remap(vb, ntiles);
    int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    tn_stats = tn;
    RowSrc arow[NPASS];
    int64_t wrow[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        arow[p] = make_row<AMODE>(d, tm * BM + r0 + p * ROWS_PER_PASS);
        int gn = tn * BN + r0 + p * ROWS_PER_PASS;
        if constexpr (sizeof(T) == 4 && BKSEL == BK_SPLIT_WPL) gn = tn * BN + (tid >> 2) + 64 * (p & 1);      // weight planes: see ld_w
        wrow[p] = (int64_t)(gn < d.N ? gn : d.N - 1) * d.ldb;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register-staged pipeline, PREFETCH DEPTH 2: two staging register sets.  While tile-slice t is multiplied out of LDS, slice t+1
    // sits in one set (loaded during the previous iteration, stored to the other LDS buffer after the MFMAs) and the loads of slice
    // t+2 are issued into the other set - two global-load latencies deep instead of one (SQ counters r01: GEMM waves parked 55 % of
    // their cycles on vmcnt / barriers with the one-deep pipeline).
    constexpr bool XF = AMODE == P3_A_AFFINE_RELU || AMODE == P3_A_PAIR_AFFINE_RELU, PAIRA = AMODE == P3_A_PAIR_AFFINE_RELU;
    constexpr int XF_MAXK = 512;                               // scale / shift of the generated operand live in LDS (host checks K)
    __shared__ float xsc[XF ? XF_MAXK : 1], xsh[XF ? XF_MAXK : 1];
    if constexpr (XF) {
        for (int i = tid; i < d.K; i += 256) { xsc[i] = d.a_scale[i]; xsh[i] = d.a_shift[i]; }
        __syncthreads();
    }
    GT(1);
    uint4 ra[2][NPASS], rb[2][NPASS];
    uint4 rv[PAIRA ? 2 : 1][PAIRA ? NPASS : 1];                // pair mode: the V rows
    auto load_v = [&](int p, int k) __attribute__((always_inline)) {
        return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(d.pair_V) + arow[p].off2 + k);
    };
    constexpr bool SPLIT = sizeof(T) == 4 && (BKSEL == BK_SPLIT || BKSEL == BK_SPLIT_WPL), WPL = sizeof(T) == 4 && BKSEL == BK_SPLIT_WPL;
    auto put = [&](T* buf, int row, const uint4& v) __attribute__((always_inline)) {
        if constexpr (SPLIT) lds_put_split(reinterpret_cast<float*>(buf), row, kq, v);
        else lds_put<T, PITCH>(buf, row, kq, v);
    };
    // W as planes (WPL): the slice of a plane is 128 rows x 64 B = 512 sixteen-byte items, 1024 for both planes = the thread's four staged vectors: vector p is
    // 8 k-values of row tid / 4 + 64 (p & 1) of plane p >> 1, loaded with ONE 16-byte load and stored with one ds_write_b128 - as many load instructions as the fp32
    // weight needs (two 8-byte loads per vector, the first form of this, doubled the W side's address-path work and LOST 0.55 ms of the step: r06_g33) and no VALU
    auto put_w = [&](T* buf, int p, const uint4& v) __attribute__((always_inline)) {
        if constexpr (WPL) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(buf) + (p >> 1) * SPLIT_IMG_B + ((tid >> 2) + 64 * (p & 1)) * SPLIT_PITCH_B + (tid & 3) * 16) = v;
        else put(buf, r0 + p * ROWS_PER_PASS, v);
    };
    auto ld_w = [&](int p, int kslice) __attribute__((always_inline)) {
        if constexpr (WPL) return *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>((p >> 1) ? d.w_lo : g.W) + wrow[p & 1] + kslice + (tid & 3) * 8);
        else return load_w<T>(W, wrow[p], kslice + kq);
    };
    auto put_a = [&](T* buf, int set, int p, int k) __attribute__((always_inline)) {
        if constexpr (XF) put(buf, r0 + p * ROWS_PER_PASS, xform_a<T, AMODE>(ra[set][p], rv[PAIRA ? set : 0][PAIRA ? p : 0], xsc, xsh, k));
        else put(buf, r0 + p * ROWS_PER_PASS, ra[set][p]);
    };
    const int nk = d.K / BK;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        ra[0][p] = load_a<T, AMODE>(d, A, arow[p], kq); rb[0][p] = ld_w(p, 0);
        if constexpr (PAIRA) rv[0][p] = load_v(p, kq);
    }
    if (nk > 1) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            ra[1][p] = load_a<T, AMODE>(d, A, arow[p], BK + kq); rb[1][p] = ld_w(p, BK);
            if constexpr (PAIRA) rv[1][p] = load_v(p, BK + kq);
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) { put_a(lds, 0, p, kq); put_w(lds + 2 * LDSE, p, rb[0][p]); }
    __syncthreads();

    // FULL = steady state (slices t+1 and t+2 exist): no conditions around the loads / LDS stores.  With the conditions inside the loop the
    // compiler's s_waitcnt pass sees a path on which a set's loads were never consumed and drains vmcnt BEFORE re-issuing loads into that
    // set - i.e. the loads of slice t+2 waited for slice t+1 to land, one slice of prefetch instead of two (ISA of r02: vmcnt(3..0) at
    // the loop head).  The last slices run the conditional form after the loop.
    auto step = [&](auto SET, auto FULLT, int t) __attribute__((always_inline)) {
        constexpr int s0 = decltype(SET)::value;     // register set that held slice t (free now); slice t+1 is in set s0 ^ 1
        constexpr bool FULL = decltype(FULLT)::value;
        const int cur = t & 1;
        if (FULL || t + 2 < nk) {
            const int k = (t + 2) * BK + kq;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                ra[s0][p] = load_a<T, AMODE>(d, A, arow[p], k); rb[s0][p] = ld_w(p, (t + 2) * BK);
                if constexpr (PAIRA) rv[s0][p] = load_v(p, k);
            }
        }
        const T* as = lds + cur * LDSE;
        const T* bs = lds + (2 + cur) * LDSE;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                s16x8 af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = *reinterpret_cast<const s16x8*>(as + (wm * 64 + i * 32 + l31) * PITCH + kk * 16 + 8 * hi);
                    bf[i] = *reinterpret_cast<const s16x8*>(bs + (wn * 64 + i * 32 + l31) * PITCH + kk * 16 + 8 * hi);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), af[i]),
                            __builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), bf[j]), acc[i][j], 0, 0, 0);
            }
        } else if constexpr (SPLIT) {
            const unsigned char* ab = reinterpret_cast<const unsigned char*>(as);
            const unsigned char* bb = reinterpret_cast<const unsigned char*>(bs);
            typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bfx8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ao = (wm * 64 + i * 32 + l31) * SPLIT_PITCH_B + 32 * kk + 16 * hi, bo = (wn * 64 + i * 32 + l31) * SPLIT_PITCH_B + 32 * kk + 16 * hi;
                    ah[i] = *reinterpret_cast<const bfx8*>(ab + ao); al[i] = *reinterpret_cast<const bfx8*>(ab + SPLIT_IMG_B + ao);
                    bh[i] = *reinterpret_cast<const bfx8*>(bb + bo); bl[i] = *reinterpret_cast<const bfx8*>(bb + SPLIT_IMG_B + bo);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {      // small terms first
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = as[(kk * 2 + hi) * PITCH + wm * 64 + i * 32 + l31];
                    bf[i] = bs[(kk * 2 + hi) * PITCH + wn * 64 + i * 32 + l31];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (FULL || t + 1 < nk) {
            T* an = lds + (cur ^ 1) * LDSE;
            T* bn = lds + (2 + (cur ^ 1)) * LDSE;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) { put_a(an, s0 ^ 1, p, (t + 1) * BK + kq); put_w(bn, p, rb[s0 ^ 1][p]); }
        }
        __syncthreads();
    };
    GT(2);
    {
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        int t = 0;
        for (; t + 3 < nk; t += 2) { step(S0{}, std::true_type{}, t); step(S1{}, std::true_type{}, t + 1); }
        for (; t < nk; t += 2) {
            step(S0{}, std::false_type{}, t);
            if (t + 1 < nk) step(S1{}, std::false_type{}, t + 1);
        }
    }

    GT(3);
    // ---- epilogue: accumulators (+bias) -> LDS as fp32 [128][132] -> whole 8-element row chunks per thread ----------
    // A row-per-lane epilogue would issue 64 two-byte stores and 64 dependent residual loads per thread (measured: the
    // epilogue, not the MFMA loop, bounded the K <= 1536 GEMMs of this path); staging through the now idle operand LDS
    // makes every residual load / aux store / output store a 16-byte access (guide T21 idea, done through LDS).
    constexpr int EP = 132;
    float* stage = reinterpret_cast<float*>(lds);
    TO* C = reinterpret_cast<TO*>(g.C);
    TO* aux = reinterpret_cast<TO*>(d.aux);
    const bool has_res = d.residual != nullptr;
    const bool res_bf = d.dtype_res == P3_BF16;
    const TO* bwd_saved = reinterpret_cast<const TO*>(d.bwd_saved);
    const bool aux_grad = d.aux_mode == 1;
    const int act = d.act;
    const DropKey dk = drop_key(d.drop);
    // Plain bf16 tile (bias only: no activation / aux copy / dropout / saved-activation factor / residual, N % 8 == 0): ONE pass through a
    // bf16 image [128][136] - neighbouring lanes (= neighbouring columns of the MFMA fragment) exchange one value so that every lane packs
    // a column PAIR of one row (32 ds_write_b32 instead of 64 per lane, all four waves at once), one barrier, 16-byte LDS reads, 16-byte
    // stores.  No flag is tested inside the loops.  (The general form below: two passes of 64 rows through an fp32 image, four barriers.)
    if constexpr (sizeof(TO) == 2 && sizeof(T) == 2 && !STATS) {
        const bool simple = act == P3_ACT_NONE && !aux && !dk.on && !bwd_saved && !has_res && g.vec_epi && (d.N & 7) == 0;
        if (__builtin_amdgcn_readfirstlane((int)simple)) {
            constexpr int P16 = 68;                              // row pitch in dwords (136 bf16): 16-byte aligned rows
            uint32_t* st16 = reinterpret_cast<uint32_t*>(lds);
            const bool odd = l31 & 1;
            const bool ts = g.tile_stats != nullptr;
            const bool full_rows = (tm + 1) * BM <= d.M;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + l31;
                const int col = tn * BN + cl;
                const float bias = (d.bias && col < d.N) ? d.bias[col] : 0.f;
                float t1 = 0.f, t2 = 0.f;                 // this lane's column over the wave's 64 rows (tile_stats)
                if (__builtin_amdgcn_readfirstlane((int)ts)) {           // a loop of its own: the packing loop below stays free of selects
                    if (__builtin_amdgcn_readfirstlane((int)full_rows)) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r] + bias; t1 += v; t2 = fmaf(v, v, t2); }
                    } else {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = (tm * BM + wm * 64 + i * 32 + crow32(r, hi) < d.M) ? acc[i][j][r] + bias : 0.f;
                                t1 += v; t2 = fmaf(v, v, t2);
                            }
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float v0 = acc[i][j][r] + bias, v1 = acc[i][j][r + 1] + bias;      // rows rw, rw + 1 of this lane's column
                        // the partner's (lane ^ 1) value of the row this lane packs: DPP quad_perm [1,0,3,2] - one VALU move, no LDS crossbar
                        const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(odd ? v0 : v1), 0xB1, 0xf, 0xf, true));
                        const int rw = wm * 64 + i * 32 + crow32(r, hi) + (odd ? 1 : 0);
                        st16[rw * P16 + (cl >> 1)] = odd ? pack_bf2(recv, v1) : pack_bf2(v0, recv);
                    }
                }
                if (ts) {      // BatchNorm column sums as per-tile partials (pre-rounding fp32 values, bias included): no persistence, no atomics
                    t1 += __shfl_xor(t1, 32, 64); t2 += __shfl_xor(t2, 32, 64);
                    if (hi == 0 && col < d.N) {
                        float* part = g.tile_stats + ((int64_t)(tm * 2 + wm) * 2) * d.N;
                        part[col] = t1; part[d.N + col] = t2;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int id = tid + 256 * c;
                const int rl = id >> 4, cl = (id & 15) * 8;
                const int row = tm * BM + rl, col = tn * BN + cl;
                const uint4 o = *reinterpret_cast<const uint4*>(st16 + rl * P16 + (cl >> 1));
                if (row < d.M && col < d.N) *reinterpret_cast<uint4*>(C + (int64_t)row * d.ldc + col) = o;
            }
            GT(4);
            return;
        }
    }
#pragma unroll
    for (int pass = 0; pass < EPI_PASSES; ++pass) {
        if (pass > 0) __syncthreads();               // the previous pass's readers are done with the staging buffer
        if (EPI_PASSES == 1 || wm == pass) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = wn * 64 + j * 32 + l31;
                const int col = tn * BN + cl;
                const bool cok = col < d.N;
                const float bias = (d.bias && cok) ? d.bias[col] : 0.f;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rw = wm * 64 + i * 32 + crow32(r, hi);     // row inside the 128-row tile
                        const float v = acc[i][j][r] + bias;
                        stage[(rw - pass * EPI_ROWS) * EP + cl] = v;
                        if (tm * BM + rw < d.M && cok) { s1 += v; s2 += v * v; }
                    }
                }
                if constexpr (STATS) { cs1[j] += s1; cs2[j] += s2; }
                else if (g.tile_stats) {
                    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
                    if (hi == 0 && cok) {
                        float* part = g.tile_stats + ((int64_t)(tm * 2 + wm) * 2) * d.N;
                        part[col] = s1; part[d.N + col] = s2;
                    }
                }
            }
        }
        __syncthreads();
    #pragma unroll 2
        for (int c = 0; c < 8 / EPI_PASSES; ++c) {
            const int id = tid + 256 * c;
            const int rl = id >> 4, cl = (id & 15) * 8;
            const int row = tm * BM + pass * EPI_ROWS + rl, col = tn * BN + cl;
            if (row >= d.M || col >= d.N) continue;
            float v[8];
            {
                const float4 v0 = *reinterpret_cast<const float4*>(stage + rl * EP + cl);
                const float4 v1 = *reinterpret_cast<const float4*>(stage + rl * EP + cl + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            }
            const int64_t co = (int64_t)row * d.ldc + col;
            if (g.vec_epi && col + 8 <= d.N) {
                if (act == P3_ACT_GELU) {
                    float gd[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { const float x = v[k]; gelu_and_grad(x, v[k], gd[k]); if (!aux_grad) gd[k] = x; }
                    if (aux) {   // pre-activation (aux_mode 0) or GELU'(pre) (aux_mode 1)
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
                    if (d.bwd_act == P3_ACT_BN_RELU) {
                        const float* bn = d.bwd_bn + col;
    #pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float sc_ = bn[k];
                            v[k] = (sv[k] * sc_ + bn[d.N + k] > 0.f ? v[k] * sc_ : 0.f) + bn[2 * d.N + k] + bn[3 * d.N + k] * sv[k];
                        }
                    } else {
    #pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] *= act_grad(sv[k], d.bwd_act) * d.bwd_scale;
                    }
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
            } else {
                for (int k = 0; k < 8 && col + k < d.N; ++k) {
                    float x = v[k];
                    if (act == P3_ACT_GELU) {
                        const float pre = x; float gd;
                        gelu_and_grad(pre, x, gd);
                        if (aux) aux[co + k] = Cvt<TO>::from_f(aux_grad ? gd : pre);
                    } else {
                        if (aux) aux[co + k] = Cvt<TO>::from_f(x);
                        if (act == P3_ACT_RELU) x = fmaxf(x, 0.f);
                    }
                    if (dk.on) x = drop_keep(dk, (uint64_t)row, (uint32_t)(col + k)) ? x * dk.inv_keep : 0.f;
                    if (bwd_saved && d.bwd_act == P3_ACT_BN_RELU) {
                        const float hv = Cvt<TO>::to_f(bwd_saved[co + k]);
                        const float* bn = d.bwd_bn + col + k;
                        x = (hv * bn[0] + bn[d.N] > 0.f ? x * bn[0] : 0.f) + bn[2 * d.N] + bn[3 * d.N] * hv;
                    } else if (bwd_saved) x *= act_grad(Cvt<TO>::to_f(bwd_saved[co + k]), d.bwd_act) * d.bwd_scale;
                    if (has_res) {
                        const int64_t ri = (int64_t)row * d.ldr + col + k;
                        x += res_bf ? bf2f(reinterpret_cast<const bf16_t*>(d.residual)[ri]) : reinterpret_cast<const float*>(d.residual)[ri];
                    }
                    C[co + k] = Cvt<TO>::from_f(x);
                }
            }
        }
    }
    if constexpr (STATS) __syncthreads();            // the staging buffer aliases the next tile's operand buffers
    vb += gridDim.x;
  } while (STATS && vb < ntiles);
    if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tn_stats * BN + wn * 64 + j * 32 + l31;
            const float s1 = cs1[j] + __shfl_xor(cs1[j], 32, 64), s2 = cs2[j] + __shfl_xor(cs2[j], 32, 64);
            if (hi == 0 && col < d.N && blockIdx.x < (unsigned)ntiles) {
                if (g.stat_slab) {                    // parts of a column: (workgroup of its tile column, row half wm) (det_reduce.hip)
                    float* part = g.stat_slab + ((int64_t)(blockIdx.x / g.tiles_n) * 2 + wm) * 2 * d.N;
                    part[col] = s1; part[d.N + col] = s2;
                } else { atomicAdd(d.colsum + col, s1); atomicAdd(d.colsumsq + col, s2); }
            }
        }
    }
}

// ---- skinny rows (M <= 128: the KV-cached decode step, one token per tile) -------------------------------------------------------
// A 128x128 tile is mostly padding there and its LDS pipeline turns K = 2048 into 32 dependent global->LDS->MFMA round trips of one
// or two workgroups (r01: 35 us for the decoder's linear2 at M = 64).  Here ONE WAVE owns a 32x32 output block and feeds the MFMA
// straight from global memory (a lane's 16-byte operand chunk is exactly its fragment: row l31, k-offset 8*hi), eight K-steps of
// loads in flight ahead of the MFMAs, no LDS, no barrier; N/32 x M/32 single-wave workgroups spread over the CUs.  Same instruction,
// same k order and the same fp32 epilogue arithmetic as gemm_kernel => bit-identical outputs for K < 1024 (tests).
template <typename TO, int SPLIT>
__global__ __launch_bounds__(64 * SPLIT) void gemm_skinny_kernel(GemmArgs g) {
    const p3_gemm_desc& d = g.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const int tn = blockIdx.x, tm = blockIdx.y;
    const int ar = min(tm * 32 + l31, d.M - 1), wr = min(tn * 32 + l31, d.N - 1);
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)ar * d.lda + 8 * hi;
    const bf16_t* wp = reinterpret_cast<const bf16_t*>(g.W) + (int64_t)wr * d.ldb + 8 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 8;
    const int nk_all = d.K / 16;
    // SPLIT > 1 (K >= 1024: the decoder's linear2): the waves of the workgroup take consecutive K ranges - a single wave cannot keep
    // enough loads in flight to stream 32 x 4 KB of weights at memory latency (r01: 28 us at K = 2048) - and the partial sums are
    // added in wave order through LDS (deterministic; fp32 rounding differs from the one-pass order in the last bits)
    const int per = (nk_all + SPLIT - 1) / SPLIT;
    const int kbeg = SPLIT > 1 ? wave * per : 0;
    const int nk = SPLIT > 1 ? min(nk_all, kbeg + per) : nk_all;
    uint4 a[2][U], b[2][U];
    auto load = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < nk) {
                a[set][u] = *reinterpret_cast<const uint4*>(ap + (k0 + u) * 16);
                b[set][u] = *reinterpret_cast<const uint4*>(wp + (k0 + u) * 16);
            }
        }
    };
    auto mma = [&](int set, int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < nk)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), a[set][u]),
                                                              __builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), b[set][u]), acc, 0, 0, 0);
        }
    };
    load(0, kbeg);
    for (int k0 = kbeg; k0 < nk; k0 += 2 * U) {
        load(1, k0 + U);
        mma(0, k0);
        load(0, k0 + 2 * U);
        mma(1, k0 + U);
    }
    if constexpr (SPLIT > 1) {
        __shared__ float part[SPLIT - 1][16][64];
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) part[wave - 1][r][lane] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SPLIT - 1; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += part[w][r][lane];
    }
    const int col = tn * 32 + l31;
    if (col >= d.N) return;
    const float bias = d.bias ? d.bias[col] : 0.f;
    TO* C = reinterpret_cast<TO*>(g.C);
    const bool res_bf = d.dtype_res == P3_BF16;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 32 + crow32(r, hi);
        if (row >= d.M) continue;
        float v = acc[r] + bias;
        if (d.act == P3_ACT_GELU) v = gelu_erf(v);
        else if (d.act == P3_ACT_RELU) v = fmaxf(v, 0.f);
        if (d.residual) {
            const int64_t ri = (int64_t)row * d.ldr + col;
            v += res_bf ? bf2f(reinterpret_cast<const bf16_t*>(d.residual)[ri]) : reinterpret_cast<const float*>(d.residual)[ri];
        }
        C[(int64_t)row * d.ldc + col] = Cvt<TO>::from_f(v);
    }
}

template <typename T, typename TO, int BKSEL, bool STATS>
int launch_bk2(const GemmArgs& g, hipStream_t s) {
    const int ntiles = g.tiles_m * g.tiles_n;
    int nwg = ntiles;
    if (STATS) {                                      // persistent: <= ~2048 workgroups, a multiple of tiles_n
        const int cap = (2048 / g.tiles_n) * g.tiles_n;
        if (cap >= g.tiles_n && nwg > cap) nwg = cap;
    }
    dim3 grid(nwg), block(256);
    if (p3_tracing()) {
        char nm[96];
        snprintf(nm, sizeof(nm), "gemm_kernel<%s, %s, %d, %d, %s>", sizeof(T) == 2 ? "bf16" : "float", sizeof(TO) == 2 ? "bf16" : "float", g.d.a_mode, BKSEL, STATS ? "true" : "false");
        p3_note_kernel(nm);
    }
    GemmArgs gs = g;
    const int nparts = 2 * (nwg / g.tiles_n);       // two row halves (wave rows) per workgroup
    if (STATS) gs.stat_slab = (nwg % g.tiles_n == 0) ? p3_det_scratch((int64_t)nparts * 2 * g.d.N, g.d.dtype_in) : nullptr;
    // (a column-major tile walk for weight matrices larger than one XCD's L2 - the fusion conv's 5.3 MB - measured SLOWER, 408 -> 423 us: the
    // re-fetch the PMC pass shows is served by the MALL and does not bound the launch; r03, removed in r04)
    {
    const GemmArgs& g = gs;
    switch (g.d.a_mode) {
        case P3_A_PLAIN: hipLaunchKernelGGL((gemm_kernel<T, TO, P3_A_PLAIN, BKSEL, STATS>), grid, block, 0, s, g); break;
        case P3_A_CONV3X3: hipLaunchKernelGGL((gemm_kernel<T, TO, P3_A_CONV3X3, BKSEL, STATS>), grid, block, 0, s, g); break;
        case P3_A_AFFINE_RELU: hipLaunchKernelGGL((gemm_kernel<T, TO, P3_A_AFFINE_RELU, BKSEL, STATS>), grid, block, 0, s, g); break;
        case P3_A_PAIR_AFFINE_RELU: hipLaunchKernelGGL((gemm_kernel<T, TO, P3_A_PAIR_AFFINE_RELU, BKSEL, STATS>), grid, block, 0, s, g); break;
        case P3_A_CONV3X3_AFFINE_RELU: hipLaunchKernelGGL((gemm_kernel<T, TO, P3_A_CONV3X3_AFFINE_RELU, BKSEL, STATS>), grid, block, 0, s, g); break;
        default: p3_set_error("p3_gemm: bad a_mode"); return P3_EINVAL;
    }
    }
    P3_LAUNCH_CHECK();
    if (STATS && gs.stat_slab) {
        int rc = p3_det_reduce(gs.stat_slab, nparts, 2 * (int64_t)g.d.N, g.d.colsum, g.d.N, 1, s);
        if (rc != P3_OK) return rc;
        return p3_det_reduce(gs.stat_slab + g.d.N, nparts, 2 * (int64_t)g.d.N, g.d.colsumsq, g.d.N, 1, s);
    }
    return P3_OK;
}

template <typename T, typename TO, int BKSEL>
int launch_bk(const GemmArgs& g, hipStream_t s) {
    if (!g.d.colsum) return launch_bk2<T, TO, BKSEL, false>(g, s);
    // BatchNorm column sums.  Preferred form (r03): the ordinary one-tile-per-workgroup kernel (3 workgroups / CU, single-pass bf16 epilogue)
    // writes per-tile, per-row-half partial sums into the registered scratch and a two-level fixed-order float64 reduction adds them -
    // tools/mb_sn_fwd.py: the persistent register-resident form below costs 573 vs 431 us (ScoreNet conv2) and 391 vs 215 us (conv3) over the
    // same product without sums.  Also bit-reproducible in bf16.  No / too small a scratch: the persistent form.
    const int nparts = g.tiles_m * 2, nch = (nparts + 127) / 128;
    const int64_t slab_f = (int64_t)nparts * 2 * g.d.N;
    float* scratch = p3_reduce_scratch(slab_f + (int64_t)nch * 2 * g.d.N);
    if (!scratch) return launch_bk2<T, TO, BKSEL, true>(g, s);
    GemmArgs gt = g;
    gt.tile_stats = scratch;
    gt.d.colsum = nullptr; gt.d.colsumsq = nullptr;
    int rc = launch_bk2<T, TO, BKSEL, false>(gt, s);
    if (rc != P3_OK) return rc;
    return p3_det_reduce2(scratch, nparts, 2 * (int64_t)g.d.N, scratch + slab_f, g.d.colsum, g.d.colsumsq, g.d.N, 2 * g.d.N, 1, s);   // (sum | sum of squares) in one go
}

template <typename T, typename TO>
int launch_mode(const GemmArgs& g, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        const bool conv = g.d.a_mode == P3_A_CONV3X3 || g.d.a_mode == P3_A_CONV3X3_AFFINE_RELU;
        const bool can64 = g.d.K % 64 == 0 && (!conv || g.d.conv_C % 64 == 0);
        const bool deep = can64 && g.d.K >= 2048;       // 64-deep K slices from K = 2048 on (r01 sweep)
        return deep ? launch_bk<T, TO, 64>(g, s) : launch_bk<T, TO, 32>(g, s);
    } else {
        // the split form walks K in 32-deep slices; a K (or conv channel count) that is only a multiple of 16 stays on the exact kernel
        const bool conv = g.d.a_mode == P3_A_CONV3X3 || g.d.a_mode == P3_A_CONV3X3_AFFINE_RELU;
        const bool can_split = g.split && g.d.K % SPLIT_BK == 0 && (!conv || g.d.conv_C % SPLIT_BK == 0);
        if (g.d.w_lo) {
            if (!can_split) { p3_set_error("p3_gemm: w_lo (weight planes) needs P3_F32X3 and K (conv: C) % 32 == 0"); return P3_EUNSUP; }
            return launch_bk<T, TO, BK_SPLIT_WPL>(g, s);
        }
        return can_split ? launch_bk<T, TO, BK_SPLIT>(g, s) : launch_bk<T, TO, 16>(g, s);
    }
}

}  // namespace

// gemm_dma.hip: LDS-DMA kernels for the plain bf16 products (variant 4: 128 x 128 tile, 64-deep slices, two in LDS; 6: 32-deep, two = 4 workgroups / CU;
// 9: 128 x 384 tile, 8 waves)
int p3_rows_gemm_try(const void* A, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s);   // rows_gemm.hip: 0x7fffffff = not one of its shapes
int p3_pair_fwd_try(const void* U, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s);    // pair_fwd_mma.hip: same convention
int p3_pair_fwd_x3_try(const void* U, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s); // pair_fwd_x3.hip: the P3_F32X3 form
int p3_rows_x3_try(const void* A, const void* W, void* C, const p3_gemm_desc* d, hipStream_t s);     // rows_x3.hip: conv3 forward, P3_F32X3
int p3_gemm_dma_eligible(const p3_gemm_desc* d, const void* A, const void* W, const void* C);
int p3_gemm_dma_launch(const void* A, const void* W, void* C, const p3_gemm_desc* d, int variant, hipStream_t s);
static int gemm_dma_mode() { static int m = -1; if (m < 0) { const char* e = getenv("P3_GEMM_DMA"); m = (e && e[0] == '0') ? 0 : 1; } return m; }

extern "C" int p3_gemm_dma(const void* A, const void* W, void* C, const p3_gemm_desc* d, int variant, void* stream) {
    P3_CHECK(A && W && C && d, P3_EINVAL, "p3_gemm_dma: null pointer");
    P3_CHECK(d->M > 0 && d->N > 0 && d->K > 0, P3_ESHAPE, "p3_gemm_dma: empty problem");
    P3_CHECK(variant == 4 || variant == 6 || variant == 9, P3_EINVAL, "p3_gemm_dma: variant 4, 6 or 9");
    P3_CHECK(p3_gemm_dma_eligible(d, A, W, C), P3_EUNSUP, "p3_gemm_dma: plain bf16 A, K % 64 == 0, N % 8 == 0, 16-byte aligned rows, no column sums");
    return p3_gemm_dma_launch(A, W, C, d, variant, (hipStream_t)stream);
}

extern "C" int p3_gemm(const void* A, const void* W, void* C, const p3_gemm_desc* d_in, void* stream) {
    P3_CHECK(A && W && C && d_in, P3_EINVAL, "p3_gemm: null pointer");
    p3_gemm_desc dn = *d_in;                      // P3_F32X3 = fp32 operands + the split flag: everything below sees P3_F32
    const int split = dn.dtype_in == P3_F32X3;
    if (split) dn.dtype_in = P3_F32;
    const p3_gemm_desc* d = &dn;
    P3_CHECK(d->M > 0 && d->N > 0 && d->K > 0, P3_ESHAPE, "p3_gemm: empty problem");
    const int bk = d->dtype_in == P3_BF16 ? 32 : 16;
    const int vec = d->dtype_in == P3_BF16 ? 8 : 4;
    P3_CHECK(d->dtype_in == P3_BF16 || d->dtype_in == P3_F32, P3_EUNSUP, "p3_gemm: dtype_in");
    P3_CHECK(d->dtype_out == P3_BF16 || d->dtype_out == P3_F32, P3_EUNSUP, "p3_gemm: dtype_out");
    P3_CHECK(d->K % bk == 0, P3_ESHAPE, "p3_gemm: K must be a multiple of 32 (bf16) / 16 (f32)");
    P3_CHECK(d->lda % vec == 0 && d->ldb % vec == 0, P3_EALIGN, "p3_gemm: lda/ldb must keep 16-byte row alignment");
    P3_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, P3_EALIGN, "p3_gemm: A/W must be 16-byte aligned");
    if (d->w_lo) {
        P3_CHECK(split && (d->a_mode == P3_A_PLAIN || d->a_mode == P3_A_CONV3X3 || d->a_mode == P3_A_CONV3X3_AFFINE_RELU), P3_EUNSUP,
                 "p3_gemm: w_lo (weight planes) goes with P3_F32X3 and a plain or 3x3-gathered A");
        P3_CHECK(((uintptr_t)d->w_lo % 16) == 0 && d->ldb % 8 == 0, P3_EALIGN, "p3_gemm: weight planes: 16-byte aligned, ldb % 8 == 0");
    }
    if (d->a_mode == P3_A_CONV3X3 || d->a_mode == P3_A_CONV3X3_AFFINE_RELU) {
        P3_CHECK(d->conv_C > 0 && d->conv_C % bk == 0 && d->K == 9 * d->conv_C, P3_ESHAPE, "p3_gemm: conv3x3 needs K == 9*C, C % BK == 0");
        P3_CHECK(d->M % (d->conv_H * d->conv_W) == 0, P3_ESHAPE, "p3_gemm: conv3x3 needs M == B*H*W");
        P3_CHECK(!d->conv_pad || d->a_mode == P3_A_CONV3X3, P3_EINVAL, "p3_gemm: conv_pad goes with P3_A_CONV3X3 (the border must already hold zeros)");
    }
    if (d->a_mode == P3_A_AFFINE_RELU || d->a_mode == P3_A_PAIR_AFFINE_RELU || d->a_mode == P3_A_CONV3X3_AFFINE_RELU)
        P3_CHECK(d->a_scale && d->a_shift, P3_EINVAL, "p3_gemm: affine mode needs a_scale/a_shift");
    if (d->a_mode == P3_A_AFFINE_RELU || d->a_mode == P3_A_PAIR_AFFINE_RELU)
        P3_CHECK(d->K <= 512, P3_EUNSUP, "p3_gemm: generated A operand: K <= 512 (scale / shift tables in LDS)");
    if (d->a_mode == P3_A_PAIR_AFFINE_RELU)
        P3_CHECK(d->pair_V && d->pair_n > 0 && d->M % (d->pair_n * d->pair_n) == 0, P3_ESHAPE, "p3_gemm: pair mode needs V and M == B*n*n");
    P3_CHECK((d->colsum == nullptr) == (d->colsumsq == nullptr), P3_EINVAL, "p3_gemm: colsum and colsumsq go together");
    P3_CHECK(!d->bwd_saved || d->bwd_act == P3_ACT_GELU || d->bwd_act == P3_ACT_RELU || d->bwd_act == P3_ACT_MUL || d->bwd_act == P3_ACT_BN_RELU, P3_EINVAL,
             "p3_gemm: bwd_saved needs bwd_act = GELU, RELU, MUL or BN_RELU");
    P3_CHECK(!(d->bwd_saved && d->bwd_act == P3_ACT_BN_RELU) || d->bwd_bn, P3_EINVAL, "p3_gemm: P3_ACT_BN_RELU needs bwd_bn");
    GemmArgs g;
    g.A = A; g.W = W; g.C = C; g.d = *d; g.stat_slab = nullptr; g.tile_stats = nullptr; g.split = split;
    g.tiles_m = p3_ceil_div(d->M, BM);
    g.tiles_n = p3_ceil_div(d->N, BN);
    {
        const int vo = d->dtype_out == P3_BF16 ? 8 : 4;     // elements per 16 bytes
        bool ok = (d->ldc % vo == 0) && ((uintptr_t)C % 16 == 0) && (!d->aux || (uintptr_t)d->aux % 16 == 0);
        if (d->residual) { const int vr = d->dtype_res == P3_BF16 ? 8 : 4; ok = ok && (d->ldr % vr == 0) && ((uintptr_t)d->residual % 16 == 0); }
        if (d->bwd_saved) ok = ok && ((uintptr_t)d->bwd_saved % 16 == 0);
        g.vec_epi = ok ? 1 : 0;
    }
    hipStream_t s = (hipStream_t)stream;
    {   // the ScoreNet's thin 1x1 convolutions over millions of rows: weight-stationary streaming kernels (rows_gemm.hip); its conv2 over the pair
        // grid: pair_fwd_mma.hip
        int rc = d->w_lo ? 0x7fffffff : p3_rows_gemm_try(A, W, C, d, s);
        if (rc != 0x7fffffff) return rc;
        rc = d->w_lo ? 0x7fffffff : p3_pair_fwd_try(A, W, C, d, s);
        if (rc != 0x7fffffff) return rc;
        if (split && !d->w_lo) {
            rc = p3_pair_fwd_x3_try(A, W, C, d, s);
            if (rc != 0x7fffffff) return rc;
            rc = p3_rows_x3_try(A, W, C, d, s);
            if (rc != 0x7fffffff) return rc;
        }
    }
    if (gemm_dma_mode() > 0 && d->M >= 2048 && p3_gemm_dma_eligible(d, A, W, C)) {
        // P3_GEMM_DMA=0 switches the rule off (everything on the register-staged kernel).  The rule (r03, tools/mb_gemm_shapes.py + same-box A/B of the
        // train step 40.60 -> 40.15 -> 38.78 ms): the 64-deep two-slice form from K = 1024 on (dX of fc1 / qkv, decoder linear2: 74 vs 87, 59 vs 67, 33 vs
        // 42 us); the 32-deep two-slice form, 4 workgroups / CU, on wide outputs with K <= 512 (qkv, fc1, dX of fc2, linear1: 66 vs 69, 86 vs 92, 44 vs 45 us);
        // the 128 x 384 tile for the 384-column outputs from K = 1024 on (fc2, dX of fc1 / qkv: 393 workgroups = ONE resident round instead of 2.3 rounds
        // of 128 x 128 tiles); everything else stays on the register-staged kernel.
        int v = d->K >= 1024 ? 4 : (d->K <= 512 && d->N >= 1024 ? 6 : 0);
        if (d->N == 384 && d->K >= 1024) v = 9;
        if (v) return p3_gemm_dma_launch(A, W, C, d, v, s);
    }
    constexpr int no_skinny = 0;
    if (!no_skinny && d->dtype_in == P3_BF16 && d->M <= 128 && d->a_mode == P3_A_PLAIN && !d->colsum && !d->aux && !d->bwd_saved &&
        !(d->drop.seed && d->drop.p > 0.f)) {
        dim3 grid(p3_ceil_div(d->N, 32), p3_ceil_div(d->M, 32));
        if (p3_tracing()) p3_note_kernel(d->K >= 1024 ? "gemm_skinny_kernel<8>" : "gemm_skinny_kernel<1>");
        if (d->K >= 1024) {
            if (d->dtype_out == P3_BF16) hipLaunchKernelGGL((gemm_skinny_kernel<bf16_t, 8>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((gemm_skinny_kernel<float, 8>), grid, dim3(512), 0, s, g);
        } else {
            if (d->dtype_out == P3_BF16) hipLaunchKernelGGL((gemm_skinny_kernel<bf16_t, 1>), grid, dim3(64), 0, s, g);
            else hipLaunchKernelGGL((gemm_skinny_kernel<float, 1>), grid, dim3(64), 0, s, g);
        }
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    if (d->dtype_in == P3_BF16) return d->dtype_out == P3_BF16 ? launch_mode<bf16_t, bf16_t>(g, s) : launch_mode<bf16_t, float>(g, s);
    return d->dtype_out == P3_BF16 ? launch_mode<float, bf16_t>(g, s) : launch_mode<float, float>(g, s);
}
