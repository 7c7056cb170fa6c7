// p3hip fused attention backward (flash style, recompute; MFMA 32x32), no atomics:
//   attn_delta_kernel : delta[b,h,q] = sum_d dO[q,d] * O[q,d]
//   attn_bwd_dq_kernel: one workgroup per 128-query block, walks the key tiles  -> dQ
//   attn_bwd_dkv_kernel: one workgroup per 128-key block, walks the query tiles  -> dK, dV
// Both reuse the forward's lane-local formulation: a lane owns ONE query (dQ kernel) or ONE key (dK/dV kernel), the
// score-like products (K.Q^T, V.dO^T / Q.K^T, dO.V^T) put that index on the MFMA column, and the accumulate products
// (K^T.dS^T / dO^T.P, Q^T.dS) feed P / dS straight from the score registers with the k-slots assigned to the rows the
// lane already holds.  bf16: one swizzled row image per tile; the transposed fragments come from ds_read_b64_tr_b16 (attn_tile.h).
#include "p3_common.h"
#include "attn_tile.h"
#include <type_traits>

#ifndef P3_DKV32_WAVES
#define P3_DKV32_WAVES 3     // waves / SIMD the d = 32 dK/dV kernel is compiled for (-DP3_DKV32_WAVES=4: 2.0 instead of 2.67 rounds of workgroups but 23 spilled registers - same-box step 38.28 -> 38.79 ms, r03: stays 3)
#endif

namespace {

using p3attn::u32x4;
using p3attn::f32s;
using p3attn::Kind;

struct BwdArgs {
    const void* Q; const void* K; const void* V; const void* O; const void* dO;
    void* dQ; void* dK; void* dV;
    const float* lse; float* delta;
    p3_attn_desc d;
    int order;          // attn_block_of mode
};

template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(BwdArgs a, int D) {
    const p3_attn_desc& d = a.d;
    const int64_t total = (int64_t)d.B * d.H * d.Lq;
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < total; row += (int64_t)gridDim.x * 4) {
        const int q = (int)(row % d.Lq), h = (int)((row / d.Lq) % d.H), b = (int)(row / ((int64_t)d.Lq * d.H));
        const T* o = reinterpret_cast<const T*>(a.O) + (int64_t)b * d.o_bs + (int64_t)q * d.o_rs + h * D;
        const T* g = reinterpret_cast<const T*>(a.dO) + (int64_t)b * d.o_bs + (int64_t)q * d.o_rs + h * D;
        float s = lane < D ? Cvt<T>::to_f(o[lane]) * Cvt<T>::to_f(g[lane]) : 0.f;
        s = wave_sum(s);
        if (lane == 0) a.delta[row] = s;
    }
}

// ---- LDS tile: R rows x D.  bf16: ONE swizzled row image (attn_tile.h) serves both the score products (ds_read_b128) and the
// accumulate products (transposing ds_read_b64_tr_b16); f32: row image [R][D + 1].
// f32s (fp32x3 mode): TWO swizzled bf16 images, hi at the tile's start and lo R * D bf16 behind it (R * D elements of 4 bytes in all).
template <typename T, int D, int R> struct Tile {
    static constexpr bool BF = Kind<T>::BF, X3 = Kind<T>::X3, IMG = Kind<T>::IMG;
    static constexpr int PR = IMG ? D : D + 1;
    static constexpr int ROW_ELEMS = R * PR;
};

// Staging of R rows (row0.., clamped to nvalid-1) of a [*, row_stride] tensor is split in two (guide T14): load() issues the
// global loads of tile t+1 into registers BEFORE the MFMAs of tile t, store() writes them to LDS after the barrier that retires
// tile t's readers.  (PMC r01, synchronous staging: waves parked 47-58 % of their cycles in the dQ / dKV kernels.)
template <typename T, int D, int R> struct Stage;
template <int D, int R> struct Stage<bf16_t, D, R> : p3attn::RowStage<D, R> {};
template <int D, int R> struct Stage<float, D, R> {
    static constexpr int ITEMS = R * (D / 4), N = (ITEMS + 255) / 256;
    u32x4 v[N];
    __device__ __forceinline__ void load(const float* __restrict__ src, int row0, int nvalid, int64_t row_stride, int tid) {
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int item = tid + 256 * it;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                const int cv = item % (D / 4), r = item / (D / 4);
                int rr = row0 + r; if (rr >= nvalid) rr = nvalid - 1;
                v[it] = *reinterpret_cast<const u32x4*>(src + (int64_t)rr * row_stride + cv * 4);
            }
        }
    }
    __device__ __forceinline__ void store(float* rowimg, int tid) const {
        constexpr int PR = Tile<float, D, R>::PR;
#pragma unroll
        for (int it = 0; it < N; ++it) {
            const int item = tid + 256 * it;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                const int cv = item % (D / 4), r = item / (D / 4);
                float* p = rowimg + r * PR + cv * 4;
                p[0] = __uint_as_float(v[it].x); p[1] = __uint_as_float(v[it].y);
                p[2] = __uint_as_float(v[it].z); p[3] = __uint_as_float(v[it].w);
            }
        }
    }
};

template <int D, int R> struct Stage<f32s, D, R> {
    p3attn::SplitStage<D, R> st;
    __device__ __forceinline__ void load(const f32s* __restrict__ src, int row0, int nvalid, int64_t row_stride, int tid) {
        st.load(reinterpret_cast<const float*>(src), row0, nvalid, row_stride, tid);
    }
    __device__ __forceinline__ void store(f32s* rowimg, int tid) const { st.store(reinterpret_cast<bf16_t*>(rowimg), tid); }
};

// lane-constant LDS offsets of the bf16 fragment reads (empty for f32)
template <typename T, int D> struct FragAddr { __device__ __forceinline__ void init(int, int, int) {} };
template <int D> struct FragAddr<bf16_t, D> {
    p3attn::ScoreAddr<D> s;
    p3attn::TrAddr<D> t;
    __device__ __forceinline__ void init(int lane, int l31, int hi) { s.init(l31, hi); t.init(lane); }
};
template <int D> struct FragAddr<f32s, D> : FragAddr<bf16_t, D> {};

// score-like product: acc[rows32 x cols32] = X[rows from LDS row image] . Y[cols held in regs]^T   (contract over D)
template <typename T, int D, int R>
__device__ __forceinline__ f32x16 score_mma(const T* rowimg, int sub, const s16x8 (&yb)[D / 16 > 0 ? D / 16 : 1], const s16x8 (&yl)[D / 16 > 0 ? D / 16 : 1],
                                            const float (&yf)[D / 2], const FragAddr<T, D>& fa, int l31, int hi) {
    using TL = Tile<T, D, R>;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (TL::BF) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) acc = p3attn::mfma_bf16(fa.s.frag(rowimg, sub, ks), yb[ks], acc);
    } else if constexpr (TL::X3) {
        const bf16_t* ih = reinterpret_cast<const bf16_t*>(rowimg);
        const bf16_t* il = ih + R * D;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {          // small terms first
            const s16x8 xh = fa.s.frag(ih, sub, ks), xl = fa.s.frag(il, sub, ks);
            acc = p3attn::mfma_bf16(xl, yb[ks], acc);
            acc = p3attn::mfma_bf16(xh, yl[ks], acc);
            acc = p3attn::mfma_bf16(xh, yb[ks], acc);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < D / 2; ++ks) {
            const float xf = reinterpret_cast<const float*>(rowimg)[(sub * 32 + l31) * TL::PR + 2 * ks + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, yf[ks], acc, 0, 0, 0);
        }
    }
    return acc;
}

// fp32x3: the two score-like products of a sub-tile (S from one image with one set of row fragments, dP from the other) as two interleaved accumulator chains, the
// fragments of k-step ks + 1 read while the six MFMAs of step ks run (score_mma above compiles to read -> s_waitcnt lgkmcnt(0) -> three MFMAs on one chain, eight
// times per product).  -DP3_ATTN_PAIR=0: the two separate calls (A/B)
#ifndef P3_ATTN_PAIR
#define P3_ATTN_PAIR 1
#endif
template <typename T, int D, int R>
__device__ __forceinline__ void score_mma_pair(const T* img1, const T* img2, int sub, const s16x8 (&y1b)[D / 16], const s16x8 (&y1l)[D / 16], const s16x8 (&y2b)[D / 16],
                                               const s16x8 (&y2l)[D / 16], const FragAddr<T, D>& fa, f32x16& a1, f32x16& a2) {
    const bf16_t* h1 = reinterpret_cast<const bf16_t*>(img1);
    const bf16_t* h2 = reinterpret_cast<const bf16_t*>(img2);
    s16x8 f[2][4];
    auto rd = [&](int bsel, int ks) __attribute__((always_inline)) {
        f[bsel][0] = fa.s.frag(h1, sub, ks); f[bsel][1] = fa.s.frag(h1 + R * D, sub, ks);
        f[bsel][2] = fa.s.frag(h2, sub, ks); f[bsel][3] = fa.s.frag(h2 + R * D, sub, ks);
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) { a1[r] = 0.f; a2[r] = 0.f; }
    rd(0, 0);
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        const int bsel = ks & 1;
        if (ks + 1 < D / 16) rd(bsel ^ 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        a1 = p3attn::mfma_bf16(f[bsel][1], y1b[ks], a1); a2 = p3attn::mfma_bf16(f[bsel][3], y2b[ks], a2);      // small terms first
        a1 = p3attn::mfma_bf16(f[bsel][0], y1l[ks], a1); a2 = p3attn::mfma_bf16(f[bsel][2], y2l[ks], a2);
        a1 = p3attn::mfma_bf16(f[bsel][0], y1b[ks], a1); a2 = p3attn::mfma_bf16(f[bsel][2], y2b[ks], a2);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// accumulate product: out[dj][d32 x cols32] += X^T[d, rows(sub)] . W[rows(sub), cols]   with W lane-local (f32x16 of sub-tile `sub`)
template <typename T, int D, int R>
__device__ __forceinline__ void accum_mma(const T* rowimg, int sub, const f32x16& w, f32x16 (&out)[D / 32], const FragAddr<T, D>& fa, int l31, int hi) {
    using TL = Tile<T, D, R>;
    if constexpr (TL::BF) {
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            uint32_t pw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[i] = pack_bf2(w[8 * c2 + 2 * i], w[8 * c2 + 2 * i + 1]);
            const s16x8 wb = __builtin_bit_cast(s16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
#pragma unroll
            for (int j = 0; j < D / 32; ++j)      // X^T[d, rows 4hi + {0..3, 8..11}] of the 16-row group: the rows whose W this lane holds
                out[j] = p3attn::mfma_bf16(fa.t.frag(rowimg, sub * 32 + 16 * c2, j), wb, out[j]);
        }
    } else if constexpr (TL::X3) {
        const bf16_t* ih = reinterpret_cast<const bf16_t*>(rowimg);
        const bf16_t* il = ih + R * D;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = w[8 * c2 + i];
            s16x8 wh, wl;
            p3attn::split8(wv, wh, wl);
#pragma unroll
            for (int j = 0; j < D / 32; ++j) {
                const s16x8 xh = fa.t.frag(ih, sub * 32 + 16 * c2, j), xl = fa.t.frag(il, sub * 32 + 16 * c2, j);
                out[j] = p3attn::mfma_bf16(xl, wh, out[j]);
                out[j] = p3attn::mfma_bf16(xh, wl, out[j]);
                out[j] = p3attn::mfma_bf16(xh, wh, out[j]);
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = sub * 32 + crow32(r, hi);
#pragma unroll
            for (int j = 0; j < D / 32; ++j) {
                const float xf = reinterpret_cast<const float*>(rowimg)[row * TL::PR + j * 32 + l31];
                out[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(xf, w[r], out[j], 0, 0, 0);
            }
        }
    }
}

template <typename T, int D>
__device__ __forceinline__ void load_rowfrags(const T* base, int64_t row, int row_stride, s16x8 (&yb)[D / 16 > 0 ? D / 16 : 1], s16x8 (&yl)[D / 16 > 0 ? D / 16 : 1],
                                              float (&yf)[D / 2], int hi) {
    if constexpr (Kind<T>::BF) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) yb[ks] = *reinterpret_cast<const s16x8*>(base + row * row_stride + ks * 16 + 8 * hi);
    } else if constexpr (Kind<T>::X3) {
        const float* fb = reinterpret_cast<const float*>(base) + row * row_stride + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(fb + ks * 16), b = *reinterpret_cast<const float4*>(fb + ks * 16 + 4);
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            p3attn::split8(x, yb[ks], yl[ks]);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < D / 2; ++ks) yf[ks] = base[row * row_stride + 2 * ks + hi];
    }
}

template <typename T, int D>
__device__ __forceinline__ void store_T_acc(T* dst_row, const f32x16 (&acc)[D / 32], float mul, int hi, int64_t planes_lo = 0) {
    // acc[j][r] holds element d = j*32 + crow32(r, hi) of this lane's row
    // planes_lo != 0 (p3_attn_desc.grad_planes): dst_row is a bf16 hi-plane row and the lo plane lies planes_lo elements behind it - the gradient leaves as the
    // operand of the qkv projection's dX / dW GEMMs on planes (p3_gemm_x3 / p3_gemm_tn_x3), no fp32 tensor and no conversion pass in between
#pragma unroll
    for (int j = 0; j < D / 32; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int dd = j * 32 + 8 * rg + 4 * hi;
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = acc[j][4 * rg + i] * mul;
            if (planes_lo != 0) {
                bf16_t* ph = reinterpret_cast<bf16_t*>(dst_row) + dd;
                uint2 h, l;
                h.x = pack_bf2(o[0], o[1]); h.y = pack_bf2(o[2], o[3]);
                l.x = pack_bf2(o[0] - __uint_as_float(h.x << 16), o[1] - __uint_as_float(h.x & 0xffff0000u));
                l.y = pack_bf2(o[2] - __uint_as_float(h.y << 16), o[3] - __uint_as_float(h.y & 0xffff0000u));
                *reinterpret_cast<uint2*>(ph) = h;
                *reinterpret_cast<uint2*>(ph + planes_lo) = l;
                continue;
            }
            if constexpr (Kind<T>::BF) {
                uint2 pk; pk.x = pack_bf2(o[0], o[1]); pk.y = pack_bf2(o[2], o[3]);
                *reinterpret_cast<uint2*>(dst_row + dd) = pk;
            } else {
                *reinterpret_cast<float4*>(dst_row + dd) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
}

// ------------------------------------------------------------------------------------------------ dQ
template <typename T, int D, int DROP>
__global__ __launch_bounds__(256, (D == 32 && !Kind<T>::X3 ? 3 : 2)) void attn_bwd_dq_kernel(BwdArgs a) {
    constexpr bool BF = Kind<T>::BF, X3 = Kind<T>::X3;
    constexpr int KT = Kind<T>::IMG ? 64 : 32;
    using TK = Tile<T, D, KT>;
    __shared__ __attribute__((aligned(16))) T Krow[TK::ROW_ELEMS];
    __shared__ __attribute__((aligned(16))) T Vrow[TK::ROW_ELEMS];
    __shared__ float Kb[KT];                                 // key bias of the staged tile, log2 units
    const p3_attn_desc& d = a.d;
    const int nqb = (d.Lq + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);      // blocks of one (batch, head) share an XCD's L2 (see attention.hip)
    int blk_, pair_;
    attn_block_of(lid, nqb, d.Lq, d.B * d.H, a.order, blk_, pair_);
    const int qblk = blk_ * 128, h = pair_ % d.H, b = pair_ / d.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const T* Qp = reinterpret_cast<const T*>(a.Q) + (int64_t)b * d.q_bs + h * D;
    const T* Kp = reinterpret_cast<const T*>(a.K) + (int64_t)b * d.k_bs + h * D;
    const T* Vp = reinterpret_cast<const T*>(a.V) + (int64_t)b * d.v_bs + h * D;
    const T* dOp = reinterpret_cast<const T*>(a.dO) + (int64_t)b * d.o_bs + h * D;
    const bool gpl = d.grad_planes != 0;          // gradients as bf16 planes with their own strides (g_bs, g_rs) and lo-plane offset g_lo
    T* dQp = gpl ? reinterpret_cast<T*>(reinterpret_cast<bf16_t*>(a.dQ) + (int64_t)b * d.g_bs + h * D) : reinterpret_cast<T*>(a.dQ) + (int64_t)b * d.q_bs + h * D;
    const int q = qblk + wave * 32 + l31;
    const int qc = q < d.Lq ? q : d.Lq - 1;
    s16x8 qb[D / 16], gb[D / 16], ql[D / 16], gl[D / 16];       // ql / gl: the lo fragments of the fp32x3 mode
    float qf[D / 2], gf[D / 2];
    load_rowfrags<T, D>(Qp, qc, d.q_rs, qb, ql, qf, hi);
    load_rowfrags<T, D>(dOp, qc, d.o_rs, gb, gl, gf, hi);
    constexpr float LOG2E = 1.4426950408889634f;
    const float lse2 = a.lse[((int64_t)b * d.H + h) * d.Lq + qc] * LOG2E;    // log2 domain: p = exp2(s*scale*log2e + bias*log2e - lse2)
    const float c2 = d.scale * LOG2E;
    // delta[q] = sum_d dO[q,d] * O[q,d], computed here from the same row fragments the lane already holds for dO (each half-wave
    // lane has 32 of the 64 / 16 of the 32 elements) and published for the dK/dV kernel, which runs after this one on the stream:
    // no separate delta kernel (r01: 24 launches, 1.1 ms per step).
    float dlt;
    {
        const T* Op = reinterpret_cast<const T*>(a.O) + (int64_t)b * d.o_bs + h * D;
        float part = 0.f;
        if constexpr (BF) {
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks) {
                const s16x8 ob = *reinterpret_cast<const s16x8*>(Op + (int64_t)qc * d.o_rs + ks * 16 + 8 * hi);
#pragma unroll
                for (int e = 0; e < 8; ++e) part += bf2f((bf16_t)ob[e]) * bf2f((bf16_t)gb[ks][e]);
            }
        } else if constexpr (X3) {                      // dO = hi + lo (to 2^-18), O read as fp32
            const float* of = reinterpret_cast<const float*>(Op) + (int64_t)qc * d.o_rs + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < D / 16; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) part += of[ks * 16 + e] * (bf2f((bf16_t)gb[ks][e]) + bf2f((bf16_t)gl[ks][e]));
        } else {
#pragma unroll
            for (int ks = 0; ks < D / 2; ++ks) part += Op[(int64_t)qc * d.o_rs + 2 * ks + hi] * gf[ks];
        }
        dlt = part + __shfl_xor(part, 32, 64);
        if (hi == 0 && q < d.Lq) a.delta[((int64_t)b * d.H + h) * d.Lq + q] = dlt;
    }
    f32x16 dq[D / 32];
#pragma unroll
    for (int j = 0; j < D / 32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[j][r] = 0.f;
    int kv_end = d.Lk;
    if (d.causal) { const int lim = qblk + 128; if (lim < kv_end) kv_end = lim; }
    const int ntiles = (kv_end + KT - 1) / KT;
    const float* kbias = d.key_bias ? d.key_bias + (int64_t)b * d.Lk : nullptr;
    DropKey dkey; uint32_t drop_rk = 0;
    if constexpr (DROP) { dkey = drop_key(d.drop); drop_rk = drop_rowkey(dkey, (uint64_t)((int64_t)b * d.H + h) * (uint64_t)d.Lq + (uint64_t)qc); }
    Stage<T, D, KT> kreg, vreg;
    // fp32x3: K / V rows by LDS-DMA into raw fp32 images (dynamic LDS), split at the tile switch (attn_tile.h SplitDma) - no staging registers across the tile
    using SD = p3attn::SplitDma<X3 ? D : 64, 64>;
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_raw[];
    const uint32_t raw_lds = X3 ? (uint32_t)(uintptr_t)attn_raw : 0u;
    FragAddr<T, D> fa; fa.init(lane, l31, hi);
    float kbreg = 0.f;                               // the next tile's key bias rides with its prefetch (see attention.hip)
    auto load_tile = [&](int row0) __attribute__((always_inline)) {
        if constexpr (X3) {
            SD::issue(reinterpret_cast<const float*>(Kp), row0, d.Lk, d.k_rs, raw_lds, wave, lane);
            SD::issue(reinterpret_cast<const float*>(Vp), row0, d.Lk, d.v_rs, raw_lds + SD::RAW_B, wave, lane);
        } else {
            kreg.load(Kp, row0, d.Lk, d.k_rs, tid); vreg.load(Vp, row0, d.Lk, d.v_rs, tid);
        }
        if (kbias && tid < KT) { const int kvb = row0 + tid; kbreg = kbias[kvb < d.Lk ? kvb : d.Lk - 1]; }
    };
    if (ntiles > 0) load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        const int kv0 = t * KT;
        if constexpr (X3) p3attn::wait_vm0();
        __syncthreads();
        if constexpr (X3) { SD::split_store(attn_raw, reinterpret_cast<bf16_t*>(Krow), tid); SD::split_store(attn_raw + SD::RAW_B, reinterpret_cast<bf16_t*>(Vrow), tid); }
        else { kreg.store(Krow, tid); vreg.store(Vrow, tid); }
        if (kbias && tid < KT) Kb[tid] = kbreg * LOG2E;
        __syncthreads();
        // keep-bit words of this tile's 32-key halves: loaded BEFORE the next tile's prefetch is issued - vmcnt retires in order, so a
        // small load issued after the prefetch would make its consumer wait for the whole prefetch (the ISA showed vmcnt(0) there)
        uint32_t dwords[KT / 32];
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            dwords[sub] = 0;
            if constexpr (DROP == 2) {
                const int nkw = (d.Lk + 31) >> 5, kw = (kv0 >> 5) + sub;     // kw == nkw: the all-masked half of the last tile
                if (kw < nkw) dwords[sub] = d.drop_rows[(((int64_t)b * d.H + h) * nkw + kw) * d.Lq + qc];
            }
        }
        if (t + 1 < ntiles) load_tile(kv0 + KT);
        if (qblk + wave * 32 >= d.Lq) continue;      // tail q-block: this wave has no live query, it only stages
        int vis_end = kv_end;                        // keys this wave's queries can see (causal: up to its last query)
        if (d.causal) { const int wl = qblk + wave * 32 + 32; if (wl < vis_end) vis_end = wl; }
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            if (kv0 + sub * 32 >= vis_end) continue;                          // 32-key half with no visible key: dS == 0
            f32x16 s, dp;                                                         // S^T[kv, q], dP^T[kv, q]
            if constexpr (X3 && P3_ATTN_PAIR) {
                score_mma_pair<T, D, KT>(Krow, Vrow, sub, qb, ql, gb, gl, fa, s, dp);
            } else {
                s = score_mma<T, D, KT>(Krow, sub, qb, ql, qf, fa, l31, hi);
                dp = score_mma<T, D, KT>(Vrow, sub, gb, gl, gf, fa, l31, hi);
            }
            // tile entirely inside [0, Lk), below the causal diagonal of this wave's first query, no key bias: no per-element masks
            const bool full = (kv0 + sub * 32 + 32 <= d.Lk) && (!d.causal || kv0 + sub * 32 + 31 <= qblk + wave * 32) && !kbias;
            const uint32_t dword = dwords[sub];
            // FULL tiles run WITHOUT the mask arithmetic: written as `if (full) ... else ...` per element the compiler if-converts both sides
            // into compares + selects that every tile pays (ISA r02: ~230 mask instructions per 64-query tile, 40 % of the VALU work)
            auto elements = [&](auto FULLT) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + sub * 32 + crow32(r, hi);
                    float p;
                    if constexpr (FULL) {
                        p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -lse2));
                    } else {
                        float t2 = -lse2;
                        if (kbias) t2 += Kb[sub * 32 + crow32(r, hi)];
                        const bool masked = kv >= d.Lk || (d.causal && kv > q);
                        p = masked ? 0.f : __builtin_amdgcn_exp2f(fmaf(s[r], c2, t2));
                    }
                    float dpv = dp[r];
                    if constexpr (DROP == 2) {
                        dpv = ((dword >> crow32(r, hi)) & 1u) ? dpv * dkey.inv_keep : 0.f;   // mask published by the forward kernel
                    } else if constexpr (DROP == 1) {
                        const uint32_t bits = drop_bits(drop_rk, drop_colkey(dkey, (uint32_t)kv));   // pairs (r, r+1) share it: CSE'd
                        dpv = ((r & 1) ? drop_keep_hi(dkey, bits) : drop_keep_lo(dkey, bits)) ? dpv * dkey.inv_keep : 0.f;
                    }
                    s[r] = p * (dpv - dlt);                                   // dS^T
                }
            };
            if (__builtin_amdgcn_readfirstlane((int)full)) elements(std::true_type{}); else elements(std::false_type{});
            accum_mma<T, D, KT>(Krow, sub, s, dq, fa, l31, hi);               // dQ^T[d, q] += K^T . dS^T
        }
    }
    if (q < d.Lq) {
        if (gpl) store_T_acc<T, D>(reinterpret_cast<T*>(reinterpret_cast<bf16_t*>(dQp) + (int64_t)q * d.g_rs), dq, d.scale, hi, d.g_lo);
        else store_T_acc<T, D>(dQp + (int64_t)q * d.q_rs, dq, d.scale, hi);
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <typename T, int D, int DROP>
__global__ __launch_bounds__(256, (D == 32 && !Kind<T>::X3 ? P3_DKV32_WAVES : 2)) void attn_bwd_dkv_kernel(BwdArgs a) {
    constexpr int QT = Kind<T>::IMG ? 64 : 32;
    using TQ = Tile<T, D, QT>;
    __shared__ __attribute__((aligned(16))) T Qrow[TQ::ROW_ELEMS];
    __shared__ __attribute__((aligned(16))) T Grow[TQ::ROW_ELEMS];
    __shared__ float Ls[QT], Ds[QT];
    const p3_attn_desc& d = a.d;
    const int nkb = (d.Lk + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int blk_, pair_;
    attn_block_of(lid, nkb, d.Lk, d.B * d.H, a.order, blk_, pair_);
    const int kblk = blk_ * 128, h = pair_ % d.H, b = pair_ / d.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const T* Qp = reinterpret_cast<const T*>(a.Q) + (int64_t)b * d.q_bs + h * D;
    const T* Kp = reinterpret_cast<const T*>(a.K) + (int64_t)b * d.k_bs + h * D;
    const T* Vp = reinterpret_cast<const T*>(a.V) + (int64_t)b * d.v_bs + h * D;
    const T* dOp = reinterpret_cast<const T*>(a.dO) + (int64_t)b * d.o_bs + h * D;
    const bool gpl = d.grad_planes != 0;
    T* dKp = gpl ? reinterpret_cast<T*>(reinterpret_cast<bf16_t*>(a.dK) + (int64_t)b * d.g_bs + h * D) : reinterpret_cast<T*>(a.dK) + (int64_t)b * d.k_bs + h * D;
    T* dVp = gpl ? reinterpret_cast<T*>(reinterpret_cast<bf16_t*>(a.dV) + (int64_t)b * d.g_bs + h * D) : reinterpret_cast<T*>(a.dV) + (int64_t)b * d.v_bs + h * D;
    const int kv = kblk + wave * 32 + l31;
    const int kvc = kv < d.Lk ? kv : d.Lk - 1;
    s16x8 kb[D / 16], vb[D / 16], kl[D / 16], vl[D / 16];
    float kf[D / 2], vf[D / 2];
    load_rowfrags<T, D>(Kp, kvc, d.k_rs, kb, kl, kf, hi);
    load_rowfrags<T, D>(Vp, kvc, d.v_rs, vb, vl, vf, hi);
    const float bias2 = (d.key_bias ? d.key_bias[(int64_t)b * d.Lk + kvc] : 0.f) * 1.4426950408889634f;
    const float c2 = d.scale * 1.4426950408889634f;
    f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int j = 0; j < D / 32; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[j][r] = 0.f; dv[j][r] = 0.f; }
    const int q_begin = d.causal ? (kblk / QT) * QT : 0;   // queries before the block's first key never see it
    DropKey dkey; uint32_t drop_ck = 0; uint64_t drop_bh = 0;
    if constexpr (DROP != 0) { dkey = drop_key(d.drop); drop_ck = drop_colkey(dkey, (uint32_t)kvc); drop_bh = (uint64_t)((int64_t)b * d.H + h) * (uint64_t)d.Lq; }
    const int64_t stat_base = ((int64_t)b * d.H + h) * d.Lq;
    Stage<T, D, QT> qreg, greg;
    constexpr bool X3 = Kind<T>::X3;
    using SD = p3attn::SplitDma<X3 ? D : 64, 64>;       // fp32x3: Q / dO rows by LDS-DMA into raw images, split at the tile switch (see the dQ kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_raw[];
    const uint32_t raw_lds = X3 ? (uint32_t)(uintptr_t)attn_raw : 0u;
    auto load_tile = [&](int row0) __attribute__((always_inline)) {
        if constexpr (X3) {
            SD::issue(reinterpret_cast<const float*>(Qp), row0, d.Lq, d.q_rs, raw_lds, wave, lane);
            SD::issue(reinterpret_cast<const float*>(dOp), row0, d.Lq, d.o_rs, raw_lds + SD::RAW_B, wave, lane);
        } else {
            qreg.load(Qp, row0, d.Lq, d.q_rs, tid); greg.load(dOp, row0, d.Lq, d.o_rs, tid);
        }
    };
    FragAddr<T, D> fa; fa.init(lane, l31, hi);
    // the tile's log-sum-exp / delta rows ride with the prefetch (registers of the first QT threads): loaded inside the staging phase they were a global-load latency
    // in front of barrier 2 in every tile - 20 % of this kernel's loop time (profiles/r06_attn_phases.txt)
    float lreg = 0.f, dreg = 0.f;
    auto load_stats = [&](int qrow0) __attribute__((always_inline)) {
        if (tid < QT) { const int qq = qrow0 + tid < d.Lq ? qrow0 + tid : d.Lq - 1; lreg = a.lse[stat_base + qq]; dreg = a.delta[stat_base + qq]; }
    };
    if (q_begin < d.Lq) { load_tile(q_begin); load_stats(q_begin); }
    const int64_t word_row = (((int64_t)b * d.H + h) * ((d.Lk + 31) >> 5) + ((kblk + wave * 32) >> 5)) * d.Lq;
    // -DP3_ATTN_TIMING (diagnostic build, tools/mb_attn_phases.py, as in attention.hip): [0] barrier 1, [1] split + LDS stores, [2] barrier 2, [3] next tile's loads
    // issued, [4] S and dP, [5] element-wise, [6] dV / dK products + loop tail, [7] 32-query halves counted
#ifdef P3_ATTN_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define TM(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tacc[k] += n_ - tprev; tprev = n_; } while (0)
#else
#define TM(k) do { } while (0)
#endif
    for (int q0 = q_begin; q0 < d.Lq; q0 += QT) {
        TM(6);
        if constexpr (X3) p3attn::wait_vm0();
        __syncthreads();
        TM(0);
        if constexpr (X3) { SD::split_store(attn_raw, reinterpret_cast<bf16_t*>(Qrow), tid); SD::split_store(attn_raw + SD::RAW_B, reinterpret_cast<bf16_t*>(Grow), tid); }
        else { qreg.store(Qrow, tid); greg.store(Grow, tid); }
        if (tid < QT) { Ls[tid] = lreg * 1.4426950408889634f; Ds[tid] = dreg; }   // lse in log2 units
        TM(1);
        __syncthreads();
        TM(2);
        // DROP == 2: lane L loads the keep-bit word of query row q0 + sub*32 + (L & 31) for this wave's 32-key block, for both halves of
        // the tile and BEFORE the next tile's prefetch (vmcnt retires in order: a load issued after the prefetch waits for all of it)
        uint32_t mywords[QT / 32];
#pragma unroll
        for (int sub = 0; sub < QT / 32; ++sub) {
            mywords[sub] = 0;
            if constexpr (DROP == 2) {
                const int qr = q0 + sub * 32 + l31;
                if (kblk + wave * 32 < d.Lk) mywords[sub] = d.drop_rows[word_row + (qr < d.Lq ? qr : d.Lq - 1)];
            }
        }
        if (q0 + QT < d.Lq) { load_tile(q0 + QT); load_stats(q0 + QT); }
        TM(3);
        if (kblk + wave * 32 >= d.Lk) continue;      // tail k-block: this wave has no live key, it only stages
#pragma unroll
        for (int sub = 0; sub < QT / 32; ++sub) {
            // 32-query half beyond Lq, or (causal) entirely before this wave's first key: P == dS == 0
            if (q0 + sub * 32 >= d.Lq || (d.causal && q0 + sub * 32 + 31 < kblk + wave * 32)) continue;
            f32x16 s, dp;                                                         // S[q, kv], dP[q, kv]
            if constexpr (X3 && P3_ATTN_PAIR) {
                score_mma_pair<T, D, QT>(Qrow, Grow, sub, kb, kl, vb, vl, fa, s, dp);
            } else {
                s = score_mma<T, D, QT>(Qrow, sub, kb, kl, kf, fa, l31, hi);
                dp = score_mma<T, D, QT>(Grow, sub, vb, vl, vf, fa, l31, hi);
            }
            TM(4);
#ifdef P3_ATTN_TIMING
            tacc[7] += 1;
#endif
            f32x16 ds;
            // all 32 queries of the sub-tile and all 32 keys of the wave valid, non-causal: no per-element masks
            const bool full = (q0 + sub * 32 + 32 <= d.Lq) && (kblk + wave * 32 + 32 <= d.Lk) && !d.causal;
            // element r needs the word of row crow32(r, hi), fetched from that lane with two uniform-index readlanes + a select on hi
            const uint32_t myword = mywords[sub];
            auto elements = [&](auto FULLT) __attribute__((always_inline)) {       // see the dQ kernel: full tiles skip the mask arithmetic
                constexpr bool FULL = decltype(FULLT)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = sub * 32 + crow32(r, hi), qq = q0 + ql;
                    float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, bias2 - Ls[ql]));
                    if constexpr (!FULL) { const bool masked = qq >= d.Lq || kv >= d.Lk || (d.causal && kv > qq); p = masked ? 0.f : p; }
                    float pd = p, dpv = dp[r];
                    if constexpr (DROP != 0) {
                        bool keep;
                        if constexpr (DROP == 2) {
                            const uint32_t w0 = __builtin_amdgcn_readlane(myword, crow32(r, 0)), w1 = __builtin_amdgcn_readlane(myword, crow32(r, 1));
                            keep = ((hi ? w1 : w0) >> l31) & 1u;
                        } else {
                            const int64_t qrow = (int64_t)drop_bh + (qq < d.Lq ? qq : d.Lq - 1);
                            const uint32_t bits = drop_bits(drop_rowkey(dkey, (uint64_t)qrow), drop_ck);
                            keep = (kvc & 1) ? drop_keep_hi(dkey, bits) : drop_keep_lo(dkey, bits);
                        }
                        pd = keep ? p * dkey.inv_keep : 0.f;
                        dpv = keep ? dpv * dkey.inv_keep : 0.f;
                    }
                    s[r] = pd;
                    ds[r] = p * (dpv - Ds[ql]);
                }
            };
            if (__builtin_amdgcn_readfirstlane((int)full)) elements(std::true_type{}); else elements(std::false_type{});
            TM(5);
            accum_mma<T, D, QT>(Grow, sub, s, dv, fa, l31, hi);               // dV^T[d, kv] += dO^T . P
            accum_mma<T, D, QT>(Qrow, sub, ds, dk, fa, l31, hi);              // dK^T[d, kv] += Q^T . dS
            TM(6);
        }
    }
#ifdef P3_ATTN_TIMING
    if (lane == 0 && d.drop_rows && kblk + wave * 32 < d.Lk)
        for (int k = 0; k < 8; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(d.drop_rows) + 8 + k, tacc[k]);
#endif
#undef TM
    if (kv < d.Lk) {
        if (gpl) {
            store_T_acc<T, D>(reinterpret_cast<T*>(reinterpret_cast<bf16_t*>(dKp) + (int64_t)kv * d.g_rs), dk, d.scale, hi, d.g_lo);
            store_T_acc<T, D>(reinterpret_cast<T*>(reinterpret_cast<bf16_t*>(dVp) + (int64_t)kv * d.g_rs), dv, 1.f, hi, d.g_lo);
        } else {
            store_T_acc<T, D>(dKp + (int64_t)kv * d.k_rs, dk, d.scale, hi);
            store_T_acc<T, D>(dVp + (int64_t)kv * d.v_rs, dv, 1.f, hi);
        }
    }
}

template <typename T, int D>
int launch_bwd(const BwdArgs& a, hipStream_t s) {
    const p3_attn_desc& d = a.d;
    const int64_t rows = (int64_t)d.B * d.H * d.Lq;
    int g = (int)((rows + 3) / 4); if (g > 4096) g = 4096;
    (void)g;   // delta is produced by the dQ kernel (attn_delta_kernel kept for reference / standalone use)
    const dim3 gq(p3_ceil_div(d.Lq, 128) * d.H * d.B), gk(p3_ceil_div(d.Lk, 128) * d.H * d.B), blk(256);
    // fp32x3: the raw fp32 images of the two staged operands in dynamic LDS (with the 32 KB of bf16 images above the 64 KB a kernel gets without asking for D = 64)
    const size_t dyn = Kind<T>::X3 ? (size_t)2 * 64 * D * 4 : 0;
    if constexpr (Kind<T>::X3 && D == 64) {
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<T, D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T, D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T, D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<T, D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            attr_set = true;
        }
    }
    if (d.drop.seed != nullptr && d.drop.p > 0.f) {
        if (d.drop_rows) {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, 2>), gq, blk, dyn, s, a);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, 2>), gk, blk, dyn, s, a);
        } else {
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, 1>), gq, blk, dyn, s, a);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, 1>), gk, blk, dyn, s, a);
        }
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, 0>), gq, blk, dyn, s, a);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, D, 0>), gk, blk, dyn, s, a);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

}  // namespace

extern "C" int p3_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse, void* dQ,
                                void* dK, void* dV, float* delta_ws, const p3_attn_desc* d, void* stream) {
    P3_CHECK(Q && K && V && O && dO && lse && dQ && dK && dV && delta_ws && d, P3_EINVAL, "p3_attention_bwd: null pointer");
    P3_CHECK(d->head_dim == 32 || d->head_dim == 64, P3_EUNSUP, "p3_attention_bwd: head_dim must be 32 or 64");
    P3_CHECK(d->dtype == P3_F32 || d->dtype == P3_BF16 || d->dtype == P3_F32X3, P3_EUNSUP, "p3_attention_bwd: dtype");
    const int al = d->dtype == P3_BF16 ? 8 : 4;
    P3_CHECK(d->q_rs % al == 0 && d->k_rs % al == 0 && d->v_rs % al == 0 && d->o_rs % al == 0, P3_EALIGN, "p3_attention_bwd: row strides");
    P3_CHECK(!d->grad_planes || (d->dtype == P3_F32X3 && d->g_lo != 0 && d->g_rs % 4 == 0 && d->g_bs % 4 == 0 && d->g_lo % 4 == 0 && ((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) % 8 == 0),
             P3_EINVAL, "p3_attention_bwd: grad_planes goes with P3_F32X3 and needs g_bs / g_rs / g_lo (multiples of 4 bf16 elements)");
    BwdArgs a; a.Q = Q; a.K = K; a.V = V; a.O = O; a.dO = dO; a.dQ = dQ; a.dK = dK; a.dV = dV; a.lse = lse; a.delta = delta_ws; a.d = *d; a.order = p3_attn_order();
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == P3_BF16) return d->head_dim == 64 ? launch_bwd<bf16_t, 64>(a, s) : launch_bwd<bf16_t, 32>(a, s);
    if (d->dtype == P3_F32X3) return d->head_dim == 64 ? launch_bwd<f32s, 64>(a, s) : launch_bwd<f32s, 32>(a, s);      // fp32x3 mode: bf16 x 3 products (attn_tile.h)
    return d->head_dim == 64 ? launch_bwd<float, 64>(a, s) : launch_bwd<float, 32>(a, s);
}
