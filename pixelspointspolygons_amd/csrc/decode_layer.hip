// p3_decode_layer: ONE launch per nn.TransformerDecoderLayer for the greedy decode step (one new position per sample, KV caches),
// the inner body of Decoder.predict's loop (model_pix2poly.py:187-219; post-norm layer, ReLU feed-forward, eval mode).
//
// The unfused step is 11 dependent launches per layer (in_proj, attention, out_proj + residual, LayerNorm, q proj, attention, out_proj,
// LayerNorm, linear1 + ReLU, linear2 + residual, LayerNorm) of 4-15 us each: launch latency, not work, is what the 385-step loop pays.
// Here a CLUSTER of C workgroups (512 threads each; C = 4, or 1) owns one sample and walks the whole layer:
//   * heads and hidden units are split over the cluster: member c owns heads [8c/C, 8(c+1)/C) = channels [256c/C, ...) and hidden units
//     [2048c/C, ...).  It computes the q|k|v rows of ITS heads, attends with them, and multiplies its slice of the attention output into
//     a PARTIAL out_proj result over all 256 rows (split over K, not over rows) - so nothing is exchanged before the projection; the
//     feed-forward is split the same way (own linear1 rows -> own hidden slice -> partial linear2 over that K slice);
//   * the three partial [256] vectors per layer (self-attention, cross-attention, feed-forward) are exchanged through global memory:
//     plain stores, a sense-reversing cluster barrier (one counter + one generation word per sample, agent-scope fences), then every
//     member sums the C partials in member order (deterministic) and computes bias + residual + LayerNorm redundantly for the full row;
//   * the members of a cluster sit on ONE XCD (workgroup id -> XCD is round robin: id = (k*C + c)*8 + xcd), so the exchange stays in
//     that XCD's L2; B*C <= 512 workgroups are co-resident (2 per CU fit), and a spin limit turns a barrier that never completes into an
//     error flag instead of a hang;
//   * GEMV: the lanes of a half-wave (K = 256 blocks) or of 8 lanes (64-wide K slices) share a weight row, 4-16 rows per lane in flight
//     (16-byte coalesced loads, v_dot2_f32_bf16 against the activation chunk held in registers), partial sums folded by a multi-value
//     butterfly (about one shuffle per row; compile-time recursion - a run-time-indexed version cost 4x the whole kernel in selects);
//     attention (cluster form): 8 threads per key (2 heads x 4 chunks), the K and V chunks of all <= 13 passes of 64 keys loaded in one
//     round, scores / probabilities kept in registers (the score thread of (key, chunk) is the P.V thread of (key, chunk)), softmax
//     maxima / sums by shuffles + one LDS hop; LayerNorm: two-pass mean / centred variance.
// Measured (MI355X, 385 steps, hipGraph replay per step): B = 1 111 -> 73 ms, B = 64 235 -> 126 ms against the 11-launch chain; plain
// launches B = 1: 509 -> 111 ms.  Stage times of one layer at B = 1: GEMVs 1.1-3.2 us, attention 4.4 us each, exchange + LayerNorm 2.4-2.9 us.
// Activations between stages are rounded to bf16 exactly where the unfused bf16 path stores them (q|k|v rows, attention outputs,
// LayerNorm outputs, the hidden layer), accumulation is fp32.  bf16 weights, D = 256, 8 heads, FF = 2048 (the reference decoder).
#include "p3_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));

constexpr int DL_D = 256, DL_H = 8, DL_DH = 32, DL_FF = 2048, DL_NT = 512, DL_NW = DL_NT / 64;
constexpr int DL_MAXK = 832;            // longest key sequence (LDS score rows): 784 memory tokens, 385 positions
constexpr unsigned DL_SPIN_LIMIT = 1u << 21;

__device__ __forceinline__ float dot2(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w), __builtin_bit_cast(bf2_t, x), acc, false);
}
__device__ __forceinline__ float dot8(const u32x4& w, const u32x4& x, float acc) {
    acc = dot2(w.x, x.x, acc); acc = dot2(w.y, x.y, acc); acc = dot2(w.z, x.z, acc); return dot2(w.w, x.w, acc);
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float round_bf(float v) { return bf2f(f2bf(v)); }
// value of lane ^ 1 (CTRL 0xB1 = quad_perm [1,0,3,2]) / lane ^ 2 (0x4E = [2,3,0,1]): a DPP move instead of an LDS-crossbar shuffle
template <int CTRL> __device__ __forceinline__ float quad_xor(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }

// Sum R per-lane values over a group of LPR lanes (lane index l inside the group): every level that still has > 1 value per lane halves
// the values (a lane keeps the half selected by its bit, the partner's half arrives by shuffle), the remaining levels are plain sums.
// Returns the total of row `row` (valid on the lanes whose low, plain-level bits are zero; `writer`).  Compile-time recursion: every
// register index is a constant.
template <int N, int M>
struct Fold {
    __device__ static __forceinline__ float run(float (&v)[N], int l, int& row, bool& writer) {
        const bool b = (l & M) != 0;
        if constexpr (N > 1) {
            float h[N / 2];
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const float give = b ? v[i] : v[i + N / 2], keep = b ? v[i + N / 2] : v[i];
                h[i] = keep + __shfl_xor(give, M, 64);
            }
            row += b ? N / 2 : 0;
            if constexpr (M > 1) return Fold<N / 2, M / 2>::run(h, l, row, writer);
            else return h[0];
        } else {
            float t[1] = {v[0] + __shfl_xor(v[0], M, 64)};
            writer = writer && !b;
            if constexpr (M > 1) return Fold<1, M / 2>::run(t, l, row, writer);
            else return t[0];
        }
    }
};
template <int R, int LPR>
__device__ __forceinline__ float fold_rows(float (&v)[R], int l, int& row, bool& writer) {
    row = 0; writer = true;
    return Fold<R, LPR / 2>::run(v, l, row, writer);
}

// out[i] = sum_k W[row(i)][k] * x[k] for i in [0, nrows) (nrows % 32 == 0), row(i) = row0 + (i / seg) * seg_stride + i % seg (seg % 16 == 0:
// `seg`-row segments `seg_stride` rows apart - the q / k / v rows of one member's heads; seg = nrows for one contiguous block).  Row pitch
// `pitch` elements, K = 256 * KB columns starting at the row's first element; x bf16 in LDS.  Half-wave per row, 16 rows per half-wave
// in flight.
template <int KB>
__device__ __forceinline__ void gemv_rows(const bf16_t* __restrict__ W, int pitch, int row0, int seg, int seg_stride, int nrows, const bf16_t* xs,
                                          float* out, int tid) {
    const int wave = tid >> 6, lane = tid & 63, hl = lane & 31, half = lane >> 5;
    for (int nb = wave * 32; nb < nrows; nb += DL_NW * 32) {
        const int i0 = nb + half * 16;
        const bf16_t* wrow = W + (int64_t)(row0 + (i0 / seg) * seg_stride + i0 % seg) * pitch + 8 * hl;
        float acc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 1
        for (int kb = 0; kb < KB; ++kb) {
            u32x4 wv[16];
            const bf16_t* wp = wrow + kb * 256;
#pragma unroll
            for (int r = 0; r < 16; ++r) { wv[r] = *reinterpret_cast<const u32x4*>(wp); wp += pitch; }
            const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + kb * 256 + 8 * hl);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = dot8(wv[r], xv, acc[r]);
        }
        int row; bool wr;
        const float tot = fold_rows<16, 32>(acc, hl, row, wr);
        if (wr) out[i0 + row] = tot;
    }
}

// out[n] = sum_{k < 64} W[n][col0 + k] * x[k] for all 256 rows n (row pitch 256): 8 lanes per row, 4 rows per lane in flight
__device__ __forceinline__ void gemv_kslice64(const bf16_t* __restrict__ W, int col0, const bf16_t* xs, float* out, int tid) {
    const int wave = tid >> 6, lane = tid & 63, l8 = lane & 7, grp = lane >> 3;      // 8 groups of 8 lanes per wave
    const int nb = wave * 32;                                                        // 8 waves x 32 rows = 256 rows
    const bf16_t* wrow = W + (int64_t)(nb + grp * 4) * DL_D + col0 + 8 * l8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + 8 * l8);
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = dot8(*reinterpret_cast<const u32x4*>(wrow + r * DL_D), xv, 0.f);
    int row; bool wr;
    const float tot = fold_rows<4, 8>(acc, l8, row, wr);
    if (wr) out[nb + grp * 4 + row] = tot;
}

// ---- element type of weights / activations / caches: bf16 (throughput mode) or fp32 (parity mode, r03) -----------------------------------
// fp32 keeps every value unrounded between the stages (where the bf16 form rounds to bf16 exactly like the unfused bf16 path stores) and
// multiplies with fmaf chains; the summation ORDER differs from the MFMA GEMMs of the unfused fp32 path (lane-split dot products folded by
// shuffles), so features agree to fp32 rounding (1e-6), not bit for bit - tokens are compared against the goldens (tests/test_decode_layer_gpu.py).
template <typename T> struct DT;
template <> struct DT<bf16_t> {
    static __device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
        const u32x4 r4 = *reinterpret_cast<const u32x4*>(p);
        v[0] = bf_lo(r4.x); v[1] = bf_hi(r4.x); v[2] = bf_lo(r4.y); v[3] = bf_hi(r4.y); v[4] = bf_lo(r4.z); v[5] = bf_hi(r4.z); v[6] = bf_lo(r4.w); v[7] = bf_hi(r4.w);
    }
    static __device__ __forceinline__ float rnd(float v) { return round_bf(v); }
    static __device__ __forceinline__ bf16_t from_f(float v) { return f2bf(v); }
    static __device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }
};
template <> struct DT<float> {
    static __device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ float rnd(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
    static __device__ __forceinline__ float to_f(float v) { return v; }
};

// fp32 form of gemv_rows: half-wave per row, a lane owns 8 of the 256 k-values of a block (two 16-byte loads), 8 rows per half-wave in flight
template <int KB>
__device__ __forceinline__ void gemv_rows(const float* __restrict__ W, int pitch, int row0, int seg, int seg_stride, int nrows, const float* xs,
                                          float* out, int tid) {
    const int wave = tid >> 6, lane = tid & 63, hl = lane & 31, half = lane >> 5;
    for (int nb = wave * 16; nb < nrows; nb += DL_NW * 16) {
        const int i0 = nb + half * 8;
        const float* wrow = W + (int64_t)(row0 + (i0 / seg) * seg_stride + i0 % seg) * pitch + 8 * hl;
        float acc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = 0.f;
#pragma unroll 1
        for (int kb = 0; kb < KB; ++kb) {
            float4 wa[8], wb[8];
            const float* wp = wrow + kb * 256;
#pragma unroll
            for (int r = 0; r < 8; ++r) { wa[r] = *reinterpret_cast<const float4*>(wp); wb[r] = *reinterpret_cast<const float4*>(wp + 4); wp += pitch; }
            const float4 xa = *reinterpret_cast<const float4*>(xs + kb * 256 + 8 * hl), xb = *reinterpret_cast<const float4*>(xs + kb * 256 + 8 * hl + 4);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float a = acc[r];
                a = fmaf(wa[r].x, xa.x, a); a = fmaf(wa[r].y, xa.y, a); a = fmaf(wa[r].z, xa.z, a); a = fmaf(wa[r].w, xa.w, a);
                a = fmaf(wb[r].x, xb.x, a); a = fmaf(wb[r].y, xb.y, a); a = fmaf(wb[r].z, xb.z, a); a = fmaf(wb[r].w, xb.w, a);
                acc[r] = a;
            }
        }
        int row; bool wr;
        const float tot = fold_rows<8, 32>(acc, hl, row, wr);
        if (wr) out[i0 + row] = tot;
    }
}

// fp32 form of gemv_kslice64: 8 lanes per row (8 x 8 = the 64-wide K slice), 4 rows per lane group
__device__ __forceinline__ void gemv_kslice64(const float* __restrict__ W, int col0, const float* xs, float* out, int tid) {
    const int wave = tid >> 6, lane = tid & 63, l8 = lane & 7, grp = lane >> 3;
    const int nb = wave * 32;
    const float* wrow = W + (int64_t)(nb + grp * 4) * DL_D + col0 + 8 * l8;
    const float4 xa = *reinterpret_cast<const float4*>(xs + 8 * l8), xb = *reinterpret_cast<const float4*>(xs + 8 * l8 + 4);
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float4 wa = *reinterpret_cast<const float4*>(wrow + r * DL_D), wb = *reinterpret_cast<const float4*>(wrow + r * DL_D + 4);
        float a = 0.f;
        a = fmaf(wa.x, xa.x, a); a = fmaf(wa.y, xa.y, a); a = fmaf(wa.z, xa.z, a); a = fmaf(wa.w, xa.w, a);
        a = fmaf(wb.x, xb.x, a); a = fmaf(wb.y, xb.y, a); a = fmaf(wb.z, xb.z, a); a = fmaf(wb.w, xb.w, a);
        acc[r] = a;
    }
    int row; bool wr;
    const float tot = fold_rows<4, 8>(acc, l8, row, wr);
    if (wr) out[nb + grp * 4 + row] = tot;
}

template <int C, typename T = bf16_t> struct Smem {
    static constexpr int CW = DL_D / C, HPB = DL_H / C, FW = DL_FF / C;
    __attribute__((aligned(16))) T xs[DL_D];             // current activation row (element type) = the operand of the full-K GEMVs
    __attribute__((aligned(16))) T as[CW];               // this member's slice of the attention output
    __attribute__((aligned(16))) T hb[FW];               // this member's slice of the hidden layer
    float xres[DL_D];                                    // the activation row as fp32 (residual)
    float g[FW > 3 * CW ? FW : 3 * CW];                  // raw GEMV sums
    float part[DL_D];                                    // partial projection over this member's K slice
    float q[CW], kcur[CW], vcur[CW];                     // query (bf16-rounded); this position's key / value (not yet readable from the cache)
    float sc[HPB][DL_MAXK];                              // scores -> probabilities
    float red[DL_NW][CW];                                // P.V partial sums per wave (key slots of a wave are folded by shuffles)
    float inv[HPB];
    float stat[32];
};

// all threads of all members of the cluster have passed when this returns; `sync` = {count, generation} of this sample.
// No agent-scope FENCE (on gfx950 that is an L2 write-back + invalidate: 23 us per barrier with 256 workgroups fencing at once, measured):
// the exchanged vectors are written and read with agent-scope atomic stores / loads (they bypass the per-CU L1), __syncthreads() retires
// the workgroup's stores (a store counts as complete when L2 has it), and the counter / generation words are agent-scope atomics.
__device__ __forceinline__ void cluster_barrier(unsigned* sync, int C, int* err, int tid) {
    __syncthreads();
    if (tid == 0) {
        const unsigned gen = __hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned old = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)C - 1) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                                        // the reset is at L2 before the generation moves
            __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned n = 0;
            while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++n > DL_SPIN_LIMIT) { if (err) *err = 1; break; }            // never hang the device: flag and go on
            }
        }
    }
    __syncthreads();
}

// softmax(q.K^T * scale + bias) . V for this member's heads; K/V rows in global memory (bf16, row stride rs elements, already offset to
// the member's channels), optionally the key / value row `cur` taken from LDS.  Result (bf16) -> s.as.
template <int C, typename T>
__device__ __forceinline__ void attend(Smem<C, T>& s, const T* __restrict__ Kp, const T* __restrict__ Vp, int rs, int Lk, int cur,
                                       const float* __restrict__ kbias, float scale, int tid) {
    constexpr int CW = Smem<C, T>::CW, HPB = Smem<C, T>::HPB, TPK = CW / 8, KPP = DL_NT / TPK;   // threads per key, keys per pass
    constexpr int NB = sizeof(T) == 2 ? 13 : 6;          // passes per batch of loads (fp32 chunks are two 16-byte loads: half the passes in flight)
    const int wave = tid >> 6, lane = tid & 63;
    {   // scores: TPK threads per key, each one 16-byte chunk (4 chunks = one head); the chunk's 8 query values live in registers
        const int part = tid % TPK, js = tid / TPK;
        float qv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) qv[i] = s.q[part * 8 + i];
        // NB passes of KPP keys per batch: all NB loads are issued before the first dot product (bf16: 13 x 64 keys = the 784 memory tokens)
        for (int j0 = 0; j0 < Lk; j0 += KPP * NB) {
            float raw[NB][8];
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int j = j0 + p * KPP + js;
                if (j < Lk && j != cur) DT<T>::load8(Kp + (int64_t)j * rs + part * 8, raw[p]);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) raw[p][i] = 0.f;
                }
            }
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int j = j0 + p * KPP + js;
                float d = 0.f;
                if (j == cur) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) d = fmaf(qv[i], s.kcur[part * 8 + i], d);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) d = fmaf(qv[i], raw[p][i], d);
                }
                d += quad_xor<0xB1>(d); d += quad_xor<0x4E>(d);             // 4 chunks = one head
                if (j < Lk && (part & 3) == 0) s.sc[part >> 2][j] = d * scale + (kbias ? kbias[j] : 0.f);
            }
        }
    }
    __syncthreads();
    if (wave < HPB) {   // softmax: one wave per head
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, s.sc[wave][j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Lk; j += 64) { const float e = __expf(s.sc[wave][j] - mx); s.sc[wave][j] = e; sum += e; }
        sum = wave_sum(sum);
        if (lane == 0) s.inv[wave] = 1.f / sum;
    }
    __syncthreads();
    {   // P.V: thread = (key slot, 8-channel chunk)
        constexpr int SLOTS = DL_NT / TPK;
        const int c = tid % TPK, slot = tid / TPK, h = c >> 2;
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
        for (int j0 = slot; j0 < Lk; j0 += SLOTS * NB) {
            float raw[NB][8];
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int j = j0 + p * SLOTS;
                if (j < Lk && j != cur) DT<T>::load8(Vp + (int64_t)j * rs + c * 8, raw[p]);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) raw[p][i] = 0.f;
                }
            }
#pragma unroll
            for (int p = 0; p < NB; ++p) {
                const int j = j0 + p * SLOTS;
                if (j >= Lk) continue;
                const float pr = s.sc[h][j];
                if (j == cur) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = fmaf(pr, s.vcur[c * 8 + i], o[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = fmaf(pr, raw[p][i], o[i]);
                }
            }
        }
        // fold the key slots that share a wave (lanes TPK apart) before going through LDS: 64 / TPK slots per wave
#pragma unroll
        for (int m = TPK; m < 64; m <<= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += __shfl_xor(o[i], m, 64);
        if (lane < TPK) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s.red[wave][c * 8 + i] = o[i];
        }
    }
    __syncthreads();
    if (tid < CW) {
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < DL_NW; ++w) a += s.red[w][tid];
        s.as[tid] = DT<T>::from_f(a * s.inv[tid >> 5]);
    }
    __syncthreads();
}

// Register form of attend() for the 4-member cluster (8 threads per key: 2 heads x 4 chunks; 64 keys per pass, at most 13 passes): the
// thread that owns chunk c of key j in the score pass is the thread that needs p(j, head(c)) in the P.V pass, so scores, probabilities
// and the K AND V chunks of all passes stay in registers - one round of loads, no LDS score table, the softmax spread over all 8 waves.
__device__ __forceinline__ void attend_reg(Smem<4, bf16_t>& s, const bf16_t* __restrict__ Kp, const bf16_t* __restrict__ Vp, int rs, int Lk, int cur,
                                           const float* __restrict__ kbias, float scale, int tid) {
    constexpr int CW = 64, TPK = 8, KPP = DL_NT / TPK, NB = DL_MAXK / KPP;          // 64 keys per pass, 13 passes
    const int wave = tid >> 6, lane = tid & 63, part = tid % TPK, js = tid / TPK, h = part >> 2;
    u32x4 kr[NB], vr[NB];
    float kbv[NB];
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int j = p * KPP + js;
        const bool ld = j < Lk && j != cur;
        kr[p] = ld ? *reinterpret_cast<const u32x4*>(Kp + (int64_t)j * rs + part * 8) : u32x4{0u, 0u, 0u, 0u};
        vr[p] = ld ? *reinterpret_cast<const u32x4*>(Vp + (int64_t)j * rs + part * 8) : u32x4{0u, 0u, 0u, 0u};
        kbv[p] = (kbias && j < Lk) ? kbias[j] : 0.f;                              // with the K / V loads: no dependent load later
    }
    float qv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) qv[i] = s.q[part * 8 + i];
    float sc[NB];
    float mx = -INFINITY;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int j = p * KPP + js;
        float d = 0.f;
        if (j == cur) {
#pragma unroll
            for (int i = 0; i < 8; ++i) d = fmaf(qv[i], s.kcur[part * 8 + i], d);
        } else {
            const u32x4 r4 = kr[p];
            d = fmaf(qv[0], bf_lo(r4.x), d); d = fmaf(qv[1], bf_hi(r4.x), d); d = fmaf(qv[2], bf_lo(r4.y), d); d = fmaf(qv[3], bf_hi(r4.y), d);
            d = fmaf(qv[4], bf_lo(r4.z), d); d = fmaf(qv[5], bf_hi(r4.z), d); d = fmaf(qv[6], bf_lo(r4.w), d); d = fmaf(qv[7], bf_hi(r4.w), d);
        }
        d += quad_xor<0xB1>(d); d += quad_xor<0x4E>(d);                     // 4 chunks = one head: all 4 lanes hold the score
        sc[p] = j < Lk ? d * scale + kbv[p] : -INFINITY;
        mx = fmaxf(mx, sc[p]);
    }
    // per-head maximum: the lanes 8 apart hold other keys of the same head; then across the waves through LDS
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lane == 0 || lane == 4) s.stat[(lane >> 2) * DL_NW + wave] = mx;
    __syncthreads();
    {
        float m = s.stat[h * DL_NW];
#pragma unroll
        for (int w = 1; w < DL_NW; ++w) m = fmaxf(m, s.stat[h * DL_NW + w]);
        mx = m;
    }
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < NB; ++p) { sc[p] = __expf(sc[p] - mx); sum += sc[p]; }    // keys beyond Lk: exp(-inf) = 0
    sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
    for (int p = 0; p < NB; ++p) {
        const int j = p * KPP + js;
        const float pr = sc[p];
        if (j == cur) {
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf(pr, s.vcur[part * 8 + i], o[i]);
        } else {
            const u32x4 r4 = vr[p];
            o[0] = fmaf(pr, bf_lo(r4.x), o[0]); o[1] = fmaf(pr, bf_hi(r4.x), o[1]); o[2] = fmaf(pr, bf_lo(r4.y), o[2]); o[3] = fmaf(pr, bf_hi(r4.y), o[3]);
            o[4] = fmaf(pr, bf_lo(r4.z), o[4]); o[5] = fmaf(pr, bf_hi(r4.z), o[5]); o[6] = fmaf(pr, bf_lo(r4.w), o[6]); o[7] = fmaf(pr, bf_hi(r4.w), o[7]);
        }
    }
#pragma unroll
    for (int m = TPK; m < 64; m <<= 1)
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += __shfl_xor(o[i], m, 64);
    __syncthreads();                                                              // every thread has read the maxima in s.stat
    if (lane < TPK) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s.red[wave][part * 8 + i] = o[i];
        if ((lane & 3) == 0) s.stat[h * DL_NW + wave] = sum;
    }
    __syncthreads();
    if (tid < CW) {
        float a = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < DL_NW; ++w) { a += s.red[w][tid]; l += s.stat[(tid >> 5) * DL_NW + w]; }
        s.as[tid] = f2bf(a / l);
    }
    __syncthreads();
}

// publish this member's partial [256], wait for the cluster, y = sum of the partials (member order) + bias + residual -> LayerNorm ->
// s.xs (bf16) / s.xres (the rounded value as fp32); member 0 optionally stores the row to global memory
template <int C, typename T>
__device__ __forceinline__ void combine_norm(Smem<C, T>& s, float* __restrict__ exch, unsigned* sync, int* err, int c, const float* __restrict__ bias,
                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps, T* __restrict__ gout, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    float y = 0.f;
    if constexpr (C > 1) {
        if (tid < DL_D) __hip_atomic_store(exch + c * DL_D + tid, s.part[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cluster_barrier(sync, C, err, tid);
        if (tid < DL_D) {
#pragma unroll
            for (int m = 0; m < C; ++m) y += __hip_atomic_load(exch + m * DL_D + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        if (tid < DL_D) y = s.part[tid];
    }
    if (tid < DL_D) y += bias[tid] + s.xres[tid];
    float p = wave_sum(tid < DL_D ? y : 0.f);
    if (lane == 0) s.stat[wave] = p;
    __syncthreads();
    const float mean = (s.stat[0] + s.stat[1] + s.stat[2] + s.stat[3]) * (1.f / DL_D);
    const float dv = tid < DL_D ? y - mean : 0.f;
    p = wave_sum(dv * dv);
    if (lane == 0) s.stat[8 + wave] = p;
    __syncthreads();
    const float rstd = rsqrtf((s.stat[8] + s.stat[9] + s.stat[10] + s.stat[11]) * (1.f / DL_D) + eps);
    if (tid < DL_D) {
        const T o = DT<T>::from_f(dv * rstd * gamma[tid] + beta[tid]);
        s.xs[tid] = o;
        s.xres[tid] = DT<T>::to_f(o);
        if (gout && c == 0) gout[tid] = o;
    }
    __syncthreads();
}

// -DDL_TIMING: workgroup 0 stores a 100 MHz timestamp per stage behind the exchange buffer (tools/mb_decode_stages.py prints them)
#ifdef DL_TIMING
#define DL_T(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<long long*>(d.exch + (int64_t)d.B * 3 * 4 * DL_D)[k] = wall_clock64(); } while (0)
#else
#define DL_T(k) do { } while (0)
#endif

template <int C, typename T>
__global__ __launch_bounds__(DL_NT) void decode_layer_kernel(p3_decode_layer_desc d) {
    constexpr int CW = Smem<C, T>::CW, FW = Smem<C, T>::FW;
    constexpr bool BF = sizeof(T) == 2;
    __shared__ Smem<C, T> s;
    const int tid = threadIdx.x;
    // cluster members on one XCD: workgroup id = (k * C + c) * 8 + xcd, sample = k * 8 + xcd
    const int xcd = blockIdx.x & 7, kc = blockIdx.x >> 3, c = kc % C, b = (kc / C) * 8 + xcd;
    if (b >= d.B) return;                                        // padding workgroups of the last group of 8 samples (whole clusters)
    const int ch0 = c * CW;
    unsigned* sync = d.sync ? d.sync + 2 * b : nullptr;
    float* exch = d.exch ? d.exch + (int64_t)b * 3 * C * DL_D : nullptr;
    const T* xin = reinterpret_cast<const T*>(d.x_in) + (int64_t)b * d.x_in_stride;
    DL_T(0);
    if (tid < DL_D) { const T v = xin[tid]; s.xs[tid] = v; s.xres[tid] = DT<T>::to_f(v); }
    __syncthreads();
    DL_T(1);
    // ---- self-attention: q|k|v rows of this member's heads -> cache, attention over positions 0..t, partial out_proj, norm1
    {
        const T* w = reinterpret_cast<const T*>(d.w_in);
        gemv_rows<1>(w, DL_D, ch0, CW, DL_D, 3 * CW, s.xs, s.g, tid);         // q, k, v rows of this member's heads: one pass over all waves
        __syncthreads();
        DL_T(2);
        T* crow = reinterpret_cast<T*>(d.kv_self) + ((int64_t)b * d.steps + d.t) * 3 * DL_D;
        for (int n = tid; n < 3 * CW; n += DL_NT) {
            const int seg = n / CW, i = n - seg * CW, col = seg * DL_D + ch0 + i;
            const T o = DT<T>::from_f(s.g[n] + d.b_in[col]);
            crow[col] = o;
            const float f = DT<T>::to_f(o);
            if (seg == 0) s.q[i] = f; else if (seg == 1) s.kcur[i] = f; else s.vcur[i] = f;
        }
        __syncthreads();
        DL_T(3);
        const T* cache = reinterpret_cast<const T*>(d.kv_self) + (int64_t)b * d.steps * 3 * DL_D;
        const float* kb = d.key_bias ? d.key_bias + (int64_t)b * d.key_bias_stride : nullptr;
        if constexpr (C == 4 && BF) attend_reg(s, cache + DL_D + ch0, cache + 2 * DL_D + ch0, 3 * DL_D, d.t + 1, d.t, kb, d.scale, tid);
        else attend<C, T>(s, cache + DL_D + ch0, cache + 2 * DL_D + ch0, 3 * DL_D, d.t + 1, d.t, kb, d.scale, tid);
        DL_T(4);
        if constexpr (C == 4) gemv_kslice64(reinterpret_cast<const T*>(d.w_so), ch0, s.as, s.part, tid);
        else gemv_rows<1>(reinterpret_cast<const T*>(d.w_so), DL_D, 0, DL_D, 0, DL_D, s.as, s.part, tid);
        __syncthreads();
        DL_T(5);
        combine_norm<C, T>(s, exch, sync, d.err, c, d.b_so, d.g1, d.be1, d.eps, nullptr, tid);
        DL_T(6);
    }
    // ---- cross-attention over the memory tokens
    {
        gemv_rows<1>(reinterpret_cast<const T*>(d.w_q), DL_D, ch0, CW, 0, CW, s.xs, s.g, tid);
        __syncthreads();
        if (tid < CW) s.q[tid] = DT<T>::rnd(s.g[tid] + d.b_q[ch0 + tid]);
        __syncthreads();
        DL_T(7);
        const T* mem = reinterpret_cast<const T*>(d.kv_mem) + (int64_t)b * d.Lmem * 2 * DL_D;
        if constexpr (C == 4 && BF) attend_reg(s, mem + ch0, mem + DL_D + ch0, 2 * DL_D, d.Lmem, -1, nullptr, d.scale, tid);
        else attend<C, T>(s, mem + ch0, mem + DL_D + ch0, 2 * DL_D, d.Lmem, -1, nullptr, d.scale, tid);
        DL_T(8);
        if constexpr (C == 4) gemv_kslice64(reinterpret_cast<const T*>(d.w_co), ch0, s.as, s.part, tid);
        else gemv_rows<1>(reinterpret_cast<const T*>(d.w_co), DL_D, 0, DL_D, 0, DL_D, s.as, s.part, tid);
        __syncthreads();
        DL_T(9);
        combine_norm<C, T>(s, exch ? exch + C * DL_D : nullptr, sync, d.err, c, d.b_co, d.g2, d.be2, d.eps, nullptr, tid);
        DL_T(10);
    }
    // ---- feed-forward: own linear1 rows -> own hidden slice -> partial linear2 over that slice
    {
        gemv_rows<1>(reinterpret_cast<const T*>(d.w1), DL_D, c * FW, FW, 0, FW, s.xs, s.g, tid);
        __syncthreads();
        DL_T(11);
        for (int n = tid; n < FW; n += DL_NT) s.hb[n] = DT<T>::from_f(fmaxf(s.g[n] + d.b1[c * FW + n], 0.f));
        __syncthreads();
        gemv_rows<FW / 256>(reinterpret_cast<const T*>(d.w2) + c * FW, DL_FF, 0, DL_D, 0, DL_D, s.hb, s.part, tid);
        __syncthreads();
        DL_T(12);
        combine_norm<C, T>(s, exch ? exch + 2 * C * DL_D : nullptr, sync, d.err, c, d.b2, d.g3, d.be3, d.eps,
                           reinterpret_cast<T*>(d.x_out) + (int64_t)b * d.x_out_stride, tid);
        DL_T(13);
    }
}

}  // namespace

extern "C" int p3_decode_layer(const p3_decode_layer_desc* d, void* stream) {
    P3_CHECK(d && d->x_in && d->x_out && d->kv_self && d->kv_mem, P3_EINVAL, "p3_decode_layer: null pointer");
    P3_CHECK(d->w_in && d->w_so && d->w_q && d->w_co && d->w1 && d->w2 && d->b_in && d->b_so && d->b_q && d->b_co && d->b1 && d->b2, P3_EINVAL, "p3_decode_layer: null weight");
    P3_CHECK(d->g1 && d->be1 && d->g2 && d->be2 && d->g3 && d->be3, P3_EINVAL, "p3_decode_layer: null LayerNorm parameter");
    P3_CHECK(d->D == DL_D && d->H == DL_H && d->FF == DL_FF, P3_EUNSUP, "p3_decode_layer: built for D = 256, 8 heads, FF = 2048");
    P3_CHECK(d->B > 0 && d->t >= 0 && d->t < d->steps && d->steps <= DL_MAXK && d->Lmem > 0 && d->Lmem <= DL_MAXK, P3_ESHAPE, "p3_decode_layer: sequence lengths");
    P3_CHECK(d->cluster == 1 || d->cluster == 4, P3_EINVAL, "p3_decode_layer: cluster must be 1 or 4");
    P3_CHECK(d->cluster == 1 || (d->exch && d->sync), P3_EINVAL, "p3_decode_layer: a cluster needs the exchange and sync buffers");
    for (const void* p : {d->x_in, (const void*)d->x_out, (const void*)d->kv_self, d->kv_mem, d->w_in, d->w_so, d->w_q, d->w_co, d->w1, d->w2})
        P3_CHECK(((uintptr_t)p % 16) == 0, P3_EALIGN, "p3_decode_layer: 16-byte alignment");
    const int groups = p3_ceil_div(d->B, 8);
    // every workgroup of a cluster must be resident while its partners spin: 2 workgroups fit a CU (60 KB LDS, 512 threads); the CU count
    // is the device's (a partition or another part has fewer than the 256 of a full MI355X)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 1;
    }
    P3_CHECK(d->cluster == 1 || groups * 8 * d->cluster <= 2 * cus, P3_ESHAPE, "p3_decode_layer: cluster launch exceeds the co-resident workgroups (use cluster = 1)");
    P3_CHECK(d->fp32 == 0 || d->fp32 == 1, P3_EINVAL, "p3_decode_layer: fp32 must be 0 or 1");
    if (d->fp32) {
        if (d->cluster == 4) hipLaunchKernelGGL((decode_layer_kernel<4, float>), dim3(groups * 8 * 4), dim3(DL_NT), 0, (hipStream_t)stream, *d);
        else hipLaunchKernelGGL((decode_layer_kernel<1, float>), dim3(groups * 8), dim3(DL_NT), 0, (hipStream_t)stream, *d);
    } else {
        if (d->cluster == 4) hipLaunchKernelGGL((decode_layer_kernel<4, bf16_t>), dim3(groups * 8 * 4), dim3(DL_NT), 0, (hipStream_t)stream, *d);
        else hipLaunchKernelGGL((decode_layer_kernel<1, bf16_t>), dim3(groups * 8), dim3(DL_NT), 0, (hipStream_t)stream, *d);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}
