// p3hip fused attention forward (flash style, online softmax, MFMA 32x32).
//
// Work split: grid = (ceil(Lq/128), H, B); a 256-thread workgroup = 4 waves, each wave owns 32 query rows
// and walks the key/value sequence in tiles of KT keys (64 for bf16, 32 for f32) staged in LDS.
//
// Formulation (chosen so that every per-query quantity is lane-local, wave = 64):
//   S^T[kv, q] = K[kv, :] . Q[q, :]      A = K rows from LDS, B = Q rows held in registers for the whole kernel
//       -> lane (q = lane&31, hi = lane>>5) holds the scores of ONE query for 16 keys per 32-key sub-tile
//          (keys crow32(r, hi)); row max / sum need only one exchange with lane^32.
//   O^T[d, q] += V^T[d, kv] . P^T[kv, q]  A = V^T from LDS, B = P of the lane's own query, fed straight from the
//       score registers: the MFMA k-slots are *assigned* to the keys the lane already holds (a sum over keys is
//       order independent), so no cross-lane shuffle of P is needed; the V^T fragment is read with the same key order.
//   bf16: v_mfma_f32_32x32x16_bf16; K and V tiles are staged as they lie in memory (attn_tile.h: unpadded, XOR-swizzled 16-byte chunks),
//         the V^T fragments come from the transposing LDS read ds_read_b64_tr_b16;
//   f32 : v_mfma_f32_32x32x2_f32 (exact fp32), V stays row-major.
// Masks: keys >= Lk, causal (key > query) -> -inf; key_bias[b, key] is ADDED (reference's float padding mask).
#include <stdlib.h>

#include "p3_common.h"
#include "attn_tile.h"

namespace {

using p3attn::u32x4;
using p3attn::f32s;
using p3attn::Kind;

struct AttnArgs {
    const void* Q; const void* K; const void* V; void* O;
    p3_attn_desc d;
    int order;          // attn_block_of mode
};

template <typename T, int D> struct ATr;
template <int D> struct ATr<bf16_t, D> {
    static constexpr int KT = 64, PK = D, PV = D;           // K [KT][D] ; V [KT][D], both swizzled (attn_tile.h)
    static constexpr int K_ELEMS = KT * D, V_ELEMS = KT * D;
};
template <int D> struct ATr<f32s, D> {                        // fp32x3 mode: two swizzled bf16 images (hi | lo) per tile, in units of 4-byte elements
    static constexpr int KT = 64, PK = D, PV = D;
    static constexpr int K_ELEMS = KT * D, V_ELEMS = KT * D;
};
template <int D> struct ATr<float, D> {
    static constexpr int KT = 32, PK = D + 1, PV = D;       // K [KT][PK] ; V [KT][PV]
    static constexpr int K_ELEMS = KT * PK, V_ELEMS = KT * PV;
};

template <typename T, int D, int KPT, int VPT>
__device__ __forceinline__ void attn_load_tile(int t, int tid, const p3_attn_desc& d, const T* Kp, const T* Vp,
                                               u32x4 (&kreg)[KPT], u32x4 (&vreg)[VPT]) {
    using TR = ATr<T, D>;
    constexpr int KT = TR::KT;
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KV16 = KT * D * (int)sizeof(T) / 16;
    constexpr int VITEMS = KV16;

        const int kv0 = t * KT;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int item = tid + 256 * i;
            if (KV16 % 256 == 0 || item < KV16) {
                constexpr int VPR = D * (int)sizeof(T) / 16;  // vectors per row
                int row = item / VPR, cv = item % VPR;
                int kv = kv0 + row; if (kv >= d.Lk) kv = d.Lk - 1;
                kreg[i] = *reinterpret_cast<const u32x4*>(Kp + (int64_t)kv * d.k_rs + cv * (16 / (int)sizeof(T)));
            }
        }
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int item = tid + 256 * i;
            if (VITEMS % 256 == 0 || item < VITEMS) {
                {
                    constexpr int VPR = D * (int)sizeof(T) / 16;
                    int row = item / VPR, cv = item % VPR;
                    int kv = kv0 + row; if (kv >= d.Lk) kv = d.Lk - 1;
                    vreg[i] = *reinterpret_cast<const u32x4*>(Vp + (int64_t)kv * d.v_rs + cv * (16 / (int)sizeof(T)));
                }
            }
        }
}

template <typename T, int D, int KPT, int VPT>
__device__ __forceinline__ void attn_store_tile(int tid, T* Ks, T* Vs, const u32x4 (&kreg)[KPT], const u32x4 (&vreg)[VPT]) {
    using TR = ATr<T, D>;
    constexpr int KT = TR::KT, PK = TR::PK, PV = TR::PV;
    constexpr bool BF = sizeof(T) == 2;
    constexpr int KV16 = KT * D * (int)sizeof(T) / 16;
    constexpr int VPR = D * (int)sizeof(T) / 16;      // 16-byte vectors per row
    static_assert(KPT == VPT, "K and V tiles have the same shape");
    // bf16: item -> (row, chunk) of the swizzled image; item + 256 is the same lane-constant offset + a multiple of 32 rows
    const int sbase = BF ? p3attn::img_off<D>(tid / VPR, tid % VPR) : 0;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int item = tid + 256 * i;
            if (KV16 % 256 == 0 || item < KV16) {
                if constexpr (BF) {
                    *reinterpret_cast<u32x4*>(Ks + sbase + i * (256 / VPR) * D) = kreg[i];
                    *reinterpret_cast<u32x4*>(Vs + sbase + i * (256 / VPR) * D) = vreg[i];
                } else {
                    int row = item / VPR, cv = item % VPR;
                    float* p = reinterpret_cast<float*>(Ks) + row * PK + cv * 4;
                    p[0] = __uint_as_float(kreg[i].x); p[1] = __uint_as_float(kreg[i].y);
                    p[2] = __uint_as_float(kreg[i].z); p[3] = __uint_as_float(kreg[i].w);
                    *reinterpret_cast<u32x4*>(reinterpret_cast<float*>(Vs) + row * PV + cv * 4) = vreg[i];
                }
            }
        }
}

template <typename T, int D, bool DROP>
__global__ __launch_bounds__(256, (Kind<T>::X3 ? 2 : 3)) void attn_fwd_kernel(AttnArgs a) {
    using TR = ATr<T, D>;
    constexpr int KT = TR::KT, PK = TR::PK, PV = TR::PV, NH2 = KT / 32, NDJ = D / 32;
    constexpr bool BF = Kind<T>::BF, X3 = Kind<T>::X3, IMG = Kind<T>::IMG;
    __shared__ __attribute__((aligned(16))) T Ks[TR::K_ELEMS];
    __shared__ __attribute__((aligned(16))) T Vs[TR::V_ELEMS];
    __shared__ float Kb[KT];

    const p3_attn_desc& d = a.d;
    // 1-D grid, XCD-aware: the q-blocks of one (batch, head) run back to back on ONE XCD, so K/V are fetched into one L2 once
    // (PMC r01: 579 MB fetched per ViT launch with the (q-block, head, batch) grid = 7x the 77 MB of K/V, one copy per XCD)
    const int nqb = (d.Lq + 127) / 128;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int blk_, pair_;
    attn_block_of(lid, nqb, d.Lq, d.B * d.H, a.order, blk_, pair_);
    const int qblk = blk_ * 128, h = pair_ % d.H, b = pair_ / d.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const T* Qp = reinterpret_cast<const T*>(a.Q) + (int64_t)b * d.q_bs + h * D;
    const T* Kp = reinterpret_cast<const T*>(a.K) + (int64_t)b * d.k_bs + h * D;
    const T* Vp = reinterpret_cast<const T*>(a.V) + (int64_t)b * d.v_bs + h * D;
    T* Op = reinterpret_cast<T*>(a.O) + (int64_t)b * d.o_bs + h * D;

    const int q = qblk + wave * 32 + l31;          // this lane's query row
    const bool wave_live = qblk + wave * 32 < d.Lq;
    const int qc = q < d.Lq ? q : d.Lq - 1;        // clamped for loads

    // ---- Q fragments (B operand of S^T) ----
    s16x8 qb[IMG ? D / 16 : 1], ql[X3 ? D / 16 : 1];
    float qf[IMG ? 1 : D / 2];
    if constexpr (BF) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks)
            qb[ks] = *reinterpret_cast<const s16x8*>(Qp + (int64_t)qc * d.q_rs + ks * 16 + 8 * hi);
    } else if constexpr (X3) {
        const float* qr = reinterpret_cast<const float*>(Qp) + (int64_t)qc * d.q_rs + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) {
            const float4 x0 = *reinterpret_cast<const float4*>(qr + ks * 16), x1 = *reinterpret_cast<const float4*>(qr + ks * 16 + 4);
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            p3attn::split8(x, qb[ks], ql[ks]);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < D / 2; ++ks) qf[ks] = Qp[(int64_t)qc * d.q_rs + 2 * ks + hi];
    }

    f32x16 oacc[NDJ];
#pragma unroll
    for (int j = 0; j < NDJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int kv_end = d.Lk;
    if (d.causal) { int lim = qblk + 128; if (lim < kv_end) kv_end = lim; }  // keys beyond the block's last query are masked
    const int ntiles = (kv_end + KT - 1) / KT;
    const float* kbias = d.key_bias ? d.key_bias + (int64_t)b * d.Lk : nullptr;
    DropKey dk; uint32_t drop_rk = 0;
    if constexpr (DROP) { dk = drop_key(d.drop); drop_rk = drop_rowkey(dk, (uint64_t)((int64_t)b * d.H + h) * (uint64_t)d.Lq + (uint64_t)qc); }

    // staging registers
    constexpr int KV16 = KT * D * (int)sizeof(T) / 16;      // 16-byte vectors per K tile
    constexpr int KPT = (KV16 + 255) / 256;                 // per thread
    constexpr int VPT = KPT;
    // LDS-DMA staging for head dim 32 only (decoder: 91 -> 84 us same-box); at head dim 64 the forward kernel has the registers for the rows in flight and the raw
    // image's extra LDS round trip made it slower (ViT: 288 -> 302 us, profiles/r06_attn_dma_ab.txt) - there the rows stay register-staged
    constexpr bool DMA = X3 && D == 32;
    u32x4 kreg[X3 ? 1 : KPT];
    u32x4 vreg[X3 ? 1 : VPT];
    p3attn::SplitStage<IMG ? D : 64, 64> ksp, vsp;  // fp32x3 without DMA: fp32 rows in flight, split into the hi / lo images at store time
    // DMA: the fp32 rows travel global -> LDS by LDS-DMA into raw images (dynamic LDS: K | V) and are split into the hi / lo images at the tile switch (attn_tile.h
    // SplitDma) - no staging registers live across the tile
    using SD = p3attn::SplitDma<IMG ? D : 64, X3 ? KT : 64>;
    extern __shared__ __attribute__((aligned(16))) unsigned char attn_raw[];
    const uint32_t raw_lds = DMA ? (uint32_t)(uintptr_t)attn_raw : 0u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    p3attn::ScoreAddr<IMG ? D : 16> sadr;         // bf16 / fp32x3: lane-constant offsets into the swizzled K / V images
    p3attn::TrAddr<IMG ? D : 32> tadr;
    if constexpr (IMG) { sadr.init(l31, hi); tadr.init(lane); }
    float kbreg = 0.f;                            // the next tile's key bias (first KT threads), loaded with the tile: see below
    auto load_tile = [&](int t) __attribute__((always_inline)) {
        if constexpr (DMA) {
            SD::issue(reinterpret_cast<const float*>(Kp), t * KT, d.Lk, d.k_rs, raw_lds, wave_u, lane);
            SD::issue(reinterpret_cast<const float*>(Vp), t * KT, d.Lk, d.v_rs, raw_lds + SD::RAW_B, wave_u, lane);
        } else if constexpr (X3) {
            ksp.load(reinterpret_cast<const float*>(Kp), t * KT, d.Lk, d.k_rs, tid);
            vsp.load(reinterpret_cast<const float*>(Vp), t * KT, d.Lk, d.v_rs, tid);
        } else {
            attn_load_tile<T, D, KPT, VPT>(t, tid, d, Kp, Vp, kreg, vreg);
        }
        if (kbias && tid < KT) { const int kvb = t * KT + tid; kbreg = kbias[kvb < d.Lk ? kvb : d.Lk - 1]; }
    };

    if (ntiles > 0) load_tile(0);
    for (int t = 0; t < ntiles; ++t) {
        if constexpr (DMA) p3attn::wait_vm0();        // this wave's DMA pieces of the tile have landed
        __syncthreads();  // previous tile's LDS reads are done (DMA: and every wave's pieces are in LDS)
        if constexpr (DMA) { SD::split_store(attn_raw, reinterpret_cast<bf16_t*>(Ks), tid); SD::split_store(attn_raw + SD::RAW_B, reinterpret_cast<bf16_t*>(Vs), tid); }
        else if constexpr (X3) { ksp.store(reinterpret_cast<bf16_t*>(Ks), tid); vsp.store(reinterpret_cast<bf16_t*>(Vs), tid); }
        else attn_store_tile<T, D, KPT, VPT>(tid, Ks, Vs, kreg, vreg);
        // the tile's key bias goes through LDS (log2 units): read per element from global memory it was 32 dependent loads per tile, each
        // followed by s_waitcnt vmcnt(0) - which also drained the next tile's prefetch (decoder self-attention: every tile has a bias).
        // r06: it is LOADED with the tile's prefetch (kbreg) - loaded here, its latency stood in front of the barrier in every tile
        if (kbias && tid < KT) Kb[tid] = kbreg * 1.4426950408889634f;
        __syncthreads();
        if (t + 1 < ntiles) load_tile(t + 1);
        const int kv0 = t * KT;
        // a wave whose 32 queries all lie beyond Lq (tail q-block: L = 785 -> 17 live rows, L = 385 -> 1) only helps with the staging
        if (!wave_live) continue;

        // ---- S^T = K . Q^T ----
        // last tile: a 32-key half that lies entirely beyond the block's last visible key is skipped (785 = 12 x 64 + 17)
        // (also per wave under the causal mask: keys beyond the wave's last query; a tile with no visible key is skipped entirely)
        int vis_end = kv_end;
        if (d.causal) { const int wl = qblk + wave * 32 + 32; if (wl < vis_end) vis_end = wl; }
        const int nh2 = (vis_end - kv0 + 31) >> 5;          // visible 32-key halves of this tile (>= NH2: all)
        if (nh2 <= 0) continue;
        f32x16 sacc[NH2];
#pragma unroll
        for (int h2 = 0; h2 < NH2; ++h2) {
            if (h2 >= nh2) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[h2][r] = 0.f;
            if constexpr (BF) {
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks)
                    sacc[h2] = p3attn::mfma_bf16(sadr.frag(reinterpret_cast<const bf16_t*>(Ks), h2, ks), qb[ks], sacc[h2]);
            } else if constexpr (X3) {
                const bf16_t* kh = reinterpret_cast<const bf16_t*>(Ks);
#pragma unroll
                for (int ks = 0; ks < D / 16; ++ks) {      // small terms first
                    const s16x8 xh = sadr.frag(kh, h2, ks), xl = sadr.frag(kh + KT * D, h2, ks);
                    sacc[h2] = p3attn::mfma_bf16(xl, qb[ks], sacc[h2]);
                    sacc[h2] = p3attn::mfma_bf16(xh, ql[ks], sacc[h2]);
                    sacc[h2] = p3attn::mfma_bf16(xh, qb[ks], sacc[h2]);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < D / 2; ++ks) {
                    float kf = reinterpret_cast<const float*>(Ks)[(h2 * 32 + l31) * PK + 2 * ks + hi];
                    sacc[h2] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf, qf[ks], sacc[h2], 0, 0, 0);
                }
            }
        }
        // ---- online softmax in the log2 domain: p = exp2(s*c - m), c = scale*log2(e); m_run, lse bookkeeping in log2 units ----
        // A tile that needs no masking (entirely inside [0, Lk), entirely at or below the causal diagonal of this wave's
        // first query) and has no key bias takes the short VALU path: max, one fma + one v_exp_f32 per score.
        const float c = d.scale * 1.4426950408889634f;
        const bool full = (kv0 + KT <= d.Lk) && (!d.causal || kv0 + KT - 1 <= qblk + wave * 32) && !kbias;
        float mx = -INFINITY;
        if (full) {
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2)       // full tiles have nh2 >= NH2
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[h2][r]);
            mx *= c;                                      // c > 0: max commutes with the scaling
        } else {
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2) {
                if (h2 >= nh2) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + h2 * 32 + crow32(r, hi);
                    float s2 = sacc[h2][r] * c;
                    if (kbias) s2 += Kb[h2 * 32 + crow32(r, hi)];
                    if (kv >= d.Lk || (d.causal && kv > q)) s2 = -INFINITY;
                    sacc[h2][r] = s2;
                    mx = fmaxf(mx, s2);
                }
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        float psum = 0.f;
        if (full) {
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#if defined(P3_ATTN_ABL) && (P3_ATTN_ABL & 4)
                    const float p = fmaf(sacc[h2][r], c, -m_use);          // timing ablation: the exponentials - wrong numbers
#else
                    const float p = __builtin_amdgcn_exp2f(fmaf(sacc[h2][r], c, -m_use));
#endif
                    sacc[h2][r] = p;
                    psum += p;
                }
        } else {
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2) {
                if (h2 >= nh2) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(sacc[h2][r] - m_use);
                    sacc[h2][r] = p;
                    psum += p;
                }
            }
        }
        // rescale the running state only when some row's maximum moved (wave-uniform test; alpha == 1 exactly otherwise)
        if (__any(m_new != m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
            l_run *= alpha;
#pragma unroll
            for (int j = 0; j < NDJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[j][r] *= alpha;
            m_run = m_new;
        }
        l_run += psum;
        if constexpr (DROP) {   // attention-probability dropout: the row sum above stays undropped (softmax first, then dropout)
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2) {
                if (h2 >= nh2) continue;            // invisible half: P == 0 there, the backward never uses its keep bits
                uint32_t wrow = 0;                  // keep bits of this lane's 16 keys of the 32-key block (bit = key % 32)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {   // registers r, r+1 hold keys 2j, 2j+1: one hash per pair
                    const uint32_t bits = drop_bits(drop_rk, drop_colkey(dk, (uint32_t)(kv0 + h2 * 32 + crow32(r, hi))));
                    const bool k0 = drop_keep_lo(dk, bits), k1 = drop_keep_hi(dk, bits);
                    sacc[h2][r] = k0 ? sacc[h2][r] * dk.inv_keep : 0.f;
                    sacc[h2][r + 1] = k1 ? sacc[h2][r + 1] * dk.inv_keep : 0.f;
                    wrow |= ((uint32_t)k0 | ((uint32_t)k1 << 1)) << crow32(r, hi);
                }
                if (d.drop_rows) {                  // publish the mask for the backward kernels (they then skip the hashing)
                    wrow |= __shfl_xor(wrow, 32, 64);
                    const int nkw = (d.Lk + 31) >> 5, kw = (kv0 >> 5) + h2;
                    if (hi == 0 && q < d.Lq && kw < nkw)      // kw == nkw: the 32-key half beyond Lk of the last tile
                        d.drop_rows[(((int64_t)b * d.H + h) * nkw + kw) * d.Lq + q] = wrow;     // [b*H + h][key word][q]: 32 lanes = 128 contiguous bytes
                }
            }
        }

        // ---- O^T += V^T . P^T ----
        if constexpr (BF) {
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2) {
                if (h2 >= nh2) continue;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    uint32_t pw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pw[i] = pack_bf2(sacc[h2][8 * c2 + 2 * i], sacc[h2][8 * c2 + 2 * i + 1]);
                    s16x8 pb = __builtin_bit_cast(s16x8, make_uint4(pw[0], pw[1], pw[2], pw[3]));
#pragma unroll
                    for (int j = 0; j < NDJ; ++j)      // V^T[d, keys 4hi + {0..3, 8..11}] of the 16-key group: the keys this lane's P registers hold
                        oacc[j] = p3attn::mfma_bf16(tadr.frag(reinterpret_cast<const bf16_t*>(Vs), h2 * 32 + 16 * c2, j), pb, oacc[j]);
                }
            }
        } else if constexpr (X3) {
            const bf16_t* vh = reinterpret_cast<const bf16_t*>(Vs);
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2) {
                if (h2 >= nh2) continue;
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    float pv8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv8[i] = sacc[h2][8 * c2 + i];
                    s16x8 ph, pl;
                    p3attn::split8(pv8, ph, pl);
#pragma unroll
                    for (int j = 0; j < NDJ; ++j) {
                        const s16x8 xh = tadr.frag(vh, h2 * 32 + 16 * c2, j), xl = tadr.frag(vh + KT * D, h2 * 32 + 16 * c2, j);
                        oacc[j] = p3attn::mfma_bf16(xl, ph, oacc[j]);
                        oacc[j] = p3attn::mfma_bf16(xh, pl, oacc[j]);
                        oacc[j] = p3attn::mfma_bf16(xh, ph, oacc[j]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int h2 = 0; h2 < NH2; ++h2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = sacc[h2][r];
                    const int kvl = h2 * 32 + crow32(r, hi);
#pragma unroll
                    for (int j = 0; j < NDJ; ++j) {
                        float vf = reinterpret_cast<const float*>(Vs)[kvl * PV + j * 32 + l31];
                        oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, pv, oacc[j], 0, 0, 0);
                    }
                }
        }
    }

    // ---- finalize ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    if (q < d.Lq) {
#pragma unroll
        for (int j = 0; j < NDJ; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int dd = j * 32 + 8 * rg + 4 * hi;
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma clang fp contract(off)          // a rounded product: the planes copy below must split the STORED value (no fma of this multiply with the split's subtraction)
                    o[i] = oacc[j][4 * rg + i] * inv;
                }
                T* dst = Op + (int64_t)q * d.o_rs + dd;
                if constexpr (BF) {
                    uint2 pk; pk.x = pack_bf2(o[0], o[1]); pk.y = pack_bf2(o[2], o[3]);
                    *reinterpret_cast<uint2*>(dst) = pk;
                } else {
                    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
                }
                if constexpr (X3) {
                    if (d.o_planes) {          // the same four values as planes: the output projection's operand (p3_attn_desc.o_planes)
                        bf16_t* ph = reinterpret_cast<bf16_t*>(d.o_planes) + (int64_t)b * d.op_bs + (int64_t)q * d.op_rs + h * D + dd;
                        uint2 hh, ll;
                        hh.x = pack_bf2(o[0], o[1]); hh.y = pack_bf2(o[2], o[3]);
                        {
#pragma clang fp contract(off)
                            ll.x = pack_bf2(o[0] - __uint_as_float(hh.x << 16), o[1] - __uint_as_float(hh.x & 0xffff0000u));
                            ll.y = pack_bf2(o[2] - __uint_as_float(hh.y << 16), o[3] - __uint_as_float(hh.y & 0xffff0000u));
                        }
                        *reinterpret_cast<uint2*>(ph) = hh;
                        *reinterpret_cast<uint2*>(ph + d.op_lo) = ll;
                    }
                }
            }
        if (d.lse && hi == 0) d.lse[((int64_t)b * d.H + h) * d.Lq + q] = (m_run + __log2f(l_tot)) * 0.6931471805599453f;
    }
}

}  // namespace

// ---- decode step: ONE query per (batch, head) against a key/value cache (bf16) -------------------------------------------------
// The tiled kernel above gives a 1-row problem a 128-query block that walks the keys 64 at a time through LDS: 13 dependent tile
// round trips for the 784 memory tokens (r01: 17 us per call, x12 calls per decode step).  Here one 256-thread workgroup owns one
// (b, h): every thread scores keys tid, tid + 256, ... with plain fp32 dot products (K rows are 64 / 128 contiguous bytes), the
// softmax is one block-wide max + sum, and P V runs as 64 key slots x D/8 sixteen-byte channel chunks reduced through LDS.
// fp32 probabilities (no bf16 rounding of P), so it differs from the tiled kernel in the last bf16 bit only.
template <int D>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
    constexpr int CH = D / 8;                   // 16-byte chunks per row
    constexpr int SLOTS = 256 / CH;             // key slots of the P V phase
    const p3_attn_desc& d = a.d;
    extern __shared__ float dsm[];
    float* p = dsm;                             // [Lk] scores -> probabilities
    float* red = dsm + ((d.Lk + 3) & ~3);       // [SLOTS][D] partial outputs, then scratch for the block reductions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / d.H, h = blockIdx.x - b * d.H;
    const bf16_t* Qp = reinterpret_cast<const bf16_t*>(a.Q) + (int64_t)b * d.q_bs + h * D;
    const bf16_t* Kp = reinterpret_cast<const bf16_t*>(a.K) + (int64_t)b * d.k_bs + h * D;
    const bf16_t* Vp = reinterpret_cast<const bf16_t*>(a.V) + (int64_t)b * d.v_bs + h * D;
    float q[D];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint4 raw = *reinterpret_cast<const uint4*>(Qp + c * 8);
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { q[c * 8 + 2 * i] = __uint_as_float(w[i] << 16); q[c * 8 + 2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    const float* kbias = d.key_bias ? d.key_bias + (int64_t)b * d.Lk : nullptr;
    float mx = -INFINITY;
    for (int j = tid; j < d.Lk; j += 256) {
        const bf16_t* kr = Kp + (int64_t)j * d.k_rs;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const uint4 raw = *reinterpret_cast<const uint4*>(kr + c * 8);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s = fmaf(q[c * 8 + 2 * i], __uint_as_float(w[i] << 16), s);
                s = fmaf(q[c * 8 + 2 * i + 1], __uint_as_float(w[i] & 0xffff0000u), s);
            }
        }
        s = s * d.scale + (kbias ? kbias[j] : 0.f);
        p[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < d.Lk; j += 256) { const float e = __expf(p[j] - mx); p[j] = e; sum += e; }
    sum = wave_sum(sum);
    __syncthreads();                            // everyone has read red[0..3]; the p[] writes are visible after the next barrier
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    __syncthreads();
    // P V: thread = (key slot, channel chunk)
    const int slot = tid / CH, ch = tid - slot * CH;
    float o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    for (int j = slot; j < d.Lk; j += SLOTS) {
        const uint4 raw = *reinterpret_cast<const uint4*>(Vp + (int64_t)j * d.v_rs + ch * 8);
        const float pj = p[j];
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = fmaf(pj, __uint_as_float(w[i] << 16), o[2 * i]); o[2 * i + 1] = fmaf(pj, __uint_as_float(w[i] & 0xffff0000u), o[2 * i + 1]); }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[slot * D + ch * 8 + i] = o[i];
    __syncthreads();
    if (tid < D) {
        float acc = 0.f;
        for (int sidx = 0; sidx < SLOTS; ++sidx) acc += red[sidx * D + tid];
        reinterpret_cast<bf16_t*>(a.O)[(int64_t)b * d.o_bs + h * D + tid] = f2bf(acc * inv);
    }
}

int p3_attn_order(void) { return 1; }      // tail blocks first (attn_block_of mode 1; r04 A/B: profiles/r04_mb_attn_ab.txt - the pair-major order is gone)

extern "C" int p3_attention(const void* Q, const void* K, const void* V, void* O, const p3_attn_desc* d, void* stream) {
    P3_CHECK(Q && K && V && O && d, P3_EINVAL, "p3_attention: null pointer");
    P3_CHECK(d->B > 0 && d->H > 0 && d->Lq > 0 && d->Lk > 0, P3_ESHAPE, "p3_attention: empty problem");
    P3_CHECK(d->head_dim == 32 || d->head_dim == 64, P3_EUNSUP, "p3_attention: head_dim must be 32 or 64");
    P3_CHECK(d->dtype == P3_F32 || d->dtype == P3_BF16 || d->dtype == P3_F32X3, P3_EUNSUP, "p3_attention: dtype");
    const int al = d->dtype == P3_BF16 ? 8 : 4;
    P3_CHECK(d->q_rs % al == 0 && d->k_rs % al == 0 && d->v_rs % al == 0 && d->o_rs % 4 == 0, P3_EALIGN, "p3_attention: row strides");
    P3_CHECK(d->q_bs % al == 0 && d->k_bs % al == 0 && d->v_bs % al == 0 && d->o_bs % 4 == 0, P3_EALIGN, "p3_attention: batch strides");
    P3_CHECK(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)V % 16) == 0 && ((uintptr_t)O % 16) == 0, P3_EALIGN, "p3_attention: 16-byte base alignment");
    P3_CHECK(!d->o_planes || (d->dtype == P3_F32X3 && d->op_lo != 0 && d->op_rs % 4 == 0 && d->op_bs % 4 == 0 && d->op_lo % 4 == 0 && ((uintptr_t)d->o_planes % 8) == 0), P3_EINVAL,
             "p3_attention: o_planes goes with P3_F32X3 and needs op_bs / op_rs / op_lo (multiples of 4 bf16 elements)");
    AttnArgs a; a.Q = Q; a.K = K; a.V = V; a.O = O; a.d = *d; a.order = p3_attn_order();
    dim3 grid(p3_ceil_div(d->Lq, 128) * d->H * d->B), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool drop = d->drop.seed != nullptr && d->drop.p > 0.f;
    P3_CHECK(!drop || d->drop.p < 1.f, P3_EINVAL, "p3_attention: dropout p must be < 1");
    constexpr int no_decode = 0;
    if (!no_decode && d->dtype == P3_BF16 && d->Lq == 1 && !d->causal && !drop && !d->lse && d->Lk <= 8192 && d->v_rs % 8 == 0 && d->v_bs % 8 == 0) {
        const size_t lds = (size_t)(((d->Lk + 3) & ~3) + (256 / (d->head_dim / 8)) * d->head_dim) * sizeof(float);
        if (d->head_dim == 64) hipLaunchKernelGGL((attn_decode_kernel<64>), dim3(d->B * d->H), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((attn_decode_kernel<32>), dim3(d->B * d->H), dim3(256), lds, s, a);
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    // fp32x3 at head dim 32: raw fp32 images of the K and V tile in dynamic LDS (2 x 8 KB; the attribute below also covers the diagnostic padding)
    size_t dyn = d->dtype == P3_F32X3 && d->head_dim == 32 ? (size_t)2 * ATr<f32s, 32>::KT * 32 * 4 : 0;
    static int pad_lds = -1;              // P3_ATTN_PAD_LDS=<bytes> (diagnostic): extra dynamic LDS per workgroup of the fp32x3 kernels - 40000 leaves ONE workgroup per CU
    if (pad_lds < 0) { const char* e = getenv("P3_ATTN_PAD_LDS"); pad_lds = e ? atoi(e) : 0; }
    if (d->dtype == P3_F32X3) dyn += (size_t)pad_lds;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<f32s, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 64 * 4 + pad_lds);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<f32s, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 64 * 4 + pad_lds);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<f32s, 32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 64 * 4 + pad_lds);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<f32s, 32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 64 * 4 + pad_lds);
        attr_set = true;
    }
#define P3_ATTN_FWD(T, D)                                                                                   \
    do {                                                                                                    \
        if (drop) hipLaunchKernelGGL((attn_fwd_kernel<T, D, true>), grid, block, dyn, s, a);                \
        else hipLaunchKernelGGL((attn_fwd_kernel<T, D, false>), grid, block, dyn, s, a);                    \
    } while (0)
    if (d->dtype == P3_BF16) {
        if (d->head_dim == 64) P3_ATTN_FWD(bf16_t, 64); else P3_ATTN_FWD(bf16_t, 32);
    } else if (d->dtype == P3_F32X3) {                   // fp32x3 mode: bf16 x 3 products on split images (attn_tile.h)
        if (d->head_dim == 64) P3_ATTN_FWD(f32s, 64); else P3_ATTN_FWD(f32s, 32);
    } else {
        if (d->head_dim == 64) P3_ATTN_FWD(float, 64); else P3_ATTN_FWD(float, 32);
    }
#undef P3_ATTN_FWD
    P3_LAUNCH_CHECK();
    return P3_OK;
}
