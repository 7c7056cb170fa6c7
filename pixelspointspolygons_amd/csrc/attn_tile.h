// LDS tiles of the bf16 attention kernels (attention.hip, attention_bwd.hip): ONE image per operand tile, rows as they lie in memory.
//
//   image  : R rows x D bf16, unpadded (128 B rows for D = 64, 64 B for D = 32); the 16-byte chunk c of row r sits at chunk
//            c ^ sw(r).  Staged with 16-byte global loads and ds_write_b128 (no packing VALU, no 4-byte transposing stores).
//   score  : MFMA A operand with k = head dim: lane (row l31, half hi) reads the 16 bytes of logical chunk 2*ks + hi (ds_read_b128).
//   accum  : MFMA A operand with k = rows (V^T, K^T, Q^T, dO^T): gfx950's transposing read ds_read_b64_tr_b16 - per 16-lane group
//            lanes 4j..4j+3 point at row j's 16 elements and lane i receives column i of those 4 rows (pinned by tools/probe/tr_probe.hip)
//            - delivers rows rb + 4*hi + {0..3} and rb + 4*hi + {8..11}, the rows whose P / dS values the lane already holds in its
//            score registers (crow32), so no transposed second image is built (r01 kept both: 2x the LDS bytes, 4 ds_write_b32 + 8
//            VALU per 16 staged bytes).
//   sw(r)  : D = 64: ((r>>1)&1)<<2 | (r>>2)&3 ; D = 32: (r>>2)&3.  Both read patterns are bank-conflict free: the 16 lanes of a
//            ds_read_b128 group ({0-3,12-15,20-27}, ...) land on 16 distinct (row parity / row&3, chunk) bank sets, and the 4 rows x 64 B
//            of a transposing read's 32-lane cycle land in 4 distinct 64-byte bank quarters.
//   All per-read addresses are lane constants (computed once) plus compile-time immediates: sw() of the rows a lane touches depends only
//   on the lane, because every tile / sub-tile base is a multiple of 16 rows.
#pragma once
#include "p3_common.h"


namespace p3attn {

// element-type tag of the "fp32x3" mode (p3_set_gemm_split): fp32 in global memory; in LDS every row image exists twice - hi = bf16(x) and lo = bf16(x - hi),
// the same swizzled layout each - and every product runs as a_lo b_hi + a_hi b_lo + a_hi b_hi on the bf16 MFMA (2^-17 per product, fp32 accumulation)
struct f32s { float v; };
template <typename T> struct Kind { static constexpr bool BF = sizeof(T) == 2, X3 = false, IMG = BF; };
template <> struct Kind<f32s> { static constexpr bool BF = false, X3 = true, IMG = true; };

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int D> __device__ __forceinline__ int sw(int row) {
    return D == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
}
// element offset of logical 16-byte chunk c of row `row`
template <int D> __device__ __forceinline__ int img_off(int row, int c) { return row * D + ((c ^ sw<D>(row)) << 3); }

// registers of one R x D bf16 tile in flight between global memory and LDS (256 threads)
template <int D, int R> struct RowStage {
    static constexpr int CPR = D / 8, ITEMS = R * CPR, N = (ITEMS + 255) / 256;
    u32x4 v[N];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ src, int row0, int nvalid, int64_t row_stride, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int item = tid + 256 * i;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                int r = row0 + item / CPR; if (r >= nvalid) r = nvalid - 1;
                v[i] = *reinterpret_cast<const u32x4*>(src + (int64_t)r * row_stride + (item % CPR) * 8);
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* img, int tid) const {
        const int base = img_off<D>(tid / CPR, tid % CPR);       // row + 256/CPR * i keeps sw(): the step is a multiple of 32 rows
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int item = tid + 256 * i;
            if (ITEMS % 256 == 0 || item < ITEMS) *reinterpret_cast<u32x4*>(img + base + i * (256 / CPR) * D) = v[i];
        }
    }
};

// lane-constant offsets of the score fragments: row sub*32 + l31, k-step ks  ->  img + sub*32*D + off[ks]
template <int D> struct ScoreAddr {
    int off[D / 16];
    __device__ __forceinline__ void init(int l31, int hi) {
#pragma unroll
        for (int ks = 0; ks < D / 16; ++ks) off[ks] = img_off<D>(l31, 2 * ks + hi);
    }
    __device__ __forceinline__ s16x8 frag(const bf16_t* img, int sub, int ks) const {
        return *reinterpret_cast<const s16x8*>(img + sub * 32 * D + off[ks]);
    }
};

// lane-constant offsets of the transposing reads: rows rb16 + 4*hi + {0..3} (sec 0) / + {8..11} (sec 1), column j*32 + l31
template <int D> struct TrAddr {
    int off[D / 32][2];
    __device__ __forceinline__ void init(int lane) {
        const int li = lane & 15, g4 = lane >> 4, hi = lane >> 5;
#pragma unroll
        for (int j = 0; j < D / 32; ++j)
#pragma unroll
            for (int sec = 0; sec < 2; ++sec)
                off[j][sec] = img_off<D>(4 * hi + (li >> 2) + 8 * sec, j * 4 + (g4 & 1) * 2 + ((li & 3) >> 1)) + (li & 1) * 4;
    }
    // X^T fragment (8 k-slots) of column j*32 + l31 over the 16 rows starting at rb16 (multiple of 16)
    __device__ __forceinline__ s16x8 frag(const bf16_t* img, int rb16, int j) const {
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + rb16 * D + off[j][0]));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + rb16 * D + off[j][1]));
        return s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    }
};

// split 8 fp32 values (two float4) into the hi / lo bf16 fragments: one v_cvt_pk_bf16_f32 per PAIR for hi, the two halves of that word back as fp32 by a shift / a
// mask, two subtractions, one more pack for lo - 3 VALU ops per value (r06: the form that rounded every value separately and then packed the rounded floats
// again cost 4; the bits are the same)
__device__ __forceinline__ void split8(const float (&x)[8], s16x8& h, s16x8& l) {
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hw[i] = pack_bf2(x[2 * i], x[2 * i + 1]);
#if defined(P3_ATTN_ABL) && (P3_ATTN_ABL & 2)
        lw[i] = hw[i];                        // timing ablation (tools/r06_g25.sh): what the split of P costs - wrong numbers
#else
        lw[i] = pack_bf2(x[2 * i] - __uint_as_float(hw[i] << 16), x[2 * i + 1] - __uint_as_float(hw[i] & 0xffff0000u));
#endif
    }
    h = __builtin_bit_cast(s16x8, u32x4{hw[0], hw[1], hw[2], hw[3]});
    l = __builtin_bit_cast(s16x8, u32x4{lw[0], lw[1], lw[2], lw[3]});
}

// registers of one R x D fp32 tile on its way into the two bf16 images (hi at img, lo at img + R * D)
template <int D, int R> struct SplitStage {
    static constexpr int VPR = D / 4, ITEMS = R * VPR, N = (ITEMS + 255) / 256;
    u32x4 v[N];
    __device__ __forceinline__ void load(const float* __restrict__ src, int row0, int nvalid, int64_t row_stride, int tid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int item = tid + 256 * i;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                int r = row0 + item / VPR; if (r >= nvalid) r = nvalid - 1;
                v[i] = *reinterpret_cast<const u32x4*>(src + (int64_t)r * row_stride + (item % VPR) * 4);
            }
        }
    }
    __device__ __forceinline__ void store(bf16_t* img, int tid) const {
        // item -> (row, float4 cv): 16-byte chunk cv / 2, half cv & 1; row + 256 / VPR * i keeps sw(): the step is a multiple of 16 rows
        const int base = img_off<D>(tid / VPR, (tid % VPR) >> 1) + ((tid % VPR) & 1) * 4;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int item = tid + 256 * i;
            if (ITEMS % 256 == 0 || item < ITEMS) {
                const float x0 = __uint_as_float(v[i][0]), x1 = __uint_as_float(v[i][1]), x2 = __uint_as_float(v[i][2]), x3 = __uint_as_float(v[i][3]);
                const uint32_t h0 = pack_bf2(x0, x1), h1 = pack_bf2(x2, x3);              // 3 VALU ops per value (see split8)
#if defined(P3_ATTN_ABL) && (P3_ATTN_ABL & 1)
                const uint32_t l0 = h0, l1 = h1;      // timing ablation: what the split of the staged rows costs - wrong numbers
#else
                const uint32_t l0 = pack_bf2(x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = pack_bf2(x2 - __uint_as_float(h1 << 16), x3 - __uint_as_float(h1 & 0xffff0000u));
#endif
                bf16_t* p = img + base + i * (256 / VPR) * D;
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + R * D) = make_uint2(l0, l1);
            }
        }
    }
};

// fp32x3, r06: the fp32 rows of a tile go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers held across the tile's MFMAs, no s_waitcnt vmcnt in
// front of anything but the tile switch) into a RAW image - R rows x D fp32, linear, 1 KB pieces of 1024 / (4 D) rows, wave w of the four issues pieces w, w + 4, ... -
// and are split into the hi / lo bf16 images LDS -> registers -> LDS at the tile switch.  The register-staged form (SplitStage) kept 32 registers per thread in flight
// across the whole tile: the dK / dV kernel spilled, and every reload of a spilled register (scratch_load + s_waitcnt vmcnt(0)) drained the prefetch it was issued behind.
__device__ __forceinline__ void lds_dma16(const void* sbase, uint32_t lds_dst, uint32_t voff) {      // 64 lanes x 16 B -> LDS [lds_dst, lds_dst + 1 KB), lane order
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int D, int R> struct SplitDma {
    static constexpr int VPR = D / 4, ITEMS = R * VPR, N = ITEMS / 256, RPP = 1024 / (D * 4), PIECES = R / RPP, PW = PIECES / 4, RAW_B = R * D * 4;
    static_assert(ITEMS % 256 == 0 && PIECES % 4 == 0, "whole pieces per wave");
    // src: the (batch, head) base of the operand (wave-uniform); rows row0 .. row0 + R - 1, clamped to nvalid - 1
    __device__ __forceinline__ static void issue(const float* __restrict__ src, int row0, int nvalid, int row_stride, uint32_t raw_lds, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int piece = wave + 4 * i;
            int r = row0 + piece * RPP + lane / VPR; if (r >= nvalid) r = nvalid - 1;
            lds_dma16(src, raw_lds + piece * 1024, (uint32_t)(r * row_stride + (lane % VPR) * 4) * 4u);
        }
    }
    // raw image -> the two swizzled bf16 images (hi at img, lo at img + R * D): the item -> (row, chunk) map of SplitStage::store
    __device__ __forceinline__ static void split_store(const unsigned char* raw, bf16_t* img, int tid) {
        const int base = img_off<D>(tid / VPR, (tid % VPR) >> 1) + ((tid % VPR) & 1) * 4;
        u32x4 v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const u32x4*>(raw + (tid + 256 * i) * 16);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float x0 = __uint_as_float(v[i][0]), x1 = __uint_as_float(v[i][1]), x2 = __uint_as_float(v[i][2]), x3 = __uint_as_float(v[i][3]);
            const uint32_t h0 = pack_bf2(x0, x1), h1 = pack_bf2(x2, x3);
            const uint32_t l0 = pack_bf2(x0 - __uint_as_float(h0 << 16), x1 - __uint_as_float(h0 & 0xffff0000u));
            const uint32_t l1 = pack_bf2(x2 - __uint_as_float(h1 << 16), x3 - __uint_as_float(h1 & 0xffff0000u));
            bf16_t* p = img + base + i * (256 / VPR) * D;
            *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(p + R * D) = make_uint2(l0, l1);
        }
    }
};

__device__ __forceinline__ f32x16 mfma_bf16(const s16x8& a, const s16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), a),
                                                   __builtin_bit_cast(__bf16 __attribute__((ext_vector_type(8))), b), c, 0, 0, 0);
}

}  // namespace p3attn
