// p3hip LayerNorm forward / backward (HBM-bound; one wave per row, 4 rows per 256-thread block).
// cols <= 1024 and cols % 4 == 0 (the path uses 384, 768 and 256).
#include <stdlib.h>

#include "p3_common.h"

namespace {

// lane-group sums without the LDS crossbar where gfx9 DPP can do it: quad_perm [1,0,3,2] / [2,3,0,1] (lane ^ 1, lane ^ 2), row_half_mirror and
// row_mirror (every lane of a 16-lane row ends up with the row sum); only the hops across rows (16, 32) use ds_bpermute.  A LayerNorm row
// pays two such reductions; __shfl_xor compiles to ds_bpermute for every level (5-6 dependent LDS round trips).
template <int CTRL> __device__ __forceinline__ float ln_dpp(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float ln_row_sum(float v) {       // all 16 lanes of each DPP row <- sum over the row
    v += ln_dpp<0xB1>(v); v += ln_dpp<0x4E>(v); v += ln_dpp<0x141>(v); v += ln_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float ln_half_sum(float v) { v = ln_row_sum(v); return v + __shfl_xor(v, 16, 64); }                 // 32-lane halves
__device__ __forceinline__ float ln_wave_sum(float v) { v = ln_half_sum(v); return v + __shfl_xor(v, 32, 64); }                // whole wave

constexpr int MAXV = 4;  // up to 4 float4-chunks per lane -> cols <= 1024

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        uint2 raw = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        float4 raw = *reinterpret_cast<const float4*>(p);
        v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
    }
}
template <typename T>
__device__ __forceinline__ void store4(T* p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        uint2 raw; raw.x = pack_bf2(v[0], v[1]); raw.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = raw;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// planes (include/p3hip.h): hi = bf16(x), lo = bf16(x - hi) of four values
__device__ __forceinline__ void store4_planes(bf16_t* ph, bf16_t* pl, const float (&v)[4]) {
    uint2 h, l;
    h.x = pack_bf2(v[0], v[1]); h.y = pack_bf2(v[2], v[3]);
    l.x = pack_bf2(v[0] - __uint_as_float(h.x << 16), v[1] - __uint_as_float(h.x & 0xffff0000u));
    l.y = pack_bf2(v[2] - __uint_as_float(h.y << 16), v[3] - __uint_as_float(h.y & 0xffff0000u));
    *reinterpret_cast<uint2*>(ph) = h;
    *reinterpret_cast<uint2*>(pl) = l;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TO* __restrict__ y, int64_t rows,
                                                     int cols, int ldx, int ldy, float eps, float* __restrict__ smean,
                                                     float* __restrict__ srstd, bf16_t* __restrict__ y_lo = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = cols >> 2;
    float v[MAXV][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            load4<TI>(x + row * ldx + ci * 4, v[c]);
            s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
        }
    }
    const float mean = ln_wave_sum(s) / (float)cols;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float dlt = v[c][i] - mean; q += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(ln_wave_sum(q) / (float)cols + eps);
#pragma unroll
    for (int c = 0; c < MAXV; ++c) {
        const int ci = lane + 64 * c;
        if (ci < nchunk) {
            float g[4], b[4], o[4];
            load4<float>(gamma + ci * 4, g);
            load4<float>(beta + ci * 4, b);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (v[c][i] - mean) * rstd * g[i] + b[i];
            if constexpr (sizeof(TO) == 2) {
                if (y_lo) { store4_planes(reinterpret_cast<bf16_t*>(y) + row * ldy + ci * 4, y_lo + row * ldy + ci * 4, o); continue; }
            }
            store4<TO>(y + row * ldy + ci * 4, o);
        }
    }
    if (lane == 0 && smean) { smean[row] = mean; srstd[row] = rstd; }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma/dbeta accumulated with one atomic per block-column
// NV = float4-chunks per lane (1: cols <= 256, 2: <= 512, 4: <= 1024): the per-row register arrays scale with it
template <typename TDY, typename TX, typename TDX, int NV>
__global__ __launch_bounds__(256, (NV == 4 ? 2 : 4)) void ln_bwd_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const TDX* __restrict__ dres, TDX* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows,
                                                     int cols, int rows_per_block) {
    __shared__ float sg[4][NV * 256], sb[4][NV * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchunk = cols >> 2;
    float ag[NV][4], ab[NV][4];
#pragma unroll
    for (int c = 0; c < NV; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) { ag[c][i] = 0.f; ab[c][i] = 0.f; }
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    int64_t rend = rbeg + rows_per_block;
    if (rend > rows) rend = rows;
    for (int64_t row = rbeg + w; row < rend; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        float g[NV][4], xh[NV][4], d[NV][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk) {
                float xv[4];
                load4<TX>(x + row * cols + ci * 4, xv);
                load4<TDY>(dy + row * cols + ci * 4, d[c]);
                load4<float>(gamma + ci * 4, g[c]);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[c][i] = (xv[i] - mu) * rs;
                    const float gd = g[c][i] * d[c][i];
                    s1 += gd; s2 += gd * xh[c][i];
                    ag[c][i] += d[c][i] * xh[c][i];
                    ab[c][i] += d[c][i];
                }
            }
        }
        s1 = ln_wave_sum(s1) / (float)cols;
        s2 = ln_wave_sum(s2) / (float)cols;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = rs * (g[c][i] * d[c][i] - s1 - xh[c][i] * s2);
                if (dres) {   // gradient of the residual branch that forked off x: summed here instead of by a separate add pass
                    float rr[4];
                    load4<TDX>(dres + row * cols + ci * 4, rr);
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += rr[i];
                }
                store4<TDX>(dx + row * cols + ci * 4, o);
            }
        }
    }
    if (dgamma) {
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const int ci = lane + 64 * c;
            if (ci < nchunk)
#pragma unroll
                for (int i = 0; i < 4; ++i) { sg[w][ci * 4 + i] = ag[c][i]; sb[w][ci * 4 + i] = ab[c][i]; }
        }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < cols; cidx += 256) {
            atomicAdd(dgamma + cidx, (sg[0][cidx] + sg[1][cidx]) + (sg[2][cidx] + sg[3][cidx]));
            atomicAdd(dbeta + cidx, (sb[0][cidx] + sb[1][cidx]) + (sb[2][cidx] + sb[3][cidx]));
        }
    }
}

// Half-wave-per-row form for cols % 128 == 0 (the path's 256 / 384 / 768): 32 lanes x CPL float4 chunks cover a row exactly (the one-wave-
// per-row form leaves a quarter of the lanes idle at 384 columns and keeps ONE row per wave in flight: 3.6 TB/s, 91 % of the wave time
// waiting, r01 SQ counters), so a wave carries two rows at once, gamma stays in registers for the whole block, and an optional bf16
// copy of dx (`dx_lo`) is written from the same registers - the residual-gradient stream is fp32, but the GEMMs of the sublayer
// below consume it in bf16: this saves their separate cast pass (a 77 MB read + a launch per sublayer).
// HAS_RES: the incoming residual gradient is loaded WITH x / dy at the top of the row (r02 loaded each chunk right before its store,
// behind `if (dres)`: three serialised load -> s_waitcnt vmcnt(0) -> store round trips per row).
// LODROP: the bf16 copy additionally carries the dropout mask of the sublayer below (the Linear that produced this LayerNorm's input
// applied dropout to its output: its backward wants mask(dx)/(1-p) in bf16 - same (seed, site, row, col) hash as the forward epilogue),
// which replaces that sublayer's dropout_apply pass over the fp32 stream.  dx itself (the residual path) stays unmasked.
template <typename TDY, typename TX, typename TDX, int CPL, bool HAS_RES, bool LODROP = false>
__global__ __launch_bounds__(256, 4) void ln_bwd_half_kernel(const TDY* __restrict__ dy, const TX* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const TDX* __restrict__ dres, TDX* __restrict__ dx, bf16_t* __restrict__ dx_lo,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int rows_per_block,
                                                             float* __restrict__ slab, p3_dropout lo_drop = p3_dropout{nullptr, 0u, 0.f},
                                                             bf16_t* __restrict__ dx_lo2 = nullptr, int ld_lo = CPL * 128) {
    constexpr int cols = CPL * 128;
    DropKey dk;
    if constexpr (LODROP) dk = drop_key(lo_drop);
    __shared__ float sg[8][cols], sb[8][cols];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    float g[CPL][4], ag[CPL][4], ab[CPL][4];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        load4<float>(gamma + (l31 + 32 * c) * 4, g[c]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ag[c][i] = 0.f; ab[c][i] = 0.f; }
    }
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    int64_t rend = rbeg + rows_per_block;
    if (rend > rows) rend = rows;
    for (int64_t r0 = rbeg + w * 2; r0 < rend; r0 += 8) {
        const int64_t row = r0 + half;
        const bool ok = row < rend;
        const int64_t rr = ok ? row : rend - 1;
        const float mu = mean[rr], rs = rstd[rr];
        float xh[CPL][4], d[CPL][4], rres[HAS_RES ? CPL : 1][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int64_t o = rr * cols + (l31 + 32 * c) * 4;
            float xv[4];
            load4<TX>(x + o, xv);
            load4<TDY>(dy + o, d[c]);
            if constexpr (HAS_RES) load4<TDX>(dres + o, rres[c]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xh[c][i] = (xv[i] - mu) * rs;
                const float gd = g[c][i] * d[c][i];
                s1 += gd; s2 += gd * xh[c][i];
                if (ok) { ag[c][i] += d[c][i] * xh[c][i]; ab[c][i] += d[c][i]; }
            }
        }
        s1 = ln_half_sum(s1); s2 = ln_half_sum(s2);                                                      // within the 32-lane half
        s1 *= 1.f / (float)cols; s2 *= 1.f / (float)cols;
        if (ok) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int64_t o = row * cols + (l31 + 32 * c) * 4;
                float ov[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[i] = rs * (g[c][i] * d[c][i] - s1 - xh[c][i] * s2);
                if constexpr (HAS_RES) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ov[i] += rres[c][i];
                }
                store4<TDX>(dx + o, ov);
                if constexpr (LODROP) {
                    const uint32_t rk = drop_rowkey(dk, (uint64_t)row), cb = (uint32_t)(l31 + 32 * c) * 4u;
                    const uint32_t b0 = drop_bits(rk, drop_colkey(dk, cb)), b1 = drop_bits(rk, drop_colkey(dk, cb + 2u));
                    float lv[4];
                    lv[0] = drop_keep_lo(dk, b0) ? ov[0] * dk.inv_keep : 0.f; lv[1] = drop_keep_hi(dk, b0) ? ov[1] * dk.inv_keep : 0.f;
                    lv[2] = drop_keep_lo(dk, b1) ? ov[2] * dk.inv_keep : 0.f; lv[3] = drop_keep_hi(dk, b1) ? ov[3] * dk.inv_keep : 0.f;
                    store4<bf16_t>(dx_lo + o, lv);
                } else {
                    // dx_lo2: the copy is PLANES (hi -> dx_lo, lo -> dx_lo2, row stride ld_lo): the operand of the fp32x3 block's backward GEMMs
                    if (dx_lo2) store4_planes(dx_lo + row * ld_lo + (l31 + 32 * c) * 4, dx_lo2 + row * ld_lo + (l31 + 32 * c) * 4, ov);
                    else if (dx_lo) store4<bf16_t>(dx_lo + row * ld_lo + (l31 + 32 * c) * 4, ov);
                }
            }
        }
    }
    if (dgamma) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) { sg[w * 2 + half][(l31 + 32 * c) * 4 + i] = ag[c][i]; sb[w * 2 + half][(l31 + 32 * c) * 4 + i] = ab[c][i]; }
        __syncthreads();
        for (int cidx = threadIdx.x; cidx < cols; cidx += 256) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) { a += sg[q][cidx]; b += sb[q][cidx]; }
            // slab: this workgroup's partial row [dgamma | dbeta], summed in workgroup order by p3_det_reduce2 - ~1000 workgroups x 2 cols
            // same-address atomics per launch otherwise (r03: the chains, not the 200 MB of rows, bounded this kernel)
            if (slab) { slab[(int64_t)blockIdx.x * (2 * cols) + cidx] = a; slab[(int64_t)blockIdx.x * (2 * cols) + cols + cidx] = b; }
            else { atomicAdd(dgamma + cidx, a); atomicAdd(dbeta + cidx, b); }
        }
    }
}

}  // namespace

// Forward, half-wave-per-row form for cols = CPL x 128 (256 / 384 / 768), planes output (r06): the one-wave-per-row kernel above keeps ONE row per wave in flight and
// leaves half its lanes without a second chunk at 384 columns (4.5 TB/s on the ViT's 50 240 x 384 rows); here a half-wave owns a row (32 lanes x CPL float4 chunks cover it
// exactly), every half-wave walks rows r, r + 8, ... of its block TWO at a time (2 x CPL loads in flight before the first reduction), gamma / beta stay in registers.
// Same arithmetic per element; the row sums are taken over 32 lanes x CPL x 4 values instead of 64 lanes - equal to the rounding of an fp32 sum.
template <int CPL>
__global__ __launch_bounds__(256) void ln_fwd_half_planes_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 bf16_t* __restrict__ y_hi, bf16_t* __restrict__ y_lo, int64_t rows, int ldx, int ldy, float eps,
                                                                 float* __restrict__ smean, float* __restrict__ srstd, int rows_per_block) {
    constexpr int cols = CPL * 128;
    const int lane = threadIdx.x & 63, l31 = lane & 31, hw = (threadIdx.x >> 6) * 2 + (lane >> 5);      // half-wave 0..7 of the block
    float g[CPL][4], b[CPL][4];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { load4<float>(gamma + (l31 + 32 * c) * 4, g[c]); load4<float>(beta + (l31 + 32 * c) * 4, b[c]); }
    const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block;
    int64_t rend = rbeg + rows_per_block;
    if (rend > rows) rend = rows;
    for (int64_t r0 = rbeg + hw; r0 < rend; r0 += 16) {
        float v[2][CPL][4];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t row = r0 + 8 * u;
            ok[u] = row < rend;
            const int64_t rr = ok[u] ? row : rend - 1;
#pragma unroll
            for (int c = 0; c < CPL; ++c) load4<float>(x + rr * ldx + (l31 + 32 * c) * 4, v[u][c]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c) s += (v[u][c][0] + v[u][c][1]) + (v[u][c][2] + v[u][c][3]);
            const float mean = ln_half_sum(s) * (1.f / (float)cols);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const float dlt = v[u][c][i] - mean; q += dlt * dlt; }
            const float rstd = rsqrtf(ln_half_sum(q) * (1.f / (float)cols) + eps);
            if (!ok[u]) continue;
            const int64_t row = r0 + 8 * u;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (v[u][c][i] - mean) * rstd * g[c][i] + b[c][i];
                store4_planes(y_hi + row * ldy + (l31 + 32 * c) * 4, y_lo + row * ldy + (l31 + 32 * c) * 4, o);
            }
            if (l31 == 0 && smean) { smean[row] = mean; srstd[row] = rstd; }
        }
    }
}

extern "C" int p3_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t rows, int cols, int ldx,
                            int ldy, float eps, int dtype_in, int dtype_out, float* save_mean, float* save_rstd, void* stream) {
    P3_CHECK(x && gamma && beta && y, P3_EINVAL, "p3_layernorm: null pointer");
    P3_CHECK(cols > 0 && cols <= 1024 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, P3_ESHAPE, "p3_layernorm: cols must be <=1024 and %4");
    if (rows <= 0) return P3_OK;
    dim3 grid(p3_ceil_div(rows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LN_LAUNCH(TI, TO) \
    hipLaunchKernelGGL((ln_fwd_kernel<TI, TO>), grid, block, 0, s, (const TI*)x, gamma, beta, (TO*)y, rows, cols, ldx, ldy, eps, save_mean, save_rstd)
    if (dtype_in == P3_F32 && dtype_out == P3_F32) LN_LAUNCH(float, float);
    else if (dtype_in == P3_F32 && dtype_out == P3_BF16) LN_LAUNCH(float, bf16_t);
    else if (dtype_in == P3_BF16 && dtype_out == P3_BF16) LN_LAUNCH(bf16_t, bf16_t);
    else if (dtype_in == P3_BF16 && dtype_out == P3_F32) LN_LAUNCH(bf16_t, float);
    else { p3_set_error("p3_layernorm: dtype"); return P3_EUNSUP; }
#undef LN_LAUNCH
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                                void* dx, float* dgamma, float* dbeta, int64_t rows, int cols, int dtype_dy, int dtype_x,
                                int dtype_dx, void* stream) {
    return p3_layernorm_bwd_res(dy, x, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, rows, cols, dtype_dy, dtype_x, dtype_dx, stream);
}

extern "C" int p3_layernorm_bwd_res(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                                    void* dx, float* dgamma, float* dbeta, int64_t rows, int cols, int dtype_dy, int dtype_x,
                                    int dtype_dx, void* stream) {
    return p3_layernorm_bwd_lo(dy, x, gamma, mean, rstd, dres, dx, nullptr, dgamma, dbeta, rows, cols, dtype_dy, dtype_x, dtype_dx, stream);
}

extern "C" int p3_layernorm_bwd_lo(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                                   void* dx, void* dx_lo, float* dgamma, float* dbeta, int64_t rows, int cols, int dtype_dy, int dtype_x,
                                   int dtype_dx, void* stream) {
    return p3_layernorm_bwd_lo_drop(dy, x, gamma, mean, rstd, dres, dx, dx_lo, nullptr, dgamma, dbeta, rows, cols, dtype_dy, dtype_x, dtype_dx, stream);
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                       void* dx, void* dx_lo, void* dx_lo2, int ld_lo, const p3_dropout* lo_drop, float* dgamma, float* dbeta, int64_t rows, int cols,
                       int dtype_dy, int dtype_x, int dtype_dx, void* stream);

extern "C" int p3_layernorm_bwd_lo_drop(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                                        void* dx, void* dx_lo, const p3_dropout* lo_drop, float* dgamma, float* dbeta, int64_t rows, int cols,
                                        int dtype_dy, int dtype_x, int dtype_dx, void* stream) {
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dx_lo, nullptr, cols, lo_drop, dgamma, dbeta, rows, cols, dtype_dy, dtype_x, dtype_dx, stream);
}

extern "C" int p3_layernorm_bwd_planes(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres, float* dx,
                                       void* dx_hi, void* dx_lo, int ld_planes, float* dgamma, float* dbeta, int64_t rows, int cols, void* stream) {
    P3_CHECK(dx_hi && dx_lo && ld_planes % 4 == 0 && ((uintptr_t)dx_hi | (uintptr_t)dx_lo) % 8 == 0, P3_EINVAL, "p3_layernorm_bwd_planes: planes arguments");
    P3_CHECK(cols == 256 || cols == 384 || cols == 768, P3_EUNSUP, "p3_layernorm_bwd_planes: 256 / 384 / 768 columns");
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dres, dx, dx_hi, dx_lo, ld_planes, nullptr, dgamma, dbeta, rows, cols, P3_F32, P3_F32, P3_F32, stream);
}

extern "C" int p3_layernorm_planes(const float* x, const float* gamma, const float* beta, void* y_hi, void* y_lo, int64_t rows, int cols, int ldx, int ldy, float eps,
                                   float* save_mean, float* save_rstd, void* stream) {
    P3_CHECK(x && gamma && beta && y_hi && y_lo, P3_EINVAL, "p3_layernorm_planes: null pointer");
    P3_CHECK(cols > 0 && cols <= 1024 && cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, P3_ESHAPE, "p3_layernorm_planes: cols must be <=1024 and %4");
    P3_CHECK((save_mean == nullptr) == (save_rstd == nullptr), P3_EINVAL, "p3_layernorm_planes: save_mean / save_rstd go together");
    if (rows <= 0) return P3_OK;
    static int half_on = -1;                // P3_LN_HALF=0: the one-wave-per-row kernel for every width (A/B switch)
    if (half_on < 0) { const char* e = getenv("P3_LN_HALF"); half_on = (e && atoi(e) == 0) ? 0 : 1; }
    if (half_on && (cols == 256 || cols == 384 || cols == 768)) {          // at EVERY row count: a tile alone must give the bits it gives in a batch
        constexpr int RPB = 32;             // rows per block: four rows per half-wave, two at a time
        const dim3 grid(p3_ceil_div(rows, RPB)), block(256);
#define LNH(CPL) hipLaunchKernelGGL((ln_fwd_half_planes_kernel<CPL>), grid, block, 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y_hi, (bf16_t*)y_lo, rows, ldx, ldy, eps, save_mean, save_rstd, RPB)
        if (cols == 256) LNH(2); else if (cols == 384) LNH(3); else LNH(6);
#undef LNH
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    hipLaunchKernelGGL((ln_fwd_kernel<float, bf16_t>), dim3(p3_ceil_div(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y_hi, rows, cols, ldx, ldy, eps,
                       save_mean, save_rstd, (bf16_t*)y_lo);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

static int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* dres,
                       void* dx, void* dx_lo, void* dx_lo2, int ld_lo, const p3_dropout* lo_drop, float* dgamma, float* dbeta, int64_t rows, int cols,
                       int dtype_dy, int dtype_x, int dtype_dx, void* stream) {
    P3_CHECK(dy && x && gamma && mean && rstd && dx, P3_EINVAL, "p3_layernorm_bwd: null pointer");
    const bool lod = lo_drop && lo_drop->seed && lo_drop->p > 0.f;
    P3_CHECK(!lod || (dx_lo && !dres && dtype_dy == P3_BF16 && dtype_x == P3_F32 && dtype_dx == P3_F32 && (cols == 256 || cols == 384 || cols == 768)),
             P3_EUNSUP, "p3_layernorm_bwd: a masked bf16 copy needs dx_lo, no dres, bf16 dy, fp32 x / dx and 256 / 384 / 768 columns");
    P3_CHECK(!dx_lo || ((cols == 256 || cols == 384 || cols == 768) && dtype_dx == P3_F32), P3_EUNSUP,
             "p3_layernorm_bwd: the bf16 copy is written by the half-wave kernel only: 256 / 384 / 768 columns and an fp32 dx");
    P3_CHECK(cols > 0 && cols <= 1024 && cols % 4 == 0, P3_ESHAPE, "p3_layernorm_bwd: cols must be <=1024 and %4");
    P3_CHECK((dgamma == nullptr) == (dbeta == nullptr), P3_EINVAL, "p3_layernorm_bwd: dgamma/dbeta go together");
    if (rows <= 0) return P3_OK;
    // rows per block trades resident waves (one row in flight per wave) against dgamma / dbeta atomics per address; same-box sweep of
    // the train step (r01, P3_LN_RPB): 16 -> 60.9 ms, 32 -> 60.2, 48 -> 59.8, 96 -> 60.3, 128 -> 60.5, 256 -> 62.7
    // r03, with the parameter sums going through the slab (no atomic chains to shorten): 32 -> 35.2 us, 40 -> 37.7, 48 -> 39.3, 56 -> 37.3, 64 -> 40.4
    const bool half_cols = cols == 256 || cols == 384 || cols == 768;
    const int rpb = (half_cols && dgamma && p3_reduce_scratch(1)) ? 32 : 48;
    dim3 grid(p3_ceil_div(rows, rpb)), block(256);
    hipStream_t s = (hipStream_t)stream;
    // dgamma / dbeta partials of the half-wave kernel go through the registered scratch (none registered / too small: atomics)
    const bool halfk = half_cols;
    const int nblk = (int)grid.x;
    const int64_t slab_floats = (int64_t)nblk * 2 * cols, tmp_floats = (int64_t)p3_ceil_div(nblk, 128) * 2 * cols;
    // p3_reduce_defer active: the partials are parked and added with every other parked set by ONE p3_reduce_flush launch (42 reduce launches less per train step)
    float* parked = (halfk && dgamma && nblk > 8 && p3_reduce_scratch(1)) ? p3_reduce_park(slab_floats, nblk, 2 * cols, cols, dgamma, dbeta) : nullptr;
    float* slab = parked ? parked : ((halfk && dgamma && nblk > 8) ? p3_reduce_scratch(slab_floats + tmp_floats) : nullptr);
    auto finish = [&]() { return (slab && !parked) ? p3_det_reduce2(slab, nblk, 2 * cols, slab + slab_floats, dgamma, dbeta, cols, 2 * cols, 1, s) : P3_OK; };
    if (lod) {
#define LNH_D(CPL) hipLaunchKernelGGL((ln_bwd_half_kernel<bf16_t, float, float, CPL, false, true>), grid, block, 0, s, (const bf16_t*)dy, (const float*)x, gamma, mean, rstd, (const float*)nullptr, (float*)dx, (bf16_t*)dx_lo, dgamma, dbeta, rows, rpb, slab, *lo_drop)
        if (cols == 256) LNH_D(2); else if (cols == 384) LNH_D(3); else LNH_D(6);
#undef LNH_D
        P3_LAUNCH_CHECK();
        return finish();
    }
    if (half_cols) {
#define LNH_R(TDY, TX, TDX, CPL, RES) \
    hipLaunchKernelGGL((ln_bwd_half_kernel<TDY, TX, TDX, CPL, RES>), grid, block, 0, s, (const TDY*)dy, (const TX*)x, gamma, mean, rstd, (const TDX*)dres, (TDX*)dx, (bf16_t*)dx_lo, dgamma, dbeta, rows, rpb, slab, p3_dropout{nullptr, 0u, 0.f}, (bf16_t*)dx_lo2, ld_lo)
#define LNH_C(TDY, TX, TDX, CPL) \
    do { if (dres) LNH_R(TDY, TX, TDX, CPL, true); else LNH_R(TDY, TX, TDX, CPL, false); } while (0)
#define LNH(TDY, TX, TDX) \
    do { if (cols == 256) LNH_C(TDY, TX, TDX, 2); else if (cols == 384) LNH_C(TDY, TX, TDX, 3); else LNH_C(TDY, TX, TDX, 6); } while (0)
        if (dtype_dy == P3_F32 && dtype_x == P3_F32 && dtype_dx == P3_F32) LNH(float, float, float);
        else if (dtype_dy == P3_BF16 && dtype_x == P3_F32 && dtype_dx == P3_F32) LNH(bf16_t, float, float);
        else if (dtype_dy == P3_BF16 && dtype_x == P3_BF16 && dtype_dx == P3_BF16) LNH(bf16_t, bf16_t, bf16_t);
        else if (dtype_dy == P3_BF16 && dtype_x == P3_F32 && dtype_dx == P3_BF16) LNH(bf16_t, float, bf16_t);      // fp32 stream, bf16 gradient stream
        else if (dtype_dy == P3_F32 && dtype_x == P3_BF16 && dtype_dx == P3_F32) LNH(float, bf16_t, float);
        else { p3_set_error("p3_layernorm_bwd: dtype combination"); return P3_EUNSUP; }
#undef LNH
#undef LNH_C
#undef LNH_R
        P3_LAUNCH_CHECK();
        return finish();
    }
#define LNB_NV(TDY, TX, TDX, NV) \
    hipLaunchKernelGGL((ln_bwd_kernel<TDY, TX, TDX, NV>), grid, block, 0, s, (const TDY*)dy, (const TX*)x, gamma, mean, rstd, (const TDX*)dres, (TDX*)dx, dgamma, dbeta, rows, cols, rpb)
#define LNB(TDY, TX, TDX) \
    do { if (cols <= 256) LNB_NV(TDY, TX, TDX, 1); else if (cols <= 512) LNB_NV(TDY, TX, TDX, 2); else LNB_NV(TDY, TX, TDX, 4); } while (0)
    if (dtype_dy == P3_F32 && dtype_x == P3_F32 && dtype_dx == P3_F32) LNB(float, float, float);
    else if (dtype_dy == P3_BF16 && dtype_x == P3_F32 && dtype_dx == P3_F32) LNB(bf16_t, float, float);
    else if (dtype_dy == P3_BF16 && dtype_x == P3_BF16 && dtype_dx == P3_BF16) LNB(bf16_t, bf16_t, bf16_t);
    else if (dtype_dy == P3_BF16 && dtype_x == P3_F32 && dtype_dx == P3_BF16) LNB(bf16_t, float, bf16_t);
    else if (dtype_dy == P3_F32 && dtype_x == P3_BF16 && dtype_dx == P3_F32) LNB(float, bf16_t, float);
    else { p3_set_error("p3_layernorm_bwd: dtype combination"); return P3_EUNSUP; }
#undef LNB
#undef LNB_NV
    P3_LAUNCH_CHECK();
    return P3_OK;
}
