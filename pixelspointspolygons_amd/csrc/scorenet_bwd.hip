// p3hip ScoreNet backward (models/pix2poly/model_pix2poly.py:69-112), never materialising the [B,512,N,N] pair tensor.
//
// The forward keeps, per ScoreNet: U, V [B*N,256] (conv1 split over (i,j)), H2 [R,128], H3 [R,64] (pre-BatchNorm conv
// outputs, R = B*N*N rows) and the BN (scale, shift, mean, rstd) triples.  Backward chain, all HBM-streaming kernels +
// MFMA GEMMs (p3_gemm for dA = dH . W, p3_gemm_tn_ex with a generated B operand for dW = dH^T . relu(bn(prev))):
//   row_affine_bwd<TAIL>   dS -> dH3' = dz3*sc3, d(sc3, sh3, w4, b4)
//   bn_bwd_coeffs          d(scale, shift) -> d(gamma, beta) and the per-channel (a, b) of the statistics path
//   affine_fix             dH = dH' + a + b*H                         (gradient through the batch mean / variance)
//   row_affine_bwd<MAT>    dA -> dH' = dA*(z>0)*sc, d(sc, sh)          (C = 128)
//   pair_bwd               dA2 [R,256] -> dU, dV [B*N,256], d(sc1, sh1) (sums over j / i of the pair grid)
//   pair_stats_bwd         closed-form BN1 statistics path into dU, dV
#include <stdlib.h>

#include "p3_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void ld4(const T* p, float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        const uint2 raw = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        const float4 raw = *reinterpret_cast<const float4*>(p);
        v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
    }
}
template <typename T>
__device__ __forceinline__ void st4(T* p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        uint2 raw; raw.x = pack_bf2(v[0], v[1]); raw.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = raw;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// C channels, LPR = C/4 lanes per row, 64/LPR rows per wave pass; every lane owns 4 fixed channels -> register accumulators.
// TAIL: upstream is a per-row scalar g[r] (dS, optionally transposed) times w4[c];  else upstream is the matrix dA [R,C].
// Writes dHd = dz * sc (may alias dA).  acc layout: [dsc(C) | dsh(C) | dw4(C) | db4(1)].
template <typename T, int C, bool TAIL>
__global__ __launch_bounds__(256) void row_affine_bwd_kernel(const T* __restrict__ dA, const float* __restrict__ dS, const T* __restrict__ H,
                                                             const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ mean,
                                                             const float* __restrict__ w4, T* __restrict__ dHd, float* __restrict__ acc,
                                                             int64_t R, int N, int transpose, const float* __restrict__ fix_a,
                                                             const float* __restrict__ fix_b, float* __restrict__ slab) {
    // Two-pass use (train-mode BatchNorm): pass 1 with dHd == NULL only accumulates the sums (no store), pass 2 with fix_a / fix_b
    // writes the FINAL gradient dz*sc + a + b*H in one go (acc == NULL: no sums) - one read of H less and no separate affine_fix pass.
    constexpr int LPR = C / 4, RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR, w = threadIdx.x >> 6;
    const int c0 = sub * 4;
    float s[4], h[4], wv[4], mu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; mu[k] = mean[c0 + k]; wv[k] = TAIL ? w4[c0 + k] : 0.f; }
    float a_sc[4] = {0, 0, 0, 0}, a_sh[4] = {0, 0, 0, 0}, a_w[4] = {0, 0, 0, 0}, a_b = 0.f;
    float fa[4] = {0, 0, 0, 0}, fb[4] = {0, 0, 0, 0};
    if (fix_a) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { fa[k] = fix_a[c0 + k]; fb[k] = fix_b[c0 + k]; }
    }
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + w, nwaves = (int64_t)gridDim.x * 4;
    // r03: UNR row groups per step with all their loads issued first (the one-group-per-step loop was a chain of two dependent loads per
    // iteration: the C = 64 tail form streamed its 302 MB at 2.2 TB/s)
    constexpr int UNR = TAIL ? 4 : 1;          // rocprofv3 A/B: tail form 135 -> 113 us; the C = 128 matrix form got slower unrolled (298 -> 317 us)
    for (int64_t r0 = wave_id * RPW; r0 < R; r0 += nwaves * RPW * UNR) {
        float hv[UNR][4], g[UNR][4], gr[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t r = r0 + (int64_t)u * nwaves * RPW + rsel;
            ok[u] = r < R;
            const int64_t rc = ok[u] ? r : R - 1;
            ld4<T>(H + rc * C + c0, hv[u]);
            gr[u] = 0.f;
            if constexpr (TAIL) {
                int64_t rs = rc;
                if (transpose) { const int64_t nn = (int64_t)N * N, b = rc / nn, p = rc - b * nn; const int i = (int)(p / N), j = (int)(p - (int64_t)i * N); rs = b * nn + (int64_t)j * N + i; }
                gr[u] = dS[rs];
            } else {
                ld4<T>(dA + rc * C + c0, g[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (!ok[u]) continue;
            const int64_t r = r0 + (int64_t)u * nwaves * RPW + rsel;
            if constexpr (TAIL) {
                if (sub == 0) a_b += gr[u];
#pragma unroll
                for (int k = 0; k < 4; ++k) { g[u][k] = gr[u] * wv[k]; a_w[k] += gr[u] * fmaxf(hv[u][k] * s[k] + h[k], 0.f); }
            }
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dz = (hv[u][k] * s[k] + h[k] > 0.f) ? g[u][k] : 0.f;
                a_sc[k] += dz * (hv[u][k] - mu[k]); a_sh[k] += dz;   // centred: sum dz*(H - mean) is what BN backward needs (no cancellation)
                o[k] = dz * s[k] + fa[k] + fb[k] * hv[u][k];
            }
            if (dHd) st4<T>(dHd + r * C + c0, o);
        }
    }
    if (!acc) return;
    // combine the row groups of the wave, then the 4 waves, then one atomic per value per block
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { a_sc[k] += __shfl_xor(a_sc[k], o, 64); a_sh[k] += __shfl_xor(a_sh[k], o, 64); if (TAIL) a_w[k] += __shfl_xor(a_w[k], o, 64); }
        if (TAIL) a_b += __shfl_xor(a_b, o, 64);
    }
    __shared__ float red[4][3 * C + 1];
    if (rsel == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[w][c0 + k] = a_sc[k]; red[w][C + c0 + k] = a_sh[k]; red[w][2 * C + c0 + k] = a_w[k]; }
        if (sub == 0) red[w][3 * C] = a_b;
    }
    __syncthreads();
    const int nvals = TAIL ? 3 * C + 1 : 2 * C;
    // slab != NULL (deterministic mode): the block's partials are stored and det_reduce_kernel adds them in block order in float64
    for (int i = threadIdx.x; i < nvals; i += 256) {
        const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (slab) slab[(int64_t)blockIdx.x * nvals + i] = v;
        else atomicAdd(acc + i, v);
    }
}

// scale = gamma*rstd, shift = beta - mean*scale, mean = S1/n, var = S2/n - mean^2  ->  dgamma, dbeta and (a, b) such that
// d(pre-BN value) = direct + a[c] + b[c]*value  is the full train-mode BatchNorm backward.  Eval mode: a = b = 0.
__global__ void bn_bwd_coeffs_kernel(const float* __restrict__ dscale, const float* __restrict__ dshift, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ rstd, float count, int training, int C,
                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ a, float* __restrict__ b) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    // dscale here is the CENTRED sum  sum dz*(pre - mean)  (= dscale_raw - dshift*mean, accumulated without cancellation)
    const float ds = dscale[c], dh = dshift[c], g = gamma[c], m = mean[c], rs = rstd[c];
    if (training & 2) { dgamma[c] += ds * rs; dbeta[c] += dh; }      // accumulate into the gradient arena (single writer per channel)
    else { dgamma[c] = ds * rs; dbeta[c] = dh; }
    if (training & 1) {
        const float dmean = -dh * g * rs;
        const float drstd = ds * g;
        const float dvar = -0.5f * drstd * rs * rs * rs;
        a[c] = (dmean - 2.f * m * dvar) / count;
        b[c] = 2.f * dvar / count;
    } else { a[c] = 0.f; b[c] = 0.f; }
}

template <typename T>
__global__ void affine_fix_kernel(T* __restrict__ dH, const T* __restrict__ H, const float* __restrict__ a, const float* __restrict__ b,
                                  int64_t n4, int C, int ldh) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)((i * 4) % C);
        const int64_t r = (i * 4) / C;
        float d[4], h[4];
        ld4<T>(dH + i * 4, d); ld4<T>(H + r * ldh + c0, h);
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] += a[c0 + k] + b[c0 + k] * h[k];
        st4<T>(dH + i * 4, d);
    }
}

// pair grid backward: rows (b,i,j).  Block = (b, chunk of IC rows i), 4 wave-groups split the j range, a lane owns 4 channels
// (8-byte loads): dU rows are completed inside the block (LDS fold over the groups), dV[b,j] gets one atomic per block.
template <typename T, int IC>
__global__ __launch_bounds__(256) void pair_bwd_kernel(const T* __restrict__ dA, const T* __restrict__ U, const T* __restrict__ V,
                                                       const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ mean,
                                                       float* __restrict__ dU, float* __restrict__ dV, float* __restrict__ acc, int N, int C,
                                                       float* __restrict__ slab, float* __restrict__ acc_slab) {
    // slab / acc_slab != NULL (deterministic mode): dV partial rows -> slab[b][blockIdx.x][N][C] (summed by pair_dv_reduce_kernel in block
    // order), (dscale, dshift) partials -> acc_slab[b * gridDim.x + blockIdx.x][2C] (det_reduce_kernel): no atomics at all
    __shared__ float red[4][IC + 2][256];     // C == 256
    const int b = blockIdx.y, i0 = blockIdx.x * IC;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, c0 = lane * 4;
    float s[4], h[4], mu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; mu[k] = mean[c0 + k]; }
    float u[IC][4], au[IC][4];
#pragma unroll
    for (int i = 0; i < IC; ++i) {
        if (i0 + i < N) ld4<T>(U + ((int64_t)b * N + i0 + i) * C + c0, u[i]);
        else { u[i][0] = u[i][1] = u[i][2] = u[i][3] = 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) au[i][k] = 0.f;
    }
    float a_sc[4] = {0, 0, 0, 0}, a_sh[4] = {0, 0, 0, 0};
    for (int j = grp; j < N; j += 4) {
        float v[4], av[4] = {0, 0, 0, 0};
        ld4<T>(V + ((int64_t)b * N + j) * C + c0, v);
#pragma unroll
        for (int i = 0; i < IC; ++i) {
            if (i0 + i < N) {
                float g[4];
                ld4<T>(dA + (((int64_t)b * N + i0 + i) * N + j) * C + c0, g);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float p = u[i][k] + v[k];
                    const float dz = (p * s[k] + h[k] > 0.f) ? g[k] : 0.f;
                    a_sc[k] += dz * (p - mu[k]); a_sh[k] += dz;
                    au[i][k] += dz * s[k]; av[k] += dz * s[k];
                }
            }
        }
        if (slab) {
            *reinterpret_cast<float4*>(slab + (((int64_t)b * gridDim.x + blockIdx.x) * N + j) * C + c0) = make_float4(av[0], av[1], av[2], av[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(dV + ((int64_t)b * N + j) * C + c0 + k, av[k]);
        }
    }
#pragma unroll
    for (int i = 0; i < IC; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[grp][i][c0 + k] = au[i][k];
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[grp][IC][c0 + k] = a_sc[k]; red[grp][IC + 1][c0 + k] = a_sh[k]; }
    __syncthreads();
    const int c = threadIdx.x;
    for (int i = 0; i < IC; ++i)
        if (i0 + i < N) dU[((int64_t)b * N + i0 + i) * C + c] = (red[0][i][c] + red[1][i][c]) + (red[2][i][c] + red[3][i][c]);
    const float t_sc = (red[0][IC][c] + red[1][IC][c]) + (red[2][IC][c] + red[3][IC][c]);
    const float t_sh = (red[0][IC + 1][c] + red[1][IC + 1][c]) + (red[2][IC + 1][c] + red[3][IC + 1][c]);
    if (acc_slab) {
        float* part = acc_slab + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C;
        part[c] = t_sc; part[C + c] = t_sh;
    } else { atomicAdd(acc + c, t_sc); atomicAdd(acc + C + c, t_sh); }
}

// bf16 form with 16-byte accesses (r02: the 8-byte form above ran at 1.4 TB/s on the 1.2 GB of dA; 8-byte accesses reach 0.54-0.70x the
// rate of 16-byte ones on this chip): a lane owns EIGHT channels, the two half-waves take the two halves of the block's IC rows i, the
// dV partial sums of the halves are paired with v_permlane32_swap so that the atomic count per dV element stays N / IC.
template <int IC, bool PRE>
__global__ __launch_bounds__(256, 2) void pair_bwd_kernel16(const bf16_t* __restrict__ dA, const bf16_t* __restrict__ U, const bf16_t* __restrict__ V,
                                                         const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ mean,
                                                         float* __restrict__ dU, float* __restrict__ dV, float* __restrict__ acc, int N, int C,
                                                         float* __restrict__ slab, float* __restrict__ acc_slab) {
    // slab != NULL: the block's dV partial rows go to slab[b][blockIdx.x][N][C] with plain 16-byte stores and pair_dv_reduce_kernel sums
    // the N / IC slabs afterwards - no global atomics at all (r02: ~50 M fp32 atomics per launch bound this kernel at ~87 G atomics/s).
    constexpr int IH = IC / 2;
    __shared__ float red[4][IC + 2][256];     // C == 256; rows IC, IC+1: (dscale, dshift) partials
    const int b = blockIdx.y, i0 = blockIdx.x * IC;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5, c0 = l31 * 8;
    float s[8], h[8], mu[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; mu[k] = mean[c0 + k]; }
    auto unpack8 = [](const uint4& raw, float (&v)[8]) __attribute__((always_inline)) {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(w[q] << 16); v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u); }
    };
    uint4 upk[IH];                            // U rows stay packed (bf16 pairs): unpacked per use, 32 registers instead of 64
    float au[IH][8];
    const int ib = i0 + half * IH;
#pragma unroll
    for (int i = 0; i < IH; ++i) {
        upk[i] = (ib + i < N) ? *reinterpret_cast<const uint4*>(U + ((int64_t)b * N + ib + i) * C + c0) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) au[i][k] = 0.f;
    }
    float a_sc[8], a_sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a_sc[k] = 0.f; a_sh[k] = 0.f; }
    // r03: the rows of step j + 4 are loaded BEFORE step j is worked on (two register sets, SQ counters r03: 78 % of the wave cycles were
    // spent waiting with one set of IH loads in flight per wave - 601 us for a 1.2 GB stream)
    uint4 gnext[IH], vnext;
    auto load_j = [&](int j, uint4 (&g)[IH], uint4& vv) __attribute__((always_inline)) {
        const int jc = j < N ? j : N - 1;                  // clamped: the surplus loads of the last step are never used
        vv = *reinterpret_cast<const uint4*>(V + ((int64_t)b * N + jc) * C + c0);
#pragma unroll
        for (int i = 0; i < IH; ++i)
            g[i] = (ib + i < N) ? *reinterpret_cast<const uint4*>(dA + (((int64_t)b * N + ib + i) * N + jc) * C + c0) : make_uint4(0, 0, 0, 0);
    };
    if constexpr (PRE) load_j(grp, gnext, vnext);
    for (int j = grp; j < N; j += 4) {
        float v[8], av[8];
        uint4 graw[IH];
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < IH; ++i) graw[i] = gnext[i];
            unpack8(vnext, v);
            load_j(j + 4, gnext, vnext);
        } else {
            uint4 vv;
            load_j(j, graw, vv);
            unpack8(vv, v);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) av[k] = 0.f;
#pragma unroll
        for (int i = 0; i < IH; ++i) {
            float g[8], ui[8];
            unpack8(graw[i], g);
            asm volatile("" : "+v"(upk[i].x), "+v"(upk[i].y), "+v"(upk[i].z), "+v"(upk[i].w));   // keep the unpack inside the loop (no hoisting)
            unpack8(upk[i], ui);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float p = ui[k] + v[k];
                const float dz = (p * s[k] + h[k] > 0.f) ? g[k] : 0.f;
                a_sc[k] += dz * (p - mu[k]); a_sh[k] += dz;
                au[i][k] += dz * s[k]; av[k] += dz * s[k];
            }
        }
        // both halves hold partial dV[b, j, c0 .. c0+7] sums over their rows i: pair them, half 0 adds channels 0..3, half 1 channels 4..7
        float pr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(av[k]), __float_as_uint(av[k + 4]), false, false);
            // lo lane: r2[0] = own av[k], r2[1] = hi's av[k];  hi lane: r2[0] = lo's av[k+4], r2[1] = own av[k+4]
            pr[k] = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
        }
        if (slab) {
            *reinterpret_cast<float4*>(slab + (((int64_t)b * gridDim.x + blockIdx.x) * N + j) * C + c0 + 4 * half) = make_float4(pr[0], pr[1], pr[2], pr[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(dV + ((int64_t)b * N + j) * C + c0 + k + 4 * half, pr[k]);
        }
    }
#pragma unroll
    for (int i = 0; i < IH; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) red[grp][half * IH + i][c0 + k] = au[i][k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {            // the halves hold the same channels for different rows: fold, half 0 writes
        a_sc[k] += __shfl_xor(a_sc[k], 32, 64); a_sh[k] += __shfl_xor(a_sh[k], 32, 64);
        if (half == 0) { red[grp][IC][c0 + k] = a_sc[k]; red[grp][IC + 1][c0 + k] = a_sh[k]; }
    }
    __syncthreads();
    const int c = threadIdx.x;
    for (int i = 0; i < IC; ++i)
        if (i0 + i < N) dU[((int64_t)b * N + i0 + i) * C + c] = (red[0][i][c] + red[1][i][c]) + (red[2][i][c] + red[3][i][c]);
    const float t_sc = (red[0][IC][c] + red[1][IC][c]) + (red[2][IC][c] + red[3][IC][c]);
    const float t_sh = (red[0][IC + 1][c] + red[1][IC + 1][c]) + (red[2][IC + 1][c] + red[3][IC + 1][c]);
    if (acc_slab) {
        float* part = acc_slab + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C;
        part[c] = t_sc; part[C + c] = t_sh;
    } else { atomicAdd(acc + c, t_sc); atomicAdd(acc + C + c, t_sh); }
}

// dV[b, j, :] += sum over the nblk slabs of a tile (16-byte accesses, one thread per 4 channels)
__global__ void pair_dv_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dV, int nblk, int64_t per_tile /* N*C/4 */, int64_t total /* B*N*C/4 */) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per_tile, r = i - b * per_tile;
        const float4* src = reinterpret_cast<const float4*>(slab) + b * nblk * per_tile + r;
        float4 a = reinterpret_cast<float4*>(dV)[i];
        for (int k = 0; k < nblk; ++k) { const float4 v = src[(int64_t)k * per_tile]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
        reinterpret_cast<float4*>(dV)[i] = a;
    }
}

// (r02, measured and dropped: accumulating dU in LDS with ds_add_f32 over IC = 48 rows per block to cut the dV atomics to N / 48 per
// element ran 4x slower, 3.0 ms - LDS fp32 atomics retire far below the LDS load / store rate.  The register form stays bound by its
// ~50 M global fp32 atomics per launch (~87 G atomics/s on this chip); per-block dV slabs reduced by pair_stats_bwd are the next step.)

// BN1 statistics were closed-form in U, V: dU[b,i] += N*a + b1*(N*U[b,i] + sum_j V[b,j]), dV symmetric
template <typename T>
__global__ void pair_stats_bwd_kernel(const T* __restrict__ U, const T* __restrict__ V, const float* __restrict__ a, const float* __restrict__ bco,
                                      float* __restrict__ dU, float* __restrict__ dV, int N, int C) {
    // grid = (B, row chunks): every block recomputes the column sums over the N rows of its sample (N*C cached loads) and then
    // updates its own chunk of rows, so the kernel fills the chip instead of running on B = 64 workgroups
    const int b = blockIdx.x;
    const int per = (N + gridDim.y - 1) / gridDim.y, i0 = blockIdx.y * per, i1 = min(N, i0 + per);
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        // eight rows in flight per step (four partial sums per operand): a one-row-at-a-time loop is a chain of N dependent load latencies
        float pu[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f};
        const T* up = U + (int64_t)b * N * C + c;
        const T* vp = V + (int64_t)b * N * C + c;
        int i = 0;
        for (; i + 8 <= N; i += 8) {
            float tu[8], tv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { tu[q] = Cvt<T>::to_f(up[(int64_t)(i + q) * C]); tv[q] = Cvt<T>::to_f(vp[(int64_t)(i + q) * C]); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { pu[q & 3] += tu[q]; pv[q & 3] += tv[q]; }
        }
        for (; i < N; ++i) { pu[0] += Cvt<T>::to_f(up[(int64_t)i * C]); pv[0] += Cvt<T>::to_f(vp[(int64_t)i * C]); }
        const float su = (pu[0] + pu[1]) + (pu[2] + pu[3]), sv = (pv[0] + pv[1]) + (pv[2] + pv[3]);
        const float ac = (float)N * a[c], bc = bco[c];
#pragma unroll 4
        for (int i = i0; i < i1; ++i) {
            const int64_t o = ((int64_t)b * N + i) * C + c;
            dU[o] += ac + bc * ((float)N * Cvt<T>::to_f(U[o]) + sv);
            dV[o] += ac + bc * ((float)N * Cvt<T>::to_f(V[o]) + su);
        }
    }
}

inline int grid_rows(int64_t rows, int rows_per_block) {
    constexpr int cap = 1024;                        // workgroup cap of the grid-stride row kernels; r03 sweep: 1024 -> 288 / 102 us, 2048 -> 297 / 108, 3072 -> 306 / 105, 4096 -> 315 / 127
    int64_t g = (rows + rows_per_block - 1) / rows_per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int p3_row_affine_bwd(const void* dA, const float* dS, const void* H, const float* scale, const float* shift, const float* mean, const float* w4,
                                 void* dHd, float* acc, int64_t R, int C, int N, int transpose, int dtype, void* stream) {
    return p3_row_affine_bwd2(dA, dS, H, scale, shift, mean, w4, dHd, acc, nullptr, nullptr, R, C, N, transpose, dtype, stream);
}

extern "C" int p3_row_affine_bwd2(const void* dA, const float* dS, const void* H, const float* scale, const float* shift, const float* mean,
                                  const float* w4, void* dHd, float* acc, const float* fix_a, const float* fix_b, int64_t R, int C, int N,
                                  int transpose, int dtype, void* stream) {
    P3_CHECK(H && scale && shift && mean && (dHd || acc) && R > 0, P3_EINVAL, "p3_row_affine_bwd: bad arguments");
    P3_CHECK((fix_a == nullptr) == (fix_b == nullptr), P3_EINVAL, "p3_row_affine_bwd: fix_a and fix_b go together");
    P3_CHECK((dS != nullptr) != (dA != nullptr), P3_EINVAL, "p3_row_affine_bwd: exactly one of dS (tail) / dA (matrix)");
    P3_CHECK(dS ? (C == 64 && w4) : (C == 128), P3_EUNSUP, "p3_row_affine_bwd: tail needs C = 64, matrix mode C = 128");
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_rows(R, 64)), b(256);
    const int nvals = dS ? 3 * 64 + 1 : 2 * 128;
    float* slab = acc ? p3_det_scratch((int64_t)g.x * nvals, dtype) : nullptr;
    if (dS) {
        if (dtype == P3_BF16) hipLaunchKernelGGL((row_affine_bwd_kernel<bf16_t, 64, true>), g, b, 0, s, nullptr, dS, (const bf16_t*)H, scale, shift, mean, w4, (bf16_t*)dHd, acc, R, N, transpose, fix_a, fix_b, slab);
        else hipLaunchKernelGGL((row_affine_bwd_kernel<float, 64, true>), g, b, 0, s, nullptr, dS, (const float*)H, scale, shift, mean, w4, (float*)dHd, acc, R, N, transpose, fix_a, fix_b, slab);
    } else {
        if (dtype == P3_BF16) hipLaunchKernelGGL((row_affine_bwd_kernel<bf16_t, 128, false>), g, b, 0, s, (const bf16_t*)dA, nullptr, (const bf16_t*)H, scale, shift, mean, nullptr, (bf16_t*)dHd, acc, R, N, 0, fix_a, fix_b, slab);
        else hipLaunchKernelGGL((row_affine_bwd_kernel<float, 128, false>), g, b, 0, s, (const float*)dA, nullptr, (const float*)H, scale, shift, mean, nullptr, (float*)dHd, acc, R, N, 0, fix_a, fix_b, slab);
    }
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce(slab, (int)g.x, nvals, acc, nvals, 1, s);
    return P3_OK;
}

// G = A^T [y > 0], G2 = A^T ([y > 0] H) from p3_gemm_tn_ex(P3_A_AFFINE_MASK2) -> the weight gradient of the layer behind the BatchNorm / ReLU
// (dW[k, c] += scale[c] G2[k, c] + shift[c] G[k, c]  =  A^T relu(H scale + shift)) and the two sums the BatchNorm backward of the layer's INPUT gradient
// dA = A W needs (dz = [y > 0] dA):  sum_r dz = sum_k W[k, c] G[k, c],  sum_r dz (H - mean) = sum_k W[k, c] (G2[k, c] - mean[c] G[k, c]).
// A workgroup = 16 channels x 16 row lanes (row lane q takes k = q, q + 16, ...); float64 partials folded in lane order (bit-reproducible).
__global__ __launch_bounds__(256) void bn_sums_from_g_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ W, const float* __restrict__ sc,
                                                             const float* __restrict__ sh, const float* __restrict__ mean, float* __restrict__ dW,
                                                             float* __restrict__ acc, int N, int K) {
    __shared__ double r1[16][16], r2[16][16];
    const int cl = threadIdx.x & 15, q = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < K) {
        const float s = sc[c], h = sh[c], mu = mean[c];
        for (int k = q; k < N; k += 16) {
            const float g = G[(int64_t)k * ldg + c], g2 = G[(int64_t)k * ldg + K + c], w = W[(int64_t)k * K + c];
            s1 += (double)w * (double)g;
            s2 += (double)w * ((double)g2 - (double)mu * (double)g);
            dW[(int64_t)k * K + c] += s * g2 + h * g;
        }
    }
    r1[q][cl] = s1; r2[q][cl] = s2;
    __syncthreads();
    if (q == 0 && c < K) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a1 += r1[i][cl]; a2 += r2[i][cl]; }
        acc[c] += (float)a2;            // centred scale sums
        acc[K + c] += (float)a1;        // shift sums
    }
}

extern "C" int p3_bn_sums_from_g(const float* G, int ldg, const float* W, const float* scale, const float* shift, const float* mean, float* dW, float* acc,
                                 int N, int K, void* stream) {
    P3_CHECK(G && W && scale && shift && mean && dW && acc && N > 0 && K > 0 && ldg >= 2 * K, P3_EINVAL, "p3_bn_sums_from_g: bad arguments");
    hipLaunchKernelGGL(bn_sums_from_g_kernel, dim3((K + 15) / 16), dim3(256), 0, (hipStream_t)stream, G, ldg, W, scale, shift, mean, dW, acc, N, K);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_bn_bwd_coeffs(const float* dscale, const float* dshift, const float* gamma, const float* mean, const float* rstd, float count,
                                int training, int C, float* dgamma, float* dbeta, float* a, float* b, void* stream) {
    P3_CHECK(dscale && dshift && gamma && mean && rstd && dgamma && dbeta && a && b && C > 0, P3_EINVAL, "p3_bn_bwd_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, dscale, dshift, gamma, mean, rstd, count, training, C,
                       dgamma, dbeta, a, b);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_affine_fix(void* dH, const void* H, const float* a, const float* b, int64_t R, int C, int dtype, void* stream) {
    return p3_affine_fix_ld(dH, H, C, a, b, R, C, dtype, stream);
}

extern "C" int p3_affine_fix_ld(void* dH, const void* H, int ldh, const float* a, const float* b, int64_t R, int C, int dtype, void* stream) {
    P3_CHECK(dH && H && a && b && R > 0 && C % 4 == 0 && ldh >= C && ldh % 4 == 0, P3_EINVAL, "p3_affine_fix: bad arguments");
    const int64_t n4 = R * C / 4;
    int64_t g = (n4 + 255) / 256; if (g > 8192) g = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P3_BF16) hipLaunchKernelGGL((affine_fix_kernel<bf16_t>), dim3((int)g), dim3(256), 0, s, (bf16_t*)dH, (const bf16_t*)H, a, b, n4, C, ldh);
    else hipLaunchKernelGGL((affine_fix_kernel<float>), dim3((int)g), dim3(256), 0, s, (float*)dH, (const float*)H, a, b, n4, C, ldh);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

void p3_pair_dv_reduce_launch(const float* slab, float* dV, int nblk, int B, int N, int C, hipStream_t s) {
    const int64_t per = (int64_t)N * C / 4, total = (int64_t)B * per;
    hipLaunchKernelGGL(pair_dv_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, slab, dV, nblk, per, total);
}

extern "C" int64_t p3_pair_bwd_workspace_bytes(int B, int N, int C) { return (int64_t)B * ((N + 11) / 12) * N * C * 4; }
// per dtype: the fp32 kernel takes 8 rows i per block (more, smaller slabs), the bf16 one 12
extern "C" int64_t p3_pair_bwd_workspace_bytes_dt(int B, int N, int C, int dtype) {
    (void)dtype;                                       // both forms may run with 8 rows i per block (bf16: P3_PAIR_IC16=8)
    return (int64_t)B * ((N + 7) / 8) * N * C * 4;
}

static int pair_bwd_impl(const void* dA, const void* U, const void* V, const float* scale, const float* shift, const float* mean, float* dU, float* dV, float* acc,
                         int B, int N, int C, int dtype, float* slab, void* stream);

extern "C" int p3_pair_bwd(const void* dA, const void* U, const void* V, const float* scale, const float* shift, const float* mean, float* dU, float* dV, float* acc,
                           int B, int N, int C, int dtype, void* stream) {
    return pair_bwd_impl(dA, U, V, scale, shift, mean, dU, dV, acc, B, N, C, dtype, nullptr, stream);
}

extern "C" int p3_pair_bwd_ws(const void* dA, const void* U, const void* V, const float* scale, const float* shift, const float* mean, float* dU, float* dV, float* acc,
                              int B, int N, int C, int dtype, void* workspace, void* stream) {
    return pair_bwd_impl(dA, U, V, scale, shift, mean, dU, dV, acc, B, N, C, dtype, (float*)workspace, stream);
}

static int pair_bwd_impl(const void* dA, const void* U, const void* V, const float* scale, const float* shift, const float* mean, float* dU, float* dV, float* acc,
                         int B, int N, int C, int dtype, float* slab, void* stream) {
    P3_CHECK(dA && U && V && scale && shift && mean && dU && dV && acc && B > 0, P3_EINVAL, "p3_pair_bwd: bad arguments");
    P3_CHECK(C == 256, P3_EUNSUP, "p3_pair_bwd: ScoreNet conv1 width must be 256 (model_pix2poly.py:74)");
    hipStream_t s = (hipStream_t)stream;
    const int ic = 16;     // rows i per block: dV gets N/IC atomic adds per element (same-box sweep r01: 4 -> 60.2 ms, 8 -> 58.0, 12 -> 57.7, 16 -> 57.6)
#define PB(T, IC) do { nblk = (N + IC - 1) / IC; acc_slab = p3_det_scratch((int64_t)B * nblk * 2 * C, dtype); if (!acc_slab) dslab = nullptr; \
                       hipLaunchKernelGGL((pair_bwd_kernel<T, IC>), dim3(nblk, B), dim3(256), 0, s, (const T*)dA, (const T*)U, (const T*)V, scale, shift, mean, dU, dV, acc, N, C, dslab, acc_slab); } while (0)
    int nblk = 0;
    float* acc_slab = nullptr;
    float* dslab = slab;          // the plain forms use the dV slab only together with the deterministic (dscale, dshift) partials
    if (dtype == P3_BF16) {
        // IC = 12 rows i per block (6 per half-wave): the register budget of two waves per SIMD without spills (IC = 16 spills 270 B / lane)
        // 8 rows i per block (4 per half-wave) with the next step's rows prefetched: 238 VGPRs, no spills (the 12-row form needs 256 + 35
        // spilled and cannot hold a second register set).  rocprofv3 A/B (r03): 592 -> 348 us per launch, the dV slab sum 44 -> 68 us.
        // Without a dV slab (caller's choice) the 12-row form runs.
        const int icb = slab ? 8 : 12;
        nblk = (N + icb - 1) / icb;
        acc_slab = p3_det_scratch((int64_t)B * nblk * 2 * C, dtype);
        if (icb == 8) hipLaunchKernelGGL((pair_bwd_kernel16<8, true>), dim3(nblk, B), dim3(256), 0, s, (const bf16_t*)dA, (const bf16_t*)U, (const bf16_t*)V, scale, shift,
                                         mean, dU, dV, acc, N, C, slab, acc_slab);
        else hipLaunchKernelGGL((pair_bwd_kernel16<12, false>), dim3(nblk, B), dim3(256), 0, s, (const bf16_t*)dA, (const bf16_t*)U, (const bf16_t*)V, scale, shift,
                           mean, dU, dV, acc, N, C, slab, acc_slab);
        P3_LAUNCH_CHECK();
        if (slab) {
            const int64_t per = (int64_t)N * C / 4, total = (int64_t)B * per;
            hipLaunchKernelGGL(pair_dv_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, slab, dV, nblk, per, total);
            P3_LAUNCH_CHECK();
        }
        if (acc_slab) return p3_det_reduce(acc_slab, B * nblk, 2 * (int64_t)C, acc, 2 * C, 1, s);
        return P3_OK;
    }
    if (dtype == P3_BF16) { dslab = nullptr; if (ic == 16) PB(bf16_t, 16); else if (ic == 12) PB(bf16_t, 12); else if (ic == 4) PB(bf16_t, 4); else PB(bf16_t, 8); }
    else PB(float, 8);
#undef PB
    P3_LAUNCH_CHECK();
    if (dslab) {
        const int64_t per = (int64_t)N * C / 4, total = (int64_t)B * per;
        hipLaunchKernelGGL(pair_dv_reduce_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, dslab, dV, nblk, per, total);
        P3_LAUNCH_CHECK();
    }
    if (acc_slab) return p3_det_reduce(acc_slab, B * nblk, 2 * (int64_t)C, acc, 2 * C, 1, s);
    return P3_OK;
}

extern "C" int p3_pair_stats_bwd(const void* U, const void* V, const float* a, const float* b, float* dU, float* dV, int B, int N, int C, int dtype,
                                 void* stream) {
    P3_CHECK(U && V && a && b && dU && dV && B > 0, P3_EINVAL, "p3_pair_stats_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P3_BF16) hipLaunchKernelGGL((pair_stats_bwd_kernel<bf16_t>), dim3(B, 8), dim3(256), 0, s, (const bf16_t*)U, (const bf16_t*)V, a, b, dU, dV, N, C);
    else hipLaunchKernelGGL((pair_stats_bwd_kernel<float>), dim3(B, 8), dim3(256), 0, s, (const float*)U, (const float*)V, a, b, dU, dV, N, C);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
