// p3hip fused log-space Sinkhorn + dustbin padding + slice + row softmax in ONE launch.
// Replaces log_optimal_transport / log_sinkhorn_iterations (models/pix2poly/model_pix2poly.py:35-66) and the
// `[:, :m, :n]` slice + F.softmax(dim=-1) of EncoderDecoder.forward (:261-264): ~600 tiny launches in the reference.
// One 1024-thread workgroup per sample keeps the whole (m+1)x(n+1) coupling matrix in LDS (193x193 fp32 = 149 KB of
// the CU's 160 KB) for all iterations.  Row pass: one wave per row (lanes stride the columns, conflict free);
// column pass: one wave per column (row stride n+1 is odd for the reference's 193 -> conflict free for ds_read_b32).
// Final softmax: softmax_j(Z + u_i + v_j - norm) over j < n  ==  softmax_j(Z_ij + v_j): u and norm cancel.
#include <stdlib.h>

#include "p3_common.h"
#include "sinkhorn_tile.h"

namespace {

__device__ __forceinline__ float lse_wave(float mx_local, float (&vals)[8], int cnt) {
    const float mx = wave_max(mx_local);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < cnt) s += __expf(vals[k] - mx);   // arguments <= 0: v_exp_f32 path, rel. error ~1e-7
    s = wave_sum(s);
    return mx + __logf(s);
}

template <int RA, int CB>
__global__ __launch_bounds__(1024) void sinkhorn_kernel(const float* __restrict__ scores, const float* __restrict__ alpha_p, int m, int n,
                                                        int iters, float* __restrict__ perm, float* __restrict__ zfull,
                                                        float* __restrict__ uv_hist, int force_log) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M1 = m + 1, N1 = n + 1;
    float* Z = sm;                 // [M1][N1]; the linear-domain path reuses the space as its column-partial slab once E sits in registers
    float* u = sm + max(M1 * N1, sk::Slab<CB>::FLOATS);       // [M1]
    float* v = u + M1;             // [N1]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float alpha = alpha_p[0];
    for (int i = tid; i < M1 * N1; i += 1024) {
        const int r = i / N1, c = i - r * N1;
        Z[i] = (r < m && c < n) ? scores[((int64_t)b * m + r) * n + c] : alpha;
    }
    for (int i = tid; i < M1; i += 1024) u[i] = 0.f;
    for (int i = tid; i < N1; i += 1024) v[i] = 0.f;
    const float norm = -logf((float)(m + n));
    const float lmu_last = logf((float)n) + norm, lnu_last = logf((float)m) + norm;
    __syncthreads();
    // Each LSE is split over 4 thread groups (thread = (row | column, quarter)): sequential online max/sum over a quarter of
    // the line (one exp per element, no cross-lane shuffles), partials folded through LDS.  Row walk: lanes = consecutive rows,
    // stride N1 words (odd -> conflict free); column walk: lanes = consecutive columns.
    float* pm = v + N1;            // [4][256] partial maxima
    float* ps = pm + 1024;         // [4][256] partial sums
    float* rmax = ps + 1024;       // [M1] row maxima of Z (linear-domain path)
    const int idx = tid & 255, part = tid >> 8;
    const int chj = (N1 + 3) / 4, chi = (M1 + 3) / 4;

    // ---- Linear-domain fast path -------------------------------------------------------------------------------------------------
    // u_i = log_mu_i - LSE_j(Z_ij + v_j) is  U_i = mu_i / sum_j E_ij V_j  with E_ij = exp(Z_ij - rmax_i), U_i = exp(u_i + rmax_i),
    // V_j = exp(v_j): the same iterates, but an iteration is two 193 x 193 matrix-vector products (one LDS read + one FMA per element)
    // instead of two online log-sum-exps (one v_exp_f32 and ~8 VALU ops per element; r01: 1.08 ms for 100 iterations, one workgroup
    // per tile, 64 of the 256 CUs busy).  Safe while no E_ij underflows: the tile takes this path only if every row's spread
    // max_j Z_ij - min_j Z_ij stays below 60 (E >= 1e-26; the products E U V are plan entries <= 1), otherwise the log-domain loop
    // below runs unchanged.  Every column contains the dustbin row's E = 1, so no column sum can vanish.
    {
        float mx = -INFINITY, mn = INFINITY;
        if (idx < M1) {
            const int c0 = part * chj, c1 = min(N1, c0 + chj);
            const float* zr = Z + idx * N1;
            for (int c = c0; c < c1; ++c) { const float x = zr[c]; mx = fmaxf(mx, x); mn = fminf(mn, x); }
        }
        pm[part * 256 + idx] = mx; ps[part * 256 + idx] = mn;
    }
    __syncthreads();
    int wide = 0;
    if (tid < M1) {
        const float mm = fmaxf(fmaxf(pm[tid], pm[256 + tid]), fmaxf(pm[512 + tid], pm[768 + tid]));
        const float nn = fminf(fminf(ps[tid], ps[256 + tid]), fminf(ps[512 + tid], ps[768 + tid]));
        rmax[tid] = mm;
        wide = !(mm - nn < 60.f);              // also catches NaN / inf scores
    }
    const bool fast = !__syncthreads_or(wide) && !force_log && iters > 0;     // iters == 0: u = v = 0 stay log-domain quantities
    if (fast) {
        // register tiling (sinkhorn_tile.h): thread (ty, tx) owns rows ty + 64 a, columns tx + 16 b of E
        const int tx = tid & 15, ty = tid >> 4;
        float e[RA][CB], rmx[RA];
#pragma unroll
        for (int a = 0; a < RA; ++a) {
            const int i = ty + 64 * a;
            rmx[a] = i < M1 ? rmax[i] : 0.f;
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) {
                const int j = tx + 16 * bb;
                e[a][bb] = (i < M1 && j < N1) ? __expf(Z[i * N1 + j] - rmx[a]) : 0.f;
            }
        }
        __syncthreads();                    // Z is dead from here on
        float* P = sm;
        float vr[CB], ur[RA];
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) vr[bb] = (tx + 16 * bb < N1) ? 1.f : 0.f;
#pragma unroll
        for (int a = 0; a < RA; ++a) ur[a] = 0.f;
        const float mu = 1.f / (float)(m + n), mu_last = (float)n / (float)(m + n), nu_last = (float)m / (float)(m + n);
        const int rc = tid >> 2, rp = tid & 3;               // the column this thread helps to reduce, and its part of the 64 partials
        // per-row / per-column marginals with the validity folded in: an invalid row has E = 0 (sum 0), marginal 0 and pad 1 -> U = 0 * rcp(1),
        // no select in the loop.  U = mu * rcp(sum): v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division; the iteration is a
        // contraction, the tests hold Z + u + v to 1e-5 of the oracle.
        float mua[RA], pada[RA];
#pragma unroll
        for (int a = 0; a < RA; ++a) { const int i = ty + 64 * a; mua[a] = i < m ? mu : (i < M1 ? mu_last : 0.f); pada[a] = i < M1 ? 0.f : 1.f; }
        const float nuc = rc < n ? mu : (rc < N1 ? nu_last : 0.f), padc = rc < N1 ? 0.f : 1.f;
        for (int it = 0; it < iters; ++it) {
            // the duals the backward pass reads are stored in the LINEAR domain (U_i, V_j) for a tile on this path: its backward kernel
            // (sinkhorn_bwd_fast_kernel makes the same per-tile decision) needs exactly these - the log / exp round trip is gone
            float* h = uv_hist ? uv_hist + ((int64_t)b * iters + it) * (M1 + N1) : nullptr;
            // U_i = mu_i / sum_j E_ij V_j: the 16 lanes of a DPP row share the row, every lane ends with the same U_i
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                float p = 0.f;
#pragma unroll
                for (int bb = 0; bb < CB; ++bb) p = fmaf(e[a][bb], vr[bb], p);
                p = sk::row16_sum(p);
                ur[a] = mua[a] * __builtin_amdgcn_rcpf(p + pada[a]);
                if (h && tx == 0 && pada[a] == 0.f) h[ty + 64 * a] = ur[a];
            }
            // V_j = nu_j / sum_i E_ij U_i
            float q[CB];
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) {
                float t = 0.f;
#pragma unroll
                for (int a = 0; a < RA; ++a) t = fmaf(e[a][bb], ur[a], t);
                q[bb] = t;
            }
            const float cs = sk::col_reduce<CB>(P, q, tid, N1);
            const float vn = nuc * __builtin_amdgcn_rcpf(cs + padc);
            if (rp == 0 && rc < N1) {
                v[rc] = vn;
                if (h) h[M1 + rc] = vn;
            }
            sk::lds_barrier();
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) { const int j = tx + 16 * bb; vr[bb] = j < N1 ? v[j] : 0.f; }
        }
        if (zfull) {
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                const int i = ty + 64 * a;
#pragma unroll
                for (int bb = 0; bb < CB; ++bb) {
                    const int j = tx + 16 * bb;
                    if (i < M1 && j < N1) {
                        const float z0 = (i < m && j < n) ? scores[((int64_t)b * m + i) * n + j] : alpha;
                        zfull[(int64_t)b * M1 * N1 + i * N1 + j] = z0 + (__logf(ur[a]) - rmx[a]) + __logf(vr[bb]) - norm;
                    }
                }
            }
        }
        if (perm) {                             // softmax_j(Z_ij + v_j) = E_ij V_j / sum_{j < n} E_ij V_j
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                const int i = ty + 64 * a;
                float vals[CB], ssum = 0.f;
#pragma unroll
                for (int bb = 0; bb < CB; ++bb) {
                    vals[bb] = (tx + 16 * bb < n) ? e[a][bb] * vr[bb] : 0.f;
                    ssum += vals[bb];
                }
                ssum = sk::row16_sum(ssum);
                const float inv = 1.f / ssum;
#pragma unroll
                for (int bb = 0; bb < CB; ++bb) {
                    const int j = tx + 16 * bb;
                    if (i < m && j < n) perm[((int64_t)b * m + i) * n + j] = vals[bb] * inv;
                }
            }
        }
        return;
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        {   // u = log_mu - LSE_j(Z + v)
            float m = -INFINITY, sacc = 0.f;
            if (idx < M1) {
                const int c0 = part * chj, c1 = min(N1, c0 + chj);
                const float* zr = Z + idx * N1;
                for (int c = c0; c < c1; ++c) {
                    const float x = zr[c] + v[c];
                    const float e = __expf(-fabsf(x - m));
                    if (x > m) { sacc = sacc * e + 1.f; m = x; } else { sacc += e; }
                }
            }
            pm[part * 256 + idx] = m; ps[part * 256 + idx] = sacc;
        }
        __syncthreads();
        if (tid < M1) {
            const float m0 = pm[tid], m1 = pm[256 + tid], m2 = pm[512 + tid], m3 = pm[768 + tid];
            const float mm = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            const float ss = ps[tid] * __expf(m0 - mm) + ps[256 + tid] * __expf(m1 - mm) + ps[512 + tid] * __expf(m2 - mm) + ps[768 + tid] * __expf(m3 - mm);
            u[tid] = (tid < m ? norm : lmu_last) - (mm + __logf(ss));
        }
        __syncthreads();
        {   // v = log_nu - LSE_i(Z + u)
            float m = -INFINITY, sacc = 0.f;
            if (idx < N1) {
                const int r0 = part * chi, r1 = min(M1, r0 + chi);
                for (int r = r0; r < r1; ++r) {
                    const float x = Z[r * N1 + idx] + u[r];
                    const float e = __expf(-fabsf(x - m));
                    if (x > m) { sacc = sacc * e + 1.f; m = x; } else { sacc += e; }
                }
            }
            pm[part * 256 + idx] = m; ps[part * 256 + idx] = sacc;
        }
        __syncthreads();
        if (tid < N1) {
            const float m0 = pm[tid], m1 = pm[256 + tid], m2 = pm[512 + tid], m3 = pm[768 + tid];
            const float mm = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            const float ss = ps[tid] * __expf(m0 - mm) + ps[256 + tid] * __expf(m1 - mm) + ps[512 + tid] * __expf(m2 - mm) + ps[768 + tid] * __expf(m3 - mm);
            v[tid] = (tid < n ? norm : lnu_last) - (mm + __logf(ss));
        }
        __syncthreads();
        if (uv_hist) {
            float* h = uv_hist + ((int64_t)b * iters + it) * (M1 + N1);
            for (int i = tid; i < M1; i += 1024) h[i] = u[i];
            for (int i = tid; i < N1; i += 1024) h[M1 + i] = v[i];
        }
    }
    if (zfull) {   // Z + u + v - norm, full (m+1)x(n+1) (== log_optimal_transport's return value)
        for (int i = tid; i < M1 * N1; i += 1024) {
            const int r = i / N1, c = i - r * N1;
            zfull[(int64_t)b * M1 * N1 + i] = Z[i] + u[r] + v[c] - norm;
        }
    }
    if (perm) {
        const int nc = (n + 63) / 64;
        for (int r = w; r < m; r += 16) {
            float vals[8]; float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = lane + 64 * k;
                if (k < nc) { vals[k] = c < n ? Z[r * N1 + c] + v[c] : -INFINITY; mx = fmaxf(mx, vals[k]); }
            }
            mx = wave_max(mx);
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k < nc) { vals[k] = expf(vals[k] - mx); s += vals[k]; }
            s = wave_sum(s);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = lane + 64 * k;
                if (k < nc && c < n) perm[((int64_t)b * m + r) * n + c] = vals[k] / s;
            }
        }
    }
}

}  // namespace

extern "C" int p3_sinkhorn(const float* scores, const float* alpha, int B, int m, int n, int iters, float* perm, float* z_full,
                           float* uv_hist, void* stream) {
    P3_CHECK(scores && alpha && B > 0 && m > 0 && n > 0 && iters >= 0, P3_EINVAL, "p3_sinkhorn: bad arguments");
    P3_CHECK(m < 255 && n < 255, P3_EUNSUP, "p3_sinkhorn: m, n must be < 255");
    static int force_log = -1;                        // P3_SINKHORN_LOG=1: always the log-domain loop (A/B, tests of the fallback)
    if (force_log < 0) { const char* e = getenv("P3_SINKHORN_LOG"); force_log = (e && e[0] == '1') ? 1 : 0; }
    hipStream_t s = (hipStream_t)stream;
    // register tile of the linear-domain path: RA rows (64 apart) x CB columns (16 apart) per thread; the smallest instantiation that covers
    // (m + 1) x (n + 1) - 4 x 13 for the reference's 193 x 193 (config/model/pix2poly.yaml:13), 1 x 4 for debug_rgd.yaml's 61 x 61
#define P3_SK_LAUNCH(RA, CB)                                                                                                              \
    do {                                                                                                                                  \
        const size_t zf = (size_t)(m + 1) * (n + 1) > (size_t)sk::Slab<CB>::FLOATS ? (size_t)(m + 1) * (n + 1) : (size_t)sk::Slab<CB>::FLOATS; \
        const size_t lds = (zf + 2 * (size_t)(m + 1) + (n + 1) + 2048) * sizeof(float);                                                   \
        P3_CHECK(lds <= 160 * 1024 - 512, P3_EUNSUP, "p3_sinkhorn: coupling matrix does not fit the 160 KB LDS");                         \
        static bool attr_set = false;                                                                                                     \
        if (!attr_set) {                                                                                                                  \
            hipError_t e = hipFuncSetAttribute((const void*)sinkhorn_kernel<RA, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); \
            if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }                                                   \
            attr_set = true;                                                                                                              \
        }                                                                                                                                 \
        hipLaunchKernelGGL((sinkhorn_kernel<RA, CB>), dim3(B), dim3(1024), lds, s, scores, alpha, m, n, iters, perm, z_full, uv_hist, force_log); \
    } while (0)
    const int M1 = m + 1, N1 = n + 1;
    if (M1 <= 64 && N1 <= 32) P3_SK_LAUNCH(1, 2);
    else if (M1 <= 64 && N1 <= 64) P3_SK_LAUNCH(1, 4);
    else if (M1 <= 128 && N1 <= 128) P3_SK_LAUNCH(2, 8);
    else if (M1 <= 256 && N1 <= 208) P3_SK_LAUNCH(4, 13);
    else P3_SK_LAUNCH(4, 16);
#undef P3_SK_LAUNCH
    P3_LAUNCH_CHECK();
    return P3_OK;
}
