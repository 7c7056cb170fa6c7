// p3hip fused log-space Sinkhorn + dustbin padding + slice + row softmax in ONE launch.
// Replaces log_optimal_transport / log_sinkhorn_iterations (models/pix2poly/model_pix2poly.py:35-66) and the
// `[:, :m, :n]` slice + F.softmax(dim=-1) of EncoderDecoder.forward (:261-264): ~600 tiny launches in the reference.
// One 1024-thread workgroup per sample keeps the whole (m+1)x(n+1) coupling matrix in LDS (193x193 fp32 = 149 KB of
// the CU's 160 KB) for all iterations.  Row pass: one wave per row (lanes stride the columns, conflict free);
// column pass: one wave per column (row stride n+1 is odd for the reference's 193 -> conflict free for ds_read_b32).
// Final softmax: softmax_j(Z + u_i + v_j - norm) over j < n  ==  softmax_j(Z_ij + v_j): u and norm cancel.
#include <stdlib.h>

#include "p3_common.h"

namespace {

__device__ __forceinline__ float lse_wave(float mx_local, float (&vals)[8], int cnt) {
    const float mx = wave_max(mx_local);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) if (k < cnt) s += __expf(vals[k] - mx);   // arguments <= 0: v_exp_f32 path, rel. error ~1e-7
    s = wave_sum(s);
    return mx + __logf(s);
}

__global__ __launch_bounds__(1024) void sinkhorn_kernel(const float* __restrict__ scores, const float* __restrict__ alpha_p, int m, int n,
                                                        int iters, float* __restrict__ perm, float* __restrict__ zfull,
                                                        float* __restrict__ uv_hist, int force_log) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M1 = m + 1, N1 = n + 1;
    float* Z = sm;                 // [M1][N1]
    float* u = sm + M1 * N1;       // [M1]
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
        for (int i = tid; i < M1 * N1; i += 1024) { const int r = i / N1; Z[i] = __expf(Z[i] - rmax[r]); }
        for (int i = tid; i < N1; i += 1024) v[i] = 1.f;
        const float mu = 1.f / (float)(m + n), mu_last = (float)n / (float)(m + n), nu_last = (float)m / (float)(m + n);
        __syncthreads();
        for (int it = 0; it < iters; ++it) {
            {   // U_i = mu_i / sum_j E_ij V_j
                float sacc = 0.f;
                if (idx < M1) {
                    const int c0 = part * chj, c1 = min(N1, c0 + chj);
                    const float* zr = Z + idx * N1;
                    for (int c = c0; c < c1; ++c) sacc = fmaf(zr[c], v[c], sacc);
                }
                ps[part * 256 + idx] = sacc;
            }
            __syncthreads();
            if (tid < M1) u[tid] = (tid < m ? mu : mu_last) / ((ps[tid] + ps[256 + tid]) + (ps[512 + tid] + ps[768 + tid]));
            __syncthreads();
            {   // V_j = nu_j / sum_i E_ij U_i
                float sacc = 0.f;
                if (idx < N1) {
                    const int r0 = part * chi, r1 = min(M1, r0 + chi);
                    for (int r = r0; r < r1; ++r) sacc = fmaf(Z[r * N1 + idx], u[r], sacc);
                }
                ps[part * 256 + idx] = sacc;
            }
            __syncthreads();
            if (tid < N1) v[tid] = (tid < n ? mu : nu_last) / ((ps[tid] + ps[256 + tid]) + (ps[512 + tid] + ps[768 + tid]));
            __syncthreads();
            if (uv_hist) {                      // the backward pass reads log-domain duals
                float* h = uv_hist + ((int64_t)b * iters + it) * (M1 + N1);
                for (int i = tid; i < M1; i += 1024) h[i] = __logf(u[i]) - rmax[i];
                for (int i = tid; i < N1; i += 1024) h[M1 + i] = __logf(v[i]);
            }
        }
        if (zfull) {
            for (int i = tid; i < M1 * N1; i += 1024) {
                const int r = i / N1, c = i - r * N1;
                const float z0 = (r < m && c < n) ? scores[((int64_t)b * m + r) * n + c] : alpha;
                zfull[(int64_t)b * M1 * N1 + i] = z0 + (__logf(u[r]) - rmax[r]) + __logf(v[c]) - norm;
            }
        }
        if (perm) {                             // softmax_j(Z_ij + v_j) = E_ij V_j / sum_j E_ij V_j
            const int nc = (n + 63) / 64;
            for (int r = w; r < m; r += 16) {
                float vals[8]; float ssum = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int c = lane + 64 * k;
                    vals[k] = (k < nc && c < n) ? Z[r * N1 + c] * v[c] : 0.f;
                    ssum += vals[k];
                }
                ssum = wave_sum(ssum);
                const float inv = 1.f / ssum;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int c = lane + 64 * k;
                    if (k < nc && c < n) perm[((int64_t)b * m + r) * n + c] = vals[k] * inv;
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
    const size_t lds = ((size_t)(m + 1) * (n + 1) + 2 * (size_t)(m + 1) + (n + 1) + 2048) * sizeof(float);
    P3_CHECK(lds <= 160 * 1024 - 512, P3_EUNSUP, "p3_sinkhorn: coupling matrix does not fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)sinkhorn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    static int force_log = -1;                        // P3_SINKHORN_LOG=1: always the log-domain loop (A/B, tests of the fallback)
    if (force_log < 0) { const char* e = getenv("P3_SINKHORN_LOG"); force_log = (e && e[0] == '1') ? 1 : 0; }
    hipLaunchKernelGGL(sinkhorn_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, scores, alpha, m, n, iters, perm, z_full, uv_hist, force_log);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
