// p3hip fused log-space Sinkhorn + dustbin padding + slice + row softmax in ONE launch.
// Replaces log_optimal_transport / log_sinkhorn_iterations (models/pix2poly/model_pix2poly.py:35-66) and the
// `[:, :m, :n]` slice + F.softmax(dim=-1) of EncoderDecoder.forward (:261-264): ~600 tiny launches in the reference.
// One 1024-thread workgroup per sample keeps the whole (m+1)x(n+1) coupling matrix in LDS (193x193 fp32 = 149 KB of
// the CU's 160 KB) for all iterations.  Row pass: one wave per row (lanes stride the columns, conflict free);
// column pass: one wave per column (row stride n+1 is odd for the reference's 193 -> conflict free for ds_read_b32).
// Final softmax: softmax_j(Z + u_i + v_j - norm) over j < n  ==  softmax_j(Z_ij + v_j): u and norm cancel.
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
                                                        float* __restrict__ uv_hist) {
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
    const int ncj = (N1 + 63) / 64, nci = (M1 + 63) / 64;   // <= 8 supported (dims <= 511)
    for (int it = 0; it < iters; ++it) {
        for (int r = w; r < M1; r += 16) {
            float vals[8]; float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = lane + 64 * k;
                if (k < ncj) { vals[k] = c < N1 ? Z[r * N1 + c] + v[c] : -INFINITY; mx = fmaxf(mx, vals[k]); }
            }
            const float l = lse_wave(mx, vals, ncj);
            if (lane == 0) u[r] = (r < m ? norm : lmu_last) - l;
        }
        __syncthreads();
        for (int c = w; c < N1; c += 16) {
            float vals[8]; float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = lane + 64 * k;
                if (k < nci) { vals[k] = r < M1 ? Z[r * N1 + c] + u[r] : -INFINITY; mx = fmaxf(mx, vals[k]); }
            }
            const float l = lse_wave(mx, vals, nci);
            if (lane == 0) v[c] = (c < n ? norm : lnu_last) - l;
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
    P3_CHECK(m < 511 && n < 511, P3_EUNSUP, "p3_sinkhorn: m, n must be < 511");
    const size_t lds = ((size_t)(m + 1) * (n + 1) + (m + 1) + (n + 1)) * sizeof(float);
    P3_CHECK(lds <= 160 * 1024, P3_EUNSUP, "p3_sinkhorn: coupling matrix does not fit the 160 KB LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)sinkhorn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipLaunchKernelGGL(sinkhorn_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, scores, alpha, m, n, iters, perm, z_full, uv_hist);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
