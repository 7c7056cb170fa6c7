// p3hip training-step kernels: Sinkhorn backward (single launch), CE / BCE losses (forward + backward), fused AdamW.
#include <stdlib.h>

#include "p3_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ Sinkhorn backward
// Reverse-mode through all `iters` log-Sinkhorn iterations + slice + row softmax (what autograd does over ~1200 launches in
// the reference, model_pix2poly.py:35-66,261-264).  One 1024-thread workgroup per sample: Z stays in LDS, the gradient dZ
// lives in registers with a fixed (row = wave + 16k, col = lane + 64c) ownership; the LSE terms are recovered from the saved
// dual iterates (u_t, v_t):  softmax_i(Z + u_t)[i,j] = exp(Z_ij + u_t[i] + v_t[j] - log_nu[j]), etc. - no re-reduction.
constexpr int SK_MAXK = 13, SK_MAXC = 4;   // rows <= 16*13 = 208, cols <= 256  (reference: 193 x 193)

__global__ __launch_bounds__(1024) void sinkhorn_bwd_kernel(const float* __restrict__ scores, const float* __restrict__ alpha_p, int m, int n,
                                                            int iters, const float* __restrict__ perm, const float* __restrict__ uv_hist,
                                                            const float* __restrict__ dperm, float* __restrict__ dscores,
                                                            float* __restrict__ dalpha, const int* __restrict__ tile_flags) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M1 = m + 1, N1 = n + 1;
    float* Z = sm;                   // [M1][N1]
    float* ut = sm + M1 * N1;        // [M1] u_t
    float* vt = ut + M1;             // [N1] v_t
    float* vp = vt + N1;             // [N1] v_{t-1}
    float* du = vp + N1;             // [M1]
    float* dv = du + M1;             // [N1]
    float* dvn = dv + N1;            // [N1] next dv (accumulated with LDS atomics)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tile_flags[b] == 0) return;       // this tile was done by sinkhorn_bwd_fast_kernel (linear domain)
    const float alpha = alpha_p[0];
    for (int i = tid; i < M1 * N1; i += 1024) {
        const int r = i / N1, c = i - r * N1;
        Z[i] = (r < m && c < n) ? scores[((int64_t)b * m + r) * n + c] : alpha;
    }
    for (int i = tid; i < N1; i += 1024) { dv[i] = 0.f; dvn[i] = 0.f; }
    for (int i = tid; i < M1; i += 1024) du[i] = 0.f;
    const float norm = -logf((float)(m + n));
    const float a_last = logf((float)n) + norm, b_last = logf((float)m) + norm;
    float dZ[SK_MAXK][SK_MAXC];
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k)
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) dZ[k][c] = 0.f;
    __syncthreads();
    // ---- softmax backward: G = perm * (dperm - rowdot); dZ += G; dv[j] = sum_i G ----
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) {
        const int i = w + 16 * k;
        float pv[SK_MAXC], gv[SK_MAXC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            const bool ok = i < m && j < n;
            pv[c] = ok ? perm[((int64_t)b * m + i) * n + j] : 0.f;
            gv[c] = ok ? dperm[((int64_t)b * m + i) * n + j] : 0.f;
            dot += pv[c] * gv[c];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            const float G = pv[c] * (gv[c] - dot);
            dZ[k][c] += G;
            if (i < m && j < n && G != 0.f) atomicAdd(&dv[j], G);
        }
    }
    __syncthreads();
    for (int t = iters; t >= 1; --t) {
        const float* h = uv_hist + ((int64_t)b * iters + (t - 1)) * (M1 + N1);
        const float* hp = t > 1 ? h - (M1 + N1) : nullptr;
        for (int i = tid; i < M1; i += 1024) ut[i] = h[i];
        for (int i = tid; i < N1; i += 1024) { vt[i] = h[M1 + i]; vp[i] = hp ? hp[M1 + i] : 0.f; }
        __syncthreads();
        // pass A: v_t = log_nu - LSE_i(Z + u_t):  q = softmax_i * dv[j];  dZ -= q;  du[i] = -sum_j q
#pragma unroll
        for (int k = 0; k < SK_MAXK; ++k) {
            const int i = w + 16 * k;
            float acc = 0.f;
            if (i < M1) {
                const float ui = ut[i];
#pragma unroll
                for (int c = 0; c < SK_MAXC; ++c) {
                    const int j = lane + 64 * c;
                    if (j < N1) {
                        const float q = __expf(Z[i * N1 + j] + ui + vt[j] - (j < n ? norm : b_last)) * dv[j];
                        dZ[k][c] -= q; acc += q;
                    }
                }
            }
            acc = wave_sum(acc);
            if (lane == 0 && i < M1) du[i] = -acc;
        }
        __syncthreads();
        // pass B: u_t = log_mu - LSE_j(Z + v_{t-1}):  r = softmax_j * du[i];  dZ -= r;  dv_{t-1}[j] = -sum_i r
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            float acc = 0.f;
            if (j < N1) {
                const float vj = vp[j];
#pragma unroll
                for (int k = 0; k < SK_MAXK; ++k) {
                    const int i = w + 16 * k;
                    if (i < M1) {
                        const float r = __expf(Z[i * N1 + j] + vj + ut[i] - (i < m ? norm : a_last)) * du[i];
                        dZ[k][c] -= r; acc += r;
                    }
                }
                atomicAdd(&dvn[j], -acc);
            }
        }
        __syncthreads();
        for (int i = tid; i < N1; i += 1024) { dv[i] = dvn[i]; dvn[i] = 0.f; }
        __syncthreads();
    }
    float da = 0.f;
#pragma unroll
    for (int k = 0; k < SK_MAXK; ++k) {
        const int i = w + 16 * k;
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            if (i < m && j < n) dscores[((int64_t)b * m + i) * n + j] = dZ[k][c];
            else if (i < M1 && j < N1) da += dZ[k][c];
        }
    }
    da = wave_sum(da);
    if (lane == 0 && da != 0.f) atomicAdd(dalpha, da);
}

// ---- linear-domain backward (tiles whose row spread allows E = exp(Z - rowmax), see sinkhorn.hip) -------------------------------
// With E fixed, every exp(Z + u + v - c) of the reverse sweep factorises: q_ij = E_ij A_i B_j, r_ij = E_ij D_i C_j.  The dual
// gradients therefore need only two matrix-vector products per iteration (du = -A (E B), dv' = -C (E^T D)) - the same loop shape
// as the forward kernel - and the 37 k-element gradient dZ = G - E (sum_t A^t B^t^T + D^t C^t^T) is accumulated ONCE at the end
// from the per-iteration vectors (kept in a global scratch slab, 309 KB per tile, L2 resident) instead of being read-modify-written
// in registers twice per iteration (r01: 1.49 ms for that form, 1.72 ms for the log-domain one, 64 x 192 x 192 x 100).
__global__ __launch_bounds__(1024) void sinkhorn_bwd_fast_kernel(const float* __restrict__ scores, const float* __restrict__ alpha_p, int m, int n,
                                                                 int iters, const float* __restrict__ perm, const float* __restrict__ uv_hist,
                                                                 const float* __restrict__ dperm, float* __restrict__ dscores,
                                                                 float* __restrict__ dalpha, int* __restrict__ tile_flags, float* __restrict__ vecs,
                                                                 int forced) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M1 = m + 1, N1 = n + 1, VS = 2 * (M1 + N1);
    float* E = sm;                   // [M1][N1]
    float* rmax = sm + M1 * N1;      // [M1]
    float* Ai = rmax + M1;           // [M1] exp(u_t + rmax)
    float* Di = Ai + M1;             // [M1] Ai / mu_i * du_i
    float* Bj = Di + M1;             // [N1] exp(v_t - log_nu) * dv
    float* Cj = Bj + N1;             // [N1] exp(v_{t-1})
    float* dv = Cj + N1;             // [N1]
    float* pm = dv + N1;             // [1024] partials
    float* ps = pm + 1024;           // [1024] partials
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int idx = tid & 255, part = tid >> 8;
    const int chj = (N1 + 3) / 4, chi = (M1 + 3) / 4;
    // forced (kernel argument; P3_SINKHORN_LOG=1): leave every tile to the log-domain kernel.  tile_flags is WRITE-only here: r03's
    // rocprofv3 runs of the captured step showed the flags non-zero on entry (a memset node ahead of this kernel had cleared them before:
    // every tile then took the 1.2 ms log-domain path under the profiler only) - nothing is read that this launch did not write
    const float alpha = alpha_p[0];
    for (int i = tid; i < M1 * N1; i += 1024) {
        const int r = i / N1, c = i - r * N1;
        E[i] = (r < m && c < n) ? scores[((int64_t)b * m + r) * n + c] : alpha;
    }
    for (int i = tid; i < N1; i += 1024) dv[i] = 0.f;
    const float norm = -logf((float)(m + n));
    const float b_last = logf((float)m) + norm;
    const float inv_mu = (float)(m + n), inv_mu_last = (float)(m + n) / (float)n;      // 1 / exp(log_mu)
    __syncthreads();
    {   // row maxima / spread test (same rule as the forward kernel)
        float mx = -INFINITY, mn = INFINITY;
        if (idx < M1) {
            const int c0 = part * chj, c1 = min(N1, c0 + chj);
            const float* zr = E + idx * N1;
            for (int c = c0; c < c1; ++c) { const float x = zr[c]; mx = fmaxf(mx, x); mn = fminf(mn, x); }
        }
        pm[part * 256 + idx] = mx; ps[part * 256 + idx] = mn;
    }
    __syncthreads();
    int wide = forced;
    if (tid < M1) {
        const float mm = fmaxf(fmaxf(pm[tid], pm[256 + tid]), fmaxf(pm[512 + tid], pm[768 + tid]));
        const float nn = fminf(fminf(ps[tid], ps[256 + tid]), fminf(ps[512 + tid], ps[768 + tid]));
        rmax[tid] = mm;
        wide |= !(mm - nn < 60.f);
    }
    const bool wide_tile = __syncthreads_or(wide);
    if (tid == 0) tile_flags[b] = wide_tile ? 1 : 0;
    if (wide_tile) return;
    for (int i = tid; i < M1 * N1; i += 1024) { const int r = i / N1; E[i] = __expf(E[i] - rmax[r]); }
    // ---- softmax backward: G = perm * (dperm - rowdot); dv[j] = sum_i G_ij (G itself is rebuilt in the last phase)
    for (int i = w; i < m; i += 16) {
        float pv[SK_MAXC], gv[SK_MAXC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            const bool ok = j < n;
            pv[c] = ok ? perm[((int64_t)b * m + i) * n + j] : 0.f;
            gv[c] = ok ? dperm[((int64_t)b * m + i) * n + j] : 0.f;
            dot += pv[c] * gv[c];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            const float G = pv[c] * (gv[c] - dot);
            if (j < n && G != 0.f) atomicAdd(&dv[j], G);
        }
    }
    __syncthreads();
    float* vt_all = vecs + (int64_t)b * iters * VS;
    for (int t = iters; t >= 1; --t) {
        const float* h = uv_hist + ((int64_t)b * iters + (t - 1)) * (M1 + N1);
        float* vs = vt_all + (int64_t)(t - 1) * VS;            // [A (M1) | D (M1) | B (N1) | C (N1)] of this iteration
        if (tid < M1) { const float a = __expf(h[tid] + rmax[tid]); Ai[tid] = a; vs[tid] = a; }
        if (tid < N1) {
            const float bj = __expf(h[M1 + tid] - (tid < n ? norm : b_last)) * dv[tid];
            const float cj = t > 1 ? __expf(h[M1 + tid - (M1 + N1)]) : 1.f;                        // v_0 = 0
            Bj[tid] = bj; Cj[tid] = cj; vs[2 * M1 + tid] = bj; vs[2 * M1 + N1 + tid] = cj;
        }
        __syncthreads();
        {   // s_i = sum_j E_ij B_j
            float sacc = 0.f;
            if (idx < M1) {
                const int c0 = part * chj, c1 = min(N1, c0 + chj);
                const float* er = E + idx * N1;
                for (int c = c0; c < c1; ++c) sacc = fmaf(er[c], Bj[c], sacc);
            }
            ps[part * 256 + idx] = sacc;
        }
        __syncthreads();
        if (tid < M1) {
            const float du = -Ai[tid] * ((ps[tid] + ps[256 + tid]) + (ps[512 + tid] + ps[768 + tid]));
            const float d = Ai[tid] * (tid < m ? inv_mu : inv_mu_last) * du;
            Di[tid] = d; vs[M1 + tid] = d;
        }
        __syncthreads();
        {   // w_j = sum_i E_ij D_i
            float sacc = 0.f;
            if (idx < N1) {
                const int r0 = part * chi, r1 = min(M1, r0 + chi);
                for (int r = r0; r < r1; ++r) sacc = fmaf(E[r * N1 + idx], Di[r], sacc);
            }
            pm[part * 256 + idx] = sacc;
        }
        __syncthreads();
        if (tid < N1) dv[tid] = -Cj[tid] * ((pm[tid] + pm[256 + tid]) + (pm[512 + tid] + pm[768 + tid]));
        // no barrier needed here: dv[tid] is read by the same thread next iteration; Ai/Bj/Cj/Di writes of the next iteration come
        // after this iteration's last reads of them (before the barrier above) except Cj, read just now by its own writer thread
    }
    __threadfence_block();
    __syncthreads();
    // ---- dZ = G - E * sum_t (A^t_i B^t_j + D^t_i C^t_j); rows w, w + 16, ..., columns lane, lane + 64, ...
    float acc[13][SK_MAXC];
#pragma unroll
    for (int k = 0; k < 13; ++k)
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) acc[k][c] = 0.f;
    for (int t = 0; t < iters; ++t) {
        const float* vs = vt_all + (int64_t)t * VS;
        float bb[SK_MAXC], cc[SK_MAXC];
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            bb[c] = j < N1 ? vs[2 * M1 + j] : 0.f;
            cc[c] = j < N1 ? vs[2 * M1 + N1 + j] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 13; ++k) {
            const int i = w + 16 * k;
            const float a = i < M1 ? vs[i] : 0.f, d = i < M1 ? vs[M1 + i] : 0.f;
#pragma unroll
            for (int c = 0; c < SK_MAXC; ++c) acc[k][c] = fmaf(a, bb[c], fmaf(d, cc[c], acc[k][c]));
        }
    }
    float da = 0.f;
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const int i = w + 16 * k;
        float pv[SK_MAXC], gv[SK_MAXC];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            const bool ok = i < m && j < n;
            pv[c] = ok ? perm[((int64_t)b * m + i) * n + j] : 0.f;
            gv[c] = ok ? dperm[((int64_t)b * m + i) * n + j] : 0.f;
            dot += pv[c] * gv[c];
        }
        dot = wave_sum(dot);
#pragma unroll
        for (int c = 0; c < SK_MAXC; ++c) {
            const int j = lane + 64 * c;
            if (i < M1 && j < N1) {
                const float dz = pv[c] * (gv[c] - dot) - E[i * N1 + j] * acc[k][c];
                if (i < m && j < n) dscores[((int64_t)b * m + i) * n + j] = dz;
                else da += dz;
            }
        }
    }
    da = wave_sum(da);
    if (lane == 0 && da != 0.f) atomicAdd(dalpha, da);
}

// ------------------------------------------------------------------------------------------------ losses
// CrossEntropyLoss(ignore_index) over rows: acc[0] += sum(lse - logit[target]), acc[1] += #valid rows; row_lse saved
// grid-stride over rows (one wave per row), ONE pair of atomics per block: 24640 same-address atomics from per-row lanes cost 0.3 ms
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ tgt, int R, int V,
                                                     int ignore, float* __restrict__ row_lse, float* __restrict__ acc) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float loss = 0.f, cnt = 0.f;
    for (int row = blockIdx.x * 4 + wv; row < R; row += gridDim.x * 4) {
        const float* x = logits + (int64_t)row * ld;
        float mx = -INFINITY;
        for (int c = lane; c < V; c += 64) mx = fmaxf(mx, x[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < V; c += 64) s += expf(x[c] - mx);
        s = wave_sum(s);
        const float lse = mx + logf(s);
        if (lane == 0) {
            row_lse[row] = lse;
            const int64_t t = tgt[row];
            if (t != ignore) { loss += lse - x[t]; cnt += 1.f; }
        }
    }
    __shared__ float red[2][4];
    if (lane == 0) { red[0][wv] = loss; red[1][wv] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(acc + 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
}

template <typename TO>
__global__ void ce_bwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ tgt, int R, int V, int ignore,
                              const float* __restrict__ row_lse, const float* __restrict__ acc, const float* __restrict__ gscale,
                              TO* __restrict__ dlogits, int ld_out, int Vpad) {
    const int64_t total = (int64_t)R * Vpad;
    const float g = gscale[0] / fmaxf(acc[1], 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Vpad);
        const int64_t r = i / Vpad;
        float d = 0.f;
        const int64_t t = tgt[r];
        if (c < V && t != ignore) d = g * (expf(logits[r * ld + c] - row_lse[r]) - (c == t ? 1.f : 0.f));
        dlogits[r * ld_out + c] = Cvt<TO>::from_f(d);
    }
}

// BCELoss (mean): acc[0] += sum -(y*max(log p,-100) + (1-y)*max(log(1-p),-100))
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n, float* __restrict__ acc) {
    // four elements per thread in flight; ONE atomic per workgroup (r03: one per wave from 1024 workgroups was a 4096-deep same-address chain
    // - 58 us for 19 MB of input)
    float s = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float pv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pv[u] = p[i + u * stride]; yv[u] = y[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) s -= yv[u] * fmaxf(logf(pv[u]), -100.f) + (1.f - yv[u]) * fmaxf(logf(1.f - pv[u]), -100.f);
    }
    for (; i < n; i += stride) {
        const float pi = p[i], yi = y[i];
        s -= yi * fmaxf(logf(pi), -100.f) + (1.f - yi) * fmaxf(logf(1.f - pi), -100.f);
    }
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, (red[0] + red[1]) + (red[2] + red[3]));
}

__global__ void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n, const float* __restrict__ gscale,
                               float* __restrict__ dp) {
    const float g = gscale[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        dp[i] = g * (pi - y[i]) / fmaxf((1.f - pi) * pi, 1e-12f);
    }
}

// ------------------------------------------------------------------------------------------------ AdamW (torch semantics)
// hyper = {lr, 1 - beta1^t, 1 - beta2^t} in device memory so that one captured graph serves every step
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                             const float* __restrict__ hyper, float beta1, float beta2, float eps, float wd, float grad_scale,
                             bf16_t* __restrict__ shadow) {
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
    const float step = lr / bc1, rbc2 = rsqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) * rbc2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (shadow) shadow[i] = f2bf(pi);
    }
}

// Step counter and schedule live on the device: one thread advances the counter and writes {lr, 1 - beta1^t, 1 - beta2^t} for the
// update kernel that follows it in the stream, so a captured graph (or a host that runs many steps ahead of the GPU) can never pair
// step t's update with another step's bias corrections.  kind 0: constant lr; kind 1: linear warm-up then linear decay to zero
// (transformers.get_linear_schedule_with_warmup, the reference's scheduler: train/trainer_pix2poly.py:62-77).
__global__ void adamw_schedule_kernel(long long* __restrict__ step, float* __restrict__ hyper, float base_lr, int kind, int warmup, int total,
                                      float beta1, float beta2) {
    const long long s = step[0];
    double lam = 1.0;
    if (kind == 1) {
        if (s < warmup) lam = (double)s / (double)(warmup > 1 ? warmup : 1);
        else { const int den = total - warmup; lam = (double)(total - s) / (double)(den > 1 ? den : 1); if (lam < 0.0) lam = 0.0; }
    }
    const double t = (double)(s + 1);
    hyper[0] = (float)((double)base_lr * lam);
    hyper[1] = (float)(1.0 - pow((double)beta1, t));
    hyper[2] = (float)(1.0 - pow((double)beta2, t));
    step[0] = s + 1;
}

// same, with (base_lr, kind, warm-up steps, total steps) read from a 4-float device buffer: a captured step graph then follows a schedule or
// learning rate the host changes AFTER the capture (the scalar-argument form bakes them into the graph)
__global__ void adamw_schedule_dev_kernel(long long* __restrict__ step, float* __restrict__ hyper, const float* __restrict__ sched, float beta1, float beta2) {
    const long long s = step[0];
    const float base_lr = sched[0];
    const int kind = (int)sched[1], warmup = (int)sched[2], total = (int)sched[3];
    double lam = 1.0;
    if (kind == 1) {
        if (s < warmup) lam = (double)s / (double)(warmup > 1 ? warmup : 1);
        else { const int den = total - warmup; lam = (double)(total - s) / (double)(den > 1 ? den : 1); if (lam < 0.0) lam = 0.0; }
    }
    const double t = (double)(s + 1);
    hyper[0] = (float)((double)base_lr * lam);
    hyper[1] = (float)(1.0 - pow((double)beta1, t));
    hyper[2] = (float)(1.0 - pow((double)beta2, t));
    step[0] = s + 1;
}

inline int grid_for(int64_t work) {
    int64_t g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" int p3_adamw_schedule_dev(long long* step, float* hyper, const float* sched, float beta1, float beta2, void* stream) {
    P3_CHECK(step && hyper && sched, P3_EINVAL, "p3_adamw_schedule_dev: bad arguments");
    hipLaunchKernelGGL(adamw_schedule_dev_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, hyper, sched, beta1, beta2);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int64_t p3_sinkhorn_bwd_workspace_bytes(int B, int m, int n, int iters) {
    return 256 + ((int64_t)B * 4 + 255) / 256 * 256 + (int64_t)B * iters * 2 * (m + n + 2) * 4;
}

extern "C" int p3_sinkhorn_bwd(const float* scores, const float* alpha, int B, int m, int n, int iters, const float* perm,
                               const float* uv_hist, const float* dperm, float* dscores, float* dalpha, void* workspace, void* stream) {
    P3_CHECK(scores && alpha && perm && uv_hist && dperm && dscores && dalpha && workspace && B > 0, P3_EINVAL, "p3_sinkhorn_bwd: bad arguments");
    P3_CHECK(m + 1 <= 16 * SK_MAXK && n + 1 <= 64 * SK_MAXC, P3_EUNSUP, "p3_sinkhorn_bwd: m <= 207, n <= 255");
    const size_t lds = ((size_t)(m + 1) * (n + 1) + 2 * (size_t)(m + 1) + 4 * (size_t)(n + 1)) * sizeof(float);
    const size_t lds_fast = ((size_t)(m + 1) * (n + 1) + 3 * (size_t)(m + 1) + 3 * (size_t)(n + 1) + 2048) * sizeof(float);
    P3_CHECK(lds <= 160 * 1024 && lds_fast <= 160 * 1024 - 512, P3_EUNSUP, "p3_sinkhorn_bwd: does not fit the 160 KB LDS");
    static int force_log = -1;                        // P3_SINKHORN_LOG=1: log-domain loop only (A/B, tests of the fallback)
    if (force_log < 0) { const char* e = getenv("P3_SINKHORN_LOG"); force_log = (e && e[0] == '1') ? 1 : 0; }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)sinkhorn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sinkhorn_bwd_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    int* tile_flags = reinterpret_cast<int*>(workspace);
    float* vecs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((size_t)B * 4 + 255) / 256 * 256);
    // two launches: the linear-domain kernel takes every tile whose row spread allows it and flags the others for the log-domain
    // kernel, which returns at once for the tiles already done (one kernel holding both loops spilled 40 more registers)
    hipLaunchKernelGGL(sinkhorn_bwd_fast_kernel, dim3(B), dim3(1024), lds_fast, s, scores, alpha, m, n, iters, perm, uv_hist, dperm, dscores, dalpha, tile_flags, vecs, force_log);
    hipLaunchKernelGGL(sinkhorn_bwd_kernel, dim3(B), dim3(1024), lds, s, scores, alpha, m, n, iters, perm, uv_hist, dperm, dscores, dalpha, tile_flags);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_ce_loss_fwd(const float* logits, int ld, const int64_t* targets, int R, int V, int ignore_index, float* row_lse, float* acc,
                              void* stream) {
    P3_CHECK(logits && targets && row_lse && acc && R > 0 && V > 0, P3_EINVAL, "p3_ce_loss_fwd: bad arguments");
    const int gr = (R + 3) / 4 < 512 ? (R + 3) / 4 : 512;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(gr), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, R, V, ignore_index, row_lse, acc);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_ce_loss_bwd(const float* logits, int ld, const int64_t* targets, int R, int V, int ignore_index, const float* row_lse,
                              const float* acc, const float* gscale, void* dlogits, int dtype_out, int ld_out, int Vpad, void* stream) {
    P3_CHECK(logits && targets && row_lse && acc && gscale && dlogits && Vpad >= V && ld_out >= Vpad, P3_EINVAL, "p3_ce_loss_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)R * Vpad;
    if (dtype_out == P3_BF16) hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, logits, ld, targets, R, V, ignore_index, row_lse, acc, gscale, (bf16_t*)dlogits, ld_out, Vpad);
    else if (dtype_out == P3_F32) hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, logits, ld, targets, R, V, ignore_index, row_lse, acc, gscale, (float*)dlogits, ld_out, Vpad);
    else { p3_set_error("p3_ce_loss_bwd: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_bce_loss_fwd(const float* p, const float* y, int64_t n, float* acc, void* stream) {
    P3_CHECK(p && y && acc && n > 0, P3_EINVAL, "p3_bce_loss_fwd: bad arguments");
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(grid_for(n) > 512 ? 512 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, y, n, acc);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_bce_loss_bwd(const float* p, const float* y, int64_t n, const float* gscale, float* dp, void* stream) {
    P3_CHECK(p && y && gscale && dp && n > 0, P3_EINVAL, "p3_bce_loss_bwd: bad arguments");
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, y, n, gscale, dp);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_adamw_schedule(long long* step, float* hyper, float base_lr, int kind, int warmup_steps, int total_steps, float beta1,
                                 float beta2, void* stream) {
    P3_CHECK(step && hyper && (kind == 0 || kind == 1), P3_EINVAL, "p3_adamw_schedule: bad arguments");
    hipLaunchKernelGGL(adamw_schedule_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, hyper, base_lr, kind, warmup_steps, total_steps, beta1,
                       beta2);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* hyper, float beta1,
                        float beta2, float eps, float weight_decay, float grad_scale, void* bf16_shadow, void* stream) {
    P3_CHECK(params && grads && exp_avg && exp_avg_sq && hyper && n > 0, P3_EINVAL, "p3_adamw: bad arguments");
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, hyper, beta1,
                       beta2, eps, weight_decay, grad_scale, (bf16_t*)bf16_shadow);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
