// p3hip training-step kernels: Sinkhorn backward (single launch), CE / BCE losses (forward + backward), fused AdamW.
#include <stdlib.h>

#include "p3_common.h"
#include "sinkhorn_tile.h"

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
                                                            float* __restrict__ dalpha, const int* __restrict__ tile_flags,
                                                            float* __restrict__ da_slab = nullptr /* deterministic mode: this tile's dalpha share, slot blockIdx.x */) {
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
    if (tile_flags[b] == 0) {             // this tile was done by sinkhorn_bwd_fast_kernel (linear domain)
        if (da_slab && tid == 0) da_slab[b] = 0.f;
        return;
    }
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
    if (da_slab) {                        // fixed order: the 16 waves fold through LDS, one plain store per tile; p3_det_reduce adds the tiles in tile order
        __syncthreads();
        if (lane == 0) sm[w] = da;
        __syncthreads();
        if (tid == 0) { float t = 0.f; for (int k = 0; k < 16; ++k) t += sm[k]; da_slab[b] = t; }
    } else if (lane == 0 && da != 0.f) atomicAdd(dalpha, da);
}

// ---- linear-domain backward (tiles whose row spread allows E = exp(Z - rowmax), see sinkhorn.hip) -------------------------------
// With E fixed, every exp(Z + u + v - c) of the reverse sweep factorises: q_ij = E_ij A_i B_j, r_ij = E_ij D_i C_j.  The dual
// gradients therefore need only two matrix-vector products per iteration (du = -A (E B), dv' = -C (E^T D)) - the same loop shape
// as the forward kernel - and the 37 k-element gradient dZ = G - E (sum_t A^t B^t^T + D^t C^t^T) is accumulated ONCE at the end
// from the per-iteration vectors (kept in a global scratch slab, 309 KB per tile, L2 resident) instead of being read-modify-written
// in registers twice per iteration (r01: 1.49 ms for that form, 1.72 ms for the log-domain one, 64 x 192 x 192 x 100).
// r04: E lives in REGISTERS in the 64 x 16 thread tiling of sinkhorn_tile.h (row products by DPP, column products through a small LDS
// slab; r03 read E and the vector from LDS for every FMA and was bound by the ds_read_b32 rate: 730 us).
template <int RA, int CB>
__global__ __launch_bounds__(1024) void sinkhorn_bwd_fast_kernel(const float* __restrict__ scores, const float* __restrict__ alpha_p, int m, int n,
                                                                 int iters, const float* __restrict__ perm, const float* __restrict__ uv_hist,
                                                                 const float* __restrict__ dperm, float* __restrict__ dscores,
                                                                 float* __restrict__ dalpha, int* __restrict__ tile_flags, float* __restrict__ vecs,
                                                                 float* __restrict__ rmax_out, int forced) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int M1 = m + 1, N1 = n + 1, VS = 2 * (M1 + N1);
    float* E = sm;                   // [M1][N1] while the tile is loaded and tested; then the column-partial slab
    float* rmax = sm + max(M1 * N1, sk::Slab<CB>::FLOATS);      // [M1]
    float* Bl = rmax + M1;           // [N1] B_j of the current iteration
    float* pm = Bl + N1;             // [1024] partials
    float* ps = pm + 1024;           // [1024] partials
    const int b = blockIdx.x, tid = threadIdx.x;
    const int idx = tid & 255, part = tid >> 8;
    const int chj = (N1 + 3) / 4;
    // forced (kernel argument; P3_SINKHORN_LOG=1): leave every tile to the log-domain kernel.  tile_flags is WRITE-only here: r03's
    // rocprofv3 runs of the captured step showed the flags non-zero on entry (a memset node ahead of this kernel had cleared them before:
    // every tile then took the 1.2 ms log-domain path under the profiler only) - nothing is read that this launch did not write
    const float alpha = alpha_p[0];
    for (int i = tid; i < M1 * N1; i += 1024) {
        const int r = i / N1, c = i - r * N1;
        E[i] = (r < m && c < n) ? scores[((int64_t)b * m + r) * n + c] : alpha;
    }
    const float norm = -logf((float)(m + n));
    const float b_last = logf((float)m) + norm;
    const float inv_mu = (float)(m + n), inv_mu_last = (float)(m + n) / (float)n;      // 1 / exp(log_mu)
    const float inv_nu_last = (float)(m + n) / (float)m;
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
    if (tid < M1) rmax_out[(int64_t)b * M1 + tid] = rmax[tid];      // sinkhorn_bwd_dz_kernel rebuilds E from the scores
    const int tx = tid & 15, ty = tid >> 4;
    const int rc = tid >> 2, rp = tid & 3;                   // reducer role: column rc, part rp of its 64 partials
    // tile bases are wave-uniform (SGPR pair) and element offsets 32-bit: one address VGPR per load instead of a 64-bit pair (the 104
    // perm / dperm loads of a thread otherwise spill)
    const float* __restrict__ permb = perm + (int64_t)b * m * n;
    const float* __restrict__ dpermb = dperm + (int64_t)b * m * n;
    float* __restrict__ dscb = dscores + (int64_t)b * m * n;
    float e[RA][CB], rmx[RA];
#pragma unroll
    for (int a = 0; a < RA; ++a) {
        const int i = ty + 64 * a;
        rmx[a] = i < M1 ? rmax[i] : 0.f;
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) {
            const int j = tx + 16 * bb;
            e[a][bb] = (i < M1 && j < N1) ? __expf(E[i * N1 + j] - rmx[a]) : 0.f;
        }
    }
    __syncthreads();                        // the LDS image is dead: its space is the slab now
    float* P = sm;
    // ---- softmax backward: G = perm * (dperm - rowdot); dv[j] = sum_i G_ij (G itself is rebuilt in the last phase)
    float dvj;                              // reducer threads: dv of column rc
    {
        float gq[CB];
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) gq[bb] = 0.f;
#pragma unroll
        for (int a = 0; a < RA; ++a) {
            const int i = ty + 64 * a;
            float pv[CB], gv[CB], dot = 0.f;
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) {
                const int j = tx + 16 * bb;
                const bool ok = i < m && j < n;
                const uint32_t off = ok ? (uint32_t)(i * n + j) * 4u : 0u;
                pv[bb] = sk::ld_off(permb, off); gv[bb] = sk::ld_off(dpermb, off);
                if (!ok) { pv[bb] = 0.f; gv[bb] = 0.f; }
                dot = fmaf(pv[bb], gv[bb], dot);
            }
            dot = sk::row16_sum(dot);
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) gq[bb] += pv[bb] * (gv[bb] - dot);
            asm volatile("" ::: "memory");      // one row's 2 CB loads in flight at a time (hoisting all RA rows' loads spills)
        }
        dvj = sk::col_reduce<CB>(P, gq, tid, N1);
        if (rc >= n) dvj = 0.f;             // the dustbin column takes no softmax gradient
    }
    float* vt_all = vecs + (int64_t)b * iters * VS;
    // the dual iterates of step t - 1 are loaded while step t is worked on (software pipeline: no load is waited for inside an iteration, and
    // the barriers order LDS traffic only - sk::lds_barrier - so the vector stores of an iteration stay in flight too)
    float hu[RA], hv = 0.f, hvp = 0.f;
    {
        const float* h = uv_hist + ((int64_t)b * iters + (iters - 1)) * (M1 + N1);
#pragma unroll
        for (int a = 0; a < RA; ++a) { const int i = ty + 64 * a; hu[a] = (iters > 0 && i < M1) ? h[i] : 0.f; }
        if (iters > 0 && rc < N1) { hv = h[M1 + rc]; if (iters > 1) hvp = h[M1 + rc - (M1 + N1)]; }
    }
    for (int t = iters; t >= 1; --t) {
        float* vs = vt_all + (int64_t)(t - 1) * VS;            // [A (M1) | D (M1) | B (N1) | C (N1)] of this iteration
        float hu_n[RA], hvp_n = 0.f;
        {
            const float* hn = uv_hist + ((int64_t)b * iters + (t - 2)) * (M1 + N1);      // iteration t - 1 (read only when t > 1)
#pragma unroll
            for (int a = 0; a < RA; ++a) { const int i = ty + 64 * a; hu_n[a] = (t > 1 && i < M1) ? hn[i] : 0.f; }
            if (t > 2 && rc < N1) hvp_n = hn[M1 + rc - (M1 + N1)];
        }
        float cj = 1.f;
        if (rc < N1) {
            const float bj = hv * (rc < n ? inv_mu : inv_nu_last) * dvj;        // the forward's linear-domain duals: V_j / nu_j * dv_j
            if (t > 1) cj = hvp;                                                 // V of the previous iteration (V_0 = 1)
            if (rp == 0) { Bl[rc] = bj; vs[2 * M1 + rc] = bj; vs[2 * M1 + N1 + rc] = cj; }
        }
        sk::lds_barrier();                  // also separates this iteration's slab use from the previous one's
        float br[CB];
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) { const int j = tx + 16 * bb; br[bb] = j < N1 ? Bl[j] : 0.f; }
        // s_i = sum_j E_ij B_j;  du_i = -A_i s_i;  D_i = A_i / mu_i * du_i
        float dr[RA];
#pragma unroll
        for (int a = 0; a < RA; ++a) {
            const int i = ty + 64 * a;
            float sacc = 0.f;
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) sacc = fmaf(e[a][bb], br[bb], sacc);
            sacc = sk::row16_sum(sacc);
            const float ai = hu[a];                        // A_i = exp(u_i + rowmax_i) = U_i, stored as such by the forward
            const float du = -ai * sacc;
            dr[a] = i < M1 ? ai * (i < m ? inv_mu : inv_mu_last) * du : 0.f;
            if (tx == 0 && i < M1) { vs[i] = ai; vs[M1 + i] = dr[a]; }
        }
        // w_j = sum_i E_ij D_i;  dv_{t-1}[j] = -C_j w_j
        float q[CB];
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) {
            float w_ = 0.f;
#pragma unroll
            for (int a = 0; a < RA; ++a) w_ = fmaf(e[a][bb], dr[a], w_);
            q[bb] = w_;
        }
        dvj = -cj * sk::col_reduce<CB>(P, q, tid, N1);
#pragma unroll
        for (int a = 0; a < RA; ++a) hu[a] = hu_n[a];
        hv = hvp; hvp = hvp_n;              // v of iteration t - 1 is this iteration's v_{t-1}
    }
}

// ---- dZ = G - E * sum_t (A^t_i B^t_j + D^t_i C^t_j) from the sweep's per-iteration vectors: no sequential dependency, so this half runs on
// the WHOLE chip (8 row blocks per tile: 512 workgroups for 64 tiles) instead of on the 64 CUs the sweep occupies.  A workgroup = 16 x 16
// threads over its row block x all columns (rows r0 + ty + 16 a, columns tx + 16 b); the vectors of 16 iterations at a time are staged through
// LDS with coalesced loads, E_ij = exp(Z_ij - rowmax_i) is rebuilt from the scores, G from perm / dperm (rowdot by DPP).
constexpr int SK_RS = 8, SK_TC = 16;
template <int RA2, int CB>
__global__ __launch_bounds__(256) void sinkhorn_bwd_dz_kernel(const float* __restrict__ scores, const float* __restrict__ alpha_p, int m, int n, int iters,
                                                              const float* __restrict__ perm, const float* __restrict__ dperm, float* __restrict__ dscores,
                                                              float* __restrict__ dalpha, const int* __restrict__ tile_flags, const float* __restrict__ vecs,
                                                              const float* __restrict__ rmax_in, float* __restrict__ da_slab = nullptr /* slot blockIdx.x */) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.x / SK_RS, rs = blockIdx.x - b * SK_RS;
    const int M1 = m + 1, N1 = n + 1, VS = 2 * (M1 + N1);
    const int RPW = (M1 + SK_RS - 1) / SK_RS, r0 = rs * RPW, nr = min(M1 - r0, RPW);
    if (tile_flags[b] != 0 || nr <= 0) {                  // a wide tile (the log-domain kernel computes it) / a row block past the end
        if (da_slab && threadIdx.x == 0) da_slab[blockIdx.x] = 0.f;
        return;
    }
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int W = 2 * RPW + 2 * N1;                       // floats staged per iteration: [A rows | D rows | B | C]
    const float* vt_all = vecs + (int64_t)b * iters * VS;
    float acc[RA2][CB];
#pragma unroll
    for (int a = 0; a < RA2; ++a)
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) acc[a][bb] = 0.f;
    // LDS offsets of this thread's rows / columns inside one iteration's record; rows / columns past the end read a valid word (their
    // products are never stored)
    int la[RA2], lb[CB];
#pragma unroll
    for (int a = 0; a < RA2; ++a) la[a] = min(ty + 16 * a, nr - 1);
#pragma unroll
    for (int bb = 0; bb < CB; ++bb) lb[bb] = 2 * RPW + min(tx + 16 * bb, N1 - 1);
    for (int t0 = 0; t0 < iters; t0 += SK_TC) {
        const int tc = min(SK_TC, iters - t0);
        __syncthreads();
        for (int x = tid; x < tc * W; x += 256) {
            const int t = x / W, kk = x - t * W;
            const float* vs = vt_all + (int64_t)(t0 + t) * VS;
            float val;
            if (kk < RPW) val = kk < nr ? vs[r0 + kk] : 0.f;
            else if (kk < 2 * RPW) val = (kk - RPW) < nr ? vs[M1 + r0 + kk - RPW] : 0.f;
            else val = vs[2 * M1 + kk - 2 * RPW];
            sm[x] = val;
        }
        __syncthreads();
        for (int t = 0; t < tc; ++t) {
            const float* L = sm + t * W;
            float av[RA2], dd[RA2];
#pragma unroll
            for (int a = 0; a < RA2; ++a) { av[a] = L[la[a]]; dd[a] = L[RPW + la[a]]; }
#pragma unroll
            for (int bb = 0; bb < CB; ++bb) {
                const float bj = L[lb[bb]], cc = L[N1 + lb[bb]];
#pragma unroll
                for (int a = 0; a < RA2; ++a) acc[a][bb] = fmaf(av[a], bj, fmaf(dd[a], cc, acc[a][bb]));
            }
        }
    }
    const float alpha = alpha_p[0];
    const float* __restrict__ permb = perm + (int64_t)b * m * n;
    const float* __restrict__ dpermb = dperm + (int64_t)b * m * n;
    const float* __restrict__ scb = scores + (int64_t)b * m * n;
    float* __restrict__ dscb = dscores + (int64_t)b * m * n;
    float da = 0.f;
#pragma unroll
    for (int a = 0; a < RA2; ++a) {
        const int li = ty + 16 * a, i = r0 + li;
        const bool rok = li < nr;
        const float rm = rok ? rmax_in[(int64_t)b * M1 + i] : 0.f;
        float pv[CB], gv[CB], zz[CB], dot = 0.f;
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) {
            const int j = tx + 16 * bb;
            const bool ok = rok && i < m && j < n;
            const uint32_t off = ok ? (uint32_t)(i * n + j) * 4u : 0u;
            pv[bb] = sk::ld_off(permb, off); gv[bb] = sk::ld_off(dpermb, off); zz[bb] = sk::ld_off(scb, off);
            if (!ok) { pv[bb] = 0.f; gv[bb] = 0.f; zz[bb] = alpha; }
            dot = fmaf(pv[bb], gv[bb], dot);
        }
        dot = sk::row16_sum(dot);
#pragma unroll
        for (int bb = 0; bb < CB; ++bb) {
            const int j = tx + 16 * bb;
            if (rok && j < N1) {
                const float dz = pv[bb] * (gv[bb] - dot) - __expf(zz[bb] - rm) * acc[a][bb];
                if (i < m && j < n) sk::st_off(dscb, (uint32_t)(i * n + j) * 4u, dz);
                else da += dz;
            }
        }
    }
    da = wave_sum(da);
    if (da_slab) {
        __syncthreads();
        if ((tid & 63) == 0) sm[tid >> 6] = da;
        __syncthreads();
        if (tid == 0) da_slab[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    } else if ((tid & 63) == 0 && da != 0.f) atomicAdd(dalpha, da);
}

// ------------------------------------------------------------------------------------------------ losses
// CrossEntropyLoss(ignore_index) over rows: acc[0] += sum(lse - logit[target]), acc[1] += #valid rows; row_lse saved
// grid-stride over rows (one wave per row), ONE pair of atomics per block: 24640 same-address atomics from per-row lanes cost 0.3 ms
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int ld, const int64_t* __restrict__ tgt, int R, int V,
                                                     int ignore, float* __restrict__ row_lse, float* __restrict__ acc, float* __restrict__ slab) {
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
        p3_commit(acc, slab, 2, 0, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        p3_commit(acc, slab, 2, 1, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
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
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n, float* __restrict__ acc, float* __restrict__ slab) {
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
    if (threadIdx.x == 0) p3_commit(acc, slab, 1, 0, (red[0] + red[1]) + (red[2] + red[3]));
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
    // [tile flags | per-iteration vectors A, D, B, C of every tile | row maxima]
    return 256 + ((int64_t)B * 4 + 255) / 256 * 256 + (int64_t)B * iters * 2 * (m + n + 2) * 4 + (int64_t)B * (m + 1) * 4 + 256;
}

extern "C" int p3_sinkhorn_bwd(const float* scores, const float* alpha, int B, int m, int n, int iters, const float* perm,
                               const float* uv_hist, const float* dperm, float* dscores, float* dalpha, void* workspace, void* stream) {
    P3_CHECK(scores && alpha && perm && uv_hist && dperm && dscores && dalpha && workspace && B > 0, P3_EINVAL, "p3_sinkhorn_bwd: bad arguments");
    P3_CHECK(m + 1 <= 16 * SK_MAXK && n + 1 <= 64 * SK_MAXC, P3_EUNSUP, "p3_sinkhorn_bwd: m <= 207, n <= 255");
    const size_t lds = ((size_t)(m + 1) * (n + 1) + 2 * (size_t)(m + 1) + 4 * (size_t)(n + 1)) * sizeof(float);
    P3_CHECK(lds <= 160 * 1024, P3_EUNSUP, "p3_sinkhorn_bwd: does not fit the 160 KB LDS");
    static int force_log = -1;                        // P3_SINKHORN_LOG=1: log-domain loop only (A/B, tests of the fallback)
    if (force_log < 0) { const char* e = getenv("P3_SINKHORN_LOG"); force_log = (e && e[0] == '1') ? 1 : 0; }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)sinkhorn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    hipStream_t s = (hipStream_t)stream;
    int* tile_flags = reinterpret_cast<int*>(workspace);
    float* vecs = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + ((size_t)B * 4 + 255) / 256 * 256);
    // two launches: the linear-domain kernel takes every tile whose row spread allows it and flags the others for the log-domain
    // kernel, which returns at once for the tiles already done (one kernel holding both loops spilled 40 more registers)
    float* rmaxs = vecs + (int64_t)B * iters * 2 * (m + n + 2);
    // deterministic mode: dalpha (the gradient of bin_score) = the tiles' / row blocks' shares added in a fixed order instead of by fp32 atomics
    float* da_slab = p3_det_scratch((int64_t)B * SK_RS + B, P3_F32);
#define P3_SKB_LAUNCH(RA, CB, RA2)                                                                                                        \
    do {                                                                                                                                  \
        const size_t zf = (size_t)(m + 1) * (n + 1) > (size_t)sk::Slab<CB>::FLOATS ? (size_t)(m + 1) * (n + 1) : (size_t)sk::Slab<CB>::FLOATS; \
        const size_t lds_fast = (zf + (size_t)(m + 1) + (size_t)(n + 1) + 2048) * sizeof(float);                                          \
        P3_CHECK(lds_fast <= 160 * 1024 - 512, P3_EUNSUP, "p3_sinkhorn_bwd: does not fit the 160 KB LDS");                                \
        static bool fattr = false;                                                                                                        \
        if (!fattr) {                                                                                                                     \
            hipError_t e = hipFuncSetAttribute((const void*)sinkhorn_bwd_fast_kernel<RA, CB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512); \
            if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }                                                   \
            fattr = true;                                                                                                                 \
        }                                                                                                                                 \
        hipLaunchKernelGGL((sinkhorn_bwd_fast_kernel<RA, CB>), dim3(B), dim3(1024), lds_fast, s, scores, alpha, m, n, iters, perm, uv_hist, dperm, dscores, \
                           dalpha, tile_flags, vecs, rmaxs, force_log);                                                                   \
        const int rpw = (m + 1 + SK_RS - 1) / SK_RS;                                                                                      \
        const size_t lds_dz = (size_t)SK_TC * (2 * rpw + 2 * (n + 1)) * sizeof(float);                                                    \
        hipLaunchKernelGGL((sinkhorn_bwd_dz_kernel<RA2, CB>), dim3(B * SK_RS), dim3(256), lds_dz, s, scores, alpha, m, n, iters, perm, dperm, dscores, \
                           dalpha, tile_flags, vecs, rmaxs, da_slab);                                                                     \
    } while (0)
    {
        const int M1 = m + 1, N1 = n + 1;      // RA2 = ceil(ceil(M1 / 8) / 16)
        if (M1 <= 64 && N1 <= 32) P3_SKB_LAUNCH(1, 2, 1);
        else if (M1 <= 64 && N1 <= 64) P3_SKB_LAUNCH(1, 4, 1);
        else if (M1 <= 128 && N1 <= 128) P3_SKB_LAUNCH(2, 8, 1);
        else if (M1 <= 256 && N1 <= 208) P3_SKB_LAUNCH(4, 13, 2);
        else P3_SKB_LAUNCH(4, 16, 2);
    }
#undef P3_SKB_LAUNCH
    hipLaunchKernelGGL(sinkhorn_bwd_kernel, dim3(B), dim3(1024), lds, s, scores, alpha, m, n, iters, perm, uv_hist, dperm, dscores, dalpha, tile_flags,
                       da_slab ? da_slab + (int64_t)B * SK_RS : nullptr);
    P3_LAUNCH_CHECK();
    if (da_slab) return p3_det_reduce(da_slab, B * SK_RS + B, 1, dalpha, 1, 1, s);
    return P3_OK;
}

extern "C" int p3_ce_loss_fwd(const float* logits, int ld, const int64_t* targets, int R, int V, int ignore_index, float* row_lse, float* acc,
                              void* stream) {
    P3_CHECK(logits && targets && row_lse && acc && R > 0 && V > 0, P3_EINVAL, "p3_ce_loss_fwd: bad arguments");
    const int gr = (R + 3) / 4 < 512 ? (R + 3) / 4 : 512;
    float* slab = p3_det_scratch((int64_t)gr * 2, P3_F32);       // deterministic mode: (loss sum | valid rows) per workgroup, added in workgroup order
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(gr), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, R, V, ignore_index, row_lse, acc, slab);
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce(slab, gr, 2, acc, 2, 1, (hipStream_t)stream);
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
    const int gb = grid_for(n) > 512 ? 512 : grid_for(n);
    float* slab = p3_det_scratch(gb, P3_F32);
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, p, y, n, acc, slab);
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce(slab, gb, 1, acc, 1, 1, (hipStream_t)stream);
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
