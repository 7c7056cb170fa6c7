// p3hip backward glue (HBM-bound elementwise / gather-scatter kernels of the training step).
#include "p3_common.h"

namespace {

inline int grid_for(int64_t work) {
    int64_t g = (work + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// dpre = dy * act'(.) ; GELU: saved = pre-activation ; ReLU: saved = output
template <typename TDY, typename TS, typename TO>
__global__ void act_bwd_kernel(const TDY* __restrict__ dy, const TS* __restrict__ saved, TO* __restrict__ out, int64_t n, int act, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = Cvt<TDY>::to_f(dy[i]) * scale, x = Cvt<TS>::to_f(saved[i]);
        float d;
        if (act == P3_ACT_GELU) {
            const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
            d = g * (cdf + x * pdf);
        } else {
            d = x > 0.f ? g : 0.f;
        }
        out[i] = Cvt<TO>::from_f(d);
    }
}

// embedding + positional gradients: demb[tok[b,t], :] += dx[b,t,:] ; dpos[t,:] += dx[b,t,:]
template <typename T>
__global__ void embed_bwd_kernel(const T* __restrict__ dx, const int64_t* __restrict__ tok, float* __restrict__ demb, float* __restrict__ dpos,
                                 int B, int L, int D) {
    const int64_t total = (int64_t)B * L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int64_t bt = i / D;
        const int t = (int)(bt % L);
        const float g = Cvt<T>::to_f(dx[i]);
        atomicAdd(demb + tok[bt] * D + c, g);
        if (dpos) atomicAdd(dpos + (int64_t)t * D + c, g);   // NULL: the caller takes dpos = sum over the batch with p3_colsum (no atomics)
    }
}

// The same for a small vocabulary (V * 64 floats fit LDS; the Pix2Poly tokenizer has 227 entries): a workgroup owns a 64-channel slice and a
// row range and accumulates into an LDS table [V][64] (ds_add_f32; rows of one token - PAD dominates - no longer serialise on 58 k global
// addresses), then adds its non-zero table entries to demb once.  r02: the global-atomic form took 297 us for 6.3 M atomics.
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_lds_kernel(const T* __restrict__ dx, const int64_t* __restrict__ tok, float* __restrict__ demb,
                                                            int64_t rows, int D, int V, int rows_per_block, float* __restrict__ slab) {
    // slab != NULL (deterministic mode): launched with ONE wave per workgroup - every table entry has a single adder that walks its rows in order - and the
    // whole table (zeros included) goes to slab[blockIdx.x][V][D]; p3_det_reduce adds the workgroups' tables in workgroup order.  Otherwise four waves share
    // the table through LDS atomics and the non-zero entries are added to demb with global atomics.
    extern __shared__ float tab[];                     // [V][64]
    const int tid = threadIdx.x, c = tid & 63, slot = tid >> 6, nslot = blockDim.x >> 6;
    const int c0 = blockIdx.y * 64;
    for (int i = tid; i < V * 64; i += blockDim.x) tab[i] = 0.f;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    if (c0 + c < D) {
        for (int64_t r = r0 + slot; r < r1; r += nslot) {
            const int64_t t = tok[r];
            if (t >= 0 && t < V) atomicAdd(&tab[(int)t * 64 + c], Cvt<T>::to_f(dx[r * D + c0 + c]));
        }
    }
    __syncthreads();
    for (int i = tid; i < V * 64; i += blockDim.x) {
        const float v = tab[i];
        const int cc = c0 + (i & 63);
        if (cc >= D) continue;
        if (slab) slab[((int64_t)blockIdx.x * V + (i >> 6)) * D + cc] = v;
        else if (v != 0.f) atomicAdd(demb + (int64_t)(i >> 6) * D + cc, v);
    }
}

// backward of tokens_assemble: dsrc = dz * scale (or dx), dscale += sum dz*src, dshift += sum dz, with dz = dx * (src*scale+shift > 0)
template <typename TS>
__global__ __launch_bounds__(256) void assemble_bwd_kernel(const float* __restrict__ dx, const TS* __restrict__ src, int src_ld,
                                                           const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                                           TS* __restrict__ dsrc, float* __restrict__ dscale, float* __restrict__ dshift,
                                                           int B, int np, int D, int rows_per_block, float* __restrict__ slab) {
    const int64_t rows = (int64_t)B * np;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int c = threadIdx.x; c < D; c += 256) {
        const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f, mu = mean ? mean[c] : 0.f;   // centred sum if mean given
        float a1 = 0.f, a2 = 0.f;
        int64_t b = r0 / np;                                  // (b, p) tracked incrementally: no 64-bit division per row
        int p = (int)(r0 - b * np);
        for (int64_t rq = r0; rq < r1; rq += 4) {              // four rows in flight: loads first, then the arithmetic
            float g[4], sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool ok = rq + q < r1;
                g[q] = ok ? dx[(b * (np + 1) + 1 + p) * D + c] : 0.f;
                sv[q] = (ok && scale) ? Cvt<TS>::to_f(src[(rq + q) * src_ld + c]) : 0.f;
                if (ok && ++p == np) { p = 0; ++b; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (rq + q >= r1) break;
                float gg = g[q];
                if (scale) {
                    if (sv[q] * sc + sh <= 0.f) gg = 0.f;
                    a1 += gg * (sv[q] - mu); a2 += gg;
                    gg *= sc;
                }
                dsrc[(rq + q) * D + c] = Cvt<TS>::from_f(gg);
            }
        }
        if (scale) {
            // slab: this workgroup's partial row [dscale | dshift], added in workgroup order by p3_det_reduce2 instead of ~1000-deep atomic chains
            if (slab) { slab[(int64_t)blockIdx.x * (2 * D) + c] = a1; slab[(int64_t)blockIdx.x * (2 * D) + D + c] = a2; }
            else { atomicAdd(dscale + c, a1); atomicAdd(dshift + c, a2); }
        }
    }
}

// backward of pool_pos (drop CLS + adaptive average pool over channels)
template <typename TG, typename TY>
__global__ void pool_bwd_kernel(const TG* __restrict__ dout, TY* __restrict__ dy, int B, int np, int Din, int Dout) {
    const int64_t total = (int64_t)B * (np + 1) * Din;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Din);
        const int t = (int)((i / Din) % (np + 1));
        const int64_t b = i / ((int64_t)Din * (np + 1));
        float g = 0.f;
        if (t > 0) {
            int c0 = (int)(((int64_t)k * Dout) / Din) - 1; if (c0 < 0) c0 = 0;
            for (int c = c0; c < Dout && c <= c0 + 3; ++c) {
                const int s = (int)(((int64_t)c * Din) / Dout), e = (int)((((int64_t)(c + 1)) * Din + Dout - 1) / Dout);
                if (k >= s && k < e) g += Cvt<TG>::to_f(dout[(b * np + (t - 1)) * Dout + c]) / (float)(e - s);
            }
        }
        dy[i] = Cvt<TY>::from_f(g);
    }
}

// The same with one thread per input channel k (blockDim = Din rounded up to a wave, Din <= 1024): the <= 4 pooling windows that contain k and
// their 1 / width weights are worked out ONCE per thread, the workgroup then walks its rows with 32-bit arithmetic only.  (The flat form
// above pays ten 64-bit divisions per element: 163 us for the 19 M elements of the ViT output gradient, r02.)
template <typename TG, typename TY>
__global__ void pool_bwd_rows_kernel(const TG* __restrict__ dout, TY* __restrict__ dy, int64_t rows, int np, int Din, int Dout, int rows_per_block) {
    const int k = threadIdx.x;
    int c0 = 0;
    float w[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < Din) {
        c0 = (k * Dout) / Din - 1; if (c0 < 0) c0 = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            if (c < Dout) {
                const int s = (c * Din) / Dout, e = ((c + 1) * Din + Dout - 1) / Dout;
                if (k >= s && k < e) w[j] = 1.f / (float)(e - s);
            }
        }
    }
    int cj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cj[j] = c0 + j < Dout ? c0 + j : Dout - 1;          // clamped: the weight is 0 there
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    int64_t b = r0 / (np + 1);
    int t = (int)(r0 - b * (np + 1));
    for (int64_t r = r0; r < r1; ++r) {
        if (k < Din) {
            float g = 0.f;
            if (t > 0) {
                const TG* src = dout + (b * np + (t - 1)) * Dout;
#pragma unroll
                for (int j = 0; j < 4; ++j) g = fmaf(w[j], Cvt<TG>::to_f(src[cj[j]]), g);
            }
            dy[r * Din + k] = Cvt<TY>::from_f(g);
        }
        if (++t > np) { t = 0; ++b; }
    }
}

// d(pair_mean): dfeats[b, 1+2v+{0,1}, :] += 0.5 * dF[b, v, :] ; position 0 gets nothing from this path
template <typename T>
__global__ void pair_mean_bwd_kernel(const float* __restrict__ dF, T* __restrict__ dfeats, int B, int L, int N, int D, int accumulate) {
    const int64_t total = (int64_t)B * L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int t = (int)((i / D) % L);
        const int64_t b = i / ((int64_t)D * L);
        float g = 0.f;
        if (t >= 1 && t <= 2 * N) g = 0.5f * dF[(b * N + (t - 1) / 2) * D + c];
        dfeats[i] = Cvt<T>::from_f(accumulate ? Cvt<T>::to_f(dfeats[i]) + g : g);
    }
}

}  // namespace

extern "C" int p3_act_bwd(const void* dy, int dtype_dy, const void* saved, int dtype_saved, void* out, int dtype_out, int64_t n, int act,
                          float scale, void* stream) {
    P3_CHECK(dy && saved && out && (act == P3_ACT_GELU || act == P3_ACT_RELU), P3_EINVAL, "p3_act_bwd: bad arguments");
    if (n <= 0) return P3_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(n)), b(256);
#define AB(TDY, TS, TO) hipLaunchKernelGGL((act_bwd_kernel<TDY, TS, TO>), g, b, 0, s, (const TDY*)dy, (const TS*)saved, (TO*)out, n, act, scale)
    const int key = dtype_dy * 4 + dtype_saved * 2 + dtype_out;
    switch (key) {
        case 0: AB(float, float, float); break;
        case 1: AB(float, float, bf16_t); break;
        case 2: AB(float, bf16_t, float); break;
        case 3: AB(float, bf16_t, bf16_t); break;
        case 4: AB(bf16_t, float, float); break;
        case 5: AB(bf16_t, float, bf16_t); break;
        case 6: AB(bf16_t, bf16_t, float); break;
        case 7: AB(bf16_t, bf16_t, bf16_t); break;
        default: p3_set_error("p3_act_bwd: dtype"); return P3_EUNSUP;
    }
#undef AB
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_embed_tokens_bwd(const void* dx, int dtype, const int64_t* tokens, float* demb, float* dpos, int B, int L, int D, void* stream) {
    P3_CHECK(dx && tokens && demb && B > 0, P3_EINVAL, "p3_embed_tokens_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * L * D;
    if (dtype == P3_BF16) hipLaunchKernelGGL((embed_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)dx, tokens, demb, dpos, B, L, D);
    else if (dtype == P3_F32) hipLaunchKernelGGL((embed_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)dx, tokens, demb, dpos, B, L, D);
    else { p3_set_error("p3_embed_tokens_bwd: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_embed_tokens_bwd_v(const void* dx, int dtype, const int64_t* tokens, float* demb, int B, int L, int D, int V, void* stream) {
    P3_CHECK(dx && tokens && demb && B > 0 && V > 0, P3_EINVAL, "p3_embed_tokens_bwd_v: bad arguments");
    if ((size_t)V * 64 * sizeof(float) > 64 * 1024) return p3_embed_tokens_bwd(dx, dtype, tokens, demb, nullptr, B, L, D, stream);
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * L;
    const int rpb = (int)p3_ceil_div(rows, (int64_t)64) < 32 ? 32 : (int)p3_ceil_div(rows, (int64_t)64);     // ~64 row groups x D/64 channel groups
    dim3 grid((unsigned)p3_ceil_div(rows, (int64_t)rpb), (unsigned)p3_ceil_div(D, 64)), block(256);
    const size_t lds = (size_t)V * 64 * sizeof(float);
    // deterministic mode: one wave per workgroup, 128 row groups, the workgroups' tables through the scratch (128 x V x D floats: 30 MB at V = 227, D = 256)
    const int det_groups = 128;
    const int det_rpb = (int)p3_ceil_div(rows, (int64_t)det_groups);
    const int det_grid = (int)p3_ceil_div(rows, (int64_t)det_rpb);
    float* slab = p3_det_scratch((int64_t)det_grid * V * D, dtype);
    if (slab) { grid = dim3((unsigned)det_grid, (unsigned)p3_ceil_div(D, 64)); block = dim3(64); }
    const int rpb_used = slab ? det_rpb : rpb;
    if (dtype == P3_BF16) hipLaunchKernelGGL((embed_bwd_lds_kernel<bf16_t>), grid, block, lds, s, (const bf16_t*)dx, tokens, demb, rows, D, V, rpb_used, slab);
    else if (dtype == P3_F32) hipLaunchKernelGGL((embed_bwd_lds_kernel<float>), grid, block, lds, s, (const float*)dx, tokens, demb, rows, D, V, rpb_used, slab);
    else { p3_set_error("p3_embed_tokens_bwd_v: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce(slab, det_grid, (int64_t)V * D, demb, V * D, 1, s);
    return P3_OK;
}

extern "C" int p3_tokens_assemble_bwd(const float* dx, const void* src, int src_ld, int dtype_src, const float* scale, const float* shift, const float* mean,
                                      void* dsrc, float* dscale, float* dshift, int B, int np, int D, void* stream) {
    P3_CHECK(dx && src && dsrc && B > 0, P3_EINVAL, "p3_tokens_assemble_bwd: bad arguments");
    P3_CHECK(!scale || (shift && dscale && dshift), P3_EINVAL, "p3_tokens_assemble_bwd: scale needs shift/dscale/dshift");
    hipStream_t s = (hipStream_t)stream;
    // rows per workgroup (P3_ASM_RPB): fewer rows = more same-address atomics on the per-channel sums, more rows = a longer serial walk per
    // workgroup; measured 8: 200 us, 16: 125, 32: 94, 64: 99, 128: 142, 192: 205
    // r03: with the partial sums going through the scratch slab (no atomics) short walks win: 16 rows per workgroup (no scratch: atomics, 48 rows)
    int rpb = 48;
    float* slab = nullptr;
    int64_t slab_floats = 0;
    if (scale) {
        const int rpb_s = 16;
        const int64_t nb = p3_ceil_div((int64_t)B * np, rpb_s);
        slab_floats = nb * 2 * D;
        slab = p3_reduce_scratch(slab_floats + p3_ceil_div(nb, 128) * 2 * D);
        if (slab) rpb = rpb_s;
    }
    dim3 g(p3_ceil_div((int64_t)B * np, rpb)), b(256);
    if (dtype_src == P3_BF16) hipLaunchKernelGGL((assemble_bwd_kernel<bf16_t>), g, b, 0, s, dx, (const bf16_t*)src, src_ld, scale, shift, mean, (bf16_t*)dsrc, dscale, dshift, B, np, D, rpb, slab);
    else if (dtype_src == P3_F32) hipLaunchKernelGGL((assemble_bwd_kernel<float>), g, b, 0, s, dx, (const float*)src, src_ld, scale, shift, mean, (float*)dsrc, dscale, dshift, B, np, D, rpb, slab);
    else { p3_set_error("p3_tokens_assemble_bwd: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce2(slab, (int)g.x, 2 * D, slab + slab_floats, dscale, dshift, D, 2 * D, 1, s);
    return P3_OK;
}

extern "C" int p3_pool_pos_bwd(const void* dout, int dtype_dout, void* dy, int dtype_dy, int B, int np, int Din, int Dout, void* stream) {
    P3_CHECK(dout && dy && B > 0, P3_EINVAL, "p3_pool_pos_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * (np + 1) * Din;
    if (Din <= 1024 && (int64_t)Din * Dout < (1ll << 30)) {      // thread-per-channel form
        const int64_t rows = (int64_t)B * (np + 1);
        const int rpb = (int)p3_ceil_div(rows, 2048) < 8 ? 8 : (int)p3_ceil_div(rows, 2048);
        dim3 g2((unsigned)p3_ceil_div(rows, rpb)), b2((unsigned)((Din + 63) / 64 * 64));
#define P3_POOL_ROWS(TG, TY) hipLaunchKernelGGL((pool_bwd_rows_kernel<TG, TY>), g2, b2, 0, s, (const TG*)dout, (TY*)dy, rows, np, Din, Dout, rpb)
        if (dtype_dout == P3_BF16 && dtype_dy == P3_BF16) P3_POOL_ROWS(bf16_t, bf16_t);
        else if (dtype_dout == P3_F32 && dtype_dy == P3_F32) P3_POOL_ROWS(float, float);
        else if (dtype_dout == P3_F32 && dtype_dy == P3_BF16) P3_POOL_ROWS(float, bf16_t);
        else if (dtype_dout == P3_BF16 && dtype_dy == P3_F32) P3_POOL_ROWS(bf16_t, float);
        else { p3_set_error("p3_pool_pos_bwd: dtype"); return P3_EUNSUP; }
#undef P3_POOL_ROWS
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    dim3 g(grid_for(total)), b(256);
    if (dtype_dout == P3_BF16 && dtype_dy == P3_BF16) hipLaunchKernelGGL((pool_bwd_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)dout, (bf16_t*)dy, B, np, Din, Dout);
    else if (dtype_dout == P3_F32 && dtype_dy == P3_F32) hipLaunchKernelGGL((pool_bwd_kernel<float, float>), g, b, 0, s, (const float*)dout, (float*)dy, B, np, Din, Dout);
    else if (dtype_dout == P3_F32 && dtype_dy == P3_BF16) hipLaunchKernelGGL((pool_bwd_kernel<float, bf16_t>), g, b, 0, s, (const float*)dout, (bf16_t*)dy, B, np, Din, Dout);
    else if (dtype_dout == P3_BF16 && dtype_dy == P3_F32) hipLaunchKernelGGL((pool_bwd_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)dout, (float*)dy, B, np, Din, Dout);
    else { p3_set_error("p3_pool_pos_bwd: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_pair_mean_bwd(const float* dF, void* dfeats, int dtype, int B, int L, int N, int D, int accumulate, void* stream) {
    P3_CHECK(dF && dfeats && B > 0, P3_EINVAL, "p3_pair_mean_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * L * D;
    if (dtype == P3_BF16) hipLaunchKernelGGL((pair_mean_bwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, dF, (bf16_t*)dfeats, B, L, N, D, accumulate);
    else if (dtype == P3_F32) hipLaunchKernelGGL((pair_mean_bwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, dF, (float*)dfeats, B, L, N, D, accumulate);
    else { p3_set_error("p3_pair_mean_bwd: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}
