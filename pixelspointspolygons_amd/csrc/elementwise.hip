// p3hip HBM-bound glue kernels (coalesced, vectorised where the layout allows; grid-stride, <= 2048 blocks).
#include "p3_common.h"

namespace {

inline int grid_for(int64_t work, int per_block = 256) {
    int64_t g = (work + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// ---- patchify: NCHW f32 image -> [B*np, 3*P*P] rows (k = c*P*P + py*P + px), the im2col of a stride-P conv ----
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int Cin, int H, int W, int P) {
    const int gw = W / P, gh = H / P, K = Cin * P * P;
    const int64_t total = (int64_t)B * gh * gw * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t row = i / K;
        const int px = k % P, py = (k / P) % P, c = k / (P * P);
        const int gx = (int)(row % gw), gy = (int)((row / gw) % gh), b = (int)(row / ((int64_t)gw * gh));
        out[i] = Cvt<T>::from_f(img[(((int64_t)b * Cin + c) * H + gy * P + py) * W + gx * P + px]);
    }
}

// Row form (K = Cin*P*P <= 1024): one workgroup per patch, thread k = (c, py, px); the patch coordinates are worked out once per workgroup
template <typename T>
__global__ void patchify_rows_kernel(const float* __restrict__ img, T* __restrict__ out, int Cin, int H, int W, int P) {
    const int gw = W / P, gh = H / P, K = Cin * P * P;
    const int k = threadIdx.x;
    if (k >= K) return;
    const int row = blockIdx.x;
    const int gx = row % gw, gy = (row / gw) % gh, b = row / (gw * gh);
    const int px = k % P, py = (k / P) % P, c = k / (P * P);
    out[(int64_t)row * K + k] = Cvt<T>::from_f(img[(((int64_t)b * Cin + c) * H + gy * P + py) * W + gx * P + px]);
}

// ---- tokens_assemble: x[b,0,:] = cls + pos[0]; x[b,1+p,:] = f(src[b,p,:]) + pos[1+p]  (f = BN+ReLU affine or identity) ----
template <typename TS>
__global__ void tokens_assemble_kernel(const TS* __restrict__ src, int src_ld, const float* __restrict__ scale,
                                       const float* __restrict__ shift, const float* __restrict__ cls, const float* __restrict__ pos,
                                       float* __restrict__ x, int B, int np, int D) {
    const int64_t total = (int64_t)B * (np + 1) * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int t = (int)((i / D) % (np + 1));
        const int b = (int)(i / ((int64_t)D * (np + 1)));
        float v;
        if (t == 0) v = cls[c];
        else {
            v = Cvt<TS>::to_f(src[((int64_t)b * np + (t - 1)) * src_ld + c]);
            if (scale) v = fmaxf(v * scale[c] + shift[c], 0.f);
        }
        x[i] = v + pos[(int64_t)t * D + c];
    }
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&v)[4]) {       // 4 consecutive elements (8 / 16 bytes, aligned) as floats
    if constexpr (sizeof(T) == 2) {
        const uint2 raw = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
        v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    } else {
        const float4 raw = *reinterpret_cast<const float4*>(p);
        v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
    }
}

// Row form of tokens_assemble (D % 4 == 0): one workgroup per token row, 4 channels per thread, one division per WORKGROUP instead of three
// 64-bit divisions per element (61 -> ~25 us on the ViT input, r02)
template <typename TS>
__global__ __launch_bounds__(128) void tokens_assemble_rows_kernel(const TS* __restrict__ src, int src_ld, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, const float* __restrict__ cls,
                                                                   const float* __restrict__ pos, float* __restrict__ x, int np, int D) {
    const int row = blockIdx.x, b = row / (np + 1), t = row - b * (np + 1);
    float* xr = x + (int64_t)row * D;
    const float* pr = pos + (int64_t)t * D;
    const TS* sr = t > 0 ? src + ((int64_t)b * np + (t - 1)) * src_ld : nullptr;
    for (int c = threadIdx.x * 4; c < D; c += 128 * 4) {
        float v[4];
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = cls[c + i];
        } else {
            load4<TS>(sr + c, v);
            if (scale) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i] * scale[c + i] + shift[c + i], 0.f);
            }
        }
        const float4 pv = *reinterpret_cast<const float4*>(pr + c);
        *reinterpret_cast<float4*>(xr + c) = make_float4(v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w);
    }
}

// ---- pool_pos: AdaptiveAvgPool1d(Din -> Dout) over channels of tokens 1..np (CLS dropped) + encoder pos embed ----
template <typename TI, typename TO>
__global__ void pool_pos_kernel(const TI* __restrict__ y, const float* __restrict__ pos, TO* __restrict__ out, TO* __restrict__ out_nopos,
                                int B, int np, int Din, int Dout) {
    const int64_t total = (int64_t)B * np * Dout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Dout);
        const int p = (int)((i / Dout) % np);
        const int b = (int)(i / ((int64_t)Dout * np));
        const int s = (int)(((int64_t)c * Din) / Dout), e = (int)((((int64_t)(c + 1)) * Din + Dout - 1) / Dout);
        const TI* row = y + ((int64_t)b * (np + 1) + 1 + p) * Din;
        float a = 0.f;
        for (int k = s; k < e; ++k) a += Cvt<TI>::to_f(row[k]);
        a /= (float)(e - s);
        if (out_nopos) out_nopos[i] = Cvt<TO>::from_f(a);
        out[i] = Cvt<TO>::from_f(pos ? a + pos[(int64_t)p * Dout + c] : a);
    }
}

// Row form of pool_pos (Dout <= 1024): one thread per output channel, its pooling window [s, e) and 1 / width worked out once, the
// workgroup walks its rows with 32-bit arithmetic (the flat form divides 64-bit indices five times per element: 53 -> ~20 us, r02)
template <typename TI, typename TO>
__global__ void pool_pos_rows_kernel(const TI* __restrict__ y, const float* __restrict__ pos, TO* __restrict__ out, TO* __restrict__ out_nopos,
                                     int64_t rows, int np, int Din, int Dout, int rows_per_block) {
    const int c = threadIdx.x;
    int s = 0, e = 1;
    if (c < Dout) { s = (c * Din) / Dout; e = ((c + 1) * Din + Dout - 1) / Dout; }
    const float inv = 1.f / (float)(e - s);
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    int64_t b = r0 / np;
    int p = (int)(r0 - b * np);
    for (int64_t r = r0; r < r1; ++r) {
        if (c < Dout) {
            const TI* row = y + (b * (np + 1) + 1 + p) * Din;
            float a = 0.f;
            for (int k = s; k < e; ++k) a += Cvt<TI>::to_f(row[k]);
            a *= inv;
            if (out_nopos) out_nopos[r * Dout + c] = Cvt<TO>::from_f(a);
            out[r * Dout + c] = Cvt<TO>::from_f(pos ? a + pos[(int64_t)p * Dout + c] : a);
        }
        if (++p == np) { p = 0; ++b; }
    }
}

// ---- embed_tokens: x[b,t,:] = emb[tok[b,t]] + pos[t]; key_bias[b,t] = (tok == pad) ? 1 : 0 ----
template <typename TO>
__global__ void embed_tokens_kernel(const int64_t* __restrict__ tok, const float* __restrict__ emb, const float* __restrict__ pos,
                                    TO* __restrict__ x, float* __restrict__ key_bias, int B, int L, int D, int pad_idx) {
    const int64_t total = (int64_t)B * L * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int64_t bt = i / D;
        const int t = (int)(bt % L);
        const int64_t id = tok[bt];
        x[i] = Cvt<TO>::from_f(emb[id * D + c] + pos[(int64_t)t * D + c]);
        if (c == 0 && key_bias) key_bias[bt] = id == pad_idx ? 1.f : 0.f;
    }
}

// ---- pair_mean: F[b,v,:] = 0.5*(feats[b,1+2v,:] + feats[b,2+2v,:])  (ScoreNet.forward, model_pix2poly.py:87-90) ----
template <typename T>
__global__ void pair_mean_kernel(const T* __restrict__ feats, T* __restrict__ out, int B, int L, int N, int D) {
    const int64_t total = (int64_t)B * N * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const int v = (int)((i / D) % N);
        const int b = (int)(i / ((int64_t)D * N));
        const T* r = feats + ((int64_t)b * L + 1 + 2 * v) * D + c;
        out[i] = Cvt<T>::from_f((Cvt<T>::to_f(r[0]) + Cvt<T>::to_f(r[D])) / 2.f);
    }
}

// ---- ScoreNet BN1 batch statistics in closed form: h[b,i,j,c] = U[b,i,c] + V[b,j,c] ----
template <typename T>
__global__ void pair_stats_kernel(const T* __restrict__ U, const T* __restrict__ V, int N, int C, float* __restrict__ sums, float* __restrict__ slab) {
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        // eight rows in flight per step, four partial sums per moment (a one-row loop is N dependent load latencies on 64 workgroups)
        float pu[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f}, pu2[4] = {0.f, 0.f, 0.f, 0.f}, pv2[4] = {0.f, 0.f, 0.f, 0.f};
        const T* up = U + (int64_t)b * N * C + c;
        const T* vp = V + (int64_t)b * N * C + c;
        int i = 0;
        for (; i + 8 <= N; i += 8) {
            float tu[8], tv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { tu[q] = Cvt<T>::to_f(up[(int64_t)(i + q) * C]); tv[q] = Cvt<T>::to_f(vp[(int64_t)(i + q) * C]); }
#pragma unroll
            for (int q = 0; q < 8; ++q) { pu[q & 3] += tu[q]; pv[q & 3] += tv[q]; pu2[q & 3] += tu[q] * tu[q]; pv2[q & 3] += tv[q] * tv[q]; }
        }
        for (; i < N; ++i) {
            const float u = Cvt<T>::to_f(up[(int64_t)i * C]), v = Cvt<T>::to_f(vp[(int64_t)i * C]);
            pu[0] += u; pv[0] += v; pu2[0] += u * u; pv2[0] += v * v;
        }
        const float su = (pu[0] + pu[1]) + (pu[2] + pu[3]), sv = (pv[0] + pv[1]) + (pv[2] + pv[3]);
        const float su2 = (pu2[0] + pu2[1]) + (pu2[2] + pu2[3]), sv2 = (pv2[0] + pv2[1]) + (pv2[2] + pv2[3]);
        const float t1 = (float)N * (su + sv), t2 = (float)N * (su2 + sv2) + 2.f * su * sv;
        if (slab) { slab[(int64_t)b * 2 * C + c] = t1; slab[(int64_t)b * 2 * C + C + c] = t2; }     // deterministic mode: per-sample parts
        else { atomicAdd(sums + c, t1); atomicAdd(sums + C + c, t2); }
    }
}

// ---- ScoreNet tail: score[b,i,j] (+)= sum_c w4[c]*relu(H3[(b,i,j),c]*sc[c]+sh[c]) + b4 ; transposed accumulate for scorenet2 ----
template <typename T>
__global__ void score_out_kernel(const T* __restrict__ H3, const float* __restrict__ sc, const float* __restrict__ sh,
                                 const float* __restrict__ w4, const float* __restrict__ b4, float* __restrict__ out, int B, int N, int C,
                                 int transpose_acc) {
    // 16 lanes per row (C = 64: 4 channels per lane, one 8/16-byte load), 4 rows per wave pass: fully coalesced
    const int64_t total = (int64_t)B * N * N;
    const int lane = threadIdx.x & 63, sub = lane & 15, rsel = lane >> 4;
    const int c0 = sub * 4;
    float w[4], s[4], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { w[k] = c0 + k < C ? w4[c0 + k] : 0.f; s[k] = c0 + k < C ? sc[c0 + k] : 0.f; h[k] = c0 + k < C ? sh[c0 + k] : 0.f; }
    const float bias = b4[0];
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r0 = wave_id * 4; r0 < total; r0 += nwaves * 4) {
        const int64_t r = r0 + rsel;
        float a = 0.f;
        if (r < total) {
            const T* row = H3 + r * C + c0;
            float v[4];
            if constexpr (sizeof(T) == 2) {
                const uint2 raw = *reinterpret_cast<const uint2*>(row);
                v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
                v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
            } else {
                const float4 raw = *reinterpret_cast<const float4*>(row);
                v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) a += w[k] * fmaxf(v[k] * s[k] + h[k], 0.f);
        }
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64); a += __shfl_xor(a, 8, 64);
        if (sub == 0 && r < total) {
            a += bias;
            if (transpose_acc) {
                const int j = (int)(r % N), i = (int)((r / N) % N);
                const int64_t b = r / ((int64_t)N * N);
                out[(b * N + j) * N + i] += a;
            } else {
                out[r] = a;
            }
        }
    }
}

// C = 64, bf16: 8 lanes per row (16-byte loads), 8 rows per wave pass and 4 passes in flight per iteration; the 8-lane sums by DPP
// (quad_perm x2 + row_half_mirror) instead of four ds_bpermute.  (r02 form: 8-byte loads, one pass in flight: 106 us = 2.9 TB/s.)
__global__ __launch_bounds__(256) void score_out64_kernel(const bf16_t* __restrict__ H3, const float* __restrict__ sc, const float* __restrict__ sh,
                                                          const float* __restrict__ w4, const float* __restrict__ b4, float* __restrict__ out,
                                                          int B, int N, int transpose_acc) {
    constexpr int C = 64, UN = 4;
    const int64_t total = (int64_t)B * N * N;
    const int lane = threadIdx.x & 63, sub = lane & 7, rsel = lane >> 3;
    const int c0 = sub * 8;
    float w[8], s[8], h[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { w[k] = w4[c0 + k]; s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; }
    const float bias = b4[0];
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r0 = wave_id * 8 * UN; r0 < total; r0 += nwaves * 8 * UN) {
        uint4 raw[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t r = r0 + u * 8 + rsel;
            raw[u] = r < total ? *reinterpret_cast<const uint4*>(H3 + r * C + c0) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int64_t r = r0 + u * 8 + rsel;
            const uint32_t wd[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a += w[2 * k] * fmaxf(__uint_as_float(wd[k] << 16) * s[2 * k] + h[2 * k], 0.f);
                a += w[2 * k + 1] * fmaxf(__uint_as_float(wd[k] & 0xffff0000u) * s[2 * k + 1] + h[2 * k + 1], 0.f);
            }
            a += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0xB1, 0xf, 0xf, true));      // lane ^ 1
            a += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x4E, 0xf, 0xf, true));      // lane ^ 2
            a += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(a), 0x141, 0xf, 0xf, true));     // row_half_mirror: the other quad
            if (sub == 0 && r < total) {
                a += bias;
                if (transpose_acc) {
                    const int j = (int)(r % N), i = (int)((r / N) % N);
                    const int64_t b = r / ((int64_t)N * N);
                    out[(b * N + j) * N + i] += a;
                } else {
                    out[r] = a;
                }
            }
        }
    }
}

__global__ void bn_finalize2_kernel(const float* __restrict__ sums, int C, float count, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar, float eps,
                                    float momentum, int training, float* __restrict__ scale, float* __restrict__ shift,
                                    float* __restrict__ save_mean, float* __restrict__ save_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    double rstd_d = 0.0;
    if (training) {
        // E[x^2] - mean^2 in float64: in fp32 the subtraction loses log2(1 + mean^2 / var) bits (the sums are fp32 values; with the
        // deterministic reduction they are float64 sums of workgroup partials rounded once)
        const double md = (double)sums[c] / (double)count;
        const double vd = fmax((double)sums[C + c] / (double)count - md * md, 0.0);
        mean = (float)md; var = (float)vd;
        rstd_d = 1.0 / sqrt(vd + (double)eps);
        if (rmean) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
        }
    } else { mean = rmean[c]; var = rvar[c]; }
    const float rstd = training ? (float)rstd_d : (float)(1.0 / sqrt((double)var + (double)eps)), s = gamma[c] * rstd;
    scale[c] = s; shift[c] = beta[c] - mean * s;
    if (save_mean) { save_mean[c] = mean; save_rstd[c] = rstd; }
}

// ---- argmax over the last dim (greedy decode: softmax -> argmax == argmax of logits; first max wins like torch) ----
__global__ void argmax_kernel(const float* __restrict__ x, int64_t* __restrict__ out, int rows, int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < cols; c += 64) {
        const float v = x[(int64_t)row * ld + c];
        if (v > best || (v == best && c < bi)) { best = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) out[row] = bi;
}

// ---- add_pos: out[b,t,:] = x[b,t,:] + pos[t,:] ----
template <typename T>
__global__ void add_pos_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ out, int64_t total, int64_t LD) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = Cvt<T>::from_f(Cvt<T>::to_f(x[i]) + pos[i % LD]);
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ a, TO* __restrict__ b, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        b[i] = Cvt<TO>::from_f(Cvt<TI>::to_f(a[i]));
}

template <typename TI, typename TO>
__global__ void dropout_apply_kernel(const TI* __restrict__ a, TO* __restrict__ b, int64_t n, int64_t ncols, p3_dropout dr) {
    const DropKey k = drop_key(dr);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ncols;
        b[i] = Cvt<TO>::from_f((!k.on || drop_keep(k, (uint64_t)row, (uint32_t)(i - row * ncols))) ? Cvt<TI>::to_f(a[i]) * k.inv_keep : 0.f);
    }
}

// Batched 2-D transposes inside one bf16 arena: entry e = {src offset, dst offset, rows, cols, first tile}; one 32x32 tile per block
// through LDS.  Refreshes the W^T copies every dX GEMM reads after the optimizer step with ONE launch (r01: 96 strided-copy launches).
struct TransEntry { long long src, dst; int rows, cols; int tile0, tiles_c; };
__global__ __launch_bounds__(256) void transpose_many_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const TransEntry* __restrict__ tab, int n) {
    __shared__ bf16_t tile[32][33];
    int lo = 0, hi = n - 1;                       // last entry whose tile0 <= blockIdx.x
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const TransEntry e = tab[lo];
    const int t = blockIdx.x - e.tile0, tr = t / e.tiles_c, tc = t - tr * e.tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = tr * 32 + ty + 8 * k, c = tc * 32 + tx;
        if (r < e.rows && c < e.cols) tile[ty + 8 * k][tx] = src[e.src + (long long)r * e.cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tc * 32 + ty + 8 * k, r = tr * 32 + tx;      // dst is [cols, rows]
        if (r < e.rows && c < e.cols) dst[e.dst + (long long)c * e.rows + r] = tile[tx][ty + 8 * k];
    }
}

__global__ void rng_advance_kernel(unsigned long long* seed) {
    seed[0] = seed[0] * 6364136223846793005ull + 1442695040888963407ull;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_BF, CALL_F32, name)                 \
    if ((dtype) == P3_BF16) { CALL_BF; }                           \
    else if ((dtype) == P3_F32) { CALL_F32; }                      \
    else { p3_set_error(name ": dtype"); return P3_EUNSUP; }

extern "C" int p3_patchify(const float* img, void* out, int B, int Cin, int H, int W, int P, int dtype_out, void* stream) {
    P3_CHECK(img && out && B > 0 && H % P == 0 && W % P == 0, P3_ESHAPE, "p3_patchify: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * Cin * H * W;
    if (Cin * P * P <= 1024 && (int64_t)B * (H / P) * (W / P) < (1ll << 31)) {
        const dim3 gr((unsigned)(B * (H / P) * (W / P))), bl((unsigned)((Cin * P * P + 63) / 64 * 64));
        DISPATCH_T(dtype_out, hipLaunchKernelGGL((patchify_rows_kernel<bf16_t>), gr, bl, 0, s, img, (bf16_t*)out, Cin, H, W, P),
                   hipLaunchKernelGGL((patchify_rows_kernel<float>), gr, bl, 0, s, img, (float*)out, Cin, H, W, P), "p3_patchify");
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    DISPATCH_T(dtype_out, hipLaunchKernelGGL((patchify_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, img, (bf16_t*)out, B, Cin, H, W, P),
               hipLaunchKernelGGL((patchify_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, img, (float*)out, B, Cin, H, W, P), "p3_patchify");
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_tokens_assemble(const void* src, int src_ld, int dtype_src, const float* scale, const float* shift, const float* cls,
                                  const float* pos, float* x, int B, int np, int D, void* stream) {
    P3_CHECK(src && cls && pos && x && B > 0, P3_EINVAL, "p3_tokens_assemble: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * (np + 1) * D;
    if (D % 4 == 0 && src_ld % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)pos % 16) == 0) {
        const dim3 gr((unsigned)(B * (np + 1)));
        DISPATCH_T(dtype_src, hipLaunchKernelGGL((tokens_assemble_rows_kernel<bf16_t>), gr, dim3(128), 0, s, (const bf16_t*)src, src_ld, scale, shift, cls, pos, x, np, D),
                   hipLaunchKernelGGL((tokens_assemble_rows_kernel<float>), gr, dim3(128), 0, s, (const float*)src, src_ld, scale, shift, cls, pos, x, np, D), "p3_tokens_assemble");
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    DISPATCH_T(dtype_src, hipLaunchKernelGGL((tokens_assemble_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)src, src_ld, scale, shift, cls, pos, x, B, np, D),
               hipLaunchKernelGGL((tokens_assemble_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)src, src_ld, scale, shift, cls, pos, x, B, np, D), "p3_tokens_assemble");
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_pool_pos(const void* y, int dtype_in, const float* pos, void* out, void* out_nopos, int dtype_out, int B, int np, int Din,
                           int Dout, void* stream) {
    P3_CHECK(y && out && B > 0 && Dout > 0 && Din >= Dout, P3_EINVAL, "p3_pool_pos: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * np * Dout;
    if (Dout <= 1024 && (int64_t)Din * Dout < (1ll << 30)) {
        const int64_t rows = (int64_t)B * np;
        const int rpb = (int)p3_ceil_div(rows, 2048) < 8 ? 8 : (int)p3_ceil_div(rows, 2048);
        dim3 g2((unsigned)p3_ceil_div(rows, rpb)), b2((unsigned)((Dout + 63) / 64 * 64));
#define P3_POOL_ROWS(TI, TO) hipLaunchKernelGGL((pool_pos_rows_kernel<TI, TO>), g2, b2, 0, s, (const TI*)y, pos, (TO*)out, (TO*)out_nopos, rows, np, Din, Dout, rpb)
        if (dtype_in == P3_F32 && dtype_out == P3_F32) P3_POOL_ROWS(float, float);
        else if (dtype_in == P3_F32 && dtype_out == P3_BF16) P3_POOL_ROWS(float, bf16_t);
        else if (dtype_in == P3_BF16 && dtype_out == P3_BF16) P3_POOL_ROWS(bf16_t, bf16_t);
        else if (dtype_in == P3_BF16 && dtype_out == P3_F32) P3_POOL_ROWS(bf16_t, float);
        else { p3_set_error("p3_pool_pos: dtype"); return P3_EUNSUP; }
#undef P3_POOL_ROWS
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    dim3 g(grid_for(total)), b(256);
    if (dtype_in == P3_F32 && dtype_out == P3_F32) hipLaunchKernelGGL((pool_pos_kernel<float, float>), g, b, 0, s, (const float*)y, pos, (float*)out, (float*)out_nopos, B, np, Din, Dout);
    else if (dtype_in == P3_F32 && dtype_out == P3_BF16) hipLaunchKernelGGL((pool_pos_kernel<float, bf16_t>), g, b, 0, s, (const float*)y, pos, (bf16_t*)out, (bf16_t*)out_nopos, B, np, Din, Dout);
    else if (dtype_in == P3_BF16 && dtype_out == P3_BF16) hipLaunchKernelGGL((pool_pos_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)y, pos, (bf16_t*)out, (bf16_t*)out_nopos, B, np, Din, Dout);
    else if (dtype_in == P3_BF16 && dtype_out == P3_F32) hipLaunchKernelGGL((pool_pos_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)y, pos, (float*)out, (float*)out_nopos, B, np, Din, Dout);
    else { p3_set_error("p3_pool_pos: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_embed_tokens(const int64_t* tokens, const float* emb, const float* pos, void* x, float* key_bias, int B, int L, int D,
                               int pad_idx, int dtype_out, void* stream) {
    P3_CHECK(tokens && emb && pos && x && B > 0, P3_EINVAL, "p3_embed_tokens: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * L * D;
    DISPATCH_T(dtype_out, hipLaunchKernelGGL((embed_tokens_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, tokens, emb, pos, (bf16_t*)x, key_bias, B, L, D, pad_idx),
               hipLaunchKernelGGL((embed_tokens_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, tokens, emb, pos, (float*)x, key_bias, B, L, D, pad_idx), "p3_embed_tokens");
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_pair_mean(const void* feats, void* out, int B, int L, int N, int D, int dtype, void* stream) {
    P3_CHECK(feats && out && B > 0 && L >= 2 * N + 1, P3_ESHAPE, "p3_pair_mean: need L >= 2N+1");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * N * D;
    DISPATCH_T(dtype, hipLaunchKernelGGL((pair_mean_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)feats, (bf16_t*)out, B, L, N, D),
               hipLaunchKernelGGL((pair_mean_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)feats, (float*)out, B, L, N, D), "p3_pair_mean");
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_pair_stats(const void* U, const void* V, int B, int N, int C, int dtype, float* sums, void* stream) {
    P3_CHECK(U && V && sums && B > 0, P3_EINVAL, "p3_pair_stats: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* slab = p3_det_scratch((int64_t)B * 2 * C, dtype);
    DISPATCH_T(dtype, hipLaunchKernelGGL((pair_stats_kernel<bf16_t>), dim3(B), dim3(256), 0, s, (const bf16_t*)U, (const bf16_t*)V, N, C, sums, slab),
               hipLaunchKernelGGL((pair_stats_kernel<float>), dim3(B), dim3(256), 0, s, (const float*)U, (const float*)V, N, C, sums, slab), "p3_pair_stats");
    P3_LAUNCH_CHECK();
    if (slab) return p3_det_reduce(slab, B, 2 * (int64_t)C, sums, 2 * C, 1, s);
    return P3_OK;
}

extern "C" int p3_bn_finalize(const float* sums, int C, float count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float eps, float momentum, int training, float* scale, float* shift, float* save_mean,
                              float* save_rstd, void* stream) {
    P3_CHECK(gamma && beta && scale && shift && C > 0, P3_EINVAL, "p3_bn_finalize: bad arguments");
    P3_CHECK(training ? sums != nullptr : (running_mean && running_var), P3_EINVAL, "p3_bn_finalize: missing statistics");
    hipLaunchKernelGGL(bn_finalize2_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, C, count, gamma, beta, running_mean,
                       running_var, eps, momentum, training, scale, shift, save_mean, save_rstd);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_score_out(const void* H3, int dtype, const float* scale, const float* shift, const float* w4, const float* b4, float* out,
                            int B, int N, int C, int transpose_accumulate, void* stream) {
    P3_CHECK(H3 && scale && shift && w4 && b4 && out && B > 0, P3_EINVAL, "p3_score_out: bad arguments");
    P3_CHECK(C == 64, P3_EUNSUP, "p3_score_out: ScoreNet conv3 width must be 64 (model_pix2poly.py:78)");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * N * N;
    if (dtype == P3_BF16 && C == 64 && ((uintptr_t)H3 % 16) == 0) {
        hipLaunchKernelGGL(score_out64_kernel, dim3(grid_for(total * 8)), dim3(256), 0, s, (const bf16_t*)H3, scale, shift, w4, b4, out, B, N, transpose_accumulate);
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((score_out_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)H3, scale, shift, w4, b4, out, B, N, C, transpose_accumulate),
               hipLaunchKernelGGL((score_out_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)H3, scale, shift, w4, b4, out, B, N, C, transpose_accumulate), "p3_score_out");
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_argmax(const float* x, int64_t* out, int rows, int cols, int ld, void* stream) {
    P3_CHECK(x && out && rows > 0 && cols > 0, P3_EINVAL, "p3_argmax: bad arguments");
    hipLaunchKernelGGL(argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, out, rows, cols, ld);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_cast(const void* a, int dtype_a, void* b, int dtype_b, int64_t n, void* stream) {
    P3_CHECK(a && b && n >= 0, P3_EINVAL, "p3_cast: bad arguments");
    if (n == 0) return P3_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(n)), blk(256);
    if (dtype_a == P3_F32 && dtype_b == P3_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, blk, 0, s, (const float*)a, (bf16_t*)b, n);
    else if (dtype_a == P3_BF16 && dtype_b == P3_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, blk, 0, s, (const bf16_t*)a, (float*)b, n);
    else if (dtype_a == P3_F32 && dtype_b == P3_F32) hipLaunchKernelGGL((cast_kernel<float, float>), g, blk, 0, s, (const float*)a, (float*)b, n);
    else if (dtype_a == P3_BF16 && dtype_b == P3_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, blk, 0, s, (const bf16_t*)a, (bf16_t*)b, n);
    else { p3_set_error("p3_cast: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_add_pos(const void* x, const float* pos, void* out, int B, int L, int D, int dtype, void* stream) {
    P3_CHECK(x && pos && out && B > 0, P3_EINVAL, "p3_add_pos: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * L * D, LD = (int64_t)L * D;
    DISPATCH_T(dtype, hipLaunchKernelGGL((add_pos_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)x, pos, (bf16_t*)out, total, LD),
               hipLaunchKernelGGL((add_pos_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)x, pos, (float*)out, total, LD), "p3_add_pos");
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_rng_advance(unsigned long long* seed, void* stream) {
    P3_CHECK(seed, P3_EINVAL, "p3_rng_advance: null pointer");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_dropout_apply(const void* in, int dtype_in, void* out, int dtype_out, int64_t n, int64_t ncols, const p3_dropout* drop,
                                void* stream) {
    P3_CHECK(in && out && drop && n >= 0 && ncols > 0, P3_EINVAL, "p3_dropout_apply: bad arguments");
    P3_CHECK(drop->p >= 0.f && drop->p < 1.f, P3_EINVAL, "p3_dropout_apply: p must be in [0, 1)");
    if (n == 0) return P3_OK;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(n)), blk(256);
    const p3_dropout dr = *drop;
    if (dtype_in == P3_F32 && dtype_out == P3_BF16) hipLaunchKernelGGL((dropout_apply_kernel<float, bf16_t>), g, blk, 0, s, (const float*)in, (bf16_t*)out, n, ncols, dr);
    else if (dtype_in == P3_BF16 && dtype_out == P3_F32) hipLaunchKernelGGL((dropout_apply_kernel<bf16_t, float>), g, blk, 0, s, (const bf16_t*)in, (float*)out, n, ncols, dr);
    else if (dtype_in == P3_F32 && dtype_out == P3_F32) hipLaunchKernelGGL((dropout_apply_kernel<float, float>), g, blk, 0, s, (const float*)in, (float*)out, n, ncols, dr);
    else if (dtype_in == P3_BF16 && dtype_out == P3_BF16) hipLaunchKernelGGL((dropout_apply_kernel<bf16_t, bf16_t>), g, blk, 0, s, (const bf16_t*)in, (bf16_t*)out, n, ncols, dr);
    else { p3_set_error("p3_dropout_apply: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_transpose_many(const void* src, void* dst, const void* table, int n_entries, int total_tiles, void* stream) {
    P3_CHECK(src && dst && table && n_entries > 0 && total_tiles > 0, P3_EINVAL, "p3_transpose_many: bad arguments");
    hipLaunchKernelGGL(transpose_many_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst,
                       (const TransEntry*)table, n_entries);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
