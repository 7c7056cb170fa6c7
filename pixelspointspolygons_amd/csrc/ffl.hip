// p3hip FFL / *CNN-encoder tail kernels (models/fusion_layers/early_fusion_vit_cnn.py:87-104, models/ffl/model_ffl.py:53-96):
//   upsample_bilinear   tokens [B, 1+h*w, C] (CLS dropped) -> NHWC map [B, H, W, ld]   (nn.Upsample(size, 'bilinear', align_corners=False))
//   head1x1             relu(bn(x)) . W^T + b -> sigmoid | 2*tanh, NCHW fp32 out (+ optional copy of the result into an NHWC channel)
//   nhwc_to_nchw        BN+ReLU'd feature map in the reference's NCHW layout (API parity of the *CNN encoders' forward)
// The 3x3 convolutions run on p3_gemm (P3_A_CONV3X3 / P3_A_CONV3X3_AFFINE_RELU, BatchNorm statistics in the epilogue).
#include "p3_common.h"

namespace {

template <typename TI, typename TO>
__global__ void upsample_bilinear_kernel(const TI* __restrict__ src, TO* __restrict__ dst, int B, int h, int w, int C, int H, int W, int ld,
                                         int src_tok_off, int src_tok_per_img) {
    // 4 channels per thread
    const int C4 = C / 4;
    const int64_t total = (int64_t)B * H * W * C4;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const int64_t pix = i / C4;
        const int X = (int)(pix % W), Y = (int)((pix / W) % H);
        const int64_t b = pix / ((int64_t)W * H);
        // PyTorch area_pixel_compute_source_index(align_corners=False): max(0, (dst + 0.5) * scale - 0.5)
        float fy = fmaxf(((float)Y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)X + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        const TI* base = src + (b * src_tok_per_img + src_tok_off) * C + c;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v00 = Cvt<TI>::to_f(base[((int64_t)y0 * w + x0) * C + k]), v01 = Cvt<TI>::to_f(base[((int64_t)y0 * w + x1) * C + k]);
            const float v10 = Cvt<TI>::to_f(base[((int64_t)y1 * w + x0) * C + k]), v11 = Cvt<TI>::to_f(base[((int64_t)y1 * w + x1) * C + k]);
            o[k] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        }
        TO* d = dst + pix * ld + c;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = Cvt<TO>::from_f(o[k]);
    }
}

// one wave per pixel row-vector of C = 256 channels (4 per lane); NOUT <= 4 outputs
template <typename T, int NOUT>
__global__ __launch_bounds__(256) void head1x1_kernel(const T* __restrict__ X, int ld, const float* __restrict__ sc, const float* __restrict__ sh,
                                                      const float* __restrict__ Wt, const float* __restrict__ bias, int act, float post_mul,
                                                      float* __restrict__ out_nchw, T* __restrict__ copy_dst, int copy_ld, int64_t R, int64_t HW) {
    const int lane = threadIdx.x & 63, c0 = lane * 4;
    float s[4], h[4], w[NOUT][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; }
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int k = 0; k < 4; ++k) w[o][k] = Wt[o * 256 + c0 + k];
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave_id; r < R; r += nwaves) {
        float v[4];
        if constexpr (sizeof(T) == 2) {
            const uint2 raw = *reinterpret_cast<const uint2*>(X + r * ld + c0);
            v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
            v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
        } else {
            const float4 raw = *reinterpret_cast<const float4*>(X + r * ld + c0);
            v[0] = raw.x; v[1] = raw.y; v[2] = raw.z; v[3] = raw.w;
        }
        float a[NOUT];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) a[o] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x = fmaxf(v[k] * s[k] + h[k], 0.f);
#pragma unroll
            for (int o = 0; o < NOUT; ++o) a[o] += w[o][k] * x;
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) a[o] = wave_sum(a[o]);
        if (lane < NOUT) {
            float y = a[0];
#pragma unroll
            for (int o = 1; o < NOUT; ++o) if (lane == o) y = a[o];
            y += bias[lane];
            y = act == 0 ? 1.f / (1.f + __expf(-y)) : tanhf(y);
            y *= post_mul;
            const int64_t b = r / HW, p = r - b * HW;
            out_nchw[(b * NOUT + lane) * HW + p] = y;
            if (copy_dst && lane == 0) copy_dst[r * copy_ld] = Cvt<T>::from_f(y);
        }
    }
}

// out[b, c, p] = relu(x[b*HW + p, c] * sc[c] + sh[c])   (32x32 tile transpose through LDS)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ X, int ld, const float* __restrict__ sc, const float* __restrict__ sh,
                                                           float* __restrict__ out, int C, int64_t HW) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        const int64_t p = p0 + k; const int c = c0 + tx;
        float v = 0.f;
        if (p < HW && c < C) { v = Cvt<T>::to_f(X[(b * HW + p) * ld + c]); if (sc) v = fmaxf(v * sc[c] + sh[c], 0.f); }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k; const int64_t p = p0 + tx;
        if (p < HW && c < C) out[(b * C + c) * HW + p] = tile[tx][k];
    }
}

inline int grid_for(int64_t work) { int64_t g = (work + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }


// ------------------------------------------------------------------------------------------------ backward
template <typename T>
__device__ __forceinline__ void ffl_ld4(const T* p, float (&v)[4]) {
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
__device__ __forceinline__ void ffl_st4(T* p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        uint2 raw; raw.x = pack_bf2(v[0], v[1]); raw.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = raw;
    } else {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Backward of head1x1 + the BatchNorm/ReLU in front of it: one wave per pixel (C = 256, 4 channels per lane).
//   do_k = dout_k * act'(out_k);  dW[k] += do_k * z;  db[k] += do_k;  dz = sum_k do_k W[k];  dy = dz * [z > 0]
//   dHd = dy * scale  (the "direct" part; the batch-statistics part is added by p3_affine_fix once the sums are complete)
//   acc = [dscale centred (256) | dshift (256) | dW (NOUT*256) | db (NOUT)]
template <typename T, int NOUT>
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(const T* __restrict__ H, const float* __restrict__ sc, const float* __restrict__ sh,
                                                          const float* __restrict__ mean, const float* __restrict__ Wt,
                                                          const float* __restrict__ out_nchw, const float* __restrict__ dout_nchw, int act,
                                                          float post_mul, T* __restrict__ dHd, float* __restrict__ acc, int64_t R, int64_t HW) {
    const int lane = threadIdx.x & 63, c0 = lane * 4, wv = threadIdx.x >> 6;
    float s[4], h[4], mu[4], w[NOUT][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; mu[k] = mean[c0 + k]; }
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int k = 0; k < 4; ++k) w[o][k] = Wt[o * 256 + c0 + k];
    float a_sc[4] = {0, 0, 0, 0}, a_sh[4] = {0, 0, 0, 0}, a_w[NOUT][4], a_b[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) { a_b[o] = 0.f; for (int k = 0; k < 4; ++k) a_w[o][k] = 0.f; }
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave_id; r < R; r += nwaves) {
        float hv[4], z[4], g[4] = {0, 0, 0, 0};
        ffl_ld4<T>(H + r * 256 + c0, hv);
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = fmaxf(hv[k] * s[k] + h[k], 0.f);
        const int64_t b = r / HW, p = r - b * HW;
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const int64_t oi = (b * NOUT + o) * HW + p;
            const float y = out_nchw[oi], dy = dout_nchw[oi];
            float dov;
            if (act == 0) dov = dy * y * (1.f - y);                         // sigmoid
            else { const float t = y / post_mul; dov = dy * post_mul * (1.f - t * t); }   // post_mul * tanh
            a_b[o] += dov;
#pragma unroll
            for (int k = 0; k < 4; ++k) { a_w[o][k] += dov * z[k]; g[k] += dov * w[o][k]; }
        }
        float ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = z[k] > 0.f ? g[k] : 0.f;
            a_sc[k] += dz * (hv[k] - mu[k]); a_sh[k] += dz;
            ov[k] = dz * s[k];
        }
        ffl_st4<T>(dHd + r * 256 + c0, ov);
    }
    __shared__ float red[4][2 * 256 + NOUT * 256 + NOUT];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[wv][c0 + k] = a_sc[k]; red[wv][256 + c0 + k] = a_sh[k];
#pragma unroll
        for (int o = 0; o < NOUT; ++o) red[wv][512 + o * 256 + c0 + k] = a_w[o][k];
    }
    if (lane == 0) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) red[wv][512 + NOUT * 256 + o] = a_b[o];
    }
    __syncthreads();
    constexpr int NV = 512 + NOUT * 256 + NOUT;
    for (int i = threadIdx.x; i < NV; i += 256) atomicAdd(acc + i, (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
}

// Backward through a BatchNorm + ReLU that sits in front of a consumer (C = 256): dA = gradient w.r.t. relu(bn(H)).
//   dy = dA * [bn(H) > 0];  dHd = dy * scale;  acc = [dscale centred (256) | dshift (256)].   H may be strided (ldh), dA / dHd dense.
template <typename T>
__global__ __launch_bounds__(256) void affine_relu_bwd256_kernel(const T* __restrict__ dA, const T* __restrict__ H, int ldh, const float* __restrict__ sc,
                                                                 const float* __restrict__ sh, const float* __restrict__ mean, T* __restrict__ dHd,
                                                                 float* __restrict__ acc, int64_t R) {
    const int lane = threadIdx.x & 63, c0 = lane * 4, wv = threadIdx.x >> 6;
    float s[4], h[4], mu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { s[k] = sc[c0 + k]; h[k] = sh[c0 + k]; mu[k] = mean[c0 + k]; }
    float a_sc[4] = {0, 0, 0, 0}, a_sh[4] = {0, 0, 0, 0};
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t r = wave_id; r < R; r += nwaves) {
        float hv[4], g[4], ov[4];
        ffl_ld4<T>(H + r * ldh + c0, hv);
        ffl_ld4<T>(dA + r * 256 + c0, g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = (hv[k] * s[k] + h[k] > 0.f) ? g[k] : 0.f;
            a_sc[k] += dz * (hv[k] - mu[k]); a_sh[k] += dz;
            ov[k] = dz * s[k];
        }
        ffl_st4<T>(dHd + r * 256 + c0, ov);
    }
    __shared__ float red[4][512];
#pragma unroll
    for (int k = 0; k < 4; ++k) { red[wv][c0 + k] = a_sc[k]; red[wv][256 + c0 + k] = a_sh[k]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) atomicAdd(acc + i, (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
}

// [R = B*H*W, ld_src] (first C channels) -> interior of the zero-bordered [B, H+2, W+2, Cp] image the shifted-row weight-gradient
// GEMMs read.  Channels < c_aff get relu(x*scale + shift) (scale != NULL) or a plain copy, channels [c_aff, C) a plain copy,
// [C, Cp) zeros.  The border must have been cleared by the caller (hipMemsetAsync).
template <typename T>
__global__ void pad_nhwc_kernel(const T* __restrict__ src, int ld_src, const float* __restrict__ sc, const float* __restrict__ sh, int c_aff, int C,
                                int Cp, T* __restrict__ dst, int B, int H, int W) {
    const int Cq = Cp / 4;
    const int64_t total = (int64_t)B * H * W * Cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cq) * 4;
        const int64_t pix = i / Cq;
        const int x = (int)(pix % W), y = (int)((pix / W) % H);
        const int64_t b = pix / ((int64_t)W * H);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c + k < C) {
                float t = Cvt<T>::to_f(src[pix * ld_src + c + k]);
                if (sc && c + k < c_aff) t = fmaxf(t * sc[c + k] + sh[c + k], 0.f);
                v[k] = t;
            }
        }
        T* d = dst + ((b * (H + 2) + (y + 1)) * (int64_t)(W + 2) + (x + 1)) * Cp + c;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = Cvt<T>::from_f(v[k]);
    }
}

// bf16, 8-channel (16-byte) form: one pass over the WHOLE destination, every element written once (zero border, interior, zero channel
// padding) - the scalar form above pays a 2 GB memset plus 2-byte accesses on the bs-64 FFL maps (1.5 ms; this one moves 3.7 GB)
__global__ __launch_bounds__(256) void pad_nhwc_vec_kernel(const bf16_t* __restrict__ src, int ld_src, const float* __restrict__ sc, const float* __restrict__ sh,
                                                           int c_aff, int C, int Cp, bf16_t* __restrict__ dst, int B, int H, int W) {
    const int Cq = Cp / 8, P = W + 2;
    const int64_t total = (int64_t)B * (H + 2) * P * Cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cq) * 8;
        const int64_t pp = i / Cq;
        const int xp = (int)(pp % P), yp = (int)((pp / P) % (H + 2));
        const int64_t b = pp / ((int64_t)P * (H + 2));
        uint4 out = make_uint4(0, 0, 0, 0);
        if (xp >= 1 && xp <= W && yp >= 1 && yp <= H && c < C) {
            const int64_t pix = (b * H + (yp - 1)) * W + (xp - 1);
            out = *reinterpret_cast<const uint4*>(src + pix * ld_src + c);
            if (sc && c < c_aff) {
                const uint32_t w[4] = {out.x, out.y, out.z, out.w};
                float v[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[2 * k] = __uint_as_float(w[k] << 16); v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
                const float4 s0 = *reinterpret_cast<const float4*>(sc + c), s1 = *reinterpret_cast<const float4*>(sc + c + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sh + c), h1 = *reinterpret_cast<const float4*>(sh + c + 4);
                const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, hh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] * ss[k] + hh[k], 0.f);
                out = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            }
        }
        *reinterpret_cast<uint4*>(dst + i * 8) = out;
    }
}

// Adjoint of upsample_bilinear, separable: pass X folds the W output columns onto the w source columns, pass Y the H rows onto h.
// Exactly the forward's index / weight rule, gathered per source cell (deterministic, no atomics).
__device__ __forceinline__ float bilinear_weight(int o, int srcn, int outn, int cell) {
    const float f = fmaxf(((float)o + 0.5f) * ((float)srcn / (float)outn) - 0.5f, 0.f);
    const int i0 = (int)f, i1 = i0 + (i0 < srcn - 1 ? 1 : 0);
    const float l = f - (float)i0;
    return (i0 == cell ? 1.f - l : 0.f) + (i1 == cell ? l : 0.f);
}

template <typename T>
__global__ void upsample_bwd_x_kernel(const T* __restrict__ dUp, float* __restrict__ tmp, int B, int H, int W, int w, int C) {
    // tmp[b, Y, xs, c] = sum_X wx(X, xs) * dUp[b, Y, X, c]
    const int Cq = C / 4;
    const int64_t total = (int64_t)B * H * w * Cq;
    const int span = (W + w - 1) / w;               // output columns per source cell
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cq) * 4;
        const int64_t t = i / Cq;
        const int xs = (int)(t % w);
        const int64_t row = t / w;                  // b*H + Y
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int lo = max(0, xs * span - span), hi = min(W - 1, xs * span + 2 * span - 1);
        for (int X = lo; X <= hi; ++X) {
            const float wt = bilinear_weight(X, w, W, xs);
            if (wt != 0.f) {
                float v[4];
                ffl_ld4<T>(dUp + (row * W + X) * C + c, v);
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += wt * v[k];
            }
        }
        *reinterpret_cast<float4*>(tmp + (row * w + xs) * C + c) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

template <typename T>
__global__ void upsample_bwd_y_kernel(const float* __restrict__ tmp, T* __restrict__ dtok, int B, int H, int h, int w, int C, int tok_off,
                                      int tok_per_img) {
    // dtok[b, tok_off + ys*w + xs, c] = sum_Y wy(Y, ys) * tmp[b, Y, xs, c]
    const int Cq = C / 4;
    const int64_t total = (int64_t)B * h * w * Cq;
    const int span = (H + h - 1) / h;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cq) * 4;
        const int64_t t = i / Cq;
        const int xs = (int)(t % w), ys = (int)((t / w) % h);
        const int64_t b = t / ((int64_t)w * h);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int lo = max(0, ys * span - span), hi = min(H - 1, ys * span + 2 * span - 1);
        for (int Y = lo; Y <= hi; ++Y) {
            const float wt = bilinear_weight(Y, h, H, ys);
            if (wt != 0.f) {
                const float4 v = *reinterpret_cast<const float4*>(tmp + ((b * H + Y) * (int64_t)w + xs) * C + c);
                a[0] += wt * v.x; a[1] += wt * v.y; a[2] += wt * v.z; a[3] += wt * v.w;
            }
        }
        ffl_st4<T>(dtok + (b * tok_per_img + tok_off + (int64_t)ys * w + xs) * C + c, a);
    }
}

}  // namespace

extern "C" int p3_upsample_bilinear(const void* src, int dtype_src, void* dst, int dtype_dst, int B, int h, int w, int C, int H, int W, int ld,
                                    int src_tok_off, int src_tok_per_img, void* stream) {
    P3_CHECK(src && dst && B > 0 && C % 4 == 0 && ld >= C, P3_EINVAL, "p3_upsample_bilinear: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W * (C / 4);
    dim3 g(grid_for(total)), b(256);
    if (dtype_src == P3_BF16 && dtype_dst == P3_BF16) hipLaunchKernelGGL((upsample_bilinear_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, B, h, w, C, H, W, ld, src_tok_off, src_tok_per_img);
    else if (dtype_src == P3_F32 && dtype_dst == P3_F32) hipLaunchKernelGGL((upsample_bilinear_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, B, h, w, C, H, W, ld, src_tok_off, src_tok_per_img);
    else { p3_set_error("p3_upsample_bilinear: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_head1x1(const void* X, int ld, int dtype, const float* scale, const float* shift, const float* W, const float* bias, int n_out,
                          int act /*0 sigmoid, 1 tanh*/, float post_mul, float* out_nchw, void* copy_dst, int copy_ld, int64_t R, int64_t HW,
                          void* stream) {
    P3_CHECK(X && scale && shift && W && bias && out_nchw && R > 0 && (n_out == 1 || n_out == 4), P3_EINVAL, "p3_head1x1: bad arguments (C = 256, n_out in {1, 4})");
    hipStream_t s = (hipStream_t)stream;
    dim3 g(grid_for(R * 64)), b(256);
#define H1(T, N) hipLaunchKernelGGL((head1x1_kernel<T, N>), g, b, 0, s, (const T*)X, ld, scale, shift, W, bias, act, post_mul, out_nchw, (T*)copy_dst, copy_ld, R, HW)
    if (dtype == P3_BF16) { if (n_out == 1) H1(bf16_t, 1); else H1(bf16_t, 4); }
    else if (dtype == P3_F32) { if (n_out == 1) H1(float, 1); else H1(float, 4); }
    else { p3_set_error("p3_head1x1: dtype"); return P3_EUNSUP; }
#undef H1
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_nhwc_to_nchw(const void* X, int ld, int dtype, const float* scale, const float* shift, float* out, int B, int C, int64_t HW,
                               void* stream) {
    P3_CHECK(X && out && B > 0 && C > 0, P3_EINVAL, "p3_nhwc_to_nchw: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)((HW + 31) / 32), (C + 31) / 32, B), b(256);
    if (dtype == P3_BF16) hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t>), g, b, 0, s, (const bf16_t*)X, ld, scale, shift, out, C, HW);
    else if (dtype == P3_F32) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float>), g, b, 0, s, (const float*)X, ld, scale, shift, out, C, HW);
    else { p3_set_error("p3_nhwc_to_nchw: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_head1x1_bwd(const void* H, int dtype, const float* scale, const float* shift, const float* mean, const float* W, int n_out,
                              const float* out_nchw, const float* dout_nchw, int act, float post_mul, void* dHd, float* acc, int64_t R, int64_t HW,
                              void* stream) {
    P3_CHECK(H && scale && shift && mean && W && out_nchw && dout_nchw && dHd && acc && R > 0 && HW > 0, P3_EINVAL, "p3_head1x1_bwd: bad arguments");
    P3_CHECK(n_out == 1 || n_out == 4, P3_EUNSUP, "p3_head1x1_bwd: n_out must be 1 or 4");
    P3_CHECK(dtype == P3_BF16 || dtype == P3_F32, P3_EUNSUP, "p3_head1x1_bwd: dtype");
    hipStream_t s = (hipStream_t)stream;
    int64_t gr = (R + 3) / 4; if (gr > 2048) gr = 2048;
    dim3 g((int)gr), b(256);
#define HB(T, NO) hipLaunchKernelGGL((head1x1_bwd_kernel<T, NO>), g, b, 0, s, (const T*)H, scale, shift, mean, W, out_nchw, dout_nchw, act, post_mul, (T*)dHd, acc, R, HW)
    if (dtype == P3_BF16) { if (n_out == 1) HB(bf16_t, 1); else HB(bf16_t, 4); }
    else { if (n_out == 1) HB(float, 1); else HB(float, 4); }
#undef HB
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_affine_relu_bwd256(const void* dA, const void* H, int ldh, int dtype, const float* scale, const float* shift, const float* mean,
                                     void* dHd, float* acc, int64_t R, void* stream) {
    P3_CHECK(dA && H && scale && shift && mean && dHd && acc && R > 0 && ldh >= 256 && ldh % 4 == 0, P3_EINVAL, "p3_affine_relu_bwd256: bad arguments");
    P3_CHECK(dtype == P3_BF16 || dtype == P3_F32, P3_EUNSUP, "p3_affine_relu_bwd256: dtype");
    hipStream_t s = (hipStream_t)stream;
    int64_t gr = (R + 3) / 4; if (gr > 2048) gr = 2048;
    if (dtype == P3_BF16) hipLaunchKernelGGL((affine_relu_bwd256_kernel<bf16_t>), dim3((int)gr), dim3(256), 0, s, (const bf16_t*)dA, (const bf16_t*)H, ldh, scale, shift, mean, (bf16_t*)dHd, acc, R);
    else hipLaunchKernelGGL((affine_relu_bwd256_kernel<float>), dim3((int)gr), dim3(256), 0, s, (const float*)dA, (const float*)H, ldh, scale, shift, mean, (float*)dHd, acc, R);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_pad_nhwc(const void* src, int ld_src, int dtype, const float* scale, const float* shift, int c_aff, int C, int Cp, void* dst, int B, int H,
                           int W, void* stream) {
    P3_CHECK(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && Cp >= C && Cp % 4 == 0 && c_aff <= C, P3_EINVAL, "p3_pad_nhwc: bad arguments");
    P3_CHECK((scale == nullptr) == (shift == nullptr), P3_EINVAL, "p3_pad_nhwc: scale and shift go together");
    P3_CHECK(dtype == P3_BF16 || dtype == P3_F32, P3_EUNSUP, "p3_pad_nhwc: dtype");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P3_BF16 && C % 8 == 0 && c_aff % 8 == 0 && Cp % 8 == 0 && ld_src % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 &&
        (!scale || (((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0))) {
        const int64_t tot = (int64_t)B * (H + 2) * (W + 2) * (Cp / 8);
        int64_t g = (tot + 255) / 256; if (g > 65536) g = 65536;
        hipLaunchKernelGGL(pad_nhwc_vec_kernel, dim3((int)g), dim3(256), 0, s, (const bf16_t*)src, ld_src, scale, shift, c_aff, C, Cp, (bf16_t*)dst, B, H, W);
        P3_LAUNCH_CHECK();
        return P3_OK;
    }
    const size_t es = dtype == P3_BF16 ? 2 : 4;
    hipError_t e = hipMemsetAsync(dst, 0, (size_t)B * (H + 2) * (W + 2) * Cp * es, s);
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    const int64_t total = (int64_t)B * H * W * (Cp / 4);
    int64_t gr = (total + 255) / 256; if (gr > 16384) gr = 16384;
    if (dtype == P3_BF16) hipLaunchKernelGGL((pad_nhwc_kernel<bf16_t>), dim3((int)gr), dim3(256), 0, s, (const bf16_t*)src, ld_src, scale, shift, c_aff, C, Cp, (bf16_t*)dst, B, H, W);
    else hipLaunchKernelGGL((pad_nhwc_kernel<float>), dim3((int)gr), dim3(256), 0, s, (const float*)src, ld_src, scale, shift, c_aff, C, Cp, (float*)dst, B, H, W);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_upsample_bilinear_bwd(const void* dUp, int dtype, float* tmp, void* dtok, int B, int h, int w, int C, int H, int W, int tok_off,
                                        int tok_per_img, void* stream) {
    P3_CHECK(dUp && tmp && dtok && B > 0 && C % 4 == 0 && H >= h && W >= w, P3_EINVAL, "p3_upsample_bilinear_bwd: bad arguments");
    P3_CHECK(dtype == P3_BF16 || dtype == P3_F32, P3_EUNSUP, "p3_upsample_bilinear_bwd: dtype");
    hipStream_t s = (hipStream_t)stream;
    const int64_t t1 = (int64_t)B * H * w * (C / 4), t2 = (int64_t)B * h * w * (C / 4);
    int64_t g1 = (t1 + 255) / 256, g2 = (t2 + 255) / 256;
    if (g1 > 16384) g1 = 16384;
    if (g2 > 16384) g2 = 16384;
    if (dtype == P3_BF16) {
        hipLaunchKernelGGL((upsample_bwd_x_kernel<bf16_t>), dim3((int)g1), dim3(256), 0, s, (const bf16_t*)dUp, tmp, B, H, W, w, C);
        hipLaunchKernelGGL((upsample_bwd_y_kernel<bf16_t>), dim3((int)g2), dim3(256), 0, s, tmp, (bf16_t*)dtok, B, H, h, w, C, tok_off, tok_per_img);
    } else {
        hipLaunchKernelGGL((upsample_bwd_x_kernel<float>), dim3((int)g1), dim3(256), 0, s, (const float*)dUp, tmp, B, H, W, w, C);
        hipLaunchKernelGGL((upsample_bwd_y_kernel<float>), dim3((int)g2), dim3(256), 0, s, tmp, (float*)dtok, B, H, h, w, C, tok_off, tok_per_img);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}
