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
