// p3hip - the element-wise pieces of the HiSup head set (models/hisup/model_hisup.py:38-64,122-226) that the GEMM / pad / BatchNorm entries do
// not cover: NCHW -> token-major input conversion, the ECA channel gate (global average pool of two activated maps, conv1d over the
// channel axis, sigmoid) and a row-wise "activated map (x gate) (+ second activated map)" mixer that feeds the next convolution.
// All maps are token-major [B*H*W, ld] in the compute dtype with the producer's BatchNorm + ReLU still pending as per-channel (scale,
// shift): it is applied where the map is consumed, so no activated copy is ever written just to be read once.  HBM-bound, tiny.
#include "p3_common.h"

namespace {

inline int hs_grid(int64_t work) { int64_t g = (work + 255) / 256; return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g)); }

// out[(b*HW + p), c] = in[b, c, p]   (32 x 32 tiles through LDS; fp32 NCHW in, compute dtype out)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ X, T* __restrict__ out, int ld, int C, int64_t HW) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z, p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k; const int64_t p = p0 + tx;
        tile[k][tx] = (p < HW && c < C) ? X[(b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int64_t p = p0 + k; const int c = c0 + tx;
        if (p < HW && c < C) out[(b * HW + p) * ld + c] = Cvt<T>::from_f(tile[tx][k]);
    }
}

// out[r, c] = f(a[r, c]; sa, ha) * gate[b(r), c] + g(b2[r, c]; sb, hb)
//   f(x) = relu(x*sa[c] + ha[c]) when sa != NULL, else x;  gate optional;  second source optional (same rule with sb / hb)
template <typename T>
__global__ void mix_kernel(T* __restrict__ out, int ld_out, const T* __restrict__ a, int ld_a, const float* __restrict__ sa, const float* __restrict__ ha,
                           const float* __restrict__ gate, const T* __restrict__ b2, int ld_b, const float* __restrict__ sb, const float* __restrict__ hb,
                           int64_t R, int C, int64_t HW) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C; const int c = (int)(i - r * C);
        float v = Cvt<T>::to_f(a[r * ld_a + c]);
        if (sa) v = fmaxf(v * sa[c] + ha[c], 0.f);
        if (gate) v *= gate[(r / HW) * C + c];
        if (b2) {
            float w = Cvt<T>::to_f(b2[r * ld_b + c]);
            if (sb) w = fmaxf(w * sb[c] + hb[c], 0.f);
            v += w;
        }
        out[r * ld_out + c] = Cvt<T>::from_f(v);
    }
}

// y[b, c] = mean over the HW rows of sample b of relu(a1*s1 + h1) + relu(a2*s2 + h2)   (ECA.avg_pool(x1 + x2), model_hisup.py:58)
// one workgroup per (b, 64-channel group): 4 row lanes x 64 channels, fixed-order fold -> deterministic
template <typename T>
__global__ __launch_bounds__(256) void eca_pool_kernel(const T* __restrict__ a1, int ld1, const float* __restrict__ s1, const float* __restrict__ h1,
                                                       const T* __restrict__ a2, int ld2, const float* __restrict__ s2, const float* __restrict__ h2,
                                                       float* __restrict__ y, int64_t HW, int C) {
    __shared__ float red[4][64];
    const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float acc = 0.f;
    if (c < C) {
        const float sc1 = s1[c], sh1 = h1[c], sc2 = s2[c], sh2 = h2[c];
        for (int64_t p = q; p < HW; p += 4) {
            const int64_t r = (int64_t)b * HW + p;
            acc += fmaxf(Cvt<T>::to_f(a1[r * ld1 + c]) * sc1 + sh1, 0.f) + fmaxf(Cvt<T>::to_f(a2[r * ld2 + c]) * sc2 + sh2, 0.f);
        }
    }
    red[q][threadIdx.x & 63] = acc;
    __syncthreads();
    if (q == 0 && c < C) y[(int64_t)b * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) / (float)HW;
}

// gate[b, c] = sigmoid(sum_t w[t] * y[b, c + t - k/2])   (Conv1d(1, 1, k, padding = k/2, bias = False) over the channel axis + Sigmoid)
__global__ void eca_gate_kernel(const float* __restrict__ y, const float* __restrict__ w, int k, float* __restrict__ gate, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    float s = 0.f;
    for (int t = 0; t < k; ++t) {
        const int cc = c + t - k / 2;
        if (cc >= 0 && cc < C) s += w[t] * y[b * C + cc];
    }
    gate[i] = 1.f / (1.f + __expf(-s));
}

}  // namespace

extern "C" int p3_nchw_to_nhwc(const float* X, void* out, int ld, int dtype, int B, int C, int64_t HW, void* stream) {
    P3_CHECK(X && out && B > 0 && C > 0 && HW > 0 && ld >= C, P3_EINVAL, "p3_nchw_to_nhwc: bad arguments");
    dim3 g((unsigned)((HW + 31) / 32), (C + 31) / 32, B), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == P3_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), g, b, 0, s, X, (bf16_t*)out, ld, C, HW);
    else if (dtype == P3_F32) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), g, b, 0, s, X, (float*)out, ld, C, HW);
    else { p3_set_error("p3_nchw_to_nhwc: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_affine_relu_mix(void* out, int ld_out, const void* a, int ld_a, const float* scale_a, const float* shift_a, const float* gate,
                                  const void* b, int ld_b, const float* scale_b, const float* shift_b, int64_t R, int C, int64_t HW, int dtype,
                                  void* stream) {
    P3_CHECK(out && a && R > 0 && C > 0 && HW > 0 && ld_out >= C && ld_a >= C, P3_EINVAL, "p3_affine_relu_mix: bad arguments");
    P3_CHECK((scale_a == nullptr) == (shift_a == nullptr) && (scale_b == nullptr) == (shift_b == nullptr), P3_EINVAL, "p3_affine_relu_mix: scale and shift go together");
    P3_CHECK(b || (!scale_b && ld_b == 0), P3_EINVAL, "p3_affine_relu_mix: second source missing");
    hipStream_t s = (hipStream_t)stream;
    const int g = hs_grid(R * C);
    if (dtype == P3_BF16) hipLaunchKernelGGL((mix_kernel<bf16_t>), dim3(g), dim3(256), 0, s, (bf16_t*)out, ld_out, (const bf16_t*)a, ld_a, scale_a, shift_a, gate, (const bf16_t*)b, ld_b, scale_b, shift_b, R, C, HW);
    else if (dtype == P3_F32) hipLaunchKernelGGL((mix_kernel<float>), dim3(g), dim3(256), 0, s, (float*)out, ld_out, (const float*)a, ld_a, scale_a, shift_a, gate, (const float*)b, ld_b, scale_b, shift_b, R, C, HW);
    else { p3_set_error("p3_affine_relu_mix: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_eca_gate(const void* a1, int ld1, const float* scale1, const float* shift1, const void* a2, int ld2, const float* scale2,
                           const float* shift2, const float* conv_w, int k, float* pooled, float* gate, int B, int64_t HW, int C, int dtype, void* stream) {
    P3_CHECK(a1 && a2 && scale1 && shift1 && scale2 && shift2 && conv_w && pooled && gate && B > 0 && HW > 0 && C > 0 && k > 0 && (k & 1), P3_EINVAL,
             "p3_eca_gate: bad arguments (odd kernel size)");
    hipStream_t s = (hipStream_t)stream;
    dim3 g(B, (C + 63) / 64), b(256);
    if (dtype == P3_BF16) hipLaunchKernelGGL((eca_pool_kernel<bf16_t>), g, b, 0, s, (const bf16_t*)a1, ld1, scale1, shift1, (const bf16_t*)a2, ld2, scale2, shift2, pooled, HW, C);
    else if (dtype == P3_F32) hipLaunchKernelGGL((eca_pool_kernel<float>), g, b, 0, s, (const float*)a1, ld1, scale1, shift1, (const float*)a2, ld2, scale2, shift2, pooled, HW, C);
    else { p3_set_error("p3_eca_gate: dtype"); return P3_EUNSUP; }
    P3_LAUNCH_CHECK();
    hipLaunchKernelGGL(eca_gate_kernel, dim3((B * C + 255) / 256), dim3(256), 0, s, pooled, conv_w, k, gate, B, C);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
