// p3hip input pipeline: what the reference does per sample on 16 CPU workers (albumentations D4 + Normalize +
// ToTensorV2 on the image, datasets/build_datasets.py:53-75; apply_d4_augmentations_to_lidar, datasets/p3_coco.py:115-164) done per
// batch on the device, so that the host ships uint8 HWC tiles (a quarter of the fp32 bytes over PCIe) and untransformed points.
//   p3_image_prepare : u8 [B,H,W,C] -> f32 [B,C,H,W], fused D4 group element (index permutation), Normalize ((x - mean*max) * 1/(std*max))
//   p3_points_d4     : in-place D4 of the jagged point list around the tile centre, the reference's fp32 arithmetic step for step
// Both are HBM / byte work: one thread per output pixel / point, x fastest so the fp32 stores are coalesced; the uint8 gathers of the
// transposing elements are column walks that the 150 KB tile keeps in L2.
#include "p3_common.h"

namespace {

// source pixel of destination pixel (i, j) for albumentations' D4 elements on an n x n tile (numpy rot90 / flips / transpose)
__device__ __forceinline__ void d4_source(int g, int i, int j, int n, int& si, int& sj) {
    switch (g) {
        case 1: si = j; sj = n - 1 - i; break;            // r90  = rot90(x, 1)
        case 2: si = n - 1 - i; sj = n - 1 - j; break;    // r180
        case 3: si = n - 1 - j; sj = i; break;            // r270 = rot90(x, 3)
        case 4: si = n - 1 - i; sj = j; break;            // v    = vflip
        case 5: si = n - 1 - j; sj = n - 1 - i; break;    // hvt  = transpose(rot90(x, 2))
        case 6: si = i; sj = n - 1 - j; break;            // h    = hflip
        case 7: si = j; sj = i; break;                    // t    = transpose
        default: si = i; sj = j; break;                   // e
    }
}

struct Norm4 { float sub[4], mul[4]; };

__global__ __launch_bounds__(256) void image_prepare_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ group, float* __restrict__ dst,
                                                            int B, int H, int W, int C, Norm4 nm) {
    const int64_t hw = (int64_t)H * W;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * hw) return;
    const int b = (int)(idx / hw);
    const int p = (int)(idx - (int64_t)b * hw);
    const int i = p / W, j = p - i * W;
    int si = i, sj = j;
    if (group) d4_source(group[b], i, j, H, si, sj);
    const uint8_t* s = src + (((int64_t)b * H + si) * W + sj) * C;
    for (int c = 0; c < C; ++c) dst[((int64_t)b * C + c) * hw + p] = ((float)s[c] - nm.sub[c]) * nm.mul[c];
}

__global__ __launch_bounds__(256) void points_d4_kernel(float* __restrict__ values, const int64_t* __restrict__ offsets, const int32_t* __restrict__ group,
                                                        int B, int64_t total, float cx, float cy) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    int lo = 0, hi = B;                                   // tile b with offsets[b] <= t < offsets[b + 1]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offsets[mid] <= t) lo = mid; else hi = mid; }
    const int g = group[lo];                             // 'e' still takes the centre round trip, (x - c) + c, as the reference does
    float x = values[t * 3] - cx, y = values[t * 3 + 1] - cy;
    float nx = x, ny = y;
    switch (g) {                                          // p3_coco.py:135-158
        case 1: nx = y; ny = -x; break;                   // r90 : swap, y = -y
        case 2: nx = -x; ny = -y; break;                  // r180
        case 3: nx = -y; ny = x; break;                   // r270: swap, x = -x
        case 4: ny = -y; break;                           // v
        case 5: nx = -y; ny = -x; break;                  // hvt : swap, negate both
        case 6: nx = -x; break;                           // h
        case 7: nx = y; ny = x; break;                    // t
        default: break;
    }
    values[t * 3] = nx + cx;
    values[t * 3 + 1] = ny + cy;
}

}  // namespace

// floored modulo like numpy's `%` on float32 (npy_divmodf): result carries the sign of the divisor
__device__ __forceinline__ float np_mod(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.f) { if ((b < 0.f) != (m < 0.f)) m += b; } else { m = copysignf(0.f, b); }
    return m;
}

// FFL ground truth of one batch (datasets/p3_coco.py:254-296): the six uint8 / float masks go through the same D4 permutation as the
// image; gt_polygons_image = clamp(u8 / 255, 0, 1); the crossfield angle = ((u8 * pi / 255 + pi/2) % pi) rotated / mirrored with the
// tile (apply_augmentations_to_ffl_crossfield_angle, :166-205); distances / sizes are permuted only.  float32, the reference's operation order.
__global__ __launch_bounds__(256) void ffl_targets_kernel(const uint8_t* __restrict__ gt_u8, const uint8_t* __restrict__ ang_u8, const float* __restrict__ dist,
                                                          const float* __restrict__ sizes, const int32_t* __restrict__ group, int B, int H, int W,
                                                          float* __restrict__ out_gt, float* __restrict__ out_ang, float* __restrict__ out_dist,
                                                          float* __restrict__ out_sizes) {
    const int64_t hw = (int64_t)H * W;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)B * hw) return;
    const int b = (int)(idx / hw);
    const int p = (int)(idx - (int64_t)b * hw);
    const int i = p / W, j = p - i * W;
    const int g = group ? group[b] : 0;
    int si = i, sj = j;
    if (group) d4_source(g, i, j, H, si, sj);
    const int64_t sp = ((int64_t)b * H + si) * W + sj;
    if (gt_u8) {
#pragma unroll
        for (int c = 0; c < 3; ++c) out_gt[((int64_t)b * 3 + c) * hw + p] = fminf(fmaxf((float)gt_u8[sp * 3 + c] / 255.f, 0.f), 1.f);
    }
    if (dist) out_dist[idx] = dist[sp];
    if (sizes) out_sizes[idx] = sizes[sp];
    if (ang_u8) {
        // the reference's float64 constants (np.pi, np.pi / 2, 3 * np.pi / 2) rounded to float32 when they meet the float32 mask
        constexpr float PI = (float)3.14159265358979323846, PI_2 = (float)(3.14159265358979323846 / 2.0), PI3_2 = (float)(3.0 * 3.14159265358979323846 / 2.0);
        float a = (float)ang_u8[sp] * PI / 255.0f;
        a = np_mod(a + PI_2, PI);                         // normals are stored, tangents are used (:286-287)
        if (group) {
            switch (g) {
                case 1: a = np_mod(a + PI_2, PI); break;              // r90
                case 2: a = np_mod(a + PI, PI); break;                // r180
                case 3: a = np_mod(a + PI3_2, PI); break;             // r270
                case 4: a = np_mod(PI - a, PI); break;                // v
                case 5: a = np_mod(PI3_2 - a, PI); break;             // hvt
                case 6: a = np_mod(-a, PI); break;                    // h
                case 7: a = np_mod(PI_2 - a, PI); break;              // t
                default: break;
            }
        }
        out_ang[idx] = a;
    }
}

extern "C" int p3_image_prepare(const uint8_t* src, const int32_t* group, float* dst, int B, int H, int W, int C, const float* sub,
                                const float* mul, void* stream) {
    P3_CHECK(src && dst && sub && mul, P3_EINVAL, "p3_image_prepare: null pointer");
    P3_CHECK(B > 0 && H > 0 && W > 0 && C > 0 && C <= 4, P3_ESHAPE, "p3_image_prepare: need 1..4 channels");
    P3_CHECK(!group || H == W, P3_ESHAPE, "p3_image_prepare: D4 needs square tiles");
    Norm4 nm;
    for (int c = 0; c < 4; ++c) { nm.sub[c] = c < C ? sub[c] : 0.f; nm.mul[c] = c < C ? mul[c] : 1.f; }
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(image_prepare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, group, dst, B, H, W, C, nm);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_points_d4(float* values, const int64_t* offsets, const int32_t* group, int B, int64_t total, float cx, float cy, void* stream) {
    P3_CHECK(offsets && group, P3_EINVAL, "p3_points_d4: null pointer");
    P3_CHECK(B > 0 && total >= 0, P3_ESHAPE, "p3_points_d4: bad sizes");
    if (total == 0) return P3_OK;
    P3_CHECK(values, P3_EINVAL, "p3_points_d4: null values");
    hipLaunchKernelGGL(points_d4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, values, offsets, group, B, total, cx, cy);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int p3_ffl_targets_prepare(const uint8_t* gt_polygons_u8, const uint8_t* crossfield_angle_u8, const float* distances, const float* sizes,
                                      const int32_t* group, int B, int H, int W, float* out_gt, float* out_angle, float* out_distances,
                                      float* out_sizes, void* stream) {
    P3_CHECK(B > 0 && H > 0 && W > 0, P3_ESHAPE, "p3_ffl_targets_prepare: bad sizes");
    P3_CHECK((!gt_polygons_u8 || out_gt) && (!crossfield_angle_u8 || out_angle) && (!distances || out_distances) && (!sizes || out_sizes), P3_EINVAL,
             "p3_ffl_targets_prepare: every given input needs its output");
    P3_CHECK(!group || H == W, P3_ESHAPE, "p3_ffl_targets_prepare: D4 needs square tiles");
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(ffl_targets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gt_polygons_u8, crossfield_angle_u8, distances,
                       sizes, group, B, H, W, out_gt, out_angle, out_distances, out_sizes);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
