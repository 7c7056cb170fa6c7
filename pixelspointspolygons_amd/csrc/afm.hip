// p3hip attraction field map (the reference's only native kernel, models/hisup/afm_module/afm_op/cuda/afm.cu:29-84).
// For every pixel of tile n: the closest point on any of the tile's line segments -> log-encoded offset (2 planes) + segment label.
// The reference runs one thread per pixel that re-reads every segment from global memory and stores on every improvement; here a
// 256-pixel workgroup stages the tile's segments (pre-scaled end point + direction + 1 / (|d|^2 + 1e-6)) through LDS in chunks that all
// lanes read as broadcasts, keeps the running minimum in registers and stores once.  Arithmetic follows the source expression by
// expression, including its mixed precision: the projection parameter is divided in double (the source adds the double literal 1e-6),
// the log encoding is a double log; multiply-adds are written as the fused forms nvcc's default contraction produces.
#include "p3_common.h"

namespace {

constexpr int SEG_CHUNK = 512;

__global__ __launch_bounds__(256) void afm_kernel(const float* __restrict__ lines, const int32_t* __restrict__ shape_info, int height, int width,
                                                  float* __restrict__ afmap, int32_t* __restrict__ aflabel) {
    __shared__ float sx1[SEG_CHUNK], sy1[SEG_CHUNK], sdx[SEG_CHUNK], sdy[SEG_CHUNK];
    __shared__ double sden[SEG_CHUNK];
    const int n = blockIdx.y;
    const int hw = height * width;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    const int start = shape_info[n * 4], end = shape_info[n * 4 + 1];
    const float xs = (float)width / (float)shape_info[n * 4 + 3];
    const float ys = (float)height / (float)shape_info[n * 4 + 2];
    const float px = (float)(pix % width), py = (float)(pix / width);
    float min_dis = 1e30f, bax = 0.f, bay = 0.f;
    int best = -1;
    for (int c0 = start; c0 < end; c0 += SEG_CHUNK) {
        const int cnt = min(SEG_CHUNK, end - c0);
        __syncthreads();
        for (int s = threadIdx.x; s < cnt; s += 256) {
            const float* l = lines + 4 * (int64_t)(c0 + s);
            const float x1 = l[0] * xs, y1 = l[1] * ys, x2 = l[2] * xs, y2 = l[3] * ys;
            const float dx = x2 - x1, dy = y2 - y1;
            sx1[s] = x1; sy1[s] = y1; sdx[s] = dx; sdy[s] = dy;
            sden[s] = (double)fmaf(dx, dx, dy * dy) + 1e-6;
        }
        __syncthreads();
        if (pix < hw) {
            for (int s = 0; s < cnt; ++s) {
                const float x1 = sx1[s], y1 = sy1[s], dx = sdx[s], dy = sdy[s];
                float t = (float)((double)fmaf(px - x1, dx, (py - y1) * dy) / sden[s]);
                t = t < 1.0f ? t : 1.0f;
                t = t > 0.0f ? t : 0.0f;
                const float ax = fmaf(t, dx, x1) - px;
                const float ay = fmaf(t, dy, y1) - py;
                const float dis = fmaf(ax, ax, ay * ay);
                if (dis < min_dis) { min_dis = dis; bax = ax; bay = ay; best = c0 + s - start; }
            }
        }
    }
    if (pix >= hw) return;
    float ox = 0.f, oy = 0.f;
    int lab = 0;
    if (best >= 0) {
        ox = (float)(-(double)(bax > 0.f ? 1.0f : -1.0f) * log((double)fabsf(bax / (float)width) + 1e-6));
        oy = (float)(-(double)(bay > 0.f ? 1.0f : -1.0f) * log((double)fabsf(bay / (float)height) + 1e-6));
        lab = best;
    }
    afmap[((int64_t)n * 2 + 0) * hw + pix] = ox;
    afmap[((int64_t)n * 2 + 1) * hw + pix] = oy;
    aflabel[(int64_t)n * hw + pix] = lab;
}

}  // namespace

extern "C" int p3_afm(const float* lines, const int32_t* shape_info, int B, int height, int width, float* afmap, int32_t* aflabel, void* stream) {
    P3_CHECK(shape_info && afmap && aflabel, P3_EINVAL, "p3_afm: null pointer");
    P3_CHECK(B > 0 && height > 0 && width > 0, P3_ESHAPE, "p3_afm: bad sizes");
    dim3 grid((unsigned)(((int64_t)height * width + 255) / 256), B);
    hipLaunchKernelGGL(afm_kernel, grid, dim3(256), 0, (hipStream_t)stream, lines, shape_info, height, width, afmap, aflabel);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
