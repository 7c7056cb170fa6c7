// p3hip FFL frame-field loss, forward + gradients fused.
//
// models/ffl/losses.py:220-461 + frame_field_utils.py:9-40 for the shipped config/model/ffl.yaml (seg = interior channel, crossfield on,
// no frequency / distance / size weights, seg.type "bool"): SegLoss (BCE on gt > 0.98 + dice on the float target), CrossfieldAlign,
// CrossfieldAlign90, CrossfieldSmooth (|Laplacian|), SegCrossfield coupling (Scharr gradient of seg, replicate padding) - in the
// reference a dozen elementwise / stencil passes and as many again in autograd over [B, 1 + 4 + 3 + 1, H, W].  Here:
//   pass 1  ffl_loss_sums   every per-pixel term once, block-reduced, fp64 atomics: 5 loss sums + per-tile dice sums
//   pass 2  ffl_loss_point  analytic per-pixel gradients (the dice term needs pass 1's sums): dseg, dcf pointwise parts, plus the two
//                           stencil pre-images dG = dL/d(scharr output) [B,2,H,W] and sL = dL/d(laplacian output) [B,4,H,W]
//   pass 3  ffl_loss_stencil  adjoint Scharr (replicate padding => gather with clamped-index test) into dseg, Laplacian (symmetric,
//                           zero padding) of sL into dcf
// All HBM-bound byte work: 13 fp32 planes read, 5 written, 6 scratch planes; coalesced along W.
#include "p3_common.h"

namespace {

constexpr float SCH_A = 47.f / 512.f, SCH_M = 162.f / 512.f;    // torch_lydorn scharr 3x3 normalised by sum |k| = 512

struct LossCoef {            // weight_i / norm_i of MultiLoss (host side), BCE / dice mix of SegLoss
    float seg, align, align90, smooth, couple, bce_coef, dice_coef;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Scharr gradient (d/di, d/dj) of one seg plane at (i, j), replicate padding, cross-correlation (torch conv)
__device__ __forceinline__ void scharr_ij(const float* __restrict__ s, int H, int W, int i, int j, float& gi, float& gj) {
    const int i0 = clampi(i - 1, 0, H - 1), i2 = clampi(i + 1, 0, H - 1), j0 = clampi(j - 1, 0, W - 1), j2 = clampi(j + 1, 0, W - 1);
    const float a00 = s[i0 * W + j0], a01 = s[i0 * W + j], a02 = s[i0 * W + j2];
    const float a10 = s[i * W + j0], a12 = s[i * W + j2];
    const float a20 = s[i2 * W + j0], a21 = s[i2 * W + j], a22 = s[i2 * W + j2];
    // kernel_x = [[-a,0,a],[-m,0,m],[-a,0,a]] (d/dj); kernel_y = its transpose (d/di)
    gj = SCH_A * (a02 - a00) + SCH_M * (a12 - a10) + SCH_A * (a22 - a20);
    gi = SCH_A * (a20 - a00) + SCH_M * (a21 - a01) + SCH_A * (a22 - a02);
}

__device__ __forceinline__ float laplacian(const float* __restrict__ c, int H, int W, int i, int j) {
    auto at = [&](int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? c[y * W + x] : 0.f; };
    const float corners = at(i - 1, j - 1) + at(i - 1, j + 1) + at(i + 1, j - 1) + at(i + 1, j + 1);
    const float edges = at(i - 1, j) + at(i + 1, j) + at(i, j - 1) + at(i, j + 1);
    return (0.5f * corners + edges - 6.f * at(i, j)) * (1.f / 12.f);
}

// f = z^4 + c2 z^2 + c0 (complex); returns |f|^2 and, optionally, f and z^2
__device__ __forceinline__ float align_err(float c0r, float c0i, float c2r, float c2i, float zr, float zi, float& fr, float& fi, float& z2r, float& z2i) {
    z2r = zr * zr - zi * zi; z2i = zr * zi + zi * zr;
    const float z4r = z2r * z2r - z2i * z2i, z4i = z2r * z2i + z2i * z2r;
    fr = z4r + (c2r * z2r - c2i * z2i) + c0r;
    fi = z4i + (c2r * z2i + c2i * z2r) + c0i;
    return fr * fr + fi * fi;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// acc (double): [0] bce sum, [1] align, [2] align90, [3] smooth, [4] couple, then per tile b: [5 + 3b + {0,1,2}] = sum y*p, sum y, sum p
__global__ __launch_bounds__(256) void ffl_loss_sums_kernel(const float* __restrict__ seg, const float* __restrict__ cf, const float* __restrict__ gt,
                                                            const float* __restrict__ angle, const float* __restrict__ segw, int B, int H, int W,
                                                            double* __restrict__ acc) {
    __shared__ float red[4];
    const int HW = H * W;
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p < HW) {
        const int i = p / W, j = p - i * W;
        const float* sg = seg + (int64_t)b * HW;
        const float* c = cf + (int64_t)b * 4 * HW;
        const float* g = gt + (int64_t)b * 3 * HW;
        const float pr = sg[p], y0 = g[p], ed = g[HW + p], vx = g[2 * HW + p];
        const float yb = y0 > 0.98f ? 1.f : 0.f;
        v[0] = -(yb * fmaxf(logf(pr), -100.f) + (1.f - yb) * fmaxf(logf(1.f - pr), -100.f));      // F.binary_cross_entropy's clamped logs
        if (segw) v[0] *= segw[(int64_t)b * HW + p];                                                // seg_loss_weights (losses.py:150-205)
        v[5] = y0 * pr; v[6] = y0; v[7] = pr;
        const float c0r = c[p], c0i = c[HW + p], c2r = c[2 * HW + p], c2i = c[3 * HW + p];
        float sn, cs; sincosf(angle[(int64_t)b * HW + p], &sn, &cs);
        float fr, fi, z2r, z2i;
        v[1] = align_err(c0r, c0i, c2r, c2i, cs, sn, fr, fi, z2r, z2i) * ed;
        v[2] = align_err(c0r, c0i, c2r, c2i, -sn, cs, fr, fi, z2r, z2i) * fminf(fmaxf(ed - vx, 0.f), 1.f);
        float lap = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) lap += fabsf(laplacian(c + (int64_t)k * HW, H, W, i, j));
        v[3] = lap * (1.f - ed);
        float gi, gj; scharr_ij(sg, H, W, i, j, gi, gj);
        gi *= 2.f; gj *= 2.f;
        const float n = sqrtf(gi * gi + gj * gj), inv = 1.f / (n + 1e-6f);
        v[4] = align_err(c0r, c0i, c2r, c2i, gi * inv, gj * inv, fr, fi, z2r, z2i) * n;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float s = block_sum(v[k], red);
        if (threadIdx.x == 0) atomicAdd(acc + (k < 5 ? k : 5 + 3 * b + (k - 5)), (double)s);
    }
}

// losses[0..4] = the five raw losses (before weight / norm), losses[5] = total = sum coef_i * loss_i with coef_i = weight_i / norm_i
__global__ void ffl_loss_finalize_kernel(const double* __restrict__ acc, int B, int H, int W, LossCoef k, float* __restrict__ losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double npix = (double)B * H * W;
    double dice = 0.0;
    for (int b = 0; b < B; ++b) {
        const float num = 2.f * (float)acc[5 + 3 * b] + 1.f, den = (float)acc[5 + 3 * b + 1] + (float)acc[5 + 3 * b + 2] + 1.f + 1e-7f;
        dice += (double)(1.f - num / den);
    }
    const float seg = k.bce_coef * (float)(acc[0] / npix) + k.dice_coef * (float)(dice / B);
    const float al = (float)(acc[1] / npix), a90 = (float)(acc[2] / npix), sm = (float)(acc[3] / (4.0 * npix)), cp = (float)(acc[4] / npix);
    losses[0] = seg; losses[1] = al; losses[2] = a90; losses[3] = sm; losses[4] = cp;
    losses[5] = k.seg * seg + k.align * al + k.align90 * a90 + k.smooth * sm + k.couple * cp;
}

// gradient of F = |f|^2 w.r.t. (c0, c2) accumulated with factor s
__device__ __forceinline__ void align_grad_c(float fr, float fi, float z2r, float z2i, float s, float& d0r, float& d0i, float& d2r, float& d2i) {
    d0r += s * 2.f * fr; d0i += s * 2.f * fi;
    d2r += s * 2.f * (fr * z2r + fi * z2i);
    d2i += s * 2.f * (fi * z2r - fr * z2i);
}

__global__ __launch_bounds__(256) void ffl_loss_point_kernel(const float* __restrict__ seg, const float* __restrict__ cf, const float* __restrict__ gt,
                                                             const float* __restrict__ angle, const float* __restrict__ segw, int B, int H, int W,
                                                             const double* __restrict__ acc, LossCoef k,
                                                             float* __restrict__ dseg, float* __restrict__ dcf, float* __restrict__ dG, float* __restrict__ sL) {
    const int HW = H * W;
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int i = p / W, j = p - i * W;
    const float inv_n = 1.f / ((float)B * (float)HW);
    const float* sg = seg + (int64_t)b * HW;
    const float* c = cf + (int64_t)b * 4 * HW;
    const float* g = gt + (int64_t)b * 3 * HW;
    const float pr = sg[p], y0 = g[p], ed = g[HW + p], vx = g[2 * HW + p];
    // ---- seg: BCE (torch's backward: (p - y) / max((1 - p) p, 1e-12)) + dice
    const float yb = y0 > 0.98f ? 1.f : 0.f;
    float ds = k.seg * k.bce_coef * inv_n * (pr - yb) / fmaxf((1.f - pr) * pr, 1e-12f);
    if (segw) ds *= segw[(int64_t)b * HW + p];
    {
        const float num = 2.f * (float)acc[5 + 3 * b] + 1.f, den = (float)acc[5 + 3 * b + 1] + (float)acc[5 + 3 * b + 2] + 1.f + 1e-7f;
        ds += k.seg * k.dice_coef * (1.f / (float)B) * -((2.f * y0 * den - num) / (den * den));
    }
    // ---- crossfield: align, align90
    const float c0r = c[p], c0i = c[HW + p], c2r = c[2 * HW + p], c2i = c[3 * HW + p];
    float sn, cs; sincosf(angle[(int64_t)b * HW + p], &sn, &cs);
    float d0r = 0.f, d0i = 0.f, d2r = 0.f, d2i = 0.f, fr, fi, z2r, z2i;
    align_err(c0r, c0i, c2r, c2i, cs, sn, fr, fi, z2r, z2i);
    align_grad_c(fr, fi, z2r, z2i, k.align * inv_n * ed, d0r, d0i, d2r, d2i);
    align_err(c0r, c0i, c2r, c2i, -sn, cs, fr, fi, z2r, z2i);
    align_grad_c(fr, fi, z2r, z2i, k.align90 * inv_n * fminf(fmaxf(ed - vx, 0.f), 1.f), d0r, d0i, d2r, d2i);
    // ---- coupling: z = g / (|g| + 1e-6), g = 2 scharr(seg); weight |g| detached
    float gi, gj; scharr_ij(sg, H, W, i, j, gi, gj);
    gi *= 2.f; gj *= 2.f;
    const float n = sqrtf(gi * gi + gj * gj), inv = 1.f / (n + 1e-6f);
    const float zr = gi * inv, zi = gj * inv;
    align_err(c0r, c0i, c2r, c2i, zr, zi, fr, fi, z2r, z2i);
    const float wc = k.couple * inv_n * n;
    align_grad_c(fr, fi, z2r, z2i, wc, d0r, d0i, d2r, d2i);
    {
        // f'(z) = 4 z^3 + 2 c2 z ; h = conj(f) f' ; dF/dzr = 2 Re h, dF/dzi = -2 Im h
        const float z3r = z2r * zr - z2i * zi, z3i = z2r * zi + z2i * zr;
        const float fpr = 4.f * z3r + 2.f * (c2r * zr - c2i * zi), fpi = 4.f * z3i + 2.f * (c2r * zi + c2i * zr);
        const float hr = fr * fpr + fi * fpi, hi = fr * fpi - fi * fpr;
        const float dzr = wc * 2.f * hr, dzi = wc * -2.f * hi;
        // z = g / (n + eps): dz_k/dg_l = delta_kl / (n + eps) - g_k g_l / (n (n + eps)^2)   (norm backward is 0 at n = 0)
        const float dot = dzr * gi + dzi * gj;
        const float t = n > 0.f ? dot / (n * (n + 1e-6f) * (n + 1e-6f)) : 0.f;
        dG[((int64_t)b * 2 + 0) * HW + p] = 2.f * (dzr * inv - gi * t);      // g = 2 * scharr
        dG[((int64_t)b * 2 + 1) * HW + p] = 2.f * (dzi * inv - gj * t);
    }
    dseg[(int64_t)b * HW + p] = ds;
    float* dc = dcf + (int64_t)b * 4 * HW;
    dc[p] = d0r; dc[HW + p] = d0i; dc[2 * HW + p] = d2r; dc[3 * HW + p] = d2i;
    // ---- smooth: d|L|/dL = sign(L) (0 at 0), mean over B*4*H*W
    const float ws = k.smooth * inv_n * 0.25f * (1.f - ed);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float L = laplacian(c + (int64_t)q * HW, H, W, i, j);
        sL[((int64_t)b * 4 + q) * HW + p] = L > 0.f ? ws : (L < 0.f ? -ws : 0.f);
    }
}

__global__ __launch_bounds__(256) void ffl_loss_stencil_kernel(int B, int H, int W, const float* __restrict__ dG, const float* __restrict__ sL,
                                                               float* __restrict__ dseg, float* __restrict__ dcf) {
    const int HW = H * W;
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int i = p / W, j = p - i * W;
    // adjoint of the replicate-padded Scharr cross-correlation: output pixel q = (qi, qj) read input clamp(q + (a, b)); this input
    // pixel collects k[a][b] * dG[q] from every (q, a, b) whose clamped tap is (i, j)
    const float* gI = dG + ((int64_t)b * 2 + 0) * HW;
    const float* gJ = dG + ((int64_t)b * 2 + 1) * HW;
    float acc = 0.f;
    for (int qi = max(i - 1, 0); qi <= min(i + 1, H - 1); ++qi)
        for (int qj = max(j - 1, 0); qj <= min(j + 1, W - 1); ++qj) {
            const float di = gI[qi * W + qj], dj = gJ[qi * W + qj];
#pragma unroll
            for (int a = -1; a <= 1; ++a)
#pragma unroll
                for (int bb = -1; bb <= 1; ++bb) {
                    if (clampi(qi + a, 0, H - 1) != i || clampi(qj + bb, 0, W - 1) != j) continue;
                    const float kx = (bb == 0 ? 0.f : (float)bb) * (a == 0 ? SCH_M : SCH_A);    // d/dj kernel: column sign, row weight
                    const float ky = (a == 0 ? 0.f : (float)a) * (bb == 0 ? SCH_M : SCH_A);     // d/di kernel: row sign, column weight
                    acc += ky * di + kx * dj;
                }
        }
    dseg[(int64_t)b * HW + p] += acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) dcf[((int64_t)b * 4 + q) * HW + p] += laplacian(sL + ((int64_t)b * 4 + q) * HW, H, W, i, j);
}

}  // namespace

extern "C" int p3_ffl_loss(const float* seg, const float* crossfield, const float* gt_polygons_image, const float* gt_crossfield_angle,
                           const float* seg_weights, int B, int H, int W, const float* coef, float bce_coef, float dice_coef, float* losses,
                           float* dseg, float* dcrossfield, void* workspace, void* stream) {
    P3_CHECK(seg && crossfield && gt_polygons_image && gt_crossfield_angle && coef && losses && workspace, P3_EINVAL, "p3_ffl_loss: null pointer");
    P3_CHECK(B > 0 && H > 1 && W > 1, P3_ESHAPE, "p3_ffl_loss: bad sizes");
    P3_CHECK((dseg == nullptr) == (dcrossfield == nullptr), P3_EINVAL, "p3_ffl_loss: dseg and dcrossfield go together");
    hipStream_t s = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    double* acc = reinterpret_cast<double*>(workspace);
    float* dG = reinterpret_cast<float*>(acc + ((5 + 3 * (int64_t)B + 1) & ~(int64_t)1));
    float* sL = dG + 2 * B * HW;
    hipError_t e = hipMemsetAsync(acc, 0, sizeof(double) * (5 + 3 * (size_t)B), s);
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    LossCoef k; k.seg = coef[0]; k.align = coef[1]; k.align90 = coef[2]; k.smooth = coef[3]; k.couple = coef[4]; k.bce_coef = bce_coef; k.dice_coef = dice_coef;
    dim3 grid((unsigned)((HW + 255) / 256), B), block(256);
    hipLaunchKernelGGL(ffl_loss_sums_kernel, grid, block, 0, s, seg, crossfield, gt_polygons_image, gt_crossfield_angle, seg_weights, B, H, W, acc);
    hipLaunchKernelGGL(ffl_loss_finalize_kernel, dim3(1), dim3(64), 0, s, acc, B, H, W, k, losses);
    if (dseg) {
        hipLaunchKernelGGL(ffl_loss_point_kernel, grid, block, 0, s, seg, crossfield, gt_polygons_image, gt_crossfield_angle, seg_weights, B, H, W, acc, k, dseg, dcrossfield, dG, sL);
        hipLaunchKernelGGL(ffl_loss_stencil_kernel, grid, block, 0, s, B, H, W, dG, sL, dseg, dcrossfield);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}

extern "C" int64_t p3_ffl_loss_workspace_bytes(int B, int H, int W) {
    const int64_t accn = (5 + 3 * (int64_t)B + 1) & ~(int64_t)1;
    return accn * 8 + 6 * (int64_t)B * H * W * 4;
}
