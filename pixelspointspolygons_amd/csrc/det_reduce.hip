// p3hip - deterministic (run-to-run bit-reproducible) reductions.
//
// Several kernels of the path finish in fp32 atomicAdd's over workgroup partials (BatchNorm sums of the ScoreNet, split-M weight
// gradients, column sums): the order of the adds differs from run to run, so results differ in the last bits - and through the ReLU
// decisions behind a train-mode BatchNorm those bits select different elements near the kink (the ScoreNet backward test at
// N = 192 passed or failed on that order).  With a scratch region registered by the host (p3_set_deterministic) those kernels store their
// workgroup partials with plain stores instead and det_reduce_kernel adds them in workgroup order in float64: same bits every run, and the
// 10^3..10^5-term BatchNorm sums lose nothing to cancellation.  The region is used launch by launch in stream order (one compute stream).
#include <stdlib.h>

#include "p3_common.h"

static float* g_det = nullptr;
static int64_t g_det_floats = 0;          // floats of ONE region
static int g_det_all = 0;
// Launch streams: the scratch is cut into equal regions: region 0 serves every stream that was not registered as a SIDE stream (the default
// stream, and the capture stream torch.cuda.graph() runs a step on - r04: keying regions on "streams in first-come order" sent the captured
// step's launches to a region that did not exist, i.e. silently back to the atomics paths: ln_bwd 34 -> 55 us), region 1 + i the i-th registered
// side stream (r04: ScoreNet 2 / the weight-gradient GEMMs / the pillar stem may run beside the main stream, ops.SIDE).  The host names the
// stream of the launches that follow (p3_scratch_stream, called by the binding whenever its current stream changes); a side stream beyond the
// region count gets no scratch (its launches take their atomics path).
static int g_regions = 1, g_region = 0;
static int64_t g_total_floats = 0;
static void* g_side[15];
static int g_nside = 0;

extern "C" int p3_set_deterministic(void* scratch, int64_t bytes, int all_dtypes) {
    P3_CHECK((scratch == nullptr) == (bytes == 0) && bytes >= 0 && ((uintptr_t)scratch % 16) == 0, P3_EINVAL,
             "p3_set_deterministic: scratch and bytes go together, 16-byte aligned");
    g_det = (float*)scratch;
    g_total_floats = bytes / 4;
    g_det_floats = (g_total_floats / g_regions) & ~(int64_t)63;
    g_det_all = all_dtypes;
    return P3_OK;
}

extern "C" int p3_scratch_regions(int n) {
    P3_CHECK(n >= 1 && n <= 16, P3_EINVAL, "p3_scratch_regions: 1..16");
    g_regions = n;
    g_det_floats = (g_total_floats / g_regions) & ~(int64_t)63;
    return P3_OK;
}

extern "C" int p3_scratch_side_stream(void* stream) {
    for (int i = 0; i < g_nside; ++i)
        if (g_side[i] == stream) return i + 1;
    P3_CHECK(g_nside < 15, P3_EINVAL, "p3_scratch_side_stream: too many side streams");
    g_side[g_nside++] = stream;
    return g_nside;
}

extern "C" int p3_scratch_stream(void* stream) {
    g_region = 0;
    for (int i = 0; i < g_nside; ++i)
        if (g_side[i] == stream) { g_region = i + 1; break; }
    return g_region < g_regions ? g_region : -1;
}

extern "C" int p3_get_deterministic(void) { return g_det ? (g_det_all ? 2 : 1) : 0; }

static inline float* region_base(int64_t floats) {
    if (!g_det || g_region >= g_regions || floats > g_det_floats) return nullptr;
    return g_det + (int64_t)g_region * g_det_floats;
}

// fp32 (parity mode) launches always take the deterministic path when a scratch is registered; bf16 ones only with all_dtypes
float* p3_det_scratch(int64_t floats, int dtype) {
    if (dtype != P3_F32 && !g_det_all) return nullptr;
    return region_base(floats);
}

// any launch (bf16 included) may borrow the registered scratch for per-tile partials that replace a slower scheme (the GEMM's BatchNorm column
// sums: gemm.hip); NULL when none is registered or it is too small - the caller then takes its other path
float* p3_reduce_scratch(int64_t floats) { return region_base(floats); }

namespace {
// level 1 of a long reduction: tmp[chunk][i] = sum of the parts [chunk*CH, (chunk+1)*CH) (float64, part order)
__global__ __launch_bounds__(256) void det_reduce_chunk_kernel(const float* __restrict__ parts, int nparts, int64_t stride, float* __restrict__ tmp, int nvals, int ch) {
    __shared__ double red[4][64];
    const int v = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int c0 = blockIdx.y * ch, c1 = min(nparts, c0 + ch);
    const int per = (c1 - c0 + 3) / 4, p0 = c0 + q * per, p1 = min(c1, p0 + per);
    double a = 0.0;
    if (v < nvals) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {            // eight loads in flight, added in part order
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = parts[(int64_t)(p + u) * stride + v];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)x[u];
        }
        for (; p < p1; ++p) a += (double)parts[(int64_t)p * stride + v];
    }
    red[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && v < nvals) tmp[(int64_t)blockIdx.y * nvals + v] = (float)(((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x]);
}

// out[i] (+)= sum_p parts[p*stride + i], p ascending, float64 accumulation.  Block = 64 values x 4 part lanes (each lane takes a
// contiguous quarter of the parts), the four partial sums are combined in lane order.
template <int NQ>
__global__ __launch_bounds__(64 * NQ) void det_reduce_kernel(const float* __restrict__ parts, int nparts, int64_t stride, float* __restrict__ out,
                                                            int nvals, int accumulate, float* __restrict__ out2 = nullptr, int split = 0) {
    __shared__ double red[NQ][64];
    const int v = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int per = (nparts + NQ - 1) / NQ, p0 = q * per, p1 = min(nparts, p0 + per);
    double a = 0.0;
    if (v < nvals) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {           // eight loads in flight, added in part order
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = parts[(int64_t)(p + u) * stride + v];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)x[u];
        }
        for (; p < p1; ++p) a += (double)parts[(int64_t)p * stride + v];
    }
    red[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && v < nvals) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NQ; ++k) s += red[k][threadIdx.x];                 // lane order: fixed
        float* o = (out2 && v >= split) ? out2 + (v - split) : out + v;       // values >= split go to the second output (sum | sum of squares)
        *o = accumulate ? (float)((double)*o + s) : (float)s;
    }
}
}  // namespace

// ---- deferred reduces (r04): a launch whose partials only feed PARAMETER gradients (LayerNorm dgamma / dbeta: 42 launches of 7 us per train step) may park its
// partials in a caller-provided arena instead of reducing them at once; p3_reduce_flush adds all parked sets in ONE launch (grid.y = set), the same
// 16-lane fixed-order float64 sum per value as det_reduce_kernel<16>: bit-identical results.
namespace {
constexpr int DEF_MAX = 48;                 // sets per flush launch (kernel-argument table: 48 x 36 B)
constexpr int DEF_CAP = 6 * DEF_MAX;        // sets parked at once (r06: the bias-gradient column sums of the weight-gradient GEMMs park here too: ~140 per train step)
struct DefTable { const float* parts[DEF_MAX]; float* out[DEF_MAX]; float* out2[DEF_MAX]; int nparts[DEF_MAX]; int nvals[DEF_MAX]; int split[DEF_MAX]; };
struct DefAll { const float* parts[DEF_CAP]; float* out[DEF_CAP]; float* out2[DEF_CAP]; int nparts[DEF_CAP]; int nvals[DEF_CAP]; int split[DEF_CAP]; };
DefAll g_def;
float* g_def_arena = nullptr;
int64_t g_def_cap = 0, g_def_used = 0;
int g_def_n = 0, g_def_maxvals = 0, g_def_on = 0;

__global__ __launch_bounds__(1024) void det_reduce_many_kernel(DefTable t) {
    constexpr int NQ = 16;
    __shared__ double red[NQ][64];
    const int e = blockIdx.y, nvals = t.nvals[e], nparts = t.nparts[e];
    if ((int)blockIdx.x * 64 >= nvals) return;
    const float* parts = t.parts[e];
    const int64_t stride = nvals;
    const int v = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int per = (nparts + NQ - 1) / NQ, p0 = q * per, p1 = min(nparts, p0 + per);
    double a = 0.0;
    if (v < nvals) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = parts[(int64_t)(p + u) * stride + v];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += (double)x[u];
        }
        for (; p < p1; ++p) a += (double)parts[(int64_t)p * stride + v];
    }
    red[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && v < nvals) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NQ; ++k) s += red[k][threadIdx.x];
        float* o = v >= t.split[e] ? t.out2[e] + (v - t.split[e]) : t.out[e] + v;
        *o = (float)((double)*o + s);
    }
}
}  // namespace

extern "C" int p3_reduce_defer(float* arena, int64_t floats) {
    if (g_def_n != 0) { p3_set_error("p3_reduce_defer: parked reduces pending - call p3_reduce_flush first"); return P3_EINVAL; }
    g_def_arena = arena; g_def_cap = arena ? floats : 0; g_def_used = 0; g_def_maxvals = 0;
    return P3_OK;
}
extern "C" int p3_reduce_pending(void) { return g_def_n; }
extern "C" int p3_reduce_drop(void) { const int n = g_def_n; g_def_n = 0; g_def_used = 0; g_def_maxvals = 0; return n; }     // forget parked sets (a backward pass that raised)
extern "C" int p3_reduce_defer_enable(int on) { const int was = g_def_on; g_def_on = on ? 1 : 0; return was; }

// a slot of `floats` for the partials of one set (nparts x nvals, stride nvals; values [0, split) += out, the rest += out2), or NULL: the caller reduces now
float* p3_reduce_park(int64_t floats, int nparts, int nvals, int split, float* out, float* out2) {
    if (!g_def_on || !g_def_arena || g_def_n >= DEF_CAP || g_def_used + floats > g_def_cap || nparts > 16 * 128) return nullptr;
    // the flush adds every parked set from its own block row without atomics: two sets with a common target (a LayerNorm module applied twice in one forward,
    // a second backward before the flush) would race there - the later one reduces immediately instead (stream order serialises it behind... the flush adds
    // to whatever it finds, so the order of the two additions does not matter, only that they are not concurrent)
    for (int e = 0; e < g_def_n; ++e)
        if (g_def.out[e] == out || g_def.out2[e] == out || (out2 && (g_def.out[e] == out2 || g_def.out2[e] == out2))) return nullptr;
    float* slot = g_def_arena + g_def_used;
    g_def_used += (floats + 63) / 64 * 64;
    const int e = g_def_n++;
    g_def.parts[e] = slot; g_def.out[e] = out; g_def.out2[e] = out2; g_def.nparts[e] = nparts; g_def.nvals[e] = nvals; g_def.split[e] = split;
    if (nvals > g_def_maxvals) g_def_maxvals = nvals;
    return slot;
}

// [splits][N] partial column sums of a weight-gradient GEMM's bias gradient: parked for p3_reduce_flush when parking is on (the target is a parameter gradient that stays
// unread until then: FlatAdamW(direct_grad)) - *parked = 1, the caller launches no reduce - else the deterministic scratch and a p3_det_reduce right behind the launch
float* p3_colsum_parts(int splits, int N, float* colsum, int dtype, int* parked) {
    float* slot = p3_reduce_park((int64_t)splits * N, splits, N, N, colsum, nullptr);
    *parked = slot ? 1 : 0;
    return slot ? slot : p3_det_scratch((int64_t)splits * N, dtype);
}

extern "C" int p3_reduce_flush(void* stream) {
    if (g_def_n == 0) return P3_OK;
    for (int base = 0; base < g_def_n; base += DEF_MAX) {
        DefTable t;
        const int n = g_def_n - base < DEF_MAX ? g_def_n - base : DEF_MAX;
        int maxvals = 0;
        for (int i = 0; i < n; ++i) {
            t.parts[i] = g_def.parts[base + i]; t.out[i] = g_def.out[base + i]; t.out2[i] = g_def.out2[base + i];
            t.nparts[i] = g_def.nparts[base + i]; t.nvals[i] = g_def.nvals[base + i]; t.split[i] = g_def.split[base + i];
            if (t.nvals[i] > maxvals) maxvals = t.nvals[i];
        }
        hipLaunchKernelGGL(det_reduce_many_kernel, dim3((maxvals + 63) / 64, n), dim3(1024), 0, (hipStream_t)stream, t);
    }
    g_def_n = 0; g_def_used = 0; g_def_maxvals = 0;
    P3_LAUNCH_CHECK();
    return P3_OK;
}

int p3_det_reduce(const float* parts, int nparts, int64_t stride, float* out, int nvals, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(det_reduce_kernel<4>, dim3((nvals + 63) / 64), dim3(256), 0, s, parts, nparts, stride, out, nvals, accumulate, (float*)nullptr, 0);
    P3_LAUNCH_CHECK();
    return P3_OK;
}

// many parts (tens of thousands of tiles): chunks of 256 parts are summed by their own workgroups into `tmp` ([ceil(nparts / 256)][nvals]
// floats, caller-provided), then those chunk sums in chunk order - same bits every run, the whole chip at work
// values [0, split) are added to out, [split, nvals) to out2 (one pass for a (sum | sum of squares) pair); tmp: [ceil(nparts / 128)][nvals]
int p3_det_reduce2(const float* parts, int nparts, int64_t stride, float* tmp, float* out, float* out2, int split, int nvals, int accumulate, hipStream_t s) {
    constexpr int CH = 128;
    const float* src = parts;
    int n = nparts;
    int64_t st = stride;
    constexpr int one_level = 16 * CH;              // parts up to which one 16-lane pass does it (r03 A/B on the LayerNorm partials)
    if (nparts > one_level) {
        const int nch = (nparts + CH - 1) / CH;
        hipLaunchKernelGGL(det_reduce_chunk_kernel, dim3((nvals + 63) / 64, nch), dim3(256), 0, s, parts, nparts, stride, tmp, nvals, CH);
        P3_LAUNCH_CHECK();
        src = tmp; n = nch; st = nvals;
    }
    if (n > 64) hipLaunchKernelGGL(det_reduce_kernel<16>, dim3((nvals + 63) / 64), dim3(1024), 0, s, src, n, st, out, nvals, accumulate, out2, split);
    else hipLaunchKernelGGL(det_reduce_kernel<4>, dim3((nvals + 63) / 64), dim3(256), 0, s, src, n, st, out, nvals, accumulate, out2, split);
    P3_LAUNCH_CHECK();
    return P3_OK;
}
