// p3hip linear sum assignment (f-1: the CPU Hungarian of predictor_pix2poly.py:307-319 moved onto the GPU).
//
// The reference calls scipy.optimize.linear_sum_assignment(-scores[b]) per tile (scipy 1.15: rectangular_lsap, the shortest
// augmenting path algorithm of Crouse 2016 in float64).  This is the same algorithm, one 64-lane wave per tile:
//   * the dual variables, shortest-path costs and the `remaining` list live in LDS, and for N <= 195 so does the whole fp32 cost
//     matrix (4 N^2 + 48 N bytes <= 160 KB), so one step of the inner loop is LDS reads + one 6-stage butterfly, no global access;
//   * each lane owns the columns j = lane, lane + 64, ...; the sequential column scan of the CPU code becomes a lane-local scan
//     followed by a wave reduction whose ORDER KEY reproduces the CPU scan's tie rule exactly: among columns at the minimum an
//     unassigned column wins over assigned ones (the LAST unassigned one in `remaining` order), otherwise the FIRST in `remaining`
//     order; `remaining` is kept with the same swap-with-last removal, initialised in reverse order;
//   * all arithmetic is float64 in the CPU code's operation order (no multiplies: nothing for the compiler to contract),
// so the assignment is bit-identical to scipy's, ties included (tests/test_assignment_gpu.py: integer cost matrices, constant
// matrices, duplicated rows).  HBM traffic is the 4 N^2 bytes of the scores read once; the kernel is latency bound (a dependent
// chain of <= N^2 steps per tile), which is why it runs one wave per tile and tiles in parallel across the CUs.
#include "p3_common.h"

namespace {

struct Best { double val; int key; int j; };

__device__ __forceinline__ bool better(double v, int k, const Best& b) { return v < b.val || (v == b.val && k < b.key); }

template <bool COST_LDS>
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ scores, int N, int maximize, int32_t* __restrict__ col4row_out,
                                                  float* __restrict__ perm, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* v = reinterpret_cast<double*>(smem);
    double* spc = v + N;                       // shortestPathCosts
    double* u = spc + N;
    int* path = reinterpret_cast<int*>(u + N);
    int* row4col = path + N;
    int* pos = row4col + N;                    // position of column j in `remaining`, -1 once j is in SC
    int* remaining = pos + N;
    int* col4row = remaining + N;
    int* SR = col4row + N;
    float* cl = reinterpret_cast<float*>(SR + N);
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const float* S = scores + (int64_t)b * N * N;
    const float sgn = maximize ? -1.f : 1.f;   // cost = -scores (exact in fp32, as numpy's unary minus)

    int bad = 0;
    for (int e = lane; e < N * N; e += 64) {
        const float c = sgn * S[e];
        if (c != c || c == -INFINITY) bad = 1;  // scipy: "matrix contains invalid numeric entries"
        if (COST_LDS) cl[e] = c;
    }
    for (int j = lane; j < N; j += 64) { v[j] = 0.0; u[j] = 0.0; path[j] = -1; row4col[j] = -1; col4row[j] = -1; }
    bad = __any(bad);
    __syncthreads();
    int st = bad ? 1 : 0;

    for (int curRow = 0; curRow < N && st == 0; ++curRow) {
        for (int j = lane; j < N; j += 64) { pos[j] = N - 1 - j; remaining[N - 1 - j] = j; spc[j] = INFINITY; SR[j] = 0; }
        __syncthreads();
        int num_remaining = N, i = curRow, sink = -1;
        double minVal = 0.0;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            Best best; best.val = INFINITY; best.key = 0x7fffffff; best.j = -1;
            const float* crow = COST_LDS ? cl + i * N : S + (int64_t)i * N;
            for (int j = lane; j < N; j += 64) {
                const int p = pos[j];
                if (p < 0) continue;
                const double c = COST_LDS ? (double)crow[j] : (double)(sgn * crow[j]);
                const double r = ((minVal + c) - ui) - v[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const int key = row4col[j] == -1 ? -(p + 1) : (p + 1);
                if (better(s, key, best)) { best.val = s; best.key = key; best.j = j; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(best.val, o, 64);
                const int ok = __shfl_xor(best.key, o, 64);
                const int oj = __shfl_xor(best.j, o, 64);
                if (better(ov, ok, best)) { best.val = ov; best.key = ok; best.j = oj; }
            }
            minVal = best.val;
            if (minVal == INFINITY || best.j < 0) { st = 2; break; }   // infeasible cost matrix
            const int j = best.j;
            const int r4c = row4col[j];
            if (r4c == -1) sink = j; else i = r4c;
            if (lane == 0) {                   // SC[j] = true; remaining[index] = remaining[--num_remaining]
                const int idx = pos[j];
                const int last = remaining[num_remaining - 1];
                remaining[idx] = last;
                pos[last] = idx;
                pos[j] = -1;
            }
            --num_remaining;
            __syncthreads();
        }
        if (st != 0) break;
        // dual updates (col4row still the previous solution), then augment along the path
        for (int r = lane; r < N; r += 64) {
            if (r == curRow) u[r] += minVal;
            else if (SR[r]) u[r] += minVal - spc[col4row[r]];
        }
        for (int j = lane; j < N; j += 64)
            if (pos[j] < 0) v[j] -= minVal - spc[j];
        __syncthreads();
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int r = path[j];
                row4col[j] = r;
                const int t = col4row[r]; col4row[r] = j; j = t;
                if (r == curRow) break;
            }
        }
        __syncthreads();
    }

    if (lane == 0) status[b] = st;
    for (int r = lane; r < N; r += 64) col4row_out[(int64_t)b * N + r] = st == 0 ? col4row[r] : -1;
    if (perm) {
        float* P = perm + (int64_t)b * N * N;
        for (int r = 0; r < N; ++r) {
            const int c = st == 0 ? col4row[r] : -1;
            for (int j = lane; j < N; j += 64) P[(int64_t)r * N + j] = j == c ? 1.f : 0.f;
        }
    }
}

}  // namespace

extern "C" int p3_assignment(const float* scores, int B, int N, int maximize, int32_t* col4row, float* perm, int32_t* status, void* stream) {
    P3_CHECK(scores && col4row && status, P3_EINVAL, "p3_assignment: null pointer");
    P3_CHECK(B > 0 && N > 0, P3_ESHAPE, "p3_assignment: empty problem");
    const size_t aux = (size_t)N * 48;
    const size_t full = aux + (size_t)N * N * 4;
    const size_t lds_max = 160 * 1024;
    P3_CHECK(aux <= lds_max, P3_ESHAPE, "p3_assignment: N too large (dual variables must fit 160 KB of LDS)");
    hipStream_t s = (hipStream_t)stream;
    if (full <= lds_max) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsap_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
        hipLaunchKernelGGL((lsap_kernel<true>), dim3(B), dim3(64), full, s, scores, N, maximize, col4row, perm, status);
    } else {
        static bool attr_set = false;
        if (!attr_set && aux > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lsap_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
            if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
            attr_set = true;
        }
        hipLaunchKernelGGL((lsap_kernel<false>), dim3(B), dim3(64), aux, s, scores, N, maximize, col4row, perm, status);
    }
    P3_LAUNCH_CHECK();
    return P3_OK;
}
