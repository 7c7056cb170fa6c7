// p3hip LiDAR pillar stem: pillarize -> PillarFeatureNet (2 PFN layers, BatchNorm1d, max) -> scatter.
// Replaces PointPillarsEncoder.forward (models/pointpillars/pointpillars_o3d.py:85-107), i.e. Open3D-ML
// PointPillars.voxelize / PillarFeatureNet / PointPillarsScatter, including the reference's per-sample python loops.
//
// Pipeline (one launch each, all samples at once, no host sync):
//   1 pillar_sort      one workgroup per sample: stable counting sort of the points by pillar hash in LDS
//                      (16 per-wave histograms -> cross-wave prefix -> in-order fill), so that the first
//                      max_points of each pillar are the LOWEST point indices (Open3D keeps those), bit-exact.
//   2 pfn_l1_stats     (train) BatchNorm1d(32) batch statistics over all V*max_points slots (padded slots are zeros)
//   3 pfn_l1_apply     decorate (xyz, xyz - pillar mean, xy - pillar centre) -> Linear(8,32) -> BN -> ReLU -> max;
//                      writes the layer-2 input rows [x | xmax] for the real points + ONE representative padded slot
//   4 p3_gemm          H2 = X2 . W2^T   (MFMA; K = 64)
//   5 pfn_l2_reduce    per pillar max / min over its rows (+ BN statistics, padded slot weighted by its multiplicity)
//   6 bn_finalize      scale / shift (+ running statistics update)
//   7 pfn_scatter      relu(scale * (scale > 0 ? hmax : hmin) + shift) -> token-major canvas (empty pillars = 0)
#include <stdlib.h>

#include "p3_common.h"

namespace {

constexpr int C1 = 32;      // PFN layer-0 units (feat_channels[0] / 2)
constexpr int K2 = 64;      // layer-1 input = [x | xmax]
// pfn_l2_reduce / pfn_bwd_l2_stats grid cap (512 with per-wave atomics: 0.38 ms; see the kernel).  P3_PFN_BLOCKS: sweeps.
// r03 sweep (same box): reduce8 2048 -> 93 us, 1024 -> 77, 512 -> 99; bwd_l2_stats8 2048 -> 102 us, 1024 -> 83, 512 -> 70 (per-workgroup atomics on 2 C addresses)
#define L2R_BLOCKS 1024
#define L2S_BLOCKS 512
constexpr int SORT_THREADS = 1024, SORT_WAVES = 16;
constexpr int MAX_CELLS = 1900;
// layer-0 backward accumulators: S_dyf[32][8] | S_xf[32][8] | S_f[8] | dbeta[32] | dgamma[32]
constexpr int ACC_DYF = 0, ACC_XF = 256, ACC_F = 512, ACC_DB = 520, ACC_DG = 552, ACC1_FLOATS = 584;   // (16 + 5) * nc * 4 B of LDS must stay below 160 KB

struct PillarGeom {
    int nx, ny, ncx, ncy, nc;  // ncx = nx+1 (cell nx holds x == xmax), nc = ncx*ncy*2
    float invx, invy, invz, xmax, ymax, zmax, vx, vy;
};

__device__ __forceinline__ int cell_of(const PillarGeom& g, float x, float y, float z) {
    if (!(x >= 0.f && x <= g.xmax && y >= 0.f && y <= g.ymax && z >= 0.f && z <= g.zmax)) return -1;
    int cx = (int)(x * g.invx), cy = (int)(y * g.invy), cz = (int)(z * g.invz);
    return cx + cy * g.ncx + cz * g.ncx * g.ncy;
}

// exclusive scan of arr[0..n) in LDS by the whole 1024-thread block; returns total. tmp: >= 17 words
__device__ unsigned block_scan_excl(unsigned* arr, int n, unsigned* tmp) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = (n + SORT_THREADS - 1) / SORT_THREADS;
    unsigned loc[4];
    unsigned s = 0;
    for (int i = 0; i < per; ++i) { int idx = tid * per + i; loc[i] = idx < n ? arr[idx] : 0u; s += loc[i]; }
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) tmp[w] = incl;
    __syncthreads();
    if (tid == 0) { unsigned r = 0; for (int i = 0; i < SORT_WAVES; ++i) { unsigned t = tmp[i]; tmp[i] = r; r += t; } tmp[SORT_WAVES] = r; }
    __syncthreads();
    unsigned run = tmp[w] + incl - s;
    for (int i = 0; i < per; ++i) { int idx = tid * per + i; if (idx < n) arr[idx] = run; run += loc[i]; }
    unsigned total = tmp[SORT_WAVES];
    __syncthreads();
    return total;
}

// rank of this lane among the lanes of the wave that hold the same cell (earlier lanes only) and the number of such lanes: 64 scalar broadcasts + compares, no LDS
// (r05: the match loop this replaces - leader, shuffle, ballot, LDS update per DISTINCT cell of the chunk - ran ~60 dependent rounds per chunk on a dense cloud)
__device__ __forceinline__ void sort_rank_in_wave(int c, int lane, unsigned& rank, unsigned& same) {
    rank = 0; same = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const int cj = __builtin_amdgcn_readlane(c, j);
        const unsigned eq = (cj == c) ? 1u : 0u;
        same += eq;
        rank += (j < lane) ? eq : 0u;
    }
}

struct SortOut {
    int* sorted;      // [total_points] global point index, grouped by pillar hash, ascending index inside a pillar
    int* vox_xy;      // [B*MV] cy*nx+cx of kept pillar, bit 30 = "skip in scatter" (overwritten by the top-z pillar)
    int* vox_start;   // [B*MV] position of the pillar's first point in `sorted`
    int* vox_cnt;     // [B*MV] min(count, max_points)
    int* vox_row;     // [B*MV] first row in X2/H2
    int* nvox;        // [B]
    int* totals;      // [0] = kept pillars over the batch
};

__global__ __launch_bounds__(SORT_THREADS) void pillar_sort_kernel(const float* __restrict__ pts, const int64_t* __restrict__ offs,
                                                                   PillarGeom g, int max_points, int max_voxels, SortOut o) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    const int nc = g.nc;
    unsigned* hist = sm;                      // [16][nc] per-wave histograms -> per-wave bases
    unsigned* tot = sm + SORT_WAVES * nc;     // [nc] points per cell
    unsigned* start = tot + nc;               // [nc] first position in the sorted list
    unsigned* kept = start + nc;              // [nc] 1 if the pillar is emitted
    unsigned* slot = kept + nc;               // [nc] output slot of a kept pillar
    unsigned* rows = slot + nc;               // [nc] first X2 row (sample local)
    unsigned* tmp = rows + nc;                // [32]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t p0 = offs[b];
    const int n = (int)(offs[b + 1] - p0);
    for (int i = tid; i < SORT_WAVES * nc; i += SORT_THREADS) hist[i] = 0;
    __syncthreads();
    int chunk = (n + SORT_WAVES - 1) / SORT_WAVES;
    chunk = (chunk + 63) / 64 * 64;
    const int beg = w * chunk, end = min(n, beg + chunk);
    for (int i = beg + lane; i < end; i += 64) {
        const float* p = pts + 3 * (p0 + i);
        int c = cell_of(g, p[0], p[1], p[2]);
        if (c >= 0) atomicAdd(&hist[w * nc + c], 1u);
    }
    __syncthreads();
    for (int c = tid; c < nc; c += SORT_THREADS) {
        unsigned run = 0;
        for (int k = 0; k < SORT_WAVES; ++k) { unsigned t = hist[k * nc + c]; hist[k * nc + c] = run; run += t; }
        tot[c] = run; start[c] = run; slot[c] = run > 0 ? 1u : 0u;
    }
    __syncthreads();
    block_scan_excl(start, nc, tmp);
    block_scan_excl(slot, nc, tmp);   // rank among non-empty cells in hash order: max_voxels counts BEFORE the bounds filter
    for (int c = tid; c < nc; c += SORT_THREADS) {
        const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy;
        kept[c] = (tot[c] > 0 && slot[c] < (unsigned)max_voxels && cx < g.nx && cy < g.ny) ? 1u : 0u;
    }
    __syncthreads();
    for (int c = tid; c < nc; c += SORT_THREADS) {
        slot[c] = kept[c];
        const unsigned cnt = min(tot[c], (unsigned)max_points);
        rows[c] = kept[c] ? cnt + (cnt < (unsigned)max_points ? 1u : 0u) : 0u;
    }
    __syncthreads();
    const unsigned nkept = block_scan_excl(slot, nc, tmp);
    block_scan_excl(rows, nc, tmp);
    const int plane = g.ncx * g.ncy;
    for (int c = tid; c < nc; c += SORT_THREADS) {
        if (kept[c]) {
            const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy, cz = c / plane;
            // PointPillarsScatter writes pillars in order; a top-z pillar (z == zmax) at the same (x,y) comes later and wins
            const bool overwritten = (cz == 0) && kept[c + plane];
            const int idx = b * max_voxels + (int)slot[c];
            o.vox_xy[idx] = (cy * g.nx + cx) | (overwritten ? (1 << 30) : 0);
            o.vox_start[idx] = (int)(p0 + start[c]);
            o.vox_cnt[idx] = (int)min(tot[c], (unsigned)max_points);
            o.vox_row[idx] = (int)(p0 + (int64_t)b * max_voxels + rows[c]);
        }
    }
    if (tid == 0) { o.nvox[b] = (int)nkept; atomicAdd(o.totals, (int)nkept); }
    // ---- in-order fill: every wave walks its range again; rank inside a 64-point chunk = the earlier lanes of the chunk in the same cell ----
    for (int i0 = beg; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        int c = -1;
        if (i < end) { const float* p = pts + 3 * (p0 + i); c = cell_of(g, p[0], p[1], p[2]); }
        unsigned rank, same;
        sort_rank_in_wave(c, lane, rank, same);
        const unsigned base = c >= 0 ? hist[w * nc + c] : 0u;
        __builtin_amdgcn_wave_barrier();                       // every lane has read its cell's position before a cell's first lane moves it on
        if (c >= 0) {
            o.sorted[p0 + start[c] + base + rank] = (int)(p0 + i);
            if (rank == 0) hist[w * nc + c] = base + same;     // one lane per distinct cell of the chunk
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the same sort for DENSE clouds, a tile's points over SORT_SPLIT workgroups (r05: one workgroup per tile = 64 of 256 CUs was 0.54 ms at 40 k points per tile) --
// count: workgroup (k, b) histograms chunk k of tile b;  tables: one workgroup per tile scans the chunk histograms and writes the pillar tables exactly as
// pillar_sort_kernel does;  fill: workgroup (k, b) walks its chunk again in order.  Chunks are contiguous point ranges in ascending order and a chunk's waves take
// contiguous sub-ranges, so "ascending point index inside a pillar" (the <= max_points lowest indices survive) holds as in the one-workgroup form.
constexpr int SORT_SPLIT = 4;

__device__ __forceinline__ void sort_chunk_range(int n, int k, int w, int& beg, int& end) {
    int cw = (n + SORT_SPLIT - 1) / SORT_SPLIT;
    cw = (cw + SORT_THREADS - 1) / SORT_THREADS * SORT_THREADS;          // per-wave sub-ranges stay multiples of 64
    const int kb = min(n, k * cw), ke = min(n, kb + cw);
    int sub = (ke - kb + SORT_WAVES - 1) / SORT_WAVES;
    sub = (sub + 63) / 64 * 64;
    beg = min(ke, kb + w * sub); end = min(ke, beg + sub);
}

// per-wave histograms of the workgroup's chunk into hist[16][nc] (LDS), as step 1 of pillar_sort_kernel
__device__ __forceinline__ void sort_wave_hist(const float* __restrict__ pts, int64_t p0, const PillarGeom& g, unsigned* hist, int beg, int end) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nc = g.nc;
    for (int i = tid; i < SORT_WAVES * nc; i += SORT_THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = beg + lane; i < end; i += 64) {
        const float* p = pts + 3 * (p0 + i);
        const int c = cell_of(g, p[0], p[1], p[2]);
        if (c >= 0) atomicAdd(&hist[w * nc + c], 1u);
    }
    __syncthreads();
}

__global__ __launch_bounds__(SORT_THREADS) void pillar_sort_count_kernel(const float* __restrict__ pts, const int64_t* __restrict__ offs, PillarGeom g,
                                                                         unsigned* __restrict__ ghist /*[B][SORT_SPLIT][nc]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    const int nc = g.nc, k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int64_t p0 = offs[b];
    const int n = (int)(offs[b + 1] - p0);
    int beg, end;
    sort_chunk_range(n, k, tid >> 6, beg, end);
    sort_wave_hist(pts, p0, g, sm, beg, end);
    for (int c = tid; c < nc; c += SORT_THREADS) {
        unsigned run = 0;
        for (int q = 0; q < SORT_WAVES; ++q) run += sm[q * nc + c];
        ghist[((int64_t)b * SORT_SPLIT + k) * nc + c] = run;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void pillar_sort_tables_kernel(const int64_t* __restrict__ offs, PillarGeom g, int max_points, int max_voxels, SortOut o,
                                                                          unsigned* __restrict__ ghist /* in: chunk counts, out: chunk bases */,
                                                                          unsigned* __restrict__ gstart /*[B][nc]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    const int nc = g.nc;
    unsigned* tot = sm;                       // [nc] points per cell
    unsigned* start = tot + nc;               // [nc] first position in the sorted list
    unsigned* kept = start + nc;              // [nc] 1 if the pillar is emitted
    unsigned* slot = kept + nc;               // [nc] output slot of a kept pillar
    unsigned* rows = slot + nc;               // [nc] first X2 row (sample local)
    unsigned* tmp = rows + nc;                // [32]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t p0 = offs[b];
    for (int c = tid; c < nc; c += SORT_THREADS) {
        unsigned run = 0;
        for (int k = 0; k < SORT_SPLIT; ++k) { unsigned* h = ghist + ((int64_t)b * SORT_SPLIT + k) * nc + c; const unsigned t = *h; *h = run; run += t; }
        tot[c] = run; start[c] = run; slot[c] = run > 0 ? 1u : 0u;
    }
    __syncthreads();
    block_scan_excl(start, nc, tmp);
    block_scan_excl(slot, nc, tmp);   // rank among non-empty cells in hash order: max_voxels counts BEFORE the bounds filter
    for (int c = tid; c < nc; c += SORT_THREADS) {
        const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy;
        kept[c] = (tot[c] > 0 && slot[c] < (unsigned)max_voxels && cx < g.nx && cy < g.ny) ? 1u : 0u;
        gstart[(int64_t)b * nc + c] = start[c];
    }
    __syncthreads();
    for (int c = tid; c < nc; c += SORT_THREADS) {
        slot[c] = kept[c];
        const unsigned cnt = min(tot[c], (unsigned)max_points);
        rows[c] = kept[c] ? cnt + (cnt < (unsigned)max_points ? 1u : 0u) : 0u;
    }
    __syncthreads();
    const unsigned nkept = block_scan_excl(slot, nc, tmp);
    block_scan_excl(rows, nc, tmp);
    const int plane = g.ncx * g.ncy;
    for (int c = tid; c < nc; c += SORT_THREADS) {
        if (kept[c]) {
            const int cx = c % g.ncx, cy = (c / g.ncx) % g.ncy, cz = c / plane;
            const bool overwritten = (cz == 0) && kept[c + plane];      // see pillar_sort_kernel
            const int idx = b * max_voxels + (int)slot[c];
            o.vox_xy[idx] = (cy * g.nx + cx) | (overwritten ? (1 << 30) : 0);
            o.vox_start[idx] = (int)(p0 + start[c]);
            o.vox_cnt[idx] = (int)min(tot[c], (unsigned)max_points);
            o.vox_row[idx] = (int)(p0 + (int64_t)b * max_voxels + rows[c]);
        }
    }
    if (tid == 0) { o.nvox[b] = (int)nkept; atomicAdd(o.totals, (int)nkept); }
}

__global__ __launch_bounds__(SORT_THREADS) void pillar_sort_fill_kernel(const float* __restrict__ pts, const int64_t* __restrict__ offs, PillarGeom g,
                                                                        const unsigned* __restrict__ gbase /*[B][SORT_SPLIT][nc]*/,
                                                                        const unsigned* __restrict__ gstart /*[B][nc]*/, int* __restrict__ sorted) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    const int nc = g.nc, k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned* hist = sm;                      // [16][nc] per-wave histograms -> positions of the wave's next point of a cell
    const int64_t p0 = offs[b];
    const int n = (int)(offs[b + 1] - p0);
    int beg, end;
    sort_chunk_range(n, k, w, beg, end);
    sort_wave_hist(pts, p0, g, hist, beg, end);
    for (int c = tid; c < nc; c += SORT_THREADS) {
        unsigned run = gstart[(int64_t)b * nc + c] + gbase[((int64_t)b * SORT_SPLIT + k) * nc + c];
        for (int q = 0; q < SORT_WAVES; ++q) { const unsigned t = hist[q * nc + c]; hist[q * nc + c] = run; run += t; }
    }
    __syncthreads();
    // in-order fill: every wave walks its range again (as pillar_sort_kernel)
    for (int i0 = beg; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        int c = -1;
        if (i < end) { const float* p = pts + 3 * (p0 + i); c = cell_of(g, p[0], p[1], p[2]); }
        unsigned rank, same;
        sort_rank_in_wave(c, lane, rank, same);
        const unsigned base = c >= 0 ? hist[w * nc + c] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (c >= 0) {
            sorted[p0 + base + rank] = (int)(p0 + i);
            if (rank == 0) hist[w * nc + c] = base + same;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// PFN layer 0
// ------------------------------------------------------------------------------------------------
struct VoxTab { const int* sorted; const int* xy; const int* start; const int* cnt; const int* row; const int* nvox; };

// decorated features of slot `lane` of a pillar (valid iff lane < cnt); h = W1 . f
// pillar mean over ALL its slots (max_points > 64: several 64-slot chunks per wave)
__device__ __forceinline__ void pfn_mean(const float* __restrict__ pts, const VoxTab& t, int v, int lane, float& mx, float& my, float& mz) {
    const int cnt = t.cnt[v];
    float x = 0.f, y = 0.f, z = 0.f;
    for (int s = lane; s < cnt; s += 64) { const float* p = pts + 3 * (int64_t)t.sorted[t.start[v] + s]; x += p[0]; y += p[1]; z += p[2]; }
    mx = wave_sum(x) / (float)cnt; my = wave_sum(y) / (float)cnt; mz = wave_sum(z) / (float)cnt;
}

// MULTI = false: slot == lane (max_points <= 64), the mean is taken here.  MULTI = true: slot = s0 + lane, mean passed in (pfn_mean).
template <bool MULTI = false>
__device__ __forceinline__ void pfn_l0(const float* __restrict__ pts, const VoxTab& t, int v, int lane, const PillarGeom& g,
                                       const float* __restrict__ w1s, float (&h)[C1], int& cnt_out, bool& valid_out, float (&fo)[8],
                                       int s0 = 0, float gmx = 0.f, float gmy = 0.f, float gmz = 0.f) {
    const int cnt = t.cnt[v];
    const bool valid = s0 + lane < cnt;
    float x = 0.f, y = 0.f, z = 0.f;
    if (valid) { const float* p = pts + 3 * (int64_t)t.sorted[t.start[v] + s0 + lane]; x = p[0]; y = p[1]; z = p[2]; }
    // same operation order as the reference: sum over slots, then divide
    float mx, my, mz;
    if constexpr (MULTI) { mx = gmx; my = gmy; mz = gmz; }
    else { mx = wave_sum(x) / (float)cnt; my = wave_sum(y) / (float)cnt; mz = wave_sum(z) / (float)cnt; }
    const int xy = t.xy[v] & 0xffffff;
    const int cx = xy % g.nx, cy = xy / g.nx;
    float f[8] = {x, y, z, x - mx, y - my, z - mz, x - ((float)cx * g.vx + 0.5f * g.vx), y - ((float)cy * g.vy + 0.5f * g.vy)};
#pragma unroll
    for (int c = 0; c < C1; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += w1s[c * 8 + k] * f[k];
        h[c] = valid ? a : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) fo[k] = valid ? f[k] : 0.f;
    cnt_out = cnt; valid_out = valid;
}


// ---- quad layout for max_points <= 64 ------------------------------------------------------------------------------------------
// A wave per pillar with lane == slot leaves most lanes idle (3 k points over 784 pillars: ~4 points per pillar) and makes every lane
// evaluate all 32 layer-0 channels from LDS-resident weights (256 LDS reads + 256 FMAs + 32 full-wave max reductions per pillar; r01:
// 0.58 ms for pfn_l1_apply).  Here a lane is (slot s = lane & 15, channel group cg = lane >> 4): it keeps the 8 x 8 weights of its 8
// channels in registers, a pass covers 16 slots (one pass for almost every pillar) and the per-channel max is a 16-lane butterfly.
// The point loads and the pillar mean stay in the 64-lane layout (one load per slot, same summation order as before), the slot's
// coordinates reach its four lanes by shuffle.  Same per-element arithmetic as pfn_l0: identical h, xmax and rows.
struct QuadCtx { float w[8][8]; float sc[8], sh[8]; };

__device__ __forceinline__ void quad_init(QuadCtx& q, const float* __restrict__ w1s, const float* __restrict__ ss, int cg) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int k = 0; k < 8; ++k) q.w[c][k] = w1s[(cg * 8 + c) * 8 + k];
        q.sc[c] = ss ? ss[cg * 8 + c] : 1.f;
        q.sh[c] = ss ? ss[C1 + cg * 8 + c] : 0.f;
    }
}

struct QuadPillar { float x, y, z, mx, my, mz, ccx, ccy; int cnt; };

__device__ __forceinline__ QuadPillar quad_pillar(const float* __restrict__ pts, const VoxTab& t, int v, int lane, const PillarGeom& g) {
    QuadPillar p;
    p.cnt = t.cnt[v];
    p.x = 0.f; p.y = 0.f; p.z = 0.f;
    if (lane < p.cnt) { const float* q = pts + 3 * (int64_t)t.sorted[t.start[v] + lane]; p.x = q[0]; p.y = q[1]; p.z = q[2]; }
    p.mx = wave_sum(p.x) / (float)p.cnt; p.my = wave_sum(p.y) / (float)p.cnt; p.mz = wave_sum(p.z) / (float)p.cnt;   // sum over slots, then divide
    const int xy = t.xy[v] & 0xffffff;
    const int cx = xy % g.nx, cy = xy / g.nx;
    p.ccx = (float)cx * g.vx + 0.5f * g.vx; p.ccy = (float)cy * g.vy + 0.5f * g.vy;
    return p;
}

// layer-0 pre-activation of this lane's 8 channels for slot `slot` (0 for padded slots) and the slot's decorated features
__device__ __forceinline__ void quad_l0(const QuadCtx& q, const QuadPillar& p, int slot, bool valid, float (&h)[8], float (&f)[8]) {
    const float x = __shfl(p.x, slot & 63, 64), y = __shfl(p.y, slot & 63, 64), z = __shfl(p.z, slot & 63, 64);
    const float ff[8] = {x, y, z, x - p.mx, y - p.my, z - p.mz, x - p.ccx, y - p.ccy};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += q.w[c][k] * ff[k];
        h[c] = valid ? a : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = valid ? ff[k] : 0.f;
}

__device__ __forceinline__ float quad_max16(float v) {      // over the 16 slots of a pass (lanes that share cg)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float quad_sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool MULTI>
__global__ __launch_bounds__(256) void pfn_l1_stats_kernel(const float* __restrict__ pts, VoxTab t, PillarGeom g, int max_voxels,
                                                           int nslots, const float* __restrict__ w1, float* __restrict__ sums /*[2*C1]*/, float* __restrict__ slab) {
    __shared__ float w1s[C1 * 8];
    for (int i = threadIdx.x; i < C1 * 8; i += 256) w1s[i] = w1[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int qs = lane & 15, qcg = lane >> 4;
    QuadCtx qc;
    if constexpr (!MULTI) quad_init(qc, w1s, nullptr, qcg);
    float s1[MULTI ? C1 : 8], s2[MULTI ? C1 : 8];
#pragma unroll
    for (int c = 0; c < (MULTI ? C1 : 8); ++c) { s1[c] = 0.f; s2[c] = 0.f; }
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nslots; v += gridDim.x * 4) {
        if ((v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        if constexpr (MULTI) {
            float h[C1]; int cnt; bool valid; float f8[8];
            float mx, my, mz;
            pfn_mean(pts, t, v, lane, mx, my, mz);
            const int cn = t.cnt[v];
            for (int s0 = 0; s0 < cn; s0 += 64) {
                pfn_l0<true>(pts, t, v, lane, g, w1s, h, cnt, valid, f8, s0, mx, my, mz);
#pragma unroll
                for (int c = 0; c < C1; ++c) { s1[c] += h[c]; s2[c] += h[c] * h[c]; }
            }
        } else {
            const QuadPillar qp = quad_pillar(pts, t, v, lane, g);
            for (int s0 = 0; s0 < qp.cnt; s0 += 16) {
                float hq[8], fq[8];
                quad_l0(qc, qp, s0 + qs, s0 + qs < qp.cnt, hq, fq);
#pragma unroll
                for (int c = 0; c < 8; ++c) { s1[c] += hq[c]; s2[c] += hq[c] * hq[c]; }
            }
        }
    }
    // block-level reduction first: one atomic per channel per block (8k waves hammering 64 addresses cost 3 ms)
    __shared__ float red[4][2 * C1];
    if constexpr (MULTI) {
#pragma unroll
        for (int c = 0; c < C1; ++c) {
            const float a = wave_sum(s1[c]), b2 = wave_sum(s2[c]);
            if (lane == 0) { red[threadIdx.x >> 6][c] = a; red[threadIdx.x >> 6][C1 + c] = b2; }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float a = quad_sum16(s1[c]), b2 = quad_sum16(s2[c]);
            if (qs == 0) { red[threadIdx.x >> 6][qcg * 8 + c] = a; red[threadIdx.x >> 6][C1 + qcg * 8 + c] = b2; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * C1) p3_commit(sums, slab, 2 * C1, threadIdx.x, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// scale/shift from batch statistics (train) or running statistics (eval); updates running stats like torch BatchNorm
__global__ void bn_finalize_kernel(const float* __restrict__ sums, int C, const int* __restrict__ count_ptr, float count_mul,
                                   float count_fixed, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float eps, float momentum, int training,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mean, var;
    if (training) {
        const float n = count_ptr ? (float)(*count_ptr) * count_mul : count_fixed;
        mean = sums[c] / n;
        var = fmaxf(sums[C + c] / n - mean * mean, 0.f);
        if (rmean) {
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
        }
    } else {
        mean = rmean[c]; var = rvar[c];
    }
    const float rstd = rsqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    scale[c] = sc; shift[c] = beta[c] - mean * sc;
    if (save_mean) { save_mean[c] = mean; save_rstd[c] = rstd; }
}

// one X2 row [x (32) | xmax (32)] + its 8 decorated features with 16-byte stores (64 two-byte stores per lane before)
template <typename T>
__device__ __forceinline__ void store_row(float* __restrict__ f8dst, T* __restrict__ dst, const float (&f8)[8], const float (&h)[C1], const float (&xm)[C1]) {
    *reinterpret_cast<float4*>(f8dst) = make_float4(f8[0], f8[1], f8[2], f8[3]);
    *reinterpret_cast<float4*>(f8dst + 4) = make_float4(f8[4], f8[5], f8[6], f8[7]);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<uint4*>(dst + 8 * q) = make_uint4(pack_bf2(h[8 * q], h[8 * q + 1]), pack_bf2(h[8 * q + 2], h[8 * q + 3]),
                                                                pack_bf2(h[8 * q + 4], h[8 * q + 5]), pack_bf2(h[8 * q + 6], h[8 * q + 7]));
            *reinterpret_cast<uint4*>(dst + C1 + 8 * q) = make_uint4(pack_bf2(xm[8 * q], xm[8 * q + 1]), pack_bf2(xm[8 * q + 2], xm[8 * q + 3]),
                                                                     pack_bf2(xm[8 * q + 4], xm[8 * q + 5]), pack_bf2(xm[8 * q + 6], xm[8 * q + 7]));
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
            *reinterpret_cast<float4*>(dst + C1 + 4 * q) = make_float4(xm[4 * q], xm[4 * q + 1], xm[4 * q + 2], xm[4 * q + 3]);
        }
    }
}

template <typename T, bool MULTI>
__global__ __launch_bounds__(256) void pfn_l1_apply_kernel(const float* __restrict__ pts, VoxTab t, PillarGeom g, int max_voxels,
                                                           int max_points, int nslots, const float* __restrict__ w1,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           T* __restrict__ X2, float* __restrict__ F8, int* __restrict__ row_vox,
                                                           float* __restrict__ row_w) {
    __shared__ float w1s[C1 * 8];
    __shared__ float ss[2 * C1];
    for (int i = threadIdx.x; i < C1 * 8; i += 256) w1s[i] = w1[i];
    if (threadIdx.x < C1) { ss[threadIdx.x] = scale[threadIdx.x]; ss[C1 + threadIdx.x] = shift[threadIdx.x]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int qs = lane & 15, qcg = lane >> 4;
    QuadCtx qc;
    if constexpr (!MULTI) quad_init(qc, w1s, ss, qcg);
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nslots; v += gridDim.x * 4) {
        if ((v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        if constexpr (MULTI) {
            float h[C1]; int cnt; bool valid; float f8[8];
            // max_points > 64: pass 1 = per-channel max over all 64-slot chunks, pass 2 = recompute and write the rows
            float mx, my, mz;
            pfn_mean(pts, t, v, lane, mx, my, mz);
            const int cn = t.cnt[v];
            const bool pad = cn < max_points;
            float xm[C1];
#pragma unroll
            for (int c = 0; c < C1; ++c) xm[c] = pad ? fmaxf(ss[C1 + c], 0.f) : -INFINITY;
            for (int s0 = 0; s0 < cn; s0 += 64) {
                pfn_l0<true>(pts, t, v, lane, g, w1s, h, cnt, valid, f8, s0, mx, my, mz);
#pragma unroll
                for (int c = 0; c < C1; ++c) xm[c] = fmaxf(xm[c], wave_max(valid ? fmaxf(h[c] * ss[c] + ss[C1 + c], 0.f) : -INFINITY));
            }
            for (int s0 = 0; s0 <= cn; s0 += 64) {     // <= : the chunk that holds the representative padded slot (index cn)
                pfn_l0<true>(pts, t, v, lane, g, w1s, h, cnt, valid, f8, s0, mx, my, mz);
                const int slot = s0 + lane;
                if (valid || (pad && slot == cn)) {
                    const int64_t row = (int64_t)t.row[v] + slot;
                    row_vox[row] = v; row_w[row] = valid ? 1.f : (float)(max_points - cn);
#pragma unroll
                    for (int c = 0; c < C1; ++c) h[c] = fmaxf(h[c] * ss[c] + ss[C1 + c], 0.f);
                    store_row<T>(F8 + row * 8, X2 + row * K2, f8, h, xm);
                }
            }
            continue;
        }
        // quad layout: lane = (slot s0 + qs, channels 8 qcg .. 8 qcg + 7)
        const QuadPillar qp = quad_pillar(pts, t, v, lane, g);
        const int cn = qp.cnt;
        const bool has_pad = cn < max_points;
        const int last = has_pad ? cn : cn - 1;                  // highest slot that owns a row (the representative padded slot is cn)
        float xq[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xq[c] = -INFINITY;
        for (int s0 = 0; s0 < cn; s0 += 16) {                    // pass 1: per-channel max over the real slots
            float hq[8], fq[8];
            const bool ok = s0 + qs < cn;
            quad_l0(qc, qp, s0 + qs, ok, hq, fq);
#pragma unroll
            for (int c = 0; c < 8; ++c) xq[c] = fmaxf(xq[c], ok ? fmaxf(hq[c] * qc.sc[c] + qc.sh[c], 0.f) : -INFINITY);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            xq[c] = quad_max16(xq[c]);
            if (has_pad) xq[c] = fmaxf(xq[c], fmaxf(qc.sh[c], 0.f));     // padded slots: relu(shift)
        }
        for (int s0 = 0; s0 <= last; s0 += 16) {                 // pass 2: rows
            float hq[8], fq[8];
            const int slot = s0 + qs;
            const bool ok = slot < cn;
            quad_l0(qc, qp, slot, ok, hq, fq);
            if (ok || (has_pad && slot == cn)) {
                const int64_t row = (int64_t)t.row[v] + slot;
#pragma unroll
                for (int c = 0; c < 8; ++c) hq[c] = fmaxf(hq[c] * qc.sc[c] + qc.sh[c], 0.f);
                if (qcg == 0) {
                    // training path: per-row pillar id / BN weight / decorated features (padded representative: zeros, weight P - cnt)
                    row_vox[row] = v; row_w[row] = ok ? 1.f : (float)(max_points - cn);
                    *reinterpret_cast<float4*>(F8 + row * 8) = make_float4(fq[0], fq[1], fq[2], fq[3]);
                    *reinterpret_cast<float4*>(F8 + row * 8 + 4) = make_float4(fq[4], fq[5], fq[6], fq[7]);
                }
                T* dst = X2 + row * K2 + qcg * 8;
                if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf2(hq[0], hq[1]), pack_bf2(hq[2], hq[3]), pack_bf2(hq[4], hq[5]), pack_bf2(hq[6], hq[7]));
                    *reinterpret_cast<uint4*>(dst + C1) = make_uint4(pack_bf2(xq[0], xq[1]), pack_bf2(xq[2], xq[3]), pack_bf2(xq[4], xq[5]), pack_bf2(xq[6], xq[7]));
                } else {
                    *reinterpret_cast<float4*>(dst) = make_float4(hq[0], hq[1], hq[2], hq[3]);
                    *reinterpret_cast<float4*>(dst + 4) = make_float4(hq[4], hq[5], hq[6], hq[7]);
                    *reinterpret_cast<float4*>(dst + C1) = make_float4(xq[0], xq[1], xq[2], xq[3]);
                    *reinterpret_cast<float4*>(dst + C1 + 4) = make_float4(xq[4], xq[5], xq[6], xq[7]);
                }
            }
        }
    }
}

// per pillar: max / min over its H2 rows and BN statistics (padded representative weighted by its multiplicity)
template <typename T>
__global__ __launch_bounds__(256) void pfn_l2_reduce_kernel(const T* __restrict__ H2, VoxTab t, int max_voxels, int max_points, int nslots,
                                                            int C, float* __restrict__ hmax, float* __restrict__ hmin,
                                                            float* __restrict__ sums /*[2C] or null*/, float* __restrict__ slab = nullptr) {
    const int lane = threadIdx.x & 63;
    constexpr int MAXJ = 12;
    float s1[MAXJ], s2[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int nj = C / 64;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nslots; v += gridDim.x * 4) {
        if ((v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        const int cnt = t.cnt[v];
        const int nrow = cnt + (cnt < max_points ? 1 : 0);
        const T* base = H2 + (int64_t)t.row[v] * C;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (j < nj) {
                float mx = -INFINITY, mn = INFINITY;
                for (int r = 0; r < nrow; ++r) {
                    const float val = Cvt<T>::to_f(base[(int64_t)r * C + lane + 64 * j]);
                    const float wgt = r < cnt ? 1.f : (float)(max_points - cnt);
                    mx = fmaxf(mx, val); mn = fminf(mn, val);
                    s1[j] += wgt * val; s2[j] += wgt * val * val;
                }
                hmax[(int64_t)v * C + lane + 64 * j] = mx;
                hmin[(int64_t)v * C + lane + 64 * j] = mn;
            }
        }
    }
    if (sums) {
        // block-level fold first (4 waves -> 1): the grid can then be 4x larger (8 waves / SIMD for this latency-bound walk) at the same
        // number of same-address atomics
        __shared__ float red[4][2 * 64 * MAXJ];
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j)
            if (j < nj) { red[wv][lane + 64 * j] = s1[j]; red[wv][64 * MAXJ + lane + 64 * j] = s2[j]; }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            p3_commit(sums, slab, 2 * C, c, (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
            p3_commit(sums, slab, 2 * C, C + c, (red[0][64 * MAXJ + c] + red[1][64 * MAXJ + c]) + (red[2][64 * MAXJ + c] + red[3][64 * MAXJ + c]));
        }
    }
}

#ifndef P3_PFN_NR
#define P3_PFN_NR 8       // rows of a pillar in flight per lane group in the rows8 kernels
#endif
// eight consecutive channels of one row: ONE 16-byte access in bf16, two in fp32
template <typename T> struct Row8;
template <> struct Row8<bf16_t> {
    uint4 raw;
    __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float (&v)[8]) const {
        const uint32_t wd[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (k & 1) ? __uint_as_float(wd[k >> 1] & 0xffff0000u) : __uint_as_float(wd[k >> 1] << 16);
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
        *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    }
};
template <> struct Row8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void get(float (&v)[8]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
    static __device__ __forceinline__ void store(float* p, const float (&o)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]); *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
};

// C % 8 == 0, C <= 512: a lane owns 8 consecutive channels (ONE 16-byte load per row in bf16, two in fp32, instead of C / 64 scalar loads), four rows in
// flight; same outputs as pfn_l2_reduce_kernel.  (r02: 241 us in the column-group form, whose row loop was a chain of dependent loads; r05: the fp32 family
// used to stay on that form - 2.05 ms at 40 k points per tile against this one's 0.50 in bf16 - and takes this kernel too.)
template <typename T>
__global__ __launch_bounds__(256) void pfn_l2_reduce8_kernel(const T* __restrict__ H2, VoxTab t, int max_voxels, int max_points, int nslots,
                                                             int C, float* __restrict__ hmax, float* __restrict__ hmin,
                                                             float* __restrict__ sums /*[2C] or null*/, float* __restrict__ slab = nullptr) {
    // a row takes C / 8 lanes: the wave walks G = 64 / (C / 8) pillars at once, one per lane group (C = 128: 4 groups of 16 lanes) - no cross-lane traffic, and G x
    // the rows in flight (r05: with one pillar per wave 48 of the 64 lanes sat idle at C = 128 and the launch ran at 1.5 TB/s at 40 k points per tile)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int lpr = C >> 3, G = 64 / lpr, grp = lane / lpr, cl = lane - grp * lpr;
    const bool act = grp < G;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
    for (int vb = (blockIdx.x * 4 + wv) * G; vb < nslots; vb += gridDim.x * 4 * G) {
        const int v = vb + grp;
        if (!act || v >= nslots || (v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        const int cnt = t.cnt[v];
        const int nrow = cnt + (cnt < max_points ? 1 : 0);
        const T* base = H2 + (int64_t)t.row[v] * C + cl * 8;
        float mx[8], mn[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { mx[k] = -INFINITY; mn[k] = INFINITY; }
        {
            for (int r0 = 0; r0 < nrow; r0 += P3_PFN_NR) {
                Row8<T> raw[P3_PFN_NR];
#pragma unroll
                for (int q = 0; q < P3_PFN_NR; ++q) { if (r0 + q < nrow) raw[q].load(base + (int64_t)(r0 + q) * C); else raw[q].zero(); }
#pragma unroll
                for (int q = 0; q < P3_PFN_NR; ++q) {
                    if (r0 + q >= nrow) break;
                    const float wgt = r0 + q < cnt ? 1.f : (float)(max_points - cnt);
                    float vals[8];
                    raw[q].get(vals);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float val = vals[k];
                        mx[k] = fmaxf(mx[k], val); mn[k] = fminf(mn[k], val);
                        s1[k] += wgt * val; s2[k] += wgt * val * val;
                    }
                }
            }
            float* hx = hmax + (int64_t)v * C + cl * 8;
            float* hn = hmin + (int64_t)v * C + cl * 8;
            *reinterpret_cast<float4*>(hx) = make_float4(mx[0], mx[1], mx[2], mx[3]); *reinterpret_cast<float4*>(hx + 4) = make_float4(mx[4], mx[5], mx[6], mx[7]);
            *reinterpret_cast<float4*>(hn) = make_float4(mn[0], mn[1], mn[2], mn[3]); *reinterpret_cast<float4*>(hn + 4) = make_float4(mn[4], mn[5], mn[6], mn[7]);
        }
    }
    if (sums) {
        __shared__ float red[4][2 * 512];               // [wave][lane group x C | the same for the squares]: lane * 8 = grp * C + cl * 8
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[wv][lane * 8 + k] = act ? s1[k] : 0.f; red[wv][512 + lane * 8 + k] = act ? s2[k] : 0.f; }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float a1 = 0.f, a2 = 0.f;
            for (int gg = 0; gg < G; ++gg) {               // fixed order: groups, then waves
                a1 += (red[0][gg * C + c] + red[1][gg * C + c]) + (red[2][gg * C + c] + red[3][gg * C + c]);
                a2 += (red[0][512 + gg * C + c] + red[1][512 + gg * C + c]) + (red[2][512 + gg * C + c] + red[3][512 + gg * C + c]);
            }
            p3_commit(sums, slab, 2 * C, c, a1);
            p3_commit(sums, slab, 2 * C, C + c, a2);
        }
    }
}

// ---- PFN layer 1 in ONE launch for DENSE clouds (r05; C = 32 NBW WPP: 384 = 3 x 4 - the path's width - and 128 = 2 x 2; bf16 or fp32 rows with the products as
// bf16 x 3): H2 = X2 W2^T, the per-pillar max / min over its rows and the weighted BatchNorm sums - what p3_gemm + pfn_l2_reduce8_kernel do in two passes over a
// [rows, C] matrix (C = 384 at 40 k points per tile: 4 GB written, then read).  The weight-stationary streaming form of rows_x3.hip with the pillar as the unit: a
// wave owns 32 NBW output channels (its slice of W2 as MFMA B fragments in registers: 32 NBW VGPRs for hi + lo) and one pillar at a time, 32 rows per MFMA group;
// the WPP waves of a group take the same pillar (the later reads of its rows are L1 / L2 hits).  In the accumulator layout a lane holds one channel and 16 rows:
// max / min / sums are per-lane operations, the two row halves meet in one exchange per pillar - no atomics.  STORE: also write H2 (the backward reads it); without
// it (no backward follows) the matrix never exists.  bf16: the value is rounded to bf16 BEFORE the max, as the two-pass form took it from the stored matrix - the
// backward finds the arg-max row by equality with the stored value.
// Measured (40 k points per tile, ~51 rows per pillar, training): 1.46 ms + 0.26 ms (zeroing the rows of no pillar) against 1.51 + 0.92 ms for the two passes; at
// 3 k points per tile (~4 rows per pillar) a pillar fills an eighth of an MFMA group and pays its own latency chain: 0.39 against 0.27 ms - the host therefore takes
// this path from 16 points per pillar slot on.  (Also measured and dropped: walking 32-row groups of the whole row range with per-lane running (pillar, max, min)
// handed over by atomic max / min - full MFMA groups at any density, but ~100 M atomic lane-operations per launch at 40 k: 2.46 ms.)
template <typename T, bool STORE, int NBW, int WPP>
__global__ __launch_bounds__(256, 2) void pfn_l2_fused_kernel(const T* __restrict__ X2, const T* __restrict__ W2, VoxTab t, int max_voxels, int max_points,
                                                              int nslots, T* __restrict__ H2, float* __restrict__ hmax, float* __restrict__ hmin,
                                                              float* __restrict__ sums /*[2 C] or null*/, float* __restrict__ slab) {
    constexpr int C = 32 * NBW * WPP, NS = 4 / WPP;      // NS: pillars in flight per workgroup
    constexpr bool X3 = sizeof(T) == 4;
    constexpr int XR = X3 ? 8 : 4;                       // 16-byte registers of one lane's share of a 32 x 64 row group
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const int cg = w % WPP, stream = w / WPP, ch0 = cg * 32 * NBW + l31;      // this lane's channels: ch0 + 32 nb
    auto split8 = [](const float (&v)[8], u32x4_t& h, u32x4_t& l) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t hw = pack_bf2(v[2 * k], v[2 * k + 1]);
            h[k] = hw;
            l[k] = pack_bf2(v[2 * k] - __uint_as_float(hw << 16), v[2 * k + 1] - __uint_as_float(hw & 0xffff0000u));
        }
    };
    // W2 rows (channels) ch0 + 32 nb, k = 16 s + 8 hi .. + 8
    bf16x8_t wh[NBW][4], wl[X3 ? NBW : 1][X3 ? 4 : 1];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            const T* wp = W2 + (int64_t)(ch0 + nb * 32) * K2 + 16 * sk + 8 * hi;
            if constexpr (X3) {
                const float4 x0 = *reinterpret_cast<const float4*>(wp), x1 = *reinterpret_cast<const float4*>(wp + 4);
                const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                u32x4_t h, l;
                split8(v, h, l);
                wh[nb][sk] = __builtin_bit_cast(bf16x8_t, h); wl[nb][sk] = __builtin_bit_cast(bf16x8_t, l);
            } else {
                wh[nb][sk] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wp));
            }
        }
    float s1[NBW], s2[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) { s1[nb] = 0.f; s2[nb] = 0.f; }
    // pillars of this stream: v, v + vstride, ... (slots beyond a tile's kept pillars are skipped).  The NEXT pillar's table entries are loaded while this one is
    // multiplied, and its first row group is fetched between the last group's products and epilogue: no pillar starts with an exposed chain of dependent loads
    const int vstride = gridDim.x * NS;
    auto next_valid = [&](int v) __attribute__((always_inline)) {
        while (v < nslots && (v % max_voxels) >= t.nvox[v / max_voxels]) v += vstride;
        return v;
    };
    u32x4_t xa[XR];
    auto fetch = [&](int64_t row0, int nrow, int g0) __attribute__((always_inline)) {
        const int rl = min(g0 + l31, nrow - 1);                       // rows beyond the pillar: a valid address, masked in the epilogue
        const T* ap = X2 + (row0 + rl) * K2 + 8 * hi;
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) {
            if constexpr (X3) { xa[2 * sk] = *reinterpret_cast<const u32x4_t*>(ap + 16 * sk); xa[2 * sk + 1] = *reinterpret_cast<const u32x4_t*>(ap + 16 * sk + 4); }
            else xa[sk] = *reinterpret_cast<const u32x4_t*>(ap + 16 * sk);
        }
    };
    int v = next_valid(blockIdx.x * NS + stream);
    int cnt = 0; int64_t row0 = 0;
    if (v < nslots) { cnt = t.cnt[v]; row0 = t.row[v]; fetch(row0, cnt + (cnt < max_points ? 1 : 0), 0); }
    while (v < nslots) {
        const int nrow = cnt + (cnt < max_points ? 1 : 0);
        const float padw = (float)(max_points - cnt);
        const int vn = next_valid(v + vstride);
        int cntn = 0; int64_t row0n = 0;
        if (vn < nslots) { cntn = t.cnt[vn]; row0n = t.row[vn]; }
        float mx[NBW], mn[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) { mx[nb] = -INFINITY; mn[nb] = INFINITY; }
        for (int g0 = 0; g0 < nrow; g0 += 32) {
            f32x16 acc[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) {
                if constexpr (X3) {
                    const float a[8] = {__uint_as_float(xa[2 * sk][0]), __uint_as_float(xa[2 * sk][1]), __uint_as_float(xa[2 * sk][2]), __uint_as_float(xa[2 * sk][3]),
                                        __uint_as_float(xa[2 * sk + 1][0]), __uint_as_float(xa[2 * sk + 1][1]), __uint_as_float(xa[2 * sk + 1][2]), __uint_as_float(xa[2 * sk + 1][3])};
                    u32x4_t ah_, al_;
                    split8(a, ah_, al_);
                    const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, ah_), al = __builtin_bit_cast(bf16x8_t, al_);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh[nb][sk], acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl[nb][sk], acc[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh[nb][sk], acc[nb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, xa[sk]), wh[nb][sk], acc[nb], 0, 0, 0);
                }
            }
            // the next row group - of this pillar, or the first one of the next pillar: issued here, in flight during the epilogue (one register set: the products
            // above were its last readers)
            __builtin_amdgcn_sched_barrier(0);
            if (g0 + 32 < nrow) fetch(row0, nrow, g0 + 32);
            else if (vn < nslots) fetch(row0n, cntn + (cntn < max_points ? 1 : 0), 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = g0 + crow32(r, hi);
                if (row < nrow) {
                    const float wgt = row < cnt ? 1.f : padw;
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        float val = acc[nb][r];
                        if constexpr (!X3) val = bf2f(f2bf(val));
                        mx[nb] = fmaxf(mx[nb], val); mn[nb] = fminf(mn[nb], val);
                        s1[nb] += wgt * val; s2[nb] += wgt * val * val;
                        if constexpr (STORE) H2[(row0 + row) * C + ch0 + nb * 32] = Cvt<T>::from_f(val);
                    }
                }
            }
        }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const float a = fmaxf(mx[nb], __shfl_xor(mx[nb], 32, 64)), b = fminf(mn[nb], __shfl_xor(mn[nb], 32, 64));
            if (hi == 0) { hmax[(int64_t)v * C + ch0 + nb * 32] = a; hmin[(int64_t)v * C + ch0 + nb * 32] = b; }
        }
        v = vn; cnt = cntn; row0 = row0n;
    }
    if (sums) {
        // the two row halves of a lane pair, then the workgroup's pillar streams, in a fixed order
        __shared__ float red[NS][2 * C];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            const float a1 = s1[nb] + __shfl_xor(s1[nb], 32, 64), a2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
            if (hi == 0) { red[stream][ch0 + nb * 32] = a1; red[stream][C + ch0 + nb * 32] = a2; }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += 256) {
            float a = red[0][i];
#pragma unroll
            for (int q = 1; q < NS; ++q) a += red[q][i];
            p3_commit(sums, slab, 2 * C, i, a);
        }
    }
}

// rows of H2 that belong to no kept pillar (points beyond the per-pillar cap, outside the range, in pillars beyond max_voxels): the weight-gradient and input-gradient
// products of the backward run over ALL rows, so they must hold zeros (the two-pass form's GEMM writes them as the product of zero rows)
template <typename T>
__global__ __launch_bounds__(256) void pfn_zero_unused_rows_kernel(T* __restrict__ H2, const int* __restrict__ row_vox, int64_t rows, int C) {
    // a wave looks at 64 consecutive rows at once (one coalesced load of their owners) and zeroes the few that have none
    const int lane = threadIdx.x & 63;
    for (int64_t rb = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; rb < rows; rb += (int64_t)gridDim.x * 4 * 64) {
        const int64_t r = rb + lane;
        unsigned long long m = __ballot(r < rows && row_vox[r] < 0);
        while (m) {
            const int k = __ffsll((long long)m) - 1;
            m &= m - 1;
            for (int c = lane; c < C; c += 64) H2[(rb + k) * C + c] = Cvt<T>::from_f(0.f);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pfn_scatter_kernel(VoxTab t, int max_voxels, int nslots, int C, int ncell,
                                                          const float* __restrict__ hmax, const float* __restrict__ hmin,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          T* __restrict__ out, int out_ld) {
    const int lane = threadIdx.x & 63;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nslots; v += gridDim.x * 4) {
        const int b = v / max_voxels;
        if ((v % max_voxels) >= t.nvox[b]) continue;
        const int xyf = t.xy[v];
        if (xyf & (1 << 30)) continue;
        const int xy = xyf & 0xffffff;
        T* dst = out + ((int64_t)b * ncell + xy) * out_ld;
        for (int c = lane; c < C; c += 64) {
            const float sc = scale[c];
            const float hv = sc > 0.f ? hmax[(int64_t)v * C + c] : hmin[(int64_t)v * C + c];
            dst[c] = Cvt<T>::from_f(fmaxf(sc * hv + shift[c], 0.f));
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Backward through the PFN parameters.  The dense [V, max_points] slot tensor of the reference never exists here either: the
// "real rows + one representative padded row (multiplicity P - cnt)" layout of the forward carries through, a padded row's
// gradient being the SUM over the identical slots it stands for.
// ------------------------------------------------------------------------------------------------
// step 1: per pillar / channel  g = dout * [relu'(y2 at the arg-max row)], BatchNorm-2 dbeta / dgamma.  hmax <- g, hmin <- selected h
template <typename T>
__global__ __launch_bounds__(256) void pfn_bwd_l2_stats_kernel(VoxTab t, int max_voxels, int nslots, int C, int ncell,
                                                               const T* __restrict__ dcanvas, int dld, const float* __restrict__ sc2,
                                                               const float* __restrict__ sh2, const float* __restrict__ mean2,
                                                               const float* __restrict__ rstd2, float* __restrict__ hmax,
                                                               float* __restrict__ hmin, float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                               float* __restrict__ slab = nullptr /* [gridDim.x][dbeta(C) | dgamma(C)] */) {
    const int lane = threadIdx.x & 63;
    constexpr int MAXJ = 12;
    float s1[MAXJ], s2[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const int nj = C / 64;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nslots; v += gridDim.x * 4) {
        const int b = v / max_voxels;
        if ((v % max_voxels) >= t.nvox[b]) continue;
        const int xyf = t.xy[v];
        const bool live = (xyf & (1 << 30)) == 0;
        const T* src = dcanvas + ((int64_t)b * ncell + (xyf & 0xffffff)) * dld;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            if (j < nj) {
                const int c = lane + 64 * j;
                const float sc = sc2[c];
                const int64_t i = (int64_t)v * C + c;
                const float hs = sc > 0.f ? hmax[i] : hmin[i];
                const float y = sc * hs + sh2[c];
                const float gv = (live && y > 0.f) ? Cvt<T>::to_f(src[c]) : 0.f;
                hmax[i] = gv; hmin[i] = hs;
                s1[j] += gv; s2[j] += gv * (hs - mean2[c]) * rstd2[c];
            }
        }
    }
    // block-level fold first, so that the grid can be 4x larger at the same number of same-address atomics (cf. pfn_l2_reduce_kernel)
    __shared__ float red[4][2 * 64 * MAXJ];
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
        if (j < nj) { red[wv][lane + 64 * j] = s1[j]; red[wv][64 * MAXJ + lane + 64 * j] = s2[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float vb = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        const float vg = (red[0][64 * MAXJ + c] + red[1][64 * MAXJ + c]) + (red[2][64 * MAXJ + c] + red[3][64 * MAXJ + c]);
        if (slab) { slab[(int64_t)blockIdx.x * 2 * C + c] = vb; slab[(int64_t)blockIdx.x * 2 * C + C + c] = vg; }
        else { atomicAdd(dbeta + c, vb); atomicAdd(dgamma + c, vg); }
    }
}

// bf16 canvas gradient, C % 8 == 0, C <= 512, 16-byte aligned rows: a lane owns 8 channels - the four per-channel BatchNorm vectors live in
// registers for the whole walk (the column-group form re-read them for every pillar), one 16-byte / two 16-byte accesses per tensor
// (r05: templated on the canvas gradient's type - the fp32 family used to stay on the column-group form: 207 us at 3 k points per tile against 70 for bf16 -
// and, with a slab, its workgroup partials go to the fixed-order reduce instead of atomics)
template <typename T>
__global__ __launch_bounds__(256) void pfn_bwd_l2_stats8_kernel(VoxTab t, int max_voxels, int nslots, int C, int ncell,
                                                                const T* __restrict__ dcanvas, int dld, const float* __restrict__ sc2,
                                                                const float* __restrict__ sh2, const float* __restrict__ mean2,
                                                                const float* __restrict__ rstd2, float* __restrict__ hmax,
                                                                float* __restrict__ hmin, float* __restrict__ dbeta, float* __restrict__ dgamma,
                                                                float* __restrict__ slab = nullptr /* [gridDim.x][dbeta(C) | dgamma(C)] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c0 = lane * 8;
    const bool act = c0 < C;
    float sc[8], sh[8], mu[8], rs[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = act ? sc2[c0 + k] : 0.f; sh[k] = act ? sh2[c0 + k] : 0.f; mu[k] = act ? mean2[c0 + k] : 0.f; rs[k] = act ? rstd2[c0 + k] : 0.f;
        s1[k] = 0.f; s2[k] = 0.f;
    }
    for (int v = blockIdx.x * 4 + wv; v < nslots; v += gridDim.x * 4) {
        const int b = v / max_voxels;
        if ((v % max_voxels) >= t.nvox[b]) continue;
        if (!act) continue;
        const int xyf = t.xy[v];
        const bool live = (xyf & (1 << 30)) == 0;
        Row8<T> raw;
        raw.load(dcanvas + ((int64_t)b * ncell + (xyf & 0xffffff)) * dld + c0);
        float gin[8];
        raw.get(gin);
        float* hx = hmax + (int64_t)v * C + c0;
        float* hn = hmin + (int64_t)v * C + c0;
        const float4 x0 = *reinterpret_cast<const float4*>(hx), x1 = *reinterpret_cast<const float4*>(hx + 4);
        const float4 n0 = *reinterpret_cast<const float4*>(hn), n1 = *reinterpret_cast<const float4*>(hn + 4);
        const float hxv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w}, hnv[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
        float gv[8], hs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            hs[k] = sc[k] > 0.f ? hxv[k] : hnv[k];
            const float y = sc[k] * hs[k] + sh[k];
            gv[k] = (live && y > 0.f) ? gin[k] : 0.f;
            s1[k] += gv[k]; s2[k] += gv[k] * (hs[k] - mu[k]) * rs[k];
        }
        *reinterpret_cast<float4*>(hx) = make_float4(gv[0], gv[1], gv[2], gv[3]); *reinterpret_cast<float4*>(hx + 4) = make_float4(gv[4], gv[5], gv[6], gv[7]);
        *reinterpret_cast<float4*>(hn) = make_float4(hs[0], hs[1], hs[2], hs[3]); *reinterpret_cast<float4*>(hn + 4) = make_float4(hs[4], hs[5], hs[6], hs[7]);
    }
    __shared__ float red[4][2 * 512];
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[wv][c0 + k] = s1[k]; red[wv][512 + c0 + k] = s2[k]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float vb = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        const float vg = (red[0][512 + c] + red[1][512 + c]) + (red[2][512 + c] + red[3][512 + c]);
        if (slab) { slab[(int64_t)blockIdx.x * 2 * C + c] = vb; slab[(int64_t)blockIdx.x * 2 * C + C + c] = vg; }
        else { atomicAdd(dbeta + c, vb); atomicAdd(dgamma + c, vg); }
    }
}

// step 2: H2 rows -> dH2 rows in place:  gamma*rstd * (g [first arg-max row] - w_row * (dbeta + xhat * dgamma) / n)
template <typename T>
__global__ __launch_bounds__(256) void pfn_bwd_l2_rows_kernel(VoxTab t, int max_voxels, int max_points, int nslots, int C, T* __restrict__ H2,
                                                              const float* __restrict__ g, const float* __restrict__ hsel,
                                                              const float* __restrict__ gamma2, const float* __restrict__ mean2,
                                                              const float* __restrict__ rstd2, const float* __restrict__ dbeta,
                                                              const float* __restrict__ dgamma, const int* __restrict__ totals, int training) {
    const int lane = threadIdx.x & 63;
    const int nj = C / 64;
    const float inv_n = training ? 1.f / fmaxf((float)totals[0] * (float)max_points, 1.f) : 0.f;
    for (int v = blockIdx.x * 4 + (threadIdx.x >> 6); v < nslots; v += gridDim.x * 4) {
        if ((v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        const int cnt = t.cnt[v];
        const int nrow = cnt + (cnt < max_points ? 1 : 0);
        T* base = H2 + (int64_t)t.row[v] * C;
        for (int j = 0; j < nj; ++j) {
            const int c = lane + 64 * j;
            const float mean = mean2[c], rstd = rstd2[c], gm = gamma2[c] * rstd;
            const float a = dbeta[c] * inv_n, bb = dgamma[c] * inv_n;
            const float gv = g[(int64_t)v * C + c], hs = hsel[(int64_t)v * C + c];
            bool found = false;
            for (int r = 0; r < nrow; ++r) {
                const float val = Cvt<T>::to_f(base[(int64_t)r * C + c]);
                const float wgt = r < cnt ? 1.f : (float)(max_points - cnt);
                float dv = -wgt * (a + (val - mean) * rstd * bb);
                if (!found && val == hs) { dv += gv; found = true; }
                base[(int64_t)r * C + c] = Cvt<T>::from_f(gm * dv);
            }
        }
    }
}

// C % 8 == 0, C <= 512: a lane owns 8 channels (16-byte row accesses, the per-channel constants in registers for the whole walk); r05: fp32 too
template <typename T>
__global__ __launch_bounds__(256) void pfn_bwd_l2_rows8_kernel(VoxTab t, int max_voxels, int max_points, int nslots, int C, T* __restrict__ H2,
                                                               const float* __restrict__ g, const float* __restrict__ hsel,
                                                               const float* __restrict__ gamma2, const float* __restrict__ mean2,
                                                               const float* __restrict__ rstd2, const float* __restrict__ dbeta,
                                                               const float* __restrict__ dgamma, const int* __restrict__ totals, int training) {
    // G = 64 / (C / 8) pillars per wave, one per lane group (see pfn_l2_reduce8_kernel)
    const int lane = threadIdx.x & 63, lpr = C >> 3, G = 64 / lpr, grp = lane / lpr, c0 = (lane - grp * lpr) * 8;
    if (grp >= G) return;
    const float inv_n = training ? 1.f / fmaxf((float)totals[0] * (float)max_points, 1.f) : 0.f;
    float mean[8], rstd[8], gm[8], a[8], bb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mean[k] = mean2[c0 + k]; rstd[k] = rstd2[c0 + k]; gm[k] = gamma2[c0 + k] * rstd[k];
        a[k] = dbeta[c0 + k] * inv_n; bb[k] = dgamma[c0 + k] * inv_n;
    }
    for (int vb = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G; vb < nslots; vb += gridDim.x * 4 * G) {
        const int v = vb + grp;
        if (v >= nslots || (v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        const int cnt = t.cnt[v];
        const int nrow = cnt + (cnt < max_points ? 1 : 0);
        T* base = H2 + (int64_t)t.row[v] * C + c0;
        const float* gp = g + (int64_t)v * C + c0;
        const float* hp = hsel + (int64_t)v * C + c0;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(hp), h1 = *reinterpret_cast<const float4*>(hp + 4);
        const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, hs[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        uint32_t found = 0;                              // bit k: the arg-max row of channel k has been credited
        for (int r0 = 0; r0 < nrow; r0 += P3_PFN_NR) {
            Row8<T> raw[P3_PFN_NR];
#pragma unroll
            for (int q = 0; q < P3_PFN_NR; ++q) { if (r0 + q < nrow) raw[q].load(base + (int64_t)(r0 + q) * C); else raw[q].zero(); }
#pragma unroll
            for (int q = 0; q < P3_PFN_NR; ++q) {
                if (r0 + q >= nrow) break;
                const float wgt = r0 + q < cnt ? 1.f : (float)(max_points - cnt);
                float vals[8], o[8];
                raw[q].get(vals);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float val = vals[k];
                    float dv = -wgt * (a[k] + (val - mean[k]) * rstd[k] * bb[k]);
                    if (!((found >> k) & 1u) && val == hs[k]) { dv += gv[k]; found |= 1u << k; }
                    o[k] = gm[k] * dv;
                }
                Row8<T>::store(base + (int64_t)(r0 + q) * C, o);
            }
        }
    }
}

// step 3: layer 0.  lane = (channel c = lane & 31, row parity = lane >> 5); per-lane sequential walk over the pillar's rows.
template <typename T>
__global__ __launch_bounds__(256) void pfn_bwd_l1_kernel(VoxTab t, int max_voxels, int max_points, int nslots, const float* __restrict__ F8,
                                                         const T* __restrict__ dX2, const float* __restrict__ w1, const float* __restrict__ sc1,
                                                         const float* __restrict__ sh1, const float* __restrict__ mean1,
                                                         const float* __restrict__ rstd1, float* __restrict__ acc, float* __restrict__ slab = nullptr) {
    const int lane = threadIdx.x & 63, c = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    float wr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wr[k] = w1[c * 8 + k];
    const float s = sc1[c], sh = sh1[c], m1 = mean1[c], r1 = rstd1[c];
    float a_dyf[8], a_xf[8], a_f[8], db = 0.f, dg = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a_dyf[k] = 0.f; a_xf[k] = 0.f; a_f[k] = 0.f; }
    for (int v = blockIdx.x * 4 + wave; v < nslots; v += gridDim.x * 4) {
        if ((v % max_voxels) >= t.nvox[v / max_voxels]) continue;
        const int cnt = t.cnt[v];
        const bool has_pad = cnt < max_points;
        const int nrow = cnt + (has_pad ? 1 : 0);
        const int64_t row0 = t.row[v];
        float best = -INFINITY; int bi = 0x7fffffff;
        float dxm = 0.f;
        // four rows per parity in flight (r05: one row per trip was a chain of dependent loads - 748 us at 40 k points per tile for ~0.8 GB)
        for (int r0 = half; r0 < nrow; r0 += 8) {
            float dxv[4]; float4 fa[4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 2 * q;
                dxv[q] = r < nrow ? Cvt<T>::to_f(dX2[(row0 + r) * K2 + C1 + c]) : 0.f;
                if (r < cnt) { fa[q] = *reinterpret_cast<const float4*>(F8 + (row0 + r) * 8); fb[q] = *reinterpret_cast<const float4*>(F8 + (row0 + r) * 8 + 4); }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 2 * q;
                dxm += dxv[q];
                if (r < cnt) {
                    const float f[8] = {fa[q].x, fa[q].y, fa[q].z, fa[q].w, fb[q].x, fb[q].y, fb[q].z, fb[q].w};
                    float h = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) h += wr[k] * f[k];
                    const float x = fmaxf(h * s + sh, 0.f);
                    if (x > best) { best = x; bi = r; }
                }
            }
        }
        {
            const float ob = __shfl_xor(best, 32, 64); const int oi = __shfl_xor(bi, 32, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            dxm += __shfl_xor(dxm, 32, 64);
        }
        const bool pad_arg = has_pad && fmaxf(sh, 0.f) > best;
        for (int r0 = half; r0 < cnt; r0 += 8) {
            float dyv[4]; float4 fa[4], fb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 2 * q;
                if (r < cnt) {
                    dyv[q] = Cvt<T>::to_f(dX2[(row0 + r) * K2 + c]);
                    fa[q] = *reinterpret_cast<const float4*>(F8 + (row0 + r) * 8); fb[q] = *reinterpret_cast<const float4*>(F8 + (row0 + r) * 8 + 4);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = r0 + 2 * q;
                if (r >= cnt) break;
                const float f[8] = {fa[q].x, fa[q].y, fa[q].z, fa[q].w, fb[q].x, fb[q].y, fb[q].z, fb[q].w};
                float h = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) h += wr[k] * f[k];
                const float y = h * s + sh;
                float dy = dyv[q] + ((r == bi && !pad_arg) ? dxm : 0.f);
                dy = y > 0.f ? dy : 0.f;
                const float xh = (h - m1) * r1;
#pragma unroll
                for (int k = 0; k < 8; ++k) { a_dyf[k] += dy * f[k]; a_xf[k] += xh * f[k]; a_f[k] += f[k]; }
                db += dy; dg += dy * xh;
            }
        }
        if (has_pad && half == 0) {   // the padded slots: h = 0, y = shift; their summed gradient sits in the representative row
            float dp = Cvt<T>::to_f(dX2[(row0 + cnt) * K2 + c]) + (pad_arg ? dxm : 0.f);
            dp = sh > 0.f ? dp : 0.f;
            db += dp; dg += dp * (0.f - m1) * r1;
        }
    }
    // fold the two row parities, then the block's 4 waves, then one atomic per value per block
    __shared__ float red[4][ACC1_FLOATS];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a_dyf[k] += __shfl_xor(a_dyf[k], 32, 64); a_xf[k] += __shfl_xor(a_xf[k], 32, 64); a_f[k] += __shfl_xor(a_f[k], 32, 64);
    }
    db += __shfl_xor(db, 32, 64); dg += __shfl_xor(dg, 32, 64);
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[wave][ACC_DYF + c * 8 + k] = a_dyf[k]; red[wave][ACC_XF + c * 8 + k] = a_xf[k]; }
        red[wave][ACC_DB + c] = db; red[wave][ACC_DG + c] = dg;
        if (c == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[wave][ACC_F + k] = a_f[k];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ACC1_FLOATS; i += 256) p3_commit(acc, slab, ACC1_FLOATS, i, (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]));
}

// step 4: dW1 = gamma*rstd * (S_dyf - dbeta/n * S_f - dgamma/n * S_xf), dgamma1, dbeta1
__global__ void pfn_bwd_l1_finalize_kernel(const float* __restrict__ acc, const float* __restrict__ stat /*[dbeta(32) | dgamma(32)]*/,
                                           const float* __restrict__ gamma1, const float* __restrict__ rstd1,
                                           const int* __restrict__ totals, int max_points, int training, float* __restrict__ dw1,
                                           float* __restrict__ dg1, float* __restrict__ db1) {
    const int i = threadIdx.x, c = i >> 3, k = i & 7;
    const float inv_n = training ? 1.f / fmaxf((float)totals[0] * (float)max_points, 1.f) : 0.f;
    const float dbv = stat[c], dgv = stat[C1 + c];
    dw1[i] = gamma1[c] * rstd1[c] * (acc[ACC_DYF + i] - dbv * inv_n * acc[ACC_F + k] - dgv * inv_n * acc[ACC_XF + i]);
    if (k == 0) { dg1[c] = acc[ACC_DG + c]; db1[c] = acc[ACC_DB + c]; }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Ws {
    int* sorted; int* vox_xy; int* vox_start; int* vox_cnt; int* vox_row; int* nvox; int* totals;
    float* sums1; float* sums2; float* sc1; float* sh1; float* sc2; float* sh2; float* hmax; float* hmin;
    float* m1; float* r1; float* m2; float* r2;   // saved BatchNorm mean / rstd (backward)
    float* acc1;                                  // backward accumulators of PFN layer 0 (ACC1_FLOATS)
    void* X2; void* H2; float* F8; int* row_vox; float* row_w;
    unsigned* sort_tmp;
    size_t bytes;
};

Ws carve(void* base, const p3_pillar_desc* d) {
    Ws w; char* p = (char*)base; size_t off = 0;
    const size_t nv = (size_t)d->B * d->max_voxels;
    const size_t rows = (size_t)d->total_points + nv;
    const size_t es = d->dtype == P3_BF16 ? 2 : 4;
    auto take = [&](size_t bytes) { void* r = p ? p + off : nullptr; off += align256(bytes); return r; };
    w.totals = (int*)take(256);
    w.sums1 = (float*)take(2 * C1 * 4);
    w.sums2 = (float*)take(2 * (size_t)d->C * 4);
    const size_t zero_end = off;   // [0, zero_end) is cleared every call
    w.sc1 = (float*)take(C1 * 4); w.sh1 = (float*)take(C1 * 4);
    w.sc2 = (float*)take((size_t)d->C * 4); w.sh2 = (float*)take((size_t)d->C * 4);
    w.m1 = (float*)take(C1 * 4); w.r1 = (float*)take(C1 * 4);
    w.m2 = (float*)take((size_t)d->C * 4); w.r2 = (float*)take((size_t)d->C * 4);
    w.acc1 = (float*)take(ACC1_FLOATS * 4);
    w.sorted = (int*)take((size_t)(d->total_points > 0 ? d->total_points : 1) * 4);
    w.vox_xy = (int*)take(nv * 4); w.vox_start = (int*)take(nv * 4); w.vox_cnt = (int*)take(nv * 4); w.vox_row = (int*)take(nv * 4);
    w.nvox = (int*)take((size_t)d->B * 4);
    w.hmax = (float*)take(nv * d->C * 4); w.hmin = (float*)take(nv * d->C * 4);
    w.F8 = (float*)take(rows * 8 * 4); w.row_w = (float*)take(rows * 4); w.row_vox = (int*)take(rows * 4);
    w.X2 = take(rows * K2 * es);
    w.H2 = take(rows * (size_t)d->C * es);
    w.sort_tmp = (unsigned*)take((size_t)d->B * (SORT_SPLIT + 1) * (size_t)(d->nx + 1) * (d->ny + 1) * 2 * 4);   // the split sort's chunk histograms + cell starts
    w.bytes = off;
    (void)zero_end;
    return w;
}

}  // namespace

// zero `per_row` 16-byte chunks at the start of every row (row pitch ldb bytes): the canvas columns the pillars scatter into
__global__ __launch_bounds__(256) void zero_cols16_kernel(char* __restrict__ base, int64_t ldb, int per_row, int64_t tot) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        *reinterpret_cast<uint4*>(base + r * ldb + (i - r * per_row) * 16) = make_uint4(0, 0, 0, 0);
    }
}

// does PFN layer 1 run as one launch (pfn_l2_fused_kernel)?  Dense clouds only - >= 16 points per pillar slot on average (the bench's 3 k points per tile are 3.8,
// BASELINE's real tiles ~51): below that a pillar fills a fraction of a 32-row MFMA group and the two-pass form is faster - at the widths the kernel is built for
static bool pfn_layer1_fused(const p3_pillar_desc* d, const void* w2) {
    return (d->C == 128 || d->C == 384) && (d->dtype == P3_BF16 || d->dtype == P3_F32X3) && ((uintptr_t)w2 % 16) == 0 &&
           d->total_points >= 16 * (int64_t)d->B * d->max_voxels;
}

extern "C" int64_t p3_pillar_stem_workspace_bytes(const p3_pillar_desc* d) {
    if (!d) return -1;
    return (int64_t)carve(nullptr, d).bytes;
}

// phases (bit mask): 1 = pillarize + layer-0 BatchNorm sums | 2 = layer 0 apply, layer-1 GEMM, reduce + BatchNorm sums |
// 4 = finalize + scatter.  SyncBatchNorm all-reduces the workspace's `totals`/`sums1` after phase 1 and `sums2` after phase 2.
extern "C" int p3_pillar_stem_phased(const float* values, const int64_t* offsets, const float* w1, const float* bn1_gamma,
                                     const float* bn1_beta, float* bn1_rmean, float* bn1_rvar, const void* w2, const float* bn2_gamma,
                                     const float* bn2_beta, float* bn2_rmean, float* bn2_rvar, void* out, void* workspace,
                                     const p3_pillar_desc* d, int phases, void* stream) {
    P3_CHECK(values && offsets && w1 && w2 && out && workspace && d, P3_EINVAL, "p3_pillar_stem: null pointer");
    P3_CHECK(d->B > 0 && d->nx > 0 && d->ny > 0 && d->max_points > 0 && d->max_points <= 4096, P3_ESHAPE, "p3_pillar_stem: max_points must be 1..4096");
    P3_CHECK(d->C % 64 == 0 && d->C <= 768, P3_ESHAPE, "p3_pillar_stem: C must be a multiple of 64, <= 768");
    P3_CHECK(d->dtype == P3_F32 || d->dtype == P3_BF16 || d->dtype == P3_F32X3, P3_EUNSUP, "p3_pillar_stem: dtype");
    P3_CHECK(d->vz >= d->zmax, P3_EUNSUP, "p3_pillar_stem: only one z cell (voxel z size == z range) is supported");
    PillarGeom g;
    g.nx = d->nx; g.ny = d->ny; g.ncx = d->nx + 1; g.ncy = d->ny + 1; g.nc = g.ncx * g.ncy * 2;
    P3_CHECK(g.nc <= MAX_CELLS, P3_EUNSUP, "p3_pillar_stem: pillar grid too large for the LDS counting sort");
    g.invx = 1.0f / d->vx; g.invy = 1.0f / d->vy; g.invz = 1.0f / d->vz;
    g.xmax = d->nx * d->vx; g.ymax = d->ny * d->vy; g.zmax = d->zmax; g.vx = d->vx; g.vy = d->vy;
    hipStream_t s = (hipStream_t)stream;
    const int kdt = d->dtype == P3_BF16 ? P3_BF16 : P3_F32;       // the dtype p3_det_scratch decides by (P3_F32X3 is fp32 storage)
    Ws w = carve(workspace, d);
    const int nslots = d->B * d->max_voxels;
    const size_t es = d->dtype == P3_BF16 ? 2 : 4;
    const size_t rows = (size_t)d->total_points + (size_t)nslots;
    hipError_t e;
    VoxTab t{w.sorted, w.vox_xy, w.vox_start, w.vox_cnt, w.vox_row, w.nvox};
    const int vgrid = (nslots + 3) / 4 < 2048 ? (nslots + 3) / 4 : 2048;
    if (phases & 1) {
    e = hipMemsetAsync(w.totals, 0, (char*)w.sc1 - (char*)w.totals, s);
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    // the one-launch layer 1 without a backward reads the rows of kept pillars only: the rows of no pillar need no defined content then (0.67 GB of memset at 40 k
    // points per tile); every other path multiplies ALL rows (the GEMM, the backward's products) and needs them finite / zero
    const bool lean_rows = d->no_backward && pfn_layer1_fused(d, w2);
    if (!lean_rows) {
        e = hipMemsetAsync(w.X2, 0, rows * K2 * es, s);   // unused rows must be finite for the GEMM
        if (e == hipSuccess) e = hipMemsetAsync(w.F8, 0, (char*)w.row_vox - (char*)w.F8, s);
    }
    if (e == hipSuccess) e = hipMemsetAsync(w.row_vox, 0xFF, rows * 4, s);   // -1 = unused row
    if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    // zero canvas columns [col_off, col_off + C) of every token row (empty pillars stay exactly 0)
    {
        const size_t rowb = (size_t)d->C * es, ldb = (size_t)d->out_ld * es, offb = (size_t)d->out_col_off * es;
        const int64_t nrows = (int64_t)d->B * d->nx * d->ny;
        if (rowb % 16 == 0 && ldb % 16 == 0 && offb % 16 == 0 && ((uintptr_t)out % 16) == 0) {   // 16-byte stores (the 2-D memset runs at < 1 TB/s)
            const int per_row = (int)(rowb / 16);
            const int64_t tot = nrows * per_row;
            hipLaunchKernelGGL(zero_cols16_kernel, dim3((unsigned)((tot + 255) / 256 < 16384 ? (tot + 255) / 256 : 16384)), dim3(256), 0, s,
                               (char*)out + offb, (int64_t)ldb, per_row, tot);
            P3_LAUNCH_CHECK();
        } else {
            e = hipMemset2DAsync((char*)out + offb, ldb, 0, rowb, (size_t)nrows, s);
            if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
        }
    }

    SortOut so{w.sorted, w.vox_xy, w.vox_start, w.vox_cnt, w.vox_row, w.nvox, w.totals};
    const size_t lds = (size_t)(SORT_WAVES + 5) * g.nc * 4 + 32 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pillar_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (d->total_points >= 16 * (int64_t)nslots) {
        // dense clouds: a tile's points over SORT_SPLIT workgroups (count / tables / fill)
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute((const void*)pillar_sort_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)pillar_sort_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr2 = true;
        }
        unsigned* ghist = w.sort_tmp;
        unsigned* gstart = w.sort_tmp + (size_t)d->B * SORT_SPLIT * g.nc;
        const size_t lds_h = (size_t)SORT_WAVES * g.nc * 4, lds_t = (size_t)5 * g.nc * 4 + 32 * 4;
        hipLaunchKernelGGL(pillar_sort_count_kernel, dim3(SORT_SPLIT, d->B), dim3(SORT_THREADS), lds_h, s, values, offsets, g, ghist);
        P3_LAUNCH_CHECK();
        hipLaunchKernelGGL(pillar_sort_tables_kernel, dim3(d->B), dim3(SORT_THREADS), lds_t, s, offsets, g, d->max_points, d->max_voxels, so, ghist, gstart);
        P3_LAUNCH_CHECK();
        hipLaunchKernelGGL(pillar_sort_fill_kernel, dim3(SORT_SPLIT, d->B), dim3(SORT_THREADS), lds_h, s, values, offsets, g, ghist, gstart, w.sorted);
        P3_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(pillar_sort_kernel, dim3(d->B), dim3(SORT_THREADS), lds, s, values, offsets, g, d->max_points, d->max_voxels, so);
        P3_LAUNCH_CHECK();
    }
    if (d->training) {
        // deterministic mode (p3_set_deterministic covers this dtype): the workgroups' partial sums go to the scratch and are added in workgroup order in float64
        const int g1 = vgrid < 512 ? vgrid : 512;
        float* slab1 = p3_det_scratch((int64_t)g1 * 2 * C1, kdt);
        if (d->max_points > 64) hipLaunchKernelGGL(pfn_l1_stats_kernel<true>, dim3(g1), dim3(256), 0, s, values, t, g, d->max_voxels, nslots, w1, w.sums1, slab1);
        else hipLaunchKernelGGL(pfn_l1_stats_kernel<false>, dim3(g1), dim3(256), 0, s, values, t, g, d->max_voxels, nslots, w1, w.sums1, slab1);
        P3_LAUNCH_CHECK();
        if (slab1) { int rc1 = p3_det_reduce(slab1, g1, 2 * C1, w.sums1, 2 * C1, 1, s); if (rc1 != P3_OK) return rc1; }
    }
    }
    if (phases & 2) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, s, w.sums1, C1, w.totals, (float)d->max_points, 0.f, bn1_gamma, bn1_beta,
                       bn1_rmean, bn1_rvar, d->bn_eps, d->bn_momentum, d->training, w.sc1, w.sh1, w.m1, w.r1);
    P3_LAUNCH_CHECK();
#define P3_L1_APPLY(T, MULTI) hipLaunchKernelGGL((pfn_l1_apply_kernel<T, MULTI>), dim3(vgrid), dim3(256), 0, s, values, t, g, d->max_voxels, d->max_points, nslots, w1, w.sc1, w.sh1, (T*)w.X2, w.F8, w.row_vox, w.row_w)
    if (d->dtype == P3_BF16) { if (d->max_points > 64) P3_L1_APPLY(bf16_t, true); else P3_L1_APPLY(bf16_t, false); }
    else { if (d->max_points > 64) P3_L1_APPLY(float, true); else P3_L1_APPLY(float, false); }
#undef P3_L1_APPLY
    P3_LAUNCH_CHECK();
    float* sums2 = d->training ? w.sums2 : nullptr;
    const bool fused2 = pfn_layer1_fused(d, w2);
    if (fused2) {
        // layer 1 in one launch: product, per-pillar max / min, BatchNorm sums (pfn_l2_fused_kernel); H2 is written only when a backward pass will read it
        const int ns = d->C == 128 ? 2 : 1;                          // pillars in flight per workgroup
        const int gf = (nslots + ns - 1) / ns < 1024 ? (nslots + ns - 1) / ns : 1024;
        float* slab2 = sums2 ? p3_det_scratch((int64_t)gf * 2 * d->C, kdt) : nullptr;
        const bool keep = !d->no_backward;
#define P3_L2_FUSED(T, ST) do { if (d->C == 128) hipLaunchKernelGGL((pfn_l2_fused_kernel<T, ST, 2, 2>), dim3(gf), dim3(256), 0, s, (const T*)w.X2, (const T*)w2, t, d->max_voxels, d->max_points, nslots, (T*)w.H2, w.hmax, w.hmin, sums2, slab2); \
        else hipLaunchKernelGGL((pfn_l2_fused_kernel<T, ST, 3, 4>), dim3(gf), dim3(256), 0, s, (const T*)w.X2, (const T*)w2, t, d->max_voxels, d->max_points, nslots, (T*)w.H2, w.hmax, w.hmin, sums2, slab2); } while (0)
        if (d->dtype == P3_BF16) { if (keep) P3_L2_FUSED(bf16_t, true); else P3_L2_FUSED(bf16_t, false); }
        else { if (keep) P3_L2_FUSED(float, true); else P3_L2_FUSED(float, false); }
#undef P3_L2_FUSED
        P3_LAUNCH_CHECK();
        if (p3_tracing()) p3_note_kernel(keep ? "pfn_l2_fused_kernel<store>" : "pfn_l2_fused_kernel");
        if (keep) {
            const int gz = (int)((rows + 255) / 256 < 1024 ? (rows + 255) / 256 : 1024);
            if (d->dtype == P3_BF16) hipLaunchKernelGGL((pfn_zero_unused_rows_kernel<bf16_t>), dim3(gz), dim3(256), 0, s, (bf16_t*)w.H2, w.row_vox, (int64_t)rows, d->C);
            else hipLaunchKernelGGL((pfn_zero_unused_rows_kernel<float>), dim3(gz), dim3(256), 0, s, (float*)w.H2, w.row_vox, (int64_t)rows, d->C);
            P3_LAUNCH_CHECK();
        }
        if (slab2) { int rc2 = p3_det_reduce(slab2, gf, 2 * d->C, sums2, 2 * d->C, 1, s); if (rc2 != P3_OK) return rc2; }
    } else {
    p3_gemm_desc gd;
    memset(&gd, 0, sizeof(gd));
    gd.M = (int)rows; gd.N = d->C; gd.K = K2; gd.lda = K2; gd.ldb = K2; gd.ldc = d->C;
    gd.dtype_in = d->dtype; gd.dtype_out = d->dtype == P3_F32X3 ? P3_F32 : d->dtype; gd.act = P3_ACT_NONE; gd.a_mode = P3_A_PLAIN;    // P3_F32X3: the PFN products as bf16 x 3, storage fp32
    int rc = p3_gemm(w.X2, w2, w.H2, &gd, stream);
    if (rc != P3_OK) return rc;
    if (d->C % 8 == 0 && d->C <= 512) {
        const int g2 = vgrid < L2R_BLOCKS ? vgrid : L2R_BLOCKS;
        float* slab2 = sums2 ? p3_det_scratch((int64_t)g2 * 2 * d->C, kdt) : nullptr;
        if (d->dtype == P3_BF16) hipLaunchKernelGGL((pfn_l2_reduce8_kernel<bf16_t>), dim3(g2), dim3(256), 0, s, (const bf16_t*)w.H2, t, d->max_voxels, d->max_points, nslots, d->C, w.hmax, w.hmin, sums2, slab2);
        else hipLaunchKernelGGL((pfn_l2_reduce8_kernel<float>), dim3(g2), dim3(256), 0, s, (const float*)w.H2, t, d->max_voxels, d->max_points, nslots, d->C, w.hmax, w.hmin, sums2, slab2);
        P3_LAUNCH_CHECK();
        if (slab2) { int rc2 = p3_det_reduce(slab2, g2, 2 * d->C, sums2, 2 * d->C, 1, s); if (rc2 != P3_OK) return rc2; }
    } else if (d->dtype == P3_BF16)
        hipLaunchKernelGGL((pfn_l2_reduce_kernel<bf16_t>), dim3(vgrid < L2R_BLOCKS ? vgrid : L2R_BLOCKS), dim3(256), 0, s, (const bf16_t*)w.H2, t, d->max_voxels, d->max_points, nslots, d->C, w.hmax, w.hmin, sums2);
    else {
        const int g2 = vgrid < L2R_BLOCKS ? vgrid : L2R_BLOCKS;
        float* slab2 = sums2 ? p3_det_scratch((int64_t)g2 * 2 * d->C, kdt) : nullptr;
        hipLaunchKernelGGL((pfn_l2_reduce_kernel<float>), dim3(g2), dim3(256), 0, s, (const float*)w.H2, t, d->max_voxels, d->max_points, nslots, d->C, w.hmax, w.hmin, sums2, slab2);
        P3_LAUNCH_CHECK();
        if (slab2) { int rc2 = p3_det_reduce(slab2, g2, 2 * d->C, sums2, 2 * d->C, 1, s); if (rc2 != P3_OK) return rc2; }
    }
    }
    P3_LAUNCH_CHECK();
    }
    if (phases & 4) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->C + 63) / 64), dim3(64), 0, s, w.sums2, d->C, w.totals, (float)d->max_points, 0.f, bn2_gamma,
                       bn2_beta, bn2_rmean, bn2_rvar, d->bn_eps, d->bn_momentum, d->training, w.sc2, w.sh2, w.m2, w.r2);
    P3_LAUNCH_CHECK();
    if (d->dtype == P3_BF16)
        hipLaunchKernelGGL((pfn_scatter_kernel<bf16_t>), dim3(vgrid), dim3(256), 0, s, t, d->max_voxels, nslots, d->C, d->nx * d->ny, w.hmax, w.hmin, w.sc2, w.sh2, (bf16_t*)out + d->out_col_off, d->out_ld);
    else
        hipLaunchKernelGGL((pfn_scatter_kernel<float>), dim3(vgrid), dim3(256), 0, s, t, d->max_voxels, nslots, d->C, d->nx * d->ny, w.hmax, w.hmin, w.sc2, w.sh2, (float*)out + d->out_col_off, d->out_ld);
    P3_LAUNCH_CHECK();
    }
    return P3_OK;
}

extern "C" int p3_pillar_stem(const float* values, const int64_t* offsets, const float* w1, const float* bn1_gamma,
                              const float* bn1_beta, float* bn1_rmean, float* bn1_rvar, const void* w2, const float* bn2_gamma,
                              const float* bn2_beta, float* bn2_rmean, float* bn2_rvar, void* out, void* workspace,
                              const p3_pillar_desc* d, void* stream) {
    return p3_pillar_stem_phased(values, offsets, w1, bn1_gamma, bn1_beta, bn1_rmean, bn1_rvar, w2, bn2_gamma, bn2_beta, bn2_rmean, bn2_rvar,
                                 out, workspace, d, 7, stream);
}



// Backward of p3_pillar_stem w.r.t. the PFN parameters.  `workspace` is the buffer the matching forward call filled; it is consumed.
// phases: 1 = layer-1 arg-max gradients + local dgamma2 / dbeta2 | 2 = dH2 rows, both GEMMs, layer-0 accumulation | 4 = layer-0 finalize.
// stat2 = [dbeta2 (C) | dgamma2 (C)] and stat1 = [dbeta1 (32) | dgamma1 (32)] are the sums the BatchNorm input gradients use: NULL =
// this rank's own (plain BatchNorm); SyncBatchNorm passes their all-reduced copies (the parameter gradients stay local, like torch).
extern "C" int p3_pillar_stem_bwd_phased(const void* dcanvas, int dcanvas_ld, const float* w1, const float* bn1_gamma, const void* w2t,
                                         const float* bn2_gamma, void* workspace, const p3_pillar_desc* d, float* dw1, float* dg1,
                                         float* db1, float* dw2, float* dg2, float* db2, const float* stat2, const float* stat1, int phases,
                                         void* stream) {
    P3_CHECK(dcanvas && w1 && bn1_gamma && w2t && bn2_gamma && workspace && d && dw1 && dg1 && db1 && dw2 && dg2 && db2, P3_EINVAL,
             "p3_pillar_stem_bwd: null pointer");
    P3_CHECK(d->C % 64 == 0 && d->C <= 768 && d->max_points > 0 && d->max_points <= 4096, P3_ESHAPE, "p3_pillar_stem_bwd: shape");
    P3_CHECK(!d->no_backward, P3_EINVAL, "p3_pillar_stem_bwd: the forward ran with no_backward set - its workspace holds no layer-1 activations");
    hipStream_t s = (hipStream_t)stream;
    const int kdt = d->dtype == P3_BF16 ? P3_BF16 : P3_F32;       // the dtype p3_det_scratch decides by (P3_F32X3 is fp32 storage)
    Ws w = carve(workspace, d);
    const int nslots = d->B * d->max_voxels;
    const size_t rows = (size_t)d->total_points + (size_t)nslots;
    const int C = d->C;
    if (phases & 1) {
        hipError_t e = hipMemsetAsync(w.acc1, 0, ACC1_FLOATS * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(dg2, 0, (size_t)C * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(db2, 0, (size_t)C * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(dw2, 0, (size_t)C * K2 * 4, s);
        if (e != hipSuccess) { p3_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const float* s2_db = stat2 ? stat2 : db2;
    const float* s2_dg = stat2 ? stat2 + C : dg2;
    VoxTab t{w.sorted, w.vox_xy, w.vox_start, w.vox_cnt, w.vox_row, w.nvox};
    const int vgrid = (nslots + 3) / 4 < 2048 ? (nslots + 3) / 4 : 2048;
    const int sgrid = vgrid < 512 ? vgrid : 512;
    const int ncell = d->nx * d->ny;
    const bool bf = d->dtype == P3_BF16;
    if (phases & 1) {
    if (bf)
        if (C % 8 == 0 && C <= 512 && d->out_col_off % 8 == 0 && dcanvas_ld % 8 == 0 && ((uintptr_t)dcanvas % 16) == 0)
            hipLaunchKernelGGL(pfn_bwd_l2_stats8_kernel<bf16_t>, dim3(vgrid < L2S_BLOCKS ? vgrid : L2S_BLOCKS), dim3(256), 0, s, t, d->max_voxels, nslots, C, ncell, (const bf16_t*)dcanvas + d->out_col_off, dcanvas_ld, w.sc2, w.sh2, w.m2, w.r2, w.hmax, w.hmin, db2, dg2, (float*)nullptr);
        else
        hipLaunchKernelGGL((pfn_bwd_l2_stats_kernel<bf16_t>), dim3(vgrid < L2S_BLOCKS ? vgrid : L2S_BLOCKS), dim3(256), 0, s, t, d->max_voxels, nslots, C, ncell, (const bf16_t*)dcanvas + d->out_col_off, dcanvas_ld, w.sc2, w.sh2, w.m2, w.r2, w.hmax, w.hmin, db2, dg2);
    else {
        const int g3 = vgrid < L2S_BLOCKS ? vgrid : L2S_BLOCKS;
        float* slab3 = p3_det_scratch((int64_t)g3 * 2 * C, kdt);
        if (C % 8 == 0 && C <= 512 && d->out_col_off % 4 == 0 && dcanvas_ld % 4 == 0 && ((uintptr_t)dcanvas % 16) == 0)
            hipLaunchKernelGGL(pfn_bwd_l2_stats8_kernel<float>, dim3(g3), dim3(256), 0, s, t, d->max_voxels, nslots, C, ncell, (const float*)dcanvas + d->out_col_off, dcanvas_ld, w.sc2, w.sh2, w.m2, w.r2, w.hmax, w.hmin, db2, dg2, slab3);
        else
        hipLaunchKernelGGL((pfn_bwd_l2_stats_kernel<float>), dim3(g3), dim3(256), 0, s, t, d->max_voxels, nslots, C, ncell, (const float*)dcanvas + d->out_col_off, dcanvas_ld, w.sc2, w.sh2, w.m2, w.r2, w.hmax, w.hmin, db2, dg2, slab3);
        P3_LAUNCH_CHECK();
        if (slab3) {
            int rc3 = p3_det_reduce(slab3, g3, 2 * C, db2, C, 1, s);
            if (rc3 == P3_OK) rc3 = p3_det_reduce(slab3 + C, g3, 2 * C, dg2, C, 1, s);
            if (rc3 != P3_OK) return rc3;
        }
    }
    P3_LAUNCH_CHECK();
    }
    if (phases & 2) {
    if (C % 8 == 0 && C <= 512) {
        if (bf) hipLaunchKernelGGL((pfn_bwd_l2_rows8_kernel<bf16_t>), dim3(vgrid), dim3(256), 0, s, t, d->max_voxels, d->max_points, nslots, C, (bf16_t*)w.H2, w.hmax, w.hmin, bn2_gamma, w.m2, w.r2, s2_db, s2_dg, w.totals, d->training);
        else hipLaunchKernelGGL((pfn_bwd_l2_rows8_kernel<float>), dim3(vgrid), dim3(256), 0, s, t, d->max_voxels, d->max_points, nslots, C, (float*)w.H2, w.hmax, w.hmin, bn2_gamma, w.m2, w.r2, s2_db, s2_dg, w.totals, d->training);
    } else if (bf)
        hipLaunchKernelGGL((pfn_bwd_l2_rows_kernel<bf16_t>), dim3(vgrid), dim3(256), 0, s, t, d->max_voxels, d->max_points, nslots, C, (bf16_t*)w.H2, w.hmax, w.hmin, bn2_gamma, w.m2, w.r2, s2_db, s2_dg, w.totals, d->training);
    else
        hipLaunchKernelGGL((pfn_bwd_l2_rows_kernel<float>), dim3(vgrid), dim3(256), 0, s, t, d->max_voxels, d->max_points, nslots, C, (float*)w.H2, w.hmax, w.hmin, bn2_gamma, w.m2, w.r2, s2_db, s2_dg, w.totals, d->training);
    P3_LAUNCH_CHECK();
    // dW2[C, 64] = dH2^T . X2
    // deterministic mode: the split-M partial tiles go through the scratch and are added in split order (the host wrappers of p3_gemm_tn hand their own slabs in)
    const int tn_slabs = 64;
    float* tslab = p3_det_scratch((int64_t)tn_slabs * C * K2, kdt);
    int rc = p3_gemm_tn_ex(w.H2, w.X2, dw2, (int)rows, C, K2, C, K2, K2, d->dtype, 0, nullptr, nullptr, nullptr, 0, nullptr, tslab, tslab ? tn_slabs : 0, stream);
    if (rc != P3_OK) return rc;
    // dX2[rows, 64] = dH2 . W2  (written over X2, which the weight-gradient GEMM above has finished with)
    p3_gemm_desc gd;
    memset(&gd, 0, sizeof(gd));
    gd.M = (int)rows; gd.N = K2; gd.K = C; gd.lda = C; gd.ldb = C; gd.ldc = K2;
    gd.dtype_in = d->dtype; gd.dtype_out = d->dtype == P3_F32X3 ? P3_F32 : d->dtype; gd.act = P3_ACT_NONE; gd.a_mode = P3_A_PLAIN;    // P3_F32X3: the PFN products as bf16 x 3, storage fp32
    rc = p3_gemm(w.H2, w2t, w.X2, &gd, stream);
    if (rc != P3_OK) return rc;
    if (bf)
        hipLaunchKernelGGL((pfn_bwd_l1_kernel<bf16_t>), dim3(sgrid), dim3(256), 0, s, t, d->max_voxels, d->max_points, nslots, w.F8, (const bf16_t*)w.X2, w1, w.sc1, w.sh1, w.m1, w.r1, w.acc1);
    else {
        float* slab4 = p3_det_scratch((int64_t)sgrid * ACC1_FLOATS, kdt);
        hipLaunchKernelGGL((pfn_bwd_l1_kernel<float>), dim3(sgrid), dim3(256), 0, s, t, d->max_voxels, d->max_points, nslots, w.F8, (const float*)w.X2, w1, w.sc1, w.sh1, w.m1, w.r1, w.acc1, slab4);
        P3_LAUNCH_CHECK();
        if (slab4) { int rc4 = p3_det_reduce(slab4, sgrid, ACC1_FLOATS, w.acc1, ACC1_FLOATS, 1, s); if (rc4 != P3_OK) return rc4; }
    }
    P3_LAUNCH_CHECK();
    }
    if (phases & 4) {
    hipLaunchKernelGGL(pfn_bwd_l1_finalize_kernel, dim3(1), dim3(256), 0, s, w.acc1, stat1 ? stat1 : w.acc1 + ACC_DB, bn1_gamma, w.r1, w.totals,
                       d->max_points, d->training, dw1, dg1, db1);
    P3_LAUNCH_CHECK();
    }
    return P3_OK;
}

extern "C" int p3_pillar_stem_bwd(const void* dcanvas, int dcanvas_ld, const float* w1, const float* bn1_gamma, const void* w2t,
                                  const float* bn2_gamma, void* workspace, const p3_pillar_desc* d, float* dw1, float* dg1, float* db1,
                                  float* dw2, float* dg2, float* db2, void* stream) {
    return p3_pillar_stem_bwd_phased(dcanvas, dcanvas_ld, w1, bn1_gamma, w2t, bn2_gamma, workspace, d, dw1, dg1, db1, dw2, dg2, db2, nullptr,
                                     nullptr, 7, stream);
}

// byte offsets of the workspace sections (for the training path, which re-reads the pillar tables in backward)
extern "C" int p3_pillar_stem_layout(const p3_pillar_desc* d, int64_t* off /*[17]*/) {
    P3_CHECK(d && off, P3_EINVAL, "p3_pillar_stem_layout: null pointer");
    char* base = (char*)256;   // any non-null base: only differences are used
    Ws w = carve(base, d);
    off[0] = (char*)w.sorted - base; off[1] = (char*)w.vox_xy - base; off[2] = (char*)w.vox_start - base;
    off[3] = (char*)w.vox_cnt - base; off[4] = (char*)w.vox_row - base; off[5] = (char*)w.nvox - base;
    off[6] = (char*)w.X2 - base; off[7] = (char*)w.H2 - base; off[8] = (char*)w.hmax - base; off[9] = (char*)w.hmin - base;
    off[10] = (char*)w.F8 - base; off[11] = (char*)w.row_vox - base; off[12] = (char*)w.row_w - base;
    off[13] = (char*)w.totals - base; off[14] = (char*)w.sums1 - base; off[15] = (char*)w.sums2 - base; off[16] = (char*)w.acc1 - base;
    return P3_OK;
}
