/*
 * ORACLE (test infrastructure only - never linked or called by the product path).
 *
 * CPU restatement of the hard voxelisation ("pillarize") that the reference reaches through
 *   pixelspointspolygons/models/pointpillars/pointpillars_o3d.py:92   self.voxelize(x_lidar)
 * -> Open3D-ML 0.19.0  ml3d/torch/models/point_pillars.py  PointPillars.voxelize /
 *    PointPillarsVoxelization.forward  -> open3d ops `voxelize` + `ragged_to_dense`
 *    (cpp/open3d/ml/impl/misc/Voxelize.h).
 * Open3D is a third-party dependency that is NOT vendored under /root/reference
 * (pyproject.toml:23 pins open3d==0.19.0); its published algorithm is restated here:
 *
 *   per sample:
 *     a point is kept iff  range_min <= p <= range_max  on every axis (inclusive),
 *     pillar coord c = (int)((p - range_min) * (1/voxel_size)) (per axis; so p == range_max
 *         lands in index `extent`, one past the grid),
 *     points are grouped per pillar in ascending point index; at most `max_points`
 *         (the lowest indices) are kept; pillars are emitted in ascending linear hash
 *         h = cx + cy*ex + cz*ex*ey and at most `max_voxels` (lowest hashes) are kept,
 *     (python side) pillars with cx >= nx or cy >= ny are filtered out; coords are
 *         re-ordered to (z, y, x) and the batch index is prepended.
 *
 * PARITY UNPINNED for this function: the reference ships no test vector at this boundary
 * and open3d cannot be imported here; the decisions above are documented in DESIGN.md.
 *
 * Outputs (dense, capacity = batch * max_voxels pillars):
 *   coors      [V,4] int32  (b, z, y, x)
 *   num_points [V]   int32
 *   point_idx  [V,max_points] int32  index into the sample-local point list, -1 = empty slot
 * returns V.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int64_t hash; int32_t idx; } hp_t;

static int cmp_hp(const void* a, const void* b) {
    const hp_t* x = (const hp_t*)a; const hp_t* y = (const hp_t*)b;
    if (x->hash != y->hash) return x->hash < y->hash ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int p3o_pillarize(const float* points,      /* [sumN,3] */
                  const int64_t* offsets,   /* [B+1]    */
                  int batch,
                  const float* voxel_size,  /* [3] */
                  const float* range_min,   /* [3] */
                  const float* range_max,   /* [3] */
                  int max_points, int max_voxels,
                  int32_t* coors, int32_t* num_points, int32_t* point_idx)
{
    int ext[3];
    for (int a = 0; a < 3; ++a) {
        float e = (range_max[a] - range_min[a]) / voxel_size[a];
        int ie = (int)e; if ((float)ie < e) ie += 1;   /* ceil */
        ext[a] = ie;
    }
    int V = 0;
    for (int b = 0; b < batch; ++b) {
        int64_t lo = offsets[b], hi = offsets[b + 1];
        int64_t n = hi - lo;
        hp_t* hp = (hp_t*)malloc(sizeof(hp_t) * (size_t)(n > 0 ? n : 1));
        int64_t m = 0;
        for (int64_t i = 0; i < n; ++i) {
            const float* p = points + 3 * (lo + i);
            int ok = 1; int c[3];
            for (int a = 0; a < 3; ++a) {
                if (!(p[a] >= range_min[a] && p[a] <= range_max[a])) ok = 0;
                c[a] = (int)((p[a] - range_min[a]) * (1.0f / voxel_size[a]));
            }
            if (!ok) continue;
            /* extents + 1 so that the p == range_max cell gets its own hash */
            hp[m].hash = (int64_t)c[0] + (int64_t)c[1] * (ext[0] + 1) + (int64_t)c[2] * (ext[0] + 1) * (ext[1] + 1);
            hp[m].idx = (int32_t)i;
            ++m;
        }
        qsort(hp, (size_t)m, sizeof(hp_t), cmp_hp);
        int nvox = 0;
        for (int64_t s = 0; s < m;) {
            int64_t e = s; while (e < m && hp[e].hash == hp[s].hash) ++e;
            if (nvox < max_voxels) {
                int64_t h = hp[s].hash;
                int cx = (int)(h % (ext[0] + 1)); h /= (ext[0] + 1);
                int cy = (int)(h % (ext[1] + 1)); int cz = (int)(h / (ext[1] + 1));
                ++nvox; /* counts against max_voxels before the python-side bounds filter */
                if (cx < ext[0] && cy < ext[1]) {
                    int cnt = (int)(e - s); if (cnt > max_points) cnt = max_points;
                    coors[4 * V + 0] = b; coors[4 * V + 1] = cz; coors[4 * V + 2] = cy; coors[4 * V + 3] = cx;
                    num_points[V] = cnt;
                    for (int k = 0; k < max_points; ++k)
                        point_idx[(int64_t)V * max_points + k] = k < cnt ? hp[s + k].idx : -1;
                    ++V;
                }
            }
            s = e;
        }
        free(hp);
    }
    return V;
}
