/* CPU oracle (TEST INFRASTRUCTURE) for the hard voxeliser: a plain C restatement of the serial loop of
 * det3d/ops/point_cloud/point_cloud_ops.py:7-55 (_points_to_voxel_reverse_kernel) with the set-up of :112-184
 * (points_to_voxel, reverse_index=True).  Pinned against tests/golden/voxelize.npz (outputs of the reference itself).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Returns the number of voxels.  coor_to_voxelidx must hold gz*gy*gx int32 all set to -1 on entry; entries touched by
 * this call are reset to -1 before returning (so the caller can reuse the map; the reference re-allocates it).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

int shasta_oracle_points_to_voxel(const float* points, int n, int ndim, const float* voxel_size,
                                  const float* coors_range, int max_points, int max_voxels, float* voxels,
                                  int32_t* coors, int32_t* num_points_per_voxel, int32_t* coor_to_voxelidx) {
    int32_t grid[3];
    for (int j = 0; j < 3; ++j) {
        float span = coors_range[3 + j] - coors_range[j];
        grid[j] = (int32_t)lrintf(span / voxel_size[j]); /* np.round: half to even */
    }
    int voxel_num = 0;
    for (int i = 0; i < n; ++i) {
        int32_t coor[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            float c = floorf((points[(size_t)i * ndim + j] - coors_range[j]) / voxel_size[j]);
            if (c < 0 || c >= (float)grid[j]) {
                failed = 1;
                break;
            }
            coor[2 - j] = (int32_t)c; /* reversed: z, y, x */
        }
        if (failed) continue;
        size_t cell = ((size_t)coor[0] * grid[1] + coor[1]) * grid[0] + coor[2];
        int32_t vid = coor_to_voxelidx[cell];
        if (vid == -1) {
            if (voxel_num >= max_voxels) continue;
            vid = voxel_num++;
            coor_to_voxelidx[cell] = vid;
            coors[vid * 3 + 0] = coor[0];
            coors[vid * 3 + 1] = coor[1];
            coors[vid * 3 + 2] = coor[2];
        }
        int32_t num = num_points_per_voxel[vid];
        if (num < max_points) {
            memcpy(voxels + ((size_t)vid * max_points + num) * ndim, points + (size_t)i * ndim, sizeof(float) * ndim);
            num_points_per_voxel[vid] = num + 1;
        }
    }
    for (int v = 0; v < voxel_num; ++v) {
        size_t cell = ((size_t)coors[v * 3] * grid[1] + coors[v * 3 + 1]) * grid[0] + coors[v * 3 + 2];
        coor_to_voxelidx[cell] = -1;
    }
    return voxel_num;
}

/* det3d/models/readers/voxel_encoder.py:18-28: sum over the point slots / count */
void shasta_oracle_voxel_mean(const float* voxels, const int32_t* num, int nvox, int max_points, int ndim, float* mean) {
    for (int v = 0; v < nvox; ++v)
        for (int c = 0; c < ndim; ++c) {
            float s = 0.0f;
            for (int r = 0; r < max_points; ++r) s += voxels[((size_t)v * max_points + r) * ndim + c];
            mean[(size_t)v * ndim + c] = s / (float)num[v];
        }
}
