"""ctypes wrapper of oracle/voxelize_oracle.c (TEST INFRASTRUCTURE; see that file's header)."""
import ctypes as C

import numpy as np

from . import build_oracle

_lib = None
_maps = {}


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle.build())
        _lib.shasta_oracle_points_to_voxel.restype = C.c_int
        _lib.shasta_oracle_points_to_voxel.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.shasta_oracle_voxel_mean.restype = None
        _lib.shasta_oracle_voxel_mean.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    return _lib


def points_to_voxel(points, voxel_size, coors_range, max_points, max_voxels, with_mean=False):
    """Same contract as the reference's points_to_voxel(points, voxel_size, coors_range, max_points, True, max_voxels)
    (det3d/ops/point_cloud/point_cloud_ops.py:112-184): returns voxels (V,max_points,ndim), coors (V,3) zyx, num (V,)."""
    lib = _load()
    points = np.ascontiguousarray(points, np.float32)
    vs = np.ascontiguousarray(voxel_size, np.float32)
    rg = np.ascontiguousarray(coors_range, np.float32)
    n, ndim = points.shape
    grid = np.round((rg[3:] - rg[:3]) / vs).astype(np.int64)
    key = tuple(grid.tolist())
    if key not in _maps:
        _maps[key] = -np.ones(int(grid.prod()), np.int32)
    cmap = _maps[key]
    voxels = np.zeros((max_voxels, max_points, ndim), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    v = lib.shasta_oracle_points_to_voxel(points.ctypes.data, n, ndim, vs.ctypes.data, rg.ctypes.data, max_points,
                                          max_voxels, voxels.ctypes.data, coors.ctypes.data, num.ctypes.data,
                                          cmap.ctypes.data)
    out = (voxels[:v], coors[:v], num[:v])
    if with_mean:
        mean = np.zeros((v, ndim), np.float32)
        if v:
            lib.shasta_oracle_voxel_mean(voxels.ctypes.data, num.ctypes.data, v, max_points, ndim, mean.ctypes.data)
        out = out + (mean,)
    return out
