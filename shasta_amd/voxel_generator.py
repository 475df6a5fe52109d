"""VoxelGenerator / points_to_voxel with the reference's interface, computed on the GPU.

Mirrors det3d/core/input/voxel_generator.py:5-46 and det3d/ops/point_cloud/point_cloud_ops.py:112-184
(`points_to_voxel(points, voxel_size, coors_range, max_points, reverse_index=True, max_voxels)`), the CPU numba loop
that the reference runs in 48 DataLoader workers.  Here the scatter (and the per-voxel mean of
VoxelFeatureExtractorV3, det3d/models/readers/voxel_encoder.py:18-28) is one call of `shasta_voxelize_mean_f32` - or, for the
current and previous clouds of a whole batch, of `shasta_voxelize_mean_batch_f32`; outputs are bit-identical to the serial loop
(voxel order, kept points, coordinates, counts).  The reference's dense cell map (332 MB per call) is a hash table of 8 MB per
cloud inside the call's workspace.
"""
import ctypes as C

import numpy as np
import torch

from . import hip


def points_to_voxel_device(points, voxel_size, coors_range, max_points=35, max_voxels=20000, with_mean=False, sync=True):
    """points: (P, ndim) fp32 DEVICE tensor.  Returns device tensors (voxels (V,max_points,ndim), coors (V,3) zyx int32,
    num_points (V,) int32[, mean (V,ndim)]) -- one host sync to read V.  sync=False: nothing is read back - the tensors come with all
    max_voxels rows and the count as a (1,) int32 device tensor appended (rows >= V are not written)."""
    lib = hip.load()
    if not points.is_cuda:
        raise hip.ShastaHipError("points_to_voxel_device needs a device tensor (no CPU path)")
    points = points.float().contiguous()
    P, ndim = points.shape
    vs = np.ascontiguousarray(voxel_size, np.float32)
    rg = np.ascontiguousarray(coors_range, np.float32)
    dev = points.device
    voxels = torch.empty(max_voxels, max_points, ndim, device=dev)
    coors = torch.empty(max_voxels, 3, dtype=torch.int32, device=dev)
    num = torch.empty(max_voxels, dtype=torch.int32, device=dev)
    mean = torch.empty(max_voxels, ndim, device=dev) if with_mean else None
    nv = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = lib.shasta_voxelize_workspace_bytes(P, max_voxels, max_points)
    ws = torch.empty((wsb + 3) // 4, dtype=torch.int32, device=dev)
    hip.check(lib.shasta_voxelize_mean_f32(hip.ptr(points), P, ndim, rg.ctypes.data_as(C.c_void_p),
                                           vs.ctypes.data_as(C.c_void_p), max_points, max_voxels, hip.ptr(voxels),
                                           hip.ptr(coors), hip.ptr(num), hip.ptr(mean), hip.ptr(nv),
                                           hip.ptr(ws), wsb, hip.stream_ptr()), "shasta_voxelize_mean_f32")
    if not sync:
        return (voxels, coors, num) + ((mean,) if with_mean else ()) + (nv,)
    V = int(nv.item())
    out = (voxels[:V], coors[:V], num[:V])
    if with_mean:
        out = out + (mean[:V],)
    return out


def points_to_voxel_batch_device(clouds, voxel_size, coors_range, max_points=35, max_voxels=20000, with_mean=True):
    """The clouds of a batch (the current and the previous cloud of every sample: preprocess.py:179-208 voxelises both) in ONE chain of
    launches, nothing read back.  clouds: list of (P_i, ndim) fp32 device tensors, or a tuple (points (sum P, ndim), offsets) with the
    clouds back to back and `offsets` a host sequence of len(clouds) + 1 row indices.  Returns device tensors with a leading cloud axis:
    voxels (n, max_voxels, max_points, ndim), coors (n, max_voxels, 3) zyx, num_points (n, max_voxels), mean (n, max_voxels, ndim) or
    None, num_voxels (n,) int32 - rows >= num_voxels[c] of cloud c are not written.  Up to 32 clouds per call."""
    lib = hip.load()
    if isinstance(clouds, tuple):
        points, offsets = clouds
        offsets = [int(o) for o in offsets]
    else:
        offsets = [0]
        for c in clouds:
            offsets.append(offsets[-1] + int(c.shape[0]))
        points = torch.cat([c.float() for c in clouds], dim=0) if len(clouds) > 1 else clouds[0]
    if not points.is_cuda:
        raise hip.ShastaHipError("points_to_voxel_batch_device needs device tensors (no CPU path)")
    points = points.float().contiguous()
    n, ndim = len(offsets) - 1, points.shape[1]
    if not 1 <= n <= 32:
        raise ValueError("1 to 32 clouds per call")
    vs = np.ascontiguousarray(voxel_size, np.float32)
    rg = np.ascontiguousarray(coors_range, np.float32)
    dev = points.device
    voxels = torch.empty(n, max_voxels, max_points, ndim, device=dev)
    coors = torch.empty(n, max_voxels, 3, dtype=torch.int32, device=dev)
    num = torch.empty(n, max_voxels, dtype=torch.int32, device=dev)
    mean = torch.empty(n, max_voxels, ndim, device=dev) if with_mean else None
    nv = torch.empty(n, dtype=torch.int32, device=dev)
    off = (C.c_int * (n + 1))(*offsets)
    wsb = lib.shasta_voxelize_batch_workspace_bytes(off, n, max_voxels, max_points)
    ws = torch.empty((wsb + 3) // 4, dtype=torch.int32, device=dev)
    hip.check(lib.shasta_voxelize_mean_batch_f32(hip.ptr(points), off, n, ndim, rg.ctypes.data_as(C.c_void_p), vs.ctypes.data_as(C.c_void_p),
                                                 max_points, max_voxels, hip.ptr(voxels), hip.ptr(coors), hip.ptr(num), hip.ptr(mean),
                                                 hip.ptr(nv), hip.ptr(ws), wsb, hip.stream_ptr()), "shasta_voxelize_mean_batch_f32")
    return voxels, coors, num, mean, nv


def points_to_voxel(points, voxel_size, coors_range, max_points=35, reverse_index=True, max_voxels=20000):
    """Drop-in for the reference function: numpy in, numpy out (voxels, coordinates zyx, num_points_per_voxel)."""
    if not reverse_index:
        raise NotImplementedError("only reverse_index=True is used by the reference pipeline (voxel_generator.py:28)")
    dev = torch.device("cuda", torch.cuda.current_device())
    pts = torch.from_numpy(np.ascontiguousarray(points, np.float32)).to(dev)
    v, c, n = points_to_voxel_device(pts, voxel_size, coors_range, max_points, max_voxels)
    return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()


class VoxelGenerator:
    """det3d/core/input/voxel_generator.py:5-46."""

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        voxel_size = np.array(voxel_size, dtype=np.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        self._grid_size = np.round(grid_size).astype(np.int64)
        self._voxel_size = voxel_size
        self._point_cloud_range = point_cloud_range
        self._max_num_points = max_num_points
        self._max_voxels = max_voxels

    def generate(self, points, max_voxels=-1):
        if max_voxels == -1:
            max_voxels = self._max_voxels
        return points_to_voxel(points, self._voxel_size, self._point_cloud_range, self._max_num_points, True, max_voxels)

    def generate_device(self, points, max_voxels=-1, with_mean=True, sync=True):
        """sync=False: the voxel count stays on the device (last element of the returned tuple), the tensors keep all max_voxels rows."""
        if max_voxels == -1:
            max_voxels = self._max_voxels
        return points_to_voxel_device(points, self._voxel_size, self._point_cloud_range, self._max_num_points,
                                      max_voxels, with_mean, sync=sync)

    def generate_batch_device(self, clouds, max_voxels=-1, with_mean=True):
        """Every cloud of a batch (current + previous cloud of every sample) in one chain of launches; the voxel counts stay on the
        device (points_to_voxel_batch_device): no host synchronisation unless the caller reads them."""
        if max_voxels == -1:
            max_voxels = self._max_voxels
        return points_to_voxel_batch_device(clouds, self._voxel_size, self._point_cloud_range, self._max_num_points, max_voxels, with_mean)

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size
