"""VoxelFeatureExtractorV3 (det3d/models/readers/voxel_encoder.py:9-28): the reader that averages the points of a voxel.

On the HIP path the mean comes out of the voxeliser itself (`shasta_voxelize_mean_f32`, voxel_generator.generate_device);
this module keeps the reference's reader interface - constructor keywords, `forward(features, num_voxels, coors)` - for
callers that already hold zero-padded (V, max_points, C) voxel tensors and their point counts."""
import torch
from torch import nn

from . import hip
from .registry import READERS


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name, self.num_input_features = name, num_input_features

    def forward(self, features, num_voxels, coors=None):
        c = self.num_input_features
        if features.shape[-1] != c:
            raise AssertionError("expected %d point features, got %d" % (c, features.shape[-1]))
        if not features.is_cuda:
            raise hip.ShastaHipError("VoxelFeatureExtractorV3 needs device tensors (example_to_device); there is no CPU path")
        feats = features.float().contiguous()
        counts = num_voxels.to(device=feats.device, dtype=torch.float32).reshape(-1).contiguous()
        V, P, nd = feats.shape
        out = torch.empty(V, c, device=feats.device)
        hip.check(hip.load().shasta_voxel_mean_f32(hip.ptr(feats), hip.ptr(counts), V, P, nd, c, hip.ptr(out), hip.stream_ptr()),
                  "shasta_voxel_mean_f32")
        return out
