"""VoxelFeatureExtractorV3 (det3d/models/readers/voxel_encoder.py:9-28): per-voxel mean of the point slots.

On the HIP path the mean is produced by the voxeliser itself (shasta_voxelize_mean_f32); this module keeps the
reference's reader interface for callers that already hold (voxels, num_points) tensors."""
import torch
from torch import nn

from .registry import READERS


@READERS.register_module
class VoxelFeatureExtractorV3(nn.Module):
    def __init__(self, num_input_features=4, norm_cfg=None, name="VoxelFeatureExtractorV3"):
        super().__init__()
        self.name = name
        self.num_input_features = num_input_features

    def forward(self, features, num_voxels, coors=None):
        assert self.num_input_features == features.shape[-1]
        s = features[:, :, : self.num_input_features].sum(dim=1, keepdim=False)
        return (s / num_voxels.type_as(features).view(-1, 1)).contiguous()
