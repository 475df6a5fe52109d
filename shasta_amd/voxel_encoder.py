"""VoxelFeatureExtractorV3 (det3d/models/readers/voxel_encoder.py:9-28): the reader that averages the points of a voxel.

On the HIP path the mean comes out of the voxeliser itself (`shasta_voxelize_mean_f32`, voxel_generator.generate_device);
this module keeps the reference's reader interface - constructor keywords, `forward(features, num_voxels, coors)` - for
callers that already hold zero-padded (V, max_points, C) voxel tensors and their point counts."""
from torch import nn

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
        counts = num_voxels.to(features.dtype).reshape(-1, 1)  # padded slots are zero: the slot sum is the point sum
        return (features[..., :c].sum(dim=1) / counts).contiguous()
