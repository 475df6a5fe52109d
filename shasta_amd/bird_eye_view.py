"""BEVFeatureExtractor: same constructor and forward contract as
det3d/models/second_stage/bird_eye_view.py:10-41, computed by the HIP gather kernel (csrc/bev_gather.hip)."""
import torch
from torch import nn

from . import hip
from .registry import SECOND_STAGE


@SECOND_STAGE.register_module
class BEVFeatureExtractor(nn.Module):
    def __init__(self, pc_start, voxel_size, out_stride):
        super().__init__()
        self.pc_start = pc_start
        self.voxel_size = voxel_size
        self.out_stride = out_stride

    def _geom(self):
        return (float(self.pc_start[0]), float(self.pc_start[1]), float(self.voxel_size[0]),
                float(self.voxel_size[1]), float(self.out_stride))

    def gather_boxes(self, bev_nhwc, boxes, num_point, out):
        """Fused path used by Shasta.forward.  bev_nhwc (B,H,W,C) fp32, boxes (B,N,S>=7) fp32 rows
        [x,y,z,w,l,h,yaw,...]; writes rows [0,N) of the (B, T, num_point*C) table `out` in place."""
        lib = hip.load()
        B, H, W, Cc = bev_nhwc.shape
        N, S = boxes.shape[1], boxes.shape[2]
        assert out.shape[0] == B and out.shape[2] == num_point * Cc and out.shape[1] >= N
        x0, y0, vx, vy, st = self._geom()
        hip.check(lib.shasta_bev_gather_f32(hip.ptr(bev_nhwc), B, H, W, Cc, hip.ptr(boxes), N, S, N * S, num_point,
                                            x0, y0, vx, vy, st, hip.ptr(out), out.shape[2],
                                            out.shape[1] * out.shape[2], hip.stream_ptr()), "shasta_bev_gather_f32")
        return out

    def forward(self, example, batch_centers, num_point):
        """example['bev_feature'] (B,H,W,C); batch_centers: list[B] of (num_point*N, 3) points, point-type-major
        (Shasta.get_box_center).  Returns list[B] of (N, num_point*C), as bird_eye_view.py:24-41."""
        bev = example["bev_feature"]
        lib = hip.load()
        ret = []
        x0, y0, vx, vy, st = self._geom()
        for b in range(len(bev)):
            im = bev[b].contiguous().float()
            pts = batch_centers[b].contiguous().float()
            H, W, Cc = im.shape
            n = pts.shape[0]
            fm = torch.empty(n, Cc, device=im.device, dtype=torch.float32)
            # every point is a 1-point "box": the kernel only reads x, y
            hip.check(lib.shasta_bev_gather_f32(hip.ptr(im), 1, H, W, Cc, hip.ptr(pts), n, pts.shape[1],
                                                n * pts.shape[1], 1, x0, y0, vx, vy, st, hip.ptr(fm), Cc, n * Cc,
                                                hip.stream_ptr()), "shasta_bev_gather_f32")
            if num_point > 1:
                sec = n // num_point
                fm = torch.cat([fm[i * sec:(i + 1) * sec] for i in range(num_point)], dim=1)
            ret.append(fm)
        return ret
