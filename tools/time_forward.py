"""Times the affinity forward (rows 4-16) at an arbitrary configuration, e.g. the reference's shipped car config
(max_obj 90, 3 features, 5 points -> F = 320).  usage: python tools/time_forward.py [--max-obj 90] [--feats 3] [--points 5] [--batch 1 8 64]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=90)
ap.add_argument("--feats", type=int, default=3)
ap.add_argument("--points", type=int, default=5)
ap.add_argument("--batch", type=int, nargs="+", default=[1, 8, 64])
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph of the step")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
with torch.device(dev):
    model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                             bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                             max_obj=a.max_obj, num_feats=a.feats, num_point=a.points)).eval()
N = a.max_obj
for B in a.batch:
    g = torch.Generator(device=dev).manual_seed(B)
    bev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
    pbev = torch.relu(torch.randn(B, 180, 180, 64, device=dev, generator=g))
    det0 = torch.zeros(B, N, 11, device=dev)
    det0[..., :2] = torch.rand(B, N, 2, device=dev, generator=g) * 100 - 50
    det0[..., 3:6] = torch.rand(B, N, 3, device=dev, generator=g) * 4 + 0.5
    det0[..., 6] = torch.rand(B, N, device=dev, generator=g) * 6.28 - 3.14
    det0[..., 7:9] = torch.randn(B, N, 2, device=dev, generator=g)
    det0[..., 9] = 0.5
    prev = det0.roll(1, 1).contiguous()
    det = det0.clone()
    def step():
        det.copy_(det0)
        return model.affinity_from_bev(bev, pbev, det, prev)

    graph = None
    with torch.no_grad():
        if a.graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                m1, m2 = step()
        for it in range(a.steps + 5):
            if it == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if graph is not None:
                graph.replay()
            else:
                m1, m2 = step()
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print("max_obj=%d F=%d nf=%d B=%-3d %8.3f ms/step %10.1f frame-pairs/s" % (N, model.aug_shape_output, a.feats, B, dt * 1e3, B / dt), flush=True)
