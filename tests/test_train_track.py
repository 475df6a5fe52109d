"""Row 17 of SURVEY.md 8(a): the caller contract `example_to_device` / `track_batch_processor`
(det3d/torchie/apis/train_track.py:29-74, :109-130): which keys are cast to fp32 (integer coordinates and counts included),
which are moved element-wise, which pass through, and the two return forms."""
import numpy as np
import pytest
import torch

from shasta_amd import train_track as TT


def _example(B=2, N=6):
    g = torch.Generator().manual_seed(0)
    return dict(
        voxels=torch.rand(40, 10, 5, generator=g, dtype=torch.float64), prev_voxels=torch.rand(30, 10, 5, generator=g),
        coordinates=torch.randint(0, 100, (40, 4), generator=g, dtype=torch.int32),
        prev_coordinates=torch.randint(0, 100, (30, 4), generator=g, dtype=torch.int32),
        num_points=torch.randint(1, 10, (40,), generator=g, dtype=torch.int32),
        prev_num_points=torch.randint(1, 10, (30,), generator=g, dtype=torch.int32),
        num_voxels=torch.tensor([25, 15], dtype=torch.int64), prev_num_voxels=torch.tensor([20, 10], dtype=torch.int64),
        det_boxes=torch.rand(B, N, 11, generator=g, dtype=torch.float64), prev_det_boxes=torch.rand(B, N, 11, generator=g, dtype=torch.float64),
        gt=(torch.rand(B, N + 2, N + 2, generator=g) < 0.2).double(),
        points=[torch.rand(7, 5, generator=g), torch.rand(9, 5, generator=g)], prev_points=[torch.rand(3, 5, generator=g)] * 2,
        shape=np.array([[1440, 1440, 40]] * B), metadata=[{"token": "a"}, {"token": "b"}], prev_metadata=[{"token": ""}, {"token": "a"}],
        cls_det_boxes=[[{"x": 1}], []], prev_cls_det_boxes=[[], [{"x": 1}]], num_det_boxes=[1, 0], num_prev_det_boxes=[0, 1],
        calib={"rect": np.eye(4)})


def test_example_to_device_casts_and_passthrough():
    ex = _example()
    out = TT.example_to_device(ex, torch.device("cpu"))
    assert set(out.keys()) == set(ex.keys())
    for k in ("voxels", "prev_voxels", "coordinates", "prev_coordinates", "num_points", "prev_num_points", "num_voxels", "prev_num_voxels",
              "det_boxes", "prev_det_boxes", "gt"):
        assert out[k].dtype == torch.float32, k                       # train_track.py:60: every listed tensor becomes fp32
        np.testing.assert_allclose(out[k].double().numpy(), ex[k].double().numpy(), rtol=1e-7)
    assert out["coordinates"].dtype == torch.float32 and ex["coordinates"].dtype == torch.int32  # integers included, input untouched
    for k in ("num_det_boxes", "num_prev_det_boxes"):                  # :69-70 lists become fp32 tensors
        assert torch.is_tensor(out[k]) and out[k].dtype == torch.float32 and out[k].tolist() == [float(v) for v in ex[k]]
    for k in ("points", "prev_points"):                                # :36-37 moved element-wise, dtype kept
        assert isinstance(out[k], list) and all(torch.equal(a, b) for a, b in zip(out[k], ex[k]))
    for k in ("shape", "metadata", "prev_metadata", "cls_det_boxes", "prev_cls_det_boxes"):
        assert out[k] is ex[k]                                         # :71-72 everything else is passed through
    assert torch.is_tensor(out["calib"]["rect"])                       # :62-68


class _Echo(torch.nn.Module):
    def forward(self, example, train_mode=True):
        self.seen = (example, train_mode)
        B, N = example["det_boxes"].shape[:2]
        return torch.zeros(B, N, N + 2), torch.ones(B, N + 2, N), example


def test_track_batch_processor_return_forms():
    """(matched1, matched2, gt) in train mode, (matched1, matched2, example) otherwise (train_track.py:122-130).  The device
    move is pointed at the CPU here; the GPU test below runs the real thing."""
    ex = _example()
    model = _Echo()
    real = TT.example_to_device
    TT.example_to_device = lambda data, device, non_blocking=False: real(data, torch.device("cpu"))
    try:
        m1, m2, gt = TT.track_batch_processor(model, ex, train_mode=True)
        assert gt.dtype == torch.float32 and torch.equal(gt, ex["gt"].float()) and model.seen[1] is True
        m1, m2, out = TT.track_batch_processor(model, ex, train_mode=False)
        assert isinstance(out, dict) and out is model.seen[0] and model.seen[1] is False
        assert out["det_boxes"].dtype == torch.float32 and m1.shape == (2, 6, 8) and m2.shape == (2, 8, 6)
    finally:
        TT.example_to_device = real


@pytest.mark.gpu
def test_track_batch_processor_drives_the_hip_model():
    """The reference's call (eval.py:113 / train.py:198) on the device: host batch in, fp32 device tensors to the model, the
    in-place back-projection of det_boxes visible in the returned example, gt returned in train mode."""
    from oracle import shasta_oracle as O
    from tests.helpers import build_model, load_golden
    z, c, sums = load_golden("small_32_3_5_pad")
    m = build_model(c)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    bev, pbev, det, prev = O.synth_case(c["B"], c["max_obj"], c["n_real"], c["cin"], c["hw"], c["hw"], c["seed"])
    m = m.cuda()
    data = dict(bev_map=bev.double(), prev_bev_map=pbev.double(), det_boxes=det.double(), prev_det_boxes=prev.double(),
                gt=torch.zeros(c["B"], c["max_obj"] + 2, c["max_obj"] + 2, dtype=torch.float64), metadata=[{"token": "t"}])
    with torch.no_grad():
        m1, m2, ex = TT.track_batch_processor(m, data, train_mode=False, local_rank=0)
    assert ex["det_boxes"].is_cuda and ex["det_boxes"].dtype == torch.float32 and ex["metadata"] is data["metadata"]
    np.testing.assert_allclose(m1.cpu().numpy(), z["m1"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(m2.cpu().numpy(), z["m2"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ex["det_boxes"].cpu().numpy(), z["det_boxes_out"], rtol=0, atol=1e-6)  # shasta.py:270 side effect
    assert torch.equal(data["det_boxes"], det.double())  # the host batch itself is not touched (a device copy was)
    m.train()
    for p in m.shared_conv.parameters():
        p.requires_grad_(False)
    m1, m2, gt = TT.track_batch_processor(m, data, train_mode=True, local_rank=0)
    assert gt.is_cuda and gt.dtype == torch.float32 and m1.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_reader_on_device_matches_reference(case):
    """VoxelFeatureExtractorV3 through shasta_voxel_mean_f32 against the reference reader's output (voxelize.npz)."""
    import os
    import shasta_amd
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "voxelize.npz"))
    reader = shasta_amd.VoxelFeatureExtractorV3(num_input_features=5)
    v = torch.from_numpy(z[case + "_voxels"]).cuda()
    n = torch.from_numpy(z[case + "_num"]).cuda().float()  # example_to_device has cast the counts
    out = reader(v, n)
    np.testing.assert_allclose(out.cpu().numpy(), z[case + "_mean"], rtol=1e-6, atol=1e-6)
    from shasta_amd import hip
    with pytest.raises(hip.ShastaHipError):
        reader(v.cpu(), n.cpu())
