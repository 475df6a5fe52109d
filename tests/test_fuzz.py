"""Seeded random configurations of the whole path against the oracle (SHASTA_FUZZ_SEED / SHASTA_FUZZ_CASES and SHASTA_FUZZ_GRAD_SEED /
SHASTA_FUZZ_GRAD_CASES widen the forward / backward sweeps for a one-off run): table sizes, feature counts, point counts, batch sizes,
padding, map sizes, convolution shapes.  The fixed-shape tests cover the shipped configurations; this sweep is there for the
seams between kernel variants (batch-size and width thresholds, partial tiles, odd sizes)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import shasta_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _model(N, nf, npnt, cin, seed, stride=8):
    import shasta_amd
    torch.manual_seed(seed)
    return shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                            bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075],
                                                               out_stride=stride),
                                            max_obj=N, num_feats=nf, num_point=npnt, in_channels=cin)).eval()


def _cases(n, seed):
    rnd = random.Random(seed)
    out = []
    for k in range(n):
        N = rnd.choice([1, 2, 3, 5, 7, 11, 16, 17, 23, 31, 32, 33, 47, 63, 64, 65, 70])
        nf = rnd.randint(1, 7)
        npnt = rnd.choice([1, 4, 5])
        B = rnd.choice([1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 31, 32, 33, 40, 64, 65])
        if N * B > 1200:
            B = max(1, 1200 // N)
        n_real = rnd.choice([None, None, 0, max(0, N // 2), max(0, N - 1)])
        hw = rnd.choice([24, 45, 90, 180])
        out.append((N, nf, npnt, B, n_real, hw, 1000 + k))
    return out


BIGGER = [(90, 3, 5, 17, 40, 180, 1), (100, 7, 4, 33, None, 90, 2), (130, 5, 1, 3, 100, 180, 3), (77, 2, 5, 65, 70, 45, 4), (129, 7, 4, 2, None, 180, 5),
          (100, 3, 1, 165, 90, 45, 6), (126, 7, 1, 131, None, 24, 7), (50, 6, 4, 129, 10, 90, 8), (64, 1, 5, 70, None, 180, 9),
          (200, 7, 1, 16, 150, 180, 10), (40, 4, 4, 48, 39, 45, 11)]


@pytest.mark.parametrize("N,nf,npnt,B,n_real,hw,seed", _cases(int(os.environ.get("SHASTA_FUZZ_CASES", 28)), int(os.environ.get("SHASTA_FUZZ_SEED", 2024))) + BIGGER)
def test_forward_random_configs_vs_oracle(N, nf, npnt, B, n_real, hw, seed):
    dev = torch.device("cuda:0")
    stride = 8 * 180 // hw  # the map always spans the same metric extent
    m = _model(N, nf, npnt, 8, seed, stride)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    bev = torch.relu(torch.randn(B, hw, hw, 64, generator=g))
    pbev = torch.relu(torch.randn(B, hw, hw, 64, generator=g))
    det, prev = O.synth_boxes(g, B, N, n_real), O.synth_boxes(g, B, N, n_real)
    d_ref = det.clone()
    r1, r2 = O.forward_from_bev(w, bev, pbev, d_ref, prev.clone(), nf, npnt, out_stride=stride)
    m = m.to(dev)
    d_dev = det.to(dev)
    with torch.no_grad():
        m1, m2 = m.affinity_from_bev(bev.to(dev), pbev.to(dev), d_dev, prev.to(dev))
    assert bool(torch.isfinite(m1).all()) and bool(torch.isfinite(m2).all())
    np.testing.assert_allclose(m1.cpu().numpy(), r1.numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(m2.cpu().numpy(), r2.numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(d_dev.cpu().numpy(), d_ref.numpy(), rtol=0, atol=1e-6)  # in-place back-projection


def _conv_cases(n, seed):
    rnd = random.Random(seed)
    return [(rnd.choice([1, 2, 3]), rnd.choice([8, 16, 24, 40, 64]), rnd.randint(1, 40), rnd.choice([1, 2, 3, 7, 31, 32, 33, 64, 100, 127, 128, 129, 199,
                                                                                                      255, 256, 257, 320]), 3000 + k) for k in range(n)]


@pytest.mark.parametrize("B,cin,H,W,seed", _conv_cases(16, 77))
def test_shared_conv_random_shapes_vs_oracle(B, cin, H, W, seed):
    dev = torch.device("cuda:0")
    m = _model(4, 7, 1, cin, seed)
    with torch.no_grad():
        m.shared_conv[1].running_mean.copy_(torch.randn(64) * 0.3)
        m.shared_conv[1].running_var.copy_(torch.rand(64) + 0.5)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    x, xp = torch.relu(torch.randn(B, cin, H, W, generator=g)), torch.relu(torch.randn(B, cin, H, W, generator=g))
    ref, refp = O.shared_conv_nhwc(w, x), O.shared_conv_nhwc(w, xp)
    m = m.to(dev)
    with torch.no_grad():
        got, gotp = m.shared_conv_nhwc(x.to(dev), xp.to(dev))
    scale = max(1.0, float(ref.abs().max()))
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(gotp.cpu().numpy(), refp.numpy(), rtol=1e-4, atol=2e-5 * scale)


def _grad_cases(n, seed):
    rnd = random.Random(seed)
    sizes = [1, 3, 5, 9, 16, 21] + ([33, 47, 62, 63, 64, 65, 90, 127, 129] if os.environ.get("SHASTA_FUZZ_GRAD_BIG") else [])  # (one-off sweeps: tile seams of pair_bwd.hip)
    return [(rnd.choice(sizes), rnd.randint(1, 7), rnd.choice([1, 4, 5]), rnd.choice([1, 2, 5, 17]), 4000 + k) for k in range(n)]


@pytest.mark.parametrize("N,nf,npnt,B,seed", _grad_cases(int(os.environ.get("SHASTA_FUZZ_GRAD_CASES", 8)), int(os.environ.get("SHASTA_FUZZ_GRAD_SEED", 9))) + [(100, 7, 4, 2, 4100), (130, 3, 5, 3, 4101), (64, 7, 1, 20, 4102)])
def test_backward_random_configs_vs_oracle_autograd(N, nf, npnt, B, seed):
    from shasta_amd import training
    dev = torch.device("cuda:0")
    m = _model(N, nf, npnt, 8, seed, 64)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    a = torch.relu(torch.randn(B, 24, 24, 64, generator=g))
    b = torch.relu(torch.randn(B, 24, 24, 64, generator=g))
    det, prev = O.synth_boxes(g, B, N, None), O.synth_boxes(g, B, N, max(0, N - 2))
    gt = (torch.rand(B, N + 2, N + 2, generator=g) < 0.2).float()
    gt[:, 0, 0] = 1.0
    wl = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.startswith("shared_conv")) for k, v in w.items()}
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    r1, r2 = O.forward_from_bev(wl, ar, br, det.clone(), prev.clone(), nf, npnt, out_stride=64, grad=True)
    O.affinity_loss(r1, r2, gt).backward()
    m = m.to(dev).train()
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    m1, m2 = training.affinity_train(m, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
    training.affinity_loss(m1, m2, gt.to(dev)).backward()
    named = dict(m.named_parameters())
    for k, v in wl.items():
        if v.grad is None or v.numel() == 0:  # max_obj < 5: the aug_dets hidden layer has width 0
            continue
        got, want = named[k].grad.double().cpu(), v.grad.double()
        # identical anchor yaws (hidden width 0: anchor = bias) make d sqrt(0) = NaN in torch autograd - and here, at the same places
        assert torch.equal(torch.isnan(got), torch.isnan(want)), k
        got, want = torch.nan_to_num(got), torch.nan_to_num(want)
        assert float((got - want).abs().max()) <= 3e-3 * max(float(want.abs().max()), 1e-7) + 1e-9, k
    for got, want in ((ad.grad, ar.grad), (bd.grad, br.grad)):
        assert float((got.double().cpu() - want.double()).abs().max()) <= 3e-3 * max(float(want.abs().max()), 1e-7) + 1e-9


def _vox_cases(n, seed):
    rnd = random.Random(seed)
    out = []
    for k in range(n):
        P = rnd.choice([1, 2, 63, 64, 65, 255, 1000, 4097, 20000, 65536])
        ndim = rnd.choice([3, 4, 5, 6])
        mp = rnd.choice([1, 2, 3, 10, 35])
        mv = rnd.choice([1, 7, 100, 5000, 70000])
        grid = rnd.choice(["nusc", "coarse", "tiny"])
        out.append((P, ndim, mp, mv, grid, 5000 + k))
    return out


@pytest.mark.parametrize("P,ndim,mp,mv,grid,seed", _vox_cases(16, 31))
def test_voxelizer_random_clouds_bit_exact_vs_c_oracle(P, ndim, mp, mv, grid, seed):
    """Random point counts, point widths, capacities and grids (points outside the range, duplicates, many points per cell):
    voxel order, contents, counts and coordinates bit-identical to the serial C restatement of the reference loop."""
    from oracle import voxelize_oracle as VO
    from shasta_amd.voxel_generator import points_to_voxel_device
    dev = torch.device("cuda:0")
    vs, rg = {"nusc": ([0.075, 0.075, 0.2], [-54, -54, -5, 54, 54, 3]), "coarse": ([0.5, 0.5, 8.0], [-20, -30, -5, 20, 30, 3]),
              "tiny": ([1.0, 2.0, 4.0], [0, 0, 0, 4, 6, 4])}[grid]
    vs, rg = np.array(vs, np.float32), np.array(rg, np.float32)
    rng = np.random.default_rng(seed)
    pts = rng.normal(0, 1, (P, ndim)).astype(np.float32)
    span = (rg[3:] - rg[:3])
    pts[:, :3] = (rg[:3] + span * rng.uniform(-0.1, 1.1, (P, 3))).astype(np.float32)  # ~1/4 of the points fall outside
    if P > 10:
        pts[: P // 7] = pts[P // 2: P // 2 + P // 7]  # exact duplicates
    v, c, n, mean = points_to_voxel_device(torch.from_numpy(pts).to(dev), vs, rg, mp, mv, with_mean=True)
    rv, rc, rn, rmean = VO.points_to_voxel(pts, vs, rg, mp, mv, with_mean=True)
    assert v.shape[0] == rv.shape[0]
    assert np.array_equal(c.cpu().numpy(), rc) and np.array_equal(n.cpu().numpy(), rn) and np.array_equal(v.cpu().numpy(), rv)
    np.testing.assert_allclose(mean.cpu().numpy(), rmean, rtol=1e-6, atol=1e-6)


def _vox_batch_cases(n, seed):
    rnd = random.Random(seed)
    out = []
    for k in range(n):
        nc = rnd.choice([1, 2, 3, 7, 16, 32])
        sizes = tuple(rnd.choice([0, 1, 255, 256, 257, 3000, 20000]) for _ in range(nc))
        out.append((sizes, rnd.choice([3, 4, 5, 6, 9]), rnd.choice([1, 2, 10, 35]), rnd.choice([1, 50, 4000]), rnd.choice(["nusc", "coarse", "tiny"]), 7000 + k))
    return out


@pytest.mark.parametrize("sizes,ndim,mp,mv,grid,seed", _vox_batch_cases(10, 77))
def test_voxelizer_random_batches_bit_exact_vs_c_oracle(sizes, ndim, mp, mv, grid, seed):
    """shasta_voxelize_mean_batch_f32 on random batches: 1 - 32 clouds of random sizes (empty ones, ones that end on a workgroup
    boundary), point widths 3 - 9, capacities and grids; every cloud bit for bit the serial C restatement, counts on the device."""
    from oracle import voxelize_oracle as VO
    from shasta_amd.voxel_generator import points_to_voxel_batch_device
    dev = torch.device("cuda:0")
    vs, rg = {"nusc": ([0.075, 0.075, 0.2], [-54, -54, -5, 54, 54, 3]), "coarse": ([0.5, 0.5, 8.0], [-20, -30, -5, 20, 30, 3]),
              "tiny": ([1.0, 2.0, 4.0], [0, 0, 0, 4, 6, 4])}[grid]
    vs, rg = np.array(vs, np.float32), np.array(rg, np.float32)
    rng = np.random.default_rng(seed)
    clouds = []
    for P in sizes:
        pts = rng.normal(0, 1, (P, ndim)).astype(np.float32)
        pts[:, :3] = (rg[:3] + (rg[3:] - rg[:3]) * rng.uniform(-0.1, 1.1, (P, 3))).astype(np.float32)
        if P > 10:
            pts[: P // 7] = pts[P // 2: P // 2 + P // 7]
        clouds.append(pts)
    if sum(sizes) == 0:
        clouds[0] = np.zeros((0, ndim), np.float32)
    v, c, n, mean, nv = points_to_voxel_batch_device([torch.from_numpy(x).to(dev) for x in clouds], vs, rg, mp, mv, with_mean=True)
    nvh = nv.cpu().numpy()
    for i, pts in enumerate(clouds):
        rv, rc, rn, rmean = VO.points_to_voxel(pts, vs, rg, mp, mv, with_mean=True)
        V = int(nvh[i])
        assert V == rv.shape[0], i
        assert np.array_equal(c[i, :V].cpu().numpy(), rc) and np.array_equal(n[i, :V].cpu().numpy(), rn), i
        assert np.array_equal(v[i, :V].cpu().numpy(), rv), i
        np.testing.assert_allclose(mean[i, :V].cpu().numpy(), rmean, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("npnt,seed", [(1, 1), (4, 2), (5, 3)])
def test_boxes_outside_the_map_follow_the_reference_clamping(npnt, seed):
    """bilinear_interpolate_torch clamps the corner INDICES to the map but takes the weights from the clamped indices
    (center_utils.py:92-121): boxes beyond the map edge (or far outside) must reproduce that, not be zeroed or rejected."""
    dev = torch.device("cuda:0")
    N, nf, B = 24, 7, 3
    m = _model(N, nf, npnt, 8, seed)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    bev = torch.relu(torch.randn(B, 180, 180, 64, generator=g))
    pbev = torch.relu(torch.randn(B, 180, 180, 64, generator=g))
    det, prev = O.synth_boxes(g, B, N, None), O.synth_boxes(g, B, N, None)
    for t in (det, prev):
        t[:, ::3, 0] = torch.empty(B, len(range(0, N, 3))).uniform_(-75, 75, generator=g)   # up to 21 m outside
        t[:, 1::4, 1] = torch.empty(B, len(range(1, N, 4))).uniform_(-56, -53.5, generator=g)  # straddling the edge
    r1, r2, im = O.forward_from_bev(w, bev, pbev, det.clone(), prev.clone(), nf, npnt, return_intermediates=True)
    m = m.to(dev)
    m.keep_intermediates = True
    with torch.no_grad():
        m1, m2 = m.affinity_from_bev(bev.to(dev), pbev.to(dev), det.to(dev), prev.to(dev))
    feat = m.last_intermediates["feature"][:, :N].cpu()
    scale = float(im["feature"].abs().max())
    assert float((feat - im["feature"]).abs().max()) <= 2e-4 * max(1.0, scale)
    np.testing.assert_allclose(m1.cpu().numpy(), r1.numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(m2.cpu().numpy(), r2.numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("N,B,alpha,seed", [(1, 3, 0.5, 1), (63, 5, 0.05, 2), (64, 4, 0.02, 3), (65, 3, 0.3, 4), (130, 6, 0.01, 5), (500, 2, 0.004, 6)])
def test_device_decode_decisions_random_sizes(N, B, alpha, seed):
    """The batched decision kernel against the host restatement of the eval loop at table sizes around the 64-lane tiles and at
    the headline size; peaked random rows / columns so that every branch (match, dead, FN, newborn, FP) occurs."""
    import copy
    from shasta_amd import decode as Dm
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    m1 = rng.dirichlet(np.full(N + 2, alpha), size=(B, N)).astype(np.float32)
    m2 = np.swapaxes(rng.dirichlet(np.full(N + 2, alpha), size=(B, N)).astype(np.float32), 1, 2).copy()
    n_prev, n_cur = rng.integers(0, N + 1, size=B), rng.integers(0, N + 1, size=B)
    n_prev[0], n_cur[0] = N, N
    pc, ps, df, ds = Dm.decode_flags_device(torch.from_numpy(m1).to(dev), torch.from_numpy(m2).to(dev), n_prev, n_cur)

    def boxes(n, tag):
        return [dict(sample_token=tag, translation=[float(i), float(-i), 0.5], velocity=[0.5 * i, -0.25 * i]) for i in range(n)]

    events = 0
    for b in range(B):
        c1, p1 = boxes(int(n_cur[b]), "c"), boxes(int(n_prev[b]), "p")
        c2, p2 = copy.deepcopy(c1), copy.deepcopy(p1)
        ref = Dm.decode_frame(m1[b], m2[b], c1, p1, "tok", 0.5)
        got = Dm.decode_frame_from_flags(pc[b], ps[b], df[b], ds[b], c2, p2, "tok", 0.5)
        assert got[1] == ref[1] and got[2] == ref[2] and got[0] == ref[0], b
        events += len(ref[1]) + sum(1 for a in ref[0] if a.get("FN") or a.get("newborn"))
    if alpha <= 0.05 and N >= 63:
        assert events > 0  # the peaked matrices do trigger the anchor branches


@pytest.mark.parametrize("N,nf,B,n_real", [(600, 7, 2, None), (1000, 3, 1, 900), (2046, 7, 1, 1500)])
def test_large_tables_up_to_the_documented_limit(N, nf, B, n_real):
    """max_obj up to 2046 (the documented limit; F = 64 so that the anchor weights stay small): every kernel's size-dependent
    choice (LDS footprints, column-softmax tiling, K splits) at its far end, against the oracle."""
    dev = torch.device("cuda:0")
    m = _model(N, nf, 1, 8, N)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(N)
    bev = torch.relu(torch.randn(B, 180, 180, 64, generator=g))
    pbev = torch.relu(torch.randn(B, 180, 180, 64, generator=g))
    det, prev = O.synth_boxes(g, B, N, n_real), O.synth_boxes(g, B, N, n_real)
    r1, r2 = O.forward_from_bev(w, bev, pbev, det.clone(), prev.clone(), nf, 1)
    m = m.to(dev)
    with torch.no_grad():
        m1, m2 = m.affinity_from_bev(bev.to(dev), pbev.to(dev), det.to(dev), prev.to(dev))
    np.testing.assert_allclose(m1.cpu().numpy(), r1.numpy(), rtol=0, atol=TOL)
    np.testing.assert_allclose(m2.cpu().numpy(), r2.numpy(), rtol=0, atol=TOL)
    assert torch.equal(m1.argmax(-1).cpu(), r1.argmax(-1)) or float((torch.sort(r1, -1).values[..., -1] - torch.sort(r1, -1).values[..., -2]).min()) < 1e-6
