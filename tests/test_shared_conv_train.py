"""K0 in train() mode on the GPU (shasta_amd/shared_conv_train.py, csrc/shared_conv_train.hip): conv + batch-statistics BatchNorm + ReLU ->
NHWC for both maps of a frame pair and the backward for shared_conv.0.{weight,bias} / shared_conv.1.{weight,bias}, against a float64
evaluation of det3d/models/tracker/shasta.py:42-47 as applied at :223-228 with torch autograd on the host (and against the module's own
nn.Sequential on the device).  The reference-gradient goldens (tests/test_training_golden.py, incl. the 512 x 180 x 180 car case) pin the
same path end to end on the reference's own autograd."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import build_model

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch.device("cuda", 0)


def _model(cin, seed=3):
    return build_model(dict(max_obj=4, np=1, nf=3, seed=seed, cin=cin, stride=8))


def _float64(model, x, xp, g, gp):
    """out, out_prev (NHWC), the four parameter gradients of sum(out g) + sum(out_prev gp) and the running statistics after the two
    BatchNorm calls, all in float64 on the host."""
    conv, bn = model.shared_conv[0], model.shared_conv[1]
    W, b = conv.weight.detach().double().cpu().requires_grad_(True), conv.bias.detach().double().cpu().requires_grad_(True)
    ga, be = bn.weight.detach().double().cpu().requires_grad_(True), bn.bias.detach().double().cpu().requires_grad_(True)
    rm, rv = bn.running_mean.detach().double().cpu().clone(), bn.running_var.detach().double().cpu().clone()
    outs = []
    for t in (x, xp):
        y = F.conv2d(t.double().cpu(), W, b, padding=1)
        o = torch.relu(F.batch_norm(y, rm, rv, ga, be, True, bn.momentum, bn.eps))
        outs.append(o.permute(0, 2, 3, 1).contiguous())
    loss = (outs[0] * g.double().cpu()).sum() + (outs[1] * gp.double().cpu()).sum()
    loss.backward()
    return outs[0].detach(), outs[1].detach(), dict(weight=W.grad, bias=b.grad, gamma=ga.grad, beta=be.grad), rm, rv


def _run(model, x, xp, g, gp):
    model.zero_grad(set_to_none=True)
    out, outp = model.shared_conv_nhwc(x, xp)
    ((out * g).sum() + (outp * gp).sum()).backward()
    conv, bn = model.shared_conv[0], model.shared_conv[1]
    return out.detach(), outp.detach(), dict(weight=conv.weight.grad.clone(), bias=conv.bias.grad.clone(), gamma=bn.weight.grad.clone(),
                                             beta=bn.bias.grad.clone())


def _case(B, cin, H, W, seed, sparse=False, offset=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(B, cin, H, W, generator=g)) + offset
    xp = torch.relu(torch.randn(B, cin, H, W, generator=g)) * 1.7
    go, gpo = torch.randn(B, H, W, 64, generator=g), torch.randn(B, H, W, 64, generator=g)
    if sparse:  # the gradient the BEV gather sends back: a few hundred pixels per map
        keep = torch.rand(B, H, W, 1, generator=g) < 0.05
        go, gpo = go * keep, gpo * keep
    return x, xp, go, gpo


@pytest.mark.parametrize("B,cin,H,W,arith", [(2, 512, 180, 180, "f16x2"), (1, 64, 180, 180, "f16x2"), (3, 40, 33, 47, "f16x2"), (2, 8, 24, 24, "f16x2"),
                                            (1, 16, 12, 200, "f16x2"), (2, 32, 50, 187, "f32"), (1, 48, 7, 5, "f16x2")])
def test_train_mode_shared_conv_against_float64(B, cin, H, W, arith):
    dev = _dev()
    model = _model(cin)
    with torch.no_grad():  # a BatchNorm that does something
        model.shared_conv[1].weight.uniform_(0.5, 1.5)
        model.shared_conv[1].bias.uniform_(-0.3, 0.3)
    x, xp, g, gp = _case(B, cin, H, W, seed=B * 100 + cin, sparse=(H == 180))
    want0, want1, wgrads, rm, rv = _float64(model, x, xp, g, gp)
    model = model.to(dev).train()
    model.arithmetic = arith
    assert model.hand_written_train_conv
    out, outp, grads = _run(model, x.to(dev), xp.to(dev), g.to(dev), gp.to(dev))
    assert getattr(model, "_conv_raw", None) is not None, "the hand-written path did not run"
    for got, want in ((out, want0), (outp, want1)):
        assert got.shape == want.shape
        assert float((got.double().cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    for k, want in wgrads.items():
        got = grads[k].double().cpu()
        scale = max(float(want.abs().max()), 1e-12)
        # (the conv bias in front of a train-mode BatchNorm: zero gradient up to rounding - absolute slack of the sums' own rounding)
        slack = 1e-6 * float(g.abs().sum() + gp.abs().sum()) ** 0.5 if k == "bias" else 0.0
        assert float((got - want).abs().max()) <= 2e-5 * scale + slack, (k, float((got - want).abs().max()), scale)
    bn = model.shared_conv[1]
    assert torch.allclose(bn.running_mean.double().cpu(), rm, rtol=1e-5, atol=1e-7)
    assert torch.allclose(bn.running_var.double().cpu(), rv, rtol=1e-5, atol=1e-7)
    assert int(bn.num_batches_tracked) == 2


def test_train_mode_shared_conv_matches_the_modules_own_sequential_and_is_deterministic():
    """Same step through nn.Sequential (MIOpen conv, ATen BatchNorm, autograd) on the device: outputs, gradients, running statistics
    agree; two runs of the hand-written path give the same bits (fixed summation order in every reduction)."""
    dev = _dev()
    x, xp, g, gp = (t.to(dev) for t in _case(2, 64, 90, 180, seed=5))
    a = _model(64).to(dev).train()
    b = copy.deepcopy(a)
    b.hand_written_train_conv = False
    c = copy.deepcopy(a)
    oa, opa, ga = _run(a, x, xp, g, gp)
    ob, opb, gb = _run(b, x, xp, g, gp)
    oc, opc, gc = _run(c, x, xp, g, gp)
    assert getattr(b, "_conv_raw", None) is None and getattr(a, "_conv_raw", None) is not None
    assert torch.equal(oa, oc) and torch.equal(opa, opc) and all(torch.equal(ga[k], gc[k]) for k in ga)
    assert float((oa - ob).abs().max()) <= 1e-4 and float((opa - opb).abs().max()) <= 1e-4
    for k in ga:
        scale = max(float(gb[k].abs().max()), 1e-12)
        # (bias: zero up to rounding in both; the same absolute slack as against float64)
        slack = 1e-6 * float(g.abs().sum() + gp.abs().sum()) ** 0.5 if k == "bias" else 0.0
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-4 * scale + slack, k
    for k in ("running_mean", "running_var"):
        assert torch.allclose(getattr(a.shared_conv[1], k), getattr(b.shared_conv[1], k), rtol=1e-4, atol=1e-6)


def test_train_mode_conv_with_mean_far_from_zero():
    """Activations with |mean| >> std (the case sync_bn.py merges partial variances for): the float64 accumulation of the statistics
    pass keeps the variance."""
    dev = _dev()
    model = _model(16)
    with torch.no_grad():
        model.shared_conv[0].bias.fill_(300.0)
    x, xp, g, gp = _case(2, 16, 20, 30, seed=9)
    want0, want1, wgrads, rm, rv = _float64(model, x * 1e-3, xp * 1e-3, g, gp)
    model = model.to(dev).train()
    out, outp, grads = _run(model, (x * 1e-3).to(dev), (xp * 1e-3).to(dev), g.to(dev), gp.to(dev))
    # y = 300 +- 1e-3: xhat carries the fp32 rounding of y itself (2^-15 of a unit) - compare with that in mind
    assert float((out.double().cpu() - want0).abs().max()) <= 5e-2 and float((outp.double().cpu() - want1).abs().max()) <= 5e-2
    assert torch.allclose(model.shared_conv[1].running_var.double().cpu(), rv, rtol=2e-2, atol=1e-9)


def test_no_grad_train_mode_forward_and_fallbacks():
    """train() mode under no_grad still normalises with batch statistics and moves the running ones; a map that requires grad and an
    eval-mode BatchNorm under autograd stay on nn.Sequential."""
    dev = _dev()
    x, xp, g, gp = (t.to(dev) for t in _case(1, 16, 16, 20, seed=2))
    m = _model(16).to(dev).train()
    with torch.no_grad():
        out, outp = m.shared_conv_nhwc(x, xp)
    ref = copy.deepcopy(m)
    ref.shared_conv[1].reset_running_stats()
    m2 = _model(16).to(dev).train()
    m2.hand_written_train_conv = False
    with torch.no_grad():
        o2, op2 = m2.shared_conv_nhwc(x, xp)
    assert float((out - o2).abs().max()) <= 1e-4 and int(m.shared_conv[1].num_batches_tracked) == 2
    assert torch.allclose(m.shared_conv[1].running_var, m2.shared_conv[1].running_var, rtol=1e-4)
    m3 = _model(16).to(dev).train()
    xg = x.clone().requires_grad_(True)
    o3, _ = m3.shared_conv_nhwc(xg, xp)
    o3.sum().backward()
    assert xg.grad is not None and getattr(m3, "_conv_raw", None) is None


def test_training_from_the_neck_follows_the_sequential_path_step_by_step():
    """The reference's training step from the neck outputs (Shasta.forward in train mode -> loss -> backward -> Adam on every trainable
    tensor incl. shared_conv.0 / .1) with K0 hand-written and, beside it, through nn.Sequential: the same losses and the same weights
    after eight steps (the two paths differ by summation order only, Adam amplifies nothing at this step size)."""
    import shasta_amd
    from shasta_amd import training
    dev = _dev()
    torch.manual_seed(5)
    cfg = dict(type="Shasta", reader=None, backbone=None, neck=None,
               bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
               max_obj=12, num_feats=3, num_point=5, in_channels=32)
    a = shasta_amd.build_simp_track(cfg).to(dev).train()
    b = copy.deepcopy(a)
    b.hand_written_train_conv = False
    g = torch.Generator().manual_seed(6)
    B, N, HW = 3, 12, 180
    x = torch.relu(torch.randn(B, 32, HW, HW, generator=g)).to(dev)
    xp = torch.relu(torch.randn(B, 32, HW, HW, generator=g)).to(dev)

    def boxes():
        t = torch.zeros(B, N, 11)
        t[:, :, :2] = (torch.rand(B, N, 2, generator=g) - 0.5) * 100
        t[:, :, 2] = torch.randn(B, N, generator=g)
        t[:, :, 3:6] = torch.rand(B, N, 3, generator=g) * 3 + 0.5
        t[:, :, 6] = (torch.rand(B, N, generator=g) - 0.5) * 6.28
        t[:, :, 7:9] = torch.randn(B, N, 2, generator=g)
        t[:, :, 9] = 0.5
        return t.to(dev)
    det, prev = boxes(), boxes()
    gt = torch.zeros(B, N + 2, N + 2)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    gt[torch.arange(B)[:, None], torch.arange(N)[None, :], perm] = 1.0
    gt = gt.to(dev)
    losses = []
    for m in (a, b):
        opt = training.FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        ls = []
        for _ in range(8):
            opt.zero_grad(set_to_none=True)
            m1, m2, _ = m(dict(det_boxes=det.clone(), prev_det_boxes=prev.clone(), bev_map=x, prev_bev_map=xp), train_mode=True)
            loss = training.affinity_loss(m1, m2, gt)
            loss.backward()
            opt.step()
            ls.append(float(loss.detach()))
        losses.append(ls)
    assert getattr(a, "_conv_raw", None) is not None and getattr(b, "_conv_raw", None) is None
    assert losses[0][-1] < losses[0][0]  # it learns
    for la, lb in zip(*losses):
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(lb)), losses
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert float((p - q).abs().max()) <= 2e-4 * max(1.0, float(q.abs().max())), k
    for k in ("running_mean", "running_var"):
        assert torch.allclose(getattr(a.shared_conv[1], k), getattr(b.shared_conv[1], k), rtol=1e-3, atol=1e-5)


def test_training_from_the_neck_is_bit_reproducible():
    """Six training steps from the neck outputs (K0 hand-written, boxes crowded so that the gather's backward adds several terms into
    the same pixels), twice from the same start: the same losses and the same weights bit for bit - every sum of the step, the
    scatter-add into the BEV maps' gradient included, has a fixed order."""
    import shasta_amd
    from shasta_amd import training
    dev = _dev()
    torch.manual_seed(9)
    cfg = dict(type="Shasta", reader=None, backbone=None, neck=None,
               bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
               max_obj=40, num_feats=3, num_point=5, in_channels=32)
    base = shasta_amd.build_simp_track(cfg).to(dev).train()
    g = torch.Generator().manual_seed(10)
    B, N, HW = 2, 40, 180
    x = torch.relu(torch.randn(B, 32, HW, HW, generator=g)).to(dev)
    xp = torch.relu(torch.randn(B, 32, HW, HW, generator=g)).to(dev)

    def boxes():
        t = torch.zeros(B, N, 11)
        t[:, :, :2] = (torch.rand(B, N, 2, generator=g) - 0.5) * 6 + 10  # 40 boxes within 6 m: shared pixels
        t[:, :, 2] = torch.randn(B, N, generator=g)
        t[:, :, 3:6] = torch.rand(B, N, 3, generator=g) * 3 + 0.5
        t[:, :, 6] = (torch.rand(B, N, generator=g) - 0.5) * 6.28
        t[:, :, 7:9] = torch.randn(B, N, 2, generator=g)
        t[:, :, 9] = 0.5
        return t.to(dev)
    det, prev = boxes(), boxes()
    gt = torch.zeros(B, N + 2, N + 2)
    perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
    gt[torch.arange(B)[:, None], torch.arange(N)[None, :], perm] = 1.0
    gt = gt.to(dev)
    runs = []
    for _ in range(2):
        m = copy.deepcopy(base)
        opt = training.FusedAdam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
        ls = []
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            m1, m2, _ = m(dict(det_boxes=det.clone(), prev_det_boxes=prev.clone(), bev_map=x, prev_bev_map=xp), train_mode=True)
            loss = training.affinity_loss(m1, m2, gt)
            loss.backward()
            opt.step()
            ls.append(loss.detach().clone())
        assert getattr(m, "_conv_raw", None) is not None
        runs.append((torch.stack(ls), {k: p.detach().clone() for k, p in m.named_parameters()}))
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert float(runs[0][0][-1]) < float(runs[0][0][0])
    for k, p in runs[0][1].items():
        assert torch.equal(p, runs[1][1][k]), k
    assert not torch.equal(runs[0][1]["shared_conv.0.weight"], dict(base.named_parameters())["shared_conv.0.weight"])  # it moved
