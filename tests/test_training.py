"""Training path: the hand-written backward against torch autograd of the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import shasta_oracle as O
from tests.helpers import build_model


def _case(N, nf, npnt, B, seed, n_real=None, H=24, W=24):
    c = dict(max_obj=N, nf=nf, np=npnt, B=B, n_real=n_real, cin=16, hw=H, stride=8, seed=seed)
    model = build_model(c)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bev, pbev, det, prev = O.synth_case(B, N, n_real, c["cin"], H, W, seed)
    # boxes inside the (small) map so that the bilinear weights are not all clamped
    span = H * 8 * 0.075
    for t in (det, prev):
        t[:, :, 0] = (t[:, :, 0] % (span * 0.8)) - 54 + 0.1 * span
        t[:, :, 1] = (t[:, :, 1] % (span * 0.8)) - 54 + 0.1 * span
    a = O.shared_conv_nhwc(w, bev)
    b = O.shared_conv_nhwc(w, pbev)
    gen = torch.Generator().manual_seed(seed)
    gt = (torch.rand(B, N + 2, N + 2, generator=gen) < 0.15).float()
    gt[:, 0, 0] = 1.0
    return c, model, w, a, b, det, prev, gt


def _oracle_grads(c, w, a, b, det, prev, gt):
    wl = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.startswith("shared_conv")) for k, v in w.items()}
    a = a.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    m1, m2 = O.forward_from_bev(wl, a, b, det.clone(), prev.clone(), c["nf"], c["np"], out_stride=c["stride"], grad=True)
    loss = O.affinity_loss(m1, m2, gt)
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in wl.items() if v.grad is not None}, a.grad, b.grad, m1.detach(), m2.detach()


def _close(name, got, want, rtol=2e-3):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= rtol * max(scale, 1e-7), "%s: max |diff| %.3e vs scale %.3e" % (name, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("N,nf,npnt,B,n_real", [(6, 7, 1, 2, None), (12, 3, 4, 3, 9), (20, 7, 5, 2, None)])
def test_backward_matches_autograd_of_oracle(N, nf, npnt, B, n_real):
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(N, nf, npnt, B, seed=5, n_real=n_real)
    loss_ref, gref, ga_ref, gb_ref, m1_ref, m2_ref = _oracle_grads(c, w, a, b, det, prev, gt)

    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    ad = a.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    detd, prevd = det.to(dev).contiguous(), prev.to(dev).contiguous()
    m1, m2 = training.affinity_train(model, ad, bd, detd, prevd)
    np.testing.assert_allclose(m1.detach().cpu().numpy(), m1_ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(m2.detach().cpu().numpy(), m2_ref.numpy(), atol=2e-5)
    loss = training.affinity_loss(m1, m2, gt.to(dev))
    assert abs(float(loss.detach()) - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref))
    loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    checked = 0
    for k, g in gref.items():
        assert named[k].grad is not None, "no gradient for " + k
        _close(k, named[k].grad, g)
        checked += 1
    assert checked == 2 * (8 + 4 + 8 + 3 + 3 + 6)
    _close("d bev", ad.grad, ga_ref)
    _close("d prev_bev", bd.grad, gb_ref)


@pytest.mark.gpu
def test_training_step_reduces_loss_and_inference_sees_new_weights():
    """A few Adam steps (tools/nusc_shasta/train.py:213-218) on one batch: the loss goes down, and the inference path picks
    up the updated parameters (packed weights are re-packed when a parameter version changes)."""
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 2, seed=9)
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    params = training.affinity_params(model)
    opt = torch.optim.Adam(params, lr=1e-3)
    ad, bd, gtd = a.to(dev), b.to(dev), gt.to(dev)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        m1, m2 = training.affinity_train(model, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
        loss = training.affinity_loss(m1, m2, gtd)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    model.eval()
    with torch.no_grad():
        e1, _ = model.affinity_from_bev(ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
    w2 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    r1, _ = O.forward_from_bev(w2, a, b, det.clone(), prev.clone(), c["nf"], c["np"], out_stride=c["stride"])
    np.testing.assert_allclose(e1.cpu().numpy(), r1.numpy(), atol=2e-5)


@pytest.mark.gpu
def test_model_forward_is_differentiable_like_the_reference_module():
    """model(example, train_mode=True) under autograd (det3d/torchie/apis/train_track.py:109-130 -> train.py:198-213):
    gradients reach the affinity parameters AND, through the NHWC maps, the torch shared_conv block."""
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 2, seed=3)
    bev, pbev, _, _ = O.synth_case(2, 12, None, c["cin"], 24, 24, 3)
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    example = dict(bev_map=bev.to(dev), prev_bev_map=pbev.to(dev), det_boxes=det.to(dev).contiguous(), prev_det_boxes=prev.to(dev).contiguous())
    model.extract_feat = lambda ex: (ex["bev_map"], None, ex["prev_bev_map"], None)  # neck outputs are given
    m1, m2, _ = model(example, train_mode=True)
    assert m1.requires_grad and m2.requires_grad
    training.affinity_loss(m1, m2, gt.to(dev)).backward()
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    assert float(model.shared_conv[0].weight.grad.abs().sum()) > 0
