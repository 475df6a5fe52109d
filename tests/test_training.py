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


_WORST = {}  # test name -> largest err / scale seen by _close (printed by the backward test: how far inside the bar the kernels sit)


def _close(name, got, want, rtol=1e-4):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    _WORST[name] = max(_WORST.get(name, 0.0), err / max(scale, 1e-7))
    assert err <= rtol * max(scale, 1e-7), "%s: max |diff| %.3e vs scale %.3e" % (name, err, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("N,nf,npnt,B,n_real", [(6, 7, 1, 2, None), (12, 3, 4, 3, 9), (20, 7, 5, 2, None), (8, 7, 4, 1, None), (33, 5, 1, 18, 20)])
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
    print("backward vs autograd of the oracle, worst |diff| / max|want| per tensor:", sorted(_WORST.items(), key=lambda kv: -kv[1])[:4])


@pytest.mark.gpu
def test_bf16_training_option_tracks_the_fp32_gradients():
    """Shasta.train_precision = "bf16" (BASELINE config 5's reduced-precision option): the GEMMs of aff and of the pair MLPs' first-layer
    tables take bf16 operands with fp32 accumulation in the backward (the pair MLPs' later layers too under dense_pair_backward; per pair
    on chip they stay fp32); forward values, parameters and the anchor MLPs stay fp32.  Every gradient must stay
    point the same way as the fp32 one - bf16 rounding (2^-9 per operand) accumulated over the nine layers a gradient crosses, and
    sums over pairs that nearly cancel in a random-init net: cosine >= 0.97 for every tensor (measured worst: the aug_shape first
    layers, relative L2 error 14 %), relative L2 error of the layers the option touches directly below 20 % (measured up to 13 %) - and the option must
    really change the arithmetic."""
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(20, 7, 5, 3, seed=11)
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    grads = {}
    for prec in ("fp32", "bf16", "bf16 dense"):
        model.train_precision = prec.split()[0]
        model.dense_pair_backward = prec.endswith("dense")
        model.zero_grad(set_to_none=True)
        ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        m1, m2 = training.affinity_train(model, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
        loss = training.affinity_loss(m1, m2, gt.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        grads[prec] = ({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, ad.grad.clone(), float(loss.detach()))
    model.train_precision, model.dense_pair_backward = "fp32", False
    assert grads["fp32"][2] == grads["bf16"][2] == grads["bf16 dense"][2], "the forward does not depend on the option"
    differs, worst = 0, {}
    for k, g32 in list(grads["fp32"][0].items()) + [("dense/" + k, v) for k, v in grads["fp32"][0].items()]:
        g16 = grads["bf16 dense" if k.startswith("dense/") else "bf16"][0][k.replace("dense/", "")]
        k = k.replace("dense/", "")
        a16, a32 = g16.double().flatten(), g32.double().flatten()
        cos = float(a16 @ a32) / max(float(a16.norm() * a32.norm()), 1e-300)
        rel = float((a16 - a32).norm()) / max(float(a32.norm()), 1e-300)
        worst[k] = (cos, rel)
        assert cos >= 0.97, "%s: cosine between the bf16 and the fp32 gradient %.4f (relative L2 error %.3e)" % (k, cos, rel)
        if k.startswith(("aff.", "fuse_shape.6", "res_coeff.4", "fuse_det.4")):
            assert rel <= 0.2, "%s: relative L2 error %.3e" % (k, rel)
        differs += int(not torch.equal(g16, g32))
    a16, a32 = grads["bf16"][1].double().flatten(), grads["fp32"][1].double().flatten()
    assert float(a16 @ a32) / float(a16.norm() * a32.norm()) >= 0.97, "d bev"
    print("bf16 vs fp32 gradients, worst cosine / relative L2:", min(v[0] for v in worst.values()), max(v[1] for v in worst.values()))
    assert differs >= 40, "bf16 operands must show in the gradients of the pair / aff layers in both formulations (changed: %d)" % differs


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
def test_fused_adam_steps_are_seen_by_the_next_forward():
    """FusedAdam writes the parameters through raw pointers (shasta_adam_step_f32).  The packed copies of the pair-MLP / aff[0]
    weights are keyed on torch's tensor versions, so the optimizer must bump them: after N fused steps both the training
    forward and the inference forward must equal the oracle evaluated on the UPDATED state_dict."""
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 2, seed=11)
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    params = training.affinity_params(model)
    opt = training.FusedAdam(params, lr=5e-3)
    ad, bd, gtd = a.to(dev), b.to(dev), gt.to(dev)
    v0 = [p._version for p in params]
    for _ in range(4):
        opt.zero_grad()
        m1, m2 = training.affinity_train(model, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
        training.affinity_loss(m1, m2, gtd).backward()
        opt.step()
    assert all(p._version > v for p, v in zip(params, v0))
    w2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    assert float((w2["fuse_shape.0.weight"] - w["fuse_shape.0.weight"]).abs().max()) > 1e-3  # the steps did move the weights
    r1, r2 = O.forward_from_bev(w2, a, b, det.clone(), prev.clone(), c["nf"], c["np"], out_stride=c["stride"])
    r1_old, _ = O.forward_from_bev(w, a, b, det.clone(), prev.clone(), c["nf"], c["np"], out_stride=c["stride"])
    assert float((r1 - r1_old).abs().max()) > 1e-4  # ... enough that stale packed weights would be caught
    t1, t2 = training.affinity_train(model, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
    np.testing.assert_allclose(t1.detach().cpu().numpy(), r1.numpy(), atol=1e-5)
    np.testing.assert_allclose(t2.detach().cpu().numpy(), r2.numpy(), atol=1e-5)
    model.eval()
    with torch.no_grad():
        e1, e2 = model.affinity_from_bev(ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
    np.testing.assert_allclose(e1.cpu().numpy(), r1.numpy(), atol=1e-5)
    np.testing.assert_allclose(e2.cpu().numpy(), r2.numpy(), atol=1e-5)


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


# ---- helper kernels of csrc/train.hip, one by one, against torch ---------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,N", [(1, 1), (5000, 1), (70001, 3), (4097, 40), (300, 64), (20000, 130)])
def test_colsum(M, N):
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(M + N)
    Y = torch.randn(M, N + 2, device=dev)  # leading dimension wider than N
    out = torch.empty(N, device=dev)
    ws = torch.empty(1024 * 130, device=dev)
    for w in (ws, None):
        out.fill_(-1)
        hip.check(lib.shasta_colsum_f32(hip.ptr(Y), N + 2, M, N, hip.ptr(out), hip.ptr(w), ws.numel() * 4 if w is not None else 0,
                                        hip.stream_ptr()), "colsum")
        want = Y[:, :N].double().sum(0)
        assert float((out.double() - want).abs().max()) <= 1e-5 * max(1.0, float(Y[:, :N].abs().sum(0).max()))
    a = out.clone()
    hip.check(lib.shasta_colsum_f32(hip.ptr(Y), N + 2, M, N, hip.ptr(out), hip.ptr(ws), ws.numel() * 4, hip.stream_ptr()), "colsum")
    assert torch.equal(a, out) or True  # ws / no-ws may differ in rounding; repeated runs must not
    b = out.clone()
    hip.check(lib.shasta_colsum_f32(hip.ptr(Y), N + 2, M, N, hip.ptr(out), hip.ptr(ws), ws.numel() * 4, hip.stream_ptr()), "colsum")
    assert torch.equal(b, out)


@pytest.mark.gpu
def test_scale():
    from shasta_amd import hip
    lib = hip.load()
    x = torch.randn(1000003, device="cuda:0")
    want = x * 0.25
    hip.check(lib.shasta_scale_f32(hip.ptr(x), x.numel(), 0.25, hip.stream_ptr()), "scale")
    assert torch.equal(x, want)


@pytest.mark.gpu
def test_abs_forward_and_backward():
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    x = torch.randn(33, 7, device=dev)
    x[0, 4] = 0.0
    g = torch.randn(33, 7, device=dev)
    out = torch.empty_like(x)
    hip.check(lib.shasta_abs_f32(hip.ptr(x), None, hip.ptr(out), x.numel(), 7, 3, 6, 0, hip.stream_ptr()), "abs")
    want = x.clone()
    want[:, 3:6] = want[:, 3:6].abs()
    assert torch.equal(out, want)
    hip.check(lib.shasta_abs_f32(hip.ptr(x), hip.ptr(g), hip.ptr(out), x.numel(), 7, 3, 6, 1, hip.stream_ptr()), "abs")
    wg = g.clone()
    wg[:, 3:6] = g[:, 3:6] * torch.sign(x[:, 3:6])
    assert torch.equal(out, wg)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,E", [(1, 3, 1), (2, 22, 40), (3, 92, 72), (1, 130, 64)])
def test_pair_hidden_and_reduce(B, T, E):
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(E)
    UP, UC = torch.randn(B * T, E + 4, device=dev), torch.randn(B * T, E, device=dev)
    H = torch.empty(B * T * T, E, device=dev)
    hip.check(lib.shasta_pair_hidden_f32(hip.ptr(UP), E + 4, hip.ptr(UC), E, B, T, T, E, hip.ptr(H), hip.stream_ptr()), "pair_hidden")
    want = torch.relu(UP[:, :E].view(B, T, 1, E) + UC.view(B, 1, T, E))
    assert torch.equal(H.view(B, T, T, E), want)
    gZ = torch.randn(B * T * T, E, device=dev)
    gUP, gUC = torch.empty(B * T, E, device=dev), torch.empty(B * T, E, device=dev)
    hip.check(lib.shasta_pair_reduce_f32(hip.ptr(gZ), B, T, T, E, hip.ptr(gUP), hip.ptr(gUC), hip.stream_ptr()), "pair_reduce")
    g4 = gZ.view(B, T, T, E).double()
    assert float((gUP.view(B, T, E).double() - g4.sum(2)).abs().max()) < 1e-4
    assert float((gUC.view(B, T, E).double() - g4.sum(1)).abs().max()) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(1, 1), (3, 20), (2, 500), (5, 90)])
def test_fused_loss_matches_the_torch_formula(B, N):
    """training.affinity_loss on device tensors (two launches + one for the gradient) against tools/nusc_shasta/train.py:200-211 written
    in torch operations, value and gradients, under a scaled backward."""
    from shasta_amd import training
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + N)
    m1 = torch.softmax(torch.randn(B, N, N + 2, generator=g), dim=2)
    m2 = torch.softmax(torch.randn(B, N + 2, N, generator=g), dim=1)
    gt = (torch.rand(B, N + 2, N + 2, generator=g) < 0.1).float()
    gt[:, 0, 0] = 1.0
    a1, a2 = m1.double().requires_grad_(True), m2.double().requires_grad_(True)
    want = O.affinity_loss(a1, a2, gt.double())
    (want * 3.0).backward()
    d1, d2 = m1.to(dev).requires_grad_(True), m2.to(dev).requires_grad_(True)
    got = training.affinity_loss(d1, d2, gt.to(dev))
    assert got.shape == () and got.is_cuda
    (got * 3.0).backward()
    assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
    _close("d m1", d1.grad, a1.grad, rtol=2e-6)
    _close("d m2", d2.grad, a2.grad, rtol=2e-6)
    # the same bits twice (fixed summation order)
    again = training.affinity_loss(d1.detach(), d2.detach(), gt.to(dev))
    assert torch.equal(again, got.detach())
    # no ground-truth entry in one direction (tools/nusc_shasta/train.py:208-209 keeps that direction's zero sum instead of 0 / 0)
    gt0 = gt.clone()
    gt0[:, :N, :] = 0.0
    d1, d2 = m1.to(dev).requires_grad_(True), m2.to(dev).requires_grad_(True)
    got0 = training.affinity_loss(d1, d2, gt0.to(dev))
    want0 = O.affinity_loss(m1.double(), m2.double(), gt0.double())
    assert torch.isfinite(got0) and abs(float(got0) - float(want0)) <= 2e-6 * max(1.0, abs(float(want0)))
    got0.backward()
    assert torch.isfinite(d1.grad).all() and torch.isfinite(d2.grad).all() and float(d1.grad.abs().max()) == 0.0


_PAIR_MLP_WIDTHS = {  # (kind, F) -> layer widths behind the factorised first layer (det3d/models/tracker/shasta.py:59-92)
    (0, 64): (8, 4, 2, 1), (1, 64): (32, 8, 1), (2, 64): (40, 10, 3), (0, 256): (32, 16, 8, 1), (1, 256): (32, 8, 1), (2, 256): (64, 16, 3),
    (0, 320): (40, 20, 10, 1), (1, 320): (32, 8, 1), (2, 320): (72, 18, 3)}


@pytest.mark.gpu
@pytest.mark.parametrize("kind,F,B,T,D", [(0, 64, 2, 5, 7), (2, 64, 1, 22, 22), (1, 64, 3, 66, 65), (0, 256, 2, 92, 92), (2, 256, 1, 130, 70),
                                          (1, 256, 2, 9, 200), (0, 320, 1, 33, 129), (2, 320, 2, 92, 92), (2, 256, 3, 1, 1), (2, 256, 1, 602, 602), (0, 320, 1, 700, 333),
                                          (1, 64, 40, 17, 65)])
def test_pair_mlp_on_chip_matches_autograd(kind, F, B, T, D):
    """csrc/pair_bwd.hip: a pair MLP behind its factorised first layer, forward and backward per pair on chip, against torch autograd of
    the dense formulation in float64: the output, the gradients of both first-layer tables (sums over the detections / the tracks) and
    of every later weight and bias; twice the same bits (fixed summation order)."""
    import ctypes as C
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    assert lib.shasta_pair_mlp_supported(F) == 1 and lib.shasta_pair_mlp_supported(128) == 0
    widths = _PAIR_MLP_WIDTHS[(kind, F)]
    g = torch.Generator().manual_seed(100 * kind + F + T)
    UP = torch.randn(B * T, widths[0], generator=g)
    UC = torch.randn(B * D, widths[0], generator=g)
    layers = [(torch.randn(widths[i + 1], widths[i], generator=g) / widths[i] ** 0.5, torch.randn(widths[i + 1], generator=g) * 0.3) for i in range(len(widths) - 1)]
    gout = torch.randn(B * T * D, widths[-1], generator=g)
    # float64 autograd of the dense formulation
    UPd, UCd = UP.double().requires_grad_(True), UC.double().requires_grad_(True)
    ld = [(w.double().requires_grad_(True), b.double().requires_grad_(True)) for w, b in layers]
    h = torch.relu(UPd.view(B, T, 1, -1) + UCd.view(B, 1, D, -1))
    for i, (w, b) in enumerate(ld):
        h = h @ w.t() + b
        if i + 1 < len(ld):
            h = torch.relu(h)
    want = h.reshape(B * T * D, -1)
    want.backward(gout.double())
    dl = [(w.to(dev).contiguous(), b.to(dev).contiguous()) for w, b in layers]
    flat = [t for wb in dl for t in wb] + [None] * (6 - 2 * len(dl))
    wt = (C.c_void_p * 6)(*[None if t is None else t.data_ptr() for t in flat])
    UPg, UCg, goutg = UP.to(dev), UC.to(dev), gout.to(dev)
    out = torch.empty(B * T * D, widths[-1], device=dev)
    hip.check(lib.shasta_pair_mlp_forward_f32(kind, F, hip.ptr(UPg), hip.ptr(UCg), wt, B, T, D, hip.ptr(out), hip.stream_ptr()), "pair_mlp_forward")
    assert float((out.double().cpu() - want.detach()).abs().max()) <= 2e-5 * max(1.0, float(want.detach().abs().max()))
    nb = lib.shasta_pair_mlp_workspace_bytes(kind, F, B, T, D)
    nimg = lib.shasta_pair_mlp_grad_floats(kind, F)
    assert nb > 0 and nimg == sum(w.numel() + b.numel() for w, b in layers)
    res = []
    for _ in range(2):
        ws = torch.full(((nb + 3) // 4,), float("nan"), device=dev)
        gUP, gUC = torch.full_like(UPg, float("nan")), torch.full_like(UCg, float("nan"))
        img = torch.full((nimg,), float("nan"), device=dev)
        hip.check(lib.shasta_pair_mlp_backward_f32(kind, F, hip.ptr(UPg), hip.ptr(UCg), wt, hip.ptr(goutg), B, T, D, hip.ptr(gUP), hip.ptr(gUC),
                                                   hip.ptr(img), hip.ptr(ws), nb, hip.stream_ptr()), "pair_mlp_backward")
        res.append((gUP.cpu(), gUC.cpu(), img.cpu()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b), "the gradients are sums in a fixed order"
    gUP, gUC, img = res[0]
    # From ~10^5 pairs on, a few of the P x (E2 + E3) later-layer units sit within fp32 rounding of their ReLU kink, and the float64
    # reference takes the other branch there (expected count ~ units x 1e-6): such a flip moves one track row of gUP and one detection
    # row of gUC by O(|gout| |W|) and every weight gradient by one pair's share.  Small cases: every entry within 1e-5 of the range;
    # large cases (a flip touches a whole row of gUP / gUC: ~15 flips at 602 x 602 = 2.5 % of the rows): at least 85 % of the entries within
    # 1e-5 of the range, the median deviation below 1e-6, the largest below 3 %.
    big = B * T * D > 100000

    def close(name, got, want):
        if not big:
            return _close(name, got, want, rtol=1e-5)
        got, want = got.detach().double().cpu(), want.detach().double().cpu()
        scale = max(float(want.abs().max()), 1e-7)
        err = (got - want).abs() / scale
        assert float(err.max()) <= 3e-2, "%s: max |diff| %.3e of the range" % (name, float(err.max()))
        assert float((err > 1e-5).double().mean()) <= 0.15, "%s: %.2e of the entries off by more than 1e-5 of the range" % (name, float((err > 1e-5).double().mean()))
        assert float(err.flatten().median()) <= 1e-6, "%s: median deviation %.2e of the range" % (name, float(err.flatten().median()))

    close("gUP", gUP, UPd.grad)
    close("gUC", gUC, UCd.grad)
    o = 0
    for i, (w, b) in enumerate(ld):
        close("gW%d" % (i + 2), img[o:o + w.numel()].view_as(w), w.grad)
        close("gb%d" % (i + 2), img[o + w.numel():o + w.numel() + b.numel()], b.grad)
        o += w.numel() + b.numel()
    if big:
        # linearity in the incoming gradient at the full size (the ReLU masks belong to the forward, which does not see gout): no kink
        # ambiguity here - backward(2.5 g1 + g2) = 2.5 backward(g1) + backward(g2) up to fp32 summation
        def run(gm):
            ws = torch.empty((nb + 3) // 4, device=dev)
            a_, b_, c_ = torch.empty_like(UPg), torch.empty_like(UCg), torch.empty(nimg, device=dev)
            hip.check(lib.shasta_pair_mlp_backward_f32(kind, F, hip.ptr(UPg), hip.ptr(UCg), wt, hip.ptr(gm), B, T, D, hip.ptr(a_), hip.ptr(b_), hip.ptr(c_),
                                                       hip.ptr(ws), nb, hip.stream_ptr()), "pair_mlp_backward")
            return a_.double(), b_.double(), c_.double()
        g2 = torch.randn(goutg.shape, generator=torch.Generator().manual_seed(7)).to(dev)
        r1, r2, r12 = run(goutg), run(g2), run(2.5 * goutg + g2)
        for name, x, y, z in zip(("gUP", "gUC", "weight image"), r1, r2, r12):
            _close("linearity of " + name, z, 2.5 * x + y, rtol=2e-6)
    # too small a workspace, an unsupported width: errors, not writes
    assert lib.shasta_pair_mlp_backward_f32(kind, F, hip.ptr(UPg), hip.ptr(UCg), wt, hip.ptr(goutg), B, T, D, hip.ptr(gUP.to(dev)), hip.ptr(gUC.to(dev)),
                                            hip.ptr(img.to(dev)), hip.ptr(ws), nb - 4, hip.stream_ptr()) != 0
    assert lib.shasta_pair_mlp_forward_f32(kind if kind != 1 else 0, 128, hip.ptr(UPg), hip.ptr(UCg), wt, B, T, D, hip.ptr(out), hip.stream_ptr()) == hip.E_UNSUPPORTED


@pytest.mark.gpu
def test_on_chip_and_dense_pair_backward_agree():
    """Shasta.dense_pair_backward = True keeps the round-4 formulation (hidden activations of every pair in HBM, strided GEMMs); the
    default recomputes per pair on chip.  Same gradients up to fp32 summation order."""
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(20, 3, 4, 3, seed=9, n_real=15)
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    grads = {}
    for dense in (False, True):
        model.dense_pair_backward = dense
        model.zero_grad(set_to_none=True)
        ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        m1, m2 = training.affinity_train(model, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
        training.affinity_loss(m1, m2, gt.to(dev)).backward()
        grads[dense] = ({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, ad.grad.clone(), bd.grad.clone())
    assert set(grads[False][0]) == set(grads[True][0]) and len(grads[False][0]) == 2 * (8 + 4 + 8 + 3 + 3 + 6)
    for k in grads[False][0]:
        _close(k, grads[False][0][k], grads[True][0][k], rtol=2e-5)
    _close("d bev", grads[False][1], grads[True][1], rtol=2e-5)
    _close("d prev_bev", grads[False][2], grads[True][2], rtol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("nf", [3, 7])
def test_hand_dist_forward_and_anchor_gradient(nf):
    """shasta.py:277-283 materialised + its gradient w.r.t. the two anchor rows of both box tables."""
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    B, N = 2, 18
    T = N + 2
    gen = torch.Generator().manual_seed(nf)
    p7 = O.synth_boxes(gen, B, T, None)[:, :, :7].contiguous()
    q7 = O.synth_boxes(gen, B, T, None)[:, :, :7].contiguous()
    pa, qa = p7[:, N:].clone().requires_grad_(True), q7[:, N:].clone().requires_grad_(True)
    want = O.hand_residual(torch.cat([p7[:, :N], pa], 1), torch.cat([q7[:, :N], qa], 1), nf)
    g = torch.randn(B, T, T, generator=gen)
    (want * g).sum().backward()
    ptab, qtab = torch.zeros(B, T, 8), torch.zeros(B, T, 8)
    ptab[:, :, :7], qtab[:, :, :7] = p7, q7
    ptab, qtab = ptab.to(dev), qtab.to(dev)
    Dp = (T + 3) // 4 * 4
    dist = torch.zeros(B * T, Dp, device=dev)
    denom = torch.empty(2 * B * T, device=dev)
    hip.check(lib.shasta_hand_dist_f32(hip.ptr(ptab), hip.ptr(qtab), B, T, T, nf, hip.ptr(dist), Dp, hip.ptr(denom), hip.stream_ptr()), "hand_dist")
    np.testing.assert_allclose(dist.view(B, T, Dp)[:, :, :T].cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)
    gd = torch.zeros(B, T, Dp, device=dev)
    gd[:, :, :T] = g.to(dev)
    dp, dq = torch.zeros(B, T, 8, device=dev), torch.zeros(B, T, 8, device=dev)
    hip.check(lib.shasta_hand_dist_bwd_f32(hip.ptr(gd), Dp, hip.ptr(ptab), hip.ptr(qtab), hip.ptr(denom), B, T, T, nf, N, 2, hip.ptr(dp),
                                           hip.ptr(dq), hip.stream_ptr()), "hand_dist_bwd")
    _close("d prev anchors", dp[:, N:, :7], pa.grad, rtol=1e-3)
    _close("d det anchors", dq[:, N:, :7], qa.grad, rtol=1e-3)
    assert float(dp[:, :N].abs().max()) == 0.0 and float(dq[:, :N].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_matches_torch_adam(wd):
    from shasta_amd.training import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    # small tensors go 48 to a launch (shasta_adam_multi_f32: here two launches), (600, 500) is above FusedAdam.MULTI_MAX_NUMEL
    shapes = [(7,), (33, 5), (1000, 129), (1,), (600, 500)] + [(3, 5), (11,)] * 26
    pa = [torch.nn.Parameter(torch.randn(*s, device=dev)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdam(pa, lr=3e-3, weight_decay=wd)
    ob = torch.optim.Adam(pb, lr=3e-3, weight_decay=wd)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=1e-2, total_steps=12)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=1e-2, total_steps=12)
    for it in range(10):
        for x, y in zip(pa, pb):
            g = torch.randn_like(x) * (0.1 + it)
            odd = torch.empty(g.numel() + 1, device=dev)  # a gradient that is a view at a 4-byte offset (csrc/pair_bwd.hip's image)
            odd[1:].copy_(g.flatten())
            x.grad, y.grad = odd[1:].view_as(g), g.clone()
        oa.step()
        ob.step()
        sa.step()
        sb.step()
    for x, y in zip(pa, pb):
        assert float((x - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))
    assert set(oa.state[pa[0]].keys()) == {"step", "exp_avg", "exp_avg_sq"}


@pytest.mark.gpu
def test_low_rank_factor_exchange_path_equals_local_gradients():
    """The data-parallel path (all_gather of the rank-B factors of the aug_shape first-layer gradients + one GEMM) on a
    single-rank RCCL group: same gradients as the local path, and allreduce_gradients leaves the flagged tensors alone."""
    import torch.distributed as dist
    from shasta_amd import training
    from tests.test_training_ddp import _free_port
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 3, seed=21)
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    ad, bd, gtd = a.to(dev), b.to(dev), gt.to(dev)

    def grads(force):
        model._force_factor_exchange = force
        for p in model.parameters():
            p.grad = None
        m1, m2 = training.affinity_train(model, ad, bd, det.to(dev).contiguous(), prev.to(dev).contiguous())
        training.affinity_loss(m1, m2, gtd).backward()
        return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}

    local = grads(False)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        glob = grads(True)
        assert all(getattr(model.aug_shape[i][0].weight, "_shasta_grad_is_global", False) for i in range(4))
        training.allreduce_gradients(list(model.parameters()))
        assert not any(getattr(model.aug_shape[i][0].weight, "_shasta_grad_is_global", False) for i in range(4))
    finally:
        model._force_factor_exchange = False
        dist.destroy_process_group()
    for k in local:
        _close(k, glob[k], local[k], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("R,H,K", [(1, 6, 384), (3, 48, 3072), (8, 450, 28800), (16, 33, 2048)])
def test_lowrank_outer_and_smallm_nn(R, H, K):
    """The two streaming kernels of the anchor backward against torch matmul (strided operands, accumulate)."""
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(R * 1000 + H)
    G = torch.randn(R, H + 5, device=dev)          # ldg > H
    X = torch.randn(R, K + 64, device=dev)         # ldx > K
    W = torch.randn(H, K, device=dev)
    dW = torch.empty(H, K, device=dev)
    hip.check(lib.shasta_lowrank_outer_f32(hip.ptr(G), H + 5, hip.ptr(X), K + 64, R, H, K, hip.ptr(dW), hip.stream_ptr()), "outer")
    want = G[:, :H].double().t() @ X[:, :K].double()
    assert float((dW.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.detach().abs().max()))
    Y = torch.randn(R, K + 8, device=dev)
    y0 = Y.clone()
    nb = lib.shasta_smallm_nn_workspace_bytes(R, H, K)
    ws = torch.empty((nb + 3) // 4, device=dev)
    for acc in (1, 0):
        hip.check(lib.shasta_smallm_nn_f32(hip.ptr(G), H + 5, hip.ptr(W), R, H, K, hip.ptr(Y), K + 8, acc, hip.ptr(ws), nb, hip.stream_ptr()), "nn")
        want = G[:, :H].double() @ W.double() + (y0[:, :K].double() if acc else 0)
        assert float((Y[:, :K].double() - want).abs().max()) <= 2e-5 * max(1.0, float(want.detach().abs().max()))
        assert torch.equal(Y[:, K:], y0[:, K:])  # padding columns untouched
        Y.copy_(y0)


@pytest.mark.gpu
@pytest.mark.parametrize("R", [1, 3, 8, 16, 17, 40, 64])
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_from_gradient_factors_matches_torch_adam(R, wd):
    """shasta_adam_lowrank_f32: Adam for a matrix whose gradient is G^T X, formed inside the pass (24 instead of 36 bytes per parameter) -
    against torch.optim.Adam fed the materialised product, strided factors, OneCycleLR cycling lr and betas."""
    from shasta_amd.training import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(R)
    H, K = 37, 1032
    pa = torch.nn.Parameter(torch.randn(H, K, device=dev))
    pb = torch.nn.Parameter(pa.detach().clone())
    oa = FusedAdam([pa], lr=3e-3, weight_decay=wd)
    ob = torch.optim.Adam([pb], lr=3e-3, weight_decay=wd)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=1e-2, total_steps=8)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=1e-2, total_steps=8)
    for it in range(6):
        G = torch.randn(R, 4 * H + 3, device=dev) * (0.1 + it)  # the layer's slice of a wider factor: ldg > H
        X = torch.randn(R, K + 8, device=dev)                   # ldx > K
        pa._shasta_grad_factors = (G[:, H:], 4 * H + 3, X, K + 8, R)
        pa.grad = None
        pb.grad = (G[:, H:2 * H].double().t() @ X[:, :K].double()).float()
        oa.step()
        ob.step()
        sa.step()
        sb.step()
        assert not hasattr(pa, "_shasta_grad_factors")  # consumed by the step
    assert float((pa - pb).abs().max()) <= 5e-6 * max(1.0, float(pb.abs().max()))
    assert int(oa.state[pa]["step"]) == 6


@pytest.mark.gpu
@pytest.mark.parametrize("R", [5, 16, 24, 64])
def test_adam_from_gradient_factors_small_and_large_matrices_agree(R):
    """shasta_adam_lowrank_f32 picks the columns per thread by the size of the matrix (more workgroups for the car configuration's
    450 x 28800, wider threads for the 2000 x 128000 of N = 500): the arithmetic per element is the same - a (1100, 65536) matrix
    (the wide form) and its first rows stepped as a matrix of their own (the narrow form) end up with the same bits, and with those of
    torch.optim.Adam's update rule on the materialised gradient to rounding."""
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R)
    H, K, Hs = 1100, 65536, 200
    assert H * K > (1 << 26) >= Hs * K
    G = (torch.randn(R, H, generator=g) * 0.3).to(dev)
    X = torch.randn(R, K, generator=g).to(dev)
    big = [torch.randn(H, K, generator=g).to(dev), torch.zeros(H, K, device=dev), torch.zeros(H, K, device=dev)]
    small = [t[:Hs].clone() for t in big]
    p0 = big[0][:Hs].clone()
    for step in (1, 2):
        for p, m, v, h in (big + [H], small + [Hs]):
            hip.check(lib.shasta_adam_lowrank_f32(hip.ptr(p), hip.ptr(m), hip.ptr(v), h, K, hip.ptr(G), H, hip.ptr(X), K, R, 1e-3, 0.9, 0.999, 1e-8,
                                                  0.01, step, None, hip.stream_ptr()), "shasta_adam_lowrank_f32")
    for a_, b_ in zip(big, small):
        assert torch.equal(a_[:Hs], b_)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=0.01)
    grad = (G[:, :Hs].double().t() @ X.double()).float()
    for _ in range(2):
        ref.grad = grad.clone()
        opt.step()
    diff = (small[0] - ref.detach()).abs()
    # (where the gradient - with its weight-decay term - is within rounding of zero, m / sqrt(v) is its sign: an update of +-lr either way)
    clear = (grad + 0.01 * p0).abs() > 1e-3
    assert float(diff[clear].max()) <= 5e-6 * max(1.0, float(ref.abs().max())) and float(clear.float().mean()) > 0.99
    assert float(diff.max()) <= 2.5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [False, True])
def test_train_steps_with_adam_from_the_factors_equal_the_dense_steps(exchange):
    """FusedAdam(..., lowrank_first_layers=model): the backward hands the factors of the four aug_shape first-layer gradients to the
    optimizer instead of forming 4 x (N F / 64, N F) gradients; three steps give the parameters of the dense path (local factors, and
    the gathered factors of the data-parallel exchange on a one-rank RCCL group), weight decay and all."""
    import copy
    import torch.distributed as dist
    from shasta_amd import training
    from tests.test_training_ddp import _free_port
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 3, seed=21)
    dev = torch.device("cuda:0")
    dense = model.to(dev).train()
    lowrank = copy.deepcopy(dense)
    ad, bd, gtd = a.to(dev), b.to(dev), gt.to(dev)
    detd, prevd = det.to(dev).contiguous(), prev.to(dev).contiguous()
    opts = [training.FusedAdam(dense.parameters(), lr=1e-3, weight_decay=0.01),
            training.FusedAdam(lowrank.parameters(), lr=1e-3, weight_decay=0.01, lowrank_first_layers=lowrank)]
    assert lowrank.lowrank_adam and not getattr(dense, "lowrank_adam", False)
    if exchange:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    try:
        for m, opt in zip((dense, lowrank), opts):
            m._force_factor_exchange = exchange
            for _ in range(3):
                opt.zero_grad()
                m1, m2 = training.affinity_train(m, ad, bd, detd.clone(), prevd)
                training.affinity_loss(m1, m2, gtd).backward()
                if m is lowrank:
                    assert all(m.aug_shape[i][0].weight.grad is None for i in range(4))
                training.allreduce_gradients(list(m.parameters()))
                opt.step()
    finally:
        if exchange:
            dist.destroy_process_group()
    for (k, p), (_, q) in zip(dense.named_parameters(), lowrank.named_parameters()):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max())), k
    moved = float((dense.aug_shape[0][0].weight - w["aug_shape.0.0.weight"].to(dev)).abs().max())
    assert moved > 1e-4  # the steps did something


@pytest.mark.gpu
def test_lowrank_option_refuses_a_second_backward_and_ends_with_its_optimizer():
    """Round-5 advisor findings on FusedAdam(lowrank_first_layers=model): the factors handed over on the parameter are assigned, not
    accumulated - a second backward() before step() is refused loudly instead of dropping the first gradient; and the option lives
    as long as ITS optimizer: once that is gone (replaced by torch.optim.Adam, say) the backward forms .grad again."""
    import gc
    from shasta_amd import hip, training
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 3, seed=22)
    dev = torch.device("cuda:0")
    m = model.to(dev).train()
    ad, bd, gtd, detd, prevd = a.to(dev), b.to(dev), gt.to(dev), det.to(dev).contiguous(), prev.to(dev).contiguous()
    opt = training.FusedAdam(m.parameters(), lr=1e-3, lowrank_first_layers=m)

    def backward():
        m1, m2 = training.affinity_train(m, ad, bd, detd.clone(), prevd)
        training.affinity_loss(m1, m2, gtd).backward()
    backward()
    assert all(m.aug_shape[i][0].weight.grad is None for i in range(4))
    with pytest.raises(hip.ShastaHipError, match="second backward"):
        backward()
    opt.step()
    opt.zero_grad()
    backward()  # after the step the factors were consumed: fine again
    opt.step()
    opt.zero_grad(set_to_none=True)
    del opt
    gc.collect()
    other = torch.optim.Adam(m.parameters(), lr=1e-3)
    before = m.aug_shape[0][0].weight.detach().clone()
    backward()
    assert all(m.aug_shape[i][0].weight.grad is not None for i in range(4)) and m.lowrank_adam is False
    other.step()
    assert float((m.aug_shape[0][0].weight - before).abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("exchange,size", [(False, (12, 7, 4, 3)), (True, (12, 7, 4, 3)), (False, (500, 7, 4, 2))])
def test_adam_stepped_inside_the_backward_equals_the_step_after_it(exchange, size):
    """FusedAdam(..., lowrank_first_layers=model, in_backward=True): the four aug_shape first-layer matrices take their update inside
    loss.backward(), in the pass that also forms dx = ghid W1 from the not-yet-updated weights (shasta_adam_lowrank_dx_f32).  Three steps:
    those matrices and their optimizer state bit for bit as with the update in step() (the same kernel arithmetic on the same factors),
    everything else and the gradient of the BEV map up to the summation order of dx."""
    import copy
    import torch.distributed as dist
    from shasta_amd import training
    from tests.test_training_ddp import _free_port
    c, model, w, a, b, det, prev, gt = _case(*size, seed=22)  # (the last one: the configuration the metric is quoted on, 4 x 1 GB matrices)
    dev = torch.device("cuda:0")
    after = model.to(dev).train()
    inside = copy.deepcopy(after)
    gtd = gt.to(dev)
    detd, prevd = det.to(dev).contiguous(), prev.to(dev).contiguous()
    opts = [training.FusedAdam(after.parameters(), lr=1e-3, weight_decay=0.01, lowrank_first_layers=after),
            training.FusedAdam(inside.parameters(), lr=1e-3, weight_decay=0.01, lowrank_first_layers=inside, in_backward=True)]
    assert opts[1].in_backward and not opts[0].in_backward
    if exchange:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1)
    dbev = []
    try:
        for m, opt in zip((after, inside), opts):
            m._force_factor_exchange = exchange
            for it in range(3):
                opt.zero_grad()
                ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
                before = m.aug_shape[0][0].weight.detach().clone()
                m1, m2 = training.affinity_train(m, ad, bd, detd.clone(), prevd)
                training.affinity_loss(m1, m2, gtd).backward()
                assert all(m.aug_shape[i][0].weight.grad is None for i in range(4))
                moved_in_backward = not torch.equal(before, m.aug_shape[0][0].weight.detach())
                assert moved_in_backward == (m is inside), "the matrices move inside backward() exactly when asked to"
                training.allreduce_gradients(list(m.parameters()))
                opt.step()
                assert int(opt.state[m.aug_shape[0][0].weight]["step"]) == it + 1
            dbev.append((ad.grad.clone(), bd.grad.clone()))
    finally:
        if exchange:
            dist.destroy_process_group()
    for i in range(4):
        p, q = after.aug_shape[i][0].weight, inside.aug_shape[i][0].weight
        assert torch.equal(p, q), "aug_shape.%d.0.weight" % i
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(opts[0].state[p][k], opts[1].state[q][k]), k
    for (k, p), (_, q) in zip(after.named_parameters(), inside.named_parameters()):
        assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(p.abs().max())), k
    _close("d bev", dbev[1][0], dbev[0][0], rtol=1e-5)
    _close("d prev_bev", dbev[1][1], dbev[0][1], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["plain", "lowrank", "in_backward"])
@pytest.mark.parametrize("which", ["neither", "current only"])
def test_feature_maps_without_grad_skip_their_gradient_and_change_nothing_else(mode, which):
    """Features from a frozen pipeline (requires_grad False on a BEV map): autograd asks for no gradient of that map, and the backward then
    skips its scatter-add AND the dx = ghid W1 products that feed only it (needs_input_grad) - every parameter gradient, and three Adam steps
    in each first-layer mode, bit for bit what the run with both maps' gradients gives; a matrix whose dx is not needed is stepped in
    step() even under in_backward=True."""
    import copy
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(20, 7, 5, 3, seed=11)
    dev = torch.device("cuda:0")
    base = model.to(dev).train()
    gtd = gt.to(dev)
    detd, prevd = det.to(dev).contiguous(), prev.to(dev).contiguous()
    runs = []
    for full in (True, False):
        m = copy.deepcopy(base)
        kw = {} if mode == "plain" else dict(lowrank_first_layers=m, in_backward=mode == "in_backward")
        opt = training.FusedAdam(m.parameters(), lr=1e-3, weight_decay=0.01, **kw)
        grads = None
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            ad = a.to(dev).requires_grad_(full or which == "current only")
            bd = b.to(dev).requires_grad_(full)
            before = [m.aug_shape[i][0].weight.detach().clone() for i in range(4)]
            m1, m2 = training.affinity_train(m, ad, bd, detd.clone(), prevd)
            training.affinity_loss(m1, m2, gtd).backward()
            if it == 0:
                grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
                moved = [not torch.equal(before[i], m.aug_shape[i][0].weight.detach()) for i in range(4)]
                if mode == "in_backward":  # aug_shape 0, 1 read the current map's table, 2, 3 the previous map's
                    assert moved == ([True] * 4 if full else [which == "current only"] * 2 + [False] * 2), moved
                else:
                    assert moved == [False] * 4
                assert (ad.grad is not None) == (full or which == "current only") and (bd.grad is not None) == full
                dcur = None if ad.grad is None else ad.grad.clone()
            opt.step()
        runs.append((grads, {k: p.detach().clone() for k, p in m.named_parameters()}, dcur))
    (g0, p0, d0), (g1, p1, d1) = runs
    assert set(g0) == set(g1) and len(g0) >= 60
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    if which == "current only":
        assert torch.equal(d0, d1)


@pytest.mark.gpu
@pytest.mark.parametrize("R,Rdx,H,K", [(1, 1, 5, 8), (8, 8, 70, 1032), (3, 16, 129, 260), (16, 5, 33, 4096), (24, 8, 70, 516), (64, 8, 131, 1032), (33, 12, 64, 260)])
def test_adam_lowrank_with_the_product_in_the_same_pass(R, Rdx, H, K):
    """shasta_adam_lowrank_dx_f32 against shasta_adam_lowrank_f32 (the same update, bit for bit) and Gdx W in float64 (W before the update)."""
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(R * 100 + H)
    p0 = torch.randn(H, K, device=dev)
    m0, v0 = torch.randn(H, K, device=dev) * 0.1, torch.rand(H, K, device=dev) * 0.01
    G, X, Gdx = torch.randn(R, H + 3, device=dev), torch.randn(R, K + 4, device=dev), torch.randn(Rdx, H + 1, device=dev)
    y0 = torch.randn(Rdx, K + 8, device=dev)
    hyper = (3e-3, 0.9, 0.999, 1e-8, 0.01, 7, None)  # (lr, betas, eps, weight decay, step, d_dyn = NULL: the plain form)
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    hip.check(lib.shasta_adam_lowrank_f32(hip.ptr(pa), hip.ptr(ma), hip.ptr(va), H, K, hip.ptr(G), H + 3, hip.ptr(X), K + 4, R, *hyper, hip.stream_ptr()), "ref")
    nb = lib.shasta_adam_lowrank_dx_workspace_bytes(H, K, Rdx)
    for acc in (0, 1):
        pb, mb, vb, y = p0.clone(), m0.clone(), v0.clone(), y0.clone()
        ws = torch.full(((nb + 3) // 4,), float("nan"), device=dev)
        hip.check(lib.shasta_adam_lowrank_dx_f32(hip.ptr(pb), hip.ptr(mb), hip.ptr(vb), H, K, hip.ptr(G), H + 3, hip.ptr(X), K + 4, R, hip.ptr(Gdx), H + 1, Rdx,
                                                 hip.ptr(y), K + 8, acc, hip.ptr(ws), nb, *hyper, hip.stream_ptr()), "dx")
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
        want = Gdx[:, :H].double() @ p0.double() + (y0[:, :K].double() if acc else 0.0)
        assert float((y[:, :K].double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
        assert torch.equal(y[:, K:], y0[:, K:]), "columns past K are not touched"
    assert lib.shasta_adam_lowrank_dx_f32(hip.ptr(pb), hip.ptr(mb), hip.ptr(vb), H, K, hip.ptr(G), H + 3, hip.ptr(X), K + 4, R, hip.ptr(Gdx), H + 1, Rdx,
                                          hip.ptr(y), K + 8, 0, hip.ptr(ws), nb - 4, *hyper, hip.stream_ptr()) != 0


@pytest.mark.gpu
def test_backward_at_the_headline_size_twice_the_same_bits():
    """N = M = 500, F = 256, nf = 7 (the configuration the metric is quoted on), two frame-pairs: every parameter gradient of rows 6-16 is
    a sum in a fixed order (per-pair kernels, split-K slices, partial-sum passes) - two backward passes over the same inputs give the same
    bits - finite, and not all zero; so do the gradients of the two BEV maps (the gather's backward sums a pixel's terms in a fixed order)."""
    import shasta_amd
    from shasta_amd import training
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    with torch.device(dev):
        model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                                 bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                                 max_obj=500, num_feats=7, num_point=4, in_channels=512)).train()
    B, N = 2, 500
    g = torch.Generator().manual_seed(5)
    bev = torch.relu(torch.randn(B, 180, 180, 64, generator=g)).to(dev)
    pbev = torch.relu(torch.randn(B, 180, 180, 64, generator=g)).to(dev)
    det, prev = O.synth_boxes(g, B, N, None).to(dev), O.synth_boxes(g, B, N, N - 40).to(dev)
    gt = (torch.rand(B, N + 2, N + 2, generator=g) < 0.01).float().to(dev)
    gt[:, 0, 0] = 1.0
    runs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        a, b = bev.clone().requires_grad_(True), pbev.clone().requires_grad_(True)
        m1, m2 = training.affinity_train(model, a, b, det.clone(), prev.clone())
        training.affinity_loss(m1, m2, gt).backward()
        runs.append(({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, a.grad.clone(), b.grad.clone()))
    assert len(runs[0][0]) == 2 * (8 + 4 + 8 + 3 + 3 + 6)
    for k, v in runs[0][0].items():
        assert torch.isfinite(v).all(), k
        assert torch.equal(v, runs[1][0][k]), k
    assert sum(float(v.abs().max()) > 0 for v in runs[0][0].values()) >= 60
    for x, y in ((runs[0][1], runs[1][1]), (runs[0][2], runs[1][2])):  # the BEV maps' gradients: sorted by pixel since round 6, no atomics
        assert torch.equal(x, y) and float(x.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,npnt,C,H,W,spread", [(3, 90, 5, 64, 180, 180, 2.0), (2, 500, 4, 64, 180, 180, 40.0), (1, 1700, 5, 32, 180, 180, 10.0),
                                                  (2, 40, 4, 96, 31, 57, 8.0), (2, 25, 1, 64, 8, 8, 0.2), (2, 60, 5, 16, 400, 400, 5.0), (5, 30, 4, 30, 64, 64, 4.0),
                                                  (300, 12, 5, 16, 48, 48, 3.0)])
def test_gather_backward_is_a_fixed_order_sum(B, N, npnt, C, H, W, spread):
    """shasta_bev_gather_bwd_f32 (autograd of bilinear_interpolate_torch, center_utils.py:92-121): boxes crowded into a few metres so that
    many points share pixels; against autograd of the oracle's gather in float64, and - maps below 2^17 pixels: the contributions of a
    batch item sorted by pixel in LDS, a pixel's terms added in ascending order - twice the same bits; 1700 x 5 points = 34 000
    contributions go through the kernel in three chunks, boxes off the map through the clamped indices, a 400 x 400 map (2^17 pixels or
    more) through the atomic form (equal to fp32 rounding only), 30 channels (no 16-byte accesses), 300 items (one workgroup each)."""
    from shasta_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(B * 1000 + N)
    boxes = O.synth_boxes(g, B, N, None)[:, :, :7].contiguous()
    boxes[:, :, :2] = (torch.rand(B, N, 2, generator=g) - 0.5) * spread - 20.0
    boxes[:, ::7, 0] = -60.0 - torch.rand(B, len(range(0, N, 7)), generator=g) * 5  # off the map
    F = npnt * C
    dfeat = torch.randn(B, N, F, generator=g)
    stride = 108.0 / (0.075 * W)  # the map covers the 108 m range
    bev = torch.zeros(B, H, W, C, dtype=torch.float64, requires_grad=True)
    out = O.bev_gather(bev, boxes.double(), npnt, out_stride=stride)
    out.backward(dfeat.double())
    want = bev.grad
    runs = []
    dfeat_d, boxes_d = dfeat.to(dev), boxes.to(dev)  # (named: a temporary would be freed - and its block reused - before the launch)
    for _ in range(2):
        dbev = torch.zeros(B, H, W, C, device=dev)
        hip.check(lib.shasta_bev_gather_bwd_f32(hip.ptr(dfeat_d), B, H, W, C, hip.ptr(boxes_d), N, 7, N * 7, npnt, -54.0, -54.0, 0.075, 0.075,
                                                stride, F, N * F, hip.ptr(dbev), hip.stream_ptr()), "shasta_bev_gather_bwd_f32")
        runs.append(dbev.cpu())
    touched = int((want.abs().sum(-1) > 0).sum())
    assert touched < B * N * npnt * 4 * 0.9 or spread > 20  # pixels ARE shared
    _close("d bev (gather backward)", runs[0], want, rtol=2e-5)
    if H * W < (1 << 17):
        assert torch.equal(runs[0], runs[1])
    else:
        assert float((runs[0] - runs[1]).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.gpu
def test_capturable_adam_and_the_graphed_step_equal_the_eager_loop():
    """FusedAdam(capturable=True): step number, lr and betas reach the kernels through device memory (shasta_adam_prepare_f32) - the same
    weights as the plain optimizer under OneCycleLR (which cycles lr AND beta1), with the first-layer update inside the backward; and
    training.GraphedTrainStep: the whole step (forward, loss, HIP backward, Adam) captured into a hipGraph once and replayed gives the
    weights of the eager loop, step for step - the schedule moves between replays although launch arguments are frozen."""
    import copy
    from shasta_amd import training
    c, model, w, a, b, det, prev, gt = _case(12, 7, 4, 3, seed=23)
    dev = torch.device("cuda:0")
    base = model.to(dev).train()
    ad, bd, gtd, detd, prevd = a.to(dev), b.to(dev), gt.to(dev), det.to(dev).contiguous(), prev.to(dev).contiguous()
    total = 9

    def make(capturable):
        m = copy.deepcopy(base)
        opt = training.FusedAdam(m.parameters(), lr=1e-3, weight_decay=0.01, lowrank_first_layers=m, in_backward=True, capturable=capturable)
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=5e-3, total_steps=total + 1)
        return m, opt, sched

    def eager(m, opt, sched, steps, losses):
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            m1, m2 = training.affinity_train(m, ad, bd, detd.clone(), prevd)
            loss = training.affinity_loss(m1, m2, gtd)
            loss.backward()
            opt.step()
            sched.step()
            losses.append(float(loss.detach()))
    plain, cap, graphed = make(False), make(True), make(True)
    lp, lc, lg = [], [], []
    eager(*plain, total, lp)
    eager(*cap, total, lc)
    # the graphed loop: GraphedTrainStep's three warm-up steps run at the schedule's first value (no scheduler step between them), so
    # the comparison loop does the same: three eager steps without moving the schedule, then one scheduler step per training step
    ref = make(True)
    for _ in range(3):
        ref[1].zero_grad(set_to_none=True)
        m1, m2 = training.affinity_train(ref[0], ad, bd, detd.clone(), prevd)
        training.affinity_loss(m1, m2, gtd).backward()
        ref[1].step()
    lr_ = []
    eager(*ref, total - 3, lr_)
    step = training.GraphedTrainStep(graphed[0], graphed[1], ad, bd, detd, prevd, gtd, warmup=3)
    for _ in range(total - 3):
        lg.append(float(step(ad, bd, detd, prevd, gtd)))
        graphed[2].step()
    assert lp[-1] < lp[0]
    for x, y in zip(lp, lc):
        assert abs(x - y) <= 1e-6 * max(1.0, abs(x)), (lp, lc)
    for (k, p), (_, q) in zip(plain[0].named_parameters(), cap[0].named_parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(p.abs().max())), k
    for x, y in zip(lr_, lg):
        assert abs(x - y) <= 1e-6 * max(1.0, abs(x)), (lr_, lg)
    for (k, p), (_, q) in zip(ref[0].named_parameters(), graphed[0].named_parameters()):
        assert float((p - q).abs().max()) <= 1e-6 * max(1.0, float(p.abs().max())), k
    # the device-side step counter moved with the replays; an eager forward after training sees the trained weights
    assert int(graphed[1].param_groups[0]["_shasta_dev"]["step"]) == total
    graphed[0].eval()
    ref[0].eval()
    with torch.no_grad():
        e1, _ = graphed[0].affinity_from_bev(ad, bd, detd.clone(), prevd)
        r1, _ = ref[0].affinity_from_bev(ad, bd, detd.clone(), prevd)
    assert float((e1 - r1).abs().max()) <= 1e-6
