"""Generates tests/golden/train_*.npz: loss and parameter gradients of the REFERENCE affinity network
(det3d/models/tracker/shasta.py forward in train() mode + the loss of tools/nusc_shasta/train.py:200-211 + autograd) on
seeded synthetic inputs, for the hand-written backward of shasta_amd/training.py.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_train_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_import as R  # noqa: E402
from oracle import shasta_oracle as O  # noqa: E402  (only for the shared synthetic-input generator)

CONFIGS = [dict(name="train_6_7_1", max_obj=6, nf=7, np=1, B=2, n_real=None, seed=31),
           dict(name="train_12_3_4", max_obj=12, nf=3, np=4, B=3, n_real=9, seed=32),
           dict(name="train_10_7_5", max_obj=10, nf=7, np=5, B=1, n_real=None, seed=33),
           # the shipped car shape end to end (configs/nusc/car.py: max_obj 90, nf 3, np 5; neck maps 512 x 180 x 180, stride 8), padded
           # tables, two frame pairs: 132 MB per map - the inputs are re-created from the seed by the tests (checksums stored)
           dict(name="train_90_3_5", max_obj=90, nf=3, np=5, B=2, n_real=35, seed=34, cin=512, hw=180, stride=8)]
CIN, HW, STRIDE = 8, 24, 64


def run(c):
    torch.manual_seed(c["seed"])
    CIN, HW, STRIDE = c.get("cin", 8), c.get("hw", 24), c.get("stride", 64)
    m = R.build_ref_model(c["max_obj"], c["nf"], c["np"], in_channels=CIN, out_stride=STRIDE).train()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    bev, pbev, det, prev = O.synth_case(c["B"], c["max_obj"], c["n_real"], CIN, HW, HW, c["seed"])
    gen = torch.Generator().manual_seed(c["seed"])
    N = c["max_obj"]
    gt = (torch.rand(c["B"], N + 2, N + 2, generator=gen) < 0.15).float()
    gt[:, 0, 0] = 1.0
    m.extract_feat = lambda ex: (bev, None, pbev, None)
    example = dict(det_boxes=det.clone(), prev_det_boxes=prev.clone())
    matched1, matched2, _ = m(example, train_mode=True)
    # tools/nusc_shasta/train.py:200-211
    gt1, gt2 = gt[:, :-2, :], gt[:, :, :-2]
    loss_f = (gt1 * (-torch.log(matched1 + 1e-10))).sum() / gt1.sum()
    loss_b = (gt2 * (-torch.log(matched2 + 1e-10))).sum() / gt2.sum()
    loss = (loss_f + loss_b) / 2
    loss.backward()
    out = dict(cfg=np.array([c["max_obj"], c["nf"], c["np"], c["B"], -1 if c["n_real"] is None else c["n_real"], CIN, HW, STRIDE, c["seed"]]),
               det=det.numpy(), prev=prev.numpy(), gt=gt.numpy(), loss=np.array(float(loss)),
               matched1=matched1.detach().numpy(), matched2=matched2.detach().numpy())
    if bev.numel() <= 1 << 20:
        out.update(bev=bev.numpy(), pbev=pbev.numpy())
    else:  # oracle.synth_case(B, max_obj, n_real, cin, hw, hw, seed) re-creates them; [sum, sum |.|, sum of squares] in float64
        for k, v in (("bevc", bev), ("pbevc", pbev)):
            out[k] = np.array([float(v.double().sum()), float(v.double().abs().sum()), float((v.double() ** 2).sum())])
    # the batch statistics the two BatchNorm calls left behind (shasta.py:223-228: current map, then previous map)
    out["running_mean"] = m.shared_conv[1].running_mean.numpy().copy()
    out["running_var"] = m.shared_conv[1].running_var.numpy().copy()
    n = 0
    small = sum(v.numel() for v in sd.values()) <= 200000
    if small:  # tiny case: the weights travel with the fixture; otherwise they are re-created from the seed (checksums stored)
        for k, v in sd.items():
            out["w::" + k] = v.numpy()
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            out["wc::" + k] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.reshape(-1)
        out["gc::" + k] = np.array([float(g.double().sum()), float(g.double().abs().sum()), float(g.abs().max())])
        if g.numel() <= 20000:
            out["g::" + k] = p.grad.numpy()
        else:  # large tensors: checksums + every stride-th element
            stride = g.numel() // 5000
            out["gs::" + k] = g[::stride].numpy().copy()
            out["gstride::" + k] = np.array([stride])
        n += 1
    path = os.path.join(HERE, c["name"] + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", n, "gradients; loss", float(loss))


if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CONFIGS:
        if not only or c["name"] in only:
            run(c)
