"""Decode pinned on the reference: the annotation lists its own eval loop produced for
seeded synthetic matrices (tests/golden/decode_golden.json.gz, made by tests/golden/make_decode_golden.py by executing
tools/nusc_shasta/eval.py:111-181 in place) against `AffinityDecoder` (host) and the device decision kernel."""
import copy
import gzip
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_golden.json.gz")


def _load():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)


def _batch(fr, N):
    pb = torch.zeros(1, N, 11)
    pb[0, 0, 9] = fr["time_lag"]
    return dict(metadata=[{"token": fr["token"]}], prev_metadata=[{"token": fr["prev_token"]}], cls_det_boxes=[copy.deepcopy(fr["cls"])],
                prev_cls_det_boxes=[copy.deepcopy(fr["prev_cls"])], prev_det_boxes=pb)


def _same(got, want):
    assert list(got.keys()) == list(want.keys())
    for tok in want:
        assert len(got[tok]) == len(want[tok]), tok
        for g, w in zip(got[tok], want[tok]):
            assert set(g.keys()) == set(w.keys()), (tok, g.get("uid"))
            for k in w:
                if isinstance(w[k], float):
                    assert abs(g[k] - w[k]) <= 1e-12, (tok, k)
                elif isinstance(w[k], list):
                    assert np.allclose(g[k], w[k], rtol=0, atol=1e-12), (tok, k)
                else:
                    assert g[k] == w[k], (tok, k)


def test_golden_covers_every_branch():
    g = _load()
    annos = sum(g["results"].values(), [])
    assert sum(1 for a in annos if a.get("FN")) >= 5 and sum(1 for a in annos if a.get("newborn")) >= 5
    assert sum(1 for a in annos if a.get("dead")) >= 2 and any(len(fr["cls"]) == 0 for fr in g["frames"])


def test_host_decoder_matches_reference_loop():
    from shasta_amd.decode import AffinityDecoder
    g = _load()
    dec = AffinityDecoder()
    for fr in g["frames"]:
        dec.add(torch.tensor(fr["m1"]), torch.tensor(fr["m2"]), _batch(fr, g["N"]))
    out = dec.finalize()
    _same(out["results"], g["results"])
    assert {k: v for k, v in dec.dead_tracker.items()} == g["dead_tracker"]


@pytest.mark.gpu
def test_device_decisions_match_reference_loop():
    """All frames decoded from ONE launch of the decision kernel (shasta_decode_flags_f32), then the dict bookkeeping."""
    from shasta_amd.decode import decode_flags_device, decode_frame_from_flags
    g = _load()
    N = g["N"]
    dev = torch.device("cuda:0")
    m1 = torch.tensor([fr["m1"][0] for fr in g["frames"]], device=dev)
    m2 = torch.tensor([fr["m2"][0] for fr in g["frames"]], device=dev)
    n_prev = [len(fr["prev_cls"]) for fr in g["frames"]]
    n_cur = [len(fr["cls"]) for fr in g["frames"]]
    pc, ps, df, ds = decode_flags_device(m1, m2, n_prev, n_cur)
    results, dead = {}, {}
    for b, fr in enumerate(g["frames"]):
        cls, prev_cls = copy.deepcopy(fr["cls"]), copy.deepcopy(fr["prev_cls"])
        dead.setdefault(fr["token"], {"dead_idx": [], "keep_idx": []})
        annos, dead_prev, keep = decode_frame_from_flags(pc[b], ps[b], df[b], ds[b], cls, prev_cls, fr["token"], fr["time_lag"])
        if prev_cls:
            dead.setdefault(fr["prev_token"], {"dead_idx": [], "keep_idx": []})["dead_idx"].extend(dead_prev)
        if cls:
            dead[fr["token"]]["keep_idx"] = keep
        results[fr["token"]] = annos
    for tok, annos in results.items():
        for i in dead[tok]["dead_idx"]:
            if i in dead[tok]["keep_idx"]:
                annos[dead[tok]["keep_idx"].index(i)]["dead"] = True
    _same(results, g["results"])
