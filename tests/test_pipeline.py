"""BASELINE configs 2-4 as a runnable whole on the synthetic-scene stand-in (shasta_amd/scenes.py writes the reference's file
schemas): loader -> batched forward -> device decode -> cp_<split>.json -> merge -> tracker -> tracking_result.json
(shasta_amd/pipeline.py) against the reference's flow restated one frame at a time on the CPU (oracle/pipeline_oracle.py).
Weights are the sharpened seeded set (tests/helpers.py), so the decode really takes newborn / dead / false-positive decisions;
the scene seed is one whose smallest decision margin (pipeline_oracle.decision_margin) is far above the kernels' error."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from oracle import pipeline_oracle as PO
from shasta_amd import pipeline, scenes
from tests.helpers import sharpen_state_dict

CLASSES = ("car", "bus", "pedestrian")
SCENE_SEED, GAINS = 7, (4.0, 2.0)


def _split(tmp_path, n_scenes=2, frames=4, seed=SCENE_SEED):
    return scenes.write_synthetic_split(str(tmp_path), n_scenes=n_scenes, frames_per_scene=frames, seed=seed)


def _models(classes=CLASSES):
    import shasta_amd
    ms = {}
    for name in classes:
        torch.manual_seed(3)
        m = shasta_amd.build_simp_track(pipeline.class_model_cfg(name)).eval()
        sharpen_state_dict(m.state_dict(), *GAINS)
        ms[name] = m
    return ms


def _weights(models):
    return {n: {k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for n, m in models.items()}


def _same_cp(got, want, tol=2e-3):
    assert set(got["results"]) == set(want["results"])
    for token, w in want["results"].items():
        g = got["results"][token]
        assert len(g) == len(w), token
        for a, b in zip(g, w):
            assert a["detection_name"] == b["detection_name"] and a["translation"] == b["translation"]
            assert a["detection_score"] == b["detection_score"]
            for flag in ("newborn", "FN", "dead"):
                assert (flag in a) == (flag in b), (token, flag)
            assert abs(a["ref_detection_score"] - b["ref_detection_score"]) <= tol


def _same_tracking(got, want, tol=2e-3):
    assert set(got["results"]) == set(want["results"])
    for token, w in want["results"].items():
        g = got["results"][token]
        key = lambda rows: [(r["tracking_id"], r["tracking_name"], r["translation"]) for r in rows]  # noqa: E731
        assert key(g) == key(w), token
        assert all(abs(a["tracking_score"] - b["tracking_score"]) <= tol for a, b in zip(g, w))


def test_synthetic_split_has_the_reference_file_schemas(tmp_path):
    paths, sc = _split(tmp_path)
    fi = json.load(open(paths["frame_info_path"]))
    tokens = [t for _, toks in sc for t in toks]
    assert set(fi) == set(tokens)
    for (_, toks) in sc:
        assert fi[toks[0]]["prev"] == "" and fi[toks[1]]["prev"] == toks[0] and fi[toks[-1]]["next"] == ""
        assert fi[toks[1]]["timestamp"] - fi[toks[1]]["prev_timestamp"] == 500000  # microseconds (get_frame_info.py:38-44)
    rows = json.load(open(os.path.join(paths["det_path"], tokens[1] + ".json")))
    cls = json.load(open(os.path.join(paths["cls_info_path"], tokens[1] + ".json")))
    assert len(rows) == len(cls) > 10 and all(len(r) == 13 for r in rows)
    assert {"sample_token", "translation", "size", "rotation", "velocity", "detection_name", "detection_score", "attribute_name"} <= set(cls[0])
    meta = json.load(open(paths["frames_meta_path"]))["frames"]
    assert [m["token"] for m in meta] == tokens and sum(m["first"] for m in meta) == len(sc)
    from shasta_amd import frames
    s = frames.FramePairs(paths["det_path"], paths["cls_info_path"], paths["frame_info_path"], det_type=["car"], max_objects=90,
                          test_mode=True).load(tokens[1], known_tokens=set(tokens))
    assert 5 < s["num_det_boxes"] <= 90 and abs(s["det_boxes"][0, 9] - 0.5) < 1e-9


def test_oracle_chain_takes_real_decisions_with_a_wide_margin(tmp_path):
    paths, sc = _split(tmp_path)
    per_class, merged, tracking, margin = PO.run_split(_weights(_models()), pipeline.CLASS_CONFIGS, paths, sc, scenes.TokenBev())
    assert margin > 2e-2  # the GPU comparison below is only meaningful when no decision sits on a threshold
    flags = {f: sum(1 for c in per_class.values() for v in c["results"].values() for a in v if f in a) for f in ("newborn", "dead")}
    assert flags["newborn"] >= 5 and flags["dead"] >= 5
    ids = {a["tracking_id"] for v in tracking["results"].values() for a in v}
    assert len(ids) > 20 and set(merged["results"]) == set(tracking["results"])


def _oracle_forward(weights, nf=3, npnt=5):
    from oracle import shasta_oracle as O

    def fwd(batch):
        det, prev = batch["det_boxes"].float(), batch["prev_det_boxes"].float()
        m1, m2 = O.forward_from_bev(weights, batch["bev_feature"], batch["prev_bev_feature"], det, prev, nf, npnt)
        batch["det_boxes"] = det
        return m1, m2, batch
    return fwd


def test_batched_chain_equals_frame_by_frame_chain_on_cpu(tmp_path):
    """The package's chain (batches of 5 frame pairs, per-scene trackers stepped together is GPU-only, so cp + merge here) with
    the oracle forward plugged in equals the reference-style frame-by-frame chain: batching, collate, decode bookkeeping, merge."""
    paths, sc = _split(tmp_path)
    W = _weights(_models())
    per_class, merged, _, _ = PO.run_split(W, pipeline.CLASS_CONFIGS, paths, sc, scenes.TokenBev())
    got = pipeline.run_split({n: None for n in W}, paths, sc, scenes.TokenBev(), torch.device("cpu"), batch_pairs=5,
                             forward_override={n: _oracle_forward(W[n]) for n in W}, tracker_on_device=False, work_dir=str(tmp_path / "out"))
    for name in W:
        _same_cp(got[0][name], per_class[name], tol=1e-4)  # torch-CPU GEMMs block differently per batch size
        assert json.load(open(tmp_path / "out" / name / "cp_val.json"))["results"].keys() == per_class[name]["results"].keys()
    _same_cp(got[1], merged, tol=1e-4)


def _gloo_worker(rank, world, port, root, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    paths = scenes.split_paths(root)
    meta = json.load(open(paths["frames_meta_path"]))["frames"]
    sc, cur = [], None
    for fr in meta:
        if fr["first"]:
            cur = ("scene-%04d" % len(sc), [])
            sc.append(cur)
        cur[1].append(fr["token"])
    W = _weights(_models(("car", "bus")))
    res = pipeline.run_split({n: None for n in W}, paths, sc, scenes.TokenBev(), torch.device("cpu"), batch_pairs=3, rank=rank, world=world,
                             forward_override={n: _oracle_forward(W[n]) for n in W}, tracker_on_device=False)
    if rank == 0:
        json.dump(res[1], open(out, "w"))
    else:
        assert res is None
    dist.destroy_process_group()


def test_two_rank_scene_sharded_chain_equals_single_rank(tmp_path):
    """BASELINE config 4 on the CPU (gloo, world_size 2): scenes sharded over the ranks, per-class results gathered on rank 0,
    merged json identical to the single-rank run (and so to the frame-by-frame oracle chain, previous test)."""
    import socket
    import torch.multiprocessing as mp
    paths, sc = _split(tmp_path, n_scenes=3, frames=3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "merged2.json")
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    W = _weights(_models(("car", "bus")))
    single = pipeline.run_split({n: None for n in W}, paths, sc, scenes.TokenBev(), torch.device("cpu"), batch_pairs=3,
                                forward_override={n: _oracle_forward(W[n]) for n in W}, tracker_on_device=False)
    _same_cp(json.load(open(out)), single[1], tol=0.0)


def test_eight_rank_scene_sharded_chain_with_uneven_scenes_equals_single_rank(tmp_path):
    """BASELINE config 4's world size on the CPU (gloo, 8 ranks): 11 scenes of 2 to 5 frames, so the ranks hold different numbers of
    scenes and frames (three ranks two scenes, five ranks one) and batches of 3 frame pairs end ragged; the merged json gathered on
    rank 0 is identical to the single-rank run.  What is left for the first 8-GPU lease is RCCL itself."""
    import socket
    import torch.multiprocessing as mp
    lengths = [3, 5, 2, 4, 3, 2, 5, 3, 4, 2, 3]
    paths, sc = _split(tmp_path, n_scenes=len(lengths), frames=lengths)
    assert [len(t) for _, t in sc] == lengths
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "merged8.json")
    mp.spawn(_gloo_worker, args=(8, port, str(tmp_path), out), nprocs=8, join=True)
    W = _weights(_models(("car", "bus")))
    single = pipeline.run_split({n: None for n in W}, paths, sc, scenes.TokenBev(), torch.device("cpu"), batch_pairs=3,
                                forward_override={n: _oracle_forward(W[n]) for n in W}, tracker_on_device=False)
    _same_cp(json.load(open(out)), single[1], tol=0.0)


@pytest.mark.gpu
def test_split_pipeline_on_device_matches_the_reference_style_chain(tmp_path):
    """Configs 2-3 on the GPU: HIP forward (batches of 8 frame pairs), device decode decisions, device tracker step for all
    scenes at once - cp json per class, merged json and tracking json equal the CPU oracle chain (flags, ids, order exactly;
    scores within 2e-3: sharpened logits)."""
    paths, sc = _split(tmp_path)
    dev = torch.device("cuda", 0)
    models = _models()
    W = _weights(models)
    want_pc, want_merged, want_trk, margin = PO.run_split(W, pipeline.CLASS_CONFIGS, paths, sc, scenes.TokenBev())
    assert margin > 2e-2
    models = {n: m.to(dev) for n, m in models.items()}
    got_pc, got_merged, got_trk = pipeline.run_split(models, paths, sc, scenes.TokenBev(), dev, work_dir=str(tmp_path / "work"), batch_pairs=8)
    for name in CLASSES:
        _same_cp(got_pc[name], want_pc[name])
    _same_cp(got_merged, want_merged)
    _same_tracking(got_trk, want_trk)
    assert os.path.exists(tmp_path / "work" / "tracking_result.json") and os.path.exists(tmp_path / "work" / "merged_cp_val.json")
    # host decode (one copy of the matrices per batch, the restated loop) gives the same lists as the device decisions
    host_pc, host_merged, _ = pipeline.run_split(models, paths, sc, scenes.TokenBev(), dev, batch_pairs=8, decode_on_device=False)
    _same_cp(host_merged, got_merged, tol=1e-7)
    # and the result does not depend on how the frame pairs are batched beyond fp32 summation order
    b3_pc, b3_merged, _ = pipeline.run_split(models, paths, sc, scenes.TokenBev(), dev, batch_pairs=3)
    _same_cp(b3_merged, got_merged, tol=1e-4)


@pytest.mark.gpu
def test_all_seven_classes_run_through_the_chain(tmp_path):
    """Config 3: the seven class configurations (tables 90 / 60 / 50 / 20, nf=3, np=5) in one split run, random-init weights."""
    paths, sc = _split(tmp_path, n_scenes=2, frames=3, seed=1)
    dev = torch.device("cuda", 0)
    models = {n: pipeline.build_class_model(n, dev, seed=5) for n in pipeline.CLASS_CONFIGS}
    pc, merged, trk = pipeline.run_split(models, paths, sc, scenes.TokenBev(), dev, batch_pairs=6)
    assert set(pc) == set(pipeline.CLASS_CONFIGS)
    n_in = sum(len(json.load(open(os.path.join(paths["cls_info_path"], t + ".json")))) for _, toks in sc for t in toks)
    n_out = sum(len(v) for v in merged["results"].values())
    assert 0 < n_out <= n_in + 50 and set(trk["results"]) == set(merged["results"])


def test_the_chain_runs_with_the_cyclic_collector_off_and_restores_it():
    """pipeline._no_cyclic_gc: the collector is off inside (the chain's heap of live dicts would be re-scanned at every generation-2 pass)
    and back in its previous state afterwards, also after an exception."""
    import gc
    assert gc.isenabled()
    with pipeline._no_cyclic_gc():
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(RuntimeError):
        with pipeline._no_cyclic_gc():
            raise RuntimeError("x")
    assert gc.isenabled()
    gc.disable()
    try:
        with pipeline._no_cyclic_gc():
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
