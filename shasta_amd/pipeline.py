"""The inference chain of BASELINE configs 2-4 as one runnable whole (the reference spreads it over three CLIs,
official_val.sh):

  tools/nusc_shasta/eval.py:90-193        per class: dataset -> batch -> track_batch_processor -> decode loop -> cp_<split>.json
  tools/nusc_shasta/merge_results.py:37-59  the seven per-class files merged per sample token            -> merged_cp_<split>.json
  tools/nusc_shasta/pub_test.py:88-162     merged detections -> PubTrackerMerged per scene                  -> tracking_result.json

Here: `FramePairs` (the dataset's detection side) -> `collate_pairs` (any number of frames in one batch: the reference runs
batch_size=1) -> `Shasta.forward` through `track_batch_processor` -> device decode decisions -> `AffinityDecoder`; scenes are
sharded over the ranks (one process per GPU, replica.shard_scenes) and only the decoded per-token lists travel to rank 0
(replica.gather_decoded, no collective in the forward); the tracker advances ALL scenes of the split in lock step, one launch of
the centre-distance / greedy kernel per frame index.  The spconv backbone and the neck are out of scope (SURVEY.md section 2 rows
10-11): `bev` is any callable token -> (H, W, C) NHWC feature map after shared_conv (`scenes.TokenBev` for the synthetic split).
"""
import contextlib
import copy
import gc
import json
import os
import time

import numpy as np
import torch

from . import builder, decode, frames, replica
from .pub_tracker import NUSCENES_TRACKING_NAMES, PubTracker, PubTrackerMerged, step_batch, step_batch_merged
from .train_track import track_batch_processor

# configs/nusc/<class>.py:26-29,66-71: table size per class; every shipped config has num_feats=3, num_point=5 (F=320)
CLASS_CONFIGS = {"bicycle": 50, "bus": 20, "car": 90, "motorcycle": 50, "pedestrian": 90, "trailer": 60, "truck": 60}
META = {"use_camera": False, "use_lidar": True, "use_radar": False, "use_map": False, "use_external": False}


class StageTimer:
    """Wall time per stage of the chain (bench.py `extra.pipeline`): `with timer.stage("forward"): ...`; with sync=True the device is
    synchronised on both sides of a stage, so that asynchronous launches are charged to the stage that issued them."""

    def __init__(self, sync=True):
        self.sync, self.seconds = sync, {}

    @contextlib.contextmanager
    def stage(self, name):
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            yield
        finally:
            if self.sync and torch.cuda.is_available():
                torch.cuda.synchronize()
            self.seconds[name] = self.seconds.get(name, 0.0) + time.perf_counter() - t0


_NO_TIMER = StageTimer(sync=False)


@contextlib.contextmanager
def _no_cyclic_gc():
    """The chain builds hundreds of thousands of small dicts and lists that stay alive until the split is done (nuScenes-format detections:
    acyclic trees).  CPython's cyclic collector re-scans that growing heap at every generation-2 pass - measured: the loader alone 1.0 - 1.3 s
    per 800 frames with the collector on, 0.45 - 0.5 s with it off, and most of the run-to-run spread of the whole chain.  Reference counting
    still frees everything the chain drops; the collector is switched back on (if it was on) when the split is done."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def class_model_cfg(name, num_feats=3, num_point=5):
    """The `model = dict(type="Shasta", ...)` block of configs/nusc/<class>.py without the out-of-scope reader / backbone / neck."""
    return dict(type="Shasta", reader=None, backbone=None, neck=None,
                bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                max_obj=CLASS_CONFIGS[name], num_feats=num_feats, num_point=num_point)


def build_class_model(name, device, checkpoint=None, seed=0, **kw):
    """tools/nusc_shasta/eval.py:80-88: build from the class config, load the checkpoint when there is one (random init
    under `seed` otherwise: the shipped *.pth files are external downloads)."""
    torch.manual_seed(seed)
    model = builder.build_simp_track(class_model_cfg(name, **kw)).eval()
    if checkpoint is not None:
        from .shasta import load_state_dict_permissive
        sd = torch.load(checkpoint, map_location="cpu")
        load_state_dict_permissive(model, sd.get("state_dict", sd) if isinstance(sd, dict) else sd)
    return model.to(device)


def eval_class(model, name, paths, tokens, bev, device, known_tokens=None, batch_pairs=32, decode_on_device=True, forward=None, timer=None):
    """eval.py:96-181 for one class over `tokens` (in time order inside every scene).  Returns the AffinityDecoder (not finalized:
    the `dead` post-pass needs the whole split, replica.gather_decoded / finalize do it).  `forward`: replaces the device model
    call - the tests pass the CPU oracle here to obtain the reference-side result of the same chain."""
    ds = frames.FramePairs(paths["det_path"], paths["cls_info_path"], paths["frame_info_path"], det_type=[name],
                           max_objects=CLASS_CONFIGS[name], test_mode=True)
    known = set(ds.frame_info.keys()) if known_tokens is None else set(known_tokens)
    dec = decode.AffinityDecoder()
    timer = _NO_TIMER if timer is None else timer
    for i in range(0, len(tokens), batch_pairs):
        with timer.stage("loader + collate"):
            samples = [ds.load(t, known_tokens=known) for t in tokens[i:i + batch_pairs]]
            batch = frames.collate_pairs(samples)
        with timer.stage("bev maps"):
            batch["bev_feature"] = torch.stack([bev(s["token"]) for s in samples])
            batch["prev_bev_feature"] = torch.stack([bev(s["prev_token"] or s["token"]) for s in samples])  # nuscenes.py:399-406
        if forward is not None:
            m1, m2, example = forward(batch)
            dec.add_batch(m1, m2, example, on_device=False)
            continue
        with timer.stage("h2d + forward"):
            with torch.no_grad():
                m1, m2, example = track_batch_processor(model, batch, train_mode=False, local_rank=device.index or 0)
        with timer.stage("decode"):
            dec.add_batch(m1, m2, example, on_device=decode_on_device)
    return dec


def merge_results(per_class):
    """merge_results.py:37-59: per sample token, the class lists concatenated in NUSCENES_TRACKING_NAMES order."""
    out = {"meta": dict(META), "results": {}}
    for name in NUSCENES_TRACKING_NAMES:
        if name not in per_class:
            continue
        for token, annos in per_class[name]["results"].items():
            out["results"].setdefault(token, []).extend(annos)
    return out


def _scenes_fit_device_tracker(results, max_dets=512):
    """A cheap necessary condition for the whole-scene kernel (512 detections per frame); the kernel itself reports track overflow."""
    return all(len(v) <= max_dets for v in results.values())


def _scene_frames(predictions, sc, merged):
    """[(detections of the frame, time since the previous frame)] of one scene (a list of its frames' meta entries), or None when - plain
    tracker - a frame has detections but none of a tracking class (the per-frame path then raises like the reference)."""
    last, fr = None, []
    for m in sc:
        if m["first"]:
            last = m["timestamp"]
        dets = predictions[m["token"]]
        if not merged and dets and not any(d["detection_name"] in NUSCENES_TRACKING_NAMES for d in dets):
            return None
        fr.append((dets, m["timestamp"] - last))
        last = m["timestamp"]
    return fr


def _tracking_rows(results, sc, rows, merged, refine_confidence):
    """The result rows of one scene (pub_test.py:125-140 / eval.py:263-280) from the whole-scene kernel's (dict, id, score) triples."""
    for m, items in zip(sc, rows):
        token = m["token"]
        if merged:
            results[token] = [
                {"sample_token": token, "translation": d["translation"], "size": d["size"], "rotation": d["rotation"], "velocity": d["velocity"],
                 "tracking_id": str(tid), "tracking_name": d["detection_name"], "tracking_score": ref} for d, tid, ref in items]
        else:
            results[token] = [
                {"sample_token": token, "translation": d["translation"], "size": d["size"], "rotation": d["rotation"], "velocity": d["velocity"],
                 "tracking_id": str(tid), "tracking_name": d["detection_name"],
                 "tracking_score": ref if refine_confidence else d["detection_score"], "attribute_name": d["attribute_name"]} for d, tid, ref in items]


def _track_scenes_on_device(predictions, scenes, max_age, merged=True, refine_confidence=False, alpha=0.5, beta=0.5):
    """run_tracking's fast path: the greedy tracker (merged or plain) of every scene in one launch
    (pub_tracker.track_scenes_merged_device); None when a scene exceeds the kernel's capacities or - plain tracker - a frame has
    detections but none of a tracking class (the per-frame path then raises like the reference)."""
    from .pub_tracker import track_scenes_merged_device
    frames = []
    for sc in scenes:
        fr = _scene_frames(predictions, sc, merged)
        if fr is None:
            return None
        frames.append(fr)
    out = track_scenes_merged_device(frames, max_age=max_age, plain=not merged, refine_confidence=refine_confidence, alpha=alpha, beta=beta)
    if any(o is None for o in out):
        return None
    annos = {"results": {}, "meta": dict(META)}
    for sc, rows in zip(scenes, out):
        _tracking_rows(annos["results"], sc, rows, merged, refine_confidence)
    return annos


class _SceneTrackers:
    """The chain's tracker, scene by scene while the device still works on later scenes: as soon as every frame of a scene has been decoded
    for all classes (its `dead` marks included - they come from the NEXT frame's decode, eval.py:175-181), the scene's class lists are
    merged and its whole merged-tracker run is queued on a side stream (pub_tracker.track_scenes_launch); collect() waits for the copies
    back and builds the rows in the order of the frames file.  None from collect(): a scene did not fit the kernel - the caller runs
    run_tracking over the finished split instead."""

    def __init__(self, meta, decs, names, max_age, device):
        self.scenes = []
        for fr in meta:
            if fr["first"]:
                self.scenes.append([])
            self.scenes[-1].append(fr)
        self.scene_of = {m["token"]: k for k, sc in enumerate(self.scenes) for m in sc}
        self.left = [len(sc) for sc in self.scenes]
        self.decs, self.names, self.max_age, self.device = decs, names, max_age, device
        self.stream = torch.cuda.Stream(device=device)
        self.handles = [None] * len(self.scenes)
        self.merged = {}

    def frames_done(self, tokens):
        from .pub_tracker import track_scenes_launch
        for t in tokens:
            k = self.scene_of.get(t)
            if k is None:
                continue
            self.left[k] -= 1
            if self.left[k] != 0:
                continue
            sc = self.scenes[k]
            toks = [m["token"] for m in sc]
            for n in self.names:
                self.decs[n].finalize_tokens(toks)
            for tok in toks:  # merge_results.py:37-59 for these tokens
                row = []
                for n in self.names:
                    row.extend(self.decs[n].results.get(tok, ()))
                self.merged[tok] = row
            if all(len(self.merged[tok]) <= 512 for tok in toks):
                self.handles[k] = track_scenes_launch([_scene_frames(self.merged, sc, True)], max_age=self.max_age, device=self.device, stream=self.stream)

    def collect(self):
        from .pub_tracker import track_scenes_collect
        if any(h is None for h in self.handles):
            return None
        annos = {"results": {}, "meta": dict(META)}
        for sc, h in zip(self.scenes, self.handles):
            rows = track_scenes_collect(h)[0]
            if rows is None:
                return None
            _tracking_rows(annos["results"], sc, rows, True, False)
        return annos


def run_tracking(predictions, frames_meta, max_age=4, hungarian=False, merged=True, refine_confidence=False, alpha=0.5, beta=0.5,
                 tracker_factory=None, batch_step=None, whole_scenes=False):
    """pub_test.py:88-162 (merged=True, PubTrackerMerged, tracking_score = ref_detection_score) or eval.py:226-300
    (merged=False, PubTracker).  The reference walks the frames of all scenes in file order with one tracker that is reset at
    every scene start; scenes are independent, so here every scene has its own tracker and all scenes advance together, one
    kernel launch per frame index.  whole_scenes=True (greedy assignment, either tracker): every scene's whole run is ONE kernel launch and the
    detection dicts are left untouched (the per-frame path annotates them in place, as the reference does); same rows."""
    scenes = []
    for fr in frames_meta:
        if fr["first"]:
            scenes.append([])
        scenes[-1].append(fr)
    if whole_scenes:
        if not hungarian and tracker_factory is None and _scenes_fit_device_tracker(predictions):
            annos = _track_scenes_on_device(predictions, scenes, max_age, merged, refine_confidence, alpha, beta)
            if annos is not None:
                return annos
        # beyond the kernel's capacities (or another tracker was asked for): the per-frame path below, on shallow copies - it writes
        # top-level keys into the detection dicts and never into their lists
        predictions = {tok: [dict(d) for d in annos] for tok, annos in predictions.items()}
    if tracker_factory is None:
        tracker_factory = (lambda: PubTrackerMerged(max_age=max_age, hungarian=hungarian)) if merged else \
            (lambda: PubTracker(max_age=max_age, hungarian=hungarian, refine_confidence=refine_confidence, alpha=alpha, beta=beta))
        batch_step = step_batch_merged if merged else step_batch
    trackers = [tracker_factory() for _ in scenes]
    last = [None] * len(scenes)
    annos = {"results": {}, "meta": dict(META)}
    for fi in range(max(len(s) for s in scenes) if scenes else 0):
        live = [k for k, s in enumerate(scenes) if fi < len(s)]
        lags, preds = [], []
        for k in live:
            fr = scenes[k][fi]
            if fr["first"]:
                last[k] = fr["timestamp"]
            lags.append(fr["timestamp"] - last[k])
            last[k] = fr["timestamp"]
            preds.append(predictions[fr["token"]])
        if batch_step is not None:
            outs = batch_step([trackers[k] for k in live], preds, lags)
        else:
            outs = [trackers[k].step_centertrack(p, lag) for k, p, lag in zip(live, preds, lags)]
        for k, out in zip(live, outs):
            token = scenes[k][fi]["token"]
            rows = []
            for item in out:
                if item["active"] == 0:
                    continue
                row = {"sample_token": token, "translation": item["translation"], "size": item["size"], "rotation": item["rotation"],
                       "velocity": item["velocity"], "tracking_id": str(item["tracking_id"]), "tracking_name": item["detection_name"],
                       "tracking_score": item["ref_detection_score"] if (merged or refine_confidence) else item["detection_score"]}
                if not merged:
                    row["attribute_name"] = item["attribute_name"]
                rows.append(row)
            annos["results"][token] = rows
    return annos


def _scene_runs(scenes, mine, known, frame_info, batch_pairs):
    """The frames of this rank's scenes cut into runs of at most batch_pairs CONSECUTIVE frames of one scene.  A run is a list of
    (token, prev token or ""); inside a run the previous frame of element i is element i - 1, so one stack of maps
    [prev of the first, frame 0, frame 1, ...] serves the whole run: current maps = stack[1:], previous maps = stack[:-1] (views)."""
    for name, toks in scenes:
        if name not in mine:
            continue
        run = []
        for t in toks:
            prev = frame_info[t]["prev"]
            prev = prev if prev in known else ""
            # a frame whose previous frame is not the list's preceding element (a token list that skips frames, a frame whose `prev`
            # is missing from the split: eval.py then falls back to the frame itself) starts a run of its own: its previous maps are
            # looked up by token (row 0 of the run's stack), exactly as the class-major chain does frame by frame
            if run and (len(run) == batch_pairs or prev != run[-1][0]):
                yield run
                run = []
            run.append((t, prev))
        if run:
            yield run


def _run_features(bev, models, names, run, device, bank_cache, timer=_NO_TIMER):
    """NHWC feature maps of the run's frames for every class: {class: (len(run) + 1, H, W, C) device tensor}; row 0 is the previous frame
    of the run's first frame (that frame itself at a scene start, nuscenes.py:399-406).  Three kinds of `bev`:
      * has `neck_batch(tokens, device)` -> (n, Cin, H, W) neck outputs on the device: every class head's shared_conv runs in ONE launch
        over them (shared_conv.SharedConvBank), once per frame - the reference convolves every frame twice per class;
      * has `device_batch(tokens, device)` -> (n, H, W, C) features on the device, shared by all classes;
      * a plain callable token -> (H, W, C) CPU tensor (tests: the CPU oracle must see the same bytes): stacked, copied once per run."""
    toks = [run[0][1] or run[0][0]] + [t for t, _ in run]
    if hasattr(bev, "neck_batch"):
        if "bank" not in bank_cache:
            from .shared_conv import SharedConvBank
            bank_cache["bank"] = SharedConvBank([models[n] for n in names])
        with timer.stage("maps: neck outputs (stand-in for backbone + neck)"):
            x = bev.neck_batch(toks, device)
        with timer.stage("maps: shared_conv, all class heads (K0)"):
            with torch.no_grad():
                outs = bank_cache["bank"](x, bound=getattr(bev, "neck_bound", None))  # the producer's bound of its maps, if it has one
        return dict(zip(names, outs))
    with timer.stage("maps: features (stand-in for backbone + neck + shared_conv)"):
        if hasattr(bev, "device_batch"):
            f = bev.device_batch(toks, device)
        else:
            f = torch.stack([bev(t) for t in toks])
            f = (f.pin_memory() if device.type == "cuda" else f).to(device, non_blocking=True)
    return {n: f for n in names}


def run_split(models, paths, scenes, bev, device, manage_gc=True, **kw):
    """Configs 2-4 end to end (see _run_split for the arguments).  manage_gc=True (default): the cyclic garbage collector of the PROCESS is
    switched off for the duration (_no_cyclic_gc: the loader is 2x faster without its generation-2 passes) and restored afterwards - not
    thread-safe; a caller that runs other Python threads, or manages the collector itself, passes manage_gc=False."""
    if not manage_gc:
        return _run_split(models, paths, scenes, bev, device, **kw)
    with _no_cyclic_gc():
        return _run_split(models, paths, scenes, bev, device, **kw)


def _run_split(models, paths, scenes, bev, device, work_dir=None, split="val", max_age=4, batch_pairs=32, decode_on_device=True,
               rank=0, world=1, group=None, forward_override=None, tracker_on_device=True, timer=None):
    """Configs 2-4 end to end.  models: {class name: Shasta on `device`}; scenes: [(scene name, [tokens])] of the WHOLE split.
    Scenes are sharded over `world` ranks; rank 0 returns (per-class cp dicts, merged dict, tracking dict) and, with work_dir,
    writes <class>/cp_<split>.json, merged_cp_<split>.json and tracking_result.json like the reference CLIs; other ranks
    return None.  forward_override ({class name: callable(batch) -> (m1, m2, example)}) and tracker_on_device=False exist for the
    CPU tests of the sharding / gather / merge logic (gloo, no GPU in the build container); the product path leaves them alone.

    Device path, frame-major: the reference runs one CLI per class over the whole split (eval.py), re-reading every frame's files and
    re-convolving its maps per class.  Here a run of up to batch_pairs consecutive frames of one scene is loaded ONCE for all classes
    (frames.SharedFrames), its maps are produced once per frame
    (_run_features), every class's forward and decode kernel are launched back to back, and ONE event wait per run brings all
    classes' decisions to the host - after the next run has been queued, so the device works while the host decodes and loads.  Same per-class results as the class-major chain (tests/test_pipeline.py)."""
    mine = set(replica.shard_scenes([(n, len(t)) for n, t in scenes], rank, world))
    tokens = [t for n, toks in scenes if n in mine for t in toks]
    all_tokens = [t for _, toks in scenes for t in toks]
    per_class = {}
    timer = _NO_TIMER if timer is None else timer
    names = [n for n in NUSCENES_TRACKING_NAMES if n in models]
    if forward_override is not None:
        for name in names:
            dec = eval_class(models[name], name, paths, tokens, bev, device, known_tokens=all_tokens, batch_pairs=batch_pairs,
                             decode_on_device=decode_on_device, forward=forward_override[name], timer=timer)
            per_class[name] = replica.gather_decoded(dec, dst=0, group=group)
    else:
        with open(paths["frames_meta_path"]) as f:
            meta = json.load(f)["frames"]
        # one rank on a GPU: every scene's tracker run is queued as soon as the scene is decoded, behind the next scenes' launches
        early = []
        hook = None
        if world == 1 and tracker_on_device and device.type == "cuda":
            def hook(decs_):
                early.append(_SceneTrackers(meta, decs_, names, max_age, device))
                return early[0].frames_done
        decs = _frame_major(models, names, paths, scenes, mine, all_tokens, bev, device, batch_pairs, decode_on_device, timer, on_decs=hook)
        for name in names:
            per_class[name] = replica.gather_decoded(decs[name], dst=0, group=group)
    if rank != 0:
        return None
    if world > 1:
        # the gather appended the other ranks' tokens behind rank 0's own: put every class's dictionary back into the split's frame
        # order, so that the files rank 0 writes are byte for byte those of a one-rank run
        for name, cp in per_class.items():
            res = cp["results"]
            cp["results"] = {t: res[t] for t in all_tokens if t in res}
    with timer.stage("merge"):
        merged = merge_results(per_class)
        if forward_override is not None:
            with open(paths["frames_meta_path"]) as f:
                meta = json.load(f)["frames"]
    if tracker_on_device:
        with timer.stage("tracker"):
            tracking = early[0].collect() if (forward_override is None and early) else None
            if tracking is None:
                # whole scenes in one launch (or, beyond the kernel's capacities, frame by frame); `merged` stays as decoded either way
                tracking = run_tracking(merged["results"], meta, max_age=max_age, whole_scenes=True)
    else:
        tracking = None
    if work_dir is not None:
        for name, cp in per_class.items():
            os.makedirs(os.path.join(work_dir, name), exist_ok=True)
            with open(os.path.join(work_dir, name, "cp_%s.json" % split), "w") as f:
                json.dump(cp, f)
        with open(os.path.join(work_dir, "merged_cp_%s.json" % split), "w") as f:
            json.dump(merged, f)
        if tracking is not None:
            with open(os.path.join(work_dir, "tracking_result.json"), "w") as f:
                json.dump(tracking, f)
    return per_class, merged, tracking


def forward_only_seconds(models, paths, scenes, bev, device, batch_pairs=32):
    """Device time of the chain's forward alone at the chain's batching: for every run of the split, shared_conv of all class heads,
    every class's affinity forward and the decode kernel, bracketed by HIP events on the launch stream (the stand-in that produces the
    neck outputs is outside the bracket, the loader runs before it, nothing is decoded or tracked).  The yardstick `run_split`'s
    frames/s is compared with in bench.py's `extra.pipeline`."""
    names = [n for n in NUSCENES_TRACKING_NAMES if n in models]
    all_tokens = [t for _, toks in scenes for t in toks]
    _loader_init(paths["det_path"], paths["cls_info_path"], paths["frame_info_path"], {n: CLASS_CONFIGS[n] for n in names}, all_tokens)
    runs = list(_scene_runs(scenes, {n for n, _ in scenes}, _LOADER["known"], _LOADER["frames"].frame_info, batch_pairs))
    bank_cache, spans = {}, []
    for run in runs:
        batches = _loader_run(run, True)
        dev_in = {n: (torch.from_numpy(b["det_boxes"]).to(device), torch.from_numpy(b["prev_det_boxes"]).to(device)) for n, b in batches.items()}
        toks = [run[0][1] or run[0][0]] + [t for t, _ in run]
        x = bev.neck_batch(toks, device)
        if "bank" not in bank_cache:
            from .shared_conv import SharedConvBank
            bank_cache["bank"] = SharedConvBank([models[n] for n in names])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            feats = dict(zip(names, bank_cache["bank"](x, bound=getattr(bev, "neck_bound", None))))
            for n in names:
                b = batches[n]
                ex = dict(b, det_boxes=dev_in[n][0], prev_det_boxes=dev_in[n][1], bev_feature=feats[n][1:], prev_bev_feature=feats[n][:-1])
                m1, m2, _ = models[n](ex, train_mode=False)
                decode.decode_flags_launch(m1, m2, [len(p) for p in b["prev_cls_det_boxes"]], [len(c) for c in b["cls_det_boxes"]])
        e1.record()
        spans.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in spans) * 1e-3


# ---- loader side of the frame-major chain ---------------------------------------------------------------------------------------------
_LOADER = {}


def _loader_init(det_path, cls_info_path, frame_info_path, max_objects, known):
    _LOADER["frames"] = frames.SharedFrames(det_path, cls_info_path, frame_info_path, max_objects)
    _LOADER["known"] = set(known)
    _LOADER["names"] = list(max_objects)


def _loader_run(run, share_prev=False, staging=None):
    """All classes' samples of one run as plain numpy / list data (frames.collate_pairs' keys).  staging: a 1-D fp32 numpy array (the
    chain's pinned host buffer) that receives every class's (n, N, 11) current and previous box stacks back to back - ONE host-to-device
    copy per run then carries them all; each class's entry `_slot` = (offset of its current stack, of its previous stack) in floats."""
    import numpy as np
    sf, known, out = _LOADER["frames"], _LOADER["known"], {}
    off = 0
    for n in _LOADER["names"]:
        views = None
        if staging is not None:
            sz = len(run) * sf.max_objects[n] * 11
            views = (staging[off:off + sz].reshape(len(run), sf.max_objects[n], 11), staging[off + sz:off + 2 * sz].reshape(len(run), sf.max_objects[n], 11))
        fast = sf.load_run(n, run, share_prev=share_prev, out=views)
        if fast is not None:
            if views is not None:
                fast["_slot"] = (off, off + sz)
                off += 2 * sz
            out[n] = fast
            continue
        samples = [sf.load(n, t, known_tokens=known) for t, _ in run]  # a frame with more than max_obj detections of the class
        out[n] = dict(det_boxes=np.stack([s["det_boxes"] for s in samples]).astype(np.float32),
                      prev_det_boxes=np.stack([s["prev_det_boxes"] for s in samples]).astype(np.float32),
                      num_det_boxes=[s["num_det_boxes"] for s in samples], num_prev_det_boxes=[s["num_prev_det_boxes"] for s in samples],
                      cls_det_boxes=[s["cls_det_boxes"] for s in samples], prev_cls_det_boxes=[s["prev_cls_det_boxes"] for s in samples],
                      metadata=[dict(token=s["token"]) for s in samples],
                      prev_metadata=[dict(token=s["prev_token"] or s["token"]) for s in samples])
    return out


def _frame_major(models, names, paths, scenes, mine, all_tokens, bev, device, batch_pairs, decode_on_device, timer, on_decs=None):
    """The loader runs in line, between the launches of one run and the host half of the previous one (the device is busy with the queued
    run meanwhile).  Alternatives measured on the 20 x 40 split when this form ran at 800 - 867 frames/s (MI355X box, 256 host cores): a loader
    THREAD 526 - 632 (parsing is pure Python: the thread takes the GIL from the launches); multiprocessing pools 248 - 303 (every child
    imports the main module and torch); 2 - 4 child processes that import numpy and json only, parse the files and pipe the parsed frames
    back 617 - 840 (unpickling the class dicts here costs what decoding their json costs; ~0.2 s of start-up); two runs in flight
    instead of one: the event waits vanish, the total does not move."""
    max_obj = {n: CLASS_CONFIGS[n] for n in names}
    init = (paths["det_path"], paths["cls_info_path"], paths["frame_info_path"], max_obj, list(all_tokens))
    _loader_init(*init)
    known = _LOADER["known"]
    decs = {n: decode.AffinityDecoder() for n in names}
    frames_done = on_decs(decs) if on_decs is not None else None  # (the chain's scene-by-scene tracker: told which frames are decoded)
    cuda = device.type == "cuda"
    runs = list(_scene_runs(scenes, mine, known, _LOADER["frames"].frame_info, batch_pairs))
    if not runs:
        return decs
    share = bool(decode_on_device)  # the device-decision decode copies a previous-frame box before it writes to it
    # Box stacks of a run: every class's (n, N, 11) current / previous arrays are written by the loader straight into ONE pinned host
    # buffer and cross to the device in ONE copy (420 pin_memory() calls and as many copies per 800 frames took 0.25 s of host time
    # before).  Three buffers in rotation: a run's copy has completed before its buffer comes round again (finish() of run r waits for
    # run r's event before run r + 2 is loaded).
    per_frame = 2 * 11 * sum(max_obj.values())
    cap = per_frame * max(len(r) for r in runs)
    if cuda:
        stage_host = [torch.empty(cap, dtype=torch.float32, pin_memory=True) for _ in range(3)]
        stage_dev = [torch.empty(cap, dtype=torch.float32, device=device) for _ in range(3)]
    stream_of_batches = ((_loader_run(r, share, stage_host[i % 3].numpy() if cuda else None), i) for i, r in enumerate(runs))

    def tensors(raw_i):
        raw, i = raw_i
        used = 0
        for b in raw.values():
            b["_lags"] = b["prev_det_boxes"][:, 0, 9].copy()
            if "_slot" in b:
                used = max(used, b["_slot"][1] + b["prev_det_boxes"].size)
        if cuda and used:
            stage_dev[i % 3][:used].copy_(stage_host[i % 3][:used], non_blocking=True)
        for b in raw.values():
            slot = b.pop("_slot", None)
            for j, k in enumerate(("det_boxes", "prev_det_boxes")):
                if cuda and slot is not None:
                    b[k] = stage_dev[i % 3][slot[j]:slot[j] + b[k].size].view(b[k].shape)
                else:  # a frame with more detections than max_obj (sampled rows) or the CPU tests: the array as it is
                    t = torch.from_numpy(np.ascontiguousarray(b[k], dtype=np.float32))
                    b[k] = t.to(device, non_blocking=True) if cuda else t
        return raw

    def finish(pending, ev):
        if ev is not None:
            ev.synchronize()  # this run's decision copies have landed (the next run's launches are already queued behind them)
        for n, b, host, m1, m2 in pending:
            if host is not None:
                decs[n].add_batch(None, None, b, flags=decode.decode_flags_unpack(host), lags=b["_lags"], copy_fn=share)
            else:
                decs[n].add_batch(m1, m2, b, on_device=False, lags=b["_lags"])
        if frames_done is not None and pending:
            frames_done([md["token"] for md in pending[0][1]["metadata"]])

    bank_cache = {}
    waiting = []  # runs whose launches are queued and whose decisions have not been read yet
    for run in runs:
        with timer.stage("loader + collate (wait)"):
            batches = tensors(next(stream_of_batches))
        feats = _run_features(bev, models, names, run, device, bank_cache, timer)
        pending = []
        with timer.stage("h2d + forward + decode kernel"):
            for n in names:
                b = batches[n]
                ex = {k: v for k, v in b.items() if k != "_lags"}
                ex["bev_feature"], ex["prev_bev_feature"] = feats[n][1:], feats[n][:-1]
                with torch.no_grad():
                    m1, m2, ex = models[n](ex, train_mode=False)
                if decode_on_device:
                    buf = decode.decode_flags_launch(m1, m2, [len(p) for p in b["prev_cls_det_boxes"]], [len(c) for c in b["cls_det_boxes"]])
                    host = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=cuda)
                    host.copy_(buf, non_blocking=True)
                    pending.append((n, b, host, None, None))
                else:
                    pending.append((n, b, None, m1, m2))
            ev = None
            if cuda:
                ev = torch.cuda.Event()
                ev.record()
        # the host half of the PREVIOUS run is done while the device works on this one: one event wait per run, no stream-wide stall
        with timer.stage("decode (host)"):
            waiting.append((pending, ev))
            if len(waiting) > 1:
                finish(*waiting.pop(0))
    with timer.stage("decode (host)"):
        for w in waiting:
            finish(*w)
    return decs
