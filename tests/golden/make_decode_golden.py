"""Generates tests/golden/decode_golden.json.gz: what the REFERENCE's decode loop (the body of `main()` in
tools/nusc_shasta/eval.py:111-181: the per-frame loop under torch.no_grad() and the dead-track post-pass) produces for seeded
synthetic affinity matrices and detection lists.  The loop is inline script code, so it is run IN PLACE: the two statements
are located in the parsed source with `ast`, compiled from the reference file itself and executed with a synthetic
`data_loader` / `track_batch_processor`; no reference text is copied.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_decode_golden.py
"""
import ast
import copy
import gzip
import json
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SHASTA_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "tools", "nusc_shasta", "eval.py")


def reference_loop_code():
    tree = ast.parse(open(SRC).read(), SRC)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    with_idx = next(i for i, n in enumerate(main.body) if isinstance(n, ast.With) and any(isinstance(x, ast.For) for x in n.body))
    post = next(n for n in main.body[with_idx + 1:] if isinstance(n, ast.For))  # the dead-track post-pass
    mod = ast.Module(body=[main.body[with_idx], post], type_ignores=[])
    return compile(mod, SRC, "exec")


def synth_frames(rng, n_frames, N):
    frames = []
    prev_token, prev_cls = None, []
    for f in range(n_frames):
        token = "tok%03d" % f
        n_det = int(rng.integers(0, N + 1)) if f != 2 else 0
        cls = [dict(sample_token=token, translation=[float(v) for v in rng.uniform(-50, 50, 3)], velocity=[float(v) for v in rng.normal(size=2)],
                    detection_name="car", detection_score=float(rng.uniform()), uid="%d_%d" % (f, k)) for k in range(n_det)]
        # peaked matrices: most rows / columns have a clear winner among [matches | dead / newborn | FN / FP]
        l1, l2 = rng.normal(0, 3, (1, N, N + 2)), rng.normal(0, 3, (1, N + 2, N))
        for r in range(N):  # make dead tracks / false negatives / newborns / false positives frequent
            u = rng.uniform()
            if u < 0.2:
                l1[0, r, -2] += 9
            elif u < 0.4:
                l1[0, r, -1] += 9
            u = rng.uniform()
            if u < 0.2:
                l2[0, -2, r] += 9
            elif u < 0.35:
                l2[0, -1, r] += 9
        m1 = torch.softmax(torch.from_numpy(l1).float(), dim=2)
        m2 = torch.softmax(torch.from_numpy(l2).float(), dim=1)
        prev_boxes = torch.zeros(1, N, 11)
        prev_boxes[0, 0, 9] = float(rng.uniform(0.4, 0.6))
        frames.append(dict(token=token, prev_token=prev_token, cls=cls, prev_cls=copy.deepcopy(prev_cls) if prev_token else [], m1=m1, m2=m2,
                           prev_boxes=prev_boxes))
        prev_token, prev_cls = token, cls
    return frames


def main():
    code = reference_loop_code()
    rng = np.random.default_rng(5)
    N = 12
    frames = synth_frames(rng, 10, N)
    batches = [dict(metadata=[{"token": fr["token"]}], prev_metadata=[{"token": fr["prev_token"]}], cls_det_boxes=[copy.deepcopy(fr["cls"])],
                    prev_cls_det_boxes=[copy.deepcopy(fr["prev_cls"])], prev_det_boxes=fr["prev_boxes"], _m=(fr["m1"], fr["m2"])) for fr in frames]
    env = dict(torch=torch, model=None, cfg=types.SimpleNamespace(local_rank=0), data_loader=batches, nusc_annos={"results": {}, "meta": None},
               dead_tracker={}, track_batch_processor=lambda model, b, train_mode=False, local_rank=0: (b["_m"][0], b["_m"][1], b))
    exec(code, env)
    out = dict(N=N, frames=[dict(token=fr["token"], prev_token=fr["prev_token"], cls=fr["cls"], prev_cls=fr["prev_cls"], m1=fr["m1"].tolist(),
                                 m2=fr["m2"].tolist(), time_lag=float(fr["prev_boxes"][0, 0, 9])) for fr in frames],
               results=env["nusc_annos"]["results"], dead_tracker=env["dead_tracker"])
    path = os.path.join(HERE, "decode_golden.json.gz")
    with gzip.open(path, "wt", compresslevel=9) as f:
        json.dump(out, f)
    kinds = dict(fn=sum(1 for a in sum(out["results"].values(), []) if a.get("FN")), newborn=sum(1 for a in sum(out["results"].values(), []) if a.get("newborn")),
                 dead=sum(1 for a in sum(out["results"].values(), []) if a.get("dead")), total=sum(len(v) for v in out["results"].values()))
    print("wrote", path, os.path.getsize(path), "bytes", kinds)


if __name__ == "__main__":
    main()
