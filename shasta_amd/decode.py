"""Decode of the affinity matrices into per-frame detection lists, restating the inline loop of
tools/nusc_shasta/eval.py:112-181 (== validate.py:53-122) as functions, so the consumer side of the hot path is callable
and testable.  One device->host copy per frame (the reference does one .item() per element)."""
import numpy as np
import torch


def decode_frame(matched1, matched2, cls_det_boxes, prev_cls_det_boxes, token, time_lag):
    """matched1 (N, N+2), matched2 (N+2, N) for ONE frame pair (tensor or array).  The box dict lists are mutated exactly
    like the reference does (FN boxes are moved forward by time_lag*velocity and re-tokened, flags added).
    Returns (annos, dead_prev_idx, keep_det_idx)."""
    m1 = matched1.detach().cpu().numpy() if torch.is_tensor(matched1) else np.asarray(matched1)
    m2 = matched2.detach().cpu().numpy() if torch.is_tensor(matched2) else np.asarray(matched2)
    n_prev, n_cur = len(prev_cls_det_boxes), len(cls_det_boxes)
    annos, fn_annos, dead_prev = [], [], []
    if n_prev > 0:
        keep_prev = []
        A = np.concatenate([m1[:n_prev, :n_cur], m1[:n_prev, -2:]], axis=1)
        col_dead, col_fn = A.shape[1] - 2, A.shape[1] - 1
        best = A.argmax(axis=1)
        for n in range(n_prev):
            k = int(best[n])
            val = float(A[n, k])
            if val > 0.5 and k == col_dead:
                dead_prev.append(n)
                continue
            if val > 0.5 and k == col_fn:
                box = prev_cls_det_boxes[n]
                box["translation"][:2] = [t + time_lag * v for t, v in zip(box["translation"][:2], box["velocity"])]
                box["FN"] = True
                box["token"] = token
                box["ref_detection_score"] = 1 - float(A[n, -2])
                fn_annos.append(box)
                continue
            keep_prev.append(n)
        Bm = np.concatenate([m2[keep_prev, :n_cur], m2[-2:, :n_cur]], axis=0)
    else:
        Bm = m2[-2:, :n_cur]
    keep_dets = []
    if n_cur > 0:
        row_fp, row_newborn = Bm.shape[0] - 1, Bm.shape[0] - 2
        best = Bm.argmax(axis=0)
        for k in range(n_cur):
            n = int(best[k])
            val = float(Bm[n, k])
            if val > 0.7 and n == row_fp:
                continue
            if val > 0.5 and n == row_newborn:
                cls_det_boxes[k]["newborn"] = True
            cls_det_boxes[k]["ref_detection_score"] = 1 - float(Bm[-1, k])
            keep_dets.append(k)
            annos.append(cls_det_boxes[k])
    annos.extend(fn_annos)
    return annos, dead_prev, keep_dets


def decode_flags_launch(matched1, matched2, n_prev, n_cur):
    """The decode kernel for a batch, asynchronous: returns ONE device int32 tensor (4, B, N) = [prev_class, prev_score bits, det_flags,
    det_score bits] (a single device->host copy fetches all four decision arrays; decode_flags_unpack splits it)."""
    from . import hip
    lib = hip.load()
    B, N = matched1.shape[0], matched1.shape[1]
    dev = matched1.device
    counts = torch.tensor([list(n_prev), list(n_cur)], dtype=torch.int32)
    counts = (counts.pin_memory() if dev.type == "cuda" else counts).to(dev, non_blocking=True)
    buf = torch.empty(4, B, N, dtype=torch.int32, device=dev)
    m1c, m2c = matched1.contiguous(), matched2.contiguous()  # named: a copy made inside the call would be freed - and its block reused by the next one - before the launch
    hip.check(lib.shasta_decode_flags_f32(hip.ptr(m1c), hip.ptr(m2c), hip.ptr(counts[0]), hip.ptr(counts[1]),
                                          B, N, hip.ptr(buf[0]), hip.ptr(buf[1].view(torch.float32)), hip.ptr(buf[2]),
                                          hip.ptr(buf[3].view(torch.float32)), hip.stream_ptr()), "shasta_decode_flags_f32")
    return buf


def decode_flags_unpack(host_buf):
    """(4, B, N) int32 host tensor of decode_flags_launch -> numpy (prev_class, prev_score, det_flags, det_score)."""
    a = host_buf.numpy()
    return a[0], a[1].view(np.float32), a[2], a[3].view(np.float32)


def decode_flags_device(matched1, matched2, n_prev, n_cur):
    """Batched decisions of the decode loop on the GPU (csrc/decode.hip): returns CPU numpy arrays
    (prev_class (B,N), prev_score (B,N), det_flags (B,N), det_score (B,N)) after ONE device->host copy."""
    return decode_flags_unpack(decode_flags_launch(matched1, matched2, n_prev, n_cur).cpu())


def decode_frame_from_flags(prev_class, prev_score, det_flags, det_score, cls_det_boxes, prev_cls_det_boxes, token, time_lag, copy_fn=False):
    """Same result as decode_frame, built from the device decisions of one frame (rows of decode_flags_device).  copy_fn: a propagated
    (false-negative) box of the previous frame is copied before it is moved and flagged, so prev_cls_det_boxes may hold shared dicts."""
    annos, fn_annos, dead_prev, keep_dets = [], [], [], []
    n_prev, n_cur = len(prev_cls_det_boxes), len(cls_det_boxes)
    # (plain Python numbers: indexing a numpy array element by element costs more than the decisions themselves)
    pc = prev_class[:n_prev].tolist() if hasattr(prev_class, "tolist") else prev_class
    df = det_flags[:n_cur].tolist() if hasattr(det_flags, "tolist") else det_flags
    for n in range(n_prev):
        c = pc[n]
        if c == 1:
            dead_prev.append(n)
        elif c == 2:
            box = prev_cls_det_boxes[n]
            if copy_fn:
                box = dict(box, translation=list(box["translation"]))
            box["translation"][:2] = [t + time_lag * v for t, v in zip(box["translation"][:2], box["velocity"])]
            box["FN"] = True
            box["token"] = token
            box["ref_detection_score"] = 1 - float(prev_score[n])
            fn_annos.append(box)
    if n_cur:
        ds = det_score[:n_cur].tolist() if hasattr(det_score, "tolist") else det_score
        for k in range(n_cur):
            f = df[k]
            if f & 1:
                box = cls_det_boxes[k]
                if f & 2:
                    box["newborn"] = True
                box["ref_detection_score"] = 1 - ds[k]
                keep_dets.append(k)
                annos.append(box)
    annos.extend(fn_annos)
    return annos, dead_prev, keep_dets


class AffinityDecoder:
    """Accumulates frames like eval.py's main loop: results dict token -> annos, plus the dead-track post-pass."""

    def __init__(self):
        self.results = {}
        self.dead_tracker = {}

    def add(self, matched1, matched2, processed_batch, b=0):
        token = processed_batch["metadata"][b]["token"]
        self.dead_tracker.setdefault(token, {"dead_idx": [], "keep_idx": []})
        cls = processed_batch["cls_det_boxes"][b]
        prev_cls = processed_batch["prev_cls_det_boxes"][b]
        time_lag = float(processed_batch["prev_det_boxes"][b, 0, 9]) if len(prev_cls) else 0.0
        annos, dead_prev, keep = decode_frame(matched1[b], matched2[b], cls, prev_cls, token, time_lag)
        if len(prev_cls):
            prev_token = processed_batch["prev_metadata"][b]["token"]
            self.dead_tracker.setdefault(prev_token, {"dead_idx": [], "keep_idx": []})
            self.dead_tracker[prev_token]["dead_idx"].extend(dead_prev)
        if len(cls):
            self.dead_tracker[token]["keep_idx"] = keep
        self.results[token] = annos
        return annos

    def add_batch(self, matched1, matched2, processed_batch, on_device=True, flags=None, lags=None, copy_fn=False):
        """Every frame pair of a batch (any number of frames x classes).  on_device: the per-row / per-column decisions of the
        whole batch come from ONE launch of the decode kernel and ONE device->host copy; otherwise the matrices are copied to the
        host once and the restated reference loop runs per frame.  flags: the decisions already on the host (decode_flags_unpack of
        a copy the caller made for several batches at once), lags: prev_det_boxes[:, 0, 9] from the host side of the batch - with
        both given this call touches no device memory."""
        cls_all, prev_all = processed_batch["cls_det_boxes"], processed_batch["prev_cls_det_boxes"]
        B = len(cls_all)
        if lags is None:
            lags = processed_batch["prev_det_boxes"][:, 0, 9].detach().float().cpu().numpy()
        if flags is not None:
            pc, ps, df, ds = flags
            on_device = True
        elif on_device:
            pc, ps, df, ds = decode_flags_device(matched1, matched2, [len(p) for p in prev_all], [len(c) for c in cls_all])
        else:
            m1h, m2h = matched1.detach().cpu().numpy(), matched2.detach().cpu().numpy()
        for b in range(B):
            token = processed_batch["metadata"][b]["token"]
            self.dead_tracker.setdefault(token, {"dead_idx": [], "keep_idx": []})
            cls, prev_cls = cls_all[b], prev_all[b]
            time_lag = float(lags[b]) if len(prev_cls) else 0.0
            if on_device:
                annos, dead_prev, keep = decode_frame_from_flags(pc[b], ps[b], df[b], ds[b], cls, prev_cls, token, time_lag, copy_fn=copy_fn)
            else:
                annos, dead_prev, keep = decode_frame(m1h[b], m2h[b], cls, prev_cls, token, time_lag)
            if len(prev_cls):
                prev_token = processed_batch["prev_metadata"][b]["token"]
                self.dead_tracker.setdefault(prev_token, {"dead_idx": [], "keep_idx": []})
                self.dead_tracker[prev_token]["dead_idx"].extend(dead_prev)
            if len(cls):
                self.dead_tracker[token]["keep_idx"] = keep
            self.results[token] = annos

    def finalize_tokens(self, tokens):
        """The `dead` post-pass of finalize() for some frames only (all decodes that can mark them - their own and the following
        frame's - must be in); idempotent, finalize() may run over them again."""
        for token in tokens:
            annos, info = self.results.get(token), self.dead_tracker.get(token)
            if annos is None or info is None:
                continue
            for i in info["dead_idx"]:
                if i in info["keep_idx"]:
                    annos[info["keep_idx"].index(i)]["dead"] = True

    def finalize(self):
        for token, annos in self.results.items():
            info = self.dead_tracker[token]
            for i in info["dead_idx"]:
                if i in info["keep_idx"]:
                    annos[info["keep_idx"].index(i)]["dead"] = True
        return {"results": self.results,
                "meta": {"use_camera": False, "use_lidar": True, "use_radar": False, "use_map": False,
                         "use_external": False}}
