"""`mot_3d.association` call surface (mot_3d/association.py:9-120) with a learned-affinity mode added.

`associate_dets_to_tracks(dets, tracks, mode, asso, dist_threshold, trk_innovation_matrix)` keeps the reference's
argument meaning and return triple `(matches: list of np.ndarray (det_idx, trk_idx), unmatched_dets, unmatched_tracks)`
so SimpleTrack-style bookkeeping (mot_3d/mot.py:149-150,208-209) consumes the result unchanged.  The distance matrix is
`[len(dets), len(tracks)]` (detections are rows: transposed w.r.t. ShaSTA's matched1).

  asso = 'affinity' : dist = 1 - affinity[:n_trk, :n_det].T with `affinity` = matched1 of the Shasta forward for this
                      frame pair (pass it as `affinity=`); this is how the learned matrix plugs into the reference
                      matchers.
  asso = 'euler' / 'm_dis' : L2 / Mahalanobis on the 7-vector [x,y,z,o,l,w,h] with the reference's yaw folding
                      (mot_3d/utils/geometry.py:246-271).
  asso = 'iou' / 'giou' : 1 - rotated 3-D IoU / GIoU for all pairs at once on the GPU (csrc/iou3d.hip, float64).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment


def _array7(b):
    if hasattr(b, "x"):
        return np.array([b.x, b.y, b.z, b.o, b.l, b.w, b.h], dtype=np.float64)
    return np.asarray(b, dtype=np.float64)[:7]


def compute_m_distance(dets, tracks, trk_innovation_matrix):
    """mot_3d/association.py:87-105 + utils/geometry.py:258-271."""
    D = np.stack([_array7(d) for d in dets]) if len(dets) else np.zeros((0, 7))
    T = np.stack([_array7(t) for t in tracks]) if len(tracks) else np.zeros((0, 7))
    diff = D[:, None, :] - T[None, :, :]
    yaw = diff[..., 3]
    yaw = np.where(yaw > np.pi / 2, yaw - np.pi, yaw)
    yaw = np.where(yaw < -np.pi / 2, yaw + np.pi, yaw)
    diff[..., 3] = yaw
    if trk_innovation_matrix is None:
        return np.sqrt((diff * diff).sum(-1))
    inv = np.stack([np.linalg.inv(m) for m in trk_innovation_matrix])  # (T,7,7)
    return np.sqrt(np.einsum("dti,tij,dtj->dt", diff, inv, diff))


def compute_affinity_distance(dets, tracks, affinity):
    a = np.asarray(affinity, dtype=np.float64)
    return 1.0 - a[:len(tracks), :len(dets)].T


def compute_iou_distance(dets, tracks, asso="iou"):
    """mot_3d/association.py:108-120 for all pairs at once on the GPU (csrc/iou3d.hip, float64): 1 - iou3d / 1 - giou3d.
    There is no CPU path: without a device this raises."""
    import torch

    from . import hip
    lib = hip.load()
    nd, nt = len(dets), len(tracks)
    if nd == 0 or nt == 0:
        return np.zeros((nd, nt))
    if not torch.cuda.is_available():
        raise hip.ShastaHipError("asso=%r needs a GPU (rotated IoU kernel); there is no CPU fallback" % asso)
    dev = torch.device("cuda", torch.cuda.current_device())
    D = torch.from_numpy(np.stack([_array7(d) for d in dets])).to(dev)
    T = torch.from_numpy(np.stack([_array7(t) for t in tracks])).to(dev)
    out = torch.empty(nd, nt, dtype=torch.float64, device=dev)
    hip.check(lib.shasta_iou3d_distance_f64(hip.ptr(D), nd, hip.ptr(T), nt, 7, 1 if asso == "giou" else 0, hip.ptr(out),
                                            hip.stream_ptr()), "shasta_iou3d_distance_f64")
    return out.cpu().numpy()


def _dist_matrix(dets, tracks, asso, trk_innovation_matrix, affinity):
    if asso == "affinity":
        if affinity is None:
            raise ValueError("asso='affinity' needs the matched1 matrix of this frame pair (affinity=...)")
        return compute_affinity_distance(dets, tracks, affinity)
    if asso == "m_dis":
        return compute_m_distance(dets, tracks, trk_innovation_matrix)
    if asso == "euler":
        return compute_m_distance(dets, tracks, None)
    if asso in ("iou", "giou"):
        return compute_iou_distance(dets, tracks, asso)
    raise ValueError("unknown asso %r" % (asso,))


def bipartite_matcher(dets, tracks, asso, dist_threshold, trk_innovation_matrix, affinity=None):
    dist = _dist_matrix(dets, tracks, asso, trk_innovation_matrix, affinity)
    r, c = linear_sum_assignment(dist)
    return np.stack([r, c], axis=1), dist


def greedy_matcher(dets, tracks, asso, dist_threshold, trk_innovation_matrix, affinity=None):
    """Global argsort of the flattened matrix, first come first served (mot_3d/association.py:52-84)."""
    dist = _dist_matrix(dets, tracks, asso, trk_innovation_matrix, affinity)
    nd, nt = dist.shape
    det_of_trk, trk_of_det, matched = [-1] * nt, [-1] * nd, []
    for idx in np.argsort(dist.reshape(-1)):
        d, t = int(idx // nt), int(idx % nt)
        if det_of_trk[t] == -1 and trk_of_det[d] == -1:
            det_of_trk[t], trk_of_det[d] = d, t
            matched.append([d, t])
    matched = np.asarray(matched) if matched else np.empty((0, 2))
    return matched, dist


def associate_dets_to_tracks(dets, tracks, mode, asso, dist_threshold=0.9, trk_innovation_matrix=None, affinity=None):
    if mode == "bipartite":
        matched, dist = bipartite_matcher(dets, tracks, asso, dist_threshold, trk_innovation_matrix, affinity)
    elif mode == "greedy":
        matched, dist = greedy_matcher(dets, tracks, asso, dist_threshold, trk_innovation_matrix, affinity)
    else:
        raise ValueError("unknown mode %r" % (mode,))
    unmatched_dets = [d for d in range(len(dets)) if d not in matched[:, 0]]
    unmatched_tracks = [t for t in range(len(tracks)) if t not in matched[:, 1]]
    matches = []
    for m in matched:
        if dist[int(m[0]), int(m[1])] > dist_threshold:
            unmatched_dets.append(m[0])
            unmatched_tracks.append(m[1])
        else:
            matches.append(m.reshape(2))
    return matches, np.array(unmatched_dets), np.array(unmatched_tracks)
