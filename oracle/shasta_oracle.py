"""CPU oracle for the ShaSTA affinity hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This is a from-scratch CPU restatement (torch-CPU fp32 / numpy) of the reference algorithm
for the path named in BASELINE.json.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; `shasta_amd/` never does.

Parity pin: the reference (tsadja/ShaSTA) ships NO tests, golden vectors or fixtures for
this path.  The oracle is therefore pinned against outputs of the
reference itself, generated in the build container by importing the reference forward
(`tests/golden/make_golden.py`, vectors committed under `tests/golden/*.npz`) and checked
by `tests/test_oracle_golden.py`.

Each function cites the reference file:line it restates (paths relative to the reference
repository root).  Weights are passed as a plain dict with the reference's state_dict key
names (`aug_shape.0.0.weight`, `fuse_shape.2.bias`, ...).
"""
import math

import numpy as np
import torch
import torch.nn.functional as TF

EPS_LOG = 1e-10  # det3d/models/tracker/shasta.py:277


# --------------------------------------------------------------------------------------
# row 2: VoxelFeatureExtractorV3.forward  (det3d/models/readers/voxel_encoder.py:18-28)
# --------------------------------------------------------------------------------------
def voxel_mean(voxels, num_points):
    """(V, max_points, C) zero padded, (V,) -> (V, C): sum over the point slots / count."""
    voxels = torch.as_tensor(voxels, dtype=torch.float32)
    cnt = torch.as_tensor(num_points).to(torch.float32).view(-1, 1)
    return (voxels.sum(dim=1) / cnt).contiguous()


# --------------------------------------------------------------------------------------
# row 1: points_to_voxel (det3d/ops/point_cloud/point_cloud_ops.py:112-184, kernel :7-55)
# pure-numpy restatement for small clouds; oracle/voxelize_oracle.c is the fast C twin.
# --------------------------------------------------------------------------------------
def points_to_voxel_np(points, voxel_size, coors_range, max_points, max_voxels):
    points = np.ascontiguousarray(points, dtype=np.float32)
    vs = np.asarray(voxel_size, dtype=np.float32)
    rng = np.asarray(coors_range, dtype=np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int32)  # x, y, z cells
    n, ndim = points.shape
    voxels = np.zeros((max_voxels, max_points, ndim), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    counts = np.zeros((max_voxels,), np.int32)
    table = {}
    nvox = 0
    # cell index per axis, fp32 arithmetic exactly as the reference: floor((p - lo) / vs)
    cell = np.floor((points[:, :3] - rng[:3]) / vs)
    ok = np.all((cell >= 0) & (cell < grid.astype(np.float32)), axis=1)
    cell = cell.astype(np.int64)
    for i in range(n):
        if not ok[i]:
            continue
        key = (int(cell[i, 2]), int(cell[i, 1]), int(cell[i, 0]))  # z, y, x
        vid = table.get(key, -1)
        if vid == -1:
            if nvox >= max_voxels:
                continue
            vid = nvox
            nvox += 1
            table[key] = vid
            coors[vid] = key
        c = counts[vid]
        if c < max_points:
            voxels[vid, c] = points[i]
            counts[vid] = c + 1
    return voxels[:nvox], coors[:nvox], counts[:nvox]


# --------------------------------------------------------------------------------------
# row 4: Shasta.get_box_center (shasta.py:121-161) + center_to_corner_box2d
# (det3d/core/bbox/box_torch_ops.py:184-203, corners_nd :24-59, rotation_2d :145-158)
# --------------------------------------------------------------------------------------
def box_points(boxes7, num_point):
    """boxes7 (N,7) [x,y,z,w,l,h,yaw] -> (num_point*N, 3), point-type-major."""
    if num_point == 1:
        return boxes7[:, :3]
    cx, cy, z = boxes7[:, 0:1], boxes7[:, 1:2], boxes7[:, 2:3]
    w, l, yaw = boxes7[:, 3], boxes7[:, 4], boxes7[:, 6]
    # unit-square corners, clockwise from the minimum point, origin 0.5
    ux = torch.tensor([-0.5, -0.5, 0.5, 0.5], dtype=boxes7.dtype)
    uy = torch.tensor([-0.5, 0.5, 0.5, -0.5], dtype=boxes7.dtype)
    px = w[:, None] * ux[None, :]
    py = l[:, None] * uy[None, :]
    s, c = torch.sin(yaw)[:, None], torch.cos(yaw)[:, None]
    rx = px * c + py * s + cx
    ry = -px * s + py * c + cy
    corners = torch.stack([rx, ry], dim=-1)  # (N,4,2)

    def mid(a, b):
        return torch.cat([(corners[:, a] + corners[:, b]) / 2, z], dim=-1)

    pts = [mid(0, 1), mid(2, 3), mid(0, 3), mid(1, 2)]  # front, back, left, right
    if num_point == 5:
        pts = [boxes7[:, :3]] + pts
    elif num_point != 4:
        raise ValueError("num_point must be 1, 4 or 5")
    return torch.cat(pts, dim=0)


# --------------------------------------------------------------------------------------
# row 5: BEVFeatureExtractor.forward (det3d/models/second_stage/bird_eye_view.py:18-41)
#        bilinear_interpolate_torch (det3d/core/utils/center_utils.py:92-121)
# --------------------------------------------------------------------------------------
def bilinear_nhwc(im, x, y):
    """im (H,W,C); indices clamped to the map, weights taken from the CLAMPED indices."""
    H, W = im.shape[0], im.shape[1]
    x0 = torch.floor(x).long()
    y0 = torch.floor(y).long()
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
    y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
    wa = (x1.to(x.dtype) - x) * (y1.to(y.dtype) - y)
    wb = (x1.to(x.dtype) - x) * (y - y0.to(y.dtype))
    wc = (x - x0.to(x.dtype)) * (y1.to(y.dtype) - y)
    wd = (x - x0.to(x.dtype)) * (y - y0.to(y.dtype))
    return (im[y0, x0] * wa[:, None] + im[y1, x0] * wb[:, None]
            + im[y0, x1] * wc[:, None] + im[y1, x1] * wd[:, None])


def bev_gather(bev_nhwc, boxes7, num_point, pc_start=(-54.0, -54.0), voxel_size=(0.075, 0.075),
               out_stride=8):
    """bev_nhwc (B,H,W,C), boxes7 (B,N,7) -> (B,N,num_point*C)."""
    out = []
    for b in range(bev_nhwc.shape[0]):
        pts = box_points(boxes7[b], num_point)
        xs = (pts[:, 0] - pc_start[0]) / voxel_size[0] / out_stride
        ys = (pts[:, 1] - pc_start[1]) / voxel_size[1] / out_stride
        f = bilinear_nhwc(bev_nhwc[b], xs, ys)  # (np*N, C)
        n = f.shape[0] // num_point
        out.append(torch.cat([f[i * n:(i + 1) * n] for i in range(num_point)], dim=1))
    return torch.stack(out)


# --------------------------------------------------------------------------------------
# generic Sequential(Linear, ReLU, Linear, ...) evaluation from the weight dict
# --------------------------------------------------------------------------------------
def _mlp(w, prefix, x):
    idx = sorted({int(k[len(prefix) + 1:].split(".")[0]) for k in w if k.startswith(prefix + ".")
                  and k.endswith(".weight")})
    for j, i in enumerate(idx):
        x = TF.linear(x, w[f"{prefix}.{i}.weight"], w[f"{prefix}.{i}.bias"])
        if j + 1 < len(idx):
            x = torch.relu(x)
    return x


# row 3: shared_conv (shasta.py:42-47 applied :223-228): conv3x3+BN(eval)+ReLU -> NHWC
def shared_conv_nhwc(w, bev_nchw, bn_eps=1e-5, batch_stats=False):
    """batch_stats=True: BatchNorm as in train() mode (statistics of this batch; the running buffers are not updated here)."""
    y = TF.conv2d(bev_nchw, w["shared_conv.0.weight"], w["shared_conv.0.bias"], padding=1)
    if batch_stats:
        y = TF.batch_norm(y, None, None, w["shared_conv.1.weight"], w["shared_conv.1.bias"], True, 0.0, bn_eps)
    else:
        y = TF.batch_norm(y, w["shared_conv.1.running_mean"], w["shared_conv.1.running_var"],
                          w["shared_conv.1.weight"], w["shared_conv.1.bias"], False, 0.0, bn_eps)
    return torch.relu(y).permute(0, 2, 3, 1).contiguous()


# rows 6+8: anchor MLPs (shasta.py:241-244, :260-267)
def anchor_shape(w, i, table):
    """table (B,N,F) -> (B,1,F): abs(MLP(flattened table))."""
    B = table.shape[0]
    return torch.abs(_mlp(w, f"aug_shape.{i}", table.reshape(B, -1))).reshape(B, 1, -1)


def anchor_box(w, i, boxes7):
    """boxes7 (B,N,7) -> (B,1,7) with dims [3:6] abs'd."""
    B = boxes7.shape[0]
    # (.clone(): at max_obj = 1 the reshape is a VIEW of the box table, which the back-projection then writes in place - autograd
    # refuses the saved input; for every other size the reshape copies anyway)
    a = _mlp(w, f"aug_dets.{i}", boxes7.reshape(B, -1).clone()).reshape(B, 1, 7)
    return torch.cat([a[:, :, :3], torch.abs(a[:, :, 3:6]), a[:, :, 6:]], dim=-1)


# row 11: hand-designed residuals (shasta.py:277-283)
def hand_residual(prev7, det7, nf):
    """prev7 (B,T,7), det7 (B,D,7) -> (B,T,D)."""
    d2 = ((prev7[:, :, None, :nf] - det7[:, None, :, :nf]) ** 2).sum(-1)
    r = TF.normalize(d2)  # L2 over dim=1 (tracks), eps 1e-12
    r = r + torch.abs(torch.log(prev7[:, :, None, 3:6] + EPS_LOG)
                      - torch.log(det7[:, None, :, 3:6] + EPS_LOG)).sum(-1)
    yp, yd = prev7[:, :, None, 6], det7[:, None, :, 6]
    r = r + torch.sqrt((torch.cos(yp) - torch.cos(yd)) ** 2 + (torch.sin(yp) - torch.sin(yd)) ** 2)
    return r


# rows 12-15: pair MLPs in the reference's dense formulation (shasta.py:286-319), chunked over T
def pair_residual(w, prev_feat, feat, prev7, det7, nf, chunk=64):
    """prev_feat (B,T,F), feat (B,D,F), prev7 (B,T,7), det7 (B,D,7) (post back-projection,
    anchors appended) -> residual (B,T,D)."""
    B, T, F = prev_feat.shape
    D = feat.shape[1]
    dist = hand_residual(prev7, det7, nf)
    out = torch.empty(B, T, D, dtype=prev_feat.dtype)
    fe = feat[:, None].expand(B, 1, D, F)
    db = det7[:, None, :, :nf]
    for t0 in range(0, T, chunk):
        t1 = min(T, t0 + chunk)
        n = t1 - t0
        pf = prev_feat[:, t0:t1, None].expand(B, n, D, F)
        pb = prev7[:, t0:t1, None, :nf].expand(B, n, D, nf)
        cf = fe.expand(B, n, D, F)
        cb = db.expand(B, n, D, nf)
        shape = _mlp(w, "fuse_shape", torch.cat([pf, cf], dim=3))[..., 0]
        fused = _mlp(w, "fuse_det", torch.cat([pb, cb], dim=3))[..., 0]
        coeff = _mlp(w, "res_coeff", torch.cat([pf, pb, cf, cb], dim=3))
        out[:, t0:t1] = (coeff[..., 0] * fused + coeff[..., 1] * dist[:, t0:t1]
                         + coeff[..., 2] * shape)
    return out


# row 16: aff row-MLP + the two softmaxes (shasta.py:323-325)
def affinity(w, residual):
    matched = _mlp(w, "aff", residual)
    m1 = torch.softmax(matched[:, :-2, :], dim=2)
    m2 = torch.softmax(matched[:, :, :-2], dim=1)
    return matched, m1, m2


# --------------------------------------------------------------------------------------
# Shasta.forward after shared_conv (shasta.py:213-327), rows 4-16.
# --------------------------------------------------------------------------------------
def forward_from_bev(w, bev_nhwc, prev_bev_nhwc, det_boxes, prev_det_boxes, num_feats, num_point,
                     pc_start=(-54.0, -54.0), voxel_size=(0.075, 0.075), out_stride=8,
                     return_intermediates=False, grad=False):
    """det_boxes / prev_det_boxes: (B,N,11) fp32.  det_boxes[:,:,:2] is back-projected IN PLACE,
    like the reference does to example["det_boxes"] (shasta.py:216,270).
    Returns (matched1 (B,N,N+2), matched2 (B,N+2,N)[, intermediates]).  grad=True keeps the autograd graph
    (tests of the training path differentiate this restatement with torch autograd, as the reference's train.py does
    with the original)."""
    with (torch.enable_grad() if grad else torch.no_grad()):
        prev7 = prev_det_boxes[:, :, :7]
        det7 = det_boxes[:, :, :7]  # a view: the in-place update below reaches the caller
        vel = det_boxes[:, :, 7:9]
        dt = det_boxes[:, :, 9].unsqueeze(-1)
        feat = bev_gather(bev_nhwc, det7, num_point, pc_start, voxel_size, out_stride)
        pfeat = bev_gather(prev_bev_nhwc, prev7, num_point, pc_start, voxel_size, out_stride)
        newborn_g, fp_g = anchor_shape(w, 0, feat), anchor_shape(w, 1, feat)
        dead_g, fn_g = anchor_shape(w, 2, pfeat), anchor_shape(w, 3, pfeat)
        feat_a = torch.cat([feat, dead_g, fn_g], dim=1)
        pfeat_a = torch.cat([pfeat, newborn_g, fp_g], dim=1)
        newborn, fp = anchor_box(w, 0, det7), anchor_box(w, 1, det7)
        dead, fn = anchor_box(w, 2, prev7), anchor_box(w, 3, prev7)
        det7[:, :, :2] = det7[:, :, :2] - vel * dt
        prev_a = torch.cat([prev7, newborn, fp], dim=1)
        det_a = torch.cat([det7, dead, fn], dim=1)
        residual = pair_residual(w, pfeat_a, feat_a, prev_a, det_a, num_feats)
        matched, m1, m2 = affinity(w, residual)
    if return_intermediates:
        return m1, m2, dict(feature=feat, prev_feature=pfeat, newborn_geom=newborn_g, fp_geom=fp_g,
                            dead_trk_geom=dead_g, fn_geom=fn_g, newborn=newborn, fp=fp,
                            dead_trk=dead, fn=fn, residual=residual, matched=matched)
    return m1, m2


def forward(w, bev_nchw, prev_bev_nchw, det_boxes, prev_det_boxes, num_feats, num_point, **kw):
    """Full Shasta.forward after extract_feat: shared_conv + rows 4-16."""
    with torch.no_grad():
        a = shared_conv_nhwc(w, bev_nchw)
        b = shared_conv_nhwc(w, prev_bev_nchw)
    return forward_from_bev(w, a, b, det_boxes, prev_det_boxes, num_feats, num_point, **kw)


# --------------------------------------------------------------------------------------
# row 19: training loss (tools/nusc_shasta/train.py:200-211)
# --------------------------------------------------------------------------------------
def affinity_loss(m1, m2, gt):
    gt1, gt2 = gt[:, :-2, :], gt[:, :, :-2]
    lf, lb = (gt1 * (-torch.log(m1 + 1e-10))).sum(), (gt2 * (-torch.log(m2 + 1e-10))).sum()
    if gt1.sum() > 0:  # (train.py:208-209: a direction without a ground-truth entry keeps its plain - zero - sum)
        lf = lf / gt1.sum()
    if gt2.sum() > 0:
        lb = lb / gt2.sum()
    return (lf + lb) / 2


# --------------------------------------------------------------------------------------
# row 18: decode loop (tools/nusc_shasta/eval.py:112-181), one frame
# --------------------------------------------------------------------------------------
def decode_frame(m1, m2, cls_det_boxes, prev_cls_det_boxes, token, time_lag):
    """m1 (N,N+2), m2 (N+2,N) for ONE frame; *_cls_det_boxes: lists of nuScenes-style dicts
    (mutated like the reference).  Returns (annos, dead_prev_idx, keep_dets)."""
    m1 = np.asarray(m1)
    m2 = np.asarray(m2)
    n_prev, n_cur = len(prev_cls_det_boxes), len(cls_det_boxes)
    annos, fn_annos, dead_prev = [], [], []
    if n_prev > 0:
        keep_prev = []
        A = np.concatenate([m1[:n_prev, :n_cur], m1[:n_prev, -2:]], axis=1)
        for n in range(n_prev):
            k = int(np.argmax(A[n]))
            val = float(A[n, k])
            if val > 0.5 and k == A.shape[1] - 2:  # dead track
                dead_prev.append(n)
                continue
            if val > 0.5 and k == A.shape[1] - 1:  # false negative: propagate forward
                box = prev_cls_det_boxes[n]
                box["translation"][:2] = [t + time_lag * v for t, v in
                                          zip(box["translation"][:2], box["velocity"])]
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
        for k in range(n_cur):
            n = int(np.argmax(Bm[:, k]))
            val = float(Bm[n, k])
            if val > 0.7 and n == Bm.shape[0] - 1:  # false positive
                continue
            if val > 0.5 and n == Bm.shape[0] - 2:
                cls_det_boxes[k]["newborn"] = True
            cls_det_boxes[k]["ref_detection_score"] = 1 - float(Bm[-1, k])
            keep_dets.append(k)
            annos.append(cls_det_boxes[k])
    annos.extend(fn_annos)
    return annos, dead_prev, keep_dets


def mark_dead(results, dead_tracker):
    """post-pass of eval.py:175-181: flag `dead` on the kept annos of frame t-1."""
    for token, annos in results.items():
        info = dead_tracker.get(token, {"dead_idx": [], "keep_idx": []})
        for i in info["dead_idx"]:
            if i in info["keep_idx"]:
                annos[info["keep_idx"].index(i)]["dead"] = True
    return results


# --------------------------------------------------------------------------------------
# synthetic inputs of SURVEY.md 8(d) (shared by tests, smoke and bench so seeds agree)
# --------------------------------------------------------------------------------------
def synth_boxes(gen, B, N, n_real=None):
    b = torch.zeros(B, N, 11)
    b[..., 0:2] = torch.rand(B, N, 2, generator=gen) * 100 - 50
    b[..., 2] = torch.randn(B, N, generator=gen)
    b[..., 3:6] = torch.rand(B, N, 3, generator=gen) * 4 + 0.5
    b[..., 6] = (torch.rand(B, N, generator=gen) * 2 - 1) * math.pi
    b[..., 7:9] = torch.randn(B, N, 2, generator=gen)
    b[..., 9] = 0.5
    b[..., 10] = torch.rand(B, N, generator=gen)
    if n_real is not None:
        b[:, n_real:] = 0
    return b


def synth_case(B, N, n_real, cin, H, W, seed):
    """Seeded synthetic inputs shared by make_golden.py and the tests: neck-output BEV maps
    (B,cin,H,W) for frames t and t-1, det_boxes and prev_det_boxes (B,N,11)."""
    gen = torch.Generator().manual_seed(1000 + seed)
    bev = torch.relu(torch.randn(B, cin, H, W, generator=gen))
    pbev = torch.relu(torch.randn(B, cin, H, W, generator=gen))
    det = synth_boxes(gen, B, N, n_real)
    prev = synth_boxes(gen, B, N, n_real)
    return bev, pbev, det, prev
