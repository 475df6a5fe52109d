"""Shared test helpers: golden fixtures, seeded model construction, weight checksums."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FORWARD_CASES = ["tiny_4_7_5", "tiny_6_3_1", "small_32_7_4", "small_32_3_5_pad", "car_90_3_5", "truck_60_3_5", "bicycle_50_3_5",
                 "bus_20_3_5", "sharp_90_3_5", "sharp_90_3_5_pad", "headline_500_7_4", "sharp_500_7_4", "classes_500_3_5_pad",
                 "mod_90_3_5", "mod_500_7_4", "heavy_90_3_5", "heavy_500_7_4"]
BIG_CASES = ("headline_500_7_4", "sharp_500_7_4", "classes_500_3_5_pad", "mod_500_7_4", "heavy_500_7_4")  # 1.03 / 1.6 G parameters: 4 - 6.4 GB and ~10 s to build on the CPU
PROBE_IDX = [0, 1, 249, 250, 499, 500, 501]  # rows / columns of the (502, 502) tables stored in full for the N=500 goldens


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = z["cfg"].tolist()
    keys = ["max_obj", "nf", "np", "B", "n_real", "cin", "hw", "stride", "seed"]
    c = dict(zip(keys, cfg))
    c["n_real"] = None if c["n_real"] < 0 else c["n_real"]
    c["sharp"] = tuple(float(v) for v in z["sharp"]) if "sharp" in z.files else None
    c["heavy"] = (int(z["heavy"][0]), int(z["heavy"][1])) if "heavy" in z.files else None
    with open(os.path.join(GOLDEN, name + ".weights.json")) as f:
        sums = json.load(f)
    return z, c, sums


def model_cfg(c):
    return dict(type="Shasta", reader=None, backbone=None, neck=None,
                bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075],
                                   out_stride=c["stride"]),
                max_obj=c["max_obj"], num_feats=c["nf"], num_point=c["np"], in_channels=c["cin"])


def sharpen_state_dict(sd, aff_gain, pair_gain):
    """The "sharpened" seeded weight set of SURVEY.md section 7 / 8(d): default-init outputs are almost flat (top-1 ~ uniform),
    so the weight matrices (not the biases) of the six `aff` layers are scaled by `aff_gain` and those of the three pair MLPs by
    `pair_gain`.  The outputs become peaked (median top-1 probability 1.0, smallest top-2 margin > 1e-2), which is what a
    trained network looks like and what makes an arg-max comparison meaningful.  Applied IN PLACE to a state_dict (reference
    model in make_golden.py, shasta_amd model in the tests)."""
    with torch.no_grad():
        for k, v in sd.items():
            if not k.endswith(".weight"):
                continue
            head = k.split(".")[0]
            if head == "aff":
                v.mul_(aff_gain)
            elif head in ("fuse_shape", "fuse_det", "res_coeff"):
                v.mul_(pair_gain)
    return sd


HEAVY_KEYS = ("aug_shape.%d.0.weight", "fuse_shape.0.weight", "fuse_shape.2.weight", "res_coeff.0.weight", "res_coeff.2.weight",
              "fuse_det.0.weight", "fuse_det.2.weight")


def heavy_tail_state_dict(sd, bits, seed):
    """Heavy-tailed weights for the range tests of the fp16 piece arithmetic: every element of the matrices that
    the two-piece fp16 kernels read - the four aug_shape first layers (one range exponent per 128 000-entry row at N=500), the first and
    second layers of the three pair MLPs - is multiplied by 2^(k - bits/2), k = the number of set bits among `bits` random bits (bits =
    32: a log-normal-like multiplier with sigma = 1.96 in natural-log units, up to 2^+-16; the largest of 128 000 entries is ~1e3 x the
    mean magnitude of its row instead of ~2 x for the default uniform init), and the matrix is rescaled to its old root-mean-square.
    Integer randomness and exact powers of two only, so the weights are bit-identical on every host (exp / randn are not: their last
    bit depends on the CPU's vector path).  Applied IN PLACE to a state_dict, identically to the reference model (make_golden.py) and to
    the shasta_amd model (tests)."""
    keys = [k % i for k in HEAVY_KEYS[:1] for i in range(4)] + list(HEAVY_KEYS[1:])
    bits = int(bits)
    with torch.no_grad():
        for n, k in enumerate(keys):
            w = sd[k]
            g = torch.Generator().manual_seed(1000 * seed + n)
            rows = max(1, (1 << 24) // max(1, w.shape[1]))  # in slabs of 16 M elements: the N=500 matrices hold 256 M each
            ms = 0.0
            for r0 in range(0, w.shape[0], rows):
                x = torch.randint(0, 1 << bits, w[r0:r0 + rows].shape, generator=g, dtype=torch.int64)
                x = x - ((x >> 1) & 0x55555555)
                x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
                x = (x + (x >> 4)) & 0x0F0F0F0F
                x = ((x * 0x01010101) >> 24) & 0xFF      # population count
                mlt = torch.ldexp(torch.ones((), dtype=torch.float32), (x - bits // 2).to(torch.int32))
                ms += float(mlt.double().pow(2).sum())
                w[r0:r0 + rows].mul_(mlt.to(w.device))
            w.mul_(float((ms / w.numel()) ** -0.5))
    return sd


def build_model(c):
    """Seeded default init: the module mirrors the reference constructor's RNG consumption, so the weights equal the
    ones the reference had when the golden was generated (checked against the stored checksums).  Goldens with a `sharp`
    entry get the same sharpening the reference model got in make_golden.py."""
    import shasta_amd
    torch.manual_seed(c["seed"])
    m = shasta_amd.build_simp_track(model_cfg(c)).eval()
    if c.get("heavy") is not None:
        heavy_tail_state_dict(m.state_dict(), *c["heavy"])
    if c.get("sharp") is not None:
        sharpen_state_dict(m.state_dict(), *c["sharp"])
    return m


def check_weight_sums(sd, sums, rtol=1e-9):
    for k, (s, a) in sums.items():
        v = sd[k].detach().double().cpu()
        assert abs(float(v.sum()) - s) <= rtol * max(1.0, a), "weight checksum mismatch for " + k
        assert abs(float(v.abs().sum()) - a) <= rtol * max(1.0, a), "weight |checksum| mismatch for " + k


def golden_weights(z, c, sums):
    """state_dict for the case: stored in the npz for the tiny cases, re-created from the seed otherwise."""
    stored = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    if stored:
        return stored
    sd = {k: v.detach() for k, v in build_model(c).state_dict().items()}
    check_weight_sums(sd, sums)
    return sd


def row_argmax_agreement(a, b, ref_margin_tol):
    """Compares argmax along the last axis; rows whose reference top-2 margin is <= ref_margin_tol are reported
    separately (an fp32 re-association may legitimately flip them)."""
    a = np.asarray(a)
    b = np.asarray(b)
    ia, ib = a.argmax(-1), b.argmax(-1)
    srt = np.sort(b, axis=-1)
    margin = srt[..., -1] - srt[..., -2]
    decided = margin > ref_margin_tol
    return (ia == ib), decided


# ---- pins on the reference's intermediates (shasta.py:241-247 geom, :260-267 anchor boxes, :319 residual, :323 matched) ----
# Tolerances: relative to the largest magnitude of the tensor, i.e. |got - ref| <= rtol * |ref| + rel_atol * max|ref|.
# The HIP path measures ~1e-7 (tables, residual) ... 1e-6 (geom: two dot products of 128 000 and 2 000 terms in another
# summation order than ATen's); a 1 % error in any first-layer weight matrix moves these tensors by 1e-3 ... 1e-2.
PIN_TOL = dict(feature=(1e-5, 2e-5), geom=(1e-5, 1e-5), anchors=(1e-5, 1e-5), residual=(1e-5, 1e-5), matched=(1e-5, 1e-5))
M_ATOL = 1e-6        # matched1 / matched2 of the default-init goldens (values ~1/N: measured 1e-9 ... 1e-8)
# sharpened goldens: probabilities up to 1 from logits of magnitude 1e3.  `matched` itself is pinned to 1e-5 of its largest entry
# (a few 1e-3 absolute: an fp32 sum in another order cannot do better, one ulp of such a logit is 6e-5 ... 2.4e-4), and the softmax
# passes a logit error on scaled by p(1-p) <= 1/4: measured 2e-4 against the reference, 6e-4 between two batch sizes.
M_ATOL_SHARP = 1e-3
ARGMAX_AGREEMENT = []  # filled by check_outputs for default-init goldens; printed by tests/test_hip_parity.py::test_zz_argmax_agreement_report


def _close(name, got, ref, tol):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, got.shape, ref.shape)
    rtol, rel_atol = tol
    bound = rtol * np.abs(ref) + rel_atol * max(float(np.abs(ref).max()), 1e-30)
    err = np.abs(got - ref)
    bad = err > bound
    assert not bad.any(), "%s: %d of %d elements off, max |diff| %.3e (max |ref| %.3e)" % (name, int(bad.sum()), bad.size, float(err.max()),
                                                                                             float(np.abs(ref).max()))
    return float((err / np.maximum(bound, 1e-300)).max())


def check_intermediates(z, got, frames=None, report=None):
    """Pin the intermediates of a forward on the golden `z`.  `got`: dict with numpy arrays feature / prev_feature (B, N+2, F)
    tables (rows N, N+1 = the aug_shape anchors), det_tab / prev_tab (B, N+2, >=7) (rows N, N+1 = the aug_dets anchors),
    residual, matched (B, N+2, N+2).  `frames`: batch indices of `got` that correspond to the golden's frames (default: all).
    Works for full goldens (`feature`, `residual`, ... stored) and for the N=500 probe goldens (rows / columns PROBE_IDX stored
    in full, float64 row and column sums for everything else)."""
    N = int(z["cfg"][0])
    sel = slice(None) if frames is None else frames
    g = {k: np.asarray(v)[sel] for k, v in got.items()}
    worst = {}
    geom = np.stack([g["prev_feature"][:, N], g["prev_feature"][:, N + 1], g["feature"][:, N], g["feature"][:, N + 1]])
    worst["geom"] = _close("geom (aug_shape anchors)", geom, z["geom"], PIN_TOL["geom"])
    for k, tab, row in (("newborn", "prev_tab", N), ("fp", "prev_tab", N + 1), ("dead_trk", "det_tab", N), ("fn", "det_tab", N + 1)):
        worst[k] = _close(k, g[tab][:, row:row + 1, :7], z[k], PIN_TOL["anchors"])
    if "residual" in z.files:
        worst["feature"] = _close("feature", g["feature"][:, :N], z["feature"], PIN_TOL["feature"])
        worst["prev_feature"] = _close("prev_feature", g["prev_feature"][:, :N], z["prev_feature"], PIN_TOL["feature"])
        worst["residual"] = _close("residual", g["residual"], z["residual"], PIN_TOL["residual"])
        worst["matched"] = _close("matched", g["matched"], z["matched"], PIN_TOL["matched"])
    else:
        idx = [i for i in PROBE_IDX if i < N]
        for k in ("feature", "prev_feature"):
            worst[k] = _close(k + " probe rows", g[k][:, idx], z[k + "_rows"], PIN_TOL["feature"])
            # a sum over F entries of magnitude m carries ~sqrt(F) * tol * m of admissible error: same relative bound on the sum
            _close(k + " row sums", g[k][:, :N].astype(np.float64).sum(-1), z[k + "_rowsum"], PIN_TOL["feature"])
        for k in ("residual", "matched"):
            worst[k] = max(_close(k + " probe rows", g[k][:, PROBE_IDX], z[k + "_rows"], PIN_TOL[k]),
                           _close(k + " probe columns", g[k][:, :, PROBE_IDX], z[k + "_cols"], PIN_TOL[k]))
            # checksums of every row and column: |sum| grows like the entries, errors average out -> same relative bound
            _close(k + " row abs-sums", np.abs(g[k].astype(np.float64)).sum(-1), z[k + "_rowabs"], PIN_TOL[k])
            _close(k + " column abs-sums", np.abs(g[k].astype(np.float64)).sum(-2), z[k + "_colabs"], PIN_TOL[k])
        # padded golden: zero rows carry log(1e-10) terms (|residual| up to 38 against 4 between real boxes), which sets the scale of the
        # bounds above; the real x real block of the probe rows is pinned against its OWN scale, so that the learned terms keep their teeth
        n_real = int(z["cfg"][4])
        real_rows = [j for j, i in enumerate(PROBE_IDX) if 0 <= i < n_real]
        if 0 < n_real < N and real_rows:
            _close("residual, real x real block of the probe rows", g["residual"][:, [PROBE_IDX[j] for j in real_rows], :n_real],
                   z["residual_rows"][:, real_rows, :n_real], PIN_TOL["residual"])
    if report is not None:
        report.update(worst)
    return worst


def check_outputs(z, m1, m2, sharp=None, frames=None):
    """matched1 / matched2 against the golden: absolute tolerance M_ATOL (default init) or M_ATOL_SHARP, and the arg-max of
    EVERY row of matched1 / every column of matched2 (no "decided" mask) for sharpened goldens; default-init goldens (flat
    outputs, top-2 margins down to 1e-9) keep the margin mask at 10 x the tolerance."""
    sel = slice(None) if frames is None else frames
    a1, a2 = np.asarray(m1)[sel], np.asarray(m2)[sel]
    sharp = ("sharp" in z.files) if sharp is None else sharp
    atol = M_ATOL_SHARP if sharp else M_ATOL
    if "matol" in z.files:  # "moderately sharp" goldens (logits O(10)): the contract itself, 1e-4 AND every arg-max
        atol = float(z["matol"])
    np.testing.assert_allclose(a1, z["m1"], rtol=0, atol=atol)
    np.testing.assert_allclose(a2, z["m2"], rtol=0, atol=atol)
    if sharp:
        for name, a, r in (("row arg-max of matched1", a1, z["m1"]), ("column arg-max of matched2", np.swapaxes(a2, 1, 2), np.swapaxes(z["m2"], 1, 2))):
            ia, ir = a.argmax(-1), r.argmax(-1)
            srt = np.sort(r, axis=-1)
            tied = srt[..., -1] == srt[..., -2]  # exact ties in the reference: identical zero-padded rows (never between real rows)
            assert np.array_equal(ia[~tied], ir[~tied]), name + " differs"
            # on an exact tie the reference's pick must be a maximum of ours too (identical inputs -> identical outputs)
            assert np.array_equal(np.take_along_axis(a, ir[..., None], -1)[..., 0][tied], a.max(-1)[tied]), name + ": tie broken differently"
    else:
        same, decided = row_argmax_agreement(a1, z["m1"], 10 * atol)
        assert same[decided].all(), "row argmax differs on a decided row"
        same2, decided2 = row_argmax_agreement(np.swapaxes(a2, 1, 2), np.swapaxes(z["m2"], 1, 2), 10 * atol)
        assert same2[decided2].all(), "column argmax differs on a decided column"
        # the UN-MASKED agreement rate: default-init outputs are flat, top-2 margins go down to 1e-9, so a row whose
        # arg-max differs is a tie of the reference's own rounding; reported, not asserted
        ARGMAX_AGREEMENT.append(dict(rows=int(same.size), rows_same=int(same.sum()), rows_undecided=int((~decided).sum()),
                                     cols=int(same2.size), cols_same=int(same2.sum()), cols_undecided=int((~decided2).sum()),
                                     N=int(z["cfg"][0]), seed=int(z["cfg"][8])))
    return float(np.abs(a1 - z["m1"]).max()), float(np.abs(a2 - z["m2"]).max())


def oracle_tables(im, det_out, prev):
    """Intermediates of oracle.forward_from_bev in the layout check_intermediates expects."""
    cat = lambda *t: torch.cat(t, dim=1).numpy()  # noqa: E731
    return dict(feature=cat(im["feature"], im["dead_trk_geom"], im["fn_geom"]),
                prev_feature=cat(im["prev_feature"], im["newborn_geom"], im["fp_geom"]),
                det_tab=cat(det_out[:, :, :7], im["dead_trk"], im["fn"]), prev_tab=cat(prev[:, :, :7], im["newborn"], im["fp"]),
                residual=im["residual"].numpy(), matched=im["matched"].numpy())
