"""CPU study (no GPU): how much accuracy do different representations of the per-pair hidden activations h1 = relu(UP[t] + UC[d])
cost in the pair stage (shasta.py:286-319) when their second layers run on the f16 matrix path?  Everything is compared with a
float64 evaluation of the reference formulation on the same tables (oracle = checker only).

 f32      : the factorised evaluation in fp32 (what pair_mfma4_kernel computes, up to summation order)
 cut      : h1 in fp32, cut per pair into two round-to-nearest fp16 pieces under a per-(track, 64-detection tile) power-of-two scale
            (pair_f16.hip, round 2)
 grid     : UP[t] and UC[d] cut ONCE per row into two fp16 pieces on a common fixed grid (high piece = multiple of G, low piece =
            multiple of G 2^-11, G from the largest |UP| of the track group + the largest |UC| of the tile); the per-pair work is then
            packed fp16 adds / maxima only, all exact
usage: python tools/pair_quant_sim.py [--max-obj 100] [--gain 1] [--spread 0] [--group 64]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import shasta_amd  # noqa: E402
from oracle import shasta_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-obj", type=int, default=100)
ap.add_argument("--feats", type=int, default=7)
ap.add_argument("--points", type=int, default=4)
ap.add_argument("--gain", type=float, default=1.0)
ap.add_argument("--spread", type=float, default=0.0)
ap.add_argument("--group", type=int, default=64, help="tracks that share one grid with a detection tile")
ap.add_argument("--hbits", type=int, default=11)
ap.add_argument("--lbits", type=int, default=11)
a = ap.parse_args()
torch.manual_seed(0)
N, nf, npnt = a.max_obj, a.feats, a.points
F = 64 * npnt
model = shasta_amd.build_simp_track(dict(type="Shasta", reader=None, backbone=None, neck=None,
                                         bev_extractor=dict(type="BEVFeatureExtractor", pc_start=[-54, -54], voxel_size=[0.075, 0.075], out_stride=8),
                                         max_obj=N, num_feats=nf, num_point=npnt)).eval()
if a.gain != 1.0:
    with torch.no_grad():
        for m in (model.fuse_shape, model.fuse_det, model.res_coeff):
            for l in m:
                if hasattr(l, "weight"):
                    l.weight.mul_(a.gain)
w = {k: v.detach().clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(3)
B = 1
bev = torch.relu(torch.randn(B, 180, 180, 64, generator=g))
pbev = torch.relu(torch.randn(B, 180, 180, 64, generator=g))
if a.spread:
    ramp = torch.pow(10.0, torch.linspace(-a.spread, a.spread, 180)).view(1, 1, 180, 1)
    bev, pbev = bev * ramp, pbev * ramp
det, prev = O.synth_boxes(g, B, N), O.synth_boxes(g, B, N)
detc = det.clone()
_, _, im = O.forward_from_bev(w, bev, pbev, detc, prev.clone(), nf, npnt, return_intermediates=True)
pf = torch.cat([im["prev_feature"], im["newborn_geom"], im["fp_geom"]], 1)[0]      # (T, F)
cf = torch.cat([im["feature"], im["dead_trk_geom"], im["fn_geom"]], 1)[0]          # (D, F)
p7 = torch.cat([prev[:, :, :7], im["newborn"], im["fp"]], 1)[0]
d7 = torch.cat([detc[:, :, :7], im["dead_trk"], im["fn"]], 1)[0]                   # back-projected in place by the oracle
T, D = pf.shape[0], cf.shape[0]
w64 = {k: v.double() for k, v in w.items()}
ref = O.pair_residual(w64, pf[None].double(), cf[None].double(), p7[None].double(), d7[None].double(), nf, chunk=16)[0]
scale = float(ref.abs().max())


def rows(dt):
    """row embeddings UP (T, 128), UC (D, 128) in the k order [fuse_shape 32 | res_coeff 64 | fuse_det 32], bias on the UC side"""
    W = {k: v.to(dt) for k, v in w.items()}
    pfe, cfe, pb, cb = pf.to(dt), cf.to(dt), p7[:, :nf].to(dt), d7[:, :nf].to(dt)
    fs, rc, fd = W["fuse_shape.0.weight"], W["res_coeff.0.weight"], W["fuse_det.0.weight"]
    UP = torch.cat([pfe @ fs[:, :F].t(), pfe @ rc[:, :F].t() + pb @ rc[:, F:F + nf].t(), pb @ fd[:, :nf].t()], 1)
    UC = torch.cat([cfe @ fs[:, F:].t() + W["fuse_shape.0.bias"],
                    cfe @ rc[:, F + nf:2 * F + nf].t() + cb @ rc[:, 2 * F + nf:].t() + W["res_coeff.0.bias"],
                    cb @ fd[:, nf:].t() + W["fuse_det.0.bias"]], 1)
    return UP, UC


def tails(z_fs, z_rc, z_fd, dt):
    """layers 3-4 + hand residual + combine from the layer-2 pre-activations (without bias), (T, D, .)"""
    W = {k: v.to(dt) for k, v in w.items()}
    h = torch.relu(z_fs + W["fuse_shape.2.bias"])
    h = torch.relu(h @ W["fuse_shape.4.weight"].t() + W["fuse_shape.4.bias"])
    shape = (h @ W["fuse_shape.6.weight"].t() + W["fuse_shape.6.bias"])[..., 0]
    h = torch.relu(z_rc + W["res_coeff.2.bias"])
    coeff = h @ W["res_coeff.4.weight"].t() + W["res_coeff.4.bias"]
    h = torch.relu(z_fd + W["fuse_det.2.bias"])
    fused = (h @ W["fuse_det.4.weight"].t() + W["fuse_det.4.bias"])[..., 0]
    dist = O.hand_residual(p7[None].to(dt), d7[None].to(dt), nf)[0]
    return coeff[..., 0] * fused + coeff[..., 1] * dist + coeff[..., 2] * shape


def cut16(x):
    """two round-to-nearest fp16 pieces of float64 x (already scaled into range)"""
    h = x.to(torch.float16).double()
    l = (x - h).to(torch.float16).double()
    return h, l


def wpieces(name):
    W = w[name].double()
    e = np.floor(np.log2(float(W.abs().max())))
    s = 2.0 ** (13 - e)
    h, l = cut16(W * s)
    return h / s, l / s


W2 = {n: wpieces(n) for n in ("fuse_shape.2.weight", "res_coeff.2.weight", "fuse_det.2.weight")}
SL = {"fuse_shape.2.weight": slice(0, 32), "res_coeff.2.weight": slice(32, 96), "fuse_det.2.weight": slice(96, 128)}


def layer2(xh, xl):
    """three piece products per fp32 product, accumulated exactly (float64) and rounded to fp32 once"""
    out = []
    for n in ("fuse_shape.2.weight", "res_coeff.2.weight", "fuse_det.2.weight"):
        wh, wl = W2[n]
        a_h, a_l = xh[..., SL[n]], xl[..., SL[n]]
        z = a_h @ wh.t() + a_l @ wh.t() + a_h @ wl.t()
        out.append(z.float())
    return out


def report(name, res):
    err = (res.double() - ref).abs()
    print("%-28s max %.3e  rms %.3e   (relative to max|residual| = %.3g: %.2e / %.2e)" % (
        name, float(err.max()), float(err.pow(2).mean().sqrt()), scale, float(err.max()) / scale, float(err.pow(2).mean().sqrt()) / scale))


# f32 factorised
UP32, UC32 = rows(torch.float32)
h1 = torch.relu(UP32[:, None, :] + UC32[None, :, :])
W32 = w
z = [h1[..., SL[n]] @ W32[n].t() for n in ("fuse_shape.2.weight", "res_coeff.2.weight", "fuse_det.2.weight")]
report("f32 factorised", tails(z[0], z[1], z[2], torch.float32))

# cut per pair (round 2): scale per (track, 64-detection tile)
UPd, UCd = UP32.double(), UC32.double()
res = torch.empty(T, D)
xh = torch.empty(T, D, 128, dtype=torch.float64)
xl = torch.empty_like(xh)
for d0 in range(0, D, 64):
    d1 = min(D, d0 + 64)
    mc = float(UCd[d0:d1].abs().max())
    for t in range(T):
        m = float(UPd[t].abs().max()) + mc
        e = 13 - np.floor(np.log2(m)) if m > 0 else 0
        s = 2.0 ** e
        hh = torch.relu(UP32[t][None] + UC32[d0:d1]).double() * s   # fp32 sum, as the kernel forms it
        ph, pl = cut16(hh)
        xh[t, d0:d1], xl[t, d0:d1] = ph / s, pl / s
z = layer2(xh, xl)
report("cut per pair (round 2)", tails(z[0], z[1], z[2], torch.float32))


def grid_pieces(x, G, hbits, lbits):
    """x = h + l + err, h a multiple of G, l a multiple of G 2^-lbits with |l| <= G / 2 (round to nearest both times)"""
    h = torch.round(x / G) * G
    gl = G * 2.0 ** (-lbits)
    l = torch.round((x - h) / gl) * gl
    return h, l


for hb, lb in ((a.hbits, a.lbits), (10, 10), (11, 10)):
    for d0 in range(0, D, 64):
        d1 = min(D, d0 + 64)
        mc = float(UCd[d0:d1].abs().max())
        for t0 in range(0, T, a.group):
            t1 = min(T, t0 + a.group)
            S = float(UPd[t0:t1].abs().max()) + mc
            # G: the smallest power of two with S / G <= 2^hbits
            G = 2.0 ** np.ceil(np.log2(S) - hb) if S > 0 else 1.0
            uph, upl = grid_pieces(UPd[t0:t1], G, hb, lb)
            uch, ucl = grid_pieces(UCd[d0:d1], G, hb, lb)
            sh = uph[:, None, :] + uch[None, :, :]
            sl = upl[:, None, :] + ucl[None, :, :]
            assert float((sh.to(torch.float16).double() - sh).abs().max()) == 0.0 or hb > 11
            h2 = torch.clamp(sh, min=-G)
            l2 = torch.maximum(sl, -h2)
            xh[t0:t1, d0:d1], xl[t0:t1, d0:d1] = h2, l2
    z = layer2(xh, xl)
    report("grid %d+%d bits, %d tracks" % (hb, lb, a.group), tails(z[0], z[1], z[2], torch.float32))

# per-MLP grids: the three k ranges (fuse_shape | res_coeff | fuse_det) get their own G (their accumulators are separate)
print("row maxima per k range: UP", [float(UPd[:, SL[n]].abs().max()) for n in SL], "UC", [float(UCd[:, SL[n]].abs().max()) for n in SL])
print("row rms per k range: UP", [float(UPd[:, SL[n]].pow(2).mean().sqrt()) for n in SL], "UC", [float(UCd[:, SL[n]].pow(2).mean().sqrt()) for n in SL])
for hb, lb in ((11, 11), (11, 10), (10, 10)):
    for d0 in range(0, D, 64):
        d1 = min(D, d0 + 64)
        for t0 in range(0, T, a.group):
            t1 = min(T, t0 + a.group)
            for n in SL:
                k = SL[n]
                S = float(UPd[t0:t1, k].abs().max()) + float(UCd[d0:d1, k].abs().max())
                G = 2.0 ** np.ceil(np.log2(S) - hb) if S > 0 else 1.0
                uph, upl = grid_pieces(UPd[t0:t1, k], G, hb, lb)
                uch, ucl = grid_pieces(UCd[d0:d1, k], G, hb, lb)
                sh = uph[:, None, :] + uch[None, :, :]
                sl = upl[:, None, :] + ucl[None, :, :]
                h2 = torch.clamp(sh, min=-G)
                l2 = torch.maximum(sl, -h2)
                xh[t0:t1, d0:d1, k], xl[t0:t1, d0:d1, k] = h2, l2
    z = layer2(xh, xl)
    report("per-MLP grid %d+%d, %d tracks" % (hb, lb, a.group), tails(z[0], z[1], z[2], torch.float32))

# per-feature grids: G_k from the largest |UP[., k]| of the track group + the largest |UC[., k]| of the tile (the weight columns take the
# inverse power of two, exact)
for hb, lb, grp in ((11, 11, a.group), (11, 11, T), (11, 10, T)):
    for d0 in range(0, D, 64):
        d1 = min(D, d0 + 64)
        for t0 in range(0, T, grp):
            t1 = min(T, t0 + grp)
            S = UPd[t0:t1].abs().amax(0) + UCd[d0:d1].abs().amax(0)          # (128,)
            G = torch.pow(2.0, torch.ceil(torch.log2(S.clamp_min(1e-30)) - hb))
            gl = G * 2.0 ** (-lb)
            def gp(x):
                h = torch.round(x / G) * G
                return h, torch.round((x - h) / gl) * gl
            uph, upl = gp(UPd[t0:t1])
            uch, ucl = gp(UCd[d0:d1])
            sh = uph[:, None, :] + uch[None, :, :]
            sl = upl[:, None, :] + ucl[None, :, :]
            h2 = torch.maximum(sh, -G)
            l2 = torch.maximum(sl, -h2)
            xh[t0:t1, d0:d1], xl[t0:t1, d0:d1] = h2, l2
    z = layer2(xh, xl)
    report("per-feature grid %d+%d, %d trk" % (hb, lb, grp), tails(z[0], z[1], z[2], torch.float32))
S = UPd.abs().amax(0) + UCd.abs().amax(0)
print("per-feature S, sorted, per MLP:", [np.round(np.sort(S[SL[n]].numpy()), 2).tolist() for n in SL])
