"""Training path of the affinity network (BASELINE config 5): `affinity_train(model, example)`
returns (matched1, matched2) that carry autograd history, so the reference's training loop
(tools/nusc_shasta/train.py:198-218: masked NLL in both directions, Adam) runs unchanged on top of it.

Forward = the inference kernels (identical values).  Backward (`_AffinityTrainFn.backward`) recomputes the activations it
needs with the REFERENCE (dense) formulation of the pair MLPs - the pair tensor is materialised - and back-propagates
with hand-written HIP kernels only: every nn.Linear through the strided matrix-core GEMM (dX = dY W, dW = dY^T X, fixed
order split-K), everything else through csrc/train.hip.  No torch autograd op runs inside the Function; torch is used for
buffer allocation.  Gradients flow to every parameter of rows 6-16 and to the NHWC BEV maps (so `shared_conv`, which stays
a torch module in train() mode with batch statistics like the reference, trains through ordinary autograd).
The first layer of each pair MLP is factorised over the table rows in both directions (forward UP[t] + UC[d], backward
row/column sums of the hidden gradient), so the (B, T*D, 2F) pair tensor of the reference is not materialised in training
either; only the narrow hidden activations (F/8, F/8+32, 32 floats per pair) are.  Checked against torch autograd of the CPU
oracle (tests/test_training.py).
"""
import ctypes as C

import torch

from . import hip


def _gemm(lib, A, sa, W, sw, M, N, K, out, ldc=None, bias=None, act=0, mask=None, ldmask=0, accum=False, ws=None, bf16=False):
    """bf16: operands rounded to bf16 on chip, bf16 matrix path, fp32 accumulation and output (`Shasta.train_precision = "bf16"`)"""
    hip.check(lib.shasta_gemm_strided_f32(hip.ptr_view(A), sa[0], sa[1], hip.ptr_view(W), sw[0], sw[1], hip.ptr(bias), hip.ptr_view(mask), ldmask,
                                          hip.ptr_view(out), ldc if ldc is not None else N, M, N, K, act + (4 if accum else 0) + (8 if bf16 else 0),
                                          hip.ptr(ws), ws.numel() * 4 if ws is not None else 0, hip.stream_ptr()),
              "shasta_gemm_strided_f32")
    return out


def _gemm_group(lib, As, sa, Ws, sw, M, N, K, outs, ldc=None, biases=None, act=0, masks=None, ldmask=0, accum=False, ws=None, bf16=False):
    """len(As) <= 8 products of one shape in one launch (shasta_gemm_strided_group_f32): As / Ws / outs (/ biases / masks) are lists of
    tensors or views, strides and sizes are shared; each member exactly as _gemm would compute it."""
    n = len(As)
    if n == 1:
        return [_gemm(lib, As[0], sa, Ws[0], sw, M, N, K, outs[0], ldc=ldc, bias=biases[0] if biases else None, act=act,
                      mask=masks[0] if masks else None, ldmask=ldmask, accum=accum, ws=ws, bf16=bf16)]
    arr = lambda ts: (C.c_void_p * n)(*[hip.ptr_view(t).value if t is not None else None for t in ts])
    hip.check(lib.shasta_gemm_strided_group_f32(n, arr(As), arr(Ws), arr(biases) if biases else None, arr(masks) if masks else None, arr(outs),
                                                sa[0], sa[1], sw[0], sw[1], ldmask, ldc if ldc is not None else N, M, N, K,
                                                act + (4 if accum else 0) + (8 if bf16 else 0), hip.ptr(ws),
                                                ws.numel() * 4 if ws is not None else 0, hip.stream_ptr()), "shasta_gemm_strided_group_f32")
    return outs


def _outer(lib, G, ldg, X, ldx, R, H, K, dW):
    """dW (H, K) = G[:R, :H]^T X[:R, :K]: the rank-R update of an anchor first-layer weight gradient.  Small R: a streaming
    kernel (one pass over the 1 GB output); otherwise the MFMA GEMM."""
    if R <= 16 and K % 4 == 0 and ldx % 4 == 0:
        hip.check(lib.shasta_lowrank_outer_f32(hip.ptr_view(G), ldg, hip.ptr_view(X), ldx, R, H, K, hip.ptr(dW), hip.stream_ptr()),
                  "shasta_lowrank_outer_f32")
    else:
        _gemm(lib, G, (1, ldg), X, (1, ldx), H, K, R, dW)


def _colsum(lib, Y, ldy, M, N, out, ws):
    hip.check(lib.shasta_colsum_f32(hip.ptr(Y), ldy, M, N, hip.ptr(out), hip.ptr(ws), ws.numel() * 4 if ws is not None else 0,
                                    hip.stream_ptr()), "shasta_colsum_f32")


class _Mlp:
    """Recompute-and-backprop helper for a Sequential(Linear, ReLU, ..., Linear) applied to the rows of X (M, ldx)."""

    def __init__(self, lib, layers, ws, bf16=False):
        self.lib, self.layers, self.ws, self.bf16 = lib, layers, ws, bf16  # layers: list of (weight (out,in), bias (out,))

    def forward(self, X, ldx, M):
        acts = [X]
        ld = ldx
        for i, (w, b) in enumerate(self.layers):
            out = torch.empty(M, w.shape[0], device=X.device)
            last = i + 1 == len(self.layers)
            _gemm(self.lib, acts[-1], (ld, 1), w, (w.shape[1], 1), M, w.shape[0], w.shape[1], out, bias=b, act=0 if last else 1, bf16=self.bf16)
            acts.append(out)
            ld = w.shape[0]
        self.acts, self.ldx, self.M = acts, ldx, M
        return acts[-1]

    def backward(self, gY, need_gx=True, gx_ld=None, mask_input=False):
        """gY (M, out_last).  Returns (list of (gW, gb)), gX (M, gx_ld) or None.  mask_input: X is itself a ReLU output and
        the returned gradient is the one of its pre-activation."""
        lib, M, bf = self.lib, self.M, self.bf16
        grads = [None] * len(self.layers)
        g = gY
        for i in range(len(self.layers) - 1, -1, -1):
            w, b = self.layers[i]
            x = self.acts[i]
            ldx = self.ldx if i == 0 else self.layers[i - 1][0].shape[0]
            nout, nin = w.shape
            gW = torch.empty_like(w)
            _gemm(lib, g, (1, nout), x, (1, ldx), nout, nin, M, gW, ws=self.ws, bf16=bf)  # dW = dY^T X
            gb = torch.empty_like(b)
            _colsum(lib, g, nout, M, nout, gb, self.ws)
            grads[i] = (gW, gb)
            if i > 0:
                gx = torch.empty(M, nin, device=g.device)
                _gemm(lib, g, (nout, 1), w, (1, nin), M, nin, nout, gx, mask=x, ldmask=ldx, bf16=bf)  # dX = (dY W) * relu'(x)
                g = gx
            elif need_gx:
                ld = gx_ld if gx_ld is not None else nin
                gx = torch.zeros(M, ld, device=g.device) if ld != nin else torch.empty(M, nin, device=g.device)
                _gemm(lib, g, (nout, 1), w, (1, nin), M, nin, nout, gx, ldc=ld, mask=x if mask_input else None, ldmask=ldx, bf16=bf)
                g = gx
            else:
                g = None
        return grads, g


class _AffinityTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, bev, prev_bev, det_boxes, prev_det_boxes, *params):
        det_pre = det_boxes.detach().clone()  # the gather and aug_dets see the boxes BEFORE back-projection
        keep = {}
        with torch.no_grad():
            m1, m2 = model.affinity_from_bev(bev.detach(), prev_bev.detach(), det_boxes, prev_det_boxes, _train_keep=keep)
        ctx.saved = dict(bev_shape=bev.shape, det_pre=det_pre, prev=prev_det_boxes.detach().clone(), feat=keep["feat"],
                         prev_feat=keep["prev_feat"], det_tab=keep["det_tab"], prev_tab=keep["prev_tab"],
                         residual=keep["residual"], shape_hidden=keep["shape_hidden"], m1=m1.clone(), m2=m2.clone())
        ctx.model = model
        return m1, m2

    @staticmethod
    def backward(ctx, g1, g2):
        model, S = ctx.model, ctx.saved
        lib = hip.load()
        dev = S["m1"].device
        B, N = S["m1"].shape[0], model.max_obj
        T = D = N + 2
        F, nf = model.aug_shape_output, model.num_feats
        Dp = (D + 3) // 4 * 4
        P = B * T * D
        ws = torch.empty(64 * 1024 * 1024 // 4, device=dev)  # split-K scratch
        # BASELINE config 5's reduced-precision option: the GEMMs of `aff` and of the pair MLPs' first-layer tables (recomputation and
        # gradients) take bf16 operands with fp32 accumulation; parameters, their gradients, Adam and the anchor MLPs stay fp32
        bf = getattr(model, "train_precision", "fp32") == "bf16"
        st = hip.stream_ptr
        g1 = g1.contiguous().float()
        g2 = g2.contiguous().float()

        # ---- softmaxes (shasta.py:324-325) ----
        gm = torch.empty(B * T, Dp, device=dev)
        hip.check(lib.shasta_softmax_bwd_f32(hip.ptr(S["m1"]), hip.ptr(g1), hip.ptr(S["m2"]), hip.ptr(g2), B, N, hip.ptr(gm), Dp, st()),
                  "shasta_softmax_bwd_f32")
        # ---- aff (shasta.py:323): recompute the hidden activations, then back-propagate ----
        residual = S["residual"].reshape(B * T, D).contiguous()
        aff = _Mlp(lib, [(model.aff[k].weight.detach(), model.aff[k].bias.detach()) for k in (0, 2, 4, 6, 8, 10)], ws, bf16=bf)
        aff.forward(residual, D, B * T)
        gmc = gm[:, :D].contiguous()
        aff_grads, gres = aff.backward(gmc, need_gx=True, gx_ld=Dp)  # gres (B*T, Dp)

        # ---- pair MLPs (shasta.py:286-316): first layers factorised over the table rows, later layers dense over pairs ----
        pf, cf, pt, ct = S["prev_feat"], S["feat"], S["prev_tab"], S["det_tab"]
        dfeat = torch.zeros(B, T, F, device=dev)
        dprev_feat = torch.zeros(B, T, F, device=dev)
        ddet_tab = torch.zeros(B, T, 8, device=dev)
        dprev_tab = torch.zeros(B, T, 8, device=dev)
        # column blocks of each first layer: (source table, its row stride, width, gradient table) for the previous / current side
        parts = {
            "fs": ([(pf, F, F, dprev_feat)], [(cf, F, F, dfeat)]),
            "rc": ([(pf, F, F, dprev_feat), (pt, 8, nf, dprev_tab)], [(cf, F, F, dfeat), (ct, 8, nf, ddet_tab)]),
            "fd": ([(pt, 8, nf, dprev_tab)], [(ct, 8, nf, ddet_tab)]),
        }
        lin = lambda m, ks: [(m[k].weight.detach(), m[k].bias.detach()) for k in ks]  # noqa: E731
        mods = {"fs": lin(model.fuse_shape, (0, 2, 4, 6)), "rc": lin(model.res_coeff, (0, 2, 4)), "fd": lin(model.fuse_det, (0, 2, 4))}
        R = B * T

        def first_layer_tables(name):
            """The first layer's two table products: UP (B*T, E) over the previous side's rows, UC (B*D, E) over the current side's
            (bias in UC); and the column offsets of the parts inside W0."""
            w0, b0 = mods[name][0]
            E, kin = w0.shape
            UP, UC = torch.empty(R, E, device=dev), torch.empty(R, E, device=dev)
            col, offs = 0, []
            for side in (0, 1):
                for _, _, wd, _ in parts[name][side]:
                    offs.append(col)
                    col += wd
            npart = len(parts[name][0])
            for k in range(npart):  # the two sides have the same shapes: one launch for both (the parts of a side accumulate in turn)
                (tp, ld, wd, _), (tc, _, _, _) = parts[name][0][k], parts[name][1][k]
                _gemm_group(lib, [tp, tc], (ld, 1), [w0[:, offs[k]:], w0[:, offs[npart + k]:]], (kin, 1), R, E, wd, [UP, UC],
                            biases=[None, b0] if k == 0 else None, accum=k > 0, bf16=bf)
            return UP, UC, offs

        def first_layer(name):
            """H (P, E) = relu(UP[t] + UC[d]); returns H and the column offsets of the parts inside W0."""
            UP, UC, offs = first_layer_tables(name)
            E = UP.shape[1]
            H = torch.empty(P, E, device=dev)
            hip.check(lib.shasta_pair_hidden_f32(hip.ptr(UP), E, hip.ptr(UC), E, B, T, D, E, hip.ptr(H), st()), "shasta_pair_hidden_f32")
            return H, offs

        def first_layer_bwd(name, gZ, offs):
            """gZ (P, E) gradient of the first layer's pre-activation -> (gW0, gb0); table gradients accumulated."""
            E = mods[name][0][0].shape[0]
            gUP, gUC = torch.empty(R, E, device=dev), torch.empty(R, E, device=dev)
            hip.check(lib.shasta_pair_reduce_f32(hip.ptr(gZ), B, T, D, E, hip.ptr(gUP), hip.ptr(gUC), st()), "shasta_pair_reduce_f32")
            return first_layer_bwd_rows(name, gUP, gUC, offs)

        def first_layer_bwd_rows(name, gUP, gUC, offs):
            """gUP (B*T, E), gUC (B*D, E): the pre-activation's gradient summed over the detections / the tracks -> (gW0, gb0)."""
            w0, b0 = mods[name][0]
            E, kin = w0.shape
            gW0, gb0 = torch.empty_like(w0), torch.empty_like(b0)
            _colsum(lib, gUC, E, R, E, gb0, ws)
            npart = len(parts[name][0])
            for k in range(npart):  # both sides per launch (distinct column blocks of gW0, distinct gradient tables)
                (tp, ld, wd, gtp), (tc, _, _, gtc) = parts[name][0][k], parts[name][1][k]
                cp, cc = offs[k], offs[npart + k]
                _gemm_group(lib, [gUP, gUC], (1, E), [tp, tc], (1, ld), E, wd, R, [gW0[:, cp:], gW0[:, cc:]], ldc=kin, ws=ws, bf16=bf)   # dW0 block = gU^T X
                _gemm_group(lib, [gUP, gUC], (E, 1), [w0[:, cp:], w0[:, cc:]], (1, kin), R, wd, E, [gtp, gtc], ldc=ld, accum=True, bf16=bf)  # dX += gU W0 block
            return gW0, gb0

        # The later layers of a pair MLP.  Default (F = 64 | 256 | 320): recomputed and back-propagated per pair on chip in fp32
        # (csrc/pair_bwd.hip) - per pair only the MLP's output is written and its gradient read; train_precision = "bf16" then shows in
        # the GEMMs around them (first-layer tables, aff).  Otherwise (other widths, Shasta.dense_pair_backward): the dense formulation -
        # hidden activations of every pair materialised, strided GEMMs (bf16 operands there too under the option).
        on_chip = bool(lib.shasta_pair_mlp_supported(F)) and not getattr(model, "dense_pair_backward", False)
        kinds = {"fs": 0, "fd": 1, "rc": 2}
        tails, Hs, offs_, tabs, wts = {}, {}, {}, {}, {}
        outs = {}
        for name in ("fs", "rc", "fd"):
            if on_chip:
                UP_, UC_, offs_[name] = first_layer_tables(name)
                tabs[name] = (UP_, UC_)
                later = [t for wb in mods[name][1:] for t in wb] + [None] * (8 - 2 * len(mods[name]))
                wts[name] = ((C.c_void_p * 6)(*[None if t is None else t.data_ptr() for t in later]), later)  # (the tensors kept alive)
                outs[name] = torch.empty(P, mods[name][-1][0].shape[0], device=dev)
                hip.check(lib.shasta_pair_mlp_forward_f32(kinds[name], F, hip.ptr(UP_), hip.ptr(UC_), wts[name][0], B, T, D,
                                                          hip.ptr(outs[name]), st()), "shasta_pair_mlp_forward_f32")
                continue
            Hs[name], offs_[name] = first_layer(name)
            tails[name] = _Mlp(lib, mods[name][1:], ws, bf16=bf)
            outs[name] = tails[name].forward(Hs[name], Hs[name].shape[1], P)
        shape, coeff, fused = outs["fs"], outs["rc"], outs["fd"]  # (P,1), (P,3), (P,1)
        dist = torch.empty(B * T, Dp, device=dev)
        denom = torch.empty(2 * B * D, device=dev)
        hip.check(lib.shasta_hand_dist_f32(hip.ptr(pt), hip.ptr(ct), B, T, D, nf, hip.ptr(dist), Dp, hip.ptr(denom), st()),
                  "shasta_hand_dist_f32")
        gcoeff, gfused, gshape = torch.empty(P, 3, device=dev), torch.empty(P, 1, device=dev), torch.empty(P, 1, device=dev)
        gdist = torch.zeros(B * T, Dp, device=dev)
        hip.check(lib.shasta_combine_bwd_f32(hip.ptr(gres), hip.ptr(coeff), 3, hip.ptr(fused), 1, hip.ptr(shape), 1, hip.ptr(dist), B, T, D, Dp,
                                             hip.ptr(gcoeff), hip.ptr(gfused), hip.ptr(gshape), hip.ptr(gdist), st()), "shasta_combine_bwd_f32")
        pair_grads = {}
        for name, gout in (("fs", gshape), ("rc", gcoeff), ("fd", gfused)):
            if on_chip:
                UP_, UC_ = tabs[name]
                E = UP_.shape[1]
                nb = lib.shasta_pair_mlp_workspace_bytes(kinds[name], F, B, T, D)
                pws = torch.empty((nb + 3) // 4, device=dev)
                gUP, gUC = torch.empty(R, E, device=dev), torch.empty(R, E, device=dev)
                img = torch.empty(lib.shasta_pair_mlp_grad_floats(kinds[name], F), device=dev)
                hip.check(lib.shasta_pair_mlp_backward_f32(kinds[name], F, hip.ptr(UP_), hip.ptr(UC_), wts[name][0], hip.ptr(gout), B, T, D,
                                                           hip.ptr(gUP), hip.ptr(gUC), hip.ptr(img), hip.ptr(pws), nb, st()),
                          "shasta_pair_mlp_backward_f32")
                tail_grads, o = [], 0
                for w_, b_ in mods[name][1:]:  # the image: [gW2 | gb2 | gW3 | gb3 | gW4 | gb4]
                    tail_grads.append((img[o:o + w_.numel()].view_as(w_), img[o + w_.numel():o + w_.numel() + b_.numel()]))
                    o += w_.numel() + b_.numel()  # (views at any 4-byte offset: FusedAdam updates small tensors without alignment demands)
                pair_grads[name] = [first_layer_bwd_rows(name, gUP, gUC, offs_[name])] + tail_grads
                del pws
                continue
            tail_grads, gZ = tails[name].backward(gout, mask_input=True)
            pair_grads[name] = [first_layer_bwd(name, gZ, offs_[name])] + tail_grads
            del gZ
        fs_grads, rc_grads, fd_grads = pair_grads["fs"], pair_grads["rc"], pair_grads["fd"]
        del Hs, tails, outs
        # ---- hand-designed residual -> anchor boxes (rows N, N+1 of both tables) ----
        hip.check(lib.shasta_hand_dist_bwd_f32(hip.ptr(gdist), Dp, hip.ptr(S["prev_tab"]), hip.ptr(S["det_tab"]), hip.ptr(denom), B, T, D, nf,
                                               N, 2, hip.ptr(dprev_tab), hip.ptr(ddet_tab), st()), "shasta_hand_dist_bwd_f32")

        # ---- anchor MLPs (shasta.py:241-247, 260-267) ----
        def anchor_bwd(seq, x, sx_m, K, g_out, c0, c1, hid=None, defer_w1=False):
            """seq = Sequential(Linear, ReLU, Linear); x rows at stride sx_m; g_out (B, out) gradient of the |.| output;
            hid: the hidden activations when the forward kept them (recomputed otherwise); defer_w1: leave the first
            layer's weight gradient to the caller (returned as None), who gets its factor ghid."""
            w1, b1, w2, b2 = seq[0].weight.detach(), seq[0].bias.detach(), seq[2].weight.detach(), seq[2].bias.detach()
            H, nout = w1.shape[0], w2.shape[0]
            if hid is None:
                hid = torch.empty(B, max(H, 1), device=dev)
                if H > 0:
                    _gemm(lib, x, (sx_m, 1), w1, (K, 1), B, H, K, hid, ldc=max(H, 1), bias=b1, act=1, ws=ws)  # (few rows, long K: split)
            pre = torch.empty(B, nout, device=dev)
            if H > 0:
                _gemm(lib, hid, (max(H, 1), 1), w2, (H, 1), B, nout, H, pre, bias=b2, ws=ws)
            else:
                pre.copy_(b2.expand(B, nout))
            gpre = torch.empty(B, nout, device=dev)
            hip.check(lib.shasta_abs_f32(hip.ptr(pre), hip.ptr(g_out), hip.ptr(gpre), B * nout, nout, c0, c1, 1, st()), "shasta_abs_f32")
            gb2 = torch.empty_like(b2)
            _colsum(lib, gpre, nout, B, nout, gb2, ws)
            gW2, gb1 = torch.zeros_like(w2), torch.zeros_like(b1)
            gW1 = None if (defer_w1 and H > 0) else (torch.empty_like(w1) if H > 0 else torch.zeros_like(w1))
            gx = None
            if H > 0:
                _gemm(lib, gpre, (1, nout), hid, (1, max(H, 1)), nout, H, B, gW2)
                ghid = torch.empty(B, H, device=dev)
                _gemm(lib, gpre, (nout, 1), w2, (1, H), B, H, nout, ghid, mask=hid, ldmask=max(H, 1))
                if gW1 is not None:
                    _outer(lib, ghid, H, x, sx_m, B, H, K, gW1)
                _colsum(lib, ghid, H, B, H, gb1, ws)
                gx = ghid
            return (gW1, gb1, gW2, gb2), gx, w1

        # Data-parallel training: dW1 of an aug_shape MLP is ghid^T x, a rank-B update of a (N*F/64, N*F) matrix (1 GB at
        # N=500).  Instead of all-reducing 4 x 1 GB of gradients, the ranks exchange the FACTORS (all_gather of B x (4H + 2K)
        # floats per rank, ~1 MB per frame-pair) and every rank forms the averaged gradient with one GEMM over world*B rows.
        for p_ in (model.aug_shape[i][0].weight for i in range(4)):  # a stale flag from a step reduced some other way
            p_._shasta_grad_is_global = False
        world, group, exchange, lowrank, stepper = first_layer_plan(model, B, N * F)
        if world > 1:
            _check_equal_local_batch(B, world, group, dev)
        # FusedAdam(..., in_backward=True): the optimizer steps the four matrices HERE, in the pass that also forms dx = ghid W1 (with the
        # weights as they are before the update): the 1 GB matrix is read once for both (shasta_adam_lowrank_dx_f32)
        in_bwd = lowrank and stepper is not None and stepper.in_backward and world * B <= 64 and B <= 16 and N * F >= 4
        shape_grads, box_grads, ghids = [None] * 4, [None] * 4, [None] * 4

        def anchor_bwd_group(seqs, xs, sx_m, K, g_list, c0, c1, hid_cat=None, defer_w1=False):
            """The four MLPs of an anchor kind at once (they are equal in shape): every nn.Linear product of the four in ONE launch
            (shasta_gemm_strided_group_f32), hidden activations / their gradients / the pre-|.| outputs side by side in (B, 4 H) and
            (B, 4 out) matrices so that one abs and one column-sum launch serve all four.  Returns (grads per MLP, ghid blocks, ghid)."""
            n = len(seqs)
            w1s, b1s = [q[0].weight.detach() for q in seqs], [q[0].bias.detach() for q in seqs]
            w2s, b2s = [q[2].weight.detach() for q in seqs], [q[2].bias.detach() for q in seqs]
            H, nout = w1s[0].shape[0], w2s[0].shape[0]
            blocks = lambda t, w: [t[:, i * w:(i + 1) * w] for i in range(n)]  # noqa: E731
            if hid_cat is None:
                hid_cat = torch.empty(B, n * H, device=dev)
                _gemm_group(lib, xs, (sx_m, 1), w1s, (K, 1), B, H, K, blocks(hid_cat, H), ldc=n * H, biases=b1s, act=1, ws=ws)
            hids = blocks(hid_cat, H)
            pre = torch.empty(B, n * nout, device=dev)
            _gemm_group(lib, hids, (n * H, 1), w2s, (H, 1), B, nout, H, blocks(pre, nout), ldc=n * nout, biases=b2s, ws=ws)
            g_cat = torch.cat(g_list, dim=1)
            gpre = torch.empty_like(pre)
            hip.check(lib.shasta_abs_f32(hip.ptr(pre), hip.ptr(g_cat), hip.ptr(gpre), B * n * nout, nout, c0, c1, 1, st()), "shasta_abs_f32")
            gb2 = torch.empty(n * nout, device=dev)
            _colsum(lib, gpre, n * nout, B, n * nout, gb2, ws)
            gpres = blocks(gpre, nout)
            gW2 = torch.empty(n, nout, H, device=dev)
            _gemm_group(lib, gpres, (1, n * nout), hids, (1, n * H), nout, H, B, [gW2[i] for i in range(n)])
            ghid = torch.empty(B, n * H, device=dev)
            gh = blocks(ghid, H)
            _gemm_group(lib, gpres, (n * nout, 1), w2s, (1, H), B, H, nout, gh, ldc=n * H, masks=hids, ldmask=n * H)
            gb1 = torch.empty(n * H, device=dev)
            _colsum(lib, ghid, n * H, B, n * H, gb1, ws)
            gW1s = [None] * n
            if not defer_w1:
                gW1s = [torch.empty_like(w) for w in w1s]
                if B <= 16 and K % 4 == 0 and sx_m % 4 == 0:  # rank-B updates: the streaming kernel, one pass over each output
                    for i in range(n):
                        _outer(lib, gh[i], n * H, xs[i], sx_m, B, H, K, gW1s[i])
                else:
                    _gemm_group(lib, gh, (1, n * H), xs, (1, sx_m), H, K, B, gW1s)
            return [(gW1s[i], gb1[i * H:(i + 1) * H], gW2[i], gb2[i * nout:(i + 1) * nout]) for i in range(n)], gh, ghid

        # aug_shape[i]: input = rows < N of feat (i<2) / prev_feat (i>=2); output row N + (i&1) of prev_feat (i<2) / feat (i>=2)
        Hs_ = N * F // 64
        ghid_cat = None
        shape_x = [S["feat"], S["feat"], S["prev_feat"], S["prev_feat"]]
        shape_gout = [(dprev_feat if i < 2 else dfeat)[:, N + (i & 1), :] for i in range(4)]
        if Hs_ > 0:
            grads, ghids, ghid_cat = anchor_bwd_group([model.aug_shape[i] for i in range(4)], shape_x, T * F, N * F, shape_gout, 0, F,
                                                      hid_cat=S["shape_hidden"], defer_w1=exchange or lowrank)
            shape_grads = list(grads)
        else:
            for i in range(4):
                shape_grads[i], ghids[i], _ = anchor_bwd(model.aug_shape[i], shape_x[i], T * F, N * F, shape_gout[i].contiguous(), 0, F, hid=None,
                                                         defer_w1=exchange or lowrank)
        # dx = ghid W1 into the rows < N of the input table's gradient feeds the gather's backward only: skipped for a map whose gradient
        # nobody asked for (features from a frozen pipeline; autograd's needs_input_grad), with the gather's backward itself
        need_dx = [bool(ctx.needs_input_grad[1])] * 2 + [bool(ctx.needs_input_grad[2])] * 2
        if ghids[0] is not None and not in_bwd and any(need_dx):  # dx = ghid W1 accumulated into the rows < N of the INPUT table's gradient
            w1s = [model.aug_shape[i][0].weight.detach() for i in range(4)]
            gins = [dfeat, dfeat, dprev_feat, dprev_feat]
            if B <= 16 and (N * F) % 4 == 0:  # small batch: stream the 1 GB matrix once (csrc/train.hip)
                for i in (j for j in range(4) if need_dx[j]):
                    nb = lib.shasta_smallm_nn_workspace_bytes(B, Hs_, N * F)
                    sws = torch.empty((nb + 3) // 4, device=dev)
                    hip.check(lib.shasta_smallm_nn_f32(hip.ptr_view(ghids[i]), ghids[i].stride(0), hip.ptr(w1s[i]), B, Hs_, N * F, hip.ptr(gins[i]),
                                                       T * F, 1, hip.ptr(sws), nb, st()), "shasta_smallm_nn_f32")
            else:  # two MLPs add into each table: one launch for the first of each pair, one for the second
                for pair in ((0, 2), (1, 3)):
                    pair = [i for i in pair if need_dx[i]]
                    if pair:
                        _gemm_group(lib, [ghids[i] for i in pair], (ghids[0].stride(0), 1), [w1s[i] for i in pair], (1, N * F), B, N * F, Hs_,
                                    [gins[i] for i in pair], ldc=T * F, accum=True)
        # aug_dets[i]: input = boxes[:, :, :7] before back-projection (det for i<2, prev for i>=2), output anchor box row
        xb_det, xb_prev = S["det_pre"][:, :, :7].contiguous(), S["prev"][:, :, :7].contiguous()
        box_x = [xb_det, xb_det, xb_prev, xb_prev]
        box_gout = [(dprev_tab if i < 2 else ddet_tab)[:, N + (i & 1), :7] for i in range(4)]
        if 7 * N // 32 > 0:
            box_grads, _, _ = anchor_bwd_group([model.aug_dets[i] for i in range(4)], box_x, 7 * N, 7 * N, box_gout, 3, 6)
        else:
            for i in range(4):
                box_grads[i], _, _ = anchor_bwd(model.aug_dets[i], box_x[i], 7 * N, 7 * N, box_gout[i].contiguous(), 3, 6)

        if exchange and ghids[0] is not None:
            Hs_, K = N * F // 64, N * F
            gh_all = _all_gather_rows(ghid_cat if ghid_cat is not None else torch.cat(ghids, dim=1), world, group)             # (world*B, 4H)
            xs = [_all_gather_rows(S[k][:, :N, :].reshape(B, K), world, group) for k in ("feat", "prev_feat")]  # (world*B, K)
            hip.check(lib.shasta_scale_f32(hip.ptr(gh_all), gh_all.numel(), 1.0 / world, st()), "shasta_scale_f32")
            for i in range(4):
                w1p = model.aug_shape[i][0].weight
                if lowrank:  # (the parameter's .grad stays None; allreduce_gradients skips it)
                    w1p._shasta_grad_factors = (gh_all[:, i * Hs_:], 4 * Hs_, xs[0 if i < 2 else 1], K, world * B)
                    continue
                gW1 = torch.empty_like(w1p)
                _outer(lib, gh_all[:, i * Hs_:], 4 * Hs_, xs[0 if i < 2 else 1], K, world * B, Hs_, K, gW1)
                g = shape_grads[i]
                shape_grads[i] = (gW1, g[1], g[2], g[3])
                w1p._shasta_grad_is_global = True  # allreduce_gradients must not reduce it again
        elif lowrank and ghids[0] is not None:
            for i in range(4):  # one rank: the local factors as they are (the input rows lie T * F apart in the feature table)
                model.aug_shape[i][0].weight._shasta_grad_factors = (ghids[i], ghids[i].stride(0), S["feat"] if i < 2 else S["prev_feat"], T * F, B)

        if in_bwd and ghids[0] is not None:
            for i in (j for j in range(4) if need_dx[j]):  # (a matrix whose dx nobody needs keeps its factors for step())
                w1p = model.aug_shape[i][0].weight
                gin = dfeat if i < 2 else dprev_feat
                stepper.step_in_backward(w1p, w1p.__dict__.pop("_shasta_grad_factors"), ghids[i], ghids[i].stride(0), B, gin, T * F)

        # ---- gather (shasta.py:231-238) -> gradient of the two NHWC maps ----
        def gather_bwd(gtab, boxes):
            Bb, H_, W_, Cc = S["bev_shape"]
            dbev = torch.zeros(Bb, H_, W_, Cc, device=dev)
            x0, y0, vx, vy, stv = model.bev_extractor._geom()
            hip.check(lib.shasta_bev_gather_bwd_f32(hip.ptr(gtab), B, H_, W_, Cc, hip.ptr(boxes), N, boxes.shape[2], N * boxes.shape[2],
                                                    model.num_point, x0, y0, vx, vy, stv, F, T * F, hip.ptr(dbev), st()),
                      "shasta_bev_gather_bwd_f32")
            return dbev

        dbev = gather_bwd(dfeat, S["det_pre"]) if ctx.needs_input_grad[1] else None
        dprev_bev = gather_bwd(dprev_feat, S["prev"]) if ctx.needs_input_grad[2] else None

        # ---- gradients in the order of affinity_params(model) ----
        out = []
        for i in range(4):
            out += list(shape_grads[i])
        for gW, gb in fs_grads:
            out += [gW, gb]
        for i in range(4):
            out += list(box_grads[i])
        for gW, gb in fd_grads:
            out += [gW, gb]
        for gW, gb in rc_grads:
            out += [gW, gb]
        for gW, gb in aff_grads:
            out += [gW, gb]
        return (None, dbev, dprev_bev, None, None) + tuple(out)


def first_layer_plan(model, B, K):
    """How the backward treats the four aug_shape first-layer matrices ((K/64, K) each, K = N*F; 1 GB at N = 500) for a step of B local
    frame-pairs: (world, group, exchange, lowrank, stepper).
      exchange: data-parallel run - the ranks all-gather the rank-B FACTORS of dW1 = ghid^T x and every rank forms the averaged gradient
                (or hands the gathered factors to the optimizer) instead of all-reducing 4 x 1 GB;
      lowrank:  the gradient is not formed at all, its factors go to FusedAdam(lowrank_first_layers=model), which builds it in registers
                inside its pass (shasta_adam_lowrank_f32) - only while that optimizer is alive, and never with rank-LOCAL factors in a
                data-parallel run (model.low_rank_grad_exchange = False there means: dense gradient, averaged by allreduce_gradients);
      stepper:  the live FusedAdam, or None."""
    world, group = _exchange_world(model)
    exchange = world > 1 or bool(getattr(model, "_force_factor_exchange", False))  # the latter: single-rank test of the path
    opt_ref = getattr(model, "_lowrank_adam_opt", None)
    stepper = opt_ref() if opt_ref is not None else None
    if stepper is None and getattr(model, "lowrank_adam", False):
        model.lowrank_adam = False  # its optimizer is gone: .grad must come back, or the matrices would silently stop training
    lowrank = bool(getattr(model, "lowrank_adam", False)) and world * B <= 64 and K % 4 == 0
    if lowrank and _dist_world(model) > 1 and not exchange:
        lowrank = False
    if lowrank:
        for i in range(4):
            if "_shasta_grad_factors" in model.aug_shape[i][0].weight.__dict__:
                raise hip.ShastaHipError(
                    "FusedAdam(lowrank_first_layers=model): a second backward() before step() - the factors of the first-layer gradients "
                    "are handed over, not accumulated (no gradient accumulation or clipping with this option; build the optimizer "
                    "without it for those)")
    return world, group, exchange, lowrank, stepper


def _exchange_world(model):
    """(world size, group) of the low-rank gradient exchange; (1, None) when not running data-parallel or switched off
    with model.low_rank_grad_exchange = False."""
    import torch.distributed as dist
    if not getattr(model, "low_rank_grad_exchange", True) or not (dist.is_available() and dist.is_initialized()):
        return 1, None
    group = getattr(model, "grad_exchange_group", None)
    return dist.get_world_size(group), group


def _dist_world(model):
    """World size of the data-parallel job itself, whatever model.low_rank_grad_exchange says."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(getattr(model, "grad_exchange_group", None))


def _all_gather_rows(t, world, group=None):
    """(B, C) on every rank -> (world*B, C), rank-major (RCCL all_gather on the GPUs)."""
    import torch.distributed as dist
    t = t.contiguous()
    out = torch.empty(world * t.shape[0], t.shape[1], dtype=t.dtype, device=t.device)
    dist.all_gather(list(out.chunk(world, dim=0)), t, group=group)
    return out


def _check_equal_local_batch(B, world, group, device):
    """The factor exchange gathers equal-size chunks: every rank must bring the same number of frame-pairs (the reference's
    DistributedGroupSampler pads the epoch so that it does, sampler.py:177-196).  A ragged last batch would hang or corrupt the
    gather, so it is refused loudly."""
    import torch.distributed as dist
    sizes = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather(list(sizes.chunk(world)), torch.tensor([B], dtype=torch.int64, device=device), group=group)
    sizes = sizes.tolist()
    if any(s != B for s in sizes):
        raise hip.ShastaHipError("low-rank gradient exchange needs the same local batch on every rank, got %s: pad the sampler "
                                 "(shasta_amd.sampler.DistributedGroupSampler does) or set model.low_rank_grad_exchange = False" % sizes)


def affinity_params(model):
    """Trainable parameters of rows 6-16 in the order the backward returns their gradients."""
    ps = []
    for i in range(4):
        ps += [model.aug_shape[i][0].weight, model.aug_shape[i][0].bias, model.aug_shape[i][2].weight, model.aug_shape[i][2].bias]
    for k in (0, 2, 4, 6):
        ps += [model.fuse_shape[k].weight, model.fuse_shape[k].bias]
    for i in range(4):
        ps += [model.aug_dets[i][0].weight, model.aug_dets[i][0].bias, model.aug_dets[i][2].weight, model.aug_dets[i][2].bias]
    for k in (0, 2, 4):
        ps += [model.fuse_det[k].weight, model.fuse_det[k].bias]
    for k in (0, 2, 4):
        ps += [model.res_coeff[k].weight, model.res_coeff[k].bias]
    for k in (0, 2, 4, 6, 8, 10):
        ps += [model.aff[k].weight, model.aff[k].bias]
    return ps


def affinity_train(model, bev_nhwc, prev_bev_nhwc, det_boxes, prev_det_boxes):
    """Differentiable rows 4-16: returns (matched1, matched2) with autograd history to the affinity parameters and to the two
    NHWC BEV maps.  det_boxes[:, :, :2] is back-projected in place like the reference forward does."""
    return _AffinityTrainFn.apply(model, bev_nhwc, prev_bev_nhwc, det_boxes, prev_det_boxes, *affinity_params(model))


class _AffinityLossFn(torch.autograd.Function):
    """The loss below in two launches and its gradient in one (csrc/train.hip) instead of ~25 elementwise / reduction launches."""

    @staticmethod
    def forward(ctx, m1, m2, gt):
        lib = hip.load()
        B, N = m1.shape[0], m1.shape[1]
        m1, m2, gt = m1.contiguous(), m2.contiguous(), gt.contiguous()
        ws = torch.empty(4 * B * (N + 2), device=m1.device)
        sums = torch.empty(8, device=m1.device)
        hip.check(lib.shasta_affinity_loss_f32(hip.ptr(m1), hip.ptr(m2), hip.ptr(gt), B, N, hip.ptr(ws), hip.ptr(sums), hip.stream_ptr()),
                  "shasta_affinity_loss_f32")
        ctx.save_for_backward(m1, m2, gt, sums)
        return sums[4].clone()

    @staticmethod
    def backward(ctx, gloss):
        m1, m2, gt, sums = ctx.saved_tensors
        lib = hip.load()
        B, N = m1.shape[0], m1.shape[1]
        g1, g2 = torch.empty_like(m1), torch.empty_like(m2)
        gl = gloss.detach().float().reshape(1).contiguous()
        hip.check(lib.shasta_affinity_loss_bwd_f32(hip.ptr(m1), hip.ptr(m2), hip.ptr(gt), hip.ptr(sums), hip.ptr(gl), B, N, hip.ptr(g1), hip.ptr(g2),
                                                   hip.stream_ptr()), "shasta_affinity_loss_bwd_f32")
        return g1, g2, None


def affinity_loss(m1, m2, gt):
    """tools/nusc_shasta/train.py:200-211.  Device fp32 tensors of the shapes affinity_train returns: the fused kernels; anything else
    (the reference loop's own tensors, CPU tensors of the host-side tests): the same formula in torch operations."""
    N = m1.shape[1]
    if (m1.is_cuda and m1.dtype == m2.dtype == gt.dtype == torch.float32 and m1.dim() == 3 and tuple(m1.shape) == (m1.shape[0], N, N + 2)
            and tuple(m2.shape) == (m1.shape[0], N + 2, N) and tuple(gt.shape) == (m1.shape[0], N + 2, N + 2) and not gt.requires_grad):
        return _AffinityLossFn.apply(m1, m2, gt)
    gt1, gt2 = gt[:, :-2, :], gt[:, :, :-2]
    lf, lb = (gt1 * (-torch.log(m1 + 1e-10))).sum(), (gt2 * (-torch.log(m2 + 1e-10))).sum()
    if gt1.sum() > 0:  # (train.py:208-209: a direction without a ground-truth entry keeps its plain - zero - sum)
        lf = lf / gt1.sum()
    if gt2.sum() > 0:
        lb = lb / gt2.sum()
    return (lf + lb) / 2


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) of tools/nusc_shasta/train.py:147 with the update done by one
    HIP kernel per tensor (`shasta_adam_step_f32`): a single pass over p, g, m, v (28 bytes per parameter) instead of the
    unfused optimizer's seven.  Same param_groups keys (so OneCycleLR, which cycles `lr` and `betas`, train.py:172, drives
    it unchanged) and the same state names (`step`, `exp_avg`, `exp_avg_sq`)."""

    MULTI_MAX_NUMEL = 1 << 18  # tensors up to this size are updated together, 48 per launch

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lowrank_first_layers=None, in_backward=False,
                 capturable=False):
        """lowrank_first_layers = the Shasta model: its four aug_shape first-layer matrices (4 x 1 GB at N = 500) are updated straight from
        the FACTORS of their gradient (shasta_adam_lowrank_f32: 24 bytes per parameter instead of 36) - the backward then leaves their
        .grad None and hands the factors over on the parameter; same update, the gradient's sum over the step's frame-pairs in another
        order."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # in_backward (with lowrank_first_layers, steps of at most 16 frame-pairs per rank and 64 over all ranks): those four matrices take their Adam
        # update INSIDE loss.backward(), in the one pass over each that also forms the backward's dx = ghid W1 from the not-yet-updated
        # weights (the matrix is read once instead of twice); step() then updates everything else.  Same weights after the step as
        # without the option; for loops that are backward() -> step() like tools/nusc_shasta/train.py:213-215 (no gradient clipping or
        # accumulation over several backward passes, whose updates must wait for step()).
        self.in_backward = bool(in_backward) and lowrank_first_layers is not None
        # capturable: the step number, the learning rate and the betas reach the kernels through DEVICE memory (one tiny kernel per
        # optimizer step advances a device-side counter and turns {lr, beta1, beta2} into the bias-corrected factors), so that a whole
        # training step - forward, loss, backward, this update - can be captured into a hipGraph once and replayed (GraphedTrainStep):
        # launch arguments are frozen at capture time.  A scheduler keeps writing group["lr"] / group["betas"] on the host;
        # sync_hyper() copies them over (called by step() itself outside a capture, by GraphedTrainStep before every replay).  One
        # step counter per parameter group (every tensor of a group is stepped together, as in the reference's loop).
        self.capturable = bool(capturable)
        self._armed = {}  # id(group) -> the group's factors are prepared for the current optimizer step
        if lowrank_first_layers is not None:
            lowrank_first_layers.lowrank_adam = True
            import weakref
            lowrank_first_layers._lowrank_adam_opt = weakref.ref(self)

    def _state_of(self, p):
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        return st

    def _dev(self, group, device):
        d = group.get("_shasta_dev")
        if d is None or d["step"].device != device:
            d = group["_shasta_dev"] = dict(step=torch.zeros(1, dtype=torch.int32, device=device), hyper=torch.zeros(3, device=device),
                                            dyn=torch.zeros(4, device=device), host=None)
        return d

    def sync_hyper(self):
        """capturable: copy every group's lr / betas (the scheduler's host values) to the device; a no-op when nothing changed."""
        for group in self.param_groups:
            d = group.get("_shasta_dev")
            if d is None:
                continue
            host = (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]))
            if d["host"] != host:
                d["hyper"].copy_(torch.tensor(host, dtype=torch.float32), non_blocking=False)
                d["host"] = host

    def _dyn(self, group, device):
        """Device pointer of the group's step factors for the CURRENT optimizer step (None in the plain mode).  The first use in a step -
        inside the backward when the first layers step there, else in step() - advances the device-side counter."""
        if not self.capturable:
            return None
        d = self._dev(group, device)
        if not self._armed.get(id(group)):
            if d["host"] is None or not torch.cuda.is_current_stream_capturing():
                if torch.cuda.is_current_stream_capturing() and d["host"] is None:
                    raise hip.ShastaHipError("FusedAdam(capturable=True): run one eager step (or sync_hyper()) before capturing")
                self.sync_hyper()
            hip.check(hip.load().shasta_adam_prepare_f32(hip.ptr(d["step"]), hip.ptr(d["hyper"]), hip.ptr(d["dyn"]), hip.stream_ptr()),
                      "shasta_adam_prepare_f32")
            self._armed[id(group)] = True
        return hip.ptr(d["dyn"])

    @torch.no_grad()
    def step_in_backward(self, p, factors, gdx, ldgdx, rdx, y, ldy):
        """The Adam update of matrix p from the factors of its gradient, and y (+)= gdx . p (p before the update), in one pass."""
        group = next((g for g in self.param_groups if any(q is p for q in g["params"])), None)
        if group is None:
            raise hip.ShastaHipError("FusedAdam(in_backward=True): the matrix is not one of this optimizer's parameters")
        lib = hip.load()
        st = self._state_of(p)
        st["step"] = int(st["step"]) + 1
        G, ldg, X, ldx, R = factors
        H, K = p.shape
        nb = lib.shasta_adam_lowrank_dx_workspace_bytes(H, K, rdx)
        ws = torch.empty((nb + 3) // 4, device=p.device)
        b1, b2 = group["betas"]
        hip.check(lib.shasta_adam_lowrank_dx_f32(hip.ptr(p), hip.ptr(st["exp_avg"]), hip.ptr(st["exp_avg_sq"]), H, K, hip.ptr_view(G), ldg,
                                                 hip.ptr_view(X), ldx, R, hip.ptr_view(gdx), ldgdx, rdx, hip.ptr_view(y), ldy, 1, hip.ptr(ws), nb,
                                                 float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                 st["step"], self._dyn(group, p.device), hip.stream_ptr()), "shasta_adam_lowrank_dx_f32")
        torch.autograd.graph.increment_version(p)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = hip.load()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            small = {}  # step count -> the small tensors that share it: one launch for all of them (shasta_adam_multi_f32)
            for p in group["params"]:
                factors = p.__dict__.pop("_shasta_grad_factors", None)
                if p.grad is None and factors is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise hip.ShastaHipError("FusedAdam needs contiguous fp32 device parameters (no CPU path)")
                st = self._state_of(p)
                st["step"] = int(st["step"]) + 1
                if p.grad is None:  # a matrix whose gradient is G^T X: formed inside the pass
                    G, ldg, X, ldx, R = factors
                    hip.check(lib.shasta_adam_lowrank_f32(hip.ptr(p), hip.ptr(st["exp_avg"]), hip.ptr(st["exp_avg_sq"]), p.shape[0], p.shape[1],
                                                          hip.ptr_view(G), ldg, hip.ptr_view(X), ldx, R, float(group["lr"]), float(b1), float(b2),
                                                          float(group["eps"]), float(group["weight_decay"]), st["step"], self._dyn(group, p.device),
                                                          hip.stream_ptr()), "shasta_adam_lowrank_f32")
                    torch.autograd.graph.increment_version(p)
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if g.dtype != torch.float32 or not g.is_cuda:
                    raise hip.ShastaHipError("FusedAdam needs fp32 device gradients")
                if p.numel() <= self.MULTI_MAX_NUMEL:
                    small.setdefault(st["step"], []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
                    continue
                if g.data_ptr() % 16:
                    g = g.clone()
                hip.check(lib.shasta_adam_step_f32(hip.ptr(p), hip.ptr(g), hip.ptr(st["exp_avg"]), hip.ptr(st["exp_avg_sq"]), p.numel(),
                                                   float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                   float(group["weight_decay"]), st["step"], self._dyn(group, p.device), hip.stream_ptr()),
                          "shasta_adam_step_f32")
                # the kernel wrote through the raw pointer: tell torch (Shasta._ensure_packed and the conv-weight cache key
                # their packed copies on (data_ptr, _version); autograd's saved-tensor checks rely on it too)
                torch.autograd.graph.increment_version(p)
            for step_no, items in small.items():
                k = len(items)
                arr = lambda j: (C.c_void_p * k)(*[it[j].data_ptr() for it in items])  # noqa: E731
                hip.check(lib.shasta_adam_multi_f32(k, arr(0), arr(1), arr(2), arr(3), (C.c_long * k)(*[it[0].numel() for it in items]),
                                                    float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                                    step_no, self._dyn(group, items[0][0].device), hip.stream_ptr()), "shasta_adam_multi_f32")
                for it in items:
                    torch.autograd.graph.increment_version(it[0])
            self._armed[id(group)] = False  # the next optimizer step prepares its own factors
        return loss


class GraphedTrainStep:
    """One training step of tools/nusc_shasta/train.py:198-218 - forward (from the NHWC features, or from the neck outputs with
    shared_conv in train mode), the masked NLL, the HIP backward, FusedAdam(capturable=True) - captured ONCE into a hipGraph and
    replayed per step: the host's launch work (~240 launches at the car configuration) is out of the way.  Measured (tools/time_train.py
    --graph): no faster than the eager loop on this stack - 2.65 vs 2.71 ms at 16 frame pairs, 2.8 at 64 once the input copies are
    discounted - the step is bound by the GPU-side dispatch of its ~240 dependent small kernels (8 - 12 us apiece end to end), which a
    replay does not shorten; what helps is fewer kernels.  Kept for hosts whose Python is the slower side.  Inputs of every step must have the shapes of the ones given here (they are copied into static buffers); the
    scheduler keeps stepping on the host, its lr / betas are copied to the device before each replay.  One process, one GPU (a
    data-parallel step synchronises on the host to compare batch sizes and stays eager).

        step = GraphedTrainStep(model, opt, bev, pbev, det, prev, gt)        # three eager warm-up steps + the capture (they DO update the weights)
        for batch in loader: loss = step(*batch); scheduler.step()
    """

    def __init__(self, model, opt, bev, pbev, det_boxes, prev_det_boxes, gt, from_neck=False, warmup=3):
        if not getattr(opt, "capturable", False):
            raise hip.ShastaHipError("GraphedTrainStep needs FusedAdam(..., capturable=True): step number, lr and betas from device memory")
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise hip.ShastaHipError("GraphedTrainStep: single-process steps only")
        self.model, self.opt, self.from_neck = model, opt, from_neck
        self.static = [t.detach().clone() for t in (bev, pbev, det_boxes, prev_det_boxes, gt)]
        self._work = self.static[2].clone()  # the forward back-projects det_boxes in place: every step starts from a fresh copy

        def body():
            opt.zero_grad(set_to_none=True)
            sb, sp, sd, spv, sg = self.static
            self._work.copy_(sd)
            if from_neck:
                m1, m2, _ = model(dict(det_boxes=self._work, prev_det_boxes=spv, bev_map=sb, prev_bev_map=sp), train_mode=True)
            else:
                m1, m2 = affinity_train(model, sb, sp, self._work, spv)
            loss = affinity_loss(m1, m2, sg)
            loss.backward()
            opt.step()
            return loss.detach()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                body()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = body()
        self.steps = 0

    def __call__(self, bev, pbev, det_boxes, prev_det_boxes, gt):
        for s, t in zip(self.static, (bev, pbev, det_boxes, prev_det_boxes, gt)):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        self.opt.sync_hyper()
        self.graph.replay()
        self.steps += 1
        # the replayed kernels wrote the weights without the host noticing: whatever the host derived from them (packed copies, keyed
        # on version counters that only move at capture time) is dropped, so that an eager forward after training packs afresh
        self.model.invalidate_weights_cache()
        return self.loss


def allreduce_gradients(params, world_size=None, bucket_bytes=256 << 20, group=None):
    """Data-parallel gradient averaging over torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the
    CPU tests).  The four 1 GB aug_shape first-layer gradients were already averaged inside the backward by exchanging their
    rank-B factors (_AffinityTrainFn.backward) and are skipped.  The rest is packed into flat buckets (few, large
    collectives: the per-link bound of the point-to-point xGMI topology favours large messages), summed with one all_reduce
    per bucket and divided by the world size, like apex DDP does for the reference (tools/nusc_shasta/train.py:156)."""
    import torch.distributed as dist
    params = list(params)
    grads = []
    for p in params:
        if getattr(p, "_shasta_grad_is_global", False):  # already the average over the ranks (low-rank factor exchange)
            p._shasta_grad_is_global = False
            continue
        if p.grad is not None:
            grads.append(p.grad)
    if not (dist.is_available() and dist.is_initialized()):
        return
    world = world_size or dist.get_world_size(group)
    if world == 1:
        return
    i = 0
    while i < len(grads):
        j, nbytes = i, 0
        while j < len(grads) and (j == i or nbytes + grads[j].numel() * grads[j].element_size() <= bucket_bytes):
            nbytes += grads[j].numel() * grads[j].element_size()
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, group=group)
        flat.div_(world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        i = j
