"""K0 in train() mode: `relu(bn(conv3x3(map))) -> NHWC` with BATCH statistics and its backward, hand-written (csrc/shared_conv_train.hip).

The reference trains `shared_conv` (det3d/models/tracker/shasta.py:42-47, applied :223-228 - once on the current neck output, once on
the previous one): tools/nusc_shasta/train.py:183-191 freezes children 1, 2 (backbone, neck) only and keeps every BatchNorm in train
mode, the optimizer holds `shared_conv.0.{weight,bias}` and `shared_conv.1.{weight,bias}`.  This module is that piece of autograd:

  forward   conv + bias (the implicit-GEMM kernels of the inference operator on a RAW pack: no BatchNorm folded in, no ReLU) -> batch
            mean / variance per BatchNorm call (float64 accumulation; merged over the ranks for a `sync_bn.SyncBatchNorm`) -> running
            statistics updated like nn.BatchNorm2d -> normalise + affine + ReLU, NHWC
  backward  dgamma, dbeta, dbias, and dweight = an implicit GEMM over all pixels of both maps on the fp16 matrix path (three fp16 piece
            products per fp32 product, fp32 accumulation, fixed summation order).  No input gradient: the producer of the maps is frozen.

No CPU path.  A map that requires grad (somebody trains the neck) or an eval-mode BatchNorm under autograd stay on the module's own
nn.Sequential (Shasta.shared_conv_nhwc)."""
import ctypes as C

import torch

from . import hip
from .sync_bn import SyncBatchNorm, _world


def supported(model, bev_map):
    """The hand-written train-mode path serves this call: BatchNorm in train mode, fp32 device maps that need no gradient, a map width the
    weight-gradient kernel holds in LDS."""
    conv, bn = model.shared_conv[0], model.shared_conv[1]
    if not bn.training or bev_map.requires_grad or not bev_map.is_cuda or conv.out_channels != 64 or not bn.affine:
        return False
    if conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.stride != (1, 1) or conv.bias is None:
        return False
    return bool(hip.load().shasta_conv_train_supported(conv.in_channels, bev_map.shape[2], bev_map.shape[3]))


class _Raw:
    """The raw pack of a model's conv weights (conv + bias, no BN, no ReLU) for the forward kernels, cached on the module."""

    @staticmethod
    def get(model, dev, f16, cin_pad):
        conv = model.shared_conv[0]
        key = (conv.weight.data_ptr(), conv.weight._version, conv.bias.data_ptr(), conv.bias._version, str(dev), f16, cin_pad)
        cache = getattr(model, "_conv_raw", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        lib = hip.load()
        w = conv.weight.detach().contiguous()
        if cin_pad != conv.in_channels:  # zero channels add exactly
            wp = torch.zeros(64, cin_pad, 3, 3, device=dev)
            wp[:, :conv.in_channels] = w
            w = wp
        if f16:
            nbytes = lib.shasta_shared_conv_f16x2_packed_bytes(cin_pad)
            packed = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
            hip.check(lib.shasta_shared_conv_pack_raw_f16x2(hip.ptr(w), hip.ptr(conv.bias.detach()), cin_pad, hip.ptr(packed), nbytes,
                                                            hip.stream_ptr()), "shasta_shared_conv_pack_raw_f16x2")
        else:
            nbytes = lib.shasta_shared_conv_packed_bytes(cin_pad)
            packed = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=dev)
            hip.check(lib.shasta_shared_conv_pack_raw_f32(hip.ptr(w), hip.ptr(conv.bias.detach()), cin_pad, hip.ptr(packed), nbytes,
                                                          hip.stream_ptr()), "shasta_shared_conv_pack_raw_f32")
        model._conv_raw = (key, (packed, nbytes))
        return packed, nbytes


def _pad_channels(x, cin_pad):
    if x.shape[1] == cin_pad:
        return x
    xp = torch.zeros(x.shape[0], cin_pad, x.shape[2], x.shape[3], device=x.device)
    xp[:, :x.shape[1]] = x
    return xp


def _raw_conv(model, x, xp):
    """conv + bias of both maps, NHWC, and the image maxima (2B uint32 bit patterns: current maps, then previous maps)."""
    lib = hip.load()
    conv = model.shared_conv[0]
    dev = x.device
    B, cin, H, W = x.shape
    y, yp = torch.empty(B, H, W, 64, device=dev), torch.empty(B, H, W, 64, device=dev)
    c16 = (cin + 15) // 16 * 16
    if model.arithmetic in ("f16x2", "f16grid") and lib.shasta_shared_conv_f16x2_supported(c16, H, W):
        packed, nbytes = _Raw.get(model, dev, True, c16)
        wsb = lib.shasta_shared_conv_multi_workspace_bytes(B)
        ws = torch.empty((wsb + 3) // 4, dtype=torch.int32, device=dev)
        a, b = (C.c_void_p * 1)(y.data_ptr()), (C.c_void_p * 1)(yp.data_ptr())
        xa, xb = _pad_channels(x, c16), _pad_channels(xp, c16)  # (named: a temporary freed before the launch would hand its block to the next)
        hip.check(lib.shasta_shared_conv_multi_f32(hip.ptr(xa), hip.ptr(xb), B, c16, H, W, hip.ptr(packed),
                                                   (nbytes + 255) // 256 * 256, 1, a, b, hip.ptr(ws), ws.numel() * 4, hip.stream_ptr()),
                  "shasta_shared_conv_multi_f32 (raw)")
        return y, yp, ws[:2 * B]
    c8 = (cin + 7) // 8 * 8
    packed, _ = _Raw.get(model, dev, False, c8)
    xa, xb = _pad_channels(x, c8), _pad_channels(xp, c8)
    hip.check(lib.shasta_shared_conv_f32(hip.ptr(xa), hip.ptr(xb), B, c8, H, W, hip.ptr(packed), hip.ptr(y),
                                         hip.ptr(yp), hip.stream_ptr()), "shasta_shared_conv_f32 (raw)")
    xmax = torch.cat([x.abs().amax(dim=(1, 2, 3)), xp.abs().amax(dim=(1, 2, 3))]).contiguous().view(torch.int32)
    return y, yp, xmax


def _batch_stats(bn, y, ws):
    """mean_m2 (128,) of the whole batch of this BatchNorm call and its pixel count; merged over the ranks for a synchronised BatchNorm
    exactly as sync_bn._SyncBNFn merges them (per-rank mean / M2 / count, Chan's formula)."""
    lib = hip.load()
    M = y.numel() // 64
    mm = torch.empty(128, device=y.device)
    hip.check(lib.shasta_bn_stats_f32(hip.ptr(y), M, hip.ptr(mm), hip.ptr(ws), ws.numel() * 8, hip.stream_ptr()), "shasta_bn_stats_f32")
    group = getattr(bn, "process_group", None)
    world = _world(group) if isinstance(bn, SyncBatchNorm) else 1
    if world == 1:
        return mm, float(M), 1, None
    import torch.distributed as dist
    mine = torch.cat([mm, mm.new_full((1,), float(M))])
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    st = torch.stack(parts)
    ns = st[:, -1:]
    n = ns.sum()
    mean = (st[:, :64] * ns).sum(0) / n
    m2 = (st[:, 64:128] + ns * (st[:, :64] - mean).square()).sum(0)
    return torch.cat([mean, m2]).contiguous(), float(n.item()), world, group


class _SharedConvTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, xp, weight, bias, gamma, beta):
        lib = hip.load()
        bn = model.shared_conv[1]
        dev = x.device
        B, cin, H, W = x.shape
        y, yp, xmax = _raw_conv(model, x, xp)
        ws = torch.empty(lib.shasta_bn_workspace_bytes() // 8, dtype=torch.float64, device=dev)
        outs, stats, counts = [], [], []
        group, world = None, 1
        g_, b_ = gamma.detach().contiguous(), beta.detach().contiguous()
        for yy in (y, yp):  # two BatchNorm calls, current map first (shasta.py:223-228)
            mm, n, world, group = _batch_stats(bn, yy, ws)
            track = bn.track_running_stats and bn.running_mean is not None
            if track and bn.momentum is None:
                mom = 1.0 / (int(bn.num_batches_tracked) + 1)  # cumulative average: needs the count on the host
            else:
                mom = 0.0 if bn.momentum is None else float(bn.momentum)
            stat = torch.empty(128, device=dev)
            hip.check(lib.shasta_bn_finalize_f32(hip.ptr(mm), n, float(bn.eps), mom, hip.ptr(stat), hip.ptr(bn.running_mean) if track else None,
                                                 hip.ptr(bn.running_var) if track else None, C.c_void_p(bn.num_batches_tracked.data_ptr()) if track else None,
                                                 hip.stream_ptr()), "shasta_bn_finalize_f32")
            if track:
                for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
                    torch.autograd.graph.increment_version(t)
            out = torch.empty_like(yy)
            hip.check(lib.shasta_bn_relu_apply_f32(hip.ptr(yy), yy.numel() // 64, hip.ptr(stat), hip.ptr(g_), hip.ptr(b_), hip.ptr(out),
                                                   hip.stream_ptr()), "shasta_bn_relu_apply_f32")
            outs.append(out)
            stats.append(stat)
            counts.append(n)
        ctx.save_for_backward(x, xp, y, yp, stats[0], stats[1], g_, b_, xmax.clone())
        ctx.counts, ctx.group, ctx.world = counts, group, world
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, g, gp):
        lib = hip.load()
        x, xp, y, yp, st0, st1, gamma, beta, xmax = ctx.saved_tensors
        dev = x.device
        B, cin, H, W = x.shape
        nimg = 2 * B
        ws = torch.empty(lib.shasta_bn_workspace_bytes() // 8, dtype=torch.float64, device=dev)
        dyb = lib.shasta_conv_dy_bytes(nimg, H, W)
        dy = torch.empty((dyb + 3) // 4, dtype=torch.int32, device=dev)
        edy = torch.empty(2, 64, device=dev)
        dbias = torch.empty(64, device=dev)
        dgamma, dbeta = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        for i, (yy, gg, st) in enumerate(((y, g, st0), (yp, gp, st1))):
            gg = gg.contiguous()
            M = yy.numel() // 64
            sums = torch.empty(256, device=dev)
            hip.check(lib.shasta_bn_relu_bwd_reduce_f32(hip.ptr(yy), hip.ptr(gg), M, hip.ptr(st), hip.ptr(gamma), hip.ptr(beta), hip.ptr(sums),
                                                        hip.ptr(ws), ws.numel() * 8, hip.stream_ptr()), "shasta_bn_relu_bwd_reduce_f32")
            glob = sums
            if ctx.world > 1:  # the means of g' and g' xhat are those of the whole batch (sync_bn._SyncBNFn.backward)
                import torch.distributed as dist
                glob = sums[:128].clone()
                dist.all_reduce(glob, group=ctx.group)
            hip.check(lib.shasta_bn_relu_bwd_dy_f16x2(hip.ptr(yy), hip.ptr(gg), B, i * B, nimg, H, W, hip.ptr(st), hip.ptr(gamma), hip.ptr(beta),
                                                      hip.ptr(glob), hip.ptr(sums), ctx.counts[i], hip.ptr(dy), dy.numel() * 4, hip.ptr(edy[i]),
                                                      hip.ptr(dbias), i, hip.stream_ptr()), "shasta_bn_relu_bwd_dy_f16x2")
            dbeta += sums[:64]       # this rank's sums: the data-parallel averaging reduces them with every other gradient
            dgamma += sums[64:128]
        dw = torch.empty(64, cin, 3, 3, device=dev)
        wsb = lib.shasta_conv_wgrad_workspace_bytes(nimg, cin, H, W)
        wws = torch.empty((wsb + 3) // 4, dtype=torch.float32, device=dev)
        hip.check(lib.shasta_conv_wgrad_f16x2(hip.ptr(x), hip.ptr(xp), B, cin, H, W, hip.ptr(xmax), hip.ptr(dy), hip.ptr(edy), hip.ptr(dw),
                                              hip.ptr(wws), wws.numel() * 4, hip.stream_ptr()), "shasta_conv_wgrad_f16x2")
        return None, None, None, dw, dbias, dgamma, dbeta


def shared_conv_train(model, bev_map, prev_bev_map):
    """(out, out_prev), both (B, H, W, 64) NHWC, of model.shared_conv in train() mode; differentiable w.r.t. the four shared_conv
    parameters when autograd is on."""
    conv, bn = model.shared_conv[0], model.shared_conv[1]
    x = bev_map.detach().float().contiguous()
    xp = prev_bev_map.detach().float().contiguous()
    if x.shape != xp.shape:
        raise ValueError("bev_map and prev_bev_map must have the same shape")
    return _SharedConvTrainFn.apply(model, x, xp, conv.weight, conv.bias, bn.weight, bn.bias)
