"""Synchronised BatchNorm for the data-parallel training of `shared_conv.1` (BASELINE config 5).

The reference converts every BatchNorm of the model with `apex.parallel.convert_syncbn_model` before wrapping it in apex DDP
(tools/nusc_shasta/train.py:155-156): in train() mode the statistics are those of the GLOBAL batch (all ranks), so N ranks with
B/N frame pairs each normalise exactly like one process with B.  `SyncBatchNorm2d` does the same over torch.distributed
("nccl" = RCCL on the GPUs, "gloo" in the CPU tests; torch.nn.SyncBatchNorm refuses CPU tensors): forward all-gathers
each rank's [mean, sum of squared deviations, count] per channel (one small collective: 2C+1 floats per rank) and merges them as
partial variances are merged (no E[x^2] - mean^2 cancellation), backward all-reduces [sum dy, sum dy*xhat].
Counts are exchanged, so ranks may hold different numbers of pixels.  Same parameter / buffer names as nn.BatchNorm2d
(`weight, bias, running_mean, running_var, num_batches_tracked`): state_dict keys are unchanged.  The structure follows the
all-gather / all-reduce pairing of the reference's own NaiveSyncBatchNorm (det3d/models/utils/norm.py:9-56)."""
import torch
import torch.distributed as dist
import torch.nn as nn


def _world(group):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _bshape(x):
    return [1, x.shape[1]] + [1] * (x.dim() - 2)


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        xs = x.transpose(0, 1).reshape(C, -1)
        # per-rank (count, mean, M2 = sum of squared deviations from the rank's own mean), merged over ranks as Chan et al. merge
        # partial variances - what apex's welford kernels do; E[x^2] - mean^2 would cancel for |mean| >> std
        cnt = x.new_full((1,), float(xs.shape[1]))
        mean_l = xs.mean(1)
        m2_l = (xs - mean_l.unsqueeze(1)).square().sum(1)
        world = _world(group)
        if world > 1:
            mine = torch.cat([mean_l, m2_l, cnt])
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            st = torch.stack(parts)                                  # (world, 2C + 1)
            ns = st[:, -1:]
            n = ns.sum()
            mean = (st[:, :C] * ns).sum(0) / n
            m2 = (st[:, C:2 * C] + ns * (st[:, :C] - mean).square()).sum(0)
        else:
            n, mean, m2 = cnt[0], mean_l, m2_l
        var = m2 / n  # biased, as BatchNorm normalises
        invstd = torch.rsqrt(var + eps)
        sh = _bshape(x)
        xhat = (x - mean.view(sh)) * invstd.view(sh)
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.group = group
        ctx.mark_non_differentiable(mean, var, n)
        y = xhat
        if weight is not None:
            y = y * weight.view(sh)
        if bias is not None:
            y = y + bias.view(sh)
        return y, mean, var, n

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, weight, invstd, n = ctx.saved_tensors
        C = gy.shape[1]
        sh = _bshape(gy)
        gys = gy.transpose(0, 1).reshape(C, -1)
        xh = xhat.transpose(0, 1).reshape(C, -1)
        gbias, gweight = gys.sum(1), (gys * xh).sum(1)  # local sums: the data-parallel gradient averaging reduces them later
        red = torch.cat([gbias, gweight])
        if _world(ctx.group) > 1:
            dist.all_reduce(red, group=ctx.group)
        sum_dy, sum_dy_xhat = red[:C] / n, red[C:] / n
        scale = invstd if weight is None else weight * invstd
        gx = (gy - sum_dy.view(sh) - xhat * sum_dy_xhat.view(sh)) * scale.view(sh)
        return gx, (gweight if weight is not None else None), (gbias if ctx.needs_input_grad[2] else None), None, None


class SyncBatchNorm(nn.modules.batchnorm._BatchNorm):
    """BatchNorm over (N, C, *) whose train()-mode statistics are those of all ranks of `process_group`."""

    def __init__(self, *a, process_group=None, **k):
        super().__init__(*a, **k)
        self.process_group = process_group

    def _check_input_dim(self, x):
        if x.dim() < 2:
            raise ValueError("expected at least 2D input (got %dD input)" % x.dim())

    def forward(self, x):
        # eval mode with running statistics, or a single process: exactly the parent class
        if _world(self.process_group) == 1 or not (self.training or self.running_mean is None):
            return super().forward(x)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.training and self.track_running_stats and self.running_mean is not None:
            with torch.no_grad():  # running statistics as nn.BatchNorm keeps them: unbiased variance, momentum update
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp_min(1.0)), alpha=m)
        return y


class SyncBatchNorm2d(SyncBatchNorm):
    """The 4-D case (`shared_conv.1`); kept as its own name for checkpoints / configs that refer to it."""


def convert_syncbn_model(module, process_group=None):
    """apex.parallel.convert_syncbn_model as train.py:155 applies it: every BatchNorm (1d / 2d / 3d) of `module` - for the affinity
    network: `shared_conv.1` - is replaced by a synchronised one that shares its parameters and buffers; a root module that is itself
    a BatchNorm is returned converted."""
    if isinstance(module, nn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNorm):
        cls = SyncBatchNorm2d if isinstance(module, nn.BatchNorm2d) else SyncBatchNorm
        new = cls(module.num_features, eps=module.eps, momentum=module.momentum, affine=module.affine,
                  track_running_stats=module.track_running_stats, process_group=process_group)
        if module.affine:
            new.weight, new.bias = module.weight, module.bias
        if module.track_running_stats:
            new.running_mean, new.running_var, new.num_batches_tracked = module.running_mean, module.running_var, module.num_batches_tracked
        new.train(module.training)
        return new
    for name, child in list(module.named_children()):
        setattr(module, name, convert_syncbn_model(child, process_group))
    return module
