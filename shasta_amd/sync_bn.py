"""Synchronised BatchNorm for the data-parallel training of `shared_conv.1` (BASELINE config 5).

The reference converts every BatchNorm of the model with `apex.parallel.convert_syncbn_model` before wrapping it in apex DDP
(tools/nusc_shasta/train.py:155-156): in train() mode the statistics are those of the GLOBAL batch (all ranks), so N ranks with
B/N frame pairs each normalise exactly like one process with B.  `SyncBatchNorm2d` does the same over torch.distributed
("nccl" = RCCL on the GPUs, "gloo" in the CPU tests; torch.nn.SyncBatchNorm refuses CPU tensors): forward all-reduces
[sum, sum of squares, count] per channel (one small collective: 2C+1 floats), backward all-reduces [sum dy, sum dy*xhat].
Counts are exchanged, so ranks may hold different numbers of pixels.  Same parameter / buffer names as nn.BatchNorm2d
(`weight, bias, running_mean, running_var, num_batches_tracked`): state_dict keys are unchanged.  The structure follows the
all-gather / all-reduce pairing of the reference's own NaiveSyncBatchNorm (det3d/models/utils/norm.py:9-56)."""
import torch
import torch.distributed as dist
import torch.nn as nn


def _world(group):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class _SyncBNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        xs = x.transpose(0, 1).reshape(C, -1)
        stat = torch.cat([xs.sum(1), (xs * xs).sum(1), torch.tensor([float(xs.shape[1])], device=x.device, dtype=x.dtype)])
        if _world(group) > 1:
            dist.all_reduce(stat, group=group)
        n = stat[-1]
        mean = stat[:C] / n
        var = (stat[C:2 * C] / n - mean * mean).clamp_min(0.0)  # biased, as BatchNorm normalises
        invstd = torch.rsqrt(var + eps)
        xhat = (x - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.group = group
        ctx.mark_non_differentiable(mean, var, n)
        return xhat * weight.view(1, C, 1, 1) + bias.view(1, C, 1, 1), mean, var, n

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, weight, invstd, n = ctx.saved_tensors
        C = gy.shape[1]
        gys = gy.transpose(0, 1).reshape(C, -1)
        xh = xhat.transpose(0, 1).reshape(C, -1)
        gbias, gweight = gys.sum(1), (gys * xh).sum(1)  # local sums: the data-parallel gradient averaging reduces them later
        red = torch.cat([gbias, gweight])
        if _world(ctx.group) > 1:
            dist.all_reduce(red, group=ctx.group)
        sum_dy, sum_dy_xhat = red[:C] / n, red[C:] / n
        gx = (gy - sum_dy.view(1, C, 1, 1) - xhat * sum_dy_xhat.view(1, C, 1, 1)) * (weight * invstd).view(1, C, 1, 1)
        return gx, gweight, gbias, None, None


class SyncBatchNorm2d(nn.BatchNorm2d):
    def __init__(self, *a, process_group=None, **k):
        super().__init__(*a, **k)
        self.process_group = process_group

    def forward(self, x):
        if not self.training or _world(self.process_group) == 1:
            return super().forward(x)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
        with torch.no_grad():  # running statistics as nn.BatchNorm2d keeps them: unbiased variance, momentum update
            m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked + 1)
            self.num_batches_tracked += 1
            self.running_mean.mul_(1 - m).add_(mean, alpha=m)
            self.running_var.mul_(1 - m).add_(var * (n / (n - 1).clamp_min(1.0)), alpha=m)
        return y


def convert_syncbn_model(module, process_group=None):
    """apex.parallel.convert_syncbn_model as train.py:155 applies it: every nn.BatchNorm2d of `module` (for the affinity network:
    `shared_conv.1`) is replaced in place by a SyncBatchNorm2d that shares its parameters and buffers."""
    for name, child in list(module.named_children()):
        if isinstance(child, nn.BatchNorm2d) and not isinstance(child, SyncBatchNorm2d):
            new = SyncBatchNorm2d(child.num_features, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                  track_running_stats=child.track_running_stats, process_group=process_group)
            new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var, new.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            new.train(child.training)
            setattr(module, name, new)
        else:
            convert_syncbn_model(child, process_group)
    return module
