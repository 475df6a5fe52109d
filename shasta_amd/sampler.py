"""Index plan of the reference's data-parallel training sampler (`DistributedGroupSampler`, det3d/datasets/loader/sampler.py:139-223,
built by det3d/datasets/loader/build_loader.py:34-35 when `dist=True`), in this package's own form: the whole epoch is ONE array
computed by `epoch_plan` - (world, samples per rank) - and the sampler object only hands out its row.

What the plan guarantees (and what the rank-B factor exchange of the training backward relies on): every rank gets the same number
of samples, a per-GPU batch never mixes aspect-ratio groups (`dataset.flag`), and all ranks derive the same plan from the epoch
number alone - no communication.

RNG-order contract (the only thing that ties the plan to the reference's index lists, tests/golden/sampler_golden.json): one
`torch.Generator` seeded with the epoch; it is consumed by exactly one `torch.randperm(len(group))` per NON-EMPTY group in increasing
group id, then by one `torch.randperm(number of per-GPU batches)`."""
import numpy as np
import torch
from torch.utils.data.sampler import Sampler


def epoch_plan(flag, samples_per_gpu, world, epoch):
    """(world, n) int64 array: row r = the dataset indices rank r visits in this epoch, in order.

    Each group is permuted, padded to a multiple of samples_per_gpu * world with its own leading entries, cut into per-GPU batches;
    the batches of all groups are then permuted together and dealt to the ranks as contiguous runs."""
    flag = np.asarray(flag)
    gen = torch.Generator()
    gen.manual_seed(int(epoch))
    per_step = samples_per_gpu * world
    # members of every group in dataset order: one stable sort instead of one scan per group
    by_group = np.argsort(flag, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(np.bincount(flag))])
    blocks = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        size = int(hi - lo)
        if size == 0:
            continue
        order = by_group[lo:hi][torch.randperm(size, generator=gen).numpy()]
        pad = -(-size // per_step) * per_step - size
        # one repetition of the group is all the padding there is (the reference's length assertion fires in the same case)
        assert pad <= size, "group of %d samples cannot be padded to %d" % (size, size + pad)
        blocks.append(np.concatenate([order, order[:pad]]))
    if not blocks:
        return np.zeros((world, 0), np.int64)
    batches = np.concatenate(blocks).reshape(-1, samples_per_gpu)
    batches = batches[torch.randperm(len(batches), generator=gen).numpy()]
    return batches.reshape(world, -1).astype(np.int64)


def samples_per_rank(flag, samples_per_gpu, world):
    per_step = samples_per_gpu * world
    return int(sum(-(-int(n) // per_step) * samples_per_gpu for n in np.bincount(np.asarray(flag))))


class DistributedGroupSampler(Sampler):
    """Same constructor, `set_epoch` and iteration protocol as the reference class; rank and world default to torch.distributed's."""

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            live = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if live else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if live else 0) if rank is None else rank
        self.dataset, self.samples_per_gpu, self.num_replicas, self.rank, self.epoch = dataset, samples_per_gpu, num_replicas, rank, 0
        # a data set without aspect-ratio groups is one group
        self.flag = np.asarray(dataset.flag) if hasattr(dataset, "flag") else np.zeros(len(dataset), np.uint8)
        self.num_samples = samples_per_rank(self.flag, samples_per_gpu, num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        return iter(epoch_plan(self.flag, self.samples_per_gpu, self.num_replicas, self.epoch)[self.rank].tolist())
