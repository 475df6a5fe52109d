"""`DistributedGroupSampler` of the reference's data-parallel training (det3d/datasets/loader/sampler.py:139-223, built by
det3d/datasets/loader/build_loader.py:34-35 when `dist=True`): every epoch a deterministic, epoch-seeded permutation inside each
aspect-ratio group (`dataset.flag`), every group padded to a multiple of samples_per_gpu * num_replicas by repeating its first
indices, whole per-GPU batches shuffled, rank r takes the r-th contiguous slice - so every rank sees the same number of samples
(what the rank-B factor exchange of the training backward relies on) and a batch never mixes groups.
No communication: all ranks derive the same permutation from the epoch."""
import math

import numpy as np
import torch
from torch.utils.data.sampler import Sampler


class DistributedGroupSampler(Sampler):
    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            inited = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if inited else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if inited else 0) if rank is None else rank
        self.dataset = dataset
        self.samples_per_gpu = samples_per_gpu
        self.num_replicas = num_replicas
        self.rank = rank
        self.epoch = 0
        # the reference requires dataset.flag (0 / 1 by aspect ratio); a data set without groups is one group
        self.flag = np.asarray(dataset.flag if hasattr(dataset, "flag") else np.zeros(len(dataset), np.uint8))
        self.group_sizes = np.bincount(self.flag)
        per = self.samples_per_gpu * self.num_replicas
        self.num_samples = sum(int(math.ceil(int(n) / per)) * self.samples_per_gpu for n in self.group_sizes)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        per = self.samples_per_gpu * self.num_replicas
        indices = []
        for i, size in enumerate(self.group_sizes):
            if size > 0:
                idx = np.where(self.flag == i)[0]
                idx = idx[list(torch.randperm(int(size), generator=g))].tolist()
                extra = int(math.ceil(int(size) / per)) * per - len(idx)
                idx += idx[:extra]
                indices += idx
        # like the reference: a group smaller than half its padded size cannot be padded by one self-concatenation
        assert len(indices) == self.total_size
        spg = self.samples_per_gpu
        indices = [indices[j] for i in list(torch.randperm(len(indices) // spg, generator=g)) for j in range(i * spg, (i + 1) * spg)]
        offset = self.num_samples * self.rank
        indices = indices[offset:offset + self.num_samples]
        assert len(indices) == self.num_samples
        return iter(indices)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
