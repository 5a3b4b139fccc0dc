"""`EnlargedSampler` -- the data-parallel partition of the reference (data/data_sampler.py:6-49):
every rank draws the SAME permutation of `ceil(len * ratio / world) * world` slots from a generator seeded with
the epoch, maps the slots back onto the dataset with `% len`, and keeps the slots `rank, rank + world, ...`.
The ranks' index lists are therefore disjoint slices of one shuffled, `ratio`-times enlarged epoch."""
import torch
from torch.utils.data.sampler import Sampler


class EnlargedSampler(Sampler):
    def __init__(self, dataset, num_replicas, rank, ratio=1):
        self.dataset = dataset
        self.num_replicas = int(num_replicas)
        self.rank = int(rank)
        self.epoch = 0
        n = len(dataset) * ratio
        self.num_samples = -(-n // self.num_replicas) if float(n).is_integer() else int(-(-n // self.num_replicas))
        self.num_samples = int(self.num_samples)
        self.total_size = self.num_samples * self.num_replicas

    def indices(self, epoch=None):
        """this rank's dataset indices for `epoch` (default: the epoch last given to set_epoch)"""
        gen = torch.Generator()
        gen.manual_seed(self.epoch if epoch is None else int(epoch))
        slots = torch.randperm(self.total_size, generator=gen)
        mine = slots[self.rank::self.num_replicas] % len(self.dataset)
        assert mine.numel() == self.num_samples
        return mine.tolist()

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
