"""Disjoint per-rank shards of an epoch of (drug, target) pairs for batch-sharded data parallelism.

The reference trains on one GPU with ``DataLoader(shuffle=True)`` (proj/dti/train_joint_attn.py:259-262) and has no
distributed sampler; this is the counterpart of ``torch.utils.data.DistributedSampler`` for the N-rank step of
jova_amd.train.Trainer: one seeded permutation per epoch, identical on every rank; the tail that does not fill a whole
global batch (world x batch pairs) is dropped, so every rank runs the same number of equally sized steps and
mean-of-shard-means equals the global mean; rank r takes positions r, r + world, r + 2 world, ... of the permutation
(DistributedSampler's interleaving), cut into batches of ``batch_size``.
"""
import numpy as np


class ShardSampler(object):
    def __init__(self, n_pairs, world_size=1, rank=0, batch_size=128, seed=0, shuffle=True, drop_last=True):
        if not 0 <= rank < world_size:
            raise ValueError("rank %d outside world of %d" % (rank, world_size))
        if not drop_last:
            raise NotImplementedError("ShardSampler keeps shards equal: drop_last=True only")
        self.n_pairs, self.world, self.rank = int(n_pairs), int(world_size), int(rank)
        self.batch_size, self.seed, self.shuffle = int(batch_size), int(seed), bool(shuffle)
        self.epoch = 0
        self.steps_per_epoch = self.n_pairs // (self.world * self.batch_size)
        if self.steps_per_epoch == 0:
            raise ValueError("%d pairs do not fill one global batch of %d x %d" % (n_pairs, world_size, batch_size))

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.steps_per_epoch

    def indices(self):
        """This rank's pair indices of the current epoch, [steps_per_epoch, batch_size]."""
        if self.shuffle:
            perm = np.random.RandomState((self.seed + self.epoch) & 0x7FFFFFFF).permutation(self.n_pairs)
        else:
            perm = np.arange(self.n_pairs)
        used = perm[:self.steps_per_epoch * self.world * self.batch_size]
        mine = used[self.rank::self.world]
        return mine.reshape(self.steps_per_epoch, self.batch_size)

    def __iter__(self):
        return iter(self.indices())
