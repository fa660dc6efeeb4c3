"""Resolution-cluster batch samplers: drop-in for the reference's ``src/utils/cluster_random_sampler.py`` plus the
rank-aligned variant data-parallel training on MI355X needs (BASELINE config 5).

``ImnetDataset`` groups its images into resolution clusters -- (384,512), (512,512), (512,384); datasets.py:331-335 -- and
exposes ``cluster_indices`` (one list of dataset indices per cluster).  ``ClusterRandomSampler`` yields a flat index
stream in which every consecutive run of ``batch_size`` indices comes from ONE cluster, so a ``DataLoader(batch_size=B,
sampler=...)`` collates same-shape batches (train.py:156-177).  Ragged tails are dropped, batches are shuffled at batch
level (cluster_random_sampler.py:31-50), ``oversampling_indices`` repeat items (:21-29).

Same RNG contract as the reference: the module-level ``random`` generator, consumed in the same order, so
``random.seed(s)`` reproduces the reference's index stream exactly (pinned by tests/golden/sampler.json).

``DistributedClusterSampler`` is new: with one process per GPU, ranks that drew different clusters in the same step would
run 3.92x vs 5.22x the 224^2 cost and the slow rank would hold the gradient all-reduce back
(x1.33 step time).  Here a global step is ``world_size`` batches of the SAME cluster, one per rank; every rank derives the
identical schedule from (seed, epoch) with a private generator -- no communication, equal lengths on all ranks.
"""
from __future__ import annotations

import random
from typing import Iterator, List, Sequence

from torch.utils.data.sampler import Sampler

__all__ = ["ClusterRandomSampler", "DistributedClusterSampler", "ProgressiveResize"]


def _full_batches(indices: Sequence[int], batch_size: int) -> List[List[int]]:
    n_full = len(indices) // batch_size
    return [list(indices[b * batch_size:(b + 1) * batch_size]) for b in range(n_full)]


def _expand_oversampled(indices: Sequence[int], repeats: Sequence[int]) -> List[int]:
    if len(repeats) != len(indices):
        raise AssertionError("oversampling_indices must match cluster_indices item for item")
    out: List[int] = []
    for idx, rep in zip(indices, repeats):
        out.extend([idx] * rep)
    return out


class ClusterRandomSampler(Sampler):
    """Single-process sampler with the reference's behaviour (cluster_random_sampler.py:4-55)."""

    def __init__(self, data_source, batch_size: int, shuffle: bool = True):
        self.data_source, self.batch_size, self.shuffle = data_source, int(batch_size), bool(shuffle)
        oversample = getattr(data_source, "oversampling_indices", None)
        per_cluster: List[List[List[int]]] = []
        for j, members in enumerate(data_source.cluster_indices):
            members = list(members)
            if oversample is not None:
                print("Oversampling initiated")
                members = _expand_oversampled(members, oversample[j])
                if self.shuffle:                      # one image must not dominate a batch
                    random.shuffle(members)
            batches = _full_batches(members, self.batch_size)
            if self.shuffle:
                random.shuffle(batches)
            per_cluster.append(batches)
        self.lst = [b for batches in per_cluster for b in batches]       # batch level from here on
        if self.shuffle:
            random.shuffle(self.lst)

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            random.shuffle(self.lst)
        return iter([i for b in self.lst for i in b])

    def __len__(self) -> int:
        return sum(len(b) for b in self.lst)


class DistributedClusterSampler(Sampler):
    """One process per GPU: step s of every rank draws a batch from the SAME cluster.

    ``cluster_of_step()`` lists the cluster id of every step of the current epoch (identical on all ranks), e.g. to look up
    the (H, W) the step will run at.  Call ``set_epoch(e)`` before each epoch, as with torch's DistributedSampler."""

    def __init__(self, data_source, batch_size: int, num_replicas: int = None, rank: int = None, shuffle: bool = True,
                 seed: int = 0):
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("pass num_replicas/rank or initialise torch.distributed first")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError("rank %d outside [0, %d)" % (rank, num_replicas))
        self.data_source, self.batch_size = data_source, int(batch_size)
        self.num_replicas, self.rank, self.shuffle, self.seed = int(num_replicas), int(rank), bool(shuffle), int(seed)
        self.epoch = 0
        self._oversample = getattr(data_source, "oversampling_indices", None)
        # schedule length is data-independent of the shuffle: full global steps per cluster
        self._steps_per_cluster = []
        for j, members in enumerate(data_source.cluster_indices):
            n = sum(self._oversample[j]) if self._oversample is not None else len(members)
            self._steps_per_cluster.append(n // (self.batch_size * self.num_replicas))

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _schedule(self):
        """[(cluster, [batch of rank 0, ..., batch of rank R-1])] for this epoch; identical on every rank."""
        rng = random.Random(self.seed * 1000003 + self.epoch)
        steps = []
        group = self.batch_size * self.num_replicas
        for j, members in enumerate(self.data_source.cluster_indices):
            members = list(members)
            if self._oversample is not None:
                members = _expand_oversampled(members, self._oversample[j])
            if self.shuffle:
                rng.shuffle(members)
            for s in range(len(members) // group):
                chunk = members[s * group:(s + 1) * group]
                steps.append((j, _full_batches(chunk, self.batch_size)))
        if self.shuffle:
            rng.shuffle(steps)
        return steps

    def cluster_of_step(self) -> List[int]:
        return [j for j, _ in self._schedule()]

    def __iter__(self) -> Iterator[int]:
        return iter([i for _, batches in self._schedule() for i in batches[self.rank]])

    def __len__(self) -> int:
        return sum(self._steps_per_cluster) * self.batch_size


class ProgressiveResize:
    """The progressive-resizing schedule of the reference's epoch loop (train.py:304-317): every ``epochs_grow_size`` epochs,
    while the dataset's ``size_ratio`` is still below 1.0, the ratio doubles and the batch size is divided by 4 (the loaders are
    rebuilt with the new values).  Pure host logic; the HIP engine keeps one compiled program per (batch, height, width), so a
    size switch costs one program build and nothing else (the kernels take H, W at run time).

        sched = ProgressiveResize(size_ratio=0.25, batch_size=256, epochs_grow_size=10)
        for epoch in range(epochs):
            ratio, bs, changed = sched.step(epoch)          # call at the top of the epoch, like train.py:306-317
            if changed: rebuild the loaders with size_ratio=ratio, batch_size=bs
    """

    def __init__(self, size_ratio: float, batch_size: int, epochs_grow_size: int):
        self.size_ratio, self.batch_size, self.epochs_grow_size = float(size_ratio), int(batch_size), int(epochs_grow_size)

    def step(self, epoch: int):
        changed = False
        if self.epochs_grow_size > 0 and (epoch + 1) % self.epochs_grow_size == 0 and self.size_ratio < 1.0:
            self.size_ratio *= 2                       # train.py:315
            self.batch_size = int(self.batch_size // 4)  # train.py:313
            changed = True
        return self.size_ratio, self.batch_size, changed

    def cluster_shapes(self, clusters=((384, 512), (512, 512), (512, 384))):
        """(H, W) of every resolution cluster (datasets.py:331-335) at the current size_ratio"""
        return [(int(h * self.size_ratio), int(w * self.size_ratio)) for h, w in clusters]
