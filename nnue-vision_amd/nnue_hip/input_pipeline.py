"""GPU-resident input pipeline.

The reference feeds training through torchvision + albumentations + DataLoader workers
(data/datasets.py:173-372, data/loaders.py:95-120): per image, on the CPU.  At the step rates this build
reaches (a CIFAR-10 epoch is ~17 ms of GPU time) that path cannot keep up by orders of magnitude, so the
dataset (50 000 x 32 x 32 x 3 uint8 = 154 MB for CIFAR) lives in HBM and one kernel per batch gathers,
augments ("light" policy), normalises and transposes it straight into the trainer's input slot.

Datasets themselves (download / decode) are out of scope: construct ``GpuImageDataset`` from any uint8
``[N,H,W,3]`` array and integer labels.
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch

from . import lib


class GpuImageDataset:
    def __init__(self, images_u8, labels, device=None, augment: bool = False, seed: int = 0,
                 num_classes: Optional[int] = None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        images_u8 = torch.as_tensor(images_u8)
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3:
            raise ValueError(f"images: expected uint8 [N,H,W,3], got {images_u8.dtype} {tuple(images_u8.shape)}")
        labels = torch.as_tensor(labels).to(torch.int64).reshape(-1)
        if labels.numel() != images_u8.shape[0]:
            raise ValueError("labels: one label per image expected")
        # validated once, on the host, outside every timed region: inside the step -1 is the padding sentinel of a
        # short last batch and the loss kernel ignores any out-of-range label, so a corrupt label must not get that far
        self.label_range = (int(labels.min()), int(labels.max())) if labels.numel() else (0, 0)
        if self.label_range[0] < 0 or (num_classes is not None and self.label_range[1] >= num_classes):
            raise ValueError(f"labels must lie in [0, {num_classes if num_classes is not None else 'C'}): "
                             f"found {self.label_range[0]} .. {self.label_range[1]}")
        self.images = images_u8.contiguous().to(device)
        self.labels = labels.to(device)
        self.augment, self.seed = augment, seed
        self.step = 0  # advances with every batch: a fresh draw for every visit of a sample

    def __len__(self) -> int:
        return self.images.shape[0]

    @property
    def image_hw(self) -> Tuple[int, int]:
        return int(self.images.shape[1]), int(self.images.shape[2])

    def batch(self, indices: torch.Tensor, out: Optional[torch.Tensor] = None, labels_out: Optional[torch.Tensor] = None):
        """One normalised (and, if enabled, augmented) batch for the given dataset indices (device int64)."""
        self.step += 1
        return lib.load_batch(self.images, self.labels, indices.to(self.images.device), self.augment, self.seed, self.step,
                              out=out, labels_out=labels_out)

    def loader(self, batch_size: int, shuffle: bool = False, drop_last: bool = False,
               generator: Optional[torch.Generator] = None) -> "GpuLoader":
        return GpuLoader(self, batch_size, shuffle, drop_last, generator)


class GpuLoader:
    """Iterable of (images float32 [b,3,H,W], labels int64 [b]) device batches; re-iterable (one pass = one epoch);
    ``len()`` = number of batches, like a DataLoader.  Everything stays on the device; nothing synchronises."""

    def __init__(self, dataset: GpuImageDataset, batch_size: int, shuffle: bool, drop_last: bool, generator):
        self.dataset, self.batch_size, self.shuffle, self.drop_last, self.generator = dataset, batch_size, shuffle, drop_last, generator

    def __len__(self) -> int:
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        for idx in self.index_batches():
            yield self.dataset.batch(idx)

    def index_batches(self):
        """The epoch's index tensors (device), one per batch -- for callers that place batches themselves."""
        n, dev = len(self.dataset), self.dataset.images.device
        if self.shuffle:
            order = torch.randperm(n, generator=self.generator).to(dev) if self.generator is not None \
                else torch.randperm(n, device=dev)
        else:
            order = torch.arange(n, device=dev)
        return [order[i * self.batch_size:(i + 1) * self.batch_size] for i in range(len(self))]


def train_epoch(trainer, loader: GpuLoader):
    """One epoch of ``trainer.step`` fed in place: each batch is written by the pipeline kernel straight into a trainer
    input slot (no staging copy), on the trainer's stream, then the step's graph is replayed on that slot.  With more than
    one input slot, whole rounds of full batches go through ``trainer.step_many``: one pipeline kernel per slot, then ONE
    graph replay for the round's steps (no gap between graph launches).  Returns (sum of per-step mean losses as a
    device scalar, number of steps); the short last batch of an epoch goes through ``trainer.step(images, labels)``
    (padded, see NnueTrainer.step).

    A side-stream double buffer was measured and is slower here (0.218 vs 0.177 ms per C2 step): the 9 us kernel is
    not worth two event waits per step on a host-launch-bound loop."""
    ds, slots = loader.dataset, len(trainer.inputs)
    if ds.label_range[1] >= trainer.C:
        raise ValueError(f"dataset labels reach {ds.label_range[1]} but the model has {trainer.C} classes")
    total = torch.zeros((), dtype=torch.float32, device=trainer.dev)
    batches = loader.index_batches()
    n, i, round_slots = len(batches), 0, tuple(range(slots))
    while i < n:
        if slots > 1 and i + slots <= n and batches[i + slots - 1].numel() == trainer.B:  # only the last batch can be short
            for s in round_slots:
                ds.batch(batches[i + s], out=trainer.inputs[s][0], labels_out=trainer.inputs[s][1])
            total += trainer.step_many(round_slots).sum()
            i += slots
            continue
        s, idx = i % slots, batches[i]
        if idx.numel() == trainer.B:
            ds.batch(idx, out=trainer.inputs[s][0], labels_out=trainer.inputs[s][1])
            total += trainer.step(slot=s)
        else:
            images, labels = ds.batch(idx)
            count = None
            if trainer.dp.world > 1:
                # short last batch with ranks: the exact mean needs the number of real samples over ALL ranks (every
                # rank's loader yields the same number of batches; a rank may hold fewer or no samples in the last one)
                import torch.distributed as dist
                cnt = torch.tensor([int(idx.numel())], dtype=torch.int64, device=trainer.dev if trainer.dp.backend == "nccl" else "cpu")
                dist.all_reduce(cnt, group=trainer.dp.group)
                count = int(cnt.item())
            total += trainer.step(images, labels, slot=s, global_count=count)
        i += 1
    return total, n
