"""Throughput of the GPU input pipeline alone and feeding the C2 training step.
Prints one JSON object; run on the GPU box:  python tools/bench_input_pipeline.py > gpurun_out/input_pipeline.json"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))

import nnue  # noqa: E402
from nnue_hip.input_pipeline import GpuImageDataset, train_epoch  # noqa: E402
from nnue_hip.trainer import NnueTrainer  # noqa: E402


def timed(fn, iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    n, b = 50000, 512  # CIFAR-10 train split shape, BASELINE configs[1] batch
    rng = np.random.RandomState(0)
    ds_plain = GpuImageDataset(rng.randint(0, 256, (n, 32, 32, 3), dtype=np.uint8), rng.randint(0, 10, n))
    ds_aug = GpuImageDataset(ds_plain.images, ds_plain.labels, augment=True, seed=1)
    order = torch.randperm(n, device="cuda")
    batches = [order[i * b:(i + 1) * b] for i in range(n // b)]
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 1024, 128, 32, num_classes=10).cuda()
    tr = NnueTrainer(model, b, (32, 32), lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0, input_slots=8)
    out, lab = tr.inputs[0]
    res = {"dataset": [n, 32, 32, 3], "batch": b, "input_slots": 8}
    for name, ds in (("plain", ds_plain), ("light_aug", ds_aug)):
        fn = lambda i: ds.batch(batches[i % len(batches)], out=out, labels_out=lab)
        timed(fn, 50)
        dt = timed(fn, 500)
        res[f"load_batch_{name}_us"] = dt * 1e6
        res[f"load_batch_{name}_images_per_s"] = b / dt
        res[f"load_batch_{name}_GBps"] = b * 32 * 32 * 3 * 5 / dt / 1e9  # 1 B read + 4 B written per value

    def step_only(i):
        tr.step(slot=i & 1)

    def step_fed(i):
        s = i & 1
        ds_aug.batch(batches[i % len(batches)], out=tr.inputs[s][0], labels_out=tr.inputs[s][1])
        tr.step(slot=s)

    for name, fn in (("step_resident_input", step_only), ("step_fed_by_pipeline", step_fed)):
        timed(fn, 50)
        dt = timed(fn, 500)
        res[f"{name}_ms"] = dt * 1e3
        res[f"{name}_images_per_s"] = b / dt
    loader = ds_aug.loader(b, shuffle=True, drop_last=True)
    train_epoch(tr, loader)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    epochs = 5
    for _ in range(epochs):
        _, steps = train_epoch(tr, loader)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (epochs * steps)
    res["train_epoch_ms"] = dt * 1e3
    res["train_epoch_images_per_s"] = b / dt
    res["epoch_seconds_cifar10"] = dt * steps
    print(json.dumps(res))


if __name__ == "__main__":
    main()
