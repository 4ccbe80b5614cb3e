"""Throughput of evaluate.evaluate_model on a GPU-resident dataset.  Prints one JSON object."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
import evaluate, nnue  # noqa: E402
from nnue_hip.input_pipeline import GpuImageDataset  # noqa: E402

torch.manual_seed(0)
model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 1024, 128, 32, num_classes=10).cuda().eval()
rng = np.random.RandomState(0)
ds = GpuImageDataset(rng.randint(0, 256, (10240, 32, 32, 3), dtype=np.uint8), rng.randint(0, 10, 10240))
loader = ds.loader(512)
evaluate.evaluate_model(model, loader)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss, metrics = evaluate.evaluate_model(model, loader)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"images": len(ds), "batch": 512, "seconds_per_pass": dt, "images_per_s": len(ds) / dt, "ms_per_batch": dt / len(loader) * 1e3,
                  "loss": loss, "acc": metrics["acc"]}))
