"""Throughput of the engine's integer inference on the GPU beside the reference's way of getting the
same numbers: one `nnue_inference` subprocess per image (evaluate.py:143-176), timed on this host with oracle/_ref.
Prints one JSON object:  python tools/bench_engine.py > gpurun_out/engine_bench.json"""
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "nnue-vision_amd"))

import nnue  # noqa: E402
import serialize  # noqa: E402
from nnue_hip.engine import EngineModel  # noqa: E402


def main():
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 1024, 128, 32, num_classes=10)
    res = {"model": "C2 architecture (800 -> 1024/128/32 -> 10), 32x32 images"}
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / "m.nnue"
        serialize.serialize_model(model, path)
        engine = EngineModel.load(path)
        images = torch.randn(4096, 3, 32, 32).cuda()
        for b in (512, 4096):
            x = images[:b]
            for _ in range(3):
                engine.evaluate_logits(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                engine.evaluate_logits(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            res[f"gpu_batch{b}_ms"] = dt * 1e3
            res[f"gpu_batch{b}_images_per_s"] = b / dt
        exe = ROOT / "oracle" / "_ref" / "nnue_inference"
        if exe.exists():
            img = Path(tmp) / "img.bin"
            images[0].cpu().numpy().tofile(img)
            subprocess.run([str(exe), str(path), str(img), "32", "32"], capture_output=True)
            t0 = time.perf_counter()
            n = 30
            for _ in range(n):
                subprocess.run([str(exe), str(path), str(img), "32", "32"], capture_output=True, text=True, timeout=10)
            dt = (time.perf_counter() - t0) / n
            res["reference_subprocess_ms_per_image"] = dt * 1e3
            res["reference_subprocess_images_per_s"] = 1.0 / dt
    print(json.dumps(res))


if __name__ == "__main__":
    main()
