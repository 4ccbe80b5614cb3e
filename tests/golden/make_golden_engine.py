"""Generates tests/golden/engine_cases.npz: inputs and printed outputs of the REFERENCE C++ engine
(`oracle/_ref/nnue_inference`, compiled by oracle/Makefile from /root/reference/engine where the sources lie) on
`.nnue` files written by this repo's serialize.py from the committed golden model states (byte-identical to the
reference writer's output, tests/test_serialize_bytes.py).

Run in the build container only (needs the reference tree for `make -C oracle`):
    python tests/golden/make_golden_engine.py
The fixture holds data only: the flat float32 image buffers handed to the engine, H, W, the logits and the density it
printed (std::fixed, 10 decimals), and the name of the `.nnue` fixture each case used.
"""
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent
ROOT = GOLDEN.parent.parent
EXE = ROOT / "oracle" / "_ref" / "nnue_inference"

# (.nnue fixture, H, W, number of images, image scale): sizes the engine handles without overrunning its grid buffer
CASES = [
    ("nnue_tiny4x4.nnue", 32, 32, 5, 1.0),
    ("nnue_tiny4x4.nnue", 17, 17, 3, 1.5),
    ("nnue_grid8.nnue", 32, 32, 5, 1.0),
    ("nnue_grid8.nnue", 64, 64, 3, 2.0),
    ("nnue_saturated.nnue", 32, 32, 4, 1.0),
    ("nnue_c1arch.nnue", 32, 32, 6, 1.0),
    ("nnue_c1arch.nnue", 96, 96, 3, 1.0),
    ("nnue_c1arch.nnue", 28, 28, 3, 3.0),
]


def run_engine(model: Path, image: np.ndarray, h: int, w: int):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        image.astype(np.float32).tofile(f.name)
        res = subprocess.run([str(EXE), str(model), f.name, str(h), str(w)], capture_output=True, text=True, timeout=30)
    if res.returncode != 0:
        raise RuntimeError(f"engine failed: {res.stderr}")
    parts = res.stdout.strip().split(",")
    return np.array([float(x) for x in parts[:-1]], dtype=np.float64), float(parts[-1])


def main():
    if not EXE.exists():
        sys.exit(f"{EXE} missing: run `make -C oracle` in the build container first")
    out, index = {}, []
    rng = np.random.RandomState(20251004)
    for k, (name, h, w, count, scale) in enumerate(CASES):
        images = (rng.randn(count, 3 * h * w) * scale).astype(np.float32)
        logits, density = [], []
        for img in images:
            lg, dn = run_engine(GOLDEN / name, img, h, w)
            logits.append(lg)
            density.append(dn)
        out[f"case{k}/images"] = images
        out[f"case{k}/logits"] = np.stack(logits)
        out[f"case{k}/density"] = np.array(density, dtype=np.float64)
        index.append({"model": name, "h": h, "w": w, "count": count})
        print(name, h, w, "logits[0] =", logits[0][:4], "density", density[:3])
    out["index"] = np.array(json.dumps(index))
    np.savez_compressed(GOLDEN / "engine_cases.npz", **out)
    print("wrote", GOLDEN / "engine_cases.npz")


if __name__ == "__main__":
    main()
