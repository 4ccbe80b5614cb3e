"""Shared test plumbing.

* registers the ``gpu`` marker (tests that need a real MI355X);
* puts the product package directory (``nnue-vision_amd/``: drop-in ``nnue`` and
  ``serialize`` modules + the ``nnue_hip`` host layer) and ``oracle/`` on sys.path;
* golden-vector loaders (fixtures generated from the real reference by
  ``tests/golden/make_golden.py``; the reference itself is never read by a test).
"""
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "nnue-vision_amd"
GOLDEN = ROOT / "tests" / "golden"
for p in (str(PKG), str(ROOT / "oracle"), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MODEL_CASES = ("tiny4x4", "grid8", "c1arch", "tiny96", "odd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    return torch.cuda.is_available()


def load_npz(name):
    with np.load(GOLDEN / name) as z:
        return {k: z[k] for k in z.files}


def golden_model(name):
    """Returns (cfg, params{str: tensor}, grads{str: tensor}, data{str: tensor})."""
    z = load_npz(f"model_{name}.npz")
    cfg = json.loads(str(z["cfg"]))
    params = {k[6:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state/")}
    grads = {k[5:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("grad/")}
    data = {k: torch.from_numpy(np.asarray(v)) for k, v in z.items()
            if "/" not in k and k != "cfg"}
    return cfg, params, grads, data


@pytest.fixture(scope="session")
def nnue_index():
    return json.loads((GOLDEN / "nnue_index.json").read_text())


# Tolerances (stated once, used everywhere):
#   logits / activations: |d| <= 1e-4 * max(1, |ref|)       
#   gradients:            |d| <= 1e-4 * max|ref grad| per tensor (hard parts)
def assert_close_logits(got, ref, what="logits", rtol=1e-4):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    bound = rtol * torch.clamp(ref.abs(), min=1.0)
    bad = (got - ref).abs() > bound
    assert not bool(bad.any()), f"{what}: max err {float((got - ref).abs().max()):.3e}, {int(bad.sum())} outside tol"


def assert_close_grad(got, ref, what="grad", rtol=1e-4):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = max(float(ref.abs().max()), 1e-12)
    err = float((got - ref).abs().max())
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (ratio {err / scale:.2e})"
