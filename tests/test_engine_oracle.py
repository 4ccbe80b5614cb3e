"""oracle/nnue_engine_oracle.py (numpy restatement of the C++ engine's evaluate_logits) against outputs of the REAL
engine: tests/golden/engine_cases.npz was produced by oracle/_ref/nnue_inference, the reference's own sources
compiled here (recipe: tests/golden/make_golden_engine.py).  CPU only."""
import json

import numpy as np
import pytest

import nnue_engine_oracle as eo
from conftest import GOLDEN


def cases():
    z = np.load(GOLDEN / "engine_cases.npz")
    return z, json.loads(str(z["index"]))


def test_oracle_reproduces_the_reference_engine_exactly():
    z, index = cases()
    assert sum(c["count"] for c in index) == 32
    for k, c in enumerate(index):
        m = eo.load_nnue(GOLDEN / c["model"])
        for i in range(c["count"]):
            logits, density = eo.evaluate_logits(m, z[f"case{k}/images"][i], c["h"], c["w"])
            # the engine prints with 10 decimals; its logits are multiples of 1/64, exactly representable
            assert np.array_equal(logits.astype(np.float64), z[f"case{k}/logits"][i]), (c, i)
            assert abs(float(density) - float(z[f"case{k}/density"][i])) < 5e-10, (c, i)


def test_engine_stride_rule_differs_from_training():
    assert eo.conv_stride(32, 10) == 4 and (32 - 1) // (10 - 1) == 3  # ceil vs floor: 8x8 map inside a 10x10 grid
    assert eo.conv_stride(28, 10) == 3 and eo.conv_stride(96, 10) == 11 and eo.conv_stride(7, 1) == 7


def test_loader_rejections(tmp_path):
    good = (GOLDEN / "nnue_tiny4x4.nnue").read_bytes()
    for name, data, msg in (("magic", b"XNUE" + good[4:], "magic"), ("version", good[:4] + b"\x03\x00\x00\x00" + good[8:], "version"),
                            ("tail", good + b"\x00", "trailing")):
        p = tmp_path / f"{name}.nnue"
        p.write_bytes(data)
        with pytest.raises(ValueError, match=msg):
            eo.load_nnue(p)
    m = eo.load_nnue(GOLDEN / "nnue_c1arch.nnue")
    assert (m["num_features"], m["l1"], m["l2"], m["l3"], m["grid"], m["oc"], m["stacks"][0]["classes"]) == (800, 64, 32, 8, 10, 8, 10)
