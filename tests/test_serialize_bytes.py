"""The .nnue writer against bytes produced by the reference's serialize_model (golden files and
sha256), plus load acceptance by the reference C++ engine (oracle/_ref, when built).  CPU only."""
import hashlib
import json
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

import nnue
import nnue_oracle as orc
import serialize
from conftest import GOLDEN, ROOT, golden_model, load_npz

ENGINE = ROOT / "oracle" / "_ref" / "nnue_inference"


def build_model(cfg, state=None, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    m = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"],
                  num_classes=cfg["classes"], input_size=cfg["input_size"])
    if state is not None:
        m.load_state_dict(state)
    return m


def blob(model, tmp_path, name="m.nnue"):
    path = tmp_path / name
    serialize.serialize_model(model, path)
    return path.read_bytes()


@pytest.mark.parametrize("name", ("tiny4x4", "grid8", "c1arch"))
def test_bytes_identical_to_reference(name, tmp_path, nnue_index):
    cfg, params, _, _ = golden_model(name)
    got = blob(build_model(cfg, params), tmp_path)
    want = (GOLDEN / f"nnue_{name}.nnue").read_bytes()
    assert hashlib.sha256(want).hexdigest() == nnue_index[f"nnue_{name}.nnue"]["sha256"]
    assert len(got) == len(want) == orc.nnue_file_size(cfg["grid"] ** 2 * cfg["fps"], cfg["fps"], cfg["l1"], cfg["l2"], cfg["l3"], cfg["classes"])
    assert got == want


def test_saturated_weights_clip_and_clamp(tmp_path, nnue_index):
    state = {k: torch.from_numpy(v) for k, v in load_npz("nnue_saturated_state.npz").items()}
    cfg = nnue_index["nnue_saturated.nnue"]["cfg"]
    m = build_model(cfg, state)
    got = blob(m, tmp_path)
    assert got == (GOLDEN / "nnue_saturated.nnue").read_bytes()
    assert float(m.input.weight.abs().max()) <= 1.0 and not m.training  # serialisation mutates, like the reference


def test_c2_architecture_hash(tmp_path, nnue_index):
    e = nnue_index["c2arch_seed0"]
    got = blob(build_model(dict(e["cfg"]), seed=e["seed"]), tmp_path)
    assert len(got) == e["size"] == 2836856
    assert hashlib.sha256(got).hexdigest() == e["sha256"]


@pytest.mark.parametrize("thr", (-0.5, 0.0, 0.5, 1.0))
def test_threshold_sweep_hashes(thr, tmp_path, nnue_index):
    # the reference's own serialisation test sweeps these thresholds (tests/test_model.py:499-545)
    e = nnue_index[f"tiny4x4_thr{thr}"]
    m = build_model(e["cfg"], seed=e["seed"])
    with torch.no_grad():
        m.visual_threshold.fill_(thr)
    got = blob(m, tmp_path)
    assert hashlib.sha256(got).hexdigest() == e["sha256"]
    assert struct.unpack_from("<f", got, 36)[0] == np.float32(thr)


def test_header_fields_and_determinism(tmp_path):
    cfg, params, _, _ = golden_model("grid8")
    a = blob(build_model(cfg, params), tmp_path, "a.nnue")
    b = blob(build_model(cfg, params), tmp_path, "b.nnue")
    assert a == b
    assert a[:4] == b"NNUE"
    version, f, l1, l2, l3, buckets = struct.unpack_from("<6I", a, 4)
    assert (version, f, l1, l2, l3, buckets) == (2, 256, 64, 4, 8, 1)
    score, one, thr = struct.unpack_from("<3f", a, 28)
    assert (score, one) == (600.0, 127.0) and abs(thr - 0.1) < 1e-7
    layer_type, scale, oc, ic, kh, kw = struct.unpack_from("<IfIIII", a, 40)
    assert (layer_type, scale, oc, ic, kh, kw) == (0, 64.0, 4, 3, 3, 3)


def test_checkpoint_round_trip(tmp_path):
    cfg, params, _, _ = golden_model("c1arch")
    torch.save(params, tmp_path / "bare.pt")
    torch.save({"state_dict": params}, tmp_path / "wrapped.pt")
    for name in ("bare.pt", "wrapped.pt"):
        assert serialize.detect_model_type(tmp_path / name) == "nnue"
        m = serialize.load_model_from_checkpoint(tmp_path / name)
        assert (m.feature_set.grid_size, m.feature_set.num_features_per_square) == (10, 8)
        assert (m.l1_size, m.l2_size, m.l3_size, m.num_classes) == (64, 32, 8, 10)
        assert blob(m, tmp_path, name + ".nnue") == (GOLDEN / "nnue_c1arch.nnue").read_bytes()
    fs, l1, l2, l3, c = serialize.infer_architecture_from_state_dict(params)
    assert (fs.num_features, l1, l2, l3, c) == (800, 64, 32, 8, 10)
    with pytest.raises(ValueError):
        serialize.infer_architecture_from_state_dict({"foo": torch.zeros(1)})


def test_missing_metadata_is_rejected(tmp_path):
    with pytest.raises(ValueError, match="Missing required NNUE metadata"):
        with open(tmp_path / "x", "wb") as f:
            serialize.write_nnue_header(f, {"L1": 1})


@pytest.mark.skipif(not ENGINE.exists(), reason="oracle/_ref/nnue_inference not built (make -C oracle)")
@pytest.mark.parametrize("name", ("tiny4x4", "grid8", "c1arch"))
def test_reference_engine_loads_our_file(name, tmp_path):
    """Acceptance: engine/nnue_inference.cpp (compiled from the reference sources) loads the file we wrote
    and prints C logits + density (engine/nnue_inference.cpp:56-62)."""
    cfg, params, _, _ = golden_model(name)
    path = tmp_path / "m.nnue"
    serialize.serialize_model(build_model(cfg, params), path)
    img = tmp_path / "img.bin"
    np.random.RandomState(1).rand(32 * 32 * 3).astype(np.float32).tofile(img)
    outs = []
    for model_file in (path, GOLDEN / f"nnue_{name}.nnue"):
        r = subprocess.run([str(ENGINE), str(model_file), str(img), "32", "32"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        outs.append([float(v) for v in r.stdout.strip().split(",")])
    assert len(outs[0]) == cfg["classes"] + 1
    assert outs[0] == outs[1]  # same bytes -> same engine output as the reference's own file


@pytest.mark.skipif(not ENGINE.exists(), reason="oracle/_ref/nnue_inference not built (make -C oracle)")
def test_reference_engine_rejects_corrupt_header(tmp_path):
    cfg, params, _, _ = golden_model("tiny4x4")
    data = bytearray(blob(build_model(cfg, params), tmp_path))
    data[4] = 9  # version != 2 (engine/src/nnue_engine.cpp:560-565)
    bad = tmp_path / "bad.nnue"
    bad.write_bytes(bytes(data))
    img = tmp_path / "img.bin"
    np.zeros(32 * 32 * 3, dtype=np.float32).tofile(img)
    r = subprocess.run([str(ENGINE), str(bad), str(img), "32", "32"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0


def bucketed_model(k=8):
    torch.manual_seed(5)
    return nnue.NNUE(nnue.GridFeatureSet(10, 8), 64, 32, 8, num_classes=10, num_ls_buckets=k)


def test_bucketed_file_layout(tmp_path):
    """K layer-stack records behind a header that says K (what engine/src/nnue_engine.cpp:619-635 loops over); everything
    before the stacks, and stack k itself, are the bytes a K = 1 model with bucket k's weights gives."""
    import nnue_engine_oracle as eo
    import nnue_oracle as orc
    m = bucketed_model()
    data = blob(m, tmp_path)
    assert len(data) == orc.nnue_file_size(800, 8, 64, 32, 8, 10, buckets=8)
    assert int.from_bytes(data[24:28], "little") == 8
    (tmp_path / "k8.nnue").write_bytes(data)
    parsed = eo.load_nnue(tmp_path / "k8.nnue")
    assert parsed["buckets"] == 8 and len(parsed["stacks"]) == 8
    stack_bytes = (len(data) - orc.nnue_file_size(800, 8, 64, 32, 8, 10, buckets=0))
    assert stack_bytes % 8 == 0
    per = stack_bytes // 8
    head = len(data) - stack_bytes
    for k in (0, 3, 7):
        one = nnue.NNUE(nnue.GridFeatureSet(10, 8), 64, 32, 8, num_classes=10)
        sd = {key: (v[k] if key.startswith("classifier.") else v) for key, v in m.state_dict().items()}
        one.load_state_dict(sd)
        single = blob(one, tmp_path)
        assert single[28:head] == data[28:head] and single[:24] == data[:24]
        assert single[head:] == data[head + k * per: head + (k + 1) * per], k
    # checkpoint -> model keeps K
    torch.save(m.state_dict(), tmp_path / "k8.pt")
    again = serialize.load_model_from_checkpoint(tmp_path / "k8.pt")
    assert again.num_ls_buckets == 8 and blob(again, tmp_path) == data


@pytest.mark.skipif(not ENGINE.exists(), reason="oracle/_ref/nnue_inference not built (make -C oracle)")
def test_reference_engine_loads_the_bucketed_file(tmp_path):
    """The reference engine reads all K stacks and evaluates bucket 0 (engine/nnue_inference.cpp:44): same output as for
    the single-stack file holding bucket 0's weights."""
    m = bucketed_model()
    path = tmp_path / "k8.nnue"
    serialize.serialize_model(m, path)
    one = nnue.NNUE(nnue.GridFeatureSet(10, 8), 64, 32, 8, num_classes=10)
    one.load_state_dict({key: (v[0] if key.startswith("classifier.") else v) for key, v in m.state_dict().items()})
    serialize.serialize_model(one, tmp_path / "k1.nnue")
    img = tmp_path / "img.bin"
    np.random.RandomState(2).rand(32 * 32 * 3).astype(np.float32).tofile(img)
    outs = []
    for f in (path, tmp_path / "k1.nnue"):
        r = subprocess.run([str(ENGINE), str(f), str(img), "32", "32"], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1] and len(outs[0].split(",")) == 11
    # a truncated last stack is rejected by the loader
    (tmp_path / "short.nnue").write_bytes(path.read_bytes()[:-40])
    r = subprocess.run([str(ENGINE), str(tmp_path / "short.nnue"), str(img), "32", "32"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0
