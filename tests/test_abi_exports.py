"""The C-ABI library: loads, exports every symbol include/nnue_hip.h declares, binding table matches the
header, and argument validation rejects bad calls before anything is launched (so these calls are safe
without a GPU).  CPU only."""
import ctypes
import re

import pytest

from conftest import ROOT
from nnue_hip import lib

HEADER = (ROOT / "include" / "nnue_hip.h").read_text()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(nnue_[a-z0-9_]+)\s*\(", body)))


def test_library_is_built_in_tree():
    assert lib.LIB_PATH.exists(), "run __graft_entry__.build()"
    assert lib.LIB_PATH.parent == ROOT / "nnue-vision_amd" / "nnue_hip"


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 17
    raw = ctypes.CDLL(str(lib.LIB_PATH))
    for n in names:
        assert hasattr(raw, n), f"{n} declared in nnue_hip.h but not exported"
    assert sorted(lib.SIGNATURES) == names, "binding table and header disagree"
    assert lib.load().nnue_hip_abi_version() == lib.ABI_VERSION == int(re.search(r"NNUE_HIP_ABI_VERSION (\d+)", HEADER).group(1))


def test_header_cites_the_reference_for_each_entry_point():
    for n in declared_functions():
        if n in ("nnue_hip_abi_version", "nnue_hip_last_error", "nnue_ftm_uses_bf16") or n.endswith(("_scratch", "_scratch_bucketed", "_supported", "_list_tiles", "_offset", "_chunks", "_count", "_bytes")):  # size / capability queries
            continue
        m = re.search(r"/\*((?:(?!/\*).)*?)\*/\s*(?:int64_t[^;]*;\s*)*int\s+" + n + r"\(", HEADER, flags=re.S)
        assert m, f"{n}: no doc comment"
        assert re.search(r"(nnue|train|serialize|evaluate|datasets|loaders)\.py:\d+", m.group(1)), f"{n}: comment cites no reference file:line"


def test_argument_validation_returns_codes_without_launching():
    L = lib.load()
    # null pointers -> NNUE_E_ARG, message set
    assert L.nnue_ft_forward(0, 0, 0, 0, 0, 8, 2, 10, 256, 0, None) == -1
    assert b"null pointer" in L.nnue_hip_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    # non-positive sizes
    assert L.nnue_ft_forward(p, p, p, p, p, 0, 2, 10, 256, p, None) == -1
    assert L.nnue_conv3x3_forward(p, p, p, 0, 32, 32, 8, 3, None) == -1
    # ldb not a multiple of 64 / smaller than B -> NNUE_E_SHAPE
    assert L.nnue_ft_backward_weight(p, p, 100, 2, 10, 256, p, p, None) == -2
    assert L.nnue_ft_prepare(p, p, 128, 4, 10, p, p, p, p, p, 64, None) == -2
    # scratch too small -> NNUE_E_SCRATCH
    assert L.nnue_sgd_step(p, p, p, 1000, 0.1, 0.9, 0.0, 1.0, 1.0, 1, None, p, 4, None, 0, 0, None, None, None, 0, 0, 0, None, 0, None, None) == -4
    assert L.nnue_classifier_forward(p, 1, p, p, p, p, p, p, 0.0, 512, 1024, 128, 32, 10, p, p, p, p, 16, None) == -4
    # odd L1 with the pairwise block
    assert L.nnue_classifier_forward(p, 1, p, p, p, p, p, p, 0.0, 2, 7, 4, 4, 3, p, p, p, p, 1 << 20, None) == -2
    # momentum without a buffer
    assert L.nnue_sgd_step(p, p, 0, 10, 0.1, 0.9, 0.0, 1.0, 1.0, 1, None, p, 1 << 20, None, 0, 0, None, None, None, 0, 0, 0, None, 0, None, None) == -1


def test_scratch_queries():
    L = lib.load()
    assert L.nnue_sgd_scratch(956106) > 0
    assert L.nnue_classifier_scratch(512, 1024, 128, 32) >= 512 * 128 * 4
    assert L.nnue_classifier_scratch(0, 1024, 128, 32) == 0
    assert L.nnue_ste_conv_backward_scratch(512, 8, 11, 11) >= 8 * 28 * 4


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", tmp_path / "libnnue_hip.so")
    with pytest.raises(lib.NnueHipError, match="no CPU or eager fallback"):
        lib.load()


def test_feature_transformer_path_policy(monkeypatch):
    """mfma > bits > list by shape; a shape the product kernels cannot address falls back instead of failing."""
    monkeypatch.delenv("NNUE_FT_PATH", raising=False)
    assert lib.ft_path(800, 968, 1024, 512) == "mfma" and lib.ft_path(65536, 65536, 1024, 128) == "mfma"
    assert lib.ft_path(800, 27 * 5, 1024, 8) == "bits"        # map size not a multiple of 4
    assert lib.ft_path(800, 968, 100, 8) == "mfma" and lib.ft_path(800, 27 * 5, 100, 8) == "list"  # width the LDS gather kernels lack
    assert lib.ft_path(600000, 65536, 1024, 8) == "bits"      # table beyond 32-bit byte offsets
    assert lib.ft_path(800, 968, 1024, 600000) == "bits"      # batch beyond 32-bit byte offsets into the map
    monkeypatch.setenv("NNUE_FT_PATH", "bits")
    assert lib.ft_path(800, 968, 1024, 512) == "bits"
    monkeypatch.setenv("NNUE_FT_PATH", "list")
    assert lib.ft_path(800, 968, 1024, 512) == "list"
