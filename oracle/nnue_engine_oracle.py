"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy, integer arithmetic) of the reference C++ engine's
`NNUEEvaluator::evaluate_logits` (engine/src/nnue_engine.cpp:704-734) on a `.nnue` file.

Pinned against outputs of the real engine: `oracle/_ref/nnue_inference` (the reference's own sources compiled by
oracle/Makefile) was run on the committed `.nnue` fixtures; tests/golden/make_golden_engine.py holds the recipe and
tests/golden/engine_cases.npz the inputs and printed outputs.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

The engine has three behaviours that are easy to miss and are kept here exactly as they are:
  * the image buffer is indexed as HWC (`input[(h*W + w)*3 + c]`, nnue_engine.cpp:66) although the Python side
    hands it the bytes of a CHW tensor (evaluate.py:150-153) -- the buffer is taken as given, flat;
  * the conv weights are the OIHW bytes of the file read as [oc][kh][kw][ic] (nnue_engine.cpp:67, :121);
  * the conv writes a dense [out_h][out_w][oc] image into a zero-filled [g][g][oc] buffer which the feature grid
    then reads flat with row length g (nnue_engine.cpp:679-683, nnue_engine.h:236-252): feature f is the f-th byte
    of that flat buffer; cells the conv did not produce are 0 and therefore active whenever threshold < 0.
"""
from __future__ import annotations

import struct
from pathlib import Path
from typing import Dict, Tuple

import numpy as np


def load_nnue(path) -> Dict[str, object]:
    """Parse a `.nnue` file the way NNUEEvaluator::load_model does (nnue_engine.cpp:544-657, :11-46, :161-186, :283-380)."""
    data = Path(path).read_bytes()
    off = 0

    def take(fmt):
        nonlocal off
        vals = struct.unpack_from("<" + fmt, data, off)
        off += struct.calcsize("<" + fmt)
        return vals if len(vals) > 1 else vals[0]

    def arr(dtype, count):
        nonlocal off
        a = np.frombuffer(data, dtype=dtype, count=count, offset=off).copy()
        off += a.nbytes
        return a

    if data[:4] != b"NNUE":
        raise ValueError("Invalid magic number")
    off = 4
    version = take("I")
    if version != 2:
        raise ValueError(f"Unsupported version: {version}")
    m: Dict[str, object] = {}
    m["num_features"], m["l1"], m["l2"], m["l3"], m["buckets"] = take("5I")
    m["nnue2score"], m["quantized_one"], m["threshold"] = take("3f")
    _layer_type = take("I")
    m["conv_scale"] = take("f")
    oc, ic, kh, kw = take("4I")
    if ic != 3 or kh != 3 or kw != 3:
        raise ValueError("conv layer must be 3x3 over 3 channels")
    m["oc"] = oc
    m["conv_w"] = arr(np.int8, oc * 27)
    if take("I") != oc:
        raise ValueError("conv bias count")
    m["conv_b"] = arr(np.int32, oc)
    if oc <= 0 or m["num_features"] % oc:
        raise ValueError("Invalid feature/channel configuration")
    g = int(np.sqrt(m["num_features"] // oc))
    if g * g * oc != m["num_features"]:
        raise ValueError("Invalid feature grid calculation")
    m["grid"] = g
    m["ft_scale"] = take("f")
    f, l1 = take("2I")
    if f != m["num_features"] or l1 != m["l1"]:
        raise ValueError("Feature transformer architecture mismatch")
    m["ft_w"] = arr(np.int16, f * l1).reshape(f, l1)
    if take("I") != l1:
        raise ValueError("ft bias count")
    m["ft_b"] = arr(np.int32, l1)
    stacks = []
    for _ in range(m["buckets"]):
        s: Dict[str, object] = {}
        s["l1_scale"], s["l2_scale"], s["out_scale"], s["l1_fact_scale"] = take("4f")
        o, i = take("2I")
        if i != m["l1"] or o - 1 != m["l2"]:
            raise ValueError("Layer stack architecture mismatch")
        s["l1_w"] = arr(np.int8, o * i).reshape(o, i)
        s["l1_b"] = arr(np.int32, take("I"))
        o, i = take("2I")
        s["l1_fact_w"] = arr(np.int8, o * i).reshape(o, i)
        s["l1_fact_b"] = arr(np.int32, take("I"))
        o, i = take("2I")
        if i != 2 * m["l2"] or o != m["l3"]:
            raise ValueError("Layer stack architecture mismatch")
        s["l2_w"] = arr(np.int8, o * i).reshape(o, i)
        s["l2_b"] = arr(np.int32, take("I"))
        o, i = take("2I")
        if i != m["l3"] or o < 1:
            raise ValueError("Invalid output layer dimensions")
        s["out_w"] = arr(np.int8, o * i).reshape(o, i)
        s["out_b"] = arr(np.int32, take("I"))
        s["classes"] = o
        stacks.append(s)
    m["stacks"] = stacks
    if off != len(data):
        raise ValueError(f"{len(data) - off} trailing bytes")
    return m


def conv_stride(h: int, grid: int) -> int:
    """ceil((H-1)/(g-1)), at least 1 (nnue_engine.cpp:710-718) -- not the training stride (H-1)//(g-1)."""
    if grid > 1:
        return max(1, (h - 1 + grid - 2) // (grid - 1))
    return max(1, h)


def _trunc_div(a: np.ndarray, d: int) -> np.ndarray:
    """C++ integer division (toward zero) of int arrays by a positive int."""
    a = a.astype(np.int64)
    return np.sign(a) * (np.abs(a) // d)


def conv_forward(m, image_flat: np.ndarray, h: int, w: int) -> Tuple[np.ndarray, int]:
    """ConvLayer::forward (nnue_engine.cpp:48-158): int8 [out_h][out_w][oc], and the stride used."""
    oc, scale = m["oc"], np.float32(m["conv_scale"])
    s = conv_stride(h, m["grid"])
    oh, ow = (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1
    x = np.trunc(image_flat.astype(np.float32).reshape(h, w, 3) * scale).astype(np.int64)  # static_cast<int32_t>(input * scale)
    xp = np.zeros((h + 2, w + 2, 3), dtype=np.int64)
    xp[1:-1, 1:-1] = x
    wq = m["conv_w"].astype(np.int64).reshape(oc, 3, 3, 3)  # [oc][kh][kw][ic] as the engine indexes the bytes
    acc = np.tile(m["conv_b"].astype(np.int64), (oh, ow, 1))
    for kh in range(3):
        for kw in range(3):
            patch = xp[kh:kh + (oh - 1) * s + 1:s, kw:kw + (ow - 1) * s + 1:s]  # [oh][ow][ic]
            acc += np.einsum("hwi,oi->hwo", patch, wq[:, kh, kw, :])
    out = np.clip(_trunc_div(acc, int(scale)), -127, 127).astype(np.int8)
    return out, s


def active_features(m, conv_out: np.ndarray) -> np.ndarray:
    """from_conv_output + extract_features (nnue_engine.h:236-283): ascending ids of the flat [g][g][oc] buffer."""
    g, oc = m["grid"], m["oc"]
    flat = np.zeros(g * g * oc, dtype=np.int8)
    produced = conv_out.reshape(-1)
    if produced.size > flat.size:
        raise ValueError("conv output exceeds the engine's grid buffer (undefined behaviour in the reference)")
    flat[:produced.size] = produced
    on = flat.astype(np.float32) > np.float32(m["threshold"])
    if oc > 64:
        on.reshape(g * g, oc)[:, 64:] = False  # only 64 channels per cell are bit-packed (nnue_engine.h:243)
    return np.nonzero(on)[0].astype(np.int64)


def ft_forward(m, ids: np.ndarray) -> np.ndarray:
    """FeatureTransformer::forward with int16 wrap-around (simd_scalar.cpp:78-96), then the clipped ReLU of
    nnue_engine.cpp:726-729."""
    acc = m["ft_b"].astype(np.int64) + m["ft_w"][ids].astype(np.int64).sum(axis=0)
    acc = ((acc + 32768) % 65536 - 32768).astype(np.int64)  # int16 arithmetic wraps
    return np.clip(acc, 0, int(np.int16(m["quantized_one"])))


def forward_multiclass(s, ft: np.ndarray, l1: int, l2: int, l3: int) -> np.ndarray:
    """LayerStack::forward_multiclass (nnue_engine.cpp:480-539)."""
    half = l1 // 2
    a, b = ft[:half], ft[half:2 * half]
    pair = np.zeros(l1, dtype=np.int64)
    pair[:half] = np.clip(_trunc_div(a * b, 128), 0, 127)
    pair[half:2 * half] = np.clip(a, 0, 127)
    acc1 = s["l1_b"][:l2].astype(np.int64) + s["l1_w"][:l2].astype(np.int64) @ pair
    # dense_forward_scalar: float division, truncation, clamp to [0, 127] (simd_scalar.cpp:117-136)
    h1 = np.clip(np.trunc(acc1.astype(np.float32) / np.float32(s["l1_scale"])).astype(np.int64), 0, 127)
    acc2 = s["l2_b"].astype(np.int64) + s["l2_w"][:, :l2].astype(np.int64) @ h1
    h2 = np.maximum(np.clip(_trunc_div(acc2, int(np.float32(s["l2_scale"]))), -127, 127), 0)
    acc3 = s["out_b"].astype(np.int64) + s["out_w"].astype(np.int64) @ h2
    return (acc3.astype(np.float32) / np.float32(s["out_scale"])).astype(np.float32)


def evaluate_logits(m, image_flat: np.ndarray, h: int, w: int, bucket: int = 0) -> Tuple[np.ndarray, np.float32]:
    """(logits [C] float32, density float32) as engine/nnue_inference.cpp:42-60 prints them."""
    if bucket >= m["buckets"]:
        bucket = 0
    conv_out, _ = conv_forward(m, image_flat, h, w)
    ids = active_features(m, conv_out)
    ft = ft_forward(m, ids)
    logits = forward_multiclass(m["stacks"][bucket], ft, m["l1"], m["l2"], m["l3"])
    density = np.float32(ids.size) / np.float32(m["num_features"])
    return logits, density
