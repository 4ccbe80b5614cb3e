#!/usr/bin/env python3
"""Drop-in replacement for the NNUE half of the reference's ``serialize.py``: writes the version-2
``.nnue`` container that ``engine/nnue_inference.cpp`` loads, byte for byte.

Host-side code (numpy + struct on CPU tensors); nothing here is on the training hot path.

File layout, little-endian (writer: reference serialize.py:30-63, :103-136, :394-491; reader:
engine/src/nnue_engine.cpp:544-657):

  header   "NNUE" | u32 2 | u32 F | u32 L1 | u32 L2 | u32 L3 | u32 buckets (1 in the reference; K for the bucketed extension) |
           f32 nnue2score | f32 quantized_one | f32 visual_threshold(mean)
  conv     u32 0 | f32 scale | u32 oc, ic, kh, kw | i8[oc*ic*kh*kw] | u32 oc | i32[oc]
  FT       f32 scale | u32 F | u32 L1 | i16[F*L1] | u32 L1 | i32[L1]
  stack    f32 x4 scales | (L2+1) x L1 i8 (+1 zero row) + i32 bias | L1 x L1 identity*127 + zero bias |
           L3 x 2*L2 i8 (second half zero) + bias | C x L3 i8 + bias   -- each as u32 out, u32 in,
           weights, u32 count, biases

EtinyNet export (.etiny) is out of scope and stays with the reference.
"""
from __future__ import annotations

import argparse
import struct
from pathlib import Path
from typing import Any, BinaryIO, Dict, Tuple

import numpy as np
import torch

from nnue import NNUE, GridFeatureSet

NNUE_MAGIC = b"NNUE"
NNUE_VERSION = 2
QUANT_SCALE = 64.0


def _u32(f: BinaryIO, *values: int) -> None:
    f.write(struct.pack(f"<{len(values)}I", *values))


def _f32(f: BinaryIO, *values: float) -> None:
    f.write(struct.pack(f"<{len(values)}f", *values))


def _array(f: BinaryIO, t, dtype: str) -> None:
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    f.write(np.ascontiguousarray(a).astype(dtype).tobytes())


# ---------------------------------------------------------------------------- quantisers
def _quantize(weight: torch.Tensor, bias: torch.Tensor, scale: float) -> Dict[str, Any]:
    """int8 weights = clamp(round_half_even(w*scale), +-127); int32 bias = round(b*scale), unclamped
    (serialize.py:218-222, :234-237)."""
    w_q = torch.round(weight * scale).clamp(-127, 127).to(torch.int8)
    b_q = torch.round(bias * scale).to(torch.int32)
    return {"weight": w_q, "bias": b_q, "scale": scale}


def quantize_conv_layer(conv_layer, scale: float = QUANT_SCALE) -> Dict[str, Any]:
    bias = conv_layer.bias.data if conv_layer.bias is not None else torch.zeros(conv_layer.out_channels)
    return _quantize(conv_layer.weight.data, bias, scale)


def quantize_linear_layer(linear_layer, scale: float = QUANT_SCALE) -> Dict[str, Any]:
    """Works for nn.Linear and for FeatureTransformer (anything with .weight/.bias)."""
    w = linear_layer.weight.data
    bias = linear_layer.bias.data if linear_layer.bias is not None else torch.zeros(linear_layer.out_features)
    return _quantize(w, bias, scale)


# ---------------------------------------------------------------------------- section writers
def write_nnue_header(f: BinaryIO, metadata: Dict[str, Any]) -> None:
    need = ("feature_set", "L1", "L2", "L3", "nnue2score", "quantized_one", "visual_threshold")
    missing = [k for k in need if k not in metadata]
    if missing:
        raise ValueError("Missing required NNUE metadata keys: " + ", ".join(missing))
    f.write(NNUE_MAGIC)
    _u32(f, NNUE_VERSION, metadata["feature_set"].num_features, metadata["L1"], metadata["L2"], metadata["L3"],
         int(metadata.get("num_ls_buckets", 1)))
    _f32(f, metadata["nnue2score"], metadata["quantized_one"], float(metadata["visual_threshold"]))


def write_conv_layer(f: BinaryIO, conv_data: Dict[str, Any]) -> None:
    w, b = conv_data["weight"], conv_data["bias"]
    _u32(f, 0)  # layer type: standard conv
    _f32(f, conv_data["scale"])
    _u32(f, *w.shape)  # oc, ic, kh, kw
    _array(f, w, "i1")
    _u32(f, b.shape[0])
    _array(f, b, "<i4")


def write_feature_transformer(f: BinaryIO, ft_data: Dict[str, Any]) -> None:
    w, b = ft_data["weight"], ft_data["bias"]
    _f32(f, ft_data["scale"])
    _u32(f, w.shape[0], w.shape[1])
    _array(f, w, "<i2")  # the engine accumulates in int16; values are int8-range
    _u32(f, b.shape[0])
    _array(f, b, "<i4")


def _dense(f: BinaryIO, weight: np.ndarray, bias: np.ndarray) -> None:
    _u32(f, weight.shape[0], weight.shape[1])
    f.write(weight.astype("i1").tobytes())
    _u32(f, bias.shape[0])
    f.write(bias.astype("<i4").tobytes())


def write_layer_stack(f: BinaryIO, classifier_data: Dict[str, Any]) -> None:
    """One engine LayerStack from the three quantised Linear layers (serialize.py:423-491): the engine's
    chess-era slots are filled with a zero extra output row, an identity "factoriser" and a zero second
    half of the L2 input."""
    l1, l2, l3 = classifier_data["layers"][:3]
    _f32(f, l1["scale"], l2["scale"], l3["scale"], l1["scale"])
    w1 = l1["weight"].cpu().numpy()
    n2, n1 = w1.shape
    w1x = np.zeros((n2 + 1, n1), dtype=np.int8)
    w1x[:n2] = w1
    b1x = np.zeros(n2 + 1, dtype=np.int32)
    b1x[:n2] = l1["bias"].cpu().numpy()
    _dense(f, w1x, b1x)
    _dense(f, np.eye(n1, dtype=np.int8) * 127, np.zeros(n1, dtype=np.int32))
    w2 = l2["weight"].cpu().numpy()
    n3 = w2.shape[0]
    w2x = np.zeros((n3, 2 * n2), dtype=np.int8)
    w2x[:, :n2] = w2
    _dense(f, w2x, l2["bias"].cpu().numpy())
    _dense(f, l3["weight"].cpu().numpy(), l3["bias"].cpu().numpy())


def write_classifier(f: BinaryIO, classifier_data: Dict[str, Any]) -> None:
    """One LayerStack record per bucket, in bucket order (the reader loops over the header's num_ls_buckets:
    engine/src/nnue_engine.cpp:619-635); the reference's own writer always has exactly one (serialize.py:57, :494-497)."""
    for stack in classifier_data.get("stacks", [classifier_data]):
        write_layer_stack(f, stack)


def serialize_model(model: NNUE, output_path) -> None:
    """eval() + clip weights to [-1,1] (mutates the model, as the reference does: serialize.py:509-510),
    quantise, write."""
    model.eval()
    model._clip_weights()
    q = model.get_quantized_model_data()
    with open(output_path, "wb") as f:
        write_nnue_header(f, q["metadata"])
        write_conv_layer(f, q["conv_layer"])
        write_feature_transformer(f, q["feature_transformer"])
        write_classifier(f, q["classifier"])
    print(f"Successfully serialized model to {output_path}")


# ---------------------------------------------------------------------------- checkpoint -> model
def infer_architecture_from_state_dict(state_dict) -> Tuple[GridFeatureSet, int, int, int, int]:
    """(feature_set, L1, L2, L3, classes) from tensor shapes (serialize.py:715-788)."""
    if "input.weight" not in state_dict:
        raise ValueError("Cannot find NNUE model weights in state dict. Available keys: "
                         + str(list(state_dict.keys())[:10]))
    num_features, l1 = state_dict["input.weight"].shape
    fps = state_dict["conv.weight"].shape[0]
    grid = int((num_features / fps) ** 0.5)
    if grid * grid * fps != num_features:
        for g, c in ((4, 6), (8, 12), (16, 8), (32, 64)):
            if g * g * c == num_features:
                grid, fps = g, c
                break
    def rows(key, default):
        return state_dict[key].shape[-2] if key in state_dict else default  # [out, in] or stacked [K, out, in]
    l2 = rows("classifier.classifier.0.weight", 16)
    l3 = rows("classifier.classifier.2.weight", 32)
    classes = rows("classifier.classifier.4.weight", 10)
    return GridFeatureSet(grid, fps), l1, l2, l3, classes


def num_ls_buckets_of(state_dict) -> int:
    """1 for the reference's checkpoints; K for the bucketed extension (stacked [K, out, in] classifier weights)."""
    w = state_dict.get("classifier.classifier.0.weight")
    return int(w.shape[0]) if w is not None and w.dim() == 3 else 1


def load_model_from_checkpoint(checkpoint_path) -> NNUE:
    """Accepts {"state_dict": ...} (optionally with saved hyper-parameters) or a bare state dict
    (serialize.py:531-585).  Safe loader only."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    state = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    saved = [ckpt.get(k) if "state_dict" in ckpt else None
             for k in ("feature_set", "l1_size", "l2_size", "l3_size", "num_classes")]
    if all(v is not None for v in saved):
        feature_set, l1, l2, l3, classes = saved
    else:
        feature_set, l1, l2, l3, classes = infer_architecture_from_state_dict(state)
    model = NNUE(feature_set=feature_set, l1_size=l1, l2_size=l2, l3_size=l3, num_classes=classes,
                 num_ls_buckets=num_ls_buckets_of(state))
    model.load_state_dict(state)
    return model


def detect_model_type(checkpoint_path) -> str:
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    state = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    if any(k in ("input.weight", "input.bias", "conv.weight") or "layer_stacks" in k for k in state):
        if not any(k.startswith(("stage", "global_pool")) for k in state):
            return "nnue"
    raise ValueError(f"Not an NNUE checkpoint (EtinyNet export stays with the reference): {checkpoint_path}")


def main() -> None:
    ap = argparse.ArgumentParser(description="Serialize an NNUE checkpoint to the .nnue format")
    ap.add_argument("checkpoint", type=Path)
    ap.add_argument("output", type=Path)
    args = ap.parse_args()
    detect_model_type(args.checkpoint)
    serialize_model(load_model_from_checkpoint(args.checkpoint), args.output)


if __name__ == "__main__":
    main()
