"""The compiled engine's integer inference on the GPU.

``EngineModel.load(path)`` parses a ``.nnue`` file the way ``NNUEEvaluator::load_model`` does
(engine/src/nnue_engine.cpp:544-657; same rejections) and keeps its quantised tensors in device memory;
``evaluate_logits(images)`` is ``NNUEEvaluator::evaluate_logits`` (nnue_engine.cpp:704-734) for a whole batch --
bit-identical to the C++ engine (tests/golden/engine_cases.npz holds outputs of the real engine).
"""
from __future__ import annotations

import ctypes
import struct
from pathlib import Path
from typing import Optional, Tuple

import numpy as np
import torch

from . import lib


class _CModel(ctypes.Structure):  # include/nnue_hip.h: nnue_engine_model
    _fields_ = ([(n, ctypes.c_int32) for n in ("num_features", "l1", "l2", "l3", "classes", "grid", "oc")]
                + [(n, ctypes.c_float) for n in ("conv_scale", "threshold", "quantized_one", "l1_scale", "l2_scale", "out_scale")]
                + [(n, ctypes.c_void_p) for n in ("conv_w", "conv_b", "ft_w", "ft_b", "l1_w", "l1_b", "l2_w", "l2_b", "out_w", "out_b")])


class EngineFormatError(ValueError):
    pass


class _Reader:
    def __init__(self, data: bytes):
        self.data, self.off = data, 0

    def take(self, fmt: str):
        try:
            vals = struct.unpack_from("<" + fmt, self.data, self.off)
        except struct.error as e:
            raise EngineFormatError(f"truncated file: {e}") from None
        self.off += struct.calcsize("<" + fmt)
        return vals if len(vals) > 1 else vals[0]

    def array(self, dtype, count: int) -> np.ndarray:
        nbytes = np.dtype(dtype).itemsize * count
        if count < 0 or self.off + nbytes > len(self.data):
            raise EngineFormatError("truncated file")
        a = np.frombuffer(self.data, dtype=dtype, count=count, offset=self.off).copy()
        self.off += nbytes
        return a


class EngineModel:
    """Quantised tensors of one `.nnue` file on the device + the scalars of its header."""

    def __init__(self, header: dict, tensors: dict, device):
        self.header = header
        self.device = torch.device(device)
        self.tensors = {k: torch.from_numpy(v).to(self.device) for k, v in tensors.items()}
        c = _CModel()
        for k in ("num_features", "l1", "l2", "l3", "classes", "grid", "oc"):
            setattr(c, k, int(header[k]))
        for k in ("conv_scale", "threshold", "quantized_one", "l1_scale", "l2_scale", "out_scale"):
            setattr(c, k, float(header[k]))
        for k, t in self.tensors.items():
            setattr(c, k, t.data_ptr())
        self._c = c
        self._scratch: Optional[torch.Tensor] = None

    @property
    def num_classes(self) -> int:
        return int(self.header["classes"])

    @staticmethod
    def load(path, device=None, bucket: int = 0) -> "EngineModel":
        if not torch.cuda.is_available():
            raise lib.NnueHipError("the engine restatement runs on the GPU only (no CPU fallback in this build)")
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        r = _Reader(Path(path).read_bytes())
        if r.data[:4] != b"NNUE":
            raise EngineFormatError("Invalid magic number")
        r.off = 4
        version = r.take("I")
        if version != 2:
            raise EngineFormatError(f"Unsupported version: {version}")
        h = {}
        h["num_features"], h["l1"], h["l2"], h["l3"], h["buckets"] = r.take("5I")
        h["nnue2score"], h["quantized_one"], h["threshold"] = r.take("3f")
        r.take("I")  # layer type
        h["conv_scale"] = r.take("f")
        oc, ic, kh, kw = r.take("4I")
        if ic != 3 or kh != 3 or kw != 3 or oc <= 0:
            raise EngineFormatError("Failed to load conv layer")
        t = {"conv_w": r.array(np.int8, oc * 27)}
        if r.take("I") != oc:
            raise EngineFormatError("Failed to load conv layer")
        t["conv_b"] = r.array(np.int32, oc)
        if h["num_features"] == 0 or h["num_features"] % oc:
            raise EngineFormatError("Invalid feature/channel configuration")
        g = int(np.sqrt(h["num_features"] // oc))
        if g * g * oc != h["num_features"]:
            raise EngineFormatError("Invalid feature grid calculation")
        h["oc"], h["grid"] = oc, g
        r.take("f")  # ft scale (unused by the engine's forward)
        f, l1 = r.take("2I")
        if f != h["num_features"] or l1 != h["l1"]:
            raise EngineFormatError("Feature transformer architecture mismatch")
        t["ft_w"] = r.array(np.int16, f * l1)
        if r.take("I") != l1:
            raise EngineFormatError("Failed to load feature transformer")
        t["ft_b"] = r.array(np.int32, l1)
        if h["buckets"] < 1:
            raise EngineFormatError("no layer stack in the file")
        chosen = bucket if bucket < h["buckets"] else 0  # nnue_engine.cpp:705-707
        for i in range(h["buckets"]):
            scales = r.take("4f")
            o, n = r.take("2I")
            if n != h["l1"] or o - 1 != h["l2"]:
                raise EngineFormatError("Layer stack architecture mismatch")
            l1_w, l1_b = r.array(np.int8, o * n), r.array(np.int32, r.take("I"))
            o, n = r.take("2I")
            if n != h["l1"] or o <= h["l2"]:
                raise EngineFormatError(f"Failed to load layer stack {i}")
            r.array(np.int8, o * n)
            r.array(np.int32, r.take("I"))  # factoriser: not on the multiclass path
            o, n = r.take("2I")
            if n != 2 * h["l2"] or o != h["l3"]:
                raise EngineFormatError("Layer stack architecture mismatch")
            l2_w, l2_b = r.array(np.int8, o * n), r.array(np.int32, r.take("I"))
            o, n = r.take("2I")
            if n != h["l3"] or o < 1:
                raise EngineFormatError(f"Invalid output layer dimensions: {n} -> {o}")
            out_w, out_b = r.array(np.int8, o * n), r.array(np.int32, r.take("I"))
            if i == chosen:
                h["l1_scale"], h["l2_scale"], h["out_scale"], _ = scales
                h["classes"] = o
                t.update(l1_w=l1_w, l1_b=l1_b, l2_w=l2_w, l2_b=l2_b, out_w=out_w, out_b=out_b)
        return EngineModel(h, t, device)

    def evaluate_logits(self, images: torch.Tensor, height: Optional[int] = None, width: Optional[int] = None
                        ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(logits [B, C] float32, density [B] float32).  ``images`` is what the reference hands the engine: per sample
        a flat buffer of 3*H*W floats which the engine indexes as HWC -- for a [B,3,H,W] tensor that is its memory as
        it stands (evaluate.py:150-161 passes shape[1], shape[2] as H, W), which is reproduced, not corrected."""
        images = lib._need(images, torch.float32, "images")
        if images.dim() == 4:
            b, h, w = images.shape[0], images.shape[2], images.shape[3]
            if images.shape[1] != 3:
                raise ValueError(f"images: expected [B,3,H,W], got {tuple(images.shape)}")
        elif images.dim() == 2 and height and width and images.shape[1] == 3 * height * width:
            b, h, w = images.shape[0], height, width
        else:
            raise ValueError("images: expected [B,3,H,W], or [B,3*H*W] with height and width")
        need = int(lib.load().nnue_engine_scratch(ctypes.byref(self._c), b))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty((max(16, need),), dtype=torch.uint8, device=self.device)
        logits = torch.empty((b, self.num_classes), dtype=torch.float32, device=self.device)
        density = torch.empty((b,), dtype=torch.float32, device=self.device)
        lib._call("nnue_engine_evaluate_logits", ctypes.addressof(self._c), images.data_ptr(), b, h, w, logits.data_ptr(),
                  density.data_ptr(), self._scratch.data_ptr(), self._scratch.numel(), lib._stream(images))
        return logits, density
