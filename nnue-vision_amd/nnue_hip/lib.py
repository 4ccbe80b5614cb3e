"""ctypes binding of libnnue_hip.so (C ABI: include/nnue_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every wrapper checks
device / dtype / contiguity / shapes on the host, then passes raw pointers.  There is no
fallback of any kind: if the library is missing or a tensor is not on the GPU the call raises.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libnnue_hip.so"
ABI_VERSION = 30

_c_int, _c_i64, _c_f, _c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class NnueBuckets(ctypes.Structure):
    """struct nnue_buckets of include/nnue_hip.h (read on the host by the *_bucketed entry points)."""
    _fields_ = [("K", ctypes.c_int32), ("bucket", _c_p), ("rows", _c_p), ("tile_bucket", _c_p), ("seg", _c_p),
                ("tiles", ctypes.c_int32)]


_c_bk = ctypes.POINTER(NnueBuckets)


class NnueClsRider(ctypes.Structure):
    """struct nnue_cls_rider of include/nnue_hip.h: the arguments of the classifier's small-gradient tile family when it rides
    in nnue_ftm_backward's launch (opaque; filled by nnue_classifier_train_rider)."""
    _fields_ = [("opaque", ctypes.c_ubyte * 256)]


_c_rider = ctypes.POINTER(NnueClsRider)

# name -> (restype, argtypes); mirrors include/nnue_hip.h one to one
SIGNATURES = {
    "nnue_hip_abi_version": (_c_int, []),
    "nnue_hip_last_error": (ctypes.c_char_p, []),
    "nnue_conv3x3_forward": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p]),
    "nnue_conv3x3_backward_input": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_sparse_values": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_sparse_values_backward": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_binarize_features": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p]),
    "nnue_act_to_padded": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "nnue_ste_conv_backward_scratch": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_ste_conv_backward_chunks": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_ste_conv_backward": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                                        _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "nnue_ft_prepare": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p]),
    "nnue_ft_forward": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_ft_backward_weight": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "nnue_ft_backward_values": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int,
                                         _c_p, _c_int, _c_p]),
    "nnue_ftb_supported": (_c_int, [_c_int]),
    "nnue_ftb_list_tiles": (_c_int, [_c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_ftb_scratch": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_binarize_bits": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_int, _c_p, _c_int,
                                    _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_p]),
    "nnue_ftb_forward": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_i64, _c_p]),
    "nnue_ftb_backward_weight": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p,
                                          _c_p, _c_i64, _c_p]),
    "nnue_ftb_backward_values": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_ftm_supported": (_c_int, [_c_int, _c_int, _c_int]),
    "nnue_ftm_scratch": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_ftm_binarize": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p]),
    "nnue_ftm_conv_binarize": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "nnue_ftm_conv_binarize_patches": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p,
                                                _c_p]),
    "nnue_ste_conv_backward_patches": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_i64, _c_int,
                                                _c_p]),
    "nnue_ftm_forward": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_i64, _c_p]),
    "nnue_ftm_forward_grouping": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_i64, _c_p, _c_int,
                                           _c_p, _c_p, _c_p, _c_p, _c_p]),
    "nnue_ftm_forward_l1_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_ftm_forward_l1": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "nnue_ftm_backward_weight": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "nnue_ftm_backward_values": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_ftm_backward_values_scratch": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_ftm_backward_values_ws": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_i64, _c_p]),
    "nnue_ftm_backward_cw_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_ftm_backward_sq_count": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_ftm_backward": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p,
                                   _c_p, _c_p, _c_int, _c_p, _c_p, _c_rider, _c_p]),
    "nnue_ftm_backward_bucketed": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p,
                                            _c_p, _c_p, _c_int, _c_p, _c_p, _c_int, _c_p, _c_int, _c_rider, _c_p]),
    "nnue_classifier_train_rider": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                             _c_p, _c_p, _c_p, _c_i64, _c_bk, _c_rider]),
    "nnue_classifier_train_dz1_grouped_offset": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_train_x_grouped_offset": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_scratch": (_c_i64, [_c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_forward": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f,
                                         _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p,
                                         _c_p, _c_i64, _c_p]),
    "nnue_classifier_backward": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p,
                                          _c_int, _c_int, _c_int, _c_int, _c_int,
                                          _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "nnue_classifier_train_scratch": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_train_dz1_offset": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_train_step": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p, _c_f,
                                            _c_int, _c_int, _c_int, _c_int, _c_int,
                                            _c_p, _c_p, _c_p, _c_p, _c_p,
                                            _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_p]),
    "nnue_bucket_tile_count": (_c_int, [_c_int, _c_int]),
    "nnue_bucket_group": (_c_int, [_c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "nnue_classifier_scratch_bucketed": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_train_scratch_bucketed": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_classifier_forward_bucketed": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f,
                                                  _c_int, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p,
                                                  _c_p, _c_i64, _c_bk, _c_p]),
    "nnue_classifier_backward_bucketed": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_f, _c_p, _c_p, _c_p,
                                                   _c_int, _c_int, _c_int, _c_int, _c_int,
                                                   _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_bk, _c_p]),
    "nnue_classifier_train_step_bucketed": (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p, _c_f,
                                                     _c_int, _c_int, _c_int, _c_int, _c_int,
                                                     _c_p, _c_p, _c_p, _c_p, _c_p,
                                                     _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_int, _c_bk, _c_p]),
    "nnue_cross_entropy": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_f, _c_p, _c_p, _c_p, _c_p]),
    "nnue_confusion_accumulate": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_p, _c_p]),
    "nnue_load_batch": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_i64, _c_int, ctypes.c_uint64, ctypes.c_uint64,
                                 _c_p, _c_p, _c_p]),
    "nnue_engine_scratch": (_c_i64, [_c_p, _c_int]),
    "nnue_engine_evaluate_logits": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "nnue_sgd_scratch": (_c_i64, [_c_i64]),
    "nnue_adam_step": (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f, _c_f,
                                _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "nnue_sgd_step": (_c_int, [_c_p, _c_p, _c_p, _c_i64, _c_f, _c_f, _c_f, _c_f, _c_f, _c_int,
                               _c_p, _c_p, _c_i64, _c_p, _c_int, _c_int, _c_p, _c_p, _c_p, _c_int, _c_i64, _c_i64, _c_p, _c_int, _c_p, _c_p]),
    "nnue_ftm_uses_bf16": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_sqnorm_partials": (_c_int, [_c_p, _c_i64, _c_p, _c_int, _c_p]),
    "nnue_dp_factor_chunk_bytes": (_c_i64, [_c_int, _c_int, _c_int, _c_i64]),
    "nnue_dp_factor_offset": (_c_i64, [_c_int, _c_int, _c_int, _c_int, _c_i64]),
    "nnue_dp_factor_pack": (_c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_p, _c_p]),
    "nnue_dp_factor_unpack": (_c_int, [_c_p, _c_int, _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "nnue_ftm_gram_sq_count": (_c_i64, [_c_int, _c_int]),
    "nnue_ftm_gram_scratch": (_c_i64, [_c_int, _c_int, _c_int]),
    "nnue_ftm_gram_sqnorm": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "nnue_ftm_gram_sqnorm_tail": (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "nnue_ftm_backward_tail_rows": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p]),
    "nnue_ftm_backward_weight_update": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_f, _c_f, _c_f, _c_f,
                                                 _c_int, _c_p, _c_p]),
    "nnue_ftm_update_forward_supported": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "nnue_ftm_backward_weight_update_forward": (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_int, _c_p, _c_p, _c_p, _c_f, _c_f, _c_f, _c_f,
                                                         _c_int, _c_p, _c_p, _c_p, _c_int, _c_p, _c_p, _c_p, _c_i64, _c_p]),
}

_lib: Optional[ctypes.CDLL] = None


class NnueHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Loads the library (once).  Raises loudly if it was not built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NnueHipError(
            f"{LIB_PATH} is missing: build it with `python nnue-vision_amd/csrc/build.py` "
            "(or __graft_entry__.build()).  There is no CPU or eager fallback.")
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    got = lib.nnue_hip_abi_version()
    if got != ABI_VERSION:
        raise NnueHipError(f"libnnue_hip.so ABI {got} != binding ABI {ABI_VERSION}: rebuild")
    _lib = lib
    return lib


_recording: Optional[list] = None  # when a list: every C call is also appended as (name, fn, args-without-stream)


class record_calls:
    """Context manager: collects the raw C calls made by the wrappers so that a fixed kernel sequence
    can later be replayed with no Python between launches (``run_plan``) -- eagerly or under hipGraph
    capture.  The last argument of every launching entry point is the stream; it is re-read at replay."""

    def __enter__(self):
        global _recording
        self.prev, _recording = _recording, []
        self.calls = _recording
        return self.calls

    def __exit__(self, *exc):
        global _recording
        _recording = self.prev
        return False


_timing: Optional[dict] = None  # when a dict {entry point: list}: direct calls of those entry points are bracketed with events


class time_calls:
    """Context manager: like ``run_plan(..., timers)`` for calls made directly through the wrappers (the exchange-and-update
    tail of a data-parallel step is not a recorded plan): (start, end) events on the current stream around each named call."""

    def __init__(self, timers):
        self.timers = timers

    def __enter__(self):
        global _timing
        self.prev, _timing = _timing, self.timers
        return self.timers

    def __exit__(self, *exc):
        global _timing
        _timing = self.prev
        return False


def _call(name: str, *args) -> None:
    lib = load()
    fn = getattr(lib, name)
    if _timing is not None and name in _timing:
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        rc = fn(*args)
        t1.record()
        _timing[name].append((t0, t1))
    else:
        rc = fn(*args)
    if rc != 0:
        raise NnueHipError(f"{name} failed (code {rc}): {lib.nnue_hip_last_error().decode()}")
    if _recording is not None:
        _recording.append((name, fn, args[:-1]))


def run_plan(plan, stream_ptr: int, timers=None) -> None:
    """Replays recorded calls on `stream_ptr`.  timers: {entry-point name: list} -- a (start, end) pair of
    torch events is recorded around each matching call on the current stream and appended to the list."""
    for name, fn, args in plan:
        if timers is not None and name in timers:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            rc = fn(*args, stream_ptr)
            t1.record()
            timers[name].append((t0, t1))
        else:
            rc = fn(*args, stream_ptr)
        if rc != 0:
            raise NnueHipError(f"{name} failed (code {rc}): {load().nnue_hip_last_error().decode()}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _need(t: torch.Tensor, dtype, what: str, shape: Optional[Tuple[int, ...]] = None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise NnueHipError(f"{what}: tensor is on {t.device}; the NNUE hot path runs on the GPU only "
                           "(no CPU fallback in this build)")
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class ActList:
    """Active-feature list + transposed coefficients (layout: include/nnue_hip.h)."""
    rows: torch.Tensor   # int32 [B, cap]
    pos: torch.Tensor    # int32 [B, cap]
    coef: torch.Tensor   # float32 [B, cap]
    n: torch.Tensor      # int32 [B]
    coefT: torch.Tensor  # float32 [F, ldb]
    cap: int
    ldb: int

    @property
    def batch(self) -> int:
        return self.n.shape[0]

    @staticmethod
    def empty(batch: int, cap: int, num_rows: int, device) -> "ActList":
        ldb = round_up(batch, 64)
        i32 = dict(dtype=torch.int32, device=device)
        return ActList(torch.empty((batch, cap), **i32), torch.empty((batch, cap), **i32),
                       torch.empty((batch, cap), dtype=torch.float32, device=device),
                       torch.empty((batch,), **i32),
                       torch.empty((num_rows, ldb), dtype=torch.float32, device=device), cap, ldb)


# ---------------------------------------------------------------------------- front end
def conv_out_hw(h: int, w: int, stride: int) -> Tuple[int, int]:
    return (h - 1) // stride + 1, (w - 1) // stride + 1


def conv3x3_forward(images: torch.Tensor, weight: torch.Tensor, stride: int,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if images.dim() != 4 or images.shape[1] != 3:
        raise ValueError(f"images: expected [B,3,H,W], got {tuple(images.shape)}")
    images = _need(images, torch.float32, "images")
    b, _, h, w = images.shape
    fps = weight.shape[0]
    weight = _need(weight, torch.float32, "conv weight", (fps, 3, 3, 3))
    gh, gw = conv_out_hw(h, w, stride)
    if out is None:
        out = torch.empty((b, fps, gh, gw), dtype=torch.float32, device=images.device)
    else:
        _need(out, torch.float32, "conv_out", (b, fps, gh, gw))
    _call("nnue_conv3x3_forward", images.data_ptr(), weight.data_ptr(), out.data_ptr(), b, h, w, fps, stride,
          _stream(images))
    return out


def conv3x3_backward_input(d_conv_out: torch.Tensor, weight: torch.Tensor, image_shape, stride: int) -> torch.Tensor:
    """d(images) of the 3x3 / pad 1 conv: [B,3,H,W] from d_conv_out [B,fps,Gh,Gw]."""
    b, _, h, w = image_shape
    fps = weight.shape[0]
    gh, gw = conv_out_hw(h, w, stride)
    d_conv_out = _need(d_conv_out, torch.float32, "d_conv_out", (b, fps, gh, gw))
    weight = _need(weight, torch.float32, "conv weight", (fps, 3, 3, 3))
    d_images = torch.empty((b, 3, h, w), dtype=torch.float32, device=d_conv_out.device)
    _call("nnue_conv3x3_backward_input", d_conv_out.data_ptr(), weight.data_ptr(), b, h, w, fps, int(stride), d_images.data_ptr(),
          _stream(d_conv_out))
    return d_images


def sparse_values(flat_map: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """val[b,i] = map[b, idx[b,i]] (0 where idx < 0); map float32 [B,P], idx int64 [B,M]."""
    flat_map = _need(flat_map, torch.float32, "feature map")
    idx = _need(idx, torch.int64, "feature indices")
    b, p = flat_map.shape
    val = torch.empty(idx.shape, dtype=torch.float32, device=flat_map.device)
    _call("nnue_sparse_values", flat_map.data_ptr(), idx.data_ptr(), b, p, idx.shape[1], val.data_ptr(), _stream(flat_map))
    return val


def sparse_values_backward(d_val: torch.Tensor, idx: torch.Tensor, positions: int) -> torch.Tensor:
    d_val = _need(d_val, torch.float32, "d_val")
    idx = _need(idx, torch.int64, "feature indices", tuple(d_val.shape))
    b, m = idx.shape
    d_map = torch.empty((b, positions), dtype=torch.float32, device=d_val.device)
    _call("nnue_sparse_values_backward", d_val.data_ptr(), idx.data_ptr(), b, positions, m, d_map.data_ptr(), _stream(d_val))
    return d_map


def binarize_features(conv_out: torch.Tensor, thr: torch.Tensor, num_rows: int,
                      act: Optional[ActList] = None) -> ActList:
    conv_out = _need(conv_out, torch.float32, "conv_out")
    if conv_out.dim() != 4:
        raise ValueError(f"conv_out: expected [B,fps,Gh,Gw], got {tuple(conv_out.shape)}")
    b, fps, gh, gw = conv_out.shape
    thr = _need(thr.reshape(-1), torch.float32, "threshold", (fps,))
    cap = fps * gh * gw
    if act is None:
        act = ActList.empty(b, cap, num_rows, conv_out.device)
    elif act.cap != cap or act.batch != b or act.coefT.shape != (num_rows, act.ldb):
        raise ValueError("binarize_features: act list does not match the map")
    _call("nnue_binarize_features", conv_out.data_ptr(), thr.data_ptr(), b, fps, gh, gw, num_rows,
          act.rows.data_ptr(), act.pos.data_ptr(), act.coef.data_ptr(), act.n.data_ptr(),
          act.coefT.data_ptr(), act.ldb, _stream(conv_out))
    return act


def act_to_padded(act: ActList, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    b = act.batch
    idx = torch.empty((b, width), dtype=torch.int64, device=act.n.device)
    val = torch.empty((b, width), dtype=torch.float32, device=act.n.device)
    _call("nnue_act_to_padded", act.pos.data_ptr(), act.coef.data_ptr(), act.n.data_ptr(), act.cap, b, width,
          idx.data_ptr(), val.data_ptr(), _stream(act.n))
    return idx, val


def ste_conv_backward(images, conv_out, thr, d_conv_out, stride: int,
                      d_thr: Optional[torch.Tensor] = None, d_weight: Optional[torch.Tensor] = None,
                      scratch: Optional[torch.Tensor] = None, stages: int = 3):
    """stages=1 leaves the per-workgroup partials at the start of `scratch` (finish with stages=2 or hand them to
    sgd_step(ste=...), whose norm launch then carries the second stage)."""
    images = _need(images, torch.float32, "images")
    b, _, h, w = images.shape
    fps = conv_out.shape[1]
    gh, gw = conv_out_hw(h, w, stride)
    conv_out = _need(conv_out, torch.float32, "conv_out", (b, fps, gh, gw))
    d_conv_out = _need(d_conv_out, torch.float32, "d_conv_out").view(b, fps, gh, gw)
    thr = _need(thr.reshape(-1), torch.float32, "threshold", (fps,))
    dev = images.device
    if d_thr is None:
        d_thr = torch.empty((fps,), dtype=torch.float32, device=dev)
    if d_weight is None:
        d_weight = torch.empty((fps, 3, 3, 3), dtype=torch.float32, device=dev)
    need = load().nnue_ste_conv_backward_scratch(b, fps, gh, gw)
    if scratch is None:
        scratch = torch.empty((need,), dtype=torch.uint8, device=dev)
    _call("nnue_ste_conv_backward", images.data_ptr(), conv_out.data_ptr(), thr.data_ptr(), d_conv_out.data_ptr(),
          b, h, w, fps, stride, d_thr.data_ptr(), d_weight.data_ptr(), scratch.data_ptr(), scratch.numel(), int(stages),
          _stream(images))
    return d_thr, d_weight


def ste_conv_backward_patches(patches, weight, thr, d_conv_out, gh: int, gw: int,
                              d_thr: Optional[torch.Tensor] = None, d_weight: Optional[torch.Tensor] = None,
                              scratch: Optional[torch.Tensor] = None, stages: int = 3, conv_out: Optional[torch.Tensor] = None):
    """ste_conv_backward from the im2col form ftm_conv_binarize(patches=...) left (no images; conv_out is read when given,
    else re-formed from the patches and the conv weights); bitwise the same d_thr / d_weight / partials."""
    weight = _need(weight, torch.float32, "conv.weight")
    fps = weight.shape[0]
    b = d_conv_out.numel() // (fps * gh * gw)
    patches = _need(patches, torch.float32, "patches", (27, b * gh * gw))
    d_conv_out = _need(d_conv_out, torch.float32, "d_conv_out").view(b, fps, gh, gw)
    thr = _need(thr.reshape(-1), torch.float32, "threshold", (fps,))
    dev = patches.device
    if d_thr is None:
        d_thr = torch.empty((fps,), dtype=torch.float32, device=dev)
    if d_weight is None:
        d_weight = torch.empty((fps, 3, 3, 3), dtype=torch.float32, device=dev)
    need = load().nnue_ste_conv_backward_scratch(b, fps, gh, gw)
    if scratch is None:
        scratch = torch.empty((need,), dtype=torch.uint8, device=dev)
    if conv_out is not None:
        conv_out = _need(conv_out, torch.float32, "conv_out", (b, fps, gh, gw))
    _call("nnue_ste_conv_backward_patches", patches.data_ptr(), weight.data_ptr(), _ptr(conv_out), thr.data_ptr(), d_conv_out.data_ptr(), b, fps, gh, gw,
          d_thr.data_ptr(), d_weight.data_ptr(), scratch.data_ptr(), scratch.numel(), int(stages), _stream(patches))
    return d_thr, d_weight


def ste_conv_backward_chunks(b: int, fps: int, gh: int, gw: int) -> int:
    """Partials per output a stages=1 call leaves in its scratch."""
    return int(load().nnue_ste_conv_backward_chunks(b, fps, gh, gw))


# ---------------------------------------------------------------------------- FeatureTransformer
def ft_prepare(idx: torch.Tensor, val: torch.Tensor, num_rows: int) -> ActList:
    if idx.dim() != 2 or idx.shape != val.shape:
        raise ValueError(f"feature_indices / feature_values: expected matching [B,M], got "
                         f"{tuple(idx.shape)} and {tuple(val.shape)}")
    idx = _need(idx, torch.int64, "feature_indices")
    val = _need(val, torch.float32, "feature_values")
    b, m = idx.shape
    act = ActList.empty(b, m, num_rows, idx.device)
    _call("nnue_ft_prepare", idx.data_ptr(), val.data_ptr(), b, m, num_rows, act.rows.data_ptr(), act.pos.data_ptr(),
          act.coef.data_ptr(), act.n.data_ptr(), act.coefT.data_ptr(), act.ldb, _stream(idx))
    return act


def ft_forward(weight: torch.Tensor, bias: torch.Tensor, act: ActList,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    weight = _need(weight, torch.float32, "input.weight")
    f, l1 = weight.shape
    bias = _need(bias, torch.float32, "input.bias", (l1,))
    if act.coefT.shape[0] != f:
        raise ValueError("ft_forward: act list was built for a different table")
    if out is None:
        out = torch.empty((act.batch, l1), dtype=torch.float32, device=weight.device)
    _call("nnue_ft_forward", weight.data_ptr(), bias.data_ptr(), act.rows.data_ptr(), act.coef.data_ptr(),
          act.n.data_ptr(), act.cap, act.batch, f, l1, out.data_ptr(), _stream(weight))
    return out


def ft_backward_weight(d_out: torch.Tensor, act: ActList, num_rows: int,
                       d_weight: Optional[torch.Tensor] = None, d_bias: Optional[torch.Tensor] = None,
                       want_weight: bool = True, want_bias: bool = True):
    d_out = _need(d_out, torch.float32, "d_out")
    b, l1 = d_out.shape
    if b != act.batch or act.coefT.shape[0] != num_rows:
        raise ValueError("ft_backward_weight: act list does not match d_out / table")
    if want_weight and d_weight is None:
        d_weight = torch.empty((num_rows, l1), dtype=torch.float32, device=d_out.device)
    if want_bias and d_bias is None:
        d_bias = torch.empty((l1,), dtype=torch.float32, device=d_out.device)
    _call("nnue_ft_backward_weight", d_out.data_ptr(), act.coefT.data_ptr(), act.ldb, b, num_rows, l1,
          _ptr(d_weight if want_weight else None), _ptr(d_bias if want_bias else None), _stream(d_out))
    return d_weight, d_bias


def ft_backward_values(d_out: torch.Tensor, weight: torch.Tensor, act: ActList, dst_ld: int,
                       dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    d_out = _need(d_out, torch.float32, "d_out")
    weight = _need(weight, torch.float32, "input.weight")
    b, l1 = d_out.shape
    f = weight.shape[0]
    if weight.shape[1] != l1 or b != act.batch:
        raise ValueError("ft_backward_values: shape mismatch")
    if dst is None:
        dst = torch.empty((b, dst_ld), dtype=torch.float32, device=d_out.device)
    elif dst.numel() != b * dst_ld:
        raise ValueError("ft_backward_values: dst has the wrong size")
    _call("nnue_ft_backward_values", d_out.data_ptr(), weight.data_ptr(), act.rows.data_ptr(), act.pos.data_ptr(),
          act.n.data_ptr(), act.cap, b, f, l1, dst.data_ptr(), dst_ld, _stream(d_out))
    return dst


# ---------------------------------------------------------------------------- FeatureTransformer, binary features
@dataclass
class FeatureBits:
    """Bit-mask + tile-list form of the active features of one batch (layout: include/nnue_hip.h)."""
    maskW: torch.Tensor  # int64 [B, pw64]           position bits per sample
    maskT: torch.Tensor  # int64 [F+1, bw64]         sample bits per table row (+ bias row)
    sink: torch.Tensor   # float32 [B]               active ids >= F-1
    n: torch.Tensor      # int32 [B]                 active positions
    tlW: torch.Tensor    # int16 [B, tiles_fwd, 128] LDS offsets of the rows to add per sample and table tile
    tcW: torch.Tensor    # uint8 [B, tiles_fwd]
    tlT: torch.Tensor    # int16 [F+1, tiles_bwd, 128] LDS offsets of the samples to add per output row and batch tile
    tcT: torch.Tensor    # uint8 [F+1, tiles_bwd]
    scratch: torch.Tensor  # uint8, split-slab workspace of the gather kernels
    positions: int       # P = fps*Gh*Gw
    num_rows: int        # F

    @property
    def batch(self) -> int:
        return self.n.shape[0]

    @staticmethod
    def empty(batch: int, positions: int, num_rows: int, l1: int, device) -> "FeatureBits":
        pw64, bw64 = round_up(round_up(positions, 64) // 64, 2), round_up(round_up(batch, 64) // 64, 2)
        tf, tb = ctypes.c_int(), ctypes.c_int()
        if load().nnue_ftb_list_tiles(batch, num_rows, positions, ctypes.byref(tf), ctypes.byref(tb)) != 0:
            raise ValueError("FeatureBits: sizes must be positive")
        u8 = dict(dtype=torch.uint8, device=device)
        return FeatureBits(torch.empty((batch, pw64), dtype=torch.int64, device=device),
                           torch.empty((num_rows + 1, bw64), dtype=torch.int64, device=device),
                           torch.empty((batch,), dtype=torch.float32, device=device),
                           torch.empty((batch,), dtype=torch.int32, device=device),
                           torch.empty((batch, tf.value, 128), dtype=torch.int16, device=device), torch.empty((batch, tf.value), **u8),
                           torch.empty((num_rows + 1, tb.value, 128), dtype=torch.int16, device=device), torch.empty((num_rows + 1, tb.value), **u8),
                           torch.empty((max(16, int(load().nnue_ftb_scratch(batch, num_rows, positions, l1))),), **u8),
                           positions, num_rows)


def ftb_supported(l1: int) -> bool:
    return bool(load().nnue_ftb_supported(int(l1)))


def ft_path(num_rows: int, positions: int, l1: int, batch: int = 1) -> str:
    """Which FeatureTransformer kernels the fused path uses for the binary map: "mfma" | "bits" | "list".

    "mfma" = dense f32-MFMA products over the float {0,1} map (fastest at the reference's ~43 % density and still
    ahead at 1 %); "bits" = tile-list kernels that stage table tiles in LDS once per sample tile; "list" = id-list
    gather kernels (any shape).  NNUE_FT_PATH=mfma|bits|list forces one where the shape allows it; the default
    takes the first of mfma, bits, list that supports the shape."""
    mode = os.environ.get("NNUE_FT_PATH", "auto")
    # the product kernels address every operand with 32-bit byte offsets (batch-sized ones included)
    fits = (batch + 256) * positions * 4 < 2 ** 31 and (batch + 256) * l1 * 4 < 2 ** 31
    can_mfma, can_bits = fits and ftm_supported(num_rows, positions, l1), ftb_supported(l1)
    if mode == "list":
        return "list"
    if mode == "bits":
        return "bits" if can_bits else "list"
    if can_mfma:
        return "mfma"
    return "bits" if can_bits else "list"


def use_bit_path(num_rows: int, l1: int) -> bool:
    """True when the LDS-staged tile-list kernels are allowed for this width (see ft_path)."""
    return os.environ.get("NNUE_FT_PATH", "auto") != "list" and ftb_supported(l1)


def binarize_bits(conv_out: torch.Tensor, thr: torch.Tensor, num_rows: int, l1: int,
                  bits: Optional[FeatureBits] = None, stages: int = 3) -> FeatureBits:
    conv_out = _need(conv_out, torch.float32, "conv_out")
    if conv_out.dim() != 4:
        raise ValueError(f"conv_out: expected [B,fps,Gh,Gw], got {tuple(conv_out.shape)}")
    b, fps, gh, gw = conv_out.shape
    thr = _need(thr.reshape(-1), torch.float32, "threshold", (fps,))
    if bits is None:
        bits = FeatureBits.empty(b, fps * gh * gw, num_rows, l1, conv_out.device)
    elif bits.batch != b or bits.positions != fps * gh * gw or bits.num_rows != num_rows:
        raise ValueError("binarize_bits: bit buffers do not match the map")
    _call("nnue_binarize_bits", conv_out.data_ptr(), thr.data_ptr(), b, fps, gh, gw, num_rows, bits.maskW.data_ptr(),
          bits.maskW.shape[1], bits.maskT.data_ptr(), bits.maskT.shape[1], bits.sink.data_ptr(), bits.n.data_ptr(),
          bits.tlW.data_ptr(), bits.tcW.data_ptr(), bits.tlT.data_ptr(), bits.tcT.data_ptr(), int(stages), _stream(conv_out))
    return bits


def ftb_forward(weight: torch.Tensor, bias: torch.Tensor, bits: FeatureBits, out: Optional[torch.Tensor] = None):
    weight = _need(weight, torch.float32, "input.weight")
    f, l1 = weight.shape
    bias = _need(bias, torch.float32, "input.bias", (l1,))
    if f != bits.num_rows:
        raise ValueError("ftb_forward: bits were built for a different table")
    if out is None:
        out = torch.empty((bits.batch, l1), dtype=torch.float32, device=weight.device)
    _call("nnue_ftb_forward", weight.data_ptr(), bias.data_ptr(), bits.tlW.data_ptr(), bits.tcW.data_ptr(),
          bits.sink.data_ptr(), bits.batch, f, bits.positions, l1, out.data_ptr(), bits.scratch.data_ptr(),
          bits.scratch.numel(), _stream(weight))
    return out


def ftb_backward_weight(d_out: torch.Tensor, bits: FeatureBits, d_weight: Optional[torch.Tensor] = None,
                        d_bias: Optional[torch.Tensor] = None, want_weight: bool = True, want_bias: bool = True):
    d_out = _need(d_out, torch.float32, "d_out")
    b, l1 = d_out.shape
    if b != bits.batch:
        raise ValueError("ftb_backward_weight: bits do not match d_out")
    if want_weight and d_weight is None:
        d_weight = torch.empty((bits.num_rows, l1), dtype=torch.float32, device=d_out.device)
    if want_bias and d_bias is None:
        d_bias = torch.empty((l1,), dtype=torch.float32, device=d_out.device)
    _call("nnue_ftb_backward_weight", d_out.data_ptr(), bits.tlT.data_ptr(), bits.tcT.data_ptr(), bits.sink.data_ptr(),
          b, bits.num_rows, bits.positions, l1, _ptr(d_weight if want_weight else None),
          _ptr(d_bias if want_bias else None), bits.scratch.data_ptr(), bits.scratch.numel(), _stream(d_out))
    return d_weight, d_bias


def ftb_backward_values(d_out: torch.Tensor, weight: torch.Tensor, bits: FeatureBits,
                        dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    d_out = _need(d_out, torch.float32, "d_out")
    weight = _need(weight, torch.float32, "input.weight")
    b, l1 = d_out.shape
    if weight.shape != (bits.num_rows, l1) or b != bits.batch:
        raise ValueError("ftb_backward_values: shape mismatch")
    if dst is None:
        dst = torch.empty((b, bits.positions), dtype=torch.float32, device=d_out.device)
    elif dst.numel() != b * bits.positions:
        raise ValueError("ftb_backward_values: dst has the wrong size")
    _call("nnue_ftb_backward_values", d_out.data_ptr(), weight.data_ptr(), bits.maskW.data_ptr(), bits.maskW.shape[1],
          b, bits.num_rows, bits.positions, l1, dst.data_ptr(), _stream(d_out))
    return dst


# ---------------------------------------------------------------------------- FeatureTransformer, dense MFMA form
def ftm_supported(num_rows: int, positions: int, l1: int) -> bool:
    return bool(load().nnue_ftm_supported(int(num_rows), int(positions), int(l1)))


def ftm_scratch_bytes(b: int, num_rows: int, positions: int, l1: int) -> int:
    return int(load().nnue_ftm_scratch(int(b), int(num_rows), int(positions), int(l1)))


class FeatureMatrix:
    """The binary map of one batch as a byte {0,1} matrix (layout: include/nnue_hip.h, nnue_ftm_*)."""
    bits: torch.Tensor     # uint8 [B, P]
    n: torch.Tensor        # int32 [B]    active positions
    sink: torch.Tensor     # float32 [B]  active positions >= F-1
    scratch: torch.Tensor  # uint8, split-K slabs of the forward
    positions: int
    num_rows: int

    def __init__(self, bits, n, sink, scratch, positions, num_rows):
        self.bits, self.n, self.sink, self.scratch, self.positions, self.num_rows = bits, n, sink, scratch, positions, num_rows

    @property
    def batch(self) -> int:
        return self.n.shape[0]

    @staticmethod
    def empty(batch: int, positions: int, num_rows: int, l1: int, device) -> "FeatureMatrix":
        return FeatureMatrix(torch.empty((batch, positions), dtype=torch.uint8, device=device),
                             torch.empty((batch,), dtype=torch.int32, device=device),
                             torch.empty((batch,), dtype=torch.float32, device=device),
                             torch.empty((max(16, ftm_scratch_bytes(batch, num_rows, positions, l1),
                                              int(load().nnue_ftm_backward_values_scratch(batch, num_rows, positions, l1))),),
                                         dtype=torch.uint8, device=device),
                             positions, num_rows)


def ftm_binarize(conv_out: torch.Tensor, thr: torch.Tensor, num_rows: int, l1: int,
                 fm: Optional[FeatureMatrix] = None) -> FeatureMatrix:
    conv_out = _need(conv_out, torch.float32, "conv_out")
    if conv_out.dim() != 4:
        raise ValueError(f"conv_out: expected [B,fps,Gh,Gw], got {tuple(conv_out.shape)}")
    b, fps, gh, gw = conv_out.shape
    thr = _need(thr.reshape(-1), torch.float32, "threshold", (fps,))
    if fm is None:
        fm = FeatureMatrix.empty(b, fps * gh * gw, num_rows, l1, conv_out.device)
    elif fm.batch != b or fm.positions != fps * gh * gw or fm.num_rows != num_rows:
        raise ValueError("ftm_binarize: buffers do not match the map")
    _call("nnue_ftm_binarize", conv_out.data_ptr(), thr.data_ptr(), b, fps, gh, gw, int(num_rows), fm.bits.data_ptr(),
          fm.n.data_ptr(), fm.sink.data_ptr(), _stream(conv_out))
    return fm


def ftm_conv_binarize(images: torch.Tensor, weight: torch.Tensor, thr: torch.Tensor, stride: int, num_rows: int, l1: int,
                      conv_out: Optional[torch.Tensor] = None, fm: Optional[FeatureMatrix] = None,
                      patches: Optional[torch.Tensor] = None, write_conv_out: bool = True):
    """conv3x3_forward + ftm_binarize in one launch; returns (conv_out, fm), bitwise the two separate calls.
    patches (float32 [27, B*Gh*Gw]): the launch also leaves the im2col form of the images (ste_conv_backward_patches reads
    it); with write_conv_out=False conv_out is then not written (and None is returned in its place)."""
    images = _need(images, torch.float32, "images")
    if images.dim() != 4 or images.shape[1] != 3:
        raise ValueError(f"images: expected [B,3,H,W], got {tuple(images.shape)}")
    b, _, h, w = images.shape
    weight = _need(weight, torch.float32, "conv.weight")
    fps = weight.shape[0]
    if tuple(weight.shape) != (fps, 3, 3, 3):
        raise ValueError("conv.weight: expected [fps,3,3,3]")
    thr = _need(thr.reshape(-1), torch.float32, "threshold", (fps,))
    gh, gw = conv_out_hw(h, w, stride)
    if patches is None and not write_conv_out:
        raise ValueError("ftm_conv_binarize: without conv_out the patches are needed")
    if not write_conv_out:
        conv_out = None
    elif conv_out is None:
        conv_out = torch.empty((b, fps, gh, gw), dtype=torch.float32, device=images.device)
    elif tuple(conv_out.shape) != (b, fps, gh, gw):
        raise ValueError("ftm_conv_binarize: conv_out has the wrong shape")
    if fm is None:
        fm = FeatureMatrix.empty(b, fps * gh * gw, num_rows, l1, images.device)
    elif fm.batch != b or fm.positions != fps * gh * gw or fm.num_rows != num_rows:
        raise ValueError("ftm_conv_binarize: buffers do not match the map")
    if patches is not None:
        patches = _need(patches, torch.float32, "patches", (27, b * gh * gw))
        _call("nnue_ftm_conv_binarize_patches", images.data_ptr(), weight.data_ptr(), thr.data_ptr(), b, h, w, fps, int(stride), int(num_rows),
              patches.data_ptr(), _ptr(conv_out), fm.bits.data_ptr(), fm.n.data_ptr(), fm.sink.data_ptr(), _stream(images))
        return conv_out, fm
    _call("nnue_ftm_conv_binarize", images.data_ptr(), weight.data_ptr(), thr.data_ptr(), b, h, w, fps, int(stride), int(num_rows),
          conv_out.data_ptr(), fm.bits.data_ptr(), fm.n.data_ptr(), fm.sink.data_ptr(), _stream(images))
    return conv_out, fm


def ftm_forward(weight: torch.Tensor, bias: torch.Tensor, fm: FeatureMatrix, out: Optional[torch.Tensor] = None,
                group=None) -> torch.Tensor:
    """group (a BucketPlan): the bucket grouping of this batch (from fm.n) rides in the launch as one extra workgroup."""
    weight = _need(weight, torch.float32, "input.weight")
    f, l1 = weight.shape
    bias = _need(bias, torch.float32, "input.bias", (l1,))
    if f != fm.num_rows:
        raise ValueError("ftm_forward: the map was built for a different table")
    if out is None:
        out = torch.empty((fm.batch, l1), dtype=torch.float32, device=weight.device)
    args = (fm.bits.data_ptr(), fm.sink.data_ptr(), weight.data_ptr(), bias.data_ptr(), fm.batch, f,
            fm.positions, l1, out.data_ptr(), fm.scratch.data_ptr(), fm.scratch.numel())
    if group is None:
        _call("nnue_ftm_forward", *args, _stream(weight))
    else:
        if group.batch != fm.batch:
            raise ValueError("ftm_forward: the bucket plan was built for another batch size")
        _call("nnue_ftm_forward_grouping", *args, fm.n.data_ptr(), group.K, group.bucket.data_ptr(), group.rows.data_ptr(),
              group.tile_bucket.data_ptr(), group.seg.data_ptr(), _stream(weight))
    return out


def ftm_forward_l1_supported(batch: int, num_rows: int, positions: int, l1: int, l2: int) -> bool:
    return bool(load().nnue_ftm_forward_l1_supported(int(batch), int(num_rows), int(positions), int(l1), int(l2)))


def ftm_forward_l1(weight: torch.Tensor, bias: torch.Tensor, fm: FeatureMatrix, w1: torch.Tensor, part: torch.Tensor,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ftm_forward + the classifier's layer-1 slabs part[L1/64][B][L2] (written at the start of `part`, the
    classifier's scratch buffer; consumed by classifier_train_step(phases | 8))."""
    weight = _need(weight, torch.float32, "input.weight")
    f, l1 = weight.shape
    bias = _need(bias, torch.float32, "input.bias", (l1,))
    l2 = w1.shape[0]
    w1 = _need(w1, torch.float32, "classifier.0.weight", (l2, l1))
    if f != fm.num_rows:
        raise ValueError("ftm_forward_l1: the map was built for a different table")
    if not part.is_cuda or part.numel() * part.element_size() < (l1 // 64) * fm.batch * l2 * 4:
        raise ValueError("ftm_forward_l1: part buffer too small for [L1/64][B][L2] floats")
    if out is None:
        out = torch.empty((fm.batch, l1), dtype=torch.float32, device=weight.device)
    _call("nnue_ftm_forward_l1", fm.bits.data_ptr(), fm.sink.data_ptr(), weight.data_ptr(), bias.data_ptr(), w1.data_ptr(),
          fm.batch, f, fm.positions, l1, l2, out.data_ptr(), part.data_ptr(), _stream(weight))
    return out


def ftm_backward_weight(d_out: torch.Tensor, fm: FeatureMatrix, d_weight: Optional[torch.Tensor] = None,
                        d_bias: Optional[torch.Tensor] = None, want_weight: bool = True, want_bias: bool = True):
    d_out = _need(d_out, torch.float32, "d_out")
    if d_out.dim() != 2 or d_out.shape[0] != fm.batch:
        raise ValueError("ftm_backward_weight: d_out does not match the map")
    l1 = d_out.shape[1]
    if want_weight and d_weight is None:
        d_weight = torch.empty((fm.num_rows, l1), dtype=torch.float32, device=d_out.device)
    if want_bias and d_bias is None:
        d_bias = torch.empty((l1,), dtype=torch.float32, device=d_out.device)
    _call("nnue_ftm_backward_weight", fm.bits.data_ptr(), fm.sink.data_ptr(), d_out.data_ptr(), fm.batch, fm.num_rows,
          fm.positions, l1, _ptr(d_weight if want_weight else None), _ptr(d_bias if want_bias else None), _stream(d_out))
    return d_weight, d_bias


def ftm_backward_values(d_out: torch.Tensor, weight: torch.Tensor, fm: FeatureMatrix,
                        dst: Optional[torch.Tensor] = None) -> torch.Tensor:
    d_out = _need(d_out, torch.float32, "d_out")
    weight = _need(weight, torch.float32, "input.weight")
    b, l1 = d_out.shape
    if tuple(weight.shape) != (fm.num_rows, l1) or b != fm.batch:
        raise ValueError("ftm_backward_values: shape mismatch")
    if dst is None:
        dst = torch.empty((b, fm.positions), dtype=torch.float32, device=d_out.device)
    elif dst.numel() != b * fm.positions:
        raise ValueError("ftm_backward_values: dst has the wrong size")
    # big maps: d_out is split once into bf16 planes in a workspace (csrc/ftv_kernels.hip) -- the forward's split-K scratch,
    # free at this point of a step, serves when it is large enough
    need = int(load().nnue_ftm_backward_values_scratch(b, fm.num_rows, fm.positions, l1))
    ws = fm.scratch if (need and fm.scratch is not None and fm.scratch.numel() >= need) else None
    if need and ws is None:
        ws = fm.scratch = torch.empty((need,), dtype=torch.uint8, device=d_out.device)
    _call("nnue_ftm_backward_values_ws", fm.bits.data_ptr(), d_out.data_ptr(), weight.data_ptr(), b, fm.num_rows, fm.positions,
          l1, dst.data_ptr(), _ptr(ws) if need else None, need, _stream(d_out))
    return dst


def ftm_backward_sq_count(b: int, f: int, p: int, l1: int) -> int:
    """Floats ftm_backward's sq_partial receives (one sum of squares per weight-gradient tile); 0 = not available."""
    return int(load().nnue_ftm_backward_sq_count(b, f, p, l1))


def ftm_backward_cw_supported(b: int, f: int, p: int, l1: int, l2: int) -> bool:
    """Shapes whose merged backward launch can also carry the classifier's first-layer weight gradient."""
    return bool(load().nnue_ftm_backward_cw_supported(b, f, p, l1, l2))


def ftm_backward(d_out: torch.Tensor, weight: torch.Tensor, fm: FeatureMatrix, d_weight: Optional[torch.Tensor] = None,
                 d_bias: Optional[torch.Tensor] = None, dst: Optional[torch.Tensor] = None, ft: Optional[torch.Tensor] = None,
                 d_z1: Optional[torch.Tensor] = None, d_w1: Optional[torch.Tensor] = None,
                 sq_partial: Optional[torch.Tensor] = None, buckets=None, small: Optional["NnueClsRider"] = None):
    """(d_weight, d_bias, d_conv_out) in one launch; bitwise the results of ftm_backward_weight + ftm_backward_values.
    small (classifier_train_rider(...)): the classifier's small gradients + mean loss run as one more tile family of the launch.
    With ft [B, L1], d_z1 [B, L2] and d_w1 [L2, L1] the launch also writes d_w1 = d_z1^T l0 (the pairwise block of ft).
    buckets (a BucketPlan): d_w1 [K, L2, L1]; ft / d_z1 are then the GROUPED-row copies [tiles*16, .] the bucketed
    classifier step left in its scratch."""
    d_out = _need(d_out, torch.float32, "d_out")
    weight = _need(weight, torch.float32, "input.weight")
    b, l1 = d_out.shape
    if weight.shape != (fm.num_rows, l1) or b != fm.batch:
        raise ValueError("ftm_backward: shape mismatch")
    if d_weight is None:
        d_weight = torch.empty((fm.num_rows, l1), dtype=torch.float32, device=d_out.device)
    if d_bias is None:
        d_bias = torch.empty((l1,), dtype=torch.float32, device=d_out.device)
    if dst is None:
        dst = torch.empty((b, fm.positions), dtype=torch.float32, device=d_out.device)
    elif dst.numel() != b * fm.positions:
        raise ValueError("ftm_backward: dst has the wrong size")
    l2 = 0
    rows = b if buckets is None else buckets.tiles * 16
    if d_w1 is not None:
        ft = _need(ft, torch.float32, "ft")
        d_z1 = _need(d_z1, torch.float32, "d_z1")
        l2 = d_w1.shape[-2]
        want = (l2, l1) if buckets is None else (buckets.K, l2, l1)
        if ft.shape != (rows, l1) or d_z1.numel() != rows * l2 or tuple(d_w1.shape) != want or not d_w1.is_contiguous():
            raise ValueError("ftm_backward: ft / d_z1 / d_w1 shape mismatch")
    args = (fm.bits.data_ptr(), fm.sink.data_ptr(), d_out.data_ptr(), weight.data_ptr(), b, fm.num_rows,
            fm.positions, l1, d_weight.data_ptr(), d_bias.data_ptr(), dst.data_ptr(),
            _ptr(ft if d_w1 is not None else None), _ptr(d_z1 if d_w1 is not None else None), l2, _ptr(d_w1), _ptr(sq_partial))
    rider = ctypes.pointer(small) if small is not None else None
    if buckets is None:
        _call("nnue_ftm_backward", *args, rider, _stream(d_out))
    else:
        _call("nnue_ftm_backward_bucketed", *args, buckets.K, buckets.seg.data_ptr(), rows, rider, _stream(d_out))
    return d_weight, d_bias, dst


def classifier_train_rider(pairwise: bool, b: int, l1: int, l2: int, l3: int, c: int, h1, h2, sample_loss, loss, grads, scratch,
                           buckets: Optional["BucketPlan"] = None) -> "NnueClsRider":
    """Arguments of the small-gradient tile family for ftm_backward(small=...) when classifier_train_step runs with phases bit 32.
    grads: the six classifier gradients (d_w1, d_b1, d_w2, d_b2, d_w3, d_b3) as classifier_train_step takes them.  The returned
    struct lives on the host and holds device pointers: keep it alive as long as a recorded plan refers to it."""
    out = NnueClsRider()
    _, d_b1, d_w2, d_b2, d_w3, d_b3 = grads
    rc = load().nnue_classifier_train_rider(int(bool(pairwise)), b, l1, l2, l3, c, h1.data_ptr(), h2.data_ptr(), sample_loss.data_ptr(),
                                            loss.data_ptr(), d_b1.data_ptr(), d_w2.data_ptr(), d_b2.data_ptr(), d_w3.data_ptr(),
                                            d_b3.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                            buckets.ref if buckets is not None else None, ctypes.byref(out))
    if rc != 0:
        raise NnueHipError(f"nnue_classifier_train_rider failed (code {rc}): {load().nnue_hip_last_error().decode()}")
    return out


# ---------------------------------------------------------------------------- bucketed layer stacks
class BucketPlan:
    """Per-batch bucket assignment + grouping (struct nnue_buckets): which of the K layer stacks each sample uses and
    the bucket-homogeneous 16-row tiles the first-layer MFMA kernels walk.  Buffers are static (hipGraph-friendly);
    ``bucket_group`` refills them from the active-feature counts of a batch."""

    def __init__(self, batch: int, buckets: int, device):
        self.batch, self.K = int(batch), int(buckets)
        self.tiles = int(load().nnue_bucket_tile_count(self.batch, self.K))
        i32 = dict(dtype=torch.int32, device=device)
        self.bucket = torch.zeros((self.batch,), **i32)
        self.rows = torch.full((self.tiles * 16,), -1, **i32)
        self.tile_bucket = torch.full((self.tiles,), -1, **i32)
        self.seg = torch.zeros((self.K + 1,), **i32)
        self.cstruct = NnueBuckets(self.K, self.bucket.data_ptr(), self.rows.data_ptr(), self.tile_bucket.data_ptr(),
                                   self.seg.data_ptr(), self.tiles)

    @property
    def ref(self):
        return ctypes.byref(self.cstruct)


def bucket_group(n: torch.Tensor, positions: int, buckets: int, plan: Optional[BucketPlan] = None) -> BucketPlan:
    """bucket[b] = min(K-1, n[b]*K // (positions+1)) and the grouping; positions == 0: n holds the bucket ids themselves."""
    n = _need(n, torch.int32, "active-feature counts")
    b = n.numel()
    if plan is None:
        plan = BucketPlan(b, buckets, n.device)
    elif plan.batch != b or plan.K != buckets:
        raise ValueError("bucket_group: plan was built for another batch size / bucket count")
    _call("nnue_bucket_group", n.data_ptr(), b, int(positions), int(buckets), plan.bucket.data_ptr(), plan.rows.data_ptr(),
          plan.tile_bucket.data_ptr(), plan.seg.data_ptr(), _stream(n))
    return plan


# ---------------------------------------------------------------------------- classifier
def classifier_scratch_bytes(b: int, l1: int, l2: int, l3: int, buckets: int = 1) -> int:
    return int(load().nnue_classifier_scratch_bucketed(b, l1, l2, l3, buckets))


def _cls_shapes(w1, b1, w2, b2, w3, b3, l1: int, buckets: Optional[BucketPlan]):
    """Checks the six (possibly stacked) tensors; returns them contiguous with (l2, l3, c, K)."""
    k = buckets.K if buckets is not None else 1
    lead = (k,) if w1.dim() == 3 else ()
    if (w1.dim() == 3) != (buckets is not None) or (lead and w1.shape[0] != k):
        raise ValueError("stacked [K, out, in] classifier weights need a BucketPlan with the same K (and vice versa)")
    l2, l3, c = w1.shape[-2], w2.shape[-2], w3.shape[-2]
    w1 = _need(w1, torch.float32, "classifier.0.weight", (*lead, l2, l1))
    w2 = _need(w2, torch.float32, "classifier.2.weight", (*lead, l3, l2))
    w3 = _need(w3, torch.float32, "classifier.4.weight", (*lead, c, l3))
    if b1 is not None:
        b1 = _need(b1, torch.float32, "classifier.0.bias", (*lead, l2))
        b2 = _need(b2, torch.float32, "classifier.2.bias", (*lead, l3))
        b3 = _need(b3, torch.float32, "classifier.4.bias", (*lead, c))
    return w1, b1, w2, b2, w3, b3, l2, l3, c, k


def classifier_forward(x, pairwise: bool, w1, b1, w2, b2, w3, b3, clip: float = 0.0,
                       scratch: Optional[torch.Tensor] = None, out=None, buckets: Optional[BucketPlan] = None):
    x = _need(x, torch.float32, "classifier input")
    if x.dim() != 2:
        raise ValueError(f"classifier input: expected [B,L1], got {tuple(x.shape)}")
    b, l1 = x.shape
    w1, b1, w2, b2, w3, b3, l2, l3, c, k = _cls_shapes(w1, b1, w2, b2, w3, b3, l1, buckets)
    dev = x.device
    if scratch is None:
        scratch = torch.empty((classifier_scratch_bytes(b, l1, l2, l3, k),), dtype=torch.uint8, device=dev)
    if out is None:
        h1 = torch.empty((b, l2), dtype=torch.float32, device=dev)
        h2 = torch.empty((b, l3), dtype=torch.float32, device=dev)
        logits = torch.empty((b, c), dtype=torch.float32, device=dev)
    else:
        h1, h2, logits = out
    args = (x.data_ptr(), int(bool(pairwise)), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
            b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), float(clip), b, l1, l2, l3, c, h1.data_ptr(), h2.data_ptr(),
            logits.data_ptr(), scratch.data_ptr(), scratch.numel())
    if buckets is None:
        _call("nnue_classifier_forward", *args, _stream(x))
    else:
        _call("nnue_classifier_forward_bucketed", *args, buckets.ref, _stream(x))
    return h1, h2, logits


def classifier_backward(x, pairwise: bool, w1, w2, w3, h1, h2, d_logits, clip: float = 0.0,
                        want_dx: bool = True, scratch: Optional[torch.Tensor] = None, grads=None, d_x=None,
                        buckets: Optional[BucketPlan] = None):
    x = _need(x, torch.float32, "classifier input")
    b, l1 = x.shape
    w1, _, w2, _, w3, _, l2, l3, c, k = _cls_shapes(w1, None, w2, None, w3, None, l1, buckets)
    h1 = _need(h1, torch.float32, "h1", (b, l2))
    h2 = _need(h2, torch.float32, "h2", (b, l3))
    d_logits = _need(d_logits, torch.float32, "d_logits", (b, c))
    dev = x.device
    lead = (k,) if buckets is not None else ()
    if scratch is None:
        scratch = torch.empty((classifier_scratch_bytes(b, l1, l2, l3, k),), dtype=torch.uint8, device=dev)
    if grads is None:
        mk = lambda *s: torch.empty((*lead, *s), dtype=torch.float32, device=dev)  # noqa: E731
        grads = (mk(l2, l1), mk(l2), mk(l3, l2), mk(l3), mk(c, l3), mk(c))
    if want_dx and d_x is None:
        d_x = torch.empty((b, l1), dtype=torch.float32, device=dev)
    args = (x.data_ptr(), int(bool(pairwise)), w1.data_ptr(), w2.data_ptr(), w3.data_ptr(),
            float(clip), h1.data_ptr(), h2.data_ptr(), d_logits.data_ptr(), b, l1, l2, l3, c,
            _ptr(d_x if want_dx else None), *[g.data_ptr() for g in grads], scratch.data_ptr(), scratch.numel())
    if buckets is None:
        _call("nnue_classifier_backward", *args, _stream(x))
    else:
        _call("nnue_classifier_backward_bucketed", *args, buckets.ref, _stream(x))
    return (d_x if want_dx else None), grads


def classifier_train_scratch_bytes(b: int, l1: int, l2: int, l3: int, c: int, buckets: int = 1) -> int:
    return int(load().nnue_classifier_train_scratch_bucketed(b, l1, l2, l3, c, buckets))


def classifier_train_dz1_offset(b: int, l1: int, l2: int, l3: int, c: int, pairwise: bool) -> int:
    """Byte offset of d_z1 [B, L2] inside the classifier's training scratch (valid after phase 1)."""
    return int(load().nnue_classifier_train_dz1_offset(b, l1, l2, l3, c, int(bool(pairwise))))


def classifier_train_grouped_offsets(b: int, l1: int, l2: int, l3: int, c: int, buckets: int):
    """Byte offsets of (d_z1 grouped [tiles*16, L2], x grouped [tiles*16, L1]) inside the bucketed training scratch."""
    lib = load()
    return (int(lib.nnue_classifier_train_dz1_grouped_offset(b, l1, l2, l3, c, buckets)),
            int(lib.nnue_classifier_train_x_grouped_offset(b, l1, l2, l3, c, buckets)))


def classifier_train_step(x, pairwise: bool, w1, b1, w2, b2, w3, b3, labels, grad_scale: float = 1.0, clip: float = 0.0,
                          want_dx: bool = True, scratch: Optional[torch.Tensor] = None, out=None, loss_out=None,
                          grads=None, d_x=None, phases: int = 3, buckets: Optional[BucketPlan] = None):
    """Forward + mean cross-entropy + backward of the classifier block in one C call.
    Returns (h1, h2, logits), (sample_loss, loss), d_x, grads."""
    x = _need(x, torch.float32, "classifier input")
    b, l1 = x.shape
    w1, b1, w2, b2, w3, b3, l2, l3, c, k = _cls_shapes(w1, b1, w2, b2, w3, b3, l1, buckets)
    labels = _need(labels, torch.int64, "labels", (b,))
    dev = x.device
    lead = (k,) if buckets is not None else ()
    mk = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
    if scratch is None:
        scratch = torch.empty((classifier_train_scratch_bytes(b, l1, l2, l3, c, k),), dtype=torch.uint8, device=dev)
    h1, h2, logits = out if out is not None else (mk(b, l2), mk(b, l3), mk(b, c))
    sample_loss, loss = loss_out if loss_out is not None else (mk(b), mk())
    if grads is None:
        grads = (mk(*lead, l2, l1), mk(*lead, l2), mk(*lead, l3, l2), mk(*lead, l3), mk(*lead, c, l3), mk(*lead, c))
    if want_dx and d_x is None:
        d_x = mk(b, l1)
    args = (x.data_ptr(), int(bool(pairwise)), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
            b2.data_ptr(), w3.data_ptr(), b3.data_ptr(), float(clip), labels.data_ptr(), float(grad_scale), b, l1, l2, l3, c,
            h1.data_ptr(), h2.data_ptr(), logits.data_ptr(), sample_loss.data_ptr(), loss.data_ptr(),
            _ptr(d_x if want_dx else None), *[g.data_ptr() for g in grads], scratch.data_ptr(), scratch.numel(), int(phases))
    if buckets is None:
        _call("nnue_classifier_train_step", *args, _stream(x))
    else:
        _call("nnue_classifier_train_step_bucketed", *args, buckets.ref, _stream(x))
    return (h1, h2, logits), (sample_loss, loss), (d_x if want_dx else None), grads


# ---------------------------------------------------------------------------- loss + step tail
def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, grad_scale: float = 1.0, want_grad: bool = True,
                  out=None):
    logits = _need(logits, torch.float32, "logits")
    b, c = logits.shape
    labels = _need(labels, torch.int64, "labels", (b,))
    dev = logits.device
    if out is None:
        sample_loss = torch.empty((b,), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        d_logits = torch.empty((b, c), dtype=torch.float32, device=dev) if want_grad else None
    else:
        sample_loss, loss, d_logits = out
    _call("nnue_cross_entropy", logits.data_ptr(), labels.data_ptr(), b, c, float(grad_scale), sample_loss.data_ptr(),
          loss.data_ptr(), _ptr(d_logits), _stream(logits))
    return sample_loss, loss, d_logits


def confusion_accumulate(logits: torch.Tensor, labels: torch.Tensor, confusion: Optional[torch.Tensor] = None) -> torch.Tensor:
    """confusion[truth, pred] += 1 for every row of the batch (int64 [K, K], K = C or 2 when C == 1)."""
    logits = _need(logits, torch.float32, "logits")
    b, c = logits.shape
    labels = _need(labels, torch.int64, "labels", (b,))
    k = 2 if c == 1 else c
    if confusion is None:
        confusion = torch.zeros((k, k), dtype=torch.int64, device=logits.device)
    else:
        _need(confusion, torch.int64, "confusion", (k, k))
    _call("nnue_confusion_accumulate", logits.data_ptr(), labels.data_ptr(), b, c, confusion.data_ptr(), _stream(logits))
    return confusion


def sgd_scratch_bytes(count: int) -> int:
    return int(load().nnue_sgd_scratch(count))


def sqnorm_partials(grads: torch.Tensor, partial: torch.Tensor) -> torch.Tensor:
    """partial[i] = block partial sums of grads^2 (their sum is ||grads||^2); fixed order."""
    grads = _need(grads, torch.float32, "gradient shard")
    _need(partial, torch.float32, "partials")
    _call("nnue_sqnorm_partials", grads.data_ptr(), grads.numel(), partial.data_ptr(), partial.numel(), _stream(grads))
    return partial


def ftm_gram_scratch(fm: "FeatureMatrix") -> int:
    """Floats of scratch ftm_gram_sqnorm needs for this map."""
    return int(load().nnue_ftm_gram_scratch(fm.bits.shape[0], fm.num_rows, fm.positions))


def ftm_gram_sqnorm(fm: "FeatureMatrix", d_out: torch.Tensor, gram: torch.Tensor, sq_partial: torch.Tensor,
                    tail=None) -> torch.Tensor:
    """Partial sums of ||A^T d_out||_F^2 over the table rows the map reaches, from two B x B Gram matrices.  ``gram``:
    ``ftm_gram_scratch(fm)`` floats of scratch; its first B*B hold A A^T afterwards.  tail = (d_weight, d_bias): the workgroups of
    ftm_backward_tail_rows ride in the first launch (same results, one launch fewer)."""
    d_out = _need(d_out, torch.float32, "d_out")
    b, l1 = d_out.shape
    _need(gram, torch.float32, "gram scratch", (ftm_gram_scratch(fm),))
    _need(sq_partial, torch.float32, "gram partials", (int(load().nnue_ftm_gram_sq_count(b, l1)),))
    if tail is not None:
        d_weight, d_bias = tail
        _call("nnue_ftm_gram_sqnorm_tail", fm.bits.data_ptr(), fm.sink.data_ptr(), d_out.data_ptr(), b, fm.num_rows, fm.positions, l1,
              gram.data_ptr(), sq_partial.data_ptr(), d_weight.data_ptr(), d_bias.data_ptr(), _stream(d_out))
        return sq_partial
    _call("nnue_ftm_gram_sqnorm", fm.bits.data_ptr(), d_out.data_ptr(), b, fm.num_rows, fm.positions, l1, gram.data_ptr(),
          sq_partial.data_ptr(), _stream(d_out))
    return sq_partial


def ftm_backward_tail_rows(d_out: torch.Tensor, fm: "FeatureMatrix", d_weight: torch.Tensor, d_bias: torch.Tensor) -> None:
    d_out = _need(d_out, torch.float32, "d_out")
    b, l1 = d_out.shape
    _call("nnue_ftm_backward_tail_rows", fm.sink.data_ptr(), d_out.data_ptr(), b, fm.num_rows, fm.positions, l1, d_weight.data_ptr(),
          d_bias.data_ptr(), _stream(d_out))


def ftm_backward_weight_update(d_out: torch.Tensor, fm: "FeatureMatrix", weight: torch.Tensor, momentum_rows: Optional[torch.Tensor],
                               coef: torch.Tensor, lr: float, momentum: float, weight_decay: float, grad_scale: float,
                               first_step: bool, lr_dev: Optional[torch.Tensor] = None) -> None:
    """weight rows [0, direct) <- SGD update with d_W = A^T d_out formed and consumed in the product's epilogue.
    lr_dev (device float32 scalar): the learning rate is read from it instead of `lr`."""
    d_out = _need(d_out, torch.float32, "d_out")
    b, l1 = d_out.shape
    weight = _need(weight, torch.float32, "input.weight", (fm.num_rows, l1))
    _need(coef, torch.float32, "clip coefficient")
    _call("nnue_ftm_backward_weight_update", fm.bits.data_ptr(), d_out.data_ptr(), b, fm.num_rows, fm.positions, l1, weight.data_ptr(),
          _ptr(momentum_rows), coef.data_ptr(), float(lr), float(momentum), float(weight_decay), float(grad_scale),
          int(bool(first_step)), _ptr(lr_dev), _stream(d_out))


def ftm_update_forward_supported(batch: int, num_rows: int, positions: int, l1: int, batch_next: Optional[int] = None) -> bool:
    """batch: rows of the gradient's factors (the global batch under the factor exchange); batch_next: rows of the next map."""
    return bool(load().nnue_ftm_update_forward_supported(int(batch), int(batch if batch_next is None else batch_next), int(num_rows),
                                                         int(positions), int(l1)))


def ftm_backward_weight_update_forward(d_out: torch.Tensor, fm: "FeatureMatrix", weight: torch.Tensor,
                                       momentum_rows: Optional[torch.Tensor], coef: torch.Tensor, lr: float, momentum: float,
                                       weight_decay: float, grad_scale: float, first_step: bool, fm_next: "FeatureMatrix",
                                       bias: torch.Tensor, out_next: torch.Tensor, lr_dev: Optional[torch.Tensor] = None) -> None:
    """ftm_backward_weight_update(d_out, fm, ...) and ftm_forward(weight, bias, fm_next, out=out_next) in one pass over the
    table (bitwise the two calls).  `bias` and table row F-1 must already be updated (sgd_step)."""
    d_out = _need(d_out, torch.float32, "d_out")
    b, l1 = d_out.shape
    weight = _need(weight, torch.float32, "input.weight", (fm.num_rows, l1))
    bias = _need(bias, torch.float32, "input.bias", (l1,))
    out_next = _need(out_next, torch.float32, "out_next", (fm_next.batch, l1))
    _need(coef, torch.float32, "clip coefficient")
    if fm.batch != b or fm_next.positions != fm.positions or fm_next.num_rows != fm.num_rows:
        raise ValueError("ftm_backward_weight_update_forward: the maps do not match d_out / each other")
    _call("nnue_ftm_backward_weight_update_forward", fm.bits.data_ptr(), d_out.data_ptr(), b, fm.num_rows, fm.positions, l1,
          weight.data_ptr(), _ptr(momentum_rows), coef.data_ptr(), float(lr), float(momentum), float(weight_decay), float(grad_scale),
          int(bool(first_step)), _ptr(lr_dev), fm_next.bits.data_ptr(), fm_next.sink.data_ptr(), fm_next.batch, bias.data_ptr(), out_next.data_ptr(),
          fm_next.scratch.data_ptr(), fm_next.scratch.numel(), _stream(d_out))


class FactorExchange:
    """Buffers of the factor exchange (include/nnue_hip.h, nnue_dp_factor_*): ``chunks`` uint8 [world][chunk_bytes] is the
    all-gather's receive buffer, ``own`` this rank's slice of it (sent in place); ``d_ft`` / ``sink`` are float views of the
    own chunk that the step's kernels write directly; ``g_*`` the global factors every rank rebuilds after the gather."""

    def __init__(self, world: int, rank: int, batch: int, positions: int, num_rows: int, l1: int, head: int, tail_lo: int,
                 tail: int, device):
        L = load()
        self.world, self.rank, self.batch, self.positions, self.l1 = world, rank, batch, positions, l1
        self.head, self.tail_lo, self.tail = int(head), int(tail_lo), int(tail)
        small = self.head + self.tail
        self.chunk_bytes = int(L.nnue_dp_factor_chunk_bytes(batch, positions, l1, small))
        if self.chunk_bytes <= 0 or self.chunk_bytes % 16:
            raise NnueHipError("nnue_dp_factor_chunk_bytes: unusable shape")
        self.chunks = torch.zeros((world, self.chunk_bytes), dtype=torch.uint8, device=device)
        self.own = self.chunks[rank]
        off = [int(L.nnue_dp_factor_offset(w, batch, positions, l1, small)) for w in range(4)]
        self.d_ft = self.own[off[0]:off[0] + batch * l1 * 4].view(torch.float32).view(batch, l1)
        self.sink = self.own[off[1]:off[1] + batch * 4].view(torch.float32)
        rows = world * batch
        self.g_dft = torch.empty((rows, l1), dtype=torch.float32, device=device)
        # the global map as a FeatureMatrix: what nnue_ftm_gram_sqnorm / nnue_ftm_backward_weight_update take (no forward scratch)
        self.g_fm = FeatureMatrix(torch.empty((rows, positions), dtype=torch.uint8, device=device),
                                  torch.zeros((rows,), dtype=torch.int32, device=device),
                                  torch.empty((rows,), dtype=torch.float32, device=device),
                                  torch.empty((16,), dtype=torch.uint8, device=device), positions, num_rows)

    def pack(self, fm: "FeatureMatrix", flat_grads: torch.Tensor) -> None:
        """own chunk <- map bits + the small gradients (d_ft and sink are already there)."""
        _call("nnue_dp_factor_pack", fm.bits.data_ptr(), flat_grads.data_ptr(), self.head, self.tail_lo, self.tail, self.batch,
              self.positions, self.l1, self.own.data_ptr(), _stream(flat_grads))

    def unpack(self, flat_grads: torch.Tensor) -> None:
        """gathered chunks -> global factors; flat_grads' small parts <- their sum over the ranks (rank order)."""
        _call("nnue_dp_factor_unpack", self.chunks.data_ptr(), self.world, self.batch, self.positions, self.l1, self.head, self.tail_lo,
              self.tail, self.g_fm.bits.data_ptr(), self.g_fm.sink.data_ptr(), self.g_dft.data_ptr(), flat_grads.data_ptr(),
              _stream(flat_grads))


def sgd_step(params: torch.Tensor, grads: torch.Tensor, momentum_buf: Optional[torch.Tensor], lr: float,
             momentum: float, weight_decay: float, max_norm: float, grad_scale: float, first_step: bool,
             norm_out: Optional[torch.Tensor], scratch: torch.Tensor, ste=None, ext=None,
             coef_out: Optional[torch.Tensor] = None, ext_applied_elsewhere: bool = False,
             lr_dev: Optional[torch.Tensor] = None) -> None:
    """ste = (partial scratch of ste_conv_backward(stages=1), chunks, fps, d_thr, d_weight): the deferred second stage
    runs inside the norm launch; d_thr / d_weight must be the first elements of `grads`.  lr_dev (device float32 scalar): the
    learning rate is read from it instead of `lr` (changes between steps need no re-recording / re-capture)."""
    params = _need(params, torch.float32, "flat params")
    grads = _need(grads, torch.float32, "flat grads", tuple(params.shape))
    if momentum_buf is not None:
        _need(momentum_buf, torch.float32, "momentum buffer", tuple(params.shape))
    if ste is not None:
        part, chunks, fps, d_thr, d_weight = ste
        ste_args = (part.data_ptr(), int(chunks), int(fps), d_thr.data_ptr(), d_weight.data_ptr())
    else:
        ste_args = (None, 0, 0, None, None)
    # ext = (partials, lo, hi): a producer's sums of squares for grads[lo:hi] (e.g. ftm_backward(sq_partial=...))
    ext_args = (ext[0].data_ptr(), ext[0].numel(), int(ext[1]), int(ext[2])) if ext is not None else (None, 0, 0, 0)
    _call("nnue_sgd_step", params.data_ptr(), grads.data_ptr(), _ptr(momentum_buf), params.numel(), float(lr),
          float(momentum), float(weight_decay), float(max_norm), float(grad_scale), int(bool(first_step)),
          _ptr(norm_out), scratch.data_ptr(), scratch.numel(), *ste_args, *ext_args, _ptr(coef_out), int(bool(ext_applied_elsewhere)),
          _ptr(lr_dev), _stream(params))


def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
              step_counter: torch.Tensor, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
              max_norm: float = 0.0, grad_scale: float = 1.0, norm_out: Optional[torch.Tensor] = None,
              scratch: Optional[torch.Tensor] = None, lr_dev: Optional[torch.Tensor] = None) -> None:
    params = _need(params, torch.float32, "flat params")
    shape = tuple(params.shape)
    grads = _need(grads, torch.float32, "flat grads", shape)
    _need(exp_avg, torch.float32, "exp_avg", shape)
    _need(exp_avg_sq, torch.float32, "exp_avg_sq", shape)
    _need(step_counter, torch.int32, "step counter", (1,))
    if scratch is None:
        scratch = torch.empty((sgd_scratch_bytes(params.numel()),), dtype=torch.uint8, device=params.device)
    _call("nnue_adam_step", params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
          step_counter.data_ptr(), params.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
          float(max_norm), float(grad_scale), _ptr(norm_out), scratch.data_ptr(), scratch.numel(), _ptr(lr_dev), _stream(params))


def load_batch(images_u8: torch.Tensor, labels_all: torch.Tensor, indices: torch.Tensor, augment: bool, seed: int, step: int,
               out: Optional[torch.Tensor] = None, labels_out: Optional[torch.Tensor] = None):
    """uint8 [N,H,W,3] dataset + indices [B] -> normalised float32 [B,3,H,W] (+ light augmentation) and labels [B]."""
    images_u8 = _need(images_u8, torch.uint8, "dataset images")
    if images_u8.dim() != 4 or images_u8.shape[3] != 3:
        raise ValueError(f"dataset images: expected uint8 [N,H,W,3], got {tuple(images_u8.shape)}")
    n, h, w, _ = images_u8.shape
    labels_all = _need(labels_all, torch.int64, "dataset labels", (n,))
    indices = _need(indices, torch.int64, "indices")
    b = indices.numel()
    if out is None:
        out = torch.empty((b, 3, h, w), dtype=torch.float32, device=images_u8.device)
    elif tuple(out.shape) != (b, 3, h, w) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("load_batch: out must be a contiguous float32 [B,3,H,W] tensor")
    if labels_out is None:
        labels_out = torch.empty((b,), dtype=torch.int64, device=images_u8.device)
    _call("nnue_load_batch", images_u8.data_ptr(), labels_all.data_ptr(), indices.data_ptr(), b, h, w, n, int(bool(augment)),
          int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1), out.data_ptr(), labels_out.data_ptr(), _stream(images_u8))
    return out, labels_out
