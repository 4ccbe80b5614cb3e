"""Drop-in for the PyTorch half of the reference's ``evaluate.py``: ``compute_metrics`` and
``evaluate_model`` (evaluate.py:23-59, :62-87), which ``train.py:378-398`` runs over the whole train AND
validation set after every epoch.

The reference moves every batch's logits to the host and calls four sklearn scorers on the concatenation.
Here a batch costs one forward pass, one cross-entropy kernel and one confusion-matrix kernel (integer
atomics) on the GPU; the per-batch mean losses and the K x K matrix stay on the device and are read back
once.  Accuracy and the support-weighted precision / recall / F1 (``zero_division=0``) are then formed from
the matrix in float64 -- the same numbers sklearn produces, without sklearn.

``evaluate_compiled_model`` (one subprocess of the C++ engine per image) is out of scope and stays with the
reference.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from nnue_hip import lib as _lib


def metrics_from_confusion(confusion) -> Dict[str, float]:
    """confusion[truth, pred] (any integer array-like) -> {"acc", "f1", "precision", "recall"} as sklearn's
    accuracy_score and the average="weighted", zero_division=0 scorers compute them."""
    m = np.asarray(confusion, dtype=np.float64)
    total = m.sum()
    if total == 0:
        return {"acc": 0.0, "f1": 0.0, "precision": 0.0, "recall": 0.0}
    tp = np.diag(m)
    support = m.sum(axis=1)       # occurrences in the targets
    predicted = m.sum(axis=0)     # occurrences in the predictions
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.where(predicted > 0, tp / predicted, 0.0)
        recall = np.where(support > 0, tp / support, 0.0)
        f1 = np.where(precision + recall > 0, 2 * precision * recall / (precision + recall), 0.0)
    w = support / total  # labels that only occur in the predictions carry weight 0
    return {"acc": float(tp.sum() / total), "f1": float((w * f1).sum()), "precision": float((w * precision).sum()),
            "recall": float((w * recall).sum())}


def _host_confusion(outputs: torch.Tensor, targets: torch.Tensor) -> np.ndarray:
    """Host-side bookkeeping for tensors that already live on the CPU (metric logic, not the hot path)."""
    out = outputs.detach().numpy()
    tgt = targets.detach().numpy().reshape(-1)
    if out.ndim == 1:
        out = out.reshape(-1, 1)
    if out.shape[1] == 1:
        pred, truth, k = (out[:, 0] > 0.5).astype(np.int64), (tgt > 0.5).astype(np.int64), 2
    else:
        pred, truth = out.argmax(axis=1), tgt.astype(np.int64)
        k = int(max(out.shape[1], truth.max(initial=0) + 1))
    return np.bincount(truth * k + pred, minlength=k * k).reshape(k, k)


def _checked(confusion: np.ndarray, samples: int) -> np.ndarray:
    """The HIP confusion kernel skips a sample whose label is outside [0, K): the matrix then holds fewer samples than
    were scored.  sklearn would count such a sample and F.cross_entropy would raise -- so a corrupt label tensor is an
    error here, found with the read-back the caller makes anyway (no extra launch or synchronisation)."""
    if int(confusion.sum()) != samples:
        raise ValueError(f"{samples - int(confusion.sum())} of {samples} labels lie outside [0, {confusion.shape[0]})")
    return confusion


def compute_metrics(outputs: torch.Tensor, targets: torch.Tensor) -> Dict[str, float]:
    """Same contract as the reference (evaluate.py:23-59).  GPU tensors are scored by the HIP kernel."""
    if outputs.is_cuda:
        logits = outputs.detach().to(torch.float32)
        if logits.dim() == 1:
            logits = logits.reshape(-1, 1)
        labels = targets.detach().reshape(-1)
        labels = (labels > 0.5).to(torch.int64) if logits.shape[1] == 1 else labels.to(torch.int64)
        conf = _lib.confusion_accumulate(logits.contiguous(), labels.to(logits.device)).cpu().numpy()
        return metrics_from_confusion(_checked(conf, int(logits.shape[0])))
    return metrics_from_confusion(_host_confusion(outputs, targets))


@torch.no_grad()
def evaluate_model(model: torch.nn.Module, loader, loss_fn=None, device: Optional[torch.device] = None
                   ) -> Tuple[float, Dict[str, float]]:
    """Mean of the per-batch mean cross-entropies and the metrics over the whole loader (evaluate.py:62-87).
    ``loss_fn`` is accepted and ignored, exactly like the reference.  One host read-back at the end.

    An NNUE on the GPU goes through a replayed hipGraph per batch shape (nnue_hip/eval_plan.py): same kernels as the
    eager forward, no per-call Python between them; anything else takes the eager path below."""
    if device is None:
        device = next(model.parameters()).device
    import nnue as _nnue
    if isinstance(model, _nnue.NNUE) and torch.device(device).type == "cuda" and next(model.parameters()).is_cuda \
            and os.environ.get("NNUE_EVAL_GRAPH", "1") != "0":
        return _evaluate_nnue_graph(model, loader, device)
    confusion = None
    losses = []
    seen = 0
    for images, labels in loader:
        seen += int(labels.shape[0])
        images = images.to(device, non_blocking=True)
        labels = labels.to(device, non_blocking=True).long()
        logits = model(images).float().contiguous()
        _, loss, _ = _lib.cross_entropy(logits, labels, want_grad=False)
        losses.append(loss)
        if logits.shape[1] == 1:
            labels = (labels > 0).long()
        confusion = _lib.confusion_accumulate(logits, labels, confusion)
    if not losses:
        raise ValueError("evaluate_model: empty loader")
    total = torch.stack(losses).double().sum().item()  # the single synchronisation
    return total / len(losses), metrics_from_confusion(_checked(confusion.cpu().numpy(), seen))


def _evaluate_nnue_graph(model, loader, device) -> Tuple[float, Dict[str, float]]:
    from nnue_hip.eval_plan import plan_for
    plans, batches, seen = [], 0, 0
    for images, labels in loader:
        seen += int(labels.shape[0])
        if images.dim() != 4 or images.shape[1] != 3:
            raise ValueError(f"images: expected [B,3,H,W], got {tuple(images.shape)}")
        plan = plan_for(model, int(images.shape[0]), (int(images.shape[2]), int(images.shape[3])))
        if plan not in plans:
            plan.reset()
            plans.append(plan)
        plan.run(images.to(device, non_blocking=True).float(), labels.to(device, non_blocking=True).long())
        batches += 1
    if not batches:
        raise ValueError("evaluate_model: empty loader")
    total = sum(p.loss_sum for p in plans)      # per-batch means, summed in float64 on the device
    confusion = sum(p.confusion for p in plans)
    mean_loss = float(total.item()) / batches  # the single synchronisation
    return mean_loss, metrics_from_confusion(_checked(confusion.cpu().numpy(), seen))


def evaluate_compiled_model(model: torch.nn.Module, loader, model_type: str) -> Dict[str, float]:
    """The reference's compiled-engine evaluation (evaluate.py:88-385) without the per-image subprocess: the model is
    serialised to a temporary `.nnue` file exactly as there, the file is loaded the way the C++ engine loads it, and
    the engine's integer `evaluate_logits` runs for whole batches on the GPU (bit-identical logits and densities,
    include/nnue_hip.h: nnue_engine_evaluate_logits).  Returns the reference's keys: the compute_metrics dict plus
    `ms_per_sample` (GPU time of the engine call per sample) and `latent_density` (mean active fraction)."""
    import tempfile
    import time
    from pathlib import Path

    from nnue_hip.engine import EngineModel
    from serialize import serialize_model
    if model_type == "etinynet":
        raise NotImplementedError("EtinyNet is outside this build's scope (SURVEY section 8): use the reference's evaluate.py")
    if model_type != "nnue":
        raise ValueError(f"Unknown model type: {model_type}")
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / "model.nnue"
        serialize_model(model, path)
        engine = EngineModel.load(path)
    outputs, targets, densities = [], [], []
    seconds, samples = 0.0, 0
    for images, labels in loader:
        images = images.to(engine.device, non_blocking=True).float().contiguous()
        torch.cuda.synchronize(engine.device)
        t0 = time.perf_counter()
        logits, density = engine.evaluate_logits(images)
        torch.cuda.synchronize(engine.device)
        seconds += time.perf_counter() - t0
        samples += images.shape[0]
        outputs.append(logits)
        densities.append(density)
        targets.append(labels.reshape(-1).to(engine.device))
    if not outputs:
        raise RuntimeError("No outputs generated during compiled model evaluation")
    outputs, targets = torch.cat(outputs), torch.cat(targets)
    inferred = int(targets.max().item()) + 1 if targets.numel() else 1
    if inferred > 2 and outputs.shape[1] == 1:
        raise RuntimeError(f"Compiled NNUE produced shape {tuple(outputs.shape)} for {inferred}-class labels.")
    metrics = compute_metrics(outputs, targets)
    metrics["ms_per_sample"] = seconds / samples * 1000.0 if samples else 0.0
    metrics["latent_density"] = float(torch.cat(densities).double().mean().item())
    return metrics
