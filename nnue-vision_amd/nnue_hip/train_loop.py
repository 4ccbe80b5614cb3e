"""The training driver around the fused step: what ``train.py:257-454`` does for the
NNUE model, minus the parts that are out of scope here (W&B, RunPod, dataset download, C++-engine compile and
per-image compiled evaluation).

* ``load_config`` executes a Python file as the config module (config/config_loader.py:16-50) -- the same
  ``config/train_*.py`` files work unchanged; only the attributes ``train_model`` reads are used.
* ``train_model`` builds the model from the config (train.py:289-302), picks SGD(lr, momentum, weight_decay)
  for ``optimizer_type == "sgd"`` and Adam(lr, weight_decay) otherwise (train.py:457-471), clips with
  ``max_grad_norm`` only if the attribute exists and is > 0 (train.py:363-364), trains ``max_epochs`` epochs,
  evaluates the train and validation loaders after each (train.py:378-387) and keeps the checkpoint with the
  best validation F1 in the reference's layout (checkpoint_manager.py:45-51):
  ``{"epoch", "model_state_dict", "optimizer_state_dict", "metrics", "config_name"}``.
  (serialize.py:536 expects "state_dict" or a bare state dict instead -- the reference's own inconsistency is
  kept: pass ``ckpt["model_state_dict"]`` to it.)

Loaders are any iterables of ``(images float32 [b,3,H,W], labels int [b])`` batches; the data pipeline itself
(torchvision / albumentations) is out of scope.
"""
from __future__ import annotations

import importlib.util
from dataclasses import dataclass, field
from pathlib import Path
from types import ModuleType
from typing import Callable, Dict, Iterable, List, Optional

import os

import torch

from . import lib
from .trainer import NnueTrainer
from .input_pipeline import GpuLoader, train_epoch


class ConfigError(Exception):
    """Raised when a configuration file cannot be loaded (config/config_loader.py:10-13)."""


def load_config(config_path) -> ModuleType:
    path = Path(config_path)
    if not path.exists():
        raise ConfigError(f"Configuration file not found: {path}")
    if path.suffix != ".py":
        raise ConfigError(f"Configuration file must be a Python file (.py): {path}")
    try:
        spec = importlib.util.spec_from_file_location("config", path)
        if spec is None or spec.loader is None:
            raise ConfigError(f"Failed to create module spec for: {path}")
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
        return module
    except ConfigError:
        raise
    except Exception as e:  # same wrapping as the reference
        raise ConfigError(f"Failed to load configuration from {path}: {e}")


@dataclass
class TrainResult:
    best_val_f1: float = 0.0
    best_epoch: int = -1
    history: List[Dict[str, float]] = field(default_factory=list)
    test: Optional[Dict[str, float]] = None
    checkpoint_path: Optional[Path] = None
    steps: int = 0


def build_model(config, device):
    import nnue
    feature_set = nnue.GridFeatureSet(grid_size=config.grid_size, num_features_per_square=config.num_features_per_square)
    # the two build extensions are read from the config when present (reference configs never set them: train.py:289-302)
    return nnue.NNUE(feature_set=feature_set, l1_size=config.l1_size, l2_size=config.l2_size, l3_size=config.l3_size,
                     num_classes=config.num_classes, input_size=config.input_size, weight_decay=config.weight_decay,
                     num_ls_buckets=int(getattr(config, "num_ls_buckets", 1)),
                     clip_activations=getattr(config, "clip_activations", None)).to(device)


def run_training(config, train_loader: Iterable, val_loader: Iterable, test_loader: Optional[Iterable] = None, model=None,
                 checkpoint_dir=None, log: Callable[[str], None] = print, use_graph: bool = True) -> TrainResult:
    import evaluate
    if not torch.cuda.is_available():
        raise lib.NnueHipError("training runs on the GPU only (no CPU fallback in this build)")
    device = torch.device("cuda", torch.cuda.current_device())
    model = build_model(config, device) if model is None else model.to(device)
    in_place = isinstance(train_loader, GpuLoader)  # GPU-resident dataset: batches are written straight into the input slots
    if in_place:
        hw = train_loader.dataset.image_hw
    else:
        first_images, _ = next(iter(train_loader))
        hw = tuple(first_images.shape[2:])
    batch = int(config.batch_size)
    clip = float(config.max_grad_norm) if hasattr(config, "max_grad_norm") and config.max_grad_norm > 0 else 0.0
    if config.optimizer_type == "sgd":
        opt = dict(optimizer="sgd", momentum=float(config.momentum))
    else:
        opt = dict(optimizer="adam")
    trainer = NnueTrainer(model, batch, hw, lr=float(config.learning_rate), weight_decay=float(config.weight_decay),
                          max_grad_norm=clip, use_graph=use_graph, input_slots=8 if in_place else 1, **opt)
    result = TrainResult()
    ckpt_dir = Path(checkpoint_dir) if checkpoint_dir is not None else None
    for epoch in range(int(config.max_epochs)):
        model.train()
        if in_place:
            result.steps += train_epoch(trainer, train_loader)[1]
        else:
            for images, labels in train_loader:
                trainer.step(images.to(device, non_blocking=True), labels.to(device, non_blocking=True).long())
                result.steps += 1
        trainer = _density_check(trainer, log)
        model.eval()
        train_loss, train_metrics = evaluate.evaluate_model(model, train_loader, None, device)
        val_loss, val_metrics = evaluate.evaluate_model(model, val_loader, None, device)
        model.train()
        row = {"epoch": epoch, "train/epoch_loss": train_loss, "train/epoch_f1": train_metrics["f1"],
               "train/epoch_accuracy": train_metrics["acc"], "val/loss": val_loss, "val/f1": val_metrics["f1"],
               "val/accuracy": val_metrics["acc"]}
        result.history.append(row)
        log(f"Epoch {epoch + 1}/{config.max_epochs} - Train Loss: {train_loss:.4f}, Train F1: {train_metrics['f1']:.4f}, "
            f"Train Acc: {train_metrics['acc']:.4f} | Val Loss: {val_loss:.4f}, Val F1: {val_metrics['f1']:.4f}, "
            f"Val Acc: {val_metrics['acc']:.4f}")
        if val_metrics["f1"] > result.best_val_f1:
            result.best_val_f1, result.best_epoch = val_metrics["f1"], epoch
            if ckpt_dir is not None:
                ckpt_dir.mkdir(parents=True, exist_ok=True)
                result.checkpoint_path = ckpt_dir / "best-model.ckpt"
                torch.save({"epoch": epoch,
                            "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
                            "optimizer_state_dict": _to_cpu(trainer.optimizer_state_dict()),
                            "metrics": {"val_f1": val_metrics["f1"], "val_loss": val_loss},
                            "config_name": config.name}, result.checkpoint_path)
    if test_loader is not None:
        model.eval()
        test_loss, test_metrics = evaluate.evaluate_model(model, test_loader, None, device)
        result.test = {"test/f1": test_metrics["f1"], "test/loss": test_loss}
    return result


# Active-feature density below which the gather kernels beat the dense products, by table size (measured on MI355X with
# `bench.py --density D`, which times both forms in one process; profiles/r03*_dens_*.json).  The thresholds
# are learnable (nnue.py:505-507), so the density is a property of the run, not of the shape: the check runs once per epoch on
# the counts the last step left, costs one read-back, and rebuilds the trainer (plans and graphs) only when it switches.
# NNUE_FT_DENSITY_SWITCH=0 disables it; a forced NNUE_FT_PATH is respected.
GATHER_BELOW_DENSITY = {"cache_resident_table": 0.0, "streamed_table": 0.0}  # 0 = the product form won at every measured density


def _density_check(trainer, log=print):
    if os.environ.get("NNUE_FT_DENSITY_SWITCH", "1") == "0" or os.environ.get("NNUE_FT_PATH", "auto") != "auto":
        return trainer
    if trainer.ft_path not in ("mfma", "bits") or not lib.ftb_supported(trainer.L1):
        return trainer
    density = trainer.active_stats()[0] / max(1, trainer.P)
    regime = "streamed_table" if trainer.F * trainer.L1 * 4 >= (200 << 20) else "cache_resident_table"
    want = "bits" if density < GATHER_BELOW_DENSITY[regime] else "mfma"
    if trainer.dp.world > 1:  # the same decision on every rank: take rank 0's
        import torch.distributed as dist
        flag = torch.tensor([1 if want == "bits" else 0], dtype=torch.int32, device=trainer.dev if trainer.dp.backend == "nccl" else "cpu")
        dist.broadcast(flag, src=dist.get_global_rank(trainer.dp.group, 0) if trainer.dp.group is not None else 0, group=trainer.dp.group)
        want = "bits" if int(flag.item()) else "mfma"
    if want == trainer.ft_path or (want == "mfma" and lib.ft_path(trainer.F, trainer.P, trainer.L1, trainer.B) != "mfma"):
        return trainer
    log(f"active-feature density {density:.4f}: FeatureTransformer kernels {trainer.ft_path} -> {want}")
    return trainer.rebuilt(want)


def train_model(config, model_type: str = "nnue", train_loader=None, val_loader=None, test_loader=None, **kwargs) -> int:
    """Same call shape and return value (0) as the reference's train_model; loaders are passed in."""
    if model_type != "nnue":
        raise ValueError(f"Unknown model type: {model_type} (EtinyNet training stays with the reference)")
    if train_loader is None or val_loader is None:
        raise ValueError("train_model needs train_loader and val_loader (the data pipeline is out of scope)")
    run_training(config, train_loader, val_loader, test_loader, **kwargs)
    return 0


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj
