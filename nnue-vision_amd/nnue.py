"""Drop-in replacement for the NNUE half of the reference's ``nnue.py`` on MI355X.

Put this directory in front of the reference root on ``sys.path`` and ``from nnue import NNUE,
GridFeatureSet, ...`` (train.py:21, serialize.py:20) resolves here: same class names, constructor
signatures, attribute names, parameter order and state-dict keys, same initial weights for the same
seed.  What differs is underneath: ``forward`` runs hand-written HIP kernels for gfx950 through
``nnue_hip`` (C ABI in include/nnue_hip.h); there is no per-sample Python loop, no data-dependent
tensor shape and no host synchronisation in ``NNUE.forward``.

Dispatch is by the *tensor's device*, never by what happens to be installed: a GPU tensor always takes the HIP
kernels and raises ``NnueHipError`` when libnnue_hip.so is missing or stale (nothing falls back to anything).  A
CPU tensor -- the reference's ``device`` fixture without a GPU (tests/conftest.py:157-160), BASELINE configs[0] --
runs the same formulas as vectorised stock torch with autograd (``_host_*`` below): a convenience path for
plumbing runs, not a measured one, never used by the trainer or bench.py, and independent of ``oracle/``.

Reference map (file:line under the reference root):
  GridFeatureSet nnue.py:81-90 | LossParams :62-72 | StraightThroughBinary :15-59
  NNUE :447-671 | FeatureTransformer :674-710 | SimpleClassifier :713-738
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from nnue_hip import lib as _lib
from nnue_hip import ops as _ops

DEFAULT_L1 = 1024
DEFAULT_L2 = 128
DEFAULT_L3 = 32


@dataclass
class LossParams:
    """Chess-era loss constants; stored on the model, unused by the vision loss (nnue.py:62-72)."""
    in_offset: float = 270
    out_offset: float = 270
    in_scaling: float = 340
    out_scaling: float = 380
    start_lambda: float = 1.0
    end_lambda: float = 1.0
    pow_exp: float = 2.5
    qp_asymmetry: float = 0.0


@dataclass
class GridFeatureSet:
    grid_size: int = 10
    num_features_per_square: int = 8

    @property
    def num_features(self) -> int:
        return self.grid_size ** 2 * self.num_features_per_square


class StraightThroughBinary(torch.autograd.Function):
    """(x > t) forward, identity gradient to x, sigmoid-slope gradient to t (nnue.py:15-54).

    Stand-alone helper kept for API compatibility; NNUE.forward does not go through it (the fused
    node computes the same forward in nnue_binarize_features and the same threshold gradient in
    nnue_ste_conv_backward)."""

    SHARPNESS = 10.0

    @staticmethod
    def forward(ctx, input, threshold=0.0):
        if not torch.is_tensor(threshold):
            threshold = torch.as_tensor(threshold, dtype=input.dtype, device=input.device)
        ctx.save_for_backward(input, threshold)
        return (input > threshold).to(torch.float32)

    @staticmethod
    def backward(ctx, grad_output):
        x, t = ctx.saved_tensors
        g_t = None
        if t.requires_grad:
            s = torch.sigmoid(StraightThroughBinary.SHARPNESS * (x - t))
            g_t = -(grad_output * (StraightThroughBinary.SHARPNESS * s * (1 - s))).sum(dim=(0, 2, 3), keepdim=True)
        return grad_output, g_t


def binary_activation_ste(x, threshold=0.0):
    return StraightThroughBinary.apply(x, threshold)


# ---- CPU tensors: the module formulas in stock torch (device dispatch, see the module docstring) -------------------
def _host_feature_transformer(idx, val, weight, bias):
    """out[b] = bias + sum_i val[b,i] * weight[clamp(idx[b,i], 0, F-1)] over idx >= 0 (nnue.py:686-710) as a
    coefficient matrix times the table; repeats accumulate, the values keep their gradient."""
    keep = idx >= 0
    rows = torch.where(keep, idx.clamp(0, weight.shape[0] - 1), torch.zeros_like(idx))
    coef = torch.zeros(idx.shape[0], weight.shape[0], dtype=val.dtype, device=val.device)
    coef = coef.scatter_add(1, rows, torch.where(keep, val, torch.zeros_like(val)))
    return coef @ weight + bias


def _host_classifier(x, linears, clip, bucket=None):
    """Linear / ReLU (or clamp(0, clip)) x 2 / Linear (nnue.py:728-738); stacked weights take each sample's own stack."""
    act = (lambda t: t.clamp(0.0, clip)) if clip else F.relu
    (a, b, c) = linears
    if a.weight.dim() == 2:
        return F.linear(act(F.linear(act(F.linear(x, a.weight, a.bias)), b.weight, b.bias)), c.weight, c.bias)
    out = x.new_zeros(x.shape[0], c.weight.shape[1])
    for k in range(a.weight.shape[0]):
        rows = (bucket == k).nonzero(as_tuple=True)[0]
        if rows.numel():
            h = act(F.linear(act(F.linear(x[rows], a.weight[k], a.bias[k])), b.weight[k], b.bias[k]))
            out = out.index_add(0, rows, F.linear(h, c.weight[k], c.bias[k]))
    return out


class FeatureTransformer(nn.Module):
    """Sparse input layer: out[b] = bias + sum_i val[b,i] * weight[clamp(idx[b,i])]  (nnue.py:674-710)."""

    def __init__(self, num_features: int, output_size: int):
        super().__init__()
        self.num_features = num_features
        self.output_size = output_size
        self.weight = nn.Parameter(torch.randn(num_features, output_size) * 0.1)
        self.bias = nn.Parameter(torch.zeros(output_size))

    def forward(self, feature_indices: torch.Tensor, feature_values: torch.Tensor) -> torch.Tensor:
        """feature_indices int64 [B,M] (negative = padding, >= num_features clamps to the last row, any
        order, repeats accumulate); feature_values float32 [B,M].  Returns float32 [B, output_size]."""
        if not feature_values.is_cuda:
            return _host_feature_transformer(feature_indices, feature_values, self.weight, self.bias)
        return _ops.FeatureTransformerFn.apply(feature_indices, feature_values, self.weight, self.bias)


class SimpleClassifier(nn.Module):
    """Linear-ReLU-Linear-ReLU-Linear head (nnue.py:713-738).  ``classifier`` stays an nn.Sequential of
    stock layers so that state-dict keys, init and serialisation match; forward runs the fused kernels."""

    def __init__(self, l1_size: int, l2_size: int, l3_size: int, num_classes: int):
        super().__init__()
        self.num_classes = num_classes
        self.classifier = nn.Sequential(
            nn.Linear(l1_size, l2_size),
            nn.ReLU(),
            nn.Linear(l2_size, l3_size),
            nn.ReLU(),
            nn.Linear(l3_size, num_classes),
        )
        self.clip_activations: Optional[float] = None  # build extension; None = reference ReLU

    def _linears(self):
        seq = self.classifier
        return seq[0], seq[2], seq[4]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        a, b, c = self._linears()
        lead = x.shape[:-1]
        if not x.is_cuda:
            return _host_classifier(x, (a, b, c), self.clip_activations)
        y = _ops.ClassifierFn.apply(x.reshape(-1, x.shape[-1]), a.weight, a.bias, b.weight, b.bias, c.weight, c.bias,
                                    False, float(self.clip_activations or 0.0), None)
        return y.reshape(*lead, y.shape[-1])


class BucketedLinear(nn.Module):
    """K independent Linear layers stacked to weight [K, out, in] / bias [K, out]; slice k is initialised exactly like
    ``nn.Linear(in, out)`` (the draws happen bucket by bucket).  Build extension -- the reference has one stack."""

    def __init__(self, num_buckets: int, in_features: int, out_features: int):
        super().__init__()
        self.num_buckets, self.in_features, self.out_features = num_buckets, in_features, out_features
        stacks = [nn.Linear(in_features, out_features) for _ in range(num_buckets)]
        self.weight = nn.Parameter(torch.stack([m.weight.detach() for m in stacks]))
        self.bias = nn.Parameter(torch.stack([m.bias.detach() for m in stacks]))

    def stack(self, k: int) -> nn.Linear:
        """Bucket k as a stock nn.Linear (a copy: for export / inspection)."""
        m = nn.Linear(self.in_features, self.out_features)
        with torch.no_grad():
            m.weight.copy_(self.weight[k])
            m.bias.copy_(self.bias[k])
        return m

    def extra_repr(self) -> str:
        return f"buckets={self.num_buckets}, in_features={self.in_features}, out_features={self.out_features}"


class BucketedClassifier(nn.Module):
    """``num_ls_buckets`` SimpleClassifier weight sets, one selected per sample (BASELINE configs[2]:
    the engine's LayerStack vector, engine/src/nnue_engine.cpp:619-635, brought back to training).  Same attribute and
    state-dict key names as SimpleClassifier (``classifier.{0,2,4}.{weight,bias}``), the tensors carry a leading K."""

    def __init__(self, l1_size: int, l2_size: int, l3_size: int, num_classes: int, num_buckets: int):
        super().__init__()
        self.num_classes, self.num_buckets = num_classes, num_buckets
        self.classifier = nn.Sequential(
            BucketedLinear(num_buckets, l1_size, l2_size),
            nn.ReLU(),
            BucketedLinear(num_buckets, l2_size, l3_size),
            nn.ReLU(),
            BucketedLinear(num_buckets, l3_size, num_classes),
        )
        self.clip_activations: Optional[float] = None

    def _linears(self):
        seq = self.classifier
        return seq[0], seq[2], seq[4]

    def forward(self, x: torch.Tensor, bucket: torch.Tensor) -> torch.Tensor:
        """x [B, L1] (the pairwise block's output), bucket integer [B] in [0, K): the stack of each sample."""
        a, b, c = self._linears()
        if not x.is_cuda:
            return _host_classifier(x.reshape(-1, x.shape[-1]), (a, b, c), self.clip_activations, bucket.reshape(-1))
        return _ops.ClassifierFn.apply(x.reshape(-1, x.shape[-1]), a.weight, a.bias, b.weight, b.bias, c.weight, c.bias,
                                       False, float(self.clip_activations or 0.0), bucket.reshape(-1))


def bucket_of(active_counts: torch.Tensor, num_buckets: int, flat_ids: int) -> torch.Tensor:
    """The selector: min(K-1, n*K // (flat_ids+1)) with n the sample's active-feature count and flat_ids = fps*Gh*Gw
    of the map it was counted on.  NNUE.forward computes it on the device from the binarise
    kernel's counts; this is the same rule for callers that drive the stand-alone modules."""
    return torch.clamp((active_counts.long() * num_buckets) // (flat_ids + 1), max=num_buckets - 1)


class NNUE(nn.Module):
    """NNUE image classifier (nnue.py:447-671): 3x3 conv -> per-channel threshold -> active grid
    features -> FeatureTransformer -> pairwise product -> SimpleClassifier.

    Two keyword extensions with reference-preserving defaults: ``num_ls_buckets`` (1 = the reference's
    SimpleClassifier, bit for bit; K > 1 = BucketedClassifier) and ``clip_activations`` (None = plain ReLU)."""

    def __init__(self, feature_set: Optional[GridFeatureSet] = None, l1_size: int = DEFAULT_L1,
                 l2_size: int = DEFAULT_L2, l3_size: int = DEFAULT_L3, loss_params=LossParams(), num_classes=1,
                 weight_decay=5e-4, input_size=32, num_ls_buckets: int = 1,
                 clip_activations: Optional[float] = None):
        super().__init__()
        if not 1 <= int(num_ls_buckets) <= 64:
            raise ValueError("num_ls_buckets must be in 1..64")
        if l1_size % 2:
            raise ValueError("l1_size must be even (pairwise product splits it in two)")
        if feature_set is None:
            feature_set = GridFeatureSet(grid_size=10, num_features_per_square=8)
        self.feature_set = feature_set
        self.l1_size, self.l2_size, self.l3_size = l1_size, l2_size, l3_size
        self.num_classes = num_classes
        self.loss_params = loss_params
        self.weight_decay = weight_decay
        self.input_size = input_size
        self.num_ls_buckets = num_ls_buckets

        fps, stride = self._calculate_conv_params(input_size, feature_set.grid_size, feature_set.num_features_per_square)
        # creation order == the reference's, so torch.manual_seed(s); NNUE(...) draws identical weights
        self.conv = nn.Conv2d(3, fps, kernel_size=3, stride=stride, padding=1, bias=False)
        self.input = FeatureTransformer(feature_set.num_features, l1_size)
        self.classifier = (SimpleClassifier(l1_size, l2_size, l3_size, num_classes) if num_ls_buckets == 1 else
                           BucketedClassifier(l1_size, l2_size, l3_size, num_classes, num_ls_buckets))
        self.classifier.clip_activations = clip_activations
        self.nnue2score = nn.Parameter(torch.tensor(600.0))
        self.visual_threshold = nn.Parameter(torch.full((fps,), 0.1))

    # ---- geometry ----------------------------------------------------------------------------
    def _calculate_conv_params(self, input_size, target_grid_size, num_features_per_square):
        """stride = max(1, (input_size-1) // (grid-1)); the map may exceed the grid (32/10 -> 11x11),
        ids beyond the table then clamp to its last row (nnue.py:509-526, :701)."""
        return num_features_per_square, max(1, (input_size - 1) // (target_grid_size - 1))

    # ---- hot path ----------------------------------------------------------------------------
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        a, b, c = self.classifier._linears()
        if not images.is_cuda:
            return self._host_forward(images)
        return _ops.NnueFn.apply(images, self.visual_threshold, self.conv.weight, self.input.weight, self.input.bias,
                                 a.weight, a.bias, b.weight, b.bias, c.weight, c.bias, int(self.conv.stride[0]),
                                 float(self.classifier.clip_activations or 0.0))

    def _host_forward(self, images: torch.Tensor) -> torch.Tensor:
        """NNUE.forward for CPU tensors (nnue.py:637-671) without the per-sample loops: the active-feature gather is the
        {0,1} map times the table, ids past the table fold into its last row (nnue.py:701).  As in the reference the
        map's gradient exists at active positions only (the values are gathered from them, nnue.py:628-633)."""
        x = self.conv(images)
        bits = StraightThroughBinary.apply(x, self.visual_threshold.view(1, -1, 1, 1))
        flat = bits.reshape(bits.shape[0], -1)
        flat = flat * flat.detach()  # same values; gradient only where the feature is active
        w, rows = self.input.weight, self.input.weight.shape[0]
        if flat.shape[1] < rows:
            ft = flat @ w[:flat.shape[1]] + self.input.bias
        else:
            ft = flat[:, :rows - 1] @ w[:rows - 1] + flat[:, rows - 1:].sum(dim=1, keepdim=True) * w[rows - 1] + self.input.bias
        s0, s1 = torch.split(ft, self.l1_size // 2, dim=1)
        l0 = torch.cat([s0 * s1, s0], dim=1)
        if self.num_ls_buckets == 1:
            return self.classifier(l0)
        counts = (flat.detach() > 0.5).sum(dim=1)
        return self.classifier(l0, bucket_of(counts, self.num_ls_buckets, flat.shape[1]))

    def _to_sparse_features(self, binary_features: torch.Tensor):
        """Reference-format sparse view of a {0,1} map: (idx int64 [B,M] padded -1, val float32 [B,M]),
        M = max(max active, 1) (nnue.py:590-635).  The width is data dependent, so this call reads one
        integer back from the device; NNUE.forward itself never calls it.  As in the reference the ids carry
        no gradient and the values stay attached to the map (nnue.py:628-633): gradients flow back to
        ``binary_features`` at the active positions."""
        bsz = binary_features.shape[0]
        flat = binary_features.reshape(bsz, -1).to(torch.float32)
        if not flat.is_cuda:
            on = flat.detach() > 0.5
            n = on.sum(dim=1)
            width = max(int(n.max()) if bsz else 0, 1)
            order = torch.argsort((~on).to(torch.int8), dim=1, stable=True)[:, :width]  # active positions first, ascending
            live = torch.arange(width).unsqueeze(0) < n.unsqueeze(1)
            idx = torch.where(live, order, torch.full_like(order, -1))
            val = torch.where(live, flat.gather(1, order), flat.new_zeros(()))
            return idx, val
        with torch.no_grad():
            half = torch.full((1,), 0.5, dtype=torch.float32, device=flat.device)
            act = _lib.binarize_features(flat.detach().reshape(bsz, 1, 1, -1), half, max(int(flat.shape[-1]), 1))
            width = max(int(act.n.max().item()), 1)
            idx, _ = _lib.act_to_padded(act, width)
        return idx, _ops.SparseValuesFn.apply(flat, idx)

    # ---- export ------------------------------------------------------------------------------
    def _clip_weights(self):
        """In-place clamp of the FT table and every classifier Linear weight to [-1, 1]; the conv is
        left alone (nnue.py:528-539)."""
        with torch.no_grad():
            self.input.weight.clamp_(-1.0, 1.0)
            for m in self.classifier.modules():
                if isinstance(m, (nn.Linear, BucketedLinear)):
                    m.weight.clamp_(-1.0, 1.0)

    def get_quantized_model_data(self):
        """eval + clip, then int8 / int32 tensors and metadata exactly as serialize.py expects
        (nnue.py:541-588)."""
        from serialize import quantize_conv_layer, quantize_linear_layer

        self.eval()
        self._clip_weights()
        meta = {
            "feature_set": self.feature_set,
            "L1": self.l1_size,
            "L2": self.l2_size,
            "L3": self.l3_size,
            "num_classes": self.num_classes,
            "nnue2score": self.nnue2score.item(),
            "quantized_one": 127.0,
            "visual_threshold": float(self.visual_threshold.detach().mean().cpu().item()),
        }
        if self.num_ls_buckets == 1:
            layers = [quantize_linear_layer(m) for m in self.classifier.classifier if isinstance(m, nn.Linear)]
            classifier = {"layers": layers}
        else:  # one record per layer stack, bucket order (what engine/src/nnue_engine.cpp:619-635 reads back)
            meta["num_ls_buckets"] = self.num_ls_buckets
            stacks = [{"layers": [quantize_linear_layer(m.stack(k)) for m in self.classifier._linears()]}
                      for k in range(self.num_ls_buckets)]
            classifier = {"layers": stacks[0]["layers"], "stacks": stacks}
        return {
            "metadata": meta,
            "conv_layer": quantize_conv_layer(self.conv),
            "feature_transformer": quantize_linear_layer(self.input),
            "classifier": classifier,
        }


class EtinyNet(nn.Module):
    """Out of scope for this build: the CNN baseline is stock dense PyTorch and stays
    with the reference.  The name exists so that ``from nnue import NNUE, EtinyNet, ...`` keeps working."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("EtinyNet is not part of the MI355X NNUE hot path; use the reference's class")
