"""CPU oracle for the NNUE training hot path  --  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the reference hot path
(marict/nnue-vision, ``nnue.py`` / ``train.py``).  It is the *checker* for the
HIP kernels in ``nnue-vision_amd/csrc``.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product path never does (it fails loudly without the HIP
extension instead of falling back to anything here).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (``/root/reference/nnue.py``, ``serialize.py``) in the build
container and stores its inputs/outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against them.

Two forms are kept on purpose:

* the *loop form* follows the reference's per-sample Python loops step by step
  (``nnue.py:601-633`` and ``nnue.py:694-708``) and lets autograd build the
  backward exactly as the reference does.  It is what ``bench.py`` times as
  ``cpu_baseline`` (kind "port");
* the *explicit form* spells out forward and backward as closed formulas
  (dense-mask products, gather-dot, scatter-add).  These are the formulas the
  HIP kernels implement, so the tests can compare intermediate tensors too.

All arithmetic is float32 unless ``dtype=torch.float64`` is passed (used by the
tests to measure how far float32 summation order can move a result).

Bucketed layer stacks (``num_ls_buckets = K > 1``, BASELINE configs[2]) and the
clipped-ReLU option are BUILD EXTENSIONS: the reference trains exactly one stack
with plain ReLU (nnue.py:499-500, :728-734; serialize.py:57), so for them this
file is the definition rather than a restatement -- **parity unpinned** (no
reference output exists to pin against; K = 1 / clip None reduces to the pinned
functions, which the tests check).  Definition:
stacked weights ``[K, out, in]``, one stack per sample chosen by
``bucket = min(K-1, n*K // (flat_ids+1))`` with ``n`` the sample's number of
active features and ``flat_ids = fps*Gh*Gw``.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

STE_SHARPNESS = 10.0  # k in nnue.py:41


# --------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------
def conv_stride(input_size: int, grid_size: int) -> int:
    """Stride of the 3x3/pad-1 front-end conv (nnue.py:519)."""
    return max(1, (input_size - 1) // (grid_size - 1))


def conv_out_hw(h: int, w: int, stride: int) -> Tuple[int, int]:
    """Output map of a 3x3, pad 1 conv (nnue.py:522)."""
    return (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1


def num_features(grid_size: int, fps: int) -> int:
    """GridFeatureSet.num_features (nnue.py:88-90)."""
    return grid_size * grid_size * fps


# --------------------------------------------------------------------------
# forward pieces, loop form
# --------------------------------------------------------------------------
def conv_forward(images: torch.Tensor, conv_w: torch.Tensor, stride: int) -> torch.Tensor:
    """nn.Conv2d(3, fps, 3, stride, padding=1, bias=False) (nnue.py:486-493, :640)."""
    return F.conv2d(images, conv_w, None, stride=stride, padding=1)


class _SteBinary(torch.autograd.Function):
    """(x > t) forward; identity grad to x, sigmoid-slope grad to t (nnue.py:15-54)."""

    @staticmethod
    def forward(ctx, x, thr):
        ctx.save_for_backward(x, thr)
        return (x > thr).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        x, thr = ctx.saved_tensors
        g_thr = None
        if thr.requires_grad:
            g_thr = -(g * ste_slope(x, thr)).sum(dim=(0, 2, 3), keepdim=True)
        return g, g_thr


def ste_slope(x: torch.Tensor, thr: torch.Tensor) -> torch.Tensor:
    """k * s * (1 - s), s = sigmoid(k (x - t))  (nnue.py:41-46)."""
    s = torch.sigmoid(STE_SHARPNESS * (x - thr))
    return STE_SHARPNESS * s * (1 - s)


def binarize(conv_out: torch.Tensor, thr: torch.Tensor) -> torch.Tensor:
    """Straight-through binarisation with a per-channel threshold (nnue.py:646-647)."""
    return _SteBinary.apply(conv_out, thr.view(1, -1, 1, 1))


def to_sparse_features_loop(bits: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-sample ascending active ids, -1 padded to the batch maximum (nnue.py:590-635).

    Flat id = c*Gh*Gw + h*Gw + w (the view(B, -1) of a [B, C, Gh, Gw] map).
    The returned values stay attached to ``bits`` for autograd.
    """
    bsz = bits.shape[0]
    flat = bits.reshape(bsz, -1)
    ids = [torch.nonzero(flat[b] > 0.5).squeeze(-1) for b in range(bsz)]
    width = max([1] + [int(i.numel()) for i in ids])
    idx = torch.full((bsz, width), -1, dtype=torch.long)
    val = torch.zeros((bsz, width), dtype=bits.dtype)
    for b, i in enumerate(ids):
        if i.numel():
            idx[b, : i.numel()] = i
            val[b, : i.numel()] = flat[b][i]
    return idx, val


def ft_forward_loop(weight, bias, idx, val) -> torch.Tensor:
    """FeatureTransformer.forward, sample by sample (nnue.py:686-710)."""
    rows = weight.shape[0]
    out = bias.unsqueeze(0).repeat(idx.shape[0], 1)
    for b in range(idx.shape[0]):
        keep = idx[b] >= 0
        if bool(keep.any()):
            r = idx[b][keep].clamp(0, rows - 1)
            out[b] = out[b] + (weight[r] * val[b][keep].unsqueeze(-1)).sum(dim=0)
    return out


def pairwise(ft: torch.Tensor) -> torch.Tensor:
    """cat(s0*s1, s0) with s0, s1 the two halves of ft (nnue.py:660-666)."""
    half = ft.shape[1] // 2
    s0, s1 = ft[:, :half], ft[:, half : 2 * half]
    return torch.cat([s0 * s1, s0], dim=1)


def classifier_forward(x, w1, b1, w2, b2, w3, b3, clip: Optional[float] = None):
    """Linear-ReLU-Linear-ReLU-Linear (nnue.py:728-734).

    ``clip`` is the build's opt-in clipped ReLU; None = reference.
    """
    act = (lambda t: t.clamp(0, clip)) if clip is not None else F.relu
    return F.linear(act(F.linear(act(F.linear(x, w1, b1)), w2, b2)), w3, b3)


def bucket_index(n: torch.Tensor, buckets: int, flat_ids: int) -> torch.Tensor:
    """Layer stack of each sample from its active-feature count (build extension): the
    vision analogue of the piece-count buckets the engine's LayerStack vector descends from
    (engine/src/nnue_engine.cpp:619-635)."""
    return torch.clamp((n.long() * buckets) // (flat_ids + 1), max=buckets - 1)


def classifier_forward_bucketed(x, bucket, w1, b1, w2, b2, w3, b3, clip: Optional[float] = None):
    """classifier_forward with stacked weights [K, out, in]: sample b goes through stack bucket[b] (loop form)."""
    rows = [classifier_forward(x[b:b + 1], w1[k], b1[k], w2[k], b2[k], w3[k], b3[k], clip)
            for b, k in enumerate(bucket.tolist())]
    return torch.cat(rows, dim=0)


def classifier_backward_bucketed(x, bucket, w1, b1, w2, b2, w3, b3, d_logits, clip: Optional[float] = None):
    """Closed-form backward of classifier_forward_bucketed: every stack sums over its own samples only."""
    d_x = torch.zeros_like(x)
    grads = [torch.zeros_like(t) for t in (w1, b1, w2, b2, w3, b3)]
    for k in range(w1.shape[0]):
        sel = (bucket == k).nonzero().squeeze(-1)
        if sel.numel() == 0:
            continue
        dx_k, g_k = classifier_backward(x[sel], w1[k], b1[k], w2[k], b2[k], w3[k], b3[k], d_logits[sel], clip)
        d_x[sel] = dx_k
        for dst, g in zip(grads, g_k):
            dst[k] = g
    return d_x, grads


PARAM_KEYS = (
    "nnue2score",
    "visual_threshold",
    "conv.weight",
    "input.weight",
    "input.bias",
    "classifier.classifier.0.weight",
    "classifier.classifier.0.bias",
    "classifier.classifier.2.weight",
    "classifier.classifier.2.bias",
    "classifier.classifier.4.weight",
    "classifier.classifier.4.bias",
)
TRAINABLE_KEYS = PARAM_KEYS[1:]  # nnue2score never receives a gradient


def model_forward_loop(p: Dict[str, torch.Tensor], images: torch.Tensor, stride: int,
                       keep: Optional[dict] = None, clip: Optional[float] = None) -> torch.Tensor:
    """NNUE.forward in the reference's loop form (nnue.py:637-671).  Stacked classifier weights ([K, out, in]) select
    the bucketed extension; ``clip`` the clipped-ReLU one."""
    x = conv_forward(images, p["conv.weight"], stride)
    bits = binarize(x, p["visual_threshold"])
    idx, val = to_sparse_features_loop(bits)
    ft = ft_forward_loop(p["input.weight"], p["input.bias"], idx, val)
    cls = [p[f"classifier.classifier.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias")]
    if cls[0].dim() == 3:
        bucket = bucket_index((idx >= 0).sum(dim=1), cls[0].shape[0], bits[0].numel())
        logits = classifier_forward_bucketed(pairwise(ft), bucket, *cls, clip)
    else:
        bucket = None
        logits = classifier_forward(pairwise(ft), *cls, clip)
    if keep is not None:
        keep.update(conv_out=x, bits=bits, idx=idx, val=val, ft=ft, bucket=bucket)
    return logits


def loss_and_grads_loop(p: Dict[str, torch.Tensor], images, labels, stride: int, clip: Optional[float] = None):
    """compute_loss + backward (train.py:250-254, :360-361) through autograd."""
    q = {k: v.detach().clone().requires_grad_(k != "nnue2score") for k, v in p.items()}
    keep: dict = {}
    logits = model_forward_loop(q, images, stride, keep, clip)
    loss = F.cross_entropy(logits, labels.long())
    loss.backward()
    grads = {k: q[k].grad for k in TRAINABLE_KEYS}
    return logits.detach(), loss.detach(), grads, {k: v.detach() for k, v in keep.items() if v is not None}


# --------------------------------------------------------------------------
# explicit (closed-form) forward and backward  --  what the HIP kernels compute
# --------------------------------------------------------------------------
def active_lists(conv_out: torch.Tensor, thr: torch.Tensor):
    """Bit-exact feature ids: ascending flat ids with conv_out > thr, per sample.

    Returns (idx [B, P] int64 padded with -1, n [B] int64) with P = C*Gh*Gw
    (fixed capacity, no data-dependent shape -- the layout the HIP path keeps).
    """
    bsz = conv_out.shape[0]
    on = (conv_out > thr.view(1, -1, 1, 1)).reshape(bsz, -1)
    n = on.sum(dim=1)
    order = torch.argsort((~on).to(torch.int8), dim=1, stable=True)  # active first, ascending
    ar = torch.arange(on.shape[1]).unsqueeze(0)
    idx = torch.where(ar < n.unsqueeze(1), order, torch.full_like(order, -1))
    return idx, n


def coefficient_matrix(idx: torch.Tensor, val: torch.Tensor, rows: int) -> torch.Tensor:
    """C[b, f] = sum of val[b, i] over entries with clamp(idx[b, i], 0, rows-1) == f, idx >= 0.

    Duplicates accumulate and ids >= rows fold into the last row, exactly as the
    gather in nnue.py:695-707 does.
    """
    keep = idx >= 0
    r = idx.clamp(0, rows - 1)
    c = torch.zeros(idx.shape[0], rows, dtype=val.dtype)
    c.scatter_add_(1, torch.where(keep, r, torch.zeros_like(r)), torch.where(keep, val, torch.zeros_like(val)))
    return c


def ft_forward(weight, bias, idx, val) -> torch.Tensor:
    """out = bias + C @ W  (same result as ft_forward_loop up to fp32 summation order)."""
    return bias.unsqueeze(0) + coefficient_matrix(idx, val, weight.shape[0]) @ weight


def ft_backward(weight, idx, val, d_out):
    """Closed-form gradients of ft_forward.

    dW   = C^T @ dOut            (embedding scatter-add; duplicates accumulate)
    db   = sum_b dOut
    dval[b, i] = <dOut[b], W[clamp(idx[b, i])]> for idx >= 0, else 0   (gather-dot)
    """
    rows = weight.shape[0]
    c = coefficient_matrix(idx, val, rows)
    d_w = c.t() @ d_out
    d_b = d_out.sum(dim=0)
    g = d_out @ weight.t()  # [B, rows]
    keep = idx >= 0
    d_val = torch.where(keep, torch.gather(g, 1, idx.clamp(0, rows - 1)), torch.zeros_like(val))
    return d_w, d_b, d_val


def pairwise_backward(ft, d_l0):
    """Gradient of cat(s0*s1, s0) w.r.t. ft."""
    half = ft.shape[1] // 2
    s0, s1 = ft[:, :half], ft[:, half : 2 * half]
    d_ft = torch.zeros_like(ft)
    d_ft[:, :half] = d_l0[:, :half] * s1 + d_l0[:, half : 2 * half]
    d_ft[:, half : 2 * half] = d_l0[:, :half] * s0
    return d_ft


def classifier_backward(x, w1, b1, w2, b2, w3, b3, d_logits, clip: Optional[float] = None):
    """Closed-form backward of classifier_forward; returns (dx, [dw1, db1, dw2, db2, dw3, db3])."""
    z1 = F.linear(x, w1, b1)
    h1 = z1.clamp(0, clip) if clip is not None else F.relu(z1)
    z2 = F.linear(h1, w2, b2)
    h2 = z2.clamp(0, clip) if clip is not None else F.relu(z2)

    def gate(z):
        m = z > 0
        if clip is not None:
            m = m & (z < clip)
        return m.to(z.dtype)

    d_w3 = d_logits.t() @ h2
    d_b3 = d_logits.sum(0)
    d_z2 = (d_logits @ w3) * gate(z2)
    d_w2 = d_z2.t() @ h1
    d_b2 = d_z2.sum(0)
    d_z1 = (d_z2 @ w2) * gate(z1)
    d_w1 = d_z1.t() @ x
    d_b1 = d_z1.sum(0)
    return d_z1 @ w1, [d_w1, d_b1, d_w2, d_b2, d_w3, d_b3]


def cross_entropy_backward(logits, labels):
    """Mean cross-entropy and its gradient w.r.t. logits (train.py:254)."""
    lse = torch.logsumexp(logits, dim=1)
    picked = logits.gather(1, labels.long().unsqueeze(1)).squeeze(1)
    loss = (lse - picked).mean()
    d = torch.softmax(logits, dim=1)
    d[torch.arange(logits.shape[0]), labels.long()] -= 1
    return loss, d / logits.shape[0]


def conv_weight_grad(images, d_conv_out, stride: int, kshape) -> torch.Tensor:
    """Gradient of the front-end conv w.r.t. its weight: correlation of the padded
    input with d_conv_out at the conv's stride."""
    fps = d_conv_out.shape[1]
    xp = F.pad(images, (1, 1, 1, 1))
    gh, gw = d_conv_out.shape[2:]
    g = torch.zeros(kshape, dtype=images.dtype)
    for kh in range(3):
        for kw in range(3):
            patch = xp[:, :, kh : kh + (gh - 1) * stride + 1 : stride, kw : kw + (gw - 1) * stride + 1 : stride]
            g[:, :, kh, kw] = torch.einsum("bchw,bihw->ci", d_conv_out, patch)
    assert g.shape[0] == fps
    return g


def loss_and_grads_explicit(p: Dict[str, torch.Tensor], images, labels, stride: int, clip: Optional[float] = None):
    """Whole step without autograd: the exact chain of products the HIP path runs.

    conv -> ids -> ft -> pairwise -> classifier -> CE, then back through
    classifier, pairwise, FT (scatter-add / gather-dot), the dense scatter of
    dval to the active positions (autograd of nnue.py:601-633), the STE
    (nnue.py:28-54) and the conv weight.
    """
    w, b = p["input.weight"], p["input.bias"]
    cls = [p[f"classifier.classifier.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias")]
    thr = p["visual_threshold"]
    x = conv_forward(images, p["conv.weight"], stride)
    idx, n = active_lists(x, thr)
    val = (idx >= 0).to(x.dtype)
    ft = ft_forward(w, b, idx, val)
    l0 = pairwise(ft)
    bucket = None
    if cls[0].dim() == 3:  # bucketed layer stacks (build extension)
        bucket = bucket_index(n, cls[0].shape[0], x[0].numel())
        logits = classifier_forward_bucketed(l0, bucket, *cls, clip)
        loss, d_logits = cross_entropy_backward(logits, labels)
        d_l0, d_cls = classifier_backward_bucketed(l0, bucket, *cls, d_logits, clip)
    else:
        logits = classifier_forward(l0, *cls, clip)
        loss, d_logits = cross_entropy_backward(logits, labels)
        d_l0, d_cls = classifier_backward(l0, *cls, d_logits, clip)
    d_ft = pairwise_backward(ft, d_l0)
    d_w, d_b, d_val = ft_backward(w, idx, val, d_ft)
    # dval lands on the *unclamped* flat position of each active id; inactive positions get 0
    d_bits = torch.zeros(x.shape[0], x[0].numel(), dtype=x.dtype)
    keep = idx >= 0
    d_bits.scatter_(1, torch.where(keep, idx, torch.zeros_like(idx)),
                    torch.where(keep, d_val, d_bits[:, :1].expand_as(d_val)).clone())
    # position 0 may have been clobbered by the padding lanes above; rewrite it exactly
    first_active = keep[:, 0] & (idx[:, 0] == 0)
    d_bits[:, 0] = torch.where(first_active, d_val[:, 0], torch.zeros_like(d_val[:, 0]))
    d_x = d_bits.view_as(x)
    d_thr = -(d_x * ste_slope(x, thr.view(1, -1, 1, 1))).sum(dim=(0, 2, 3))
    d_conv = conv_weight_grad(images, d_x, stride, p["conv.weight"].shape)
    grads = {
        "visual_threshold": d_thr,
        "conv.weight": d_conv,
        "input.weight": d_w,
        "input.bias": d_b,
    }
    for k, g in zip(TRAINABLE_KEYS[4:], d_cls):
        grads[k] = g
    keep_t = dict(conv_out=x, idx=idx, n=n, ft=ft, d_ft=d_ft, d_conv_out=d_x, d_logits=d_logits, bucket=bucket)
    return logits, loss, grads, keep_t


# --------------------------------------------------------------------------
# step tail: clip_grad_norm_ + SGD(momentum, weight_decay)  (train.py:363-366, :457-464)
# --------------------------------------------------------------------------
def clip_coefficient(grads: Sequence[torch.Tensor], max_norm: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch.nn.utils.clip_grad_norm_: global L2 norm, coef = min(1, max_norm/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def sgd_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
             bufs: Dict[str, Optional[torch.Tensor]], lr: float, momentum: float,
             weight_decay: float, max_grad_norm: float) -> torch.Tensor:
    """One optimizer step as the reference's loop performs it.  Returns the pre-clip norm.

    g <- g * clip;  g <- g + wd * p;  buf <- g (first step) | momentum*buf + g;  p <- p - lr*buf
    (torch.optim.SGD, dampening 0, nesterov off.)  nnue2score has no gradient and is skipped.
    """
    keys = [k for k in TRAINABLE_KEYS if k in grads]
    total = torch.zeros(())
    if max_grad_norm and max_grad_norm > 0:
        total, coef = clip_coefficient([grads[k] for k in keys], max_grad_norm)
    else:
        coef = torch.ones(())
    for k in keys:
        g = grads[k] * coef
        if weight_decay:
            g = g + weight_decay * params[k]
        if momentum:
            bufs[k] = g.clone() if bufs.get(k) is None else momentum * bufs[k] + g
            g = bufs[k]
        params[k] = params[k] - lr * g
    return total


def adam_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state: Dict[str, dict], lr: float,
              weight_decay: float, max_grad_norm: float, betas=(0.9, 0.999), eps: float = 1e-8) -> torch.Tensor:
    """clip_grad_norm_ + one torch.optim.Adam(lr, weight_decay) step, the optimizer create_optimizer builds when
    optimizer_type != "sgd" (train.py:363-366, :465-470).  Returns the pre-clip norm."""
    keys = [k for k in TRAINABLE_KEYS if k in grads]
    total = torch.zeros(())
    coef = torch.ones(())
    if max_grad_norm and max_grad_norm > 0:
        total, coef = clip_coefficient([grads[k] for k in keys], max_grad_norm)
    b1, b2 = betas
    for k in keys:
        st = state.setdefault(k, {"step": 0, "m": torch.zeros_like(params[k]), "v": torch.zeros_like(params[k])})
        st["step"] += 1
        g = grads[k] * coef
        if weight_decay:
            g = g + weight_decay * params[k]
        st["m"] = b1 * st["m"] + (1 - b1) * g
        st["v"] = b2 * st["v"] + (1 - b2) * g * g
        bc1, bc2 = 1 - b1 ** st["step"], 1 - b2 ** st["step"]
        denom = st["v"].sqrt() / math.sqrt(bc2) + eps
        params[k] = params[k] - (lr / bc1) * st["m"] / denom
    return total


# --------------------------------------------------------------------------
# parameter initialisation in the reference's RNG order (nnue.py:486-507, :683-684, :728-734)
# --------------------------------------------------------------------------
def init_params(grid_size: int, fps: int, l1: int, l2: int, l3: int, num_classes: int,
                seed: int, buckets: int = 1) -> Dict[str, torch.Tensor]:
    """Draws parameters with the same generator calls, in the same order, as
    ``torch.manual_seed(seed); NNUE(...)`` does in the reference.  buckets > 1 (extension): every layer holds
    ``buckets`` independent nn.Linear draws stacked to [K, out, in], layer by layer."""
    torch.manual_seed(seed)
    conv = torch.nn.Conv2d(3, fps, 3, stride=1, padding=1, bias=False)  # stride does not touch the RNG
    ft_w = torch.randn(num_features(grid_size, fps), l1) * 0.1

    class _Stack:
        def __init__(self, fan_in, fan_out):
            ms = [torch.nn.Linear(fan_in, fan_out) for _ in range(buckets)]
            self.weight = torch.stack([m.weight.detach() for m in ms])
            self.bias = torch.stack([m.bias.detach() for m in ms])

    if buckets > 1:
        lin = [_Stack(l1, l2), _Stack(l2, l3), _Stack(l3, num_classes)]
    else:
        lin = [torch.nn.Linear(l1, l2), torch.nn.Linear(l2, l3), torch.nn.Linear(l3, num_classes)]
    p = {
        "nnue2score": torch.tensor(600.0),
        "visual_threshold": torch.full((fps,), 0.1),
        "conv.weight": conv.weight.detach().clone(),
        "input.weight": ft_w,
        "input.bias": torch.zeros(l1),
    }
    for i, m in zip((0, 2, 4), lin):
        p[f"classifier.classifier.{i}.weight"] = m.weight.detach().clone()
        p[f"classifier.classifier.{i}.bias"] = m.bias.detach().clone()
    return p


# --------------------------------------------------------------------------
# .nnue quantiser (serialize.py:210-239); the byte writer itself is product code
# --------------------------------------------------------------------------
def quantize(t: torch.Tensor, scale: float = 64.0, bias: bool = False) -> torch.Tensor:
    """round-half-even(t*scale); weights additionally clamp to +-127 (serialize.py:218-222, :234-237)."""
    q = torch.round(t * scale)
    return q.to(torch.int32) if bias else q.clamp(-127, 127).to(torch.int8)


def nnue_file_size(f: int, fps: int, l1: int, l2: int, l3: int, c: int, buckets: int = 1) -> int:
    """Size in bytes of a version-2 .nnue file (layout: serialize.py:30-63, :103-136, :394-491)."""
    header = 4 + 4 + 5 * 4 + 3 * 4
    conv = 4 + 4 + 4 * 4 + fps * 27 + 4 + fps * 4
    ft = 4 + 8 + f * l1 * 2 + 4 + l1 * 4
    stack = 16 + (8 + (l2 + 1) * l1 + 4 + (l2 + 1) * 4) + (8 + l1 * l1 + 4 + l1 * 4) \
        + (8 + l3 * 2 * l2 + 4 + l3 * 4) + (8 + c * l3 + 4 + c * 4)
    return header + conv + ft + buckets * stack
