"""Logits of ``NNUE.forward`` (the drop-in module, not the trainer) at the full BASELINE shapes against the reference's
PyTorch-CPU forward, element by element: ``|d| <= 1e-4 * max(1, |logit|)`` (the north star's
acceptance sentence; nnue.py:637-671).  ``-m gpu``.

Inputs are what bench.py feeds: unfiltered ``randn`` images, model after ``torch.manual_seed(0)``.  The only
filter is the one bit-exact feature ids need: a sample with a conv output within 1e-5 of its threshold (float64 check)
is redrawn, because ``conv_out > thr`` is a discontinuity on whose two sides a float32 conv is equally right (our
fixed-order fmaf chain and MKL-DNN differ by ~1e-7); nothing downstream is filtered -- the logits are continuous in
everything else.

Two references per shape:
* the oracle's *loop form* in float32 -- the reference's own arithmetic, per-sample gathers and ``sum(dim=0)``
  (nnue.py:601-633, :694-708): this is "the reference PyTorch-CPU forward on the same inputs";
* the same chain in float64 -- the exact value, which shows how much of a difference is ours.
Every comparison also leaves its error distribution (in units of the bar) under gpurun_out/ when that directory
exists, so that profiles/ can carry the histogram the test passed with.
"""
import json
import os
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

import nnue
import nnue_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = Path(__file__).resolve().parent.parent

SHAPES = {
    # name: NNUE arguments, batch for the float32 loop form, batch for the float64 chain
    "c2": dict(grid=10, fps=8, image=32, classes=10, loop_batch=512, batch=512),
    "c3": dict(grid=10, fps=8, image=32, classes=100, loop_batch=1024, batch=1024),
    "c3k8": dict(grid=10, fps=8, image=32, classes=100, loop_batch=1024, batch=1024, buckets=8, clip=1.0),
    "c4": dict(grid=32, fps=64, image=224, classes=1000, loop_batch=8, batch=128),
}


def draw_images(cfg, params, stride, count, gen, margin=1e-5):
    """randn images; a sample with a conv output within `margin` of its threshold is redrawn (ids must be bit-exact)."""
    hw = cfg["image"]
    images = torch.randn(count, 3, hw, hw, generator=gen)
    w64, t64 = params["conv.weight"].double(), params["visual_threshold"].double().view(1, -1, 1, 1)
    for _ in range(200):
        x = F.conv2d(images.double(), w64, stride=stride, padding=1)
        dirty = ((x - t64).abs() < margin).flatten(1).any(dim=1)
        if not bool(dirty.any()):
            return images
        images[dirty] = torch.randn(int(dirty.sum()), 3, hw, hw, generator=gen)
    raise AssertionError("could not draw images that keep the threshold margin")


def error_report(got, ref, name):
    """Errors in units of the bar 1e-4 * max(1, |ref|); returns (max ratio, report dict)."""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape
    err = (got - ref).abs()
    ratio = (err / (1e-4 * ref.abs().clamp(min=1.0))).flatten()
    q = torch.quantile(ratio[:1 << 22], torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64))
    edges = [0.0, 0.001, 0.003, 0.01, 0.03, 0.1, 0.3, 1.0, 3.0, float("inf")]
    hist = [int(((ratio >= lo) & (ratio < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
    rep = dict(what=name, elements=int(ratio.numel()), max_abs_err=float(err.max()), max_abs_ref=float(ref.abs().max()),
               max_ratio_to_bar=float(ratio.max()), ratio_quantiles=dict(p50=float(q[0]), p90=float(q[1]), p99=float(q[2]), p999=float(q[3])),
               hist_edges=edges[:-1], hist_counts=hist)
    out = ROOT / "gpurun_out"
    if out.is_dir():
        (out / f"logit_err_{name}.json").write_text(json.dumps(rep, indent=1))
    return float(ratio.max()), rep


def build(cfg):
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), 1024, 128, 32, num_classes=cfg["classes"],
                      input_size=cfg["image"], num_ls_buckets=cfg.get("buckets", 1), clip_activations=cfg.get("clip"))
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.to(DEV).eval(), params, orc.conv_stride(cfg["image"], cfg["grid"])


@pytest.mark.parametrize("name", tuple(SHAPES))
def test_module_logits_match_the_reference_forward(name):
    """float32 loop form (the reference's arithmetic) vs NNUE.forward on unfiltered inputs."""
    cfg = SHAPES[name]
    model, params, stride = build(cfg)
    images = draw_images(cfg, params, stride, cfg["loop_batch"], torch.Generator().manual_seed(1234))
    keep = {}
    with torch.no_grad():
        ref = orc.model_forward_loop(params, images, stride, keep, cfg.get("clip"))
        got = model(images.to(DEV))
    worst, rep = error_report(got, ref, f"{name}_vs_loop_f32")
    assert worst <= 1.0, rep


@pytest.mark.parametrize("name", tuple(SHAPES))
def test_module_logits_match_the_exact_forward(name):
    """float64 chain (the exact value of the reference formula) vs NNUE.forward at the full batch of the shape, and the
    FeatureTransformer output itself against the same element-wise bar."""
    cfg = SHAPES[name]
    model, params, stride = build(cfg)
    images = draw_images(cfg, params, stride, cfg["batch"], torch.Generator().manual_seed(4321))
    p64 = {k: v.double() for k, v in params.items()}
    x = orc.conv_forward(images.double(), p64["conv.weight"], stride)
    idx, n = orc.active_lists(x, p64["visual_threshold"])
    ft = orc.ft_forward(p64["input.weight"], p64["input.bias"], idx, (idx >= 0).double())
    cls = [p64[f"classifier.classifier.{i}.{k}"] for i in (0, 2, 4) for k in ("weight", "bias")]
    if cls[0].dim() == 3:
        ref = orc.classifier_forward_bucketed(orc.pairwise(ft), orc.bucket_index(n, cls[0].shape[0], x[0].numel()), *cls, cfg.get("clip"))
    else:
        ref = orc.classifier_forward(orc.pairwise(ft), *cls, cfg.get("clip"))
    with torch.no_grad():
        got = model(images.to(DEV))
    worst, rep = error_report(got, ref, f"{name}_vs_exact_f64")
    assert worst <= 1.0, rep
    # the FeatureTransformer output of the same batch (sums of up to 28 k float32 terms) against the same bar
    from nnue_hip import lib
    with torch.no_grad():
        conv_out = lib.conv3x3_forward(images.to(DEV), model.conv.weight, stride)
        f, l1 = model.input.weight.shape
        path = lib.ft_path(f, conv_out[0].numel(), l1, conv_out.shape[0])
        assert path == "mfma" or os.environ.get("NNUE_FT_PATH", "auto") not in ("auto", "mfma")
        if path == "mfma":
            fm = lib.ftm_binarize(conv_out, model.visual_threshold, f, l1)
            assert torch.equal(fm.n.cpu().long(), n)
            worst_ft, rep_ft = error_report(lib.ftm_forward(model.input.weight, model.input.bias, fm), ft, f"{name}_ft_vs_exact_f64")
            assert worst_ft <= 1.0, rep_ft
