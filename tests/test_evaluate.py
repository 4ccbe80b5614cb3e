"""compute_metrics / evaluate_model against golden vectors from the reference's sklearn-based
implementation.  The metric arithmetic and the host bookkeeping run everywhere; the HIP confusion kernel and
the whole evaluate_model loop need the GPU."""
import json
import os

import numpy as np
import pytest
import torch

import evaluate
from conftest import load_npz

NAMES = ("acc", "f1", "precision", "recall")


def cases():
    z = load_npz("eval_metrics.npz")
    for name in sorted({k.split("/")[0] for k in z}):
        yield name, torch.from_numpy(z[f"{name}/outputs"]), torch.from_numpy(z[f"{name}/targets"]), z[f"{name}/metrics"]


def check(got, want, what):
    for n, w in zip(NAMES, want):
        assert abs(got[n] - w) <= 1e-12 + 1e-9 * abs(w), f"{what} {n}: {got[n]} vs {w}"


def test_compute_metrics_matches_sklearn_on_host_tensors():
    seen = 0
    for name, outputs, targets, want in cases():
        check(evaluate.compute_metrics(outputs, targets), want, name)
        seen += 1
    assert seen == 7


def test_metrics_from_confusion_edge_cases():
    assert evaluate.metrics_from_confusion(np.zeros((3, 3))) == {"acc": 0.0, "f1": 0.0, "precision": 0.0, "recall": 0.0}
    m = evaluate.metrics_from_confusion([[5, 0], [0, 7]])
    assert m == {"acc": 1.0, "f1": 1.0, "precision": 1.0, "recall": 1.0}
    # a class that is only ever predicted has weight 0; a class never predicted has precision 0 (zero_division=0)
    m = evaluate.metrics_from_confusion([[0, 4, 0], [0, 4, 0], [0, 0, 0]])
    assert m["acc"] == 0.5 and abs(m["precision"] - 0.25) < 1e-15 and abs(m["recall"] - 0.5) < 1e-15


@pytest.mark.gpu
def test_compute_metrics_on_gpu_tensors():
    for name, outputs, targets, want in cases():
        check(evaluate.compute_metrics(outputs.cuda(), targets.cuda()), want, name + " (gpu)")
    # accumulation over batches equals one big batch; 1-D outputs are treated as one column
    from nnue_hip import lib
    name, outputs, targets, _ = next(cases())
    o, t = outputs.cuda(), targets.cuda()
    conf = None
    for i in range(0, o.shape[0], 50):
        conf = lib.confusion_accumulate(o[i:i + 50].contiguous(), t[i:i + 50].contiguous(), conf)
    assert torch.equal(conf, lib.confusion_accumulate(o, t)) and int(conf.sum()) == o.shape[0]
    assert evaluate.compute_metrics(torch.tensor([0.9, 0.1, 0.7]).cuda(), torch.tensor([1, 0, 0]).cuda())["acc"] == pytest.approx(2 / 3)


@pytest.mark.gpu
def test_evaluate_model_matches_reference():
    import nnue
    z = load_npz("eval_model.npz")
    cfg = json.loads(str(z["cfg"]))
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"])
    model.load_state_dict({k[6:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state/")})
    model = model.cuda().eval()
    loader = [(torch.from_numpy(z[f"images{i}"]), torch.from_numpy(z[f"labels{i}"])) for i in range(cfg["batches"])]
    loss, metrics = evaluate.evaluate_model(model, loader, None, torch.device("cuda"))
    assert abs(loss - float(z["loss"])) <= 1e-4 * max(1.0, abs(float(z["loss"])))
    check(metrics, z["metrics"], "evaluate_model")  # same predictions -> identical metrics
    with pytest.raises(ValueError):
        evaluate.evaluate_model(model, [], None)
    # a label outside [0, C) is an error (F.cross_entropy would raise, sklearn would count it), not a silently skipped sample
    bad = [(loader[0][0], loader[0][1].clone())]
    bad[0][1][3] = cfg["classes"] + 2
    for graph in ("1", "0"):
        os.environ["NNUE_EVAL_GRAPH"] = graph
        try:
            with pytest.raises(ValueError, match="labels lie outside"):
                evaluate.evaluate_model(model, bad, None, torch.device("cuda"))
        finally:
            os.environ.pop("NNUE_EVAL_GRAPH")
    with pytest.raises(ValueError, match="labels lie outside"):
        evaluate.compute_metrics(torch.randn(4, 3).cuda(), torch.tensor([0, 1, 7, 2]).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("classes", (10, 1))
def test_graph_replayed_evaluation_equals_eager(monkeypatch, classes):
    """evaluate_model's hipGraph path (one plan per batch shape, ragged last batch, weights updated in between)
    against the eager forward: same confusion matrix, same loss."""
    import nnue
    torch.manual_seed(3)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=classes).cuda()
    gen = torch.Generator().manual_seed(4)
    sizes = [64, 64, 64, 23]
    loader = [(torch.randn(n, 3, 32, 32, generator=gen), torch.randint(0, max(classes, 2), (n,), generator=gen)) for n in sizes]
    results = {}
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("NNUE_EVAL_GRAPH", mode)
        results.setdefault(mode, []).append(evaluate.evaluate_model(model, loader))
    (l1, m1), (l2, m2) = results["1"]  # first call captures, second replays everything
    l0, m0 = results["0"][0]
    for loss, metrics in ((l1, m1), (l2, m2)):
        assert abs(loss - l0) <= 1e-12 * max(1.0, abs(l0))
        assert metrics == m0
    # parameters change in place between evaluations (what an optimizer step does): the plan follows
    with torch.no_grad():
        model.classifier.classifier[4].bias.add_(torch.randn(classes, device="cuda"))
        model.visual_threshold.add_(0.05)
    monkeypatch.setenv("NNUE_EVAL_GRAPH", "1")
    la, ma = evaluate.evaluate_model(model, loader)
    monkeypatch.setenv("NNUE_EVAL_GRAPH", "0")
    lb, mb = evaluate.evaluate_model(model, loader)
    assert abs(la - lb) <= 1e-12 * max(1.0, abs(lb)) and ma == mb
    if classes > 1:
        assert (la, ma) != (l0, m0)
