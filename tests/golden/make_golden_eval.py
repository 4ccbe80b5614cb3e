#!/usr/bin/env python3
"""Golden vectors for the widened rows (SURVEY section 8f.1 evaluation, 8f.2 Adam step), generated from the REAL reference:
evaluate.compute_metrics (sklearn scorers, evaluate.py:23-59) and evaluate.evaluate_model (evaluate.py:62-87)
on the reference NNUE, CPU.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_eval.py
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import evaluate as ref_eval  # noqa: E402
import nnue as ref_nnue  # noqa: E402

OUT = Path(__file__).resolve().parent
torch.set_num_threads(1)


def metric_cases():
    g = torch.Generator().manual_seed(21)
    cases = {}
    # name: (outputs, targets)
    cases["c10_random"] = (torch.randn(257, 10, generator=g), torch.randint(0, 10, (257,), generator=g))
    cases["c3_skewed"] = (torch.randn(64, 3, generator=g) + torch.tensor([2.0, 0.0, -2.0]), torch.randint(0, 3, (64,), generator=g))
    t = torch.randint(0, 5, (40,), generator=g)
    cases["c5_perfect"] = (torch.nn.functional.one_hot(t, 5).float() * 3, t)
    cases["c6_missing_classes"] = (torch.randn(50, 6, generator=g) * torch.tensor([1, 1, 0, 1, 0, 0.0]) - torch.tensor([0, 0, 9, 0, 9, 9.0]),
                                   torch.randint(0, 2, (50,), generator=g) * 3)  # targets in {0,3}, preds in {0,1,3}
    ties = torch.zeros(8, 4)
    ties[:, 1] = 1.0
    ties[:, 3] = 1.0  # first maximum wins
    cases["c4_ties"] = (ties, torch.tensor([1, 3, 1, 3, 0, 2, 1, 1]))
    cases["binary_single_output"] = (torch.rand(33, 1, generator=g), torch.randint(0, 2, (33,), generator=g))
    cases["c100_large"] = (torch.randn(1000, 100, generator=g), torch.randint(0, 100, (1000,), generator=g))
    out = {}
    for name, (o, t) in cases.items():
        m = ref_eval.compute_metrics(o, t)
        out[f"{name}/outputs"] = o.numpy()
        out[f"{name}/targets"] = t.numpy()
        out[f"{name}/metrics"] = np.array([m["acc"], m["f1"], m["precision"], m["recall"]], dtype=np.float64)
    np.savez_compressed(OUT / "eval_metrics.npz", **out)
    print("metric cases:", ", ".join(cases))


def model_case():
    cfg = dict(grid=10, fps=8, l1=256, l2=32, l3=16, classes=10, input_size=32)
    seed = 0
    while True:  # a seed whose conv outputs all keep clear of the threshold: ids do not depend on the conv's summation order
        torch.manual_seed(500 + seed)
        model = ref_nnue.NNUE(ref_nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"])
        with torch.no_grad():
            model.classifier.classifier[4].weight.mul_(40.0)  # spread the predictions over the classes
            model.classifier.classifier[4].bias.zero_()
        model.eval()
        g = torch.Generator().manual_seed(77 + seed)
        batches = [(torch.randn(n, 3, 32, 32, generator=g), torch.randint(0, 10, (n,), generator=g)) for n in (16, 16, 16, 9)]
        with torch.no_grad():
            gap = min(float((model.conv(x) - model.visual_threshold.view(1, -1, 1, 1)).abs().min()) for x, _ in batches)
        if gap > 1e-4:
            break
        seed += 1
    with torch.no_grad():
        loss, metrics = ref_eval.evaluate_model(model, batches, None, torch.device("cpu"))
        preds = torch.cat([model(x).argmax(1) for x, _ in batches])
    out = {f"images{i}": x.numpy() for i, (x, _) in enumerate(batches)}
    out.update({f"labels{i}": y.numpy() for i, (_, y) in enumerate(batches)})
    out.update({f"state/{k}": v.numpy() for k, v in model.state_dict().items()})
    out["loss"] = np.float64(loss)
    out["metrics"] = np.array([metrics["acc"], metrics["f1"], metrics["precision"], metrics["recall"]], dtype=np.float64)
    out["cfg"] = json.dumps(dict(cfg, batches=len(batches), min_gap=gap, model_seed=500 + seed))
    np.savez_compressed(OUT / "eval_model.npz", **out)
    print(f"evaluate_model: seed+{seed} loss {loss:.6f} metrics {metrics} min_gap {gap:.2e} distinct predictions {preds.unique().numel()}")


def adam_case(steps=3):
    """train.py:359-366 with the optimizer create_optimizer builds for optimizer_type != "sgd" (train.py:465-470):
    torch.optim.Adam(model.parameters(), lr, weight_decay), plus clip_grad_norm_(1.0)."""
    cfg = dict(grid=10, fps=8, input_size=32, image=32, l1=64, l2=32, l3=8, classes=10, batch=8)
    torch.manual_seed(300)
    model = ref_nnue.NNUE(ref_nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"])
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=2e-4)
    out = {f"state0/{k}": v.numpy().copy() for k, v in model.state_dict().items()}
    out["cfg"] = json.dumps(dict(cfg, model_seed=300, lr=1e-3, weight_decay=2e-4, max_grad_norm=1.0, stride=model.conv.stride[0]))
    model.train()
    for s in range(steps):
        g = torch.Generator().manual_seed(6000 + s)
        images = torch.randn(cfg["batch"], 3, 32, 32, generator=g)
        labels = torch.randint(0, cfg["classes"], (cfg["batch"],), generator=g)
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(images), labels.long())
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        out[f"images{s}"], out[f"labels{s}"] = images.numpy(), labels.numpy()
        out[f"loss{s}"], out[f"gradnorm{s}"] = np.float32(loss.item()), np.float32(norm.item())
        for k, v in model.state_dict().items():
            out[f"state{s + 1}/{k}"] = v.numpy().copy()
    np.savez_compressed(OUT / "step_adam_c1arch.npz", **out)
    print("adam trajectory losses", [float(out[f"loss{s}"]) for s in range(steps)])


if __name__ == "__main__":
    if "--adam-only" in sys.argv:
        adam_case()
    else:
        metric_cases()
        model_case()
        adam_case()
