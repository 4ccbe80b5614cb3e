#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports /root/reference/nnue.py and serialize.py unmodified, runs them on the
CPU on seeded inputs, and stores inputs + outputs as data (npz / .nnue bytes /
sha256).  Nothing of the reference's source is copied; the fixtures are
numbers.  Every fixture records the seed and constructor arguments it came
from so that it can be regenerated.

Fixtures
--------
model_<name>.npz      whole-model cases: state dict, images, labels, conv_out,
                      idx, val, ft, logits, loss, every parameter gradient
                      (F.cross_entropy + backward, train.py:250-254, :360-361).
                      Seeds are chosen so that no conv_out element lies within
                      MARGIN of its threshold: feature ids then do not depend
                      on the conv's summation order.
ft_cases.npz          hand-built (idx, val) inputs for FeatureTransformer.forward
                      alone: all -1, M=1, repeated / unsorted ids, ids >= F,
                      non-unit values, with outputs and gradients.
step_<name>.npz       three optimizer steps (zero_grad, backward, clip_grad_norm_,
                      SGD momentum + weight decay: train.py:359-366, :457-464)
                      with the parameters after each step.
nnue_<name>.nnue      serialize_model output (serialize.py:500-528), small models.
nnue_index.json       sha256 + size of every .nnue (incl. the 2.8 MB C2-arch file and
                      the four visual_threshold variants of tests/test_model.py:499-545,
                      which are not stored as bytes).
big_c2.npz            C2 architecture (800 -> 1024/128/32 -> 10), B=16: logits, loss,
                      gradient norms and strided gradient samples; the state dict is
                      NOT stored -- it is re-drawn from the seed (init order is part
                      of the drop-in contract).
"""
import hashlib
import io
import json
import os
import sys
import tempfile
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np
import torch

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import nnue as ref_nnue  # noqa: E402
import serialize as ref_serialize  # noqa: E402

OUT = Path(__file__).resolve().parent
MARGIN = 1e-3
torch.set_num_threads(1)


def build(cfg, seed):
    torch.manual_seed(seed)
    return ref_nnue.NNUE(
        feature_set=ref_nnue.GridFeatureSet(cfg["grid"], cfg["fps"]),
        l1_size=cfg["l1"], l2_size=cfg["l2"], l3_size=cfg["l3"],
        num_classes=cfg["classes"], input_size=cfg["input_size"],
    )


def draw_batch(cfg, seed):
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(cfg["batch"], 3, cfg["image"], cfg["image"], generator=g)
    labels = torch.randint(0, cfg["classes"], (cfg["batch"],), generator=g)
    return images, labels


def margin_ok(model, images):
    with torch.no_grad():
        x = model.conv(images)
        return float((x - model.visual_threshold.view(1, -1, 1, 1)).abs().min()) > MARGIN


def np_state(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


MODEL_CASES = {
    # name: grid, fps, input_size (model), image (fed), l1, l2, l3, classes, batch
    "tiny4x4": dict(grid=4, fps=8, input_size=32, image=32, l1=32, l2=4, l3=4, classes=10, batch=3),
    "grid8": dict(grid=8, fps=4, input_size=32, image=32, l1=64, l2=4, l3=8, classes=10, batch=4),
    "c1arch": dict(grid=10, fps=8, input_size=32, image=32, l1=64, l2=32, l3=8, classes=10, batch=4),
    "tiny96": dict(grid=4, fps=8, input_size=32, image=96, l1=32, l2=4, l3=4, classes=10, batch=2),
    "odd": dict(grid=5, fps=3, input_size=32, image=40, l1=24, l2=7, l3=5, classes=3, batch=5),
}


def model_case(name, cfg):
    seed = 0
    while True:
        model = build(cfg, 100 + seed)
        images, labels = draw_batch(cfg, 1234 + seed)
        if margin_ok(model, images):
            break
        seed += 1
    state = np_state(model)
    model.train()
    logits = model(images)
    loss = torch.nn.functional.cross_entropy(logits, labels.long())
    loss.backward()
    with torch.no_grad():
        conv_out = model.conv(images)
        bits = (conv_out > model.visual_threshold.view(1, -1, 1, 1)).float()
        idx, val = model._to_sparse_features(bits)
        ft = model.input(idx, val)
    out = {f"state/{k}": v for k, v in state.items()}
    for k, p in model.named_parameters():
        assert (p.grad is None) == (k == "nnue2score")
        if p.grad is not None:
            out[f"grad/{k}"] = p.grad.numpy().copy()
    out.update(
        images=images.numpy(), labels=labels.numpy(), conv_out=conv_out.numpy(), idx=idx.numpy(),
        val=val.numpy(), ft=ft.numpy(), logits=logits.detach().numpy(), loss=np.float32(loss.item()),
        cfg=json.dumps(dict(cfg, model_seed=100 + seed, data_seed=1234 + seed, stride=model.conv.stride[0])),
    )
    np.savez_compressed(OUT / f"model_{name}.npz", **out)
    print(f"model_{name}: seed+{seed} n_mean={float((idx >= 0).sum(1).float().mean()):.1f} "
          f"M={idx.shape[1]} loss={loss.item():.6f}")
    return model


def ft_cases():
    """Stand-alone FeatureTransformer.forward inputs (nnue.py:686-710)."""
    torch.manual_seed(7)
    rows, width = 40, 24
    ft = ref_nnue.FeatureTransformer(rows, width)
    with torch.no_grad():
        ft.bias.copy_(torch.randn(width) * 0.05)
    g = torch.Generator().manual_seed(11)
    cases = {
        "all_pad": (torch.full((3, 5), -1), torch.zeros(3, 5)),
        "m1_pad": (torch.full((2, 1), -1), torch.zeros(2, 1)),
        "m1_one": (torch.tensor([[3], [39]]), torch.tensor([[1.0], [0.5]])),
        "repeat": (torch.tensor([[5, 5, 5, 7, -1], [0, 0, 39, 39, 39]]),
                   torch.tensor([[1.0, 2.0, -0.5, 1.0, 9.0], [1.0, 1.0, 0.25, 0.25, 0.25]])),
        "unsorted": (torch.stack([torch.randperm(rows, generator=g)[:9] for _ in range(4)]),
                     torch.ones(4, 9)),
        "overflow": (torch.tensor([[38, 39, 40, 41, 1000, -1, 2], [100, -1, -1, 39, 39, 0, -5]]),
                     torch.tensor([[1.0, 1.0, 1.0, 1.0, 1.0, 7.0, 1.0], [2.0, 3.0, 4.0, 1.0, 1.0, 1.0, 8.0]])),
        "values": (torch.randint(-1, rows, (6, 13), generator=g), torch.randn(6, 13, generator=g)),
        "holes": (torch.tensor([[-1, 4, -1, -1, 6, -1, 8, -1]]), torch.tensor([[5.0, 1.0, 5.0, 5.0, 2.0, 5.0, 3.0, 5.0]])),
    }
    out = {"weight": ft.weight.detach().numpy().copy(), "bias": ft.bias.detach().numpy().copy()}
    for name, (idx, val) in cases.items():
        idx = idx.long()
        val = val.float().clone().requires_grad_(True)
        ft.zero_grad()
        y = ft(idx, val)
        up = torch.randn(y.shape, generator=g)
        (y * up).sum().backward()
        out[f"{name}/idx"] = idx.numpy()
        out[f"{name}/val"] = val.detach().numpy()
        out[f"{name}/out"] = y.detach().numpy()
        out[f"{name}/upstream"] = up.numpy()
        # an all-padding batch never touches the table: the reference leaves weight.grad = None
        out[f"{name}/d_weight"] = (ft.weight.grad if ft.weight.grad is not None
                                   else torch.zeros_like(ft.weight)).numpy().copy()
        out[f"{name}/d_bias"] = ft.bias.grad.numpy().copy()
        out[f"{name}/d_val"] = (val.grad if val.grad is not None else torch.zeros_like(val)).numpy().copy()
    np.savez_compressed(OUT / "ft_cases.npz", **out)
    print("ft_cases:", ", ".join(cases))


def step_case(name, cfg, steps=3):
    """train.py:359-366 with config/train_nnue.py's SGD settings."""
    model = build(cfg, 300)
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=2e-4)
    out = {f"state0/{k}": v for k, v in np_state(model).items()}
    out["cfg"] = json.dumps(dict(cfg, model_seed=300, lr=0.01, momentum=0.9, weight_decay=2e-4,
                                 max_grad_norm=1.0, stride=model.conv.stride[0]))
    model.train()
    for s in range(steps):
        images, labels = draw_batch(cfg, 5000 + s)
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(model(images), labels.long())
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        out[f"images{s}"] = images.numpy()
        out[f"labels{s}"] = labels.numpy()
        out[f"loss{s}"] = np.float32(loss.item())
        out[f"gradnorm{s}"] = np.float32(norm.item())
        for k, v in np_state(model).items():
            out[f"state{s + 1}/{k}"] = v
    np.savez_compressed(OUT / f"step_{name}.npz", **out)
    print(f"step_{name}: losses", [float(out[f'loss{s}']) for s in range(steps)])


def serialize_bytes(model):
    with tempfile.TemporaryDirectory() as d:
        path = Path(d) / "m.nnue"
        with redirect_stdout(io.StringIO()):
            ref_serialize.serialize_model(model, path)
        return path.read_bytes()


def nnue_files(models):
    index = {}
    for name, model in models.items():
        blob = serialize_bytes(model)
        (OUT / f"nnue_{name}.nnue").write_bytes(blob)
        index[f"nnue_{name}.nnue"] = dict(sha256=hashlib.sha256(blob).hexdigest(), size=len(blob), stored=True,
                                          source=f"model_{name}.npz state")
    # C2 architecture: hash only, model re-drawn from the seed by the test
    c2 = dict(grid=10, fps=8, input_size=32, l1=1024, l2=128, l3=32, classes=10)
    blob = serialize_bytes(build(c2, 0))
    index["c2arch_seed0"] = dict(sha256=hashlib.sha256(blob).hexdigest(), size=len(blob), stored=False, cfg=c2, seed=0)
    # visual_threshold sweep of the reference's own serialisation test (tests/test_model.py:499-545)
    cfg = MODEL_CASES["tiny4x4"]
    for thr in (-0.5, 0.0, 0.5, 1.0):
        m = build(cfg, 100)
        m.visual_threshold = torch.nn.Parameter(torch.full_like(m.visual_threshold, thr))
        blob = serialize_bytes(m)
        index[f"tiny4x4_thr{thr}"] = dict(sha256=hashlib.sha256(blob).hexdigest(), size=len(blob), stored=False,
                                          cfg=cfg, seed=100, threshold=thr)
    # a model with out-of-range weights exercises the clamp(-1,1) + clamp(+-127) path
    m = build(MODEL_CASES["grid8"], 9)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(30.0)
    state = np_state(m)
    blob = serialize_bytes(m)
    (OUT / "nnue_saturated.nnue").write_bytes(blob)
    np.savez_compressed(OUT / "nnue_saturated_state.npz", **state)
    index["nnue_saturated.nnue"] = dict(sha256=hashlib.sha256(blob).hexdigest(), size=len(blob), stored=True,
                                        source="nnue_saturated_state.npz", cfg=MODEL_CASES["grid8"])
    (OUT / "nnue_index.json").write_text(json.dumps(index, indent=1, sort_keys=True))
    print("nnue files:", {k: v["size"] for k, v in index.items()})


def big_c2():
    cfg = dict(grid=10, fps=8, input_size=32, image=32, l1=1024, l2=128, l3=32, classes=10, batch=16)
    model = build(cfg, 0)
    images, labels = draw_batch(cfg, 1234)
    with torch.no_grad():
        x = model.conv(images)
        gap = float((x - model.visual_threshold.view(1, -1, 1, 1)).abs().min())
    model.train()
    logits = model(images)
    loss = torch.nn.functional.cross_entropy(logits, labels.long())
    loss.backward()
    out = dict(cfg=json.dumps(dict(cfg, model_seed=0, data_seed=1234, stride=model.conv.stride[0], min_gap=gap)),
               logits=logits.detach().numpy(), loss=np.float32(loss.item()))
    with torch.no_grad():
        bits = (x > model.visual_threshold.view(1, -1, 1, 1)).float()
        idx, _ = model._to_sparse_features(bits)
        out["idx"] = idx.numpy().astype(np.int16)
        out["ft_sample"] = model.input(idx, (idx >= 0).float())[:, ::37].numpy()
    for k, p in model.named_parameters():
        if p.grad is not None:
            out[f"gradnorm/{k}"] = np.float32(p.grad.norm().item())
            out[f"gradsample/{k}"] = p.grad.flatten()[::101].numpy().copy()
    # parameter checksums pin the "same seed -> same init" contract
    for k, v in model.state_dict().items():
        out[f"statesum/{k}"] = np.float64(v.double().sum().item())
        out[f"statesample/{k}"] = v.flatten()[::997].numpy().copy()
    np.savez_compressed(OUT / "big_c2.npz", **out)
    print(f"big_c2: loss={loss.item():.6f} min_gap={gap:.2e} M={idx.shape[1]}")


def main():
    models = {name: model_case(name, cfg) for name, cfg in MODEL_CASES.items()}
    ft_cases()
    step_case("c1arch", dict(MODEL_CASES["c1arch"], batch=8))
    step_case("tiny96", dict(MODEL_CASES["tiny96"], batch=4))
    # serialisation mutates (eval + clip), so rebuild from the stored states' seeds
    fresh = {}
    for name in ("tiny4x4", "grid8", "c1arch"):
        cfg = json.loads(str(np.load(OUT / f"model_{name}.npz")["cfg"]))
        fresh[name] = build(cfg, cfg["model_seed"])
    nnue_files(fresh)
    big_c2()
    total = sum(p.stat().st_size for p in OUT.iterdir() if p.suffix in (".npz", ".nnue", ".json"))
    print(f"total fixture bytes: {total}")


if __name__ == "__main__":
    main()
