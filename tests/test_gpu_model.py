"""Whole-model parity on the GPU through the drop-in nn.Module surface: logits, loss and every parameter
gradient against the reference's golden vectors; the reference's own structural tests restated; training
steps against the reference's SGD trajectory.  Needs an MI355X: ``-m gpu``."""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import nnue
import nnue_oracle as orc
from conftest import MODEL_CASES, assert_close_grad, assert_close_logits, golden_model, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(cfg, state=None, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    m = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"],
                  num_classes=cfg["classes"], input_size=cfg["input_size"])
    if state is not None:
        m.load_state_dict(state)
    return m.to(DEV)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_forward_backward_matches_reference(name):
    cfg, params, grads, data = golden_model(name)
    model = build(cfg, params)
    model.train()
    logits = model(data["images"].to(DEV))
    loss = F.cross_entropy(logits, data["labels"].to(DEV).long())
    loss.backward()
    assert_close_logits(logits, data["logits"], "logits")
    assert abs(float(loss) - float(data["loss"])) <= 1e-4 * max(1.0, abs(float(data["loss"])))
    assert model.nnue2score.grad is None  # reference tests/test_model.py:179-182
    for k, p in model.named_parameters():
        if k != "nnue2score":
            assert p.grad is not None, k
            assert_close_grad(p.grad, grads[k], k)


@pytest.mark.parametrize("name", ("tiny4x4", "c1arch"))
def test_standalone_modules_match_reference(name):
    """model.input(idx, val) and model.classifier(x) called on their own, as the reference's tests do
    (tests/test_model.py:595-626)."""
    cfg, params, _, data = golden_model(name)
    model = build(cfg, params)
    idx, val = data["idx"].to(DEV), data["val"].to(DEV).requires_grad_(True)
    ft = model.input(idx, val)
    assert_close_logits(ft, data["ft"], "ft")
    half = cfg["l1"] // 2
    l0 = torch.cat([ft[:, :half] * ft[:, half:], ft[:, :half]], dim=1)
    logits = model.classifier(l0)
    assert logits.shape == (idx.shape[0], cfg["classes"])
    assert_close_logits(logits, data["logits"], "logits via stand-alone modules")
    logits.square().sum().backward()
    assert val.grad is not None and model.input.weight.grad is not None
    # _to_sparse_features reproduces the reference's padded format from a {0,1} map
    bits = (data["conv_out"] > params["visual_threshold"].view(1, -1, 1, 1)).float()
    i2, v2 = model._to_sparse_features(bits.to(DEV))
    assert torch.equal(i2.cpu(), data["idx"]) and torch.equal(v2.cpu(), data["val"])


def test_big_c2_matches_reference():
    z = load_npz("big_c2.npz")
    cfg = json.loads(str(z["cfg"]))
    model = build(cfg, seed=cfg["model_seed"])
    gen = torch.Generator().manual_seed(cfg["data_seed"])
    images = torch.randn(cfg["batch"], 3, cfg["image"], cfg["image"], generator=gen)
    labels = torch.randint(0, cfg["classes"], (cfg["batch"],), generator=gen)
    logits = model(images.to(DEV))
    loss = F.cross_entropy(logits, labels.to(DEV))
    loss.backward()
    assert_close_logits(logits, torch.from_numpy(z["logits"]), "logits")
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * float(z["loss"])
    for k, p in model.named_parameters():
        if k == "nnue2score":
            continue
        ref_norm = float(z[f"gradnorm/{k}"])
        assert abs(float(p.grad.norm()) - ref_norm) <= 1e-4 * ref_norm, k
        got = p.grad.flatten()[::101].cpu()
        assert float((got - torch.from_numpy(z[f"gradsample/{k}"])).abs().max()) <= 1e-4 * float(p.grad.abs().max()), k


@pytest.mark.parametrize("name", ("c1arch", "tiny96"))
def test_training_steps_match_reference(name):
    """zero_grad / backward / clip_grad_norm_ / SGD.step exactly as train.py:359-366, on the GPU model."""
    z = load_npz(f"step_{name}.npz")
    cfg = json.loads(str(z["cfg"]))
    state0 = {k[7:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state0/")}
    model = build(cfg, state0)
    opt = torch.optim.SGD(model.parameters(), lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["weight_decay"])
    model.train()
    for s in range(3):
        opt.zero_grad()
        loss = F.cross_entropy(model(torch.from_numpy(z[f"images{s}"]).to(DEV)), torch.from_numpy(z[f"labels{s}"]).to(DEV).long())
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), cfg["max_grad_norm"])
        opt.step()
        assert abs(float(loss) - float(z[f"loss{s}"])) <= 2e-4 * max(1.0, abs(float(z[f"loss{s}"])))
        assert abs(float(norm) - float(z[f"gradnorm{s}"])) <= 2e-4 * float(z[f"gradnorm{s}"])
        for k, v in model.state_dict().items():
            assert_close_grad(v, torch.from_numpy(z[f"state{s + 1}/{k}"]), f"step {s} {k}", rtol=2e-4)


def test_reference_structural_checks():
    """Finite outputs for 96x96 images on a 32-built model, B in {1, 2} (reference tests/test_model.py:316-328);
    eval / no_grad work; gradients flow where the reference says they do (:179-226)."""
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(4, 8), 32, 4, 4, num_classes=10).to(DEV)
    for b in (1, 2):
        y = model(torch.randn(b, 3, 96, 96, device=DEV))
        assert y.shape == (b, 10) and bool(torch.isfinite(y).all())
    model.eval()
    with torch.no_grad():
        y = model(torch.randn(2, 3, 32, 32, device=DEV))
    assert not y.requires_grad
    model.train()
    F.cross_entropy(model(torch.randn(4, 3, 32, 32, device=DEV)), torch.randint(0, 10, (4,), device=DEV)).backward()
    assert model.nnue2score.grad is None
    assert float(model.input.weight.grad.norm()) > 1e-8
    assert float(model.classifier.classifier[0].weight.grad.norm()) > 1e-8
    for k in ("conv.weight", "input.bias", "classifier.classifier.0.bias", "visual_threshold"):
        assert dict(model.named_parameters())[k].grad is not None, k


def test_density_extremes_and_pixel_gradient():
    """Thresholds of -1e6 / +1e6 (everything / nothing active, reference tests/test_model.py:347-362) and
    a gradient request for the pixels (nnue_conv3x3_backward_input)."""
    torch.manual_seed(1)
    cfg = dict(grid=10, fps=8, l1=64, l2=32, l3=8, classes=10, input_size=32)
    model = build(cfg)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    images = torch.randn(6, 3, 32, 32)
    labels = torch.randint(0, 10, (6,))
    for thr in (-1e6, 1e6, 0.0):
        with torch.no_grad():
            model.visual_threshold.fill_(thr)
        params["visual_threshold"] = torch.full((8,), thr)
        model.zero_grad()
        x = images.to(DEV).requires_grad_(True)
        logits = model(x)
        F.cross_entropy(logits, labels.to(DEV)).backward()
        ref_logits, _, ref_grads, _ = orc.loss_and_grads_loop(params, images, labels, 3)
        assert_close_logits(logits, ref_logits, f"logits thr={thr}")
        for k, p in model.named_parameters():
            if k == "nnue2score":
                continue
            if ref_grads[k] is None:  # nothing active: the reference's graph never reaches conv / threshold
                assert not bool(p.grad.any()), k
            else:
                assert_close_grad(p.grad, ref_grads[k], f"{k} thr={thr}")
        # pixel gradient against autograd of the loop form
        xi = images.clone().requires_grad_(True)
        q = {k: v.clone() for k, v in params.items()}
        ref_loss = F.cross_entropy(orc.model_forward_loop(q, xi, 3), labels)
        if ref_loss.requires_grad:  # with nothing active the reference's output does not depend on the pixels
            ref_loss.backward()
        if xi.grad is None:
            assert not bool(x.grad.any())
        else:
            assert_close_grad(x.grad, xi.grad, f"d_images thr={thr}")


@pytest.mark.parametrize("name", ("tiny4x4", "c1arch", "tiny96"))
def test_composed_public_pieces_carry_the_reference_gradients(name):
    """The reference's forward spelled out with the public pieces (nnue.py:640-669): conv -> binary_activation_ste ->
    _to_sparse_features -> self.input -> pairwise -> self.classifier.  The values returned by _to_sparse_features stay
    attached to the map (nnue.py:628-633), so conv weight and threshold receive the golden gradients this way too."""
    cfg, params, grads, data = golden_model(name)
    model = build(cfg, params)
    x = F.conv2d(data["images"].to(DEV), model.conv.weight, stride=model.conv.stride, padding=1)
    bits = nnue.binary_activation_ste(x, model.visual_threshold.view(1, -1, 1, 1))
    idx, val = model._to_sparse_features(bits)
    assert torch.equal(idx.cpu(), data["idx"]) and torch.equal(val.detach().cpu(), data["val"])
    assert val.requires_grad and not idx.requires_grad
    ft = model.input(idx, val)
    half = cfg["l1"] // 2
    logits = model.classifier(torch.cat([ft[:, :half] * ft[:, half:], ft[:, :half]], dim=1))
    F.cross_entropy(logits, data["labels"].to(DEV).long()).backward()
    assert_close_logits(logits, data["logits"], "logits")
    for k, p in model.named_parameters():
        if k != "nnue2score":
            assert_close_grad(p.grad, grads[k], k)


def test_sparse_values_follow_the_loop_form():
    """_to_sparse_features on a map with non-binary entries: values are the map's own entries (nnue.py:601-606) and the
    gradient lands on exactly those positions."""
    gen = torch.Generator().manual_seed(5)
    m = torch.rand(5, 3, 4, 6, generator=gen)
    m[3] = 0.0  # a sample with nothing active
    model = nnue.NNUE(nnue.GridFeatureSet(4, 3), 8, 4, 4, num_classes=2, input_size=8).to(DEV)
    a = m.clone().requires_grad_(True)
    ref_idx, ref_val = orc.to_sparse_features_loop(a)
    w = torch.randn(ref_val.shape, generator=gen)
    (ref_val * w).sum().backward()
    b = m.to(DEV).requires_grad_(True)
    idx, val = model._to_sparse_features(b)
    assert torch.equal(idx.cpu(), ref_idx) and torch.equal(val.detach().cpu(), ref_val.detach())
    (val * w.to(DEV)).sum().backward()
    assert torch.equal(b.grad.cpu(), a.grad)


@pytest.mark.parametrize("shape", ((2, 10, 10, 1, 5), (3, 33, 41, 3, 8), (1, 96, 96, 7, 16)))
def test_pixel_gradient_kernel(shape):
    from nnue_hip import lib
    b, h, w, stride, fps = shape
    gen = torch.Generator().manual_seed(b + h)
    wt = torch.randn(fps, 3, 3, 3, generator=gen)
    gh, gw = lib.conv_out_hw(h, w, stride)
    d = torch.randn(b, fps, gh, gw, generator=gen)
    ref = torch.nn.grad.conv2d_input((b, 3, h, w), wt.double(), d.double(), stride=stride, padding=1)
    got = lib.conv3x3_backward_input(d.to(DEV), wt.to(DEV), (b, 3, h, w), stride)
    assert_close_grad(got, ref, "d_images", rtol=1e-5)


def test_serialize_from_gpu_model(tmp_path):
    """serialize_model works on a model living on the GPU and gives the reference's bytes."""
    import serialize
    from conftest import GOLDEN
    cfg, params, _, _ = golden_model("grid8")
    model = build(cfg, params)
    serialize.serialize_model(model, tmp_path / "m.nnue")
    assert (tmp_path / "m.nnue").read_bytes() == (GOLDEN / "nnue_grid8.nnue").read_bytes()
