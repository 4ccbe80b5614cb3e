"""Pins oracle/nnue_oracle.py against golden vectors produced by the real reference.

CPU only.  Every oracle function used as a checker elsewhere is exercised here:
loop form and explicit form, forward, backward, optimizer step, init order,
quantiser.
"""
import json

import numpy as np
import pytest
import torch

import nnue_oracle as orc
from conftest import MODEL_CASES, assert_close_grad, assert_close_logits, golden_model, load_npz


@pytest.mark.parametrize("name", MODEL_CASES)
def test_loop_form_matches_reference(name):
    cfg, params, grads, data = golden_model(name)
    logits, loss, g, keep = orc.loss_and_grads_loop(params, data["images"], data["labels"], cfg["stride"])
    # feature ids: bit-exact, values exact
    assert torch.equal(keep["idx"], data["idx"])
    assert torch.equal(keep["val"], data["val"])
    assert_close_logits(keep["conv_out"], data["conv_out"], "conv_out", rtol=1e-6)
    assert_close_logits(keep["ft"], data["ft"], "ft", rtol=1e-5)
    assert_close_logits(logits, data["logits"], "logits", rtol=1e-5)
    assert abs(float(loss) - float(data["loss"])) <= 1e-5 * max(1.0, abs(float(data["loss"])))
    assert set(g) == set(grads)
    for k in grads:
        assert_close_grad(g[k], grads[k], k, rtol=1e-5)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_explicit_form_matches_reference(name):
    cfg, params, grads, data = golden_model(name)
    logits, loss, g, keep = orc.loss_and_grads_explicit(params, data["images"], data["labels"], cfg["stride"])
    # fixed-capacity id lists hold the same ids as the reference's padded lists
    m = data["idx"].shape[1]
    assert torch.equal(keep["idx"][:, :m], data["idx"])
    assert bool((keep["idx"][:, m:] == -1).all())
    assert torch.equal(keep["n"], (data["idx"] >= 0).sum(1))
    assert_close_logits(keep["ft"], data["ft"], "ft", rtol=2e-5)
    assert_close_logits(logits, data["logits"], "logits", rtol=2e-5)
    assert abs(float(loss) - float(data["loss"])) <= 2e-5 * max(1.0, abs(float(data["loss"])))
    for k in grads:
        assert_close_grad(g[k], grads[k], k, rtol=2e-5)


def test_ft_standalone_cases():
    z = load_npz("ft_cases.npz")
    w, b = torch.from_numpy(z["weight"]), torch.from_numpy(z["bias"])
    names = sorted({k.split("/")[0] for k in z if "/" in k})
    assert len(names) == 8
    for name in names:
        idx = torch.from_numpy(z[f"{name}/idx"])
        val = torch.from_numpy(z[f"{name}/val"])
        up = torch.from_numpy(z[f"{name}/upstream"])
        ref_out = torch.from_numpy(z[f"{name}/out"])
        for fwd in (orc.ft_forward_loop, orc.ft_forward):
            assert_close_logits(fwd(w, b, idx, val), ref_out, f"{name} out", rtol=1e-6)
        d_w, d_b, d_val = orc.ft_backward(w, idx, val, up)
        assert_close_grad(d_w, torch.from_numpy(z[f"{name}/d_weight"]), f"{name} dW", rtol=1e-6)
        assert_close_grad(d_b, torch.from_numpy(z[f"{name}/d_bias"]), f"{name} db", rtol=1e-6)
        ref_dval = torch.from_numpy(z[f"{name}/d_val"])
        assert float((d_val - ref_dval).abs().max()) <= 1e-6 * max(1.0, float(ref_dval.abs().max())), name


@pytest.mark.parametrize("name", ("c1arch", "tiny96"))
def test_sgd_steps_match_reference(name):
    z = load_npz(f"step_{name}.npz")
    cfg = json.loads(str(z["cfg"]))
    params = {k[7:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state0/")}
    bufs = {}
    for s in range(3):
        images, labels = torch.from_numpy(z[f"images{s}"]), torch.from_numpy(z[f"labels{s}"])
        for explicit in (False, True):
            fn = orc.loss_and_grads_explicit if explicit else orc.loss_and_grads_loop
            _, loss, grads, _ = fn(params, images, labels, cfg["stride"])
            assert abs(float(loss) - float(z[f"loss{s}"])) <= 5e-5 * max(1.0, abs(float(z[f"loss{s}"])))
        norm = orc.sgd_step(params, grads, bufs, cfg["lr"], cfg["momentum"], cfg["weight_decay"], cfg["max_grad_norm"])
        assert abs(float(norm) - float(z[f"gradnorm{s}"])) <= 5e-5 * float(z[f"gradnorm{s}"])
        for k in params:
            ref = torch.from_numpy(z[f"state{s + 1}/{k}"])
            assert_close_grad(params[k], ref, f"step{s} {k}", rtol=5e-5)
    # nnue2score is never updated (no gradient)
    assert float(params["nnue2score"]) == 600.0


def test_init_order_matches_reference_seed():
    """torch.manual_seed(s); NNUE(...) in the reference == oracle.init_params(..., seed=s)."""
    for name in MODEL_CASES:
        cfg, params, _, _ = golden_model(name)
        mine = orc.init_params(cfg["grid"], cfg["fps"], cfg["l1"], cfg["l2"], cfg["l3"], cfg["classes"], cfg["model_seed"])
        assert list(mine) == list(orc.PARAM_KEYS)
        for k in orc.PARAM_KEYS:
            assert torch.equal(mine[k], params[k]), (name, k)


def test_big_c2_against_reference():
    z = load_npz("big_c2.npz")
    cfg = json.loads(str(z["cfg"]))
    params = orc.init_params(cfg["grid"], cfg["fps"], cfg["l1"], cfg["l2"], cfg["l3"], cfg["classes"], cfg["model_seed"])
    for k, v in params.items():
        assert abs(float(v.double().sum()) - float(z[f"statesum/{k}"])) <= 1e-9 * max(1.0, abs(float(z[f"statesum/{k}"])))
        assert np.array_equal(v.flatten()[::997].numpy(), z[f"statesample/{k}"])
    g = torch.Generator().manual_seed(cfg["data_seed"])
    images = torch.randn(cfg["batch"], 3, cfg["image"], cfg["image"], generator=g)
    labels = torch.randint(0, cfg["classes"], (cfg["batch"],), generator=g)
    logits, loss, grads, keep = orc.loss_and_grads_explicit(params, images, labels, cfg["stride"])
    m = z["idx"].shape[1]
    assert np.array_equal(keep["idx"][:, :m].numpy(), z["idx"].astype(np.int64))
    assert_close_logits(keep["ft"][:, ::37], torch.from_numpy(z["ft_sample"]), "ft", rtol=2e-5)
    assert_close_logits(logits, torch.from_numpy(z["logits"]), "logits", rtol=5e-5)
    assert abs(float(loss) - float(z["loss"])) <= 5e-5 * float(z["loss"])
    for k in orc.TRAINABLE_KEYS:
        ref_norm = float(z[f"gradnorm/{k}"])
        assert abs(float(grads[k].norm()) - ref_norm) <= 1e-4 * ref_norm, k
        sample = grads[k].flatten()[::101]
        assert float((sample - torch.from_numpy(z[f"gradsample/{k}"])).abs().max()) <= 1e-4 * float(grads[k].abs().max()), k


def test_quantiser_and_size_formula(nnue_index):
    t = torch.tensor([0.0078125, 0.0234375, -0.0234375, 1.9, -3.0, 0.5])
    # *64 -> 0.5, 1.5, -1.5, 121.6, -192, 32 ; round-half-even, clamp +-127
    assert orc.quantize(t).tolist() == [0, 2, -2, 122, -127, 32]
    assert orc.quantize(t, bias=True).tolist() == [0, 2, -2, 122, -192, 32]
    assert orc.nnue_file_size(128, 8, 32, 4, 4, 10) == nnue_index["nnue_tiny4x4.nnue"]["size"]
    assert orc.nnue_file_size(256, 4, 64, 4, 8, 10) == nnue_index["nnue_grid8.nnue"]["size"] == 38204
    assert orc.nnue_file_size(800, 8, 64, 32, 8, 10) == nnue_index["nnue_c1arch.nnue"]["size"] == 110312
    assert orc.nnue_file_size(800, 8, 1024, 128, 32, 10) == nnue_index["c2arch_seed0"]["size"] == 2836856


def test_adam_steps_match_reference():
    """oracle.adam_step against clip_grad_norm_ + torch.optim.Adam run on the reference model."""
    z = load_npz("step_adam_c1arch.npz")
    cfg = json.loads(str(z["cfg"]))
    params = {k[7:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state0/")}
    state = {}
    for s in range(3):
        images, labels = torch.from_numpy(z[f"images{s}"]), torch.from_numpy(z[f"labels{s}"])
        _, loss, grads, _ = orc.loss_and_grads_explicit(params, images, labels, cfg["stride"])
        assert abs(float(loss) - float(z[f"loss{s}"])) <= 5e-5 * max(1.0, abs(float(z[f"loss{s}"])))
        norm = orc.adam_step(params, grads, state, cfg["lr"], cfg["weight_decay"], cfg["max_grad_norm"])
        assert abs(float(norm) - float(z[f"gradnorm{s}"])) <= 5e-5 * float(z[f"gradnorm{s}"])
        for k in params:
            ref = torch.from_numpy(z[f"state{s + 1}/{k}"])
            # Adam divides by sqrt(v): an element whose gradient is ~0 amplifies rounding, so compare on the update scale
            assert float((params[k] - ref).abs().max()) <= 2e-5 + 1e-4 * float(ref.abs().max()), (s, k)
