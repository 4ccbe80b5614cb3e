"""The engine's integer inference on the GPU (nnue_engine_evaluate_logits): bit-identical to the real C++ engine
(golden outputs of oracle/_ref/nnue_inference) and to its numpy restatement on fresh models.  ``-m gpu``."""
import json

import numpy as np
import pytest
import torch

import nnue
import nnue_engine_oracle as eo
import serialize
from conftest import GOLDEN
from nnue_hip import lib
from nnue_hip.engine import EngineFormatError, EngineModel

pytestmark = pytest.mark.gpu


def test_golden_outputs_of_the_reference_engine():
    z = np.load(GOLDEN / "engine_cases.npz")
    index = json.loads(str(z["index"]))
    for k, c in enumerate(index):
        engine = EngineModel.load(GOLDEN / c["model"])
        images = torch.from_numpy(z[f"case{k}/images"]).cuda()
        logits, density = engine.evaluate_logits(images, c["h"], c["w"])
        assert np.array_equal(logits.cpu().numpy().astype(np.float64), z[f"case{k}/logits"]), c
        assert float(np.abs(density.cpu().numpy().astype(np.float64) - z[f"case{k}/density"]).max()) < 5e-10, c


@pytest.mark.parametrize("arch", [(10, 8, 1024, 128, 32, 10, 32), (10, 8, 256, 32, 16, 100, 32), (4, 64, 64, 8, 8, 3, 40),
                                  (8, 4, 96, 16, 8, 1000, 17), (32, 64, 512, 32, 32, 10, 224)])
@pytest.mark.parametrize("threshold", (None, -0.5, 1.0))
def test_fresh_models_against_the_oracle(tmp_path, arch, threshold):
    g, fps, l1, l2, l3, classes, size = arch
    torch.manual_seed(g * 100 + l1)
    model = nnue.NNUE(nnue.GridFeatureSet(g, fps), l1, l2, l3, num_classes=classes, input_size=size)
    with torch.no_grad():
        model.input.weight.mul_(3.0)  # spread the quantised table; some int16 sums then wrap like the engine's
        model.input.bias.uniform_(-1, 1)
        if threshold is not None:
            model.visual_threshold.fill_(threshold)
    path = tmp_path / "m.nnue"
    serialize.serialize_model(model, path)
    ref = eo.load_nnue(path)
    engine = EngineModel.load(path)
    gen = torch.Generator().manual_seed(7)
    images = torch.randn(6, 3, size, size, generator=gen) * 1.5
    logits, density = engine.evaluate_logits(images.cuda())
    for i in range(images.shape[0]):
        want_logits, want_density = eo.evaluate_logits(ref, images[i].numpy().reshape(-1), size, size)
        assert np.array_equal(logits[i].cpu().numpy(), want_logits), (arch, threshold, i)
        assert float(density[i]) == float(want_density), (arch, threshold, i)
    if threshold == -0.5:
        assert float(density.min()) > 0.3  # cells the conv never produced count as active below zero


def test_evaluate_compiled_model_contract(tmp_path):
    import evaluate
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10).cuda()
    gen = torch.Generator().manual_seed(1)
    loader = [(torch.randn(16, 3, 32, 32, generator=gen), torch.randint(0, 10, (16,), generator=gen)) for _ in range(3)]
    loader.append((torch.randn(5, 3, 32, 32, generator=gen), torch.randint(0, 10, (5,), generator=gen)))
    metrics = evaluate.evaluate_compiled_model(model, loader, "nnue")
    assert set(metrics) == {"acc", "precision", "recall", "f1", "ms_per_sample", "latent_density"}
    assert 0.0 < metrics["latent_density"] < 1.0 and metrics["ms_per_sample"] > 0.0
    # the same numbers from the oracle, image by image (the reference's per-image loop)
    path = tmp_path / "m.nnue"
    serialize.serialize_model(model, path)
    ref = eo.load_nnue(path)
    outs, dens = [], []
    for images, _ in loader:
        for img in images:
            lg, dn = eo.evaluate_logits(ref, img.numpy().reshape(-1), 32, 32)
            outs.append(lg)
            dens.append(float(dn))
    want = evaluate.compute_metrics(torch.from_numpy(np.stack(outs)), torch.cat([y for _, y in loader]))
    assert all(abs(metrics[k] - want[k]) < 1e-12 for k in want)
    assert abs(metrics["latent_density"] - sum(dens) / len(dens)) < 1e-12
    with pytest.raises(ValueError, match="Unknown model type"):
        evaluate.evaluate_compiled_model(model, loader, "resnet")


def test_engine_errors(tmp_path):
    good = (GOLDEN / "nnue_tiny4x4.nnue").read_bytes()
    (tmp_path / "bad.nnue").write_bytes(b"XNUE" + good[4:])
    with pytest.raises(EngineFormatError, match="magic"):
        EngineModel.load(tmp_path / "bad.nnue")
    (tmp_path / "short.nnue").write_bytes(good[:200])
    with pytest.raises(EngineFormatError, match="truncated"):
        EngineModel.load(tmp_path / "short.nnue")
    engine = EngineModel.load(GOLDEN / "nnue_tiny4x4.nnue")
    with pytest.raises(lib.NnueHipError, match="overruns"):
        engine.evaluate_logits(torch.zeros(1, 3, 8, 40).cuda())  # stride from H, a wide image overruns the 4x4 grid
    with pytest.raises(ValueError):
        engine.evaluate_logits(torch.zeros(1, 1, 8, 8).cuda())
    with pytest.raises(lib.NnueHipError):
        engine.evaluate_logits(torch.zeros(1, 3, 32, 32))
