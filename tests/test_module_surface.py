"""Drop-in surface of nnue.py: names, constructor, attributes, parameter order,
state-dict keys, seed-identical init, and the loud failure without a GPU.  CPU only."""
import dataclasses

import pytest
import torch
import torch.nn as nn

import nnue
import nnue_oracle as orc
from conftest import MODEL_CASES, golden_model, has_gpu
from nnue_hip.lib import NnueHipError


def test_importable_names():
    for name in ("NNUE", "FeatureTransformer", "SimpleClassifier", "GridFeatureSet", "LossParams",
                 "StraightThroughBinary", "binary_activation_ste", "DEFAULT_L1", "DEFAULT_L2", "DEFAULT_L3", "EtinyNet"):
        assert hasattr(nnue, name), name
    assert (nnue.DEFAULT_L1, nnue.DEFAULT_L2, nnue.DEFAULT_L3) == (1024, 128, 32)


def test_grid_feature_set():
    # reference tests/test_model.py:48-66
    assert nnue.GridFeatureSet().num_features == 800
    assert nnue.GridFeatureSet(8, 12).num_features == 768
    assert nnue.GridFeatureSet(grid_size=4, num_features_per_square=8).num_features == 128
    assert dataclasses.is_dataclass(nnue.GridFeatureSet) and dataclasses.is_dataclass(nnue.LossParams)
    assert nnue.LossParams().pow_exp == 2.5


def test_constructor_defaults_and_attributes():
    m = nnue.NNUE()
    assert (m.l1_size, m.l2_size, m.l3_size, m.num_classes, m.input_size) == (1024, 128, 32, 1, 32)
    assert m.weight_decay == 5e-4 and isinstance(m.loss_params, nnue.LossParams)
    assert nnue.NNUE(weight_decay=1e-3).weight_decay == 1e-3  # reference tests/test_weight_decay.py:41-72
    for attr in ("feature_set", "input", "conv", "classifier", "nnue2score", "visual_threshold"):
        assert hasattr(m, attr)  # reference tests/test_model.py:116-119
    assert isinstance(m.conv, nn.Conv2d) and m.conv.bias is None and m.conv.stride == (3, 3) and m.conv.padding == (1, 1)
    assert isinstance(m.classifier.classifier, nn.Sequential)
    assert [type(x) for x in m.classifier.classifier] == [nn.Linear, nn.ReLU, nn.Linear, nn.ReLU, nn.Linear]
    assert (m.input.num_features, m.input.output_size) == (800, 1024)
    assert float(m.nnue2score) == 600.0 and torch.all(m.visual_threshold == 0.1)
    # the threshold stays re-assignable as a Parameter (reference tests/test_model.py:347-349)
    m.visual_threshold = nn.Parameter(torch.full_like(m.visual_threshold, -1.0))
    assert "visual_threshold" in dict(m.named_parameters())


def test_parameter_order_and_shapes():
    m = nnue.NNUE(nnue.GridFeatureSet(10, 8), 1024, 128, 32, num_classes=10)
    names = [k for k, _ in m.named_parameters()]
    assert names == list(orc.PARAM_KEYS)
    assert list(m.state_dict().keys()) == list(orc.PARAM_KEYS)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes["conv.weight"] == (8, 3, 3, 3) and shapes["input.weight"] == (800, 1024)
    assert shapes["classifier.classifier.0.weight"] == (128, 1024) and shapes["classifier.classifier.4.weight"] == (10, 32)
    assert sum(p.numel() for p in m.parameters()) == 956107


def test_conv_stride_formula():
    m = nnue.NNUE()
    assert m._calculate_conv_params(32, 10, 8) == (8, 3)
    assert m._calculate_conv_params(224, 32, 64) == (64, 7)
    assert m._calculate_conv_params(4, 10, 8) == (8, 1)
    assert nnue.NNUE(nnue.GridFeatureSet(32, 64), 64, 8, 8, num_classes=5, input_size=224).conv.stride == (7, 7)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_same_seed_same_weights_as_reference(name):
    cfg, params, _, _ = golden_model(name)
    torch.manual_seed(cfg["model_seed"])
    m = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"],
                  num_classes=cfg["classes"], input_size=cfg["input_size"])
    sd = m.state_dict()
    for k in orc.PARAM_KEYS:
        assert torch.equal(sd[k], params[k]), k
    m.load_state_dict(params)  # reference state dicts load unchanged


def test_rejected_configurations():
    with pytest.raises(ValueError):
        nnue.NNUE(num_ls_buckets=0)
    with pytest.raises(ValueError):
        nnue.NNUE(num_ls_buckets=65)
    with pytest.raises(ValueError):
        nnue.NNUE(l1_size=33)
    with pytest.raises(NotImplementedError):
        nnue.EtinyNet()


def test_cpu_tensors_run_and_the_gpu_layer_refuses_them():
    """Device dispatch: CPU tensors run the stock-torch formulas (tests/test_host_modules.py pins them);
    the HIP layer itself never accepts a CPU tensor."""
    from nnue_hip import lib
    m = nnue.NNUE(nnue.GridFeatureSet(4, 8), 32, 4, 4, num_classes=10)
    assert m(torch.randn(2, 3, 32, 32)).shape == (2, 10)
    assert m.input(torch.zeros(2, 3, dtype=torch.long), torch.ones(2, 3)).shape == (2, 32)
    assert m.classifier(torch.randn(2, 32)).shape == (2, 10)
    with pytest.raises(NnueHipError, match="GPU only"):
        lib.conv3x3_forward(torch.randn(2, 3, 32, 32), m.conv.weight, 3)


def test_clip_weights_and_quantized_data_on_cpu():
    torch.manual_seed(3)
    m = nnue.NNUE(nnue.GridFeatureSet(4, 8), 32, 4, 4, num_classes=10)
    with torch.no_grad():
        m.input.weight.mul_(50)
        m.conv.weight.mul_(50)
    conv_before = m.conv.weight.clone()
    m._clip_weights()
    assert float(m.input.weight.abs().max()) <= 1.0
    assert torch.equal(m.conv.weight, conv_before)  # conv is not clipped (nnue.py:528-539)
    q = m.get_quantized_model_data()
    assert not m.training
    assert set(q) == {"metadata", "conv_layer", "feature_transformer", "classifier"}
    assert q["metadata"]["quantized_one"] == 127.0 and abs(q["metadata"]["visual_threshold"] - 0.1) < 1e-7
    assert q["feature_transformer"]["weight"].dtype == torch.int8 and q["feature_transformer"]["bias"].dtype == torch.int32
    assert len(q["classifier"]["layers"]) == 3


def test_bucketed_layer_stacks_surface():
    """num_ls_buckets = K > 1 (build extension): same key names, a leading K on the classifier tensors,
    every slice drawn like nn.Linear in the oracle's order; K = 1 stays the reference's SimpleClassifier."""
    import nnue_oracle as orc
    torch.manual_seed(11)
    m = nnue.NNUE(nnue.GridFeatureSet(4, 8), 32, 8, 4, num_classes=5, num_ls_buckets=3)
    assert isinstance(m.classifier, nnue.BucketedClassifier) and m.num_ls_buckets == 3
    ref = orc.init_params(4, 8, 32, 8, 4, 5, 11, buckets=3)
    sd = m.state_dict()
    assert list(sd) == list(orc.PARAM_KEYS)
    for k, v in ref.items():
        assert torch.equal(sd[k], v), k
    assert sd["classifier.classifier.0.weight"].shape == (3, 8, 32) and sd["classifier.classifier.4.bias"].shape == (3, 5)
    one = nnue.NNUE(nnue.GridFeatureSet(4, 8), 32, 8, 4, num_classes=5, num_ls_buckets=1)
    assert isinstance(one.classifier, nnue.SimpleClassifier)
    # the selector (also what the kernels compute): min(K-1, n*K // (flat_ids+1))
    n = torch.tensor([0, 1, 120, 121, 242, 243, 363, 364, 968, 5000])
    assert torch.equal(nnue.bucket_of(n, 8, 968), orc.bucket_index(n, 8, 968))
    assert nnue.bucket_of(n, 8, 968).tolist() == [0, 0, 0, 0, 1, 2, 2, 3, 7, 7]
    # clip: only the classifier's Linear weights and the table
    with torch.no_grad():
        m.classifier.classifier[0].weight.fill_(3.0)
    m._clip_weights()
    assert float(m.classifier.classifier[0].weight.max()) == 1.0
