"""Train-driver row: config modules, Adam step, ragged batches, optimizer state in torch's
format, the epoch loop with best-F1 checkpoints in the reference's layout."""
import json
import textwrap

import pytest
import torch

import nnue
import nnue_oracle as orc
from conftest import assert_close_grad, load_npz
from nnue_hip import train_loop

CONFIG_TEXT = """
    name = "nnue_vision_test"
    batch_size = 16
    num_workers = 0
    num_classes = 10
    l1_size = 256
    l2_size = 32
    l3_size = 16
    input_size = 32
    grid_size = 10
    num_features_per_square = 8
    learning_rate = {lr}
    weight_decay = 2e-4
    momentum = 0.9
    optimizer_type = "{opt}"
    subset = 1.0
    max_epochs = {epochs}
    max_grad_norm = 1.0
    use_augmentation = False
    keep_alive = True
    log_dir = "logs"
    project_name = "test"
"""


def write_config(tmp_path, opt="sgd", lr=0.01, epochs=2):
    path = tmp_path / f"train_{opt}.py"
    path.write_text(textwrap.dedent(CONFIG_TEXT.format(opt=opt, lr=lr, epochs=epochs)))
    return path


# ------------------------------------------------------------------------------------------ CPU
def test_load_config_like_the_reference(tmp_path):
    cfg = train_loop.load_config(write_config(tmp_path))
    assert (cfg.name, cfg.batch_size, cfg.optimizer_type, cfg.max_grad_norm) == ("nnue_vision_test", 16, "sgd", 1.0)
    with pytest.raises(train_loop.ConfigError, match="not found"):
        train_loop.load_config(tmp_path / "missing.py")
    (tmp_path / "cfg.txt").write_text("x = 1")
    with pytest.raises(train_loop.ConfigError, match="must be a Python file"):
        train_loop.load_config(tmp_path / "cfg.txt")
    (tmp_path / "broken.py").write_text("x = = 1")
    with pytest.raises(train_loop.ConfigError, match="Failed to load configuration"):
        train_loop.load_config(tmp_path / "broken.py")


def test_train_model_argument_errors():
    with pytest.raises(ValueError, match="Unknown model type"):
        train_loop.train_model(object(), "etinynet", [], [])
    with pytest.raises(ValueError, match="needs train_loader"):
        train_loop.train_model(object(), "nnue")


# ------------------------------------------------------------------------------------------ GPU
def make_loader(n_batches, batch, seed, last=None):
    g = torch.Generator().manual_seed(seed)
    sizes = [batch] * n_batches + ([last] if last else [])
    data = []
    for n in sizes:
        labels = torch.randint(0, 10, (n,), generator=g)
        images = torch.randn(n, 3, 32, 32, generator=g) + labels.view(-1, 1, 1, 1).float() * 0.3  # learnable signal
        data.append((images, labels))
    return data


@pytest.mark.gpu
@pytest.mark.parametrize("count", (37, 956112))
def test_adam_kernel_against_oracle(count):
    from nnue_hip import lib
    gen = torch.Generator().manual_seed(count)
    p0 = torch.randn(count, generator=gen)
    params, state = {"visual_threshold": p0.clone()}, {}
    dp = p0.clone().cuda()
    m, v = torch.zeros(count).cuda(), torch.zeros(count).cuda()
    step = torch.zeros(1, dtype=torch.int32).cuda()
    norm = torch.zeros(()).cuda()
    for s in range(4):
        grad = torch.randn(count, generator=gen) * (3.0 if s == 1 else 1e-3)
        ref_norm = orc.adam_step(params, {"visual_threshold": grad / 2}, state, 1e-3, 2e-4, 1.0)
        lib.adam_step(dp, grad.cuda(), m, v, step, 1e-3, weight_decay=2e-4, max_norm=1.0, grad_scale=0.5, norm_out=norm)
        assert int(step) == s + 1
        assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
        assert float((dp.cpu() - params["visual_threshold"]).abs().max()) <= 2e-6
        assert_close_grad(m, state["visual_threshold"]["m"], "exp_avg")
        assert_close_grad(v, state["visual_threshold"]["v"], "exp_avg_sq")


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", (False, True))
def test_adam_trainer_follows_reference_trajectory(use_graph):
    from nnue_hip.trainer import NnueTrainer
    z = load_npz("step_adam_c1arch.npz")
    cfg = json.loads(str(z["cfg"]))
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"])
    model.load_state_dict({k[7:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state0/")})
    tr = NnueTrainer(model.cuda(), cfg["batch"], (32, 32), lr=cfg["lr"], weight_decay=cfg["weight_decay"],
                     max_grad_norm=cfg["max_grad_norm"], optimizer="adam", use_graph=use_graph)
    for s in range(3):
        loss = tr.step(torch.from_numpy(z[f"images{s}"]).cuda(), torch.from_numpy(z[f"labels{s}"]).cuda().long())
        assert abs(float(loss) - float(z[f"loss{s}"])) <= 2e-4 * max(1.0, abs(float(z[f"loss{s}"])))
        assert abs(float(tr.grad_norm) - float(z[f"gradnorm{s}"])) <= 2e-4 * float(z[f"gradnorm{s}"])
        for k, v in model.state_dict().items():
            ref = torch.from_numpy(z[f"state{s + 1}/{k}"])
            assert float((v.cpu() - ref).abs().max()) <= 5e-5 + 2e-4 * float(ref.abs().max()), (s, k)
    # the state it hands to checkpoints loads into a real torch.optim.Adam
    sd = tr.optimizer_state_dict()
    opt = torch.optim.Adam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    opt.load_state_dict(sd)
    assert 0 not in sd["state"] and float(sd["state"][1]["step"]) == 3.0  # nnue2score (index 0) has no state


@pytest.mark.gpu
def test_ragged_last_batch_equals_exact_small_batch():
    from nnue_hip.trainer import NnueTrainer
    torch.manual_seed(3)
    a = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10).cuda()
    b = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10).cuda()
    b.load_state_dict(a.state_dict())
    opt = dict(lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0)
    big, small = NnueTrainer(a, 16, (32, 32), **opt), NnueTrainer(b, 5, (32, 32), **opt)
    (full_x, full_y), (x, y) = make_loader(1, 16, 9)[0], make_loader(1, 5, 10)[0]
    small.step(full_x[:5].cuda(), full_y[:5].cuda())
    big.step(full_x[:5].cuda(), full_y[:5].cuda())  # 5 of 16: padded
    la, lb = big.step(x.cuda(), y.cuda()), small.step(x.cuda(), y.cuda())
    assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(lb)))
    assert abs(float(big.grad_norm) - float(small.grad_norm)) <= 1e-5 * float(small.grad_norm)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert_close_grad(p, q, k, rtol=1e-5)
    with pytest.raises(ValueError):
        big.step(torch.zeros(17, 3, 32, 32).cuda(), torch.zeros(17, dtype=torch.long).cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ("sgd", "adam"))
def test_train_model_epochs_checkpoint_and_learning(tmp_path, opt):
    cfg = train_loop.load_config(write_config(tmp_path, opt=opt, lr=0.02 if opt == "sgd" else 2e-3, epochs=3))
    train, val, test = make_loader(6, 16, 1, last=7), make_loader(2, 16, 2), make_loader(1, 16, 3)
    logs = []
    torch.manual_seed(0)
    res = train_loop.run_training(cfg, train, val, test, checkpoint_dir=tmp_path / "ckpt", log=logs.append)
    assert res.steps == 3 * 7 and len(res.history) == 3 and len(logs) == 3 and res.test is not None
    assert res.history[-1]["train/epoch_loss"] < res.history[0]["train/epoch_loss"]  # it learns the synthetic signal
    assert train_loop.train_model(cfg, "nnue", train, val) == 0  # the reference's call shape and return value
    if res.checkpoint_path is None:
        pytest.skip("validation F1 never rose above 0 on this tiny synthetic set")
    ckpt = torch.load(res.checkpoint_path, weights_only=True)
    assert set(ckpt) == {"epoch", "model_state_dict", "optimizer_state_dict", "metrics", "config_name"}  # checkpoint_manager.py:45-51
    assert ckpt["config_name"] == "nnue_vision_test" and set(ckpt["metrics"]) == {"val_f1", "val_loss"}
    assert ckpt["epoch"] == res.best_epoch and abs(ckpt["metrics"]["val_f1"] - res.best_val_f1) < 1e-12
    fresh = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10)
    fresh.load_state_dict(ckpt["model_state_dict"])
    make = torch.optim.SGD if opt == "sgd" else torch.optim.Adam
    kw = dict(momentum=0.9) if opt == "sgd" else {}
    make(fresh.parameters(), lr=0.01, **kw).load_state_dict(ckpt["optimizer_state_dict"])  # checkpoint_manager.py:75-85
    # serialize.py reads a bare state dict (its documented inconsistency with the manager's layout)
    import serialize
    torch.save(ckpt["model_state_dict"], tmp_path / "bare.pt")
    serialize.serialize_model(serialize.load_model_from_checkpoint(tmp_path / "bare.pt"), tmp_path / "m.nnue")
    assert (tmp_path / "m.nnue").stat().st_size == orc.nnue_file_size(800, 8, 256, 32, 16, 10)


@pytest.mark.gpu
def test_train_model_with_bucketed_layer_stacks(tmp_path):
    """The two build extensions read from the config module (num_ls_buckets, clip_activations): epochs, evaluation,
    best-F1 checkpoint and export of a K = 4 model."""
    path = write_config(tmp_path, opt="sgd", lr=0.02, epochs=2)
    path.write_text(path.read_text() + "num_ls_buckets = 4\nclip_activations = 1.0\n")
    cfg = train_loop.load_config(path)
    train, val = make_loader(5, 16, 1, last=9), make_loader(2, 16, 2)
    torch.manual_seed(0)
    net = train_loop.build_model(cfg, "cuda")
    assert net.num_ls_buckets == 4 and net.classifier.clip_activations == 1.0
    res = train_loop.run_training(cfg, train, val, model=net, checkpoint_dir=tmp_path / "ckpt", log=lambda s: None)
    assert res.steps == 2 * 6 and all(torch.isfinite(torch.tensor(r["train/epoch_loss"])) for r in res.history)
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    assert sd["classifier.classifier.0.weight"].shape == (4, 32, 256)
    import serialize
    torch.save(sd, tmp_path / "k4.pt")
    model = serialize.load_model_from_checkpoint(tmp_path / "k4.pt")
    assert model.num_ls_buckets == 4
    serialize.serialize_model(model, tmp_path / "k4.nnue")
    assert (tmp_path / "k4.nnue").stat().st_size == orc.nnue_file_size(800, 8, 256, 32, 16, 10, buckets=4)


@pytest.mark.gpu
def test_density_check_switches_the_kernel_family_and_keeps_the_trajectory(monkeypatch):
    """The epoch-end density check (train_loop._density_check): below the measured crossover the trainer is rebuilt on the
    gather kernels, optimizer state and step count carried over; the steps that follow equal those of a trainer that never
    switched (same arithmetic, another summation order)."""
    from nnue_hip.trainer import NnueTrainer
    opt = dict(lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0)
    gen = torch.Generator().manual_seed(3)
    batches = [(torch.randn(16, 3, 32, 32, generator=gen).cuda(), torch.randint(0, 10, (16,), generator=gen).cuda()) for _ in range(4)]

    def fresh():
        torch.manual_seed(0)
        m = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10).cuda()
        return m, NnueTrainer(m, 16, (32, 32), **opt)

    m_a, tr_a = fresh()
    m_b, tr_b = fresh()
    assert tr_a.ft_path == "mfma"
    for x, y in batches[:2]:
        tr_a.step(x, y)
        tr_b.step(x, y)
    logs = []
    assert train_loop._density_check(tr_a, logs.append) is tr_a and not logs  # shipped thresholds: the product form stays
    monkeypatch.setitem(train_loop.GATHER_BELOW_DENSITY, "cache_resident_table", 1.0)
    tr_c = train_loop._density_check(tr_a, logs.append)
    assert tr_c is not tr_a and tr_c.ft_path == "bits" and tr_c.steps_done == 2 and "mfma -> bits" in logs[0]
    assert torch.equal(tr_c.flat_momentum, tr_a.flat_momentum) and torch.equal(tr_c.flat_params, tr_b.flat_params)
    for x, y in batches[2:]:
        la, lb = tr_c.step(x, y), tr_b.step(x, y)
        assert abs(float(la) - float(lb)) <= 1e-4 * max(1.0, abs(float(lb)))
    for k in tr_b.p:
        assert_close_grad(tr_c.p[k], tr_b.p[k], k, rtol=2e-4)
    monkeypatch.setenv("NNUE_FT_DENSITY_SWITCH", "0")
    assert train_loop._density_check(tr_b, logs.append) is tr_b
