"""The step bench.py times -- NnueTrainer with its default flags under hipGraph replay (fused conv+map, FT forward with
the layer-1 slabs in its epilogue, merged FT backward with the d_w1 rider and the norm partials, deferred STE stage 2,
fused clip + SGD) -- at the BASELINE shapes themselves, three optimizer steps against the CPU oracle's explicit chain
(oracle.loss_and_grads_explicit + oracle.sgd_step; train.py:359-366).  ``-m gpu``.

The step has two kinds of discontinuity, and a float32 result on either side of one is equally "right":
``conv_out > thr`` (bit-exact feature ids need identical decisions; the oracle's MKL-DNN conv sums in another order than
ours, ~1e-7 relative) and the ReLU gates ``z > 0`` of the classifier's backward (found at the 224x224 shape: one
pre-activation of 16 384 at -2e-5 against a scale of 400 and a float32 summation error of 4e-4 flipped its gate and
moved d_w1 by 1.4 % of its maximum, with loss and logits unchanged).  Every batch is therefore drawn so that *for the
parameters of the step it is used in* no conv output lies within 1e-5 of its threshold and no pre-activation within
2e-5 x (largest pre-activation of its layer) of zero; samples that do are redrawn.  The margin check runs in float64 on
the CPU.
"""
import pytest
import torch
import torch.nn.functional as F

import nnue
import nnue_oracle as orc
from conftest import assert_close_grad, assert_close_logits
from nnue_hip.trainer import NnueTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda"
OPT = dict(lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0)  # config/train_nnue.py:29-36

SHAPES = {
    # BASELINE configs[1], [2] (K = 1 form), [3]                                   
    "c2": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=10, batch=512),
    "c3": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=100, batch=1024),
    "c4": dict(grid=32, fps=64, image=224, l1=1024, l2=128, l3=32, classes=1000, batch=128),
    # BASELINE configs[2] as worded: 8 layer-stack buckets + clipped ReLU (build extension: the oracle is the definition,
    # parity unpinned); "spread" draws samples of very different density so that every stack is exercised
    "c3k8": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=100, batch=1024, buckets=8, clip=1.0),
    "c3k8_spread": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=100, batch=1024, buckets=8, clip=1.0, spread=True),
}


def clean_batch(cfg, params, stride, gen, margin=1e-5, gate_margin=2e-5):
    """randn images / randint labels that keep the margins described above."""
    b, hw = cfg["batch"], cfg["image"]
    clip = cfg.get("clip")

    def draw(count):
        x = torch.randn(count, 3, hw, hw, generator=gen)
        if cfg.get("spread"):  # per-sample offset and gain (+ all-positive conv weights): feature counts over the whole range
            x = x * (0.5 + torch.rand(count, 1, 1, 1, generator=gen)) + (3.2 * torch.rand(count, 1, 1, 1, generator=gen) - 1.6)
        return x

    images = draw(b)
    p64 = {k: v.double() for k, v in params.items()}
    cls = [p64[f"classifier.classifier.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias")]
    t64 = p64["visual_threshold"].view(1, -1, 1, 1)
    for _ in range(64):
        x = F.conv2d(images.double(), p64["conv.weight"], stride=stride, padding=1)
        dirty = ((x - t64).abs() < margin).flatten(1).any(dim=1)
        idx, n = orc.active_lists(x, p64["visual_threshold"])
        l0 = orc.pairwise(orc.ft_forward(p64["input.weight"], p64["input.bias"], idx, (idx >= 0).double()))
        act = (lambda t: t.clamp(0, clip)) if clip is not None else F.relu
        if cls[0].dim() == 3:  # each sample through its own stack
            k = orc.bucket_index(n, cls[0].shape[0], x[0].numel())
            z1 = torch.einsum("bi,boi->bo", l0, cls[0][k]) + cls[1][k]
            z2 = torch.einsum("bi,boi->bo", act(z1), cls[2][k]) + cls[3][k]
        else:
            z1 = F.linear(l0, cls[0], cls[1])
            z2 = F.linear(act(z1), cls[2], cls[3])
        for z in (z1, z2):
            scale = gate_margin * float(z.abs().max())
            dirty |= (z.abs() < scale).any(dim=1)
            if clip is not None:
                dirty |= ((z - clip).abs() < scale).any(dim=1)
        if not bool(dirty.any()):
            break
        images[dirty] = draw(int(dirty.sum()))
    else:
        raise AssertionError("could not draw a batch that keeps the margins")
    return images, torch.randint(0, cfg["classes"], (b,), generator=gen)


@pytest.fixture
def one_rank_rccl(monkeypatch):
    """A one-rank RCCL process group with the collectives forced on (NNUE_DP_FORCE_COLLECTIVES): the data-parallel code path
    -- here the factor exchange of the 224x224 shape -- with everything a second rank would add except wire time."""
    import socket
    import torch.distributed as dist
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(port))
    monkeypatch.setenv("NNUE_DP_FORCE_COLLECTIVES", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    yield
    torch.cuda.synchronize()
    dist.destroy_process_group()


def test_three_steps_at_the_224_shape_through_the_factor_exchange(one_rank_rccl):
    """BASELINE configs[3]'s shape under data parallel: bit map + d_ft all-gathered (captured in the step graph), Gram norm
    and update on the gathered factors -- against the same oracle steps as the single-rank path."""
    test_three_steps_at_the_baseline_shape_follow_the_oracle("c4", expect_factor_exchange=True)


@pytest.mark.parametrize("name", ("c2", "c3", "c4", "c3k8", "c3k8_spread"))
def test_three_steps_at_the_baseline_shape_follow_the_oracle(name, expect_factor_exchange=False):
    cfg = SHAPES[name]
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"],
                      num_classes=cfg["classes"], input_size=cfg["image"], num_ls_buckets=cfg.get("buckets", 1),
                      clip_activations=cfg.get("clip"))
    if cfg.get("spread"):
        with torch.no_grad():
            model.conv.weight.abs_()  # an image offset then moves all conv outputs of a sample the same way
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    stride = orc.conv_stride(cfg["image"], cfg["grid"])
    model = model.to(DEV)
    tr = NnueTrainer(model, cfg["batch"], (cfg["image"], cfg["image"]), use_graph=True, input_slots=2, **OPT)
    # the path under test is the one the bench line reports
    import os
    if os.environ.get("NNUE_FT_PATH", "auto") not in ("auto", "mfma"):
        pytest.skip("another FeatureTransformer kernel family is forced (NNUE_FT_PATH)")
    assert tr.ft_path == "mfma" and tr.use_graph
    assert tr.factor_exchange == expect_factor_exchange and (not expect_factor_exchange or tr.capture_collectives)
    if name in ("c2", "c3"):  # every default fusion is on (a knob set to its non-default value switches its own off)
        on = lambda k: os.environ.get(k, "1") != "0"  # noqa: E731
        merged = os.environ.get("NNUE_FTM_SPLIT_BACKWARD", "0") != "1"
        assert tr.merge_backward == merged and tr.ride_dw1 == (on("NNUE_FTM_RIDE_DW1") and merged)
        assert tr.fuse_l1 == on("NNUE_FUSE_L1") and tr.defer_ste == on("NNUE_DEFER_STE")
    gen = torch.Generator().manual_seed(77)
    bufs = {}
    for s in range(3):
        images, labels = clean_batch(cfg, params, stride, gen)
        before = {k: v.clone() for k, v in params.items()}
        ref_logits, ref_loss, ref_grads, keep = orc.loss_and_grads_explicit(params, images, labels, stride, cfg.get("clip"))
        if cfg.get("spread") and s == 0:
            assert len(set(keep["bucket"].tolist())) == cfg["buckets"], "the spread batch must reach every layer stack"
        ref_norm = orc.sgd_step(params, ref_grads, bufs, OPT["lr"], OPT["momentum"], OPT["weight_decay"], OPT["max_grad_norm"])
        slot = s % 2  # steps 1.. replay the captured full-step graph, on alternating input slots
        was = {k: v.detach().clone() for k, v in tr.p.items()}
        loss = tr.step(images.to(DEV), labels.to(DEV), slot=slot)
        torch.cuda.synchronize()
        n_mean, n_max = tr.active_stats()
        assert n_max == int(keep["n"].max()) and abs(n_mean - float(keep["n"].float().mean())) < 1e-2, "feature counts differ"
        # the north star's own acceptance sentence: every logit within 1e-4 * max(1, |logit|) of the reference forward
        assert_close_logits(tr.logits, ref_logits, f"{name} step {s} logits")
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss))), (s, float(loss), float(ref_loss))
        assert abs(float(tr.grad_norm) - float(ref_norm)) <= 1e-4 * float(ref_norm), (s, float(tr.grad_norm), float(ref_norm))
        if getattr(tr, "grads_materialised", True):
            got = tr.layout.views(tr.flat_grads)
            for k, ref in ref_grads.items():
                assert_close_grad(got[k], ref, f"{name} step {s} grad {k}")
        for k in orc.TRAINABLE_KEYS:
            # the parameters themselves and, much sharper, what the step changed
            assert_close_grad(tr.p[k], params[k], f"{name} step {s} {k}", rtol=1e-5)
            got_d, ref_d = (tr.p[k] - was[k]).cpu().double(), (params[k] - before[k]).double()
            # a float32 parameter cannot move by less than its own rounding: two ulps of the largest entry are the floor
            floor = 2 * 2.0 ** -23 * float(params[k].abs().max())
            err, scale = float((got_d - ref_d).abs().max()), float(ref_d.abs().max())
            assert err <= 2e-4 * scale + floor, f"{name} step {s} update of {k}: err {err:.3e}, scale {scale:.3e}, floor {floor:.1e}"
    assert float(model.nnue2score) == 600.0


def test_a_step_group_at_the_224_shape_through_the_factor_exchange(one_rank_rccl):
    """The same group under data parallel: factors all-gathered and the collective captured in the group's graph, the table update on
    the gathered factors also forming this rank's next forward."""
    test_a_step_group_at_the_224_shape_follows_the_oracle(expect_factor_exchange=True)


def test_a_step_group_at_the_224_shape_follows_the_oracle(expect_factor_exchange=False):
    """BASELINE configs[3]'s shape as bench.py runs it: a group of steps replayed as one hipGraph in which the table update of a
    step also forms the next step's FeatureTransformer forward (nnue_ftm_backward_weight_update_forward, two alternating maps).
    One single step (records the plans), then a group of three, against four oracle steps."""
    cfg = SHAPES["c4"]
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"],
                      input_size=cfg["image"])
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    stride = orc.conv_stride(cfg["image"], cfg["grid"])
    tr = NnueTrainer(model.to(DEV), cfg["batch"], (cfg["image"], cfg["image"]), use_graph=True, input_slots=3, **OPT)
    import os
    assert tr.factor_exchange == expect_factor_exchange and (not expect_factor_exchange or tr.capture_collectives)
    if not tr.fuse_next_forward:
        assert os.environ.get("NNUE_FUSE_NEXT_FORWARD") == "0" or os.environ.get("NNUE_FUSE_TABLE_UPDATE") == "0" \
            or os.environ.get("NNUE_FT_PATH", "auto") not in ("auto", "mfma") or os.environ.get("NNUE_FTM_BF16") == "0" \
            or os.environ.get("NNUE_FTM_BF_KT64") == "0", "the 224x224 shape must take the fused pass by default"
        pytest.skip("a knob switched the fused update + next forward off")
    gen = torch.Generator().manual_seed(78)
    bufs, ref_losses = {}, []
    for s in range(4):
        images, labels = clean_batch(cfg, params, stride, gen)
        ref_logits, ref_loss, ref_grads, keep = orc.loss_and_grads_explicit(params, images, labels, stride, None)
        orc.sgd_step(params, ref_grads, bufs, OPT["lr"], OPT["momentum"], OPT["weight_decay"], OPT["max_grad_norm"])
        ref_losses.append(float(ref_loss))
        if s == 0:
            first = tr.step(images.to(DEV), labels.to(DEV), slot=0)
            assert abs(float(first) - ref_losses[0]) <= 1e-4 * max(1.0, abs(ref_losses[0]))
            was = {k: v.detach().clone() for k, v in tr.p.items()}
            before = {k: v.clone() for k, v in params.items()}
        else:
            tr.inputs[s - 1][0].copy_(images)
            tr.inputs[s - 1][1].copy_(labels)
    losses = tr.step_many((0, 1, 2))
    torch.cuda.synchronize()
    assert (((0, 1, 2), "many") in tr._g_local) and tr.steps_done == 4
    for s in range(3):
        assert abs(float(losses[s]) - ref_losses[s + 1]) <= 1e-4 * max(1.0, abs(ref_losses[s + 1])), (s, float(losses[s]), ref_losses[s + 1])
    assert_close_logits(tr.logits, ref_logits, "logits of the group's last step")
    n_mean, n_max = tr.active_stats()
    assert n_max == int(keep["n"].max()) and abs(n_mean - float(keep["n"].float().mean())) < 1e-2
    for k in orc.TRAINABLE_KEYS:
        assert_close_grad(tr.p[k], params[k], f"after the group: {k}", rtol=1e-5)
        got_d, ref_d = (tr.p[k] - was[k]).cpu().double(), (params[k] - before[k]).double()
        floor = 4 * 2.0 ** -23 * float(params[k].abs().max())
        err, scale = float((got_d - ref_d).abs().max()), float(ref_d.abs().max())
        assert err <= 2e-4 * scale + floor, f"three steps of {k}: err {err:.3e}, scale {scale:.3e}, floor {floor:.1e}"
