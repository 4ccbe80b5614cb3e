"""Bucketed layer stacks (num_ls_buckets = K > 1; BASELINE configs[2]) on the GPU against the CPU restatement in
oracle/nnue_oracle.py.  The reference trains one stack (nnue.py:713-738, serialize.py:57), so for K > 1 the oracle is
the definition: **parity unpinned** -- what IS pinned is that K = 1 through the bucketed entry points is bit-identical
to the reference path.  ``-m gpu``."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import nnue
import nnue_oracle as orc
from conftest import assert_close_grad, assert_close_logits
from nnue_hip import lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


def group_reference(n, K, P):
    """numpy statement of nnue_bucket_group."""
    n = np.asarray(n, dtype=np.int64)
    bucket = np.minimum(K - 1, np.maximum(0, (n * K) // (P + 1) if P > 0 else n))
    tiles = (len(n) + 15) // 16 + K
    rows = np.full(tiles * 16, -1, dtype=np.int64)
    seg = [0]
    for k in range(K):
        members = np.nonzero(bucket == k)[0]
        rows[seg[-1]: seg[-1] + len(members)] = members
        seg.append(seg[-1] + (len(members) + 15) // 16 * 16)
    tile_bucket = np.full(tiles, -1, dtype=np.int64)
    for k in range(K):
        cnt = int((bucket == k).sum())
        for t in range(seg[k] // 16, (seg[k] + cnt + 15) // 16):
            tile_bucket[t] = k
    return bucket, rows, tile_bucket, np.asarray(seg)


@pytest.mark.parametrize("B,K,P", ((1, 2, 968), (17, 8, 968), (512, 8, 968), (1031, 64, 65536), (300, 3, 0), (64, 8, 100)))
def test_bucket_group_kernel(B, K, P):
    rng = np.random.RandomState(B + K)
    n = rng.randint(0, (P if P > 0 else K + 2) + 1, size=B)
    if B == 512:
        n[:] = 400  # everything in one bucket (what randn images give at the reference's threshold)
    plan = lib.bucket_group(torch.from_numpy(n).to(DEV, torch.int32), P, K)
    bucket, rows, tile_bucket, seg = group_reference(n, K, P)
    assert np.array_equal(plan.bucket.cpu().numpy(), bucket)
    assert np.array_equal(plan.seg.cpu().numpy(), seg)
    assert np.array_equal(plan.rows.cpu().numpy(), rows)
    assert np.array_equal(plan.tile_bucket.cpu().numpy(), tile_bucket)
    # refilling a static plan gives the same answer (hipGraph replays rely on it)
    again = lib.bucket_group(torch.from_numpy(n).to(DEV, torch.int32), P, K, plan=plan)
    assert again is plan and np.array_equal(plan.rows.cpu().numpy(), rows)


def stacked(K, l1, l2, l3, c, gen):
    def mk(*s):
        return torch.randn(*s, generator=gen) * 0.3
    return [mk(K, l2, l1), mk(K, l2), mk(K, l3, l2), mk(K, l3), mk(K, c, l3), mk(K, c)]


BUCKET_MIXES = ("random", "one", "sparse")


def draw_buckets(mix, B, K, gen):
    if mix == "one":
        return torch.full((B,), K // 2, dtype=torch.int64)
    if mix == "sparse":  # only two of the stacks ever see a sample
        return torch.where(torch.rand(B, generator=gen) < 0.3, 0, K - 1).long()
    return torch.randint(0, K, (B,), generator=gen)


@pytest.mark.parametrize("shape", ((96, 256, 32, 16, 10, 4), (200, 1024, 128, 32, 100, 8), (37, 24, 7, 5, 3, 3), (1024, 128, 32, 32, 10, 8)))
@pytest.mark.parametrize("mix", BUCKET_MIXES)
@pytest.mark.parametrize("pairwise,clip", ((True, 0.0), (False, 1.0)))
def test_bucketed_classifier_forward_backward(shape, mix, pairwise, clip):
    B, l1, l2, l3, c, K = shape
    gen = torch.Generator().manual_seed(B + K)
    x = torch.randn(B, l1, generator=gen)
    w = stacked(K, l1, l2, l3, c, gen)
    bucket = draw_buckets(mix, B, K, gen)
    d_logits = torch.randn(B, c, generator=gen)
    xin = orc.pairwise(x.double()) if pairwise else x.double()
    w64 = [t.double() for t in w]
    oclip = clip if clip > 0 else None
    ref = orc.classifier_forward_bucketed(xin, bucket, *w64, oclip)
    d_l0, g_ref = orc.classifier_backward_bucketed(xin, bucket, *w64, d_logits.double(), oclip)
    d_x_ref = orc.pairwise_backward(x.double(), d_l0) if pairwise else d_l0
    plan = lib.bucket_group(bucket.to(DEV, torch.int32), 0, K)
    wd = [t.to(DEV) for t in w]
    h1, h2, logits = lib.classifier_forward(x.to(DEV), pairwise, *wd, clip, buckets=plan)
    assert_close_logits(logits, ref, "bucketed logits")
    d_x, g = lib.classifier_backward(x.to(DEV), pairwise, wd[0], wd[2], wd[4], h1, h2, d_logits.to(DEV), clip, buckets=plan)
    assert_close_grad(d_x, d_x_ref, "d_x")
    for name, got, want in zip(("d_w1", "d_b1", "d_w2", "d_b2", "d_w3", "d_b3"), g, g_ref):
        assert got.shape == want.shape
        assert_close_grad(got, want, name)
        for k in range(K):  # a stack no sample selected gets exactly zero
            if not bool((bucket == k).any()):
                assert not bool(got[k].any()), (name, k)


@pytest.mark.parametrize("shape", ((96, 256, 32, 16, 10), (64, 1024, 128, 32, 1000), (21, 24, 7, 5, 3)))
def test_one_bucket_through_the_bucketed_entry_points_is_the_reference_path(shape):
    """K = 1: nnue_classifier_*_bucketed are the plain entry points, bit for bit."""
    B, l1, l2, l3, c = shape
    gen = torch.Generator().manual_seed(B)
    x = torch.randn(B, l1, generator=gen).to(DEV)
    w = [t.to(DEV) for t in stacked(1, l1, l2, l3, c, gen)]
    flat = [t[0].contiguous() for t in w]
    labels = torch.randint(0, c, (B,), generator=gen).to(DEV)
    plan = lib.bucket_group(torch.zeros(B, dtype=torch.int32, device=DEV), 0, 1)
    a = lib.classifier_train_step(x, True, *flat, labels, phases=3)
    b = lib.classifier_train_step(x, True, *w, labels, phases=3, buckets=plan)
    for ta, tb in zip(a[0] + a[1] + (a[2],) + tuple(a[3]), b[0] + b[1] + (b[2],) + tuple(b[3])):
        assert torch.equal(ta.reshape(-1), tb.reshape(-1))


@pytest.mark.parametrize("shape", ((96, 256, 32, 16, 10, 4), (256, 1024, 128, 32, 100, 8), (37, 24, 7, 5, 3, 3)))
@pytest.mark.parametrize("mix", BUCKET_MIXES)
def test_bucketed_train_step_equals_separate_calls_and_the_oracle(shape, mix):
    B, l1, l2, l3, c, K = shape
    gen = torch.Generator().manual_seed(B * 3 + K)
    x = torch.randn(B, l1, generator=gen)
    w = stacked(K, l1, l2, l3, c, gen)
    bucket = draw_buckets(mix, B, K, gen)
    labels = torch.randint(0, c, (B,), generator=gen)
    plan = lib.bucket_group(bucket.to(DEV, torch.int32), 0, K)
    wd = [t.to(DEV) for t in w]
    acts, (sample_loss, loss), d_x, grads = lib.classifier_train_step(x.to(DEV), True, *wd, labels.to(DEV), clip=1.0, buckets=plan)
    # oracle: the same chain in float64
    l0 = orc.pairwise(x.double())
    w64 = [t.double() for t in w]
    logits = orc.classifier_forward_bucketed(l0, bucket, *w64, 1.0)
    ref_loss, d_logits = orc.cross_entropy_backward(logits, labels)
    d_l0, g_ref = orc.classifier_backward_bucketed(l0, bucket, *w64, d_logits, 1.0)
    assert_close_logits(acts[2], logits, "logits")
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss)))
    assert_close_grad(d_x, orc.pairwise_backward(x.double(), d_l0), "d_x")
    for name, got, want in zip(("d_w1", "d_b1", "d_w2", "d_b2", "d_w3", "d_b3"), grads, g_ref):
        assert_close_grad(got, want, name)
    # phase splits (1 then 2; 5 then 6) are bitwise the one-call result
    for first, second in ((1, 2), (5, 6)):
        scratch = torch.empty((lib.classifier_train_scratch_bytes(B, l1, l2, l3, c, K),), dtype=torch.uint8, device=DEV)
        r1 = lib.classifier_train_step(x.to(DEV), True, *wd, labels.to(DEV), clip=1.0, buckets=plan, phases=first, scratch=scratch)
        r2 = lib.classifier_train_step(x.to(DEV), True, *wd, labels.to(DEV), clip=1.0, buckets=plan, phases=second, scratch=scratch,
                                       out=r1[0], loss_out=r1[1], grads=r1[3], d_x=r1[2])
        for ta, tb in zip((d_x, loss) + tuple(grads), (r2[2], r2[1][1]) + tuple(r2[3])):
            assert torch.equal(ta, tb), (first, second)
    with pytest.raises(lib.NnueHipError):  # products that live in the FeatureTransformer launches know one stack only
        lib.classifier_train_step(x.to(DEV), True, *wd, labels.to(DEV), buckets=plan, phases=27)


def spread_images(B, hw, gen):
    """randn images with a per-sample offset and gain.  Together with all-positive conv weights (set by the caller) the
    offset moves every conv output of a sample the same way, so the active-feature counts -- and with them the buckets
    -- spread over the whole range instead of sitting at the 43 % randn gives."""
    shift = torch.linspace(-1.6, 1.6, B)[torch.randperm(B, generator=gen)].view(B, 1, 1, 1)
    gain = 0.5 + torch.rand(B, 1, 1, 1, generator=gen)
    return torch.randn(B, 3, hw, hw, generator=gen) * gain + shift


@pytest.mark.parametrize("cfg", (dict(grid=10, fps=8, l1=256, l2=32, l3=16, classes=10, image=32, K=8, batch=96),
                                 dict(grid=4, fps=8, l1=32, l2=8, l3=4, classes=5, image=32, K=3, batch=40),
                                 dict(grid=8, fps=4, l1=64, l2=32, l3=8, classes=7, image=96, K=5, batch=33)))
@pytest.mark.parametrize("clip", (None, 1.0))
def test_bucketed_model_forward_backward_against_the_oracle(cfg, clip):
    torch.manual_seed(cfg["K"])
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"],
                      input_size=cfg["image"], num_ls_buckets=cfg["K"], clip_activations=clip)
    with torch.no_grad():
        model.conv.weight.abs_()  # see spread_images
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    stride = orc.conv_stride(cfg["image"], cfg["grid"])
    gen = torch.Generator().manual_seed(9)
    images = spread_images(cfg["batch"], cfg["image"], gen)
    labels = torch.randint(0, cfg["classes"], (cfg["batch"],), generator=gen)
    p64 = {k: v.double() for k, v in params.items()}
    ref_logits, ref_loss, ref_grads, keep = orc.loss_and_grads_explicit(p64, images.double(), labels, stride, clip)
    assert len(set(keep["bucket"].tolist())) >= min(3, cfg["K"])  # the case exercises several stacks
    # the loop form (autograd, per-sample stacks) says the same as the closed form
    _, loop_loss, loop_grads, _ = orc.loss_and_grads_loop(p64, images.double(), labels, stride, clip)
    assert abs(float(loop_loss) - float(ref_loss)) < 1e-9
    for k in ref_grads:
        assert_close_grad(loop_grads[k], ref_grads[k], f"oracle loop vs explicit {k}", rtol=1e-9)
    model = model.to(DEV)
    logits = model(images.to(DEV))
    loss = F.cross_entropy(logits, labels.to(DEV))
    loss.backward()
    assert_close_logits(logits, ref_logits, "logits")
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
    for k, p in model.named_parameters():
        if k != "nnue2score":
            assert_close_grad(p.grad, ref_grads[k], k)
    # eval() / no_grad and the replayed evaluation plan see the same logits
    import evaluate
    with torch.no_grad():
        assert torch.equal(model(images.to(DEV)), logits.detach())
    ev_loss, _ = evaluate.evaluate_model(model, [(images, labels)], None, torch.device(DEV))
    assert abs(ev_loss - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
    # stand-alone classifier with explicit bucket ids (the selector on the counts of the map)
    n = keep["n"]
    bucket = nnue.bucket_of(n, cfg["K"], cfg["fps"] * keep["conv_out"].shape[2] * keep["conv_out"].shape[3])
    assert torch.equal(bucket, keep["bucket"])
    l0 = orc.pairwise(keep["ft"]).float().to(DEV)
    assert_close_logits(model.classifier(l0, bucket.to(DEV)), ref_logits, "stand-alone bucketed classifier")


@pytest.mark.parametrize("ride", ("1", "0"))
@pytest.mark.parametrize("use_graph", (False, True))
def test_bucketed_trainer_step_against_the_oracle(monkeypatch, ride, use_graph):
    """The fused trainer with K = 4 stacks: d_w1 from the per-bucket rider of the FeatureTransformer backward launch
    (grouped-row operands) and from the classifier's own product -- both against the oracle, two SGD steps."""
    from nnue_hip.trainer import NnueTrainer
    monkeypatch.setenv("NNUE_FTM_RIDE_DW1", ride)
    torch.manual_seed(3)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10, num_ls_buckets=4, clip_activations=1.0)
    with torch.no_grad():
        model.conv.weight.abs_()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = dict(lr=0.05, momentum=0.9, weight_decay=1e-4, max_grad_norm=1.0)
    tr = NnueTrainer(model, 80, (32, 32), use_graph=use_graph, **opt)
    assert tr.K == 4 and tr.ride_dw1 == (ride == "1") and not tr.fuse_l1
    gen = torch.Generator().manual_seed(12)
    bufs = {}
    for s in range(2):
        images = spread_images(80, 32, gen)
        labels = torch.randint(0, 10, (80,), generator=gen)
        p64 = {k: v.double() for k, v in params.items()}
        _, ref_loss, ref_grads, keep = orc.loss_and_grads_explicit(p64, images.double(), labels, 3, 1.0)
        assert len(set(keep["bucket"].tolist())) == 4
        loss = tr.step(images.to(DEV), labels.to(DEV))
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))
        got = tr.layout.views(tr.flat_grads)
        for k, ref in ref_grads.items():
            assert_close_grad(got[k], ref, f"step {s} {k}")
        g32 = {k: v.float() for k, v in ref_grads.items()}
        orc.sgd_step(params, g32, bufs, opt["lr"], opt["momentum"], opt["weight_decay"], opt["max_grad_norm"])
        for k in orc.TRAINABLE_KEYS:
            assert_close_grad(tr.p[k], params[k], f"step {s} parameter {k}", rtol=2e-5)


def test_bucketed_trainer_with_the_fused_table_update_and_a_short_batch(monkeypatch):
    """Combinations: K = 4 stacks + the table update inside the weight-gradient product (Gram norm) + a short last batch
    (padded rows land in bucket 0 with label -1), against the oracle on the real samples."""
    from nnue_hip.trainer import NnueTrainer
    monkeypatch.setenv("NNUE_FUSE_TABLE_UPDATE", "1")
    torch.manual_seed(6)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10, num_ls_buckets=4, clip_activations=1.0)
    with torch.no_grad():
        model.conv.weight.abs_()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV)
    opt = dict(lr=0.05, momentum=0.9, weight_decay=1e-4, max_grad_norm=1.0)
    tr = NnueTrainer(model, 64, (32, 32), use_graph=True, **opt)
    assert tr.fuse_table_update and tr.K == 4 and not tr.ride_dw1
    gen = torch.Generator().manual_seed(21)
    bufs = {}
    for s, n in enumerate((64, 64, 41)):
        images = spread_images(n, 32, gen)
        labels = torch.randint(0, 10, (n,), generator=gen)
        p64 = {k: v.double() for k, v in params.items()}
        _, ref_loss, ref_grads, _ = orc.loss_and_grads_explicit(p64, images.double(), labels, 3, 1.0)
        loss = tr.step(images.to(DEV), labels.to(DEV))
        assert abs(float(loss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss))), (s, float(loss), float(ref_loss))
        ref_norm = orc.sgd_step(params, {k: v.float() for k, v in ref_grads.items()}, bufs, opt["lr"], opt["momentum"], opt["weight_decay"],
                                opt["max_grad_norm"])
        assert abs(float(tr.grad_norm) - float(ref_norm)) <= 1e-4 * float(ref_norm)
        for k in orc.TRAINABLE_KEYS:
            assert_close_grad(tr.p[k], params[k], f"step {s} parameter {k}", rtol=2e-5)


@pytest.mark.parametrize("shape", ((512, 8, 11, 11, 800, 1024, 8), (100, 64, 32, 32, 65536, 256, 4), (37, 4, 8, 8, 256, 64, 3)))
def test_grouping_riding_in_the_forward_launch(shape):
    """nnue_ftm_forward_grouping: the same ft as nnue_ftm_forward (bitwise) and the same plan as nnue_bucket_group."""
    b, fps, gh, gw, f, l1, K = shape
    gen = torch.Generator().manual_seed(b + K)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen) + torch.linspace(-1.5, 1.5, b).view(b, 1, 1, 1)
    thr = torch.full((fps,), 0.1)
    weight, bias = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV), torch.randn(l1, generator=gen).to(DEV)
    fm = lib.ftm_binarize(conv_out.to(DEV), thr.to(DEV), f, l1)
    want = lib.bucket_group(fm.n, fps * gh * gw, K)
    plan = lib.BucketPlan(b, K, DEV)
    out = lib.ftm_forward(weight, bias, fm, group=plan)
    assert torch.equal(out, lib.ftm_forward(weight, bias, fm))
    for name in ("bucket", "rows", "tile_bucket", "seg"):
        assert torch.equal(getattr(plan, name), getattr(want, name)), name
    assert len(set(plan.bucket.tolist())) >= 2
