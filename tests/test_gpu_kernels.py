"""Parity of every HIP kernel (called through the C ABI) with the CPU oracle and the reference's golden
vectors.  Needs an MI355X: run with ``-m gpu``.

Bars: feature ids bit-exact; float results within 1e-4*max(1,|ref|) (activations) or 1e-4*max|ref| per
tensor (gradients) -- conftest.assert_close_logits / assert_close_grad.
"""
import json

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import nnue_oracle as orc
from conftest import MODEL_CASES, assert_close_grad, assert_close_logits, golden_model, load_npz

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from nnue_hip import lib
    lib.load()
    return lib


def g(t):
    return t.to(DEV)


# ------------------------------------------------------------------------------------ FeatureTransformer
def ft_all(hip, w, b, idx, val, up):
    act = hip.ft_prepare(g(idx), g(val), w.shape[0])
    out = hip.ft_forward(g(w), g(b), act)
    d_w, d_b = hip.ft_backward_weight(g(up), act, w.shape[0])
    d_val = hip.ft_backward_values(g(up), g(w), act, idx.shape[1])
    return out.cpu(), d_w.cpu(), d_b.cpu(), d_val.cpu(), act


def test_ft_reference_cases(hip):
    """The hand-built stand-alone cases (all -1, M=1, repeats, unsorted, ids >= F, values, holes)."""
    z = load_npz("ft_cases.npz")
    w, b = torch.from_numpy(z["weight"]), torch.from_numpy(z["bias"])
    for name in sorted({k.split("/")[0] for k in z if "/" in k}):
        idx, val, up = (torch.from_numpy(z[f"{name}/{k}"]) for k in ("idx", "val", "upstream"))
        out, d_w, d_b, d_val, act = ft_all(hip, w, b, idx, val, up)
        assert_close_logits(out, torch.from_numpy(z[f"{name}/out"]), f"{name} out")
        assert_close_grad(d_w, torch.from_numpy(z[f"{name}/d_weight"]), f"{name} d_weight")
        assert_close_grad(d_b, torch.from_numpy(z[f"{name}/d_bias"]), f"{name} d_bias")
        ref = torch.from_numpy(z[f"{name}/d_val"])
        assert float((d_val - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), name
        # compaction bookkeeping is exact
        assert torch.equal(act.n.cpu().long(), (idx >= 0).sum(1))


@pytest.mark.parametrize("rows,l1,b,m", [(40, 24, 5, 17), (800, 64, 32, 426), (129, 256, 7, 300), (800, 1024, 64, 460),
                                         (1000, 512, 3, 1), (64, 2048, 9, 130), (50, 6, 4, 9)])
def test_ft_random_against_oracle(hip, rows, l1, b, m):
    gen = torch.Generator().manual_seed(rows * 7 + l1)
    w = torch.randn(rows, l1, generator=gen) * 0.1
    bias = torch.randn(l1, generator=gen) * 0.1
    idx = torch.randint(-rows // 4, rows + rows // 8, (b, m), generator=gen)  # ~20% padding, some ids >= F
    idx = torch.where(idx < 0, torch.full_like(idx, -1), idx)
    val = torch.randn(b, m, generator=gen)
    up = torch.randn(b, l1, generator=gen)
    out, d_w, d_b, d_val, act = ft_all(hip, w, bias, idx, val, up)
    ref_out = orc.ft_forward(w.double(), bias.double(), idx, val.double())
    r_w, r_b, r_val = orc.ft_backward(w.double(), idx, val.double(), up.double())
    assert_close_logits(out, ref_out, "out")
    assert_close_grad(d_w, r_w, "d_weight")
    assert_close_grad(d_b, r_b, "d_bias")
    assert_close_grad(d_val, r_val, "d_val")
    # coefT is the transposed coefficient matrix
    c = orc.coefficient_matrix(idx, val.double(), rows)
    assert_close_grad(act.coefT[:, :b].cpu().t(), c, "coefT")
    assert bool((act.coefT[:, b:] == 0).all())


def test_ft_edge_shapes(hip):
    w = torch.randn(10, 256)
    bias = torch.randn(256)
    # every entry padding: output is the bias, gradients are zero
    idx = torch.full((3, 4), -1)
    out, d_w, d_b, d_val, act = ft_all(hip, w, bias, idx, torch.ones(3, 4), torch.ones(3, 256))
    assert torch.equal(out, bias.expand(3, -1)) and not d_w.any() and not d_val.any()
    assert torch.equal(d_b, torch.full((256,), 3.0)) and not act.n.any()
    # one sample, one feature, huge id -> clamps to the last row
    out, d_w, _, d_val, _ = ft_all(hip, w, bias, torch.tensor([[2 ** 40]]), torch.tensor([[2.0]]), torch.ones(1, 256))
    assert_close_logits(out, (bias + 2 * w[9]).unsqueeze(0), "clamped")
    assert torch.equal(d_w[9], torch.full((256,), 2.0)) and not d_w[:9].any()
    assert abs(float(d_val) - float(w[9].sum())) < 1e-3


def test_ft_properties_at_full_size(hip):
    """C2-sized table and batch: linearity in the values, order invariance, duplicates == doubled value."""
    gen = torch.Generator().manual_seed(5)
    rows, l1, b, m = 800, 1024, 512, 460
    w, bias = g(torch.randn(rows, l1, generator=gen) * 0.1), g(torch.zeros(l1))
    idx = torch.stack([torch.randperm(968, generator=gen)[:m] for _ in range(b)])
    idx[:, 400:] = torch.where(torch.rand(b, m - 400, generator=gen) < 0.5, -1, idx[:, 400:])
    v1, v2 = torch.randn(b, m, generator=gen), torch.randn(b, m, generator=gen)
    f = lambda i, v: hip.ft_forward(w, bias, hip.ft_prepare(g(i), g(v), rows))  # noqa: E731
    y1, y2, y12 = f(idx, v1), f(idx, v2), f(idx, 2 * v1 - 3 * v2)
    assert_close_logits(y12, 2 * y1 - 3 * y2, "linearity", rtol=2e-4)
    perm = torch.randperm(m, generator=gen)
    assert_close_logits(f(idx[:, perm], v1[:, perm]), y1, "order invariance")
    assert_close_logits(f(torch.cat([idx, idx], 1), torch.cat([v1, v1], 1)), f(idx, 2 * v1), "duplicates accumulate")
    # adjoint identity <dOut, FT(val)> == <dval, val> + <dW, W>-free part: check dval against finite linearity
    up = g(torch.randn(b, l1, generator=gen))
    act = hip.ft_prepare(g(idx), g(v1), rows)
    d_val = hip.ft_backward_values(up, w, act, m)
    lhs = float((up.double() * (y1.double())).sum())
    rhs = float((d_val.double() * g(torch.where(idx >= 0, v1, torch.zeros_like(v1))).double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    # bitwise reproducible: the accumulation kernels use no atomics.  (ft_prepare itself adds repeated ids of
    # one sample with float atomics, so only inputs without 3+ repeats of an id are bitwise stable: use ids < F)
    uniq = torch.stack([torch.randperm(rows, generator=gen)[:m] for _ in range(b)])
    act_u = hip.ft_prepare(g(uniq), g(v1), rows)
    d_w1, _ = hip.ft_backward_weight(up, act_u, rows)
    d_w2, _ = hip.ft_backward_weight(up, hip.ft_prepare(g(uniq), g(v1), rows), rows)
    assert torch.equal(d_w1, d_w2) and torch.equal(f(idx, v1), y1)
    assert torch.equal(hip.ft_backward_values(up, w, act_u, m), hip.ft_backward_values(up, w, act_u, m))


# ------------------------------------------------------------------------------------ front end
@pytest.mark.parametrize("name", MODEL_CASES)
def test_conv_and_ids_golden(hip, name):
    cfg, params, _, data = golden_model(name)
    conv = hip.conv3x3_forward(g(data["images"]), g(params["conv.weight"]), cfg["stride"])
    assert conv.shape == data["conv_out"].shape
    assert_close_logits(conv, data["conv_out"], "conv_out", rtol=1e-5)
    f = cfg["grid"] ** 2 * cfg["fps"]
    for source in (g(data["conv_out"]), conv):  # ids from the reference's conv_out AND from ours
        act = hip.binarize_features(source, g(params["visual_threshold"]), f)
        n = act.n.cpu().long()
        ref_idx = data["idx"]
        assert torch.equal(n, (ref_idx >= 0).sum(1))  # bit-exact ids
        m = ref_idx.shape[1]
        idx, val = hip.act_to_padded(act, m)
        assert torch.equal(idx.cpu(), ref_idx) and torch.equal(val.cpu(), data["val"])
        rows = act.rows.cpu().long()
        for b in range(n.numel()):
            assert torch.equal(rows[b, : n[b]], ref_idx[b, : n[b]].clamp(max=f - 1))
        c = orc.coefficient_matrix(ref_idx, data["val"], f)
        assert torch.equal(act.coefT[:, : n.numel()].cpu().t(), c)  # small integers: exact


def test_binarize_random_geometry(hip):
    gen = torch.Generator().manual_seed(9)
    for b, fps, gh, gw, f in [(3, 8, 11, 11, 800), (70, 3, 5, 7, 200), (2, 64, 32, 32, 65536), (5, 2, 3, 3, 4), (130, 1, 1, 1, 1)]:
        x = torch.randn(b, fps, gh, gw, generator=gen)
        thr = torch.randn(fps, generator=gen) * 0.3
        act = hip.binarize_features(g(x), g(thr), f)
        idx, n = orc.active_lists(x, thr)
        assert torch.equal(act.n.cpu().long(), n)
        p = fps * gh * gw
        pos = act.pos.cpu().long()
        for s in range(b):
            assert torch.equal(pos[s, : n[s]], idx[s, : n[s]])
        c = orc.coefficient_matrix(idx, (idx >= 0).float(), f)
        assert torch.equal(act.coefT[:, :b].cpu().t(), c), (b, fps, gh, gw, f)
        assert p == act.cap


@pytest.mark.parametrize("name", MODEL_CASES)
def test_ste_conv_backward(hip, name):
    cfg, params, _, data = golden_model(name)
    gen = torch.Generator().manual_seed(3)
    x = data["conv_out"]
    d = torch.randn(x.shape, generator=gen) * (torch.rand(x.shape, generator=gen) < 0.45)
    thr = params["visual_threshold"]
    d_thr, d_w = hip.ste_conv_backward(g(data["images"]), g(x), g(thr), g(d), cfg["stride"])
    ref_thr = -(d.double() * orc.ste_slope(x.double(), thr.double().view(1, -1, 1, 1))).sum(dim=(0, 2, 3))
    ref_w = orc.conv_weight_grad(data["images"].double(), d.double(), cfg["stride"], params["conv.weight"].shape)
    assert_close_grad(d_thr, ref_thr, "d_thr")
    assert_close_grad(d_w, ref_w, "d_conv_weight")


STE_SHAPES = [  # B, H, W, fps, stride: every channel-tile count of the MFMA kernel, the generic kernel (fps > 64),
    # non-square maps, tiles crossing sample boundaries, a single position, C2's and C4's shapes
    (1, 3, 3, 1, 3), (2, 32, 32, 3, 3), (5, 17, 23, 8, 2), (64, 32, 32, 8, 3), (3, 40, 40, 17, 4), (4, 33, 31, 33, 5),
    (2, 64, 64, 50, 7), (3, 224, 224, 64, 7), (2, 20, 20, 70, 2), (7, 96, 96, 16, 1),
]


@pytest.mark.parametrize("shape", STE_SHAPES)
@pytest.mark.parametrize("density", (0.03, 0.45, 1.0))
def test_ste_conv_backward_shapes(hip, shape, density):
    b, h, w, fps, stride = shape
    gen = torch.Generator().manual_seed(b * 1000 + fps)
    images = torch.randn(b, 3, h, w, generator=gen)
    weight = torch.randn(fps, 3, 3, 3, generator=gen) * 0.2
    x = F.conv2d(images, weight, stride=stride, padding=1)
    thr = torch.randn(fps, generator=gen) * 0.1
    d = torch.randn(x.shape, generator=gen) * (torch.rand(x.shape, generator=gen) < density)
    d_thr, d_w = hip.ste_conv_backward(g(images), g(x), g(thr), g(d), stride)
    ref_thr = -(d.double() * orc.ste_slope(x.double(), thr.double().view(1, -1, 1, 1))).sum(dim=(0, 2, 3))
    ref_w = orc.conv_weight_grad(images.double(), d.double(), stride, weight.shape)
    assert_close_grad(d_thr, ref_thr, "d_thr", rtol=2e-5)
    assert_close_grad(d_w, ref_w, "d_conv_weight", rtol=2e-5)
    again = hip.ste_conv_backward(g(images), g(x), g(thr), g(d), stride)
    assert torch.equal(again[0], d_thr) and torch.equal(again[1], d_w)  # fixed summation order


# ------------------------------------------------------------------------------------ classifier
CLS_SHAPES = [  # B, L1, L2, L3, C
    (3, 32, 4, 4, 10), (5, 24, 7, 5, 3), (4, 64, 32, 8, 10), (37, 64, 16, 8, 10), (16, 128, 32, 8, 100),
    (512, 1024, 128, 32, 10), (100, 1024, 128, 32, 1000), (33, 256, 48, 16, 7),
]


@pytest.mark.parametrize("shape", CLS_SHAPES)
@pytest.mark.parametrize("pairwise", (True, False))
@pytest.mark.parametrize("clip", (0.0, 0.75))
def test_classifier_forward_backward(hip, shape, pairwise, clip):
    b, l1, l2, l3, c = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(b, l1, generator=gen)
    mk = lambda *s: torch.randn(*s, generator=gen) / (s[-1] ** 0.5)  # noqa: E731
    p = [mk(l2, l1), mk(l2) * 0.1, mk(l3, l2), mk(l3) * 0.1, mk(c, l3), mk(c) * 0.1]
    up = torch.randn(b, c, generator=gen)
    h1, h2, logits = hip.classifier_forward(g(x), pairwise, *[g(t) for t in p], clip)
    d_x, grads = hip.classifier_backward(g(x), pairwise, g(p[0]), g(p[2]), g(p[4]), h1, h2, g(up), clip)
    xd = x.double()
    pd = [t.double() for t in p]
    l0 = orc.pairwise(xd) if pairwise else xd
    cl = clip if clip > 0 else None
    ref_logits = orc.classifier_forward(l0, *pd, clip=cl)
    assert_close_logits(logits, ref_logits, "logits")
    # gate decisions can flip when a pre-activation sits within rounding of 0/clip: compare the gradients with
    # the gates taken from OUR activations (the backward kernel's contract), via a float64 replay
    z1 = torch.nn.functional.linear(l0, pd[0], pd[1])
    z2 = torch.nn.functional.linear(h1.cpu().double(), pd[2], pd[3])
    assert_close_logits(h1, z1.clamp(0, cl) if cl else z1.relu(), "h1")
    assert_close_logits(h2, z2.clamp(0, cl) if cl else z2.relu(), "h2")
    gate = lambda h: ((h > 0) & ((h < clip) if clip > 0 else torch.ones_like(h, dtype=torch.bool))).double()  # noqa: E731
    h1d, h2d, upd = h1.cpu().double(), h2.cpu().double(), up.double()
    d_z2 = (upd @ pd[4]) * gate(h2d)
    d_z1 = (d_z2 @ pd[2]) * gate(h1d)
    ref = [d_z1.t() @ l0, d_z1.sum(0), d_z2.t() @ h1d, d_z2.sum(0), upd.t() @ h2d, upd.sum(0)]
    for got, want, nm in zip(grads, ref, ("d_w1", "d_b1", "d_w2", "d_b2", "d_w3", "d_b3")):
        assert_close_grad(got, want, nm)
    d_l0 = d_z1 @ pd[0]
    assert_close_grad(d_x, orc.pairwise_backward(xd, d_l0) if pairwise else d_l0, "d_x")


# ------------------------------------------------------------------------------------ loss + step tail
@pytest.mark.parametrize("b,c", [(3, 10), (512, 10), (65, 100), (7, 1000), (1, 1)])
def test_cross_entropy(hip, b, c):
    gen = torch.Generator().manual_seed(b + c)
    logits = torch.randn(b, c, generator=gen) * 3
    labels = torch.randint(0, c, (b,), generator=gen)
    sample, loss, d = hip.cross_entropy(g(logits), g(labels), 1.0)
    ref_loss, ref_d = orc.cross_entropy_backward(logits.double(), labels)
    assert abs(float(loss) - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss)))
    assert_close_grad(d, ref_d, "d_logits")
    assert_close_logits(sample, torch.nn.functional.cross_entropy(logits.double(), labels, reduction="none"), "sample loss", rtol=1e-5)
    _, _, d2 = hip.cross_entropy(g(logits), g(labels), 0.5)
    assert_close_grad(d2, ref_d * 0.5, "scaled d_logits")


@pytest.mark.parametrize("count", (10, 4097, 956106))
def test_sgd_step(hip, count):
    gen = torch.Generator().manual_seed(count)
    p0 = torch.randn(count, generator=gen)
    params = {"visual_threshold": p0.clone()}
    bufs = {}
    dp, dm = g(p0.clone()), g(torch.zeros(count))
    scratch = torch.empty(hip.sgd_scratch_bytes(count), dtype=torch.uint8, device=DEV)
    norm = torch.zeros((), device=DEV)
    for step in range(3):
        grad = torch.randn(count, generator=gen) * (5.0 if step == 1 else 0.01 / count ** 0.5)  # clipped and unclipped steps
        ref_norm = orc.sgd_step(params, {"visual_threshold": grad / 2}, bufs, 0.01, 0.9, 2e-4, 1.0)
        hip.sgd_step(dp, g(grad), dm, 0.01, 0.9, 2e-4, 1.0, 0.5, step == 0, norm, scratch)
        assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
        assert_close_grad(dp, params["visual_threshold"], f"params step {step}", rtol=1e-5)
        assert_close_grad(dm, bufs["visual_threshold"], f"momentum step {step}", rtol=1e-5)
    # no momentum, no clipping, no buffer
    q = g(p0.clone())
    hip.sgd_step(q, g(torch.ones(count)), None, 0.1, 0.0, 0.0, 0.0, 1.0, True, None, scratch)
    assert_close_grad(q, p0 - 0.1, "plain sgd", rtol=1e-6)


@pytest.mark.parametrize("shape", [(3, 32, 4, 4, 10), (5, 24, 7, 5, 3), (512, 1024, 128, 32, 10), (100, 1024, 128, 32, 1000), (33, 256, 48, 16, 7)])
@pytest.mark.parametrize("clip", (0.0, 0.75))
def test_classifier_train_step_equals_separate_calls(hip, shape, clip):
    """The fused training entry point (narrow layers + loss + their backward in one kernel) against
    nnue_classifier_forward -> nnue_cross_entropy -> nnue_classifier_backward, and against the oracle."""
    b, l1, l2, l3, c = shape
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    x = g(torch.randn(b, l1, generator=gen))
    mk = lambda *s: torch.randn(*s, generator=gen) / (s[-1] ** 0.5)  # noqa: E731
    p = [g(t) for t in (mk(l2, l1), mk(l2) * 0.1, mk(l3, l2), mk(l3) * 0.1, mk(c, l3), mk(c) * 0.1)]
    labels = g(torch.randint(0, c, (b,), generator=gen))
    h1, h2, logits = hip.classifier_forward(x, True, *p, clip)
    sample, loss, d_logits = hip.cross_entropy(logits, labels, 0.5)
    d_x, grads = hip.classifier_backward(x, True, p[0], p[2], p[4], h1, h2, d_logits, clip)
    (f1, f2, flog), (fsample, floss), fd_x, fgrads = hip.classifier_train_step(x, True, *p, labels, 0.5, clip)
    assert_close_logits(flog, logits, "logits", rtol=1e-5)
    assert_close_logits(f1, h1, "h1", rtol=1e-5)
    assert_close_logits(f2, h2, "h2", rtol=1e-5)
    assert_close_logits(fsample, sample, "sample loss", rtol=1e-5)
    assert abs(float(floss) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    assert_close_grad(fd_x, d_x, "d_x", rtol=2e-5)
    for a, r, nm in zip(fgrads, grads, ("d_w1", "d_b1", "d_w2", "d_b2", "d_w3", "d_b3")):
        assert_close_grad(a, r, nm, rtol=2e-5)
    ref_loss, _ = orc.cross_entropy_backward(orc.classifier_forward(orc.pairwise(x.cpu().double()), *[t.cpu().double() for t in p],
                                                                    clip=clip if clip > 0 else None), labels.cpu())
    assert abs(float(floss) - float(ref_loss)) <= 1e-4 * max(1.0, abs(float(ref_loss)))


@pytest.mark.parametrize("shape", [(512, 1024, 128, 32, 10), (100, 1024, 128, 32, 1000), (33, 256, 48, 16, 7), (5, 24, 7, 5, 3)])
def test_classifier_train_step_phase_splits_are_identical(hip, shape):
    """phases 3 (one call), 1 then 2, 5 then 6 (first-layer weight product inside the d_x launch) and 7: same bits."""
    b, l1, l2, l3, c = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x = g(torch.randn(b, l1, generator=gen))
    mk = lambda *s: torch.randn(*s, generator=gen) / (s[-1] ** 0.5)  # noqa: E731
    p = [g(t) for t in (mk(l2, l1), mk(l2) * 0.1, mk(l3, l2), mk(l3) * 0.1, mk(c, l3), mk(c) * 0.1)]
    labels = g(torch.randint(0, c, (b,), generator=gen))

    def run(sequence):
        scratch = torch.empty((hip.classifier_train_scratch_bytes(b, l1, l2, l3, c),), dtype=torch.uint8, device=DEV)
        keep = None
        for ph in sequence:
            keep = hip.classifier_train_step(x, True, *p, labels, 0.5, 0.0, scratch=scratch, phases=ph,
                                             **({} if keep is None else dict(out=keep[0], loss_out=keep[1], d_x=keep[2], grads=keep[3])))
        return keep

    ref = run((3,))
    for seq in ((1, 2), (5, 6), (7,)):
        got = run(seq)
        assert torch.equal(got[2], ref[2]) and torch.equal(got[1][1], ref[1][1]) and torch.equal(got[0][2], ref[0][2]), seq
        for a, r in zip(got[3], ref[3]):
            assert torch.equal(a, r), seq
    with pytest.raises(hip.NnueHipError):
        hip.classifier_train_step(x, True, *p, labels, phases=4)


@pytest.mark.parametrize("shape", [(512, 1024, 128, 32, 10), (33, 256, 48, 16, 7)])
def test_classifier_train_step_leaves_dw1_to_the_rider(hip, shape):
    """phases bit 16: no first-layer weight product here (d_w1 untouched); d_z1 sits at the published scratch offset,
    and d_z1^T l0 formed from it is the d_w1 of the plain call; everything else keeps its bits."""
    b, l1, l2, l3, c = shape
    gen = torch.Generator().manual_seed(sum(shape) + 1)
    x = g(torch.randn(b, l1, generator=gen))
    mk = lambda *s: torch.randn(*s, generator=gen) / (s[-1] ** 0.5)  # noqa: E731
    p = [g(t) for t in (mk(l2, l1), mk(l2) * 0.1, mk(l3, l2), mk(l3) * 0.1, mk(c, l3), mk(c) * 0.1)]
    labels = g(torch.randint(0, c, (b,), generator=gen))
    nbytes = hip.classifier_train_scratch_bytes(b, l1, l2, l3, c)
    ref = hip.classifier_train_step(x, True, *p, labels, 0.5, 0.0, phases=3,
                                    scratch=torch.empty((nbytes,), dtype=torch.uint8, device=DEV))
    scratch = torch.empty((nbytes,), dtype=torch.uint8, device=DEV)
    grads = [torch.full_like(t, 7.0) for t in ref[3]]
    got = hip.classifier_train_step(x, True, *p, labels, 0.5, 0.0, scratch=scratch, phases=17, grads=grads)
    got = hip.classifier_train_step(x, True, *p, labels, 0.5, 0.0, scratch=scratch, phases=18, out=got[0], loss_out=got[1],
                                    d_x=got[2], grads=got[3])
    assert torch.equal(got[2], ref[2]) and torch.equal(got[1][1], ref[1][1]) and torch.equal(got[0][2], ref[0][2])
    assert torch.equal(got[3][0], torch.full_like(ref[3][0], 7.0))
    for a, r in zip(got[3][1:], ref[3][1:]):
        assert torch.equal(a, r)
    once = hip.classifier_train_step(x, True, *p, labels, 0.5, 0.0, phases=19, grads=[torch.full_like(t, 7.0) for t in ref[3]],
                                     scratch=torch.empty((nbytes,), dtype=torch.uint8, device=DEV))
    assert torch.equal(once[2], ref[2]) and torch.equal(once[1][1], ref[1][1]) and torch.equal(once[0][2], ref[0][2])
    for a, r in zip(once[3], got[3]):
        assert torch.equal(a, r)
    off = hip.classifier_train_dz1_offset(b, l1, l2, l3, c, True)
    assert off >= 0 and off % 16 == 0 and off + b * l2 * 4 <= nbytes
    d_z1 = scratch[off:off + b * l2 * 4].view(torch.float32).view(b, l2).double().cpu()
    x64 = x.double().cpu()
    l0 = torch.cat([x64[:, :l1 // 2] * x64[:, l1 // 2:], x64[:, :l1 // 2]], dim=1)
    assert_close_grad(ref[3][0].cpu(), d_z1.t() @ l0, "d_w1 from the published d_z1", rtol=2e-5)
    with pytest.raises(hip.NnueHipError):
        hip.classifier_train_step(x, True, *p, labels, phases=23)  # 4 and 16 exclude each other


@pytest.mark.parametrize("shape", [(64, 32, 32, 8, 3), (7, 40, 40, 3, 4), (16, 64, 64, 20, 7)])
def test_deferred_ste_sums_ride_in_the_norm_launch(hip, shape):
    """nnue_ste_conv_backward(stages=1) + nnue_sgd_step(ste=...) against the plain two calls: d_thr / d_weight bitwise
    (same sums, same order), norm and updated parameters within float rounding (other partition of the partials)."""
    b, h, w, fps, stride = shape
    gen = torch.Generator().manual_seed(b + fps)
    images = g(torch.randn(b, 3, h, w, generator=gen))
    gh, gw = hip.conv_out_hw(h, w, stride)
    conv_out = g(torch.randn(b, fps, gh, gw, generator=gen))
    thr = g(torch.randn(fps, generator=gen) * 0.1)
    d_conv = g(torch.randn(b, fps, gh, gw, generator=gen) / b)
    n_thr, n_w = (fps + 3) // 4 * 4, (fps * 27 + 3) // 4 * 4
    count = n_thr + n_w + 4096 + 4
    base_g = torch.zeros(count)
    base_g[n_thr + n_w:] = torch.randn(4096 + 4, generator=gen)
    params0 = g(torch.randn(count, generator=gen))

    def run(defer):
        grads, params, mom = g(base_g.clone()), params0.clone(), torch.zeros(count, device=DEV)
        d_thr, d_w = grads[:fps], grads[n_thr:n_thr + fps * 27].view(fps, 3, 3, 3)
        scratch = torch.empty((max(16, hip.load().nnue_ste_conv_backward_scratch(b, fps, gh, gw)),), dtype=torch.uint8, device=DEV)
        norm = torch.zeros((), device=DEV)
        sgd_scratch = torch.empty((hip.sgd_scratch_bytes(count),), dtype=torch.uint8, device=DEV)
        hip.ste_conv_backward(images, conv_out, thr, d_conv, stride, d_thr=d_thr, d_weight=d_w, scratch=scratch, stages=1 if defer else 3)
        ste = (scratch, hip.ste_conv_backward_chunks(b, fps, gh, gw), fps, d_thr, d_w) if defer else None
        hip.sgd_step(params, grads, mom, 0.05, 0.9, 1e-4, 1.0, 0.5, True, norm, sgd_scratch, ste=ste)
        return grads, params, float(norm)

    g0, p0, n0 = run(False)
    g1, p1, n1 = run(True)
    assert torch.equal(g0, g1)
    assert abs(n0 - n1) <= 1e-5 * n0
    assert_close_grad(p1, p0, "updated parameters", rtol=1e-5)
    # stage 2 on its own finishes the same partials
    grads = g(base_g.clone())
    d_thr, d_w = grads[:fps], grads[n_thr:n_thr + fps * 27].view(fps, 3, 3, 3)
    scratch = torch.empty((max(16, hip.load().nnue_ste_conv_backward_scratch(b, fps, gh, gw)),), dtype=torch.uint8, device=DEV)
    hip.ste_conv_backward(images, conv_out, thr, d_conv, stride, d_thr=d_thr, d_weight=d_w, scratch=scratch, stages=1)
    hip.ste_conv_backward(images, conv_out, thr, d_conv, stride, d_thr=d_thr, d_weight=d_w, scratch=scratch, stages=2)
    assert torch.equal(grads[:n_thr + n_w], g0[:n_thr + n_w])
    with pytest.raises(hip.NnueHipError):  # outputs that are not the head of the gradient buffer
        hip.sgd_step(p0, g0, torch.zeros(count, device=DEV), 0.05, 0.9, 0.0, 1.0, 1.0, True, None,
                     torch.empty((hip.sgd_scratch_bytes(count),), dtype=torch.uint8, device=DEV),
                     ste=(scratch, 4, fps, g0[8:8 + fps], g0[n_thr:n_thr + fps * 27]))
