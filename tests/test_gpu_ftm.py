"""FeatureTransformer for the binary map as dense f32-MFMA products (nnue_ftm_*): parity with the float64
restatement of the reference formula (nnue.py:686-710 and its autograd), with the gather kernels, and through the
golden model fixtures.  ``-m gpu``."""
import pytest
import torch
import torch.nn.functional as F

import nnue_oracle as orc
from conftest import MODEL_CASES, assert_close_grad, assert_close_logits, golden_model

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def hip():
    from nnue_hip import lib
    lib.load()
    return lib


def dense_reference(conv_out, thr, weight, bias, d_out):
    """float64: A (membership incl. the clamp sink), out, d_weight, d_bias, d_conv_out."""
    b, fps, gh, gw = conv_out.shape
    p, (f, l1) = fps * gh * gw, weight.shape
    bits = (conv_out > thr.view(1, -1, 1, 1)).double().reshape(b, p)
    rows = torch.clamp(torch.arange(p), max=f - 1)
    a = torch.zeros(b, f, dtype=torch.float64)
    a.index_add_(1, rows, bits)  # positions >= F-1 pile up on row F-1
    w, d = weight.double(), d_out.double()
    out = a @ w + bias.double()
    d_w, d_b = a.t() @ d, d.sum(0)
    d_val = (d @ w.t())[:, rows] * bits
    n = bits.sum(1).to(torch.int32)
    sink = bits[:, f - 1:].sum(1).float() if p >= f else torch.zeros(b)
    return out, d_w, d_b, d_val.reshape(conv_out.shape), n, sink


SHAPES = [  # B, fps, Gh, Gw, F, L1: the CIFAR map (clamp sink), exact-fit map, table larger than the map, ragged everything
    (512, 8, 11, 11, 800, 1024), (64, 8, 11, 11, 800, 256), (37, 4, 8, 8, 256, 64), (5, 4, 3, 3, 100, 36),
    (16, 64, 8, 8, 4096, 128), (3, 2, 2, 1, 3, 4), (130, 8, 10, 10, 800, 200), (33, 12, 5, 5, 150, 72), (1, 4, 1, 1, 4, 8),
    (24, 64, 32, 32, 65536, 256), (140, 64, 24, 24, 30000, 136),  # big maps: bf16-split tiles incl. the six-product value gradient
    (130, 64, 24, 24, 30000, 128), (128, 64, 32, 32, 65536, 64),  # ... and the LDS-DMA value gradient (ftv_kernels.hip): row tail, clamp, two row tiles
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("density", (0.02, 0.45, 1.0))
def test_ftm_kernels_against_float64(hip, shape, density):
    b, fps, gh, gw, f, l1 = shape
    assert hip.ftm_supported(f, fps * gh * gw, l1)
    gen = torch.Generator().manual_seed(b * 7 + f)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen)
    thr = torch.quantile(conv_out.transpose(0, 1).flatten(1), 1.0 - density, dim=1) if density < 1.0 else torch.full((fps,), -1e9)
    weight, bias = torch.randn(f, l1, generator=gen) * 0.1, torch.randn(l1, generator=gen)
    d_out = torch.randn(b, l1, generator=gen) / b
    ref_out, ref_dw, ref_db, ref_dval, ref_n, ref_sink = dense_reference(conv_out, thr, weight, bias, d_out)
    g = lambda t: t.to(DEV)
    fm = hip.ftm_binarize(g(conv_out), g(thr), f, l1)
    active = (conv_out > thr.view(1, -1, 1, 1)).reshape(b, -1)
    assert torch.equal(fm.n.cpu(), ref_n) and torch.equal(fm.sink.cpu(), ref_sink) and torch.equal(fm.bits.cpu(), active.to(torch.uint8))
    out = hip.ftm_forward(g(weight), g(bias), fm)
    # element by element against the float64 value.  Sums of tens of thousands of float32 terms (the big maps) are held to
    # the north star's own bar, 1e-4 * max(1, |ref|); measured there (profiles/r03a_ft_err.json): worst element 0.79 of
    # that bar, and never worse than a float32 CPU product of the same operands (1.7e-4 vs 1.3e-4 absolute at |ref| = 880).
    # Small maps keep the 5x sharper bar they have always met.
    assert_close_logits(out, ref_out, "out", rtol=1e-4 if fps * gh * gw >= 16384 else 2e-5)
    d_w, d_b = hip.ftm_backward_weight(g(d_out), fm)
    assert_close_grad(d_w, ref_dw, "d_weight", rtol=2e-5)
    assert_close_grad(d_b, ref_db, "d_bias", rtol=2e-5)
    assert not bool(d_w[min(f - 1, fps * gh * gw):f - 1].any())  # rows the map cannot reach
    d_val = hip.ftm_backward_values(g(d_out), g(weight), fm)
    assert_close_grad(d_val.view(conv_out.shape), ref_dval, "d_conv_out", rtol=2e-5)
    assert not bool(d_val.view(b, -1)[~active.to(DEV)].any())  # exact zeros where inactive
    # fixed summation order: bitwise reproducible
    assert torch.equal(hip.ftm_forward(g(weight), g(bias), fm), out)
    assert torch.equal(hip.ftm_backward_weight(g(d_out), fm)[0], d_w)
    # only one output requested
    only_b = hip.ftm_backward_weight(g(d_out), fm, want_weight=False)
    assert only_b[0] is None and torch.equal(only_b[1], d_b)


@pytest.mark.parametrize("shape", [(24, 64, 32, 32, 65536, 256), (130, 64, 24, 24, 30000, 128), (128, 64, 32, 32, 65536, 512)])
def test_value_gradient_by_lds_dma_against_float64(hip, shape, monkeypatch):
    """The optional big-map value gradient of csrc/ftv_kernels.hip (NNUE_FTM_VAL_DMA=1: d_out split once into bf16 planes in a
    workspace and staged by LDS-DMA, table fragments straight to registers): row tail, two row tiles, the F-1 clamp, and bitwise
    equality with the default six-plane tile kernel's term order is NOT expected -- both are held to the float64 value."""
    b, fps, gh, gw, f, l1 = shape
    monkeypatch.setenv("NNUE_FTM_VAL_DMA", "1")
    if int(hip.load().nnue_ftm_backward_values_scratch(b, f, fps * gh * gw, l1)) == 0:
        pytest.skip("the six-plane products are switched off (NNUE_FTM_BF16 / NNUE_FTM_VAL_BF6): this shape takes the f32 kernels")
    assert int(hip.load().nnue_ftm_backward_values_scratch(b, f, fps * gh * gw, l1)) == 3 * b * l1 * 2
    gen = torch.Generator().manual_seed(b + f)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen)
    thr = torch.full((fps,), 0.17)
    weight, bias = torch.randn(f, l1, generator=gen) * 0.1, torch.zeros(l1)
    d_out = torch.randn(b, l1, generator=gen) / b
    _, _, _, ref_dval, _, _ = dense_reference(conv_out, thr, weight, bias, d_out)
    g = lambda t: t.to(DEV)
    fm = hip.ftm_binarize(g(conv_out), g(thr), f, l1)
    d_val = hip.ftm_backward_values(g(d_out), g(weight), fm)
    assert_close_grad(d_val.view(conv_out.shape), ref_dval, "d_conv_out (LDS-DMA kernel)", rtol=2e-5)
    active = (conv_out > 0.17).reshape(b, -1)
    assert not bool(d_val.view(b, -1)[~active.to(DEV)].any())
    for _ in range(8):  # fixed order: reproducible (and a guard against the intermittent early read this kernel once had)
        assert torch.equal(hip.ftm_backward_values(g(d_out), g(weight), fm), d_val)
    # the default: the six-plane tile kernel, d_out split in every workgroup, no workspace
    monkeypatch.setenv("NNUE_FTM_VAL_DMA", "0")
    monkeypatch.setenv("NNUE_FTM_VAL_PLANES", "0")
    assert int(hip.load().nnue_ftm_backward_values_scratch(b, f, fps * gh * gw, l1)) == 0
    d_tile = hip.ftm_backward_values(g(d_out), g(weight), fm)
    assert_close_grad(d_tile.view(conv_out.shape), ref_dval, "d_conv_out (tile kernel)", rtol=2e-5)
    # ... and the same kernel fed with the d_out planes from the workspace (NNUE_FTM_VAL_PLANES=1; measured slower, kept as a
    # knob): the same terms in the same order, bit for bit
    monkeypatch.setenv("NNUE_FTM_VAL_PLANES", "1")
    assert int(hip.load().nnue_ftm_backward_values_scratch(b, f, fps * gh * gw, l1)) == (3 * b * l1 * 2 if l1 % 32 == 0 else 0)
    assert torch.equal(hip.ftm_backward_values(g(d_out), g(weight), fm), d_tile)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_ftm_on_golden_models(hip, name):
    """The reference's own tensors: conv_out -> ft, and the input.weight / input.bias gradients of a real backward."""
    cfg, params, grads, data = golden_model(name)
    conv_out, thr = data["conv_out"], params["visual_threshold"]
    w, bias = params["input.weight"], params["input.bias"]
    f, l1 = w.shape
    p = conv_out[0].numel()
    if not hip.ftm_supported(f, p, l1):
        pytest.skip(f"P={p} / L1={l1} not multiples of 4")
    g = lambda t: t.to(DEV)
    fm = hip.ftm_binarize(g(conv_out), g(thr), f, l1)
    assert torch.equal(fm.n.cpu().long(), (data["idx"] >= 0).sum(1))
    out = hip.ftm_forward(g(w), g(bias), fm)
    assert_close_logits(out, data["ft"], "ft")
    # upstream gradient of the reference run, recomputed by the oracle's classifier backward
    _, _, _, keep = orc.loss_and_grads_explicit(params, data["images"], data["labels"], cfg["stride"])
    d_w, d_b = hip.ftm_backward_weight(g(keep["d_ft"]), fm)
    assert_close_grad(d_w, grads["input.weight"], "input.weight.grad")
    assert_close_grad(d_b, grads["input.bias"], "input.bias.grad")


def test_ftm_equals_gather_kernels_at_c4_shape(hip):
    """224x224 shape (65 536-row table, 32x32x64 map): MFMA products vs the LDS-staged gather kernels."""
    gen = torch.Generator().manual_seed(5)
    b, fps, gh, gw, f, l1 = 8, 64, 32, 32, 65536, 1024
    conv_out = torch.randn(b, fps, gh, gw, generator=gen).to(DEV)
    thr = torch.full((fps,), 0.17).to(DEV)
    weight, bias = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV), torch.randn(l1, generator=gen).to(DEV)
    d_out = (torch.randn(b, l1, generator=gen) / b).to(DEV)
    bits = hip.binarize_bits(conv_out, thr, f, l1)
    fm = hip.ftm_binarize(conv_out, thr, f, l1)
    assert torch.equal(fm.n, bits.n) and torch.equal(fm.sink, bits.sink)
    out_g, out_m = hip.ftb_forward(weight, bias, bits), hip.ftm_forward(weight, bias, fm)
    assert float((out_g - out_m).abs().max()) <= 1e-3 * float(out_g.abs().max())  # 28 k fp32 terms, two summation orders
    ref = (conv_out.reshape(b, -1)[:2] > 0.17).double() @ weight.double() + bias.double()
    assert float((out_m[:2].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    dw_g, db_g = hip.ftb_backward_weight(d_out, bits)
    dw_m, db_m = hip.ftm_backward_weight(d_out, fm)
    assert_close_grad(dw_m, dw_g, "d_weight", rtol=2e-5)
    assert_close_grad(db_m, db_g, "d_bias", rtol=2e-5)
    dv_g, dv_m = hip.ftb_backward_values(d_out, weight, bits), hip.ftm_backward_values(d_out, weight, fm)
    assert_close_grad(dv_m, dv_g, "d_conv_out", rtol=2e-5)


def test_ftm_argument_errors(hip):
    conv_out, thr = torch.randn(2, 3, 3, 3, device=DEV), torch.zeros(3, device=DEV)  # P = 27: not a multiple of 4
    assert not hip.ftm_supported(27, 27, 8)
    with pytest.raises(hip.NnueHipError, match="multiple of 4"):
        hip.ftm_binarize(conv_out, thr, 27, 8)
    fm = hip.ftm_binarize(torch.randn(2, 4, 3, 3, device=DEV), torch.zeros(4, device=DEV), 36, 8)
    with pytest.raises(ValueError):
        hip.ftm_forward(torch.zeros(35, 8, device=DEV), torch.zeros(8, device=DEV), fm)
    with pytest.raises(ValueError):
        hip.ftm_backward_values(torch.zeros(3, 8, device=DEV), torch.zeros(36, 8, device=DEV), fm)
    with pytest.raises(hip.NnueHipError):
        hip.ftm_binarize(conv_out.cpu(), thr.cpu(), 27, 8)


@pytest.mark.parametrize("shape", [(512, 8, 11, 11, 800, 1024), (1024, 8, 11, 11, 800, 1024), (128, 64, 32, 32, 65536, 1024),
                                   (37, 4, 8, 8, 256, 64), (5, 4, 3, 3, 100, 36)])
@pytest.mark.parametrize("bf16", ("0", "1"))
def test_merged_backward_is_bitwise_the_two_launches(hip, shape, bf16, monkeypatch):
    """nnue_ftm_backward (one launch) against nnue_ftm_backward_weight + nnue_ftm_backward_values: every tile-shape pair
    the policy picks for the BASELINE configurations, and shapes that fall back to the two launches.  With the f32 tiles
    (NNUE_FTM_BF16=0) the merged launch runs the very tiles of the separate launches -- bitwise equal; with the bf16-split
    tiles the merged launch may take another tile height for the weight gradient than the stand-alone call (another
    summation order of the same exact terms): equal to rounding; and its 64 x 64 value-gradient tiles run as six bf16 plane
    products where the stand-alone call of that shape keeps the f32 MFMA: equal to 2e-6 of the tensor's scale."""
    monkeypatch.setenv("NNUE_FTM_BF16", bf16)
    b, fps, gh, gw, f, l1 = shape
    gen = torch.Generator().manual_seed(f + b)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen).to(DEV)
    thr = torch.full((fps,), 0.17).to(DEV)
    weight = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV)
    d_out = (torch.randn(b, l1, generator=gen) / b).to(DEV)
    fm = hip.ftm_binarize(conv_out, thr, f, l1)
    d_w, d_b = hip.ftm_backward_weight(d_out, fm)
    d_v = hip.ftm_backward_values(d_out, weight, fm)
    m_w, m_b, m_v = hip.ftm_backward(d_out, weight, fm)
    assert torch.equal(m_b, d_b)
    if bf16 == "0":
        assert torch.equal(m_w, d_w) and torch.equal(m_v, d_v)
    else:
        assert_close_grad(m_w, d_w, "d_weight, merged vs stand-alone", rtol=2e-6)
        assert_close_grad(m_v, d_v, "d_conv_out, merged vs stand-alone", rtol=2e-6)
        assert torch.equal(m_v == 0, d_v == 0)  # the same exact zeros at the inactive positions


@pytest.mark.parametrize("shape", [(512, 32, 32, 8, 3, 800), (128, 224, 224, 64, 7, 65536), (5, 17, 23, 4, 2, 300), (3, 96, 96, 8, 10, 800),
                                   (2, 40, 40, 12, 1, 100)])
def test_conv_binarize_is_bitwise_the_two_calls(hip, shape):
    b, h, w, fps, stride, f = shape
    gen = torch.Generator().manual_seed(h + fps)
    images = torch.randn(b, 3, h, w, generator=gen).to(DEV)
    weight = (torch.randn(fps, 3, 3, 3, generator=gen) * 0.2).to(DEV)
    thr = (torch.randn(fps, generator=gen) * 0.1).to(DEV)
    conv_ref = hip.conv3x3_forward(images, weight, stride)
    if conv_ref[0].numel() % 4:
        pytest.skip("map size not a multiple of 4")
    fm_ref = hip.ftm_binarize(conv_ref, thr, f, 64)
    conv, fm = hip.ftm_conv_binarize(images, weight, thr, stride, f, 64)
    assert torch.equal(conv, conv_ref) and torch.equal(fm.bits, fm_ref.bits)
    assert torch.equal(fm.n, fm_ref.n) and torch.equal(fm.sink, fm_ref.sink)
    again = hip.ftm_conv_binarize(images, weight, thr, stride, f, 64, conv_out=conv, fm=fm)
    assert torch.equal(again[1].n, fm_ref.n)  # counters are re-zeroed when a sample is split over slices


def test_random_shapes_sweep(hip):
    """Twenty random shapes (ragged batch, table larger / smaller than the map, odd widths): every product-form entry
    point against the float64 restatement, and the merged launch against the separate ones."""
    rng = torch.Generator().manual_seed(2025)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))  # noqa: E731
    done = 0
    while done < 20:
        b, fps, gh, gw = ri(1, 200), ri(1, 12), ri(1, 9), ri(1, 9)
        p = fps * gh * gw
        if p % 4:
            continue
        f = max(1, p + ri(-p // 2, p // 2))
        l1 = 4 * ri(1, 80)
        conv_out = torch.randn(b, fps, gh, gw, generator=rng)
        thr = torch.randn(fps, generator=rng) * 0.3
        weight, bias = torch.randn(f, l1, generator=rng) * 0.1, torch.randn(l1, generator=rng)
        d_out = torch.randn(b, l1, generator=rng) / b
        ref_out, ref_dw, ref_db, ref_dval, ref_n, ref_sink = dense_reference(conv_out, thr, weight, bias, d_out)
        g = lambda t: t.to(DEV)  # noqa: E731
        fm = hip.ftm_binarize(g(conv_out), g(thr), f, l1)
        tag = (b, fps, gh, gw, f, l1)
        assert torch.equal(fm.n.cpu(), ref_n) and torch.equal(fm.sink.cpu(), ref_sink), tag
        assert_close_logits(hip.ftm_forward(g(weight), g(bias), fm), ref_out, f"out {tag}", rtol=2e-5)
        d_w, d_b, d_v = hip.ftm_backward(g(d_out), g(weight), fm)
        assert_close_grad(d_w, ref_dw, f"d_weight {tag}", rtol=2e-5)
        assert_close_grad(d_b, ref_db, f"d_bias {tag}", rtol=2e-5)
        assert_close_grad(d_v.view(conv_out.shape), ref_dval, f"d_conv_out {tag}", rtol=2e-5)
        s_w, s_b = hip.ftm_backward_weight(g(d_out), fm)
        assert torch.equal(s_b, d_b) and torch.equal(hip.ftm_backward_values(g(d_out), g(weight), fm), d_v), tag
        assert_close_grad(s_w, d_w, f"d_weight merged vs stand-alone {tag}", rtol=2e-6)  # the merged launch may take bf16-split tiles
        done += 1


@pytest.mark.parametrize("shape", [(512, 8, 11, 11, 800, 1024, 128), (1024, 8, 11, 11, 800, 1024, 128), (37, 4, 8, 8, 256, 64, 32),
                                   (130, 8, 10, 10, 800, 192, 40), (64, 8, 11, 11, 800, 256, 7)])
def test_fused_forward_forms_the_layer1_slabs(hip, shape):
    """nnue_ftm_forward_l1: the FeatureTransformer output is bitwise nnue_ftm_forward's, and the slabs sum to the
    classifier's first pre-activation l0 @ w1^T (pairwise block nnue.py:660-666, Linear nnue.py:728-730)."""
    b, fps, gh, gw, f, l1, l2 = shape
    if not hip.ftm_forward_l1_supported(b, f, fps * gh * gw, l1, l2):
        pytest.skip("not a fused-forward shape")
    gen = torch.Generator().manual_seed(b + l2)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen).to(DEV)
    thr = torch.full((fps,), 0.17).to(DEV)
    weight, bias = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV), torch.randn(l1, generator=gen).to(DEV)
    w1 = (torch.randn(l2, l1, generator=gen) / l1 ** 0.5).to(DEV)
    fm = hip.ftm_binarize(conv_out, thr, f, l1)
    ref_ft = hip.ftm_forward(weight, bias, fm)
    part = torch.full(((l1 // 64) * b * l2 + 16,), float("nan"), device=DEV)
    ft = hip.ftm_forward_l1(weight, bias, fm, w1, part)
    assert torch.equal(ft, ref_ft)
    slabs = part[:(l1 // 64) * b * l2].view(l1 // 64, b, l2)
    assert not bool(torch.isnan(slabs).any()) and bool(torch.isnan(part[(l1 // 64) * b * l2:]).all())
    x = ref_ft.double()
    l0 = torch.cat([x[:, :l1 // 2] * x[:, l1 // 2:], x[:, :l1 // 2]], dim=1)
    assert_close_grad(slabs.double().sum(0), l0 @ w1.double().t(), "layer-1 pre-activation", rtol=1e-5)
    assert not hip.ftm_forward_l1_supported(128, 65536, 65536, 1024, 128)  # split-K forward: separate layer-1 launch


@pytest.mark.parametrize("shape", [(512, 8, 11, 11, 800, 1024, 128), (1024, 8, 11, 11, 800, 1024, 128), (37, 4, 8, 8, 256, 128, 32),
                                   (300, 8, 11, 11, 800, 256, 4), (64, 8, 11, 11, 2000, 384, 20)])
def test_merged_backward_carries_the_classifier_weight_gradient(hip, shape):
    """nnue_ftm_backward with the d_w1 rider: d_w1 = d_z1^T l0 against float64 (l0 = pairwise block of ft), and the
    launch's own three outputs stay bitwise what they are without the rider."""
    b, fps, gh, gw, f, l1, l2 = shape
    import os
    if os.environ.get("NNUE_FTM_SPLIT_BACKWARD", "0") == "1":
        pytest.skip("the merged launch is switched off (NNUE_FTM_SPLIT_BACKWARD=1): no rider")
    assert hip.ftm_backward_cw_supported(b, f, fps * gh * gw, l1, l2)
    gen = torch.Generator().manual_seed(b + l2)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen).to(DEV)
    thr = torch.full((fps,), 0.3).to(DEV)
    weight = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV)
    d_out = (torch.randn(b, l1, generator=gen) / b).to(DEV)
    ft = torch.rand(b, l1, generator=gen).to(DEV)
    d_z1 = (torch.randn(b, l2, generator=gen) / b).to(DEV)
    fm = hip.ftm_binarize(conv_out, thr, f, l1)
    m_w, m_b, m_v = hip.ftm_backward(d_out, weight, fm)
    d_w1 = torch.full((l2, l1), float("nan"), device=DEV)
    r_w, r_b, r_v = hip.ftm_backward(d_out, weight, fm, ft=ft, d_z1=d_z1, d_w1=d_w1)
    assert torch.equal(m_w, r_w) and torch.equal(m_b, r_b) and torch.equal(m_v, r_v)
    half = l1 // 2
    ft64 = ft.double().cpu()
    l0 = torch.cat([ft64[:, :half] * ft64[:, half:], ft64[:, :half]], dim=1)
    ref = d_z1.double().cpu().t() @ l0
    assert_close_grad(d_w1.cpu(), ref, "d_w1 (rider)")


def test_rider_is_refused_where_the_launch_is_split(hip):
    b, f, p, l1, l2 = 128, 65536, 64 * 32 * 32, 1024, 128  # C4: the two products run as separate launches
    assert not hip.ftm_backward_cw_supported(b, f, p, l1, l2)
    assert not hip.ftm_backward_cw_supported(512, 800, 968, 1000, 128)  # L1 % 128
    assert not hip.ftm_backward_cw_supported(512, 800, 968, 1024, 30)  # L2 % 4
    fm = hip.FeatureMatrix.empty(8, 968, 800, 1000, DEV)
    fm.bits.zero_(); fm.sink.zero_(); fm.n.zero_()
    z = lambda *s: torch.zeros(*s, device=DEV)  # noqa: E731
    with pytest.raises(hip.NnueHipError):
        hip.ftm_backward(z(8, 1000), z(800, 1000), fm, ft=z(8, 1000), d_z1=z(8, 128), d_w1=z(128, 1000))


@pytest.mark.parametrize("shape", [(512, 8, 11, 11, 800, 1024), (1024, 8, 11, 11, 800, 1024), (128, 64, 32, 32, 65536, 1024),
                                   (37, 4, 8, 8, 300, 64)])
def test_weight_gradient_tiles_leave_their_squared_norm(hip, shape):
    """nnue_ftm_backward(sq_partial): the per-tile sums of squares add up to ||d_weight[:min(F-1, P)]||^2, outputs keep
    their bits; nnue_sgd_step(ext=...) with them gives the norm / update of the plain call within float rounding."""
    b, fps, gh, gw, f, l1 = shape
    p = fps * gh * gw
    n_sq = hip.ftm_backward_sq_count(b, f, p, l1)
    assert n_sq > 0
    gen = torch.Generator().manual_seed(b + f)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen).to(DEV)
    thr = torch.full((fps,), 0.2).to(DEV)
    weight = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV)
    d_out = (torch.randn(b, l1, generator=gen) / b).to(DEV)
    fm = hip.ftm_binarize(conv_out, thr, f, l1)
    ref_w, ref_b, ref_v = hip.ftm_backward(d_out, weight, fm)
    sq = torch.full((n_sq,), float("nan"), device=DEV)
    d_w, d_b, d_v = hip.ftm_backward(d_out, weight, fm, sq_partial=sq)
    assert torch.equal(d_w, ref_w) and torch.equal(d_b, ref_b) and torch.equal(d_v, ref_v)
    rows = min(f - 1, p)
    want = float((d_w[:rows].double() ** 2).sum())
    got = float(sq.double().sum())
    assert abs(got - want) <= 1e-5 * max(want, 1e-30)
    if f * l1 > 4_000_000:
        return  # the optimizer part on the small tables only (keeps the test's memory small)
    count = f * l1 + 1024
    grads = torch.cat([d_w.reshape(-1), torch.randn(1024, generator=gen).to(DEV)])
    params = torch.randn(count, generator=gen).to(DEV)
    outs = []
    for ext in (None, (sq, 0, rows * l1)):
        pp, mom, norm = params.clone(), torch.zeros(count, device=DEV), torch.zeros((), device=DEV)
        scratch = torch.empty((hip.sgd_scratch_bytes(count),), dtype=torch.uint8, device=DEV)
        hip.sgd_step(pp, grads.clone(), mom, 0.05, 0.9, 1e-4, 0.5, 0.5, True, norm, scratch, ext=ext)
        outs.append((pp, float(norm)))
    assert abs(outs[0][1] - outs[1][1]) <= 1e-5 * outs[0][1]
    assert_close_grad(outs[1][0], outs[0][0], "updated parameters", rtol=1e-5)


@pytest.mark.parametrize("shape", ((128, 64, 32, 32, 65536, 1024), (5, 4, 3, 3, 100, 36), (200, 8, 11, 11, 800, 256), (300, 8, 10, 10, 800, 64),
                                   (37, 4, 8, 8, 200, 64)))
def test_gram_form_of_the_weight_gradient_norm(hip, shape):
    """||A^T D||_F^2 = sum (A A^T) . (D D^T) over the table rows the map reaches (nnue_ftm_gram_sqnorm), against the
    float64 norm of the materialised gradient."""
    b, fps, gh, gw, f, l1 = shape
    gen = torch.Generator().manual_seed(b + f)
    conv_out = torch.randn(b, fps, gh, gw, generator=gen)
    thr = torch.full((fps,), 0.17)
    d_out = torch.randn(b, l1, generator=gen) / b + 0.3 / b  # a common component: off-diagonal Gram terms matter
    fm = hip.ftm_binarize(conv_out.to(DEV), thr.to(DEV), f, l1)
    direct = min(f - 1, fps * gh * gw)
    a = (conv_out > 0.17).reshape(b, -1)[:, :direct].double()
    ref = float(((a.t() @ d_out.double()) ** 2).sum())
    gram = torch.full((hip.ftm_gram_scratch(fm),), float("nan"), device=DEV)  # nothing in the scratch needs clearing
    part = torch.empty(int(hip.load().nnue_ftm_gram_sq_count(b, l1)), device=DEV)
    for _ in range(2):
        hip.ftm_gram_sqnorm(fm, d_out.to(DEV), gram, part)
        got = float(part.double().sum())
        assert abs(got - ref) <= 2e-6 * ref, (got, ref)
    assert torch.equal(gram[:b * b].view(b, b).cpu().double(), a @ a.t())  # common active positions: exact integers
    # the tail rows' workgroups riding in the Gram product's launch (nnue_ftm_gram_sqnorm_tail): bitwise the two separate calls
    dw_ref, db_ref = torch.full((f, l1), 7.0, device=DEV), torch.empty(l1, device=DEV)
    hip.ftm_backward_tail_rows(d_out.to(DEV), fm, dw_ref, db_ref)
    dw_got, db_got = torch.full((f, l1), 7.0, device=DEV), torch.empty(l1, device=DEV)
    gram2, part2 = torch.full_like(gram, float("nan")), torch.empty_like(part)
    hip.ftm_gram_sqnorm(fm, d_out.to(DEV), gram2, part2, tail=(dw_got, db_got))
    torch.cuda.synchronize()
    assert torch.equal(part2, part) and torch.equal(gram2[:b * b], gram[:b * b])
    assert torch.equal(dw_got, dw_ref) and torch.equal(db_got, db_ref)


def test_table_update_in_the_product_epilogue_equals_the_materialised_path(hip, monkeypatch):
    """NNUE_FUSE_TABLE_UPDATE: Gram norm + optimizer pass that skips the table + product with the SGD epilogue, against the
    path that writes d_weight and reads it back -- three steps with momentum, weight decay and an active clip."""
    import os
    if os.environ.get("NNUE_FT_PATH", "auto") not in ("auto", "mfma"):
        pytest.skip("another FeatureTransformer kernel family is forced (NNUE_FT_PATH)")
    import nnue
    from nnue_hip.trainer import NnueTrainer
    runs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("NNUE_FUSE_TABLE_UPDATE", mode)
        torch.manual_seed(4)
        model = nnue.NNUE(nnue.GridFeatureSet(16, 16), 256, 32, 16, num_classes=10, input_size=64).to(DEV)
        tr = NnueTrainer(model, 96, (64, 64), lr=0.05, momentum=0.9, weight_decay=1e-3, max_grad_norm=0.5, use_graph=True)
        assert tr.fuse_table_update == (mode == "1") and tr.grads_materialised == (mode == "0")
        gen = torch.Generator().manual_seed(8)
        norms = []
        for _ in range(3):
            images, labels = torch.randn(96, 3, 64, 64, generator=gen), torch.randint(0, 10, (96,), generator=gen)
            tr.step(images.to(DEV), labels.to(DEV))
            norms.append(float(tr.grad_norm))
        runs[mode] = (norms, tr.flat_params.clone(), tr.flat_momentum.clone())
    for a, b in zip(runs["1"][0], runs["0"][0]):
        assert abs(a - b) <= 2e-6 * b and b > 0.5  # the clip is active
    assert_close_grad(runs["1"][1], runs["0"][1], "parameters", rtol=1e-6)
    assert_close_grad(runs["1"][2], runs["0"][2], "momentum", rtol=1e-5)


@pytest.mark.parametrize("span", (20, 100, 125))
def test_bf16_split_is_exact_over_the_exponent_range(hip, span):
    """The three-way truncation split x = hi + mid + lo is exact, so with ONE active position per sample the bf16-split
    products must return the operand itself, bit for bit: out[b] = W[p_b] (zero bias) and d_W[p_b] = d_out[b], for
    magnitudes from 2^-span to 2^span.  Span 125 reaches the edge of the normal range: below |x| = 2^-103 the lo (then the
    mid) part is a bf16 denormal, which the matrix unit flushes -- what is lost is smaller than the smallest normal
    float (measured 9e-41), so there the bar is an absolute 2^-126."""
    b, fps, gh, gw, f, l1 = 24, 64, 32, 32, 65536, 256
    p = fps * gh * gw
    lib = hip.load()
    if not (lib.nnue_ftm_uses_bf16(0, b, f, p, l1) == 1 and lib.nnue_ftm_uses_bf16(1, b, f, p, l1) == 1):
        pytest.skip("the bf16-split tiles are switched off (NNUE_FTM_BF16=0)")
    gen = torch.Generator().manual_seed(span)

    def wide(*shape):  # random sign, exponent uniform in [-span, span], full 24-bit mantissa
        mant = 1.0 + torch.rand(*shape, generator=gen, dtype=torch.float64)
        e = torch.randint(-span, span + 1, shape, generator=gen).double()
        sign = torch.randint(0, 2, shape, generator=gen).double() * 2 - 1
        return (sign * mant * torch.exp2(e)).float()

    pos = torch.randperm(p - 1, generator=gen)[:b]  # distinct positions below the clamp sink
    conv_out = torch.full((b, p), -1.0)
    conv_out[torch.arange(b), pos] = 1.0
    weight, d_out = wide(f, l1), wide(b, l1)
    fm = hip.ftm_binarize(conv_out.view(b, fps, gh, gw).to(DEV), torch.zeros(fps, device=DEV), f, l1)
    assert torch.equal(fm.n.cpu(), torch.ones(b, dtype=torch.int32))
    out = hip.ftm_forward(weight.to(DEV), torch.zeros(l1, device=DEV), fm)
    d_w, _ = hip.ftm_backward_weight(d_out.to(DEV), fm)
    if span <= 100:
        assert torch.equal(out.cpu(), weight[pos]), float((out.cpu() - weight[pos]).abs().max())
        assert torch.equal(d_w[pos.to(DEV)].cpu(), d_out)
    else:
        tiny = 2.0 ** -126
        assert float((out.cpu().double() - weight[pos].double()).abs().max()) < tiny
        assert float((d_w[pos.to(DEV)].cpu().double() - d_out.double()).abs().max()) < tiny
        big = weight[pos].abs() >= 2.0 ** -100
        assert torch.equal(out.cpu()[big], weight[pos][big])
    rest = torch.ones(f, dtype=torch.bool)
    rest[pos] = False
    assert not bool(d_w[rest.to(DEV)].any())


def test_random_big_map_shapes_sweep(hip):
    """Six random shapes with maps of thousands to tens of thousands of positions (ragged batches up to 300, odd widths):
    the shapes that take the bf16-split tiles of either depth, the six-plane-product value gradient, the i8 Gram product
    and the update in the product's epilogue -- each against the float64 restatement."""
    rng = torch.Generator().manual_seed(7)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=rng))  # noqa: E731
    done = 0
    while done < 6:
        b, fps, gh, gw = ri(1, 300), ri(8, 64), ri(12, 32), ri(12, 32)
        p = fps * gh * gw
        if p % 4 or p < 4096:
            continue
        f = max(2, p + ri(-p // 4, p // 8))
        l1 = 4 * ri(8, 96)
        tag = (b, fps, gh, gw, f, l1)
        conv_out = torch.randn(b, fps, gh, gw, generator=rng)
        thr = torch.randn(fps, generator=rng) * 0.3
        weight, bias = torch.randn(f, l1, generator=rng) * 0.1, torch.randn(l1, generator=rng)
        d_out = torch.randn(b, l1, generator=rng) / b + 0.2 / b
        ref_out, ref_dw, ref_db, ref_dval, ref_n, ref_sink = dense_reference(conv_out, thr, weight, bias, d_out)
        g = lambda t: t.to(DEV)  # noqa: E731
        fm = hip.ftm_binarize(g(conv_out), g(thr), f, l1)
        assert torch.equal(fm.n.cpu(), ref_n) and torch.equal(fm.sink.cpu(), ref_sink), tag
        assert_close_grad(hip.ftm_forward(g(weight), g(bias), fm), ref_out, f"out {tag}", rtol=1e-5)
        d_w, d_b = hip.ftm_backward_weight(g(d_out), fm)
        assert_close_grad(d_w, ref_dw, f"d_weight {tag}", rtol=2e-5)
        assert_close_grad(d_b, ref_db, f"d_bias {tag}", rtol=2e-5)
        d_v = hip.ftm_backward_values(g(d_out), g(weight), fm)
        assert_close_grad(d_v.view(conv_out.shape), ref_dval, f"d_conv_out {tag}", rtol=2e-5)
        active = (conv_out > thr.view(1, -1, 1, 1)).reshape(b, -1)
        assert not bool(d_v.view(b, -1)[~active.to(DEV)].any()), tag
        # Gram norm of the rows the map reaches, and the update in the product's epilogue against the float64 update
        direct = min(f - 1, p)
        gram = torch.full((hip.ftm_gram_scratch(fm),), float("nan"), device=DEV)
        part = torch.empty(int(hip.load().nnue_ftm_gram_sq_count(b, l1)), device=DEV)
        hip.ftm_gram_sqnorm(fm, g(d_out), gram, part)
        ref_sq = float((ref_dw[:direct] ** 2).sum())
        assert abs(float(part.double().sum()) - ref_sq) <= 2e-6 * ref_sq, tag
        a = active[:, :direct].double()
        assert torch.equal(gram[:b * b].view(b, b).cpu().double(), a @ a.t()), tag
        w_dev, m_dev = g(weight).clone(), (torch.randn(f, l1, generator=rng) * 0.01).to(DEV)
        mom0 = m_dev.cpu().double()
        coef = torch.tensor(0.37, device=DEV)
        lr, mu, wd, gs = 0.05, 0.9, 1e-3, 1.7
        hip.ftm_backward_weight_update(g(d_out), fm, w_dev, m_dev, coef, lr, mu, wd, gs, False)
        gg = 0.37 * gs * ref_dw[:direct] + wd * weight[:direct].double()
        mm = mu * mom0[:direct] + gg
        assert_close_grad(m_dev[:direct], mm, f"momentum rows {tag}", rtol=2e-5)
        assert_close_grad(w_dev[:direct], weight[:direct].double() - lr * mm, f"updated rows {tag}", rtol=1e-6)
        assert torch.equal(w_dev[direct:].cpu(), weight[direct:]) and torch.equal(m_dev[direct:].cpu().double(), mom0[direct:]), tag
        done += 1
