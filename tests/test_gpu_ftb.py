"""Bit-mask / LDS-staged FeatureTransformer kernels (nnue_binarize_bits, nnue_ftb_*) against the CPU oracle
and against the list kernels.  ``-m gpu``."""
import pytest
import torch

import nnue_oracle as orc
from conftest import assert_close_grad, assert_close_logits

pytestmark = pytest.mark.gpu
DEV = "cuda"

GEOMS = [  # B, fps, gh, gw, F, L1
    (512, 8, 11, 11, 800, 1024),   # C2: 968 positions, clamp sink at row 799
    (70, 3, 5, 7, 200, 256),       # table larger than the map: no sink, rows >= 105 unreachable
    (33, 8, 4, 4, 128, 512),       # F == P: row F-1 is its own (single) sink position
    (130, 1, 1, 1, 1, 256),        # one-row table: everything is sink
    (5, 2, 3, 3, 4, 256),
    (64, 4, 9, 9, 324, 1024),      # F == P, 64-aligned batch
    (16, 16, 16, 16, 4000, 1024),  # several LDS tiles, sink at a non-aligned row
    (257, 8, 11, 11, 800, 1024),   # ragged batch
]


@pytest.fixture(scope="module")
def hip():
    from nnue_hip import lib
    lib.load()
    return lib


def unpack(words, nbits):
    """int64 words [R, W] -> bool [R, nbits] (bit k of word w = column 64*w + k)."""
    w = words.cpu()
    cols = torch.arange(nbits)
    return ((w[:, cols // 64] >> (cols % 64)) & 1).bool()


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("thr_scale", (0.3, 1e6, -1e6))
def test_bits_and_products(hip, geom, thr_scale):
    b, fps, gh, gw, f, l1 = geom
    gen = torch.Generator().manual_seed(sum(geom))
    x = torch.randn(b, fps, gh, gw, generator=gen)
    thr = torch.randn(fps, generator=gen) * 0.3 if abs(thr_scale) < 1 else torch.full((fps,), float(thr_scale))
    w = torch.randn(f, l1, generator=gen) * 0.1
    bias = torch.randn(l1, generator=gen) * 0.1
    up = torch.randn(b, l1, generator=gen)
    p = fps * gh * gw
    xd, td, wd, bd, ud = (t.to(DEV) for t in (x, thr, w, bias, up))

    bits = hip.binarize_bits(xd, td, f, l1)
    on = (x > thr.view(1, -1, 1, 1)).reshape(b, p)
    assert torch.equal(unpack(bits.maskW, p), on)                       # bit-exact ids
    assert not bool(unpack(bits.maskW, bits.maskW.shape[1] * 64)[:, p:].any())  # padding bits are zero
    assert torch.equal(bits.n.cpu().long(), on.sum(1))
    sink = on[:, f - 1:].sum(1).float() if f - 1 < p else torch.zeros(b)
    assert torch.equal(bits.sink.cpu(), sink)
    direct = min(f - 1, p)
    mt = unpack(bits.maskT, b)
    assert torch.equal(mt[:direct], on[:, :direct].t())
    assert not bool(mt[direct:f - 1].any())
    assert torch.equal(mt[f - 1], sink != 0) and bool(mt[f].all())
    assert not bool(unpack(bits.maskT, bits.maskT.shape[1] * 64)[:, b:].any())

    # tile lists decode to exactly the mask bits, ascending, padded with 128
    def decode(tl, tc):
        tl, tc = tl.cpu().long(), tc.cpu().long()
        e = torch.arange(128)
        pos = ((e >> 2) & 3) * 32 + (e >> 4) * 4 + (e & 3)
        off = tl[..., pos] & 0xffff  # entry order; u16 LDS offsets = local index * 256, padding 32768
        assert bool((off % 256 == 0).all())
        return off // 256, tc
    lw, cw = decode(bits.tlW, bits.tcW)
    for t in range(lw.shape[1]):
        seg = on[:, t * 128:(t + 1) * 128].clone()
        seg[:, max(0, direct - t * 128):] = False
        assert torch.equal(cw[:, t], seg.sum(1)), t
        for s in range(0, b, max(1, b // 7)):
            k = int(cw[s, t])
            assert lw[s, t, :k].tolist() == seg[s].nonzero().flatten().tolist()
            assert bool((lw[s, t, k:] == 128).all())
    lt, ct = decode(bits.tlT, bits.tcT)
    assert torch.equal(ct.sum(1), mt.sum(1))
    for r in (0, direct // 2, f - 1, f):
        got = [int(v) + 128 * t for t in range(lt.shape[1]) for v in lt[r, t, :int(ct[r, t])]]
        assert got == mt[r].nonzero().flatten().tolist(), r

    # oracle (float64) through the id-list form
    idx, _ = orc.active_lists(x, thr)
    val = (idx >= 0).double()
    ref_out = orc.ft_forward(w.double(), bias.double(), idx, val)
    r_w, r_b, r_val = orc.ft_backward(w.double(), idx, val, up.double())
    ref_dx = torch.zeros(b, p, dtype=torch.float64)
    keep = idx >= 0
    ref_dx[keep.nonzero(as_tuple=True)[0], idx[keep]] = r_val[keep]

    out = hip.ftb_forward(wd, bd, bits)
    d_w, d_b = hip.ftb_backward_weight(ud, bits)
    d_x = hip.ftb_backward_values(ud, wd, bits)
    assert_close_logits(out, ref_out, "ftb out")
    assert_close_grad(d_w, r_w, "ftb d_weight")
    assert_close_grad(d_b, r_b, "ftb d_bias")
    assert_close_grad(d_x, ref_dx, "ftb d_conv_out")
    assert bool((d_x.cpu()[~on] == 0).all())  # inactive positions are exactly zero (written, not left over)

    # the list kernels on the same batch (different fp32 summation order: the LDS kernels keep four partial sums)
    act = hip.binarize_features(xd, td, f)
    l_w, l_b = hip.ft_backward_weight(ud, act, f)
    assert_close_grad(d_w, l_w, "ftb vs list d_weight")
    assert_close_grad(d_b, l_b, "ftb vs list d_bias")
    assert torch.equal(d_w, hip.ftb_backward_weight(ud, bits)[0])  # and is itself bitwise reproducible
    assert_close_logits(out, hip.ft_forward(wd, bd, act), "ftb vs list forward")
    assert_close_grad(d_x, hip.ft_backward_values(ud, wd, act, p), "ftb vs list values")
    # bitwise reproducible
    assert torch.equal(out, hip.ftb_forward(wd, bd, bits)) and torch.equal(d_x, hip.ftb_backward_values(ud, wd, bits))


def test_ftb_rejects_unsupported_width(hip):
    assert hip.ftb_supported(1024) and hip.ftb_supported(256) and not hip.ftb_supported(64) and not hip.ftb_supported(2048)
    bits = hip.FeatureBits.empty(4, 32, 16, 64, DEV)
    with pytest.raises(hip.NnueHipError, match="not supported"):
        hip.ftb_forward(torch.zeros(16, 64, device=DEV), torch.zeros(64, device=DEV), bits)


def test_c4_shaped_table_paths_agree(hip):
    """BASELINE config 4 geometry (64 x 32 x 32 map = 65 536 positions = table rows, L1 = 1024): the LDS-staged
    kernels (tile splitting + finish pass forward, one-slot variant backward) against the list kernels, and a
    slice of the outputs against a float64 oracle."""
    b, fps, gh, gw, f, l1 = 24, 64, 32, 32, 65536, 1024
    gen = torch.Generator().manual_seed(44)
    x = torch.randn(b, fps, gh, gw, generator=gen)
    thr = torch.full((fps,), 0.1)
    w = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV)
    bias = torch.zeros(l1, device=DEV)
    up = torch.randn(b, l1, generator=gen).to(DEV)
    xd, td = x.to(DEV), thr.to(DEV)
    bits = hip.binarize_bits(xd, td, f, l1)
    act = hip.binarize_features(xd, td, f)
    assert torch.equal(bits.n, act.n)
    out, ref_out = hip.ftb_forward(w, bias, bits), hip.ft_forward(w, bias, act)
    # two fp32 summation orders over ~28 000 terms each (sequential in the list kernel, 4 x 4 partial sums here):
    # their mutual distance is the summation-order effect at this size; the float64 slice below is the real bar
    assert_close_logits(out, ref_out, "c4 forward", rtol=1e-3)
    d_w, d_b = hip.ftb_backward_weight(up, bits)
    l_w, l_b = hip.ft_backward_weight(up, act, f)
    assert_close_grad(d_w, l_w, "c4 d_weight")
    assert_close_grad(d_b, l_b, "c4 d_bias")
    d_x = hip.ftb_backward_values(up, w, bits)
    assert_close_grad(d_x, hip.ft_backward_values(up, w, act, fps * gh * gw), "c4 d_conv_out")
    # float64 oracle on three samples
    on = (x > 0.1).reshape(b, -1)
    for s in (0, 11, 23):
        ids = on[s].nonzero().flatten().to(DEV)
        ref = w[ids].double().sum(0)
        assert float((out[s].double() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
        dots = (w[ids].double() @ up[s].double())
        assert float((d_x[s][ids].double() - dots).abs().max()) <= 1e-4 * float(dots.abs().max())
        assert int((d_x[s] != 0).sum()) <= ids.numel()
