"""nnue_ftm_backward_weight_update_forward (include/nnue_hip.h): the big table's SGD update of one step (train.py:457-464) and
the FeatureTransformer forward of the next step (nnue.py:686-710) in one pass over the table.  The contract is BITWISE
equality with the two separate entry points (same MFMA operands, accumulators and order), which themselves are held to
the oracle by tests/test_gpu_ftm.py and tests/test_gpu_step_shapes.py.  ``-m gpu``."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _map(lib, gen, b, p, f, l1, density=0.42):
    bits = (torch.rand(b, p, generator=gen) < density).to(torch.uint8)
    bits[0].fill_(1)
    if b > 1:
        bits[-1].zero_()
    fm = lib.FeatureMatrix.empty(b, p, f, l1, DEV)
    fm.bits.copy_(bits)
    fm.n.copy_(bits.sum(1).to(torch.int32))
    fm.sink.copy_(bits[:, f - 1:].sum(1).to(torch.float32))  # ids >= F-1 all hit table row F-1 (nnue.py:701)
    return fm


# (B, F, P, L1[, B_next]): split-K forward over >= 4096 table rows; ragged batch, table rows that are no multiple of a tile, a
# clamp sink (P > F - 1), the 224x224 shape itself, and the factor exchange's shapes (B = the all-gathered global batch of 2, 4
# and 8 ranks: 4, 8, 16 K tiles in the weight-gradient product; B_next = this rank's next batch)
SHAPES = [(128, 8192, 8192, 256), (100, 9001, 9000, 256), (128, 5000, 8192, 128), (64, 16400, 16400, 192), (128, 65536, 65536, 1024),
          (256, 8192, 8192, 256, 128), (500, 8192, 8192, 256, 125), (1024, 8192, 8192, 256, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("first,mom", [(False, 0.9), (True, 0.9), (False, 0.0)])
def test_bitwise_the_two_separate_calls(shape, first, mom):
    from nnue_hip import lib
    lib.load()
    b, f, p, l1 = shape[:4]
    bn = shape[4] if len(shape) > 4 else b
    if not lib.ftm_update_forward_supported(b, f, p, l1, bn):
        pytest.skip("shape is not a split-K forward over a big table")
    gen = torch.Generator().manual_seed(b * 7 + f)
    fm, fm_next = _map(lib, gen, b, p, f, l1), _map(lib, gen, bn, p, f, l1, density=0.3)
    d_out = (torch.randn(b, l1, generator=gen) * 0.05).to(DEV)
    weight = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV)
    momentum = (torch.randn(f, l1, generator=gen) * 0.01).to(DEV) if mom else None
    bias = torch.randn(l1, generator=gen).to(DEV)
    coef = torch.tensor([0.37], device=DEV)
    lr_dev = torch.tensor([0.02], device=DEV)
    direct = min(f - 1, p)

    w_ref, m_ref = weight.clone(), (momentum.clone() if mom else None)
    lib.ftm_backward_weight_update(d_out, fm, w_ref, m_ref[:direct] if mom else None, coef, 0.5, mom, 2e-4, 1.0 / b, first, lr_dev=lr_dev)
    out_ref = lib.ftm_forward(w_ref, bias, fm_next)
    torch.cuda.synchronize()

    w_got, m_got = weight.clone(), (momentum.clone() if mom else None)
    out_got = torch.full((bn, l1), float("nan"), device=DEV)
    fm_next.scratch.zero_()
    for _ in range(2 if first else 1):  # twice from the same state when nothing accumulates (first step ignores momentum)
        w_got.copy_(weight)
        lib.ftm_backward_weight_update_forward(d_out, fm, w_got, m_got[:direct] if mom else None, coef, 0.5, mom, 2e-4, 1.0 / b, first,
                                               fm_next, bias, out_got, lr_dev=lr_dev)
    torch.cuda.synchronize()
    assert torch.equal(w_got, w_ref), f"table differs: max |d| = {(w_got - w_ref).abs().max().item():.3e}"
    if mom:
        assert torch.equal(m_got, m_ref), "momentum differs"
    assert torch.equal(out_got, out_ref), f"next forward differs: max |d| = {(out_got - out_ref).abs().max().item():.3e}"
    assert not torch.equal(w_got, weight)  # it did move


def test_against_float64():
    """The fused pass against plain float64 arithmetic (not only against its two halves)."""
    from conftest import assert_close_grad, assert_close_logits
    from nnue_hip import lib
    lib.load()
    b, f, p, l1 = 128, 8192, 8192, 256
    if not lib.ftm_update_forward_supported(b, f, p, l1):
        pytest.skip("a developer knob took the forward off the tiles the fused pass is built on")
    gen = torch.Generator().manual_seed(5)
    fm, fm_next = _map(lib, gen, b, p, f, l1), _map(lib, gen, b, p, f, l1)
    d_out = (torch.randn(b, l1, generator=gen) * 0.05).to(DEV)
    weight = (torch.randn(f, l1, generator=gen) * 0.1).to(DEV)
    momentum = (torch.randn(f, l1, generator=gen) * 0.01).to(DEV)
    bias = torch.randn(l1, generator=gen).to(DEV)
    coef, lr, mom, wd, scale = 0.8, 0.05, 0.9, 2e-4, 1.0 / b
    direct = f - 1
    A, An = fm.bits.double(), fm_next.bits.double()
    g = coef * scale * (A[:, :direct].T @ d_out.double()) + wd * weight[:direct].double()
    m64 = mom * momentum[:direct].double() + g
    w64 = weight.double().clone()
    w64[:direct] -= lr * m64
    out64 = An[:, :direct] @ w64[:direct] + bias.double() + fm_next.sink.double()[:, None] * w64[f - 1]
    out = torch.empty(b, l1, device=DEV)
    lib.ftm_backward_weight_update_forward(d_out, fm, weight, momentum[:direct], torch.tensor([coef], device=DEV), lr, mom, wd, scale, False,
                                           fm_next, bias, out)
    torch.cuda.synchronize()
    assert_close_grad(weight, w64.float(), "table after the update")
    assert_close_grad(momentum[:direct], m64.float(), "momentum")
    assert_close_logits(out, out64.float(), "next forward", rtol=1e-4)


def test_argument_errors():
    from nnue_hip import lib
    L = lib.load()
    import os
    if os.environ.get("NNUE_FTM_BF16") == "0" or os.environ.get("NNUE_FTM_BF_KT64") == "0":
        pytest.skip("a developer knob took the forward off the tiles the fused pass is built on")
    assert L.nnue_ftm_update_forward_supported(128, 128, 65536, 65536, 1024) == 1
    assert L.nnue_ftm_update_forward_supported(1024, 128, 65536, 65536, 1024) == 1  # eight ranks' factors
    assert L.nnue_ftm_update_forward_supported(512, 512, 800, 968, 1024) == 0        # launch-sized table
    assert L.nnue_ftm_update_forward_supported(256, 256, 65536, 65536, 1024) == 0    # next batch wider than one forward tile
    assert L.nnue_ftm_update_forward_supported(384, 128, 65536, 65536, 1024) == 0    # six K tiles: not one of 1, 2, 4, 8, 16
    assert L.nnue_ftm_update_forward_supported(128, 128, 65536, 65536, 1000) == 0    # L1 % 64
    z = torch.zeros(1 << 16, device=DEV)
    u = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    args = (u.data_ptr(), z.data_ptr(), 8, 800, 968, 64, z.data_ptr(), 0, z.data_ptr(), 0.1, 0.0, 0.0, 1.0, 0, 0, u.data_ptr() + 4096, z.data_ptr(),
            8, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1 << 16, 0)
    assert L.nnue_ftm_backward_weight_update_forward(*args) != 0
    assert b"split-K" in L.nnue_hip_last_error()
