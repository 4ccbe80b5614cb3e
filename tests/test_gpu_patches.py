"""The im2col form of the conv front for large strides (include/nnue_hip.h: nnue_ftm_conv_binarize_patches,
nnue_ste_conv_backward_patches; self.conv of nnue.py:640 and StraightThroughBinary.backward, nnue.py:28-54): the forward launch
leaves patches[27][B*G] and the backward reads them instead of the strided pixels and of conv_out, which it re-forms with the
forward's own fmaf chain.  Contract: BITWISE the pixel form (itself held to the oracle by tests/test_gpu_kernels.py and the
whole-step tests).  ``-m gpu``."""
import pytest
import torch
import torch.nn.functional as F

import nnue
from nnue_hip.trainer import NnueTrainer

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (B, H, W, stride, fps, table rows F)
SHAPES = [(4, 32, 32, 3, 8, 800), (2, 224, 224, 7, 64, 65536), (3, 40, 56, 5, 20, 1500), (5, 33, 33, 4, 32, 2000), (130, 16, 16, 1, 4, 1024)]


@pytest.mark.parametrize("b,h,w,stride,fps,f", SHAPES)
def test_forward_leaves_the_im2col_form_and_the_same_map(b, h, w, stride, fps, f):
    from nnue_hip import lib
    lib.load()
    gen = torch.Generator().manual_seed(h * 13 + fps)
    images = torch.randn(b, 3, h, w, generator=gen).to(DEV)
    weight = (torch.randn(fps, 3, 3, 3, generator=gen) * 0.3).to(DEV)
    thr = (torch.rand(fps, generator=gen) * 0.2).to(DEV)
    gh, gw = lib.conv_out_hw(h, w, stride)
    conv_ref, fm_ref = lib.ftm_conv_binarize(images, weight, thr, stride, f, 64)
    patches = torch.full((27, b * gh * gw), float("nan"), device=DEV)
    conv_got, fm_got = lib.ftm_conv_binarize(images, weight, thr, stride, f, 64, patches=patches)
    none, fm_only = lib.ftm_conv_binarize(images, weight, thr, stride, f, 64, patches=torch.empty_like(patches), write_conv_out=False)
    torch.cuda.synchronize()
    assert none is None
    for fm in (fm_got, fm_only):
        assert torch.equal(fm.bits, fm_ref.bits) and torch.equal(fm.n, fm_ref.n) and torch.equal(fm.sink, fm_ref.sink)
    assert torch.equal(conv_got, conv_ref)
    # the pixels themselves, zero where a tap falls off the image: exactly torch's unfold
    cols = F.unfold(images, kernel_size=3, padding=1, stride=stride)  # [B, 27, G], term = ci*9 + kh*3 + kw
    assert cols.shape == (b, 27, gh * gw)
    assert torch.equal(patches, cols.permute(1, 0, 2).reshape(27, b * gh * gw))


@pytest.mark.parametrize("b,h,w,stride,fps,f", SHAPES)
@pytest.mark.parametrize("stages", (3, 1))
def test_backward_from_the_patches_is_bitwise_the_pixel_form(b, h, w, stride, fps, f, stages):
    from nnue_hip import lib
    lib.load()
    gen = torch.Generator().manual_seed(h * 17 + fps)
    images = torch.randn(b, 3, h, w, generator=gen).to(DEV)
    weight = (torch.randn(fps, 3, 3, 3, generator=gen) * 0.3).to(DEV)
    thr = (torch.rand(fps, generator=gen) * 0.2).to(DEV)
    gh, gw = lib.conv_out_hw(h, w, stride)
    patches = torch.empty((27, b * gh * gw), device=DEV)
    conv_out, fm = lib.ftm_conv_binarize(images, weight, thr, stride, f, 64, patches=patches)
    d = torch.randn(b, fps, gh, gw, generator=gen).to(DEV) * fm.bits.view(b, fps, gh, gw).float()  # zero at inactive positions, as in the step
    need = int(lib.load().nnue_ste_conv_backward_scratch(b, fps, gh, gw))
    s_ref, s_got = torch.zeros(need, dtype=torch.uint8, device=DEV), torch.zeros(need, dtype=torch.uint8, device=DEV)
    t_ref, w_ref = lib.ste_conv_backward(images, conv_out, thr, d, stride, scratch=s_ref, stages=stages)
    for given in (None, conv_out):  # conv_out re-formed from the patches / read
        s_got.zero_()
        t_got, w_got = lib.ste_conv_backward_patches(patches, weight, thr, d, gh, gw, scratch=s_got, stages=stages, conv_out=given)
        torch.cuda.synchronize()
        assert torch.equal(s_got, s_ref), "stage-1 partials differ"
        if stages == 3:
            assert torch.equal(t_got, t_ref) and torch.equal(w_got, w_ref)
            assert float(w_ref.abs().max()) > 0 and float(t_ref.abs().max()) > 0


def test_trainer_with_and_without_patches_is_bitwise_the_same(monkeypatch):
    """Three steps of the trainer at a stride-4 shape with the patch form forced on and off: identical parameters."""
    out = []
    for mode, reform in (("1", "0"), ("1", "1"), ("0", "0")):
        monkeypatch.setenv("NNUE_CONV_PATCHES", mode)
        monkeypatch.setenv("NNUE_CONV_REFORM", reform)
        torch.manual_seed(0)
        model = nnue.NNUE(nnue.GridFeatureSet(16, 8), 256, 32, 16, num_classes=10, input_size=64).to(DEV)
        tr = NnueTrainer(model, 32, (64, 64), lr=0.05, momentum=0.9, weight_decay=1e-4, max_grad_norm=1.0, use_graph=True, input_slots=2)
        assert tr.use_patches == (mode == "1") and tr.stride == 4
        gen = torch.Generator().manual_seed(3)
        losses = []
        for s in range(3):
            losses.append(float(tr.step(torch.randn(32, 3, 64, 64, generator=gen).to(DEV), torch.randint(0, 10, (32,), generator=gen).to(DEV), slot=s % 2)))
        torch.cuda.synchronize()
        out.append((tr.flat_params.clone(), tr.flat_momentum.clone(), losses))
    for o in out[1:]:
        assert out[0][2] == o[2]
        assert torch.equal(out[0][0], o[0]) and torch.equal(out[0][1], o[1])


def test_auto_policy(monkeypatch):
    """auto: the patch form where the images are more than twice their patches (224x224 at stride 7), not at the CIFAR shapes."""
    monkeypatch.delenv("NNUE_CONV_PATCHES", raising=False)
    torch.manual_seed(0)
    small = NnueTrainer(nnue.NNUE(nnue.GridFeatureSet(10, 8), 64, 32, 8, num_classes=10).to(DEV), 8, (32, 32), lr=0.01)
    assert not small.use_patches and small.patches is None
    big = NnueTrainer(nnue.NNUE(nnue.GridFeatureSet(32, 8), 64, 32, 8, num_classes=10, input_size=224).to(DEV), 4, (224, 224), lr=0.01)
    assert big.use_patches and big.stride == 7 and tuple(big.patches.shape) == (27, 4 * 32 * 32)
