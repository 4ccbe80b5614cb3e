"""GPU input pipeline.  The un-augmented path (validation / test loaders in the reference) is an
exact formula -- albumentations' Normalize(mean, std, max_pixel_value=255) followed by ToTensorV2
(data/datasets.py:358-372) -- restated here in numpy.  The augmented path draws from this build's own
counter-based generator, not from albumentations' stream (albumentations is not installed): parity unpinned,
so it is checked through the policy's properties (data/datasets.py:183-195).  ``-m gpu``."""
import numpy as np
import pytest
import torch

from nnue_hip import lib
from nnue_hip.input_pipeline import GpuImageDataset

pytestmark = pytest.mark.gpu

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def normalize_reference(u8):
    """albumentations.Normalize on uint8 HWC images + ToTensorV2, restated: (img - 255*mean) * (1 / (255*std))."""
    img = u8.astype(np.float32)
    img = (img - MEAN * 255.0) * (1.0 / (STD * 255.0))
    return np.transpose(img, (0, 3, 1, 2))


def make(n=300, h=32, w=32, seed=0):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, size=(n, h, w, 3), dtype=np.uint8), rng.randint(0, 10, size=(n,))


def test_plain_path_is_the_exact_normalisation():
    images, labels = make()
    ds = GpuImageDataset(images, labels)
    idx = torch.tensor([5, 0, 299, 17, 17, 123], device="cuda")
    out, lab = ds.batch(idx)
    ref = normalize_reference(images[idx.cpu().numpy()])
    assert out.shape == (6, 3, 32, 32) and out.dtype == torch.float32
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= 2e-6
    assert lab.cpu().tolist() == labels[idx.cpu().numpy()].tolist()
    # non-square and odd sizes, writing into a caller-provided buffer (a trainer input slot)
    images2, labels2 = make(40, 17, 23, seed=1)
    ds2 = GpuImageDataset(images2, labels2)
    buf, lbuf = torch.empty(8, 3, 17, 23, device="cuda"), torch.empty(8, dtype=torch.int64, device="cuda")
    out2, _ = ds2.batch(torch.arange(8, device="cuda"), out=buf, labels_out=lbuf)
    assert out2.data_ptr() == buf.data_ptr()
    assert float(np.abs(buf.cpu().numpy() - normalize_reference(images2[:8])).max()) <= 2e-6


def test_loader_covers_the_dataset_like_a_dataloader():
    images, labels = make(103)
    ds = GpuImageDataset(images, labels)
    assert len(ds.loader(16)) == 7 and len(ds.loader(16, drop_last=True)) == 6
    seen = torch.cat([lab for _, lab in ds.loader(16)])
    assert seen.cpu().tolist() == labels.tolist()  # unshuffled: dataset order, ragged last batch (7)
    shuffled = ds.loader(16, shuffle=True)
    sizes, all_means = [], []
    for x, lab in shuffled:
        sizes.append(x.shape[0])
        all_means.append(x.sum().item())
    assert sizes == [16] * 6 + [7]
    total = sum(all_means)
    ref_total = float(normalize_reference(images).astype(np.float64).sum())
    assert abs(total - ref_total) <= 1e-4 * abs(ref_total)  # every sample exactly once


def test_augmentation_follows_the_light_policy():
    n = 4000
    images = np.full((n, 32, 32, 3), 0, dtype=np.uint8)
    images[:, :, :, :] = np.arange(32, dtype=np.uint8)[None, None, :, None] * 4 + 60  # value encodes the column: flips are visible
    labels = np.zeros(n, dtype=np.int64)
    ds = GpuImageDataset(images, labels, augment=True, seed=7)
    idx = torch.arange(n, device="cuda")
    out, _ = ds.batch(idx)
    plain = torch.from_numpy(normalize_reference(images[:1])).cuda()[0]  # every image is the same
    flipped = torch.flip(plain, dims=[2])
    o = out
    d_plain = (o - plain).abs().amax(dim=(1, 2, 3))
    d_flip = (o - flipped).abs().amax(dim=(1, 2, 3))
    untouched = (d_plain < 1e-6) | (d_flip < 1e-6)
    is_flip = (o - flipped).abs().mean(dim=(1, 2, 3)) < (o - plain).abs().mean(dim=(1, 2, 3))  # mean: a hole does not decide it
    assert 0.46 < float(is_flip.float().mean()) < 0.54                         # HorizontalFlip p = 0.5
    # untouched = neither brightness/contrast (p=0.2) nor dropout (p=0.2): 0.8 * 0.8 = 0.64
    assert 0.60 < float(untouched.float().mean()) < 0.68
    # dropout: exactly one 1x1 hole (5% of 32 -> 1 pixel) holding Normalize(0) in all three channels
    zero = torch.from_numpy(((0 - MEAN * 255.0) / (STD * 255.0)).astype(np.float32)).cuda().view(1, 3, 1, 1)
    holes = ((o - zero).abs() < 1e-6).all(dim=1).flatten(1).sum(dim=1)
    assert set(holes.cpu().tolist()) <= {0, 1}
    assert 0.17 < float((holes == 1).float().mean()) < 0.23                      # CoarseDropout p = 0.2
    # brightness/contrast: pixels stay valid uint8 levels after the LUT (value*255-grid), within +-(0.1*v + 25.5)
    levels = (o * torch.from_numpy(STD * 255.0).cuda().view(1, 3, 1, 1) + torch.from_numpy(MEAN * 255.0).cuda().view(1, 3, 1, 1))
    assert float((levels - levels.round()).abs().max()) < 1e-2 and float(levels.min()) >= -1e-3 and float(levels.max()) <= 255.001
    ref_levels = torch.where(is_flip.view(-1, 1, 1, 1), flipped, plain) * torch.from_numpy(STD * 255.0).cuda().view(1, 3, 1, 1) \
        + torch.from_numpy(MEAN * 255.0).cuda().view(1, 3, 1, 1)
    not_hole = ~((o - zero).abs() < 1e-6).all(dim=1, keepdim=True)
    dev = ((levels - ref_levels).abs() * not_hole).amax(dim=(1, 2, 3))
    assert float(dev.max()) <= 0.1 * 184 + 25.5 + 1.0
    changed = dev > 0.5
    assert 0.16 < float(changed.float().mean()) < 0.24                             # RandomBrightnessContrast p = 0.2


def test_augmentation_is_reproducible_and_varies_by_visit():
    images, labels = make(64)
    a, b = GpuImageDataset(images, labels, augment=True, seed=3), GpuImageDataset(images, labels, augment=True, seed=3)
    idx = torch.arange(64, device="cuda")
    x1, x2 = a.batch(idx)[0], b.batch(idx)[0]
    assert torch.equal(x1, x2)                       # same seed, same step -> same draws
    assert not torch.equal(a.batch(idx)[0], x1)      # next visit of the same samples: new draws
    c = GpuImageDataset(images, labels, augment=True, seed=4)
    assert not torch.equal(c.batch(idx)[0], x1)


def test_trainer_consumes_pipeline_batches_in_place():
    import nnue
    from nnue_hip.trainer import NnueTrainer
    images, labels = make(256)
    ds = GpuImageDataset(images, labels, augment=True, seed=1)
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10).cuda()
    tr = NnueTrainer(model, 64, (32, 32), lr=0.01, momentum=0.9, input_slots=2)
    losses = []
    for epoch in range(2):
        for i, idx in enumerate(torch.randperm(256, device="cuda").split(64)):
            slot = i % 2
            ds.batch(idx, out=tr.inputs[slot][0], labels_out=tr.inputs[slot][1])  # written straight into the slot
            losses.append(float(tr.step(slot=slot)))
    assert all(np.isfinite(losses)) and len(losses) == 8
    with pytest.raises(ValueError):
        GpuImageDataset(np.zeros((4, 8, 8), dtype=np.uint8), np.zeros(4))
    # labels are validated once at construction (inside the step only the -1 padding sentinel is ignored)
    with pytest.raises(ValueError, match="labels must lie"):
        GpuImageDataset(images, np.where(np.arange(256) == 7, -1, labels))
    with pytest.raises(ValueError, match="labels must lie"):
        GpuImageDataset(images, labels, num_classes=int(np.asarray(labels).max()))
    with pytest.raises(lib.NnueHipError):
        lib.load_batch(torch.zeros(4, 8, 8, 3, dtype=torch.uint8), torch.zeros(4, dtype=torch.int64), torch.zeros(2, dtype=torch.int64), False, 0, 0)


def test_in_place_epoch_equals_copy_feeding():
    import nnue
    from nnue_hip.input_pipeline import train_epoch
    from nnue_hip.trainer import NnueTrainer
    images, labels = make(64 * 5 + 9)  # five full batches and a short one
    models, sums = [], []
    for mode in ("sequential", "in_place"):
        torch.manual_seed(0)
        model = nnue.NNUE(nnue.GridFeatureSet(10, 8), 256, 32, 16, num_classes=10).cuda()
        tr = NnueTrainer(model, 64, (32, 32), lr=0.01, momentum=0.9, input_slots=2)
        ds = GpuImageDataset(images, labels, augment=True, seed=5)
        loader = ds.loader(64, shuffle=True, generator=torch.Generator().manual_seed(11))
        total = 0.0
        for epoch in range(2):
            if mode == "sequential":
                for x, y in loader:
                    total += float(tr.step(x, y))
            else:
                s, n = train_epoch(tr, loader)
                assert n == 6
                total += float(s)
        models.append(model)
        sums.append(total)
    assert abs(sums[0] - sums[1]) <= 1e-4 * abs(sums[0])
    for (k, p), (_, q) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert float((p - q).detach().abs().max()) <= 1e-5 * max(1.0, float(q.detach().abs().max())), k


def test_run_training_on_a_gpu_resident_dataset(tmp_path):
    from nnue_hip import train_loop
    from test_train_loop import write_config
    cfg = train_loop.load_config(write_config(tmp_path, opt="sgd", lr=0.02, epochs=2))
    rng = np.random.RandomState(0)
    labels = rng.randint(0, 10, size=200)
    images = np.clip(rng.randint(0, 120, size=(200, 32, 32, 3)) + labels[:, None, None, None] * 12, 0, 255).astype(np.uint8)
    train = GpuImageDataset(images[:135], labels[:135], augment=True, seed=2).loader(16, shuffle=True)  # 8 full + one of 7
    val = GpuImageDataset(images[135:], labels[135:]).loader(16)
    logs = []
    torch.manual_seed(0)
    res = train_loop.run_training(cfg, train, val, checkpoint_dir=tmp_path / "ckpt", log=logs.append)
    assert res.steps == 2 * 9 and len(res.history) == 2
    assert all(np.isfinite(row["train/epoch_loss"]) and np.isfinite(row["val/loss"]) for row in res.history)
