"""Wall-clock of whole training epochs through run_training on a CIFAR-10-sized synthetic dataset that
lives in HBM: augmenting input pipeline -> fused step -> evaluation of the train and validation sets every epoch,
as train.py:350-398 does.  Prints one JSON object."""
import json, os, sys, tempfile, textwrap, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
from nnue_hip import train_loop  # noqa: E402
from nnue_hip.input_pipeline import GpuImageDataset  # noqa: E402

CONFIG = """
    name = "epoch_bench"
    batch_size = 512
    num_workers = 0
    num_classes = 10
    l1_size = 1024
    l2_size = 128
    l3_size = 32
    input_size = 32
    grid_size = 10
    num_features_per_square = 8
    learning_rate = 0.01
    weight_decay = 2e-4
    momentum = 0.9
    optimizer_type = "sgd"
    subset = 1.0
    max_epochs = 4
    max_grad_norm = 1.0
    use_augmentation = True
    keep_alive = True
    log_dir = "logs"
    project_name = "bench"
"""

with tempfile.TemporaryDirectory() as tmp:
    path = os.path.join(tmp, "cfg.py")
    open(path, "w").write(textwrap.dedent(CONFIG))
    cfg = train_loop.load_config(path)
    rng = np.random.RandomState(0)
    labels = rng.randint(0, 10, 60000)
    images = np.clip(rng.randint(0, 160, (60000, 32, 32, 3)) + labels[:, None, None, None] * 9, 0, 255).astype(np.uint8)
    train = GpuImageDataset(images[:50000], labels[:50000], augment=True, seed=1).loader(512, shuffle=True)
    val = GpuImageDataset(images[50000:], labels[50000:]).loader(512)
    stamps = []
    t0 = time.perf_counter()
    res = train_loop.run_training(cfg, train, val, checkpoint_dir=os.path.join(tmp, "ckpt"), log=lambda line: stamps.append(time.perf_counter()))
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    per_epoch = [b - a for a, b in zip(stamps[:-1], stamps[1:])]
    print(json.dumps({"train_images": 50000, "val_images": 10000, "epochs": len(res.history), "total_s": total,
                      "steady_epoch_s": min(per_epoch) if per_epoch else None, "per_epoch_s": per_epoch,
                      "first_epoch_s_incl_setup": stamps[0] - t0, "final_train_acc": res.history[-1]["train/epoch_accuracy"],
                      "final_val_acc": res.history[-1]["val/accuracy"], "steps": res.steps}))
