"""The overlapped data-parallel step of NnueTrainer with two ranks on ONE GPU (gloo backend moves the
gradient buckets; on a multi-GPU node the same code runs over RCCL).  Checks: both ranks end with bitwise
identical parameters, and the trajectory equals a single-process run on the global batch.  ``-m gpu``."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CFG = dict(grid=10, fps=8, l1=256, l2=32, l3=16, classes=10)
OPT = dict(lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0)
GLOBAL_BATCH, STEPS, WORLD = 32, 4, 2


def _batch(step):
    g = torch.Generator().manual_seed(4242 + step)
    return torch.randn(GLOBAL_BATCH, 3, 32, 32, generator=g), torch.randint(0, CFG["classes"], (GLOBAL_BATCH,), generator=g)


def _build():
    for p in (str(ROOT / "nnue-vision_amd"),):
        if p not in sys.path:
            sys.path.insert(0, p)
    import nnue
    torch.manual_seed(0)
    return nnue.NNUE(nnue.GridFeatureSet(CFG["grid"], CFG["fps"]), CFG["l1"], CFG["l2"], CFG["l3"], num_classes=CFG["classes"]).to("cuda")


def _worker(rank, port, out_dir, use_graph, buckets):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NNUE_DP_BUCKETS=str(buckets))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        model = _build()
        from nnue_hip.trainer import NnueTrainer
        tr = NnueTrainer(model, GLOBAL_BATCH // WORLD, (32, 32), use_graph=use_graph, input_slots=2, **OPT)
        assert tr.dp.world == WORLD and tr.bucket_split == 8 + 8 * 27 and tr.dp.buckets == buckets
        sl = tr.dp.shard(GLOBAL_BATCH)
        losses = []
        for s in range(STEPS):
            images, labels = _batch(s)
            losses.append(float(tr.step(images[sl].cuda(), labels[sl].cuda(), slot=s % 2)))
        torch.cuda.synchronize()
        torch.save({"flat": tr.flat_params.cpu(), "norm": float(tr.grad_norm), "losses": losses}, Path(out_dir) / f"rank{rank}.pt")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("buckets", (1, 2))
@pytest.mark.parametrize("use_graph", (False, True))
def test_two_ranks_match_single_process(tmp_path, use_graph, buckets):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(port, str(tmp_path), use_graph, buckets), nprocs=WORLD, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    assert r0["norm"] == r1["norm"]
    # single process, whole batch
    model = _build()
    from nnue_hip.trainer import NnueTrainer
    tr = NnueTrainer(model, GLOBAL_BATCH, (32, 32), use_graph=use_graph, **OPT)
    ref_losses = []
    for s in range(STEPS):
        images, labels = _batch(s)
        ref_losses.append(float(tr.step(images.cuda(), labels.cuda())))
    ref = tr.flat_params.cpu()
    scale = float(ref.abs().max())
    assert float((r0["flat"] - ref).abs().max()) <= 2e-4 * scale
    assert abs(r0["norm"] - float(tr.grad_norm)) <= 2e-4 * float(tr.grad_norm)
    # the two half-batch mean losses average to the whole-batch loss
    for a, b, c in zip(r0["losses"], r1["losses"], ref_losses):
        assert abs((a + b) / 2 - c) <= 2e-4 * max(1.0, abs(c))
