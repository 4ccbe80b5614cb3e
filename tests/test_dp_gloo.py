"""Data-parallel path with world size 2 over gloo on the CPU.

The GPU trainer's N>1 logic is FlatLayout (one flat buffer, aligned views) + DataParallel (equal batch
shards, ONE all-reduce(sum) on the flat gradient, 1/world scale, broadcast of the initial parameters).
Here the per-rank gradients come from the CPU oracle (test infrastructure) instead of the HIP kernels, so
the collective/sharding code that runs on RCCL is exercised unchanged:

  * averaged shard gradients == gradient of the global batch;
  * after 3 clip+SGD steps both ranks hold bitwise-identical parameters;
  * the trajectory equals a single-process run on the global batch.
"""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
WORLD = 2
CFG = dict(grid=10, fps=8, l1=64, l2=32, l3=8, classes=10, input_size=32)
OPT = dict(lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0)
GLOBAL_BATCH, STEPS = 8, 3


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(step):
    g = torch.Generator().manual_seed(777 + step)
    return torch.randn(GLOBAL_BATCH, 3, 32, 32, generator=g), torch.randint(0, CFG["classes"], (GLOBAL_BATCH,), generator=g)


def _worker(rank, port, out_dir):
    for p in (str(ROOT / "nnue-vision_amd"), str(ROOT / "oracle")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import nnue
        import nnue_oracle as orc
        from nnue_hip.trainer import DataParallel, FlatLayout

        torch.manual_seed(100 + rank)  # deliberately different seeds: the broadcast must equalise the replicas
        model = nnue.NNUE(nnue.GridFeatureSet(CFG["grid"], CFG["fps"]), CFG["l1"], CFG["l2"], CFG["l3"], num_classes=CFG["classes"])
        layout = FlatLayout.of(model)
        dp = DataParallel()
        assert (dp.world, dp.rank) == (WORLD, rank) and dp.grad_scale == 0.5
        flat_params = layout.pack({k: p.detach() for k, p in model.named_parameters() if k != "nnue2score"})
        dp.broadcast(flat_params)
        params = {k: v.clone() for k, v in layout.views(flat_params).items()}
        params["nnue2score"] = torch.tensor(600.0)
        single = {k: v.clone() for k, v in params.items()}  # reference run: whole batch in one process
        bufs, bufs_single = {}, {}
        sl = dp.shard(GLOBAL_BATCH)
        assert (sl.start, sl.stop) == (rank * 4, rank * 4 + 4)
        for step in range(STEPS):
            images, labels = _batch(step)
            _, _, g_local, _ = orc.loss_and_grads_explicit(params, images[sl], labels[sl], 3)
            flat_grads = layout.pack(g_local)
            dp.allreduce_sum(flat_grads)
            g_avg = {k: v * dp.grad_scale for k, v in layout.views(flat_grads).items()}
            _, _, g_global, _ = orc.loss_and_grads_explicit(single, images, labels, 3)
            for k in g_global:
                scale = max(float(g_global[k].abs().max()), 1e-12)
                assert float((g_avg[k] - g_global[k]).abs().max()) <= 1e-4 * scale, (step, k)
            orc.sgd_step(params, g_avg, bufs, **OPT)
            orc.sgd_step(single, g_global, bufs_single, **OPT)
        final = layout.pack({k: v for k, v in params.items() if k != "nnue2score"})
        gathered = [torch.empty_like(final) for _ in range(WORLD)]
        dist.all_gather(gathered, final)
        assert torch.equal(gathered[0], gathered[1]), "replicas diverged"
        for k in params:
            scale = max(float(single[k].abs().max()), 1e-12)
            assert float((params[k] - single[k]).abs().max()) <= 2e-4 * scale, k
        with pytest.raises(ValueError):
            dp.shard(7)
        # sharded update plumbing: reduce-scatter + all-gather of equal float4-aligned shards == all-reduce
        lay2 = FlatLayout.of(model, pad_to=4 * WORLD)
        assert lay2.count % (4 * WORLD) == 0 and lay2.count >= layout.count and lay2.offsets == layout.offsets
        flat = torch.arange(lay2.count, dtype=torch.float32) * (rank + 1)
        want = torch.arange(lay2.count, dtype=torch.float32) * 3  # ranks 1x + 2x
        shard = torch.empty(lay2.count // WORLD)
        dp.reduce_scatter_sum(flat.clone(), shard)
        assert torch.equal(shard, dp.shard_of(want))
        full = torch.zeros(lay2.count)
        dp.shard_of(full).copy_(shard)
        dp.all_gather(full, dp.shard_of(full))
        assert torch.equal(full, want)
        with pytest.raises(ValueError):
            dp.shard_of(torch.zeros(lay2.count + 1))
        (Path(out_dir) / f"ok{rank}").write_text("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process(tmp_path):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["ok0", "ok1"]


def test_single_process_defaults():
    sys.path.insert(0, str(ROOT / "nnue-vision_amd"))
    from nnue_hip.trainer import DataParallel
    dp = DataParallel()
    assert (dp.world, dp.rank, dp.grad_scale) == (1, 0, 1.0)
    assert dp.shard(512) == slice(0, 512)
    t = torch.ones(4)
    assert dp.allreduce_sum(t) is None and torch.equal(t, torch.ones(4))
