#!/usr/bin/env python3
"""NNUE training throughput on MI355X  (metric: BASELINE.json -- images/sec of the full training step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c3|c4]      (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
conv -> binarise/compact -> FeatureTransformer gather-accumulate -> pairwise + classifier -> cross-entropy
-> full backward -> (all-reduce of the flat gradient when N > 1) -> clip_grad_norm_ + SGD, i.e. what
train.py:359-366 does per batch.  Default workload = BASELINE configs[1]: CIFAR-10 shapes, batch 512 per
GPU, 800 -> 1024/128/32 -> 10, fp32, SGD(lr 0.01, momentum 0.9, wd 2e-4), clip 1.0 (config/train_nnue.py).
Weak scaling: every GPU keeps batch 512 (N=8 is BASELINE configs[4], global batch 4096).

Rank 0 prints ONE JSON line.  Beyond the contract fields it carries
  roofline      dominant kernel: algorithmic bytes per launch (one gathered / accumulated table
                row = L1*4 bytes) / its average duration, measured here with HIP events on the launch
                stream in an instrumented pass that follows the timed region (same buffers, same process);
  kernels       the same figure for every C entry point of the step;
  cpu_baseline  the oracle's loop-form port of the reference CPU path (N=1 only) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "nnue-vision_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: grid, fps, image, l1, l2, l3, classes, per-GPU batch            
    "c1": dict(grid=10, fps=8, image=32, l1=64, l2=32, l3=8, classes=10, batch=32),
    "c2": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=10, batch=512),
    # BASELINE configs[2]: 8 layer-stack buckets + clipped ReLU -- a build extension (the reference trains one stack with
    # plain ReLU: SURVEY section 0, D1/D2); "c3k1" is the same shape as the reference has it
    "c3": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=100, batch=1024, buckets=8, clip=1.0),
    "c3k1": dict(grid=10, fps=8, image=32, l1=1024, l2=128, l3=32, classes=100, batch=1024),
    "c4": dict(grid=32, fps=64, image=224, l1=1024, l2=128, l3=32, classes=1000, batch=128),
}
OPT = dict(lr=0.01, momentum=0.9, weight_decay=2e-4, max_grad_norm=1.0)  # config/train_nnue.py:29-36
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA, dense (= the f32 vector rate)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparsity headline)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--density", type=float, default=None,
                    help="override visual_threshold per channel so about this fraction of features is active (SURVEY 8d sweep); "
                         "default: the reference's 0.1 threshold (~0.43)")
    ap.add_argument("--spread", action="store_true", default=None,
                    help="give every sample its own offset and gain (and make the conv weights positive) so that the "
                         "active-feature counts, and with them the layer-stack buckets, cover the whole range.  DEFAULT for the "
                         "bucketed workload (c3): plain randn images put every sample into one bucket, which would time 8 stacks "
                         "of weights on a one-stack workload; the line then also carries config.single_bucket_images_per_sec")
    ap.add_argument("--no-spread", dest="spread", action="store_false", help="plain randn images also for the bucketed workload")
    ap.add_argument("--steps-per-graph", type=int, default=0,
                    help="consecutive steps (each on the next input slot) replayed as one hipGraph (NnueTrainer.step_many); "
                         "steps left over after the whole groups run as single-step graphs; 1 = one graph launch per step; "
                         "0 (default) = the group size in 5..32 that leaves the fewest single steps for --steps")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-compare", action="store_true",
                    help="skip the second pass with the LDS-staged gather kernels (NNUE_FT_PATH=bits) and its HBM-roofline object")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="rehearse the self-launch path without a GPU: the ranks rendezvous over gloo, all-reduce a small flat "
                         "buffer through the trainer's DataParallel plumbing and rank 0 prints the JSON line (tests/test_bench_launcher.py)")
    return ap.parse_args()


def self_launch(args) -> int:
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks ourselves as a child
    `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` (one rank per GPU over RCCL), relay rank 0's
    single JSON line and return the children's exit code.  Runs before this process touches the GPU (nothing here calls
    into HIP), so there is no exec from a process that has initialised the device."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited 0 without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def launcher_selftest(args, world, rank):
    """CPU rehearsal of the multi-rank plumbing (no kernels, no timing claim): gloo rendezvous, the trainer's DataParallel
    all-reduce on a flat buffer, max-over-ranks reduction, one JSON line from rank 0."""
    sys.path.insert(0, str(ROOT / "nnue-vision_amd"))
    from nnue_hip.trainer import DataParallel
    dist.init_process_group("gloo")
    dp = DataParallel()
    flat = torch.full((1024,), float(rank + 1))
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dp.allreduce_sum(flat)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest (no kernels)", "value": 0.0, "unit": "images/sec", "n_gpus": args.gpus,
                          "n_ranks": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(t.item()) * 1e3 / max(1, args.steps), 4), "selftest": True,
                          "allreduce_ok": bool(dp.world == world)}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def cpu_baseline(cfg, budget_s):
    """Loop-form port of the reference CPU path (oracle/nnue_oracle.py) on the host cores: full steps
    (forward, backward, clip, SGD) on the same synthetic shapes; bounded by `budget_s`."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import nnue_oracle as orc
    cores = min(os.cpu_count() or 1, 16)  # capped at 16: the box's CPU share for one GPU ("of" = what the host reports)
    torch.set_num_threads(cores)
    stride = orc.conv_stride(cfg["image"], cfg["grid"])
    params = orc.init_params(cfg["grid"], cfg["fps"], cfg["l1"], cfg["l2"], cfg["l3"], cfg["classes"], 0, buckets=cfg.get("buckets", 1))
    gen = torch.Generator().manual_seed(1234)
    batch = cfg["batch"]
    images = torch.randn(batch, 3, cfg["image"], cfg["image"], generator=gen)
    labels = torch.randint(0, cfg["classes"], (batch,), generator=gen)
    bufs = {}

    def one():
        _, _, grads, _ = orc.loss_and_grads_loop(params, images, labels, stride, cfg.get("clip"))
        orc.sgd_step(params, grads, bufs, OPT["lr"], OPT["momentum"], OPT["weight_decay"], OPT["max_grad_norm"])

    t0 = time.perf_counter()
    one()  # warm-up, also sizes the sample
    first = time.perf_counter() - t0
    steps = max(1, min(20, int(budget_s / max(first, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 1), "unit": "images/sec", "cores": cores, "of": os.cpu_count(), "kind": "port",
            "sample": f"{steps} full training steps of batch {batch} after 1 warm-up ({dt:.1f} s), "
                      f"oracle loop form (per-sample Python loops + autograd, as nnue.py:601-633/:694-708), torch CPU fp32"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.launcher_selftest:
        return launcher_selftest(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")))
    # stdout carries exactly ONE line (the JSON): libraries that chat on fd 1 (RCCL prints a version banner
    # at communicator creation) are sent to stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run, also with one rank
        dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm

    import nnue
    from nnue_hip.trainer import NnueTrainer

    cfg = WORKLOADS[args.workload]
    if args.spread is None:
        args.spread = cfg.get("buckets", 1) > 1
    torch.manual_seed(0)
    model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"],
                      num_classes=cfg["classes"], input_size=cfg["image"], num_ls_buckets=cfg.get("buckets", 1),
                      clip_activations=cfg.get("clip")).to(dev)
    if args.spread:
        with torch.no_grad():
            model.conv.weight.abs_()
    B = cfg["batch"]
    SLOTS = 4
    trainer = NnueTrainer(model, B, (cfg["image"], cfg["image"]), group=None, use_graph=not args.no_graph,
                          input_slots=SLOTS, **OPT)

    # synthetic batches, resident in HBM (the trainer's input ring) before the clock starts; every rank
    # draws its own.  Steps rotate over the slots, so consecutive steps see different data.
    gen = torch.Generator().manual_seed(1234 + rank)
    for images, labels in trainer.inputs:
        x = torch.randn(B, 3, cfg["image"], cfg["image"], generator=gen)
        if args.spread:
            x = x * (0.5 + torch.rand(B, 1, 1, 1, generator=gen)) + (3.2 * torch.rand(B, 1, 1, 1, generator=gen) - 1.6)
        images.copy_(x)
        labels.copy_(torch.randint(0, cfg["classes"], (B,), generator=gen))

    if args.density is not None:
        if not 0.0 < args.density < 1.0:
            raise SystemExit("--density must be in (0, 1)")
        with torch.no_grad():  # per-channel quantile of this rank's conv outputs over every input slot
            convs = [torch.nn.functional.conv2d(im, model.conv.weight.detach(), stride=model.conv.stride, padding=1).transpose(0, 1).flatten(1)
                     for im, _ in trainer.inputs]
            per_channel = torch.cat(convs, dim=1)
            if per_channel.shape[1] > 1 << 22:
                per_channel = per_channel[:, :: per_channel.shape[1] // (1 << 22) + 1]
            model.visual_threshold.copy_(torch.stack([torch.quantile(c, 1.0 - args.density) for c in per_channel]))
        # the sweep point must stay where it was put: training would move the thresholds (r1: 0.01 requested -> 0.07
        # realised after 220 steps), so the sweep runs the identical kernels with a zero learning rate
        trainer.lr = 0.0

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # set-up, outside warm-up and timing whatever --warmup says: the first step records the kernel plan, the following ones
    # capture one hipGraph per input slot (with the collective inside when there are ranks)
    S = args.steps_per_graph
    if S <= 0:  # fewest left-over single steps, then the larger group
        S = min(range(5, 33), key=lambda g: (args.steps % g, -g)) if args.steps >= 5 else 1

    def run_steps(tr, n):
        """n optimizer steps, step i on input slot i % SLOTS; whole groups of S steps as one graph replay."""
        i, last = 0, None
        while S > 1 and i + S <= n:
            last = tr.step_many(tuple((i + j) % SLOTS for j in range(S)))[-1]  # falls back to single steps where it must
            i += S
        for j in range(i, n):
            last = tr.step(slot=j % SLOTS)
        return last  # mean loss of the last step (device scalar)

    def set_up(tr):
        for i in range(SLOTS + 1):
            tr.step(slot=i % SLOTS)
        if S > 1:
            run_steps(tr, S * SLOTS)  # captures the S-step graph of every slot rotation the loops below replay

    set_up(trainer)
    run_steps(trainer, args.warmup)
    sync()
    t0 = time.perf_counter()
    last_loss = run_steps(trainer, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_after = float(last_loss) if last_loss is not None else float("nan")
    n_mean, n_max = trainer.active_stats()
    bucket_hist = (torch.bincount(trainer.bucket_plan.bucket.long(), minlength=trainer.K).tolist()
                   if trainer.bucket_plan is not None else None)

    # ---- instrumented pass: per-entry-point durations from HIP events on the launch stream
    # FeatureTransformer kernel family: dense MFMA products, bit-mask/LDS-staged gather kernels or id-list kernels
    ftp = {"mfma": "nnue_ftm", "bits": "nnue_ftb", "list": "nnue_ft"}[trainer.ft_path]
    names = (["nnue_ftm_conv_binarize"] if trainer.use_mfma else
             ["nnue_conv3x3_forward", {"bits": "nnue_binarize_bits", "list": "nnue_binarize_features"}[trainer.ft_path]])
    fwd_entry = f"{ftp}_forward_l1" if getattr(trainer, "fuse_l1", False) else f"{ftp}_forward"  # fused: + layer-1 slabs in the epilogue
    if trainer.K > 1 and trainer.use_mfma:
        fwd_entry = f"{ftp}_forward_grouping"  # + the bucket grouping as one extra workgroup of the launch
    names += [fwd_entry] + (["nnue_classifier_train_step_bucketed"] if trainer.K > 1 else ["nnue_classifier_train_step"])
    # weight + value gradient (+ tail rows) go through one C call; it is one launch at launch-sized shapes and the two
    # separate launches at the 224x224 shapes (policy in nnue_ftm_backward)
    merged = trainer.use_mfma and trainer.merge_backward
    val_entry = "nnue_ftm_backward_values_ws" if trainer.use_mfma else f"{ftp}_backward_values"  # (the product form takes a workspace)
    bwd_entry = f"{ftp}_backward_bucketed" if (merged and trainer.K > 1) else f"{ftp}_backward"
    names += [bwd_entry] if merged else [f"{ftp}_backward_weight", val_entry]
    factor_exchange = bool(getattr(trainer, "factor_exchange", False))  # data parallel: the gradient's factors are gathered, same fused update
    fused_update = bool(getattr(trainer, "fuse_table_update", False)) or factor_exchange
    if fused_update:  # weight gradient formed and consumed in the update (no d_weight)
        names = [n for n in names if n not in (f"{ftp}_backward", f"{ftp}_backward_bucketed", f"{ftp}_backward_weight")]
        names += [val_entry, f"{ftp}_backward_tail_rows"] + (["nnue_dp_factor_pack", "nnue_dp_factor_unpack"] if factor_exchange else [])
        names += [f"{ftp}_gram_sqnorm", f"{ftp}_gram_sqnorm_tail", f"{ftp}_backward_weight_update"]
        names = list(dict.fromkeys(names))
        merged = False
    names += ["nnue_ste_conv_backward", "nnue_sgd_step"]
    # big table inside a step group: the update of a step also forms the next step's forward (one pass over the table); the
    # instrumented pass then issues the group's launches eagerly (step_many(timers=...)) instead of single steps
    group_fused = bool(getattr(trainer, "fuse_next_forward", False)) and S > 1
    upd_fwd = f"{ftp}_backward_weight_update_forward"
    if group_fused:
        names += [upd_fwd]
    timers = {k: [] for k in names}
    isteps = max(5, min(50, args.steps))
    if group_fused:
        isteps = max(S, isteps // S * S)
        group = lambda i0: tuple((i0 + j) % SLOTS for j in range(S))  # noqa: E731
        trainer.step_many(group(0), timers={k: [] for k in names})
    else:
        for i in range(3):
            trainer.step(slot=i % SLOTS, timers={k: [] for k in names})  # settle into eager mode
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    if group_fused:
        for i0 in range(0, isteps, S):
            trainer.step_many(group(i0), timers=timers)
    else:
        for i in range(isteps):
            trainer.step(slot=i % SLOTS, timers=timers)
    torch.cuda.synchronize(dev)
    eager_ms = (time.perf_counter() - t1) * 1e3 / isteps
    dur_us = {}
    for k, pairs in timers.items():
        # an entry point may be called more than once per step (stages / phases): report its time per step
        dur_us[k] = sum(a.elapsed_time(b) * 1e3 for a, b in pairs) / isteps if pairs else 0.0
    if group_fused:  # these run once per group / S-1 times per group: report them per launch
        for k in (upd_fwd, f"{ftp}_backward_weight_update", fwd_entry):
            if timers[k]:
                dur_us[k] = sum(a.elapsed_time(b) * 1e3 for a, b in timers[k]) / len(timers[k])

    row = cfg["l1"] * 4  # bytes of one gathered / accumulated table row
    alg = {  # algorithmic bytes per launch: fwd (n+1), value grad (n+1), weight grad n rows per image
        f"{ftp}_forward": (n_mean + 1) * row * B,
        val_entry: (n_mean + 1) * row * B,
        f"{ftp}_backward_weight": n_mean * row * B,
    }
    if merged:
        alg = {f"{ftp}_forward": alg[f"{ftp}_forward"], f"{ftp}_backward": alg[val_entry] + alg[f"{ftp}_backward_weight"]}
    if fwd_entry != f"{ftp}_forward":
        alg[fwd_entry] = alg.pop(f"{ftp}_forward")
    if merged and bwd_entry != f"{ftp}_backward":
        alg[bwd_entry] = alg.pop(f"{ftp}_backward")
    if fused_update:  # value gradient as its own launch; the weight gradient is formed inside the update product
        alg = {fwd_entry: (n_mean + 1) * row * B, val_entry: (n_mean + 1) * row * B,
               f"{ftp}_backward_weight_update": n_mean * row * B * (world if factor_exchange else 1)}
        if group_fused:  # + the next forward's gathered rows
            alg[upd_fwd] = n_mean * row * B + (n_mean + 1) * row * B
    kernels = {k: {"avg_us": round(dur_us[k], 2), **({"alg_GBps": round(alg[k] / dur_us[k] * 1e-3, 1)} if k in alg and dur_us[k] > 0 else {})}
               for k in names}
    KERNEL_OF = {  # C entry point -> (kernel name prefix, substring) in rocprof / PMC summaries
        "nnue_ftm_forward": ("ftm_gemm", "FwdEpi"), "nnue_ftm_forward_grouping": ("ftm_gemm", "FwdEpi"), "nnue_ftm_forward_l1": ("ftm_forward_l1", ""),
        "nnue_ftm_backward": ("ftm_backward", ""), "nnue_ftm_backward_bucketed": ("ftm_backward", ""),
        "nnue_ftm_backward_weight": ("ftm_gemm", "BwwEpi"), "nnue_ftm_backward_values": ("ftm_gemm", "ValEpi"), "nnue_ftm_backward_values_ws": ("ftm_gemm", "ValEpi"),
        "nnue_ftm_backward_weight_update": ("ftm_gemm", "BwwSgdEpi"), "nnue_ftm_backward_weight_update_forward": ("ftm_update_forward_kernel", ""),
        "nnue_ftb_forward": ("ftb_gather_kernel", ", 0,"), "nnue_ftb_backward_weight": ("ftb_gather_kernel", ", 1,"),
        "nnue_ftb_backward_values": ("ftb_values_kernel", ""), "nnue_ft_forward": ("ft_forward_wide", ""),
        "nnue_ft_backward_weight": ("ft_backward_weight_wide", ""), "nnue_ft_backward_values": ("ft_backward_values_wide", "")}

    def match_kernel(entry, name):
        want = KERNEL_OF.get(entry)
        if not want:
            return False
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        if want[1] == "BwwEpi" and "BwwSgdEpi" in name:
            return False
        return name.startswith(want[0]) and want[1] in name

    def pmc_traffic(entry):
        """HBM-side bytes per launch of the kernel behind a C entry point, from the committed rocprofv3 --pmc passes
        (profiles/*pmc_traffic.json, FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; separate passes per
        counter group).  Counters cannot be read inside this process, so workloads without a committed pass report null."""
        files = sorted((ROOT / "profiles").glob("*pmc_traffic.json"))
        if not files:
            return None
        data = json.loads(files[-1].read_text())
        kernels = data.get("workloads", {}).get(args.workload.replace("c3k1", "c3")) or (data.get("kernels") if args.workload == "c2" else None)
        if not kernels:
            return None
        for name, v in kernels.items():
            if match_kernel(entry, name):
                return {"bytes": v["hbm_bytes_corrected"], "l2_hit_rate": v["l2_hit_rate"], "kernel": name,
                        "source": f"profiles/{files[-1].name} (committed counter passes of the eager step; not collected in this run)"}
        return None

    def rocprof_avg_us(entry):
        """Average duration of that kernel in the latest committed `rocprofv3 --kernel-trace --stats` summary of this
        workload's bench command (profiles/*kernel_stats_<workload>.csv)."""
        files = sorted((ROOT / "profiles").glob(f"*kernel_stats_{args.workload.replace('c3k1', 'c3')}.csv"))
        if not files:
            return None
        import csv
        with open(files[-1]) as fh:
            for rec in csv.DictReader(fh):
                if match_kernel(entry, rec["Name"]) and int(rec["Calls"]) >= 10:
                    return {"us": round(float(rec["AverageNs"]) / 1e3, 2), "source": f"profiles/{files[-1].name}"}
        return None

    dom = max(alg, key=lambda k: dur_us[k])
    table_mb = model.input.weight.numel() * 4 / 1e6
    alg_rate = alg[dom] / (dur_us[dom] * 1e-6) / 1e9 if dur_us[dom] > 0 else 0.0
    if trainer.ft_path == "mfma":
        # Products over the whole map (active and inactive positions).  Each product is priced on the unit it runs on
        # (nnue_ftm_uses_bf16): the f32-input MFMA (157.3 TF) or, for the two products whose A operand is the binary map,
        # three bf16 MFMAs per logical product (2.5 PF dense; 3x the flops as matrix work).  Beside it the compulsory HBM
        # bytes of the launch; the bound reported is whichever ideal time is longer.
        from nnue_hip import lib as _lib
        direct = min(trainer.F - 1, trainer.P)
        L1, L2, F, P = cfg["l1"], cfg["l2"], trainer.F, trainer.P
        uses = lambda which: bool(_lib.load().nnue_ftm_uses_bf16(which, B, F, P, L1))  # noqa: E731
        f_fwd, f_w, f_v, f_l1 = 2.0 * B * direct * L1, 2.0 * B * direct * L1, 2.0 * B * P * L1, 2.0 * B * L1 * L2
        tbl, mp, act = direct * L1 * 4.0, float(B * P), B * L1 * 4.0
        work = {  # entry -> ([(unit, useful flops)], compulsory bytes)
            f"{ftp}_forward": ([("bf16" if uses(0) else "f32", f_fwd)], tbl + mp + act),
            f"{ftp}_forward_grouping": ([("bf16" if uses(0) else "f32", f_fwd)], tbl + mp + act),
            f"{ftp}_forward_l1": ([("bf16" if uses(0) else "f32", f_fwd), ("f32", f_l1)], tbl + mp + act + (L1 // 64) * B * L2 * 4.0),
            f"{ftp}_backward_weight": ([("bf16" if uses(1) else "f32", f_w)], mp + act + tbl),
            val_entry: ([("bf16x6" if uses(4) else "f32", f_v)], act + F * L1 * 4.0 + mp + B * P * 4.0),
            # (under the factor exchange the product contracts the GLOBAL batch: world x the flops, map and d_ft rows)
            f"{ftp}_backward_weight_update": ([("bf16", f_w * (world if factor_exchange else 1))],
                                              (mp + act) * (world if factor_exchange else 1) + 2 * tbl + (2 * tbl if OPT["momentum"] else 0)),
        }
        # the update of step t + the forward of step t+1 in one pass: both products, the table and its momentum read and written
        # once, both maps (the forward's own read of the table is what the fusion removes)
        work[upd_fwd] = ([("bf16", f_w), ("bf16", f_fwd)], 2 * mp + 2 * act + 2 * tbl + (2 * tbl if OPT["momentum"] else 0))
        bw = [("bf16" if uses(2) else "f32", f_w), ("bf16x6" if uses(5) else "f32", f_v)] + ([("f32", f_l1)] if getattr(trainer, "ride_dw1", False) else [])
        work[f"{ftp}_backward"] = (bw, 2 * mp + act + F * L1 * 4.0 + tbl + B * P * 4.0)
        work[f"{ftp}_backward_bucketed"] = work[f"{ftp}_backward"]
        PEAK = {"f32": MFMA_F32_PEAK_TFLOPS, "bf16": MFMA_BF16_PEAK_TFLOPS, "bf16x6": MFMA_BF16_PEAK_TFLOPS}
        PASSES = {"f32": 1, "bf16": 3, "bf16x6": 6}  # bf16 MFMAs per logical product: exact split of one operand / six plane products of two
        units, comp_bytes = work[dom]
        useful = sum(f for _, f in units)
        t_mfma = sum(f * PASSES[u] / (PEAK[u] * 1e12) for u, f in units)  # seconds at the units' peaks
        t_hbm = comp_bytes / (HBM_PEAK_GBS * 1e9)
        dur = dur_us[dom] * 1e-6
        rp = rocprof_avg_us(dom)
        common = {"kernel": dom, "traffic": (pmc_traffic(dom) or {}).get("bytes"), "traffic_detail": pmc_traffic(dom),
                  "compulsory_bytes": int(comp_bytes), "flops_per_launch": int(useful),
                  "matrix_work": [{"unit": u, "useful_flops": int(f), "peak_TFLOPs": PEAK[u], "mfma_flops": int(f * PASSES[u])}
                                  for u, f in units],
                  "ideal_us": {"mfma": round(t_mfma * 1e6, 2), "hbm": round(t_hbm * 1e6, 2)},
                  "avg_launch_us": round(dur_us[dom], 2), "rocprof_avg_us": rp["us"] if rp else None, "rocprof_source": rp["source"] if rp else None,
                  "alg_bytes_per_launch": int(alg[dom]), "alg_GBps": round(alg_rate, 1),
                  # SURVEY 8d's "one table row per gathered id" prices a gather; a product re-uses a staged row across the
                  # samples of a tile, so for this path the algorithmic-byte rate is NOT a lower bound on traffic (it exceeds
                  # the HBM peak while the table is cache-resident) -- the bound is the one reported in bound/peak/frac
                  "alg_is_bound": False}
        if t_hbm >= t_mfma:
            rate = comp_bytes / dur / 1e9 if dur > 0 else 0.0
            roofline = {"bound": "hbm", "achieved": round(rate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(rate / HBM_PEAK_GBS, 4),
                        # the same fraction with the kernel's average duration in the committed rocprofv3 summary instead of this run's events
                        "frac_rocprof": round(comp_bytes / (rp["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rp else None, **common,
                        "regime": "compulsory bytes of the launch (table %.1f MB streamed once, parameters and momentum read and written where the "
                                  "update is fused) / HIP-event duration; the matrix work of the launch would take %.1f us at its units' peaks"
                                  % (table_mb, t_mfma * 1e6)}
        else:
            blended = useful / t_mfma / 1e12  # the peak this launch could reach given which unit each product runs on
            achieved = useful / dur / 1e12 if dur > 0 else 0.0
            roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(blended, 1), "unit": "TFLOP/s",
                        "frac": round(achieved / blended, 4),
                        "frac_rocprof": round(useful / (rp["us"] * 1e-6) / 1e12 / blended, 4) if rp else None,
                        "frac_of_f32_mfma_peak": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),  # round 1's pricing of the same launch, for comparison
                        **common,
                        "regime": "useful flops of the products in the launch (2 M N K over the whole map) / HIP-event duration, against the "
                                  "peak of the unit each product runs on (f32-input MFMA 157.3 TF; bf16 MFMA 2.5 PF at three MFMAs per "
                                  "product for the exact split of one operand, six for the plane products of two f32 operands); table %.1f MB" % table_mb}
    else:
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(alg_rate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(alg_rate / HBM_PEAK_GBS, 4), "traffic": (pmc_traffic(dom) or {}).get("bytes"),
                    "traffic_detail": pmc_traffic(dom), "alg_is_bound": True,
                    "alg_bytes_per_launch": int(alg[dom]), "avg_launch_us": round(dur_us[dom], 2),
                    "regime": ("table %.1f MB is L2/Infinity-Cache resident: algorithmic rate is cache bandwidth and may exceed the HBM peak"
                               % table_mb) if table_mb < 200 else "table %.0f MB exceeds the Infinity Cache: HBM-bound" % table_mb}

    # ---- the same step with the LDS-staged gather kernels (the form SURVEY 8d prices against the HBM roofline), same
    # process, same synthetic batches: images/s and the algorithmic-byte rate of its dominant FT kernel
    # ---- bucketed workload: the same step on plain randn images (every sample lands in ONE layer stack), same process
    single_bucket = None
    if world == 1 and args.spread and cfg.get("buckets", 1) > 1 and args.density is None:
        torch.manual_seed(0)
        model_s = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"],
                            input_size=cfg["image"], num_ls_buckets=cfg["buckets"], clip_activations=cfg.get("clip")).to(dev)
        ts = NnueTrainer(model_s, B, (cfg["image"], cfg["image"]), group=None, use_graph=not args.no_graph, input_slots=SLOTS, **OPT)
        gs = torch.Generator().manual_seed(1234)
        for images, labels in ts.inputs:
            images.copy_(torch.randn(B, 3, cfg["image"], cfg["image"], generator=gs))
            labels.copy_(torch.randint(0, cfg["classes"], (B,), generator=gs))
        set_up(ts)
        run_steps(ts, max(5, min(args.warmup, 30)))
        torch.cuda.synchronize(dev)
        ts0 = time.perf_counter()
        run_steps(ts, args.steps)
        torch.cuda.synchronize(dev)
        s_elapsed = time.perf_counter() - ts0
        single_bucket = {"images_per_sec": round(B * args.steps / s_elapsed, 1), "ms_per_step": round(s_elapsed * 1e3 / args.steps, 4),
                         "samples_per_bucket_last_batch": torch.bincount(ts.bucket_plan.bucket.long(), minlength=ts.K).tolist()}
        del ts, model_s

    gather = None
    if world == 1 and trainer.ft_path == "mfma" and not args.no_gather_compare and os.environ.get("NNUE_FT_PATH", "auto") == "auto":
        os.environ["NNUE_FT_PATH"] = "bits"
        try:
            torch.manual_seed(0)
            model_g = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"],
                                num_classes=cfg["classes"], input_size=cfg["image"], num_ls_buckets=cfg.get("buckets", 1),
                                clip_activations=cfg.get("clip")).to(dev)
            if args.spread:
                with torch.no_grad():
                    model_g.conv.weight.abs_()
            if args.density is not None:
                with torch.no_grad():
                    model_g.visual_threshold.copy_(model.visual_threshold)
            tg = NnueTrainer(model_g, B, (cfg["image"], cfg["image"]), group=None, use_graph=not args.no_graph, input_slots=SLOTS, **OPT)
            if args.density is not None:
                tg.lr = 0.0
            if tg.ft_path == "bits":
                for (gi, gl), (si, sl) in zip(tg.inputs, trainer.inputs):
                    gi.copy_(si)
                    gl.copy_(sl)
                gsteps = max(10, min(args.steps, 100))
                set_up(tg)
                run_steps(tg, max(5, min(args.warmup, 20)))
                torch.cuda.synchronize(dev)
                tg0 = time.perf_counter()
                run_steps(tg, gsteps)
                torch.cuda.synchronize(dev)
                g_elapsed = time.perf_counter() - tg0
                gn_mean, _ = tg.active_stats()
                gnames = ["nnue_ftb_forward", "nnue_ftb_backward_weight", "nnue_ftb_backward_values"]
                gt = {k: [] for k in gnames}
                for i in range(3):
                    tg.step(slot=i % SLOTS, timers={k: [] for k in gnames})
                gi_steps = max(5, min(30, gsteps))
                for i in range(gi_steps):
                    tg.step(slot=i % SLOTS, timers=gt)
                torch.cuda.synchronize(dev)
                gdur = {k: sum(a.elapsed_time(b) * 1e3 for a, b in v) / gi_steps for k, v in gt.items()}
                galg = {"nnue_ftb_forward": (gn_mean + 1) * row * B, "nnue_ftb_backward_values": (gn_mean + 1) * row * B,
                        "nnue_ftb_backward_weight": gn_mean * row * B}
                gdom = max(galg, key=lambda k: gdur[k])
                grate = galg[gdom] / (gdur[gdom] * 1e-6) / 1e9 if gdur[gdom] > 0 else 0.0
                old = sorted((ROOT / "profiles").glob("r01g_pmc_traffic.json"))
                gtraffic = None
                if old and args.workload == "c2":
                    kk = json.loads(old[-1].read_text())["kernels"]
                    pat = {"nnue_ftb_forward": ("ftb_gather_kernel", ", 0,"), "nnue_ftb_backward_weight": ("ftb_gather_kernel", ", 1,"),
                           "nnue_ftb_backward_values": ("ftb_values_kernel", "")}[gdom]
                    for name, v in kk.items():
                        if name.startswith(pat[0]) and pat[1] in name:
                            gtraffic = v["hbm_bytes_corrected"]
                gather = {"path": "NNUE_FT_PATH=bits (LDS-staged gather kernels, ftb_kernels.hip)",
                          "images_per_sec": round(B * gsteps / g_elapsed, 1), "ms_per_step": round(g_elapsed * 1e3 / gsteps, 4),
                          "roofline": {"bound": "hbm", "kernel": gdom, "achieved": round(grate, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(grate / HBM_PEAK_GBS, 4), "traffic": gtraffic,
                                       "alg_bytes_per_launch": int(galg[gdom]), "avg_launch_us": round(gdur[gdom], 2),
                                       "regime": ("table %.1f MB is L2/Infinity-Cache resident: algorithmic rate is cache bandwidth and may "
                                                  "exceed the HBM peak" % table_mb) if table_mb < 200 else
                                                 "table %.0f MB exceeds the Infinity Cache: HBM-bound" % table_mb},
                          "kernels_us": {k: round(v, 2) for k, v in gdur.items()}}
            del tg, model_g
        finally:
            os.environ["NNUE_FT_PATH"] = "auto"

    if rank == 0:
        out = {
            "metric": "NNUE training images/sec (full step: fwd+bwd+clip+SGD)",
            "value": round(B * world * args.steps / elapsed, 1),
            "unit": "images/sec",
            "n_gpus": world,
            "n_ranks": dist.get_world_size() if dist.is_initialized() else 1,  # as RCCL saw it
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
            "timed_s": round(elapsed, 6),  # wall time of the timed region (K steps between the two barriers, max over ranks)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: NNUE {cfg['image']}x{cfg['image']} grid {cfg['grid']}x{cfg['grid']}x{cfg['fps']} "
                                   f"F={cfg['grid'] ** 2 * cfg['fps']} -> {cfg['l1']}/{cfg['l2']}/{cfg['l3']} -> {cfg['classes']}, "
                                   + (f"{cfg['buckets']} layer-stack buckets (selector: active-feature count), clipped ReLU [0, {cfg['clip']}] "
                                      f"(build extension, parity unpinned), " if cfg.get("buckets", 1) > 1 else "")
                                   + ("per-sample offset/gain spread, " if args.spread else "")
                                   + f"batch {B}/GPU, SGD m0.9 wd2e-4 clip1.0" + (f", thresholds set for density {args.density} and held (lr 0)" if args.density is not None else ""),
                       "global_batch": B * world, "parallelism": f"dp{world}", "launch": "eager" if args.no_graph else "hipGraph", "steps_per_graph": S,
                       "mean_active_features": round(n_mean, 1), "max_active_features": n_max,
                       "active_density": round(n_mean / ((cfg["image"] - 1) // max(1, (cfg["image"] - 1) // (cfg["grid"] - 1)) + 1) ** 2 / cfg["fps"], 4),
                       "eager_ms_per_step_instrumented": round(eager_ms, 4), "loss_after": round(loss_after, 4),
                       **({"num_ls_buckets": trainer.K, "samples_per_bucket_last_batch": bucket_hist} if bucket_hist is not None else {}),
                       # the same 8-stack model on plain randn images, where one stack takes the whole batch
                       **({"single_bucket_images_per_sec": single_bucket["images_per_sec"], "single_bucket": single_bucket}
                          if single_bucket is not None else {})},
            "roofline": roofline,
            "kernels": kernels,
        }
        if gather is not None:
            out["gather_path"] = gather
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
