"""Training step of the NNUE hot path on MI355X, one process per GPU.

What train.py:359-366 does per batch (zero_grad, forward, cross-entropy, backward, clip_grad_norm_,
SGD step) is one fixed sequence of HIP kernels here, working on preallocated buffers:

    conv -> binarise+compact -> FT gather-accumulate -> pairwise+classifier -> cross-entropy
         -> classifier backward -> FT weight gather-sum -> FT value gather-dot -> STE/conv backward
         -> [all-reduce of ONE flat gradient buffer over RCCL when world > 1] -> fused clip + SGD

* every trainable parameter (all but nnue2score, which never has a gradient) lives in one flat fp32
  buffer; the module's parameters are views into it, so state_dict()/serialize keep working.  Gradients
  and momentum have matching flat buffers -- the all-reduce is a single call on a single buffer
  (3.8 MB at the CIFAR configs: latency-bound on xGMI, so one message, not one per tensor);
* shapes are static, nothing synchronises with the host, so the local part of the step is captured into
  a hipGraph and replayed (the C ABI launches on torch's current stream);
* data parallel = each rank takes an equal slice of the global batch; mean-loss gradients are summed
  by the all-reduce and scaled by 1/world inside the SGD kernel, where the *global* gradient norm is
  clipped -- every rank therefore applies the identical update.  Default: ONE all-reduce of the whole
  flat gradient buffer after the local step (the value gradient now leaves in the same launch as the
  weight gradient, so only ~16 us of STE kernels could still hide a first bucket -- less than a second
  collective costs).  NNUE_DP_BUCKETS=2 keeps the earlier split: everything except the threshold /
  conv-weight gradients (99.98 % of the bytes) is all-reduced while the STE/conv backward kernels run,
  the 232-float tail bucket follows.

``FlatLayout`` and ``DataParallel`` are device-agnostic plumbing (covered by gloo tests on CPU);
``NnueTrainer`` is GPU-only.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import os

import torch
import torch.distributed as dist

from . import lib

SKIP = ("nnue2score",)  # no gradient ever reaches it (reference tests/test_model.py:179-182)


@dataclass
class FlatLayout:
    names: List[str]
    shapes: List[Tuple[int, ...]]
    offsets: List[int]
    count: int  # padded element count of the flat buffers

    @staticmethod
    def of(model: torch.nn.Module, align: int = 4, pad_to: int = 1) -> "FlatLayout":
        """pad_to: the flat buffers' length becomes a multiple of it (equal, float4-aligned shards per rank)."""
        names, shapes, offsets, off = [], [], [], 0
        for k, p in model.named_parameters():
            if k in SKIP:
                continue
            names.append(k)
            shapes.append(tuple(p.shape))
            offsets.append(off)
            off += (p.numel() + align - 1) // align * align  # 16-byte aligned starts for the float4 kernels
        return FlatLayout(names, shapes, offsets, (off + pad_to - 1) // pad_to * pad_to)

    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for k, s, o in zip(self.names, self.shapes, self.offsets):
            n = 1
            for d in s:
                n *= d
            out[k] = flat[o:o + n].view(s)
        return out

    def pack(self, tensors: Dict[str, torch.Tensor], device=None) -> torch.Tensor:
        some = next(iter(tensors.values()))
        flat = torch.zeros(self.count, dtype=torch.float32, device=device if device is not None else some.device)
        for k, v in self.views(flat).items():
            v.copy_(tensors[k])
        return flat


class DataParallel:
    """Batch sharding + the one collective of the step.  Backend: "nccl" (= RCCL over xGMI) on GPUs,
    "gloo" in the CPU tests."""

    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        # NNUE_DP_FORCE_COLLECTIVES=1 keeps the bucketed all-reduce path on even with one rank (rehearses the
        # RCCL + hipGraph interplay on a single-GPU box; a 1-rank all-reduce is the identity)
        self.collectives = self.enabled and (self.world > 1 or os.environ.get("NNUE_DP_FORCE_COLLECTIVES") == "1")
        self.buckets = 2 if os.environ.get("NNUE_DP_BUCKETS", "1") == "2" else 1
        self.backend = dist.get_backend(group) if self.enabled else None

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def shard(self, global_batch: int) -> slice:
        if global_batch % self.world:
            raise ValueError(f"global batch {global_batch} is not divisible by world size {self.world}")
        per = global_batch // self.world
        return slice(self.rank * per, (self.rank + 1) * per)

    def broadcast(self, flat: torch.Tensor) -> None:
        if self.world > 1:
            dist.broadcast(flat, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)

    def allreduce_sum(self, flat: torch.Tensor, async_op: bool = False):
        if self.collectives:
            return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return None

    # ---- sharded update (reduce-scatter the gradient, update one shard per rank, all-gather the parameters): the same
    # wire bytes as the all-reduce, 1/world of the optimizer's memory traffic per rank -- for flat buffers that are
    # bandwidth- rather than latency-sized (the 269 MB of the 224x224 configuration)
    def shard_of(self, flat: torch.Tensor) -> torch.Tensor:
        per = flat.numel() // self.world
        if per * self.world != flat.numel():
            raise ValueError("flat buffer length must be a multiple of the world size (FlatLayout.of(pad_to=...))")
        return flat[self.rank * per:(self.rank + 1) * per]

    def reduce_scatter_sum(self, flat: torch.Tensor, out_shard: torch.Tensor) -> None:
        """out_shard <- this rank's shard of the sum over ranks of `flat`."""
        if not self.collectives:
            out_shard.copy_(self.shard_of(flat))
        elif self.backend == "nccl":
            dist.reduce_scatter_tensor(out_shard, flat, op=dist.ReduceOp.SUM, group=self.group)
        else:  # gloo has no reduce-scatter: all-reduce, keep the own slice (CPU tests / rehearsals only)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            out_shard.copy_(self.shard_of(flat))

    def all_gather(self, flat_out: torch.Tensor, shard: torch.Tensor) -> None:
        """flat_out <- the ranks' shards side by side (shard may be flat_out's own slice)."""
        if not self.collectives:
            if shard.data_ptr() != self.shard_of(flat_out).data_ptr():
                self.shard_of(flat_out).copy_(shard)
        elif self.backend == "nccl":
            dist.all_gather_into_tensor(flat_out, shard, group=self.group)
        else:
            per = flat_out.numel() // self.world
            parts = [flat_out[r * per:(r + 1) * per] for r in range(self.world)]
            dist.all_gather(parts, shard.clone(), group=self.group)


    def all_gather_chunks(self, chunks: torch.Tensor) -> None:
        """chunks [world][n] (any dtype): row r <- rank r's row; every rank sends its own row in place."""
        if not self.collectives:
            return
        if self.backend == "nccl":
            dist.all_gather_into_tensor(chunks.view(-1), chunks[self.rank], group=self.group)
        else:
            dist.all_gather([chunks[r] for r in range(self.world)], chunks[self.rank].clone(), group=self.group)


class NnueTrainer:
    """Owns the flat buffers and the activation workspace for one (batch, H, W) shape."""

    def __init__(self, model, batch_size: int, image_hw: Tuple[int, int], lr: float, momentum: float = 0.0,
                 weight_decay: float = 0.0, max_grad_norm: float = 0.0, group=None, use_graph: bool = True,
                 input_slots: int = 1, optimizer: str = "sgd", betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8):
        lib.load()
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise lib.NnueHipError("NnueTrainer needs the model on the GPU (no CPU fallback)")
        self.model, self.dev = model, p0.device
        self._hyper = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        if optimizer not in ("sgd", "adam"):
            raise ValueError(f"optimizer must be 'sgd' or 'adam', got {optimizer!r}")
        self.optimizer, self.betas, self.eps = optimizer, betas, eps
        self.dp = DataParallel(group)
        self.B, (self.H, self.W) = batch_size, image_hw
        self.stride = int(model.conv.stride[0])
        self.fps = model.conv.out_channels
        self.F, self.L1 = model.input.weight.shape
        lin = model.classifier._linears()
        self.L2, self.L3, self.C = lin[0].out_features, lin[1].out_features, lin[2].out_features
        # bucketed layer stacks (BASELINE configs[2]): stacked [K, out, in] classifier parameters, one stack per sample
        self.K = int(getattr(model, "num_ls_buckets", 1))
        self.clip = float(model.classifier.clip_activations or 0.0)
        self.gh, self.gw = lib.conv_out_hw(self.H, self.W, self.stride)
        self.P = self.fps * self.gh * self.gw

        # ---- flat parameter / gradient / momentum buffers; module parameters become views
        self.layout = FlatLayout.of(model, pad_to=4 * self.dp.world)
        f32 = dict(dtype=torch.float32, device=self.dev)
        state = {k: p.detach() for k, p in model.named_parameters() if k not in SKIP}
        self.flat_params = self.layout.pack(state)
        self.flat_grads = torch.zeros(self.layout.count, **f32)
        self.flat_momentum = torch.zeros(self.layout.count, **f32) if (momentum and optimizer == "sgd") else None
        # Adam state (torch.optim.Adam semantics); the step number lives on the device so the update graph replays
        self.flat_exp_avg = torch.zeros(self.layout.count, **f32) if optimizer == "adam" else None
        self.flat_exp_avg_sq = torch.zeros(self.layout.count, **f32) if optimizer == "adam" else None
        self.adam_step_count = torch.zeros(1, dtype=torch.int32, device=self.dev) if optimizer == "adam" else None
        self.dp.broadcast(self.flat_params)  # identical replicas even if ranks were seeded differently
        self.p = self.layout.views(self.flat_params)
        self.g = self.layout.views(self.flat_grads)
        # gradient buckets: [threshold, conv weight] come last out of the backward and lead the flat buffer
        assert self.layout.names[:2] == ["visual_threshold", "conv.weight"]
        self.bucket_split = self.layout.offsets[2]
        for k, prm in model.named_parameters():
            if k not in SKIP:
                prm.data = self.p[k]
                prm.grad = self.g[k]

        # ---- static activations and scratch
        B = self.B
        # input ring: a loader fills slot k while slot k-1 trains; kernels read the slots in place
        self.inputs = [(torch.empty((B, 3, self.H, self.W), **f32), torch.empty((B,), dtype=torch.int64, device=self.dev))
                       for _ in range(max(1, input_slots))]
        self.images, self.labels = self.inputs[0]
        self.conv_out = torch.empty((B, self.fps, self.gh, self.gw), **f32)
        self.patches = None  # (set below for the product form at large strides)
        # binary features: float {0,1} map + dense MFMA products when the shape allows, else bit masks + LDS-staged
        # gather kernels, else id lists
        self.ft_path = lib.ft_path(self.F, self.P, self.L1, B)
        self.use_mfma, self.use_bits = self.ft_path == "mfma", self.ft_path == "bits"
        self.fm = lib.FeatureMatrix.empty(B, self.P, self.F, self.L1, self.dev) if self.use_mfma else None
        # Large strides (the taps of neighbouring positions do not overlap: 224x224 at stride 7 reads 3 of every 7 rows and the
        # backward would gather them again): the conv launch leaves the im2col form of the images -- 27 floats per position, 0.18
        # of the image bytes -- and does not write conv_out; the STE backward reads the patches and re-forms conv_out with the
        # forward's fmaf chain (bitwise).  NNUE_CONV_PATCHES=0|1|auto (auto: images more than twice their patches).
        pm = os.environ.get("NNUE_CONV_PATCHES", "auto")
        self.reform_conv_out = os.environ.get("NNUE_CONV_REFORM", "0") == "1"  # conv_out not written; the backward re-forms it
        self.use_patches = (self.use_mfma and self.fps <= 64 and pm != "0"
                            and (pm == "1" or 3 * self.H * self.W > 2 * 27 * self.gh * self.gw))
        if self.use_patches:
            self.patches = torch.empty((27, B * self.gh * self.gw), **f32)
        # Data parallel with a bandwidth-sized table: the ranks all-gather the FACTORS of the table's gradient (the map as
        # bits + d_ft, ~1.5 MB per rank at the 224x224 shape) instead of reducing the 269 MB product, and every rank runs the
        # fused single-rank path (Gram norm, update in the product's epilogue) on the global factors: no big collective, no
        # materialised d_W, bitwise identical replicas.  The other gradients travel in the same all-gather and are summed in
        # rank order (lib.FactorExchange).  NNUE_DP_FACTOR_EXCHANGE=auto|1|0; auto = tables of 32 MB or more.
        fx_mode = os.environ.get("NNUE_DP_FACTOR_EXCHANGE", "auto")
        fx_rows = min(self.F - 1, self.P)
        fx_off = self.layout.offsets[self.layout.names.index("input.weight")]
        gb = B * self.dp.world
        self.factor_exchange = (self.dp.collectives and self.use_mfma and optimizer == "sgd" and self.dp.buckets == 1 and fx_rows > 0
                                and self.layout.names[:3] == ["visual_threshold", "conv.weight", "input.weight"]
                                and gb * self.L1 <= (1 << 24) and ((gb + 15) // 16) * ((self.L1 + 15) // 16) <= 65536
                                and fx_off % 4 == 0 and (fx_rows * self.L1) % 4 == 0 and fx_mode != "0"
                                and (fx_mode == "1" or self.F * self.L1 * 4 >= (32 << 20)))
        self.fx = None
        if self.factor_exchange:
            tail_lo = fx_off + fx_rows * self.L1
            self.fx = lib.FactorExchange(self.dp.world, self.dp.rank, B, self.P, self.F, self.L1, head=fx_off, tail_lo=tail_lo,
                                         tail=self.layout.count - tail_lo, device=self.dev)
            self.fm.sink = self.fx.sink  # written by the binarise kernel straight into this rank's chunk
        self.bits = lib.FeatureBits.empty(B, self.P, self.F, self.L1, self.dev) if self.use_bits else None
        self.act = lib.ActList.empty(B, self.P, self.F, self.dev) if self.ft_path == "list" else None
        self.ft = torch.empty((B, self.L1), **f32)
        self.h1 = torch.empty((B, self.L2), **f32)
        self.h2 = torch.empty((B, self.L3), **f32)
        self.logits = torch.empty((B, self.C), **f32)
        self.sample_loss = torch.empty((B,), **f32)
        self.loss = torch.zeros((), **f32)
        self.loss_ring = torch.zeros((max(1, input_slots),), **f32)  # step_many's per-step mean losses
        self.d_logits = torch.empty((B, self.C), **f32)
        self.d_ft = self.fx.d_ft if self.fx is not None else torch.empty((B, self.L1), **f32)  # (the chunk's view: sent in place)
        self.d_conv_out = torch.empty((B, self.P), **f32)
        self.grad_norm = torch.zeros((), **f32)
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.cls_scratch = torch.empty((max(lib.classifier_scratch_bytes(B, self.L1, self.L2, self.L3, self.K),
                                            lib.classifier_train_scratch_bytes(B, self.L1, self.L2, self.L3, self.C, self.K)),), **u8)
        self.bucket_plan = lib.BucketPlan(B, self.K, self.dev) if self.K > 1 else None
        self.ste_scratch = torch.empty((max(16, lib.load().nnue_ste_conv_backward_scratch(B, self.fps, self.gh, self.gw)),), **u8)
        self.sgd_scratch = torch.empty((lib.sgd_scratch_bytes(self.layout.count),), **u8)
        # single rank + SGD: the second stage of the STE / conv-weight gradient sum rides in the optimizer's norm launch
        # (with collectives the gradients must be complete before the all-reduce)
        self.defer_ste = (not self.dp.collectives and optimizer == "sgd" and os.environ.get("NNUE_DEFER_STE", "1") != "0"
                          and self.fps * 28 <= 4096 and self.layout.names[:2] == ["visual_threshold", "conv.weight"])
        self.ste_chunks = lib.ste_conv_backward_chunks(B, self.fps, self.gh, self.gw)
        # single rank + SGD: the FT weight-gradient tiles leave their sums of squares, so the clip norm does not read
        # those rows of the flat gradient buffer again (268 MB at the 224x224 configuration)
        self.sq_partial, self.sq_range = None, None
        # K > 1: the first layer's weights differ per sample, so its product is the classifier's own grouped launch
        # (bucket-homogeneous MFMA tiles) instead of the FeatureTransformer forward's epilogue / backward rider
        self.fuse_l1 = (self.K == 1 and self.use_mfma and os.environ.get("NNUE_FUSE_L1", "1") != "0"
                        and lib.ftm_forward_l1_supported(B, self.F, self.P, self.L1, self.L2))
        # Sharded update for bandwidth-sized flat buffers under collectives (SGD): reduce-scatter, per-shard clip + SGD with
        # the norm assembled from all-gathered block partials, all-gather of the parameters.  NNUE_DP_SHARDED_UPDATE=0|1|auto
        mode = os.environ.get("NNUE_DP_SHARDED_UPDATE", "auto")
        self.sharded_update = (self.dp.collectives and optimizer == "sgd" and self.dp.buckets == 1 and mode != "0"
                               and not self.factor_exchange and (mode == "1" or self.layout.count * 4 >= (64 << 20)))
        if self.sharded_update:
            per = self.layout.count // self.dp.world
            self.grad_shard = torch.empty((per,), **f32)
            self.norm_parts = 256
            self.local_partials = torch.empty((self.norm_parts,), **f32)
            self.all_partials = torch.empty((self.norm_parts * self.dp.world,), **f32)
        # the collective(s) captured into the step graph (torch's NCCL path is capturable): one replay per step, no host
        # round trip between the local kernels, the exchange and the update.  NNUE_DP_CAPTURE=0 keeps the eager collective.
        self.capture_collectives = (self.dp.collectives and use_graph and self.dp.backend == "nccl" and self.dp.buckets == 1
                                    and os.environ.get("NNUE_DP_CAPTURE", "1") != "0")
        # the learning rate lives in a device scalar the optimizer kernels read: ``trainer.lr = x`` (a scheduler's hook) is one
        # tiny fill, the recorded plans and captured graphs stay valid (the other hyper-parameters are constants of the plans)
        self.lr_dev = torch.full((1,), float(lr), **f32)
        self.steps_done = 0
        self.use_graph = use_graph
        self._g_local, self._g_update = {}, None
        self._plan_local = self._plan_seg = self._plan_update_first = self._plan_update = None
        self._plan_slot = 0
        self._side = torch.cuda.Stream(device=self.dev) if use_graph else None
        self._s1 = torch.cuda.Stream(device=self.dev) if use_graph else None
        self._s2 = torch.cuda.Stream(device=self.dev) if use_graph else None
        # NNUE_GRAPH_BRANCHES=1 captures the off-critical-path segments on side streams.  Measured on MI355X /
        # ROCm 7.0 (C2): 0.230 ms/step forked vs 0.175 ms/step as one chain -- the graph's fork/join edges cost
        # more than the overlap returns, so the default is the linear chain.
        self.branch = os.environ.get("NNUE_GRAPH_BRANCHES", "0") == "1"
        # forked capture runs "ft_wgrad" beside "tail", which then must not depend on it
        self.merge_backward = not self.branch and os.environ.get("NNUE_FTM_SPLIT_BACKWARD", "0") != "1"
        # the classifier's first-layer weight gradient rides in the merged FeatureTransformer backward launch (a third
        # tile family reading d_z1 out of the classifier's scratch) where that launch is used
        self.ride_dw1 = (self.use_mfma and self.merge_backward and os.environ.get("NNUE_FTM_RIDE_DW1", "1") != "0"
                         and lib.ftm_backward_cw_supported(B, self.F, self.P, self.L1, self.L2))
        n_sq = lib.ftm_backward_sq_count(B, self.F, self.P, self.L1) if (self.use_mfma and self.merge_backward) else 0
        if (n_sq > 0 and not self.dp.collectives and optimizer == "sgd" and os.environ.get("NNUE_NORM_PARTIALS", "1") != "0"):
            off = self.layout.offsets[self.layout.names.index("input.weight")]
            rows = min(self.F - 1, self.P)
            if off % 4 == 0 and (rows * self.L1) % 4 == 0:
                self.sq_partial = torch.empty((n_sq,), **f32)
                self.sq_range = (off, off + rows * self.L1)
        # Big tables on a single rank with SGD: the FeatureTransformer weight gradient is never materialised.  Its squared
        # norm comes from two B x B Gram matrices (nnue_ftm_gram_sqnorm), the optimizer's norm/apply pass skips those rows
        # and leaves the clip coefficient in a device scalar, and the product d_W = A^T d_out runs LAST, applying the update
        # to the table in its epilogue (nnue_ftm_backward_weight_update): no 268 MB write + read at the 224x224 shape.
        self.fuse_table_update = False
        rows = min(self.F - 1, self.P)
        off = self.layout.offsets[self.layout.names.index("input.weight")]
        big_table = self.F * self.L1 * 4 >= (32 << 20)
        want = os.environ.get("NNUE_FUSE_TABLE_UPDATE", "auto")
        if (self.use_mfma and not self.dp.collectives and optimizer == "sgd" and rows > 0 and B * self.L1 <= (1 << 24) and off % 4 == 0
                and (rows * self.L1) % 4 == 0 and want != "0" and (big_table or want == "1")):
            self.fuse_table_update = True
            self.ride_dw1 = False  # the rider lives in the merged launch, which this path does not use
            self.sq_partial = torch.empty((int(lib.load().nnue_ftm_gram_sq_count(B, self.L1)),), **f32)
            self.sq_range = (off, off + rows * self.L1)
            self.gram = torch.zeros((lib.ftm_gram_scratch(self.fm),), **f32)
            self.clip_coef = torch.ones((), **f32)
        # With the fused table update ``model.input.weight.grad`` (a view of flat_grads) is never written: rows the product
        # covers stay at the zeros they were allocated with.  ``grads_materialised`` says so; callers that want the table's
        # gradient for logging or custom clipping set NNUE_FUSE_TABLE_UPDATE=0.
        if self.factor_exchange:  # the same fused update, on the global factors (B * world rows)
            self.ride_dw1 = False
            self.sq_partial = torch.empty((int(lib.load().nnue_ftm_gram_sq_count(gb, self.L1)),), **f32)
            self.sq_range = (fx_off, fx_off + fx_rows * self.L1)
            self.gram = torch.zeros((lib.ftm_gram_scratch(self.fx.g_fm),), **f32)
            self.clip_coef = torch.ones((), **f32)
        self.grads_materialised = not (self.fuse_table_update or self.factor_exchange)
        # Inside a step group (step_many) the table's update of step t and the FeatureTransformer forward of step t+1 are ONE
        # pass over the table (nnue_ftm_backward_weight_update_forward: the next batch is resident, its map only needs the conv
        # weights the small tensors' update has just written, and the update leaves every new table tile in registers) -- the
        # forward's own 268 MB read of the table disappears.  Two maps alternate (the update still reads step t's while step
        # t+1's is being written).  Bitwise the separate kernels.  NNUE_FUSE_NEXT_FORWARD=0 keeps them separate.
        # (Under the factor exchange the update contracts the all-gathered GLOBAL batch, the next forward this rank's own next map.)
        self.fuse_next_forward = ((self.fuse_table_update or self.factor_exchange) and not self.fuse_l1 and self.K == 1
                                  and os.environ.get("NNUE_FUSE_NEXT_FORWARD", "1") != "0"
                                  and lib.ftm_update_forward_supported(gb if self.factor_exchange else B, self.F, self.P, self.L1, B))
        self.fm_alt = lib.FeatureMatrix.empty(B, self.P, self.F, self.L1, self.dev) if self.fuse_next_forward else None
        if self.fm_alt is not None and self.fx is not None:
            self.fm_alt.sink = self.fx.sink  # (both local maps' sink counts live in this rank's chunk: only one of them is live at a time)
        self._last_alt = False  # the last step's map is the second one (an even-length step group)
        self.d_z1 = self.ft_rider = None
        if self.ride_dw1 and self.K == 1:
            off = lib.classifier_train_dz1_offset(B, self.L1, self.L2, self.L3, self.C, True)
            self.d_z1 = self.cls_scratch[off:off + B * self.L2 * 4].view(torch.float32).view(B, self.L2)
            self.ft_rider = self.ft
        elif self.ride_dw1:
            # bucketed stacks: the rider contracts each bucket's own rows, so its operands are the grouped-row copies
            # the classifier's per-sample kernel leaves in the scratch
            rows = self.bucket_plan.tiles * 16
            o_dz, o_x = lib.classifier_train_grouped_offsets(B, self.L1, self.L2, self.L3, self.C, self.K)
            self.d_z1 = self.cls_scratch[o_dz:o_dz + rows * self.L2 * 4].view(torch.float32).view(rows, self.L2)
            self.ft_rider = self.cls_scratch[o_x:o_x + rows * self.L1 * 4].view(torch.float32).view(rows, self.L1)
        # ... and so do the classifier's small gradients + mean loss (one more tile family of that launch; the classifier's d_x
        # launch then holds only d_x tiles).  NNUE_CLS_RIDE_SMALL=0 keeps them beside d_x.
        self.ride_small = self.ride_dw1 and os.environ.get("NNUE_CLS_RIDE_SMALL", "1") != "0"
        self._riders = {}  # mean-loss destination -> host struct (kept alive: recorded plans hold pointers to them)

    # ------------------------------------------------------------------ hyper-parameters
    # The learning rate is a device scalar (``lr_dev``) every optimizer kernel reads: changing it costs one fill and keeps
    # every plan and graph -- per-step schedules are fine; with ranks every rank must set the same value.  Momentum, weight
    # decay and the clip norm are constants of the recorded update plan (and of every graph captured from it): changing one
    # re-records that plan and drops the graphs that contain it; the local (forward/backward) plan is untouched.
    # ``trainer.lr = x`` and ``trainer.set_lr(x)`` are the same thing.
    def _set_hyper(self, key: str, value: float) -> None:
        if self._hyper[key] == value:
            return
        if key == "lr":
            self._hyper[key] = value
            self.lr_dev.fill_(value)
            return
        if key == "momentum" and self.optimizer == "sgd" and bool(value) != (self.flat_momentum is not None):
            raise ValueError("momentum cannot be switched on or off after construction (the buffer layout is fixed)")
        self._hyper[key] = value
        self._plan_update_first = self._plan_update = None
        self._g_update = None
        for key_ in [k for k in self._g_local if k[1] in ("full", "full_dp", "many")]:  # every graph that contains an update
            del self._g_local[key_]

    lr = property(lambda self: self._hyper["lr"], lambda self, v: self._set_hyper("lr", float(v)))
    momentum = property(lambda self: self._hyper["momentum"], lambda self, v: self._set_hyper("momentum", float(v)))
    weight_decay = property(lambda self: self._hyper["weight_decay"], lambda self, v: self._set_hyper("weight_decay", float(v)))
    max_grad_norm = property(lambda self: self._hyper["max_grad_norm"], lambda self, v: self._set_hyper("max_grad_norm", float(v)))

    def set_lr(self, lr: float) -> None:
        """Learning rate for the following steps (a scheduler's hook)."""
        self.lr = lr

    # ------------------------------------------------------------------ kernel sequences
    def _cls_params(self):
        p = self.p
        return [p[f"classifier.classifier.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias")]

    # The step is cut into segments so that a captured graph can run the ones nothing waits for beside the
    # critical path (conv -> bits -> FT forward -> classifier activations/d_x -> value gradient -> STE):
    #   front      conv, per-sample bit masks + forward tile lists                       main
    #   transpose  transposed masks + backward tile lists (only the FT weight gradient reads them)   side 1
    #   forward    FT forward, classifier phase 1 (activations, loss per sample, d_ft)  main
    #   ft_wgrad   FT weight/bias gradient                                              side 1
    #   cls_wgrad  classifier weight/bias gradients + mean loss                         side 2
    #   tail       FT value gradient -> d(conv_out) -> threshold / conv-weight gradients main  (bucket "b")
    SEGMENTS = ("front", "transpose", "forward", "ft_wgrad", "cls_wgrad", "tail")

    def _cls_step(self, phases: int) -> None:
        g = self.g
        cls_grads = tuple(g[f"classifier.classifier.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias"))
        lib.classifier_train_step(self.ft, True, *self._cls_params(), self.labels, 1.0, self.clip, scratch=self.cls_scratch,
                                  out=(self.h1, self.h2, self.logits), loss_out=(self.sample_loss, self.loss),
                                  grads=cls_grads, d_x=self.d_ft, phases=phases, buckets=self.bucket_plan)

    def _rider(self, loss: torch.Tensor):
        """The small-gradient tile family's arguments with `loss` as the mean loss's destination (one host struct per
        destination: step_many gives every step of a group its own)."""
        key = loss.data_ptr()
        if key not in self._riders:
            g = self.g
            grads = tuple(g[f"classifier.classifier.{i}.{n}"] for i in (0, 2, 4) for n in ("weight", "bias"))
            self._riders[key] = lib.classifier_train_rider(True, self.B, self.L1, self.L2, self.L3, self.C, self.h1, self.h2,
                                                           self.sample_loss, loss, grads, self.cls_scratch, self.bucket_plan)
        return self._riders[key]

    def _segment(self, name: str) -> None:
        p, g = self.p, self.g
        if name == "front":
            if self.use_mfma:  # conv + {0,1} map + counts in one launch
                lib.ftm_conv_binarize(self.images, p["conv.weight"], p["visual_threshold"], self.stride, self.F, self.L1,
                                      conv_out=self.conv_out, fm=self.fm, patches=self.patches, write_conv_out=not (self.use_patches and self.reform_conv_out))
                return
            lib.conv3x3_forward(self.images, p["conv.weight"], self.stride, out=self.conv_out)
            if self.use_bits:
                lib.binarize_bits(self.conv_out, p["visual_threshold"], self.F, self.L1, bits=self.bits, stages=1)
            else:
                lib.binarize_features(self.conv_out, p["visual_threshold"], self.F, act=self.act)
        elif name == "transpose":
            if self.use_bits:
                lib.binarize_bits(self.conv_out, p["visual_threshold"], self.F, self.L1, bits=self.bits, stages=2)
        elif name == "forward":
            if self.bucket_plan is not None and not self.use_mfma:  # stack of each sample from the counts the binarise kernel just wrote
                feats = self.bits if self.use_bits else self.act
                lib.bucket_group(feats.n, self.P, self.K, plan=self.bucket_plan)
            # (product form: the grouping rides in the FeatureTransformer forward launch as one extra workgroup)
            if self.use_mfma and self.fuse_l1:
                # the forward's epilogue also forms the classifier's layer-1 slabs (start of its scratch)
                lib.ftm_forward_l1(p["input.weight"], p["input.bias"], self.fm, p["classifier.classifier.0.weight"], self.cls_scratch,
                                   out=self.ft)
                self._cls_step((59 if self.ride_small else 27) if self.ride_dw1 else 13)  # 27: both phases, d_w1 left to the merged backward; + 32: the small gradients too
                return
            if self.use_mfma:
                lib.ftm_forward(p["input.weight"], p["input.bias"], self.fm, out=self.ft, group=self.bucket_plan)
            elif self.use_bits:
                lib.ftb_forward(p["input.weight"], p["input.bias"], self.bits, out=self.ft)
            else:
                lib.ft_forward(p["input.weight"], p["input.bias"], self.act, out=self.ft)
            self._cls_step((51 if self.ride_small else 19) if self.ride_dw1 else 5)  # 5: + the first-layer weight product beside d_x
        elif name == "ft_wgrad":
            if self.factor_exchange:
                # this rank's share of the rows the product does not cover; they travel with the small gradients, the
                # norm and the product wait for the global factors (_exchange_and_update)
                lib.ftm_backward_tail_rows(self.d_ft, self.fm, g["input.weight"], g["input.bias"])
            elif self.fuse_table_update:
                # rows the product does not cover (bias, clamp-sink row, unreachable rows) + the Gram form of the norm;
                # the product itself is part of the update (_update)
                # (the tail rows' workgroups ride in the Gram product's launch)
                lib.ftm_gram_sqnorm(self.fm, self.d_ft, self.gram, self.sq_partial, tail=(g["input.weight"], g["input.bias"]))
            elif self.use_mfma and self.merge_backward:
                # weight gradient, value gradient and tail rows share one launch (independent work, all read d_ft)
                lib.ftm_backward(self.d_ft, p["input.weight"], self.fm, d_weight=g["input.weight"], d_bias=g["input.bias"],
                                 dst=self.d_conv_out, ft=self.ft_rider, d_z1=self.d_z1,
                                 d_w1=g["classifier.classifier.0.weight"] if self.ride_dw1 else None, sq_partial=self.sq_partial,
                                 buckets=self.bucket_plan, small=self._rider(self.loss) if self.ride_small else None)
            elif self.use_mfma:
                lib.ftm_backward_weight(self.d_ft, self.fm, d_weight=g["input.weight"], d_bias=g["input.bias"])
            elif self.use_bits:
                lib.ftb_backward_weight(self.d_ft, self.bits, d_weight=g["input.weight"], d_bias=g["input.bias"])
            else:
                lib.ft_backward_weight(self.d_ft, self.act, self.F, d_weight=g["input.weight"], d_bias=g["input.bias"])
        elif name == "cls_wgrad":
            if not self.ride_dw1:  # else the small gradients already rode in the d_x launch of "forward"
                self._cls_step(6)
        elif name == "tail":
            if self.use_mfma and self.merge_backward and not self.fuse_table_update and not self.factor_exchange:
                pass  # d_conv_out came out of the merged launch in "ft_wgrad"
            elif self.use_mfma:
                lib.ftm_backward_values(self.d_ft, p["input.weight"], self.fm, dst=self.d_conv_out)
            elif self.use_bits:
                lib.ftb_backward_values(self.d_ft, p["input.weight"], self.bits, dst=self.d_conv_out)
            else:
                lib.ft_backward_values(self.d_ft, p["input.weight"], self.act, self.P, dst=self.d_conv_out)
            if self.use_patches:
                lib.ste_conv_backward_patches(self.patches, p["conv.weight"], p["visual_threshold"], self.d_conv_out, self.gh, self.gw,
                                              d_thr=g["visual_threshold"], d_weight=g["conv.weight"], scratch=self.ste_scratch,
                                              stages=1 if self.defer_ste else 3, conv_out=None if self.reform_conv_out else self.conv_out)
            else:
                lib.ste_conv_backward(self.images, self.conv_out, p["visual_threshold"], self.d_conv_out, self.stride,
                                      d_thr=g["visual_threshold"], d_weight=g["conv.weight"], scratch=self.ste_scratch,
                                      stages=1 if self.defer_ste else 3)
        else:
            raise KeyError(name)

    def _forward(self) -> None:
        for name in ("front", "forward"):
            self._segment(name)

    def _local_step(self) -> None:
        """forward + loss + backward into the flat gradient buffer (every element is overwritten)."""
        for name in self.SEGMENTS:
            self._segment(name)

    def _update(self, first: bool, grad_scale: Optional[float] = None) -> None:
        scale = self.dp.grad_scale if grad_scale is None else grad_scale
        if self.optimizer == "adam":
            lib.adam_step(self.flat_params, self.flat_grads, self.flat_exp_avg, self.flat_exp_avg_sq, self.adam_step_count,
                          self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm, scale, self.grad_norm,
                          self.sgd_scratch, lr_dev=self.lr_dev)
        else:
            ste = ((self.ste_scratch, self.ste_chunks, self.fps, self.g["visual_threshold"], self.g["conv.weight"])
                   if self.defer_ste else None)
            lib.sgd_step(self.flat_params, self.flat_grads, self.flat_momentum, self.lr, self.momentum, self.weight_decay,
                         self.max_grad_norm, scale, first, self.grad_norm, self.sgd_scratch, ste=ste,
                         ext=(self.sq_partial, *self.sq_range) if self.sq_partial is not None else None,
                         coef_out=self.clip_coef if self.fuse_table_update else None,
                         ext_applied_elsewhere=self.fuse_table_update, lr_dev=self.lr_dev)
            if self.fuse_table_update:
                lo, hi = self.sq_range
                mom = self.flat_momentum[lo:hi] if self.flat_momentum is not None else None
                lib.ftm_backward_weight_update(self.d_ft, self.fm, self.p["input.weight"], mom, self.clip_coef, self.lr, self.momentum,
                                               self.weight_decay, scale, first, lr_dev=self.lr_dev)

    def _exchange_and_update(self, first: bool, grad_scale: Optional[float] = None, alt: bool = False,
                             next_slot: Optional[int] = None) -> None:
        """Everything after the local kernels of a data-parallel step, as launches on the current stream: the gradient
        exchange and the optimizer.  Capturable (no host synchronisation).  alt: this step's local map is the second one;
        next_slot (step groups under the factor exchange): the table update also forms the forward of the step on that slot."""
        if self.factor_exchange:
            fx, scale = self.fx, (self.dp.grad_scale if grad_scale is None else grad_scale)
            fx.pack(self.fm_alt if alt else self.fm, self.flat_grads)
            self.dp.all_gather_chunks(fx.chunks)  # the step's one collective
            fx.unpack(self.flat_grads)
            lib.ftm_gram_sqnorm(fx.g_fm, fx.g_dft, self.gram, self.sq_partial)
            lib.sgd_step(self.flat_params, self.flat_grads, self.flat_momentum, self.lr, self.momentum, self.weight_decay,
                         self.max_grad_norm, scale, first, self.grad_norm, self.sgd_scratch, ext=(self.sq_partial, *self.sq_range),
                         coef_out=self.clip_coef, ext_applied_elsewhere=True, lr_dev=self.lr_dev)
            lo, hi = self.sq_range
            mom = self.flat_momentum[lo:hi] if self.flat_momentum is not None else None
            if next_slot is not None:
                nxt = self.fm if alt else self.fm_alt
                lib.ftm_conv_binarize(self.inputs[next_slot][0], self.p["conv.weight"], self.p["visual_threshold"], self.stride, self.F, self.L1,
                                      conv_out=self.conv_out, fm=nxt, patches=self.patches,
                                      write_conv_out=not (self.use_patches and self.reform_conv_out))
                lib.ftm_backward_weight_update_forward(fx.g_dft, fx.g_fm, self.p["input.weight"], mom, self.clip_coef, self.lr, self.momentum,
                                                       self.weight_decay, scale, first, nxt, self.p["input.bias"], self.ft, lr_dev=self.lr_dev)
                return
            lib.ftm_backward_weight_update(fx.g_dft, fx.g_fm, self.p["input.weight"], mom, self.clip_coef, self.lr, self.momentum,
                                           self.weight_decay, scale, first, lr_dev=self.lr_dev)
            return
        if not self.sharded_update:
            self.dp.allreduce_sum(self.flat_grads, async_op=False)
            self._update(first, grad_scale)
            return
        dp = self.dp
        scale = dp.grad_scale if grad_scale is None else grad_scale
        dp.reduce_scatter_sum(self.flat_grads, self.grad_shard)
        lib.sqnorm_partials(self.grad_shard, self.local_partials)
        dp.all_gather(self.all_partials, self.local_partials)  # rank-major, fixed order: every rank forms the identical norm
        mom = dp.shard_of(self.flat_momentum) if self.flat_momentum is not None else None
        lib.sgd_step(dp.shard_of(self.flat_params), self.grad_shard, mom, self.lr, self.momentum, self.weight_decay, self.max_grad_norm,
                     scale, first, self.grad_norm, self.sgd_scratch, ext=(self.all_partials, 0, self.grad_shard.numel()), lr_dev=self.lr_dev)
        dp.all_gather(self.flat_params, dp.shard_of(self.flat_params))

    def _ranks_agree(self, ok: bool) -> bool:
        """True only when `ok` holds on EVERY rank (an eager MIN all-reduce; every rank reaches it at the same step).  A step
        graph with the collective inside must exist on all ranks or on none: a rank that replays while another issues the
        eager collective would pair different operations."""
        if self.dp.world <= 1:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.dev if self.dp.backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.dp.group)
        return bool(int(flag.item()))

    def _optimizer_buffers(self):
        return [t for t in (self.flat_momentum, self.flat_exp_avg, self.flat_exp_avg_sq, self.adam_step_count) if t is not None]

    def _capture(self, fn) -> torch.cuda.CUDAGraph:
        """Captures fn(main_stream) into a graph; fn may fork work onto self._s1 / self._s2 with events."""
        graph = torch.cuda.CUDAGraph()
        self._side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self._side):
            # thread_local: the collective's watchdog thread may query events while we capture
            with torch.cuda.graph(graph, stream=self._side, capture_error_mode="thread_local"):
                fn(torch.cuda.current_stream(self.dev))
        torch.cuda.current_stream(self.dev).wait_stream(self._side)
        return graph

    def _run_local(self, slot: int, part: str, main: torch.cuda.Stream, branch: bool, timers=None, loss=None, alt: bool = False,
                   forward_done: bool = False) -> None:
        """Launches the local segments of `part` ("all" | "a" | "b").  With `branch` (graph capture) the transposed
        lists, the FT weight gradient and the classifier weight gradients go to side streams and are joined at
        the end; otherwise everything is issued in order on `main`.  `loss`: a device scalar that receives the mean loss
        instead of ``self.loss`` (step_many's per-step results)."""
        seg = lambda name: self._seg_plan(slot, name, loss, alt)  # noqa: E731
        m = main.cuda_stream
        if forward_done:
            # step_many with the next forward fused into the previous step's table update: this step's map and FeatureTransformer
            # output already exist (alt: in the second map)
            assert part == "all" and not branch
            lib.run_plan([c for c in seg("forward") if c[0] != "nnue_ftm_forward"], m, timers)
            for name in ("ft_wgrad", "cls_wgrad", "tail"):
                lib.run_plan(seg(name), m, timers)
            return
        if part == "b":
            lib.run_plan(seg("tail"), m, timers)
            return
        if not branch:
            for name in ("front", "transpose", "forward", "ft_wgrad", "cls_wgrad"):
                lib.run_plan(seg(name), m, timers)
            if part == "all":
                lib.run_plan(seg("tail"), m, timers)
            return
        s1, s2 = self._s1, self._s2
        lib.run_plan(seg("front"), m)
        e0 = torch.cuda.Event()
        e0.record(main)
        s1.wait_event(e0)
        lib.run_plan(seg("transpose"), s1.cuda_stream)
        lib.run_plan(seg("forward"), m)
        e1 = torch.cuda.Event()
        e1.record(main)
        s1.wait_event(e1)
        lib.run_plan(seg("ft_wgrad"), s1.cuda_stream)
        s2.wait_event(e1)
        lib.run_plan(seg("cls_wgrad"), s2.cuda_stream)
        if part == "all":
            lib.run_plan(seg("tail"), m)
        main.wait_stream(s1)
        main.wait_stream(s2)

    def _plans(self, slot: int = 0):
        """Records the three fixed call sequences once (this also executes them once; the update plans are
        recorded on throw-away copies of the buffers' contents, which are restored afterwards).  The local plan is
        recorded on the input slot of the step that triggers it (no other slot is read or written)."""
        if self._plan_local is None:
            self._plan_seg = {}
            self._plan_slot = slot
            self.images, self.labels = self.inputs[slot]
            for name in self.SEGMENTS:
                with lib.record_calls() as calls:
                    self._segment(name)
                self._plan_seg[name] = list(calls)
            self._plan_local = [c for name in self.SEGMENTS for c in self._plan_seg[name]]
        if self._plan_update is None and (self.sharded_update or self.factor_exchange):
            self._plan_update_first = self._plan_update = []  # these modes issue exchange + update themselves (_exchange_and_update)
        if self._plan_update is None:
            live = [self.flat_params, self.flat_grads, self.grad_norm] + self._optimizer_buffers()
            keep = [t.clone() for t in live]  # recording executes the updates: put everything back afterwards
            with lib.record_calls() as calls:
                self._update(True)
            self._plan_update_first = list(calls)
            with lib.record_calls() as calls:
                self._update(False)
            self._plan_update = list(calls)
            for t, k in zip(live, keep):
                t.copy_(k)
        return self._plan_local, self._plan_update_first, self._plan_update

    def _alt_swap(self) -> dict:
        """Pointer substitutions that make a recorded call use the second map (fuse_next_forward)."""
        a, b = self.fm, self.fm_alt
        return {a.bits.data_ptr(): b.bits.data_ptr(), a.n.data_ptr(): b.n.data_ptr(), a.sink.data_ptr(): b.sink.data_ptr(),
                a.scratch.data_ptr(): b.scratch.data_ptr()}

    def _seg_plan(self, slot: int, name: str, loss: Optional[torch.Tensor] = None, alt: bool = False):
        """The recorded calls of one segment with the recorded slot's input pointers swapped for `slot`'s (and the mean
        loss's destination for `loss`; `alt`: the map's buffers for the second map's)."""
        plan = self._plan_seg[name]
        src = self._plan_slot
        if slot == src and loss is None and not alt:
            return plan
        swap = {self.inputs[src][0].data_ptr(): self.inputs[slot][0].data_ptr(),
                self.inputs[src][1].data_ptr(): self.inputs[slot][1].data_ptr()}
        if alt:
            swap.update(self._alt_swap())
        rider_swap = None
        if loss is not None:
            swap[self.loss.data_ptr()] = loss.data_ptr()
            if self.ride_small:  # the mean loss's destination sits inside the rider's host struct: use the struct made for `loss`
                rider_swap = (ctypes.addressof(self._rider(self.loss)), ctypes.pointer(self._rider(loss)))

        def sub(a):
            if isinstance(a, int):
                return swap.get(a, a)
            if rider_swap is not None and hasattr(a, "contents") and isinstance(a.contents, lib.NnueClsRider) \
                    and ctypes.addressof(a.contents) == rider_swap[0]:
                return rider_swap[1]
            return a
        return [(nm, fn, tuple(sub(a) for a in args)) for nm, fn, args in plan]

    # ------------------------------------------------------------------ public
    def step(self, images: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None, slot: int = 0,
             timers=None, global_count: Optional[int] = None) -> torch.Tensor:
        """One optimizer step on this rank's slice.  Returns the local mean loss (device scalar, no sync).

        A short batch (fewer than the B images the trainer was built for; possibly none on some ranks) is padded; with
        more than one rank pass ``global_count`` = the number of real samples over ALL ranks in this step (the loader knows
        it), so that every rank scales the summed gradient by B / global_count -- the exact mean over the real samples.

        ``images``/``labels`` are copied into input slot ``slot`` first; pass None to train on what the slot
        already holds (zero-copy: fill ``trainer.inputs[slot]`` directly).  With ``timers`` ({C entry point:
        list}) the step runs eagerly from the recorded plan and brackets the named calls with HIP events on the
        launch stream (bench.py's per-kernel durations)."""
        buf_images, buf_labels = self.inputs[slot]
        ragged = None
        if images is not None:
            if labels is None or images.shape[1:] != buf_images.shape[1:] or labels.shape[0] != images.shape[0] \
                    or not 0 <= images.shape[0] <= self.B or (images.shape[0] == 0 and self.dp.world == 1):
                raise ValueError(f"trainer was built for images {tuple(buf_images.shape)} / labels ({self.B},)")
            n = images.shape[0]
            if n == self.B:
                buf_images.copy_(images, non_blocking=True)
                buf_labels.copy_(labels, non_blocking=True)
            else:
                # short last batch of an epoch: pad with blank images whose label -1 the loss kernel ignores (zero
                # loss, zero gradient row); the kernels divide by B, the exact mean over the n real samples is
                # restored by scaling gradients and loss with B/n
                if self.dp.world > 1 and global_count is None:
                    raise ValueError("a short batch with more than one rank needs global_count (real samples over all ranks)")
                ragged = self.B / n if self.dp.world == 1 else self.B * self.dp.world / int(global_count)
                buf_images.zero_()
                buf_images[:n].copy_(images, non_blocking=True)
                buf_labels.fill_(-1)
                buf_labels[:n].copy_(labels, non_blocking=True)
        if ragged is None and global_count is not None and self.dp.world > 1 and int(global_count) != self.B * self.dp.world:
            ragged = self.B * self.dp.world / int(global_count)  # this rank's slot is full but others are short: the same global mean
        _, upd_first, upd = self._plans(slot)
        self._last_alt = False
        first = self.steps_done == 0
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        graphs = self.use_graph and timers is None

        two_buckets = self.dp.collectives and self.dp.buckets == 2
        parts = ("a", "b") if two_buckets else ("all",)
        main = torch.cuda.current_stream(self.dev)
        if graphs:  # capture everything this step replays before any collective is enqueued
            for part in parts:
                if (slot, part) not in self._g_local:
                    self._g_local[(slot, part)] = self._capture(
                        lambda st, part=part: self._run_local(slot, part, st, branch=self.branch))
            if self._g_update is None and upd:
                self._g_update = self._capture(lambda st: lib.run_plan(upd, st.cuda_stream))

        def run(part):
            if graphs:
                self._g_local[(slot, part)].replay()
            else:
                self._run_local(slot, part, main, branch=False, timers=timers)

        if graphs and self.capture_collectives and not first and ragged is None:
            # data parallel, steady state: local kernels + exchange + update are ONE graph
            if (slot, "full_dp") not in self._g_local:
                def full_dp(st):
                    self._run_local(slot, "all", st, branch=False)
                    self._exchange_and_update(False)
                captured = True
                try:
                    self._g_local[(slot, "full_dp")] = self._capture(full_dp)
                except RuntimeError as exc:  # a stack that cannot capture its collectives: keep them eager from here on
                    import warnings
                    warnings.warn(f"collectives could not be captured into the step graph ({exc}); using the eager collective")
                    captured = False
                    torch.cuda.synchronize(self.dev)
                if not self._ranks_agree(captured):  # all ranks or none
                    self._g_local.pop((slot, "full_dp"), None)
                    self.capture_collectives = False
            if self.capture_collectives:
                self._g_local[(slot, "full_dp")].replay()
                self.steps_done += 1
                return self.loss
        if graphs and not self.dp.collectives and not first and ragged is None:
            # single rank, steady state: local step + update are ONE graph (one replay per step)
            if (slot, "full") not in self._g_local:
                def full(st):
                    self._run_local(slot, "all", st, branch=self.branch)
                    lib.run_plan(upd, st.cuda_stream)
                self._g_local[(slot, "full")] = self._capture(full)
            self._g_local[(slot, "full")].replay()
            self.steps_done += 1
            return self.loss
        if not self.dp.collectives:
            run("all")
        elif self.sharded_update or self.factor_exchange:
            run("all")
            with lib.time_calls(timers):
                self._exchange_and_update(first, grad_scale=self.dp.grad_scale * ragged if ragged is not None else None)
            self.steps_done += 1
            return self.loss * ragged if ragged is not None else self.loss
        elif not two_buckets:
            # one message: the whole flat gradient buffer, after the local graph; the wait is a stream dependency
            run("all")
            # a blocking (for the stream, not the host) collective: measured 25 us/step cheaper than async_op + wait()
            # in the 1-rank RCCL rehearsal (0.1395 vs 0.1643 ms); NNUE_DP_ASYNC_OP=1 restores the latter
            if os.environ.get("NNUE_DP_ASYNC_OP") == "1":
                self.dp.allreduce_sum(self.flat_grads, async_op=True).wait()
            else:
                self.dp.allreduce_sum(self.flat_grads, async_op=False)
        else:
            # big bucket is complete after part a: its all-reduce runs on the collective's own stream while part b
            # (STE/conv backward) still computes; the tail bucket follows part b
            run("a")
            big = self.dp.allreduce_sum(self.flat_grads[self.bucket_split:], async_op=True)
            run("b")
            small = self.dp.allreduce_sum(self.flat_grads[:self.bucket_split], async_op=True)
            big.wait()
            small.wait()
        if ragged is not None:
            self._update(first, grad_scale=self.dp.grad_scale * ragged)
            self.steps_done += 1
            return self.loss * ragged
        if first:
            lib.run_plan(upd_first, stream, timers)
        elif graphs:
            self._g_update.replay()
        else:
            lib.run_plan(upd, stream, timers)
        self.steps_done += 1
        return self.loss

    def _run_many(self, st: torch.cuda.Stream, slots, ring: torch.Tensor, upd, timers=None) -> None:
        """The launches of ``len(slots)`` consecutive steps on stream `st` (the current stream): what step_many captures, and
        -- with `timers` -- what it runs eagerly with HIP events around the named entry points."""
        fuse = self.fuse_next_forward and len(slots) > 1 and (not self.dp.collectives or self.factor_exchange)
        for i, s in enumerate(slots):
            alt = fuse and i % 2 == 1
            self._run_local(s, "all", st, branch=False, timers=timers, loss=ring[i], alt=alt, forward_done=fuse and i > 0)
            if self.dp.collectives:
                with lib.time_calls(timers):
                    self._exchange_and_update(False, alt=alt, next_slot=slots[i + 1] if fuse and i + 1 < len(slots) else None)
            elif fuse and i + 1 < len(slots):
                # small tensors (and the clip coefficient) first: the next map needs the updated conv weights and thresholds,
                # the next forward's finish the updated bias and table row F-1
                lib.run_plan([c for c in upd if c[0] == "nnue_sgd_step"], st.cuda_stream, timers)
                cur, nxt = (self.fm_alt, self.fm) if alt else (self.fm, self.fm_alt)
                lo, hi = self.sq_range
                mom = self.flat_momentum[lo:hi] if self.flat_momentum is not None else None
                with lib.time_calls(timers):
                    lib.ftm_conv_binarize(self.inputs[slots[i + 1]][0], self.p["conv.weight"], self.p["visual_threshold"], self.stride,
                                          self.F, self.L1, conv_out=self.conv_out, fm=nxt, patches=self.patches,
                                          write_conv_out=not (self.use_patches and self.reform_conv_out))
                    lib.ftm_backward_weight_update_forward(self.d_ft, cur, self.p["input.weight"], mom, self.clip_coef, self.lr, self.momentum,
                                                           self.weight_decay, self.dp.grad_scale, False, nxt, self.p["input.bias"], self.ft,
                                                           lr_dev=self.lr_dev)
            elif alt:
                swap = self._alt_swap()
                lib.run_plan([(nm, fn, tuple(swap.get(a, a) if isinstance(a, int) else a for a in args)) for nm, fn, args in upd],
                             st.cuda_stream, timers)
            else:
                lib.run_plan(upd, st.cuda_stream, timers)

    def step_many(self, slots, timers=None) -> torch.Tensor:
        """``len(slots)`` consecutive optimizer steps on what the named input slots already hold (fill
        ``trainer.inputs[s]`` first; a slot may repeat), replayed as ONE hipGraph: the same kernels in the same order as
        ``step(slot=s)`` for each s, without the gap between two graph launches (measured 5 us at the CIFAR batch-512
        configuration, 5 % of its step).  With a big table (``fuse_next_forward``) the table update of every step but the last
        also forms the next step's FeatureTransformer forward -- one pass over the table instead of two, bitwise the same
        results.  Returns the mean loss of every step (device vector, no sync; a view of a ring
        the next call overwrites); ``self.loss`` is not written.  Falls back to single steps while the plans are not recorded yet, without graphs, or with an eager
        collective.  ``timers`` ({C entry point: list}): the same launches issued eagerly with HIP events around the named
        calls (bench.py's per-kernel durations of the group form)."""
        slots = tuple(int(s) for s in slots)
        if not slots or min(slots) < 0 or max(slots) >= len(self.inputs):
            raise ValueError(f"slots must name input slots 0..{len(self.inputs) - 1}")
        if timers is not None and self.steps_done > 0 and self._plan_local is not None:
            # (with ranks: the collectives are issued eagerly, in the same order on every rank)
            _, _, upd = self._plans(slots[0])
            if self.loss_ring.numel() < len(slots):
                self.loss_ring = torch.zeros((len(slots),), dtype=torch.float32, device=self.dev)
            self._run_many(torch.cuda.current_stream(self.dev), slots, self.loss_ring, upd, timers)
            self._last_alt = self.fuse_next_forward and len(slots) % 2 == 0 and (not self.dp.collectives or self.factor_exchange)
            self.steps_done += len(slots)
            return self.loss_ring[:len(slots)]
        one_graph = (self.use_graph and self.steps_done > 0 and self._plan_local is not None
                     and (not self.dp.collectives or self.capture_collectives))
        if one_graph and (slots, "many") not in self._g_local:
            _, _, upd = self._plans(slots[0])
            if self.loss_ring.numel() < len(slots):
                self.loss_ring = torch.zeros((len(slots),), dtype=torch.float32, device=self.dev)
            ring = self.loss_ring
            captured = True
            try:
                self._g_local[(slots, "many")] = (self._capture(lambda st: self._run_many(st, slots, ring, upd)), ring)
            except RuntimeError as exc:
                if not self.dp.collectives:
                    raise
                import warnings
                warnings.warn(f"collectives could not be captured into the step graph ({exc}); using the eager collective")
                captured = False
                torch.cuda.synchronize(self.dev)
            if self.dp.collectives and not self._ranks_agree(captured):  # all ranks or none
                self._g_local.pop((slots, "many"), None)
                self.capture_collectives = False
                one_graph = False
        if not one_graph:
            if self.loss_ring.numel() < len(slots):
                self.loss_ring = torch.zeros((len(slots),), dtype=torch.float32, device=self.dev)
            for i, s in enumerate(slots):
                self.loss_ring[i].copy_(self.step(slot=s), non_blocking=True)
            return self.loss_ring[:len(slots)]
        graph, ring = self._g_local[(slots, "many")]
        graph.replay()
        self._last_alt = self.fuse_next_forward and len(slots) % 2 == 0 and (not self.dp.collectives or self.factor_exchange)
        self.steps_done += len(slots)
        return ring[:len(slots)]

    def rebuilt(self, ft_path: str) -> "NnueTrainer":
        """A trainer for the same model, shapes, hyper-parameters and optimizer state that uses another FeatureTransformer
        kernel family (``"mfma"`` | ``"bits"`` | ``"list"`` | ``"auto"``, as NNUE_FT_PATH) -- the train loop's density check
        calls this at an epoch boundary when the learnable thresholds have moved the active-feature density across the
        measured crossover of the dense-product and gather forms.  This trainer must not be used afterwards (the module's
        parameters become views of the new trainer's buffers)."""
        torch.cuda.synchronize(self.dev)
        old = os.environ.get("NNUE_FT_PATH")
        os.environ["NNUE_FT_PATH"] = ft_path
        try:
            new = NnueTrainer(self.model, self.B, (self.H, self.W), group=self.dp.group, use_graph=self.use_graph,
                              input_slots=len(self.inputs), optimizer=self.optimizer, betas=self.betas, eps=self.eps, **self._hyper)
        finally:
            if old is None:
                os.environ.pop("NNUE_FT_PATH", None)
            else:
                os.environ["NNUE_FT_PATH"] = old
        if self.sharded_update and self.flat_momentum is not None and self.dp.world > 1 and self.steps_done > 0:
            self.dp.all_gather(self.flat_momentum, self.dp.shard_of(self.flat_momentum))
        for src, dst in zip(self._optimizer_buffers(), new._optimizer_buffers()):
            dst.copy_(src)
        new.steps_done = self.steps_done
        return new

    def optimizer_state_dict(self) -> dict:
        """The optimizer state in torch.optim's own state_dict format (SGD momentum buffers, or Adam's step /
        exp_avg / exp_avg_sq), indexed like ``model.parameters()`` -- what checkpoint_manager.py:45-51 stores and
        ``optimizer.load_state_dict`` (checkpoint_manager.py:75-85) expects.

        With the sharded update (more than one rank, bandwidth-sized buffers) every rank owns only its shard of the
        momentum, so this call first all-gathers the shards: it is then a COLLECTIVE and every rank must make it (the
        train loop does: all ranks reach the same best-F1 decision)."""
        if self.sharded_update and self.flat_momentum is not None and self.dp.world > 1 and self.steps_done > 0:
            torch.cuda.current_stream(self.dev).synchronize()
            self.dp.all_gather(self.flat_momentum, self.dp.shard_of(self.flat_momentum))
        params = list(self.model.parameters())
        if self.optimizer == "adam":
            opt = torch.optim.Adam(params, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)
        else:
            opt = torch.optim.SGD(params, lr=self.lr, momentum=self.momentum, weight_decay=self.weight_decay)
        sd = opt.state_dict()
        if self.steps_done == 0:
            return sd
        index = {k: i for i, (k, _) in enumerate(self.model.named_parameters())}
        for k in self.layout.names:
            if self.optimizer == "adam":
                sd["state"][index[k]] = {"step": torch.tensor(float(self.steps_done)),
                                         "exp_avg": self.layout.views(self.flat_exp_avg)[k].clone(),
                                         "exp_avg_sq": self.layout.views(self.flat_exp_avg_sq)[k].clone()}
            elif self.flat_momentum is not None:
                sd["state"][index[k]] = {"momentum_buffer": self.layout.views(self.flat_momentum)[k].clone()}
        return sd

    @torch.no_grad()
    def evaluate(self, images: torch.Tensor) -> torch.Tensor:
        """Forward only on a full-size batch; returns logits (a view of the static buffer)."""
        self.images.copy_(images, non_blocking=True)  # the slot the eager segments read (slot 0 unless a plan was recorded elsewhere)
        self._forward()
        return self.logits

    def active_stats(self) -> Tuple[float, int]:
        """(mean, max) active features per image of the last batch -- reads back; not for timed regions."""
        fm = self.fm_alt if (self.use_mfma and self._last_alt) else self.fm
        n = (fm if self.use_mfma else self.bits if self.use_bits else self.act).n.float()
        return float(n.mean()), int(n.max())
