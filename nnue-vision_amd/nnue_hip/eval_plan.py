"""Forward + loss + confusion bookkeeping of one evaluation batch as a replayed hipGraph.

`evaluate_model` runs over the whole train and validation set every epoch (train.py:378-398); with the step at
0.12 ms the per-call Python of an eager forward (a dozen library calls, each allocating its outputs) made an
evaluation batch cost more than a training step.  An `EvalPlan` owns static buffers for one (batch, H, W) shape,
records the C calls once -- conv + binary map, FeatureTransformer forward, classifier forward, mean cross-entropy,
confusion update, running loss sum -- captures them, and replays the graph per batch.  Weights are read through the
module's own parameter tensors, so the plan sees every optimizer update; it is rebuilt if a parameter is re-bound.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import lib


class EvalPlan:
    def __init__(self, model, batch: int, hw: Tuple[int, int]):
        p0 = model.input.weight
        if not p0.is_cuda:
            raise lib.NnueHipError("EvalPlan needs the model on the GPU (no CPU fallback)")
        self.dev = p0.device
        self.batch, self.hw = batch, hw
        lin = model.classifier._linears()
        self.params = [model.visual_threshold, model.conv.weight, model.input.weight, model.input.bias,
                       lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias]
        self._ptrs = [t.data_ptr() for t in self.params]
        self.stride = int(model.conv.stride[0])
        self.clip = float(model.classifier.clip_activations or 0.0)
        f, l1 = model.input.weight.shape
        fps = model.conv.out_channels
        gh, gw = lib.conv_out_hw(hw[0], hw[1], self.stride)
        self.path = lib.ft_path(f, fps * gh * gw, l1, batch)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.images = torch.empty((batch, 3, hw[0], hw[1]), **f32)
        self.labels = torch.empty((batch,), dtype=torch.int64, device=self.dev)
        self.conv_out = torch.empty((batch, fps, gh, gw), **f32)
        self.ft = torch.empty((batch, l1), **f32)
        l2, l3, c = lin[0].out_features, lin[1].out_features, lin[2].out_features
        self.acts = (torch.empty((batch, l2), **f32), torch.empty((batch, l3), **f32), torch.empty((batch, c), **f32))
        self.logits = self.acts[2]
        self.ce = (torch.empty((batch,), **f32), torch.zeros((), **f32), None)
        self.K = int(getattr(model, "num_ls_buckets", 1))
        self.positions = fps * gh * gw
        self.bucket_plan = lib.BucketPlan(batch, self.K, self.dev) if self.K > 1 else None
        self.cls_scratch = torch.empty((lib.classifier_scratch_bytes(batch, l1, l2, l3, self.K),), dtype=torch.uint8, device=self.dev)
        k = 2 if c == 1 else c
        self.classes = c
        self.confusion = torch.zeros((k, k), dtype=torch.int64, device=self.dev)
        self.loss_sum = torch.zeros((), dtype=torch.float64, device=self.dev)
        self.metric_labels = torch.empty((batch,), dtype=torch.int64, device=self.dev) if c == 1 else self.labels
        if self.path == "mfma":
            self.feats = lib.FeatureMatrix.empty(batch, fps * gh * gw, f, l1, self.dev)
        elif self.path == "bits":
            self.feats = lib.FeatureBits.empty(batch, fps * gh * gw, f, l1, self.dev)
        else:
            self.feats = lib.ActList.empty(batch, fps * gh * gw, f, self.dev)
        self.num_rows, self.l1 = f, l1
        self.graph = None

    def stale(self, model) -> bool:
        lin = model.classifier._linears()
        now = [model.visual_threshold, model.conv.weight, model.input.weight, model.input.bias,
               lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias]
        return [t.data_ptr() for t in now] != self._ptrs

    def _body(self) -> None:
        thr, conv_w, ft_w, ft_b, w1, b1, w2, b2, w3, b3 = self.params
        if self.path == "mfma":
            lib.ftm_conv_binarize(self.images, conv_w, thr, self.stride, self.num_rows, self.l1, conv_out=self.conv_out, fm=self.feats)
            lib.ftm_forward(ft_w, ft_b, self.feats, out=self.ft)
        elif self.path == "bits":
            lib.conv3x3_forward(self.images, conv_w, self.stride, out=self.conv_out)
            lib.binarize_bits(self.conv_out, thr, self.num_rows, self.l1, bits=self.feats, stages=1)
            lib.ftb_forward(ft_w, ft_b, self.feats, out=self.ft)
        else:
            lib.conv3x3_forward(self.images, conv_w, self.stride, out=self.conv_out)
            lib.binarize_features(self.conv_out, thr, self.num_rows, act=self.feats)
            lib.ft_forward(ft_w, ft_b, self.feats, out=self.ft)
        if self.bucket_plan is not None:
            lib.bucket_group(self.feats.n, self.positions, self.K, plan=self.bucket_plan)
        lib.classifier_forward(self.ft, True, w1, b1, w2, b2, w3, b3, self.clip, scratch=self.cls_scratch, out=self.acts,
                               buckets=self.bucket_plan)
        lib.cross_entropy(self.logits, self.labels, want_grad=False, out=self.ce)
        self.loss_sum.add_(self.ce[1].double())
        if self.classes == 1:  # the reference's binary rule for the metrics (evaluate.py:30-37)
            self.metric_labels.copy_(self.labels > 0)
        lib.confusion_accumulate(self.logits, self.metric_labels, self.confusion)

    def reset(self) -> None:
        self.confusion.zero_()
        self.loss_sum.zero_()

    def run(self, images: torch.Tensor, labels: torch.Tensor) -> None:
        self.images.copy_(images, non_blocking=True)
        self.labels.copy_(labels.reshape(-1), non_blocking=True)
        with torch.no_grad():
            if self.graph is None:
                self._body()  # warm-up outside capture (undo its bookkeeping below)
                keep_conf, keep_loss = self.confusion.clone(), self.loss_sum.clone()
                torch.cuda.synchronize(self.dev)
                side = torch.cuda.Stream(device=self.dev)
                side.wait_stream(torch.cuda.current_stream(self.dev))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.stream(side):
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        self._body()
                torch.cuda.current_stream(self.dev).wait_stream(side)
                self.confusion.copy_(keep_conf)
                self.loss_sum.copy_(keep_loss)
                self.graph = g
                return  # the eager warm-up already counted this batch
            self.graph.replay()


def plan_for(model, batch: int, hw: Tuple[int, int]) -> EvalPlan:
    cache: Dict[tuple, EvalPlan] = model.__dict__.setdefault("_nnue_hip_eval_plans", {})
    key = (batch, hw[0], hw[1], str(model.input.weight.device))
    plan = cache.get(key)
    if plan is None or plan.stale(model):
        plan = cache[key] = EvalPlan(model, batch, hw)
    return plan
