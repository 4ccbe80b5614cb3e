"""autograd bridges between the nn.Module surface (nnue.py) and the HIP kernels.

Four nodes, each a hand-written forward/backward pair over the C ABI:

* ``SparseValuesFn``        -- the values returned by NNUE._to_sparse_features, attached to the map (nnue.py:628-633)
* ``FeatureTransformerFn``  -- FeatureTransformer.forward called stand-alone (nnue.py:686-710)
* ``ClassifierFn``          -- SimpleClassifier.forward called stand-alone (nnue.py:736-738)
* ``NnueFn``                -- the whole NNUE.forward (nnue.py:637-671) as ONE node: conv, binarise +
                               compact, gather-accumulate, pairwise + classifier.  No data-dependent
                               shapes, no host synchronisation.
"""
from __future__ import annotations

import torch

from . import lib


class FeatureTransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, val, weight, bias):
        act = lib.ft_prepare(idx, val, weight.shape[0])
        out = lib.ft_forward(weight, bias, act)
        ctx.act = act
        ctx.width = idx.shape[1]
        ctx.save_for_backward(weight)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (weight,) = ctx.saved_tensors
        _, need_val, need_w, need_b = ctx.needs_input_grad
        d_out = d_out.contiguous()
        d_val = d_w = d_b = None
        if need_val:
            d_val = lib.ft_backward_values(d_out, weight, ctx.act, ctx.width)
        if need_w or need_b:
            d_w, d_b = lib.ft_backward_weight(d_out, ctx.act, weight.shape[0], want_weight=need_w, want_bias=need_b)
        return None, d_val, d_w, d_b


class ClassifierFn(torch.autograd.Function):
    """SimpleClassifier / BucketedClassifier.  ``bucket`` (integer [B], stacked weights only) names each sample's layer
    stack; it is grouped on the device (nnue_bucket_group with P = 0)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, pairwise, clip, bucket=None):
        x = x.contiguous()
        plan = None
        if w1.dim() == 3:
            if bucket is None:
                raise ValueError("stacked classifier weights need the per-sample bucket ids")
            plan = lib.bucket_group(bucket.to(device=x.device, dtype=torch.int32).contiguous(), 0, w1.shape[0])
        h1, h2, logits = lib.classifier_forward(x, pairwise, w1, b1, w2, b2, w3, b3, clip, buckets=plan)
        ctx.pairwise, ctx.clip, ctx.plan = pairwise, clip, plan
        ctx.save_for_backward(x, w1, w2, w3, h1, h2)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        x, w1, w2, w3, h1, h2 = ctx.saved_tensors
        d_x, g = lib.classifier_backward(x, ctx.pairwise, w1, w2, w3, h1, h2, d_logits.contiguous(), ctx.clip,
                                         want_dx=ctx.needs_input_grad[0], buckets=ctx.plan)
        return (d_x, *g, None, None, None)


class SparseValuesFn(torch.autograd.Function):
    """The differentiable half of NNUE._to_sparse_features (nnue.py:628-633): val = map[idx], gradient scattered back."""

    @staticmethod
    def forward(ctx, flat_map, idx):
        ctx.positions = flat_map.shape[1]
        ctx.save_for_backward(idx)
        return lib.sparse_values(flat_map.contiguous(), idx)

    @staticmethod
    def backward(ctx, d_val):
        (idx,) = ctx.saved_tensors
        return lib.sparse_values_backward(d_val.contiguous(), idx, ctx.positions), None


class NnueFn(torch.autograd.Function):
    """images -> logits.  Inputs: images, threshold [fps], conv weight, FT weight/bias, 3x (weight, bias)."""

    @staticmethod
    def forward(ctx, images, thr, conv_w, ft_w, ft_b, w1, b1, w2, b2, w3, b3, stride, clip):
        images = images.contiguous()
        conv_out = lib.conv3x3_forward(images, conv_w, stride)
        # binary features: dense MFMA products over the float map, else bit masks + LDS-staged tiles, else id lists
        ctx.path = lib.ft_path(ft_w.shape[0], conv_out[0].numel(), ft_w.shape[1], conv_out.shape[0])
        if ctx.path == "mfma":
            feats = lib.ftm_binarize(conv_out, thr, ft_w.shape[0], ft_w.shape[1])
            ft = lib.ftm_forward(ft_w, ft_b, feats)
        elif ctx.path == "bits":
            feats = lib.binarize_bits(conv_out, thr, ft_w.shape[0], ft_w.shape[1])
            ft = lib.ftb_forward(ft_w, ft_b, feats)
        else:
            feats = lib.binarize_features(conv_out, thr, ft_w.shape[0])
            ft = lib.ft_forward(ft_w, ft_b, feats)
        # bucketed layer stacks: each sample's stack follows from its active-feature count, grouped on the device
        plan = lib.bucket_group(feats.n, conv_out[0].numel(), w1.shape[0]) if w1.dim() == 3 else None
        h1, h2, logits = lib.classifier_forward(ft, True, w1, b1, w2, b2, w3, b3, clip, buckets=plan)
        ctx.feats, ctx.stride, ctx.clip, ctx.plan = feats, stride, clip, plan
        ctx.save_for_backward(images, thr, conv_w, conv_out, ft_w, ft, w1, w2, w3, h1, h2)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        images, thr, conv_w, conv_out, ft_w, ft, w1, w2, w3, h1, h2 = ctx.saved_tensors
        feats = ctx.feats
        need = ctx.needs_input_grad
        d_ft, g_cls = lib.classifier_backward(ft, True, w1, w2, w3, h1, h2, d_logits.contiguous(), ctx.clip, buckets=ctx.plan)
        d_ftw = d_ftb = d_thr = d_conv_w = d_images = None
        if need[3] or need[4]:
            if ctx.path == "mfma":
                d_ftw, d_ftb = lib.ftm_backward_weight(d_ft, feats, want_weight=need[3], want_bias=need[4])
            elif ctx.path == "bits":
                d_ftw, d_ftb = lib.ftb_backward_weight(d_ft, feats, want_weight=need[3], want_bias=need[4])
            else:
                d_ftw, d_ftb = lib.ft_backward_weight(d_ft, feats, ft_w.shape[0], want_weight=need[3], want_bias=need[4])
        if need[0] or need[1] or need[2]:
            # value gradient of the binary features == d(conv_out) (identity STE, nnue.py:33)
            if ctx.path == "mfma":
                d_conv_out = lib.ftm_backward_values(d_ft, ft_w, feats).view_as(conv_out)
            elif ctx.path == "bits":
                d_conv_out = lib.ftb_backward_values(d_ft, ft_w, feats).view_as(conv_out)
            else:
                d_conv_out = lib.ft_backward_values(d_ft, ft_w, feats, feats.cap).view_as(conv_out)
            if need[1] or need[2]:
                d_thr, d_conv_w = lib.ste_conv_backward(images, conv_out, thr, d_conv_out, ctx.stride)
                d_thr = d_thr.view_as(thr)
            if need[0]:  # gradient to the pixels: never needed by the training loop
                d_images = lib.conv3x3_backward_input(d_conv_out.contiguous(), conv_w, images.shape, ctx.stride)
        return (d_images, d_thr, d_conv_w, d_ftw, d_ftb, *g_cls, None, None)
