"""Probe: what makes nnue_ste_conv_backward slow at low feature density?  (d sparsity vs threshold position)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
from nnue_hip import lib

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n

for (B, H, fps, stride) in ((128, 224, 64, 7), (512, 32, 8, 3)):
    img = torch.randn(B, 3, H, H, device="cuda")
    w = torch.randn(fps, 3, 3, 3, device="cuda") * 0.2
    conv = lib.conv3x3_forward(img, w, stride)
    for thr_q in (0.1, 0.5, 0.95):
        thr = torch.quantile(conv.transpose(0, 1).flatten(1)[:, :1 << 20], thr_q, dim=1).contiguous()
        for dens in (0.02, 0.05, 0.25, 0.5, 1.0):
            d = torch.randn_like(conv) * 1e-3 * (torch.rand_like(conv) < dens)
            us = t(lambda: lib.ste_conv_backward(img, conv, thr, d, stride))
            print(f"B={B} H={H} fps={fps} thr_quantile={thr_q} d_density={dens}: {us:.1f} us", flush=True)
