"""Probe: the fused FeatureTransformer forward and the merged backward launch of the training step at a BASELINE shape,
launched eagerly (run under rocprofv3 --kernel-trace or one --pmc counter group per pass)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
from nnue_hip import lib

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfgs = {"c2": (512, 8, 11, 11, 800, 1024, 128), "c3": (1024, 8, 11, 11, 800, 1024, 128)}
b, fps, gh, gw, f, l1, l2 = cfgs[which]
gen = torch.Generator().manual_seed(0)
conv_out = torch.randn(b, fps, gh, gw, generator=gen).cuda()
thr = torch.full((fps,), 0.17).cuda()
weight, bias = (torch.randn(f, l1, generator=gen) * 0.1).cuda(), torch.zeros(l1).cuda()
w1 = (torch.randn(l2, l1, generator=gen) * 0.03).cuda()
d_out = (torch.randn(b, l1, generator=gen) / b).cuda()
fm = lib.ftm_binarize(conv_out, thr, f, l1)
part = torch.empty((l1 // 64) * b * l2 + 1024, device="cuda")
out = torch.empty(b, l1, device="cuda")
dw = torch.empty(f, l1, device="cuda"); db = torch.empty(l1, device="cuda"); dv = torch.empty(b, fps * gh * gw, device="cuda")
d_z1 = (torch.randn(b, l2, generator=gen) / b).cuda()
d_w1 = torch.empty(l2, l1, device="cuda")
ride = lib.ftm_backward_cw_supported(b, f, fps * gh * gw, l1, l2)
for _ in range(20):
    lib.ftm_forward_l1(weight, bias, fm, w1, part, out=out)
    if ride:  # the launch of the training step: + the classifier's first-layer weight gradient tiles
        lib.ftm_backward(d_out, weight, fm, dw, db, dv, ft=out, d_z1=d_z1, d_w1=d_w1)
    else:
        lib.ftm_backward(d_out, weight, fm, dw, db, dv)
torch.cuda.synchronize()
print("done", which)
