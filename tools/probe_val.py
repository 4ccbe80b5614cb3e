"""Probe: the stand-alone FeatureTransformer value gradient (nnue_ftm_backward_values) and forward at the 224x224 shape, for
rocprofv3 --kernel-trace --stats / --pmc passes.    python tools/probe_val.py [reps] [fwd]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
from nnue_hip import lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
b, fps, gh, gw, f, l1 = 128, 64, 32, 32, 65536, 1024
gen = torch.Generator().manual_seed(0)
conv_out = torch.randn(b, fps, gh, gw, generator=gen).cuda()
thr = torch.full((fps,), 0.17).cuda()
weight, bias = (torch.randn(f, l1, generator=gen) * 0.1).cuda(), torch.zeros(l1).cuda()
d_out = (torch.randn(b, l1, generator=gen) / b).cuda()
fm = lib.ftm_binarize(conv_out, thr, f, l1)
dv = torch.empty(b, fps * gh * gw, device="cuda")
out = torch.empty(b, l1, device="cuda")
for _ in range(reps):
    lib.ftm_backward_values(d_out, weight, fm, dv)
    if len(sys.argv) > 2:
        lib.ftm_forward(weight, bias, fm, out)
torch.cuda.synchronize()
print("done", float(dv.abs().sum()))
