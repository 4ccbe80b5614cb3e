"""Probe: dense-MFMA FeatureTransformer kernels vs the LDS-staged gather kernels (run under rocprofv3 --kernel-trace)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
from nnue_hip import lib

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfgs = {"c2": (512, 8, 11, 11, 800, 1024), "c3": (1024, 8, 11, 11, 800, 1024), "c4": (128, 64, 32, 32, 65536, 1024)}
b, fps, gh, gw, f, l1 = cfgs[which]
gen = torch.Generator().manual_seed(0)
conv_out = torch.randn(b, fps, gh, gw, generator=gen).cuda()
thr = torch.full((fps,), 0.17).cuda()  # ~43 % active
weight, bias = (torch.randn(f, l1, generator=gen) * 0.1).cuda(), torch.zeros(l1).cuda()
d_out = (torch.randn(b, l1, generator=gen) / b).cuda()
fm = lib.ftm_binarize(conv_out, thr, f, l1)
n = fm.n
out = torch.empty(b, l1, device="cuda"); dw = torch.empty(f, l1, device="cuda"); db = torch.empty(l1, device="cuda")
dv = torch.empty(b, fps * gh * gw, device="cuda")
do_gather = os.environ.get("PROBE_GATHER", "1") == "1"
if do_gather:
    bits = lib.binarize_bits(conv_out, thr, f, l1)
for _ in range(20):
    lib.ftm_binarize(conv_out, thr, f, l1, fm)
    lib.ftm_forward(weight, bias, fm, out)
    lib.ftm_backward_weight(d_out, fm, dw, db)
    lib.ftm_backward_values(d_out, weight, fm, dv)
    lib.ftm_backward(d_out, weight, fm, dw, db, dv)
    if do_gather:
        lib.ftb_forward(weight, bias, bits, out)
        lib.ftb_backward_weight(d_out, bits, dw, db)
        lib.ftb_backward_values(d_out, weight, bits, dv)
torch.cuda.synchronize()
print("done", which, float(n.float().mean()))
