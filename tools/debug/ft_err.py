"""Error distribution of nnue_ftm_forward at the big-map shapes of tests/test_gpu_ftm.py against float64, in units of
the element-wise bar r * max(1, |ref|) for r = 2e-5 and 1e-4.  Developer tool (GPU box): python tools/debug/ft_err.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT / "nnue-vision_amd"), str(ROOT / "tests")]
from nnue_hip import lib  # noqa: E402

lib.load()
SHAPES = [(24, 64, 32, 32, 65536, 256), (140, 64, 24, 24, 30000, 136), (128, 64, 32, 32, 65536, 1024), (8, 64, 32, 32, 65536, 1024)]
out = []
for b, fps, gh, gw, f, l1 in SHAPES:
    for density in (0.02, 0.45, 1.0):
        gen = torch.Generator().manual_seed(b * 7 + f)
        conv_out = torch.randn(b, fps, gh, gw, generator=gen)
        thr = torch.quantile(conv_out.transpose(0, 1).flatten(1), 1.0 - density, dim=1) if density < 1.0 else torch.full((fps,), -1e9)
        weight, bias = torch.randn(f, l1, generator=gen) * 0.1, torch.randn(l1, generator=gen)
        p = fps * gh * gw
        bits = (conv_out > thr.view(1, -1, 1, 1)).reshape(b, p)
        rows = torch.clamp(torch.arange(p), max=f - 1)
        a = torch.zeros(b, f, dtype=torch.float64)
        a.index_add_(1, rows, bits.double())
        ref = a @ weight.double() + bias.double()
        ref32 = (a.float() @ weight + bias).double()  # a float32 CPU product of the same operands, for scale
        fm = lib.ftm_binarize(conv_out.cuda(), thr.cuda(), f, l1)
        got = lib.ftm_forward(weight.cuda(), bias.cuda(), fm).cpu().double()
        err = (got - ref).abs()
        unit = ref.abs().clamp(min=1.0)
        rec = dict(shape=[b, fps, gh, gw, f, l1], density=density, max_err=float(err.max()), max_ref=float(ref.abs().max()),
                   ratio_2e5=float((err / (2e-5 * unit)).max()), ratio_1e4=float((err / (1e-4 * unit)).max()),
                   outside_2e5=int((err > 2e-5 * unit).sum()), outside_1e4=int((err > 1e-4 * unit).sum()),
                   rms_err=float(err.pow(2).mean().sqrt()), mean_signed=float((got - ref).mean()),
                   cpu_f32_max_err=float((ref32 - ref).abs().max()), cpu_f32_rms=float((ref32 - ref).pow(2).mean().sqrt()))
        print(json.dumps(rec), flush=True)
        out.append(rec)
d = ROOT / "gpurun_out"
if d.is_dir():
    (d / "ft_err.json").write_text(json.dumps(out, indent=1))
