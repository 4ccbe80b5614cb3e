import os, sys, torch
from pathlib import Path
R = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(R / "nnue-vision_amd"), str(R / "oracle"), str(R / "tests")]
import nnue, nnue_oracle as orc
from nnue_hip.trainer import NnueTrainer
from test_gpu_step_shapes import SHAPES, OPT, clean_batch
cfg = SHAPES["c4"]
torch.manual_seed(0)
model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"], input_size=cfg["image"])
params = {k: v.detach().clone() for k, v in model.state_dict().items()}
stride = orc.conv_stride(cfg["image"], cfg["grid"])
gen = torch.Generator().manual_seed(77)
images, labels = clean_batch(cfg, params, stride, gen)
_, ref_loss, ref_grads, keep = orc.loss_and_grads_explicit(params, images, labels, stride)
tot64 = sum((g.double() ** 2).sum() for g in ref_grads.values()).sqrt()
print("oracle norm (f64 of f32 grads)", float(tot64), {k: float(g.double().norm()) for k, g in ref_grads.items()})
# f64 oracle
p64 = {k: v.double() for k, v in params.items()}
_, l64, g64, _ = orc.loss_and_grads_explicit(p64, images.double(), labels, stride)
print("f64 oracle norm", float(sum((g ** 2).sum() for g in g64.values()).sqrt()), "loss", float(l64), float(ref_loss))
for mode in ("1", "0"):
    os.environ["NNUE_NORM_PARTIALS"] = mode
    m = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"], input_size=cfg["image"])
    m.load_state_dict(params)
    m = m.cuda()
    tr = NnueTrainer(m, cfg["batch"], (cfg["image"], cfg["image"]), use_graph=True, **OPT)
    loss = tr.step(images.cuda(), labels.cuda())
    torch.cuda.synchronize()
    got = tr.layout.views(tr.flat_grads)
    print("partials", mode, "trainer norm", float(tr.grad_norm), "norm of flat grads f64", float(tr.flat_grads.double().norm()), "loss", float(loss))
    for k, ref in g64.items():
        e = float((got[k].cpu().double() - ref).abs().max()); s = float(ref.abs().max())
        e32 = float((ref_grads[k].double() - ref).abs().max())
        print(f"  {k:34s} gpu-vs-f64 {e/s:.2e}   oracle32-vs-f64 {e32/s:.2e}  scale {s:.3e}")
    del tr, m

# ---- where does d_z1 differ?
import torch.nn.functional as F
from nnue_hip import lib
os.environ["NNUE_NORM_PARTIALS"] = "1"
m = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"], input_size=cfg["image"])
m.load_state_dict(params); m = m.cuda()
tr = NnueTrainer(m, cfg["batch"], (cfg["image"], cfg["image"]), use_graph=False, **dict(OPT, lr=0.0))
tr.step(images.cuda(), labels.cuda()); torch.cuda.synchronize()
B, L1, L2, L3, C = tr.B, tr.L1, tr.L2, tr.L3, tr.C
off = lib.classifier_train_dz1_offset(B, L1, L2, L3, C, True)
dz1 = tr.cls_scratch[off:off + B * L2 * 4].view(torch.float32).view(B, L2).cpu().double()
ft = tr.ft.cpu().double()
x = F.conv2d(images.double(), p64["conv.weight"], stride=stride, padding=1)
idx, n = orc.active_lists(x, p64["visual_threshold"])
val = (idx >= 0).double()
ft64 = orc.ft_forward(p64["input.weight"], p64["input.bias"], idx, val)
print("ft err", float((ft - ft64).abs().max()), "ft scale", float(ft64.abs().max()))
l0 = orc.pairwise(ft64)
cls = [p64[f"classifier.classifier.{i}.{n_}"] for i in (0, 2, 4) for n_ in ("weight", "bias")]
z1 = F.linear(l0, cls[0], cls[1]); h1 = F.relu(z1); z2 = F.linear(h1, cls[2], cls[3]); h2 = F.relu(z2)
logits = F.linear(h2, cls[4], cls[5])
_, dl = orc.cross_entropy_backward(logits, labels)
dz2 = (dl @ cls[4]) * (z2 > 0); dz1_ref = (dz2 @ cls[2]) * (z1 > 0)
h1g = tr.h1.cpu().double()
print("h1 err", float((h1g - h1).abs().max()), "scale", float(h1.abs().max()))
bad = ((dz1 - dz1_ref).abs() > 1e-4 * dz1_ref.abs().max()).nonzero()
print("d_z1 mismatches", bad.shape[0], "of", dz1.numel(), "scale", float(dz1_ref.abs().max()))
for b, j in bad[:20].tolist():
    print(f"  b={b} j={j} z1_f64={float(z1[b, j]):.6e} h1_gpu={float(h1g[b, j]):.6e} dz1_gpu={float(dz1[b, j]):.4e} dz1_ref={float(dz1_ref[b, j]):.4e}")
