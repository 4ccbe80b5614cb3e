import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import nnue
from nnue_hip.trainer import NnueTrainer
z = np.load(os.path.join(ROOT, "tests/golden/step_c1arch.npz"))
cfg = json.loads(str(z["cfg"]))
def build():
    st = {k[7:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("state0/")}
    m = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"], input_size=cfg["input_size"])
    m.load_state_dict(st); return m.cuda()
images0 = torch.from_numpy(z["images0"])
trs = {}
MODES = (True,) if os.environ.get('ONLY_GRAPH') else (False, True)
for ug in MODES:
    m = build()
    trs[ug] = NnueTrainer(m, images0.shape[0], tuple(images0.shape[2:]), lr=cfg["lr"], momentum=cfg["momentum"], weight_decay=cfg["weight_decay"], max_grad_norm=cfg["max_grad_norm"], use_graph=ug)
print("ft_path", trs[True].ft_path, "B", images0.shape[0], cfg)
for s in range(3):
    im = torch.from_numpy(z[f"images{s}"]).cuda(); lb = torch.from_numpy(z[f"labels{s}"]).cuda().long()
    for ug in MODES:
        loss = trs[ug].step(im, lb)
        print(s, ug, float(loss), float(trs[ug].grad_norm), float(z[f"gradnorm{s}"]))
    if len(MODES) == 1:
        b = trs[True]
        for k in b.g: print('   ', k, b.g[k].abs().max().item())
        continue
    a, b = trs[False], trs[True]
    for k in a.g:
        d = (a.g[k] - b.g[k]).abs().max().item()
        if not (d < 1e-6): print("   grad diff", k, d, a.g[k].abs().max().item(), b.g[k].abs().max().item())
