"""Probe: the shipped training step of a BASELINE workload launched eagerly (no hipGraph), for rocprofv3 -- `--kernel-trace
--stats`, or one `--pmc` counter group per pass (graph replays are not attributed per kernel by the counter passes).
    python tools/probe_step.py c2|c3|c3k1|c4 [steps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nnue-vision_amd"))
sys.path.insert(0, ROOT)
import nnue
from bench import OPT, WORKLOADS
from nnue_hip.trainer import NnueTrainer

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = WORKLOADS[which]
torch.manual_seed(0)
model = nnue.NNUE(nnue.GridFeatureSet(cfg["grid"], cfg["fps"]), cfg["l1"], cfg["l2"], cfg["l3"], num_classes=cfg["classes"],
                  input_size=cfg["image"], num_ls_buckets=cfg.get("buckets", 1), clip_activations=cfg.get("clip")).cuda()
tr = NnueTrainer(model, cfg["batch"], (cfg["image"], cfg["image"]), use_graph=False, input_slots=2, **OPT)
gen = torch.Generator().manual_seed(1234)
for im, lb in tr.inputs:
    im.copy_(torch.randn(cfg["batch"], 3, cfg["image"], cfg["image"], generator=gen))
    lb.copy_(torch.randint(0, cfg["classes"], (cfg["batch"],), generator=gen))
if getattr(tr, "fuse_next_forward", False):
    # big table: the launches of a step GROUP, eagerly (the table update of a step also forms the next step's forward)
    tr.step()
    for _ in range(max(1, steps // 4)):
        tr.step_many((0, 1, 0, 1), timers={})
else:
    for _ in range(steps):
        tr.step()
torch.cuda.synchronize()
print("done", which, tr.ft_path, "fused table update" if tr.fuse_table_update else "", float(tr.loss))
