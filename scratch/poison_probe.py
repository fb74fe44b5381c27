import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/experiment-yolo_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_poison import _batch
from conftest import CFG_DIR
from oracle import graph as og
import ultralytics.hip.engine as E
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

def run(name, B, S, poison, zero_p=False):
    cfg = os.path.join(CFG_DIR, name + ".yaml")
    g = og.build_graph(og.load_yaml(cfg))
    m = DetectionModel(cfg, verbose=False)
    sd = og.fill_state(og.state_layout(g), 11)
    if zero_p:
        for k in sd:
            if "p_conv.weight" in k: sd[k] = torch.zeros_like(sd[k])
            if "p_conv.bias" in k: sd[k] = sd[k].clamp(-0.9, 0.9)
    m.load_state_dict(sd, strict=True)
    E.POISON = poison
    m.cuda().train()
    plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=False)
    plan.set_hyper([0.0]*3, 0.9, [0.0]*3)
    plan.forward_backward(_batch(B, S, 4, 0))
    torch.cuda.synchronize()
    E.POISON = False
    return plan.rt.flat_g.cpu().clone(), plan.rt.param_off, plan.crit.scalars.cpu().clone()

for zero_p in (False, True):
    a, off, sa = run("yolov8n-LD-P2", 2, 640, False, zero_p)
    b, _, sb = run("yolov8n-LD-P2", 2, 640, False, zero_p)
    c, _, sc = run("yolov8n-LD-P2", 2, 640, True, zero_p)
    print("zero_p", zero_p, "items", sa[5:9].tolist(), sb[5:9].tolist(), sc[5:9].tolist())
    names = sorted(off, key=lambda k: off[k])
    for tag, x, y in (("clean-clean", a, b), ("clean-dirty", a, c)):
        d = (x - y).abs()
        print(tag, "max abs diff", float(d.max()), "of", float(x.abs().max()), "nonfinite", int((~torch.isfinite(y)).sum()))
        bad = []
        for i, k in enumerate(names):
            o = off[k]; e = off[names[i+1]] if i + 1 < len(names) else len(x)
            dd = float(d[o:e].max())
            if dd > 0: bad.append((k, dd, float(x[o:e].abs().max())))
        print("  params differing:", len(bad), "of", len(names))
        for k, dd, mx in bad[:12]: print("   ", k, dd, mx)
