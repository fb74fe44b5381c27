import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/experiment-yolo_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_poison import _batch
from conftest import CFG_DIR
from oracle import graph as og
import ultralytics.hip.engine as E
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

def run(name, B, S, poison, use_graph, opt, lr):
    cfg = os.path.join(CFG_DIR, name + ".yaml")
    g = og.build_graph(og.load_yaml(cfg))
    m = DetectionModel(cfg, verbose=False)
    sd = og.fill_state(og.state_layout(g), 11)
    m.load_state_dict(sd, strict=True)
    E.POISON = poison
    m.cuda().train()
    plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=use_graph)
    plan.set_hyper([lr]*3, 0.9, [0.0]*3)
    plan.forward_backward(_batch(B, S, 4, 0))
    torch.cuda.synchronize()
    g0 = plan.rt.flat_g.cpu().clone()
    if opt:
        plan.optimizer_step()
    torch.cuda.synchronize()
    E.POISON = False
    return g0, plan.rt.flat_g.cpu().clone(), plan.rt.param_off

def cmp(tag, x, y, off):
    names = sorted(off, key=lambda k: off[k])
    d = (x - y).abs()
    bad = []
    for i, k in enumerate(names):
        o = off[k]; e = off[names[i+1]] if i + 1 < len(names) else len(x)
        dd = float(d[o:e].max())
        if dd > 0: bad.append((k, dd, float(x[o:e].abs().max())))
    print(tag, "max abs diff", float(d.max()), "params differing:", len(bad), bad[:6])

ref = None
for use_graph in (False, True):
    for opt in (False, True):
        for poison in (False, True):
            g0, g1, off = run("yolov8n-LD-P2", 2, 640, poison, use_graph, opt, 0.01)
            if ref is None: ref = g0
            cmp(f"graph={use_graph} opt={opt} poison={poison}: before-opt vs ref", g0, ref, off)
            cmp(f"graph={use_graph} opt={opt} poison={poison}: after-opt vs before", g1, g0, off)
