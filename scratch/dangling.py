"""Every device pointer recorded in the step's launch list must lie inside a tensor the plan keeps alive."""
import os, sys, gc, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/experiment-yolo_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_poison import _batch
from conftest import CFG_DIR
from oracle import graph as og
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

name = sys.argv[1] if len(sys.argv) > 1 else "yolov8n-LD-P2"
cfg = os.path.join(CFG_DIR, name + ".yaml")
g = og.build_graph(og.load_yaml(cfg))
m = DetectionModel(cfg, verbose=False)
m.load_state_dict(og.fill_state(og.state_layout(g), 11), strict=True)
m.cuda().train()
B, S = 2, 320
plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=True)
plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
plan.forward_backward(_batch(B, S, 4, 0))
plan.optimizer_step()
torch.cuda.synchronize()
gc.collect()
# live CUDA tensors reachable by the garbage collector
live = []
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            st = o.untyped_storage()
            live.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
    except Exception:
        pass
live = sorted(set(live))
lo = min(a for a, _ in live); hi = max(b for _, b in live)
print("live cuda storages", len(live))
def inside(p):
    return any(a <= p < b for a, b in live)
stats = torch.cuda.memory_stats()
def scan(rec, tag):
    bad = {}
    for fn, args, nm, side in rec.ops:
        if fn is None: continue
        for i, a in enumerate(args):
            v = a.value if isinstance(a, (C.c_void_p,)) else a
            if isinstance(v, int) and v > (1 << 32):
                if not inside(v):
                    bad.setdefault(nm, []).append((i, hex(v)))
    print(tag, "launches", len(rec.ops), "launches with pointers outside every live tensor:", {k: v[:4] for k, v in bad.items()})
scan(plan.rec_fb, "fwd/bwd")
for k, r in plan.rec_opt.items(): scan(r, "optimizer")
# the loss argument block
a = plan.crit._args
for f, _t in a._fields_:
    v = getattr(a, f)
    vals = list(v) if hasattr(v, "__len__") else [v]
    for x in vals:
        if isinstance(x, int) and x > (1 << 32) and not inside(x):
            print("loss args field", f, hex(x), "NOT inside a live tensor")
