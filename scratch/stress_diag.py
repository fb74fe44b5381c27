import os, sys, threading, time, gc, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/experiment-yolo_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_poison import _batch
from conftest import CFG_DIR
from oracle import graph as og
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

name = "yolov8n-LD-P2"
nogc = "nogc" in sys.argv
if nogc:
    gc.collect = lambda *a, **k: 0
dev = torch.device("cuda", 0)
cfg = os.path.join(CFG_DIR, name + ".yaml")
g = og.build_graph(og.load_yaml(cfg))
m = DetectionModel(cfg, verbose=False)
m.load_state_dict(og.fill_state(og.state_layout(g), 11), strict=True)
m.cuda().train()
B, S = 2, 320
plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=True)
stop = threading.Event(); count = [0]; held = []
lock = threading.Lock()
def producer():
    pins = [torch.empty((2, S, S, 3), dtype=torch.uint8).pin_memory() for _ in range(4)]
    while not stop.is_set():
        k = count[0] % 4
        t = pins[k].to(dev, non_blocking=True)
        f = torch.zeros(2, dtype=torch.uint8).to(dev, non_blocking=True)
        with lock:
            held.append((t, f))
            if len(held) > 3: held.pop(0)
        count[0] += 1
th = threading.Thread(target=producer, daemon=True); th.start()
def ptr_ranges():
    out = []
    for t in plan.eng.keep:
        if torch.is_tensor(t) and t.is_cuda:
            st = t.untyped_storage(); out.append((st.data_ptr(), st.data_ptr() + st.nbytes(), t))
    return out
try:
    for it in range(3):
        plan.rec_opt.clear(); plan.graph_opt.clear()
        plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
        plan.forward_backward(_batch(B, S, 4, it % 5))
        torch.cuda.synchronize()
        nf = int((~torch.isfinite(plan.rt.flat_g)).sum())
        print("iter", it, "fb#1 nonfinite grads", nf, "state", plan.state.cpu().tolist(), flush=True)
        if nf:
            stop.set(); th.join(2)
            torch.cuda.synchronize()
            # which kept buffers hold non-finite values, and which op writes them first
            rng = ptr_ranges()
            first = {}
            for oi, (fn, args, nm, side) in enumerate(plan.rec_fb.ops):
                if fn is None: continue
                for a in args:
                    if isinstance(a, int) and a > (1 << 32):
                        for lo, hi, t in rng:
                            if lo <= a < hi and id(t) not in first: first[id(t)] = (oi, nm)
            rep = []
            for lo, hi, t in rng:
                if t.dtype in (torch.float16, torch.float32):
                    n = int((~torch.isfinite(t)).sum())
                    if n: rep.append((first.get(id(t), (99999, "?")), n, tuple(t.shape), str(t.dtype), hex(lo)))
            rep.sort()
            for r in rep[:25]: print("  nonfinite buffer: first touched by op", r)
            for i, hb in enumerate(plan.ho.dbox + plan.ho.dcls): print("  head grad", i, int((~torch.isfinite(hb)).sum()))
            print("  scalars", plan.crit.scalars.cpu().tolist())
            # producer overlap
            with lock:
                for t, f in held:
                    for x in (t, f):
                        a0 = x.untyped_storage().data_ptr(); a1 = a0 + x.untyped_storage().nbytes()
                        for lo, hi, kt in rng:
                            if a0 < hi and lo < a1: print("  PRODUCER TENSOR OVERLAPS KEPT BUFFER", hex(a0), hex(lo), tuple(kt.shape))
            break
        plan.optimizer_step()
        plan.forward_backward(_batch(B, S, 4, (it + 1) % 5))
        plan.optimizer_step()
        torch.cuda.synchronize()
        print("iter", it, "end state", plan.state.cpu().tolist(), flush=True)
finally:
    stop.set(); th.join(2)
print("done", count[0])
