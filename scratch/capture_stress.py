"""Targeted experiment for the silent no-train run: capture the optimizer / step graphs (thread_local mode) while a second
thread keeps the allocator, a copy stream and events busy exactly as HipDataLoader's producer does; after every capture the
graph is replayed and its effect checked."""
import os, sys, threading, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/experiment-yolo_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_poison import _batch
from conftest import CFG_DIR
from oracle import graph as og
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

name = sys.argv[1] if len(sys.argv) > 1 else "yolov8n-LD-P2"
legacy = len(sys.argv) > 2 and sys.argv[2] == "legacy"   # producer copies on the legacy default stream (the pre-fix loader)
dev = torch.device("cuda", 0)
cfg = os.path.join(CFG_DIR, name + ".yaml")
g = og.build_graph(og.load_yaml(cfg))
m = DetectionModel(cfg, verbose=False)
m.load_state_dict(og.fill_state(og.state_layout(g), 11), strict=True)
m.cuda().train()
B, S = 2, 320
plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=True)
stop = threading.Event()
count = [0]
def producer():
    cs = torch.cuda.Stream(dev)
    pins = [torch.empty((2, S, S, 3), dtype=torch.uint8).pin_memory() for _ in range(4)]
    rng = np.random.default_rng(0)
    held = []
    while not stop.is_set():
        k = count[0] % 4
        if legacy:
            t = pins[k].to(dev, non_blocking=True)
            f = torch.zeros(2, dtype=torch.uint8).to(dev, non_blocking=True)
        else:
            with torch.cuda.stream(cs):
                t = pins[k].to(dev, non_blocking=True)
                f = torch.zeros(int(rng.integers(1, 1 << 22)), dtype=torch.uint8).pin_memory().to(dev, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(cs)
        held.append((t, f))
        if len(held) > 3: held.pop(0)
        count[0] += 1
th = threading.Thread(target=producer, daemon=True); th.start()
bad = 0
try:
    for it in range(int(os.environ.get("ITERS", "150"))):
        plan.rec_opt.clear(); plan.graph_opt.clear()
        if it % 10 == 0:
            plan.rec_fb = None; plan.graph_fb = None
        plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
        plan.forward_backward(_batch(B, S, 4, it % 5))
        p0 = plan.rt.flat_p.clone(); st0 = plan.state.cpu()
        plan.optimizer_step()          # eager + capture
        torch.cuda.synchronize(); p1 = plan.rt.flat_p.clone(); st1 = plan.state.cpu()
        plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
        plan.forward_backward(_batch(B, S, 4, (it + 1) % 5))   # graph replay (or re-trace)
        l1 = plan.crit.scalars.cpu().clone()
        plan.optimizer_step()          # graph replay
        torch.cuda.synchronize(); p2 = plan.rt.flat_p.clone(); st2 = plan.state.cpu()
        ok = (st1[5] == st0[5] + 1) and (st2[5] == st1[5] + 1) and bool((p1 != p0).any()) and bool((p2 != p1).any()) and st2[6] == 0 and bool(torch.isfinite(l1[5:9]).all())
        if not ok:
            bad += 1
            print("ANOMALY at", it, st0.tolist(), st1.tolist(), st2.tolist(), float((p1 - p0).abs().max()), float((p2 - p1).abs().max()), l1[5:9].tolist(), flush=True)
finally:
    stop.set(); th.join(2)
print(name, "legacy" if legacy else "copy-stream", "iterations done, producer loops", count[0], "anomalies", bad)
