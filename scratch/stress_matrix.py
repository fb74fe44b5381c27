import os, sys, threading, time, gc
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/experiment-yolo_amd"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_poison import _batch
from conftest import CFG_DIR
from oracle import graph as og
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

mode, mainstream, use_graph = sys.argv[1], sys.argv[2], sys.argv[3] == "graph"
name = "yolov8n-LD-P2"
dev = torch.device("cuda", 0)
cfg = os.path.join(CFG_DIR, name + ".yaml")
g = og.build_graph(og.load_yaml(cfg))
m = DetectionModel(cfg, verbose=False)
m.load_state_dict(og.fill_state(og.state_layout(g), 11), strict=True)
m.cuda().train()
B, S = 2, 320
stop = threading.Event(); count = [0]
def producer():
    pin = torch.empty((2, S, S, 3), dtype=torch.uint8).pin_memory()
    dst = torch.empty((2, S, S, 3), dtype=torch.uint8, device=dev)
    small = torch.zeros(2, dtype=torch.uint8)
    dsmall = torch.zeros(2, dtype=torch.uint8, device=dev)
    held = []
    while not stop.is_set():
        if mode == "alloc":
            held.append(torch.empty(int(np.random.randint(1, 1 << 20)), dtype=torch.uint8, device=dev))
        elif mode == "pinned":
            dst.copy_(pin, non_blocking=True)
        elif mode == "pageable":
            dsmall.copy_(small, non_blocking=True)
        elif mode == "kernel":
            dsmall.add_(1)
        elif mode == "loader":
            held.append(small.to(dev, non_blocking=True))
        if len(held) > 3: held.pop(0)
        count[0] += 1
th = threading.Thread(target=producer, daemon=True)
ms = torch.cuda.Stream(dev) if mainstream == "own" else torch.cuda.current_stream(dev)
skipped = -1
with torch.cuda.stream(ms):
    plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=use_graph)
    th.start()
    try:
        for it in range(40):
            plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
            plan.forward_backward(_batch(B, S, 4, it % 5))
            plan.optimizer_step()
        torch.cuda.synchronize()
        st = plan.state.cpu().tolist()
    finally:
        stop.set(); th.join(2)
print(f"RESULT producer={mode:9s} main-stream={mainstream:7s} graph={use_graph}: steps {st[5]:.0f} skipped {st[6]:.0f} scale {st[0]:g} producer-loops {count[0]}")
