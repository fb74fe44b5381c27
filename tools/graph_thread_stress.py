"""Measurement behind the loader's threading rule ("Threads and hipGraphs").

A training step runs for 40 iterations in the main thread (recorded launch list, replayed eagerly or as hipGraphs; legacy default
stream or its own stream) while a second host thread does ONE kind of device work in a loop on the legacy default stream:
  loader    what HipDataLoader's producer did before round 2: a small pageable tensor .to(device, non_blocking=True)
  pageable  the same copy into a preallocated destination        pinned   a pinned 600 KB host buffer copied likewise
  alloc     device allocations only                              kernel   an element-wise kernel on two bytes
The optimizer kernels count the steps they applied and the ones they skipped for non-finite gradients (amp=False: fixed loss
scale 1).  Usage: python tools/graph_thread_stress.py <mode> <legacy|own> <graph|eager>; tools/graph_thread_matrix.sh runs all."""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "experiment-yolo_amd"))
import numpy as np
import torch

from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel

mode, mainstream, use_graph = sys.argv[1], sys.argv[2], sys.argv[3] == "graph"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = DetectionModel("yolov8n-LD-P2.yaml", verbose=False).cuda().train()
B, S = 2, 320
rng = np.random.default_rng(0)


def batch():
    nb = 4
    return dict(img=torch.from_numpy(rng.random((B, 3, S, S), dtype=np.float32)), batch_idx=torch.arange(B).repeat_interleave(nb).float(),
                cls=torch.from_numpy(rng.integers(0, 6, (B * nb, 1)).astype(np.float32)),
                bboxes=torch.from_numpy(np.concatenate([rng.random((B * nb, 2)) * 0.6 + 0.2, rng.random((B * nb, 2)) * 0.2 + 0.03], 1).astype(np.float32)))


stop, count = threading.Event(), [0]


def second_thread():
    pin = torch.empty((2, S, S, 3), dtype=torch.uint8).pin_memory()
    dst = torch.empty((2, S, S, 3), dtype=torch.uint8, device=dev)
    small, dsmall, held = torch.zeros(2, dtype=torch.uint8), torch.zeros(2, dtype=torch.uint8, device=dev), []
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
        if len(held) > 3:
            held.pop(0)
        count[0] += 1


th = threading.Thread(target=second_thread, daemon=True)
ms = torch.cuda.Stream(dev) if mainstream == "own" else torch.cuda.current_stream(dev)
with torch.cuda.stream(ms):
    plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=use_graph, dynamic_scale=False)
    if os.environ.get("DY_NO_VERIFY"):  # measure the raw behaviour: without the capture self-check a broken graph goes unnoticed
        plan._verify_capture = lambda *a: None
    if mode != "none":
        th.start()
    try:
        for it in range(40):
            plan.step(batch(), [1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
        torch.cuda.synchronize()
        st = plan.state.cpu().tolist()
    finally:
        stop.set()
        if th.is_alive():
            th.join(2)
print(f"RESULT second-thread={mode:9s} main-stream={mainstream:7s} graph={str(use_graph):5s}: optimizer steps taken {st[5]:.0f} skipped {st[6]:.0f} "
      f"(second-thread loops {count[0]})")
