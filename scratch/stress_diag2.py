import os, sys, threading
sys.path.insert(0, "/root/repo/experiment-yolo_amd")
import numpy as np, torch
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel
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
stop = threading.Event()
def second():
    d = torch.zeros(2, dtype=torch.uint8, device=dev)
    while not stop.is_set():
        d.add_(1)
th = threading.Thread(target=second, daemon=True)
ms = torch.cuda.Stream(dev)
with torch.cuda.stream(ms):
    plan = StepPlan(m, B, S, nmax=8, init_scale=1.0, use_graph=True, dynamic_scale=False)
    th.start()
    bt = batch()
    for it in range(6):
        plan.set_hyper([1e-3, 1e-4, 1e-4], 0.9, [0.0, 5e-4, 0.0])
        plan.forward_backward(bt)
        torch.cuda.synchronize()
        ho = plan.ho
        nf = [int((~torch.isfinite(t)).sum()) for t in ho.dbox + ho.dcls]
        nfb = [int((~torch.isfinite(t)).sum()) for t in ho.box + ho.cls]
        print("iter", it, "nonfinite head grads", nf, "head outputs", nfb, "flat_g", int((~torch.isfinite(plan.rt.flat_g)).sum()), "scalars", [round(x, 4) for x in plan.crit.scalars.cpu().tolist()[:10]], flush=True)
        if sum(nf):
            stop.set(); th.join(2)
            for l, t in enumerate(ho.dbox):
                bad = ~torch.isfinite(t)
                if bad.any():
                    idx = bad.nonzero()
                    print(" level", l, "shape", tuple(t.shape), "bad", int(bad.sum()), "first idx", idx[:6].tolist(), "per-channel count", bad.sum((0,1,2)).tolist()[:16], "per-image", bad.sum((1,2,3)).tolist())
                    print(" values", t[bad][:8].tolist(), " max finite", float(t[~bad].abs().max()))
            # replay the same graph again now that the second thread is stopped
            plan.graph_fb.replay(); torch.cuda.synchronize()
            print(" replay after stopping the second thread: nonfinite", [int((~torch.isfinite(t)).sum()) for t in ho.dbox + ho.dcls], int((~torch.isfinite(plan.rt.flat_g)).sum()))
            plan.eng.replay(plan.rec_fb); torch.cuda.synchronize()
            print(" eager replay of the same launch list: nonfinite", [int((~torch.isfinite(t)).sum()) for t in ho.dbox + ho.dcls], int((~torch.isfinite(plan.rt.flat_g)).sum()))
            break
        plan.optimizer_step()
stop.set()
