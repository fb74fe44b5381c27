"""The recorded eval forward under co-tenancy: repeated and compared bit for bit with its first result
(usage: infer_stress.py <model yaml> <batch> <imgsz> <iterations> [fuse]; run beside `tools/loss_stress.py heavy1 <n>`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch  # noqa: E402
from ultralytics import YOLO  # noqa: E402

name, B, S, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
m = YOLO(name).model.cuda().eval()
if len(sys.argv) > 5 and sys.argv[5] == "fuse":
    m.fuse()
x = torch.rand(B, 3, S, S, device="cuda")
with torch.no_grad():
    for _ in range(3):
        want = m(x)[0].clone()
    bad, t0 = 0, time.time()
    for it in range(iters):
        y = m(x)[0]
        if not torch.equal(y, want):
            bad += 1
            if bad <= 3:
                d = (y - want).abs()
                print("iter", it, "differing", int((d > 0).sum()), "max", float(d.max()), flush=True)
print(name, "forwards", iters, "odd", bad, f"{time.time() - t0:.1f}s", flush=True)
