"""The loss launch alone under co-tenancy: usage  loss_stress.py <tag> <iterations>   (tag heavy*: full training steps instead, the co-tenant)
see tools/loss_stress.sh; env B, S (batch, image size), WIOU, NWD."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import ultralytics.hip
import torch
from bench import CFG, synth_batch
from ultralytics.hip.train import StepPlan
from ultralytics.nn.tasks import DetectionModel
tag, iters = sys.argv[1], int(sys.argv[2])
torch.manual_seed(0)
model = DetectionModel(CFG, verbose=False).cuda().train()
B, S = int(os.environ.get("B", "4")), int(os.environ.get("S", "320"))
plan = StepPlan(model, B, S, nmax=8, use_graph=False)
batch = {k: v.cuda() for k, v in synth_batch(1, B, S, 6).items()}
plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = os.environ.get("WIOU", "1") == "1", os.environ.get("NWD", "1") == "1"
plan.set_hyper([0.0005] * 3, 0.937, [0, 5e-4, 0])
s0 = plan.crit.scalars.clone()
plan.forward_backward(batch)
torch.cuda.synchronize()
ops = plan.rec_fb.ops
if tag.startswith("heavy"):
    t0 = time.time()
    for it in range(iters):
        plan.eng.replay(plan.rec_fb)
    torch.cuda.synchronize()
    print(tag, "full steps", iters, f"{time.time() - t0:.1f}s", flush=True)
    sys.exit(0)
li = [i for i, o in enumerate(ops) if o[0] is not None and o[2] == "dy_detection_loss"][0]
dbox = plan.ho.dbox
def run_loss():
    plan.crit.scalars.copy_(s0)
    plan.eng.replay(plan.rec_fb, li, li + 1)
run_loss(); torch.cuda.synchronize()
ref = [t.clone() for t in dbox]
refs = plan.crit.scalars.clone()
bad, t0 = 0, time.time()
for it in range(iters):
    run_loss()
    torch.cuda.synchronize()
    if not all(torch.equal(a, b) for a, b in zip(dbox, ref)):
        bad += 1
        for l, (a, b) in enumerate(zip(dbox, ref)):
            nz = (a.view(-1) != b.view(-1)).nonzero().view(-1)
            if nz.numel():
                print(tag, "iter", it, "level", l, "differing", nz.numel(), "first", int(nz[0]), "ch", int(nz[0]) % 64, "scalars equal", bool(torch.equal(plan.crit.scalars, refs)), flush=True)
print(tag, "loss executions", iters, "odd", bad, f"{time.time() - t0:.1f}s", flush=True)
