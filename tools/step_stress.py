"""The whole recorded training step under co-tenancy: re-run from restored state and compared bit for bit with its first result
(usage: step_stress.py <tag> <iterations>; run beside `tools/loss_stress.py heavy1 <n>` as the busy co-tenant; env B, S, WIOU, NWD).
A step that does not repeat names the parameters whose gradients differ (round 4: the loss kernel's transient)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import ultralytics.hip  # noqa: E402,F401
import torch  # noqa: E402
from bench import CFG, synth_batch  # noqa: E402
from ultralytics.hip.train import StepPlan  # noqa: E402
from ultralytics.nn.tasks import DetectionModel  # noqa: E402

tag, iters = sys.argv[1], int(sys.argv[2])
torch.manual_seed(0)
name = os.environ.get("MODEL")
cfg = CFG if not name else os.path.join(os.path.dirname(CFG), name + ".yaml")
model = DetectionModel(cfg, verbose=False).cuda().train()
B, S = int(os.environ.get("B", "4")), int(os.environ.get("S", "320"))
plan = StepPlan(model, B, S, nmax=8, use_graph=False)
batch = {k: v.cuda() for k, v in synth_batch(1, B, S, 6).items()}
plan.crit.bbox_loss.use_wiseiou, plan.crit.bbox_loss.nwd_loss = os.environ.get("WIOU", "1") == "1", os.environ.get("NWD", "1") == "1"
plan.set_hyper([0.0005] * 3, 0.937, [0, 5e-4, 0])
b0, s0 = plan.rt.flat_b.clone(), plan.crit.scalars.clone()
plan.forward_backward(batch)
torch.cuda.synchronize()
want = plan.rt.flat_g.clone()
offs = sorted((o, n) for n, o in plan.rt.param_off.items())
bad, t0 = 0, time.time()
for it in range(iters):
    plan.rt.flat_b.copy_(b0); plan.crit.scalars.copy_(s0)
    plan.eng.replay(plan.rec_fb)
    torch.cuda.synchronize()
    if not torch.equal(plan.rt.flat_g, want):
        bad += 1
        if bad <= 5:
            d = (plan.rt.flat_g - want).abs()
            names = [n for k, (o, n) in enumerate(offs) if float(d[o:(offs[k + 1][0] if k + 1 < len(offs) else d.numel())].max()) > 0]
            print(tag, "iter", it, "rel", float(d.norm() / want.norm()), len(names), "parameters differ, e.g.", names[:4], flush=True)
print(tag, "steps", iters, "odd", bad, f"{time.time() - t0:.1f}s", flush=True)
