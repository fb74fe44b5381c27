"""-m gpu: the training loop (warm-up, accumulate, all-reduce hook, optimizer + EMA, hipGraph replay) learns on a synthetic
source, the EMA export and the predict / NMS facade run, and the vanilla P2 model of BASELINE.json configs[4] runs fused."""
import os

import pytest
import torch

from conftest import CFG_DIR

pytestmark = pytest.mark.gpu


def test_trainer_learns_and_exports():
    from ultralytics import YOLO
    from ultralytics.data import SyntheticDetection
    torch.manual_seed(0)
    y = YOLO(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"))
    src = SyntheticDetection(n_batches=12, batch=4, imgsz=64, boxes_per_image=3, wh=(0.1, 0.4), seed=3)
    hist = y.train(data=src, batch=4, imgsz=64, epochs=3, optimizer="SGD", warmup_epochs=0.0, lr0=0.01, nbs=4, hipgraph=True)
    first, last = float(sum(hist[0])), float(sum(hist[-1]))
    assert all(torch.isfinite(h).all() for h in hist)
    assert last < first, (first, last)  # it learns something in 36 steps
    plan = y.trainer.plan
    assert float(plan.state[5]) >= 30  # optimizer steps actually taken (a few may be skipped while the loss scale settles)
    ema = y.trainer.ema.ema
    w = dict(y.model.named_parameters())["model.0.conv.weight"]
    we = dict(ema.named_parameters())["model.0.conv.weight"]
    assert we.shape == w.shape and not torch.equal(we.to(w.device), w)
    out = y.predict(torch.rand(2, 3, 64, 64), conf=0.001)
    assert len(out) == 2 and all(o.boxes.data.shape[1] == 6 and o.orig_shape == (64, 64) for o in out)


def test_p2_model_fused_inference():
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-p2.yaml"), verbose=False).cuda().eval()
    m.fuse()
    with torch.no_grad():
        y, feats = m(torch.rand(2, 3, 128, 128).cuda())
    A = 32 * 32 + 16 * 16 + 8 * 8 + 4 * 4
    assert y.shape == (2, 84, A) and len(feats) == 4 and torch.isfinite(y).all()


def _tiny_plan(**kw):
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    torch.manual_seed(0)
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"), verbose=False).cuda().train()
    plan = StepPlan(m, 2, 64, nmax=8, use_graph=True, **kw)
    batch = dict(img=torch.rand(2, 3, 64, 64), batch_idx=torch.tensor([0., 0., 1.]), cls=torch.tensor([[1.], [2.], [3.]]),
                 bboxes=torch.tensor([[.5, .5, .3, .3], [.3, .6, .2, .2], [.6, .4, .4, .3]]))
    return plan, batch


def test_steps_that_do_not_take_effect_fail_loudly():
    """A run whose optimizer steps are all skipped (non-finite gradients) or not executed must raise where the trainer already
    synchronises, not show up as a flat loss curve (the silent no-training run of round 1)."""
    plan, batch = _tiny_plan(init_scale=1.0, dynamic_scale=False)  # amp=False
    for _ in range(3):
        plan.step(batch, [0.01] * 3, 0.9, [0.0, 5e-4, 0.0])
    assert plan.check_progress()[:2] == (3, 0)
    plan.rt.flat_g[5] = float("nan")  # what a corrupted gradient buffer looks like to the optimizer
    plan.set_hyper([0.01] * 3, 0.9, [0.0] * 3)
    plan.optimizer_step()
    assert float(plan.state[0]) == 1.0, "amp=False: the loss scale is a constant"
    with pytest.raises(RuntimeError, match="skipped for non-finite gradients"):
        plan.check_progress()
    plan2, batch = _tiny_plan(init_scale=1.0, dynamic_scale=False)
    plan2.step(batch, [0.01] * 3, 0.9, [0.0] * 3)
    plan2.opt_calls += 1  # a recorded optimizer launch that never ran on the device
    with pytest.raises(RuntimeError, match="out of step with the host"):
        plan2.check_progress()
    plan3, batch = _tiny_plan(init_scale=4.0, dynamic_scale=True)  # amp=True: a scale search is normal, a collapse is not
    plan3.forward_backward(batch)
    for _ in range(4):
        plan3.rt.flat_g[5] = float("inf")
        plan3.set_hyper([0.01] * 3, 0.9, [0.0] * 3)
        plan3.optimizer_step()
    with pytest.raises(RuntimeError, match="loss scale collapsed"):
        plan3.check_progress()


def test_more_labels_than_capacity_raises():
    """one image above nmax while the batch total stays under B*nmax used to train on silently truncated labels."""
    plan, batch = _tiny_plan(init_scale=1.0)
    n = 10  # nmax = 8, B*nmax = 16
    batch = dict(img=batch["img"], batch_idx=torch.zeros(n), cls=torch.zeros(n, 1),
                 bboxes=torch.cat([torch.rand(n, 2) * 0.5 + 0.25, torch.rand(n, 2) * 0.2 + 0.05], 1))
    plan.forward_backward(batch)
    with pytest.raises(RuntimeError, match="more than nmax=8 labels"):
        plan.loss_items()
    plan.forward_backward(dict(batch, batch_idx=torch.tensor([0., 1.] * 5)))  # five per image: fits, and the flag was cleared
    plan.loss_items()


def test_resume_continues_from_the_saved_optimizer_state(tmp_path):
    """reference engine/trainer.py:1050-1105: last.pt after every epoch (when project/name name a run folder) carries the optimizer
    state; ``YOLO(last.pt).train(resume=True)`` restores weights, momentum, EMA + update count, loss scale and step counters and
    continues at the next epoch of the schedule."""
    from ultralytics import YOLO
    from ultralytics.data import SyntheticDetection
    src = SyntheticDetection(n_batches=6, batch=4, imgsz=64, boxes_per_image=3, wh=(0.1, 0.4), seed=3)
    kw = dict(batch=4, imgsz=64, optimizer="SGD", warmup_epochs=0.0, lr0=0.01, nbs=4, hipgraph=True, amp=False, project=str(tmp_path), name="run")
    y = YOLO(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"))
    y.train(data=src, epochs=2, **kw)
    last = tmp_path / "run" / "weights" / "last.pt"
    assert last.exists()
    ck = torch.load(last, map_location="cpu", weights_only=False)
    assert ck["epoch"] == 1 and ck["updates"] == 12 and float(ck["optimizer"]["flat"]["state"][5]) == 12
    saved = {k: v.clone() for k, v in ck["optimizer"]["flat"].items()}
    import shutil
    keep = tmp_path / "epoch2.pt"  # the resumed run below overwrites last.pt after each of its epochs
    shutil.copy(last, keep)
    y2 = YOLO(str(last))
    hist = y2.train(data=src, resume=True, epochs=4, **kw)
    tr = y2.trainer
    assert tr.start_epoch == 2 and len(hist) == 2  # epochs 3 and 4 of 4
    st = tr.plan.state.cpu()
    assert float(st[5]) == 24 and tr.plan.ema_updates == 24 and tr.plan.opt_calls == 24
    assert not torch.equal(tr.plan.mom.cpu(), saved["mom"])  # ... and it kept training
    # the state the resumed run STARTED from is the saved one: replay the restore on a fresh trainer and compare
    y3 = YOLO(str(keep))
    y3.train(data=src, resume=True, epochs=2, **dict(kw, name="other"))  # nothing left to do: restores and returns
    assert y3.trainer.start_epoch == 2
    assert torch.equal(y3.trainer.plan.mom.cpu(), saved["mom"]) and torch.equal(y3.trainer.plan.ema.cpu(), saved["ema"])
    assert torch.equal(y3.trainer.plan.rt.flat_p.cpu(), saved["p"]) and torch.equal(y3.trainer.plan.state.cpu(), saved["state"])
    with pytest.raises(ValueError, match="no resumable optimizer state"):
        YOLO(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")).train(data=src, resume=os.path.join(os.path.dirname(CFG_DIR), "..", "..", "..", "tests", "golden", "ref_ckpt.pt"), **kw)


def test_soap_is_selectable_by_name_and_trains():
    """optimizer='SOAP' (reference engine/trainer.py:1156-1165; the optimizer itself is pinned against the reference's class on
    the CPU, tests/test_host_logic.py): unscale + clip + SOAP over the flat parameter views, EMA / counters through the kernels."""
    from ultralytics import YOLO
    from ultralytics.data import SyntheticDetection
    src = SyntheticDetection(n_batches=10, batch=4, imgsz=64, boxes_per_image=3, wh=(0.1, 0.4), seed=3)
    y = YOLO(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"))
    hist = y.train(data=src, batch=4, imgsz=64, epochs=3, optimizer="SOAP", warmup_epochs=0.0, lr0=0.003, nbs=4, amp=False)
    plan = y.trainer.plan
    assert plan.soap and not plan.use_graph and plan._soap is not None
    taken, skipped, _ = plan.check_progress()
    assert (taken, skipped) == (30, 0) and all(torch.isfinite(h).all() for h in hist)
    assert float(sum(hist[-1])) < float(sum(hist[0]))
    st = plan._soap.state[0]  # model.0.conv.weight: a Gram matrix and an eigenbasis per dimension, refreshed twice by now
    assert st.step == 29 and st.q is not None and sum(q is not None for q in st.q) == 4
    assert plan._soap.state[1].q == [None]  # its BatchNorm weight: 1-D tensors run plain Adam
    with pytest.raises(NotImplementedError, match="not found in list of available optimizers"):
        YOLO(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")).train(data=src, batch=4, imgsz=64, epochs=1, optimizer="LAMB")
