"""-m gpu: the end-to-end protocol of SURVEY.md section 8(c).  tests/golden/e2e_trainer.npz holds the results.csv of the
reference's UNMODIFIED trainer (its own dataset reader, loader, loss, optimizer, warm-up, EMA, validator; CPU, fp32) on the
synthetic set of golden.cases.write_e2e_dataset: 16 + 16 images of 640x640, 40 epochs, batch 2, no augmentation.  Here the
same files and overrides go through this package's public API -- YOLO(yaml).train(data=...) -- on the GPU, and the north-star
clause is checked: mAP50 on the held-out images within 0.2 of the reference.  The run reproduces the reference's starting point
and data order -- init_seeds(seed + 1 + RANK), a model built after it in the reference's construction order (same draws from the
global RNG, same BN buffers), the DataLoader generator seed -- so the first epochs are compared number for number (measured: epoch 1
and 2 within 0.4 %); fp16 activations against fp32 then decorrelate the two trajectories (3 % at epoch 3, 5-10 % later) and the
rest of the curve and the final metrics are compared statistically."""
import os

import numpy as np
import pytest
import torch

from golden.cases import E2E, write_e2e_dataset

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,fixture", [("yolov8n-ASF-P2P2.yaml", "e2e_trainer.npz"), ("yolov8n-LD-P2.yaml", "e2e_trainer_ld.npz")],
                         ids=["DEAL-YOLO-N", "LD"])
def test_public_api_training_reaches_the_reference_trainers_map(tmp_path, model, fixture):
    from ultralytics import YOLO
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", fixture))
    head = [str(h) for h in G["header"]]
    ref = {h: G["results"][:, j] for j, h in enumerate(head)}
    root = str(tmp_path / "e2e")
    write_e2e_dataset(root)
    zero = dict(mosaic=0.0, mixup=0.0, copy_paste=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0,
                perspective=0.0, flipud=0.0, fliplr=0.0)
    torch.manual_seed(0)
    y = YOLO(model)
    w0 = {k: v.detach().clone() for k, v in y.model.state_dict().items() if k.startswith("model.0.") and v.dtype.is_floating_point}
    hist = y.train(data=os.path.join(root, "data.yaml"), epochs=E2E["epochs"], batch=E2E["batch"], imgsz=E2E["imgsz"], workers=2,
                   optimizer="SGD", amp=False, val=True, close_mosaic=0, seed=0, deterministic=True, log_every=1, **zero)
    # amp=False as in the reference run: loss scale 1, no skipped steps.  (With the dynamic loss scale a run this short -- ~30
    # optimizer steps, gradient accumulation over 32 batches -- loses 5 of them to the scale search on the LD model.)
    hist = np.asarray(hist, dtype=np.float64)
    m = y.trainer.metrics
    st = y.trainer.plan.state.cpu().numpy()
    print(f"optimizer steps taken {st[5]:.0f}, skipped {st[6]:.0f}, loss scale {st[0]:.0f}, last grad norm {st[3]:.3f}, "
          f"nonfinite grads {int((~torch.isfinite(y.trainer.plan.rt.flat_g)).sum())}")
    plan = y.trainer.plan
    # every optimizer_step() the trainer issued took effect: none skipped at amp=False, the device counter advanced with the host's
    assert st[5] > 0 and st[6] == 0 and st[5] == plan.opt_calls and st[0] == 1.0, st.tolist()
    w1 = y.trainer.model.state_dict()
    moved = {k: float((w1[k].cpu() - w0[k].cpu()).abs().max()) for k in w0 if "running" not in k}
    assert all(v > 0 for v in moved.values()), f"first-layer parameters that never moved: {[k for k, v in moved.items() if v == 0]}"
    keys = ("train/box_loss", "train/cls_loss", "train/dfl_loss")
    dev = np.array([[abs(hist[e][j] - ref[k][e]) / ref[k][e] for j, k in enumerate(keys)] for e in range(E2E["epochs"])])
    print("relative deviation of the per-epoch mean losses from the reference's results.csv: epoch 1", np.round(dev[0], 4), " worst over epochs 1-5",
          np.round(dev[:5].max(0), 4), " worst over all", np.round(dev.max(0), 4))
    print("per-epoch deviation, epochs 1-10:\n", np.round(dev[:10], 4))
    for e in (0, 9, 19, 29, 39):
        print(f"epoch {e + 1:2d}  ours box/cls/dfl {np.round(hist[e], 3)}   reference "
              f"{[round(float(ref[k][e]), 3) for k in ('train/box_loss', 'train/cls_loss', 'train/dfl_loss')]}")
    print(f"held-out: ours P {m['metrics/precision(B)']:.3f} R {m['metrics/recall(B)']:.3f} mAP50 {m['metrics/mAP50(B)']:.3f} "
          f"mAP50-95 {m['metrics/mAP50-95(B)']:.3f}   reference P {ref['metrics/precision(B)'][-1]:.3f} R {ref['metrics/recall(B)'][-1]:.3f} "
          f"mAP50 {ref['metrics/mAP50(B)'][-1]:.3f} mAP50-95 {ref['metrics/mAP50-95(B)'][-1]:.3f}")
    assert hist.shape == (E2E["epochs"], 3) and np.isfinite(hist).all()
    # measured: epoch 1 0.3 % (N) / 0.13 % (LD); epoch 2 0.4 % ... 1.1 % (N, over this round's trees: a different pixel-to-block
    # partition of a BatchNorm sum moves last bits, the first optimizer steps amplify them) / 1.9 % (LD: LDConv's floor() turns fp16
    # roundings into moved samples)
    assert dev[0].max() < 1e-2 and dev[1].max() < (3e-2 if "LD" in model else 2e-2), "epochs 1 and 2: same initial weights, same batches"
    assert dev[:5].max() < 0.12, "epochs 3-5: decorrelating (measured up to 8 %)"
    rl = np.array([ref[k][-1] for k in ("train/box_loss", "train/cls_loss", "train/dfl_loss")])
    assert np.all(np.abs(hist[-1] - rl) / rl < 0.25), "last-epoch mean losses"
    assert abs(m["metrics/mAP50(B)"] - ref["metrics/mAP50(B)"][-1]) < 0.2  # the north-star bound
