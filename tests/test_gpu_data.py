"""-m gpu: the loader-facing side of the step -- the uint8 NHWC import kernel, a StepPlan recorded for the loader's batches,
and YOLO.train / YOLO.val driven by a dataset YAML (fixture dataset of golden.cases.write_dataset)."""
import os

import numpy as np
import pytest
import torch

from golden.cases import write_dataset

pytestmark = pytest.mark.gpu


def _hsv_reference(img, r):
    """numpy restatement of RandomHSV's arithmetic on uint8 RGB (8-bit HSV with H in [0,180), the three LUTs, and back)."""
    c = img.astype(np.float32)
    R, G, B = c[..., 0], c[..., 1], c[..., 2]
    V, mn = c.max(-1), c.min(-1)
    d = V - mn
    with np.errstate(divide="ignore", invalid="ignore"):
        H = np.where(V == R, (G - B) / d, np.where(V == G, 2 + (B - R) / d, 4 + (R - G) / d)) * 30
        H = np.where(d > 0, H, 0)
        H = np.where(H < 0, H + 180, H)
        h8 = np.rint(H)
        h8 = np.where(h8 >= 180, h8 - 180, h8)
        s8 = np.where(V > 0, np.rint(255 * d / V), 0)
    hh = np.floor(np.fmod(h8 * np.float32(r[0]), 180)).astype(np.float32)
    ss = np.floor(np.minimum(s8 * np.float32(r[1]), 255)).astype(np.float32)
    vv = np.floor(np.minimum(V * np.float32(r[2]), 255)).astype(np.float32)
    hs, sf = hh / 30, ss / 255
    sec = np.floor(hs).astype(int) % 6
    f = hs - np.floor(hs)
    p, q, t = vv * (1 - sf), vv * (1 - sf * f), vv * (1 - sf * (1 - f))
    table = [(vv, t, p), (q, vv, p), (p, vv, t), (p, q, vv), (t, p, vv), (vv, p, q)]
    out = np.zeros_like(c)
    for k, (r_, g_, b_) in enumerate(table):
        m = sec == k
        out[..., 0][m], out[..., 1][m], out[..., 2][m] = r_[m], g_[m], b_[m]
    return np.clip(np.rint(out), 0, 255)


def test_import_u8_hsv_jitter_matches_numpy_restatement():
    from ultralytics.hip import check, lib
    n, h, w = 3, 24, 20
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
    x[0, 0, :4] = torch.tensor([[0, 0, 0], [255, 255, 255], [114, 114, 114], [200, 10, 10]], dtype=torch.uint8)
    gains = torch.tensor([[1.0, 1.0, 1.0], [1.012, 0.55, 1.31], [0.99, 1.6, 0.7]], device="cuda")
    y = torch.zeros((n, h, w, 8), dtype=torch.float16, device="cuda")
    check(lib().dy_import_image_u8(x.data_ptr(), y.data_ptr(), n, h, w, 8, None, None, gains.data_ptr(), None), "dy_import_image_u8")
    torch.cuda.synchronize()
    got = (y[..., :3].float().cpu().numpy() * 255).round()
    xs = x.cpu().numpy()
    for i in range(n):
        ref = _hsv_reference(xs[i], gains[i].cpu().numpy())
        d = np.abs(got[i] - ref)
        assert d.max() <= 1 and (d > 0).mean() < 0.01, (i, d.max(), (d > 0).mean())
    # unit gains are the identity up to the 8-bit HSV round trip (hue is quantised to 2 degrees: a few levels)
    assert np.abs(got[0] - xs[0]).max() <= 6 and np.abs(got[0] - xs[0]).mean() < 0.6


def test_import_u8_matches_float_division():
    from ultralytics.hip import check, lib
    for n, h, w in ((2, 5, 7), (1, 64, 64), (3, 33, 31)):
        x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
        y = torch.full((n, h, w, 8), 7.0, dtype=torch.float16, device="cuda")
        check(lib().dy_import_image_u8(x.data_ptr(), y.data_ptr(), n, h, w, 8, None, None, None, None), "dy_import_image_u8")
        torch.cuda.synchronize()
        ref = torch.zeros((n, h, w, 8), dtype=torch.float16, device="cuda")
        ref[..., :3] = (x.float() / 255).half()
        assert torch.equal(y, ref)
        # flips folded into the conversion: bit 0 = left-right, bit 1 = up-down, per image
        flip = torch.tensor([(i * 3 + 1) % 4 for i in range(n)], dtype=torch.uint8, device="cuda")
        y.fill_(7.0)
        check(lib().dy_import_image_u8(x.data_ptr(), y.data_ptr(), n, h, w, 8, flip.data_ptr(), None, None, None), "dy_import_image_u8")
        torch.cuda.synchronize()
        for i in range(n):
            f = int(flip[i])
            r = ref[i].flip(1) if f & 1 else ref[i]
            r = r.flip(0) if f & 2 else r
            assert torch.equal(y[i], r), (i, f)


def test_plan_recorded_for_u8_batches_equals_the_float_plan():
    from ultralytics.hip.train import StepPlan
    from ultralytics.nn.tasks import DetectionModel
    g = torch.Generator().manual_seed(3)
    B, S = 4, 128
    img = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g)
    n = 12
    lab = dict(batch_idx=torch.arange(B).repeat_interleave(3).float(), cls=torch.randint(0, 6, (n, 1), generator=g).float(),
               bboxes=torch.cat([torch.rand(n, 2, generator=g) * 0.6 + 0.2, torch.rand(n, 2, generator=g) * 0.2 + 0.05], 1))
    from ultralytics.hip import engine as E
    out = []
    for fmt in ("u8", "f32", "f32-direct"):
        # a float batch reaches the stem through the import kernel (same bits as the uint8 route) or -- the default -- is read by
        # the direct stem kernels (csrc/stem.hip): the same sums in another order, i.e. fp16 rounding noise on the stem output
        direct, E.STEM_DIRECT = E.STEM_DIRECT, fmt == "f32-direct" and E.STEM_DIRECT
        try:
            torch.manual_seed(0)
            m = DetectionModel("yolov8n-ASF-P2P2.yaml", verbose=False).cuda().train()
            plan = StepPlan(m, B, S, nmax=8, optimizer="SGD", use_graph=False, init_scale=1024.0)
            batch = {k: v.cuda() for k, v in lab.items()}
            batch["img"] = img.cuda() if fmt == "u8" else (img.permute(0, 3, 1, 2).float() / 255).cuda()
            plan.forward_backward(batch)
            torch.cuda.synchronize()
            out.append((plan.crit.scalars.clone(), plan.rt.flat_g.clone()))
        finally:
            E.STEM_DIRECT = direct
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert float((out[2][0][5:9] - out[0][0][5:9]).abs().max() / out[0][0][5:9].abs().max()) < 2e-4       # loss items: 2.0e-5 measured
    # gradients: 4.3e-2 (relative L2) measured here, and the SAME at full size (round 4, scratch measurement: 3.5e-2 at 640x640 batch 2,
    # 5.1e-2 at batch 8) -- so not BatchNorm over a few small images, as round 3 guessed: at random init the task-aligned top-10
    # choice is full of near-ties, and the last-bit noise the other summation order puts on ~2 % of the stem outputs flips a few
    # assignments; each flip re-routes a whole anchor's gradient while the loss VALUES barely move.  Bound: twice the measured value.
    assert float((out[2][1] - out[0][1]).norm() / out[0][1].norm()) < 9e-2
    with pytest.raises(TypeError):  # the recorded launch list reads ONE input format (this plan: float NCHW -> fine; u8 plan: not)
        plan_u8 = StepPlan(m, B, S, nmax=8, optimizer="SGD", use_graph=False)
        lab_d = {k: v.cuda() for k, v in lab.items()}
        plan_u8.forward_backward({**lab_d, "img": img.cuda()})
        plan_u8.forward_backward({**lab_d, "img": torch.rand(B, 3, S, S).cuda()})


def test_device_flips_equal_host_flips(tmp_path):
    """A loader that leaves the flips to the import kernel feeds the network the same pixels as one that flips on the host."""
    import random
    from types import SimpleNamespace
    from ultralytics.data import build_dataloader, build_yolo_dataset, check_det_dataset
    from ultralytics.hip.engine import Engine
    root = str(tmp_path / "ds")
    write_dataset(root)
    data = check_det_dataset(os.path.join(root, "data.yaml"))
    cfg = SimpleNamespace(imgsz=64, rect=False, cache=False, fraction=1.0, fliplr=0.5, flipud=0.5)
    eng = Engine("cuda:0")
    outs = []
    for on_dev, cache in ((False, False), (True, False), (True, "hbm")):  # host flips | kernel flips | kernel flips + HBM pool
        cfg.cache = cache
        ds = build_yolo_dataset(cfg, data["train"], 4, data, mode="train", flip_on_device=on_dev)
        random.seed(3)
        acts, labs = [], []
        for b in build_dataloader(ds, 4, 2, shuffle=True, device="cuda:0", drop_last=True):
            assert ("index" in b) == (cache == "hbm")
            a = eng.import_image_u8(b["img"].contiguous(), 8, b.get("flip"), b.get("index"))
            acts.append(a.st.buf.clone())
            labs.append(b["bboxes"].clone())
        outs.append((acts, labs))
    assert len(outs[0][0]) == 2
    for other in outs[1:]:
        for (a0, l0), (a1, l1) in zip(zip(*outs[0]), zip(*other)):
            assert torch.equal(a0, a1) and torch.equal(l0, l1)


def _host_warp(ds, j, aug, S):
    """numpy composition of one record: the canvas built by slicing like Mosaic._mosaic4 (or the letterboxed image), then one float
    bilinear sample per output pixel through the inverse (affine or projective) map with a 114 border, rounded to uint8 values."""
    from ultralytics.data.dataset import letterbox_geometry
    if aug["mosaic"] is not None:
        canvas = np.full((2 * S, 2 * S, 3), 114, np.float32)
        for i, x1a, y1a, x2a, y2a, x1b, y1b in ds.mosaic_layout(j, aug):
            im = ds.load_image(i)[0]
            canvas[y1a:y2a, x1a:x2a] = im[y1b:y1b + (y2a - y1a), x1b:x1b + (x2a - x1a)]
    else:
        im, _, (h, w) = ds.load_image(j)
        _, _, _, (top, bottom, left, right) = letterbox_geometry((h, w), (S, S), scaleup=True)
        canvas = np.full((S, S, 3), 114, np.float32)
        canvas[top:top + h, left:left + w] = im
    M = ds.affine_matrix(aug, canvas.shape[1], canvas.shape[0], (S, S))
    minv = np.linalg.inv(M.astype(np.float64)).astype(np.float32)
    oy, ox = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32), indexing="ij")
    den = minv[2, 0] * ox + minv[2, 1] * oy + minv[2, 2] if ds.perspective else np.float32(1.0)
    u = (minv[0, 0] * ox + minv[0, 1] * oy + minv[0, 2]) / den
    v = (minv[1, 0] * ox + minv[1, 1] * oy + minv[1, 2]) / den
    x0, y0 = np.floor(u).astype(int), np.floor(v).astype(int)
    ax, ay = (u - np.floor(u))[..., None], (v - np.floor(v))[..., None]
    padded = np.full((canvas.shape[0] + 4, canvas.shape[1] + 4, 3), 114, np.float32)
    padded[2:-2, 2:-2] = canvas

    def at(yy, xx):
        ok = (yy >= -2) & (yy < canvas.shape[0] + 2) & (xx >= -2) & (xx < canvas.shape[1] + 2)
        res = padded[np.clip(yy + 2, 0, padded.shape[0] - 1), np.clip(xx + 2, 0, padded.shape[1] - 1)]
        res[~ok] = 114
        return res

    top_ = at(y0, x0) * (1 - ax) + at(y0, x0 + 1) * ax
    bot_ = at(y0 + 1, x0) * (1 - ax) + at(y0 + 1, x0 + 1) * ax
    return np.clip(np.rint(top_ * (1 - ay) + bot_ * ay), 0, 255)


@pytest.mark.parametrize("extra", [{}, dict(mixup=0.6, perspective=0.0008, copy_paste=0.5)], ids=["mosaic+affine", "+mixup+perspective"])
def test_device_mosaic_affine_matches_host_composition(tmp_path, extra):
    """dy_warp_import_u8 (mosaic of four pool images + affine / perspective warp + MixUp blend + flips, sampled straight into the
    fp16 stem input) against a numpy composition of the same thing.  (The labels of this pipeline are pinned against the reference
    in tests/test_host_data.py; cv2's fixed-point interpolation is not reproduced.)"""
    import random
    from types import SimpleNamespace
    from ultralytics.data import build_dataloader, build_yolo_dataset, check_det_dataset
    from ultralytics.hip.engine import Engine
    root = str(tmp_path / "ds")
    write_dataset(root)
    data = check_det_dataset(os.path.join(root, "data.yaml"))
    S = 64
    cfg = SimpleNamespace(imgsz=S, rect=False, cache="hbm", fraction=1.0, fliplr=0.5, flipud=0.25, mosaic=0.7, degrees=8.0, translate=0.1,
                          scale=0.5, shear=3.0, **extra)
    eng = Engine("cuda:0")
    ds = build_yolo_dataset(cfg, data["train"], 4, data, mode="train", flip_on_device=True)
    ref_ds = build_yolo_dataset(cfg, data["train"], 4, data, mode="train", flip_on_device=True)
    loader = build_dataloader(ds, 4, 2, shuffle=True, device="cuda:0", drop_last=True)
    random.seed(21)
    np.random.seed(3)
    got = []
    for b in loader:
        assert "warp" in b and b["img"].shape[0] == len(ds)
        got.append((eng.import_warp(b["img"], b["warp"].contiguous(), 8).st.buf.clone(), b["bboxes"].clone()))
    random.seed(21)
    np.random.seed(3)
    order = build_dataloader(ref_ds, 4, 2, shuffle=True, drop_last=True)._indices()
    worst, mosaics, mixes = 0.0, 0, 0
    for bi, (act, boxes) in enumerate(got):
        out = (act[..., :3].float().cpu().numpy() * 255).round()
        assert (act[..., 3:] == 0).all()
        for k, j in enumerate(order[bi * 4:(bi + 1) * 4]):
            aug = ref_ds.draw_augment(j)
            mosaics += aug["mosaic"] is not None
            ref = _host_warp(ref_ds, j, aug, S)
            if "mix" in aug:  # MixUp._mix_transform: (img1 * r + img2 * (1 - r)).astype(np.uint8)
                mixes += 1
                i2, aug2, r = aug["mix"]
                ref = np.floor(ref.astype(np.float64) * r + _host_warp(ref_ds, i2, aug2, S).astype(np.float64) * (1 - r))
            if aug["flip"] & 1:
                ref = ref[:, ::-1]
            if aug["flip"] & 2:
                ref = ref[::-1]
            d = np.abs(out[k] - ref)
            worst = max(worst, float((d > 1).mean()))
            assert (d > 1).mean() < 5e-3 and (d > 0).mean() < 0.05, (bi, k, (d > 1).mean(), (d > 0).mean())
    assert mosaics >= 2 and (mixes >= 2 or not extra)


def test_train_with_device_side_mosaic(tmp_path):
    from ultralytics import YOLO
    root = str(tmp_path / "ds")
    write_dataset(root)
    y = YOLO("yolov8n-ASF-P2P2.yaml")
    hist = y.train(data=os.path.join(root, "data.yaml"), batch=4, imgsz=64, epochs=3, optimizer="SGD", workers=2, hipgraph=True, val=False,
                   cache="hbm", mosaic=1.0, degrees=5.0, translate=0.1, scale=0.5, shear=2.0, fliplr=0.5, mixup=0.3, copy_paste=0.3,
                   hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, perspective=0.0005, flipud=0.0, close_mosaic=1)  # the default gains + MixUp + perspective
    assert y.trainer.plan.warp is not None and np.isfinite(np.asarray(hist, dtype=np.float64)).all()
    assert y.trainer.train_loader.dataset.mosaic == 0.0  # closed for the last epoch, the affine / HSV / flip path went on
    # a batch larger than the dataset is clamped to it (data/build.py:104), and a mosaic's four images' labels fit the plan
    y = YOLO("yolov8n-ASF-P2P2.yaml")
    hist = y.train(data=os.path.join(root, "data.yaml"), batch=32, imgsz=64, epochs=2, optimizer="SGD", workers=2, val=False, cache="hbm",
                   mixup=0.0, copy_paste=0.0, perspective=0.0)  # everything else at the reference's defaults (mosaic 1.0, HSV, ...)
    assert y.trainer.plan.B == 9 and y.trainer.plan.nmax >= 8 and np.isfinite(np.asarray(hist, dtype=np.float64)).all()
    # colour jitter + flips through the streaming loader (no pool): gains travel as a (B,3) tensor into dy_import_image_u8
    y = YOLO("yolov8n-ASF-P2P2.yaml")
    hist = y.train(data=os.path.join(root, "data.yaml"), batch=4, imgsz=64, epochs=2, optimizer="SGD", workers=2, hipgraph=True, val=False,
                   mosaic=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0, fliplr=0.5, mixup=0.0, copy_paste=0.0,
                   hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, perspective=0.0, flipud=0.0)
    assert y.trainer.plan.hsv is not None and np.isfinite(np.asarray(hist, dtype=np.float64)).all()


def test_train_and_val_from_a_dataset_yaml(tmp_path):
    from ultralytics import YOLO
    root = str(tmp_path / "ds")
    write_dataset(root)
    zero = dict(mosaic=0.0, mixup=0.0, copy_paste=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0,
                perspective=0.0, flipud=0.0, fliplr=0.0)
    y = YOLO("yolov8n-ASF-P2P2.yaml")
    hist = y.train(data=os.path.join(root, "data.yaml"), batch=4, imgsz=64, epochs=3, optimizer="SGD", workers=2, hipgraph=True,
                   **{**zero, "fliplr": 0.5, "flipud": 0.25})
    assert y.model.model[-1].nc == 4  # rebuilt for the dataset's class count
    assert len(hist) == 3 and all(np.isfinite(np.asarray(h, dtype=np.float64)).all() for h in hist)
    assert y.trainer.plan.input_u8  # the loader's uint8 NHWC batches went through dy_import_image_u8
    assert y.trainer.plan.flip is not None  # ... which also applied the flips
    m = y.trainer.metrics
    assert set(m) >= {"metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)", "fitness"}
    assert y.trainer.validator.seen == 7
    m2 = y.val(data=os.path.join(root, "data.yaml"), batch=4)
    assert set(m2) == set(m)
    # the same run with the dataset resident in HBM: identical batches (same seeds) -> identical loss history
    hists = []
    for cache in (False, "hbm"):
        y2 = YOLO("yolov8n-ASF-P2P2.yaml")
        torch.manual_seed(0)  # the model is rebuilt for the dataset's 4 classes inside train(): same initial weights both times
        hists.append(y2.train(data=os.path.join(root, "data.yaml"), batch=4, imgsz=64, epochs=2, optimizer="SGD", workers=2, hipgraph=False,
                              val=False, cache=cache, **{**zero, "fliplr": 0.5}))
        assert (y2.trainer.plan.pool is not None) == (cache == "hbm")
    np.testing.assert_array_equal(np.asarray(hists[0], dtype=np.float64), np.asarray(hists[1], dtype=np.float64))


@pytest.mark.parametrize("cache", [False, "hbm"])
def test_only_the_consumer_thread_calls_into_hip(tmp_path, monkeypatch, cache):
    """The threading rule of the loader: decode workers and the producer thread do host work only; every
    device allocation, host->device copy, pinning and event comes from the thread that launches the step's hipGraphs.  (A
    second thread enqueuing copies while hipGraphLaunch runs corrupted a gradient buffer of the replayed graph on this ROCm:
    every optimizer step was skipped and the run trained nothing.)"""
    import threading
    from ultralytics.cfg import get_cfg
    from ultralytics.data import build_dataloader, build_yolo_dataset, check_det_dataset
    root = str(tmp_path / "ds")
    write_dataset(root)
    data = check_det_dataset(os.path.join(root, "data.yaml"))
    geo = dict(mosaic=1.0, translate=0.1, scale=0.5) if cache else dict(mosaic=0.0, translate=0.0, scale=0.0)  # device-side composition
    args = get_cfg(overrides=dict(imgsz=64, cache=cache, fliplr=0.5, hsv_h=0.015, hsv_s=0.7, hsv_v=0.4, mixup=0.0, copy_paste=0.0,
                                  perspective=0.0, **geo))
    ds = build_yolo_dataset(args, data["train"], 4, data, mode="train", layout="nhwc", flip_on_device=True)
    loader = build_dataloader(ds, 4, 2, shuffle=True, device=torch.device("cuda", 0), drop_last=True)
    main = threading.get_ident()
    offenders = []

    def watch(obj, name):
        orig = getattr(obj, name)

        def wrapped(*a, **k):
            out = orig(*a, **k)
            if threading.get_ident() != main and (name != "to" or getattr(out, "is_cuda", False)):  # .to(dtype) on the host is fine
                offenders.append(name)
            return out
        monkeypatch.setattr(obj, name, wrapped)

    for obj, name in ((torch.Tensor, "to"), (torch.Tensor, "cuda"), (torch.Tensor, "pin_memory"), (torch.Tensor, "record_stream"),
                      (torch.cuda.Event, "record"), (torch.cuda.Event, "synchronize"), (torch.cuda.Event, "wait"),
                      (torch.cuda.Stream, "wait_event"), (torch.cuda.Stream, "synchronize"), (torch.cuda, "synchronize"),
                      (torch.cuda, "current_stream")):
        watch(obj, name)
    seen = 0
    for epoch in range(2):
        loader.set_epoch(epoch)
        for b in loader:
            assert b["img"].is_cuda and all(b[k].is_cuda for k in ("flip", "hsv", "index", "warp") if k in b)
            seen += 1
    assert seen == 2 * len(loader) and seen >= 4
    assert not offenders, f"device API calls from a loader thread: {sorted(set(offenders))}"


@pytest.mark.parametrize("cache", [False, "hbm"])
def test_ragged_last_batch_is_trained_on_like_the_reference(tmp_path, cache):
    """9 training images at batch 4 -> batches of 4, 4 and 1 per epoch (reference data/build.py:104-124 has no drop_last, so
    nb = ceil(n / batch) also sets the warm-up length and the schedule).  The batch of 1 runs through its own recorded launch
    list; gradient accumulation, the optimizer state and the step counters are the main plan's."""
    from ultralytics import YOLO
    root = str(tmp_path / "ds")
    write_dataset(root)
    zero = dict(mosaic=0.0, mixup=0.0, copy_paste=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0,
                perspective=0.0, flipud=0.0, fliplr=0.5)
    y = YOLO("yolov8n-ASF-P2P2.yaml")
    hist = y.train(data=os.path.join(root, "data.yaml"), batch=4, imgsz=64, epochs=3, optimizer="SGD", workers=2, hipgraph=True, val=False,
                   cache=cache, nbs=4, amp=False, **zero)
    tr = y.trainer
    assert len(tr.train_loader) == 3 and tr.nb == 3 and sorted(tr.plans) == [1, 4]
    assert np.isfinite(np.asarray(hist, dtype=np.float64)).all()
    taken, skipped, _ = tr.plan.check_progress()
    assert (taken, skipped) == (9, 0) and tr.plan.opt_calls == 9  # nbs = batch: one optimizer step per batch, the tail's included
    assert tr.plans[1].state is tr.plan.state and tr.plans[1].ema is tr.plan.ema and tr.plans[1].crit.scalars is tr.plan.crit.scalars
