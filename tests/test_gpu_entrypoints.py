"""-m gpu: the reference's entry-point call shapes against this package: ``detect.py`` --
``YOLO(weights).predict(source=<directory>, imgsz=640, project=..., name=..., verbose=True)``; ``get_FPS.py`` -- ``select_device`` /
``attempt_load_weights(weights, device=device, fuse=True)`` / ``model.fuse()`` / ``model.half()`` / timed forwards; ``train.py`` --
``YOLO(yaml).train(data=..., device='0', ...)`` and the multi-GPU form ``device='0,1'`` that re-launches itself under
torch.distributed.run (here: two ranks sharing the one GPU over gloo, DY_REHEARSE_ON_ONE_GPU=1)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from golden.cases import write_dataset
from oracle import metrics as om

pytestmark = pytest.mark.gpu
CKPT = os.path.join(ROOT, "tests", "golden", "ref_ckpt.pt")


def _write_images(d, shapes, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(d, exist_ok=True)
    paths = []
    for i, (h, w) in enumerate(shapes):
        p = os.path.join(d, f"im{i:02d}.png")
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    return paths


def test_predict_on_a_directory_like_the_reference_detect_script(tmp_path):
    from ultralytics import YOLO
    from ultralytics.engine.predictor import letterbox
    from ultralytics.utils import ops
    shapes = [(480, 640), (480, 640), (300, 200), (720, 1280)]
    paths = _write_images(str(tmp_path / "images"), shapes)
    open(os.path.join(tmp_path, "images", "notes.txt"), "w").write("not an image")
    model = YOLO(CKPT)  # a reference-format checkpoint: LetterBox(auto=True) for same-shaped batches, as with the reference's .pt
    res = model.predict(source=str(tmp_path / "images"), imgsz=640, project="runs/detect", name="exp", verbose=True, conf=0.001)
    assert [r.path for r in res] == paths and [r.orig_shape for r in res] == shapes
    from PIL import Image
    for r, p, (h, w) in zip(res, paths, shapes):
        img = np.asarray(Image.open(p).convert("RGB"))
        assert r.orig_img.shape == (h, w, 3) and r.boxes.data.shape[1] == 6 and len(r.boxes) <= 300
        # the same image through the public tensor API by hand: letterbox (one image = "same shapes" -> minimum rectangle), forward,
        # soft-NMS, ops.scale_boxes; boxes inside the image
        lb = letterbox(img, (640, 640), auto=True, stride=32)
        x = torch.from_numpy(lb.transpose(2, 0, 1)[None].copy()).float() / 255
        y, _ = model.model(x.cuda())
        det = ops.non_max_suppression(y, 0.001, 0.7, max_det=300)[0].cpu()
        want = torch.from_numpy(om.scale_boxes(lb.shape[:2], det[:, :4].numpy(), (h, w)))
        assert r.boxes.xyxy.shape == want.shape, (r.boxes.xyxy.shape, want.shape)
        print(p, "max |box diff|", float((r.boxes.xyxy.cpu() - want).abs().max()), "conf diff", float((r.boxes.conf.cpu() - det[:, 4]).abs().max()))
        assert torch.allclose(r.boxes.xyxy.cpu(), want, atol=1e-3) and torch.equal(r.boxes.cls.cpu(), det[:, 5])
        b = r.boxes.xyxy
        assert float(b[:, [0, 2]].min()) >= 0 and float(b[:, [0, 2]].max()) <= w and float(b[:, [1, 3]].max()) <= h
        assert r.boxes.xywhn.shape == (len(r), 4) and isinstance(r.verbose(), str)
    # other source kinds: one file, a list, a BGR array, a tensor; and YOLO.__call__
    one = model(paths[2], conf=0.001)
    assert len(one) == 1 and torch.equal(one[0].boxes.data, res[2].boxes.data)
    mixed = model.predict(source=[paths[0], np.zeros((64, 96, 3), np.uint8)], conf=0.001)
    assert [m.orig_shape for m in mixed] == [(480, 640), (64, 96)]
    t = model.predict(torch.rand(2, 3, 64, 64), conf=0.001)
    assert len(t) == 2 and t[0].orig_shape == (64, 64)
    with pytest.raises(FileNotFoundError):
        model.predict(source=str(tmp_path / "nothing_here"))


def test_get_fps_protocol_of_the_reference():
    """reference get_FPS.py:40-75 line by line (with 3 + 5 iterations instead of 200 + 1000)."""
    from ultralytics import YOLO
    from ultralytics.nn.tasks import attempt_load_weights
    from ultralytics.utils.torch_utils import select_device
    device = select_device("0", batch=8)
    model = attempt_load_weights(CKPT, device=device, fuse=True)
    model = model.to(device)
    model.fuse()
    example_inputs = torch.randn((8, 3, 640, 640)).to(device)
    ref = model(example_inputs)[0]
    model = model.half()
    example_inputs = example_inputs.half()
    for _ in range(3):
        model(example_inputs)
    torch.cuda.synchronize()
    y = model(example_inputs)[0]
    torch.cuda.synchronize()
    assert y.shape == (8, 10, 33600) and torch.isfinite(y).all()
    assert float((y.float() - ref.float()).abs().max()) <= 2e-2 * float(ref.abs().max())  # half() only re-rounds the fp32 masters
    assert YOLO("yolov8n-p2.yaml").model is not None  # the '.yaml' branch of the script


def test_train_call_of_the_reference_and_the_multi_gpu_relaunch(tmp_path):
    """train.py's call (device='0') and the device list form: device='0,1' spawns torch.distributed.run with one rank per entry
    before this process touches the GPU -- so it runs in a fresh interpreter here."""
    root = str(tmp_path / "ds")
    write_dataset(root)
    code = f"""
import sys, json
sys.path.insert(0, {os.path.join(ROOT, 'experiment-yolo_amd')!r})
from ultralytics import YOLO
zero = dict(mosaic=0.0, mixup=0.0, copy_paste=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0,
            perspective=0.0, flipud=0.0, fliplr=0.0)
model = YOLO('yolov8n-ASF-P2P2.yaml')
hist = model.train(data={os.path.join(root, 'data.yaml')!r}, cache=False, imgsz=64, epochs=2, batch=4, close_mosaic=10, workers=2,
                   device=DEVICE, optimizer='SGD', project={str(tmp_path / 'runs')!r}, name='exp', val=False, **dict(zero, **EXTRA))
rm = model.model.state_dict()['model.0.bn.running_mean']  # zero at construction: moved only by training steps
print('TRAINED', float(rm.abs().sum()) > 0.0, model.trainer is not None)
import torch, os
last = os.path.join({str(tmp_path / 'runs')!r}, 'exp', 'weights', 'last.pt')
if getattr(model, 'ddp_result', None) is not None and os.path.exists(last):
    # after the re-launch the parent holds what YOLO(last.pt) would: the checkpoint's EMA weights (reference nn/tasks.py:780,
    # ``ckpt.get('ema') or ckpt['model']``), not the raw trained ones
    ck = torch.load(last, map_location='cpu', weights_only=False)
    sd = model.model.state_dict()
    same = all(torch.equal(sd[k].cpu().float(), v.float()) for k, v in ck['ema'].items())
    differs = any(not torch.equal(ck['ema'][k].float(), ck['model'][k].float()) for k in ck['ema'])
    print('EMA_HANDOVER', same, differs)
print('RESULT', json.dumps([[float(x) for x in h] for h in hist]), getattr(model, 'ddp_result', None) is not None)
"""
    outs = {}
    for dev, env, extra in (("'0'", {}, "{}"), ("'0,1'", {"DY_REHEARSE_ON_ONE_GPU": "1"}, "{}"), ("'0,1' ", {"DY_REHEARSE_ON_ONE_GPU": "1"}, "dict(multi_scale=True)")):
        p = subprocess.run([sys.executable, "-c", code.replace("DEVICE", dev).replace("EXTRA", extra)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")][-1]
        outs[dev] = (np.array(eval(line.split(" ", 1)[1].rsplit(" ", 1)[0])), line.endswith("True"), p.stderr)
        # whichever way the run was launched, the caller's model object holds TRAINED weights afterwards (after the re-launch: rank
        # 0's checkpoint loaded back, reference engine/model.py:612-616) -- m.val() / m.predict() never see the untouched copy
        assert [l for l in p.stdout.splitlines() if l.startswith("TRAINED")][-1].split()[1] == "True", p.stdout[-2000:]
        if dev.startswith("'0,1'"):
            assert [l for l in p.stdout.splitlines() if l.startswith("EMA_HANDOVER")][-1].split()[1:] == ["True", "True"], p.stdout[-2000:]
    single, ddp = outs["'0'"], outs["'0,1'"]
    assert single[0].shape == (2, 3) and np.isfinite(single[0]).all() and not single[1]
    assert ddp[1], "device='0,1' must have gone through the torch.distributed.run re-launch"
    assert ddp[0].shape == (2, 3) and np.isfinite(ddp[0]).all()
    assert "torch.distributed.run" in ddp[2] and "--nproc_per_node=2" in ddp[2]
    ms = outs["'0,1' "]  # the same re-launch with multi_scale=True: every rank draws its own sizes (seed + 1 + RANK) and traces its own
    assert ms[1] and ms[0].shape == (2, 3) and np.isfinite(ms[0]).all()  # plans; the step's one collective keeps the ranks in step
