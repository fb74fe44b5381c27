"""CPU: host-side logic of the drop-in package -- YAML graph building, parameter naming/counts against the reference's
(golden), optimizer grouping, Concat in-place planning, cfg loading.  No kernels are launched."""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_DIR

MODELS = ["yolov8n-ASF-P2P2", "yolov8n-LD-P2", "yolov8n-p2"]


@pytest.mark.parametrize("name", MODELS)
def test_state_dict_matches_reference(golden, name):
    from ultralytics.nn.tasks import DetectionModel
    G = golden("models")
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), verbose=False)
    sd = m.state_dict()
    assert list(sd.keys()) == list(G[f"{name}/keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(G[f"{name}/shapes"])
    assert sum(p.numel() for p in m.parameters()) == int(G[f"{name}/n_params"])
    assert m.stride.tolist() == G[f"{name}/stride"].tolist()
    det = f"model.{len(m.model) - 1}"
    got = torch.stack([sd[f"{det}.cv3.{l}.2.bias"] for l in range(len(m.stride))])
    np.testing.assert_allclose(got.numpy(), G[f"{name}/init_cls_bias"], rtol=1e-6)
    bn = [x for x in m.modules() if type(x) is torch.nn.BatchNorm2d]
    assert all(b.eps == 1e-3 and b.momentum == 0.03 for b in bn)
    bn3 = [x for x in m.modules() if type(x) is torch.nn.BatchNorm3d]
    assert all(b.eps == 1e-5 and b.momentum == 0.1 for b in bn3)


def test_scale_letter_and_missing_module():
    from ultralytics.nn.tasks import DetectionModel, guess_model_scale, yaml_model_load
    assert guess_model_scale("yolov8s-ASF-P2P2.yaml") == "s"
    d = yaml_model_load(os.path.join(CFG_DIR, "yolov8s-ASF-P2P2.yaml"))
    assert d["scale"] == "s"
    ms = DetectionModel(d, verbose=False)
    assert ms.model[0].conv.out_channels == 32  # width 0.5 * 64
    d["backbone"][0][2] = "GhostConv"
    with pytest.raises(NotImplementedError, match="outside the DEAL-YOLO hot path"):
        DetectionModel(d, verbose=False)


def test_concat_plan_is_in_place_for_deal_yolo():
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"), verbose=False)
    plan = m._concat_plan()
    # every Concat input of DEAL-YOLO-N is produced straight into the concat buffer: no copy kernels
    assert plan == {10: (11, 0), 9: (11, 1), 15: (16, 0), 14: (16, 1), 18: (19, 0), 12: (19, 1), 21: (22, 0), 8: (22, 1)}


def test_optimizer_groups_match_reference(golden):
    """Flat layout = [bias | decayed weights | norm weights] with the reference's group sizes (trainer.npz)."""
    from oracle.trainer import param_groups
    from ultralytics.nn.tasks import DetectionModel
    G = golden("trainer")
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"), verbose=False)
    norm = tuple(v for k, v in torch.nn.__dict__.items() if "Norm" in k and isinstance(v, type))
    groups = ([], [], [])
    for mn, mod in m.named_modules():
        for pn, p in mod.named_parameters(recurse=False):
            full = f"{mn}.{pn}"
            groups[0 if "bias" in full else (2 if isinstance(mod, norm) else 1)].append(full)
    assert [len(g) for g in groups] == G["SGD/group_sizes"].tolist()
    ref = param_groups(m.state_dict().keys())
    assert [sorted(g) for g in groups] == [sorted(g) for g in ref]


def test_load_reference_format_checkpoint():
    """tests/golden/ref_ckpt.pt is a whole-module fp16 pickle written by the REFERENCE classes (make_golden.py::gen_ckpt, the
    on-disk format of engine/trainer.py:898-923).  attempt_load_weights must rebuild this package's DetectionModel from it."""
    import os
    import torch
    from oracle import graph as og
    from conftest import CFG_DIR, ROOT
    from ultralytics.nn.tasks import DetectionModel, attempt_load_weights, torch_safe_load
    path = os.path.join(ROOT, "tests", "golden", "ref_ckpt.pt")
    ckpt, _ = torch_safe_load(path)
    assert ckpt["epoch"] == 3 and ckpt["updates"] == 57 and ckpt["ema"] is None
    m = attempt_load_weights(path)
    assert isinstance(m, DetectionModel) and not m.training
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")))
    want = og.fill_state(og.state_layout(g), 21)
    got = m.state_dict()
    assert set(got) == set(want)
    for k, v in want.items():
        ref = v.half().float() if v.is_floating_point() else v  # the reference stores .half() weights
        assert torch.equal(got[k].cpu().float(), ref.float()), k
    assert m.args.get("imgsz") == 640


def test_reference_format_checkpoint_roundtrip(tmp_path):
    """save_reference_format -> attempt_load_weights: weights survive (as fp16), the pickled object carries no engine state
    and has the attribute layout the reference's classes expect (checked against the reference itself in the build container
    by tests/golden/check_export.py)."""
    import os
    import torch
    import torch.nn as nn
    from conftest import CFG_DIR
    from ultralytics.nn.tasks import DetectionModel, attempt_load_weights, save_reference_format, torch_safe_load
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-LD-P2.yaml"), ch=3, verbose=False)
    path = save_reference_format(str(tmp_path / "last.pt"), m, updates=9, epoch=2, train_args={"imgsz": 640})
    ckpt, _ = torch_safe_load(path)
    obj = ckpt["model"]
    assert ckpt["updates"] == 9 and next(obj.parameters()).dtype == torch.float16
    assert all("rt" not in mod.__dict__ and "_pn_i32" not in mod.__dict__ for mod in obj.modules())
    assert any(type(mod) is nn.Upsample for mod in obj.modules()) and obj.model[-1].anchors.numel() == 0 and obj.warehouse_manager is None
    back = attempt_load_weights(path)
    for (k, a), (_, b) in zip(m.state_dict().items(), back.state_dict().items()):
        assert torch.equal(a.half().float() if a.is_floating_point() else a, b if not b.is_floating_point() else b.float()), k


def test_reference_entry_script_arguments_are_accepted():
    """The keyword sets of the reference's entry scripts (train.py:9-24, detect.py:7-14, val.py:8-16, get_FPS.py:40-52) pass the
    cfg check, device lists parse without touching a GPU, and the multi-GPU launch is the reference's torch.distributed.run
    command (utils/dist.py:47-65) around a script that rebuilds the model and calls train with the same overrides."""
    import ast
    from ultralytics.cfg import DEFAULT_CFG_DICT, get_cfg
    from ultralytics.utils.dist import ddp_cleanup, generate_ddp_command, parse_devices
    from ultralytics.utils.torch_utils import select_device
    train_kw = dict(data="VisDrone.yaml", cache=False, imgsz=640, epochs=300, batch=8, close_mosaic=10, workers=8, device="0",
                    optimizer="SGD", project="runs/train", name="yolov8m-ASF-P2")
    a = get_cfg(overrides=train_kw)
    assert a.epochs == 300 and a.batch == 8 and a.device == "0" and a.project == "runs/train"
    get_cfg(overrides=dict(source="images/test", imgsz=640, project="runs/detect", name="exp", verbose=True, save=True, conf=0.2, visualize=False))
    get_cfg(overrides=dict(data="data.yaml", split="test", imgsz=640, batch=16, rect=False, save_json=False, project="runs/val", name="x"))
    with pytest.raises(SyntaxError):
        get_cfg(overrides=dict(not_a_key=1))
    # defaults are the reference's (cfg/default.yaml)
    assert (DEFAULT_CFG_DICT["epochs"], DEFAULT_CFG_DICT["batch"], DEFAULT_CFG_DICT["device"], DEFAULT_CFG_DICT["optimizer"]) == (200, 8, 0, "auto")
    assert parse_devices("0,1,2,3") == [0, 1, 2, 3] and parse_devices(None) == [0] and parse_devices([2, 3]) == [2, 3] and parse_devices("cuda:1") == [1]
    with pytest.raises(ValueError, match="not a rank of a distributed launch"):
        select_device("0,1")  # outside a launch a device list is an error, never a silent "first GPU"
    cmd, file, result = generate_ddp_command(4, "yolov8-ASF-P2.yaml", dict(train_kw, device="0,1,2,3"))
    try:
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc_per_node=4" in cmd and "127.0.0.1" in cmd and cmd[-1] == file
        src = open(file).read()
        ast.parse(src)
        assert "YOLO('yolov8-ASF-P2.yaml')" in src and "'device': '0,1,2,3'" in src and "model.train(**overrides)" in src
    finally:
        ddp_cleanup(file)
    assert not os.path.exists(file)


def test_import_paths_of_the_reference():
    """Module paths the reference's own code imports on this path."""
    from ultralytics import YOLO  # noqa: F401
    from ultralytics.engine.results import Boxes, Results  # noqa: F401
    from ultralytics.models.yolo.detect import DetectionPredictor, DetectionTrainer, DetectionValidator  # noqa: F401
    from ultralytics.models.yolo.detect.predict import DetectionPredictor as P2  # noqa: F401
    from ultralytics.models.yolo.detect.train import DetectionTrainer as T2  # noqa: F401
    from ultralytics.models.yolo.model import YOLO as Y2  # noqa: F401
    from ultralytics.nn.tasks import attempt_load_weights  # noqa: F401
    from ultralytics.utils.ops import clip_boxes, scale_boxes
    from ultralytics.utils.tal import TaskAlignedAssigner, bbox2dist, dist2bbox, make_anchors
    from ultralytics.utils.torch_utils import select_device  # noqa: F401
    from oracle import metrics as om, nn as onn
    # the tensor helpers against the oracle's restatements of the reference (utils/tal.py:294-324, utils/ops.py:89-124)
    feats = [torch.zeros(1, 8, 4, 6), torch.zeros(1, 8, 2, 3)]
    pts, st = make_anchors(feats, [8, 16])
    pts_o, st_o = onn.make_anchors([(4, 6), (2, 3)], [8, 16])
    assert torch.equal(pts, pts_o) and torch.equal(st, st_o)
    d = torch.rand(2, 4, 30)
    x = dist2bbox(d, pts.t().unsqueeze(0), xywh=True, dim=1)
    lt, rb = d.chunk(2, 1)
    assert torch.allclose(x[:, :2], (pts.t() - lt + pts.t() + rb) / 2) and torch.allclose(x[:, 2:], lt + rb)
    assert bbox2dist(pts, torch.cat((pts - 1, pts + 20), -1), 16).max() <= 16 - 0.01
    b = torch.tensor([[100., 50., 300., 400.], [-5., 10., 700., 650.]])
    got = scale_boxes((640, 640), b.clone(), (480, 640, 3))
    want = torch.from_numpy(om.scale_boxes((640, 640), b.numpy().copy(), (480, 640)))
    assert torch.allclose(got, want.float(), atol=1e-4) and float(clip_boxes(b.clone(), (100, 100)).max()) == 100
    assert TaskAlignedAssigner(topk=10, num_classes=6, alpha=0.5, beta=6.0).topk == 10


@pytest.mark.parametrize("name", ["yolov8n-ASF-P2P2", "yolov8n-LD-P2", "yolov8n-ASF-P2", "yolov8n-p2"])
def test_fresh_model_equals_a_fresh_reference_model(golden, name):
    """torch.manual_seed(s) + construction gives the reference's initial state entry for entry: same construction order and
    tensor shapes (so the same draws from the global RNG), same BN buffers as its stride-probe forward leaves behind -- which is
    what lets ``seed`` reproduce the reference trainer's starting point."""
    from ultralytics.nn.tasks import DetectionModel
    G = golden("init_state")
    torch.manual_seed(0)
    m = DetectionModel(name + ".yaml", ch=3, nc=6 if name != "yolov8n-p2" else 80, verbose=False)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in G[f"{name}/keys"]]
    s = np.array([float(v.double().sum()) for v in sd.values()])
    a = np.array([float(v.double().abs().sum()) for v in sd.values()])
    np.testing.assert_allclose(s, G[f"{name}/sum"], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(a, G[f"{name}/abssum"], rtol=1e-6, atol=1e-5)


def test_soap_matches_the_reference_trainers_optimizer(golden):
    """ultralytics.hip.soap.Soap against the reference trainer's SOAP class (engine/trainer.py:54-473) as build_optimizer
    configures it: 25 steps of seeded gradients over a bias, two conv weights, a matrix and a norm weight -- parameters after
    steps 1 (preconditioner seeded, nothing moves), 2, 11 (first QR refresh of the eigenbases) and 25."""
    from golden.cases import rnd
    from ultralytics.hip.soap import Soap
    G = golden("soap")
    shapes = [((16,), 0), ((16, 8, 3, 3), 1), ((8, 16, 1, 1), 1), ((24, 6), 1), ((8,), 2)]
    ps = [rnd(50 + i, *sh, scale=0.5) for i, (sh, _) in enumerate(shapes)]
    opt = Soap([(p, g) for p, (_, g) in zip(ps, shapes)], beta1=0.937, beta2=0.95)
    for step in range(1, 26):
        grads = [rnd(1000 + 31 * step + i, *p.shape, scale=1.0) * (1.0 + 0.1 * i) + 0.05 * p for i, p in enumerate(ps)]
        opt.step(grads, [0.01] * 3, [0.0, 5e-4, 0.0])
        if step in (1, 2, 11, 25):
            for i, p in enumerate(ps):
                ref = G.t(f"step{step}/p{i}")
                assert float((p - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), (step, i, float((p - ref).abs().max()))


def test_importing_the_package_turns_graph_packet_capture_off():
    """hip/__init__.py: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 must be in the environment before HIP initialises (the hipGraph defect of
    this ROCm); a user's own setting is respected and then decides whether graphs are allowed."""
    import subprocess
    import sys
    from conftest import PKG
    code = f"import sys, os; sys.path.insert(0, {PKG!r}); import ultralytics.hip as h; print(os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'), h.GRAPH_SAFE)"
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.split() == ["0", "True"]
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1")).stdout.split() == ["1", "False"]


def test_label_capacity_covers_the_outlier_image_in_every_mosaic_tile():
    """Mosaic partners are drawn with replacement from a buffer that holds the index image (data/dataset.py, reference
    data/augment.py Mosaic._mosaic4) and MixUp's partner is an independent draw: the most-labelled image can sit in all 4 (8) tiles."""
    from types import SimpleNamespace as NS
    from ultralytics.engine.trainer import DetectionTrainer
    labels = [dict(cls=np.zeros((n, 1))) for n in (3, 41, 2, 5, 1)]
    cap = lambda mosaic, mixup: DetectionTrainer._label_capacity(NS(args=NS(mixup=mixup)), NS(dataset=NS(labels=labels, mosaic=mosaic)))  # noqa: E731
    assert cap(0.0, 0.0) == 48          # one image: 41 rounded up to 8
    assert cap(1.0, 0.0) == 4 * 41 + 4  # four tiles of the outlier (164 -> 168)
    assert cap(1.0, 0.1) == 8 * 41      # MixUp of two such mosaics
    assert DetectionTrainer._label_capacity(NS(args=NS(mixup=0.0)), NS(dataset=None)) is None
