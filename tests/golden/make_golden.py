"""Generate the golden fixtures (tests/golden/*.npz) from the REFERENCE implementation.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference ships no tests or vectors for this path, so the oracle is pinned
by outputs of the reference itself.  Inputs come from numpy PCG64 streams and weights from
``oracle.graph.fill_state`` (a deterministic fill keyed by parameter name order), so fixtures hold
inputs + expected outputs only, never weights or reference source.
"""
import os
import sys
import warnings

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _refimport  # noqa: E402

_refimport.install()

from types import SimpleNamespace  # noqa: E402

from ultralytics.cfg import get_cfg  # noqa: E402
from ultralytics.engine.trainer import BaseTrainer  # noqa: E402
from ultralytics.nn.extra_modules.block import Add, ScalSeq, Zoom_cat  # noqa: E402
from ultralytics.nn.modules import SPPF, C2f, Conv, Detect, LDConv  # noqa: E402
from ultralytics.nn.tasks import DetectionModel  # noqa: E402
from ultralytics.utils import DEFAULT_CFG, ops  # noqa: E402
from ultralytics.utils.loss import v8DetectionLoss  # noqa: E402
from ultralytics.utils.metrics import WiseIouLoss  # noqa: E402
from ultralytics.utils.torch_utils import ModelEMA, initialize_weights  # noqa: E402

from oracle import graph as og  # noqa: E402
from cases import tal_filler_case, DATASET_IMGSZ, E2E, write_dataset, write_e2e_dataset, MODES, loss_cases, metric_cases, metric_geometry, module_cases, planted_batches, module_shapes, rnd, synth_batch, synth_detections  # noqa: E402

CFG_DIR = os.path.join(_refimport.REF, "ultralytics/cfg/models")
torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(out)} arrays)")


def load_filled(module, layer, seed, prefix="model.0."):
    """Fill a reference module with the shared deterministic state for a one-layer oracle graph."""
    g = og.Graph([layer], [], layer.args.get("nc", 0), 0, "", [], {})
    sd = og.fill_state(og.state_layout(g), seed)
    sd = {k[len(prefix):]: v for k, v in sd.items()}
    initialize_weights(module)  # BN eps/momentum exactly as DetectionModel.__init__ does
    module.load_state_dict(sd, strict=True)
    return module


# ----------------------------------------------------------------------------- A. modules
REF_CTORS = {
    "conv_k3s1": lambda: Conv(16, 32, 3, 1), "conv_k3s2": lambda: Conv(16, 32, 3, 2), "conv_k1": lambda: Conv(24, 16, 1, 1),
    "conv_stem": lambda: Conv(3, 16, 3, 2), "c2f_n2_sc": lambda: C2f(32, 32, 2, True), "c2f_n1": lambda: C2f(48, 32, 1, False),
    "sppf": lambda: SPPF(32, 32, 5), "scalseq": lambda: ScalSeq([16, 32, 64], 16), "scalseq_conv0": lambda: ScalSeq([32, 32, 64], 16),
    "zoom_cat": lambda: Zoom_cat(), "add": lambda: Add(), "ldconv_n3s2": lambda: LDConv(8, 16, 3, 2),
    "ldconv_n1s1": lambda: LDConv(16, 8, 1, 1), "ldconv_n5s1": lambda: LDConv(8, 8, 5, 1), "ldconv_stem": lambda: LDConv(3, 16, 3, 2),
}


def gen_modules():
    arrs = {}
    for name, (layer, ci) in module_cases().items():
        shapes = module_shapes()[name]
        torch.manual_seed(0)
        m = load_filled(REF_CTORS[name](), layer, seed=100 + ci).train()
        xs = [rnd(1000 + 10 * ci + j, *s).requires_grad_(True) for j, s in enumerate(shapes)]
        y = m(xs[0] if len(xs) == 1 else list(xs))
        gy = rnd(2000 + ci, *y.shape)
        y.backward(gy)
        arrs[f"{name}/y"] = y
        arrs[f"{name}/gy"] = gy
        for j, x in enumerate(xs):
            arrs[f"{name}/x{j}"] = x
            arrs[f"{name}/gx{j}"] = x.grad
        for k, p in m.named_parameters():
            if p.grad is not None:
                arrs[f"{name}/gp/{k}"] = p.grad
        for k, b in m.named_buffers():
            if "running" in k:
                arrs[f"{name}/buf/{k}"] = b
        # eval-mode output with the (now updated) running statistics
        m.eval()
        with torch.no_grad():
            arrs[f"{name}/y_eval"] = m(xs[0].detach() if len(xs) == 1 else [x.detach() for x in xs])
    # Detect: train list + eval decode
    layer = og.Layer(0, [0, 1, 2], "Detect", [16, 32, 64], 70, dict(nc=6))
    det = load_filled(Detect(6, (16, 32, 64)), layer, seed=150)
    det.stride = torch.tensor([4.0, 8.0, 16.0])
    xs = [rnd(3000 + j, *s).requires_grad_(True) for j, s in enumerate([(2, 16, 8, 12), (2, 32, 4, 6), (2, 64, 2, 3)])]
    det.train()
    outs = det(list(xs))
    gys = [rnd(3100 + j, *o.shape) for j, o in enumerate(outs)]
    torch.autograd.backward(outs, gys)
    for j in range(3):
        arrs[f"detect/x{j}"], arrs[f"detect/gx{j}"] = xs[j], xs[j].grad
        arrs[f"detect/y{j}"], arrs[f"detect/gy{j}"] = outs[j], gys[j]
    for k, p in det.named_parameters():
        if p.grad is not None:
            arrs[f"detect/gp/{k}"] = p.grad
    det.eval()
    with torch.no_grad():
        y, feats = det([x.detach() for x in xs])
    arrs["detect/y_eval"] = y
    npz("modules", **arrs)


# ----------------------------------------------------------------------------- B. whole models
def set_mode(model, wiou, nwd):
    crit = model.criterion
    crit.bbox_loss.use_wiseiou = wiou
    crit.bbox_loss.nwd_loss = nwd
    if wiou:
        crit.bbox_loss.wiou_loss = WiseIouLoss(ltype="WIoU", monotonous=False, inner_iou=False, focaler_iou=False)




def gen_models():
    arrs = {}
    for mi, name in enumerate(["yolov8n-ASF-P2P2", "yolov8n-LD-P2", "yolov8n-ASF-P2", "yolov8n-p2"]):
        torch.manual_seed(0)
        m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
        m.args = get_cfg(DEFAULT_CFG)
        g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
        layout = og.state_layout(g)
        keys = list(m.state_dict().keys())
        assert keys == list(layout.keys()), (name, [k for k in keys if k not in layout][:5], [k for k in layout if k not in keys][:5])
        arrs[f"{name}/n_params"] = sum(p.numel() for p in m.parameters())
        arrs[f"{name}/stride"] = m.stride
        arrs[f"{name}/keys"] = np.array(keys)
        arrs[f"{name}/shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
        # reference-initialised Detect biases and BN hyper-parameters (bias_init / initialize_weights)
        sd0 = m.state_dict()
        det = f"model.{len(m.model) - 1}"
        arrs[f"{name}/init_cls_bias"] = torch.stack([sd0[f"{det}.cv3.{l}.2.bias"] for l in range(len(m.stride))])
        m.load_state_dict(og.fill_state(layout, seed=7 + mi), strict=True)
        batch = synth_batch(50 + mi, 2, 4, g.nc)
        arrs[f"{name}/img"] = batch["img"]
        for k in ("batch_idx", "cls", "bboxes"):
            arrs[f"{name}/{k}"] = batch[k]
        m.train()
        # per-layer outputs (train mode, batch statistics)
        ys, x = [], batch["img"]
        sd_before = {k: v.clone() for k, v in m.state_dict().items()}
        for layer in m.model:
            if layer.f != -1:
                x = ys[layer.f] if isinstance(layer.f, int) else [x if j == -1 else ys[j] for j in layer.f]
            x = layer(x)
            ys.append(x)
        for i, y in enumerate(ys[:-1]):
            arrs[f"{name}/layer{i}"] = y.detach().to(torch.float16)  # fp16 storage: 2e-3 tolerance layer dump
        for l, f in enumerate(ys[-1]):
            arrs[f"{name}/feat{l}"] = f.detach()
        m.load_state_dict(sd_before)
        if name in ("yolov8n-ASF-P2P2", "yolov8n-LD-P2"):
            for mode, (wiou, nwd) in MODES.items():
                m.load_state_dict(sd_before)
                m.zero_grad()
                if hasattr(m, "criterion"):
                    del m.criterion
                m.criterion = m.init_criterion()
                set_mode(m, wiou, nwd)
                loss, items = m(batch)
                loss.backward()
                arrs[f"{name}/{mode}/loss"] = loss.detach()
                arrs[f"{name}/{mode}/items"] = items
                gn = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
                arrs[f"{name}/{mode}/grad_names"] = np.array(list(gn.keys()))
                arrs[f"{name}/{mode}/grad_l2"] = torch.stack([v.norm() for v in gn.values()])
                arrs[f"{name}/{mode}/grad_sum"] = torch.stack([v.sum() for v in gn.values()])
                if mode == "ciou":
                    first = next(iter(gn))
                    arrs[f"{name}/{mode}/grad_first"] = gn[first]
                    sdm = m.state_dict()
                    rm = [k for k in sdm if k.endswith("running_mean")]
                    arrs[f"{name}/run_mean_names"] = np.array(rm)
                    arrs[f"{name}/run_mean_sum"] = torch.stack([sdm[k].sum() for k in rm])
                    arrs[f"{name}/run_var_sum"] = torch.stack([sdm[k.replace("mean", "var")].sum() for k in rm])
        # eval + fused eval
        m.load_state_dict(sd_before)
        m.eval()
        with torch.no_grad():
            y, _ = m(batch["img"])
            arrs[f"{name}/y_eval"] = y
            m.fuse(verbose=False)
            yf, _ = m(batch["img"])
            arrs[f"{name}/y_eval_fused"] = yf
            arrs[f"{name}/n_params_fused"] = sum(p.numel() for p in m.parameters())
    npz("models", **arrs)


# ----------------------------------------------------------------------------- C. loss
class _FakeModel(torch.nn.Module):
    def __init__(self, nc, strides):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.args = get_cfg(DEFAULT_CFG)
        self.model = [SimpleNamespace(stride=torch.tensor(strides), nc=nc, no=nc + 64, reg_max=16)]


def gen_loss():
    arrs = {}
    shapes = [(16, 16), (8, 8), (4, 4)]
    for ci, (name, tg) in enumerate(loss_cases().items()):
        for mode, (wiou, nwd) in MODES.items():
            crit = v8DetectionLoss(_FakeModel(6, [4.0, 8.0, 16.0]))
            crit.bbox_loss.use_wiseiou, crit.bbox_loss.nwd_loss = wiou, nwd
            if wiou:
                crit.bbox_loss.wiou_loss = WiseIouLoss(ltype="WIoU", monotonous=False, inner_iou=False, focaler_iou=False)
            feats = [rnd(400 + 10 * ci + l, 2, 70, *s, scale=1.5).requires_grad_(True) for l, s in enumerate(shapes)]
            batch = dict(tg)
            captured = {}
            orig = crit.assigner.forward

            def spy(*a, **k):
                out = orig(*a, **k)
                captured["asg"] = out
                return out

            crit.assigner.forward = spy
            n_calls = 3 if (wiou and name == "random5") else 1
            for call in range(n_calls):
                for f in feats:
                    f.grad = None
                loss, items = crit([f for f in feats], batch)
                loss.backward()
                tag = f"{name}/{mode}" + (f"/call{call}" if n_calls > 1 else "")
                arrs[f"{tag}/loss"], arrs[f"{tag}/items"] = loss.detach(), items
                for l, f in enumerate(feats):
                    arrs[f"{tag}/gfeat{l}"] = f.grad if f.grad is not None else torch.zeros_like(f)
                if wiou:
                    arrs[f"{tag}/iou_mean"] = crit.bbox_loss.wiou_loss.iou_mean.clone()
            if mode == "ciou":
                for l, f in enumerate(feats):
                    arrs[f"{name}/feat{l}"] = f
                for k, v in tg.items():
                    arrs[f"{name}/{k}"] = v
                tl, tb, ts, fg, tgi = captured["asg"]
                arrs[f"{name}/target_labels"] = tl
                arrs[f"{name}/target_bboxes"] = tb
                arrs[f"{name}/target_scores"] = ts
                arrs[f"{name}/fg_mask"] = fg
                arrs[f"{name}/target_gt_idx"] = tgi
    npz("loss", **arrs)


# ----------------------------------------------------------------------------- D. decode + NMS
def gen_nms():
    arrs = {}
    rng = np.random.default_rng(77)
    # raw soft_nms on hand-made candidate sets
    def case(name, boxes, scores, thr=0.7):
        b, s = torch.tensor(boxes, dtype=torch.float32).view(-1, 4), torch.tensor(scores, dtype=torch.float32)
        arrs[f"soft/{name}/boxes"], arrs[f"soft/{name}/scores_in"] = b.clone(), s.clone()
        keep = ops.soft_nms(b, s, thr)
        arrs[f"soft/{name}/keep"], arrs[f"soft/{name}/scores_out"], arrs[f"soft/{name}/thr"] = keep, s, thr

    case("n0", [], [])
    case("n1", [[0, 0, 10, 10]], [0.9])
    case("n2_disjoint", [[0, 0, 10, 10], [20, 20, 30, 30]], [0.5, 0.9])
    case("n2_overlap", [[0, 0, 10, 10], [1, 1, 11, 11]], [0.9, 0.8], 0.5)
    case("n3_chain", [[0, 0, 10, 10], [1, 0, 11, 10], [2, 0, 12, 10]], [0.6, 0.9, 0.7], 0.5)
    case("first_not_top", [[0, 0, 10, 10], [50, 50, 60, 60], [0.5, 0, 10.5, 10], [51, 50, 61, 60]], [0.3, 0.95, 0.9, 0.5], 0.5)
    # borderline decay: iou just above threshold, score*exp(-iou^2/0.5) close to 0.25
    case("borderline", [[0, 0, 10, 10], [0, 0, 10, 9.1], [0, 0, 10, 7.2], [30, 30, 40, 40]], [0.9, 0.9, 0.4436, 0.3], 0.7)
    n = 400
    xy = rng.random((n, 2)) * 80
    wh = rng.random((n, 2)) * 25 + 5
    case("random400", np.concatenate([xy, xy + wh], 1), rng.random(n) * 0.75 + 0.25, 0.7)
    n = 1500
    xy = rng.random((n, 2)) * 200
    wh = rng.random((n, 2)) * 40 + 8
    case("random1500", np.concatenate([xy, xy + wh], 1), rng.random(n) * 0.7 + 0.3, 0.7)
    # full non_max_suppression on a synthetic (B, 4+nc, A) prediction
    B, nc, A = 2, 6, 1344
    ctr = rng.random((B, 2, A)) * 64
    whp = rng.random((B, 2, A)) * 20 + 4
    cls = rng.random((B, nc, A)) ** 6  # mostly small, a few confident
    pred = torch.from_numpy(np.concatenate([ctr, whp, cls], 1).astype(np.float32))
    arrs["nms/pred"] = pred
    for tag, kw in {"predict": dict(conf_thres=0.25, iou_thres=0.7, max_det=300),
                    "val": dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300),
                    "agnostic": dict(conf_thres=0.25, iou_thres=0.45, agnostic=True, max_det=300),
                    "classes": dict(conf_thres=0.2, iou_thres=0.6, classes=[1, 4], max_det=20)}.items():
        out = ops.non_max_suppression(pred.clone(), **kw)
        for i, o in enumerate(out):
            arrs[f"nms/{tag}/img{i}"] = o
    npz("nms", **arrs)


# ----------------------------------------------------------------------------- E. trainer micro-trace
def gen_trainer():
    """5 iterations of the reference's own optimizer_step/build_optimizer/ModelEMA on DEAL-YOLO-N 64x64,
    batch 2, driven exactly as engine/trainer.py:780-815 does (warm-up interpolation, loss, backward, step)."""
    arrs = {}
    name = "yolov8n-ASF-P2P2"
    for opt_name in ("SGD", "AdamW"):
        torch.manual_seed(0)
        m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
        args = get_cfg(DEFAULT_CFG)
        m.args = args
        g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
        m.load_state_dict(og.fill_state(og.state_layout(g), seed=11), strict=True)
        for k, v in m.named_parameters():
            v.requires_grad = ".dfl" not in k
        bs, nb, epochs = 2, 8, 100
        fake = SimpleNamespace(args=args, model=m)
        accumulate = max(round(args.nbs / bs), 1)
        wd = args.weight_decay * bs * accumulate / args.nbs
        lr0 = args.lr0 if opt_name == "SGD" else 0.001
        fake.optimizer = BaseTrainer.build_optimizer(fake, model=m, name=opt_name, lr=lr0, momentum=args.momentum, decay=wd)
        fake.scaler = torch.cuda.amp.GradScaler(enabled=False)
        fake.ema = ModelEMA(m)
        lf = lambda x: max(1 - x / epochs, 0) * (1.0 - args.lrf) + args.lrf  # noqa: E731
        for pg in fake.optimizer.param_groups:
            pg["initial_lr"] = pg["lr"]
        nw = max(round(args.warmup_epochs * nb), 100)
        last_opt_step = -1
        m.train()
        trace = []
        for ni in range(5):
            xi = [0, nw]
            accumulate = max(1, int(np.interp(ni, xi, [1, args.nbs / bs]).round()))
            for j, x in enumerate(fake.optimizer.param_groups):
                x["lr"] = np.interp(ni, xi, [args.warmup_bias_lr if j == 0 else 0.0, x["initial_lr"] * lf(0)])
                if "momentum" in x:
                    x["momentum"] = np.interp(ni, xi, [args.warmup_momentum, args.momentum])
            batch = synth_batch(900 + ni, bs, 4, g.nc)
            loss, items = m(batch)
            loss.backward()
            gnorm = torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.parameters() if p.grad is not None))
            stepped = 0
            if ni - last_opt_step >= accumulate:
                BaseTrainer.optimizer_step(fake)
                last_opt_step = ni
                stepped = 1
            sd = m.state_dict()
            esd = fake.ema.ema.state_dict()
            trace.append([float(loss), *[float(v) for v in items], float(gnorm), stepped,
                          *[float(pg["lr"]) for pg in fake.optimizer.param_groups],
                          float(sum(v.double().sum() for k, v in sd.items() if v.dtype.is_floating_point)),
                          float(sum(v.double().abs().sum() for k, v in sd.items() if v.dtype.is_floating_point)),
                          float(sum(v.double().abs().sum() for k, v in esd.items() if v.dtype.is_floating_point))])
        arrs[f"{opt_name}/trace"] = np.array(trace, dtype=np.float64)
        arrs[f"{opt_name}/final_w0"] = m.state_dict()["model.0.conv.weight"]
        arrs[f"{opt_name}/final_bn22"] = m.state_dict()["model.22.cv2.bn.weight"] if "model.22.cv2.bn.weight" in m.state_dict() else m.state_dict()["model.23.cv2.bn.weight"]
        arrs[f"{opt_name}/final_bias"] = m.state_dict()["model.26.cv3.0.2.bias"]
        arrs[f"{opt_name}/ema_w0"] = fake.ema.ema.state_dict()["model.0.conv.weight"]
        arrs[f"{opt_name}/group_sizes"] = np.array([len(pg["params"]) for pg in fake.optimizer.param_groups])
    arrs["trace_columns"] = np.array(["loss", "box", "cls", "dfl", "grad_norm", "stepped", "lr_bias", "lr_w", "lr_bn",
                                      "sum_state", "abs_state", "abs_ema"])
    npz("trainer", **arrs)


# ----------------------------------------------------------------------------- F. full-size scalars
def gen_fullsize():
    arrs = {}
    for mi, name in enumerate(["yolov8n-ASF-P2P2", "yolov8n-LD-P2"]):
        m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
        m.args = get_cfg(DEFAULT_CFG)
        g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
        m.load_state_dict(og.fill_state(og.state_layout(g), seed=21 + mi), strict=True)
        m.train()
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        # SURVEY.md 8(d) synthetic recipe: 8 boxes/img, wh in [0.01,0.09), xy in [0.1,0.9)
        rng = np.random.default_rng(5 + mi)
        B, nb = 2, 8
        batch = dict(img=torch.from_numpy(rng.random((B, 3, 640, 640), dtype=np.float32)),
                     batch_idx=torch.arange(B).repeat_interleave(nb).float(),
                     cls=torch.from_numpy(rng.integers(0, 6, (B * nb, 1)).astype(np.float32)),
                     bboxes=torch.from_numpy(np.concatenate([rng.random((B * nb, 2)) * 0.8 + 0.1,
                                                             rng.random((B * nb, 2)) * 0.08 + 0.01], 1).astype(np.float32)))
        for k in ("batch_idx", "cls", "bboxes"):
            arrs[f"{name}/{k}"] = batch[k]
        arrs[f"{name}/img_seed"] = 5 + mi
        for mode, (wiou, nwd) in MODES.items():
            m.load_state_dict(sd0)
            m.zero_grad()
            if hasattr(m, "criterion"):
                del m.criterion
            m.criterion = m.init_criterion()
            set_mode(m, wiou, nwd)
            loss, items = m(batch)
            arrs[f"{name}/{mode}/loss"], arrs[f"{name}/{mode}/items"] = loss.detach(), items
            if mode == "ciou":
                loss.backward()
                gn = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
                arrs[f"{name}/grad_names"] = np.array(list(gn.keys()))
                arrs[f"{name}/grad_l2"] = torch.stack([v.norm() for v in gn.values()])
    npz("fullsize", **arrs)


def gen_tal_filler():
    """The assigner's zero-metric fillers (cases.tal_filler_case): assignment tensors, losses and gradients of the reference in
    CIoU and WIoU mode."""
    arrs = {}
    feats0, batch = tal_filler_case()  # inputs are regenerated from cases.py on both sides: only outputs go into the fixture
    for mode in ("ciou", "wiou"):
        wiou, nwd = MODES[mode]
        crit = v8DetectionLoss(_FakeModel(6, [4.0, 8.0, 16.0]))
        crit.bbox_loss.use_wiseiou, crit.bbox_loss.nwd_loss = wiou, nwd
        if wiou:
            crit.bbox_loss.wiou_loss = WiseIouLoss(ltype="WIoU", monotonous=False, inner_iou=False, focaler_iou=False)
        feats = [f.clone().requires_grad_(True) for f in feats0]
        captured = {}
        orig = crit.assigner.forward

        def spy(*a, **k):
            out = orig(*a, **k)
            captured["asg"] = out
            return out

        crit.assigner.forward = spy
        loss, items = crit(feats, dict(batch))
        loss.backward()
        arrs[f"{mode}/loss"], arrs[f"{mode}/items"] = loss.detach(), items
        for l, f in enumerate(feats):  # whole-level checksums + the top-left patch of image 0 where the gt and the fillers live
            arrs[f"{mode}/gfeat{l}_sum"], arrs[f"{mode}/gfeat{l}_abssum"] = f.grad.double().sum(), f.grad.double().abs().sum()
        arrs[f"{mode}/gfeat0_patch"] = feats[0].grad[0, :, :16, :16]
        if wiou:
            arrs[f"{mode}/iou_mean"] = crit.bbox_loss.wiou_loss.iou_mean.clone()
        if mode == "ciou":
            tl, tb, ts, fg, tgi = captured["asg"]
            arrs["target_scores_sum"], arrs["fg_mask"], arrs["target_gt_idx"] = ts.sum(-1), fg, tgi
            print("foreground anchors", int(fg.sum()), "of which with zero target score", int((fg & (ts.sum(-1) == 0)).sum()),
                  "at", (fg & (ts.sum(-1) == 0)).nonzero().tolist())
    npz("tal_filler", **arrs)


def gen_soap():
    """The reference trainer's SOAP class (engine/trainer.py:54-473), built as build_optimizer builds it (:1156-1165), over a few
    small tensors for 25 steps of seeded gradients (two QR refreshes of the eigenbases): parameters after steps 1, 2, 11 and 25."""
    from ultralytics.engine.trainer import SOAP
    shapes = [((16,), 0), ((16, 8, 3, 3), 1), ((8, 16, 1, 1), 1), ((24, 6), 1), ((8,), 2)]
    ps = [torch.nn.Parameter(rnd(50 + i, *sh, scale=0.5)) for i, (sh, _) in enumerate(shapes)]
    groups = [[p for p, (_, g) in zip(ps, shapes) if g == k] for k in range(3)]
    opt = SOAP(groups[0], lr=0.01, betas=(0.937, 0.95), weight_decay=0.0)
    opt.add_param_group({"params": groups[1], "weight_decay": 5e-4})
    opt.add_param_group({"params": groups[2], "weight_decay": 0.0})
    arrs = {"shapes": np.array([str(sh) for sh, _ in shapes]), "groups": np.array([g for _, g in shapes])}
    for step in range(1, 26):
        for i, p in enumerate(ps):
            p.grad = rnd(1000 + 31 * step + i, *p.shape, scale=1.0) * (1.0 + 0.1 * i) + 0.05 * p.detach()
        opt.step()
        if step in (1, 2, 11, 25):
            for i, p in enumerate(ps):
                arrs[f"step{step}/p{i}"] = p.detach().clone()
    npz("soap", **arrs)


def gen_init():
    """What a freshly constructed reference model holds after torch.manual_seed(0): per-entry sum and abs-sum of the state dict
    (weights come from the global RNG in construction order; BN buffers carry the side effects of the stride-probe forward)."""
    arrs = {}
    for name in ["yolov8n-ASF-P2P2", "yolov8n-LD-P2", "yolov8n-ASF-P2", "yolov8n-p2"]:
        torch.manual_seed(0)
        m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, nc=6 if name != "yolov8n-p2" else 80, verbose=False)
        sd = m.state_dict()
        arrs[f"{name}/keys"] = np.array(list(sd.keys()))
        arrs[f"{name}/sum"] = np.array([float(v.double().sum()) for v in sd.values()])
        arrs[f"{name}/abssum"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
    npz("init_state", **arrs)


def gen_fullsize_ld0():
    """The LD variant at 640x640 in the regime the reference trains it in: LDConv.__init__ zero-initialises p_conv.weight
    (nn/modules/conv.py:351-359), so |offset| = |p_conv.bias| < 1.  With the random p_conv weights of ``fullsize`` the offsets
    reach tens of pixels and the reference's own gradients are not a stable function of its inputs (a 1e-4 relative change of the
    image moves its per-parameter gradient norms by a median of 34 %, measured with the oracle), so gradient parity is pinned here."""
    arrs = {}
    name, mi = "yolov8n-LD-P2", 1
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    m.args = get_cfg(DEFAULT_CFG)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    sd = og.fill_state(og.state_layout(g), seed=21 + mi)
    for k in sd:
        if k.endswith("p_conv.weight"):
            sd[k] = torch.zeros_like(sd[k])
        elif k.endswith("p_conv.bias"):
            sd[k] = sd[k].clamp(-0.9, 0.9)
    m.load_state_dict(sd, strict=True)
    m.train()
    rng = np.random.default_rng(5 + mi)
    B, nb = 2, 8
    batch = dict(img=torch.from_numpy(rng.random((B, 3, 640, 640), dtype=np.float32)),
                 batch_idx=torch.arange(B).repeat_interleave(nb).float(),
                 cls=torch.from_numpy(rng.integers(0, 6, (B * nb, 1)).astype(np.float32)),
                 bboxes=torch.from_numpy(np.concatenate([rng.random((B * nb, 2)) * 0.8 + 0.1,
                                                         rng.random((B * nb, 2)) * 0.08 + 0.01], 1).astype(np.float32)))
    for k in ("batch_idx", "cls", "bboxes"):
        arrs[f"{name}/{k}"] = batch[k]
    m.criterion = m.init_criterion()
    loss, items = m(batch)
    arrs[f"{name}/ciou/loss"], arrs[f"{name}/ciou/items"] = loss.detach(), items
    loss.backward()
    gn = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    arrs[f"{name}/grad_names"] = np.array(list(gn.keys()))
    arrs[f"{name}/grad_l2"] = torch.stack([v.norm() for v in gn.values()])
    npz("fullsize_ld0", **arrs)



def gen_metrics():
    """Validation path: the reference's own box_iou, BaseValidator.match_predictions (numpy greedy
    matching) and ap_per_class on synthetic detection sets, plus DetMetrics.mean_results."""
    from ultralytics.engine.validator import BaseValidator
    from ultralytics.utils.metrics import DetMetrics, ap_per_class, box_iou
    arrs = {}
    iouv = torch.linspace(0.5, 0.95, 10)
    holder = SimpleNamespace(iouv=iouv)
    for name, seed, n_images, nc, max_labels, max_dets, jitter in metric_cases():
        batch, preds = synth_detections(seed, n_images, nc, max_labels, max_dets, jitter)
        geo = metric_geometry(name, n_images)
        stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[])
        for si, pred in enumerate(preds):
            # DetectionValidator._prepare_batch / _prepare_pred (models/yolo/detect/val.py:93-115), verbatim call sequence
            idx = batch["batch_idx"] == si
            tcls = torch.from_numpy(batch["cls"][idx]).squeeze(-1)
            ori_shape, ratio_pad = geo[si]
            tbox = torch.from_numpy(batch["bboxes"][idx])
            if len(tcls):
                tbox = ops.xywh2xyxy(tbox) * torch.tensor((640, 640))[[1, 0, 1, 0]]
                ops.scale_boxes((640, 640), tbox, ori_shape, ratio_pad=ratio_pad)
            pred = torch.from_numpy(pred).clone()
            ops.scale_boxes((640, 640), pred[:, :4], ori_shape, ratio_pad=ratio_pad)
            tp = torch.zeros(len(pred), 10, dtype=torch.bool)
            if len(pred) and len(tcls):
                iou = box_iou(tbox, pred[:, :4])
                tp = BaseValidator.match_predictions(holder, pred[:, 5], tcls, iou)
                arrs[f"{name}/iou{si}"] = iou
            arrs[f"{name}/tp{si}"] = tp
            if len(pred) == 0 and len(tcls) == 0:
                continue
            if len(pred) == 0:
                stats["tp"].append(tp); stats["conf"].append(torch.zeros(0)); stats["pred_cls"].append(torch.zeros(0))
            else:
                stats["tp"].append(tp); stats["conf"].append(pred[:, 4]); stats["pred_cls"].append(pred[:, 5])
            stats["target_cls"].append(tcls)
        st = {k: torch.cat(v, 0).numpy() for k, v in stats.items()}
        res = ap_per_class(st["tp"], st["conf"], st["pred_cls"], st["target_cls"], plot=False, names={i: str(i) for i in range(nc)})
        for key, v in zip(("tp_c", "fp_c", "p", "r", "f1", "ap", "classes", "p_curve", "r_curve", "f1_curve"), res[:10]):
            arrs[f"{name}/{key}"] = v
        dm = DetMetrics(names={i: str(i) for i in range(nc)})
        dm.process(st["tp"], st["conf"], st["pred_cls"], st["target_cls"])
        arrs[f"{name}/mean_results"] = np.asarray(dm.mean_results(), dtype=np.float64)  # mp, mr, map50, map
        arrs[f"{name}/fitness"] = np.asarray(dm.fitness)
    npz("metrics", **arrs)



def gen_ckpt():
    """A checkpoint in the reference's on-disk format (engine/trainer.py:898-923: whole-module pickle, fp16) holding the
    REFERENCE DetectionModel built from its own YAML with the shared deterministic state: exercises the loader of
    ultralytics.nn.tasks.attempt_load_weights.  The file is data: tensors + class paths."""
    from copy import deepcopy
    name = "yolov8n-ASF-P2P2"
    m = DetectionModel(os.path.join(CFG_DIR, name + ".yaml"), ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(os.path.join(CFG_DIR, name + ".yaml")))
    m.load_state_dict(og.fill_state(og.state_layout(g), 21), strict=True)
    m.args = dict(get_cfg(DEFAULT_CFG).__dict__) if not hasattr(m, "args") else m.args
    ckpt = {"epoch": 3, "best_fitness": 0.25, "model": deepcopy(m).half(), "ema": None, "updates": 57, "optimizer": None,
            "train_args": {"imgsz": 640, "batch": 64, "model": name + ".yaml"}, "date": "2026-01-01T00:00:00", "version": "8.1.9"}
    path = os.path.join(HERE, "ref_ckpt.pt")
    torch.save(ckpt, path)
    print(f"ref_ckpt.pt  {os.path.getsize(path) / 1024:.1f} KiB")


MAP_PROTOCOL = dict(name="yolov8n-ASF-P2P2", nc=4, imgsz=320, batch=16, nb=8, epochs=30, nval=2, init_seed=5)


def gen_map(perm_seed=None, save=True):
    """mAP parity protocol (north-star: 'mAP50 on a held-out synthetic set within +-0.2 of the reference'): the REFERENCE
    model / loss / build_optimizer / optimizer_step / ModelEMA train DEAL-YOLO-N for 30 epochs x 8 batches of 16 planted-
    rectangle images (320x320, 4 classes, no augmentation), driven as engine/trainer.py:780-815 does, from the shared
    reference-like initial state; the EMA model is then scored on 32 held-out images with the reference's own
    non_max_suppression (conf 0.001, iou 0.7, multi_label) + match_predictions + ap_per_class."""
    import time
    from ultralytics.engine.validator import BaseValidator
    from ultralytics.utils.metrics import DetMetrics, box_iou
    P = MAP_PROTOCOL
    cfg = os.path.join(CFG_DIR, P["name"] + ".yaml")
    y = og.load_yaml(cfg)
    y["nc"] = P["nc"]
    g = og.build_graph(y)
    torch.manual_seed(0)
    m = DetectionModel(cfg, ch=3, nc=P["nc"], verbose=False)
    args = get_cfg(DEFAULT_CFG)
    m.args = args
    m.load_state_dict(og.default_init_state(g, seed=P["init_seed"]), strict=True)
    for k, v in m.named_parameters():
        v.requires_grad = ".dfl" not in k
    bs, nb, epochs = P["batch"], P["nb"], P["epochs"]
    fake = SimpleNamespace(args=args, model=m)
    accumulate = max(round(args.nbs / bs), 1)
    wd = args.weight_decay * bs * accumulate / args.nbs
    fake.optimizer = BaseTrainer.build_optimizer(fake, model=m, name="SGD", lr=args.lr0, momentum=args.momentum, decay=wd)
    fake.scaler = torch.cuda.amp.GradScaler(enabled=False)
    fake.ema = ModelEMA(m)
    lf = lambda x: max(1 - x / epochs, 0) * (1.0 - args.lrf) + args.lrf  # noqa: E731
    for pg in fake.optimizer.param_groups:
        pg["initial_lr"] = pg["lr"]
    nw = max(round(args.warmup_epochs * nb), 100)
    last_opt_step = -1
    train = [{k: torch.from_numpy(v) for k, v in b.items()} for b in planted_batches(1, nb, bs, P["imgsz"], P["nc"])]
    hist = []
    t0 = time.time()
    order_rng = None if perm_seed is None else np.random.default_rng(1000 + perm_seed)  # gen_map_dist: a neutral perturbation
    base_train = train
    for epoch in range(epochs):
        if order_rng is not None:  # the same eight batches in another order every epoch (what a shuffling loader does)
            train = [base_train[j] for j in order_rng.permutation(nb)]
        m.train()
        for j, x in enumerate(fake.optimizer.param_groups):  # scheduler.step() value of this epoch (LambdaLR on initial_lr)
            x["lr"] = x["initial_lr"] * lf(epoch)
        tl = None
        for i, batch in enumerate(train):
            ni = i + nb * epoch
            if ni <= nw:
                xi = [0, nw]
                accumulate = max(1, int(np.interp(ni, xi, [1, args.nbs / bs]).round()))
                for j, x in enumerate(fake.optimizer.param_groups):
                    x["lr"] = np.interp(ni, xi, [args.warmup_bias_lr if j == 0 else 0.0, x["initial_lr"] * lf(epoch)])
                    if "momentum" in x:
                        x["momentum"] = np.interp(ni, xi, [args.warmup_momentum, args.momentum])
            loss, items = m(batch)
            loss.backward()
            if ni - last_opt_step >= accumulate:
                BaseTrainer.optimizer_step(fake)
                last_opt_step = ni
            tl = items if tl is None else (tl * i + items) / (i + 1)
        hist.append(tl.detach().numpy().copy())
        print(f"epoch {epoch + 1}/{epochs} {hist[-1].round(3)}  {time.time() - t0:.0f}s", flush=True)
    ema = fake.ema.ema.eval()
    holder = SimpleNamespace(iouv=torch.linspace(0.5, 0.95, 10))
    stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[])
    with torch.no_grad():
        for b in planted_batches(2, P["nval"], bs, P["imgsz"], P["nc"]):
            preds = ops.non_max_suppression(ema(torch.from_numpy(b["img"])), 0.001, 0.7, multi_label=True, max_det=300)
            for si, pred in enumerate(preds):
                idx = b["batch_idx"] == si
                tcls = torch.from_numpy(b["cls"][idx]).squeeze(-1)
                tbox = ops.xywh2xyxy(torch.from_numpy(b["bboxes"][idx])) * P["imgsz"]
                tp = torch.zeros(len(pred), 10, dtype=torch.bool)
                if len(pred) and len(tcls):
                    tp = BaseValidator.match_predictions(holder, pred[:, 5], tcls, box_iou(tbox, pred[:, :4]))
                stats["tp"].append(tp); stats["conf"].append(pred[:, 4]); stats["pred_cls"].append(pred[:, 5]); stats["target_cls"].append(tcls)
    st = {k: torch.cat(v, 0).numpy() for k, v in stats.items()}
    dm = DetMetrics(names={i: str(i) for i in range(P["nc"])})
    dm.process(st["tp"], st["conf"], st["pred_cls"], st["target_cls"])
    print("reference mean_results (P, R, mAP50, mAP50-95):", dm.mean_results(), "perm_seed", perm_seed, flush=True)
    if not save:
        return np.stack(hist), np.asarray(dm.mean_results(), np.float64)
    npz("map_parity", loss_hist=np.stack(hist), mean_results=np.asarray(dm.mean_results(), np.float64), n_det=np.asarray(len(st["conf"])),
        protocol=np.asarray([P["nc"], P["imgsz"], P["batch"], P["nb"], P["epochs"], P["nval"], P["init_seed"]]))

def gen_map_dist(k=6):
    """The reference's DISTRIBUTION under the protocol of gen_map: the same run repeated with k neutral perturbations -- the eight
    training batches visited in another (seeded) order every epoch, nothing else changed -- so that the parity test compares means
    with a known spread instead of one trajectory of a chaotic system with another (map_parity_dist.npz: mean_results (k, 4),
    final-epoch loss items (k, 3), the order seeds)."""
    torch.set_num_threads(int(os.environ.get("DY_GOLDEN_THREADS", "4")))
    res, last = [], []
    for s in range(1, k + 1):
        hist, mr = gen_map(perm_seed=s, save=False)
        res.append(mr)
        last.append(hist[-1])
    res = np.stack(res)
    print("reference mAP50 over", k, "batch orders:", res[:, 2].round(4), "mean", res[:, 2].mean().round(4), "std", res[:, 2].std(ddof=1).round(4))
    npz("map_parity_dist", mean_results=res, last_loss=np.stack(last), order_seeds=np.arange(1, k + 1))


def gen_data():
    """Data pipeline: the REFERENCE YOLODataset + build_dataloader over the fixture dataset of
    cases.write_dataset -- train mode with every augmentation gain at zero (2 epochs, shuffled) and val mode (rect batches).
    cv2 is absent here; the three calls this path makes into it are given their documented meaning below (constant border,
    the rotation matrix formula of the OpenCV docs); images come from the *.npy siblings, and nothing is interpolated."""
    import math
    import shutil
    import tempfile
    import cv2
    from ultralytics.data import build_dataloader, build_yolo_dataset

    def copy_make_border(img, top, bottom, left, right, border_type, value=(0, 0, 0)):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
        out[...] = np.asarray(value, img.dtype)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out

    def rotation_matrix(angle, center, scale):
        a, b = scale * math.cos(math.radians(angle)), scale * math.sin(math.radians(angle))
        return np.array([[a, b, (1 - a) * center[0] - b * center[1]], [-b, a, b * center[0] + (1 - a) * center[1]]])

    def no_resize(*a, **k):
        raise RuntimeError("interpolation is outside the pinned subset")

    cv2.copyMakeBorder, cv2.getRotationMatrix2D, cv2.resize = copy_make_border, rotation_matrix, no_resize
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR = 0, 1
    root = tempfile.mkdtemp(prefix="dy_dataset_")
    write_dataset(root)
    zero = dict(mosaic=0.0, mixup=0.0, copy_paste=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0,
                perspective=0.0, flipud=0.0, fliplr=0.0)
    cfg = get_cfg(DEFAULT_CFG, overrides=dict(imgsz=DATASET_IMGSZ, task="detect", **zero))
    data = {"names": {0: "a", 1: "b", 2: "c", 3: "d"}, "nc": 4}
    arrs = {}

    def put(tag, batch):
        arrs[f"{tag}/img"] = batch["img"]
        arrs[f"{tag}/cls"] = batch["cls"].reshape(-1, 1)
        arrs[f"{tag}/bboxes"] = batch["bboxes"].reshape(-1, 4)
        arrs[f"{tag}/batch_idx"] = batch["batch_idx"]
        arrs[f"{tag}/files"] = np.array([os.path.basename(f) for f in batch["im_file"]])
        arrs[f"{tag}/ori_shape"] = np.array(batch["ori_shape"])
        arrs[f"{tag}/resized_shape"] = np.array(batch["resized_shape"])
        if "ratio_pad" in batch:
            arrs[f"{tag}/ratio_pad"] = np.array([[rp[0][0], rp[0][1], rp[1][0], rp[1][1]] for rp in batch["ratio_pad"]], dtype=np.float64)

    ds = build_yolo_dataset(cfg, os.path.join(root, "images", "train"), 4, data, mode="train")
    loader = build_dataloader(ds, 4, 0, shuffle=True, rank=-1)
    arrs["train/nb"] = len(loader)
    for ep in range(2):
        for i, batch in enumerate(loader):
            put(f"train/e{ep}/b{i}", batch)
    # the same training set with the two flips switched on: decisions come from Python's random in the pipeline's draw order
    import random
    cfg_f = get_cfg(DEFAULT_CFG, overrides=dict(imgsz=DATASET_IMGSZ, task="detect", **{**zero, "fliplr": 0.5, "flipud": 0.25}))
    ds = build_yolo_dataset(cfg_f, os.path.join(root, "images", "train"), 4, data, mode="train")
    loader = build_dataloader(ds, 4, 0, shuffle=True, rank=-1)
    random.seed(7)
    for ep in range(2):
        for i, batch in enumerate(loader):
            put(f"flip/e{ep}/b{i}", batch)
    # the full geometric pipeline: mosaic + random affine + flips.  Labels only (cv2.warpAffine is absent: the stand-in returns a
    # grey canvas of the requested size, so pixels of this section are not part of the fixture)
    def grey_warp(img, M, dsize=None, borderValue=(114, 114, 114), **k):
        return np.full((dsize[1], dsize[0], 3), 114, np.uint8)

    cv2.warpAffine = grey_warp
    geo = {**zero, "mosaic": 1.0, "degrees": 5.0, "translate": 0.1, "scale": 0.5, "shear": 2.0, "fliplr": 0.5, "flipud": 0.1}
    cfg_g = get_cfg(DEFAULT_CFG, overrides=dict(imgsz=DATASET_IMGSZ, task="detect", **geo))
    ds = build_yolo_dataset(cfg_g, os.path.join(root, "images", "train"), 4, data, mode="train")
    loader = build_dataloader(ds, 4, 0, shuffle=True, rank=-1)
    random.seed(11)
    for ep in range(3):
        for i, batch in enumerate(loader):
            put(f"geo/e{ep}/b{i}", batch)
            del arrs[f"geo/e{ep}/b{i}/img"]
    ds = build_yolo_dataset(cfg, os.path.join(root, "images", "val"), 4, data, mode="val", rect=True, stride=32)
    loader = build_dataloader(ds, 4, 0, shuffle=False, rank=-1)
    arrs["val/nb"] = len(loader)
    for i, batch in enumerate(loader):
        put(f"val/b{i}", batch)
    npz("data", **arrs)
    shutil.rmtree(root)


def gen_data_mix():
    """MixUp + perspective (+ CopyPaste, a no-op for box-only labels: augment.py's CopyPaste only acts on segments) on top of the
    mosaic / affine / flip pipeline: labels of the REFERENCE pipeline over three epochs (pixels are not part of the fixture: cv2's
    warps are absent and stand in as grey canvases, as in gen_data).  Python's random seeded 13, numpy's (MixUp's beta draw) 5."""
    import math
    import random
    import shutil
    import tempfile
    import cv2
    from ultralytics.data import build_dataloader, build_yolo_dataset

    def copy_make_border(img, top, bottom, left, right, border_type, value=(0, 0, 0)):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
        out[...] = np.asarray(value, img.dtype)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out

    def rotation_matrix(angle, center, scale):
        a, b = scale * math.cos(math.radians(angle)), scale * math.sin(math.radians(angle))
        return np.array([[a, b, (1 - a) * center[0] - b * center[1]], [-b, a, b * center[0] + (1 - a) * center[1]]])

    def grey_warp(img, M, dsize=None, borderValue=(114, 114, 114), **k):
        return np.full((dsize[1], dsize[0], 3), 114, np.uint8)

    cv2.copyMakeBorder, cv2.getRotationMatrix2D = copy_make_border, rotation_matrix
    cv2.warpAffine = cv2.warpPerspective = grey_warp
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR = 0, 1
    root = tempfile.mkdtemp(prefix="dy_dataset_")
    write_dataset(root)
    data = {"names": {0: "a", 1: "b", 2: "c", 3: "d"}, "nc": 4}
    mix = dict(mosaic=0.7, mixup=0.5, copy_paste=0.3, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=5.0, translate=0.1, scale=0.5, shear=2.0,
               perspective=0.0005, flipud=0.1, fliplr=0.5)
    cfg = get_cfg(DEFAULT_CFG, overrides=dict(imgsz=DATASET_IMGSZ, task="detect", **mix))
    ds = build_yolo_dataset(cfg, os.path.join(root, "images", "train"), 4, data, mode="train")
    loader = build_dataloader(ds, 4, 0, shuffle=True, rank=-1)
    arrs = {}
    random.seed(13)
    np.random.seed(5)
    for ep in range(3):
        for i, batch in enumerate(loader):
            tag = f"mix/e{ep}/b{i}"
            arrs[f"{tag}/cls"] = batch["cls"].reshape(-1, 1)
            arrs[f"{tag}/bboxes"] = batch["bboxes"].reshape(-1, 4)
            arrs[f"{tag}/batch_idx"] = batch["batch_idx"]
            arrs[f"{tag}/files"] = np.array([os.path.basename(f) for f in batch["im_file"]])
    npz("data_mix", **arrs)
    shutil.rmtree(root)


def gen_two_stage():
    """Two-stage inference: outputs of the reference script's OWN functions.  double_inference.py is
    a Kaggle script whose import has side effects (creates /kaggle/... folders, opens a log file), so only its function
    definitions are compiled -- read from the reference tree at generation time, nothing is copied -- and called on seeded
    inputs.  torchvision is absent here, so torchvision_nms takes its built-in fallback branch."""
    import ast
    src = open(os.path.join(_refimport.REF, "double_inference.py")).read()
    tree = ast.parse(src)
    want = {"calculate_iou_tensor", "calculate_iou", "calculate_optimal_crop_batch", "scale_boxes_vectorized", "torchvision_nms",
            "process_refined_boxes_optimized"}
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    for f in fns:
        f.decorator_list = []  # @torch.jit.script: same arithmetic, eager
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "double_inference.py", "exec"), ns)
    rng = np.random.default_rng(11)
    arrs = {}
    # crops
    W, H = 900, 600
    n = 64
    c = np.stack([rng.uniform(0, W, n), rng.uniform(0, H, n)], 1)
    wh = np.exp(rng.uniform(np.log(2), np.log(300), (n, 2)))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
    boxes[:4] = [[0, 0, 5, 5], [W - 6, H - 6, W, H], [10.5, 20.25, 14.75, 300.5], [-20, -30, 50, 40]]
    dets = [{"bbox": [float(v) for v in b], "score": 0.5, "category_id": 0} for b in boxes]
    crops = ns["calculate_optimal_crop_batch"](dets, W, H)
    arrs["crop/boxes"], arrs["crop/wh"] = boxes, np.array([W, H])
    arrs["crop/rects"] = np.array([[c["x1"], c["y1"], c["x2"], c["y2"]] for c in crops])
    # scale + refine: per case one original detection and a candidate set around it
    K = 40
    for k in range(K):
        ob = boxes[k % n].clip(0, [W, H, W, H]).astype(np.float32)
        if ob[2] - ob[0] < 2 or ob[3] - ob[1] < 2:
            ob = np.array([100, 100, 160, 150], np.float32)
        rect = arrs["crop/rects"][k % n]
        cw, ch = int(rect[2] - rect[0]), int(rect[3] - rect[1])  # Python ints, as crop.shape gives the script
        ratio = min(640 / cw, 640 / ch)
        nw, nh = int(cw * ratio), int(ch * ratio)
        px, py = (640 - nw) // 2, (640 - nh) // 2
        m = int(rng.integers(0, 12))
        # candidates in crop-canvas coordinates: jittered copies of the original box mapped into the canvas, plus strays
        base = (ob - np.array([rect[0], rect[1], rect[0], rect[1]])) * ratio + np.array([px, py, px, py])
        cand = (base[None] + rng.normal(0, 25, (m, 4))).astype(np.float32)
        if m > 2:
            cand[0] = rng.uniform(0, 640, 4)
            cand[1] = base  # exact hit
        cand = cand.clip(0, 640).astype(np.float32)
        labels = rng.integers(0, 3, m).astype(int)
        confs = rng.uniform(0.25, 1.0, m).astype(np.float32)
        if m > 4:
            confs[3] = confs[2]  # a tie in confidence
        odet = {"bbox": [float(v) for v in ob], "score": float(rng.uniform(0.25, 0.9)), "category_id": int(rng.integers(0, 3))}
        crop_info = {"x1": int(rect[0]), "y1": int(rect[1]), "x2": int(rect[2]), "y2": int(rect[3])}
        scaled = ns["scale_boxes_vectorized"](cand, px, py, crop_info, ratio)
        res = ns["process_refined_boxes_optimized"](scaled, labels, confs, odet, W, H) if m else None
        arrs[f"ref/{k}/cand"], arrs[f"ref/{k}/labels"], arrs[f"ref/{k}/confs"] = cand, labels, confs
        arrs[f"ref/{k}/orig"] = np.array(odet["bbox"] + [odet["score"], odet["category_id"]], np.float64)
        arrs[f"ref/{k}/rect"], arrs[f"ref/{k}/geom"] = rect, np.array([ratio, px, py], np.float64)
        arrs[f"ref/{k}/scaled"] = np.asarray(scaled, np.float32).reshape(-1, 4)
        arrs[f"ref/{k}/out"] = np.array(res["bbox"] + [res["score"], res["category_id"]], np.float64) if res else np.zeros(0)
    arrs["ref/n"] = K
    # per-class NMS
    for k, m in enumerate((1, 2, 7, 40, 150)):
        c = np.stack([rng.uniform(50, 400, m), rng.uniform(50, 300, m)], 1)
        wh = rng.uniform(20, 120, (m, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        sc = rng.uniform(0.25, 1, m).astype(np.float32)
        if m > 6:
            sc[5] = sc[4]
            b[6] = b[2]  # identical boxes
        lab = rng.integers(0, 3, m)
        kb, ks, kl = ns["torchvision_nms"](b.tolist(), sc.tolist(), lab.tolist(), 0.45)
        arrs[f"nms/{k}/boxes"], arrs[f"nms/{k}/scores"], arrs[f"nms/{k}/labels"] = b, sc, lab
        arrs[f"nms/{k}/kept_boxes"], arrs[f"nms/{k}/kept_scores"] = np.array(kb, np.float32).reshape(-1, 4), np.array(ks, np.float32)
        arrs[f"nms/{k}/kept_labels"] = np.array(kl, np.int64)
    arrs["nms/n"] = 5
    npz("two_stage", **arrs)


def gen_e2e_ld():
    """The same protocol for the LDConv model (BASELINE configs[3] oracle): yolov8n-LD-P2.yaml, p_conv zero-initialised."""
    gen_e2e("yolov8n-LD-P2", "e2e_trainer_ld")


def gen_e2e(model_name="yolov8n-ASF-P2P2", out_name="e2e_trainer"):
    """End-to-end protocol of SURVEY section 8(c): the reference's UNMODIFIED trainer (DetectionTrainer(overrides).train():
    its dataset reader, loader, loss, optimizer, warm-up, EMA, validator, soft-NMS) on the synthetic set of
    cases.write_e2e_dataset, CPU, fp32, batch 2, all augmentation gains zero.  The fixture keeps its results.csv (per-epoch
    train losses, val metrics) -- the mAP-parity target for `YOLO.train(data=...)` of this package on the GPU."""
    import csv
    import glob
    import math
    import shutil
    import tempfile
    import cv2
    import cpuinfo

    def copy_make_border(img, top, bottom, left, right, border_type, value=(0, 0, 0)):
        out = np.empty((img.shape[0] + top + bottom, img.shape[1] + left + right, img.shape[2]), img.dtype)
        out[...] = np.asarray(value, img.dtype)
        out[top:top + img.shape[0], left:left + img.shape[1]] = img
        return out

    def rotation_matrix(angle, center, scale):
        a, b = scale * math.cos(math.radians(angle)), scale * math.sin(math.radians(angle))
        return np.array([[a, b, (1 - a) * center[0] - b * center[1]], [-b, a, b * center[0] + (1 - a) * center[1]]])

    def no_interp(*a, **k):
        raise RuntimeError("interpolation is outside the pinned subset")

    cv2.copyMakeBorder, cv2.getRotationMatrix2D, cv2.resize, cv2.warpAffine = copy_make_border, rotation_matrix, no_interp, no_interp
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR, cv2.INTER_AREA = 0, 1, 3
    cv2.setNumThreads = lambda n: None
    cpuinfo.get_cpu_info = lambda: {"brand_raw": "cpu"}
    os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
    from ultralytics.utils import USER_CONFIG_DIR
    import matplotlib
    ttf = glob.glob(os.path.join(os.path.dirname(matplotlib.__file__), "mpl-data", "fonts", "ttf", "DejaVuSans.ttf"))[0]
    for name in ("Arial.ttf", "Arial.Unicode.ttf"):
        shutil.copy(ttf, os.path.join(str(USER_CONFIG_DIR), name))  # check_font would otherwise try to download it
    from ultralytics.models.yolo.detect import DetectionTrainer
    from ultralytics.utils import callbacks as cb_mod
    import ultralytics.engine.trainer as trainer_mod
    # experiment-tracker integrations (neptune, wandb, ...) see this harness's permissive import stubs as installed packages
    cb_mod.add_integration_callbacks = lambda trainer: None
    trainer_mod.callbacks.add_integration_callbacks = cb_mod.add_integration_callbacks
    root = tempfile.mkdtemp(prefix="dy_e2e_")
    write_e2e_dataset(root)
    zero = dict(mosaic=0.0, mixup=0.0, copy_paste=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, degrees=0.0, translate=0.0, scale=0.0, shear=0.0,
                perspective=0.0, flipud=0.0, fliplr=0.0)
    ov = dict(model=os.path.join(CFG_DIR, model_name + ".yaml"), data=os.path.join(root, "data.yaml"), epochs=E2E["epochs"], batch=E2E["batch"],
              imgsz=E2E["imgsz"], device="cpu", workers=0, optimizer="SGD", amp=False, plots=False, val=True, close_mosaic=0, seed=0,
              deterministic=True, project=os.path.join(root, "runs"), name="e2e", exist_ok=True, **zero)
    import time
    t0 = time.time()
    tr = DetectionTrainer(overrides=ov)
    tr.train()
    print(f"reference trainer: {time.time() - t0:.0f} s")
    rows = list(csv.reader(open(os.path.join(root, "runs", "e2e", "results.csv"))))
    head = [h.strip() for h in rows[0]]
    vals = np.array([[float(v) for v in r] for r in rows[1:]], dtype=np.float64)
    print(head)
    print(vals[[0, len(vals) // 2, -1]])
    npz(out_name, header=np.array(head), results=vals, protocol=np.array(sorted(f"{k}={v}" for k, v in E2E.items())))
    shutil.rmtree(root)


if __name__ == "__main__":
    which = sys.argv[1:] or ["modules", "models", "loss", "nms", "trainer", "fullsize", "metrics"]
    for w in which:
        globals()["gen_" + w]()
