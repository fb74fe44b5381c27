"""Oracle (test infrastructure): model-YAML -> flat layer specs + state_dict layout.

Restates, as plain data, what the reference's ``parse_model`` (nn/tasks.py:780-1062) and module
constructors produce for the module names used by the DEAL-YOLO YAMLs: Conv, LDConv, C2f, SPPF,
Concat, nn.Upsample, ScalSeq, Add, Zoom_cat, Detect.  No nn.Module objects are built: a model is a
list of ``Layer`` records plus an ordinary ``dict[str, Tensor]`` carrying the reference's
parameter/buffer names.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import torch
import yaml

REG_MAX = 16


def make_divisible(x, d=8):
    """utils/ops.py:127-140."""
    return int(math.ceil(x / d) * d)


@dataclass
class Layer:
    i: int
    f: object  # int or list[int]
    kind: str
    cin: object  # int or list[int]
    cout: int
    args: dict = field(default_factory=dict)


@dataclass
class Graph:
    layers: list
    save: list
    nc: int
    ch: int
    scale: str
    strides: list  # per Detect level
    yaml: dict


def guess_scale(path):
    """nn/tasks.py:1083-1099: scale letter from the file name 'yolov8[nsmlx]...'."""
    m = re.search(r"yolov\d+([nslmx])", Path(path).stem)
    return m.group(1) if m else ""


def load_yaml(path):
    """nn/tasks.py:1065-1080: 'yolov8n-X.yaml' is served by 'yolov8-X.yaml' + scale 'n'."""
    path = Path(path)
    unified = Path(re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path)))
    src = unified if unified.exists() else path
    d = yaml.safe_load(src.read_text(errors="ignore"))
    d["scale"] = guess_scale(path)
    d["yaml_file"] = str(path)
    return d


def build_graph(d, ch=3, nc=None):
    """nn/tasks.py:780-1062 restricted to the hot-path module names."""
    d = dict(d)
    if nc:
        d["nc"] = nc
    nc = d["nc"]
    scales = d.get("scales")
    depth, width, max_ch = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf")
    scale = d.get("scale") or (next(iter(scales)) if scales else "")
    if scales:
        depth, width, max_ch = scales[scale]
    chs = [ch]
    down = [1]  # cumulative down-sampling of every layer output w.r.t. the image
    layers, save = [], []
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = [nc if a == "nc" else (None if a == "None" else a) for a in args]
        n = max(round(n * depth), 1) if n > 1 else n
        fl = [f] if isinstance(f, int) else list(f)
        src = [chs[x] for x in fl]
        dsrc = [down[x] for x in fl]
        a = {}
        if m in ("Conv", "LDConv", "C2f", "SPPF"):
            c1, c2 = src[0], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_ch) * width, 8)
            if m == "Conv":
                a = dict(k=args[1] if len(args) > 1 else 1, s=args[2] if len(args) > 2 else 1)
                ds = dsrc[0] * a["s"]
            elif m == "LDConv":
                a = dict(N=args[1], s=args[2] if len(args) > 2 else 1)
                ds = dsrc[0] * a["s"]
            elif m == "C2f":
                a = dict(n=n, shortcut=bool(args[1]) if len(args) > 1 else False)
                ds = dsrc[0]
            else:
                a = dict(k=args[1] if len(args) > 1 else 5)
                ds = dsrc[0]
            cin = c1
        elif m == "nn.Upsample":
            c2, cin, a, ds = src[0], src[0], dict(scale=int(args[1])), dsrc[0] / int(args[1])
        elif m == "Concat":
            c2, cin, ds = sum(src), src, dsrc[0]
        elif m == "Zoom_cat":
            c2, cin, ds = sum(src), src, dsrc[1]
        elif m == "Add":
            c2, cin, ds = src[-1], src, dsrc[-1]
        elif m == "ScalSeq":
            c2, cin, ds = make_divisible(args[0] * width, 8), src, dsrc[0]
        elif m == "Detect":
            c2, cin, ds = nc + 4 * REG_MAX, src, dsrc[0]
            a = dict(nc=nc)
        else:
            raise NotImplementedError(f"module '{m}' is outside the DEAL-YOLO hot path (SURVEY.md 8a)")
        layers.append(Layer(i, f, m, cin, c2, a))
        save.extend(x % i for x in fl if x != -1)
        if i == 0:
            chs, down = [], []
        chs.append(c2)
        down.append(ds)
    det = layers[-1]
    strides = [float(down[x]) for x in det.f] if det.kind == "Detect" else [32.0]
    return Graph(layers, sorted(set(save)), nc, ch, scale, strides, d)


# ----------------------------------------------------------------------------- state layout
def _conv_entries(p, c1, c2, k):
    """Conv = conv(bias-free) + BatchNorm2d; nn/modules/conv.py:41-55."""
    return {
        f"{p}.conv.weight": (c2, c1, k, k),
        f"{p}.bn.weight": (c2,), f"{p}.bn.bias": (c2,),
        f"{p}.bn.running_mean": (c2,), f"{p}.bn.running_var": (c2,), f"{p}.bn.num_batches_tracked": (),
    }


def state_layout(g: Graph):
    """name -> shape for every parameter/buffer, in the reference's registration order."""
    out = {}
    for L in g.layers:
        p = f"model.{L.i}"
        if L.kind == "Conv":
            out.update(_conv_entries(p, L.cin, L.cout, L.args["k"]))
        elif L.kind == "LDConv":  # nn/modules/conv.py:351-359
            N = L.args["N"]
            out[f"{p}.p_n"] = (1, 2 * N, 1, 1)
            out[f"{p}.conv.0.weight"] = (L.cout, L.cin, N, 1)
            for s in ("weight", "bias", "running_mean", "running_var"):
                out[f"{p}.conv.1.{s}"] = (L.cout,)
            out[f"{p}.conv.1.num_batches_tracked"] = ()
            out[f"{p}.p_conv.weight"] = (2 * N, L.cin, 3, 3)
            out[f"{p}.p_conv.bias"] = (2 * N,)
        elif L.kind == "C2f":  # nn/modules/block.py:209-226, 320-335
            c = int(L.cout * 0.5)
            out.update(_conv_entries(f"{p}.cv1", L.cin, 2 * c, 1))
            out.update(_conv_entries(f"{p}.cv2", (2 + L.args["n"]) * c, L.cout, 1))
            for j in range(L.args["n"]):
                out.update(_conv_entries(f"{p}.m.{j}.cv1", c, c, 3))
                out.update(_conv_entries(f"{p}.m.{j}.cv2", c, c, 3))
        elif L.kind == "SPPF":  # nn/modules/block.py:151-171
            c_ = L.cin // 2
            out.update(_conv_entries(f"{p}.cv1", L.cin, c_, 1))
            out.update(_conv_entries(f"{p}.cv2", c_ * 4, L.cout, 1))
        elif L.kind == "ScalSeq":  # nn/extra_modules/block.py:3414-3424
            c = L.cout
            if c != L.cin[0]:
                out.update(_conv_entries(f"{p}.conv0", L.cin[0], c, 1))
            out.update(_conv_entries(f"{p}.conv1", L.cin[1], c, 1))
            out.update(_conv_entries(f"{p}.conv2", L.cin[2], c, 1))
            out[f"{p}.conv3d.weight"] = (c, c, 1, 1, 1)
            out[f"{p}.conv3d.bias"] = (c,)
            for s in ("weight", "bias", "running_mean", "running_var"):
                out[f"{p}.bn.{s}"] = (c,)
            out[f"{p}.bn.num_batches_tracked"] = ()
        elif L.kind == "Detect":  # nn/modules/head.py:27-43
            ch = L.cin
            nc = L.args["nc"]
            c2, c3 = max(16, ch[0] // 4, REG_MAX * 4), max(ch[0], min(nc, 100))
            for br, cm, co in (("cv2", c2, 4 * REG_MAX), ("cv3", c3, nc)):
                for l, x in enumerate(ch):
                    out.update(_conv_entries(f"{p}.{br}.{l}.0", x, cm, 3))
                    out.update(_conv_entries(f"{p}.{br}.{l}.1", cm, cm, 3))
                    out[f"{p}.{br}.{l}.2.weight"] = (co, cm, 1, 1)
                    out[f"{p}.{br}.{l}.2.bias"] = (co,)
            out[f"{p}.dfl.conv.weight"] = (1, REG_MAX, 1, 1)
    return out


def ld_p_n(N):
    """nn/modules/conv.py:413-432: initial sampling shape, rows first then columns."""
    base = round(math.sqrt(N))
    rows, mod = N // base, N % base
    px = [r for r in range(rows) for _ in range(base)] + [rows] * mod
    py = [c for _ in range(rows) for c in range(base)] + list(range(mod))
    return torch.tensor(px + py, dtype=torch.int64).view(1, 2 * N, 1, 1)


def is_param(name):
    return not name.endswith(("running_mean", "running_var", "num_batches_tracked", ".p_n"))


def fill_state(layout, seed=0, g: Graph | None = None):
    """Deterministic, RNG-library-independent state: numpy PCG64 keyed by seed, name order.

    Used on BOTH sides of every golden comparison (applied to the reference model by
    tests/golden/make_golden.py and to the oracle/product by the tests), so the fixtures
    need not store weights.  Scales keep activations O(1) through 27 layers.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in layout.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith(".p_n"):
            sd[name] = ld_p_n(shape[1] // 2)
        elif name.endswith("dfl.conv.weight"):
            sd[name] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
        else:
            v = rng.standard_normal(shape).astype(np.float32)
            if name.endswith("running_var"):
                v = 1.0 + 0.25 * np.abs(v)
            elif name.endswith("running_mean"):
                v = 0.1 * v
            elif ".bn." in name or ".conv.1." in name:
                v = (1.0 + 0.1 * v) if name.endswith("weight") else 0.1 * v
            elif name.endswith("p_conv.weight"):
                v = v * (0.6 / math.sqrt(shape[1] * 9))
            elif name.endswith("p_conv.bias"):
                v = 0.3 * v
            elif name.endswith("bias"):
                v = 0.1 * v
            else:  # conv weights: variance-preserving
                fan_in = int(np.prod(shape[1:]))
                v = v * (1.0 / math.sqrt(fan_in))
            sd[name] = torch.from_numpy(np.ascontiguousarray(v))
    return sd


def default_init_state(g: Graph, seed=0):
    """Reference-like random init: PyTorch default Conv init + Detect.bias_init (head.py:76-83),
    BN weight 1 / bias 0, LDConv p_conv weight 0 (conv.py:357).  Distribution-equivalent, not
    RNG-stream-equivalent, to constructing the reference model under the same seed."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in state_layout(g).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith(".p_n"):
            sd[name] = ld_p_n(shape[1] // 2)
        elif name.endswith("dfl.conv.weight"):
            sd[name] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
        elif name.endswith("running_var"):
            sd[name] = torch.ones(shape)
        elif name.endswith("running_mean"):
            sd[name] = torch.zeros(shape)
        elif (".bn." in name or ".conv.1." in name):
            sd[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif name.endswith("p_conv.weight"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("weight"):
            bound = 1.0 / math.sqrt(int(np.prod(shape[1:])))  # kaiming_uniform(a=sqrt(5))
            sd[name] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
        else:  # conv bias
            w = sd[name[: -len("bias")] + "weight"]
            bound = 1.0 / math.sqrt(int(np.prod(w.shape[1:])))
            sd[name] = (torch.rand(shape, generator=gen) * 2 - 1) * bound
    det = g.layers[-1]
    if det.kind == "Detect":
        p = f"model.{det.i}"
        for l, s in enumerate(g.strides):
            sd[f"{p}.cv2.{l}.2.bias"][:] = 1.0
            sd[f"{p}.cv3.{l}.2.bias"][: g.nc] = math.log(5 / g.nc / (640 / s) ** 2)
    return sd
