"""Oracle (test infrastructure): functional CPU forward of the DEAL-YOLO graph.

Each function restates one reference module over a plain ``dict[str, Tensor]`` state and NCHW fp32
tensors.  Gradients come from torch autograd over these functions (the reference has no hand-written
backward on this path).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .graph import REG_MAX, Graph

# Storage emulation (tests only): with STORAGE_FP16 on, every tensor the HIP engine keeps in HBM as fp16 -- the raw conv output,
# the activation after BN + SiLU, pooled / up-sampled / summed maps -- is rounded to fp16 at the same point here, while all
# arithmetic stays fp32 and BN batch statistics come from the un-rounded conv output (as the conv epilogue sums them from its
# fp32 accumulators).  Against THIS oracle the engine differs only by its kernels' arithmetic (summation order), not by storage.
STORAGE_FP16 = False


def _q(t):
    return t.half().float() if STORAGE_FP16 else t


BN2D_EPS, BN2D_MOM = 1e-3, 0.03  # utils/torch_utils.py:347-349 (applies to every nn.BatchNorm2d)
BN3D_EPS, BN3D_MOM = 1e-5, 0.1  # nn.BatchNorm3d defaults, untouched by initialize_weights


def _bn(sd, p, x, training, eps, mom):
    nbt = sd.get(f"{p}.num_batches_tracked")
    if training and nbt is not None:
        nbt += 1
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"],
                        training, mom, eps)


def _bn_q(sd, p, y, training, eps, mom):
    """BatchNorm of a conv output; under STORAGE_FP16 the normalised tensor is the fp16-rounded one, the batch statistics are the
    un-rounded one's (2-D maps only; ScalSeq's 5-D volume is handled in scalseq)."""
    if not (STORAGE_FP16 and training):
        return _bn(sd, p, _q(y), training, eps, mom)
    dims = [0] + list(range(2, y.dim()))
    mean, var = y.mean(dims), y.var(dims, unbiased=False)
    n = y.numel() / y.shape[1]
    sd[f"{p}.running_mean"].mul_(1 - mom).add_(mom * mean.detach())
    sd[f"{p}.running_var"].mul_(1 - mom).add_(mom * (var.detach() * n / max(n - 1, 1)))
    shape = [1, -1] + [1] * (y.dim() - 2)
    sc = sd[f"{p}.weight"] / torch.sqrt(var + eps)
    return _q(y) * sc.view(shape) + (sd[f"{p}.bias"] - mean * sc).view(shape)


def conv_bn_silu(sd, p, x, k, s, training, act=True, round_out=True):
    """Conv.forward / forward_fuse, nn/modules/conv.py:41-59 (autopad k//2 :32-38).  ``round_out=False``: the caller adds a
    residual before the result is stored (one rounding, as the engine's fused epilogue does)."""
    y = F.conv2d(x, sd[f"{p}.conv.weight"], sd.get(f"{p}.conv.bias"), s, k // 2)
    if f"{p}.bn.weight" in sd:  # absent after fuse (nn/tasks.py:168-195)
        y = _bn_q(sd, f"{p}.bn", y, training, BN2D_EPS, BN2D_MOM)
    y = F.silu(y) if act else y
    return _q(y) if round_out else y


def ld_sample(x, off, pn, N, s):
    """The sampling stage of LDConv.forward, nn/modules/conv.py:368-404 (_get_p :446-454, _get_x_q :456-489): offsets
    (B,2N,h,w) -> bilinear 4-corner samples (B,C,h,w,N)."""
    B, C, H, W = x.shape
    h, w = off.shape[2:]
    rows = torch.arange(0, h * s, s, dtype=x.dtype).view(1, 1, h, 1)
    cols = torch.arange(0, w * s, s, dtype=x.dtype).view(1, 1, 1, w)
    pn = pn.to(x.dtype)
    pr = rows + pn[:, :N] + off[:, :N]  # sample row coordinate (first N channels)  :446-454
    pc = cols + pn[:, N:] + off[:, N:]  # sample col coordinate (last N channels)
    pr, pc = pr.permute(0, 2, 3, 1), pc.permute(0, 2, 3, 1)  # (B,h,w,N)
    r0, c0 = pr.detach().floor(), pc.detach().floor()
    r1, c1 = r0 + 1, c0 + 1
    r0, r1 = r0.clamp(0, H - 1), r1.clamp(0, H - 1)
    c0, c1 = c0.clamp(0, W - 1), c1.clamp(0, W - 1)
    pr, pc = pr.clamp(0, H - 1), pc.clamp(0, W - 1)
    g_lt = (1 + (r0 - pr)) * (1 + (c0 - pc))  # :390-393
    g_rb = (1 - (r1 - pr)) * (1 - (c1 - pc))
    g_lb = (1 + (r0 - pr)) * (1 - (c1 - pc))
    g_rt = (1 - (r1 - pr)) * (1 + (c0 - pc))
    xf = x.reshape(B, C, H * W)

    def take(r, c):  # :456-489
        idx = (r.long() * W + c.long()).reshape(B, 1, -1).expand(-1, C, -1)
        return xf.gather(2, idx).view(B, C, h, w, N)

    return (g_lt.unsqueeze(1) * take(r0, c0) + g_rb.unsqueeze(1) * take(r1, c1)
            + g_lb.unsqueeze(1) * take(r0, c1) + g_rt.unsqueeze(1) * take(r1, c0))


def ldconv(sd, p, x, N, s, training):
    """LDConv.forward, nn/modules/conv.py:366-410 with helpers :413-503."""
    B, C, H, W = x.shape
    off = F.conv2d(x, sd[f"{p}.p_conv.weight"], sd[f"{p}.p_conv.bias"], s, 1)  # (B,2N,h,w)
    h, w = off.shape[2:]
    xo = ld_sample(x, off, sd[f"{p}.p_n"], N, s)
    xo = xo.permute(0, 1, 2, 4, 3).reshape(B, C, h * N, w)  # 'b c h w n -> b c (h n) w'  :494-503
    y = F.conv2d(_q(xo), sd[f"{p}.conv.0.weight"], sd.get(f"{p}.conv.0.bias"), (N, 1))
    y = _bn_q(sd, f"{p}.conv.1", y, training, BN2D_EPS, BN2D_MOM)
    return _q(F.silu(y))


def c2f(sd, p, x, n, shortcut, training):
    """C2f.forward + Bottleneck.forward, nn/modules/block.py:222-226, :333-335."""
    y = list(conv_bn_silu(sd, f"{p}.cv1", x, 1, 1, training).chunk(2, 1))
    for j in range(n):
        t = conv_bn_silu(sd, f"{p}.m.{j}.cv1", y[-1], 3, 1, training)
        t = conv_bn_silu(sd, f"{p}.m.{j}.cv2", t, 3, 1, training, round_out=not shortcut)
        y.append(_q(y[-1] + t) if shortcut else t)
    return conv_bn_silu(sd, f"{p}.cv2", torch.cat(y, 1), 1, 1, training)


def sppf(sd, p, x, k, training):
    """SPPF.forward, nn/modules/block.py:166-171."""
    x = conv_bn_silu(sd, f"{p}.cv1", x, 1, 1, training)
    y1 = F.max_pool2d(x, k, 1, k // 2)
    y2 = F.max_pool2d(y1, k, 1, k // 2)
    y3 = F.max_pool2d(y2, k, 1, k // 2)
    return conv_bn_silu(sd, f"{p}.cv2", torch.cat((x, y1, y2, y3), 1), 1, 1, training)


def scalseq(sd, p, xs, training):
    """ScalSeq.forward, nn/extra_modules/block.py:3426-3443."""
    p3, p4, p5 = xs
    if f"{p}.conv0.conv.weight" in sd:
        p3 = conv_bn_silu(sd, f"{p}.conv0", p3, 1, 1, training)
    size = p3.shape[2:]
    p4 = F.interpolate(conv_bn_silu(sd, f"{p}.conv1", p4, 1, 1, training), size, mode="nearest")
    p5 = F.interpolate(conv_bn_silu(sd, f"{p}.conv2", p5, 1, 1, training), size, mode="nearest")
    vol = torch.stack([p3, p4, p5], 2)  # (B,C,3,H,W)
    vol = F.conv3d(vol, sd[f"{p}.conv3d.weight"], sd[f"{p}.conv3d.bias"])
    vol = _bn_q(sd, f"{p}.bn", vol, training, BN3D_EPS, BN3D_MOM)
    vol = F.leaky_relu(vol, 0.1)
    return _q(F.max_pool3d(vol, (3, 1, 1)).squeeze(2))


def zoom_cat(xs):
    """Zoom_cat.forward, nn/extra_modules/block.py:3406-3412."""
    l, m, s = xs
    size = m.shape[2:]
    l = _q(F.adaptive_max_pool2d(l, size) + F.adaptive_avg_pool2d(l, size))
    s = F.interpolate(s, size, mode="nearest")
    return torch.cat([l, m, s], 1)


def make_anchors(shapes, strides, offset=0.5):
    """utils/tal.py:294-307 from (h, w) pairs."""
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=torch.float32) + offset
        sy = torch.arange(h, dtype=torch.float32) + offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s)))
    return torch.cat(pts), torch.cat(st)


def dfl_expect(box):
    """DFL.forward, nn/modules/block.py:52-55: (B, 4*16, A) -> (B, 4, A)."""
    b, _, a = box.shape
    proj = torch.arange(REG_MAX, dtype=box.dtype).view(1, REG_MAX, 1, 1)
    return (box.view(b, 4, REG_MAX, a).transpose(2, 1).softmax(1) * proj).sum(1)


def decode(feats, strides, nc):
    """Detect inference path, nn/modules/head.py:50-74 + utils/tal.py:310-318 -> (B, 4+nc, A)."""
    b = feats[0].shape[0]
    no = nc + 4 * REG_MAX
    x_cat = torch.cat([f.reshape(b, no, -1) for f in feats], 2)
    anchors, st = make_anchors([f.shape[2:] for f in feats], strides)
    box, cls = x_cat.split((4 * REG_MAX, nc), 1)
    d = dfl_expect(box)
    lt, rb = d.chunk(2, 1)
    anc = anchors.t().unsqueeze(0)
    x1y1, x2y2 = anc - lt, anc + rb
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * st.t()
    return torch.cat((dbox, cls.sigmoid()), 1)


def detect(sd, p, xs, nc, training):
    """Detect.forward training path, nn/modules/head.py:45-48."""
    outs = []
    for l, x in enumerate(xs):
        a = conv_bn_silu(sd, f"{p}.cv2.{l}.0", x, 3, 1, training)
        a = conv_bn_silu(sd, f"{p}.cv2.{l}.1", a, 3, 1, training)
        a = F.conv2d(a, sd[f"{p}.cv2.{l}.2.weight"], sd[f"{p}.cv2.{l}.2.bias"])
        c = conv_bn_silu(sd, f"{p}.cv3.{l}.0", x, 3, 1, training)
        c = conv_bn_silu(sd, f"{p}.cv3.{l}.1", c, 3, 1, training)
        c = F.conv2d(c, sd[f"{p}.cv3.{l}.2.weight"], sd[f"{p}.cv3.{l}.2.bias"])
        outs.append(torch.cat((a, c), 1))
    return outs


def apply_layer(L, sd, x, training=True, strides=None, p=None):
    """Dispatch one layer spec (the per-module body of ``_predict_once``, nn/tasks.py:98-126)."""
    p = p or f"model.{L.i}"
    k = L.kind
    if k == "Conv":
        return conv_bn_silu(sd, p, x, L.args["k"], L.args["s"], training)
    if k == "LDConv":
        return ldconv(sd, p, x, L.args["N"], L.args["s"], training)
    if k == "C2f":
        return c2f(sd, p, x, L.args["n"], L.args["shortcut"], training)
    if k == "SPPF":
        return sppf(sd, p, x, L.args["k"], training)
    if k == "nn.Upsample":
        return F.interpolate(x, scale_factor=float(L.args["scale"]), mode="nearest")
    if k == "Concat":
        return torch.cat(x, 1)
    if k == "Zoom_cat":
        return zoom_cat(x)
    if k == "Add":
        return _q(torch.stack(x, 0).sum(0))
    if k == "ScalSeq":
        return scalseq(sd, p, x, training)
    if k == "Detect":
        feats = detect(sd, p, x, L.args["nc"], training)
        return feats if training else (decode(feats, strides, L.args["nc"]), feats)
    raise NotImplementedError(k)


def forward(g: Graph, sd, x, training=True, return_layers=False):
    """BaseModel._predict_once, nn/tasks.py:85-126.  Returns the Detect feature list (training) or
    ``(y, feats)`` (eval).  ``sd`` running statistics are updated in place when training."""
    ys = []
    for L in g.layers:
        if L.f != -1:
            x = ys[L.f] if isinstance(L.f, int) else [x if j == -1 else ys[j] for j in L.f]
        x = apply_layer(L, sd, x, training, g.strides)
        ys.append(x)
    return (x, ys) if return_layers else x


def fuse_state(g: Graph, sd):
    """BaseModel.fuse + fuse_conv_and_bn, nn/tasks.py:168-195, utils/torch_utils.py:171-198: fold BN into
    every ``Conv`` (LDConv's inner BN and ScalSeq's BatchNorm3d are never folded)."""
    out = dict(sd)
    for name in list(sd):
        if name.endswith(".bn.weight") and name[: -len(".bn.weight")] + ".conv.weight" in sd:
            p = name[: -len(".bn.weight")]
            w = sd[f"{p}.conv.weight"]
            if w.dim() != 4:
                continue
            scale = sd[f"{p}.bn.weight"] / torch.sqrt(BN2D_EPS + sd[f"{p}.bn.running_var"])
            out[f"{p}.conv.weight"] = w * scale.view(-1, 1, 1, 1)
            out[f"{p}.conv.bias"] = sd[f"{p}.bn.bias"] - sd[f"{p}.bn.running_mean"] * scale
            for s in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
                out.pop(f"{p}.bn.{s}")
    return out
