"""Case definitions shared by the fixture generator (make_golden.py, build container only) and the tests.

Only inputs are defined here (shapes, seeds, target boxes); expected outputs live in the .npz fixtures.
"""
import os

import numpy as np
import torch

from oracle.graph import Layer as L

MODES = {"ciou": (False, False), "wiou": (True, False), "ciou_nwd": (False, True), "wiou_nwd": (True, True)}

_MODULES = {
    "conv_k3s1": (L(0, -1, "Conv", 16, 32, dict(k=3, s=1)), [(2, 16, 12, 20)]),
    "conv_k3s2": (L(0, -1, "Conv", 16, 32, dict(k=3, s=2)), [(2, 16, 13, 21)]),
    "conv_k1": (L(0, -1, "Conv", 24, 16, dict(k=1, s=1)), [(2, 24, 9, 7)]),
    "conv_stem": (L(0, -1, "Conv", 3, 16, dict(k=3, s=2)), [(2, 3, 32, 48)]),
    "c2f_n2_sc": (L(0, -1, "C2f", 32, 32, dict(n=2, shortcut=True)), [(2, 32, 10, 12)]),
    "c2f_n1": (L(0, -1, "C2f", 48, 32, dict(n=1, shortcut=False)), [(2, 48, 8, 8)]),
    "sppf": (L(0, -1, "SPPF", 32, 32, dict(k=5)), [(2, 32, 9, 11)]),
    "scalseq": (L(0, [0, 1, 2], "ScalSeq", [16, 32, 64], 16, {}), [(2, 16, 16, 24), (2, 32, 8, 12), (2, 64, 4, 6)]),
    "scalseq_conv0": (L(0, [0, 1, 2], "ScalSeq", [32, 32, 64], 16, {}), [(2, 32, 8, 8), (2, 32, 4, 4), (2, 64, 2, 2)]),
    "zoom_cat": (L(0, [0, 1, 2], "Zoom_cat", [8, 8, 8], 24, {}), [(2, 8, 16, 16), (2, 8, 8, 8), (2, 8, 4, 4)]),
    "add": (L(0, [0, 1], "Add", [8, 8], 8, {}), [(2, 8, 6, 5), (2, 8, 6, 5)]),
    "ldconv_n3s2": (L(0, -1, "LDConv", 8, 16, dict(N=3, s=2)), [(2, 8, 13, 17)]),
    "ldconv_n1s1": (L(0, -1, "LDConv", 16, 8, dict(N=1, s=1)), [(2, 16, 9, 10)]),
    "ldconv_n5s1": (L(0, -1, "LDConv", 8, 8, dict(N=5, s=1)), [(1, 8, 7, 9)]),
    "ldconv_stem": (L(0, -1, "LDConv", 3, 16, dict(N=3, s=2)), [(2, 3, 24, 32)]),
}


def module_cases():
    """name -> (oracle Layer spec, case index).  State seed = 100+ci, input seeds 1000+10*ci+j, gy seed 2000+ci."""
    return {k: (v[0], ci) for ci, (k, v) in enumerate(_MODULES.items())}


def module_shapes():
    return {k: v[1] for k, v in _MODULES.items()}


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def synth_batch(seed, B, n_per, nc, hw=(64, 64), wh=(0.05, 0.35)):
    rng = np.random.default_rng(seed)
    img = torch.from_numpy(rng.random((B, 3, *hw), dtype=np.float32))
    n = B * n_per
    bi = torch.arange(B).repeat_interleave(n_per).float()
    cls = torch.from_numpy(rng.integers(0, nc, (n, 1)).astype(np.float32))
    w = torch.from_numpy((rng.random((n, 2)) * (wh[1] - wh[0]) + wh[0]).astype(np.float32))
    xy = torch.from_numpy((rng.random((n, 2)) * 0.7 + 0.15).astype(np.float32))
    return dict(img=img, batch_idx=bi, cls=cls, bboxes=torch.cat([xy, w], 1))


def _boxes(rows):
    t = torch.tensor(rows, dtype=torch.float32).view(-1, 6)
    return dict(batch_idx=t[:, 0], cls=t[:, 1:2], bboxes=t[:, 2:6])


def loss_cases():
    """Targets for the loss fixtures: a 3-level 64x64 head (16x16, 8x8, 4x4; strides 4/8/16), nc=6, B=2.
    Feature seeds: 400 + 10*case_index + level, scale 1.5."""
    b = synth_batch(300, 2, 5, 6)
    return {
        "random5": {k: b[k] for k in ("batch_idx", "cls", "bboxes")},
        "no_gt": _boxes([]),
        "one_gt": _boxes([[1, 3, 0.5, 0.5, 0.4, 0.3]]),
        "overlap": _boxes([[0, 1, 0.50, 0.50, 0.50, 0.50], [0, 2, 0.52, 0.50, 0.48, 0.52], [0, 1, 0.45, 0.55, 0.5, 0.45],
                           [1, 0, 0.3, 0.3, 0.4, 0.4], [1, 5, 0.32, 0.3, 0.4, 0.42]]),
        "tiny": _boxes([[0, 4, 0.51, 0.49, 0.01, 0.012], [0, 2, 0.2, 0.8, 0.02, 0.02], [1, 1, 0.7, 0.3, 0.3, 0.3]]),
        "ragged": _boxes([[0, 0, 0.3, 0.3, 0.2, 0.2]] + [[1, i % 6, 0.1 + 0.08 * i, 0.5, 0.15, 0.2 + 0.02 * i] for i in range(9)]),
    }


def tal_filler_case():
    """Adversarial input for the task-aligned assigner's zero-metric fillers (reference utils/tal.py:94-98, 145-161): one ground
    truth over the top-left quarter of a 64x64 image and a head whose box logits put (almost) every anchor's four distances at
    zero, so a predicted box is its anchor point, CIoU with the gt is negative and clamps to 0, and the alignment metric is 0
    everywhere -- except three anchors that predict the gt well.  ``torch.topk(metrics, 10)`` then fills seven places with
    zero-metric anchors in an implementation-defined order; those inside the gt box become foreground with target score 0.
    Returns (feats [3 x (2, 70, h, w)], batch)."""
    import torch
    # 256x256 image: 64x64 + 32x32 + 16x16 = 5,376 anchors.  (On a row this long torch.topk's CPU kernel returns the zero-metric
    # fillers from anchor indices 0-9 -- the top-left cells of level 0, inside this gt; on the 336-anchor rows of the other loss
    # fixtures it takes them from indices ~220-231, outside every gt of those cases.)
    shapes, strides = [(64, 64), (32, 32), (16, 16)], [4.0, 8.0, 16.0]
    feats = []
    for l, (h, w) in enumerate(shapes):
        f = rnd(700 + l, 2, 70, h, w, scale=0.3)
        box = f[:, :64].view(2, 4, 16, h, w)
        box[:, :, 0] += 12.0  # softmax mass on bin 0: every distance ~ 0
        feats.append(f)
    # three anchors of level 0 in image 0 that regress the gt (x1, y1, x2, y2) = (0, 0, 48, 48) px = (0, 0, 12, 12) grid units well
    for (iy, ix) in ((5, 5), (6, 6), (5, 6)):
        cx, cy = ix + 0.5, iy + 0.5
        for side, d in enumerate((cx - 0.0, cy - 0.0, 12.0 - cx, 12.0 - cy)):
            b = feats[0][0, 16 * side:16 * side + 16, iy, ix]
            b[:] = -6.0
            lo = int(d)
            b[lo], b[min(lo + 1, 15)] = 6.0 + 3.0 * (1 - (d - lo)), 6.0 + 3.0 * (d - lo)
        feats[0][0, 64 + 2, iy, ix] = 2.0  # a confident score for the gt's class
    batch = _boxes([[0, 2, 0.09375, 0.09375, 0.1875, 0.1875], [1, 1, 0.6, 0.6, 0.3, 0.3]])
    return feats, batch


def metric_cases():
    """Synthetic detection/label sets for the validation path: (name, seed, n_images, nc, max_labels, max_dets, jitter)."""
    return [("small", 11, 4, 3, 6, 12, 0.04), ("crowded", 12, 3, 2, 24, 60, 0.08), ("sparse", 13, 5, 6, 3, 5, 0.02),
            ("large", 14, 6, 6, 40, 300, 0.06)]


def metric_geometry(name, n_images, imgsz=640):
    """Per image (ori_shape (h, w), ratio_pad ((gain, gain), (padw, padh))) as the dataloader would report it: identity for
    most cases, a letterboxed 480x640 / 1080x1920 original for the images of 'crowded' and 'large'."""
    out = []
    for i in range(n_images):
        if name == "crowded":
            out.append(((480, 640), ((1.0, 1.0), (0.0, 80.0))))
        elif name == "large" and i % 2:
            g = imgsz / 1920
            out.append(((1080, 1920), ((g, g), (0.0, round((imgsz - 1080 * g) / 2 - 0.1)))))
        else:
            out.append(((imgsz, imgsz), ((1.0, 1.0), (0.0, 0.0))))
    return out


def synth_detections(seed, n_images, nc, max_labels, max_dets, jitter, imgsz=640):
    """Per image: labels (cls, normalised xywh) and NMS-style detections (xyxy px, conf, cls) sorted by confidence: each label
    spawns 0-3 jittered detections (some with the wrong class), plus random false positives; image 0 has no detections and the
    last image has no labels when n_images > 3."""
    rng = np.random.default_rng(seed)
    batch_idx, cls, bboxes, preds = [], [], [], []
    for i in range(n_images):
        nl = 0 if (i == n_images - 1 and n_images > 3) else int(rng.integers(1, max_labels + 1))
        c = rng.integers(0, nc, nl)
        xy = rng.random((nl, 2)) * 0.7 + 0.15
        wh = rng.random((nl, 2)) * 0.25 + 0.03
        batch_idx += [i] * nl
        cls += list(c)
        bboxes += list(np.concatenate([xy, wh], 1))
        dets = []
        if i != 0:
            for k in range(nl):
                for _ in range(int(rng.integers(0, 4))):
                    d_xy = xy[k] + rng.normal(0, jitter, 2) * wh[k] * 4
                    d_wh = wh[k] * np.exp(rng.normal(0, jitter * 3, 2))
                    dc = c[k] if rng.random() > 0.15 else rng.integers(0, nc)
                    dets.append([*(d_xy - d_wh / 2) * imgsz, *(d_xy + d_wh / 2) * imgsz, rng.random() * 0.9 + 0.05, dc])
            for _ in range(int(rng.integers(0, max(2, max_dets // 4)))):
                f_xy, f_wh = rng.random(2) * 0.8 + 0.1, rng.random(2) * 0.2 + 0.02
                dets.append([*(f_xy - f_wh / 2) * imgsz, *(f_xy + f_wh / 2) * imgsz, rng.random() * 0.6 + 0.01, rng.integers(0, nc)])
        dets = np.asarray(dets, dtype=np.float32).reshape(-1, 6)
        dets = dets[np.argsort(-dets[:, 4], kind="stable")][:max_dets]
        preds.append(dets)
    batch = dict(batch_idx=np.asarray(batch_idx, dtype=np.float32), cls=np.asarray(cls, dtype=np.float32).reshape(-1, 1),
                 bboxes=np.asarray(bboxes, dtype=np.float32).reshape(-1, 4))
    return batch, preds


def planted_batches(seed, n_batches, B, imgsz, nc, k=3, wh=(0.12, 0.35)):
    """Learnable synthetic detection data: dim noise background, k axis-aligned rectangles per image whose
    colour identifies the class.  Returns a list of batch dicts of numpy arrays (img float32 in [0,1], batch_idx, cls, bboxes)."""
    palette = np.array([[0.9, 0.1, 0.1], [0.1, 0.9, 0.1], [0.1, 0.1, 0.9], [0.9, 0.9, 0.1], [0.9, 0.1, 0.9], [0.1, 0.9, 0.9],
                        [0.9, 0.5, 0.1], [0.5, 0.1, 0.9]], np.float32)
    out = []
    for b in range(n_batches):
        rng = np.random.default_rng(seed * 7919 + b)
        img = rng.random((B, 3, imgsz, imgsz), dtype=np.float32) * 0.25
        bi, cl, bb = [], [], []
        for i in range(B):
            for _ in range(k):
                c = int(rng.integers(0, nc))
                w, h = rng.random(2) * (wh[1] - wh[0]) + wh[0]
                cx, cy = rng.random() * (1 - w) + w / 2, rng.random() * (1 - h) + h / 2
                x1, x2 = int((cx - w / 2) * imgsz), int((cx + w / 2) * imgsz)
                y1, y2 = int((cy - h / 2) * imgsz), int((cy + h / 2) * imgsz)
                img[i, :, y1:y2, x1:x2] = palette[c][:, None, None] + rng.random((3, y2 - y1, x2 - x1), dtype=np.float32) * 0.1
                bi.append(i); cl.append(c); bb.append([cx, cy, w, h])
        out.append(dict(img=img, batch_idx=np.asarray(bi, np.float32), cls=np.asarray(cl, np.float32).reshape(-1, 1),
                        bboxes=np.asarray(bb, np.float32)))
    return out


# ----------------------------------------------------------------------------- YOLO-format fixture dataset (data pipeline)
DATASET_NC = 4
DATASET_IMGSZ = 64


def dataset_spec():
    """name -> (h, w, label-file text | None).  Long side == 64 everywhere (no interpolation on the reference side, which
    would need cv2).  Label edge cases follow reference data/utils.py:96-165."""
    train = {
        "t00": (64, 64, "0 0.30 0.30 0.20 0.25\n1 0.70 0.60 0.30 0.20\n3 0.50 0.85 0.40 0.10\n"),
        "t01": (64, 64, ""),                                                   # empty label file -> background
        "t02": (64, 64, None),                                                 # label file missing -> background
        "t03": (48, 64, "2 0.5 0.5 0.25 0.5\n2 0.5 0.5 0.25 0.5\n1 0.2 0.2 0.1 0.3\n"),   # duplicate row removed (rows come back sorted)
        "t04": (64, 48, "0 0.5 0.5 0.02 0.02\n1 0.4 0.6 0.5 0.3\n"),          # first box < 2 px: dropped by the candidate filter
        "t05": (64, 64, "3 0.02 0.5 0.2 0.3\n0 0.98 0.97 0.5 0.5\n"),         # boxes reaching outside: clipped
        "t06": (64, 64, "0 -0.1 0.5 0.2 0.2\n"),                              # negative value -> corrupt, image dropped
        "t07": (64, 64, "0 0.5 1.2 0.2 0.2\n"),                               # non-normalised -> corrupt
        "t08": (64, 64, "0 0.5 0.5 0.2 0.2 0.9\n"),                           # six columns -> corrupt
        "t09": (64, 64, "7 0.5 0.5 0.2 0.2\n"),                               # class id beyond nc -> corrupt
        "t10": (40, 64, "1 0.5 0.5 0.9 0.04\n2 0.25 0.5 0.3 0.6\n"),          # 0.04*40 = 1.6 px high: dropped
        "t11": (64, 40, "0 0.5 0.01 0.4 0.3\n3 0.6 0.6 0.2 0.2\n"),           # loses > 90 % ... of nothing: keeps (area ratio 0.53)
        "t12": (64, 64, "2 0.5 0.005 0.5 0.2\n1 0.5 0.5 0.1 0.1\n"),          # clipped to < 10 % of its area?  no: 0.1/0.2 -> h2 < 2 px dropped
    }
    val = {
        "v00": (64, 64, "0 0.3 0.3 0.2 0.2\n"), "v01": (48, 64, "1 0.5 0.5 0.5 0.5\n2 0.2 0.2 0.1 0.1\n"),
        "v02": (64, 48, "3 0.6 0.4 0.3 0.3\n"), "v03": (32, 64, ""), "v04": (64, 32, "0 0.5 0.5 0.9 0.9\n"),
        "v05": (64, 56, "1 0.1 0.9 0.15 0.15\n2 0.9 0.1 0.15 0.15\n"), "v06": (40, 64, None),
    }
    return dict(train=train, val=val)


def write_dataset(root):
    """Materialise the fixture dataset under ``root`` (PNG + BGR *.npy siblings + labels + data.yaml)."""
    import zlib
    from PIL import Image
    for split, items in dataset_spec().items():
        os.makedirs(os.path.join(root, "images", split), exist_ok=True)
        os.makedirs(os.path.join(root, "labels", split), exist_ok=True)
        for name, (h, w, txt) in items.items():
            rng = np.random.default_rng(zlib.crc32(name.encode()))
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)  # RGB
            Image.fromarray(img).save(os.path.join(root, "images", split, name + ".png"))
            np.save(os.path.join(root, "images", split, name + ".npy"), img[..., ::-1].copy())  # the reference's caches are BGR
            if txt is not None:
                with open(os.path.join(root, "labels", split, name + ".txt"), "w") as f:
                    f.write(txt)
    with open(os.path.join(root, "data.yaml"), "w") as f:
        f.write("path: .\ntrain: images/train\nval: images/val\nnc: 4\nnames: [a, b, c, d]\n")


# ----------------------------------------------------------------------------- end-to-end trainer protocol
E2E = dict(imgsz=640, nc=6, n_train=16, n_val=16, boxes=8, epochs=40, batch=2)
E2E_COLOURS = [(220, 40, 40), (40, 200, 60), (50, 80, 230), (230, 210, 40), (200, 60, 210), (40, 210, 220)]


def write_e2e_dataset(root):
    """The synthetic set of SURVEY.md section 8(c): 640x640 images, grey 114 + U{0..19} noise, eight solid class-coloured
    rectangles per image with w, h in [0.03, 0.09]; PNG + BGR *.npy siblings + YOLO labels + data.yaml."""
    from PIL import Image
    s = E2E["imgsz"]
    for split, n, seed in (("train", E2E["n_train"], 100), ("val", E2E["n_val"], 200)):
        os.makedirs(os.path.join(root, "images", split), exist_ok=True)
        os.makedirs(os.path.join(root, "labels", split), exist_ok=True)
        for i in range(n):
            rng = np.random.default_rng(seed + i)
            img = (114 + rng.integers(0, 20, (s, s, 3))).astype(np.uint8)
            rows = []
            for _ in range(E2E["boxes"]):
                c = int(rng.integers(0, E2E["nc"]))
                w, h = rng.uniform(0.03, 0.09, 2)
                cx, cy = rng.uniform(0.06, 0.94, 2)
                x1, y1, x2, y2 = [int(round(v * s)) for v in (cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2)]
                img[y1:y2, x1:x2] = E2E_COLOURS[c]
                rows.append(f"{c} {cx:.6f} {cy:.6f} {w:.6f} {h:.6f}")
            name = f"{split}_{i:03d}"
            Image.fromarray(img).save(os.path.join(root, "images", split, name + ".png"))
            np.save(os.path.join(root, "images", split, name + ".npy"), img[..., ::-1].copy())
            with open(os.path.join(root, "labels", split, name + ".txt"), "w") as f:
                f.write("\n".join(rows) + "\n")
    with open(os.path.join(root, "data.yaml"), "w") as f:
        f.write(f"path: {os.path.abspath(root)}\ntrain: images/train\nval: images/val\nnc: 6\nnames: [c0, c1, c2, c3, c4, c5]\n")
