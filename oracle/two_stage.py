"""Oracle (TEST INFRASTRUCTURE -- never imported by the product path): CPU restatement of the numeric functions of the
reference's two-stage inference script, double_inference.py.  Pinned by tests/golden/two_stage.npz, which holds outputs of
the reference's own functions (extracted from the script by tests/golden/make_golden.py::gen_two_stage)."""
from __future__ import annotations

import numpy as np


def calculate_iou(b1, b2):
    """calculate_iou_tensor, double_inference.py:70-87 (float32 like the torch tensors it runs on)."""
    b1, b2 = np.asarray(b1, np.float32), np.asarray(b2, np.float32)
    x1, y1 = max(b1[0], b2[0]), max(b1[1], b2[1])
    x2, y2 = min(b1[2], b2[2]), min(b1[3], b2[3])
    if x2 <= x1 or y2 <= y1:
        return 0.0
    inter = np.float32(x2 - x1) * np.float32(y2 - y1)
    a1 = np.float32(b1[2] - b1[0]) * np.float32(b1[3] - b1[1])
    a2 = np.float32(b2[2] - b2[0]) * np.float32(b2[3] - b2[1])
    if a1 <= 0 or a2 <= 0:
        return 0.0
    union = np.float32(np.float32(a1 + a2) - inter)
    return float(np.float32(inter / union)) if union > 0 else 0.0


def optimal_crops(boxes, img_width, img_height, pad_factor=0.2):
    """calculate_optimal_crop_batch, :98-126.  boxes: (n,4) python floats -> (n,4) int x1,y1,x2,y2."""
    out = []
    for x1, y1, x2, y2 in boxes:
        sw, sh = max(1, x2 - x1), max(1, y2 - y1)
        cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
        cw, ch = sw + 2 * sw * pad_factor, sh + 2 * sh * pad_factor
        n = [max(0, int(cx - cw / 2)), max(0, int(cy - ch / 2)), min(img_width, int(cx + cw / 2)), min(img_height, int(cy + ch / 2))]
        if n[2] - n[0] < 10 or n[3] - n[1] < 10:
            n = [max(0, int(cx - 16.0)), max(0, int(cy - 16.0)), min(img_width, int(cx + 16.0)), min(img_height, int(cy + 16.0))]
        out.append(n)
    return np.array(out, dtype=np.int64).reshape(-1, 4)


def crop_geometry(rect, size=640):
    """prepare_cropped_image_cv2, :129-149, without the pixels: ratio, (new_w, new_h), pad_x, pad_y."""
    w, h = int(rect[2] - rect[0]), int(rect[3] - rect[1])
    ratio = min(size / w, size / h)
    nw, nh = int(w * ratio), int(h * ratio)
    return ratio, (nw, nh), (size - nw) // 2, (size - nh) // 2


def crop_letterbox(img, rect, size=640):
    """The crop canvas with plain float bilinear sampling at cv2.resize's INTER_LINEAR positions (centre-aligned, edge-clamped);
    cv2's fixed-point rounding is NOT reproduced (no cv2 here): this checks the kernel's geometry and interpolation formula."""
    x1, y1, x2, y2 = [int(v) for v in rect]
    crop = img[y1:y2, x1:x2].astype(np.float32)
    ch, cw = crop.shape[:2]
    _, (nw, nh), px, py = crop_geometry(rect, size)
    fx = (np.arange(nw, dtype=np.float32) + np.float32(0.5)) * (np.float32(cw) / np.float32(nw)) - np.float32(0.5)
    fy = (np.arange(nh, dtype=np.float32) + np.float32(0.5)) * (np.float32(ch) / np.float32(nh)) - np.float32(0.5)
    sx, sy = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64)
    fx, fy = fx - sx, fy - sy
    lo = sx < 0; sx[lo] = 0; fx[lo] = 0
    hi = sx >= cw - 1; sx[hi] = cw - 1; fx[hi] = 0
    lo = sy < 0; sy[lo] = 0; fy[lo] = 0
    hi = sy >= ch - 1; sy[hi] = ch - 1; fy[hi] = 0
    sx1, sy1 = np.minimum(sx + 1, cw - 1), np.minimum(sy + 1, ch - 1)
    fx, fy = fx[None, :, None].astype(np.float32), fy[:, None, None].astype(np.float32)
    top = crop[sy][:, sx] * (1 - fx) + crop[sy][:, sx1] * fx
    bot = crop[sy1][:, sx] * (1 - fx) + crop[sy1][:, sx1] * fx
    out = np.full((size, size, 3), 114, np.uint8)
    out[py:py + nh, px:px + nw] = np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8)
    return out


def scale_boxes(boxes, pad_x, pad_y, rect, ratio):
    """scale_boxes_vectorized, :152-161 (float32 array, Python-float scalars)."""
    s = np.array(boxes, dtype=np.float32).reshape(-1, 4).copy()
    s[:, [0, 2]] -= int(pad_x)  # Python scalars: float32 arithmetic throughout
    s[:, [1, 3]] -= int(pad_y)
    s /= float(ratio)
    s[:, [0, 2]] += int(rect[0])
    s[:, [1, 3]] += int(rect[1])
    return s


def refine(scaled_boxes, labels, confs, orig_box, orig_score, orig_label, img_width, img_height):
    """process_refined_boxes_optimized, :263-303 -> (box, score, label) or None."""
    if len(scaled_boxes) == 0:
        return None
    m = labels == orig_label
    if not m.any():
        return None
    vb, vc, vl = scaled_boxes[m], confs[m], labels[m]
    ok = (vb[:, 2] > vb[:, 0]) & (vb[:, 3] > vb[:, 1]) & (vb[:, 0] >= 0) & (vb[:, 1] >= 0) & (vb[:, 2] <= img_width) & (vb[:, 3] <= img_height)
    if not ok.any():
        return None
    fb, fc, fl = vb[ok], vc[ok], vl[ok]
    best_i, best = -1, -1
    for i in range(len(fb)):
        iou = calculate_iou(orig_box, fb[i])
        if iou < 0.25:
            continue
        comb = fc[i] * 0.6 + iou * 0.4
        if comb > best:
            best, best_i = comb, i
    if best_i >= 0 and fc[best_i] > orig_score:
        return fb[best_i].tolist(), float(fc[best_i]), int(fl[best_i])
    return None


def nms_per_class(boxes, scores, labels, iou_threshold=0.45):
    """torchvision_nms, :164-203 (the branch taken without torchvision: descending-score greedy sweep per class, a box
    survives while IoU <= threshold with every kept box).  Returns the sorted kept indices."""
    boxes, scores, labels = np.asarray(boxes, np.float32).reshape(-1, 4), np.asarray(scores, np.float32), np.asarray(labels)
    keep = []
    for lab in np.unique(labels):
        idx = np.where(labels == lab)[0]
        order = sorted(range(len(idx)), key=lambda i: (-scores[idx[i]], i))
        remaining = list(order)
        while remaining:
            cur = remaining.pop(0)
            keep.append(int(idx[cur]))
            remaining = [r for r in remaining if calculate_iou(boxes[idx[cur]], boxes[idx[r]]) <= iou_threshold]
    return sorted(keep)
