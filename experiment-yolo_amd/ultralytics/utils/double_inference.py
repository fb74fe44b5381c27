"""Two-stage ("double") inference (drop-in for the numeric part of reference double_inference.py:98-305; SURVEY.md section
8f row 4): every first-stage detection is cut out of the image with 20 % padding, letterboxed to 640x640, sent through the
model again, and replaced when the second look finds the same class with a higher confidence close to the original box;
a per-class hard NMS merges the result.

The reference does this one detection at a time on the host (PIL crop, cv2 resize, ``model.predict`` per crop, numpy / Python
loops).  Here the image is uploaded once; ``dy_crop_letterbox_u8`` cuts all crops in one launch, the second pass runs as
batched forwards + the soft-NMS kernel, ``dy_refine_select`` picks the replacements for all detections in one launch and
``dy_nms_hard`` does the merge.  Function names and argument meaning follow the reference script; file / JSON handling,
torchmetrics scoring and the visualisations of that script are control plane.

One reference quirk is kept behind a switch: ``process_image_optimized`` zips the list of *successful* refinements with the
list of *all* refined indices (:437-441), so the k-th success overwrites the k-th candidate detection, not the one it was
computed for.  ``aligned=False`` reproduces that; ``aligned=True`` (default here) applies each refinement to its own
detection."""
from __future__ import annotations

import time

import numpy as np
import torch

from ..hip import check, lib
from . import ops
from ..hip.engine import dev_empty

CONF_THRESHOLD = 0.25       # double_inference.py:25
NMS_IOU_THRESHOLD = 0.45    # :27
CROP_SIZE = 640


def calculate_optimal_crop_batch(detections, img_width, img_height, pad_factor=0.2):
    """:98-126 (Python int/float arithmetic kept: int() truncates toward zero)."""
    crops = []
    for detection in detections:
        x1, y1, x2, y2 = detection["bbox"]
        sw, sh = max(1, x2 - x1), max(1, y2 - y1)
        cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
        crop_w, crop_h = sw + 2 * (sw * pad_factor), sh + 2 * (sh * pad_factor)
        nx1, ny1 = max(0, int(cx - crop_w / 2)), max(0, int(cy - crop_h / 2))
        nx2, ny2 = min(img_width, int(cx + crop_w / 2)), min(img_height, int(cy + crop_h / 2))
        if nx2 - nx1 < 10 or ny2 - ny1 < 10:
            m = 32
            nx1, ny1 = max(0, int(cx - m / 2)), max(0, int(cy - m / 2))
            nx2, ny2 = min(img_width, int(cx + m / 2)), min(img_height, int(cy + m / 2))
        crops.append({"x1": nx1, "y1": ny1, "x2": nx2, "y2": ny2})
    return crops


def crop_geometry(crop_info, size=CROP_SIZE):
    """The letterbox bookkeeping of prepare_cropped_image_cv2 (:129-149), or None for an empty crop."""
    w, h = crop_info["x2"] - crop_info["x1"], crop_info["y2"] - crop_info["y1"]
    if w <= 0 or h <= 0:
        return None
    ratio = min(size / w, size / h)
    new_size = (int(w * ratio), int(h * ratio))
    return {"original_size": (w, h), "new_size": new_size, "pad_x": (size - new_size[0]) // 2, "pad_y": (size - new_size[1]) // 2,
            "ratio": ratio}


def prepare_cropped_images(image, crop_infos, size=CROP_SIZE):
    """image: (H,W,3) uint8 tensor on the device.  -> ((K,size,size,3) uint8 batch, [geometry dict per crop])."""
    H, W = image.shape[:2]
    geos = [crop_geometry(c, size) for c in crop_infos]
    assert all(g is not None and g["new_size"][0] > 0 and g["new_size"][1] > 0 for g in geos), "filter empty crops first"
    K = len(crop_infos)
    out = dev_empty((K, size, size, 3), torch.uint8, image.device)
    if K:
        rects = torch.tensor([[c["x1"], c["y1"], c["x2"], c["y2"]] for c in crop_infos], dtype=torch.int32).to(image.device)
        geom = torch.tensor([[g["new_size"][0], g["new_size"][1], g["pad_x"], g["pad_y"]] for g in geos], dtype=torch.int32).to(image.device)
        check(lib().dy_crop_letterbox_u8(image.data_ptr(), H, W, rects.data_ptr(), geom.data_ptr(), K, size, out.data_ptr(),
                                         torch.cuda.current_stream(image.device).cuda_stream), "dy_crop_letterbox_u8")
    return out, geos


def scale_boxes_vectorized(boxes, pad_x, pad_y, crop_info, ratio):
    """:152-161 (host copy for callers of the reference API; the device path does this inside dy_refine_select)."""
    if boxes.size == 0:
        return np.array([])
    scaled = boxes.copy()
    scaled[:, [0, 2]] -= pad_x
    scaled[:, [1, 3]] -= pad_y
    scaled /= ratio
    scaled[:, [0, 2]] += crop_info["x1"]
    scaled[:, [1, 3]] += crop_info["y1"]
    return scaled


def _second_stage(model, crops, conf, iou, batch_size):
    """model.predict on every crop: forward + decode + soft-NMS, boxes clipped to the crop canvas (ops.scale_boxes with equal
    shapes = clip_boxes)."""
    model.eval()
    preds = []
    with torch.no_grad():
        for i in range(0, crops.shape[0], batch_size):
            x = crops[i:i + batch_size].permute(0, 3, 1, 2).float() / 255
            y, _ = model(x)
            for p in ops.non_max_suppression(y, conf, iou, max_det=300):
                p[:, :4].clamp_(0, crops.shape[1])
                preds.append(p)
    return preds


def perform_batch_double_inference(image, model, detections, use_augment=False, conf=0.25, iou=0.7, batch_size=64,
                                   return_aligned=False):
    """:206-260.  image: (H,W,3) uint8 RGB (tensor | ndarray); detections: [{'bbox': [x1,y1,x2,y2], 'score', 'category_id'}].
    Returns ([refined dicts], seconds) like the reference -- only the successful refinements, in detection order -- or, with
    ``return_aligned``, one entry (dict | None) per detection."""
    if use_augment:
        raise NotImplementedError("test-time augmentation is not on the DEAL-YOLO hot path")
    t0 = time.time()
    dev = next(model.parameters()).device
    image = torch.as_tensor(image).to(dev).contiguous()
    H, W = image.shape[:2]
    crop_infos = calculate_optimal_crop_batch(detections, W, H)
    valid = [k for k, c in enumerate(crop_infos) if c["x2"] > c["x1"] and c["y2"] > c["y1"]
             and min(crop_geometry(c)["new_size"]) > 0]
    aligned = [None] * len(detections)
    if valid:
        cinfo = [crop_infos[k] for k in valid]
        crops, geos = prepare_cropped_images(image, cinfo)
        preds = _second_stage(model, crops, conf, iou, batch_size)
        K = len(valid)
        counts = [int(p.shape[0]) for p in preds]
        off = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), dtype=torch.int32, device=dev)
        dets = torch.cat([p.reshape(-1, 6).float() for p in preds], 0).contiguous() if sum(counts) else torch.zeros((0, 6), device=dev)
        orig = torch.tensor([[*detections[k]["bbox"], detections[k]["score"], detections[k]["category_id"]] for k in valid],
                            dtype=torch.float32, device=dev)
        rects = torch.tensor([[c["x1"], c["y1"], c["x2"], c["y2"]] for c in cinfo], dtype=torch.int32, device=dev)
        scale = torch.tensor([[g["ratio"], g["pad_x"], g["pad_y"]] for g in geos], dtype=torch.float32, device=dev)
        out = torch.zeros((K, 6), dtype=torch.float32, device=dev)
        found = torch.zeros(K, dtype=torch.int32, device=dev)
        check(lib().dy_refine_select(dets.data_ptr(), off.data_ptr(), orig.data_ptr(), rects.data_ptr(), scale.data_ptr(), K,
                                     float(W), float(H), out.data_ptr(), found.data_ptr(),
                                     torch.cuda.current_stream(dev).cuda_stream), "dy_refine_select")
        out_h, found_h = out.cpu().numpy(), found.cpu().numpy()
        for j, k in enumerate(valid):
            if found_h[j]:
                aligned[k] = {"bbox": out_h[j, :4].tolist(), "score": float(out_h[j, 4]), "category_id": int(out_h[j, 5])}
    dt = time.time() - t0
    return (aligned, dt) if return_aligned else ([r for r in aligned if r is not None], dt)


def torchvision_nms(boxes, scores, labels, iou_threshold=NMS_IOU_THRESHOLD, device=None):
    """:164-203: per-class greedy NMS; returns the kept (boxes, scores, labels) as lists in their original order."""
    if not boxes or len(boxes) == 0:
        return [], [], []
    dev = torch.device(device or "cuda:0")
    b = torch.tensor(boxes, dtype=torch.float32, device=dev).reshape(-1, 4).contiguous()
    s = torch.tensor(scores, dtype=torch.float32, device=dev).contiguous()
    lab = torch.tensor(labels, dtype=torch.int64)
    lf = lab.to(dev).float().contiguous()
    keep = torch.zeros(b.shape[0], dtype=torch.uint8, device=dev)
    check(lib().dy_nms_hard(b.data_ptr(), s.data_ptr(), lf.data_ptr(), b.shape[0], float(iou_threshold), keep.data_ptr(),
                            torch.cuda.current_stream(dev).cuda_stream), "dy_nms_hard")
    k = keep.bool().cpu()
    return b.cpu()[k].tolist(), s.cpu()[k].tolist(), lab[k].tolist()


def double_inference(image, model, predictions, conf_threshold=CONF_THRESHOLD, nms_iou=NMS_IOU_THRESHOLD, aligned=True):
    """The per-image flow of process_image_optimized (:404-449) without its file handling: ``predictions`` =
    {'boxes': [[x1,y1,x2,y2]], 'scores': [...], 'labels': [...]} (first stage); returns the refined dict + seconds spent."""
    cur = {k: list(v) for k, v in predictions.items()}
    idxs = [i for i, s in enumerate(cur["scores"]) if s >= conf_threshold]
    dets = [{"bbox": cur["boxes"][i], "score": cur["scores"][i], "category_id": cur["labels"][i]} for i in idxs]
    res, dt = perform_batch_double_inference(image, model, dets, return_aligned=True)
    pairs = zip(res, idxs) if aligned else zip([r for r in res if r is not None], idxs)
    for refined, i in pairs:
        if refined is not None:
            cur["boxes"][i], cur["scores"][i], cur["labels"][i] = refined["bbox"], refined["score"], refined["category_id"]
    if cur["boxes"]:
        cur["boxes"], cur["scores"], cur["labels"] = torchvision_nms(cur["boxes"], cur["scores"], cur["labels"], nms_iou,
                                                                     device=next(model.parameters()).device)
    return cur, dt
