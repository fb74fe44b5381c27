"""Post-processing ops (drop-in for the detection part of reference utils/ops.py) issued as HIP kernels."""
from __future__ import annotations

import ctypes as C

import torch

from ..hip import check, lib
from ..hip.engine import dev_empty

__all__ = ("non_max_suppression", "soft_nms", "decode_predictions", "xywh2xyxy", "xyxy2xywh", "make_divisible", "scale_boxes", "clip_boxes")


def make_divisible(x, divisor):
    import math
    if isinstance(divisor, torch.Tensor):
        divisor = int(divisor.max())
    return math.ceil(x / divisor) * divisor


def xywh2xyxy(x):
    """(x, y, w, h) -> (x1, y1, x2, y2) (reference utils/ops.py:527-546); plain tensor plumbing for callers."""
    y = torch.empty_like(x)
    dw, dh = x[..., 2] / 2, x[..., 3] / 2
    y[..., 0], y[..., 1] = x[..., 0] - dw, x[..., 1] - dh
    y[..., 2], y[..., 3] = x[..., 0] + dw, x[..., 1] + dh
    return y


def xyxy2xywh(x):
    y = torch.empty_like(x)
    y[..., 0], y[..., 1] = (x[..., 0] + x[..., 2]) / 2, (x[..., 1] + x[..., 3]) / 2
    y[..., 2], y[..., 3] = x[..., 2] - x[..., 0], x[..., 3] - x[..., 1]
    return y


def clip_boxes(boxes, shape):
    """Clip xyxy boxes to an image of (h, w) in place (reference utils/ops.py:147-165)."""
    if isinstance(boxes, torch.Tensor):
        boxes[..., 0].clamp_(0, shape[1])
        boxes[..., 1].clamp_(0, shape[0])
        boxes[..., 2].clamp_(0, shape[1])
        boxes[..., 3].clamp_(0, shape[0])
    else:
        boxes[..., [0, 2]] = boxes[..., [0, 2]].clip(0, shape[1])
        boxes[..., [1, 3]] = boxes[..., [1, 3]].clip(0, shape[0])
    return boxes


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None, padding=True, xywh=False):
    """Rescale boxes from the letterboxed shape ``img1_shape`` (h, w) to the original ``img0_shape``, in place, then clip
    (reference utils/ops.py:89-124; a few boxes per image: plain tensor arithmetic, the validator does it inside its kernel)."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1)
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    if padding:
        boxes[..., 0] -= pad[0]
        boxes[..., 1] -= pad[1]
        if not xywh:
            boxes[..., 2] -= pad[0]
            boxes[..., 3] -= pad[1]
    boxes[..., :4] /= gain
    return clip_boxes(boxes, img0_shape)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def decode_predictions(ho):
    """Detect inference path (reference nn/modules/head.py:50-74): HeadOut -> y (B, 4+nc, A) fp32."""
    L = lib()
    nl, dev = len(ho.box), ho.box[0].device
    B = ho.box[0].shape[0]
    A = sum(b.shape[1] * b.shape[2] for b in ho.box)
    y = dev_empty((B, 4 + ho.nc, A), torch.float32, dev)
    PP = C.c_void_p * nl
    box, cls = PP(*[b.data_ptr() for b in ho.box]), PP(*[c.data_ptr() for c in ho.cls])
    H = (C.c_int * nl)(*[b.shape[1] for b in ho.box])
    W = (C.c_int * nl)(*[b.shape[2] for b in ho.box])
    st = (C.c_float * nl)(*[float(s) for s in ho.strides])
    check(L.dy_decode_predictions(box, cls, H, W, st, nl, B, ho.nc, ho.cls[0].shape[-1], y.data_ptr(), _stream(dev)),
          "dy_decode_predictions")
    return y


def soft_nms(bboxes, scores, iou_thresh=0.5, sigma=0.5, score_threshold=0.25):
    """Gaussian soft-NMS with the reference's exact sequential semantics (utils/ops.py:260-290).  ``scores`` is decayed
    IN PLACE; returns the kept indices as a CPU int64 tensor, like the reference's ``torch.LongTensor(keep)``."""
    if bboxes.device.type != "cuda":
        raise RuntimeError("soft_nms: HIP path only (no CPU fallback)")
    n, dev = scores.shape[0], bboxes.device
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    b = bboxes.reshape(n, 4).float().contiguous()
    s = scores if (scores.dtype == torch.float32 and scores.is_contiguous()) else scores.float().contiguous()
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    oa, ob, keep = (dev_empty(n, torch.int32, dev) for _ in range(3))
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().dy_soft_nms(b.data_ptr(), s.data_ptr(), 0, cnt.data_ptr(), oa.data_ptr(), ob.data_ptr(), keep.data_ptr(),
                            nk.data_ptr(), 1, n, float(iou_thresh), float(sigma), float(score_threshold), 0.0, _stream(dev)),
          "dy_soft_nms")
    if s is not scores:
        scores.copy_(s)
    return keep[: int(nk.item())].long().cpu()


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        labels=(), max_det=300, nc=0, max_time_img=0.05, max_nms=30000, max_wh=7680, rotated=False):
    """Reference utils/ops.py:292-427 for detection outputs (no masks / apriori labels / rotated boxes).

    prediction: (B, 4+nc, A) [xywh, class scores] or the (y, feats) tuple of an eval forward.  Returns a list of (k, 6)
    tensors [x1, y1, x2, y2, conf, cls] on the prediction's device.  Unlike the reference the input is not modified and
    there is no wall-clock time limit (a nondeterminism source)."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    if labels or rotated:
        raise NotImplementedError("apriori labels / rotated boxes are outside the DEAL-YOLO hot path")
    if prediction.device.type != "cuda":
        raise RuntimeError("non_max_suppression: HIP path only (no CPU fallback)")
    pred = prediction.float().contiguous()
    B, no, A = pred.shape
    nc = nc or (no - 4)
    if no != 4 + nc:
        raise NotImplementedError("mask coefficients are outside the hot path")
    dev = pred.device
    multi_label = bool(multi_label) and nc > 1
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    cls_t = torch.tensor(list(classes), dtype=torch.int32, device=dev) if classes is not None else None
    L, st = lib(), _stream(dev)

    def candidates(cbox, csc, ccl, cap):
        check(L.dy_nms_candidates(pred.data_ptr(), B, nc, A, float(conf_thres), int(multi_label), 0 if cls_t is None else cls_t.data_ptr(),
                                  0 if cls_t is None else cls_t.numel(), 0 if cbox is None else cbox.data_ptr(),
                                  0 if csc is None else csc.data_ptr(), 0 if ccl is None else ccl.data_ptr(), cnt.data_ptr(), cap, st),
              "dy_nms_candidates")

    # Pass 1 counts, pass 2 fills buffers sized by the largest count: A*nc slots per image (the multi-label worst case) would be
    # 5.6 GB for yolov8n-p2 at 1280x1280, batch 32 (BASELINE configs[4]) where a trained model yields a few thousand candidates.
    candidates(None, None, None, 0)
    counts = cnt.cpu()
    cap = max(int(counts.max()) if B else 0, 8)
    cbox = dev_empty((B, cap, 4), torch.float32, dev)
    csc = dev_empty((B, cap), torch.float32, dev)
    ccl = dev_empty((B, cap), torch.float32, dev)
    candidates(cbox, csc, ccl, cap)
    if int(counts.max()) > max_nms:
        # utils/ops.py:395-396 ``x[x[:, 4].argsort(descending=True)[:max_nms]]``: keep the max_nms most confident, in descending
        # order (dy_nms_presort: radix select + ordered compaction + bitonic sort, one workgroup per image).  The reference's argsort
        # is unstable -- the order of EQUAL confidences is unspecified there; here ties keep their candidate (anchor-major) order.
        obox, osc, ocl = dev_empty((B, max_nms, 4), torch.float32, dev), dev_empty((B, max_nms), torch.float32, dev), dev_empty((B, max_nms), torch.float32, dev)
        ws = dev_empty(L.dy_nms_presort_workspace(B, max_nms), torch.uint8, dev)
        check(L.dy_nms_presort(cbox.data_ptr(), csc.data_ptr(), ccl.data_ptr(), cnt.data_ptr(), B, cap, max_nms, obox.data_ptr(),
                               osc.data_ptr(), ocl.data_ptr(), ws.data_ptr(), st), "dy_nms_presort")
        cbox, csc, ccl, cap = obox, osc, ocl, max_nms
    oa, ob, keep = (dev_empty((B, cap), torch.int32, dev) for _ in range(3))
    nk = torch.zeros(B, dtype=torch.int32, device=dev)
    check(L.dy_soft_nms(cbox.data_ptr(), csc.data_ptr(), ccl.data_ptr(), cnt.data_ptr(), oa.data_ptr(), ob.data_ptr(), keep.data_ptr(),
                        nk.data_ptr(), B, cap, float(iou_thres), 0.5, 0.25, 0.0 if agnostic else float(max_wh), st), "dy_soft_nms")
    nks = nk.cpu()
    out = []
    for b in range(B):
        i = keep[b, : min(int(nks[b]), max_det)].long()
        out.append(torch.cat((cbox[b, i], csc[b, i, None], ccl[b, i, None]), 1))
    return out
