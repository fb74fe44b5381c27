// Validation-path kernels: the per-batch half of DetectionValidator.update_metrics.
//
// Replaces, for a whole batch in ONE launch, the reference's per-image Python loop of _prepare_batch / _prepare_pred
// (models/yolo/detect/val.py:93-115: xywh2xyxy * imgsz, scale_boxes + clip_boxes), box_iou (utils/metrics.py:53-73) and
// BaseValidator.match_predictions (engine/validator.py:217-257, numpy path) -- which moves the IoU matrix to the host and
// sorts/uniques it once per IoU threshold.  The matching rule that code implements is: every detection keeps its
// best-IoU label of the same class (if that IoU reaches the threshold); every label then keeps the lowest-index (= most
// confident, NMS output is sorted) of the detections that chose it.  A detection's best label does not depend on the
// threshold, so one pass over the (labels x detections) IoUs serves all ten thresholds.
// Exact-IoU ties between two labels go to the lower label index here (numpy's unstable argsort leaves them undefined).
#include "common.h"
#include "dealyolo_hip.h"
#pragma clang fp contract(off)  // keep the fp32 evaluation order of the torch expressions (no fused multiply-add)

#define MP_MAXL 1024  // labels of one image staged in LDS
#define MP_MAXT 16    // IoU thresholds

struct MpArgs {
  const float* preds;    // (Ntot, 6) x1 y1 x2 y2 conf cls, network-input pixels
  const int* pred_off;   // (B+1)
  const float* t_bidx;
  const float* t_cls;
  const float* t_box;    // (n,4) xywh normalised
  const float* geom;     // (B,5) gain, padw, padh, ori_h, ori_w
  const float* iouv;
  unsigned char* tp;     // (Ntot, niou)
  float* predn;          // (Ntot, 6) native-space predictions (optional)
  int* status;           // |= 1 when an image has more than MP_MAXL labels
  int n_targets, niou, B;
  float img_h, img_w;
};

static __device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ __launch_bounds__(256) void match_predictions_kernel(MpArgs a) {
  __shared__ float lbox[MP_MAXL * 4];
  __shared__ float lcls[MP_MAXL];
  __shared__ int win[MP_MAXL * MP_MAXT];
  __shared__ int nl_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float gain = a.geom[b * 5 + 0], padw = a.geom[b * 5 + 1], padh = a.geom[b * 5 + 2];
  const float oh = a.geom[b * 5 + 3], ow = a.geom[b * 5 + 4];
  // ---- this image's labels, in their original order (wave 0: ballot compaction), moved to native space
  if (tid < 64) {
    int nl = 0;
    for (int base = 0; base < a.n_targets; base += 64) {
      const int i = base + tid;
      const bool mine = i < a.n_targets && (int)a.t_bidx[i] == b;
      const unsigned long long m = __ballot(mine);
      if (mine) {
        const int slot = nl + __popcll(m & ((1ull << tid) - 1ull));
        if (slot < MP_MAXL) {
          const float x = a.t_box[i * 4 + 0], y = a.t_box[i * 4 + 1], w = a.t_box[i * 4 + 2], h = a.t_box[i * 4 + 3];
          const float dw = w / 2.f, dh = h / 2.f;  // ops.xywh2xyxy, then * (w, h, w, h), then scale_boxes + clip_boxes
          lbox[slot * 4 + 0] = clampf(((x - dw) * a.img_w - padw) / gain, 0.f, ow);
          lbox[slot * 4 + 1] = clampf(((y - dh) * a.img_h - padh) / gain, 0.f, oh);
          lbox[slot * 4 + 2] = clampf(((x + dw) * a.img_w - padw) / gain, 0.f, ow);
          lbox[slot * 4 + 3] = clampf(((y + dh) * a.img_h - padh) / gain, 0.f, oh);
          lcls[slot] = a.t_cls[i];
        }
      }
      nl += __popcll(m);
    }
    if (tid == 0) {
      if (nl > MP_MAXL) {
        atomicOr(a.status, 1);
        nl = MP_MAXL;
      }
      nl_s = nl;
    }
  }
  __syncthreads();
  const int nl = nl_s;
  for (int i = tid; i < nl * a.niou; i += 256) win[i] = 0x7fffffff;
  __syncthreads();
  const int d0 = a.pred_off[b], d1 = a.pred_off[b + 1];
  // ---- pass 1: native-space prediction, best same-class label, candidate winner per (label, threshold)
  for (int d = d0 + tid; d < d1; d += 256) {
    const float* p = a.preds + (size_t)d * 6;
    const float x1 = clampf((p[0] - padw) / gain, 0.f, ow), y1 = clampf((p[1] - padh) / gain, 0.f, oh);
    const float x2 = clampf((p[2] - padw) / gain, 0.f, ow), y2 = clampf((p[3] - padh) / gain, 0.f, oh);
    if (a.predn) {
      float* q = a.predn + (size_t)d * 6;
      q[0] = x1; q[1] = y1; q[2] = x2; q[3] = y2; q[4] = p[4]; q[5] = p[5];
    }
    const float area_d = (x2 - x1) * (y2 - y1);
    float best = 0.f;
    int bl = -1;
    for (int l = 0; l < nl; ++l) {
      if (lcls[l] != p[5]) continue;
      const float lx1 = lbox[l * 4], ly1 = lbox[l * 4 + 1], lx2 = lbox[l * 4 + 2], ly2 = lbox[l * 4 + 3];
      const float iw = fmaxf(fminf(lx2, x2) - fmaxf(lx1, x1), 0.f), ih = fmaxf(fminf(ly2, y2) - fmaxf(ly1, y1), 0.f);
      const float inter = iw * ih;
      const float iou = inter / ((lx2 - lx1) * (ly2 - ly1) + area_d - inter + 1e-7f);  // box_iou(labels, preds)
      if (iou > best) {
        best = iou;
        bl = l;
      }
    }
    if (bl >= 0)
      for (int t = 0; t < a.niou; ++t)
        if (best >= a.iouv[t]) atomicMin(&win[bl * a.niou + t], d);
    // stash (best, label) for pass 2 in the output row: tp is written below, reuse registers via recompute-free trick
    // (kept in registers: the same thread handles the same detections in pass 2)
  }
  __syncthreads();
  // ---- pass 2: a detection is a true positive at threshold t when it won its label
  for (int d = d0 + tid; d < d1; d += 256) {
    const float* p = a.preds + (size_t)d * 6;
    const float x1 = clampf((p[0] - padw) / gain, 0.f, ow), y1 = clampf((p[1] - padh) / gain, 0.f, oh);
    const float x2 = clampf((p[2] - padw) / gain, 0.f, ow), y2 = clampf((p[3] - padh) / gain, 0.f, oh);
    const float area_d = (x2 - x1) * (y2 - y1);
    float best = 0.f;
    int bl = -1;
    for (int l = 0; l < nl; ++l) {
      if (lcls[l] != p[5]) continue;
      const float lx1 = lbox[l * 4], ly1 = lbox[l * 4 + 1], lx2 = lbox[l * 4 + 2], ly2 = lbox[l * 4 + 3];
      const float iw = fmaxf(fminf(lx2, x2) - fmaxf(lx1, x1), 0.f), ih = fmaxf(fminf(ly2, y2) - fmaxf(ly1, y1), 0.f);
      const float inter = iw * ih;
      const float iou = inter / ((lx2 - lx1) * (ly2 - ly1) + area_d - inter + 1e-7f);
      if (iou > best) {
        best = iou;
        bl = l;
      }
    }
    for (int t = 0; t < a.niou; ++t)
      a.tp[(size_t)d * a.niou + t] = (bl >= 0 && best >= a.iouv[t] && win[bl * a.niou + t] == d) ? 1 : 0;
  }
}

extern "C" int dy_match_predictions(const float* preds, const int* pred_off, const float* t_batch_idx, const float* t_cls,
                                    const float* t_boxes, int n_targets, const float* geom, const float* iouv, int niou,
                                    int B, int img_h, int img_w, unsigned char* tp, float* predn, int* status,
                                    hipStream_t stream) {
  if (B < 1 || niou < 1 || niou > MP_MAXT || !preds || !pred_off || !geom || !iouv || !tp || !status || n_targets < 0) return DY_ERR_ARG;
  MpArgs a{preds, pred_off, t_batch_idx, t_cls, t_boxes, geom, iouv, tp, predn, status, n_targets, niou, B, (float)img_h, (float)img_w};
  hipLaunchKernelGGL(match_predictions_kernel, dim3(B), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// utils/metrics.py:53-73 box_iou as a standalone operator: (n,4) x (m,4) xyxy -> (n,m)
__global__ __launch_bounds__(256) void box_iou_kernel(const float* b1, int n, const float* b2, int m, float* out) {
  const long total = (long)n * m;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float* p = b1 + (i / m) * 4;
    const float* q = b2 + (i % m) * 4;
    const float iw = fmaxf(fminf(p[2], q[2]) - fmaxf(p[0], q[0]), 0.f), ih = fmaxf(fminf(p[3], q[3]) - fmaxf(p[1], q[1]), 0.f);
    const float inter = iw * ih;
    out[i] = inter / ((p[2] - p[0]) * (p[3] - p[1]) + (q[2] - q[0]) * (q[3] - q[1]) - inter + 1e-7f);
  }
}
extern "C" int dy_box_iou(const float* box1, int n, const float* box2, int m, float* out, hipStream_t stream) {
  if (n < 0 || m < 0) return DY_ERR_ARG;
  if (n == 0 || m == 0) return DY_OK;
  long blocks = ((long)n * m + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(box_iou_kernel, dim3((int)blocks), dim3(256), 0, stream, box1, n, box2, m, out);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
