// Two-stage ("double") inference, reference double_inference.py:98-305 -- the device half.
//
// The reference re-reads the image with PIL per detection, crops with numpy, resizes with cv2, pads to 640x640, runs
// model.predict one crop at a time, moves every result to the host, and loops in Python over candidates and over NMS.
// Here the source image stays in HBM and three kernels bracket the batched second forward pass:
//   crop_letterbox   K crop rectangles of one image -> (K, S, S, 3) uint8 batch, bilinear (cv2.resize INTER_LINEAR sampling
//                    positions: centre-aligned, edge-clamped), centred on a 114 canvas (prepare_cropped_image_cv2 :129-149)
//   refine_select    per first-stage detection: second-stage rows mapped back to image space (scale_boxes_vectorized
//                    :152-161), the filters and the 0.6*conf + 0.4*IoU choice of process_refined_boxes_optimized :263-303
//   nms_hard         greedy per-class NMS in score order (torchvision_nms :164-203: suppress IoU > threshold)
#include "common.h"
#include "dealyolo_hip.h"
#pragma clang fp contract(off)

static inline int grid_for(long total) {
  long b = (total + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}

struct CropArgs {
  const unsigned char* img;  // (H, W, 3)
  const int* rects;          // (K, 4) x1 y1 x2 y2, x2/y2 exclusive
  const int* geom;           // (K, 4) new_w, new_h, pad_x, pad_y
  unsigned char* out;        // (K, S, S, 3)
  int H, W, K, S;
};

__global__ __launch_bounds__(256) void crop_letterbox_kernel(CropArgs a) {
  const long per = (long)a.S * a.S, total = per * a.K;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int k = (int)(idx / per);
    const int r = (int)(idx - (long)k * per);
    const int oy = r / a.S, ox = r - oy * a.S;
    const int x1 = a.rects[k * 4 + 0], y1 = a.rects[k * 4 + 1], cw = a.rects[k * 4 + 2] - x1, ch = a.rects[k * 4 + 3] - y1;
    const int nw = a.geom[k * 4 + 0], nh = a.geom[k * 4 + 1], px = a.geom[k * 4 + 2], py = a.geom[k * 4 + 3];
    unsigned char v[3] = {114, 114, 114};
    const int dx = ox - px, dy = oy - py;
    if (dx >= 0 && dx < nw && dy >= 0 && dy < nh && cw > 0 && ch > 0) {
      float fx = ((float)dx + 0.5f) * ((float)cw / (float)nw) - 0.5f;
      float fy = ((float)dy + 0.5f) * ((float)ch / (float)nh) - 0.5f;
      int sx = (int)floorf(fx), sy = (int)floorf(fy);
      fx -= (float)sx;
      fy -= (float)sy;
      if (sx < 0) { sx = 0; fx = 0.f; }
      if (sx >= cw - 1) { sx = cw - 1; fx = 0.f; }
      if (sy < 0) { sy = 0; fy = 0.f; }
      if (sy >= ch - 1) { sy = ch - 1; fy = 0.f; }
      const int sx1 = min(sx + 1, cw - 1), sy1 = min(sy + 1, ch - 1);
      const unsigned char* p00 = a.img + ((long)(y1 + sy) * a.W + x1 + sx) * 3;
      const unsigned char* p01 = a.img + ((long)(y1 + sy) * a.W + x1 + sx1) * 3;
      const unsigned char* p10 = a.img + ((long)(y1 + sy1) * a.W + x1 + sx) * 3;
      const unsigned char* p11 = a.img + ((long)(y1 + sy1) * a.W + x1 + sx1) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float top = (float)p00[c] * (1.f - fx) + (float)p01[c] * fx, bot = (float)p10[c] * (1.f - fx) + (float)p11[c] * fx;
        v[c] = (unsigned char)fminf(fmaxf(rintf(top * (1.f - fy) + bot * fy), 0.f), 255.f);
      }
    }
    unsigned char* o = a.out + idx * 3;
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  }
}

extern "C" int dy_crop_letterbox_u8(const void* img, int H, int W, const int* rects, const int* geom, int K, int S, void* out,
                                    hipStream_t stream) {
  if (K <= 0) return DY_OK;
  CropArgs a{(const unsigned char*)img, rects, geom, (unsigned char*)out, H, W, K, S};
  hipLaunchKernelGGL(crop_letterbox_kernel, dim3(grid_for((long)K * S * S)), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// IoU of calculate_iou_tensor (:70-87): 0 for empty intersection or a non-positive area, no epsilon.
static __device__ __forceinline__ float iou_plain(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2, float by2) {
  const float x1 = fmaxf(ax1, bx1), y1 = fmaxf(ay1, by1), x2 = fminf(ax2, bx2), y2 = fminf(ay2, by2);
  if (x2 <= x1 || y2 <= y1) return 0.f;
  const float inter = (x2 - x1) * (y2 - y1);
  const float a1 = (ax2 - ax1) * (ay2 - ay1), a2 = (bx2 - bx1) * (by2 - by1);
  if (a1 <= 0.f || a2 <= 0.f) return 0.f;
  const float uni = a1 + a2 - inter;
  return uni > 0.f ? inter / uni : 0.f;
}

struct RefineArgs {
  const float* dets;   // second-stage NMS rows (x1 y1 x2 y2 conf cls) of all crops, packed
  const int* off;      // (K+1) row offsets per crop
  const float* orig;   // (K, 6) the first-stage detection each crop was cut around
  const int* rects;    // (K, 4)
  const float* scale;  // (K, 3) ratio, pad_x, pad_y
  float* out;          // (K, 6) refined detection
  int* found;          // (K) 1 when a refinement replaces the original
  int K;
  float img_w, img_h;
};

// One wave per first-stage detection.  Candidate order matters only for exact score ties: the reference keeps the FIRST
// candidate reaching the best combined score (strict >), so the reduction prefers the lower row index.
__global__ __launch_bounds__(64) void refine_select_kernel(RefineArgs a) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const float ox1 = a.orig[k * 6 + 0], oy1 = a.orig[k * 6 + 1], ox2 = a.orig[k * 6 + 2], oy2 = a.orig[k * 6 + 3];
  const float oscore = a.orig[k * 6 + 4], ocls = a.orig[k * 6 + 5];
  const float ratio = a.scale[k * 3 + 0], padx = a.scale[k * 3 + 1], pady = a.scale[k * 3 + 2];
  const float cx = (float)a.rects[k * 4 + 0], cy = (float)a.rects[k * 4 + 1];
  float best = -1.f;
  int besti = 0x7fffffff;
  for (int i = a.off[k] + lane; i < a.off[k + 1]; i += 64) {
    const float* d = a.dets + (long)i * 6;
    if (d[5] != ocls) continue;
    const float x1 = (d[0] - padx) / ratio + cx, y1 = (d[1] - pady) / ratio + cy;
    const float x2 = (d[2] - padx) / ratio + cx, y2 = (d[3] - pady) / ratio + cy;
    if (!(x2 > x1 && y2 > y1 && x1 >= 0.f && y1 >= 0.f && x2 <= a.img_w && y2 <= a.img_h)) continue;
    const float iou = iou_plain(ox1, oy1, ox2, oy2, x1, y1, x2, y2);
    if (iou < 0.25f) continue;
    const float comb = d[4] * 0.6f + iou * 0.4f;
    if (comb > best) { best = comb; besti = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
  }
  if (lane == 0) {
    int ok = 0;
    if (best >= 0.f && besti != 0x7fffffff) {
      const float* d = a.dets + (long)besti * 6;
      if (d[4] > oscore) {
        ok = 1;
        a.out[k * 6 + 0] = (d[0] - padx) / ratio + cx;
        a.out[k * 6 + 1] = (d[1] - pady) / ratio + cy;
        a.out[k * 6 + 2] = (d[2] - padx) / ratio + cx;
        a.out[k * 6 + 3] = (d[3] - pady) / ratio + cy;
        a.out[k * 6 + 4] = d[4];
        a.out[k * 6 + 5] = d[5];
      }
    }
    a.found[k] = ok;
  }
}

extern "C" int dy_refine_select(const float* dets, const int* offsets, const float* orig, const int* rects, const float* scale,
                                int K, float img_w, float img_h, float* out, int* found, hipStream_t stream) {
  if (K <= 0) return DY_OK;
  RefineArgs a{dets, offsets, orig, rects, scale, out, found, K, img_w, img_h};
  hipLaunchKernelGGL(refine_select_kernel, dim3(K), dim3(64), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

#define NH_MAX 2048
// One workgroup per image.  order[] = indices by descending score (ties: lower index first); the greedy sweep is
// sequential over that order, the suppression of each kept box parallel over the rest.
__global__ __launch_bounds__(256) void nms_hard_kernel(const float* boxes, const float* scores, const float* labels, int n, float thr,
                                                       unsigned char* keep) {
  __shared__ float sb[NH_MAX * 4];
  __shared__ float ss[NH_MAX], sl[NH_MAX];
  __shared__ short order[NH_MAX], rank[NH_MAX];
  __shared__ unsigned char alive[NH_MAX];
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 256) {
    sb[i * 4 + 0] = boxes[i * 4 + 0]; sb[i * 4 + 1] = boxes[i * 4 + 1];
    sb[i * 4 + 2] = boxes[i * 4 + 2]; sb[i * 4 + 3] = boxes[i * 4 + 3];
    ss[i] = scores[i];
    sl[i] = labels[i];
    alive[i] = 1;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    int r = 0;
    const float s = ss[i];
    for (int j = 0; j < n; ++j) r += (ss[j] > s) || (ss[j] == s && j < i);
    rank[i] = (short)r;
    order[r] = (short)i;
  }
  __syncthreads();
  for (int t = 0; t < n; ++t) {
    const int i = order[t];
    if (alive[i]) {  // uniform: every thread reads the same shared byte after the barrier below
      const float x1 = sb[i * 4], y1 = sb[i * 4 + 1], x2 = sb[i * 4 + 2], y2 = sb[i * 4 + 3], l = sl[i];
      for (int j = tid; j < n; j += 256)
        if (alive[j] && rank[j] > t && sl[j] == l && iou_plain(x1, y1, x2, y2, sb[j * 4], sb[j * 4 + 1], sb[j * 4 + 2], sb[j * 4 + 3]) > thr)
          alive[j] = 0;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += 256) keep[i] = alive[i];
}

extern "C" int dy_nms_hard(const float* boxes, const float* scores, const float* labels, int n, float iou_thr, void* keep,
                           hipStream_t stream) {
  if (n <= 0) return DY_OK;
  if (n > NH_MAX) return DY_ERR_ARG;
  hipLaunchKernelGGL(nms_hard_kernel, dim3(1), dim3(256), 0, stream, boxes, scores, labels, n, iou_thr, (unsigned char*)keep);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
