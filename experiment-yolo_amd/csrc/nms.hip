// Inference post-process on device: Detect decode (reference nn/modules/head.py:50-74, nn/modules/block.py:52-55,
// utils/tal.py:310-318) and the reference's Gaussian soft-NMS with its order-dependent behaviour reproduced exactly
// (utils/ops.py:260-290 soft_nms, :162-199 bbox_iou_for_nms, :292-427 non_max_suppression) -- see SURVEY.md 8a row N2:
// first kept box is candidate 0, surviving scores are decayed in place, arg-max survivor is SWAPPED to the front,
// the last remaining box is never kept, threshold on decayed scores is the constant 0.25.
// One workgroup per image walks the sequential rounds; every round is a parallel pass + stable compaction.
#include "common.h"
#include "dealyolo_hip.h"

#pragma clang fp contract(off)  // keep the fp32 IoU arithmetic un-fused: it must round exactly like the ATen op sequence

// ---------------------------------------------------------------------------------------------- decode
struct DecArgs {
  const float* box[4];
  const float* cls[4];
  int H[4], W[4], a0[4];
  float stride[4];
  int nl, B, A, nc, ncp;
  float* y;  // (B, 4+nc, A)
};
__global__ __launch_bounds__(256) void decode_pred_kernel(DecArgs d) {
  const long total = (long)d.B * d.A;
  for (long ba = (long)blockIdx.x * 256 + threadIdx.x; ba < total; ba += (long)gridDim.x * 256) {
    const int b = (int)(ba / d.A), a = (int)(ba - (long)b * d.A);
    int l = 0;
    for (int k = 1; k < d.nl; ++k)
      if (a >= d.a0[k]) l = k;
    const int r = a - d.a0[l], iy = r / d.W[l], ix = r - iy * d.W[l];
    const float* bp = d.box[l] + (((size_t)b * d.H[l] + iy) * d.W[l] + ix) * 64;
    float e[4];
    for (int s = 0; s < 4; ++s) {
      float m = bp[s * 16];
      for (int k = 1; k < 16; ++k) m = fmaxf(m, bp[s * 16 + k]);
      float den = 0.f, num = 0.f;
      for (int k = 0; k < 16; ++k) {
        const float ex = expf(bp[s * 16 + k] - m);
        den += ex;
        num += ex * (float)k;
      }
      e[s] = num / den;
    }
    const float ax = ix + 0.5f, ay = iy + 0.5f, st = d.stride[l];
    const float x1 = ax - e[0], y1 = ay - e[1], x2 = ax + e[2], y2 = ay + e[3];
    float* yo = d.y + (size_t)b * (4 + d.nc) * d.A + a;
    yo[0] = (x1 + x2) / 2 * st;
    yo[(size_t)d.A] = (y1 + y2) / 2 * st;
    yo[(size_t)2 * d.A] = (x2 - x1) * st;
    yo[(size_t)3 * d.A] = (y2 - y1) * st;
    const float* cp = d.cls[l] + (((size_t)b * d.H[l] + iy) * d.W[l] + ix) * d.ncp;
    for (int c = 0; c < d.nc; ++c) yo[(size_t)(4 + c) * d.A] = 1.f / (1.f + expf(-cp[c]));
  }
}
extern "C" int dy_decode_predictions(const float* const* box, const float* const* cls, const int* H, const int* W,
                                     const float* stride, int nl, int B, int nc, int ncp, float* y, hipStream_t stream) {
  if (nl < 1 || nl > 4) return DY_ERR_ARG;
  DecArgs d{};
  int a0 = 0;
  for (int l = 0; l < nl; ++l) {
    d.box[l] = box[l]; d.cls[l] = cls[l]; d.H[l] = H[l]; d.W[l] = W[l]; d.stride[l] = stride[l]; d.a0[l] = a0;
    a0 += H[l] * W[l];
  }
  d.nl = nl; d.B = B; d.A = a0; d.nc = nc; d.ncp = ncp; d.y = y;
  long blocks = ((long)B * a0 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(decode_pred_kernel, dim3((int)blocks), dim3(256), 0, stream, d);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---------------------------------------------------------------------------------------------- block scan helper
// exclusive scan of one int per thread over a 1024-thread block; returns the block total in `total`
static __device__ __forceinline__ int block_excl_scan(int v, int* sh, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) sh[w] = x;
  __syncthreads();
  if (w == 0) {
    int s = lane < 16 ? sh[lane] : 0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int y = __shfl_up(s, o, 64);
      if (lane >= o) s += y;
    }
    if (lane < 16) sh[16 + lane] = s;  // inclusive per-wave totals
  }
  __syncthreads();
  total = sh[16 + 15];
  const int base = w ? sh[16 + w - 1] : 0;
  __syncthreads();
  return base + x - v;
}

// ---------------------------------------------------------------------------------------------- candidates
struct CandArgs {
  const float* pred;  // (B, 4+nc, A) xywh + class scores
  int B, nc, A, cap, multi_label, n_classes;
  float conf;
  const int* classes;  // optional filter list
  float* cbox;         // (B,cap,4) xyxy
  float* cscore;       // (B,cap)
  float* ccls;         // (B,cap)
  int* ccount;         // (B)  number of candidates (may exceed cap: overflow)
};
__global__ __launch_bounds__(1024) void nms_candidates_kernel(CandArgs c) {
  __shared__ int sh[32];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* P = c.pred + (size_t)b * (4 + c.nc) * c.A;
  int base = 0;
  for (int a0 = 0; a0 < c.A; a0 += 1024) {
    const int a = a0 + tid;
    int k = 0;
    float best = -INFINITY;
    int bj = 0;
    if (a < c.A) {
      for (int j = 0; j < c.nc; ++j) {
        const float s = P[(size_t)(4 + j) * c.A + a];
        if (s > best) {
          best = s;
          bj = j;
        }
      }
      if (best > c.conf) {  // xc, utils/ops.py:344
        if (c.multi_label) {
          for (int j = 0; j < c.nc; ++j) {
            const float s = P[(size_t)(4 + j) * c.A + a];
            bool ok = s > c.conf;
            if (ok && c.n_classes) {
              ok = false;
              for (int q = 0; q < c.n_classes; ++q) ok |= (c.classes[q] == j);
            }
            k += ok;
          }
        } else {
          bool ok = true;
          if (c.n_classes) {
            ok = false;
            for (int q = 0; q < c.n_classes; ++q) ok |= (c.classes[q] == bj);
          }
          k = ok;
        }
      }
    }
    int total;
    int pos = base + block_excl_scan(k, sh, total);
    if (k) {
      const float x = P[a], y = P[(size_t)c.A + a], w = P[(size_t)2 * c.A + a], h = P[(size_t)3 * c.A + a];
      const float dw = w / 2, dh = h / 2;  // xywh2xyxy, utils/ops.py:527-546
      const float x1 = x - dw, y1 = y - dh, x2 = x + dw, y2 = y + dh;
      for (int j = 0; j < c.nc && k; ++j) {
        float s;
        bool take;
        if (c.multi_label) {
          s = P[(size_t)(4 + j) * c.A + a];
          take = s > c.conf;
        } else {
          s = best;
          take = (j == bj);
        }
        if (take && c.n_classes) {
          bool ok = false;
          for (int q = 0; q < c.n_classes; ++q) ok |= (c.classes[q] == j);
          take = ok;
        }
        if (take) {
          if (pos < c.cap) {
            float* bo = c.cbox + ((size_t)b * c.cap + pos) * 4;
            bo[0] = x1; bo[1] = y1; bo[2] = x2; bo[3] = y2;
            c.cscore[(size_t)b * c.cap + pos] = s;
            c.ccls[(size_t)b * c.cap + pos] = (float)j;
          }
          ++pos;
          --k;
        }
      }
    }
    base += total;
  }
  if (tid == 0) c.ccount[b] = base;
}
extern "C" int dy_nms_candidates(const float* pred, int B, int nc, int A, float conf, int multi_label,
                                 const int* classes, int n_classes, float* cbox, float* cscore, float* ccls, int* ccount,
                                 int cap, hipStream_t stream) {
  CandArgs c{pred, B, nc, A, cap, multi_label, n_classes, conf, classes, cbox, cscore, ccls, ccount};
  hipLaunchKernelGGL(nms_candidates_kernel, dim3(B), dim3(1024), 0, stream, c);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---------------------------------------------------------------------------------------------- soft-NMS
struct SnArgs {
  const float* boxes;   // (B,cap,4) xyxy
  float* scores;        // (B,cap) decayed in place
  const float* cls;     // (B,cap) or null (no class offset)
  const int* count;     // (B) candidates per image
  int* order_a;         // (B,cap) scratch
  int* order_b;         // (B,cap) scratch
  int* keep;            // (B,cap) kept candidate indices in order
  int* nkeep;           // (B)
  int cap;
  float iou_thr, sigma, score_thr, class_offset;  // class_offset = 0 (agnostic) or max_wh
};
__global__ __launch_bounds__(1024) void soft_nms_kernel(SnArgs a) {
  __shared__ int sh[32];
  __shared__ float r_s[16];
  __shared__ int r_p[16];
  __shared__ int s_m;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int n = a.count[b];
  if (n > a.cap) n = a.cap;
  const float* B_ = a.boxes + (size_t)b * a.cap * 4;
  float* S = a.scores + (size_t)b * a.cap;
  const float* C_ = a.cls ? a.cls + (size_t)b * a.cap : nullptr;
  int* cur = a.order_a + (size_t)b * a.cap;
  int* nxt = a.order_b + (size_t)b * a.cap;
  int* keep = a.keep + (size_t)b * a.cap;
  for (int t = tid; t < n; t += 1024) cur[t] = t;  // order = arange(n)
  __syncthreads();
  int m = n, nk = 0;
  while (m > 1) {
    const int i = cur[0];
    if (tid == 0) keep[nk] = i;
    ++nk;
    const float oi = C_ ? C_[i] * a.class_offset : 0.f;
    const float x1 = B_[i * 4 + 0] + oi, y1 = B_[i * 4 + 1] + oi, x2 = B_[i * 4 + 2] + oi, y2 = B_[i * 4 + 3] + oi;
    const float w1 = x2 - x1, h1 = y2 - y1 + 1e-7f;
    const int rest = m - 1;
    int base = 0;
    float best_s = -INFINITY;
    int best_p = 0x7fffffff;
    for (int t0 = 0; t0 < rest; t0 += 1024) {
      const int t = t0 + tid;
      int alive = 0, idx = 0;
      float sc = 0.f;
      if (t < rest) {
        idx = cur[1 + t];
        const float oj = C_ ? C_[idx] * a.class_offset : 0.f;
        const float X1 = B_[idx * 4 + 0] + oj, Y1 = B_[idx * 4 + 1] + oj, X2 = B_[idx * 4 + 2] + oj, Y2 = B_[idx * 4 + 3] + oj;
        const float w2 = X2 - X1, h2 = Y2 - Y1 + 1e-7f;
        float iw = fminf(x2, X2) - fmaxf(x1, X1), ih = fminf(y2, Y2) - fmaxf(y1, Y1);
        iw = iw > 0.f ? iw : 0.f;
        ih = ih > 0.f ? ih : 0.f;
        const float inter = iw * ih;
        const float uni = w1 * h1 + w2 * h2 - inter + 1e-7f;
        const float iou = inter / uni;
        sc = S[idx];
        if (rest > 1 && iou > a.iou_thr) {  // quirk: a single rival (0-d IoU in the reference) is never decayed
          const float arg = -(iou * iou) / a.sigma;
          sc = sc * (float)exp((double)arg);
          S[idx] = sc;
        }
        alive = sc > a.score_thr;
      }
      int total;
      const int pos = base + block_excl_scan(alive, sh, total);
      if (alive) {
        nxt[pos] = idx;
        if (sc > best_s || (sc == best_s && pos < best_p)) {
          best_s = sc;
          best_p = pos;
        }
      }
      base += total;
    }
    // arg-max over survivors (first maximum in the new order), then swap it to the front
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float s2 = __shfl_xor(best_s, o, 64);
      const int p2 = __shfl_xor(best_p, o, 64);
      if (s2 > best_s || (s2 == best_s && p2 < best_p)) {
        best_s = s2;
        best_p = p2;
      }
    }
    if (lane == 0) {
      r_s[w] = best_s;
      r_p[w] = best_p;
    }
    __syncthreads();
    if (tid == 0) {
      float bs = r_s[0];
      int bp = r_p[0];
      for (int k = 1; k < 16; ++k)
        if (r_s[k] > bs || (r_s[k] == bs && r_p[k] < bp)) {
          bs = r_s[k];
          bp = r_p[k];
        }
      if (base > 0 && bp != 0) {
        const int t0 = nxt[0];
        nxt[0] = nxt[bp];
        nxt[bp] = t0;
      }
      s_m = base;
    }
    __syncthreads();
    m = s_m;
    int* tmp = cur;
    cur = nxt;
    nxt = tmp;
    if (m == 0) break;
    __syncthreads();
  }
  if (tid == 0) a.nkeep[b] = nk;
}
extern "C" int dy_soft_nms(const float* boxes, float* scores, const float* cls, const int* count, int* order_a, int* order_b,
                           int* keep, int* nkeep, int B, int cap, float iou_thr, float sigma, float score_thr,
                           float class_offset, hipStream_t stream) {
  SnArgs a{boxes, scores, cls, count, order_a, order_b, keep, nkeep, cap, iou_thr, sigma, score_thr, class_offset};
  hipLaunchKernelGGL(soft_nms_kernel, dim3(B), dim3(1024), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
