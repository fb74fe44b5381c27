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
// A workgroup turns 64 consecutive anchors into their (4+nc) output rows.  Inputs are anchor-major (NHWC logits), the output
// is channel-major (B, 4+nc, A), so everything goes through an LDS tile: DFL with 16 lanes per anchor (coalesced 16-byte
// loads, two shuffles per softmax), class logits as a coalesced float4 stream, and the write phase with lane = anchor so
// that every output row receives 256-byte runs.  (The first version gave each thread one anchor: every lane walked its own
// 256-byte + ncp*4-byte row with scalar loads -- 4.0 ms for 1280^2 x 32, a quarter of the get_FPS.py iteration.)
#define DEC_TILE 64
__global__ __launch_bounds__(256) void decode_pred_kernel(DecArgs d) {
  extern __shared__ float tile[];  // [4 + ncp][DEC_TILE + 1]
  const int TS = DEC_TILE + 1;
  const long total = (long)d.B * d.A;
  const int tid = threadIdx.x;
  for (long base = (long)blockIdx.x * DEC_TILE; base < total; base += (long)gridDim.x * DEC_TILE) {
    auto locate = [&](long ba, int& b, int& a, int& l, int& iy, int& ix) {
      b = (int)(ba / d.A);
      a = (int)(ba - (long)b * d.A);
      l = 0;
      for (int k = 1; k < d.nl; ++k)
        if (a >= d.a0[k]) l = k;
      const int r = a - d.a0[l];
      iy = r / d.W[l];
      ix = r - iy * d.W[l];
    };
    // ---- boxes: 16 lanes per anchor, 16 anchors per pass
    const int sub = tid & 15, side = sub >> 2, quarter = sub & 3;
    for (int j = tid >> 4; j < DEC_TILE; j += 16) {
      const long ba = base + j;
      float e = 0.f, anc = 0.f, st = 1.f;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ba < total) {
        int b, a, l, iy, ix;
        locate(ba, b, a, l, iy, ix);
        v = *reinterpret_cast<const float4*>(d.box[l] + (((size_t)b * d.H[l] + iy) * d.W[l] + ix) * 64 + sub * 4);
        anc = (side & 1) ? iy + 0.5f : ix + 0.5f;
        st = d.stride[l];
      }
      float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
      m = fmaxf(m, __shfl_xor(m, 1, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64));
      const float e0 = expf(v.x - m), e1 = expf(v.y - m), e2 = expf(v.z - m), e3 = expf(v.w - m);
      float den = (e0 + e1) + (e2 + e3);
      const float k0 = (float)(quarter * 4);
      float num = e0 * k0 + e1 * (k0 + 1.f) + e2 * (k0 + 2.f) + e3 * (k0 + 3.f);
      den += __shfl_xor(den, 1, 64);
      num += __shfl_xor(num, 1, 64);
      den += __shfl_xor(den, 2, 64);
      num += __shfl_xor(num, 2, 64);
      e = num / den;
      const float lo = anc - e, hi = anc + e;                 // side 0/1: x1,y1 = anchor - e; side 2/3: x2,y2 = anchor + e
      const float mine = side < 2 ? lo : hi;
      const float other = __shfl_xor(mine, 8, 64);            // lane with side ^ 2 (same quarter)
      if (quarter == 0 && side < 2) {
        tile[side * TS + j] = (mine + other) * 0.5f * st;     // centre x / y
        tile[(2 + side) * TS + j] = (other - mine) * st;      // width / height
      }
    }
    // ---- classes: coalesced float4 stream over the tile's contiguous logits, sigmoid, transposed into LDS
    const int q4 = d.ncp >> 2;
    for (int i = tid; i < DEC_TILE * q4; i += 256) {
      const int j = i / q4, part = i - j * q4;
      const long ba = base + j;
      if (ba < total) {
        int b, a, l, iy, ix;
        locate(ba, b, a, l, iy, ix);
        const float4 v = *reinterpret_cast<const float4*>(d.cls[l] + (((size_t)b * d.H[l] + iy) * d.W[l] + ix) * d.ncp + part * 4);
        const float xs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[(4 + part * 4 + k) * TS + j] = 1.f / (1.f + expf(-xs[k]));
      }
    }
    __syncthreads();
    // ---- write: lane = anchor, 4 output rows per pass
    const int j = tid & (DEC_TILE - 1);
    const long ba = base + j;
    if (ba < total) {
      const int b = (int)(ba / d.A), a = (int)(ba - (long)b * d.A);
      float* yo = d.y + (size_t)b * (4 + d.nc) * d.A + a;
      for (int ch = tid >> 6; ch < 4 + d.nc; ch += 4) yo[(size_t)ch * d.A] = tile[ch * TS + j];
    }
    __syncthreads();
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
  if ((ncp & 7) || nc > ncp) return DY_ERR_ARG;
  long blocks = ((long)B * a0 + DEC_TILE - 1) / DEC_TILE;
  if (blocks > 8192) blocks = 8192;
  const size_t lds = (size_t)(4 + ncp) * (DEC_TILE + 1) * sizeof(float);
  if (lds > 64 * 1024) return DY_ERR_ARG;
  hipLaunchKernelGGL(decode_pred_kernel, dim3((int)blocks), dim3(256), lds, stream, d);
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
