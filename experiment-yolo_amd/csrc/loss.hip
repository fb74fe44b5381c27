// Detection loss on device: target packing, DFL decode, task-aligned assignment, BCE + CIoU|WIoU-v3 (+NWD) + DFL
// losses with ANALYTIC gradients w.r.t. the head logits -- no autograd graph, no (B, n_max, A) temporaries, no host
// synchronisation.  Replaces v8DetectionLoss.__call__ (reference utils/loss.py:356-457), TaskAlignedAssigner.forward
// (utils/tal.py:39-258), BboxLoss.forward (utils/loss.py:202-250), bbox_iou / wasserstein_loss / WiseIouLoss
// (utils/metrics.py:75-126, 540-565, 591-645) and their backward.
//
// Layout: per level l the head writes box logits fp32 (B,H,W,64) and class logits fp32 (B,H,W,ncp); anchors are
// numbered level-major, row-major (utils/tal.py:294-307).  Box work maps ONE anchor to ONE wave: lane = side*16+bin,
// so the 256-byte logit row is a single coalesced load and the 16-bin softmax is four DPP shuffles.
#include "common.h"
#include "dealyolo_hip.h"

#define REG_MAX 16
#define TOPK 10
#define TAL_EPS 1e-9f
#define IOU_EPS 1e-7f

struct Level {
  const float* box;  // (B,H,W,64)
  const float* cls;  // (B,H,W,ncp)
  f16* dbox;         // (B,H,W,64)
  f16* dcls;         // (B,H,W,ncp)
  int H, W, a0;
  float stride;
  // box == NULL inside a training step (dy_head_box_decode wrote pred_box from the layer input): the DFL logits of a foreground
  // anchor are recomputed from the final box convolution's input, fp32 master weight (rounded to fp16 as the packed form) and bias
  const f16* xin = nullptr;
  int ldin = 0;
  const float* w = nullptr;
  const float* bias = nullptr;
  const float* incoef = nullptr;  // non-null: xin is the RAW output of the Conv below ([4][64] scale, shift, ..): applied here
};
struct LossCtx {
  Level lv[4];
  int nl, B, A, nc, ncp, nmax;
  // targets
  float* gt_box;   // (B,nmax,4) xyxy pixels
  int* gt_cls;     // (B,nmax)
  int* gt_valid;   // (B,nmax)
  // per-anchor state
  float* pred_box;   // (B,A,4) xyxy grid units
  int* cnt;          // (B,A)
  int* owner;        // (B,A)
  int* asg_gt;       // (B,A)  -1 = background
  float* asg_metric; // (B,A)
  float* asg_ov;     // (B,A)
  float* tscore;     // (B,A) target score at the assigned class (0 for background)
  // per-gt state
  int* topk_idx;       // (B,nmax,TOPK) anchor index or -1
  unsigned* pos_align; // (B,nmax) float bits
  unsigned* pos_ov;    // (B,nmax)
  // scalars: [0]=tss_raw [1]=tss=max(sum,1) [2]=liou_sum [3]=fg_count [4]=iou_mean (persistent) [5..7]=loss box,cls,dfl
  //          [8]=total loss*B  [9]=err flags
  float* scal;
  float* partials;  // scratch for deterministic block sums
  float hyp_box, hyp_cls, hyp_dfl;
  int use_wiou, use_nwd;
  float iou_ratio;
  const float* gscale;  // device scalar multiplied into every gradient (loss scale)
  int prob_scores;      // lv[].cls holds sigmoid probabilities (dy_tal_assign: TaskAlignedAssigner.forward's pd_scores), not logits
};

static __device__ __forceinline__ void anchor_of(const LossCtx& c, int a, int& l, int& iy, int& ix) {
  l = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < c.nl && a >= c.lv[k].a0) l = k;
  const int r = a - c.lv[l].a0;
  iy = r / c.lv[l].W;
  ix = r - iy * c.lv[l].W;
}

// ---------------------------------------------------------------------------------------------- scalar/dual maths
struct Dual {  // forward-mode value + d/d(x1,y1,x2,y2) of the predicted box
  float v, d[4];
};
static __device__ __forceinline__ Dual dconst(float v) { return Dual{v, {0.f, 0.f, 0.f, 0.f}}; }
static __device__ __forceinline__ Dual dvar(float v, int i) {
  Dual r = dconst(v);
  r.d[i] = 1.f;
  return r;
}
static __device__ __forceinline__ Dual operator+(Dual a, Dual b) {
  Dual r{a.v + b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
static __device__ __forceinline__ Dual operator-(Dual a, Dual b) {
  Dual r{a.v - b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
static __device__ __forceinline__ Dual operator*(Dual a, Dual b) {
  Dual r{a.v * b.v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
// The derivative components take ONE reciprocal of b instead of four correctly rounded divisions by it (<= 1 ulp apart per
// component).  Not for speed: with the four divisions, box_loss_kernel -- from inputs that were the reference's bits -- returned a
// different LAST component for one anchor in 4-7 % of its launches while a second process kept the GPU busy (scratch/loss_stress.py:
// 503 + 799 of 24,000; WIoU only, whose path holds the most divisions in flight), never with the GPU to itself; with the reciprocal:
// 0 of 24,000 beside the same co-tenant.  The mechanism is NOT understood -- it is not the s_mov vcc / v_div_fmas adjacency hipcc
// emits when it interleaves division sequences (removing every such site with scheduling barriers left the rate unchanged) -- see
// DESIGN 9.
static __device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const float q = a.v / b.v, inv = 1.0f / b.v;
  Dual r{q, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - q * b.d[i]) * inv;
  return r;
}
static __device__ __forceinline__ Dual operator+(Dual a, float b) { a.v += b; return a; }
static __device__ __forceinline__ Dual operator*(Dual a, float b) {
  a.v *= b;
  for (int i = 0; i < 4; ++i) a.d[i] *= b;
  return a;
}
static __device__ __forceinline__ Dual dmin(Dual a, Dual b) { return a.v <= b.v ? a : b; }  // torch: grad to first on ties? see note
static __device__ __forceinline__ Dual dmax(Dual a, Dual b) { return a.v >= b.v ? a : b; }
static __device__ __forceinline__ Dual drelu(Dual a) { return a.v > 0.f ? a : dconst(0.f); }
static __device__ __forceinline__ Dual dfn(Dual a, float v, float dv) {  // f(a) with value v and derivative dv
  Dual r{v, {}};
  for (int i = 0; i < 4; ++i) r.d[i] = dv * a.d[i];
  return r;
}
static __device__ __forceinline__ Dual detach(Dual a) { return dconst(a.v); }
static __device__ __forceinline__ float val(float a) { return a; }
static __device__ __forceinline__ float val(Dual a) { return a.v; }
static __device__ __forceinline__ float dmin(float a, float b) { return fminf(a, b); }
static __device__ __forceinline__ float dmax(float a, float b) { return fmaxf(a, b); }
static __device__ __forceinline__ float drelu(float a) { return a > 0.f ? a : 0.f; }
static __device__ __forceinline__ float detach(float a) { return a; }
static __device__ __forceinline__ float t_atan(float a) { return atanf(a); }
static __device__ __forceinline__ Dual t_atan(Dual a) { return dfn(a, atanf(a.v), 1.f / (1.f + a.v * a.v)); }
static __device__ __forceinline__ float t_exp(float a) { return expf(a); }
static __device__ __forceinline__ Dual t_exp(Dual a) { const float e = expf(a.v); return dfn(a, e, e); }
static __device__ __forceinline__ float t_sqrt(float a) { return sqrtf(a); }
static __device__ __forceinline__ Dual t_sqrt(Dual a) { const float s = sqrtf(a.v); return dfn(a, s, 0.5f / s); }
static __device__ __forceinline__ float lift(float, float v) { return v; }
static __device__ __forceinline__ Dual lift(Dual, float v) { return dconst(v); }

// bbox_iou(box1, box2, xywh=False, CIoU=True), utils/metrics.py:75-126.  T = float (assigner) or Dual (loss).
template <typename T>
static __device__ __forceinline__ T ciou_t(T x1, T y1, T x2, T y2, float X1, float Y1, float X2, float Y2) {
  const T w1 = x2 - x1, h1 = (y2 - y1) + IOU_EPS;
  const float w2 = X2 - X1, h2 = Y2 - Y1 + IOU_EPS;
  const T iw = drelu(dmin(x2, lift(x1, X2)) - dmax(x1, lift(x1, X1)));
  const T ih = drelu(dmin(y2, lift(x1, Y2)) - dmax(y1, lift(x1, Y1)));
  const T inter = iw * ih;
  const T uni = (w1 * h1 + (w2 * h2)) - inter + IOU_EPS;
  const T iou = inter / uni;
  const T cw = dmax(x2, lift(x1, X2)) - dmin(x1, lift(x1, X1));
  const T ch = dmax(y2, lift(x1, Y2)) - dmin(y1, lift(x1, Y1));
  const T c2 = cw * cw + ch * ch + IOU_EPS;
  const T sx = lift(x1, X1 + X2) - x1 - x2, sy = lift(x1, Y1 + Y2) - y1 - y2;
  const T rho2 = (sx * sx + sy * sy) * 0.25f;
  const T da = lift(x1, atanf(w2 / h2)) - t_atan(w1 / h1);
  const T v = (da * da) * (4.f / (3.14159265358979323846f * 3.14159265358979323846f));
  const float alpha = val(v) / (val(v) - val(iou) + (1.f + IOU_EPS));  // no_grad in the reference
  return iou - (rho2 / c2 + v * alpha);
}
// assigner orientation: box1 = gt (constants), box2 = prediction: same formula with the roles of (w1,h1)/(w2,h2)
// swapped inside v; v is symmetric, eps placement is identical, so one routine serves both.
static __device__ __forceinline__ float ciou_gt_pred(const float* g, const float* p) {
  return ciou_t<float>(g[0], g[1], g[2], g[3], p[0], p[1], p[2], p[3]);
}

// ---------------------------------------------------------------------------------------------- 1. targets
// v8DetectionLoss.preprocess (utils/loss.py:330-345) without the host loop / counts.max() sync.
__global__ void pack_targets_kernel(LossCtx c, const float* bidx, const float* cls, const float* boxes, int n,
                                    const int* n_dev, float img_w, float img_h) {
  if (n_dev) n = *n_dev;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < c.B * c.nmax; i += gridDim.x * blockDim.x) {
    c.gt_valid[i] = 0;
    c.gt_cls[i] = 0;
    c.pos_align[i] = 0u;
    c.pos_ov[i] = 0u;
    for (int k = 0; k < 4; ++k) c.gt_box[i * 4 + k] = 0.f;
  }
  if (threadIdx.x == 0) c.scal[9] = 0.f;
  // the slot of target i = number of earlier targets of the same image: batch indices staged in LDS for that O(n^2/threads)
  // scan (it read global memory n/2 times per thread before: 31 us for 512 targets)
  __shared__ short sb[8192];
  const int ncache = n < 8192 ? n : 8192;
  for (int i = threadIdx.x; i < ncache; i += blockDim.x) sb[i] = (short)(int)bidx[i];
  __syncthreads();  // launched with ONE block: zeroing above is complete before slots are filled
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int b = (int)bidx[i];
    if (b < 0 || b >= c.B) continue;
    int slot = 0;
    for (int j = 0; j < i; ++j) slot += ((j < ncache ? (int)sb[j] : (int)bidx[j]) == b);
    if (slot >= c.nmax) {
      c.scal[9] = 1.f;   // capacity overflow in this call
      c.scal[10] = 1.f;  // sticky: cleared only by the host, which raises on it wherever it already synchronises
      continue;
    }
    const float x = boxes[i * 4 + 0] * img_w, y = boxes[i * 4 + 1] * img_h;
    const float w = boxes[i * 4 + 2] * img_w, h = boxes[i * 4 + 3] * img_h;
    const float dw = w / 2, dh = h / 2;
    float* g = c.gt_box + ((size_t)b * c.nmax + slot) * 4;
    g[0] = x - dw; g[1] = y - dh; g[2] = x + dw; g[3] = y + dh;
    c.gt_cls[b * c.nmax + slot] = (int)cls[i];
    c.gt_valid[b * c.nmax + slot] = (g[0] + g[1] + g[2] + g[3]) > 0.f ? 1 : 0;  // mask_gt, utils/loss.py:382
  }
}

// ---------------------------------------------------------------------------------------------- 2. decode
// bbox_decode (utils/loss.py:347-354): one anchor per wave iteration, lane = side*16 + bin.
static __device__ __forceinline__ float softmax16_expect(float logit, int bin, float& prob) {
  float m = logit;
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  m = fmaxf(m, __shfl_xor(m, 4, 64));
  m = fmaxf(m, __shfl_xor(m, 8, 64));
  const float e = expf(logit - m);
  const float s = quad16_sum(e);
  prob = e / s;
  return quad16_sum(prob * (float)bin);
}

// 16 lanes per anchor (lane = side*4 + quarter, 4 consecutive bins each): one 16-byte load per lane, a wave covers four
// anchors per instruction and the 16-bin softmax needs two shuffles instead of four.
__global__ __launch_bounds__(256) void decode_kernel(LossCtx c) {
  // blockIdx.y = image.  No integer division in the loop: the logits of anchor a of level l lie at pixel b*H*W + (a - a0) of that
  // level's tensor, and the row is floor((r + 0.5) / W) in fp32 (r + 0.5 is at least 0.5 / W away from a multiple of W; the
  // product with the rounded reciprocal is off by < 1e-4 of a row for any map a 32-bit anchor index can address).  The first form
  // spent a 64-bit and a 32-bit division per 16-byte load and was VALU-bound at 3.1 TB/s.
  const int sub = threadIdx.x & 15, side = sub >> 2, quarter = sub & 3;
  const int b = blockIdx.y;
  float invw[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) invw[k] = k < c.nl ? 1.0f / (float)c.lv[k].W : 0.f;
  for (int a = (int)((blockIdx.x * 256 + threadIdx.x) >> 4); a < c.A; a += (int)((gridDim.x * 256) >> 4)) {
    int l = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < c.nl && a >= c.lv[k].a0) l = k;
    const Level& L = c.lv[l];
    const int r = a - L.a0;
    const int iy = (int)(((float)r + 0.5f) * invw[l]), ix = r - iy * L.W;
    const float4 v = *reinterpret_cast<const float4*>(L.box + ((size_t)b * (L.H * L.W) + r) * 64 + sub * 4);
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
    const float e = num / den;
    const float anc = (side & 1) ? (iy + 0.5f) : (ix + 0.5f);
    if (quarter == 0) c.pred_box[((size_t)b * c.A + a) * 4 + side] = side < 2 ? anc - e : anc + e;
  }
}

// ---------------------------------------------------------------------------------------------- 3. top-k per gt
// get_box_metrics + select_topk_candidates (utils/tal.py:102-161) restricted to the anchors inside the gt box
// (everything else has metric 0 and can only enter torch.topk's result as an implementation-ordered zero filler,
// which mask_in_gts removes again, :98).  Ties: larger metric first, then LOWER anchor index.  Zero-metric
// anchors are never selected (documented divergence: DESIGN.md "TAL zero-metric fillers").
#define TAL_CAP 2048
__global__ __launch_bounds__(256) void tal_topk_kernel(LossCtx c) {
  const int bj = blockIdx.x, b = bj / c.nmax, tid = threadIdx.x;
  __shared__ float s_m[TAL_CAP + TOPK];
  __shared__ int s_a[TAL_CAP + TOPK];
  __shared__ float r_m[256];
  __shared__ int r_i[256];
  __shared__ int s_keep;
  int* out = c.topk_idx + (size_t)bj * TOPK;
  if (tid < TOPK) out[tid] = -1;
  if (!c.gt_valid[bj]) return;
  const float* g = c.gt_box + (size_t)bj * 4;
  const float gx1 = g[0], gy1 = g[1], gx2 = g[2], gy2 = g[3];
  const int label = c.gt_cls[bj];
  // candidate rectangles per level
  int x0[4], y0[4], nx[4], ny[4], cum[5];
  cum[0] = 0;
  for (int l = 0; l < 4; ++l) {
    x0[l] = y0[l] = nx[l] = ny[l] = 0;
    if (l < c.nl) {
      const float s = c.lv[l].stride;
      int xa = (int)floorf(gx1 / s - 0.5f) - 1, xb = (int)ceilf(gx2 / s - 0.5f) + 1;
      int ya = (int)floorf(gy1 / s - 0.5f) - 1, yb = (int)ceilf(gy2 / s - 0.5f) + 1;
      xa = xa < 0 ? 0 : xa; ya = ya < 0 ? 0 : ya;
      xb = xb > c.lv[l].W - 1 ? c.lv[l].W - 1 : xb;
      yb = yb > c.lv[l].H - 1 ? c.lv[l].H - 1 : yb;
      x0[l] = xa; y0[l] = ya;
      nx[l] = xb >= xa ? xb - xa + 1 : 0;
      ny[l] = yb >= ya ? yb - ya + 1 : 0;
    }
    cum[l + 1] = cum[l] + nx[l] * ny[l];
  }
  const int ncand = cum[4];
  if (tid == 0) s_keep = 0;
  __syncthreads();
  for (int base = 0; base < ncand; base += TAL_CAP) {
    const int keep = s_keep;  // running winners occupy [0, keep)
    const int chunk = (ncand - base) < TAL_CAP ? (ncand - base) : TAL_CAP;
    __syncthreads();
    for (int t = tid; t < chunk; t += 256) {
      const int k = base + t;
      int l = 0;
      for (int q = 1; q < 4; ++q)
        if (k >= cum[q]) l = q;
      const int r = k - cum[l];
      const int iy = y0[l] + r / nx[l], ix = x0[l] + r % nx[l];
      const float s = c.lv[l].stride;
      const float ax = (ix + 0.5f) * s, ay = (iy + 0.5f) * s;
      const float dmin_ = fminf(fminf(ax - gx1, ay - gy1), fminf(gx2 - ax, gy2 - ay));
      float metric = 0.f;
      const int a = c.lv[l].a0 + iy * c.lv[l].W + ix;
      if (dmin_ > TAL_EPS) {  // select_candidates_in_gts, :213-229
        const float* pb = c.pred_box + ((size_t)b * c.A + a) * 4;
        const float pp[4] = {pb[0] * s, pb[1] * s, pb[2] * s, pb[3] * s};
        float ov = ciou_gt_pred(g, pp);
        ov = ov > 0.f ? ov : 0.f;  // clamp_(0), :125
        const float logit = c.lv[l].cls[(((size_t)b * c.lv[l].H + iy) * c.lv[l].W + ix) * c.ncp + label];
        const float sc = c.prob_scores ? logit : 1.f / (1.f + expf(-logit));
        const float o3 = ov * ov * ov;
        metric = sqrtf(sc) * (o3 * o3);  // alpha 0.5, beta 6
      }
      s_m[keep + t] = metric;
      s_a[keep + t] = a;
    }
    __syncthreads();
    const int n = keep + chunk;
    // TOPK rounds of block arg-max (metric desc, anchor asc); winners are swapped to the front
    for (int round = 0; round < TOPK; ++round) {
      float bm = 0.f;
      int bi = -1;
      for (int t = round + tid; t < n; t += 256) {
        const float m = s_m[t];
        if (m > bm || (m == bm && m > 0.f && bi >= 0 && s_a[t] < s_a[bi])) {
          bm = m;
          bi = t;
        }
      }
      r_m[tid] = bm;
      r_i[tid] = bi;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
          const float m2 = r_m[tid + o];
          const int i2 = r_i[tid + o];
          const float m1 = r_m[tid];
          const int i1 = r_i[tid];
          if (i2 >= 0 && (i1 < 0 || m2 > m1 || (m2 == m1 && s_a[i2] < s_a[i1]))) {
            r_m[tid] = m2;
            r_i[tid] = i2;
          }
        }
        __syncthreads();
      }
      const int win = r_i[0];
      if (win < 0) {
        if (tid == 0) s_keep = round;
        __syncthreads();
        break;
      }
      if (tid == 0) {
        const float tm = s_m[round];
        const int ta = s_a[round];
        s_m[round] = s_m[win];
        s_a[round] = s_a[win];
        s_m[win] = tm;
        s_a[win] = ta;
        s_keep = round + 1;
      }
      __syncthreads();
    }
    __syncthreads();
  }
  const int keep = s_keep;
  if (tid < keep) out[tid] = s_a[tid];
}

// ---------------------------------------------------------------------------------------------- 4. resolve
__global__ void tal_scatter_kernel(LossCtx c) {  // count how many gts picked each anchor
  const int total = c.B * c.nmax * TOPK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int a = c.topk_idx[i];
    if (a < 0) continue;
    const int bj = i / TOPK, b = bj / c.nmax, j = bj - b * c.nmax;
    atomicAdd(&c.cnt[(size_t)b * c.A + a], 1);
    c.owner[(size_t)b * c.A + a] = j;
  }
}

// select_highest_overlaps (utils/tal.py:232-258) + the per-gt maxima used by the normalisation (:80-86)
__global__ __launch_bounds__(256) void tal_resolve_kernel(LossCtx c) {
  const long total = (long)c.B * c.A;
  for (long ba = (long)blockIdx.x * 256 + threadIdx.x; ba < total; ba += (long)gridDim.x * 256) {
    const int n = c.cnt[ba];
    int j = -1;
    float ov = 0.f, metric = 0.f;
    if (n > 0) {
      const int b = (int)(ba / c.A), a = (int)(ba - (long)b * c.A);
      int l, iy, ix;
      anchor_of(c, a, l, iy, ix);
      const float s = c.lv[l].stride;
      const float ax = (ix + 0.5f) * s, ay = (iy + 0.5f) * s;
      const float* pb = c.pred_box + ba * 4;
      const float pp[4] = {pb[0] * s, pb[1] * s, pb[2] * s, pb[3] * s};
      if (n == 1) {
        j = c.owner[ba];
        ov = ciou_gt_pred(c.gt_box + ((size_t)b * c.nmax + j) * 4, pp);
        ov = ov > 0.f ? ov : 0.f;
      } else {  // argmax over ALL gts of the image of overlaps (zero outside their boxes); first maximum wins
        float best = -1.f;
        for (int k = 0; k < c.nmax; ++k) {
          float o = 0.f;
          if (c.gt_valid[b * c.nmax + k]) {
            const float* g = c.gt_box + ((size_t)b * c.nmax + k) * 4;
            const float dm = fminf(fminf(ax - g[0], ay - g[1]), fminf(g[2] - ax, g[3] - ay));
            if (dm > TAL_EPS) {
              o = ciou_gt_pred(g, pp);
              o = o > 0.f ? o : 0.f;
            }
          }
          if (o > best) {
            best = o;
            j = k;
          }
        }
        ov = best;
      }
      const int label = c.gt_cls[b * c.nmax + j];
      const float logit = c.lv[l].cls[(((size_t)b * c.lv[l].H + iy) * c.lv[l].W + ix) * c.ncp + label];
      const float sc = c.prob_scores ? logit : 1.f / (1.f + expf(-logit));
      const float o3 = ov * ov * ov;
      metric = sqrtf(sc) * (o3 * o3);
      atomicMax(&c.pos_align[b * c.nmax + j], __float_as_uint(metric));
      atomicMax(&c.pos_ov[b * c.nmax + j], __float_as_uint(ov));
    }
    c.asg_gt[ba] = j;
    c.asg_metric[ba] = metric;
    c.asg_ov[ba] = ov;
  }
}

// ---------------------------------------------------------------------------------------------- 5. target scores
// WIoU's L_IoU (utils/metrics.py:596-622), no eps, boxes in grid units
static __device__ __forceinline__ float liou_plain(const float* p, const float* t) {
  const float iw = fmaxf(fminf(p[2], t[2]) - fmaxf(p[0], t[0]), 0.f), ih = fmaxf(fminf(p[3], t[3]) - fmaxf(p[1], t[1]), 0.f);
  const float inter = iw * ih;
  const float uni = (p[2] - p[0]) * (p[3] - p[1]) + (t[2] - t[0]) * (t[3] - t[1]) - inter;
  return 1.f - inter / uni;
}

__global__ __launch_bounds__(256) void tal_scores_kernel(LossCtx c) {
  const long total = (long)c.B * c.A;
  float s_ts = 0.f, s_li = 0.f, s_n = 0.f;
  for (long ba = (long)blockIdx.x * 256 + threadIdx.x; ba < total; ba += (long)gridDim.x * 256) {
    const int j = c.asg_gt[ba];
    float ts = 0.f;
    if (j >= 0) {
      const int b = (int)(ba / c.A), a = (int)(ba - (long)b * c.A);
      const float pa = __uint_as_float(c.pos_align[b * c.nmax + j]), po = __uint_as_float(c.pos_ov[b * c.nmax + j]);
      ts = c.asg_metric[ba] * po / (pa + TAL_EPS);
      s_ts += ts;
      if (c.use_wiou) {
        int l, iy, ix;
        anchor_of(c, a, l, iy, ix);
        const float s = c.lv[l].stride;
        const float* g = c.gt_box + ((size_t)b * c.nmax + j) * 4;
        const float t[4] = {g[0] / s, g[1] / s, g[2] / s, g[3] / s};
        s_li += liou_plain(c.pred_box + ba * 4, t);
      }
      s_n += 1.f;
    }
    c.tscore[ba] = ts;
  }
  __shared__ float red[3][256];
  red[0][threadIdx.x] = s_ts;
  red[1][threadIdx.x] = s_li;
  red[2][threadIdx.x] = s_n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) c.partials[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void loss_scalars_kernel(LossCtx c, int nparts) {
  __shared__ double red[3][256];
  double s[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < nparts; i += 256)
    for (int k = 0; k < 3; ++k) s[k] += c.partials[i * 3 + k];
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float tss = (float)red[0][0];
    c.scal[0] = tss;
    c.scal[1] = tss > 1.f ? tss : 1.f;  // max(target_scores.sum(), 1), utils/loss.py:399
    c.scal[2] = (float)red[1][0];
    c.scal[3] = (float)red[2][0];
    if (c.use_wiou && red[2][0] > 0.0)  // iou_mean update (training is always True on the criterion)
      c.scal[4] = c.scal[4] * (1.f - 1e-2f) + 1e-2f * (float)(red[1][0] / red[2][0]);
    c.scal[5] = c.scal[6] = c.scal[7] = c.scal[8] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------- 6. class loss
// BCEWithLogits over (B,A,nc) + gradient: d/dx = (sigmoid(x) - t) * hyp.cls * B / tss * gscale
__global__ __launch_bounds__(256) void cls_loss_kernel(LossCtx c, int level) {
  // one thread per anchor: its ncp (multiple of 8) class logits are two/more 16-byte loads, the assignment is looked up
  // once, and the gradient leaves as 16-byte fp16 stores.  (Was one thread per (anchor, class) with 64-bit divisions and
  // libm log1pf/expf per element: 107 us at P2; no discrete decision depends on these values, so fast intrinsics are safe.)
  const Level& L = c.lv[level];
  const int hw = L.H * L.W;
  const int npix = c.B * hw;
  const float k = c.hyp_cls * (float)c.B / c.scal[1] * c.gscale[0];
  float acc = 0.f;
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < npix; pix += gridDim.x * 256) {
    const int b = pix / hw;
    const long ba = (long)b * c.A + L.a0 + (pix - b * hw);
    const int j = c.asg_gt[ba];
    const int tc = j >= 0 ? c.gt_cls[b * c.nmax + j] : -1;
    const float ts = j >= 0 ? c.tscore[ba] : 0.f;
    const float* xp = L.cls + (size_t)pix * c.ncp;
    for (int c0 = 0; c0 < c.ncp; c0 += 8) {
      const float4 v0 = *reinterpret_cast<const float4*>(xp + c0), v1 = *reinterpret_cast<const float4*>(xp + c0 + 4);
      const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      half8 g8;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int cc = c0 + q;
        float g = 0.f;
        if (cc < c.nc) {
          const float x = xs[q], t = cc == tc ? ts : 0.f;
          const float e = __expf(-fabsf(x));  // max(x,0) - x*t + log1p(exp(-|x|)); sigmoid(x) = x >= 0 ? 1/(1+e) : e/(1+e)
          acc += fmaxf(x, 0.f) - x * t + __logf(1.f + e);
          const float r = __builtin_amdgcn_rcpf(1.f + e);
          g = ((x >= 0.f ? r : e * r) - t) * k;
        }
        g8[q] = (f16)g;
      }
      if (L.dcls) *reinterpret_cast<half8*>(L.dcls + (size_t)pix * c.ncp + c0) = g8;
    }
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) c.partials[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------------- 7. box + DFL loss
__global__ __launch_bounds__(256) void box_loss_kernel(LossCtx c) {
  const int lane = threadIdx.x & 63, side = lane >> 4, bin = lane & 15;
  const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((long)gridDim.x * 256) >> 6;
  const long total = (long)c.B * c.A;
  const float tss = c.scal[1], gs = c.gscale[0], Bf = (float)c.B;
  float l_box = 0.f, l_dfl = 0.f;
  for (long base = wave * 64; base < total; base += nw * 64) {
   // one coalesced load tells the wave which of its 64 anchors are foreground (~1 %); only those are visited
   const int jl = (base + lane < total) ? c.asg_gt[base + lane] : -1;
   unsigned long long fgm = __ballot(jl >= 0);
   while (fgm) {
    const int bit = __ffsll((long long)fgm) - 1;
    fgm &= fgm - 1;
    const long ba = base + bit;
    const int j = __shfl(jl, bit, 64);
    const int b = (int)(ba / c.A), a = (int)(ba - (long)b * c.A);
    int l, iy, ix;
    anchor_of(c, a, l, iy, ix);
    const Level& L = c.lv[l];
    const size_t off = (((size_t)b * L.H + iy) * L.W + ix) * 64 + lane;
    float logit;
    if (L.box) {
      logit = L.box[off];
    } else {
      // lane = output channel (side * 16 + bin) AND input channel: every lane brings one input value (BatchNorm + SiLU applied to it
      // when the layer input is the RAW output of the Conv in front -- the pair form of the apply kernel, so the same bits), the 64
      // values go round by shuffle against the lane's row of the weight (rounded to fp16 as the packed form the conv multiplies with)
      const float xraw = (float)L.xin[(((size_t)b * L.H + iy) * L.W + ix) * L.ldin + lane];
      float yv = xraw;
      if (L.incoef) {
        const float z = __builtin_fmaf(xraw, L.incoef[lane], L.incoef[64 + lane]);
        const f32x2 a2 = act_fwd2_fast<DY_ACT_SILU>((f32x2){z, z});
        yv = (float)(f16)a2[0];
      }
      const float4* wr = reinterpret_cast<const float4*>(L.w + lane * 64);
      float sacc = 0.f;
#pragma unroll 4
      for (int k4 = 0; k4 < 16; ++k4) {
        const float4 w4 = wr[k4];
        sacc += (float)(f16)w4.x * __shfl(yv, k4 * 4, 64);
        sacc += (float)(f16)w4.y * __shfl(yv, k4 * 4 + 1, 64);
        sacc += (float)(f16)w4.z * __shfl(yv, k4 * 4 + 2, 64);
        sacc += (float)(f16)w4.w * __shfl(yv, k4 * 4 + 3, 64);
      }
      logit = sacc + L.bias[lane];
    }
    float pr;
    const float e = softmax16_expect(logit, bin, pr);
    const float w = c.tscore[ba];  // weight = target_scores.sum(-1)
    const float s = L.stride;
    const float* g = c.gt_box + ((size_t)b * c.nmax + j) * 4;
    const float t[4] = {g[0] / s, g[1] / s, g[2] / s, g[3] / s};
    const float* pb = c.pred_box + ba * 4;
    // ---- IoU-family loss with forward-mode derivatives w.r.t. (x1,y1,x2,y2)
    Dual x1 = dvar(pb[0], 0), y1 = dvar(pb[1], 1), x2 = dvar(pb[2], 2), y2 = dvar(pb[3], 3);
    Dual lb;
    if (c.use_wiou) {  // WiseIouLoss._WIoU + _scaled_loss (non-monotonous v3)
      const Dual pw = x2 - x1, ph = y2 - y1;
      const float tw = t[2] - t[0], th = t[3] - t[1];
      const Dual iw = drelu(dmin(x2, dconst(t[2])) - dmax(x1, dconst(t[0])));
      const Dual ih = drelu(dmin(y2, dconst(t[3])) - dmax(y1, dconst(t[1])));
      const Dual inter = iw * ih;
      const Dual uni = pw * ph + (tw * th) - inter;
      const Dual liou = dconst(1.f) - inter / uni;
      const Dual bw = dmax(x2, dconst(t[2])) - dmin(x1, dconst(t[0]));
      const Dual bh = dmax(y2, dconst(t[3])) - dmin(y1, dconst(t[1]));
      const float l2box = bw.v * bw.v + bh.v * bh.v;  // detached
      const Dual dcx = (x1 + x2) * 0.5f + (-(t[0] + t[2]) * 0.5f), dcy = (y1 + y2) * 0.5f + (-(t[1] + t[3]) * 0.5f);
      const Dual l2c = dcx * dcx + dcy * dcy;
      const Dual dist = t_exp(l2c * (1.f / l2box));
      const float beta = liou.v / c.scal[4];
      const float fac = beta / (2.7f * powf(1.7f, beta - 2.7f));
      lb = (dist * liou) * fac;
    } else {
      lb = dconst(1.f) - ciou_t<Dual>(x1, y1, x2, y2, t[0], t[1], t[2], t[3]);
    }
    if (c.use_nwd) {  // wasserstein_loss, utils/metrics.py:540-565
      const Dual w1 = x2 - x1, h1 = (y2 - y1) + IOU_EPS;
      const float w2 = t[2] - t[0], h2 = t[3] - t[1] + IOU_EPS;
      const Dual cx = (x1 + w1 * 0.5f) + (-(t[0] + w2 / 2)), cy = (y1 + h1 * 0.5f) + (-(t[1] + h2 / 2));
      const Dual cd = cx * cx + cy * cy + IOU_EPS;
      const Dual dw = w1 + (-w2), dh = h1 + (-h2);
      const Dual whd = (dw * dw + dh * dh) * 0.25f;
      const Dual nw_ = t_exp(t_sqrt(cd + whd) * (-1.f / 12.8f));
      const Dual ln = dconst(1.f) - nw_;
      lb = lb * c.iou_ratio + ln * (1.f - c.iou_ratio);
    }
    const float kb = w / tss;  // (loss * weight).sum() / target_scores_sum
    // d(loss_box)/d(pred coord) -> d/dE (E = expected distance of this side): x1 = ax - E0, y1 = ay - E1, x2 = ax + E2 ...
    const float dcoord = lb.d[side] * (side < 2 ? -1.f : 1.f);
    float glogit = c.hyp_box * Bf * kb * dcoord * pr * ((float)bin - e);
    // ---- DFL (utils/loss.py:236-250, bbox2dist utils/tal.py:321-324)
    const float anc = (side & 1) ? (iy + 0.5f) : (ix + 0.5f);
    float tl = side < 2 ? anc - t[side] : t[side] - anc;
    tl = fminf(fmaxf(tl, 0.f), (float)(REG_MAX - 1) - 0.01f);
    const int il = (int)tl;
    const float wl = (float)(il + 1) - tl, wr = 1.f - wl;
    const float logp = logf(pr);
    const float lp_l = __shfl(logp, (lane & 48) + il, 64), lp_r = __shfl(logp, (lane & 48) + il + 1, 64);
    const float dfl_side = -(lp_l * wl + lp_r * wr);  // CE(tl)*wl + CE(tr)*wr
    const float ind = (bin == il ? wl : 0.f) + (bin == il + 1 ? wr : 0.f);
    glogit += c.hyp_dfl * Bf * kb * 0.25f * (pr - ind);
    if (L.dbox) L.dbox[off] = (f16)(glogit * gs);
    if (lane == 0) l_box += lb.v * kb;
    if (bin == 0) l_dfl += dfl_side * 0.25f * kb;
   }
  }
  // lanes with bin==0 hold the four side terms of dfl; lane 0 holds box
  l_dfl = wave_sum(l_dfl);
  l_box = wave_sum(l_box);
  __shared__ float red[2][4];
  if (lane == 0) {
    red[0][threadIdx.x >> 6] = l_box;
    red[1][threadIdx.x >> 6] = l_dfl;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = threadIdx.x;
    c.partials[blockIdx.x * 2 + k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}

// final: loss_items and total
__global__ __launch_bounds__(256) void loss_final_kernel(LossCtx c, const float* box_parts, int n_box,
                                                         const float* cls_parts, int n_cls) {
  __shared__ double red[3][256];
  double s[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < n_box; i += 256) {
    s[0] += box_parts[i * 2 + 0];
    s[2] += box_parts[i * 2 + 1];
  }
  for (int i = threadIdx.x; i < n_cls; i += 256) s[1] += cls_parts[i];
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float box = (float)red[0][0] * c.hyp_box, cls = (float)(red[1][0] / c.scal[1]) * c.hyp_cls,
                dfl = (float)red[2][0] * c.hyp_dfl;
    c.scal[5] = box;
    c.scal[6] = cls;
    c.scal[7] = dfl;
    c.scal[8] = (box + cls + dfl) * (float)c.B;
  }
}

// ---------------------------------------------------------------------------------------------- host entry
struct DyLossDesc;  // public mirror declared in dealyolo_hip.h

extern "C" size_t dy_loss_workspace_bytes(int B, int A, int nmax) {
  size_t n = 0;
  auto al = [&](size_t b) { n += (b + 255) / 256 * 256; };
  al((size_t)B * nmax * 4 * 4);   // gt_box
  al((size_t)B * nmax * 4);       // gt_cls
  al((size_t)B * nmax * 4);       // gt_valid
  al((size_t)B * A * 16);         // pred_box
  al((size_t)B * A * 4);          // cnt
  al((size_t)B * A * 4);          // owner
  al((size_t)B * A * 4);          // asg_gt
  al((size_t)B * A * 4);          // asg_metric
  al((size_t)B * A * 4);          // asg_ov
  al((size_t)B * A * 4);          // tscore
  al((size_t)B * nmax * TOPK * 4);
  al((size_t)B * nmax * 4);
  al((size_t)B * nmax * 4);
  al(16 * 4);                     // scalars
  al(3 * 4096 * 4);               // partials A
  al(2 * 4096 * 4);               // partials B (box)
  al(4 * 4096 * 4);               // partials C (cls, per level)
  return n;
}

// byte offsets of the per-anchor assignment state inside the workspace (test / debugging aid)
extern "C" int dy_loss_workspace_layout(int B, int A, int nmax, size_t* off_pred_box, size_t* off_asg_gt,
                                        size_t* off_tscore) {
  size_t n = 0;
  auto al = [&](size_t b) { size_t r = n; n += (b + 255) / 256 * 256; return r; };
  const size_t BA = (size_t)B * A, BN = (size_t)B * nmax;
  al(BN * 16); al(BN * 4); al(BN * 4);
  *off_pred_box = al(BA * 16);
  al(BA * 4); al(BA * 4);
  *off_asg_gt = al(BA * 4);
  al(BA * 4); al(BA * 4);
  *off_tscore = al(BA * 4);
  return DY_OK;
}

// TaskAlignedAssigner.forward (reference utils/tal.py:39-88) on its own: the assignment kernels of dy_detection_loss fed with the
// reference's arguments -- class PROBABILITIES per level, decoded boxes, padded ground truth -- instead of head logits.
extern "C" int dy_tal_assign(const float* const* scores, const int* H, const int* W, const float* stride, int nl, int B, int nc,
                             int ncp, int n, const float* pd_boxes_grid, const int* gt_labels, const float* gt_bboxes,
                             const int* mask_gt, int* asg_gt, float* tscore, void* workspace, hipStream_t stream) {
  if (nl < 1 || nl > 4 || n < 1 || (ncp & 7) || nc > ncp || !workspace || !asg_gt || !tscore) return DY_ERR_ARG;
  LossCtx c{};
  c.nl = nl; c.B = B; c.nc = nc; c.ncp = ncp; c.nmax = n; c.prob_scores = 1;
  int a0 = 0;
  for (int l = 0; l < nl; ++l) {
    c.lv[l] = Level{nullptr, scores[l], nullptr, nullptr, H[l], W[l], a0, stride[l]};
    a0 += H[l] * W[l];
  }
  c.A = a0;
  char* p = (char*)workspace;
  auto take = [&](size_t b) { char* r = p; p += (b + 255) / 256 * 256; return r; };
  const size_t BA = (size_t)B * c.A, BN = (size_t)B * n;
  take(BN * 16); take(BN * 4); take(BN * 4); take(BA * 16);
  c.gt_box = const_cast<float*>(gt_bboxes); c.gt_cls = const_cast<int*>(gt_labels); c.gt_valid = const_cast<int*>(mask_gt);
  c.pred_box = const_cast<float*>(pd_boxes_grid);
  c.cnt = (int*)take(BA * 4); c.owner = (int*)take(BA * 4);
  take(BA * 4);
  c.asg_gt = asg_gt;
  c.asg_metric = (float*)take(BA * 4); c.asg_ov = (float*)take(BA * 4);
  take(BA * 4);
  c.tscore = tscore;
  c.topk_idx = (int*)take(BN * TOPK * 4); c.pos_align = (unsigned*)take(BN * 4); c.pos_ov = (unsigned*)take(BN * 4);
  take(16 * 4);
  c.partials = (float*)take(3 * 4096 * 4);
  if (hipMemsetAsync(c.cnt, 0, BA * 4, stream) != hipSuccess || hipMemsetAsync(c.pos_align, 0, BN * 4, stream) != hipSuccess ||
      hipMemsetAsync(c.pos_ov, 0, BN * 4, stream) != hipSuccess)
    return DY_ERR_LAUNCH;
  hipLaunchKernelGGL(tal_topk_kernel, dim3((int)BN), dim3(256), 0, stream, c);
  hipLaunchKernelGGL(tal_scatter_kernel, dim3(cdiv((int)BN * TOPK, 256)), dim3(256), 0, stream, c);
  const int gridE = (int)((BA + 255) / 256 < 2048 ? (BA + 255) / 256 : 2048);
  hipLaunchKernelGGL(tal_resolve_kernel, dim3(gridE), dim3(256), 0, stream, c);
  hipLaunchKernelGGL(tal_scores_kernel, dim3(gridE), dim3(256), 0, stream, c);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// sizeof(DyLossArgs) as this library was compiled: a binding that lays the block out differently must notice before its first call
extern "C" int dy_loss_args_bytes(void) { return (int)sizeof(DyLossArgs); }
extern "C" int dy_detection_loss(const DyLossArgs* d, hipStream_t stream) {
  if (d->nl < 1 || d->nl > 4 || d->nmax < 1 || (d->ncp & 7) || d->nc > d->ncp) return DY_ERR_ARG;
  LossCtx c{};
  c.nl = d->nl; c.B = d->B; c.nc = d->nc; c.ncp = d->ncp; c.nmax = d->nmax;
  int a0 = 0;
  for (int l = 0; l < d->nl; ++l) {
    c.lv[l] = Level{d->box[l], d->cls[l], (f16*)d->dbox[l], (f16*)d->dcls[l], d->H[l], d->W[l], a0, d->stride[l]};
    if (d->box_from_input) {
      if (!d->box_in[l] || !d->box_w[l] || !d->box_b[l] || (d->box_in_ld[l] & 7)) return DY_ERR_ARG;
      c.lv[l].box = nullptr;
      c.lv[l].xin = (const f16*)d->box_in[l]; c.lv[l].ldin = d->box_in_ld[l]; c.lv[l].w = d->box_w[l]; c.lv[l].bias = d->box_b[l];
      c.lv[l].incoef = d->box_in_coef[l];
    }
    a0 += d->H[l] * d->W[l];
  }
  c.A = a0;
  char* p = (char*)d->workspace;
  auto take = [&](size_t b) { char* r = p; p += (b + 255) / 256 * 256; return r; };
  const size_t BA = (size_t)c.B * c.A, BN = (size_t)c.B * c.nmax;
  c.gt_box = (float*)take(BN * 16); c.gt_cls = (int*)take(BN * 4); c.gt_valid = (int*)take(BN * 4);
  c.pred_box = (float*)take(BA * 16); c.cnt = (int*)take(BA * 4); c.owner = (int*)take(BA * 4);
  c.asg_gt = (int*)take(BA * 4); c.asg_metric = (float*)take(BA * 4); c.asg_ov = (float*)take(BA * 4);
  c.tscore = (float*)take(BA * 4);
  c.topk_idx = (int*)take(BN * TOPK * 4); c.pos_align = (unsigned*)take(BN * 4); c.pos_ov = (unsigned*)take(BN * 4);
  float* scal_ws = (float*)take(16 * 4);
  (void)scal_ws;
  c.scal = d->scalars;  // caller-owned persistent 16-float block (holds iou_mean across steps)
  float* partA = (float*)take(3 * 4096 * 4);
  float* partB = (float*)take(2 * 4096 * 4);
  float* partC = (float*)take(4 * 4096 * 4);
  c.hyp_box = d->hyp_box; c.hyp_cls = d->hyp_cls; c.hyp_dfl = d->hyp_dfl;
  c.use_wiou = d->use_wiou; c.use_nwd = d->use_nwd; c.iou_ratio = d->iou_ratio; c.gscale = d->gscale;

  hipLaunchKernelGGL(pack_targets_kernel, dim3(1), dim3(1024), 0, stream, c, d->t_batch_idx, d->t_cls, d->t_boxes,
                     d->n_targets, d->n_targets_dev, d->img_w, d->img_h);
  if (hipMemsetAsync(c.cnt, 0, BA * 4, stream) != hipSuccess) return DY_ERR_LAUNCH;
  const int gridA = (int)((BA * 64 + 255) / 256 < 2048 ? (BA * 64 + 255) / 256 : 2048);
  if (!d->box_from_input) {  // else dy_head_box_decode has written pred_box already (same workspace, same layout)
    const long nb = ((long)c.A * 16 + 255) / 256;  // per image: 16 lanes per anchor; blockIdx.y = image
    hipLaunchKernelGGL(decode_kernel, dim3((int)(nb < 256 ? nb : 256), c.B), dim3(256), 0, stream, c);
  }
  hipLaunchKernelGGL(tal_topk_kernel, dim3((int)BN), dim3(256), 0, stream, c);
  hipLaunchKernelGGL(tal_scatter_kernel, dim3(cdiv((int)BN * TOPK, 256)), dim3(256), 0, stream, c);
  const int gridE = (int)((BA + 255) / 256 < 2048 ? (BA + 255) / 256 : 2048);
  hipLaunchKernelGGL(tal_resolve_kernel, dim3(gridE), dim3(256), 0, stream, c);
  c.partials = partA;
  hipLaunchKernelGGL(tal_scores_kernel, dim3(gridE), dim3(256), 0, stream, c);
  hipLaunchKernelGGL(loss_scalars_kernel, dim3(1), dim3(256), 0, stream, c, gridE);
  int n_cls = 0;
  for (int l = 0; l < c.nl; ++l) {
    const long tot = (long)c.B * c.lv[l].H * c.lv[l].W;  // one thread per anchor
    int g = (int)((tot + 255) / 256 < 2048 ? (tot + 255) / 256 : 2048);
    c.partials = partC + n_cls;
    hipLaunchKernelGGL(cls_loss_kernel, dim3(g), dim3(256), 0, stream, c, l);
    n_cls += g;
  }
  c.partials = partB;
  for (int l = 0; l < c.nl; ++l)  // background anchors get zero box-gradient: one memset instead of 2 M scattered stores
    if (c.lv[l].dbox && !d->dbox_rows_only && hipMemsetAsync(c.lv[l].dbox, 0, (size_t)c.B * c.lv[l].H * c.lv[l].W * 64 * 2, stream) != hipSuccess)
      return DY_ERR_LAUNCH;
  hipLaunchKernelGGL(box_loss_kernel, dim3(gridA), dim3(256), 0, stream, c);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, stream, c, partB, gridA, partC, n_cls);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
