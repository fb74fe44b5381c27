// Backward of Detect's final box convolution (reference nn/modules/head.py:38-40: Conv2d(c2, 4 * reg_max, 1) after two Convs) when
// its output gradient has ROWS -- the loss writes a box-logit gradient for foreground anchors only (utils/loss.py:436-445 selects
// fg_mask before the box / DFL terms), 0.2 % of the anchors of a DEAL-YOLO-N batch; every other row is zero.  The dense path zeroed
// the whole gradient tensor (210 MB at 160x160, batch 64), read it back twice (weight gradient, input gradient) and read the layer
// input once for a product with zeros.  Here the foreground flags (the loss's assignment, 4 bytes per anchor) decide which rows are
// touched at all: the weight / bias gradient visits foreground pixels only, the input gradient writes zeros elsewhere without
// reading anything.  Deterministic: every workgroup walks its pixels in order.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "common.h"
#include "dealyolo_hip.h"

// Several detection levels in ONE launch: the kernels below take a table of per-level argument blocks and blockIdx.y (blockIdx.z where y
// is the image) picks the level, so the small levels' workgroups -- latency-bound walks over a few thousand pixels -- run beside the
// large level's instead of after it, and three launches per kind become one.
template <typename T>
struct Levels {
  int nl;
  T lv[4];
};
struct RowsArgs {
  const f16* x;      // [npix][ldx] layer input (activated), cin = 64 channels
  const f16* dy;     // [npix][lddy] gradient of the 64 box logits; only rows whose flag >= 0 hold data
  const int* flag;   // the loss's assigned-gt index per anchor, [B][A]; pixel (b, r) of this level is anchor a0 + r
  const float* w;    // fp32 master weight [64][64] (cout, cin)
  f16* dx;           // [npix][lddx]
  float* slabs;      // [gridDim.x][64][64] fp32 partial weight gradients for dy_wgrad_reduce_batched
  double* bias_acc;  // [DY_BN_COPIES][64]
  int ldx, lddy, lddx, A, a0, hw, B, dx_acc;
  const float* xcoef;  // non-null: x is the RAW output of the Conv below ([4][64]: scale, shift, ..); BatchNorm + SiLU applied on load
  int nblk, dgx;       // workgroups of the weight-gradient walk (= slabs) / of the input-gradient kernel per image, for THIS level
};

// ---- weight + bias gradient over the foreground pixels of this workgroup's pixel range
__global__ __launch_bounds__(256) void rows_wgrad_kernel(Levels<RowsArgs> L) {
  const RowsArgs& a = L.lv[blockIdx.y];
  if ((int)blockIdx.x >= a.nblk) return;
  __shared__ int s_list[256];
  __shared__ int s_cnt[4];
  __shared__ __attribute__((aligned(16))) f16 s_x[16 * 64], s_dy[16 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int co = tid >> 2, ci0 = (tid & 3) * 16;
  const long npix = (long)a.B * a.hw;
  const long chunk = (npix + a.nblk - 1) / a.nblk;
  const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < npix ? lo + chunk : npix;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  float bs = 0.f;
  // the flags of eight trips are requested together: one trip at a time the loop was a chain of 25 dependent memory latencies at
  // 160x160 (40 us for a pass that moves 6.5 MB)
  constexpr int FB = 8;
  int fl[FB];
  for (long p0 = lo, trip = 0; p0 < hi; p0 += 256, ++trip) {
    if ((trip & (FB - 1)) == 0) {
#pragma unroll
      for (int k = 0; k < FB; ++k) {
        const long pk = p0 + (long)k * 256 + tid;
        fl[k] = -1;
        if (pk < hi) {
          const int b = (int)(pk / a.hw), r = (int)(pk - (long)b * a.hw);
          fl[k] = a.flag[(size_t)b * a.A + a.a0 + r];
        }
      }
    }
    int mine = -1;
#pragma unroll
    for (int k = 0; k < FB; ++k)
      if ((trip & (FB - 1)) == k) mine = fl[k];
    const bool fg = mine >= 0;
    const unsigned long long m = __ballot(fg);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) base += s_cnt[w];
      total += s_cnt[w];
    }
    if (fg) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = tid;  // ordered: ascending pixel
    __syncthreads();
    // the rows of up to 16 foreground pixels are staged per barrier pair: thread (i, g) brings granule g of pixel i (8 of x, 8 of dy)
    for (int i0 = 0; i0 < total; i0 += 16) {
      const int nb = total - i0 < 16 ? total - i0 : 16;
      {
        const int i = tid >> 4, g = tid & 15;
        if (i < nb) {
          const long q = p0 + s_list[i0 + i];
          if (g < 8) {
            half8 xv = *reinterpret_cast<const half8*>(a.x + q * a.ldx + g * 8);
            if (a.xcoef) xv = bn_silu_apply8(xv, a.xcoef + g * 8, a.xcoef + 64 + g * 8);
            *reinterpret_cast<half8*>(s_x + i * 64 + g * 8) = xv;
          } else {
            *reinterpret_cast<uint4*>(s_dy + i * 64 + (g - 8) * 8) = *reinterpret_cast<const uint4*>(a.dy + q * a.lddy + (g - 8) * 8);
          }
        }
      }
      __syncthreads();
      for (int i = 0; i < nb; ++i) {
        const float g = (float)s_dy[i * 64 + co];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] += g * (float)s_x[i * 64 + ci0 + k];
        if ((tid & 3) == 0) bs += g;
      }
      __syncthreads();
    }
  }
  float* slab = a.slabs + (size_t)blockIdx.x * 64 * 64 + co * 64 + ci0;
#pragma unroll
  for (int k = 0; k < 16; k += 4) *reinterpret_cast<float4*>(slab + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
  if ((tid & 3) == 0 && bs != 0.f) unsafeAtomicAdd(a.bias_acc + (size_t)(blockIdx.x % DY_BN_COPIES) * 64 + co, (double)bs);
}

// ---- input gradient: dx[p] = W^T dy[p] on foreground pixels (weights rounded to fp16 as the packed form the dense kernel multiplies
// with, fp32 sum over the output channels in ascending order), zeros elsewhere; blockIdx.y = image
__global__ __launch_bounds__(256) void rows_dgrad_kernel(Levels<RowsArgs> L) {
  const RowsArgs& a = L.lv[blockIdx.z];
  if (!a.dx || (int)blockIdx.x >= a.dgx || (int)blockIdx.y >= a.B) return;
  __shared__ float s_w[64 * 65];  // [co][ci], pitch 65
  for (int i = threadIdx.x; i < 64 * 64; i += 256) s_w[(i >> 6) * 65 + (i & 63)] = (float)(f16)a.w[i];
  __syncthreads();
  const int b = blockIdx.y, part = threadIdx.x & 7, ci0 = part * 8;
  const int* const fl = a.flag + (size_t)b * a.A + a.a0;
  for (int r = (int)(blockIdx.x * 32 + (threadIdx.x >> 3)); r < a.hw; r += a.dgx * 32) {
    const long p = (long)b * a.hw + r;
    f16* const dst = a.dx + p * a.lddx + ci0;
    if (fl[r] < 0) {
      if (!a.dx_acc) *reinterpret_cast<uint4*>(dst) = make_uint4(0, 0, 0, 0);
      continue;
    }
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    const f16* const g = a.dy + p * a.lddy;
    for (int c8 = 0; c8 < 64; c8 += 8) {
      const half8 gv = *reinterpret_cast<const half8*>(g + c8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = (float)gv[j];
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += gg * s_w[(c8 + j) * 65 + ci0 + k];
      }
    }
    half8 o;
    if (a.dx_acc) {
      const half8 old = *reinterpret_cast<const half8*>(dst);
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f16)((float)old[k] + s[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (f16)s[k];
    }
    *reinterpret_cast<half8*>(dst) = o;
  }
}

// ---- Forward of the same convolution fused with the loss's box decode (reference utils/loss.py:347-354, bbox_decode: softmax over the
// 16 DFL bins of each side, expectation, dist2bbox against the anchor point).  Inside a training step nothing needs the 64 fp32
// logits of all 2.2 M anchors: the assigner wants the decoded box (4 floats), the DFL / IoU terms want the logits of the ~0.2 %
// foreground anchors -- which the loss recomputes from this layer's input.  So the layer reads its input once and writes 16 bytes per
// anchor instead of 256 (and the 557 MB are not read back by a decode launch).
// One wave = 64 pixels: B fragments straight from global memory (a pixel's 64 channels are 128 contiguous bytes), A fragments from
// an LDS copy of the weights whose rows are permuted so that a lane ends up with the 16 bins of ONE side of its pixel: MFMA row
// 16 m + 4 q + j holds channel 16 q + 4 m + j.  The arithmetic after the accumulators is decode_kernel's, association included.
// exp(x) for x <= 0 (softmax terms after the maximum is taken off): 2^(x log2 e) with the product carried in two floats -- v_exp_f32 of
// the head, first-order correction by the tail.  About 1 ulp; 7 instructions against ~20 of the library call, and this kernel runs
// 64 of them per anchor (DY_DECODE_LIBM_EXP=1 restores expf).
static __device__ __forceinline__ float exp_nonpos(float x) {
  const float t = x * 1.44269502f;
  float r = __builtin_fmaf(x, 1.44269502f, -t);
  r = __builtin_fmaf(x, 1.92596299e-8f, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, r * 0.693147182f, e);
}
// the same for a pair (element-wise identical arithmetic: v_pk_mul / v_pk_fma around the two v_exp)
static __device__ __forceinline__ f32x2 exp_nonpos2(f32x2 x) {
  const f32x2 l2e = {1.44269502f, 1.44269502f}, l2e_lo = {1.92596299e-8f, 1.92596299e-8f}, ln2 = {0.693147182f, 0.693147182f};
  const f32x2 t = x * l2e;
  f32x2 r = __builtin_elementwise_fma(x, l2e, -t);
  r = __builtin_elementwise_fma(x, l2e_lo, r);
  const f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  return __builtin_elementwise_fma(e, r * ln2, e);
}
struct BoxDecArgs {
  const f16* x;
  const float* w;     // fp32 master [64][64]
  const float* bias;  // [64]
  float* pred_box;    // (B, A, 4) grid units
  int ldx, A, a0, hw, W, B;
  const float* xcoef;  // non-null: x is the RAW output of the Conv below; its BatchNorm + SiLU apply happens on load, no apply launch
  int nblk;
};
template <bool LIBM>
__global__ __launch_bounds__(256) void head_box_decode_kernel(Levels<BoxDecArgs> L) {
  const BoxDecArgs& a = L.lv[blockIdx.y];
  if ((int)blockIdx.x >= a.nblk) return;
  auto ex = [](float x) { return LIBM ? expf(x) : exp_nonpos(x); };
  constexpr int PITCH = 160;  // bytes per weight row: 10 sixteen-byte slots (== 2 mod 4: conflict-free for ds_read_b128's lane groups)
  __shared__ __attribute__((aligned(16))) char s_w[64 * PITCH];
  __shared__ float s_b[64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, k = i & 63;                          // channel c = 16 q' + 4 m' + j'  ->  row 16 m' + 4 q' + j'
    const int row = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3);
    *reinterpret_cast<f16*>(s_w + row * PITCH + k * 2) = (f16)a.w[i];
  }
  if (tid < 64) s_b[tid] = a.bias[tid];
  __syncthreads();
  const long npix = (long)a.B * a.hw;
  const float invw = 1.0f / (float)a.W;
  float csc[2][8], csh[2][8];  // this lane's 16 input channels: ks * 32 + q * 8 + j
  if (a.xcoef) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        csc[ks][j] = a.xcoef[ks * 32 + q * 8 + j];
        csh[ks][j] = a.xcoef[64 + ks * 32 + q * 8 + j];
      }
  }
  for (long base = ((long)blockIdx.x * 4 + wave) * 64; base < npix; base += (long)a.nblk * 256) {
    half8 bf[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long pix = base + t * 16 + p;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if (pix < npix) bf[t][ks] = *reinterpret_cast<const half8*>(a.x + pix * a.ldx + ks * 32 + q * 8);
        else bf[t][ks] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
    if (a.xcoef) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          if (base + t * 16 + p < npix) bf[t][ks] = bn_silu_apply8(bf[t][ks], csc[ks], csh[ks]);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const half8 a0f = *reinterpret_cast<const half8*>(s_w + (m * 16 + p) * PITCH + q * 16);
      const half8 a1f = *reinterpret_cast<const half8*>(s_w + (m * 16 + p) * PITCH + 64 + q * 16);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0f, bf[t][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1f, bf[t][1], acc[m][t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long pix = base + t * 16 + p;
      if (pix >= npix) continue;
      float v[16];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[m * 4 + j] = acc[m][t][j] + s_b[q * 16 + m * 4 + j];
      float mx = v[0];
#pragma unroll
      for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
      float den4[4], num4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // decode_kernel's quarters: four consecutive bins, then (q0 + q1) + (q2 + q3)
        float e0, e1, e2, e3;
        if (LIBM) {
          e0 = ex(v[g * 4] - mx); e1 = ex(v[g * 4 + 1] - mx); e2 = ex(v[g * 4 + 2] - mx); e3 = ex(v[g * 4 + 3] - mx);
        } else {
          const f32x2 m2 = {mx, mx};
          const f32x2 ea = exp_nonpos2((f32x2){v[g * 4], v[g * 4 + 1]} - m2), eb = exp_nonpos2((f32x2){v[g * 4 + 2], v[g * 4 + 3]} - m2);
          e0 = ea[0]; e1 = ea[1]; e2 = eb[0]; e3 = eb[1];
        }
        const float k0 = (float)(g * 4);
        den4[g] = (e0 + e1) + (e2 + e3);
        num4[g] = e0 * k0 + e1 * (k0 + 1.f) + e2 * (k0 + 2.f) + e3 * (k0 + 3.f);
      }
      const float den = (den4[0] + den4[1]) + (den4[2] + den4[3]), num = (num4[0] + num4[1]) + (num4[2] + num4[3]);
      const float e = num / den;
      const int b = (int)(pix / a.hw), r = (int)(pix - (long)b * a.hw);
      const int iy = (int)(((float)r + 0.5f) * invw), ix = r - iy * a.W;
      const float anc = (q & 1) ? (iy + 0.5f) : (ix + 0.5f);
      a.pred_box[((size_t)b * a.A + a.a0 + r) * 4 + q] = q < 2 ? anc - e : anc + e;
    }
  }
}
static int box_decode_fill(BoxDecArgs& a, const void* x, int ldx, const float* x_coef, const float* weight, const float* bias, float* pred_box,
                           int A, int a0, int n, int h, int w, int cin, int cout) {
  if (cin != 64 || cout != 64 || !x || !weight || !bias || !pred_box || n < 1 || h < 1 || w < 1 || a0 < 0 || a0 + h * w > A) return DY_ERR_ARG;
  if ((ldx & 7) || ((uintptr_t)x & 15)) return DY_ERR_ALIGN;
  static const int cap = getenv("DY_HEAD_DECODE_WGS") ? atoi(getenv("DY_HEAD_DECODE_WGS")) : 2048;  // measurement knob (workgroups per level)
  long blocks = ((long)n * h * w + 255) / 256;
  if (blocks > cap) blocks = cap;
  a = BoxDecArgs{(const f16*)x, weight, bias, pred_box, ldx, A, a0, h * w, w, n, x_coef, (int)blocks};
  return DY_OK;
}
static int box_decode_launch(const Levels<BoxDecArgs>& L, hipStream_t stream) {
  int gx = 1;
  for (int l = 0; l < L.nl; ++l) gx = L.lv[l].nblk > gx ? L.lv[l].nblk : gx;
  static const bool libm = getenv("DY_DECODE_LIBM_EXP") && atoi(getenv("DY_DECODE_LIBM_EXP")) != 0;
  if (libm) hipLaunchKernelGGL(head_box_decode_kernel<true>, dim3(gx, L.nl), dim3(256), 0, stream, L);
  else hipLaunchKernelGGL(head_box_decode_kernel<false>, dim3(gx, L.nl), dim3(256), 0, stream, L);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_head_box_decode(const void* x, int ldx, const float* x_coef, const float* weight, const float* bias, float* pred_box, int A,
                                  int a0, int n, int h, int w, int cin, int cout, hipStream_t stream) {
  Levels<BoxDecArgs> L{};
  L.nl = 1;
  const int rc = box_decode_fill(L.lv[0], x, ldx, x_coef, weight, bias, pred_box, A, a0, n, h, w, cin, cout);
  return rc != DY_OK ? rc : box_decode_launch(L, stream);
}
extern "C" int dy_head_box_decode_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const float* const* weight,
                                         const float* const* bias, float* pred_box, int A, const int* a0, int n, const int* h, const int* w,
                                         int cin, int cout, hipStream_t stream) {
  if (nl < 1 || nl > 4) return DY_ERR_ARG;
  Levels<BoxDecArgs> L{};
  L.nl = nl;
  for (int l = 0; l < nl; ++l) {
    const int rc = box_decode_fill(L.lv[l], x[l], ldx[l], x_coef ? x_coef[l] : nullptr, weight[l], bias[l], pred_box, A, a0[l], n, h[l], w[l], cin, cout);
    if (rc != DY_OK) return rc;
  }
  return box_decode_launch(L, stream);
}

extern "C" int dy_conv1x1_rows_supported(int cin, int cout) { return cin == 64 && cout == 64; }
// workgroups (= fp32 slabs of 16 KB) of the weight-gradient walk: four per CU, so that one's flag / row latencies lie under another's
// arithmetic (256: 144 -> 96 us per call at 160x160 with everything else in place; 1024: see profiles/r03_stage_ab.md)
extern "C" int dy_conv1x1_rows_slabs(int n, int h, int w) {
  const long ns = (long)n * h * w / 2048;  // measured per call at 160 / 80 / 40 (batch 64): 256 slabs 96 / 28 / 20 us, 1024 slabs 79 / 30 / 26 us
  return ns < 64 ? 64 : (ns > 1024 ? 1024 : (int)ns);
}
static int rows_fill(RowsArgs& a, const void* x, int ldx, const float* x_coef, const void* dy, int lddy, const int* assigned, int A, int a0,
                     const float* weight, void* dx, int lddx, int dx_accumulate, float* slabs, double* bias_acc, int n, int h, int w, int cin,
                     int cout) {
  if (!dy_conv1x1_rows_supported(cin, cout) || !x || !dy || !assigned || !weight || !slabs || !bias_acc || n < 1 || h < 1 || w < 1 ||
      a0 < 0 || a0 + h * w > A)
    return DY_ERR_ARG;
  if ((ldx & 7) || (lddy & 7) || (dx && (lddx & 7)) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) ||
      ((uintptr_t)slabs & 15))
    return DY_ERR_ALIGN;
  int gx = (h * w + 1023) / 1024;  // every input-gradient workgroup copies the 16 KB weight into LDS first: give it >= 1024 pixels
  if (gx > 16) gx = 16;
  a = RowsArgs{(const f16*)x, (const f16*)dy, assigned, weight, (f16*)dx, slabs, bias_acc, ldx, lddy, lddx, A, a0, h * w, n, dx_accumulate, x_coef,
               dy_conv1x1_rows_slabs(n, h, w), gx};
  return DY_OK;
}
static int rows_launch(const Levels<RowsArgs>& L, hipStream_t stream) {
  int gw = 1, gd = 0, nb = 1;
  for (int l = 0; l < L.nl; ++l) {
    gw = L.lv[l].nblk > gw ? L.lv[l].nblk : gw;
    if (L.lv[l].dx) gd = L.lv[l].dgx > gd ? L.lv[l].dgx : gd;
    nb = L.lv[l].B > nb ? L.lv[l].B : nb;
  }
  hipLaunchKernelGGL(rows_wgrad_kernel, dim3(gw, L.nl), dim3(256), 0, stream, L);
  if (gd) hipLaunchKernelGGL(rows_dgrad_kernel, dim3(gd, nb, L.nl), dim3(256), 0, stream, L);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_conv1x1_rows_backward(const void* x, int ldx, const float* x_coef, const void* dy, int lddy, const int* assigned, int A, int a0,
                                        const float* weight, void* dx, int lddx, int dx_accumulate, float* slabs, double* bias_acc,
                                        int n, int h, int w, int cin, int cout, hipStream_t stream) {
  Levels<RowsArgs> L{};
  L.nl = 1;
  const int rc = rows_fill(L.lv[0], x, ldx, x_coef, dy, lddy, assigned, A, a0, weight, dx, lddx, dx_accumulate, slabs, bias_acc, n, h, w, cin, cout);
  return rc != DY_OK ? rc : rows_launch(L, stream);
}
extern "C" int dy_conv1x1_rows_backward_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const void* const* dy,
                                               const int* lddy, const int* assigned, int A, const int* a0, const float* const* weight,
                                               void* const* dx, const int* lddx, const int* dx_accumulate, float* const* slabs,
                                               double* const* bias_acc, int n, const int* h, const int* w, int cin, int cout,
                                               hipStream_t stream) {
  if (nl < 1 || nl > 4) return DY_ERR_ARG;
  Levels<RowsArgs> L{};
  L.nl = nl;
  for (int l = 0; l < nl; ++l) {
    const int rc = rows_fill(L.lv[l], x[l], ldx[l], x_coef ? x_coef[l] : nullptr, dy[l], lddy[l], assigned, A, a0[l], weight[l], dx[l], lddx[l],
                             dx_accumulate[l], slabs[l], bias_acc[l], n, h[l], w[l], cin, cout);
    if (rc != DY_OK) return rc;
  }
  return rows_launch(L, stream);
}

// ================================================================================================================================
// Detect's final CLASS convolution (reference nn/modules/head.py:41-42: Conv2d(c3, nc, 1) with bias; nc <= 8 here) as stand-alone
// element-wise kernels.  Eight output channels make no matrix tile; as a generic conv launch the layer moved 157 MB at 2.7 TB/s on the
// 160x160 level, after an apply launch of its own for the Conv in front.  Here PARTS = cin / 8 lanes share a pixel (one 16-byte granule
// each, fully coalesced), apply that Conv's BatchNorm + SiLU to the RAW granule they load, multiply with the 8 x 8 weight block they own
// and meet in two / three shuffle steps.  The backward does the same walk once for all three gradients: weight (per-lane 8 x 8
// accumulators, summed over the workgroup at the end, one slab per workgroup), bias (fp64 accumulator) and input (W^T dy, written
// as the gradient of the Conv in front).
struct ClsArgs {
  const f16* x;        // [npix][ldx]: RAW output of the Conv in front when xcoef is set, else its activated output
  const float* xcoef;  // [4][cin] scale, shift, ..
  const float* w;      // fp32 master [nc][cin]
  const float* bias;   // [nc]
  float* y;            // forward: logits [npix][8] fp32
  const f16* dy;       // backward: [npix][8]
  f16* dx;             // backward: [npix][lddx] gradient of the activated input (may be NULL)
  float* slabs;        // backward: [gridDim.x][16][cin] fp32
  double* bias_acc;    // backward: [DY_BN_COPIES][8]
  int ldx, lddx, nc, dx_acc;
  long npix;
  int nblk;
};
template <int CIN>
__global__ __launch_bounds__(256) void cls_head_fwd_kernel(Levels<ClsArgs> L) {
  const ClsArgs& a = L.lv[blockIdx.y];
  if ((int)blockIdx.x >= a.nblk) return;
  constexpr int PARTS = CIN / 8, PPB = 256 / PARTS;
  __shared__ float s_w[8][CIN], s_b[8];
  for (int i = threadIdx.x; i < 8 * CIN; i += 256) s_w[i / CIN][i % CIN] = (i / CIN) < a.nc ? (float)(f16)a.w[i] : 0.f;
  if (threadIdx.x < 8) s_b[threadIdx.x] = threadIdx.x < a.nc ? a.bias[threadIdx.x] : 0.f;
  __syncthreads();
  const int part = threadIdx.x % PARTS, k0 = part * 8;
  float sc[8], sh[8];
  if (a.xcoef) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = a.xcoef[k0 + j]; sh[j] = a.xcoef[CIN + k0 + j]; }
  }
  for (long pix = (long)blockIdx.x * PPB + threadIdx.x / PARTS; pix < a.npix; pix += (long)a.nblk * PPB) {
    half8 xv = *reinterpret_cast<const half8*>(a.x + pix * a.ldx + k0);
    if (a.xcoef) xv = bn_silu_apply8(xv, sc, sh);
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float t = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) t += s_w[c][k0 + j] * (float)xv[j];
      o[c] = t;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = PARTS == 4 ? quad_sum(o[c]) : half_row_sum(o[c]);  // the pixel's lanes: one quad / one half row (DPP)
    if (part == 0) {
      float* dst = a.y + pix * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(o[0] + s_b[0], o[1] + s_b[1], o[2] + s_b[2], o[3] + s_b[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4] + s_b[4], o[5] + s_b[5], o[6] + s_b[6], o[7] + s_b[7]);
    }
  }
}
template <int CIN>
__global__ __launch_bounds__(256) void cls_head_bwd_kernel(Levels<ClsArgs> L) {
  const ClsArgs& a = L.lv[blockIdx.y];
  if ((int)blockIdx.x >= a.nblk) return;
  constexpr int PARTS = CIN / 8, PPB = 256 / PARTS;
  __shared__ float s_w[8][CIN];
  __shared__ float s_red[4][PARTS][64];
  __shared__ float s_bs[4][8];
  for (int i = threadIdx.x; i < 8 * CIN; i += 256) s_w[i / CIN][i % CIN] = (i / CIN) < a.nc ? (float)(f16)a.w[i] : 0.f;
  __syncthreads();
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, part = tid % PARTS, k0 = part * 8;
  float sc[8], sh[8];
  if (a.xcoef) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = a.xcoef[k0 + j]; sh[j] = a.xcoef[CIN + k0 + j]; }
  }
  float acc[8][8], bs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    bs[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
  }
  // a workgroup owns a contiguous pixel range (its sums have one order whatever the grid)
  const long chunk = ((a.npix + a.nblk - 1) / a.nblk + PPB - 1) / PPB * PPB;
  const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < a.npix ? lo + chunk : a.npix;
  for (long pix = lo + tid / PARTS; pix < hi; pix += PPB) {
    half8 xv = *reinterpret_cast<const half8*>(a.x + pix * a.ldx + k0);
    const half8 dv = *reinterpret_cast<const half8*>(a.dy + pix * 8);
    if (a.xcoef) xv = bn_silu_apply8(xv, sc, sh);
    float d[8], g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) d[c] = (float)dv[c];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc[c][j] += d[c] * (float)xv[j];
        g[j] += s_w[c][k0 + j] * d[c];
      }
      if (part == 0) bs[c] += d[c];
    }
    if (a.dx) {
      f16* dst = a.dx + pix * a.lddx + k0;
      half8 o;
      if (a.dx_acc) {
        const half8 old = *reinterpret_cast<const half8*>(dst);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)old[j] + g[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)g[j];
      }
      *reinterpret_cast<half8*>(dst) = o;
    }
  }
  // lanes with the same channel part (lane % PARTS) -> one lane per wave, waves -> LDS, then one slab [16][CIN] per workgroup
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[c][j];
#pragma unroll
      for (int m = PARTS; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
      acc[c][j] = v;
    }
    float b = bs[c];
#pragma unroll
    for (int m = PARTS; m < 64; m <<= 1) b += __shfl_xor(b, m, 64);
    bs[c] = b;
  }
  if (lane < PARTS) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) s_red[wave][lane][c * 8 + j] = acc[c][j];
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 8; ++c) s_bs[wave][c] = bs[c];
  }
  __syncthreads();
  float* slab = a.slabs + (size_t)blockIdx.x * 16 * CIN;
  for (int i = tid; i < 16 * CIN; i += 256) {
    const int co = i / CIN, ci = i - co * CIN;
    float v = 0.f;
    if (co < 8) {
      const int pp = ci >> 3, j = ci & 7;
      v = (s_red[0][pp][co * 8 + j] + s_red[1][pp][co * 8 + j]) + (s_red[2][pp][co * 8 + j] + s_red[3][pp][co * 8 + j]);
    }
    slab[i] = v;
  }
  if (tid < 8) {
    const float b = (s_bs[0][tid] + s_bs[1][tid]) + (s_bs[2][tid] + s_bs[3][tid]);
    if (b != 0.f) unsafeAtomicAdd(a.bias_acc + (size_t)(blockIdx.x % DY_BN_COPIES) * 8 + tid, (double)b);
  }
}
extern "C" int dy_cls_head_supported(int cin, int nc) { return (cin == 32 || cin == 64) && nc >= 1 && nc <= 8; }
extern "C" int dy_cls_head_slabs(void) { return 512; }
static int cls_fill(ClsArgs& a, bool backward, const void* x, int ldx, const float* x_coef, const float* weight, const float* bias, float* logits,
                    const void* dy, void* dx, int lddx, int dx_accumulate, float* slabs, double* bias_acc, long npix, int cin, int nc) {
  if (!dy_cls_head_supported(cin, nc) || !x || !weight || npix < 1) return DY_ERR_ARG;
  if (!backward && (!bias || !logits)) return DY_ERR_ARG;
  if (backward && (!dy || !slabs || !bias_acc)) return DY_ERR_ARG;
  if ((ldx & 7) || (dx && (lddx & 7)) || ((uintptr_t)x & 15) || ((uintptr_t)logits & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15))
    return DY_ERR_ALIGN;
  int nblk;
  if (backward) {
    nblk = dy_cls_head_slabs();
  } else {
    const int ppb = 256 / (cin / 8);
    long blocks = (npix + ppb - 1) / ppb;
    nblk = (int)(blocks > 4096 ? 4096 : blocks);
  }
  a = ClsArgs{(const f16*)x, x_coef, weight, bias, logits, (const f16*)dy, (f16*)dx, slabs, bias_acc, ldx, lddx, nc, dx_accumulate, npix, nblk};
  return DY_OK;
}
static int cls_launch(const Levels<ClsArgs>& L, bool backward, int cin, hipStream_t stream) {
  int gx = 1;
  for (int l = 0; l < L.nl; ++l) gx = L.lv[l].nblk > gx ? L.lv[l].nblk : gx;
  const dim3 grid(gx, L.nl);
  if (!backward && cin == 32) hipLaunchKernelGGL(cls_head_fwd_kernel<32>, grid, dim3(256), 0, stream, L);
  else if (!backward) hipLaunchKernelGGL(cls_head_fwd_kernel<64>, grid, dim3(256), 0, stream, L);
  else if (cin == 32) hipLaunchKernelGGL(cls_head_bwd_kernel<32>, grid, dim3(256), 0, stream, L);
  else hipLaunchKernelGGL(cls_head_bwd_kernel<64>, grid, dim3(256), 0, stream, L);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_cls_head_forward(const void* x, int ldx, const float* x_coef, const float* weight, const float* bias, float* logits,
                                   long npix, int cin, int nc, hipStream_t stream) {
  Levels<ClsArgs> L{};
  L.nl = 1;
  const int rc = cls_fill(L.lv[0], false, x, ldx, x_coef, weight, bias, logits, nullptr, nullptr, 0, 0, nullptr, nullptr, npix, cin, nc);
  return rc != DY_OK ? rc : cls_launch(L, false, cin, stream);
}
extern "C" int dy_cls_head_backward(const void* x, int ldx, const float* x_coef, const void* dy, const float* weight, void* dx, int lddx,
                                    int dx_accumulate, float* slabs, double* bias_acc, long npix, int cin, int nc, hipStream_t stream) {
  Levels<ClsArgs> L{};
  L.nl = 1;
  const int rc = cls_fill(L.lv[0], true, x, ldx, x_coef, weight, nullptr, nullptr, dy, dx, lddx, dx_accumulate, slabs, bias_acc, npix, cin, nc);
  return rc != DY_OK ? rc : cls_launch(L, true, cin, stream);
}
extern "C" int dy_cls_head_forward_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const float* const* weight,
                                          const float* const* bias, float* const* logits, const long* npix, int cin, int nc,
                                          hipStream_t stream) {
  if (nl < 1 || nl > 4) return DY_ERR_ARG;
  Levels<ClsArgs> L{};
  L.nl = nl;
  for (int l = 0; l < nl; ++l) {
    const int rc = cls_fill(L.lv[l], false, x[l], ldx[l], x_coef ? x_coef[l] : nullptr, weight[l], bias[l], logits[l], nullptr, nullptr, 0, 0, nullptr,
                            nullptr, npix[l], cin, nc);
    if (rc != DY_OK) return rc;
  }
  return cls_launch(L, false, cin, stream);
}
extern "C" int dy_cls_head_backward_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const void* const* dy,
                                           const float* const* weight, void* const* dx, const int* lddx, const int* dx_accumulate,
                                           float* const* slabs, double* const* bias_acc, const long* npix, int cin, int nc,
                                           hipStream_t stream) {
  if (nl < 1 || nl > 4) return DY_ERR_ARG;
  Levels<ClsArgs> L{};
  L.nl = nl;
  for (int l = 0; l < nl; ++l) {
    const int rc = cls_fill(L.lv[l], true, x[l], ldx[l], x_coef ? x_coef[l] : nullptr, weight[l], nullptr, nullptr, dy[l], dx[l], lddx[l],
                            dx_accumulate[l], slabs[l], bias_acc[l], npix[l], cin, nc);
    if (rc != DY_OK) return rc;
  }
  return cls_launch(L, true, cin, stream);
}
