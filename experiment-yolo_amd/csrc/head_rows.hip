// Backward of Detect's final box convolution (reference nn/modules/head.py:38-40: Conv2d(c2, 4 * reg_max, 1) after two Convs) when
// its output gradient has ROWS -- the loss writes a box-logit gradient for foreground anchors only (utils/loss.py:436-445 selects
// fg_mask before the box / DFL terms), 0.2 % of the anchors of a DEAL-YOLO-N batch; every other row is zero.  The dense path zeroed
// the whole gradient tensor (210 MB at 160x160, batch 64), read it back twice (weight gradient, input gradient) and read the layer
// input once for a product with zeros.  Here the foreground flags (the loss's assignment, 4 bytes per anchor) decide which rows are
// touched at all: the weight / bias gradient visits foreground pixels only, the input gradient writes zeros elsewhere without
// reading anything.  Deterministic: every workgroup walks its pixels in order.
#include <hip/hip_runtime.h>

#include "common.h"
#include "dealyolo_hip.h"

struct RowsArgs {
  const f16* x;      // [npix][ldx] layer input (activated), cin = 64 channels
  const f16* dy;     // [npix][lddy] gradient of the 64 box logits; only rows whose flag >= 0 hold data
  const int* flag;   // the loss's assigned-gt index per anchor, [B][A]; pixel (b, r) of this level is anchor a0 + r
  const float* w;    // fp32 master weight [64][64] (cout, cin)
  f16* dx;           // [npix][lddx]
  float* slabs;      // [gridDim.x][64][64] fp32 partial weight gradients for dy_wgrad_reduce_batched
  double* bias_acc;  // [DY_BN_COPIES][64]
  int ldx, lddy, lddx, A, a0, hw, B, dx_acc;
};

// ---- weight + bias gradient over the foreground pixels of this workgroup's pixel range
__global__ __launch_bounds__(256) void rows_wgrad_kernel(RowsArgs a) {
  __shared__ int s_list[256];
  __shared__ int s_cnt[4];
  __shared__ __attribute__((aligned(16))) f16 s_x[64], s_dy[64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int co = tid >> 2, ci0 = (tid & 3) * 16;
  const long npix = (long)a.B * a.hw;
  const long chunk = (npix + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < npix ? lo + chunk : npix;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  float bs = 0.f;
  for (long p0 = lo; p0 < hi; p0 += 256) {
    const long p = p0 + tid;
    bool fg = false;
    if (p < hi) {
      const int b = (int)(p / a.hw), r = (int)(p - (long)b * a.hw);
      fg = a.flag[(size_t)b * a.A + a.a0 + r] >= 0;
    }
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
    for (int i = 0; i < total; ++i) {
      const long q = p0 + s_list[i];
      if (tid < 8) *reinterpret_cast<uint4*>(s_x + tid * 8) = *reinterpret_cast<const uint4*>(a.x + q * a.ldx + tid * 8);
      else if (tid < 16) *reinterpret_cast<uint4*>(s_dy + (tid - 8) * 8) = *reinterpret_cast<const uint4*>(a.dy + q * a.lddy + (tid - 8) * 8);
      __syncthreads();
      const float g = (float)s_dy[co];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] += g * (float)s_x[ci0 + k];
      if ((tid & 3) == 0) bs += g;
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
__global__ __launch_bounds__(256) void rows_dgrad_kernel(RowsArgs a) {
  __shared__ float s_w[64 * 65];  // [co][ci], pitch 65
  for (int i = threadIdx.x; i < 64 * 64; i += 256) s_w[(i >> 6) * 65 + (i & 63)] = (float)(f16)a.w[i];
  __syncthreads();
  const int b = blockIdx.y, part = threadIdx.x & 7, ci0 = part * 8;
  const int* const fl = a.flag + (size_t)b * a.A + a.a0;
  for (int r = (int)(blockIdx.x * 32 + (threadIdx.x >> 3)); r < a.hw; r += (int)gridDim.x * 32) {
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

extern "C" int dy_conv1x1_rows_supported(int cin, int cout) { return cin == 64 && cout == 64; }
extern "C" int dy_conv1x1_rows_slabs(void) { return 256; }
extern "C" int dy_conv1x1_rows_backward(const void* x, int ldx, const void* dy, int lddy, const int* assigned, int A, int a0,
                                        const float* weight, void* dx, int lddx, int dx_accumulate, float* slabs, double* bias_acc,
                                        int n, int h, int w, int cin, int cout, hipStream_t stream) {
  if (!dy_conv1x1_rows_supported(cin, cout) || !x || !dy || !assigned || !weight || !slabs || !bias_acc || n < 1 || h < 1 || w < 1 ||
      a0 < 0 || a0 + h * w > A)
    return DY_ERR_ARG;
  if ((ldx & 7) || (lddy & 7) || (dx && (lddx & 7)) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) ||
      ((uintptr_t)slabs & 15))
    return DY_ERR_ALIGN;
  RowsArgs a{(const f16*)x, (const f16*)dy, assigned, weight, (f16*)dx, slabs, bias_acc, ldx, lddy, lddx, A, a0, h * w, n, dx_accumulate};
  hipLaunchKernelGGL(rows_wgrad_kernel, dim3(dy_conv1x1_rows_slabs()), dim3(256), 0, stream, a);
  if (dx) {
    int gx = (h * w + 31) / 32;
    if (gx > 128) gx = 128;
    hipLaunchKernelGGL(rows_dgrad_kernel, dim3(gx, n), dim3(256), 0, stream, a);
  }
  DY_CHECK_LAUNCH();
  return DY_OK;
}
