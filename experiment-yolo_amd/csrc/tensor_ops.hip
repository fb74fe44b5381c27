// Data-movement operators of the DEAL-YOLO graph on NHWC fp16 tensors addressed as (pointer, pixel stride):
// image import, nearest 2x up-sampling (nn.Upsample in the model YAMLs), the 5x5 stride-1 max-pool chain of SPPF
// (reference nn/modules/block.py:166-171), element-wise adds (Add / Bottleneck shortcut, gradient fan-in) and strided
// channel-slice copies (Concat, reference nn/modules/conv.py:338-348, when a producer cannot write in place).
// All are HBM-bound byte movers: 16 bytes per lane, pixel-major so that wave accesses are contiguous.
#include <cstdlib>
#include "common.h"
#include "dealyolo_hip.h"

static inline int grid_for(long total) {
  long b = (total + 255) / 256;
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- NCHW fp32 image -> NHWC fp16 with channels zero-padded to Cp (stem input)
__global__ __launch_bounds__(256) void import_image_kernel(const float* x, f16* y, int N, int C, int H, int W, int Cp,
                                                           float mul) {
  const long hw = (long)H * W, total = (long)N * hw;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const long n = pix / hw, r = pix - n * hw;
    for (int c0 = 0; c0 < Cp; c0 += 8) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c0 + j < C) ? (f16)(x[(n * C + c0 + j) * hw + r] * mul) : (f16)0.f;
      *reinterpret_cast<half8*>(y + pix * Cp + c0) = v;
    }
  }
}
extern "C" int dy_import_image(const float* x, void* y, int n, int c, int h, int w, int cp, float mul,
                               hipStream_t stream) {
  if (cp & 7) return DY_ERR_ALIGN;
  hipLaunchKernelGGL(import_image_kernel, dim3(grid_for((long)n * h * w)), dim3(256), 0, stream, x, (f16*)y, n, c, h, w,
                     cp, mul);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// uint8 NHWC RGB (what an image decoder / the data loader's pinned batch holds) -> fp16 NHWC, channels zero-padded to Cp,
// value = float(u8) / 255 exactly as the reference's preprocess_batch (models/yolo/detect/train.py:59) before the fp16 cast.
// Four pixels per thread: 12 input bytes as three dwords, four 16-byte stores per 8-channel granule.
__global__ __launch_bounds__(256) void import_image_u8_kernel(const unsigned char* x, f16* y, long npix, int Cp) {
  const long quads = npix >> 2;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long)gridDim.x * 256) {
    const unsigned* p = reinterpret_cast<const unsigned*>(x + q * 12);
    const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
    const unsigned char b[12] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                 (unsigned char)w1, (unsigned char)(w1 >> 8), (unsigned char)(w1 >> 16), (unsigned char)(w1 >> 24),
                                 (unsigned char)w2, (unsigned char)(w2 >> 8), (unsigned char)(w2 >> 16), (unsigned char)(w2 >> 24)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      half8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = j < 3 ? (f16)((float)b[3 * k + j] / 255.f) : (f16)0.f;
      f16* d = y + (q * 4 + k) * Cp;
      *reinterpret_cast<half8*>(d) = v;
      for (int c0 = 8; c0 < Cp; c0 += 8) *reinterpret_cast<half8*>(d + c0) = half8{};
    }
  }
  // tail (npix not a multiple of 4)
  for (long pix = (quads << 2) + (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < 3 ? (f16)((float)x[pix * 3 + j] / 255.f) : (f16)0.f;
    *reinterpret_cast<half8*>(y + pix * Cp) = v;
    for (int c0 = 8; c0 < Cp; c0 += 8) *reinterpret_cast<half8*>(y + pix * Cp + c0) = half8{};
  }
}
// RandomHSV (data/augment.py:605-624) on one 8-bit RGB pixel: to 8-bit HSV (H in [0,180), OpenCV's convention: V = max,
// S = 255*(V-min)/V, H = 30*sector formula, each rounded to nearest), the three look-up tables hue' = (H*r0) mod 180,
// sat' = min(S*r1, 255), val' = min(V*r2, 255) with numpy's truncating uint8 cast, and back.  OpenCV evaluates both conversions
// in fixed point / float with its own rounding, so values can differ from the reference's by an LSB or two (not pinned).
static __device__ __forceinline__ void hsv_jitter(float (&c)[3], float r0, float r1, float r2) {
  const float R = c[0], G = c[1], B = c[2];
  const float V = fmaxf(R, fmaxf(G, B)), mn = fminf(R, fminf(G, B)), d = V - mn;
  float H = 0.f;
  if (d > 0.f) {
    if (V == R) H = (G - B) / d;
    else if (V == G) H = 2.f + (B - R) / d;
    else H = 4.f + (R - G) / d;
    H *= 30.f;
    if (H < 0.f) H += 180.f;
  }
  float h8 = rintf(H);
  if (h8 >= 180.f) h8 -= 180.f;
  const float s8 = V > 0.f ? rintf(255.f * d / V) : 0.f;
  const float hh = floorf(fmodf(h8 * r0, 180.f)), ss = floorf(fminf(s8 * r1, 255.f)), vv = floorf(fminf(V * r2, 255.f));
  // HSV -> RGB (sector form), hue in degrees = 2*hh
  const float hs = hh / 30.f, sf = ss / 255.f;
  const int sec = (int)floorf(hs) % 6;
  const float f = hs - floorf(hs);
  const float p = vv * (1.f - sf), q = vv * (1.f - sf * f), t = vv * (1.f - sf * (1.f - f));
  float r, g, b;
  switch (sec) {
    case 0: r = vv; g = t; b = p; break;
    case 1: r = q; g = vv; b = p; break;
    case 2: r = p; g = vv; b = t; break;
    case 3: r = p; g = q; b = vv; break;
    case 4: r = t; g = p; b = vv; break;
    default: r = vv; g = p; b = q; break;
  }
  c[0] = fminf(fmaxf(rintf(r), 0.f), 255.f);
  c[1] = fminf(fmaxf(rintf(g), 0.f), 255.f);
  c[2] = fminf(fmaxf(rintf(b), 0.f), 255.f);
}

// The same conversion with the two flip augmentations folded in (RandomFlip, data/augment.py:651-683): flip[n] bit 0 mirrors
// image n left-right, bit 1 up-down -- the loader ships the pixels as decoded and this kernel reads them mirrored, so a flip
// costs no pass over the image anywhere.  index (optional): batch slot n reads image index[n] of x, which then is a pool of
// decoded images resident in HBM (a 100k-image 640x640 dataset is 123 GB of the 288): the step's input costs one gather-read.
// One output pixel per thread (mirrored quads would straddle the 12-byte groups).
__global__ __launch_bounds__(256) void import_image_u8_flip_kernel(const unsigned char* x, f16* y, int N, int H, int W, int Cp,
                                                                   const unsigned char* flip, const int* index, const float* hsv) {
  const long hw = (long)H * W, npix = hw * N;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
    const int n = (int)(pix / hw);
    const int r = (int)(pix - (long)n * hw);
    int yy = r / W, xx = r - yy * W;
    const unsigned f = flip ? flip[n] : 0u;
    if (f & 1) xx = W - 1 - xx;
    if (f & 2) yy = H - 1 - yy;
    const long src = index ? (long)index[n] : (long)n;
    const unsigned char* p = x + (src * hw + (long)yy * W + xx) * 3;
    float c[3] = {(float)p[0], (float)p[1], (float)p[2]};
    if (hsv) hsv_jitter(c, hsv[n * 3], hsv[n * 3 + 1], hsv[n * 3 + 2]);
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = j < 3 ? (f16)(c[j] / 255.f) : (f16)0.f;
    *reinterpret_cast<half8*>(y + pix * Cp) = v;
    for (int c0 = 8; c0 < Cp; c0 += 8) *reinterpret_cast<half8*>(y + pix * Cp + c0) = half8{};
  }
}
extern "C" int dy_import_image_u8(const void* x, void* y, int n, int h, int w, int cp, const void* flip, const int* index,
                                  const float* hsv, hipStream_t stream) {
  if ((cp & 7) || cp < 8) return DY_ERR_ALIGN;
  if ((uintptr_t)x & 3) return DY_ERR_ALIGN;
  const long npix = (long)n * h * w;
  if (flip || index || hsv)
    hipLaunchKernelGGL(import_image_u8_flip_kernel, dim3(grid_for(npix)), dim3(256), 0, stream, (const unsigned char*)x, (f16*)y, n, h, w, cp,
                       (const unsigned char*)flip, index, hsv);
  else hipLaunchKernelGGL(import_image_u8_kernel, dim3(grid_for((npix + 3) / 4)), dim3(256), 0, stream, (const unsigned char*)x, (f16*)y, npix, cp);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// Mosaic + random affine + flips composed ON THE DEVICE, straight into the fp16 stem input (Mosaic._mosaic4, data/augment.py:208-
// 241; RandomPerspective.affine_transform / cv2.warpAffine :384-435; RandomFlip :651-683).  The decoded training set lives in HBM
// (x = pool of letterboxed S x S uint8 images); the host only draws the random decisions and transforms the labels.  Per batch
// slot one WarpSlot: the inverse affine map (output pixel -> canvas), the canvas (2S x 2S mosaic of up to four pool images, each
// a destination rectangle + source offset, or the S x S letterbox of one image), the flip bits.  Every output pixel is one
// bilinear sample of that virtual canvas -- grey 114 outside the rectangles and outside the canvas (np.full / borderValue) --
// so the mosaic canvas, the warped image and the uint8 batch are never materialised.  Sampling is float bilinear with
// round-to-nearest; cv2's fixed-point INTER_LINEAR (5 fractional bits) is not reproduced.
struct WarpSlot {
  float minv[6];                   // u = minv[0]*x + minv[1]*y + minv[2];  v = minv[3]*x + minv[4]*y + minv[5]
  int canvas_w, canvas_h, xc, yc;  // mosaic centre: patch k = (u >= xc) + 2*(v >= yc); single image: xc = yc = canvas size
  int flip, npatch;
  int patch[4][7];                 // pool index, x1a, y1a, x2a, y2a (destination, exclusive ends), source x, y of the rectangle's corner
  float hsv[3];                    // RandomHSV gains (hue, saturation, value multipliers); hsv[0] == 0: off
  float pinv[3];                   // third row of the inverse map (RandomPerspective); (0, 0, 1) for an affine warp
  double mix_r;                    // MixUp ratio of THIS image against the partner record that follows; < 0: no partner
};
static_assert(sizeof(WarpSlot) == 48 * 4, "WarpSlot is 48 words (ultralytics/data/dataset.py: warp_slot)");

static __device__ __forceinline__ void warp_fetch(const unsigned char* pool, const WarpSlot& w, int S, int cx, int cy, float (&v)[3]) {
  v[0] = v[1] = v[2] = 114.f;
  if ((unsigned)cx >= (unsigned)w.canvas_w || (unsigned)cy >= (unsigned)w.canvas_h) return;
  const int k = w.npatch == 1 ? 0 : (cx >= w.xc ? 1 : 0) + (cy >= w.yc ? 2 : 0);
  const int* p = w.patch[k];
  if (cx < p[1] || cx >= p[3] || cy < p[2] || cy >= p[4]) return;
  const int sx = cx - p[1] + p[5], sy = cy - p[2] + p[6];
  if ((unsigned)sx >= (unsigned)S || (unsigned)sy >= (unsigned)S) return;
  const unsigned char* q = pool + (((long)p[0] * S + sy) * S + sx) * 3;
  v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2];
}

// one bilinear sample of a record's virtual canvas at output pixel (ox, oy), rounded and clamped to the uint8 range like
// cv2.warpAffine / warpPerspective's uint8 result
static __device__ __forceinline__ void warp_sample(const unsigned char* pool, const WarpSlot& w, int S, int ox, int oy, float (&px)[3]) {
  const float fx = (float)ox, fy = (float)oy;
  const float den = w.pinv[0] * fx + w.pinv[1] * fy + w.pinv[2];  // exactly 1 for an affine map
  const float u = (w.minv[0] * fx + w.minv[1] * fy + w.minv[2]) / den;
  const float vv = (w.minv[3] * fx + w.minv[4] * fy + w.minv[5]) / den;
  const float fu = floorf(u), fv = floorf(vv);
  const int x0 = (int)fu, y0 = (int)fv;
  const float ax = u - fu, ay = vv - fv;
  float c00[3], c01[3], c10[3], c11[3];
  warp_fetch(pool, w, S, x0, y0, c00);
  warp_fetch(pool, w, S, x0 + 1, y0, c01);
  warp_fetch(pool, w, S, x0, y0 + 1, c10);
  warp_fetch(pool, w, S, x0 + 1, y0 + 1, c11);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float top = c00[c] * (1.f - ax) + c01[c] * ax, bot = c10[c] * (1.f - ax) + c11[c] * ax;
    px[c] = fminf(fmaxf(rintf(top * (1.f - ay) + bot * ay), 0.f), 255.f);
  }
}

// slots: (N, 2) records -- the sample and its MixUp partner (read only when the sample's mix_r >= 0)
__global__ __launch_bounds__(256) void warp_import_kernel(const unsigned char* pool, const WarpSlot* slots, f16* y, int N, int S, int Cp) {
  const long per = (long)S * S, total = per * N;
  for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < total; pix += (long)gridDim.x * 256) {
    const int n = (int)(pix / per);
    const int r = (int)(pix - (long)n * per);
    int oy = r / S, ox = r - oy * S;
    const WarpSlot& w = slots[2 * n];
    if (w.flip & 1) ox = S - 1 - ox;  // RandomFlip follows MixUp: both images are mirrored together
    if (w.flip & 2) oy = S - 1 - oy;
    float px[3];
    warp_sample(pool, w, S, ox, oy, px);
    if (w.mix_r >= 0.0) {  // MixUp._mix_transform (augment.py:341): (img1 * r + img2 * (1 - r)).astype(np.uint8), float64, truncating
      float p2[3];
      warp_sample(pool, slots[2 * n + 1], S, ox, oy, p2);
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = (float)floor((double)px[c] * w.mix_r + (double)p2[c] * (1.0 - w.mix_r));
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)0.f;
    if (w.hsv[0] != 0.f) hsv_jitter(px, w.hsv[0], w.hsv[1], w.hsv[2]);  // RandomHSV follows the warp (and MixUp) in the reference too
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (f16)(px[c] / 255.f);
    *reinterpret_cast<half8*>(y + pix * Cp) = o;
    for (int c0 = 8; c0 < Cp; c0 += 8) *reinterpret_cast<half8*>(y + pix * Cp + c0) = half8{};
  }
}
extern "C" int dy_warp_import_u8(const void* pool, const void* slots, void* y, int n, int s, int cp, hipStream_t stream) {
  if ((cp & 7) || cp < 8) return DY_ERR_ALIGN;
  if (!pool || !slots || n <= 0) return DY_ERR_ARG;
  hipLaunchKernelGGL(warp_import_kernel, dim3(grid_for((long)n * s * s)), dim3(256), 0, stream, (const unsigned char*)pool, (const WarpSlot*)slots,
                     (f16*)y, n, s, cp);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_warp_slot_bytes(void) { return 2 * (int)sizeof(WarpSlot); }  // per sample: itself + its MixUp partner

// ---- generic 8-channel-granule element-wise kernels
struct EwArgs {
  const f16* a;
  const f16* b;
  const f16* c;
  f16* y;
  int lda, ldb, ldc, ldy, C;
  long npix;
};
// y = a (+ b) (+ c)
__global__ __launch_bounds__(256) void add_kernel(EwArgs e) {
  const int cpp = e.C >> 3;
  const long total = e.npix * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    half8 v = *reinterpret_cast<const half8*>(e.a + pix * e.lda + c0);
    if (e.b) {
      const half8 w = *reinterpret_cast<const half8*>(e.b + pix * e.ldb + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] + (float)w[j]);
    }
    if (e.c) {
      const half8 w = *reinterpret_cast<const half8*>(e.c + pix * e.ldc + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] + (float)w[j]);
    }
    *reinterpret_cast<half8*>(e.y + pix * e.ldy + c0) = v;
  }
}
extern "C" int dy_add(const void* a, int lda, const void* b, int ldb, const void* c, int ldc, void* y, int ldy,
                      long npix, int C, hipStream_t stream) {
  if ((C & 7) || (lda & 7) || (ldy & 7) || (b && (ldb & 7)) || (c && (ldc & 7))) return DY_ERR_ALIGN;
  EwArgs e{(const f16*)a, (const f16*)b, (const f16*)c, (f16*)y, lda, ldb, ldc, ldy, C, npix};
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(npix * (C >> 3))), dim3(256), 0, stream, e);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---- nearest 2x up-sampling, forward (y[2h+i, 2w+j] = x[h, w]) and backward (sum of the 2x2 block)
struct UpArgs {
  const f16* x;
  f16* y;
  int ldx, ldy, C, N, H, W, accumulate;  // H, W: low-resolution extent
};
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(UpArgs u) {
  const int cpp = u.C >> 3, Ho = 2 * u.H, Wo = 2 * u.W;
  const long total = (long)u.N * Ho * Wo * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int ox = (int)(pix % Wo);
    const long t = pix / Wo;
    const int oy = (int)(t % Ho);
    const long n = t / Ho;
    const long src = (n * u.H + (oy >> 1)) * u.W + (ox >> 1);
    *reinterpret_cast<uint4*>(u.y + pix * u.ldy + c0) = *reinterpret_cast<const uint4*>(u.x + src * u.ldx + c0);
  }
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(UpArgs u) {  // x = dY (high res), y = dX (low res)
  const int cpp = u.C >> 3, Ho = 2 * u.H, Wo = 2 * u.W;
  const long total = (long)u.N * u.H * u.W * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int lx = (int)(pix % u.W);
    const long t = pix / u.W;
    const int ly = (int)(t % u.H);
    const long n = t / u.H;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (u.accumulate) {
      const half8 o = *reinterpret_cast<const half8*>(u.y + pix * u.ldy + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = (float)o[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long src = (n * Ho + 2 * ly + (i >> 1)) * Wo + 2 * lx + (i & 1);
      const half8 v = *reinterpret_cast<const half8*>(u.x + src * u.ldx + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += (float)v[j];
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)s[j];
    *reinterpret_cast<half8*>(u.y + pix * u.ldy + c0) = o;
  }
}
extern "C" int dy_upsample2x(const void* x, int ldx, void* y, int ldy, int n, int h, int w, int C, int backward,
                             int accumulate, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return DY_ERR_ALIGN;
  UpArgs u{(const f16*)x, (f16*)y, ldx, ldy, C, n, h, w, accumulate};
  if (!backward)
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(grid_for((long)n * 4 * h * w * (C >> 3))), dim3(256), 0, stream, u);
  else
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(grid_for((long)n * h * w * (C >> 3))), dim3(256), 0, stream, u);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---- 5x5 / stride 1 / pad 2 max-pool with recorded arg-max (first maximum in row-major window order, as ATen)
struct PoolArgs {
  const f16* x;
  f16* y;
  uint8_t* arg;  // [N*H*W][C] window position 0..24
  const f16* dy;
  int ldx, ldy, lddy, C, N, H, W, accumulate;
};
__global__ __launch_bounds__(256) void maxpool5_fwd_kernel(PoolArgs a) {
  const int cpp = a.C >> 3;
  const long total = (long)a.N * a.H * a.W * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int x0 = (int)(pix % a.W);
    const long t = pix / a.W;
    const int y0 = (int)(t % a.H);
    const long n = t / a.H;
    float best[8];
    int bi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      best[j] = -INFINITY;
      bi[j] = 0;
    }
    // one window row (5 independent 16-byte loads, clamped address + validity flag) in flight at a time: the branchy
    // load-compare-load form ran 25 dependent L2 round trips per thread (40 us for a 6.5 MB map)
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      const int yy = y0 + dy - 2;
      const bool yok = yy >= 0 && yy < a.H;
      const int yc = yok ? yy : y0;
      half8 v[5];
      bool ok[5];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int xx = x0 + dx - 2;
        ok[dx] = yok && xx >= 0 && xx < a.W;
        v[dx] = *reinterpret_cast<const half8*>(a.x + ((n * a.H + yc) * a.W + (ok[dx] ? xx : x0)) * a.ldx + c0);
      }
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        if (!ok[dx]) continue;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[dx][j];
          if (f > best[j] || f != f) {
            best[j] = f;
            bi[j] = dy * 5 + dx;
          }
        }
      }
    }
    half8 o;
    uint8_t ai[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = (f16)best[j];
      ai[j] = (uint8_t)bi[j];
    }
    *reinterpret_cast<half8*>(a.y + pix * a.ldy + c0) = o;
    if (a.arg) *reinterpret_cast<uint2*>(a.arg + pix * a.C + c0) = *reinterpret_cast<uint2*>(ai);
  }
}
// dX[p] (+)= sum over the <=25 windows w containing p with argmax(w) == p of dY[w]   (gather form, no atomics)
__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(PoolArgs a) {
  const int cpp = a.C >> 3;
  const long total = (long)a.N * a.H * a.W * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int x0 = (int)(pix % a.W);
    const long t = pix / a.W;
    const int y0 = (int)(t % a.H);
    const long n = t / a.H;
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (a.accumulate) {
      const half8 o = *reinterpret_cast<const half8*>(a.y + pix * a.ldy + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = (float)o[j];
    }
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      const int wy = y0 - (dy - 2);  // window centre whose tap (dy,dx) lands on this pixel
      const bool yok = wy >= 0 && wy < a.H;
      const int yc = yok ? wy : y0;
      uint2 raw[5];
      half8 g[5];
      bool ok[5];
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {  // ten independent loads per window row in flight
        const int wx = x0 - (dx - 2);
        ok[dx] = yok && wx >= 0 && wx < a.W;
        const long wp = (n * a.H + yc) * a.W + (ok[dx] ? wx : x0);
        raw[dx] = *reinterpret_cast<const uint2*>(a.arg + wp * a.C + c0);
        g[dx] = *reinterpret_cast<const half8*>(a.dy + wp * a.lddy + c0);
      }
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        if (!ok[dx]) continue;
        const uint8_t* ai = reinterpret_cast<const uint8_t*>(&raw[dx]);
        const int code = dy * 5 + dx;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (ai[j] == code) s[j] += (float)g[dx][j];
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)s[j];
    *reinterpret_cast<half8*>(a.y + pix * a.ldy + c0) = o;
  }
}
// ---- SPPF's chain of three 5x5 pools (reference nn/modules/block.py:166-171: y1 = m(x), y2 = m(y1), y3 = m(y2), all four concatenated)
// in ONE launch when a whole map fits in LDS: a workgroup owns one image x CG channels, keeps the map in LDS and pools it three times
// in place.  Each pool is separable -- a row pass (max over dx, first maximum kept) and a column pass over the row results (first row
// kept) -- which is the row-major "first maximum" of the 25-tap scan above, NaN rule included (a NaN always replaces, so the last
// NaN in row-major order wins in both forms); 10 LDS reads per output instead of 25 L1 reads.  x is read once and y1..y3 + the three
// arg-max maps are written once: 1 + 3 + 1.5 tensor units instead of 3 x (1 + 1 + 0.5), and one launch instead of three.
// 16 waves per workgroup: the passes are VALU work (compare / select chains) on LDS operands, and one workgroup owns the whole CU's LDS --
// with 4 waves (one per SIMD) every LDS wait stalled its SIMD and the fused form was SLOWER than the three launches (13.08 vs 13.00 ms / step)
#define SPPF_THREADS 1024
struct SppfArgs {
  f16* cat;            // [N*H*W][ld]: slice 0 = x (C channels), slices 1..3 = y1..y3 (forward); the gradients of the same (backward)
  uint8_t* arg[3];     // [N*H*W][C] window position 0..24 of pool l
  int ld, C, N, H, W, acc[3];
};
template <int CG>
__global__ __launch_bounds__(SPPF_THREADS) void sppf_pool3_fwd_kernel(SppfArgs a) {
  constexpr int G = CG / 8;
  extern __shared__ __attribute__((aligned(16))) char sppf_lds[];
  const int HW = a.H * a.W, items = HW * G;
  half8* cur = reinterpret_cast<half8*>(sppf_lds);          // the map being pooled; overwritten by its pooled form
  half8* rowm = cur + items;                                 // row-pass maxima
  uint2* rowa = reinterpret_cast<uint2*>(rowm + items);      // row-pass arg (dx) per channel
  const int groups = a.C / CG, n = blockIdx.x / groups, c0 = (blockIdx.x - n * groups) * CG;
  f16* const base = a.cat + (size_t)n * HW * a.ld + c0;
  for (int i = threadIdx.x; i < items; i += SPPF_THREADS) {
    const int px = i / G, part = i - px * G;
    cur[i] = *reinterpret_cast<const half8*>(base + (size_t)px * a.ld + part * 8);
  }
  __syncthreads();
  for (int l = 0; l < 3; ++l) {
    for (int i = threadIdx.x; i < items; i += SPPF_THREADS) {
      const int px = i / G, part = i - px * G, y = px / a.W, x = px - y * a.W;
      float best[8];
      uint8_t bd[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bd[j] = 0; }
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {
        const int xx = x + dx - 2;
        if (xx < 0 || xx >= a.W) continue;
        const half8 v = cur[(y * a.W + xx) * G + part];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[j];
          if (f > best[j] || f != f) { best[j] = f; bd[j] = (uint8_t)dx; }
        }
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f16)best[j];
      rowm[i] = o;
      rowa[i] = *reinterpret_cast<const uint2*>(bd);
    }
    __syncthreads();
    f16* const yb = base + (size_t)(l + 1) * a.C;
    uint8_t* const ab = a.arg[l] ? a.arg[l] + (size_t)n * HW * a.C + c0 : nullptr;
    for (int i = threadIdx.x; i < items; i += SPPF_THREADS) {
      const int px = i / G, part = i - px * G, y = px / a.W, x = px - y * a.W;
      float best[8];
      uint8_t bi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
#pragma unroll
      for (int dy = 0; dy < 5; ++dy) {
        const int yy = y + dy - 2;
        if (yy < 0 || yy >= a.H) continue;
        const int k = (yy * a.W + x) * G + part;
        const half8 v = rowm[k];
        const uint2 ra = rowa[k];
        const uint8_t* rd = reinterpret_cast<const uint8_t*>(&ra);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = (float)v[j];
          if (f > best[j] || f != f) { best[j] = f; bi[j] = (uint8_t)(dy * 5 + rd[j]); }
        }
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f16)best[j];
      cur[i] = o;  // only this thread reads or writes cur[i] in this pass: the column pass reads the row results
      *reinterpret_cast<half8*>(yb + (size_t)px * a.ld + part * 8) = o;
      if (ab) *reinterpret_cast<uint2*>(ab + (size_t)px * a.C + part * 8) = *reinterpret_cast<const uint2*>(bi);
    }
    __syncthreads();
  }
}
// The backward chain of the same three pools in one launch: g2 = dcat2 + poolbwd(dcat3), g1 = dcat1 + poolbwd(g2), dcat0 += poolbwd(g1),
// gather form and summation order of maxpool5_bwd_kernel, each level rounded to fp16 as its stand-alone launch would store it.  The
// intermediate gradients never leave LDS (nothing else reads the gradients of y1..y3 once cv2's input gradient has been chained through).
template <int CG>
__global__ __launch_bounds__(SPPF_THREADS) void sppf_pool3_bwd_kernel(SppfArgs a) {
  constexpr int G = CG / 8;
  extern __shared__ __attribute__((aligned(16))) char sppf_lds[];
  const int HW = a.H * a.W, items = HW * G;
  half8* g0 = reinterpret_cast<half8*>(sppf_lds);
  half8* g1 = g0 + items;
  uint2* ar = reinterpret_cast<uint2*>(g1 + items);
  const int groups = a.C / CG, n = blockIdx.x / groups, c0 = (blockIdx.x - n * groups) * CG;
  f16* const base = a.cat + (size_t)n * HW * a.ld + c0;
  for (int i = threadIdx.x; i < items; i += SPPF_THREADS) {
    const int px = i / G, part = i - px * G;
    g0[i] = *reinterpret_cast<const half8*>(base + (size_t)3 * a.C + (size_t)px * a.ld + part * 8);
  }
  for (int l = 2; l >= 0; --l) {
    const uint8_t* const ab = a.arg[l] + (size_t)n * HW * a.C + c0;
    for (int i = threadIdx.x; i < items; i += SPPF_THREADS) {
      const int px = i / G, part = i - px * G;
      ar[i] = *reinterpret_cast<const uint2*>(ab + (size_t)px * a.C + part * 8);
    }
    __syncthreads();
    f16* const tb = base + (size_t)l * a.C;
    for (int i = threadIdx.x; i < items; i += SPPF_THREADS) {
      const int px = i / G, part = i - px * G, y = px / a.W, x = px - y * a.W;
      float s[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] = 0.f;
      if (a.acc[l]) {
        const half8 o = *reinterpret_cast<const half8*>(tb + (size_t)px * a.ld + part * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = (float)o[j];
      }
#pragma unroll
      for (int dy = 0; dy < 5; ++dy) {
        const int wy = y - (dy - 2);
        if (wy < 0 || wy >= a.H) continue;
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) {
          const int wx = x - (dx - 2);
          if (wx < 0 || wx >= a.W) continue;
          const int k = (wy * a.W + wx) * G + part;
          const uint2 raw = ar[k];
          const half8 g = g0[k];
          const uint8_t* ai = reinterpret_cast<const uint8_t*>(&raw);
          const int code = dy * 5 + dx;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (ai[j] == code) s[j] += (float)g[j];
        }
      }
      half8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f16)s[j];
      if (l > 0) g1[i] = o;
      else *reinterpret_cast<half8*>(tb + (size_t)px * a.ld + part * 8) = o;
    }
    __syncthreads();
    half8* t = g0; g0 = g1; g1 = t;
  }
}
// channels per workgroup the fused SPPF pools would use for this map (16 or 8), 0 = the map does not fit in LDS: use dy_maxpool5
extern "C" int dy_sppf_pool3_supported(int h, int w, int C) {
  if (h < 1 || w < 1 || C < 8 || (C & 7)) return 0;
  const long px = (long)h * w;
  if (!(C & 15) && px * (2 * 32 + 16) <= 156 * 1024) return 16;
  if (px * (2 * 16 + 8) <= 156 * 1024) return 8;
  return 0;
}
extern "C" int dy_sppf_pool3(void* cat, int ld, int C, void* arg0, void* arg1, void* arg2, int n, int h, int w, hipStream_t stream) {
  const int cg = dy_sppf_pool3_supported(h, w, C);
  if (!cg || !cat || ld < 4 * C) return DY_ERR_ARG;
  if ((ld & 7) || ((uintptr_t)cat & 15)) return DY_ERR_ALIGN;
  SppfArgs a{(f16*)cat, {(uint8_t*)arg0, (uint8_t*)arg1, (uint8_t*)arg2}, ld, C, n, h, w, {0, 0, 0}};
  const size_t lds = (size_t)h * w * (cg / 8) * (2 * 16 + 8);
  static bool attr[2] = {false, false};
  if (cg == 16) {
    if (!attr[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool3_fwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return DY_ERR_LAUNCH; attr[0] = true; }
    hipLaunchKernelGGL(sppf_pool3_fwd_kernel<16>, dim3(n * (C / 16)), dim3(SPPF_THREADS), lds, stream, a);
  } else {
    if (!attr[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool3_fwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return DY_ERR_LAUNCH; attr[1] = true; }
    hipLaunchKernelGGL(sppf_pool3_fwd_kernel<8>, dim3(n * (C / 8)), dim3(SPPF_THREADS), lds, stream, a);
  }
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_sppf_pool3_backward(void* gcat, int ld, int C, const void* arg0, const void* arg1, const void* arg2, int n, int h,
                                      int w, int acc0, int acc1, int acc2, hipStream_t stream) {
  const int cg = dy_sppf_pool3_supported(h, w, C);
  if (!cg || !gcat || !arg0 || !arg1 || !arg2 || ld < 4 * C) return DY_ERR_ARG;
  if ((ld & 7) || ((uintptr_t)gcat & 15)) return DY_ERR_ALIGN;
  SppfArgs a{(f16*)gcat, {(uint8_t*)arg0, (uint8_t*)arg1, (uint8_t*)arg2}, ld, C, n, h, w, {acc0, acc1, acc2}};
  const size_t lds = (size_t)h * w * (cg / 8) * (2 * 16 + 8);
  static bool attr[2] = {false, false};
  if (cg == 16) {
    if (!attr[0]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool3_bwd_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return DY_ERR_LAUNCH; attr[0] = true; }
    hipLaunchKernelGGL(sppf_pool3_bwd_kernel<16>, dim3(n * (C / 16)), dim3(SPPF_THREADS), lds, stream, a);
  } else {
    if (!attr[1]) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool3_bwd_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess) return DY_ERR_LAUNCH; attr[1] = true; }
    hipLaunchKernelGGL(sppf_pool3_bwd_kernel<8>, dim3(n * (C / 8)), dim3(SPPF_THREADS), lds, stream, a);
  }
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_maxpool5(const void* x, int ldx, void* y, int ldy, void* argmax, int n, int h, int w, int C,
                           hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return DY_ERR_ALIGN;
  PoolArgs a{(const f16*)x, (f16*)y, (uint8_t*)argmax, nullptr, ldx, ldy, 0, C, n, h, w, 0};
  hipLaunchKernelGGL(maxpool5_fwd_kernel, dim3(grid_for((long)n * h * w * (C >> 3))), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_maxpool5_backward(const void* dy, int lddy, const void* argmax, void* dx, int lddx, int n, int h,
                                    int w, int C, int accumulate, hipStream_t stream) {
  if ((C & 7) || (lddy & 7) || (lddx & 7)) return DY_ERR_ALIGN;
  PoolArgs a{nullptr, (f16*)dx, (uint8_t*)argmax, (const f16*)dy, 0, lddx, lddy, C, n, h, w, accumulate};
  hipLaunchKernelGGL(maxpool5_bwd_kernel, dim3(grid_for((long)n * h * w * (C >> 3))), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---- ScalSeq tail (reference nn/extra_modules/block.py:3437-3443): the Conv3d(1x1x1)+bias outputs r0 (full res),
// r1 (1/2), r2 (1/4) are normalised with the shared BatchNorm3d coefficients, LeakyReLU(0.1), max over the three
// scales with nearest up-sampling done by indexing; optional fused `Add` of a residual map (:3483-3484).
struct SsArgs {
  const f16* r[3];
  int ld[3];
  const f16* res;
  int ldres;
  f16* y;
  int ldy;
  const float* coef;
  int C, N, H, W;
};
__global__ __launch_bounds__(256) void scalseq_tail_kernel(SsArgs a) {
  const int cpp = a.C >> 3;
  const long total = (long)a.N * a.H * a.W * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int x0 = (int)(pix % a.W);
    const long t = pix / a.W;
    const int y0 = (int)(t % a.H);
    const long n = t / a.H;
    float best[8];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const int hl = a.H >> l, wl = a.W >> l;
      const long src = (n * hl + (y0 >> l)) * wl + (x0 >> l);
      const half8 v = *reinterpret_cast<const half8*>(a.r[l] + src * a.ld[l] + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = (float)v[j] * a.coef[c0 + j] + a.coef[a.C + c0 + j];
        z = z > 0.f ? z : 0.1f * z;
        best[j] = (l == 0) ? z : fmaxf(best[j], z);
      }
    }
    half8 o;
    if (a.res) {
      const half8 rv = *reinterpret_cast<const half8*>(a.res + pix * a.ldres + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) best[j] = (float)(f16)best[j] + (float)rv[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)best[j];
    *reinterpret_cast<half8*>(a.y + pix * a.ldy + c0) = o;
  }
}
extern "C" int dy_scalseq_tail(const void* r0, int ld0, const void* r1, int ld1, const void* r2, int ld2,
                               const void* res, int ldres, void* y, int ldy, const float* coef, int n, int h, int w,
                               int C, hipStream_t stream) {
  if ((C & 7) || (ld0 & 7) || (ld1 & 7) || (ld2 & 7) || (ldy & 7) || (h & 3) || (w & 3)) return DY_ERR_ALIGN;
  SsArgs a{{(const f16*)r0, (const f16*)r1, (const f16*)r2}, {ld0, ld1, ld2}, (const f16*)res, ldres, (f16*)y, ldy,
           coef, C, n, h, w};
  hipLaunchKernelGGL(scalseq_tail_kernel, dim3(grid_for((long)n * h * w * (C >> 3))), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// Backward of the tail.  Level l (block edge s = 1 << l) owns one low-resolution pixel per thread-granule and visits
// the s*s full-resolution positions it was replicated to.  g_l(pos) = dY(pos) * leaky'(z_l) where level l is the
// arg-max at pos (first maximum wins, as max_pool3d), else 0.
//   mode 0: partial sums of g and g*xhat over the whole (B,3,H,W) volume  -> partials [grid][2][C]
//   mode 1: dr_l = scale * (sum_block g_l - cnt*mean_g - cnt*xhat_l*mean_gxhat)
struct SsBwdArgs {
  const f16* r[3];
  int ld[3];
  const f16* dy;
  int lddy;
  f16* dr;
  int lddr;
  const float* coef;
  const float* bwdcoef;
  float* partials;
  int C, N, H, W, level, mode;
};
__global__ __launch_bounds__(256) void scalseq_bwd_kernel(SsBwdArgs a) {
  const int cpp = a.C >> 3, l = a.level, s = 1 << l;
  const int hl = a.H >> l, wl = a.W >> l;
  const long total = (long)a.N * hl * wl * cpp;
  float ps[8], px[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) ps[j] = px[j] = 0.f;
  // a thread keeps one channel granule for the whole grid-stride loop so that mode-0 sums stay per channel
  const int part = threadIdx.x % cpp, c0 = part * 8;
  const int rows = 256 / cpp, row = threadIdx.x / cpp;
  const long npix_l = (long)a.N * hl * wl;
  if (row < rows) {
    for (long lp = (long)blockIdx.x * rows + row; lp < npix_l; lp += (long)gridDim.x * rows) {
      const int lx = (int)(lp % wl);
      const long t = lp / wl;
      const int ly = (int)(t % hl);
      const long n = t / hl;
      const half8 own = *reinterpret_cast<const half8*>(a.r[l] + lp * a.ld[l] + c0);
      float gsum[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) gsum[j] = 0.f;
      for (int i = 0; i < s * s; ++i) {
        const int y0 = ly * s + i / s, x0 = lx * s + i % s;
        const long pos = (n * a.H + y0) * a.W + x0;
        const half8 g = *reinterpret_cast<const half8*>(a.dy + pos * a.lddy + c0);
        half8 v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          if (k == l) {
            v[k] = own;
          } else {
            const int hk = a.H >> k, wk = a.W >> k;
            v[k] = *reinterpret_cast<const half8*>(a.r[k] + ((n * hk + (y0 >> k)) * wk + (x0 >> k)) * a.ld[k] + c0);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float sc = a.coef[c0 + j], sh = a.coef[a.C + c0 + j];
          float z[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float zz = (float)v[k][j] * sc + sh;
            z[k] = zz > 0.f ? zz : 0.1f * zz;
          }
          int am = 0;
          if (z[1] > z[am]) am = 1;
          if (z[2] > z[am]) am = 2;
          if (am == l) {
            const float zz = (float)own[j] * sc + sh;
            gsum[j] += (float)g[j] * (zz > 0.f ? 1.f : 0.1f);
          }
        }
      }
      if (a.mode == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)own[j] - a.coef[2 * a.C + c0 + j]) * a.coef[3 * a.C + c0 + j];
          ps[j] += gsum[j];
          px[j] += gsum[j] * xh;
        }
      } else {
        half8 o;
        const float cnt = (float)(s * s);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)own[j] - a.coef[2 * a.C + c0 + j]) * a.coef[3 * a.C + c0 + j];
          o[j] = (f16)(a.coef[c0 + j] * (gsum[j] - cnt * a.bwdcoef[c0 + j] - cnt * xh * a.bwdcoef[a.C + c0 + j]));
        }
        *reinterpret_cast<half8*>(a.dr + lp * a.lddr + c0) = o;
      }
    }
  }
  if (a.mode == 0) {
    __shared__ float red[2][256][9];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[0][threadIdx.x][j] = ps[j];
      red[1][threadIdx.x][j] = px[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
      const int which = i / a.C, c = i - which * a.C, pp = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int r = 0; r < rows; ++r) sum += red[which][r * cpp + pp][j];
      a.partials[((size_t)blockIdx.x * 2 + which) * a.C + c] = sum;
    }
  }
  (void)total;
}
extern "C" int dy_scalseq_tail_backward(const void* r0, int ld0, const void* r1, int ld1, const void* r2, int ld2,
                                        const void* dy, int lddy, void* dr, int lddr, const float* coef,
                                        const float* bwdcoef, float* partials, int max_partials, int n, int h, int w,
                                        int C, int level, int mode, int* nparts, hipStream_t stream) {
  if ((C & 7) || (C >> 3) > 256 || (h & 3) || (w & 3) || level < 0 || level > 2) return DY_ERR_ARG;
  SsBwdArgs a{{(const f16*)r0, (const f16*)r1, (const f16*)r2}, {ld0, ld1, ld2}, (const f16*)dy, lddy, (f16*)dr, lddr,
              coef, bwdcoef, partials, C, n, h, w, level, mode};
  const int rows = 256 / (C >> 3);
  long blocks = ((long)n * (h >> level) * (w >> level) + rows - 1) / rows;
  if (blocks > 1024) blocks = 1024;
  if (mode == 0 && blocks > max_partials) blocks = max_partials;
  if (blocks < 1) blocks = 1;
  if (nparts) *nparts = (int)blocks;
  hipLaunchKernelGGL(scalseq_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// All three levels in one pass: a thread owns one level-2 pixel (a 4x4 block of full-resolution positions, four level-1
// pixels) of one 8-channel granule, so dY and r0/r1/r2 are read once per mode instead of once per (level, mode) --
// six launches and ~2.9 GB of reads become two launches and ~1.0 GB for the P2 ScalSeq of DEAL-YOLO-N.
struct SsBwdAllArgs {
  const f16* r[3];
  int ld[3];
  const f16* dy;
  int lddy;
  f16* dr[3];
  int lddr[3];
  const float* coef;
  const float* bwdcoef;
  float* partials;
  int C, N, H, W, mode;
};
__global__ __launch_bounds__(256) void scalseq_bwd_all_kernel(SsBwdAllArgs a) {
  const int cpp = a.C >> 3;
  const int h1 = a.H >> 1, w1 = a.W >> 1, h2 = a.H >> 2, w2 = a.W >> 2;
  const int part = threadIdx.x % cpp, c0 = part * 8;
  const int rows = 256 / cpp, row = threadIdx.x / cpp;
  const long npix2 = (long)a.N * h2 * w2;
  float sc[8], sh[8], mean[8], inv[8], mg[8], mgx[8], ps[8], px[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.coef[c0 + j];
    sh[j] = a.coef[a.C + c0 + j];
    mean[j] = a.coef[2 * a.C + c0 + j];
    inv[j] = a.coef[3 * a.C + c0 + j];
    mg[j] = a.mode ? a.bwdcoef[c0 + j] : 0.f;
    mgx[j] = a.mode ? a.bwdcoef[a.C + c0 + j] : 0.f;
    ps[j] = px[j] = 0.f;
  }
  if (row < rows) {
    for (long lp2 = (long)blockIdx.x * rows + row; lp2 < npix2; lp2 += (long)gridDim.x * rows) {
      const int lx2 = (int)(lp2 % w2);
      const long t = lp2 / w2;
      const int ly2 = (int)(t % h2);
      const long n = t / h2;
      const half8 own2 = *reinterpret_cast<const half8*>(a.r[2] + lp2 * a.ld[2] + c0);
      float z2[8], gs2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float zz = (float)own2[j] * sc[j] + sh[j];
        z2[j] = zz > 0.f ? zz : 0.1f * zz;
        gs2[j] = 0.f;
      }
      for (int q1 = 0; q1 < 4; ++q1) {
        const int ly1 = ly2 * 2 + (q1 >> 1), lx1 = lx2 * 2 + (q1 & 1);
        const long lp1 = (n * h1 + ly1) * w1 + lx1;
        const half8 own1 = *reinterpret_cast<const half8*>(a.r[1] + lp1 * a.ld[1] + c0);
        float z1[8], gs1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float zz = (float)own1[j] * sc[j] + sh[j];
          z1[j] = zz > 0.f ? zz : 0.1f * zz;
          gs1[j] = 0.f;
        }
#pragma unroll
        for (int q0 = 0; q0 < 4; ++q0) {
          const int y0 = ly1 * 2 + (q0 >> 1), x0 = lx1 * 2 + (q0 & 1);
          const long pos = (n * a.H + y0) * a.W + x0;
          const half8 own0 = *reinterpret_cast<const half8*>(a.r[0] + pos * a.ld[0] + c0);
          const half8 g = *reinterpret_cast<const half8*>(a.dy + pos * a.lddy + c0);
          half8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float zz0 = (float)own0[j] * sc[j] + sh[j];
            const float z0 = zz0 > 0.f ? zz0 : 0.1f * zz0;
            // first maximum wins (max_pool3d over the depth axis); leaky(z) > 0 <=> z > 0, so the slope follows the winner's sign
            int am = 0;
            float zb = z0;
            if (z1[j] > zb) { am = 1; zb = z1[j]; }
            if (z2[j] > zb) { am = 2; zb = z2[j]; }
            const float gg = (float)g[j] * (zb > 0.f ? 1.f : 0.1f);
            const float g0 = am == 0 ? gg : 0.f;
            gs1[j] += am == 1 ? gg : 0.f;
            gs2[j] += am == 2 ? gg : 0.f;
            const float xh0 = ((float)own0[j] - mean[j]) * inv[j];
            if (a.mode == 0) {
              ps[j] += g0;
              px[j] += g0 * xh0;
            } else {
              o[j] = (f16)(sc[j] * (g0 - mg[j] - xh0 * mgx[j]));
            }
          }
          if (a.mode) *reinterpret_cast<half8*>(a.dr[0] + pos * a.lddr[0] + c0) = o;
        }
        half8 o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = ((float)own1[j] - mean[j]) * inv[j];
          if (a.mode == 0) {
            ps[j] += gs1[j];
            px[j] += gs1[j] * xh;
          } else {
            o1[j] = (f16)(sc[j] * (gs1[j] - 4.f * mg[j] - 4.f * xh * mgx[j]));
          }
        }
        if (a.mode) *reinterpret_cast<half8*>(a.dr[1] + lp1 * a.lddr[1] + c0) = o1;
      }
      half8 o2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = ((float)own2[j] - mean[j]) * inv[j];
        if (a.mode == 0) {
          ps[j] += gs2[j];
          px[j] += gs2[j] * xh;
        } else {
          o2[j] = (f16)(sc[j] * (gs2[j] - 16.f * mg[j] - 16.f * xh * mgx[j]));
        }
      }
      if (a.mode) *reinterpret_cast<half8*>(a.dr[2] + lp2 * a.lddr[2] + c0) = o2;
    }
  }
  if (a.mode == 0) {
    __shared__ float red[2][256][9];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[0][threadIdx.x][j] = ps[j];
      red[1][threadIdx.x][j] = px[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
      const int which = i / a.C, c = i - which * a.C, pp = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int r = 0; r < rows; ++r) sum += red[which][r * cpp + pp][j];
      a.partials[((size_t)blockIdx.x * 2 + which) * a.C + c] = sum;
    }
  }
}
// The same pass with a wave = (64 / cpp) adjacent COLUMNS x the 4 rows of one level-2 row: a lane owns one column (one 8-channel granule
// of it), so every load of r0 / dY / dr0 is a row of consecutive pixels across the wave -- full lines -- where the kernel above gave a
// lane a 4x4 block and touched 16 half-used lines per instruction (2.1 TB/s in the statistics pass).  The level-1 / level-2 sums meet by
// shuffle (column pairs: lane ^ cpp; column quads: + lane ^ 2 cpp); the lane of the first column of a pair / quad adds them to the
// statistics or stores dr1 / dr2.  Needs cpp | 64 and W a multiple of 64 / cpp; the kernel above takes every other shape.
__global__ __launch_bounds__(256) void scalseq_bwd_cols_kernel(SsBwdAllArgs a) {
  const int cpp = a.C >> 3, PW = 64 / cpp;
  const int h1 = a.H >> 1, w1 = a.W >> 1, h2 = a.H >> 2, w2 = a.W >> 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int part = lane % cpp, col = lane / cpp, c0 = part * 8;
  float sc[8], sh[8], mean[8], inv[8], mg[8], mgx[8], ps[8], px[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.coef[c0 + j];
    sh[j] = a.coef[a.C + c0 + j];
    mean[j] = a.coef[2 * a.C + c0 + j];
    inv[j] = a.coef[3 * a.C + c0 + j];
    mg[j] = a.mode ? a.bwdcoef[c0 + j] : 0.f;
    mgx[j] = a.mode ? a.bwdcoef[a.C + c0 + j] : 0.f;
    ps[j] = px[j] = 0.f;
  }
  const int chunks = a.W / PW;
  const long items = (long)a.N * h2 * chunks;
  for (long it = (long)blockIdx.x * 4 + wave; it < items; it += (long)gridDim.x * 4) {
    const int xc = (int)(it % chunks);
    const long t = it / chunks;
    const int y2 = (int)(t % h2);
    const long n = t / h2;
    const int x = xc * PW + col;
    const long lp2 = (n * h2 + y2) * w2 + (x >> 2);
    const half8 own2 = *reinterpret_cast<const half8*>(a.r[2] + lp2 * a.ld[2] + c0);
    float z2[8], gs2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float zz = (float)own2[j] * sc[j] + sh[j];
      z2[j] = zz > 0.f ? zz : 0.1f * zz;
      gs2[j] = 0.f;
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int y1 = y2 * 2 + hf;
      const long lp1 = (n * h1 + y1) * w1 + (x >> 1);
      const half8 own1 = *reinterpret_cast<const half8*>(a.r[1] + lp1 * a.ld[1] + c0);
      float z1[8], gs1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float zz = (float)own1[j] * sc[j] + sh[j];
        z1[j] = zz > 0.f ? zz : 0.1f * zz;
        gs1[j] = 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const long pos = (n * a.H + (y1 * 2 + rr)) * a.W + x;
        const half8 own0 = *reinterpret_cast<const half8*>(a.r[0] + pos * a.ld[0] + c0);
        const half8 g = *reinterpret_cast<const half8*>(a.dy + pos * a.lddy + c0);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float zz0 = (float)own0[j] * sc[j] + sh[j];
          const float z0 = zz0 > 0.f ? zz0 : 0.1f * zz0;
          int am = 0;  // first maximum wins (max_pool3d over the depth axis), as above
          float zb = z0;
          if (z1[j] > zb) { am = 1; zb = z1[j]; }
          if (z2[j] > zb) { am = 2; zb = z2[j]; }
          const float gg = (float)g[j] * (zb > 0.f ? 1.f : 0.1f);
          const float g0 = am == 0 ? gg : 0.f;
          gs1[j] += am == 1 ? gg : 0.f;
          gs2[j] += am == 2 ? gg : 0.f;
          const float xh0 = ((float)own0[j] - mean[j]) * inv[j];
          if (a.mode == 0) {
            ps[j] += g0;
            px[j] += g0 * xh0;
          } else {
            o[j] = (f16)(sc[j] * (g0 - mg[j] - xh0 * mgx[j]));
          }
        }
        if (a.mode) *reinterpret_cast<half8*>(a.dr[0] + pos * a.lddr[0] + c0) = o;
      }
      half8 o1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // the column pair of this level-1 pixel.  Four parts per pixel (C = 32): a column is a quad and the lane that uses the total is
        // in the pair's first column, so "lane + 4 of my row" is the partner (one DPP add); other widths take the shuffle
        const float tot = gs1[j] + (cpp == 4 ? dpp_read<DY_DPP_ROW_SHL(4)>(gs1[j]) : __shfl_xor(gs1[j], cpp, 64));
        const float xh = ((float)own1[j] - mean[j]) * inv[j];
        if (a.mode == 0) {
          if (!(col & 1)) { ps[j] += tot; px[j] += tot * xh; }
        } else {
          o1[j] = (f16)(sc[j] * (tot - 4.f * mg[j] - 4.f * xh * mgx[j]));
        }
      }
      if (a.mode && !(col & 1)) *reinterpret_cast<half8*>(a.dr[1] + lp1 * a.lddr[1] + c0) = o1;
    }
    half8 o2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float tot;  // the four columns of this level-2 pixel (cpp == 4: one DPP row; its first quad ends with the sum of all four)
      if (cpp == 4) {
        tot = gs2[j] + dpp_read<DY_DPP_ROW_SHL(4)>(gs2[j]);
        tot += dpp_read<DY_DPP_ROW_SHL(8)>(tot);
      } else {
        tot = gs2[j] + __shfl_xor(gs2[j], cpp, 64);
        tot += __shfl_xor(tot, 2 * cpp, 64);
      }
      const float xh = ((float)own2[j] - mean[j]) * inv[j];
      if (a.mode == 0) {
        if (!(col & 3)) { ps[j] += tot; px[j] += tot * xh; }
      } else {
        o2[j] = (f16)(sc[j] * (tot - 16.f * mg[j] - 16.f * xh * mgx[j]));
      }
    }
    if (a.mode && !(col & 3)) *reinterpret_cast<half8*>(a.dr[2] + lp2 * a.lddr[2] + c0) = o2;
  }
  if (a.mode == 0) {
    __shared__ float red[2][256][9];
    const int rows = 256 / cpp;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[0][tid][j] = ps[j];
      red[1][tid][j] = px[j];
    }
    __syncthreads();
    for (int i = tid; i < 2 * a.C; i += 256) {
      const int which = i / a.C, c = i - which * a.C, pp = c >> 3, j = c & 7;
      float sum = 0.f;
      for (int r = 0; r < rows; ++r) sum += red[which][r * cpp + pp][j];  // thread index = (wave * 64 / cpp + column) * cpp + part
      a.partials[((size_t)blockIdx.x * 2 + which) * a.C + c] = sum;
    }
  }
}
extern "C" int dy_scalseq_tail_backward_all(const void* r0, int ld0, const void* r1, int ld1, const void* r2, int ld2,
                                            const void* dy, int lddy, void* dr0, int lddr0, void* dr1, int lddr1,
                                            void* dr2, int lddr2, const float* coef, const float* bwdcoef,
                                            float* partials, int max_partials, int n, int h, int w, int C, int mode,
                                            int* nparts, hipStream_t stream) {
  if ((C & 7) || (C >> 3) > 256 || (h & 3) || (w & 3) || (ld0 & 7) || (ld1 & 7) || (ld2 & 7) || (lddy & 7)) return DY_ERR_ARG;
  if (mode && ((lddr0 & 7) || (lddr1 & 7) || (lddr2 & 7) || !dr0 || !dr1 || !dr2 || !bwdcoef)) return DY_ERR_ARG;
  if (!mode && !partials) return DY_ERR_ARG;
  SsBwdAllArgs a{{(const f16*)r0, (const f16*)r1, (const f16*)r2}, {ld0, ld1, ld2}, (const f16*)dy, lddy,
                 {(f16*)dr0, (f16*)dr1, (f16*)dr2}, {lddr0, lddr1, lddr2}, coef, bwdcoef, partials, C, n, h, w, mode};
  const int rows = 256 / (C >> 3);
  long blocks = ((long)n * (h >> 2) * (w >> 2) + rows - 1) / rows;
  if (blocks > 2048) blocks = 2048;
  if (mode == 0 && blocks > max_partials) blocks = max_partials;
  if (blocks < 1) blocks = 1;
  const int cpp = C >> 3;
  if (cpp <= 16 && 64 % cpp == 0 && w % (64 / cpp) == 0) {
    long wblocks = ((long)n * (h >> 2) * (w / (64 / cpp)) + 3) / 4;
    if (wblocks > 2048) wblocks = 2048;
    if (mode == 0 && wblocks > max_partials) wblocks = max_partials;
    if (wblocks < 1) wblocks = 1;
    if (nparts) *nparts = (int)wblocks;
    hipLaunchKernelGGL(scalseq_bwd_cols_kernel, dim3((int)wblocks), dim3(256), 0, stream, a);
    DY_CHECK_LAUNCH();
    return DY_OK;
  }
  if (nparts) *nparts = (int)blocks;
  hipLaunchKernelGGL(scalseq_bwd_all_kernel, dim3((int)blocks), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---- strided channel-slice copy (Concat fallback) and zero fill
__global__ __launch_bounds__(256) void copy_slice_kernel(EwArgs e) {
  const int cpp = e.C >> 3;
  const long total = e.npix * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    *reinterpret_cast<uint4*>(e.y + pix * e.ldy + c0) = *reinterpret_cast<const uint4*>(e.a + pix * e.lda + c0);
  }
}
extern "C" int dy_copy_slice(const void* x, int ldx, void* y, int ldy, long npix, int C, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return DY_ERR_ALIGN;
  EwArgs e{(const f16*)x, nullptr, nullptr, (f16*)y, ldx, 0, 0, ldy, C, npix};
  hipLaunchKernelGGL(copy_slice_kernel, dim3(grid_for(npix * (C >> 3))), dim3(256), 0, stream, e);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_fill_zero(void* p, size_t bytes, hipStream_t stream) {
  return hipMemsetAsync(p, 0, bytes, stream) == hipSuccess ? DY_OK : DY_ERR_LAUNCH;
}

// ---- Zoom_cat's fine-level branch (reference nn/extra_modules/block.py:3406-3412): adaptive_max_pool2d + adaptive_avg_pool2d
// to exactly half resolution == 2x2 max + 2x2 mean.  Backward: the max routes to its first arg-max (row-major, as ATen),
// the mean spreads 1/4.
struct ZpArgs {
  const f16* x;
  f16* y;
  const f16* dy;
  f16* dx;
  int ldx, ldy, lddy, lddx, C, N, H, W, accumulate;  // H, W: output (half) extent
};
__global__ __launch_bounds__(256) void zoom_pool_fwd_kernel(ZpArgs a) {
  const int cpp = a.C >> 3;
  const long total = (long)a.N * a.H * a.W * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int ox = (int)(pix % a.W);
    const long t = pix / a.W;
    const int oy = (int)(t % a.H);
    const long n = t / a.H;
    float mx[8], sm[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long src = (n * 2 * a.H + 2 * oy + (i >> 1)) * 2 * a.W + 2 * ox + (i & 1);
      const half8 v = *reinterpret_cast<const half8*>(a.x + src * a.ldx + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        mx[j] = i == 0 ? f : fmaxf(mx[j], f);
        sm[j] = i == 0 ? f : sm[j] + f;
      }
    }
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)((float)(f16)mx[j] + (float)(f16)(sm[j] * 0.25f));
    *reinterpret_cast<half8*>(a.y + pix * a.ldy + c0) = o;
  }
}
__global__ __launch_bounds__(256) void zoom_pool_bwd_kernel(ZpArgs a) {
  const int cpp = a.C >> 3;
  const long total = (long)a.N * a.H * a.W * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const int ox = (int)(pix % a.W);
    const long t = pix / a.W;
    const int oy = (int)(t % a.H);
    const long n = t / a.H;
    const half8 g = *reinterpret_cast<const half8*>(a.dy + pix * a.lddy + c0);
    half8 v[4];
    long src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      src[i] = (n * 2 * a.H + 2 * oy + (i >> 1)) * 2 * a.W + 2 * ox + (i & 1);
      v[i] = *reinterpret_cast<const half8*>(a.x + src[i] * a.ldx + c0);
    }
    int am[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      am[j] = 0;
      float best = (float)v[0][j];
#pragma unroll
      for (int i = 1; i < 4; ++i)
        if ((float)v[i][j] > best) {
          best = (float)v[i][j];
          am[j] = i;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16* d = a.dx + src[i] * a.lddx + c0;
      half8 o;
      if (a.accumulate) o = *reinterpret_cast<const half8*>(d);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gi = (float)g[j] * (0.25f + (am[j] == i ? 1.f : 0.f));
        o[j] = (f16)(gi + (a.accumulate ? (float)o[j] : 0.f));
      }
      *reinterpret_cast<half8*>(d) = o;
    }
  }
}
extern "C" int dy_zoom_pool(const void* x, int ldx, void* y, int ldy, int n, int h, int w, int C, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7)) return DY_ERR_ALIGN;
  ZpArgs a{(const f16*)x, (f16*)y, nullptr, nullptr, ldx, ldy, 0, 0, C, n, h, w, 0};
  hipLaunchKernelGGL(zoom_pool_fwd_kernel, dim3(grid_for((long)n * h * w * (C >> 3))), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
extern "C" int dy_zoom_pool_backward(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int n, int h,
                                     int w, int C, int accumulate, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7)) return DY_ERR_ALIGN;
  ZpArgs a{(const f16*)x, nullptr, (const f16*)dy, (f16*)dx, ldx, 0, lddy, lddx, C, n, h, w, accumulate};
  hipLaunchKernelGGL(zoom_pool_bwd_kernel, dim3(grid_for((long)n * h * w * (C >> 3))), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
