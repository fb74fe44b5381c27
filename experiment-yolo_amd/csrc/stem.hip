// The stem of the detector read straight from the image batch: Conv(3 -> 16, k 3, s 2, p 1) + BatchNorm statistics forward, and
// its weight gradient with the BatchNorm / SiLU backward apply folded in (there is no input gradient: the input is the image).
//
// Replaces, for model.0 of every DEAL-YOLO YAML (reference nn/modules/conv.py:41-55 fed by models/yolo/detect/train.py:57-59
// ``batch["img"].float() / 255``), the three-launch form "import fp32 NCHW -> fp16 NHWC padded to 8 channels; generic conv;
// generic weight gradient", which moved 735 MB through the import kernel and read a 420 MB padded tensor twice for 3 real
// channels.  Here the fp32 NCHW planes are read once per pass (315 MB), staged as fp16 in LDS.
//
// K = 3 * 9 = 27 (padded to 32): one MFMA k-step.  Forward: plain FMAs, one output pixel per thread (432 per pixel: 36 us of vector
// time over the whole batch, under the ~100 us the memory traffic takes).  Weight gradient: reduction over pixels on MFMA
// (v_mfma_f32_16x16x32_f16, A = d(raw)^T, B = im2col patch values), two N-tiles (k index 0..15, 16..31) per 32 pixels.
#include "common.h"
#include "dealyolo_hip.h"

#define STEM_CO 16
#define STEM_TH 8
#define STEM_TW 32
#define STEM_IH (2 * STEM_TH + 1)   // 17 input rows per tile
#define STEM_IW (2 * STEM_TW + 1)   // 65 input columns per tile
#define STEM_IP 68                  // LDS row pitch in halfs

struct StemArgs {
  const float* img;    // (N, 3, H, W) fp32
  const float* w;      // (16, 3, 3, 3) fp32 master weights
  f16* raw;            // forward out: (N, Ho, Wo, ldraw) raw conv output
  double* acc;         // forward: [DY_BN_COPIES][2][16] statistic accumulator (adds); backward: sums of g and g*xhat (reads)
  const f16* dy;       // backward: gradient w.r.t. the activated output (N, Ho, Wo, lddy)
  const float* coef;   // backward: [4][16] scale, shift, mean, invstd
  float* dgamma;
  float* dbeta;
  float* slabs;        // backward: [gridDim.x][9][16][16] fp32 (tap, cout, cin) partial weight gradients
  int N, H, W, Ho, Wo, ldraw, lddy, tiles_x, tiles_y, ntiles;
  float mul, count;
};

// stage the (17 x 65) x 3 input patch of one tile as fp16, zero outside the image
static __device__ __forceinline__ void stem_stage(const StemArgs& a, f16 (*s_in)[STEM_IH][STEM_IP], int n, int oy0, int ox0) {
  const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
  for (int e = threadIdx.x; e < 3 * STEM_IH * STEM_IW; e += 256) {
    const int c = e / (STEM_IH * STEM_IW), r = e - c * (STEM_IH * STEM_IW);
    const int y = r / STEM_IW, x = r - y * STEM_IW;
    const int iy = iy0 + y, ix = ix0 + x;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) v = a.img[(((size_t)n * 3 + c) * a.H + iy) * a.W + ix] * a.mul;
    s_in[c][y][x] = (f16)v;
  }
}

__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
  __shared__ f16 s_in[3][STEM_IH][STEM_IP];
  __shared__ __attribute__((aligned(16))) float s_w[27][STEM_CO];  // [k = c*9 + ky*3 + kx][cout], rounded to fp16 like every packed weight
  __shared__ float s_red[4][2][STEM_CO];
  const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
  for (int e = tid; e < 27 * STEM_CO; e += 256) {
    const int k = e / STEM_CO, co = e - k * STEM_CO;
    s_w[k][co] = (float)(f16)a.w[co * 27 + k];
  }
  float s1[STEM_CO], s2[STEM_CO];
#pragma unroll
  for (int j = 0; j < STEM_CO; ++j) s1[j] = s2[j] = 0.f;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int bx = tile % a.tiles_x, t2 = tile / a.tiles_x;
    const int by = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const int oy0 = by * STEM_TH, ox0 = bx * STEM_TW;
    __syncthreads();
    stem_stage(a, s_in, n, oy0, ox0);
    __syncthreads();
    float o[STEM_CO];
#pragma unroll
    for (int j = 0; j < STEM_CO; ++j) o[j] = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = (float)s_in[c][2 * ty + ky][2 * tx + kx];
          const float* wk = s_w[c * 9 + ky * 3 + kx];
#pragma unroll
          for (int j = 0; j < STEM_CO; j += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wk + j);
            o[j] = fmaf(v, w4[0], o[j]);
            o[j + 1] = fmaf(v, w4[1], o[j + 1]);
            o[j + 2] = fmaf(v, w4[2], o[j + 2]);
            o[j + 3] = fmaf(v, w4[3], o[j + 3]);
          }
        }
    const int oy = oy0 + ty, ox = ox0 + tx;
    if (oy < a.Ho && ox < a.Wo) {
      union { half8 h[2]; uint4 u[2]; } pk;
#pragma unroll
      for (int j = 0; j < STEM_CO; ++j) {
        pk.h[j >> 3][j & 7] = (f16)o[j];
        s1[j] += o[j];  // statistics from the fp32 values, as the ping-pong conv epilogue takes them
        s2[j] += o[j] * o[j];
      }
      uint4* dst = reinterpret_cast<uint4*>(a.raw + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.ldraw);
      dst[0] = pk.u[0];
      dst[1] = pk.u[1];
    }
  }
  // per-workgroup sums -> one fp64 atomic add per (sum, channel) into copy blockIdx.x % DY_BN_COPIES
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int j = 0; j < STEM_CO; ++j) {
    const float r1 = wave_sum(s1[j]), r2 = wave_sum(s2[j]);
    if (lane == 0) {
      s_red[wave][0][j] = r1;
      s_red[wave][1][j] = r2;
    }
  }
  __syncthreads();
  if (tid < 2 * STEM_CO) {
    const int which = tid / STEM_CO, ch = tid - which * STEM_CO;
    const float s = (s_red[0][which][ch] + s_red[1][which][ch]) + (s_red[2][which][ch] + s_red[3][which][ch]);
    unsafeAtomicAdd(&a.acc[((size_t)(blockIdx.x % DY_BN_COPIES) * 2 + which) * STEM_CO + ch], (double)s);
  }
}

// dW[co][c][ky][kx] = sum_pix d(raw)[pix][co] * img[c][2*oy+ky-1][2*ox+kx-1], d(raw) formed from (dy, raw) on the way
__global__ __launch_bounds__(256) void stem_wgrad_bn_kernel(StemArgs a) {
  __shared__ f16 s_in[3][STEM_IH][STEM_IP];
  __shared__ __attribute__((aligned(16))) f16 s_d[STEM_TH * STEM_TW][STEM_CO + 8];  // d(raw) of the tile, pixel-major (+16 B pad)
  __shared__ float s_bn[4][STEM_CO];
  __shared__ float s_acc[4][2][16][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  if (tid < STEM_CO) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < DY_BN_COPIES; ++k) {
      t1 += a.acc[(size_t)(k * 2 + 0) * STEM_CO + tid];
      t2 += a.acc[(size_t)(k * 2 + 1) * STEM_CO + tid];
    }
    const float mg = (float)(t1 / a.count), mgx = (float)(t2 / a.count);
    const float sc = a.coef[tid], sh = a.coef[STEM_CO + tid], mean = a.coef[2 * STEM_CO + tid], inv = a.coef[3 * STEM_CO + tid];
    const float kb = sc * inv * mgx;
    s_bn[0][tid] = sc;
    s_bn[1][tid] = sh;
    s_bn[2][tid] = kb;
    s_bn[3][tid] = sc * mg - kb * mean;
    if (blockIdx.x == 0) {
      if (a.dbeta) a.dbeta[tid] = (float)t1;
      if (a.dgamma) a.dgamma[tid] = (float)t2;
    }
  }
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  // this lane's B rows: k index kk = nt*16 + p -> (tap, c) with kk = tap*3 + c; rows 27..31 are padding
  int boff[2];
  bool bok[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int kk = nt * 16 + p, tap = kk / 3, c = kk - tap * 3, ky = tap / 3, kx = tap - ky * 3;
    bok[nt] = kk < 27;
    boff[nt] = bok[nt] ? (c * STEM_IH + ky) * STEM_IP + kx : 0;
  }
  const f16* s_in_flat = &s_in[0][0][0];
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int bx = tile % a.tiles_x, t2 = tile / a.tiles_x;
    const int by = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const int oy0 = by * STEM_TH, ox0 = bx * STEM_TW;
    __syncthreads();
    stem_stage(a, s_in, n, oy0, ox0);
    {  // d(raw) of this thread's pixel (two 8-channel granules); pixels outside the map contribute zero
      const int ty = tid >> 5, tx = tid & 31, oy = oy0 + ty, ox = ox0 + tx;
      union { half8 h; uint4 u; } o0, o1;
      o0.u = o1.u = make_uint4(0, 0, 0, 0);
      if (oy < a.Ho && ox < a.Wo) {
        const size_t pix = ((size_t)n * a.Ho + oy) * a.Wo + ox;
        const half8 d0 = *reinterpret_cast<const half8*>(a.dy + pix * a.lddy), d1 = *reinterpret_cast<const half8*>(a.dy + pix * a.lddy + 8);
        const half8 r0 = *reinterpret_cast<const half8*>(a.raw + pix * a.ldraw), r1 = *reinterpret_cast<const half8*>(a.raw + pix * a.ldraw + 8);
        o0.h = bn_bwd_apply8<DY_ACT_SILU>(d0, r0, s_bn[0], s_bn[1], s_bn[2], s_bn[3]);
        o1.h = bn_bwd_apply8<DY_ACT_SILU>(d1, r1, s_bn[0] + 8, s_bn[1] + 8, s_bn[2] + 8, s_bn[3] + 8);
      }
      *reinterpret_cast<uint4*>(&s_d[tid][0]) = o0.u;
      *reinterpret_cast<uint4*>(&s_d[tid][8]) = o1.u;
    }
    __syncthreads();
    // wave w owns tile rows 2w, 2w+1 (64 pixels = two 32-pixel k-steps)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ty = wave * 2 + ks, pix0 = ty * STEM_TW + q * 8;
      half8 af, bf[2];
#pragma unroll
      for (int j = 0; j < 8; ++j) af[j] = s_d[pix0 + j][p];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f16* b = s_in_flat + boff[nt] + (2 * ty) * STEM_IP + 2 * (q * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[nt][j] = bok[nt] ? b[2 * j] : (f16)0.f;
      }
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[1], acc[1], 0, 0, 0);
    }
  }
  // D layout: lane (p, q) holds rows (= cout) q*4 .. q*4+3 of column p (= k index nt*16 + p)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_acc[wave][nt][q * 4 + r][p] = acc[nt][r];
  __syncthreads();
  // slab [tap][cout 16][cin 16]: entries with cin >= 3 are never read by the reduction
  float* slab = a.slabs + (size_t)blockIdx.x * 9 * 16 * 16;
  for (int e = tid; e < 27 * STEM_CO; e += 256) {
    const int kk = e / STEM_CO, co = e - kk * STEM_CO, tap = kk / 3, c = kk - tap * 3, nt = kk >> 4, col = kk & 15;
    const float s = (s_acc[0][nt][co][col] + s_acc[1][nt][co][col]) + (s_acc[2][nt][co][col] + s_acc[3][nt][co][col]);
    slab[((size_t)tap * 16 + co) * 16 + c] = s;
  }
}

static int stem_args(StemArgs& a, int n, int h, int w) {
  if (n < 1 || h < 2 || w < 2) return DY_ERR_ARG;
  a.N = n; a.H = h; a.W = w;
  a.Ho = (h + 2 - 3) / 2 + 1;
  a.Wo = (w + 2 - 3) / 2 + 1;
  a.tiles_x = cdiv(a.Wo, STEM_TW);
  a.tiles_y = cdiv(a.Ho, STEM_TH);
  a.ntiles = a.tiles_x * a.tiles_y * n;
  return DY_OK;
}
// persistent grid: 8 workgroups per CU keep enough loads in flight; also the number of weight-gradient slabs
extern "C" int dy_stem_grid(int n, int h, int w) {
  StemArgs a{};
  if (stem_args(a, n, h, w) != DY_OK) return DY_ERR_ARG;
  return a.ntiles < 2048 ? a.ntiles : 2048;
}

extern "C" int dy_stem_forward(const float* img_nchw, const float* weight, void* raw, int ldraw, double* acc, int n, int h, int w,
                               float mul, hipStream_t stream) {
  StemArgs a{};
  if (!img_nchw || !weight || !raw || !acc || (ldraw & 7) || ((uintptr_t)raw & 15)) return DY_ERR_ARG;
  if (stem_args(a, n, h, w) != DY_OK) return DY_ERR_ARG;
  a.img = img_nchw; a.w = weight; a.raw = (f16*)raw; a.acc = acc; a.ldraw = ldraw; a.mul = mul;
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(dy_stem_grid(n, h, w)), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_stem_wgrad_bn(const float* img_nchw, const void* dy, int lddy, const void* raw, int ldraw, const float* coef,
                                const double* acc, float* dgamma, float* dbeta, float count, float* slabs, int n, int h, int w,
                                float mul, hipStream_t stream) {
  StemArgs a{};
  if (!img_nchw || !dy || !raw || !coef || !acc || !slabs || (ldraw & 7) || (lddy & 7) || ((uintptr_t)raw & 15) || ((uintptr_t)dy & 15))
    return DY_ERR_ARG;
  if (stem_args(a, n, h, w) != DY_OK) return DY_ERR_ARG;
  a.img = img_nchw; a.dy = (const f16*)dy; a.raw = (f16*)const_cast<void*>(raw); a.coef = coef; a.acc = const_cast<double*>(acc);
  a.dgamma = dgamma; a.dbeta = dbeta; a.slabs = slabs; a.ldraw = ldraw; a.lddy = lddy; a.mul = mul; a.count = count;
  hipLaunchKernelGGL(stem_wgrad_bn_kernel, dim3(dy_stem_grid(n, h, w)), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
