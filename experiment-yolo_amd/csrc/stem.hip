// The stem of the detector read straight from the image batch: Conv(3 -> 16, k 3, s 2, p 1) + BatchNorm statistics forward, and
// its weight gradient with the BatchNorm / SiLU backward apply folded in (there is no input gradient: the input is the image).
//
// Replaces, for model.0 of every DEAL-YOLO YAML (reference nn/modules/conv.py:41-55 fed by models/yolo/detect/train.py:57-59
// ``batch["img"].float() / 255``), the three-launch form "import fp32 NCHW -> fp16 NHWC padded to 8 channels; generic conv;
// generic weight gradient", which moved 735 MB through the import kernel and read a 420 MB padded tensor twice for 3 real
// channels.  Here the fp32 NCHW planes are read once per pass (315 MB), staged as fp16 in LDS.
//
// K = 3 * 9 = 27.  Forward: v_mfma_f32_16x16x16_f16 over K groups of (channel, row) x 4 columns, so a lane's operand is one 8-byte LDS
// read (a first version with plain FMAs -- 432 per pixel, weights broadcast from LDS -- took 292 us against 277 us for the import +
// generic conv it replaces).  Weight gradient: reduction over pixels on v_mfma_f32_16x16x32_f16, A = d(raw)^T, B = im2col patch
// values, two N-tiles (k index 0..15, 16..31) per 32 pixels.
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
  const float* bias;   // forward, inference form (stem_fwd_kernel<true>): per-channel bias (BatchNorm folded into the weights) or null
  int silu;            // ... and SiLU on the way out; no statistics
};

// stage the (17 x 65) x 3 input patch of one tile as fp16, zero outside the image: one patch row per wave and trip (a coalesced
// 256-byte load), no integer division
static __device__ __forceinline__ void stem_stage(const StemArgs& a, f16 (*s_in)[STEM_IH][STEM_IP], int n, int oy0, int ox0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
  const float* img = a.img + (size_t)n * 3 * a.H * a.W;
#pragma unroll 1
  for (int r0 = 0; r0 < 3 * STEM_IH; r0 += 4 * 4) {   // four rows per wave in flight
    float v[4], ve[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * 4 + wave;
      const int c = (r >= STEM_IH) + (r >= 2 * STEM_IH), y = r - c * STEM_IH, iy = iy0 + y, ix = ix0 + lane;
      const bool rowok = r < 3 * STEM_IH && (unsigned)iy < (unsigned)a.H;
      const float* row = img + ((size_t)c * a.H + (rowok ? iy : 0)) * a.W;
      v[u] = (rowok && (unsigned)ix < (unsigned)a.W) ? row[ix] : 0.f;
      ve[u] = (rowok && lane == 0 && ix0 + 64 < a.W) ? row[ix0 + 64] : 0.f;  // the 65th column
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * 4 + wave;
      if (r < 3 * STEM_IH) {
        const int c = (r >= STEM_IH) + (r >= 2 * STEM_IH), y = r - c * STEM_IH;
        s_in[c][y][lane] = (f16)(v[u] * a.mul);
        if (lane == 0) s_in[c][y][64] = (f16)(ve[u] * a.mul);
      }
    }
  }
}

typedef _Float16 half4_ __attribute__((ext_vector_type(4)));
typedef uint2 __attribute__((aligned(4))) uint2_a4;

// Forward on MFMA 16x16x16 (f16): K is laid out in groups of four, group g = c*3 + ky holding kx = 0..3 (kx = 3 is padding with a zero
// weight), so one lane's four K values are four CONSECUTIVE halfs of a staged input row: one 8-byte LDS read per MFMA.  Nine groups
// = three MFMAs per 16 pixels.  A = weights (rows = cout), B = patches (columns = pixels): a lane ends up with 4 consecutive
// channels of one pixel.
// INFER: the eval form -- ``raw`` receives the finished activation (+ bias, SiLU: Conv.forward_fuse, reference nn/modules/conv.py:57-59)
// or the plain conv output (un-fused eval: BatchNorm with running statistics follows as its own pass); no statistics either way.
template <bool INFER>
__global__ __launch_bounds__(256) void stem_fwd_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) f16 s_in[3][STEM_IH][STEM_IP];
  __shared__ float s_red[4][2][STEM_CO];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  half4_ af[3];
  int goff[3];  // LDS half offset of this lane's K group (clamped to the last real one: its weights are zero beyond)
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int g = ks * 4 + q, gc = g < 9 ? g : 8, c = gc / 3, ky = gc - c * 3;
    goff[ks] = (c * STEM_IH + ky) * STEM_IP;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) af[ks][kx] = (g < 9 && kx < 3) ? (f16)a.w[p * 27 + c * 9 + ky * 3 + kx] : (f16)0.f;
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (INFER && a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + q * 4);
  const f16* s_flat = &s_in[0][0][0];
  // the padding slot kx = 3 of the last pixel reads column 65, which the staging never writes: its weight is zero, but 0 x (whatever
  // bits LDS holds) is NaN when those bits are a NaN -- the pad columns are zeroed once
  for (int e = tid; e < 3 * STEM_IH * (STEM_IP - STEM_IW); e += 256)
    (&s_in[0][0][0])[(e / (STEM_IP - STEM_IW)) * STEM_IP + STEM_IW + e % (STEM_IP - STEM_IW)] = (f16)0.f;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int bx = tile % a.tiles_x, t2 = tile / a.tiles_x;
    const int by = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const int oy0 = by * STEM_TH, ox0 = bx * STEM_TW;
    __syncthreads();
    stem_stage(a, s_in, n, oy0, ox0);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {  // wave w owns tile rows 2w, 2w+1: four 16-pixel N-tiles
      const int ty = wave * 2 + (nt >> 1), px = (nt & 1) * 16 + p;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        union { uint2 u; half4_ h; } bf;
        const uint2_a4* src = reinterpret_cast<const uint2_a4*>(s_flat + goff[ks] + (2 * ty) * STEM_IP + 2 * px);  // 4-byte aligned
        bf.u.x = src->x;
        bf.u.y = src->y;
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(af[ks], bf.h, acc, 0, 0, 0);
      }
      const int oy = oy0 + ty, ox = ox0 + px;
      if (oy < a.Ho && ox < a.Wo) {
        union { half4_ h; uint2 u; } o;
        if (INFER) {
          acc += b4;
          if (a.silu) acc = (f32x4){silu_f(acc[0]), silu_f(acc[1]), silu_f(acc[2]), silu_f(acc[3])};
        }
        o.h = __builtin_convertvector(acc, half4_);
        *reinterpret_cast<uint2*>(a.raw + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.ldraw + q * 4) = o.u;
        if (!INFER) {
          s1 += acc;  // statistics from the fp32 values, as the ping-pong conv epilogue takes them
          s2 += acc * acc;
        }
      }
    }
  }
  if (INFER) return;
  // lane (p, q) holds channels q*4 .. q*4+3: sum over the 16 pixels lanes, then over the waves, one fp64 atomic per (sum, channel)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float r1 = quad16_sum(s1[r]), r2 = quad16_sum(s2[r]);
    if (p == 0) {
      s_red[wave][0][q * 4 + r] = r1;
      s_red[wave][1][q * 4 + r] = r2;
    }
  }
  __syncthreads();
  if (tid < 2 * STEM_CO) {
    const int which = tid / STEM_CO, ch = tid - which * STEM_CO;
    const float s = (s_red[0][which][ch] + s_red[1][which][ch]) + (s_red[2][which][ch] + s_red[3][which][ch]);
    unsafeAtomicAdd(&a.acc[((size_t)(blockIdx.x % DY_BN_COPIES) * 2 + which) * STEM_CO + ch], (double)s);
  }
}

// dW[co][c][ky][kx] = sum_pix d(raw)[pix][co] * img[c][2*oy+ky-1][2*ox+kx-1], d(raw) formed from (dy, raw) on the way.
// MFMA 16x16x32 with K = 32 pixels of one tile row: A = d(raw)^T from a channel-major LDS image (one 16-byte read), B = patch values
// of k index kk = c*9 + ky*3 + kx (two N-tiles: kk 0..15, 16..31; 27..31 padding), eight 2-byte reads two halfs apart.
#define STEM_DP (STEM_TH * STEM_TW + 8)   // pitch of the channel-major d(raw) image in halfs
__global__ __launch_bounds__(256) void stem_wgrad_bn_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) f16 s_in[3][STEM_IH][STEM_IP];
  __shared__ __attribute__((aligned(16))) f16 s_d[STEM_CO][STEM_DP];
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
  for (int e = tid; e < 3 * STEM_IH * (STEM_IP - STEM_IW); e += 256)  // pad columns of the staged patch: never read here, kept defined
    (&s_in[0][0][0])[(e / (STEM_IP - STEM_IW)) * STEM_IP + STEM_IW + e % (STEM_IP - STEM_IW)] = (f16)0.f;
  int boff[2];
  bool bok[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int kk = nt * 16 + p, c = kk / 9, r = kk - c * 9, ky = r / 3, kx = r - ky * 3;
    bok[nt] = kk < 27;
    boff[nt] = bok[nt] ? (c * STEM_IH + ky) * STEM_IP + kx : 0;
  }
  const f16* s_in_flat = &s_in[0][0][0];
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int bx = tile % a.tiles_x, t2 = tile / a.tiles_x;
    const int by = t2 % a.tiles_y, n = t2 / a.tiles_y;
    const int oy0 = by * STEM_TH, ox0 = bx * STEM_TW;
    __syncthreads();
    // d(raw) of this thread's pixel (two 8-channel granules): the loads go first, the patch staging runs under them
    const int ty = tid >> 5, tx = tid & 31, oy = oy0 + ty, ox = ox0 + tx;
    const bool pv = oy < a.Ho && ox < a.Wo;
    const size_t pix = pv ? ((size_t)n * a.Ho + oy) * a.Wo + ox : 0;
    const half8 d0 = *reinterpret_cast<const half8*>(a.dy + pix * a.lddy), d1 = *reinterpret_cast<const half8*>(a.dy + pix * a.lddy + 8);
    const half8 r0 = *reinterpret_cast<const half8*>(a.raw + pix * a.ldraw), r1 = *reinterpret_cast<const half8*>(a.raw + pix * a.ldraw + 8);
    stem_stage(a, s_in, n, oy0, ox0);
    {
      half8 o0 = bn_bwd_apply8<DY_ACT_SILU>(d0, r0, s_bn[0], s_bn[1], s_bn[2], s_bn[3]);
      half8 o1 = bn_bwd_apply8<DY_ACT_SILU>(d1, r1, s_bn[0] + 8, s_bn[1] + 8, s_bn[2] + 8, s_bn[3] + 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // pixels outside the map contribute zero
        s_d[j][tid] = pv ? o0[j] : (f16)0.f;
        s_d[8 + j][tid] = pv ? o1[j] : (f16)0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // wave w owns tile rows 2w, 2w+1 (64 pixels = two 32-pixel k-steps)
      const int row = wave * 2 + ks, pix0 = row * STEM_TW + q * 8;
      const half8 af = *reinterpret_cast<const half8*>(&s_d[p][pix0]);
      half8 bf[2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f16* b = s_in_flat + boff[nt] + (2 * row) * STEM_IP + 2 * (q * 8);
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
    const int kk = e / STEM_CO, co = e - kk * STEM_CO, c = kk / 9, tap = kk - c * 9, nt = kk >> 4, col = kk & 15;
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
// persistent grids.  Forward: 8 workgroups per CU keep enough loads in flight.  Weight gradient: 4 per CU -- every workgroup ends
// with a 9 KB slab that the batched reduction reads back (2048 slabs made that launch 234 instead of 190 us).
static int stem_fwd_grid(int ntiles) { return ntiles < 2048 ? ntiles : 2048; }
extern "C" int dy_stem_grid(int n, int h, int w) {
  StemArgs a{};
  if (stem_args(a, n, h, w) != DY_OK) return DY_ERR_ARG;
  return a.ntiles < 1024 ? a.ntiles : 1024;
}

extern "C" int dy_stem_forward(const float* img_nchw, const float* weight, void* raw, int ldraw, double* acc, int n, int h, int w,
                               float mul, hipStream_t stream) {
  StemArgs a{};
  if (!img_nchw || !weight || !raw || !acc || (ldraw & 7) || ((uintptr_t)raw & 15)) return DY_ERR_ARG;
  if (stem_args(a, n, h, w) != DY_OK) return DY_ERR_ARG;
  a.img = img_nchw; a.w = weight; a.raw = (f16*)raw; a.acc = acc; a.ldraw = ldraw; a.mul = mul;
  hipLaunchKernelGGL(stem_fwd_kernel<false>, dim3(stem_fwd_grid(a.ntiles)), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// The same convolution in eval mode: y = act(conv(img * mul) + bias) as fp16 NHWC (bias null: none; silu 0: no activation), no
// statistics.  With BatchNorm folded into (weight, bias) this is the whole fused stem (Conv.forward_fuse, reference
// nn/modules/conv.py:57-59; get_FPS.py:49 fuses before timing); un-fused eval takes bias = null, silu = 0 and applies BatchNorm with
// running statistics afterwards.
extern "C" int dy_stem_forward_eval(const float* img_nchw, const float* weight, const float* bias, void* y, int ldy, int n, int h, int w,
                                    float mul, int silu, hipStream_t stream) {
  StemArgs a{};
  if (!img_nchw || !weight || !y || (ldy & 7) || ((uintptr_t)y & 15) || (bias && ((uintptr_t)bias & 15))) return DY_ERR_ARG;
  if (stem_args(a, n, h, w) != DY_OK) return DY_ERR_ARG;
  a.img = img_nchw; a.w = weight; a.raw = (f16*)y; a.ldraw = ldy; a.mul = mul; a.bias = bias; a.silu = silu;
  hipLaunchKernelGGL(stem_fwd_kernel<true>, dim3(stem_fwd_grid(a.ntiles)), dim3(256), 0, stream, a);
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
