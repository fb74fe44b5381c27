// Implicit-GEMM convolution on MFMA (v_mfma_f32_16x16x32_f16) for NHWC fp16 activations.
//
// Replaces the ATen sequence of Conv.forward (reference nn/modules/conv.py:49-55): conv2d -> (stats for)
// BatchNorm2d -> SiLU, and -- with transposed/flipped packed weights -- the input-gradient half of
// convolution_backward.  GEMM orientation is M = Cout (A = packed weights, straight from L2), N = pixels
// (B = activation halo tile staged in LDS), K = taps x Cin flattened in 8-channel granules, so each lane ends
// up holding 4*MT *contiguous* output channels of one pixel and stores them with one or two 16-byte writes.
#include "common.h"
#include "dealyolo_hip.h"
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

// measurement hook (tools/conv_bench.py DY_EPI=256): run the kernel without its output stores to price the store path
#define DY_EPI_DEBUG_NOSTORE 256
// add ConvArgs::res to the result after bias / activation (dy_conv_forward_res; ping-pong kernel, fast epilogue only)
#define DY_EPI_RES 64

struct ConvArgs {
  const f16* x;
  const f16* w;
  const float* bias;
  void* y;
  float* partials;
  int ldx, ldy;
  int N, H, W;    // (virtual) input extent used for bounds
  int Hr, Wr;     // real input extent (== H, W unless dil == 2)
  int Ho, Wo;
  int cout;       // real output channels (stores beyond are masked)
  int nch;        // number of Cin chunks of CC channels
  int epi;        // DY_EPI_* flags
  int dil;        // 1, or 2 = zero-dilated virtual input (stride-2 dgrad)
  int tiles_x, tiles_y;
  int npix;       // FLAT (1x1) only: N*Ho*Wo
  int xcd_map;    // 1: workgroups of one XCD (blockIdx.x % 8) own a contiguous band of tiles, so halo rows meet in one L2
  // RED (input-gradient launches only): y is the COMPLETE gradient w.r.t. the activated output of a Conv (conv + BatchNorm + SiLU) --
  // this launch is its only writer -- so the first pass of that layer's BatchNorm backward (sums of g = dy * act'(z) and g * xhat,
  // bn_act_bwd_reduce_kernel) runs here, in the epilogue, on the values being stored, and is added into racc
  // DY_EPI_RES (inference, with BIAS | SILU): a tensor added to the activated result -- Bottleneck's shortcut x + cv2(cv1(x)), reference
  // nn/modules/block.py:333-335 -- on the fp16 values about to be stored, i.e. the bits of "store, then dy_add"
  const f16* res;      // (N, Ho, Wo, ldres)
  int ldres;
  // 1x1 (FLAT) launches of the ping-pong kernel only: the input / the output is a never-materialised channel concatenation (DySegs,
  // include/dealyolo_hip.h).  xs.nseg > 0: chunk h of the input comes from the segment that holds channels [h*CC, (h+1)*CC) (the host
  // picked CC so that no chunk straddles two segments).  ys.nseg > 0: every 8-channel piece of the output goes to its segment.
  DySegs xs, ys;
  const f16* rraw;     // (N, Ho, Wo, ldrraw) raw conv output of that layer's forward
  const float* rcoef;  // [4][rC]: scale, shift, mean, invstd
  double* racc;        // [DY_BN_COPIES][2][rC]
  int ldrraw, rC;
};


// BN partial sums of one workgroup leave through here: a row of the per-workgroup partial table (DY_EPI_STATS), or -- with
// DY_EPI_STATS_ACC -- an fp64 atomic add into copy blockIdx.x % DY_BN_COPIES of the layer's accumulator, which the consuming
// kernel sums in its prologue (bn_act.hip, BnAccFwd): no finalize launch between the convolution and its BatchNorm apply.
static __device__ __forceinline__ void stats_out(const ConvArgs& a, int which, int c, float s) {
  const int ctot = (a.cout + 15) & ~15;  // rows are round16(cout) wide whatever the cout-group padding
  if (c >= ctot) return;
  if (a.epi & DY_EPI_STATS_ACC)
    unsafeAtomicAdd(reinterpret_cast<double*>(a.partials) + ((size_t)(blockIdx.x % DY_BN_COPIES) * 2 + which) * ctot + c, (double)s);
  else
    a.partials[((size_t)blockIdx.x * 2 + which) * ctot + c] = s;
}

template <int CC, int MT, int KS, int STRIDE, int TROWS>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  constexpr bool FLAT = (KS == 1);
  constexpr int NT = FLAT ? 4 : 2 * TROWS;
  constexpr int TH = 4 * TROWS, TW = 32;
  constexpr int HW_ = FLAT ? 256 : (TW - 1) * STRIDE + KS;
  constexpr int HH_ = FLAT ? 1 : (TH - 1) * STRIDE + KS;
  constexpr int PS = CC * 2 + 16;  // LDS bytes per pixel (16 B pad de-phases the banks)
  constexpr int KSTEPS = (KS * KS * CC + 31) / 32;
  constexpr int CPP = CC / 8;
  constexpr int PAD = KS / 2;
  constexpr int TILE_BYTES = HH_ * HW_ * PS;
  constexpr int RED_BYTES = 4 * 2 * 16 * MT * 4;
  __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  int n = 0, oy0 = 0, ox0 = 0, pix0 = 0;
  if (FLAT) {
    pix0 = blockIdx.x * 256;
  } else {
    const int bx = blockIdx.x % a.tiles_x;
    const int t2 = blockIdx.x / a.tiles_x;
    const int by = t2 % a.tiles_y;
    n = t2 / a.tiles_y;
    oy0 = by * TH;
    ox0 = bx * TW;
  }

  // per-lane LDS byte offset of its pixel in each N-tile
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (FLAT) {
      boff[t] = (wave * 64 + t * 16 + p) * PS;
    } else {
      const int ty = wave * TROWS + (t >> 1), tx = (t & 1) * 16 + p;
      boff[t] = ((ty * STRIDE) * HW_ + tx * STRIDE) * PS;
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int h = 0; h < a.nch; ++h) {
    // ---- stage the halo tile of Cin chunk h (zero outside the image)
    for (int id = tid; id < HH_ * HW_ * CPP; id += 256) {
      const int pixel = id / CPP, part = id - pixel * CPP;
      const f16* src = nullptr;
      if (FLAT) {
        const int gp = pix0 + pixel;
        if (gp < a.npix) src = a.x + (size_t)gp * a.ldx + h * CC + part * 8;
      } else {
        const int hy = pixel / HW_, hx = pixel - hy * HW_;
        int iy = oy0 * STRIDE - PAD + hy, ix = ox0 * STRIDE - PAD + hx;
        bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        if (a.dil == 2) {
          ok = ok && !((iy | ix) & 1);
          iy >>= 1;
          ix >>= 1;
        }
        if (ok) src = a.x + ((size_t)(n * a.Hr + iy) * a.Wr + ix) * a.ldx + h * CC + part * 8;
      }
      uint4 v = make_uint4(0, 0, 0, 0);
      if (src) v = *reinterpret_cast<const uint4*>(src);
      *reinterpret_cast<uint4*>(smem + pixel * PS + part * 16) = v;
    }
    __syncthreads();
    const f16* wh = a.w + (size_t)((blockIdx.y * a.nch + h) * KSTEPS) * (16 * MT) * 32;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int kk = ks * 32 + q * 8;
      int tap = kk / CC;
      const int c = kk - tap * CC;
      if (tap > KS * KS - 1) tap = KS * KS - 1;  // K padding: weights are zero there, keep the read in range
      const int toff = ((tap / KS) * HW_ + (tap % KS)) * PS + c * 2;
      half8 af[MT], bf[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        af[m] = *reinterpret_cast<const half8*>(wh + ((ks * (16 * MT) + m * 16 + p) * 32 + q * 8));
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[t] = *reinterpret_cast<const half8*>(smem + boff[t] + toff);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf[t], acc[m][t], 0, 0, 0);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds channels [co0, co0 + 4*MT) of one pixel per N-tile
  constexpr int NC = 4 * MT;
  const int co0 = blockIdx.y * (16 * MT) + q * NC;
  float bias[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) bias[j] = ((a.epi & DY_EPI_BIAS) && co0 + j < a.cout) ? a.bias[co0 + j] : 0.f;
  float s1[NC], s2[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) s1[j] = s2[j] = 0.f;

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    bool valid;
    size_t yoff;
    if (FLAT) {
      const int gp = pix0 + wave * 64 + t * 16 + p;
      valid = gp < a.npix;
      yoff = (size_t)gp * a.ldy;
    } else {
      const int oy = oy0 + wave * TROWS + (t >> 1), ox = ox0 + (t & 1) * 16 + p;
      valid = oy < a.Ho && ox < a.Wo;
      yoff = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.ldy;
    }
    float v[NC];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[m * 4 + r] = acc[m][t][r] + bias[m * 4 + r];
    if (a.epi & DY_EPI_SILU) {
#pragma unroll
      for (int j = 0; j < NC; ++j) v[j] = silu_f(v[j]);
    }
    if (a.epi & DY_EPI_F32OUT) {
      float* yp = reinterpret_cast<float*>(a.y) + yoff + co0;
      if (valid) {
        if (co0 + NC <= a.cout && !(a.ldy & 3)) {
#pragma unroll
          for (int j = 0; j < NC; j += 4) {
            float4 o = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
            if (a.epi & DY_EPI_ACCUM) {
              const float4 old = *reinterpret_cast<const float4*>(yp + j);
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *reinterpret_cast<float4*>(yp + j) = o;
          }
        } else {
#pragma unroll
          for (int j = 0; j < NC; ++j)
            if (co0 + j < a.cout) yp[j] = (a.epi & DY_EPI_ACCUM) ? yp[j] + v[j] : v[j];
        }
      }
    } else {
      f16* yp = reinterpret_cast<f16*>(a.y) + yoff + co0;
      __attribute__((aligned(16))) f16 hv[NC];
      if (valid && (a.epi & DY_EPI_ACCUM)) {
        if (co0 + NC <= a.cout) {
          __attribute__((aligned(16))) f16 old[NC];
          if (NC == 4) {
            *reinterpret_cast<uint2*>(old) = *reinterpret_cast<const uint2*>(yp);
          } else {
#pragma unroll
            for (int j = 0; j < NC; j += 8) *reinterpret_cast<uint4*>(old + j) = *reinterpret_cast<const uint4*>(yp + j);
          }
#pragma unroll
          for (int j = 0; j < NC; ++j) v[j] += (float)old[j];
        } else {
#pragma unroll
          for (int j = 0; j < NC; ++j)
            if (co0 + j < a.cout) v[j] += (float)yp[j];
        }
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) hv[j] = (f16)v[j];
      if (valid) {
        if (co0 + NC <= a.cout) {
          if (NC == 4) {
            *reinterpret_cast<uint2*>(yp) = *reinterpret_cast<uint2*>(hv);
          } else {
#pragma unroll
            for (int j = 0; j < NC; j += 8) *reinterpret_cast<uint4*>(yp + j) = *reinterpret_cast<uint4*>(hv + j);
          }
        } else {
#pragma unroll
          for (int j = 0; j < NC; ++j)
            if (co0 + j < a.cout) yp[j] = hv[j];
        }
        if (a.epi & DY_EPI_STATS) {
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            const float r = (float)hv[j];
            s1[j] += r;
            s2[j] += r * r;
          }
        }
      }
    }
  }

  if (a.epi & DY_EPI_STATS) {
    float* red = reinterpret_cast<float*>(smem);  // [wave][2][16*MT]
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      s1[j] = quad16_sum(s1[j]);
      s2[j] = quad16_sum(s2[j]);
    }
    if (p == 0) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        red[(wave * 2 + 0) * (16 * MT) + q * NC + j] = s1[j];
        red[(wave * 2 + 1) * (16 * MT) + q * NC + j] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * 16 * MT) {
      const int which = tid / (16 * MT), ch = tid - which * (16 * MT);
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[(w * 2 + which) * (16 * MT) + ch];
      stats_out(a, which, blockIdx.y * 16 * MT + ch, s);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// v3: persistent 8-wave workgroups (two waves per SIMD cover each other's stalls) with the cout group's packed weights
// RESIDENT in LDS and the next activation tile prefetched into registers while the current one is multiplied.
// Both LDS images are XOR-swizzled instead of padded (conflict-free ds_read_b128, and a 16x32-pixel halo tile plus
// 72 KiB of weights fits the 160 KiB LDS).  All per-granule address arithmetic of the prefetch is hoisted out of the
// tile loop.  v1 above re-fetches every A fragment from L2 inside the k-loop (an L2 round trip per k-step: rocprof
// showed ~5 % MFMA utilisation on 64->64 3x3); it remains the path for weight sets that do not fit LDS.
// activation tile image: pixel stride CC*2+16 bytes.  The 16-byte pad de-phases the banks (at most 2-way conflicts)
// and -- unlike an XOR swizzle -- keeps every tap/k-step displacement a compile-time immediate of ds_read_b128, so the
// k-loop spends no VALU or registers on addresses (an XOR-swizzled variant made hipcc hoist 72 per-lane addresses
// out of the tile loop and spill).
// LDS bytes per staged pixel.  ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...
// (MI355X_MICROARCH.md, LDS table): a group mixes 8 pixels of k-slice q with the 8 OTHER pixels of k-slice q^1, so it is
// conflict-free exactly when the pixel pitch in 16-B slots is == 2 (mod 4) -- even pitches put slice q on even slots and
// slice q^1 on odd ones.  The old odd pitch (CC*2+16) measured 2-way conflicts on every B read (SQ_LDS_BANK_CONFLICT =
// 49 % of SQ_LDS_IDX_ACTIVE).  Stride-2 tiles step two pixels per column, so there the odd pitch is the right one.
static constexpr __host__ __device__ int ps_bytes(int cc, int stride) {
  return stride == 2 ? cc * 2 + 16 : (cc >= 32 ? cc * 2 + 32 : 32);
}
template <int CC, int STRIDE>
static __device__ __forceinline__ int swz(int pixel, int part) {
  return pixel * ps_bytes(CC, STRIDE) + (part << 4);
}


// The cout group's packed weights into LDS (rows of 64 B, 16-byte slots XOR-swizzled by row group).  All of a thread's loads are
// issued before the first store: the 64-wide 3x3 set (73.7 KB) is nine 16-byte loads per thread, and as a plain load -> store loop
// they were nine DEPENDENT round trips to L2 -- in-kernel stamps put 27 % of a 40x40 launch (8.9 k of 32 k cycles) and ~10 us of every
// launch before the first MFMA.
template <int NTHR>
static __device__ __forceinline__ void weights_to_lds(char* sw, const f16* w, int wrows) {
  const uint4* src = reinterpret_cast<const uint4*>(w);
  const int total = wrows * 4;
  constexpr int WB = 10;
  for (int c0 = threadIdx.x; c0 < total; c0 += NTHR * WB) {
    uint4 t[WB];
#pragma unroll
    for (int k = 0; k < WB; ++k) {
      const int c = c0 + k * NTHR;
      t[k] = c < total ? src[c] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < WB; ++k) {
      const int c = c0 + k * NTHR;
      if (c < total) {
        const int row = c >> 2, qq = c & 3;
        *reinterpret_cast<uint4*>(sw + row * 64 + ((qq ^ ((0 - (row >> 2)) & 3)) << 4)) = t[k];
      }
    }
  }
}

#ifdef DY_CONV_TIMING
__device__ unsigned long long dy_timing[8];
extern "C" int dy_conv_timing_fetch(unsigned long long* out, int reset) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(dy_timing), sizeof(dy_timing));
  if (reset) { unsigned long long z[8] = {}; hipMemcpyToSymbol(HIP_SYMBOL(dy_timing), z, sizeof(z)); }
  return 0;
}
#define TSTAMP(i) { const unsigned long long now_ = clock64(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define TSTAMP(i)
#endif
template <int CC, int MT, int KS, int STRIDE, int TROWS, int NW>
__global__ __launch_bounds__(NW * 64) void conv_mfma_wlds_kernel(ConvArgs a, int ntiles) {
#ifdef DY_CONV_TIMING
  unsigned long long tacc[8] = {}, tlast = clock64();
#endif
  constexpr bool FLAT = (KS == 1);
  constexpr int NTHR = NW * 64;                         // NW waves per workgroup (8: two per SIMD; 4: one per SIMD, half the
                                                        // A-fragment LDS traffic per MFMA because each wave owns 64 pixels)
  constexpr int NT = 2 * TROWS;                         // 16-pixel N-tiles per wave
  constexpr int TH = NW * TROWS, TW = 32;
  constexpr int HW_ = FLAT ? NW * NT * 16 : (TW - 1) * STRIDE + KS;
  constexpr int HH_ = FLAT ? 1 : (TH - 1) * STRIDE + KS;
  constexpr int KSTEPS = (KS * KS * CC + 31) / 32;
  constexpr int CPP = CC / 8;
  constexpr int PAD = KS / 2;
  constexpr int NCHUNK16 = HH_ * HW_ * CPP;
  constexpr int NPF = (NCHUNK16 + NTHR - 1) / NTHR;
  constexpr int NC = 4 * MT;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int wrows = a.nch * KSTEPS * 16 * MT;
  char* const sw = dsm;
  char* const st = dsm + wrows * 64;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  weights_to_lds<NTHR>(sw, a.w + (size_t)blockIdx.y * wrows * 32, wrows);
  constexpr int PS = ps_bytes(CC, STRIDE);
  // LDS byte offset of this lane's pixel in each N-tile
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (FLAT) {
      boff[t] = (wave * (NT * 16) + t * 16 + p) * PS;
    } else {
      const int ty = wave * TROWS + (t >> 1), tx = (t & 1) * 16 + p;
      boff[t] = ((ty * STRIDE) * HW_ + tx * STRIDE) * PS;
    }
  }
  // weight rows are 64 B: the four rows r, r+4, r+8, r+12 a lane group touches must land on four different 16-B slots,
  // which the row-group map g -> (-g)&3 gives for the {0-3,12-15 | 4-11} grouping (g -> g does not: 2-way on every A read)
  const int aoff = p * 64 + ((q ^ ((0 - (p >> 2)) & 3)) << 4);

  float bias[NC];
  const int co0 = blockIdx.y * (16 * MT) + q * NC;
#pragma unroll
  for (int j = 0; j < NC; ++j) bias[j] = ((a.epi & DY_EPI_BIAS) && co0 + j < a.cout) ? a.bias[co0 + j] : 0.f;
  float s1[NC], s2[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) s1[j] = s2[j] = 0.f;

  // ---- tile-invariant prefetch metadata of this thread's granules
  int goff[NPF];   // element offset from the tile origin pointer
  int gyx[NPF];    // packed (ry << 16) | (rx & 0xffff), relative real coordinates; -1 row marks "never valid"
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int id = tid + i * NTHR;
    const int pixel = id / CPP, part = id - pixel * CPP;
    if (FLAT) {
      goff[i] = pixel * a.ldx + part * 8;
      gyx[i] = id < NCHUNK16 ? pixel : -1;
    } else {
      const int hy = pixel / HW_, hx = pixel - hy * HW_;
      int ry = hy, rx = hx;
      bool ok = id < NCHUNK16;
      if (a.dil == 2) {
        ok = ok && !(((hy - PAD) | (hx - PAD)) & 1);
        ry = (hy - PAD) >> 1;
        rx = (hx - PAD) >> 1;
      }
      goff[i] = (ry * a.Wr + rx) * a.ldx + part * 8;
      gyx[i] = ok ? ((ry << 16) | (rx & 0xffff)) : (int)0x80000000;
    }
  }

  uint4 pf[NPF];
  auto prefetch = [&](int tile, int h) {
    if (FLAT) {
      const int pix0 = tile * HW_;
      const f16* base = a.x + (size_t)pix0 * a.ldx + h * CC;
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const bool ok = gyx[i] >= 0 && pix0 + gyx[i] < a.npix;
        pf[i] = ok ? *reinterpret_cast<const uint4*>(base + goff[i]) : make_uint4(0, 0, 0, 0);
      }
    } else {
      const int bx = tile % a.tiles_x;
      const int t2 = tile / a.tiles_x;
      const int n = t2 / a.tiles_y;
      const int oy0 = (t2 % a.tiles_y) * TH, ox0 = bx * TW;
      const int ty0 = a.dil == 2 ? (oy0 >> 1) : oy0 * STRIDE - PAD;
      const int tx0 = a.dil == 2 ? (ox0 >> 1) : ox0 * STRIDE - PAD;
      const f16* base = a.x + ((long)(n * a.Hr + ty0) * a.Wr + tx0) * a.ldx + h * CC;
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const int ry = gyx[i] >> 16, rx = (short)(gyx[i] & 0xffff);
        const bool ok = gyx[i] != (int)0x80000000 && (unsigned)(ty0 + ry) < (unsigned)a.Hr && (unsigned)(tx0 + rx) < (unsigned)a.Wr;
        pf[i] = ok ? *reinterpret_cast<const uint4*>(base + goff[i]) : make_uint4(0, 0, 0, 0);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile, 0);
  f32x4 acc[MT][NT];
  while (tile < ntiles) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < a.nch; ++h) {
      TSTAMP(5)
      __syncthreads();
      TSTAMP(0)
#pragma unroll
      for (int i = 0; i < NPF; ++i)
        if (tid + i * NTHR < NCHUNK16) {
          const int id = tid + i * NTHR;
          *reinterpret_cast<uint4*>(st + swz<CC, STRIDE>(id / CPP, id % CPP)) = pf[i];
        }
      __syncthreads();
      TSTAMP(1)
      if (h + 1 < a.nch) prefetch(tile, h + 1);
      else if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x, 0);
      TSTAMP(2)
      const char* wh = sw + (size_t)(h * KSTEPS) * (16 * MT) * 64;
      // software-pipelined k-loop: the fragments of k-step ks+1 are requested from LDS BEFORE the MFMAs of k-step ks issue
      // (hipcc otherwise emits read -> wait -> 2 MFMAs, exposing the ~100-cycle LDS latency once per pair)
      half8 af[2][MT], bf[2][NT];
      auto load_frags = [&](int buf, int ks) {
        const int kk = ks * 32 + q * 8;
        int tap = kk / CC;
        const int c = kk - tap * CC;
        if (tap > KS * KS - 1) tap = KS * KS - 1;
        const int toff = ((tap / KS) * HW_ + (tap % KS)) * PS + c * 2;
#pragma unroll
        for (int m = 0; m < MT; ++m) af[buf][m] = *reinterpret_cast<const half8*>(wh + (ks * (16 * MT) + m * 16) * 64 + aoff);
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[buf][t] = *reinterpret_cast<const half8*>(st + boff[t] + toff);
      };
      load_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        if (ks + 1 < KSTEPS) load_frags((ks + 1) & 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks & 1][m], bf[ks & 1][t], acc[m][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      TSTAMP(3)
    }
    // ---- epilogue of this tile
    int n = 0, oy0 = 0, ox0 = 0, pix0 = 0;
    if (FLAT) {
      pix0 = tile * HW_;
    } else {
      const int bx = tile % a.tiles_x;
      const int t2 = tile / a.tiles_x;
      n = t2 / a.tiles_y;
      oy0 = (t2 % a.tiles_y) * TH;
      ox0 = bx * TW;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bool valid;
      size_t yoff;
      if (FLAT) {
        const int gp = pix0 + wave * (NT * 16) + t * 16 + p;
        valid = gp < a.npix;
        yoff = (size_t)gp * a.ldy;
      } else {
        const int oy = oy0 + wave * TROWS + (t >> 1), ox = ox0 + (t & 1) * 16 + p;
        valid = oy < a.Ho && ox < a.Wo;
        yoff = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.ldy;
      }
      float v[NC];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[m * 4 + r] = acc[m][t][r] + bias[m * 4 + r];
      if (a.epi & DY_EPI_SILU) {
#pragma unroll
        for (int j = 0; j < NC; ++j) v[j] = silu_f(v[j]);
      }
      if (a.epi & DY_EPI_F32OUT) {
        float* yp = reinterpret_cast<float*>(a.y) + yoff + co0;
        if (valid) {
          if (co0 + NC <= a.cout && !(a.ldy & 3)) {
#pragma unroll
            for (int j = 0; j < NC; j += 4) {
              float4 o = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
              if (a.epi & DY_EPI_ACCUM) {
                const float4 old = *reinterpret_cast<const float4*>(yp + j);
                o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
              }
              *reinterpret_cast<float4*>(yp + j) = o;
            }
          } else {
#pragma unroll
            for (int j = 0; j < NC; ++j)
              if (co0 + j < a.cout) yp[j] = (a.epi & DY_EPI_ACCUM) ? yp[j] + v[j] : v[j];
          }
        }
      } else {
        f16* yp = reinterpret_cast<f16*>(a.y) + yoff + co0;
        __attribute__((aligned(16))) f16 hv[NC];
        if (valid && (a.epi & DY_EPI_ACCUM)) {
          if (co0 + NC <= a.cout) {
            __attribute__((aligned(16))) f16 old[NC];
            if (NC == 4) {
              *reinterpret_cast<uint2*>(old) = *reinterpret_cast<const uint2*>(yp);
            } else {
#pragma unroll
              for (int j = 0; j < NC; j += 8) *reinterpret_cast<uint4*>(old + j) = *reinterpret_cast<const uint4*>(yp + j);
            }
#pragma unroll
            for (int j = 0; j < NC; ++j) v[j] += (float)old[j];
          } else {
#pragma unroll
            for (int j = 0; j < NC; ++j)
              if (co0 + j < a.cout) v[j] += (float)yp[j];
          }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) hv[j] = (f16)v[j];
        if (valid) {
          if (co0 + NC <= a.cout) {
            if (NC == 4) {
              *reinterpret_cast<uint2*>(yp) = *reinterpret_cast<uint2*>(hv);
            } else {
#pragma unroll
              for (int j = 0; j < NC; j += 8) *reinterpret_cast<uint4*>(yp + j) = *reinterpret_cast<uint4*>(hv + j);
            }
          } else {
#pragma unroll
            for (int j = 0; j < NC; ++j)
              if (co0 + j < a.cout) yp[j] = hv[j];
          }
          if (a.epi & DY_EPI_STATS) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
              const float r = (float)hv[j];
              s1[j] += r;
              s2[j] += r * r;
            }
          }
        }
      }
    }
    tile += gridDim.x;
    TSTAMP(4)
  }
#ifdef DY_CONV_TIMING
  if (lane == 0 && wave == 0) {
    TSTAMP(6)
    for (int i = 0; i < 7; ++i) atomicAdd(&dy_timing[i], tacc[i]);
    atomicAdd(&dy_timing[7], 1ull);
  }
#endif

  if (a.epi & DY_EPI_STATS) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(st);
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      s1[j] = quad16_sum(s1[j]);
      s2[j] = quad16_sum(s2[j]);
    }
    if (p == 0) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        red[(wave * 2 + 0) * (16 * MT) + q * NC + j] = s1[j];
        red[(wave * 2 + 1) * (16 * MT) + q * NC + j] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * 16 * MT) {
      const int which = tid / (16 * MT), ch = tid - which * (16 * MT);
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[(w * 2 + which) * (16 * MT) + ch];
      stats_out(a, which, blockIdx.y * 16 * MT + ch, s);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// v4 "ping-pong": the same weights-in-LDS persistent kernel, but the 8 waves form two groups of four (one wave per SIMD
// each) that work on DIFFERENT tiles half a period apart.  In-kernel stamps of v3 showed its waves in lock-step: 26 % of
// a wave's cycles in the k-loop, 24 % in the epilogue and 26 % waiting at the barrier for the others' epilogues -- the MFMA
// pipe idles while all eight waves convert/store, the VALU idles while all eight multiply.  Here, in every slot between
// two workgroup barriers, one group runs a pure LDS->MFMA k-loop on its own staged tile while the other does everything
// else for its tile (epilogue + stats of the finished tile, write of the prefetched next chunk into its own LDS buffer,
// address math and issue of the chunk after that); then they swap.  The matrix pipe therefore sees a k-loop in every
// slot and the vector/memory work runs beside it.
//   slot s:  group g computes chunk j = (s-g)/2 when s-g is even, and runs MEM(j) with j = (s-g-1)/2 when it is odd.
//   MEM(j):  epilogue of j's tile if j was its last chunk; LDS write of chunk j+1; global prefetch of chunk j+2.
// FW (0 / 40 / 80): "full-width" tiles for maps that are exactly FW pixels wide (3x3 stride-1, 32-channel chunks).  The usual tile is 32
// columns wide, so a 40-wide map pays for 64 columns (37.5 % of its MFMAs and staged bytes on pixels that do not exist) and an 80-wide
// one for 96.  With FW a wave owns an 80-pixel STRIP -- two rows of 40 or one of 80, contiguous in memory because the tile spans the
// whole width -- as five 16-pixel N-tiles: the epilogue addresses it like the 1x1 kernel's flat pixels, the k-loop like any 3x3 tile
// except that N-tile 2 of a 40-wide strip straddles the row end (its lanes 8-15 sit one LDS row further: one more base register).
// Columns 0 and FW+1 of the halo never exist; the descriptor's range check answers them.  Fits beside 73.7 KB of resident weights
// only with the 64-byte swizzled pixels (SWZ below): 10 x 48 x 64 B = 30.7 KB per group.
template <int CC, int MT, int KS, int STRIDE, int TROWS, bool REDK = false, int FW = 0>
__global__ __launch_bounds__(512) void conv_mfma_pp_kernel(ConvArgs a, int ntiles) {
#ifdef DY_CONV_TIMING
  unsigned long long tacc[8] = {}, tlast = clock64();
#endif
  constexpr bool FLAT = (KS == 1);
  constexpr int GW = 4, GTHR = GW * 64;                 // waves / threads per group
  static_assert(FW == 0 || (KS == 3 && STRIDE == 1 && CC == 32 && !REDK && TROWS * FW == 80), "full-width tiles: 3x3 s1, 32-channel chunks");
  constexpr bool LIN = FLAT || FW != 0;                 // a wave's N-tiles are consecutive runs of 16 pixels of ONE contiguous strip
  constexpr int NT = FW ? 5 : 2 * TROWS;                // 16-pixel N-tiles per wave
  constexpr int TH = GW * TROWS, TW = FW ? FW : 32;
  // 3x3 stride-1 tiles of 32-channel chunks: 64-byte pixels WITHOUT padding.  A B-fragment read (ds_read_b128, lanes = 16 consecutive
  // pixels x 4 channel quarters) is conflict-free on that pitch when the 16-byte slot of a pixel is XORed with 2 for pixels whose index
  // has bit 2 set (enumerated over the instruction's four 16-lane groups and every alignment of the 16 pixels; the 96-byte padded pitch
  // it replaces is the only padding <= 112 B that is conflict-free).  Rows are 40 pixels long in LDS (34 real), so that a tap's row
  // offset never changes a pixel's index mod 8 and the XOR term depends on (lane pixel + dx) only: three base registers, immediates
  // as before.  A tile takes 25.6 KB instead of 32.6: the 32-wide cout groups fit two workgroups per CU.
  constexpr bool SWZ = (KS == 3 && STRIDE == 1 && CC == 32);
  constexpr int HWR = FLAT ? GW * NT * 16 : (TW - 1) * STRIDE + KS;   // halo columns that exist
  constexpr int HW_ = SWZ ? ((HWR + 7) & ~7) : HWR;                   // pixels per tile row in LDS
  constexpr int HH_ = FLAT ? 1 : (TH - 1) * STRIDE + KS;
  constexpr int KSTEPS = (KS * KS * CC + 31) / 32;
  constexpr int CPP = CC / 8;
  constexpr int PAD = KS / 2;
  constexpr int NCHUNK16 = HH_ * HW_ * CPP;
  constexpr int NPF = (NCHUNK16 + GTHR - 1) / GTHR;
  constexpr int NC = 4 * MT;
  constexpr int PS = SWZ ? 64 : ps_bytes(CC, STRIDE);
  constexpr int RED_BYTES = 8 * 2 * 16 * MT * 4;
  constexpr int TILE_BYTES = HH_ * HW_ * PS > RED_BYTES ? HH_ * HW_ * PS : RED_BYTES;
  constexpr unsigned NEVER = 0x80000000u;               // byte offset no image reaches: the buffer range check returns 0
  constexpr int RB = 4 * NC * 2;                        // bytes of one pixel's channel block owned by this workgroup
  constexpr int XROW = RB + 16;                         // pitch of the per-wave store-transpose scratch (16 rows)
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int wrows = a.nch * KSTEPS * 16 * MT;
  char* const sw = dsm;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int g = __builtin_amdgcn_readfirstlane(wave >> 2), wg = wave & 3, gtid = tid & (GTHR - 1);
  const int wgs = __builtin_amdgcn_readfirstlane(wg);
  char* const st = dsm + wrows * 64 + g * TILE_BYTES;
  char* const xs = dsm + wrows * 64 + 2 * TILE_BYTES + wave * (16 * XROW);
  float* const bsh = reinterpret_cast<float*>(dsm + wrows * 64 + 2 * TILE_BYTES + 8 * (16 * XROW));  // this cout group's bias
  float* const rcf = bsh + 16 * MT;  // RED: [scale | shift] of this cout group's channels
  if (tid < 16 * MT) {
    const int c = blockIdx.y * (16 * MT) + tid;
    bsh[tid] = ((a.epi & DY_EPI_BIAS) && c < a.cout) ? a.bias[c] : 0.f;
    if (REDK) {
      rcf[tid] = c < a.rC ? a.rcoef[c] : 0.f;
      rcf[16 * MT + tid] = c < a.rC ? a.rcoef[a.rC + c] : 0.f;
    }
  }
  // Segmented input (1x1): where chunk h lives -- {base pointer of the chunk's first channel (lo, hi), pixel stride, bytes to the end
  // of the segment} -- found once per workgroup by thread h and parked in LDS: the staging loop reads its chunk's entry instead of
  // searching the table in the kernel arguments (a dependent scalar-load chain per chunk: the first version, +10 us on a 160x160 launch)
  unsigned* const xtab = reinterpret_cast<unsigned*>(rcf + 2 * 16 * MT);
  if (FLAT && a.xs.nseg > 0 && tid < a.nch && tid < 16) {
    const int ch0 = tid * CC;
    int sg = 0;
    while (sg + 1 < a.xs.nseg && ch0 >= a.xs.c_end[sg]) ++sg;
    const int cb = sg ? a.xs.c_end[sg - 1] : 0;
    const unsigned long long base = (unsigned long long)(reinterpret_cast<const f16*>(a.xs.ptr[sg]) + (ch0 - cb));
    xtab[tid * 4 + 0] = (unsigned)base;
    xtab[tid * 4 + 1] = (unsigned)(base >> 32);
    const bool up = (a.xs.acc[sg] & 2) != 0;  // the segment is a 2x nearest-neighbour up-sampling of a (N, H/2, W/2) tensor (DySegs)
    xtab[tid * 4 + 2] = (unsigned)a.xs.ld[sg] | (up ? 0x80000000u : 0u);
    xtab[tid * 4 + 3] = (up ? (unsigned)(a.N * (a.H >> 1) * (a.W >> 1)) : (unsigned)a.npix) * (unsigned)a.xs.ld[sg] * 2u - (unsigned)(ch0 - cb) * 2u;
  }
  // Segmented output (1x1 input gradient): this lane's 8-channel piece lives in ONE segment for the whole launch -- its base at the
  // piece's channel, its pixel stride in bytes, whether that segment accumulates
  char* sbase = nullptr;
  unsigned sld2 = 0;
  bool sacc = false;
  if (FLAT && a.ys.nseg > 0) {
    const int chn = (int)(blockIdx.y * (16 * MT)) + (lane % ((4 * NC * 2) / 16)) * 8;
    int sg = 0;
    while (sg + 1 < a.ys.nseg && chn >= a.ys.c_end[sg]) ++sg;
    const int cb = sg ? a.ys.c_end[sg - 1] : 0;
    sbase = reinterpret_cast<char*>(const_cast<void*>(a.ys.ptr[sg])) + (chn - cb) * 2;
    sld2 = (unsigned)a.ys.ld[sg] * 2u;
    sacc = a.ys.acc[sg] != 0;
  }
  bool seg_any_acc = false;  // (wave-uniform) some output segment accumulates: the epilogue variant that loads old values
  if (FLAT)
    for (int k = 0; k < a.ys.nseg; ++k) seg_any_acc = seg_any_acc || a.ys.acc[k] != 0;
  if (FLAT && a.xs.nseg > 0) __syncthreads();
  f32x2 rsg[4], rsgx[4];  // RED: this lane's running sums over its 8-channel piece (dead registers in every other variant)
#pragma unroll
  for (int j = 0; j < 4; ++j) rsg[j] = rsgx[j] = (f32x2){0.f, 0.f};
  // RED: the raw conv output of the layer whose gradient this launch writes, for the pixels / channel pieces of this lane's stores in
  // the NEXT tile this group will finish.  Requested one tile ahead (right after the previous epilogue), so that the HBM latency
  // lies under a compute slot: requested inside the epilogue it was exposed once per tile and tripled the kernel (measured).
  constexpr int R_PPR = (4 * NC * 2) / 16, R_PIXPASS = 64 / R_PPR, R_NPASS = R_PIXPASS >= 16 ? 1 : 16 / R_PIXPASS;
  union RU4 { uint4 u; half2_ h[4]; };
  RU4 rw[REDK ? NT : 1][REDK ? R_NPASS : 1];
  auto red_prefetch = [&](const auto& tc) {
    const int r_dpix = lane / R_PPR, r_piece = lane % R_PPR;
    const int n = tc.n, oy0 = tc.by * TH, ox0 = tc.bx * TW, pix0 = tc.bx * HW_;
    const int row0 = FLAT ? 0 : oy0 + wgs * TROWS;
    const long tbase = FLAT ? (long)(pix0 + wgs * (NT * 16)) * a.ldrraw : ((long)(n * a.Ho + row0) * a.Wo + ox0) * a.ldrraw;
    const char* const rbase = reinterpret_cast<const char*>(a.rraw) + tbase * 2;
    const unsigned rloff = (unsigned)((r_dpix * a.ldrraw + blockIdx.y * (16 * MT) + r_piece * 8) * 2);
    const int collim = FLAT ? a.npix - (pix0 + wgs * (NT * 16)) : a.Wo - ox0;
    const bool chok = (int)(blockIdx.y * (16 * MT) + r_piece * 8) < a.rC;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int ps = 0; ps < R_NPASS; ++ps) {
        const int c0 = (FLAT ? t * 16 : (t & 1) * 16) + ps * R_PIXPASS;
        const bool rowok = FLAT ? true : row0 + (t >> 1) < a.Ho;
        const bool valid = rowok && chok && r_dpix < 16 && c0 + r_dpix < collim;
        const long soff = FLAT ? (long)c0 * a.ldrraw : ((long)(t >> 1) * a.Wo + c0) * a.ldrraw;
        rw[REDK ? t : 0][REDK ? ps : 0].u = *reinterpret_cast<const uint4*>(valid ? rbase + soff * 2 + rloff : reinterpret_cast<const char*>(a.rraw));
      }
  };
  // LDS byte offset of this lane's pixel in N-tile 0; N-tile t lies a compile-time distance further (bdelta)
  const int boff0 = FLAT ? (wg * (NT * 16) + p) * PS : ((wg * TROWS * STRIDE) * HW_ + p * STRIDE) * PS;
  auto bdelta = [](int t) {
    constexpr int FWD = FW ? FW : 1;
    if (FW) return (((t * 16) / FWD) * HW_ + (t * 16) % FWD) * PS;  // strip pixel 16 t -> (row, column) of the wave's rows
    return FLAT ? t * 16 * PS : (((t >> 1) * STRIDE) * HW_ + (t & 1) * 16 * STRIDE) * PS;
  };
  // FW = 40: N-tile 2 covers columns 32-39 of the strip's first row and 0-7 of its second: lanes 8-15 lie HW_ - FW pixels further
  const int fwadj = (FW == 40 && p >= 8) ? (HW_ - FW) * PS : 0;
  auto straddles = [](int t) { return FW == 40 && t == 2; };
  const char* const stb = st + boff0;
  const char* stbx[3];  // SWZ: the lane's base for taps of column dx, with its channel quarter and that pixel's slot XOR folded in
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) stbx[dx] = stb + ((q * 16) ^ (((p + dx) & 4) << 3));
  (void)stbx;
  const int aoff = p * 64 + ((q ^ ((0 - (p >> 2)) & 3)) << 4);
  const int co0 = blockIdx.y * (16 * MT) + q * NC;
  f32x2 s1[NC / 2], s2[NC / 2];
#pragma unroll
  for (int j = 0; j < NC / 2; ++j) s1[j] = s2[j] = (f32x2){0.f, 0.f};

  // ---- staging: every 16-byte granule of a chunk is one buffer_load_dwordx4 through a descriptor that spans exactly
  // the image the tile lies in, so rows above/below the image (negative or too-large byte offsets) come back as zeros
  // from the hardware range check and cost no VALU; only tiles that touch the left/right border mask columns.
  // Tile-invariant per granule: byte offset from the tile's origin pixel (NEVER for granules that do not exist or are
  // the zero holes of a stride-2 dgrad), and the granule's real column relative to the origin (two per register).
  unsigned goff[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int id = gtid + i * GTHR;
    const int pixel = id / CPP, part = id - pixel * CPP;
    int rx = 0;
    if (FLAT) {
      goff[i] = id < NCHUNK16 ? (unsigned)((pixel * a.ldx + part * 8) * 2) : NEVER;
    } else {
      const int hy = pixel / HW_, hx = pixel - hy * HW_;
      int ry = hy;
      rx = hx;
      bool ok = id < NCHUNK16 && hx < HWR && !(FW && (hx == 0 || hx == FW + 1));
      if (a.dil == 2) {
        ok = ok && !(((hy - PAD) | (hx - PAD)) & 1);
        ry = (hy - PAD) >> 1;
        rx = (hx - PAD) >> 1;
      }
      goff[i] = ok ? (unsigned)(((ry * a.Wr + rx) * a.ldx + part * 8) * 2) : NEVER;
    }
    (void)rx;
  }

  uint4 pf[NPF];
  struct TileCur { int bx, by, n; };  // FLAT: bx is the tile index
  auto prefetch = [&](const TileCur& tc, int h) {
    if (FLAT) {
      const int tile = tc.bx;
      if (a.xs.nseg > 0) {
        // segmented input: the chunk's own base pointer and pixel stride from the table the prologue left in LDS
        const uint4 e = *reinterpret_cast<const uint4*>(xtab + (h < 16 ? h : 15) * 4);
        const unsigned blo = __builtin_amdgcn_readfirstlane(e.x), bhi = __builtin_amdgcn_readfirstlane(e.y);
        const unsigned ldw = __builtin_amdgcn_readfirstlane(e.z);
        const int ld = (int)(ldw & 0x7fffffffu);
        const f16* base = reinterpret_cast<const f16*>(((unsigned long long)bhi << 32) | blo);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(base), 0, (int)__builtin_amdgcn_readfirstlane(e.w), 0x00020000);
        if (ldw >> 31) {
          // an UP-SAMPLED segment (reference nn.Upsample(None, 2, 'nearest') in front of a Concat: models/*.yaml): pixel (n, y, x) of the
          // concatenation reads pixel (n, y >> 1, x >> 1) of the low-resolution tensor -- the four-times-larger copy is never written.
          // The tile's first pixel is decomposed once (wave-uniform integer divisions); a lane's pixel lies < 256 further, so its
          // row / image carries are two exact float divisions of small integers.
          const int g0 = tile * HW_;
          const int hw = a.H * a.W;
          const int n0 = g0 / hw, r0 = g0 - n0 * hw, y0 = r0 / a.W, x0 = r0 - y0 * a.W;
          const float invW = 1.0f / (float)a.W, invH = 1.0f / (float)a.H;
          const int Hh = a.H >> 1, Wh = a.W >> 1;
#pragma unroll
          for (int i = 0; i < NPF; ++i) {
            const int id = gtid + i * GTHR, lp = id / CPP;
            const int xx = x0 + lp, dy_ = (int)(((float)xx + 0.5f) * invW), x = xx - dy_ * a.W;
            const int yt = y0 + dy_, dn = (int)(((float)yt + 0.5f) * invH), y = yt - dn * a.H;
            const bool ok = id < NCHUNK16 && g0 + lp < a.npix;
            const unsigned off = ok ? (unsigned)(((((n0 + dn) * Hh + (y >> 1)) * Wh + (x >> 1)) * ld + (id % CPP) * 8) * 2) : NEVER;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
            pf[i] = *reinterpret_cast<const uint4*>(&v);
          }
          return;
        }
        const unsigned org = (unsigned)tile * HW_ * ld * 2u;
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const int id = gtid + i * GTHR;
          const unsigned off = id < NCHUNK16 ? (unsigned)(((id / CPP) * ld + (id % CPP) * 8) * 2) + org : NEVER;
          const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
          pf[i] = *reinterpret_cast<const uint4*>(&v);
        }
        return;
      }
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x), 0, (int)((unsigned)a.npix * (unsigned)a.ldx * 2u), 0x00020000);
      const unsigned org = ((unsigned)tile * HW_ * a.ldx + h * CC) * 2u;
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(goff[i] + org), 0, 0);
        pf[i] = *reinterpret_cast<const uint4*>(&v);
      }
    } else {
      const int n = tc.n;
      const int oy0 = tc.by * TH, ox0 = tc.bx * TW;
      const int ty0 = a.dil == 2 ? (oy0 >> 1) : oy0 * STRIDE - PAD;
      const int tx0 = a.dil == 2 ? (ox0 >> 1) : ox0 * STRIDE - PAD;
      const int img = a.Hr * a.Wr * a.ldx * 2;
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x) + (size_t)n * a.Hr * a.Wr * a.ldx, 0, img, 0x00020000);
      const unsigned org = (unsigned)(((ty0 * a.Wr + tx0) * a.ldx + h * CC) * 2);
      const bool xedge = !FW && (tx0 < 0 || tx0 + (a.dil == 2 ? (HWR + 1) / 2 : HWR) > a.Wr);   // wave-uniform (FW: goff holds the border)
      if (!xedge) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(goff[i] + org), 0, 0);
          pf[i] = *reinterpret_cast<const uint4*>(&v);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
          const int hx = ((gtid + i * GTHR) / CPP) % HW_;  // recomputed on border tiles only: cheaper than NPF live registers
          const int rx = a.dil == 2 ? (hx - PAD) >> 1 : hx;
          const unsigned off = (unsigned)(tx0 + rx) < (unsigned)a.Wr ? goff[i] + org : NEVER;
          const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
          pf[i] = *reinterpret_cast<const uint4*>(&v);
        }
      }
    }
  };

  const int gstride = 2 * gridDim.x;
  // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  With the identity map the tiles of one period are
  // spread over all XCDs and every halo row is fetched into two L2s; mapping XCD x to the x-th eighth of the period's tiles
  // keeps vertical neighbours on one XCD (only the band edges still meet a second L2).
  const int vb = (a.xcd_map && !(gridDim.x & 7)) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int gt0 = vb * 2 + g;
  const int ntg = gt0 < ntiles ? (ntiles - gt0 + gstride - 1) / gstride : 0;
  const int J = ntg * a.nch;
  const int Jmax = ((ntiles - vb * 2 + gstride - 1) / gstride) * a.nch;  // group 0 has the most tiles
  // tile coordinates advance by a fixed (dx, dy, dn) per period: no integer division inside the loop
  const int sdx = FLAT ? gstride : gstride % a.tiles_x;
  const int sdy = FLAT ? 0 : (gstride / a.tiles_x) % a.tiles_y;
  const int sdn = FLAT ? 0 : (gstride / a.tiles_x) / a.tiles_y;
  auto tile_next = [&](TileCur& c) {
    c.bx += sdx;
    if (!FLAT) {
      const int cx = c.bx >= a.tiles_x ? 1 : 0;
      c.bx -= cx * a.tiles_x;
      c.by += sdy + cx;
      const int cy = c.by >= a.tiles_y ? 1 : 0;
      c.by -= cy * a.tiles_y;
      c.n += sdn + cy;
    }
  };
  TileCur pcur, ecur;  // tile being prefetched / tile whose epilogue comes next
  if (FLAT) {
    pcur = TileCur{gt0, 0, 0};
  } else {
    pcur = TileCur{gt0 % a.tiles_x, (gt0 / a.tiles_x) % a.tiles_y, (gt0 / a.tiles_x) / a.tiles_y};
  }
  ecur = pcur;
  int ph = 0, jp = 0;   // prefetch cursor: chunk, flat chunk index
  int ch = 0, eh = 0;   // chunk of the next k-loop / of the k-loop this group ran last
  auto advance_pf = [&]() {
    if (jp < J) prefetch(pcur, ph);
    ++jp;
    if (++ph == a.nch) { ph = 0; tile_next(pcur); }
  };
  advance_pf();  // chunk 0 is requested first, then the weights (weights_to_lds waits for its own loads before it stores, so in the
  // other order the activation request leaves only after the weights have landed).  Measured neutral on the step: 12.56 vs 12.57 ms
  weights_to_lds<512>(sw, a.w + (size_t)blockIdx.y * wrows * 32, wrows);
  int etiles_left = ntg;  // RED: tiles of this group whose epilogue is still to come
  if (REDK && etiles_left > 0) red_prefetch(ecur);

  const int e4 = a.epi & (DY_EPI_ACCUM | DY_EPI_STATS | DY_EPI_BIAS | DY_EPI_SILU);
  const bool fast_epi = !(a.epi & DY_EPI_F32OUT) && a.cout % 8 == 0 && !(a.ldy & 7) &&
                        (e4 == 0 || e4 == DY_EPI_ACCUM || e4 == DY_EPI_STATS || e4 == (DY_EPI_STATS | DY_EPI_BIAS) ||
                         e4 == (DY_EPI_BIAS | DY_EPI_SILU));
  const bool f32_epi = (a.epi & DY_EPI_F32OUT) && !(a.epi & (DY_EPI_SILU | DY_EPI_ACCUM | DY_EPI_STATS)) && a.cout % 4 == 0 &&
                       !(a.ldy & 3) && !((uintptr_t)a.y & 15);
  f32x4 acc[MT][NT];
  for (int s = -1; s <= 2 * Jmax; ++s) {
    const int r = s - g;
    TSTAMP(0)
    if (r >= 0 && !(r & 1)) {
      // ------------------------------------------------------------ compute slot: LDS -> MFMA only
      if ((r >> 1) < J) {
        if (ch == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const char* wh = sw + (size_t)(ch * KSTEPS) * (16 * MT) * 64;
        // B fragments are double-buffered; every A fragment is reloaded in place right after the NT MFMAs that read it
        // (its next use is 3*NT MFMAs away, beyond the LDS latency), which frees 4*MT registers for the epilogue
        half8 af[MT], bf[2][NT];
        auto frag_off = [&](int ks) {
          const int kk = ks * 32 + q * 8;
          int tap = kk / CC;
          const int c = kk - tap * CC;
          if (tap > KS * KS - 1) tap = KS * KS - 1;
          return ((tap / KS) * HW_ + (tap % KS)) * PS + (SWZ ? 0 : c * 2);  // (SWZ: one tap per k-step, c = q * 8 lives in stbx)
        };
        auto bbase = [&](int ks) { return SWZ ? stbx[ks % KS] : stb; };
        auto bptr = [&](int ks, int t) { return bbase(ks) + (bdelta(t) + frag_off(ks)) + (straddles(t) ? fwadj : 0); };
#pragma unroll
        for (int m = 0; m < MT; ++m) af[m] = *reinterpret_cast<const half8*>(wh + (m * 16) * 64 + aoff);
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[0][t] = *reinterpret_cast<const half8*>(bptr(0, t));
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          if (ks + 1 < KSTEPS) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[(ks + 1) & 1][t] = *reinterpret_cast<const half8*>(bptr(ks + 1, t));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[m], bf[ks & 1][t], acc[m][t], 0, 0, 0);
            if (ks + 1 < KSTEPS) af[m] = *reinterpret_cast<const half8*>(wh + ((ks + 1) * (16 * MT) + m * 16) * 64 + aoff);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        eh = ch;
        if (++ch == a.nch) ch = 0;
        TSTAMP(1)
      }
    } else if (r >= -1 && (r & 1)) {
      // ------------------------------------------------------------ memory slot
      // This wave shares its SIMD with a wave of the other group that is streaming MFMAs; at equal priority the arbiter
      // gave this slot's ~150 instructions about one issue per 25 cycles.  Raised priority lets them go first -- they are
      // few -- and the matrix stream fills every other cycle.
      __builtin_amdgcn_s_setprio(3);
      const int j = (r - 1) >> 1;  // chunk whose k-loop this group finished in the previous slot (-1: none yet)
      if (j + 1 < J) {  // pf holds chunk j+1: its k-loop runs in the next slot
        char* const wb = st + (SWZ ? (gtid / CPP) * 64 + (((gtid % CPP) << 4) ^ (((gtid / CPP) & 4) << 3))
                                   : swz<CC, STRIDE>(gtid / CPP, gtid % CPP));  // granule i lies GTHR/CPP pixels further: immediates
#pragma unroll
        for (int i = 0; i < NPF; ++i)
          if (gtid + i * GTHR < NCHUNK16) *reinterpret_cast<uint4*>(wb + i * (GTHR / CPP) * PS) = pf[i];
        TSTAMP(3)
      }
      if (j >= 0 && j < J && eh == a.nch - 1) {
        const int n = ecur.n, oy0 = ecur.by * TH, ox0 = ecur.bx * TW, pix0 = ecur.bx * HW_;
        tile_next(ecur);
        if (fast_epi || f32_epi) {
          // hot training epilogues (raw conv + BN partial sums, dgrad store, dgrad accumulate): fp16 out, whole channel
          // groups, no bias / activation.  Two things shape it:
          //  * the matrix pipe of this SIMD is busy with the other group's k-loop and leaves the vector port about one
          //    issue per MFMA, so everything is 2-wide (v_cvt_pk_f16_f32, v_pk_add/fma_f32) and the accumulate / stats
          //    variants are separate instantiations (no per-N-tile flag branches and their register copies);
          //  * in the MFMA result layout the four lanes that own one pixel's channels are 16 lanes apart, so a direct
          //    store is 64 separate 16-byte writes per instruction (measured: 21 % of the kernel).  Each N-tile is
          //    therefore turned through a per-wave LDS scratch so that PPR consecutive lanes write one pixel's whole
          //    channel block: full-line stores, 8x fewer write requests.
          auto fast = [&](auto acc_tag, auto stats_tag, auto bias_tag, auto silu_tag, auto red_tag, auto resv_tag, auto segy_tag) {
            constexpr bool RESV = decltype(resv_tag)::value;   // the addend is ConvArgs::res (its own pointer / pitch), not the old output
            constexpr bool SEGY = decltype(segy_tag)::value;   // 1x1 only: the output is a segmented concatenation (ConvArgs::ys)
            constexpr bool ACCUM = decltype(acc_tag)::value || RESV, STATS = decltype(stats_tag)::value;
            constexpr bool BIAS = decltype(bias_tag)::value, SILU = decltype(silu_tag)::value, RED = decltype(red_tag)::value;
            constexpr int PPR = RB / 16;                       // 16-byte pieces per pixel row
            constexpr int PIXPASS = 64 / PPR;                  // pixels one store instruction covers
            constexpr int NPASS = PIXPASS >= 16 ? 1 : 16 / PIXPASS;
            // destination = wave-uniform 64-bit base (SGPRs) + a per-lane 32-bit offset that never changes, so one store
            // costs one compare; row validity, tile origin and the N-tile / pass displacement are scalar arithmetic
            const int dpix = lane / PPR, piece = lane % PPR;
            const unsigned loff = (unsigned)((dpix * a.ldy + blockIdx.y * (16 * MT) + piece * 8) * 2);
            const bool chok = (int)(blockIdx.y * (16 * MT) + piece * 8) < a.cout;  // padded cout groups: whole 8-channel pieces drop
            const int row0 = FLAT ? 0 : oy0 + wgs * TROWS;
            const long tbase = FLAT ? (long)(pix0 + wgs * (NT * 16)) * a.ldy : ((long)(n * a.Ho + row0) * a.Wo + ox0) * a.ldy;
            char* const ybase = reinterpret_cast<char*>(a.y) + tbase * 2;
            // RESV: where the addend of this lane's pieces lives (same pixel / channel walk over res with its own pitch)
            const unsigned rloff = RESV ? (unsigned)((dpix * a.ldres + blockIdx.y * (16 * MT) + piece * 8) * 2) : 0u;
            const long rtbase = !RESV ? 0 : (FLAT ? (long)(pix0 + wgs * (NT * 16)) * a.ldres : ((long)(n * a.Ho + row0) * a.Wo + ox0) * a.ldres);
            const char* const rbase = reinterpret_cast<const char*>(a.res) + rtbase * 2;
            const int fwrows = a.Ho - row0 < 0 ? 0 : (a.Ho - row0 < TROWS ? a.Ho - row0 : TROWS);  // FW: rows of the strip inside the image
            const int collim = FLAT ? a.npix - (pix0 + wgs * (NT * 16)) : (FW ? fwrows * FW : a.Wo - ox0);  // lanes' pixels below this are real
            const bool full = LIN ? collim >= NT * 16 : (collim >= TW && oy0 + TH <= a.Ho);
            char* const xw = xs + p * XROW + q * (NC * 2);
            const char* const xr = xs + dpix * XROW + piece * 16;
            typedef uint2 __attribute__((may_alias)) uint2_a;  // (the 8-byte write and the 16-byte read are different C++ types)
            typedef uint4 __attribute__((may_alias)) uint4_a;
            // lanes exchange data through LDS inside one wave.  The hardware keeps a wave's LDS operations in order, but the
            // compiler reasons per thread and would move N-tile t+1's write above N-tile t's read: wavefront-scope fences
            // restricted to the LDS address space pin the order (an unqualified fence also emits s_waitcnt vmcnt(0) and
            // waits for every earlier global store).
            auto lds_order = []() {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
              __builtin_amdgcn_wave_barrier();
              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            };
            auto tile_col = [](int t) { return LIN ? t * 16 : (t & 1) * 16; };   // first pixel column of N-tile t
            auto convert_write = [&](int t) {
              const bool rowok = LIN ? true : row0 + (t >> 1) < a.Ho;
              union { half2_ h[NC / 2]; uint4 u4[NC / 8 > 0 ? NC / 8 : 1]; uint2 u2; } hv;
              const float keep = (full || (rowok && tile_col(t) + p < collim)) ? 1.f : 0.f;
              const f32x2 k2 = {keep, keep};
#pragma unroll
              for (int m = 0; m < MT; ++m) {
                f32x2 lo = {acc[m][t][0], acc[m][t][1]}, hi = {acc[m][t][2], acc[m][t][3]};
                if (BIAS) {
                  const f32x4 b = *reinterpret_cast<const f32x4*>(bsh + q * NC + m * 4);
                  lo += (f32x2){b[0], b[1]};
                  hi += (f32x2){b[2], b[3]};
                }
                if (SILU) {  // the division written out in packed fp32 (common.h: the quotient's bits, about half the instructions of
                             // z / (1 + exp(-z)) -- this runs in the memory slot, beside the other group's MFMAs)
                  lo = act_fwd2_fast<DY_ACT_SILU>(lo);
                  hi = act_fwd2_fast<DY_ACT_SILU>(hi);
                }
                hv.h[m * 2] = __builtin_convertvector(lo, half2_);
                hv.h[m * 2 + 1] = __builtin_convertvector(hi, half2_);
                if (STATS) {  // BN partial sums from the fp32 accumulators (before the fp16 rounding of the stored tensor);
                              // branch-free 0/1 mask: a full/partial-tile branch doubles the live copies of s1/s2
                  const f32x2 l2 = lo * k2, h2 = hi * k2;
                  s1[m * 2] += l2; s1[m * 2 + 1] += h2;
                  s2[m * 2] += l2 * lo; s2[m * 2 + 1] += h2 * hi;
                }
              }
              if (NC == 4) {
                *reinterpret_cast<uint2_a*>(xw) = hv.u2;
              } else {
#pragma unroll
                for (int jj = 0; jj < NC / 8; ++jj) reinterpret_cast<uint4_a*>(xw)[jj] = hv.u4[jj];
              }
            };
            union U4 { uint4 u; half2_ h[4]; };
            auto dest = [&](int t, int ps, bool& valid) {
              const bool rowok = LIN ? true : row0 + (t >> 1) < a.Ho;
              const int c0 = tile_col(t) + ps * PIXPASS;              // compile-time column of lane group 0
              const long soff = LIN ? (long)c0 * a.ldy : ((long)(t >> 1) * a.Wo + c0) * a.ldy;  // scalar
              valid = rowok && chok && dpix < 16 && c0 + dpix < collim;
              if (SEGY) {  // (FLAT) pixel pix0 + wgs * 64 + c0 + dpix of the lane's segment (a segment stays below 4 GB: 32-bit offset)
                char* const pzs = sbase + (size_t)((unsigned)(pix0 + wgs * (NT * 16) + c0 + dpix) * sld2);
                return reinterpret_cast<uint4*>(valid ? pzs : reinterpret_cast<char*>(const_cast<void*>(a.ys.ptr[0])));
              }
              // lanes without a destination get the tensor base: the accumulate variant LOADS through this pointer before
              // it tests `valid`, and rows of a partial tile past the last image lie outside the allocation
              char* const pz = ybase + soff * 2 + loff;
              return reinterpret_cast<uint4*>(valid ? pz : reinterpret_cast<char*>(a.y));
            };
            auto addend = [&](int t, int ps) {  // what is added to the values of (t, ps): the old output, or (RESV) the residual tensor
              bool valid;
              if (SEGY) {  // only the segments that accumulate have an old value
                uint4* const pd = dest(t, ps, valid);
                return (valid && sacc) ? *pd : make_uint4(0, 0, 0, 0);
              }
              if (!RESV) return *dest(t, ps, valid);
              const bool rowok = LIN ? true : row0 + (t >> 1) < a.Ho;
              const int c0 = tile_col(t) + ps * PIXPASS;
              const long soff = LIN ? (long)c0 * a.ldres : ((long)(t >> 1) * a.Wo + c0) * a.ldres;
              valid = rowok && chok && dpix < 16 && c0 + dpix < collim;
              return *reinterpret_cast<const uint4*>(valid ? rbase + soff * 2 + rloff : reinterpret_cast<const char*>(a.res));
            };
            // software pipeline over the N-tiles: the scratch is written for tile t+1 as soon as tile t's reads have ISSUED
            // (not returned), so one LDS round trip hides behind the next tile's conversion instead of four in a row
            U4 d[2][NPASS], o[ACCUM ? NT : 1][NPASS];
            TSTAMP(6)
            if (ACCUM) {  // every old value of the tile is requested up front: one exposed HBM latency per tile, not one per N-tile
#pragma unroll
              for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) o[t][ps].u = addend(t, ps);
            }
            convert_write(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              lds_order();
#pragma unroll
              for (int ps = 0; ps < NPASS; ++ps) d[t & 1][ps].u = *reinterpret_cast<const uint4_a*>(xr + ps * PIXPASS * XROW);
              lds_order();
              if (t + 1 < NT) convert_write(t + 1);
#pragma unroll
              for (int ps = 0; ps < NPASS; ++ps) {
                bool valid;
                uint4* const yp = dest(t, ps, valid);
                U4 v = d[t & 1][ps];
                if (ACCUM) {
#pragma unroll
                  for (int k = 0; k < 4; ++k)
                    v.h[k] = __builtin_convertvector(__builtin_convertvector(v.h[k], f32x2) + __builtin_convertvector(o[ACCUM ? t : 0][ps].h[k], f32x2), half2_);
                }
                if (valid && !(a.epi & DY_EPI_DEBUG_NOSTORE)) *yp = v.u;
                if (RED) {  // g = dy * act'(x * scale + shift) on the fp16 values just stored; sums of g and g * x per channel
                  // (lanes without a destination hold whatever the transpose scratch had: selected away, never multiplied away)
                  const bool use = valid && (int)(blockIdx.y * (16 * MT) + piece * 8) < a.rC;
#pragma unroll
                  for (int k = 0; k < 4; ++k) {
                    const f32x2 x = __builtin_convertvector(rw[RED ? t : 0][RED ? ps : 0].h[k], f32x2);
                    const f32x2 sc2 = *reinterpret_cast<const f32x2*>(rcf + piece * 8 + 2 * k);
                    const f32x2 sh2 = *reinterpret_cast<const f32x2*>(rcf + 16 * MT + piece * 8 + 2 * k);
                    f32x2 gg = __builtin_convertvector(v.h[k], f32x2) * act_grad2<DY_ACT_SILU>(__builtin_elementwise_fma(x, sc2, sh2));
                    f32x2 gx = gg * x;
                    gg = use ? gg : (f32x2){0.f, 0.f};
                    gx = use ? gx : (f32x2){0.f, 0.f};
                    rsg[k] += gg;
                    rsgx[k] += gx;
                  }
                }
              }
            }
          };
          if (f32_epi) {
            // Detect-head outputs: fp32 rows (+ bias), whole channel groups.  Same scalar addressing as above; each lane
            // owns 4*MT consecutive floats of its pixel.
            f32x4 b4[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
              b4[m] = *reinterpret_cast<const f32x4*>(bsh + q * NC + m * 4);
            const int row0 = FLAT ? 0 : oy0 + wgs * TROWS;
            const long tbase = FLAT ? (long)(pix0 + wgs * (NT * 16)) * a.ldy : ((long)(n * a.Ho + row0) * a.Wo + ox0) * a.ldy;
            char* const ybase = reinterpret_cast<char*>(a.y) + tbase * 4;
            const int fwrows = a.Ho - row0 < 0 ? 0 : (a.Ho - row0 < TROWS ? a.Ho - row0 : TROWS);
            const int collim = FLAT ? a.npix - (pix0 + wgs * (NT * 16)) : (FW ? fwrows * FW : a.Wo - ox0);
            const unsigned loff = (unsigned)((p * a.ldy + co0) * 4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int tcol = LIN ? t * 16 : (t & 1) * 16;
              const bool rowok = LIN ? true : row0 + (t >> 1) < a.Ho;
              const long soff = LIN ? (long)tcol * a.ldy : ((long)(t >> 1) * a.Wo + tcol) * a.ldy;
              const bool valid = rowok && tcol + p < collim;
              f32x4* const yp = reinterpret_cast<f32x4*>(ybase + soff * 4 + (valid ? loff : 0u));
              if (valid) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                  if (co0 + m * 4 < a.cout) yp[m] = acc[m][t] + b4[m];
              }
            }
          } else {
            constexpr std::true_type Y{};
            constexpr std::false_type N_{};
            const int e = a.epi & (DY_EPI_ACCUM | DY_EPI_STATS | DY_EPI_BIAS | DY_EPI_SILU);
            if (FLAT && a.ys.nseg > 0) {
              // input gradient of a 1x1 conv over a segmented concatenation: every piece to its segment, stored or added per segment
              if constexpr (FLAT && !REDK) {
                if (seg_any_acc) fast(Y, N_, N_, N_, N_, N_, Y);   // per-lane: the segments that accumulate load their old values
                else fast(N_, N_, N_, N_, N_, N_, Y);
              }
            } else if (e == DY_EPI_ACCUM) fast(Y, N_, N_, N_, N_, N_, N_);
            else if (e == DY_EPI_STATS) fast(N_, Y, N_, N_, N_, N_, N_);
            else if (e == (DY_EPI_STATS | DY_EPI_BIAS)) fast(N_, Y, Y, N_, N_, N_, N_);
            else if (e == (DY_EPI_BIAS | DY_EPI_SILU) && (a.epi & DY_EPI_RES)) {
              // (3x3 stride-1 only -- Bottleneck.cv2 -- so that no other instantiation carries the variant's code and registers)
              if constexpr (KS == 3 && STRIDE == 1) fast(N_, N_, Y, Y, N_, Y, N_);
            } else if (e == (DY_EPI_BIAS | DY_EPI_SILU)) fast(N_, N_, Y, Y, N_, N_, N_);
            else if (REDK) fast(N_, N_, N_, N_, std::integral_constant<bool, REDK>{}, N_, N_);
            else fast(N_, N_, N_, N_, N_, N_, N_);
          }
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            bool valid;
            size_t yoff;
            if (FLAT) {
              const int gp = pix0 + wg * (NT * 16) + t * 16 + p;
              valid = gp < a.npix;
              yoff = (size_t)gp * a.ldy;
            } else {
              const int fpix = t * 16 + p;  // (FW) pixel of the wave's strip
              const int oy = oy0 + wg * TROWS + (FW ? fpix / (FW ? FW : 1) : (t >> 1)), ox = FW ? fpix % (FW ? FW : 1) : ox0 + (t & 1) * 16 + p;
              valid = oy < a.Ho && ox < a.Wo;
              yoff = ((size_t)(n * a.Ho + oy) * a.Wo + ox) * a.ldy;
            }
            float v[NC];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int rr = 0; rr < 4; ++rr)  // bias is re-read per N-tile (L1-resident): 4*MT registers less across the k-loop
                v[m * 4 + rr] = acc[m][t][rr] + (((a.epi & DY_EPI_BIAS) && co0 + m * 4 + rr < a.cout) ? a.bias[co0 + m * 4 + rr] : 0.f);
            if (a.epi & DY_EPI_SILU) {
#pragma unroll
              for (int jj = 0; jj < NC; ++jj) v[jj] = silu_f(v[jj]);
            }
            if (a.epi & DY_EPI_F32OUT) {
              float* yp = reinterpret_cast<float*>(a.y) + yoff + co0;
              if (valid) {
                if (co0 + NC <= a.cout && !(a.ldy & 3)) {
#pragma unroll
                  for (int jj = 0; jj < NC; jj += 4) {
                    float4 o = make_float4(v[jj], v[jj + 1], v[jj + 2], v[jj + 3]);
                    if (a.epi & DY_EPI_ACCUM) {
                      const float4 oldv = *reinterpret_cast<const float4*>(yp + jj);
                      o.x += oldv.x; o.y += oldv.y; o.z += oldv.z; o.w += oldv.w;
                    }
                    *reinterpret_cast<float4*>(yp + jj) = o;
                  }
                } else {
#pragma unroll
                  for (int jj = 0; jj < NC; ++jj)
                    if (co0 + jj < a.cout) yp[jj] = (a.epi & DY_EPI_ACCUM) ? yp[jj] + v[jj] : v[jj];
                }
              }
            } else {
              f16* yp = reinterpret_cast<f16*>(a.y) + yoff + co0;
              if (valid) {
#pragma unroll
                for (int jj = 0; jj < NC; ++jj)
                  if (co0 + jj < a.cout) {
                    const float o = (a.epi & DY_EPI_ACCUM) ? v[jj] + (float)yp[jj] : v[jj];
                    const f16 ho = (f16)o;
                    yp[jj] = ho;
                    if (a.epi & DY_EPI_STATS) {
                      const float rv = (float)ho;
                      s1[jj >> 1][jj & 1] += rv;
                      s2[jj >> 1][jj & 1] += rv * rv;
                    }
                  }
              }
            }
          }
        }
        if (REDK && --etiles_left > 0) red_prefetch(ecur);  // ecur is already the next tile this group finishes
      }
    }
    TSTAMP(2)
    // last in the memory slot: chunk j+2 flies during the coming k-loop, and neither the epilogue's own loads (accumulate)
    // nor its registers had to share the slot with the staging registers
    if (r >= -1 && (r & 1) && ((r - 1) >> 1) + 1 < J) advance_pf();
    TSTAMP(4)
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    TSTAMP(5)
  }
#ifdef DY_CONV_TIMING
  if (lane == 0 && (wave & 3) == 0 && g == 0) {
    for (int i = 0; i < 7; ++i) atomicAdd(&dy_timing[i], tacc[i]);
    atomicAdd(&dy_timing[7], 1ull);
  }
#endif

  if (REDK) {
    // lanes with the same channel piece (lane % PPR) -> one value; waves -> LDS; then per channel sum(g * xhat) = (sum(g x) - mean sum(g))
    // * invstd and one fp64 atomic add per (sum, channel) into copy blockIdx.x % DY_BN_COPIES of the layer's accumulator
    constexpr int PPR = (4 * NC * 2) / 16;
    float* red = reinterpret_cast<float*>(dsm + wrows * 64);  // [8 waves][2][16*MT]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float u = rsg[k][e], w = rsgx[k][e];
#pragma unroll
        for (int msk = PPR; msk < 64; msk <<= 1) {
          u += __shfl_xor(u, msk, 64);
          w += __shfl_xor(w, msk, 64);
        }
        if (lane < PPR) {
          red[(wave * 2 + 0) * (16 * MT) + lane * 8 + 2 * k + e] = u;
          red[(wave * 2 + 1) * (16 * MT) + lane * 8 + 2 * k + e] = w;
        }
      }
    __syncthreads();
    if (tid < 16 * MT) {
      const int c = blockIdx.y * (16 * MT) + tid;
      if (c < a.rC) {
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
          sg += red[(w * 2 + 0) * (16 * MT) + tid];
          sgx += red[(w * 2 + 1) * (16 * MT) + tid];
        }
        sgx = (sgx - a.rcoef[2 * a.rC + c] * sg) * a.rcoef[3 * a.rC + c];
        double* dst = a.racc + (size_t)(blockIdx.x % DY_BN_COPIES) * 2 * a.rC;
        unsafeAtomicAdd(dst + c, (double)sg);
        unsafeAtomicAdd(dst + a.rC + c, (double)sgx);
      }
    }
  }
  if (a.epi & DY_EPI_STATS) {
    float* red = reinterpret_cast<float*>(dsm + wrows * 64);
    if (true) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const float r1 = quad16_sum(s1[j >> 1][j & 1]), r2 = quad16_sum(s2[j >> 1][j & 1]);
        if (p == 0) {
          red[(wave * 2 + 0) * (16 * MT) + q * NC + j] = r1;
          red[(wave * 2 + 1) * (16 * MT) + q * NC + j] = r2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * 16 * MT) {
      const int which = tid / (16 * MT), chn = tid - which * (16 * MT);
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sum += red[(w * 2 + which) * (16 * MT) + chn];
      stats_out(a, which, blockIdx.y * 16 * MT + chn, sum);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// Weight packing: fp32 OIHW master -> fp16 [cout chunk][cin chunk][k-step][16*MT rows][32], rows permuted so that the
// MFMA D rows held by one lane are consecutive output channels; optional per-Cout scale (BN folding) and the
// transposed + spatially flipped form used by the input-gradient pass.
struct PackArgs {
  const float* w;      // (Cout, Cin, KS, KS)
  const float* scale;  // per-Cout multiplier or null
  f16* out;
  int cout, cin, ks, cc, nch, mt, ngroups, ksteps, transposed;
  int ld_taps, ld_cphys;  // LDConv column conv: reduction index n*ld_cphys + c maps to w[(o*cin + c)*ld_taps + n]
};

static __device__ __forceinline__ void pack_one(const PackArgs& a, int idx) {
  int r = idx;
  const int kk = r & 31;
  r >>= 5;
  const int m = r % (16 * a.mt);
  r /= (16 * a.mt);
  const int ks = r % a.ksteps;
  r /= a.ksteps;
  const int h = r % a.nch;
  const int g = r / a.nch;
  const int mt = m >> 4, i = m & 15, qq = i >> 2, rr = i & 3;
  const int o = g * 16 * a.mt + qq * 4 * a.mt + mt * 4 + rr;  // GEMM row = conv output channel of this pass
  const int flat = ks * 32 + kk;
  const int tap = flat / a.cc, c = h * a.cc + (flat - tap * a.cc);
  float v = 0.f;
  const int ntap = a.ks * a.ks;
  if (a.ld_taps) {  // ks == 1: `c` (forward) or `o` (transposed) runs over n*ld_cphys + channel
    const int kidx = a.transposed ? o : c, oc = a.transposed ? c : o;
    const int n = kidx / a.ld_cphys, ch = kidx - n * a.ld_cphys;
    if (tap == 0 && n < a.ld_taps && ch < a.cin && oc < a.cout) v = a.w[((size_t)oc * a.cin + ch) * a.ld_taps + n];
  } else if (tap < ntap) {
    if (!a.transposed) {
      if (o < a.cout && c < a.cin) {
        v = a.w[((size_t)o * a.cin + c) * ntap + tap];
        if (a.scale) v *= a.scale[o];
      }
    } else {  // pass output channel o = original Cin index, reduction channel c = original Cout index
      if (o < a.cin && c < a.cout) {
        v = a.w[((size_t)c * a.cin + o) * ntap + (ntap - 1 - tap)];
        if (a.scale) v *= a.scale[c];
      }
    }
  }
  a.out[idx] = (f16)v;
}

__global__ void pack_weights_kernel(PackArgs a) {
  const int total = a.ngroups * a.nch * a.ksteps * 16 * a.mt * 32;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) pack_one(a, idx);
}

// all packs of a model in ONE launch: descs[i].first_block is the exclusive prefix of per-descriptor block counts
struct PackDesc {
  PackArgs a;
  int total, first_block;
};
__global__ __launch_bounds__(256) void pack_weights_batched_kernel(const PackDesc* descs, int n) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {  // last descriptor whose first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= (int)blockIdx.x) lo = mid;
    else hi = mid - 1;
  }
  const PackDesc& d = descs[lo];
  const int base = (blockIdx.x - d.first_block) * 1024;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int idx = base + k * 256 + threadIdx.x;
    if (idx < d.total) pack_one(d.a, idx);
  }
}

// ----------------------------------------------------------------------------------------------------------------
static int pp_trows(int cc, int mt, int ks, int stride, int nch);
static int pick_cc(int cin_p, int ks, int stride) {
  static const int forced = getenv("DY_CONV_CC") ? atoi(getenv("DY_CONV_CC")) : 0;  // measurement switch, read once
  if (forced && cin_p % forced == 0) return forced;
  // 64-channel stride-2 3x3: 16-channel chunks.  A stride-2 halo tile is 9 x 65 pixels for 4 x 32 outputs, so the staged bytes per MFMA
  // are four times a stride-1 tile's; with 32-channel chunks the tile + weights only fit for 32-wide cout groups, i.e. 64->128 staged
  // its input four times.  16-channel chunks leave room for the 64-wide group: 78.1 -> 61.0 us (64->128 @80x80), 43.1 -> 32.7 (64->64);
  // narrower inputs lose with them (32->64 @160: 36.4 -> 44.2), gpurun_out/s2_sweep.log.
  if (ks == 3 && stride == 2 && cin_p == 64) return 16;
  const int cap = (ks == 3 && stride == 2) ? 32 : 64;
  if (cin_p <= cap && (cin_p == 8 || cin_p == 16 || cin_p == 32 || cin_p == 64)) return cin_p;
  if (cin_p % 64 == 0 && cap >= 64) return 64;
  if (cin_p % 32 == 0) return 32;
  if (cin_p % 16 == 0) return 16;
  return 8;
}
// cout rows per workgroup = 16*MT.  48 / 80 / 96-channel outputs take MT = 4 with a padded last group (zero weight rows, masked
// stores): these layers are memory-bound, and MT = 1 would stream the whole input once per 16 output channels
static int pick_mt(int cout16) {
  static const int forced = getenv("DY_CONV_MT") ? atoi(getenv("DY_CONV_MT")) : 0;  // measurement switch, read once
  if (forced) return forced;
  return cout16 >= 48 ? 4 : (cout16 == 32 ? 2 : 1);
}

extern "C" int dy_conv_geometry(int cin, int cout, int ks, int stride, int* cin_p, int* cout_p, int* cc, int* nch,
                                int* mt, int* ngroups, int* ksteps, int* packed_elems) {
  if (!(ks == 1 || ks == 3) || !(stride == 1 || stride == 2) || (ks == 1 && stride != 1)) return DY_ERR_ARG;
  const int cp = (cin + 7) / 8 * 8, op = (cout + 15) / 16 * 16;
  int c = pick_cc(cp, ks, stride);
  int m = pick_mt(op);
  static const bool cc64 = getenv("DY_CONV_CC64") != nullptr;  // measurement switch, read once
  // 64-channel 3x3 with a 64-wide cout group: two 32-channel chunks let the 16-row halo tile share LDS with the 72 KiB
  // of weights, so each wave owns 64 pixels (4 N-tiles) and re-reads half as many A fragments per MFMA
  if (ks == 3 && stride == 1 && c == 64 && m == 4 && !cc64) c = 32;
  // Prefer a shape the weights-in-LDS ping-pong kernel can take: wide layers (128+ channels: 3x3 weights of a 64-wide cout
  // group are 147+ KB) narrow the cout group to 32 or 16 rows and, for 3x3, the Cin chunk to 32.  The input is then
  // streamed once per cout group, but these layers sit at 40x40 / 20x20 where the whole activation tensor is L2/MALL
  // resident; the alternative (v1 kernel, A fragments from L2) measured 150-300 TFLOP/s on them.
  // (Only up to 128x128 channels: beyond, the cout groups multiply while the 20x20 maps leave each workgroup a handful of
  // tiles per 74 KB weight load -- 256->256 @20x20 and 128->256 s2 measured 15-55 % slower this way.)
  // Stride-2 3x3 over 128 input channels (yolov8n-p2's 128->256 / 128->128 down-sampling convs at 80x80): the halo tile of a stride-2
  // kernel is four times a stride-1 tile's, so only 16-channel chunks leave room for a 32-wide cout group beside it -- without them
  // 128->128 ran on 16-wide groups (input staged eight times, 84 TFLOP/s) and 128->256 on the v1 kernel (A fragments from L2).
  static const bool s2cc16 = getenv("DY_CONV_S2_CC16") == nullptr || atoi(getenv("DY_CONV_S2_CC16")) != 0;
  const bool s2wide = s2cc16 && ks == 3 && stride == 2 && cp == 128 && op <= 128;  // (128->256 on 32-wide groups measured SLOWER than v1: 227 vs 174 us)
  if (pp_trows(c, m, ks, stride, cp / c) == 0 && cp <= 128 && (op <= 128 || s2wide)) {
    bool found = false;
    for (int mm = m; mm >= 1 && !found; mm >>= 1)
      for (int cc2 = c; cc2 >= (s2wide ? 16 : 32) && !found; cc2 >>= 1) {
        if (cp % cc2) continue;
        if (pp_trows(cc2, mm, ks, stride, cp / cc2)) {
          m = mm;
          c = cc2;
          found = true;
        }
      }
  }
  *cin_p = cp;
  *ngroups = (op + 16 * m - 1) / (16 * m);
  *cout_p = *ngroups * 16 * m;
  *cc = c;
  *nch = cp / c;
  *mt = m;
  *ksteps = (ks * ks * c + 31) / 32;
  *packed_elems = (*ngroups) * (*nch) * (*ksteps) * 16 * m * 32;
  return DY_OK;
}

extern "C" int dy_pack_weights_ld(const float* w, void* out, int cout, int cin, int ld_taps, int ld_cphys, int transposed,
                                  hipStream_t stream) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  const int keff = ld_taps * ld_cphys, coutp = (cout + 7) / 8 * 8;
  const int pin = transposed ? coutp : keff, pout = transposed ? keff : cout;
  if (dy_conv_geometry(pin, pout, 1, 1, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return DY_ERR_ARG;
  PackArgs a{w, nullptr, (f16*)out, cout, cin, 1, cc, nch, mt, ng, kst, transposed, ld_taps, ld_cphys};
  const int blocks = cdiv(pe, 256) < 1024 ? cdiv(pe, 256) : 1024;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// Fill one host-side descriptor (sizeof = dy_pack_desc_bytes()) for dy_pack_weights_batched; returns its block count.
extern "C" int dy_pack_desc_bytes(void) { return (int)sizeof(PackDesc); }
extern "C" int dy_pack_desc_fill(void* desc, const float* w, const float* scale, void* out, int cout, int cin, int ks,
                                 int stride, int transposed, int ld_taps, int ld_cphys, int first_block) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  int pin, pout;
  if (ld_taps) {
    const int keff = ld_taps * ld_cphys, coutp = (cout + 7) / 8 * 8;
    pin = transposed ? coutp : keff;
    pout = transposed ? keff : cout;
    ks = 1;
    stride = 1;
  } else {
    pin = transposed ? cout : cin;
    pout = transposed ? cin : cout;
  }
  if (dy_conv_geometry(pin, pout, ks, transposed ? 1 : stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return DY_ERR_ARG;
  PackDesc* d = reinterpret_cast<PackDesc*>(desc);
  d->a = PackArgs{w, scale, (f16*)out, cout, cin, ks, cc, nch, mt, ng, kst, transposed, ld_taps, ld_cphys};
  d->total = pe;
  d->first_block = first_block;
  return cdiv(pe, 1024);
}
extern "C" int dy_pack_weights_batched(const void* descs_device, int n, int total_blocks, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return DY_ERR_ARG;
  hipLaunchKernelGGL(pack_weights_batched_kernel, dim3(total_blocks), dim3(256), 0, stream, (const PackDesc*)descs_device, n);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_pack_weights(const float* w, const float* scale, void* out, int cout, int cin, int ks, int stride,
                               int transposed, hipStream_t stream) {
  // geometry is that of the pass that will consume the pack: (cin -> cout) forward, or (cout -> cin) stride-1 dgrad
  int cp, op, cc, nch, mt, ng, kst, pe;
  const int pin = transposed ? cout : cin, pout = transposed ? cin : cout;
  if (dy_conv_geometry(pin, pout, ks, transposed ? 1 : stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK)
    return DY_ERR_ARG;
  PackArgs a{w, scale, (f16*)out, cout, cin, ks, cc, nch, mt, ng, kst, transposed, 0, 0};
  const int blocks = cdiv(pe, 256) < 1024 ? cdiv(pe, 256) : 1024;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

#define DY_WLDS_BUDGET (156 * 1024)
#define DY_WLDS_MAX_WGS 512
static bool g_force_v1 = getenv("DY_CONV_V1") != nullptr;

// LDS bytes of the v3 kernel with `nw` waves and `trows` output rows per wave (FLAT: 32*trows pixels per wave)
static size_t wlds_bytes_t(int cc, int mt, int ks, int stride, int nch, int trows, int nw = 8) {
  const bool flat = ks == 1;
  const int th = nw * trows, hw = flat ? nw * 2 * trows * 16 : 31 * stride + ks, hh = flat ? 1 : (th - 1) * stride + ks;
  size_t tile = (size_t)hh * hw * ps_bytes(cc, stride);
  const size_t red = nw * 2 * 16 * mt * 4;
  if (tile < red) tile = red;
  const size_t wts = (size_t)nch * ((ks * ks * cc + 31) / 32) * 16 * mt * 64;
  return tile + wts;
}
// v3 configuration for a geometry: 2 = 8 waves x 2 rows (preferred), 4 = 4 waves x 2 rows (same per-wave tile, fits when the
// 16-row halo does not: 64-channel 3x3), 1 = 8 waves x 1 row, 0 = v3 not applicable
static bool g_no_nw4 = getenv("DY_CONV_NW4") == nullptr;  // measured slower (224 vs 202 us on 64->64 3x3 @160^2): opt-in only
static int v3_trows(int cc, int mt, int ks, int stride, int nch) {
  if (ks == 3 && stride == 2) return wlds_bytes_t(cc, mt, ks, stride, nch, 1) <= DY_WLDS_BUDGET ? 1 : 0;
  if (wlds_bytes_t(cc, mt, ks, stride, nch, 2) <= DY_WLDS_BUDGET) return 2;
  if (!g_no_nw4 && ks == 3 && wlds_bytes_t(cc, mt, ks, stride, nch, 2, 4) <= DY_WLDS_BUDGET) return 4;
  return wlds_bytes_t(cc, mt, ks, stride, nch, 1) <= DY_WLDS_BUDGET ? 1 : 0;
}
static int v3_tile_rows(int cfg) { return cfg == 2 ? 16 : 8; }
static int v3_flat_pix(int cfg) { return cfg == 2 ? 512 : (cfg == 4 ? 256 : 256); }


template <int CC, int MT, int KS, int STRIDE, int TR3, int NW>
static int launch_v3(const ConvArgs& a, int grid_y, hipStream_t s) {
  static bool attr_set = false;
  auto kern = conv_mfma_wlds_kernel<CC, MT, KS, STRIDE, TR3, NW>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DY_WLDS_BUDGET) != hipSuccess)
      return DY_ERR_LAUNCH;
    attr_set = true;
  }
  ConvArgs b = a;
  int ntiles;
  if (KS == 1) {
    ntiles = cdiv(a.npix, NW * 2 * TR3 * 16);
  } else {
    b.tiles_y = cdiv(a.Ho, NW * TR3);
    ntiles = b.tiles_x * b.tiles_y * a.N;
  }
  const int gx = ntiles < DY_WLDS_MAX_WGS ? ntiles : DY_WLDS_MAX_WGS;
  hipLaunchKernelGGL(kern, dim3(gx, grid_y), dim3(NW * 64), wlds_bytes_t(CC, MT, KS, STRIDE, a.nch, TR3, NW), s, b, ntiles);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Input gradient of a stride-2 3x3 convolution by output-parity classes.  dX(oy,ox) = sum_t W'[t] dYv(oy+ty-1, ox+tx-1) over
// the zero-dilated dY: a tap contributes only where both coordinates are even, so an output pixel of parity (py,px) sees
// 1, 2, 2 or 4 of the 9 taps.  The generic kernel multiplied all 9 taps against a tile that is three quarters zeros; here
// a wave's four 16-pixel N-tiles ARE the four parity classes of its 2 x 32 output patch (N-tile (py,px) = row 2wg+py,
// columns px, px+2, ...), every tap (k-step) feeds exactly one of them (4x fewer MFMAs), and the staged tile is the real
// 5 x 17 dY patch (lanes read unit-stride pixels: conflict-free) instead of the 10 x 34 dilated one.
// Same ping-pong structure, packed-weight layout and store transpose as conv_mfma_pp_kernel; epilogue: store or accumulate.
template <int CC, int MT>
__global__ __launch_bounds__(512) void conv_mfma_dg2_kernel(ConvArgs a, int ntiles) {
  constexpr int GW = 4, GTHR = GW * 64, NT = 4, TH = 8, TW = 32;
  constexpr int HW_ = TW / 2 + 1, HH_ = TH / 2 + 1;
  constexpr int KPT = CC / 32, KSTEPS = 9 * KPT;         // k-steps per tap / per Cin chunk
  constexpr int CPP = CC / 8;
  constexpr int NCHUNK16 = HH_ * HW_ * CPP;
  constexpr int NPF = (NCHUNK16 + GTHR - 1) / GTHR;
  constexpr int NC = 4 * MT;
  constexpr int PS = ps_bytes(CC, 1);
  constexpr int TILE_BYTES = HH_ * HW_ * PS;
  constexpr unsigned NEVER = 0x80000000u;
  constexpr int RB = 4 * NC * 2, XROW = RB + 16;
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int wrows = a.nch * KSTEPS * 16 * MT;
  char* const sw = dsm;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int g = __builtin_amdgcn_readfirstlane(wave >> 2), wg = wave & 3, gtid = tid & (GTHR - 1);
  const int wgs = __builtin_amdgcn_readfirstlane(wg);
  char* const st = dsm + wrows * 64 + g * TILE_BYTES;
  char* const xs = dsm + wrows * 64 + 2 * TILE_BYTES + wave * (16 * XROW);
  const char* const stb = st + (wg * HW_ + p) * PS + q * 16;
  const int aoff = p * 64 + ((q ^ ((0 - (p >> 2)) & 3)) << 4);

  unsigned goff[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int id = gtid + i * GTHR;
    const int pixel = id / CPP, part = id - pixel * CPP;
    const int hy = pixel / HW_, hx = pixel - hy * HW_;
    goff[i] = id < NCHUNK16 ? (unsigned)(((hy * a.Wr + hx) * a.ldx + part * 8) * 2) : NEVER;
  }
  struct TileCur { int bx, by, n; };
  uint4 pf[NPF];
  auto prefetch = [&](const TileCur& tc, int h) {
    const int ty0 = tc.by * (TH / 2), tx0 = tc.bx * (TW / 2);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(a.x) + (size_t)tc.n * a.Hr * a.Wr * a.ldx, 0,
                                                        a.Hr * a.Wr * a.ldx * 2, 0x00020000);
    const unsigned org = (unsigned)(((ty0 * a.Wr + tx0) * a.ldx + h * CC) * 2);
    const bool xedge = tx0 + HW_ > a.Wr;   // wave-uniform
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      unsigned off = goff[i] + org;
      if (xedge) {
        const int hx = ((gtid + i * GTHR) / CPP) % HW_;
        off = tx0 + hx < a.Wr ? off : NEVER;
      }
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
      pf[i] = *reinterpret_cast<const uint4*>(&v);
    }
  };

  const int gstride = 2 * gridDim.x;
  const int vb = (a.xcd_map && !(gridDim.x & 7)) ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int gt0 = vb * 2 + g;
  const int ntg = gt0 < ntiles ? (ntiles - gt0 + gstride - 1) / gstride : 0;
  const int J = ntg * a.nch;
  const int Jmax = ((ntiles - vb * 2 + gstride - 1) / gstride) * a.nch;
  const int sdx = gstride % a.tiles_x, sdy = (gstride / a.tiles_x) % a.tiles_y, sdn = (gstride / a.tiles_x) / a.tiles_y;
  auto tile_next = [&](TileCur& c) {
    c.bx += sdx;
    const int cx = c.bx >= a.tiles_x ? 1 : 0;
    c.bx -= cx * a.tiles_x;
    c.by += sdy + cx;
    const int cy = c.by >= a.tiles_y ? 1 : 0;
    c.by -= cy * a.tiles_y;
    c.n += sdn + cy;
  };
  TileCur pcur{gt0 % a.tiles_x, (gt0 / a.tiles_x) % a.tiles_y, (gt0 / a.tiles_x) / a.tiles_y}, ecur = pcur;
  int ph = 0, jp = 0, ch = 0, eh = 0;
  auto advance_pf = [&]() {
    if (jp < J) prefetch(pcur, ph);
    ++jp;
    if (++ph == a.nch) { ph = 0; tile_next(pcur); }
  };
  advance_pf();  // before the weights, as in conv_mfma_pp_kernel
  weights_to_lds<512>(sw, a.w + (size_t)blockIdx.y * wrows * 32, wrows);

  f32x4 acc[MT][NT];
  for (int s = -1; s <= 2 * Jmax; ++s) {
    const int r = s - g;
    if (r >= 0 && !(r & 1)) {
      if ((r >> 1) < J) {
        if (ch == 0) {
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const char* wh = sw + (size_t)(ch * KSTEPS) * (16 * MT) * 64;
        half8 af[2][MT], bf[2];
        auto load_frags = [&](int buf, int ks) {
          const int tap = ks / KPT, sub = ks - tap * KPT, ty = tap / 3, tx = tap - ty * 3;
          // tap (ty,tx) reaches the parity class (ty != 1, tx != 1); its real dY pixel is one further when the tap index is 2
          bf[buf] = *reinterpret_cast<const half8*>(stb + (((ty == 2) ? HW_ : 0) + ((tx == 2) ? 1 : 0)) * PS + sub * 64);
#pragma unroll
          for (int m = 0; m < MT; ++m) af[buf][m] = *reinterpret_cast<const half8*>(wh + (ks * (16 * MT) + m * 16) * 64 + aoff);
        };
        load_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          if (ks + 1 < KSTEPS) load_frags((ks + 1) & 1, ks + 1);
          __builtin_amdgcn_sched_barrier(0);
          const int tap = ks / KPT, ty = tap / 3, tx = tap - ty * 3;
          const int t = ((ty != 1) ? 2 : 0) + ((tx != 1) ? 1 : 0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ks & 1][m], bf[ks & 1], acc[m][t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        eh = ch;
        if (++ch == a.nch) ch = 0;
      }
    } else if (r >= -1 && (r & 1)) {
      __builtin_amdgcn_s_setprio(3);
      const int j = (r - 1) >> 1;
      if (j + 1 < J) {
        char* const wb = st + (gtid / CPP) * PS + (gtid % CPP) * 16;
#pragma unroll
        for (int i = 0; i < NPF; ++i)
          if (gtid + i * GTHR < NCHUNK16) *reinterpret_cast<uint4*>(wb + i * (GTHR / CPP) * PS) = pf[i];
      }
      if (j >= 0 && j < J && eh == a.nch - 1) {
        const int n = ecur.n, oy0 = ecur.by * TH, ox0 = ecur.bx * TW;
        tile_next(ecur);
        auto epi = [&](auto acc_tag) {
          constexpr bool ACCUM = decltype(acc_tag)::value;
          constexpr int PPR = RB / 16, PIXPASS = 64 / PPR, NPASS = PIXPASS >= 16 ? 1 : 16 / PIXPASS;
          typedef uint2 __attribute__((may_alias)) uint2_a;
          typedef uint4 __attribute__((may_alias)) uint4_a;
          auto lds_order = []() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
          };
          const int dpix = lane / PPR, piece = lane % PPR;
          const unsigned loff = (unsigned)((2 * dpix * a.ldy + blockIdx.y * (16 * MT) + piece * 8) * 2);  // N-tile pixels are 2 apart
          const bool chok = (int)(blockIdx.y * (16 * MT) + piece * 8) < a.cout;
          const int row0 = oy0 + wgs * 2;
          char* const ybase = reinterpret_cast<char*>(a.y) + ((long)(n * a.Ho + row0) * a.Wo + ox0) * a.ldy * 2;
          const int collim = a.Wo - ox0;
          char* const xw = xs + p * XROW + q * (NC * 2);
          const char* const xr = xs + dpix * XROW + piece * 16;
          union U4 { uint4 u; half2_ h[4]; };
          auto convert_write = [&](int t) {
            union { half2_ h[NC / 2]; uint4 u4[NC / 8 > 0 ? NC / 8 : 1]; uint2 u2; } hv;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const f32x2 lo = {acc[m][t][0], acc[m][t][1]}, hi = {acc[m][t][2], acc[m][t][3]};
              hv.h[m * 2] = __builtin_convertvector(lo, half2_);
              hv.h[m * 2 + 1] = __builtin_convertvector(hi, half2_);
            }
            if (NC == 4) {
              *reinterpret_cast<uint2_a*>(xw) = hv.u2;
            } else {
#pragma unroll
              for (int jj = 0; jj < NC / 8; ++jj) reinterpret_cast<uint4_a*>(xw)[jj] = hv.u4[jj];
            }
          };
          auto dest = [&](int t, int ps, bool& valid) {
            const int py = t >> 1, px = t & 1;
            const int c0 = px + 2 * ps * PIXPASS;                       // column of lane group 0
            const long soff = ((long)py * a.Wo + c0) * a.ldy;          // scalar
            valid = row0 + py < a.Ho && chok && dpix < 16 && c0 + 2 * dpix < collim;
            char* const pz = ybase + soff * 2 + loff;
            return reinterpret_cast<uint4*>(valid ? pz : reinterpret_cast<char*>(a.y));
          };
          U4 d[2][NPASS], o[ACCUM ? NT : 1][NPASS];
          if (ACCUM) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int ps = 0; ps < NPASS; ++ps) {
                bool valid;
                o[t][ps].u = *dest(t, ps, valid);
              }
          }
          convert_write(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            lds_order();
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) d[t & 1][ps].u = *reinterpret_cast<const uint4_a*>(xr + ps * PIXPASS * XROW);
            lds_order();
            if (t + 1 < NT) convert_write(t + 1);
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
              bool valid;
              uint4* const yp = dest(t, ps, valid);
              U4 v = d[t & 1][ps];
              if (ACCUM) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  v.h[k] = __builtin_convertvector(__builtin_convertvector(v.h[k], f32x2) + __builtin_convertvector(o[ACCUM ? t : 0][ps].h[k], f32x2), half2_);
              }
              if (valid) *yp = v.u;
            }
          }
        };
        if (a.epi & DY_EPI_ACCUM) epi(std::true_type{});
        else epi(std::false_type{});
      }
      if (j + 1 < J) advance_pf();
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
  }
}

// ---- v4 (ping-pong) host side
#define DY_NUM_CUS 256  // MI355X
static bool g_force_v3 = getenv("DY_CONV_V3") != nullptr;
static size_t pp_lds_bytes(int cc, int mt, int ks, int stride, int nch, int trows, int fw = 0) {
  const bool flat = ks == 1;
  const bool swz = ks == 3 && stride == 1 && cc == 32;  // conv_mfma_pp_kernel, SWZ: 64-byte pixels, rows padded to a multiple of 8 pixels
  const int th = 4 * trows, hw = flat ? 4 * 2 * trows * 16 : (swz ? ((fw ? fw : 32) + 2 + 7) / 8 * 8 : 31 * stride + ks), hh = flat ? 1 : (th - 1) * stride + ks;
  size_t tile = (size_t)hh * hw * (swz ? 64 : ps_bytes(cc, stride));
  const size_t red = 8 * 2 * 16 * mt * 4;
  if (tile < red) tile = red;
  const size_t wts = (size_t)nch * ((ks * ks * cc + 31) / 32) * 16 * mt * 64;
  const size_t xpose = 8 * 16 * (size_t)(32 * mt + 16);  // per-wave store-transpose scratch
  return 2 * tile + wts + xpose + 3 * 16 * mt * 4 + (flat ? 256 : 0);  // + this cout group's bias + the RED coefficient table (scale | shift)
                                                                        // + (1x1) the per-chunk table of a segmented input
}
// rows per wave of the ping-pong kernel for a geometry (0 = does not fit: v3/v1 take it)
static int pp_trows(int cc, int mt, int ks, int stride, int nch) {
  if (g_force_v1 || g_force_v3 || (cc == 64 && stride == 2)) return 0;
  static const bool force1 = getenv("DY_PP_TROWS1") != nullptr;
  if (!force1 && !(ks == 3 && stride == 2) && pp_lds_bytes(cc, mt, ks, stride, nch, 2) <= DY_WLDS_BUDGET) return 2;
  return pp_lds_bytes(cc, mt, ks, stride, nch, 1) <= DY_WLDS_BUDGET ? 1 : 0;
}
// workgroups of a ping-pong launch: every workgroup resident at once (one per CU, two when two fit in LDS), each owning
// two tiles per period; this is also the number of BN partial rows the launch writes
static int pp_grid(int cc, int mt, int ks, int stride, int nch, int trows, int ntiles, int fw = 0) {
  // (the swizzled 3x3 tiles would let the 32-wide cout groups sit two per CU: measured SLOWER -- 32->32 @40x40 12.1 -> 15.5 us, @80x80
  // 19.5 -> 22.8, @160x160 50.2 -> 57.7 -- twice the weight preloads for half the tiles each: they stay at one)
  const bool swz_wide = ks == 3 && stride == 1 && cc == 32 && mt >= 2;
  const int per_cu = (!swz_wide && 2 * pp_lds_bytes(cc, mt, ks, stride, nch, trows, fw) <= 160 * 1024) ? 2 : 1;
  int want = cdiv(ntiles, 2);
  const int cap = DY_NUM_CUS * per_cu;
  if (want >= 8) want = (want + 7) & ~7;  // a multiple of the XCD count, so that the XCD-aware tile map applies
  return want < cap ? want : cap;
}

// full-width tiles (conv_mfma_pp_kernel, FW): maps exactly 40 / 80 pixels wide, 3x3 stride-1 with 32-channel chunks.  DY_CONV_FW=0: off
template <int MT, int FW>
static int launch_pp_fw(const ConvArgs& a, int grid_y, hipStream_t s) {
  constexpr int TR = 80 / FW;
  static bool attr_set = false;
  auto kern = conv_mfma_pp_kernel<32, MT, 3, 1, TR, false, FW>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DY_WLDS_BUDGET) != hipSuccess)
      return DY_ERR_LAUNCH;
    attr_set = true;
  }
  ConvArgs b = a;
  b.tiles_x = 1;
  b.tiles_y = cdiv(a.Ho, 4 * TR);
  const int ntiles = b.tiles_y * a.N;
  const int gx = pp_grid(32, MT, 3, 1, a.nch, TR, ntiles, FW);
  hipLaunchKernelGGL(kern, dim3(gx, grid_y), dim3(512), pp_lds_bytes(32, MT, 3, 1, a.nch, TR, FW), s, b, ntiles);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
// DY_CONV_FW: 0 = off, 40 (default) = 40-wide maps, 80 = 40- and 80-wide maps (measured on the step: 80-wide tiles gain nothing --
// 32->32 @80x80 19.3 -> 18.3 us stand-alone, 11.43 vs 11.44 ms per step -- so they stay behind the switch)
static int pp_fw_for(int cc, int mt, int ks, int stride, int nch, int wo, int dil, int epi, bool red) {
  static const int on = getenv("DY_CONV_FW") ? atoi(getenv("DY_CONV_FW")) : 40;
  if (!on || ks != 3 || stride != 1 || cc != 32 || dil == 2 || red) return 0;
  if ((epi & DY_EPI_STATS) && !(epi & DY_EPI_STATS_ACC)) return 0;  // partial-row statistics: dy_conv_num_partials sizes the rows for the 32-wide tiles
  const int fw = wo == 40 ? 40 : ((wo == 80 && on >= 80) ? 80 : 0);
  if (!fw || pp_lds_bytes(cc, mt, ks, stride, nch, 80 / fw, fw) > DY_WLDS_BUDGET) return 0;
  return fw;
}
static int pp_fw_width(const ConvArgs& a, int cc, int mt, int ks, int stride) {
  if (a.Wo != a.Wr || a.Ho != a.Hr) return 0;
  return pp_fw_for(cc, mt, ks, stride, a.nch, a.Wo, a.dil, a.epi, a.racc != nullptr);
}

template <int CC, int MT, int KS, int STRIDE, int TR, bool REDK = false>
static int launch_pp(const ConvArgs& a, int grid_y, hipStream_t s) {
  if constexpr (CC == 32 && KS == 3 && STRIDE == 1 && !REDK) {
    const int fw = pp_fw_width(a, CC, MT, KS, STRIDE);
    if (fw == 40) return launch_pp_fw<MT, 40>(a, grid_y, s);
    if (fw == 80) return launch_pp_fw<MT, 80>(a, grid_y, s);
  }
  if (!REDK && a.racc) {  // the epilogue that also runs a BatchNorm backward reduce: its own instantiation (stride-1 dgrads only), so
    if constexpr (STRIDE == 1) return launch_pp<CC, MT, KS, 1, TR, true>(a, grid_y, s);  // that its registers are not every launch's problem
    return DY_ERR_ARG;
  }
  static bool attr_set = false;
  auto kern = conv_mfma_pp_kernel<CC, MT, KS, STRIDE, TR, REDK>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DY_WLDS_BUDGET) != hipSuccess)
      return DY_ERR_LAUNCH;
    attr_set = true;
  }
  ConvArgs b = a;
  int ntiles;
  if (KS == 1) {
    ntiles = cdiv(a.npix, 4 * 2 * TR * 16);
  } else {
    b.tiles_y = cdiv(a.Ho, 4 * TR);
    ntiles = b.tiles_x * b.tiles_y * a.N;
  }
  const int gx = pp_grid(CC, MT, KS, STRIDE, a.nch, TR, ntiles);
  hipLaunchKernelGGL(kern, dim3(gx, grid_y), dim3(512), pp_lds_bytes(CC, MT, KS, STRIDE, a.nch, TR), s, b, ntiles);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

template <int CC, int MT, int KS, int STRIDE, int TROWS>
static int launch_conv(const ConvArgs& a, int grid_x, int grid_y, hipStream_t s) {
  const int pp = pp_trows(CC, MT, KS, STRIDE, a.nch);
  if (pp == 2 && !(KS == 3 && STRIDE == 2)) return launch_pp<CC, MT, KS, STRIDE, (KS == 3 && STRIDE == 2) ? 1 : 2>(a, grid_y, s);
  if (pp == 1) return launch_pp<CC, MT, KS, STRIDE, 1>(a, grid_y, s);
  const int cfg = (g_force_v1 || (CC == 64 && STRIDE == 2)) ? 0 : v3_trows(CC, MT, KS, STRIDE, a.nch);
  if (cfg == 2 && !(KS == 3 && STRIDE == 2)) return launch_v3<CC, MT, KS, STRIDE, (KS == 3 && STRIDE == 2) ? 1 : 2, 8>(a, grid_y, s);
  if (cfg == 4 && KS == 3 && STRIDE == 1) return launch_v3<CC, MT, KS, STRIDE, (KS == 3 && STRIDE == 1) ? 2 : 1, (KS == 3 && STRIDE == 1) ? 4 : 8>(a, grid_y, s);
  if (cfg == 1) return launch_v3<CC, MT, KS, STRIDE, 1, 8>(a, grid_y, s);
  hipLaunchKernelGGL((conv_mfma_kernel<CC, MT, KS, STRIDE, TROWS>), dim3(grid_x, grid_y), dim3(256), 0, s, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

template <int KS, int STRIDE, int TROWS>
static int dispatch_cc_mt(int cc, int mt, const ConvArgs& a, int gx, int gy, hipStream_t s) {
#define DY_CASE(C, M) \
  if (cc == C && mt == M) return launch_conv<C, M, KS, STRIDE, TROWS>(a, gx, gy, s);
  DY_CASE(8, 1) DY_CASE(8, 2) DY_CASE(8, 4)
  DY_CASE(16, 1) DY_CASE(16, 2) DY_CASE(16, 4)
  DY_CASE(32, 1) DY_CASE(32, 2) DY_CASE(32, 4)
  if (STRIDE == 1) { DY_CASE(64, 1) DY_CASE(64, 2) DY_CASE(64, 4) }
#undef DY_CASE
  return DY_ERR_ARG;
}

extern "C" int dy_conv_num_partials(int n, int h, int w, int cin, int cout, int ks, int stride, int dil);

static bool g_no_dg2 = getenv("DY_CONV_NO_DG2") != nullptr;
static size_t dg2_lds_bytes(int cc, int mt, int nch) {
  return (size_t)nch * 9 * (cc / 32) * 16 * mt * 64 + 2 * (size_t)(5 * 17 * ps_bytes(cc, 1)) + 8 * 16 * (size_t)(32 * mt + 16);
}
template <int CC, int MT>
static int launch_dg2(const ConvArgs& a, int grid_y, hipStream_t s) {
  static bool attr_set = false;
  auto kern = conv_mfma_dg2_kernel<CC, MT>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, DY_WLDS_BUDGET) != hipSuccess)
      return DY_ERR_LAUNCH;
    attr_set = true;
  }
  ConvArgs b = a;
  b.tiles_x = cdiv(a.Wo, 32);
  b.tiles_y = cdiv(a.Ho, 8);
  const int ntiles = b.tiles_x * b.tiles_y * a.N;
  const size_t lds = dg2_lds_bytes(CC, MT, a.nch);
  const int per_cu = 2 * lds <= 160 * 1024 ? 2 : 1;
  int gx = cdiv(ntiles, 2);
  if (gx > DY_NUM_CUS * per_cu) gx = DY_NUM_CUS * per_cu;
  hipLaunchKernelGGL(kern, dim3(gx, grid_y), dim3(512), lds, s, b, ntiles);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

struct RedHost { const void* raw; int ldraw; const float* coef; double* acc; int C; const void* res; int ldres;
                 const DySegs* xs; const DySegs* ys; int cc_override; };
static int conv_forward_impl(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                             float* partials, int n, int h, int w, int cin, int cout, int ks, int stride, int dil,
                             int out_h, int out_w, int epi, int* num_partials, hipStream_t stream, const RedHost* red);
extern "C" int dy_conv_forward(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                               float* partials, int n, int h, int w, int cin, int cout, int ks, int stride, int dil,
                               int out_h, int out_w, int epi, int* num_partials, hipStream_t stream) {
  return conv_forward_impl(x, ldx, w_packed, bias, y, ldy, partials, n, h, w, cin, cout, ks, stride, dil, out_h, out_w, epi, num_partials,
                           stream, nullptr);
}
// 1 when dy_conv_input_grad_red can take this (stride-1) input-gradient geometry: the ping-pong kernel with its transposing epilogue
extern "C" int dy_conv_red_supported(int cin, int cout, int ks) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (dy_conv_geometry(cin, cout, ks, 1, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return 0;
  return (cin == cp && cout % 8 == 0 && pp_trows(cc, mt, ks, 1, nch) != 0) ? 1 : 0;
}
// The input gradient of a stride-1 convolution (dy_conv_forward over the transposed pack, no epilogue flags) that is the ONLY writer of
// the gradient it produces -- the gradient w.r.t. the activated output of a Conv (conv + BatchNorm + SiLU) -- and therefore also runs
// the first pass of that Conv's BatchNorm backward on the values it stores: sums of g = dy * silu'(raw * scale + shift) and g * xhat,
// added into acc [DY_BN_COPIES][2][C] (what dy_bn_act_bwd_reduce_acc would compute from the stored tensor in a pass of its own).
extern "C" int dy_conv_input_grad_red(const void* dy, int lddy, const void* w_packed_t, void* dx, int lddx, int n, int h, int w, int cin,
                                      int cout, int ks, const void* raw, int ldraw, const float* coef, double* acc, int C,
                                      hipStream_t stream) {
  if (!raw || !coef || !acc || C != cout || (ldraw & 7) || ((uintptr_t)raw & 15) || !dy_conv_red_supported(cin, cout, ks) || (lddx & 7))
    return DY_ERR_ARG;
  const RedHost red{raw, ldraw, coef, acc, C, nullptr, 0, nullptr, nullptr, 0};
  return conv_forward_impl(dy, lddy, w_packed_t, nullptr, dx, lddx, nullptr, n, h, w, cin, cout, ks, 1, 1, 0, 0, 0, nullptr, stream, &red);
}
// 1 when dy_conv_forward_res can take this geometry: the ping-pong kernel's transposing epilogue (whole 8-channel pieces)
extern "C" int dy_conv_res_supported(int cin, int cout, int ks, int stride) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (dy_conv_geometry(cin, cout, ks, stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return 0;
  return (ks == 3 && stride == 1 && cin == cp && cout % 8 == 0 && pp_trows(cc, mt, ks, stride, nch) != 0) ? 1 : 0;
}
// Conv.forward_fuse followed by Bottleneck's shortcut add (reference nn/modules/conv.py:57-59, nn/modules/block.py:333-335) in one
// launch: y = fp16(fp16(SiLU(conv(x) + bias)) + res) -- the bits of dy_conv_forward(BIAS | SILU) followed by dy_add.
extern "C" int dy_conv_forward_res(const void* x, int ldx, const void* w_packed, const float* bias, const void* res, int ldres, void* y,
                                   int ldy, int n, int h, int w, int cin, int cout, int ks, int stride, hipStream_t stream) {
  if (!res || !bias || (ldres & 7) || ((uintptr_t)res & 15) || (ldy & 7) || !dy_conv_res_supported(cin, cout, ks, stride)) return DY_ERR_ARG;
  const RedHost red{nullptr, 0, nullptr, nullptr, 0, res, ldres, nullptr, nullptr, 0};
  return conv_forward_impl(x, ldx, w_packed, bias, y, ldy, nullptr, n, h, w, cin, cout, ks, stride, 1, 0, 0,
                           DY_EPI_BIAS | DY_EPI_SILU | DY_EPI_RES, nullptr, stream, &red);
}
// ---- 1x1 convolutions over a never-materialised concatenation (DySegs)
extern "C" int dy_segs_bytes(void) { return (int)sizeof(DySegs); }
static bool segs_valid(const DySegs* s, int total) {
  if (!s || s->nseg < 1 || s->nseg > DY_MAX_SEGS || s->c_end[s->nseg - 1] != total) return false;
  for (int k = 0; k < s->nseg; ++k) {
    const int cb = k ? s->c_end[k - 1] : 0;
    if (s->c_end[k] <= cb || (s->c_end[k] & 7) || (s->ld[k] & 7) || s->ld[k] < s->c_end[k] - cb || !s->ptr[k] || ((uintptr_t)s->ptr[k] & 15)) return false;
  }
  return true;
}
// the Cin chunk a forward launch over these segments stages per step: dy_conv_geometry's own, or 32 where that one (64) would straddle
// a boundary -- the packed weights are the same for both; 0: no such chunk (16-channel chunks of 48- / 80-channel inputs only work
// when they are the geometry's own)
static int segs_chunk(int cin, int cout, const DySegs* s) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (dy_conv_geometry(cin, cout, 1, 1, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK || cin != cp) return 0;
  for (int c = cc; c >= 32 || c == cc; c >>= 1) {
    bool ok = cin % c == 0 && cin / c <= 16 && pp_trows(c, mt, 1, 1, cin / c) != 0;
    for (int k = 0; ok && k < s->nseg; ++k) ok = s->c_end[k] % c == 0;
    if (ok) return c;
    if (c <= 32) break;
  }
  return 0;
}
extern "C" int dy_conv1x1_segs_supported(int cin, int cout, const DySegs* xs) {
  return (segs_valid(xs, cin) && segs_chunk(cin, cout, xs) != 0) ? 1 : 0;
}
// the instantiation dy_conv1x1_forward_segs launches for these segments, spelled as rocprofv3 prints it (see dy_conv_kernel_name)
extern "C" int dy_conv1x1_segs_kernel_name(int cin, int cout, const DySegs* xs, char* out, int cap) {
  if (!out || cap < 8 || !segs_valid(xs, cin)) return DY_ERR_ARG;
  const int c = segs_chunk(cin, cout, xs);
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (!c || dy_conv_geometry(cin, cout, 1, 1, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return DY_ERR_ARG;
  const int pp = pp_trows(c, mt, 1, 1, cin / c);
  if (!pp) return DY_ERR_ARG;
  snprintf(out, cap, "conv_mfma_pp_kernel<%d, %d, 1, 1, %d, false, 0>", c, mt, pp);
  return DY_OK;
}
// dy_conv_forward for a 1x1 convolution whose INPUT is the concatenation xs (n, h, w, cin = xs->c_end[last]): Conv.forward over
// torch.cat(...) (reference nn/modules/block.py:222-226 C2f, :166-171 SPPF, nn/modules/conv.py:338-348 Concat) without the cat.
extern "C" int dy_conv1x1_forward_segs(const DySegs* xs, const void* w_packed, const float* bias, void* y, int ldy, float* partials, int n,
                                       int h, int w, int cin, int cout, int epi, hipStream_t stream) {
  if (!segs_valid(xs, cin)) return DY_ERR_ARG;
  const int c = segs_chunk(cin, cout, xs);
  if (!c) return DY_ERR_ARG;
  for (int k = 0; k < xs->nseg; ++k)  // an up-sampled member (acc bit 1) halves both map sides
    if ((xs->acc[k] & 2) && ((h | w) & 1)) return DY_ERR_ARG;
  const RedHost red{nullptr, 0, nullptr, nullptr, 0, nullptr, 0, xs, nullptr, c};
  return conv_forward_impl(nullptr, 0, w_packed, bias, y, ldy, partials, n, h, w, cin, cout, 1, 1, 1, 0, 0, epi, nullptr, stream, &red);
}
// ... and its input gradient: dx_s (+)= (W^T dy)[channels of segment s] for every segment of dxs (total channels = cout here),
// stored or added per segment (dxs->acc): the dgrad of the same layer writing straight into the concat members' gradient tensors.
extern "C" int dy_conv1x1_input_grad_segs(const void* dy, int lddy, const void* w_packed_t, const DySegs* dxs, int n, int h, int w, int cin,
                                          int cout, hipStream_t stream) {
  if (!segs_valid(dxs, cout)) return DY_ERR_ARG;
  const RedHost red{nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, dxs, 0};
  return conv_forward_impl(dy, lddy, w_packed_t, nullptr, nullptr, 0, nullptr, n, h, w, cin, cout, 1, 1, 1, 0, 0, 0, nullptr, stream, &red);
}
static int conv_forward_impl(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy,
                             float* partials, int n, int h, int w, int cin, int cout, int ks, int stride, int dil,
                             int out_h, int out_w, int epi, int* num_partials, hipStream_t stream, const RedHost* red) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (dy_conv_geometry(cin, cout, ks, stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return DY_ERR_ARG;
  const bool segx = red && red->xs, segy = red && red->ys;
  if (segx) {  // segmented input (1x1): the chunk size the caller chose so that no chunk straddles two segments; same packed layout
    if (ks != 1 || !red->cc_override || cin % red->cc_override || (red->cc_override != cc && (red->cc_override < 32 || cc < 32))) return DY_ERR_ARG;
    cc = red->cc_override;
    nch = cin / cc;
    if (nch > 16) return DY_ERR_ARG;  // the kernel's per-chunk table
    x = red->xs->ptr[0];
    ldx = 8;
    for (int k = 0; k < red->xs->nseg; ++k)  // every segment is addressed through 32-bit buffer offsets
      if ((double)n * h * w * red->xs->ld[k] * 2.0 >= 2147483648.0) return DY_ERR_ARG;
  }
  if (segy) {
    if (ks != 1 || y || epi) return DY_ERR_ARG;
    ldy = 8;
  }
  if ((segx || segy) && pp_trows(cc, mt, ks, stride, nch) == 0) return DY_ERR_ARG;
  if (cin != cp || (ldx & 7) || ((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15)) return DY_ERR_ALIGN;
  if (!(epi & DY_EPI_F32OUT) && (((uintptr_t)y & 15) || (ldy & 3))) return DY_ERR_ALIGN;
  if (dil != 1 && !(dil == 2 && ks == 3 && stride == 1)) return DY_ERR_ARG;
  // the staging loads address the input through 32-bit buffer offsets: the whole tensor (1x1) or one image (3x3) must stay
  // below 2 GiB -- fail loudly rather than wrap
  if ((ks == 1 ? (double)n : 1.0) * h * w * ldx * 2.0 >= 2147483648.0) return DY_ERR_ARG;
  ConvArgs a{};
  a.x = (const f16*)x; a.w = (const f16*)w_packed; a.bias = bias; a.y = y; a.partials = partials;
  a.ldx = ldx; a.ldy = ldy; a.N = n; a.Hr = h; a.Wr = w;
  a.H = dil == 2 ? 2 * h : h; a.W = dil == 2 ? 2 * w : w;
  const int pad = ks / 2;
  a.Ho = (a.H + 2 * pad - ks) / stride + 1;
  a.Wo = (a.W + 2 * pad - ks) / stride + 1;
  if (out_h > 0 && out_w > 0) {  // stride-2 dgrad: the forward input extent (2*h or 2*h-1) cannot be derived from h
    if (out_h > a.Ho || out_w > a.Wo) return DY_ERR_ARG;
    a.Ho = out_h;
    a.Wo = out_w;
  }
  a.cout = cout; a.nch = nch; a.epi = epi; a.dil = dil;
  if (red) {
    a.rraw = (const f16*)red->raw; a.ldrraw = red->ldraw; a.rcoef = red->coef; a.racc = red->acc; a.rC = red->C;
    a.res = (const f16*)red->res; a.ldres = red->ldres;
    if (red->xs) a.xs = *red->xs;
    if (red->ys) a.ys = *red->ys;
  }
  static const bool xcd_map = getenv("DY_CONV_NO_XCDMAP") == nullptr;
  a.xcd_map = xcd_map && ks == 3;  // 1x1 tiles have no halo to share
  int gx;
  if (ks == 1) {
    a.npix = n * a.Ho * a.Wo;
    gx = cdiv(a.npix, 256);
  } else {
    const int th = stride == 1 ? 8 : 4;
    a.tiles_x = cdiv(a.Wo, 32);
    a.tiles_y = cdiv(a.Ho, th);
    gx = a.tiles_x * a.tiles_y * n;
  }
  if (num_partials) *num_partials = dy_conv_num_partials(n, h, w, cin, cout, ks, stride, dil);
  if ((epi & DY_EPI_STATS) && !partials) return DY_ERR_ARG;
  if (gx <= 0) return DY_ERR_ARG;
  if (dil == 2 && !g_no_dg2 && !g_force_v1 && (cc == 32 || cc == 64) && !(epi & ~DY_EPI_ACCUM) && !((uintptr_t)y & 15) && !(ldy & 7) &&
      cout % 8 == 0 && dg2_lds_bytes(cc, mt, nch) <= DY_WLDS_BUDGET && (double)h * w * ldx * 2.0 < 2147483648.0) {
#define DY_DG2(C, M) if (cc == C && mt == M) return launch_dg2<C, M>(a, ng, stream);
    DY_DG2(32, 1) DY_DG2(32, 2) DY_DG2(32, 4) DY_DG2(64, 1) DY_DG2(64, 2) DY_DG2(64, 4)
#undef DY_DG2
  }
  if (ks == 1) return dispatch_cc_mt<1, 1, 2>(cc, mt, a, gx, ng, stream);
  if (stride == 1) return dispatch_cc_mt<3, 1, 2>(cc, mt, a, gx, ng, stream);
  return dispatch_cc_mt<3, 2, 1>(cc, mt, a, gx, ng, stream);
}

// name of the kernel instantiation dy_conv_forward launches for a geometry, spelled as rocprofv3 prints it (host-side
// helper: lets bench.py group its live per-launch timings by the same kernel names as the committed profiles)
extern "C" int dy_conv_kernel_name(int cin, int cout, int ks, int stride, char* out, int cap) {
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (!out || cap < 8 || dy_conv_geometry(cin, cout, ks, stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return DY_ERR_ARG;
  const int pp = pp_trows(cc, mt, ks, stride, nch);
  if (pp) {
    snprintf(out, cap, "conv_mfma_pp_kernel<%d, %d, %d, %d, %d, false, 0>", cc, mt, ks, stride, pp);  // "true": dy_conv_input_grad_red
    return DY_OK;
  }
  const int cfg = (g_force_v1 || (cc == 64 && stride == 2)) ? 0 : v3_trows(cc, mt, ks, stride, nch);
  if (cfg == 2 && !(ks == 3 && stride == 2)) snprintf(out, cap, "conv_mfma_wlds_kernel<%d, %d, %d, %d, 2, 8>", cc, mt, ks, stride);
  else if (cfg == 4 && ks == 3 && stride == 1) snprintf(out, cap, "conv_mfma_wlds_kernel<%d, %d, %d, %d, 2, 4>", cc, mt, ks, stride);
  else if (cfg == 1) snprintf(out, cap, "conv_mfma_wlds_kernel<%d, %d, %d, %d, 1, 8>", cc, mt, ks, stride);
  else snprintf(out, cap, "conv_mfma_kernel<%d, %d, %d, %d, %d>", cc, mt, ks, stride, (ks == 3 && stride == 2) ? 1 : 2);
  return DY_OK;
}

// ... for a given output width and epilogue: maps exactly 40 (80) pixels wide take the full-width tiles of conv_mfma_pp_kernel (last
// template argument), whose launches rocprofv3 lists as a kernel of their own
extern "C" int dy_conv_kernel_name_at(int cin, int cout, int ks, int stride, int out_w, int dil, int epi, char* out, int cap) {
  const int rc = dy_conv_kernel_name(cin, cout, ks, stride, out, cap);
  if (rc != DY_OK) return rc;
  int cp, op, cc, nch, mt, ng, kst, pe;
  dy_conv_geometry(cin, cout, ks, stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe);
  const int pp = pp_trows(cc, mt, ks, stride, nch);
  const int fw = pp ? pp_fw_for(cc, mt, ks, stride, nch, out_w, dil, epi, false) : 0;
  if (fw) snprintf(out, cap, "conv_mfma_pp_kernel<%d, %d, %d, %d, %d, false, %d>", cc, mt, ks, stride, 80 / fw, fw);
  return DY_OK;
}

// number of partial rows dy_conv_forward will write for a given problem (host-side planning helper)
extern "C" int dy_conv_num_partials(int n, int h, int w, int cin, int cout, int ks, int stride, int dil) {
  const int H = dil == 2 ? 2 * h : h, W = dil == 2 ? 2 * w : w, pad = ks / 2;
  const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
  int tiles;
  if (ks == 1) tiles = cdiv(n * Ho * Wo, 256);
  else tiles = cdiv(Wo, 32) * cdiv(Ho, stride == 1 ? 8 : 4) * n;
  int cp, op, cc, nch, mt, ng, kst, pe;
  if (dy_conv_geometry(cin, cout, ks, stride, &cp, &op, &cc, &nch, &mt, &ng, &kst, &pe) != DY_OK) return tiles;
  if (const int pp = pp_trows(cc, mt, ks, stride, nch)) {
    const int nt = ks == 1 ? cdiv(n * Ho * Wo, 4 * 2 * pp * 16) : cdiv(Wo, 32) * cdiv(Ho, 4 * pp) * n;
    return pp_grid(cc, mt, ks, stride, nch, pp, nt);
  }
  const int cfg = (g_force_v1 || (cc == 64 && stride == 2)) ? 0 : v3_trows(cc, mt, ks, stride, nch);
  if (!cfg) return tiles;
  const int t3 = ks == 1 ? cdiv(n * Ho * Wo, v3_flat_pix(cfg)) : cdiv(Wo, 32) * cdiv(Ho, v3_tile_rows(cfg)) * n;
  return t3 > DY_WLDS_MAX_WGS ? DY_WLDS_MAX_WGS : t3;
}
