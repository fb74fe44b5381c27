// LDConv (linear deformable convolution, reference nn/modules/conv.py:350-503) sampling stage, forward and backward.
//
// The reference builds p = p0 + p_n + offset, floors/clamps four corner indices, expands an int64 gather index over
// all channels four times (:456-489) and blends with bilinear weights g (:390-393).  Here one kernel reads the offsets
// (fp32, from the p_conv 3x3 MFMA conv), gathers the four corners straight from the NHWC fp16 map (channel-contiguous
// 16-byte granules) and writes the N samples of every output pixel channel-major as x_off[pix][n*C + c], which turns the
// reference's (N,1)-kernel column conv (:354, 'b c h w n -> b c (h n) w' :494-503) into a plain 1x1 MFMA conv with
// K = N*C.  Backward: the offset gradient through dg/dp with the reference's clamp semantics (no gradient through floor;
// clamp passes gradient inside [0, H-1]) and the input gradient as a deterministic GATHER -- every input pixel walks the
// samples that can touch it, within a radius set by the largest |offset| of the layer (measured on the device by the
// offset-gradient kernel) and capped at rmax; the few samples whose offsets reach further are scattered with fp32 atomics
// into a side accumulator that the gather kernel folds in.
#include "common.h"
#include "dealyolo_hip.h"

struct LdArgs {
  const f16* x;
  const float* off;   // (N,h,w,2*Np): first Np = row offsets, last Np = col offsets
  f16* xo;            // (N,h,w,Np*C)
  const f16* dxo;     // backward: gradient of xo
  float* dx32;        // backward: (N,H,W,C) fp32 accumulator (zeroed by the caller)
  f16* doff;          // backward: (N,h,w,ldoff) fp16, channels [0,2Np)
  const int* pn;      // (2*Np) initial sampling shape p_n (rows then cols)
  int ldx, ldxo, lddxo, ldoff_in, lddoff;
  int N, H, W, h, w, C, Np, stride;
};

// q = x / d, r = x % d.  A 64-bit division is ~100 VALU instructions on this machine and the index decompositions below need four to
// eight of them per 16 bytes moved (the sample kernels were bound by them); every tensor these kernels see has fewer than 2^31
// elements, so the 32-bit form runs -- the 64-bit one stays for the case that one day does not.
static __device__ __forceinline__ void ld_divmod(long x, int d, long& q, int& r) {
  if (x >= 0 && x < 0x7fffffffL) {
    const unsigned xx = (unsigned)x, qq = xx / (unsigned)d;
    q = (long)qq;
    r = (int)(xx - qq * (unsigned)d);
  } else {
    q = x / d;
    r = (int)(x - q * d);
  }
}
static __device__ __forceinline__ void ld_coords(const LdArgs& a, long pix, int n, int& r0, int& r1, int& c0, int& c1,
                                                 float& pr, float& pc, bool& in_r, bool& in_c, long& img) {
  int ox, oy;
  long t;
  ld_divmod(pix, a.w, t, ox);
  ld_divmod(t, a.h, img, oy);
  const float* o = a.off + pix * a.ldoff_in;
  const float ur = (float)(oy * a.stride) + (float)a.pn[n] + o[n];            // :446-454
  const float uc = (float)(ox * a.stride) + (float)a.pn[a.Np + n] + o[a.Np + n];
  const float fr = floorf(ur), fc = floorf(uc);
  const float Hm = (float)(a.H - 1), Wm = (float)(a.W - 1);
  r0 = (int)fminf(fmaxf(fr, 0.f), Hm);
  r1 = (int)fminf(fmaxf(fr + 1.f, 0.f), Hm);
  c0 = (int)fminf(fmaxf(fc, 0.f), Wm);
  c1 = (int)fminf(fmaxf(fc + 1.f, 0.f), Wm);
  pr = fminf(fmaxf(ur, 0.f), Hm);
  pc = fminf(fmaxf(uc, 0.f), Wm);
  in_r = ur >= 0.f && ur <= Hm;  // torch.clamp backward passes gradient where min <= x <= max
  in_c = uc >= 0.f && uc <= Wm;
}

__global__ __launch_bounds__(256) void ldconv_sample_kernel(LdArgs a) {
  const int cpp = a.C >> 3;
  const long total = (long)a.N * a.h * a.w * a.Np * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    int part, n;
    long t, pix;
    ld_divmod(idx, cpp, t, part);
    ld_divmod(t, a.Np, pix, n);
    int r0, r1, c0, c1;
    float pr, pc;
    bool ir, ic;
    long img;
    ld_coords(a, pix, n, r0, r1, c0, c1, pr, pc, ir, ic, img);
    const float g_lt = (1.f + ((float)r0 - pr)) * (1.f + ((float)c0 - pc));
    const float g_rb = (1.f - ((float)r1 - pr)) * (1.f - ((float)c1 - pc));
    const float g_lb = (1.f + ((float)r0 - pr)) * (1.f - ((float)c1 - pc));
    const float g_rt = (1.f - ((float)r1 - pr)) * (1.f + ((float)c0 - pc));
    const f16* xb = a.x + img * a.H * a.W * a.ldx + part * 8;
    const half8 v_lt = *reinterpret_cast<const half8*>(xb + ((long)r0 * a.W + c0) * a.ldx);
    const half8 v_rb = *reinterpret_cast<const half8*>(xb + ((long)r1 * a.W + c1) * a.ldx);
    const half8 v_lb = *reinterpret_cast<const half8*>(xb + ((long)r0 * a.W + c1) * a.ldx);
    const half8 v_rt = *reinterpret_cast<const half8*>(xb + ((long)r1 * a.W + c0) * a.ldx);
    half8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      o[j] = (f16)(g_lt * (float)v_lt[j] + g_rb * (float)v_rb[j] + g_lb * (float)v_lb[j] + g_rt * (float)v_rt[j]);
    *reinterpret_cast<half8*>(a.xo + pix * a.ldxo + n * a.C + part * 8) = o;
  }
}

// Largest |offset| of a layer travels as the bit pattern of its absolute value (integer max orders non-negative floats and
// puts NaN above everything).  A sample is NEAR when both its offsets are <= rmax - 0.01 in magnitude (the slack covers the
// fp32 rounding of base + offset); the gather handles near samples, the atomic side pass the FAR ones (incl. inf / NaN).
static __device__ __forceinline__ bool ld_has_far(unsigned bits, int rmax) { return !(__uint_as_float(bits) <= (float)rmax - 0.01f); }
static __device__ __forceinline__ bool ld_is_far(float o_r, float o_c, float thr) { return !(fabsf(o_r) <= thr) || !(fabsf(o_c) <= thr); }

// One thread per (output pixel, sample n, channel): the C lanes of a sample are adjacent, so every atomic wave-instruction
// adds runs of C contiguous floats (the shape float atomics run at full rate in; one-lane-per-row scatter is ~17x slower,
// MI355X_MICROARCH.md "Global float atomics"), the sample's geometry is computed by the group's first lane and broadcast,
// and the offset gradient is a log2(C)-step shuffle reduction over the group.
__global__ __launch_bounds__(256) void ldconv_sample_bwd_kernel(LdArgs a, int G) {
  // G = lanes per sample: the largest power of two <= 64 dividing C; each lane walks C/G channels G apart.
  const int C = a.C;
  const long total = (long)a.N * a.h * a.w * a.Np * G;
  const long span = ((total + 255) / 256) * 256;
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), leader = lane & ~(G - 1);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < span; idx += (long)gridDim.x * 256) {
    const bool live = idx < total;
    const long t = (live ? idx : total - 1) / G;
    const int n = (int)(t % a.Np);
    const long pix = t / a.Np;
    int r0 = 0, r1 = 0, c0 = 0, c1 = 0, flags = 0;
    float pr = 0.f, pc = 0.f;
    long img = 0;
    if (gl == 0) {
      bool ir, ic;
      ld_coords(a, pix, n, r0, r1, c0, c1, pr, pc, ir, ic, img);
      flags = (ir ? 1 : 0) | (ic ? 2 : 0);
    }
    r0 = __shfl(r0, leader, 64); r1 = __shfl(r1, leader, 64);
    c0 = __shfl(c0, leader, 64); c1 = __shfl(c1, leader, 64);
    pr = __shfl(pr, leader, 64); pc = __shfl(pc, leader, 64);
    flags = __shfl(flags, leader, 64);
    img = pix / ((long)a.h * a.w);
    const float ar0 = 1.f + ((float)r0 - pr), ar1 = 1.f - ((float)r1 - pr);
    const float ac0 = 1.f + ((float)c0 - pc), ac1 = 1.f - ((float)c1 - pc);
    const long o_lt = ((long)r0 * a.W + c0), o_rb = ((long)r1 * a.W + c1), o_lb = ((long)r0 * a.W + c1), o_rt = ((long)r1 * a.W + c0);
    const f16* xb = a.x + img * a.H * a.W * a.ldx;
    float* db = a.dx32 ? a.dx32 + img * a.H * a.W * C : nullptr;
    const f16* gb = a.dxo + pix * a.lddxo + n * C;
    float dpr = 0.f, dpc = 0.f;
    for (int c = gl; c < C; c += G) {
      const float v_lt = (float)xb[o_lt * a.ldx + c], v_rb = (float)xb[o_rb * a.ldx + c];
      const float v_lb = (float)xb[o_lb * a.ldx + c], v_rt = (float)xb[o_rt * a.ldx + c];
      const float g = live ? (float)gb[c] : 0.f;
      dpr += g * (-ac0 * v_lt + ac1 * v_rb - ac1 * v_lb + ac0 * v_rt);
      dpc += g * (-ar0 * v_lt + ar1 * v_rb + ar0 * v_lb - ar1 * v_rt);
      if (db && live) {
        atomicAdd(db + o_lt * C + c, g * (ar0 * ac0));
        atomicAdd(db + o_rb * C + c, g * (ar1 * ac1));
        atomicAdd(db + o_lb * C + c, g * (ar0 * ac1));
        atomicAdd(db + o_rt * C + c, g * (ar1 * ac0));
      }
    }
    for (int o = 1; o < G; o <<= 1) {
      dpr += __shfl_xor(dpr, o, 64);
      dpc += __shfl_xor(dpc, o, 64);
    }
    if (live && gl == 0) {
      f16* d = a.doff + pix * a.lddoff;
      d[n] = (f16)((flags & 1) ? dpr : 0.f);
      d[a.Np + n] = (f16)((flags & 2) ? dpc : 0.f);
    }
  }
}

// Offset gradient only (layers whose input gradient is not needed: the stem): one lane per (sample, 8-channel granule) so that
// the four corner reads and the sample-gradient read are 16-byte loads; the channel sum is 8 in-lane terms plus a shuffle
// reduction over the C/8 lanes of the sample.  The per-channel mapping of the scatter kernel spends five 2-byte loads per
// lane on the same data (1.24 ms -> 0.23 ms for the 640^2 stem).  Where the input gradient IS needed the fp32 atomics set the
// pace (splitting that kernel into this one + a lean scatter measured 14 % slower), so the fused kernel above stays.
__global__ __launch_bounds__(256) void ldconv_doff_kernel(LdArgs a, int LG, unsigned* maxabs) {
  unsigned mbits = 0;
  const long total = (long)a.N * a.h * a.w * a.Np * LG;  // LG = C/8 rounded up to a power of two (<= 64)
  const long span = ((total + 255) / 256) * 256;
  const int cpp = a.C >> 3;
  const int lane = threadIdx.x & 63, gl = lane & (LG - 1);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < span; idx += (long)gridDim.x * 256) {
    const bool live = idx < total;
    const long t = (live ? idx : total - 1) / LG;  // LG is a power of two: a shift
    int n;
    long pix;
    ld_divmod(t, a.Np, pix, n);
    int r0, r1, c0, c1;
    float pr, pc;
    bool ir, ic;
    long img;
    ld_coords(a, pix, n, r0, r1, c0, c1, pr, pc, ir, ic, img);
    if (maxabs) {
      const float* o = a.off + pix * a.ldoff_in;
      const unsigned br = __float_as_uint(o[n]) & 0x7fffffffu, bc = __float_as_uint(o[a.Np + n]) & 0x7fffffffu;
      mbits = max(mbits, max(br, bc));
    }
    const float ar0 = 1.f + ((float)r0 - pr), ar1 = 1.f - ((float)r1 - pr);
    const float ac0 = 1.f + ((float)c0 - pc), ac1 = 1.f - ((float)c1 - pc);
    float dpr = 0.f, dpc = 0.f;
    for (int part = gl; part < cpp; part += LG) {
      const f16* xb = a.x + img * a.H * a.W * a.ldx + part * 8;
      const half8 v_lt = *reinterpret_cast<const half8*>(xb + ((long)r0 * a.W + c0) * a.ldx);
      const half8 v_rb = *reinterpret_cast<const half8*>(xb + ((long)r1 * a.W + c1) * a.ldx);
      const half8 v_lb = *reinterpret_cast<const half8*>(xb + ((long)r0 * a.W + c1) * a.ldx);
      const half8 v_rt = *reinterpret_cast<const half8*>(xb + ((long)r1 * a.W + c0) * a.ldx);
      const half8 g8 = *reinterpret_cast<const half8*>(a.dxo + pix * a.lddxo + n * a.C + part * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float g = live ? (float)g8[j] : 0.f;
        const float lt = (float)v_lt[j], rb = (float)v_rb[j], lb = (float)v_lb[j], rt = (float)v_rt[j];
        dpr += g * (-ac0 * lt + ac1 * rb - ac1 * lb + ac0 * rt);
        dpc += g * (-ar0 * lt + ar1 * rb + ar0 * lb - ar1 * rt);
      }
    }
    for (int o = 1; o < LG; o <<= 1) {
      dpr += __shfl_xor(dpr, o, 64);
      dpc += __shfl_xor(dpc, o, 64);
    }
    if (live && gl == 0) {
      f16* d = a.doff + pix * a.lddoff;
      d[n] = (f16)(ir ? dpr : 0.f);
      d[a.Np + n] = (f16)(ic ? dpc : 0.f);
    }
  }
  if (maxabs) {  // one atomic per wave, and only while it can still raise the running maximum
    for (int o = 1; o < 64; o <<= 1) mbits = max(mbits, (unsigned)__shfl_xor((int)mbits, o, 64));
    if (lane == 0 && mbits > __atomic_load_n(maxabs, __ATOMIC_RELAXED)) atomicMax(maxabs, mbits);
  }
}

// Input gradient by gather.  Thread = (input pixel, GP granules of 8 channels).  A sample (oy, ox, n) has its base point at
// (oy*s + pn_r[n], ox*s + pn_c[n]) and lands within R = ceil(max|offset|) of it, so the samples whose corner rows can equal
// r have base rows in [r-1-R, r+R] (everything below it for the clamped last row r = H-1; the clamped first row needs no
// extension because bases are >= 0), likewise for columns.  Every candidate is then tested EXACTLY with ld_coords'
// arithmetic: row weight = [r0==r](1+(r0-pr)) + [r1==r](1-(r1-pr)) (both terms when the clamp folds the two corners onto
// one row), the same for columns, and the sum of the scatter's four corner products factorises into row*col.  Over-wide
// candidate ranges only cost time.  fp32 accumulation in a fixed order: run-to-run deterministic (the far side pass, when a
// layer has far samples at all, is not).
// accumulate, for input pixel (img, r, c) and channel granules [part*GP, part*GP+GP), every near sample that touches it --
// candidates re-derived from the offsets in global memory (the path for geometries whose candidate table does not fit in LDS)
template <int GP>
static __device__ __forceinline__ void ld_gather_direct(const LdArgs& a, int R, bool has_far, float thr, long img, int r, int c, int part,
                                                        float (&acc)[GP][8]) {
  const int Np = a.Np, s = a.stride, C = a.C;
  const float Hm = (float)(a.H - 1), Wm = (float)(a.W - 1);
  for (int n = 0; n < Np; ++n) {
    const int pnr = a.pn[n], pnc = a.pn[Np + n];
    const int nr = r - 1 - R - pnr, nc_ = c - 1 - R - pnc;
    const int oy_lo = nr > 0 ? (nr + s - 1) / s : 0;
    const int ox_lo = nc_ > 0 ? (nc_ + s - 1) / s : 0;
    int oy_hi = (r + R - pnr) / s, ox_hi = (c + R - pnc) / s;
    if (r == a.H - 1 || oy_hi > a.h - 1) oy_hi = a.h - 1;
    if (c == a.W - 1 || ox_hi > a.w - 1) ox_hi = a.w - 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const long prow = (img * a.h + oy) * a.w;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const long pix = prow + ox;
        const float* o = a.off + pix * a.ldoff_in;
        if (has_far && ld_is_far(o[n], o[Np + n], thr)) continue;  // the side pass owns it
        const float ur = (float)(oy * s) + (float)pnr + o[n];  // identical to ld_coords
        const float fr = floorf(ur);
        const int r0 = (int)fminf(fmaxf(fr, 0.f), Hm), r1 = (int)fminf(fmaxf(fr + 1.f, 0.f), Hm);
        if (r0 != r && r1 != r) continue;
        const float uc = (float)(ox * s) + (float)pnc + o[Np + n];
        const float fc = floorf(uc);
        const int c0 = (int)fminf(fmaxf(fc, 0.f), Wm), c1 = (int)fminf(fmaxf(fc + 1.f, 0.f), Wm);
        if (c0 != c && c1 != c) continue;
        const float pr = fminf(fmaxf(ur, 0.f), Hm), pc = fminf(fmaxf(uc, 0.f), Wm);
        const float wr = (r0 == r ? 1.f + ((float)r0 - pr) : 0.f) + (r1 == r ? 1.f - ((float)r1 - pr) : 0.f);
        const float wc = (c0 == c ? 1.f + ((float)c0 - pc) : 0.f) + (c1 == c ? 1.f - ((float)c1 - pc) : 0.f);
        const float wgt = wr * wc;
        const f16* gp = a.dxo + pix * a.lddxo + n * C + part * (GP * 8);
#pragma unroll
        for (int g = 0; g < GP; ++g) {
          const half8 g8 = *reinterpret_cast<const half8*>(gp + g * 8);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[g][j] += wgt * (float)g8[j];
        }
      }
    }
  }
}

template <int GP>
static __device__ __forceinline__ void ld_gather_store(const LdArgs& a, f16* dx, int lddx, int accumulate, bool has_far, long img, int r, int c,
                                                       int part, float (&acc)[GP][8]) {
  const int C = a.C;
  const long ipix = (img * a.H + r) * a.W + c;
  if (has_far) {
    const float4* f = reinterpret_cast<const float4*>(a.dx32 + ipix * C + part * (GP * 8));
#pragma unroll
    for (int g = 0; g < GP; ++g) {
      const float4 f0 = f[2 * g], f1 = f[2 * g + 1];
      acc[g][0] += f0.x; acc[g][1] += f0.y; acc[g][2] += f0.z; acc[g][3] += f0.w;
      acc[g][4] += f1.x; acc[g][5] += f1.y; acc[g][6] += f1.z; acc[g][7] += f1.w;
    }
  }
  f16* d = dx + ipix * lddx + part * (GP * 8);
#pragma unroll
  for (int g = 0; g < GP; ++g) {
    half8 o8;
    if (accumulate) o8 = *reinterpret_cast<const half8*>(d + g * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = (f16)(acc[g][j] + (accumulate ? (float)o8[j] : 0.f));
    *reinterpret_cast<half8*>(d + g * 8) = o8;
  }
}

template <int GP>
__global__ __launch_bounds__(256) void ldconv_gather_bwd_kernel(LdArgs a, f16* dx, int lddx, int accumulate, const unsigned* maxabs, int rmax) {
  const unsigned mb = *maxabs;
  const bool has_far = ld_has_far(mb, rmax);
  const int R = has_far ? rmax : (int)ceilf(__uint_as_float(mb) + 0.01f);  // >= 1
  const float thr = (float)rmax - 0.01f;
  const int tpp = (a.C >> 3) / GP;
  const long total = (long)a.N * a.H * a.W * tpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int part = (int)(idx % tpp);
    long t = idx / tpp;
    const int c = (int)(t % a.W);
    t /= a.W;
    const int r = (int)(t % a.H);
    const long img = t / a.H;
    float acc[GP][8];
#pragma unroll
    for (int g = 0; g < GP; ++g)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
    ld_gather_direct<GP>(a, R, has_far, thr, img, r, c, part, acc);
    ld_gather_store<GP>(a, dx, lddx, accumulate, has_far, img, r, c, part, acc);
  }
}

// Tiled form of the gather: a workgroup owns a (TH x 32)-pixel tile of one image.  The samples that can reach the tile form a
// compact block of output pixels; their landing geometry (two clamped rows, two clamped columns, the clamped coordinates) is
// derived ONCE, by the whole workgroup, into an LDS table, and every (pixel, candidate) test then is one 16-byte LDS read and a
// dozen VALU ops instead of two strided global loads and the full floor/clamp arithmetic.  Far samples are marked in the table.
// A tile whose candidate block exceeds the table (very wide sample patterns) takes the direct path above.
#define LD_TAB_CAP 1536
struct __attribute__((aligned(16))) LdTab { short r0, r1, c0, c1; float pr, pc; };

template <int GP>
__global__ __launch_bounds__(256) void ldconv_gather_tile_kernel(LdArgs a, f16* dx, int lddx, int accumulate, const unsigned* maxabs, int rmax,
                                                                 int TH, int tiles_x, int tiles_y) {
  __shared__ LdTab tab[LD_TAB_CAP];
  const unsigned mb = *maxabs;
  const bool has_far = ld_has_far(mb, rmax);
  const int R = has_far ? rmax : (int)ceilf(__uint_as_float(mb) + 0.01f);
  const float thr = (float)rmax - 0.01f;
  const int Np = a.Np, s = a.stride, C = a.C;
  const int tpp = (C >> 3) / GP;
  const float Hm = (float)(a.H - 1), Wm = (float)(a.W - 1);
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % tiles_x;
  const int t2 = blockIdx.x / tiles_x;
  const int ty = t2 % tiles_y;
  const long img = t2 / tiles_y;
  int pr_min = 1 << 20, pr_max = -(1 << 20), pc_min = 1 << 20, pc_max = -(1 << 20);
  for (int n = 0; n < Np; ++n) {
    pr_min = min(pr_min, a.pn[n]); pr_max = max(pr_max, a.pn[n]);
    pc_min = min(pc_min, a.pn[Np + n]); pc_max = max(pc_max, a.pn[Np + n]);
  }
  const int r_first = ty * TH, r_last = min(r_first + TH - 1, a.H - 1);
  const int c_first = tx * 32, c_last = min(c_first + 31, a.W - 1);
  int v = r_first - 1 - R - pr_max;
  const int oy_base = v > 0 ? (v + s - 1) / s : 0;
  v = c_first - 1 - R - pc_max;
  const int ox_base = v > 0 ? (v + s - 1) / s : 0;
  int oy_top = (r_last + R - pr_min) / s, ox_top = (c_last + R - pc_min) / s;
  if (r_last == a.H - 1 || oy_top > a.h - 1) oy_top = a.h - 1;
  if (c_last == a.W - 1 || ox_top > a.w - 1) ox_top = a.w - 1;
  const int noy = oy_top - oy_base + 1, nox = ox_top - ox_base + 1;
  const int per_n = noy > 0 && nox > 0 ? noy * nox : 0;
  const bool use_tab = per_n * Np <= LD_TAB_CAP;
  if (use_tab) {
    for (int k = tid; k < per_n * Np; k += 256) {
      const int n = k / per_n, rem = k - n * per_n;
      const int oy = oy_base + rem / nox, ox = ox_base + rem % nox;
      const float* o = a.off + ((img * a.h + oy) * a.w + ox) * a.ldoff_in;
      const float o_r = o[n], o_c = o[Np + n];
      LdTab e;
      if (has_far && ld_is_far(o_r, o_c, thr)) {
        e.r0 = e.r1 = e.c0 = e.c1 = -1;  // never equals a pixel coordinate
        e.pr = e.pc = 0.f;
      } else {
        const float ur = (float)(oy * s) + (float)a.pn[n] + o_r;  // identical to ld_coords
        const float uc = (float)(ox * s) + (float)a.pn[Np + n] + o_c;
        const float fr = floorf(ur), fc = floorf(uc);
        e.r0 = (short)fminf(fmaxf(fr, 0.f), Hm); e.r1 = (short)fminf(fmaxf(fr + 1.f, 0.f), Hm);
        e.c0 = (short)fminf(fmaxf(fc, 0.f), Wm); e.c1 = (short)fminf(fmaxf(fc + 1.f, 0.f), Wm);
        e.pr = fminf(fmaxf(ur, 0.f), Hm); e.pc = fminf(fmaxf(uc, 0.f), Wm);
      }
      tab[k] = e;
    }
    __syncthreads();
  }
  const int ps = tid / tpp, part = tid - ps * tpp;
  const int r = r_first + (ps >> 5), c = c_first + (ps & 31);
  if (ps >= TH * 32 || r > r_last || c > c_last) return;
  float acc[GP][8];
#pragma unroll
  for (int g = 0; g < GP; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  if (!use_tab) {
    ld_gather_direct<GP>(a, R, has_far, thr, img, r, c, part, acc);
  } else {
    for (int n = 0; n < Np; ++n) {
      const int pnr = a.pn[n], pnc = a.pn[Np + n];
      const int nr = r - 1 - R - pnr, nc_ = c - 1 - R - pnc;
      const int oy_lo = nr > 0 ? (nr + s - 1) / s : 0;
      const int ox_lo = nc_ > 0 ? (nc_ + s - 1) / s : 0;
      int oy_hi = (r + R - pnr) / s, ox_hi = (c + R - pnc) / s;
      if (r == a.H - 1 || oy_hi > a.h - 1) oy_hi = a.h - 1;
      if (c == a.W - 1 || ox_hi > a.w - 1) ox_hi = a.w - 1;
      for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        const LdTab* row = tab + n * per_n + (oy - oy_base) * nox - ox_base;
        const long prow = (img * a.h + oy) * a.w;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
          const LdTab e = row[ox];
          if ((e.r0 != r && e.r1 != r) || (e.c0 != c && e.c1 != c)) continue;
          const float wr = (e.r0 == r ? 1.f + ((float)e.r0 - e.pr) : 0.f) + (e.r1 == r ? 1.f - ((float)e.r1 - e.pr) : 0.f);
          const float wc = (e.c0 == c ? 1.f + ((float)e.c0 - e.pc) : 0.f) + (e.c1 == c ? 1.f - ((float)e.c1 - e.pc) : 0.f);
          const float wgt = wr * wc;
          const f16* gp = a.dxo + (prow + ox) * a.lddxo + n * C + part * (GP * 8);
#pragma unroll
          for (int g = 0; g < GP; ++g) {
            const half8 g8 = *reinterpret_cast<const half8*>(gp + g * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[g][j] += wgt * (float)g8[j];
          }
        }
      }
    }
  }
  ld_gather_store<GP>(a, dx, lddx, accumulate, has_far, img, r, c, part, acc);
}

extern "C" int dy_ldconv_sample(const void* x, int ldx, const float* off, int ldoff, const int* pn, void* xo, int ldxo,
                                int n, int H, int W, int h, int w, int C, int Np, int stride, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldxo & 7)) return DY_ERR_ALIGN;
  LdArgs a{};
  a.x = (const f16*)x; a.off = off; a.xo = (f16*)xo; a.pn = pn; a.ldx = ldx; a.ldxo = ldxo; a.ldoff_in = ldoff;
  a.N = n; a.H = H; a.W = W; a.h = h; a.w = w; a.C = C; a.Np = Np; a.stride = stride;
  long blocks = ((long)n * h * w * Np * (C >> 3) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(ldconv_sample_kernel, dim3((int)blocks), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_ldconv_sample_backward(const void* x, int ldx, const float* off, int ldoff, const int* pn,
                                         const void* dxo, int lddxo, float* dx32, void* doff, int lddoff, int n, int H,
                                         int W, int h, int w, int C, int Np, int stride, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (lddxo & 7)) return DY_ERR_ALIGN;
  int G = 64;
  while (C % G) G >>= 1;
  LdArgs a{};
  a.x = (const f16*)x; a.off = off; a.dxo = (const f16*)dxo; a.dx32 = dx32; a.doff = (f16*)doff; a.pn = pn;
  a.ldx = ldx; a.lddxo = lddxo; a.ldoff_in = ldoff; a.lddoff = lddoff;
  a.N = n; a.H = H; a.W = W; a.h = h; a.w = w; a.C = C; a.Np = Np; a.stride = stride;
  if (!dx32) {  // no input gradient wanted (stem): offset gradient only, granule-mapped
    int LG = 1;
    while (LG < (C >> 3) && LG < 64) LG <<= 1;
    long nb = ((long)n * h * w * Np * LG + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(ldconv_doff_kernel, dim3((int)nb), dim3(256), 0, stream, a, LG, (unsigned*)nullptr);
    DY_CHECK_LAUNCH();
    return DY_OK;
  }
  long blocks = ((long)n * h * w * Np * G + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(ldconv_sample_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, a, G);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// dst(fp16, strided) (+)= src(fp32 dense): folds the fp32 scatter accumulator into the activation-gradient buffer
__global__ __launch_bounds__(256) void f32_to_f16_add_kernel(const float* src, f16* dst, int ld, int C, long npix, int accumulate) {
  const int cpp = C >> 3;
  const long total = npix * cpp;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long pix = idx / cpp;
    const int c0 = (int)(idx - pix * cpp) * 8;
    const float4 a0 = *reinterpret_cast<const float4*>(src + pix * C + c0), a1 = *reinterpret_cast<const float4*>(src + pix * C + c0 + 4);
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    f16* d = dst + pix * ld + c0;
    half8 o;
    if (accumulate) o = *reinterpret_cast<const half8*>(d);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)(v[j] + (accumulate ? (float)o[j] : 0.f));
    *reinterpret_cast<half8*>(d) = o;
  }
}
extern "C" int dy_f32_to_f16_add(const float* src, void* dst, int ld, long npix, int C, int accumulate, hipStream_t stream) {
  if ((C & 7) || (ld & 7)) return DY_ERR_ALIGN;
  long blocks = (npix * (C >> 3) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(f32_to_f16_add_kernel, dim3((int)blocks), dim3(256), 0, stream, src, (f16*)dst, ld, C, npix, accumulate);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

__global__ __launch_bounds__(256) void ld_zero_if_far_kernel(float4* p, long n16, const unsigned* maxabs, int rmax) {
  if (!ld_has_far(*maxabs, rmax)) return;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Side pass: fp32-atomic scatter of the FAR samples only (same lane mapping as ldconv_sample_bwd_kernel).
__global__ __launch_bounds__(256) void ldconv_scatter_far_kernel(LdArgs a, int G, const unsigned* maxabs, int rmax) {
  if (!ld_has_far(*maxabs, rmax)) return;
  const int C = a.C;
  const float thr = (float)rmax - 0.01f;
  const long total = (long)a.N * a.h * a.w * a.Np * G;
  const int gl = threadIdx.x & (G - 1);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long t = idx / G;
    const int n = (int)(t % a.Np);
    const long pix = t / a.Np;
    const float* o = a.off + pix * a.ldoff_in;
    if (!ld_is_far(o[n], o[a.Np + n], thr)) continue;
    int r0, r1, c0, c1;
    float pr, pc;
    bool ir, ic;
    long img;
    ld_coords(a, pix, n, r0, r1, c0, c1, pr, pc, ir, ic, img);
    const float ar0 = 1.f + ((float)r0 - pr), ar1 = 1.f - ((float)r1 - pr);
    const float ac0 = 1.f + ((float)c0 - pc), ac1 = 1.f - ((float)c1 - pc);
    float* db = a.dx32 + img * a.H * a.W * C;
    const f16* gb = a.dxo + pix * a.lddxo + n * C;
    for (int c = gl; c < C; c += G) {
      const float g = (float)gb[c];
      atomicAdd(db + ((long)r0 * a.W + c0) * C + c, g * (ar0 * ac0));
      atomicAdd(db + ((long)r1 * a.W + c1) * C + c, g * (ar1 * ac1));
      atomicAdd(db + ((long)r0 * a.W + c1) * C + c, g * (ar0 * ac1));
      atomicAdd(db + ((long)r1 * a.W + c0) * C + c, g * (ar1 * ac0));
    }
  }
}

extern "C" int dy_ldconv_sample_backward_gather(const void* x, int ldx, const float* off, int ldoff, const int* pn, const void* dxo,
                                                int lddxo, void* dx, int lddx, int accumulate, float* dx32, void* doff,
                                                int lddoff, void* scratch, int rmax, int n, int H, int W, int h, int w, int C,
                                                int Np, int stride, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (lddxo & 7) || (lddx & 7)) return DY_ERR_ALIGN;
  if (!dx || !dx32 || !scratch || !doff || rmax < 1 || rmax > 16) return DY_ERR_ARG;
  LdArgs a{};
  a.x = (const f16*)x; a.off = off; a.dxo = (const f16*)dxo; a.dx32 = dx32; a.doff = (f16*)doff; a.pn = pn;
  a.ldx = ldx; a.lddxo = lddxo; a.ldoff_in = ldoff; a.lddoff = lddoff;
  a.N = n; a.H = H; a.W = W; a.h = h; a.w = w; a.C = C; a.Np = Np; a.stride = stride;
  unsigned* maxabs = (unsigned*)scratch;
  if (hipMemsetAsync(maxabs, 0, 4, stream) != hipSuccess) return DY_ERR_LAUNCH;
  int LG = 1;
  while (LG < (C >> 3) && LG < 64) LG <<= 1;
  long nb = ((long)n * h * w * Np * LG + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(ldconv_doff_kernel, dim3((int)nb), dim3(256), 0, stream, a, LG, maxabs);
  // side pass for far samples: both kernels return at once when the layer has none (decided on the device: no host sync,
  // the same launch sequence every step, so the chain captures into a hipGraph)
  const long n16 = (long)n * H * W * C / 4;
  hipLaunchKernelGGL(ld_zero_if_far_kernel, dim3(2048), dim3(256), 0, stream, (float4*)dx32, n16, maxabs, rmax);
  int G = 64;
  while (C % G) G >>= 1;
  long blocks = ((long)n * h * w * Np * G + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ldconv_scatter_far_kernel, dim3((int)blocks), dim3(256), 0, stream, a, G, maxabs, rmax);
  const int cpp = C >> 3;
  const int GP = (cpp % 4 == 0 && cpp >= 8) ? 4 : (cpp % 2 == 0 ? 2 : 1);
  long gb = ((long)n * H * W * (cpp / GP) + 255) / 256;
  if (gb > 65536) gb = 65536;
  const int tpp = cpp / GP;
  static const bool no_tile = getenv("DY_LD_NO_TILE") != nullptr;  // measurement switch: the untiled gather
  const int TH = tpp <= 8 ? (256 / tpp) / 32 : 0;
  const long ntile = TH ? (long)n * ((H + TH - 1) / TH) * ((W + 31) / 32) : 0;
  if (TH && !no_tile && H < 32768 && W < 32768 && ntile < 2147483647L) {
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH;
    if (GP == 4) hipLaunchKernelGGL(ldconv_gather_tile_kernel<4>, dim3((int)ntile), dim3(256), 0, stream, a, (f16*)dx, lddx, accumulate, maxabs, rmax, TH, tiles_x, tiles_y);
    else if (GP == 2) hipLaunchKernelGGL(ldconv_gather_tile_kernel<2>, dim3((int)ntile), dim3(256), 0, stream, a, (f16*)dx, lddx, accumulate, maxabs, rmax, TH, tiles_x, tiles_y);
    else hipLaunchKernelGGL(ldconv_gather_tile_kernel<1>, dim3((int)ntile), dim3(256), 0, stream, a, (f16*)dx, lddx, accumulate, maxabs, rmax, TH, tiles_x, tiles_y);
  } else if (GP == 4) hipLaunchKernelGGL(ldconv_gather_bwd_kernel<4>, dim3((int)gb), dim3(256), 0, stream, a, (f16*)dx, lddx, accumulate, maxabs, rmax);
  else if (GP == 2) hipLaunchKernelGGL(ldconv_gather_bwd_kernel<2>, dim3((int)gb), dim3(256), 0, stream, a, (f16*)dx, lddx, accumulate, maxabs, rmax);
  else hipLaunchKernelGGL(ldconv_gather_bwd_kernel<1>, dim3((int)gb), dim3(256), 0, stream, a, (f16*)dx, lddx, accumulate, maxabs, rmax);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
