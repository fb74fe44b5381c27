// BatchNorm (training statistics / affine apply) + activation, forward and backward, for NHWC fp16 tensors.
//
// Replaces the ATen batch_norm + silu_ / leaky_relu launches behind Conv.forward (reference
// nn/modules/conv.py:49-55; eps 1e-3 / momentum 0.03 from utils/torch_utils.py:347-349) and their autograd
// backward.  Statistics arrive as per-workgroup partial sums written by the conv epilogue, so the raw conv
// output is never re-read just to be averaged.  HBM-bound: every kernel moves 16 bytes per lane.
#include <cstdlib>
#include "common.h"
#include "dealyolo_hip.h"

// ---------------------------------------------------------------------------------------------- finalize
struct BnFinArgs {
  const float* part[3];  // up to three partial sets [P][2][C] (ScalSeq sums three resolutions)
  int nparts[3];
  float pweight[3];
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* coef;  // [4][C]: scale, shift, mean, invstd
  int C;
  float count, eps, momentum;
  int update_running;
};

__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinArgs a) {
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < 3; ++k) {
    if (!a.part[k]) continue;
    double t1 = 0.0, t2 = 0.0;
    for (int i = tid; i < a.nparts[k]; i += 256) {
      t1 += a.part[k][((size_t)i * 2 + 0) * a.C + c];
      t2 += a.part[k][((size_t)i * 2 + 1) * a.C + c];
    }
    s1 += t1 * a.pweight[k];
    s2 += t2 * a.pweight[k];
  }
  __shared__ double r1[256], r2[256];
  r1[tid] = s1;
  r2[tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      r1[tid] += r1[tid + o];
      r2[tid] += r2[tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const double mean = r1[0] / a.count;
    double var = r2[0] / a.count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float sc = a.gamma[c] * invstd;
    a.coef[0 * a.C + c] = sc;
    a.coef[1 * a.C + c] = a.beta[c] - (float)mean * sc;
    a.coef[2 * a.C + c] = (float)mean;
    a.coef[3 * a.C + c] = invstd;
    if (a.update_running) {
      const double unbiased = a.count > 1.f ? var * a.count / (a.count - 1.0) : var;
      a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)mean;
      a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
    }
  }
}

extern "C" int dy_bn_finalize(const float* p0, int n0, float w0, const float* p1, int n1, float w1, const float* p2,
                              int n2, float w2, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float* coef, int C, float count, float eps, float momentum,
                              int update_running, hipStream_t stream) {
  BnFinArgs a{{p0, p1, p2}, {n0, n1, n2}, {w0, w1, w2}, gamma, beta, running_mean, running_var, coef, C, count, eps,
              momentum, update_running};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// coefficients from running statistics (eval mode, unfused BN)
__global__ void bn_eval_coef_kernel(const float* g, const float* b, const float* rm, const float* rv, float* coef, int C,
                                    float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps), sc = g[c] * invstd;
  coef[c] = sc;
  coef[C + c] = b[c] - rm[c] * sc;
  coef[2 * C + c] = rm[c];
  coef[3 * C + c] = invstd;
}
extern "C" int dy_bn_eval_coef(const float* gamma, const float* beta, const float* rm, const float* rv, float* coef,
                               int C, float eps, hipStream_t stream) {
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, gamma, beta, rm, rv, coef, C, eps);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

#define DY_ACT_DISPATCH(KERNEL, grid, stream, args)                                                        \
  do {                                                                                                     \
    if (act == DY_ACT_SILU) hipLaunchKernelGGL(KERNEL<DY_ACT_SILU>, grid, dim3(256), 0, stream, args);      \
    else if (act == DY_ACT_LEAKY) hipLaunchKernelGGL(KERNEL<DY_ACT_LEAKY>, grid, dim3(256), 0, stream, args); \
    else hipLaunchKernelGGL(KERNEL<DY_ACT_NONE>, grid, dim3(256), 0, stream, args);                         \
  } while (0)

// ---------------------------------------------------------------------------------------------- apply (forward)
static __device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == DY_ACT_SILU) return silu_f(z);
  if (act == DY_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
  return z;
}
static __device__ __forceinline__ float act_grad(float z, int act) {
  if (act == DY_ACT_SILU) {
    const float s = sigmoid_f(z);
    return s * (1.f + z * (1.f - s));
  }
  if (act == DY_ACT_LEAKY) return z > 0.f ? 1.f : 0.1f;
  return 1.f;
}

struct ApplyArgs {
  const f16* x;
  const f16* res;  // optional residual added AFTER the activation (Bottleneck shortcut)
  f16* y;
  const float* coef;
  int ldx, ldr, ldy, C, act;
  long npix;
  int rev;  // walk the pixels from the far end (DY_EW_REVERSE bit 0)
  // two output PLANES (C2f.cv1: reference nn/modules/block.py:223 ``cv1(x).chunk(2, 1)``): channels [csplit, C) go to y2 (its own pixel
  // stride) instead of y + csplit -- each half of the chunk is then a contiguous tensor for the kernels that read it alone
  f16* y2 = nullptr;
  int ldy2 = 0, csplit = 1 << 30;
};

// Statistics that arrive in an fp64 accumulator instead of a finished coefficient table (the *_acc entry points): the conv
// epilogue / the backward reduce ADD their per-workgroup sums into acc[copy][2][C] (copy = blockIdx.x % DY_BN_COPIES, so at most
// gridDim.x / DY_BN_COPIES adders meet on one address) and the kernel that consumes the statistics sums the copies in its own
// prologue -- every block for itself, block 0 also leaves the finished values behind (coef / running statistics, or the
// parameter gradients).  This removes the separate finalize launch (~4.7 us + a launch boundary, 127 times per step) without
// any cross-workgroup wait.  The partial sums are fp32 values added in fp64: the order of the adds changes the result by
// at most 2^-53 relative, so after the rounding to fp32 the statistics repeat bit for bit from run to run.
struct BnAccFwd {
  const double* acc;  // [DY_BN_COPIES][2][C]: sum, sum of squares
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* coef_out;    // [4][C] written by block 0 (the backward kernels read it)
  float count, eps, momentum;
};

// (bx, nb) = this workgroup's index and the workgroup count of ITS tensor: blockIdx.x / gridDim.x for a launch of one tensor, the entry's
// own count in a group launch (bn_act_apply_group_kernel)
template <int ACT, bool ACC, bool FAST = false>
static __device__ __forceinline__ void bn_act_apply_body(const ApplyArgs& a, const BnAccFwd& b, const int bx, const int nb) {
  // a thread owns one 8-channel granule for the whole launch: scale/shift live in registers
  const int cpp = a.C >> 3, rows = 256 / cpp;
  const int part = threadIdx.x % cpp, row = threadIdx.x / cpp, c0 = part * 8;
  extern __shared__ float s_coef[];  // ACC: [2][C]
  if (ACC) {
    for (int c = threadIdx.x; c < a.C; c += 256) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < DY_BN_COPIES; ++k) {
        s1 += b.acc[(size_t)(k * 2 + 0) * a.C + c];
        s2 += b.acc[(size_t)(k * 2 + 1) * a.C + c];
      }
      const double mean = s1 / b.count;
      double var = s2 / b.count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)b.eps));
      const float scl = b.gamma[c] * invstd, shf = b.beta[c] - (float)mean * scl;
      s_coef[c] = scl;
      s_coef[a.C + c] = shf;
      if (bx == 0) {
        b.coef_out[c] = scl;
        b.coef_out[a.C + c] = shf;
        b.coef_out[2 * a.C + c] = (float)mean;
        b.coef_out[3 * a.C + c] = invstd;
        const double unbiased = b.count > 1.f ? var * b.count / (b.count - 1.0) : var;
        b.running_mean[c] = (1.f - b.momentum) * b.running_mean[c] + b.momentum * (float)mean;
        b.running_var[c] = (1.f - b.momentum) * b.running_var[c] + b.momentum * (float)unbiased;
      }
    }
    __syncthreads();
  }
  if (row >= rows) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = ACC ? s_coef[c0 + j] : a.coef[c0 + j];
    sh[j] = ACC ? s_coef[a.C + c0 + j] : a.coef[a.C + c0 + j];
  }
  auto at = [&](long p) { return a.rev ? a.npix - 1 - p : p; };
  f16* const yb = c0 < a.csplit ? a.y + c0 : a.y2 + (c0 - a.csplit);  // this thread's granule of the output (its plane)
  const int ly = c0 < a.csplit ? a.ldy : a.ldy2;
  auto one = [&](long pix, const half8& xv) {
    pix = at(pix);
    half8 rv;
    if (a.res) rv = *reinterpret_cast<const half8*>(a.res + pix * a.ldr + c0);
    half8 out;
    if (FAST) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f32x2 z = act_fwd2_fast<ACT>(__builtin_elementwise_fma((f32x2){(float)xv[j], (float)xv[j + 1]}, (f32x2){sc[j], sc[j + 1]},
                                                               (f32x2){sh[j], sh[j + 1]}));
        if (a.res) z += (f32x2){(float)rv[j], (float)rv[j + 1]};
        const half2_ oh = __builtin_convertvector(z, half2_);
        out[j] = oh[0];
        out[j + 1] = oh[1];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float z = act_fwd_t<ACT>((float)xv[j] * sc[j] + sh[j]);
        if (a.res) z += (float)rv[j];
        out[j] = (f16)z;
      }
    }
    *reinterpret_cast<half8*>(yb + pix * ly) = out;
  };
  const long step = (long)nb * rows;
  long pix = (long)bx * rows + row;
  for (; pix + step < a.npix; pix += 2 * step) {  // two pixels per trip: both loads issue before the first use
    const half8 x0 = *reinterpret_cast<const half8*>(a.x + at(pix) * a.ldx + c0);
    const half8 x1 = *reinterpret_cast<const half8*>(a.x + at(pix + step) * a.ldx + c0);
    one(pix, x0);
    one(pix + step, x1);
  }
  if (pix < a.npix) one(pix, *reinterpret_cast<const half8*>(a.x + at(pix) * a.ldx + c0));
}

template <int ACT, bool ACC, bool FAST = false>
__global__ __launch_bounds__(256) void bn_act_apply_kernel(ApplyArgs a, BnAccFwd b) {
  bn_act_apply_body<ACT, ACC, FAST>(a, b, (int)blockIdx.x, (int)gridDim.x);
}
// Several tensors in ONE launch (blockIdx.y = entry): independent Convs of one stage -- the first convs of Detect's six branches, the
// zero-pixel coefficient launches of the six second convs -- whose apply passes would otherwise queue behind each other although the
// small ones cannot fill the chip.
#define DY_BN_GROUP_MAX 8
struct ApplyGroup {
  int n;
  int nblk[DY_BN_GROUP_MAX];
  ApplyArgs a[DY_BN_GROUP_MAX];
  BnAccFwd b[DY_BN_GROUP_MAX];
};
__global__ __launch_bounds__(256) void bn_act_apply_group_kernel(ApplyGroup g) {
  const int e = blockIdx.y;
  if ((int)blockIdx.x >= g.nblk[e]) return;
  bn_act_apply_body<DY_ACT_SILU, true, true>(g.a[e], g.b[e], (int)blockIdx.x, g.nblk[e]);
}

// Which passes walk their pixels from the far end (bit 0 forward apply, bit 1 the stand-alone backward apply, bit 2 backward reduce).
// Round 3 default 4: the backward reduce reads dy right after the dgrad kernels wrote it front to back -- starting at the end, where the
// most recently written lines still sit in L2 / Infinity Cache -- and finishes at the front, which is where the weight-gradient kernel
// that follows starts reading dy and raw: 12.778 -> 12.723 ms per step (alternating runs on one box; bits 0 and 1 measured nothing:
// 12.77 / 12.81 with bit 0, and the stand-alone backward apply of bit 1 is no longer launched).  Element-wise / order-free sums of
// fp32 partials in fp64, so the results do not depend on the order.
static inline int ew_reverse() {
  static const int rev = getenv("DY_EW_REVERSE") ? atoi(getenv("DY_EW_REVERSE")) : 4;
  return rev;
}

// the block-count cap of one kernel family: the environment is read once per name (the launch path runs ~120 times per step)
static long ew_cap(const char* env, long dflt) {
  static struct { const char* name; long cap; } seen[8];
  static int n = 0;
  for (int i = 0; i < n; ++i)
    if (seen[i].name == env) return seen[i].cap;  // callers pass string literals: the pointer identifies the name
  const char* e = getenv(env);
  const long cap = e ? atol(e) : dflt;
  if (n < 8) seen[n++] = {env, cap};
  return cap;
}
static inline int ew_blocks(long npix, int C, const char* env, long dflt) {
  const int rows = 256 / (C >> 3);
  long blocks = (npix + (long)rows * 4 - 1) / ((long)rows * 4);
  // Block-count caps measured per kernel (tools/ew_bench.py, round 2; env DY_EW_BLOCKS_* overrides them): the forward apply (one
  // read stream, one write stream) likes many short blocks (8192: 83-88 us against 92 at 2048 on the 105 M-element layers), the
  // backward apply (two read streams + one write stream) few long ones (512: 120 us against 136 at 2048; 10.5 against 15.7 us on
  // 6.5 M elements); the backward reduce is flat between 1024 and 2048.  Which pixels a block visits (interleaved over the grid
  // or one contiguous span per block) made no measurable difference.
  const long cap = ew_cap(env, dflt);
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

extern "C" int dy_bn_act_apply(const void* x, int ldx, const void* res, int ldr, void* y, int ldy, const float* coef,
                               long npix, int C, int act, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7) || (res && (ldr & 7))) return DY_ERR_ALIGN;
  if ((C >> 3) > 256) return DY_ERR_ARG;
  const int rev = ew_reverse();
  ApplyArgs a{(const f16*)x, (const f16*)res, (f16*)y, coef, ldx, ldr, ldy, C, act, npix, rev & 1};
  const BnAccFwd b{};
  const dim3 grid(ew_blocks(npix, C, "DY_EW_BLOCKS_APPLY", 8192));
  if (act == DY_ACT_SILU) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_SILU, false>), grid, dim3(256), 0, stream, a, b);
  else if (act == DY_ACT_LEAKY) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_LEAKY, false>), grid, dim3(256), 0, stream, a, b);
  else hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_NONE, false>), grid, dim3(256), 0, stream, a, b);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_bn_act_apply_acc(const void* x, int ldx, const void* res, int ldr, void* y, int ldy, const double* acc,
                                   const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float* coef, long npix, int C, int act, float count, float eps, float momentum,
                                   hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7) || (res && (ldr & 7))) return DY_ERR_ALIGN;
  if ((C >> 3) > 256 || !acc || !gamma || !beta || !running_mean || !running_var || !coef) return DY_ERR_ARG;
  const int rev = ew_reverse();
  ApplyArgs a{(const f16*)x, (const f16*)res, (f16*)y, nullptr, ldx, ldr, ldy, C, act, npix, rev & 1};
  const BnAccFwd b{acc, gamma, beta, running_mean, running_var, coef, count, eps, momentum};
  // every block pays the prologue (DY_BN_COPIES x 2 x C doubles from L2 + one fp64 sqrt/divide per channel); measured on one box
  // (round 3, alternating): caps 2048 / 4096 / 8192 -> 13.49 / 13.51 / 13.47 ms per step, i.e. no preference: same cap as before
  const dim3 grid(ew_blocks(npix, C, "DY_EW_BLOCKS_APPLY_ACC", 8192));
  const size_t lds = 2 * (size_t)C * sizeof(float);
  // SiLU's division written out in packed fp32 (common.h, act_fwd2_fast: Markstein's sequence, the correctly rounded quotient on
  // this range): 13.00 -> 12.87 ms per step on one box, all parity tests unchanged.  DY_SILU_FAST=0: the compiler's division.
  static const bool fast = !getenv("DY_SILU_FAST") || atoi(getenv("DY_SILU_FAST"));
  if (act == DY_ACT_SILU && fast) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_SILU, true, true>), grid, dim3(256), lds, stream, a, b);
  else if (act == DY_ACT_SILU) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_SILU, true>), grid, dim3(256), lds, stream, a, b);
  else if (act == DY_ACT_LEAKY) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_LEAKY, true>), grid, dim3(256), lds, stream, a, b);
  else hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_NONE, true>), grid, dim3(256), lds, stream, a, b);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// dy_bn_act_apply_acc (no residual) whose output lives in TWO planes: channels [0, csplit) to y, the rest to y2 (reference
// nn/modules/block.py:223: C2f's cv1(x).chunk(2, 1) -- each half is then a tensor of its own)
extern "C" int dy_bn_act_apply_acc_split(const void* x, int ldx, void* y, int ldy, void* y2, int ldy2, int csplit, const double* acc,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var, float* coef,
                                         long npix, int C, int act, float count, float eps, float momentum, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (ldy & 7) || (ldy2 & 7) || (csplit & 7) || csplit <= 0 || csplit >= C || !y2) return DY_ERR_ALIGN;
  if ((C >> 3) > 256 || !acc || !gamma || !beta || !running_mean || !running_var || !coef) return DY_ERR_ARG;
  const int rev = ew_reverse();
  ApplyArgs a{(const f16*)x, nullptr, (f16*)y, nullptr, ldx, 0, ldy, C, act, npix, rev & 1};
  a.y2 = (f16*)y2; a.ldy2 = ldy2; a.csplit = csplit;
  const BnAccFwd b{acc, gamma, beta, running_mean, running_var, coef, count, eps, momentum};
  const dim3 grid(ew_blocks(npix, C, "DY_EW_BLOCKS_APPLY_ACC", 8192));
  const size_t lds = 2 * (size_t)C * sizeof(float);
  if (act == DY_ACT_SILU) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_SILU, true, true>), grid, dim3(256), lds, stream, a, b);
  else if (act == DY_ACT_LEAKY) hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_LEAKY, true>), grid, dim3(256), lds, stream, a, b);
  else hipLaunchKernelGGL((bn_act_apply_kernel<DY_ACT_NONE, true>), grid, dim3(256), lds, stream, a, b);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_bn_group_max(void) { return DY_BN_GROUP_MAX; }
// dy_bn_act_apply_acc (SiLU, no residual) for n <= dy_bn_group_max() tensors in one launch; array arguments have n entries
extern "C" int dy_bn_act_apply_acc_group(int n, const void* const* x, const int* ldx, void* const* y, const int* ldy, const double* const* acc,
                                         const float* const* gamma, const float* const* beta, float* const* running_mean,
                                         float* const* running_var, float* const* coef, const long* npix, const int* C, const float* count,
                                         const float* eps, const float* momentum, hipStream_t stream) {
  if (n < 1 || n > DY_BN_GROUP_MAX) return DY_ERR_ARG;
  ApplyGroup g{};
  g.n = n;
  int gx = 1, cmax = 0;
  const int rev = ew_reverse();
  for (int e = 0; e < n; ++e) {
    if ((C[e] & 7) || (ldx[e] & 7) || (ldy[e] & 7)) return DY_ERR_ALIGN;
    if ((C[e] >> 3) > 256 || !acc[e] || !gamma[e] || !beta[e] || !running_mean[e] || !running_var[e] || !coef[e]) return DY_ERR_ARG;
    g.a[e] = ApplyArgs{(const f16*)x[e], nullptr, (f16*)y[e], nullptr, ldx[e], 0, ldy[e], C[e], DY_ACT_SILU, npix[e], rev & 1};
    g.b[e] = BnAccFwd{acc[e], gamma[e], beta[e], running_mean[e], running_var[e], coef[e], count[e], eps[e], momentum[e]};
    g.nblk[e] = ew_blocks(npix[e], C[e], "DY_EW_BLOCKS_APPLY_ACC", 8192);
    gx = g.nblk[e] > gx ? g.nblk[e] : gx;
    cmax = C[e] > cmax ? C[e] : cmax;
  }
  hipLaunchKernelGGL(bn_act_apply_group_kernel, dim3(gx, n), dim3(256), 2 * (size_t)cmax * sizeof(float), stream, g);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// ---------------------------------------------------------------------------------------------- backward
// pass 1: per-channel partial sums of g = dy * act'(z) and g * xhat
struct BwdRedArgs {
  const f16* dy;
  const f16* x;
  const float* coef;
  float* partials;  // [gridDim.x][2][C]
  int lddy, ldx, C, act;
  long npix;
  int rev;  // DY_EW_REVERSE bit 2
  double* acc;      // non-null: add the block's sums into acc[blockIdx.x % DY_BN_COPIES][2][C] instead (see BnAccFwd)
  f16* rg;          // RES: gradient of the residual operand (Bottleneck shortcut, y = act(bn(conv)) + res): rg (+)= dy on the way
  int ldrg, rg_acc;
  const f16* dy2 = nullptr;  // two gradient PLANES (see ApplyArgs): channels [csplit, C) of dy come from dy2
  int lddy2 = 0, csplit = 1 << 30;
};

template <int ACT, bool RES = false>
static __device__ __forceinline__ void bn_act_bwd_reduce_body(const BwdRedArgs& a, const int bx, const int nb) {
  const int cpp = a.C >> 3, rows = 256 / cpp, tid = threadIdx.x;
  const int part = tid % cpp, row = tid / cpp, c0 = part * 8;
  // The loop accumulates sum(g) and sum(g * x) on the RAW values; sum(g * xhat) = invstd * (sum(g * x) - mean * sum(g)) is formed
  // once per thread afterwards.  (Keeping mean / invstd out of the loop takes 16 registers and two VALU operations per element off a
  // kernel that sits at the VALU / memory balance point: exp + rcp per element.)  Channel pairs: packed fp32 arithmetic.
  f32x2 sg[4], sgx[4], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sg[j] = sgx[j] = (f32x2){0.f, 0.f};
    sc[j] = (f32x2){a.coef[c0 + 2 * j], a.coef[c0 + 2 * j + 1]};
    sh[j] = (f32x2){a.coef[a.C + c0 + 2 * j], a.coef[a.C + c0 + 2 * j + 1]};
  }
  if (row < rows) {
    // two pixels per trip: four 16-byte loads in flight per lane before the first use
    const long step = (long)nb * rows;
    long pix = (long)bx * rows + row;
    auto at = [&](long p) { return a.rev ? a.npix - 1 - p : p; };
    const f16* const dyb = c0 < a.csplit ? a.dy + c0 : a.dy2 + (c0 - a.csplit);  // this thread's granule of the gradient (its plane)
    const int ldd = c0 < a.csplit ? a.lddy : a.lddy2;
    auto pair = [](const half8& v, int j) { return (f32x2){(float)v[2 * j], (float)v[2 * j + 1]}; };
    // the shortcut's gradient is dy itself: written (first writer) or added (fan-in) here instead of by a launch of its own
    auto pass_on = [&](long p, const half8& dv) {
      f16* dst = a.rg + at(p) * a.ldrg + c0;
      half8 o = dv;
      if (a.rg_acc) {
        const half8 old = *reinterpret_cast<const half8*>(dst);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)old[j] + (float)dv[j]);
      }
      *reinterpret_cast<half8*>(dst) = o;
    };
    for (; pix + step < a.npix; pix += 2 * step) {
      const half8 dv0 = *reinterpret_cast<const half8*>(dyb + at(pix) * ldd);
      const half8 xv0 = *reinterpret_cast<const half8*>(a.x + at(pix) * a.ldx + c0);
      const half8 dv1 = *reinterpret_cast<const half8*>(dyb + at(pix + step) * ldd);
      const half8 xv1 = *reinterpret_cast<const half8*>(a.x + at(pix + step) * a.ldx + c0);
      if (RES) {
        pass_on(pix, dv0);
        pass_on(pix + step, dv1);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 x0 = pair(xv0, j), x1 = pair(xv1, j);
        const f32x2 g0 = pair(dv0, j) * act_grad2<ACT>(__builtin_elementwise_fma(x0, sc[j], sh[j]));
        const f32x2 g1 = pair(dv1, j) * act_grad2<ACT>(__builtin_elementwise_fma(x1, sc[j], sh[j]));
        sg[j] += g0 + g1;
        sgx[j] += __builtin_elementwise_fma(g0, x0, g1 * x1);
      }
    }
    if (pix < a.npix) {
      const half8 dv = *reinterpret_cast<const half8*>(dyb + at(pix) * ldd);
      const half8 xv = *reinterpret_cast<const half8*>(a.x + at(pix) * a.ldx + c0);
      if (RES) pass_on(pix, dv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x2 x = pair(xv, j);
        const f32x2 g = pair(dv, j) * act_grad2<ACT>(__builtin_elementwise_fma(x, sc[j], sh[j]));
        sg[j] += g;
        sgx[j] += g * x;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
        sgx[j][k] = (sgx[j][k] - a.coef[2 * a.C + c0 + 2 * j + k] * sg[j][k]) * a.coef[3 * a.C + c0 + 2 * j + k];
  }
  __shared__ float red[2][256][8 + 1];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][tid][j] = sg[j >> 1][j & 1];
    red[1][tid][j] = sgx[j >> 1][j & 1];
  }
  __syncthreads();
  for (int i = tid; i < 2 * a.C; i += 256) {
    const int which = i / a.C, c = i - which * a.C, pp = c >> 3, j = c & 7;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += red[which][r * cpp + pp][j];
    if (a.acc) unsafeAtomicAdd(&a.acc[((size_t)(bx % DY_BN_COPIES) * 2 + which) * a.C + c], (double)s);
    else a.partials[((size_t)bx * 2 + which) * a.C + c] = s;
  }
}

template <int ACT, bool RES = false>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(BwdRedArgs a) {
  bn_act_bwd_reduce_body<ACT, RES>(a, (int)blockIdx.x, (int)gridDim.x);
}
struct BwdRedGroup {  // several tensors in one launch (blockIdx.y = entry), as ApplyGroup
  int n;
  int nblk[DY_BN_GROUP_MAX];
  BwdRedArgs a[DY_BN_GROUP_MAX];
};
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_group_kernel(BwdRedGroup g) {
  const int e = blockIdx.y;
  if ((int)blockIdx.x >= g.nblk[e]) return;
  bn_act_bwd_reduce_body<DY_ACT_SILU, false>(g.a[e], (int)blockIdx.x, g.nblk[e]);
}

extern "C" int dy_bn_act_bwd_reduce(const void* dy, int lddy, const void* x, int ldx, const float* coef,
                                    float* partials, int max_partials, long npix, int C, int act, int* nparts,
                                    hipStream_t stream) {
  if ((C & 7) || C > 2048 || (ldx & 7) || (lddy & 7)) return DY_ERR_ALIGN;
  const int cpp = C >> 3;
  if (cpp > 256) return DY_ERR_ARG;
  const int rows = 256 / cpp;
  long blocks = (npix + (long)rows * 8 - 1) / ((long)rows * 8);
  if (blocks > max_partials) blocks = max_partials;
  // 1024 blocks (4 per CU): 14.20 / 14.23 ms per step against 14.34 / 14.35 at 2048 and 14.43 at 512 or 1536 (alternating runs on one
  // box, round 2) -- fewer partial rows for the finalize launch that follows, still enough loads in flight
  static const long bcap = getenv("DY_EW_BLOCKS_BRED") ? atol(getenv("DY_EW_BLOCKS_BRED")) : 1024;
  if (blocks > bcap) blocks = bcap;
  if (blocks < 1) blocks = 1;
  if (nparts) *nparts = (int)blocks;
  const int rev = ew_reverse();
  BwdRedArgs a{(const f16*)dy, (const f16*)x, coef, partials, lddy, ldx, C, act, npix, (rev >> 2) & 1, nullptr, nullptr, 0, 0};
  DY_ACT_DISPATCH(bn_act_bwd_reduce_kernel, dim3((int)blocks), stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_bn_act_bwd_reduce_acc(const void* dy, int lddy, const void* x, int ldx, const float* coef, double* acc,
                                        long npix, int C, int act, void* res_grad, int ldrg, int res_accumulate, hipStream_t stream) {
  if ((C & 7) || C > 2048 || (ldx & 7) || (lddy & 7) || (res_grad && ((ldrg & 7) || ((uintptr_t)res_grad & 15)))) return DY_ERR_ALIGN;
  const int cpp = C >> 3;
  if (cpp > 256 || !acc) return DY_ERR_ARG;
  const int rows = 256 / cpp;
  long blocks = (npix + (long)rows * 8 - 1) / ((long)rows * 8);
  static const long bcap = getenv("DY_EW_BLOCKS_BRED") ? atol(getenv("DY_EW_BLOCKS_BRED")) : 1024;
  if (blocks > bcap) blocks = bcap;
  if (blocks < 1) blocks = 1;
  const int rev = ew_reverse();
  BwdRedArgs a{(const f16*)dy, (const f16*)x, coef, nullptr, lddy, ldx, C, act, npix, (rev >> 2) & 1, acc, (f16*)res_grad, ldrg, res_accumulate};
  if (res_grad && act == DY_ACT_SILU) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<DY_ACT_SILU, true>), dim3((int)blocks), dim3(256), 0, stream, a);
  else if (res_grad && act == DY_ACT_LEAKY) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<DY_ACT_LEAKY, true>), dim3((int)blocks), dim3(256), 0, stream, a);
  else if (res_grad) hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<DY_ACT_NONE, true>), dim3((int)blocks), dim3(256), 0, stream, a);
  else DY_ACT_DISPATCH(bn_act_bwd_reduce_kernel, dim3((int)blocks), stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// dy_bn_act_bwd_reduce_acc (no shortcut gradient) for a gradient that lives in TWO planes: channels [0, csplit) in dy, the rest in dy2
extern "C" int dy_bn_act_bwd_reduce_acc_split(const void* dy, int lddy, const void* dy2, int lddy2, int csplit, const void* x, int ldx,
                                              const float* coef, double* acc, long npix, int C, int act, hipStream_t stream) {
  if ((C & 7) || C > 2048 || (ldx & 7) || (lddy & 7) || (lddy2 & 7) || (csplit & 7) || csplit <= 0 || csplit >= C || !dy2) return DY_ERR_ALIGN;
  const int cpp = C >> 3;
  if (cpp > 256 || !acc) return DY_ERR_ARG;
  const int rows = 256 / cpp;
  long blocks = (npix + (long)rows * 8 - 1) / ((long)rows * 8);
  static const long bcap = getenv("DY_EW_BLOCKS_BRED") ? atol(getenv("DY_EW_BLOCKS_BRED")) : 1024;
  if (blocks > bcap) blocks = bcap;
  if (blocks < 1) blocks = 1;
  const int rev = ew_reverse();
  BwdRedArgs a{(const f16*)dy, (const f16*)x, coef, nullptr, lddy, ldx, C, act, npix, (rev >> 2) & 1, acc, nullptr, 0, 0};
  a.dy2 = (const f16*)dy2; a.lddy2 = lddy2; a.csplit = csplit;
  DY_ACT_DISPATCH(bn_act_bwd_reduce_kernel, dim3((int)blocks), stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// dy_bn_act_bwd_reduce_acc (SiLU, no shortcut gradient) for n <= dy_bn_group_max() tensors in one launch
extern "C" int dy_bn_act_bwd_reduce_acc_group(int n, const void* const* dy, const int* lddy, const void* const* x, const int* ldx,
                                              const float* const* coef, double* const* acc, const long* npix, const int* C, hipStream_t stream) {
  if (n < 1 || n > DY_BN_GROUP_MAX) return DY_ERR_ARG;
  BwdRedGroup g{};
  g.n = n;
  int gx = 1;
  const int rev = ew_reverse();
  static const long bcap = getenv("DY_EW_BLOCKS_BRED") ? atol(getenv("DY_EW_BLOCKS_BRED")) : 1024;
  for (int e = 0; e < n; ++e) {
    if ((C[e] & 7) || C[e] > 2048 || (ldx[e] & 7) || (lddy[e] & 7)) return DY_ERR_ALIGN;
    const int cpp = C[e] >> 3;
    if (cpp > 256 || !acc[e] || !dy[e] || !x[e] || !coef[e]) return DY_ERR_ARG;
    const int rows = 256 / cpp;
    long blocks = (npix[e] + (long)rows * 8 - 1) / ((long)rows * 8);
    if (blocks > bcap) blocks = bcap;
    if (blocks < 1) blocks = 1;
    g.a[e] = BwdRedArgs{(const f16*)dy[e], (const f16*)x[e], coef[e], nullptr, lddy[e], ldx[e], C[e], DY_ACT_SILU, npix[e], (rev >> 2) & 1, acc[e],
                        nullptr, 0, 0};
    g.nblk[e] = (int)blocks;
    gx = g.nblk[e] > gx ? g.nblk[e] : gx;
  }
  hipLaunchKernelGGL(bn_act_bwd_reduce_group_kernel, dim3(gx, n), dim3(256), 0, stream, g);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// The same pass when dy has ROWS (the gradient Detect's final box conv passes down: non-zero at the loss's foreground anchors only,
// csrc/head_rows.hip): g = dy * act'(.) is zero wherever dy is, so only foreground pixels are visited -- the assignment (4 bytes per
// anchor) is read instead of dy and x (4 * C bytes per pixel).  A workgroup owns a contiguous pixel range and lists its foreground
// pixels in ascending order, so its sums have one order; workgroups meet in the fp64 accumulator as in the dense form.
struct BwdRedRowsArgs {
  const f16* dy;
  const f16* x;
  const float* coef;
  double* acc;
  const int* flag;  // (B, A) assigned-gt index, -1 = background; pixel (b, r) of this tensor is anchor a0 + r
  int lddy, ldx, C, A, a0, hw, B;
};
template <int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_rows_kernel(BwdRedRowsArgs a) {
  __shared__ int s_list[256];
  __shared__ int s_cnt[4];
  __shared__ float red[2][256][8 + 1];
  const int cpp = a.C >> 3, rows = 256 / cpp, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int part = tid % cpp, row = tid / cpp, c0 = part * 8;
  f32x2 sg[4], sgx[4], sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sg[j] = sgx[j] = (f32x2){0.f, 0.f};
    sc[j] = (f32x2){a.coef[c0 + 2 * j], a.coef[c0 + 2 * j + 1]};
    sh[j] = (f32x2){a.coef[a.C + c0 + 2 * j], a.coef[a.C + c0 + 2 * j + 1]};
  }
  const long npix = (long)a.B * a.hw;
  const long chunk = (npix + gridDim.x - 1) / gridDim.x;
  const long lo = (long)blockIdx.x * chunk, hi = lo + chunk < npix ? lo + chunk : npix;
  // the flags of eight trips are requested together: one trip at a time the loop was a chain of 25 dependent memory latencies at
  // 160x160
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
    if (fg) s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = tid;
    __syncthreads();
    if (row < rows) {
      for (int i = row; i < total; i += rows) {
        const long q = p0 + s_list[i];
        const half8 dv = *reinterpret_cast<const half8*>(a.dy + q * a.lddy + c0);
        const half8 xv = *reinterpret_cast<const half8*>(a.x + q * a.ldx + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x2 x = (f32x2){(float)xv[2 * j], (float)xv[2 * j + 1]};
          const f32x2 g = (f32x2){(float)dv[2 * j], (float)dv[2 * j + 1]} * act_grad2<ACT>(__builtin_elementwise_fma(x, sc[j], sh[j]));
          sg[j] += g;
          sgx[j] += g * x;
        }
      }
    }
    __syncthreads();  // the list is rewritten by the next trip
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k)
      sgx[j][k] = (sgx[j][k] - a.coef[2 * a.C + c0 + 2 * j + k] * sg[j][k]) * a.coef[3 * a.C + c0 + 2 * j + k];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[0][tid][j] = row < rows ? sg[j >> 1][j & 1] : 0.f;
    red[1][tid][j] = row < rows ? sgx[j >> 1][j & 1] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < 2 * a.C; i += 256) {
    const int which = i / a.C, c = i - which * a.C, pp = c >> 3, j = c & 7;
    float t = 0.f;
    for (int r = 0; r < rows; ++r) t += red[which][r * cpp + pp][j];
    if (t != 0.f) unsafeAtomicAdd(&a.acc[((size_t)(blockIdx.x % DY_BN_COPIES) * 2 + which) * a.C + c], (double)t);
  }
}
extern "C" int dy_bn_act_bwd_reduce_rows(const void* dy, int lddy, const void* x, int ldx, const float* coef, double* acc, int n,
                                         int hw, int C, int act, const int* assigned, int A, int a0, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (lddy & 7)) return DY_ERR_ALIGN;
  const int cpp = C >> 3;
  if (cpp > 256 || !acc || !assigned || n < 1 || hw < 1 || a0 < 0 || a0 + hw > A) return DY_ERR_ARG;
  BwdRedRowsArgs a{(const f16*)dy, (const f16*)x, coef, acc, assigned, lddy, ldx, C, A, a0, hw, n};
  DY_ACT_DISPATCH(bn_act_bwd_reduce_rows_kernel, dim3(1024), stream, a);  // four workgroups per CU: the walk is latency-bound
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// pass 1b: reduce partials -> dgamma, dbeta (fp32 grads, scaled by the loss scale like every gradient) and the two
// per-channel means pass 2 needs.  bwdcoef = [2][C]: mean_g, mean_gxhat.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partials, int nparts, float* dgamma,
                                                              float* dbeta, float* bwdcoef, int C, float count,
                                                              int accumulate) {
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int i = tid; i < nparts; i += 256) {
    s1 += partials[((size_t)i * 2 + 0) * C + c];
    s2 += partials[((size_t)i * 2 + 1) * C + c];
  }
  __shared__ double r1[256], r2[256];
  r1[tid] = s1;
  r2[tid] = s2;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      r1[tid] += r1[tid + o];
      r2[tid] += r2[tid + o];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)r1[0];
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)r2[0];
    bwdcoef[c] = (float)(r1[0] / count);
    bwdcoef[C + c] = (float)(r2[0] / count);
  }
}

extern "C" int dy_bn_bwd_finalize(const float* partials, int nparts, float* dgamma, float* dbeta, float* bwdcoef, int C,
                                  float count, int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, partials, nparts, dgamma, dbeta, bwdcoef, C,
                     count, accumulate);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

// pass 2: dx_raw = scale * (g - mean_g - xhat * mean_gxhat); optional second output: residual gradient pass-through
struct BwdApplyArgs {
  const f16* dy;
  const f16* x;
  f16* dx;
  const float* coef;
  const float* bwdcoef;
  int lddy, ldx, lddx, C, act, frozen_stats;
  long npix;
  int rev;  // DY_EW_REVERSE bit 1
  const double* acc;  // non-null: [DY_BN_COPIES][2][C] sums of g and g*xhat from the reduce pass (replaces bwdcoef)
  float* dgamma;      // with acc: written by block 0
  float* dbeta;
  float count;
};

template <int ACT, bool ACC>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(BwdApplyArgs a) {
  const int cpp = a.C >> 3, rows = 256 / cpp;
  const int part = threadIdx.x % cpp, row = threadIdx.x / cpp, c0 = part * 8;
  extern __shared__ float s_bw[];  // ACC: [2][C] = mean_g, mean_gxhat
  if (ACC) {
    for (int c = threadIdx.x; c < a.C; c += 256) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int k = 0; k < DY_BN_COPIES; ++k) {
        s1 += a.acc[(size_t)(k * 2 + 0) * a.C + c];
        s2 += a.acc[(size_t)(k * 2 + 1) * a.C + c];
      }
      s_bw[c] = (float)(s1 / a.count);
      s_bw[a.C + c] = (float)(s2 / a.count);
      if (blockIdx.x == 0) {
        if (a.dbeta) a.dbeta[c] = (float)s1;
        if (a.dgamma) a.dgamma[c] = (float)s2;
      }
    }
    __syncthreads();
  }
  if (row >= rows) return;
  // dx = sc*(g - mean_g - xhat*mean_gx) = sc*g - kb*x - kc  with  kb = sc*invstd*mean_gx,  kc = sc*mean_g - kb*mean
  float sc[8], sh[8], kb[8], kc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sc[j] = a.coef[c0 + j];
    sh[j] = a.coef[a.C + c0 + j];
    const float mean = a.coef[2 * a.C + c0 + j], inv = a.coef[3 * a.C + c0 + j];
    const float mg = ACC ? s_bw[c0 + j] : (a.frozen_stats ? 0.f : a.bwdcoef[c0 + j]);
    const float mgx = ACC ? s_bw[a.C + c0 + j] : (a.frozen_stats ? 0.f : a.bwdcoef[a.C + c0 + j]);
    kb[j] = sc[j] * inv * mgx;
    kc[j] = sc[j] * mg - kb[j] * mean;
  }
  auto at = [&](long p) { return a.rev ? a.npix - 1 - p : p; };
  auto one = [&](long pix, const half8& dv, const half8& xv) {
    pix = at(pix);
    *reinterpret_cast<half8*>(a.dx + pix * a.lddx + c0) = bn_bwd_apply8<ACT>(dv, xv, sc, sh, kb, kc);
  };
  const long step = (long)gridDim.x * rows;
  long pix = (long)blockIdx.x * rows + row;
  for (; pix + step < a.npix; pix += 2 * step) {
    const half8 d0 = *reinterpret_cast<const half8*>(a.dy + at(pix) * a.lddy + c0);
    const half8 x0 = *reinterpret_cast<const half8*>(a.x + at(pix) * a.ldx + c0);
    const half8 d1 = *reinterpret_cast<const half8*>(a.dy + at(pix + step) * a.lddy + c0);
    const half8 x1 = *reinterpret_cast<const half8*>(a.x + at(pix + step) * a.ldx + c0);
    one(pix, d0, x0);
    one(pix + step, d1, x1);
  }
  if (pix < a.npix)
    one(pix, *reinterpret_cast<const half8*>(a.dy + at(pix) * a.lddy + c0), *reinterpret_cast<const half8*>(a.x + at(pix) * a.ldx + c0));
}

extern "C" int dy_bn_act_bwd_apply(const void* dy, int lddy, const void* x, int ldx, void* dx, int lddx,
                                   const float* coef, const float* bwdcoef, long npix, int C, int act, int frozen_stats,
                                   hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7)) return DY_ERR_ALIGN;
  if ((C >> 3) > 256) return DY_ERR_ARG;
  const int rev = ew_reverse();
  BwdApplyArgs a{(const f16*)dy, (const f16*)x, (f16*)dx, coef, bwdcoef, lddy, ldx, lddx, C, act, frozen_stats, npix, (rev >> 1) & 1,
                 nullptr, nullptr, nullptr, 1.f};
  const dim3 grid(ew_blocks(npix, C, "DY_EW_BLOCKS_BAPPLY", 512));
  if (act == DY_ACT_SILU) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DY_ACT_SILU, false>), grid, dim3(256), 0, stream, a);
  else if (act == DY_ACT_LEAKY) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DY_ACT_LEAKY, false>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DY_ACT_NONE, false>), grid, dim3(256), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_bn_act_bwd_apply_acc(const void* dy, int lddy, const void* x, int ldx, void* dx, int lddx,
                                       const float* coef, const double* acc, float* dgamma, float* dbeta, long npix, int C,
                                       int act, float count, hipStream_t stream) {
  if ((C & 7) || (ldx & 7) || (lddy & 7) || (lddx & 7)) return DY_ERR_ALIGN;
  if ((C >> 3) > 256 || !acc) return DY_ERR_ARG;
  const int rev = ew_reverse();
  BwdApplyArgs a{(const f16*)dy, (const f16*)x, (f16*)dx, coef, nullptr, lddy, ldx, lddx, C, act, 0, npix, (rev >> 1) & 1,
                 acc, dgamma, dbeta, count};
  const dim3 grid(ew_blocks(npix, C, "DY_EW_BLOCKS_BAPPLY", 512));
  const size_t lds = 2 * (size_t)C * sizeof(float);
  if (act == DY_ACT_SILU) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DY_ACT_SILU, true>), grid, dim3(256), lds, stream, a);
  else if (act == DY_ACT_LEAKY) hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DY_ACT_LEAKY, true>), grid, dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((bn_act_bwd_apply_kernel<DY_ACT_NONE, true>), grid, dim3(256), lds, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
