// non_max_suppression's candidate cap (reference utils/ops.py:395-396):
//     if n > max_nms:  x = x[x[:, 4].argsort(descending=True)[:max_nms]]
// per image: the max_nms most confident candidates, in descending confidence.  The reference's argsort is unstable (the order of
// EQUAL confidences is unspecified there); here ties keep candidate order, which makes the result reproducible.
//
// One 1024-thread workgroup per image (validation path: tens of images, not a throughput kernel):
//   1. radix select (4 x 8 bits, LDS histograms) of the max_nms-th smallest key, key = ~bits(confidence): confidences are positive
//      floats, so their bit patterns order like the values and the complement turns "descending" into "ascending";
//   2. ordered compaction (ballot + popcount scans, chunk by chunk) of the selected candidates -- every key below the threshold
//      and the first few equal to it -- into 64-bit sort keys (key << 32 | candidate index): unique, so any sort is stable;
//   3. bitonic sort of the padded power-of-two array in global memory (L2-resident: 256 KB for max_nms = 30000);
//   4. gather of boxes / confidences / classes in sorted order into the output buffers.
// Images with n <= max_nms are copied through unchanged.
#include "common.h"
#include "dealyolo_hip.h"

struct PresortArgs {
  const float* cbox;  // (B, cap, 4)
  const float* csc;   // (B, cap)
  const float* ccl;   // (B, cap)
  int* count;         // (B) in: candidates per image, out: min(count, max_nms)
  float* obox;        // (B, max_nms, 4)
  float* osc;         // (B, max_nms)
  float* ocl;         // (B, max_nms)
  unsigned long long* keys;  // (B, P) workspace
  int B, cap, max_nms, P;
};

static __device__ __forceinline__ unsigned conf_key(float s) { return ~__float_as_uint(s); }

// exclusive prefix of a predicate over the 1024 threads of the block + the block total (wave ballots, 16 wave totals in LDS)
static __device__ __forceinline__ int block_excl_scan(bool pred, int* wave_tot, int& total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(pred);
  const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();  // previous use of wave_tot is over
  if (lane == 0) wave_tot[wave] = __popcll(m);
  __syncthreads();
  int before = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int t = wave_tot[w];
    before += w < wave ? t : 0;
    tot += t;
  }
  total = tot;
  return before + in_wave;
}

__global__ __launch_bounds__(1024) void nms_presort_kernel(PresortArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = a.count[b] < a.cap ? a.count[b] : a.cap;
  const float* sc = a.csc + (size_t)b * a.cap;
  const float* cl = a.ccl + (size_t)b * a.cap;
  const float* bx = a.cbox + (size_t)b * a.cap * 4;
  float* osc = a.osc + (size_t)b * a.max_nms;
  float* ocl = a.ocl + (size_t)b * a.max_nms;
  float* obx = a.obox + (size_t)b * a.max_nms * 4;
  if (n <= a.max_nms) {  // nothing to cut: straight copy
    for (int i = tid; i < n; i += 1024) {
      osc[i] = sc[i];
      ocl[i] = cl[i];
      *reinterpret_cast<float4*>(obx + (size_t)i * 4) = *reinterpret_cast<const float4*>(bx + (size_t)i * 4);
    }
    return;
  }
  const int K = a.max_nms;
  __shared__ int hist[256];
  __shared__ int wave_tot[16];
  __shared__ unsigned s_prefix;
  __shared__ int s_rank;
  // ---- 1. the K-th smallest key, most significant digit first
  if (tid == 0) { s_prefix = 0u; s_rank = K; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int i = tid; i < n; i += 1024) {
      const unsigned k = conf_key(sc[i]);
      if (pass == 0 || (k >> (shift + 8)) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int rank = s_rank, cum = 0, d = 0;
      for (; d < 256; ++d) {
        if (cum + hist[d] >= rank) break;
        cum += hist[d];
      }
      s_prefix = (prefix << 8) | (unsigned)d;
      s_rank = rank - cum;
    }
    __syncthreads();
  }
  const unsigned Tk = s_prefix;   // threshold key
  const int need_ties = s_rank;   // how many candidates with key == Tk belong to the selection (the first ones in candidate order)
  // ---- 2. ordered compaction into 64-bit sort keys
  unsigned long long* keys = a.keys + (size_t)b * a.P;
  int sel_base = 0, tie_base = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const unsigned k = i < n ? conf_key(sc[i]) : 0xffffffffu;
    const bool tie = i < n && k == Tk;
    int tie_tot, sel_tot;
    const int tie_rank = tie_base + block_excl_scan(tie, wave_tot, tie_tot);
    const bool sel = i < n && (k < Tk || (tie && tie_rank < need_ties));
    const int pos = sel_base + block_excl_scan(sel, wave_tot, sel_tot);
    if (sel) keys[pos] = ((unsigned long long)k << 32) | (unsigned)i;
    sel_base += sel_tot;
    tie_base += tie_tot;
  }
  for (int i = K + tid; i < a.P; i += 1024) keys[i] = ~0ull;
  __syncthreads();
  // ---- 3. bitonic sort (ascending) of P = 2^m keys; one workgroup: barriers order the global-memory passes
  for (int k = 2; k <= a.P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (a.P >> 1); t += 1024) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const unsigned long long x = keys[lo], y = keys[hi];
        const bool up = (lo & k) == 0;
        if ((x > y) == up) {
          keys[lo] = y;
          keys[hi] = x;
        }
      }
      __syncthreads();
    }
  // ---- 4. gather in sorted order
  for (int i = tid; i < K; i += 1024) {
    const int src = (int)(keys[i] & 0xffffffffull);
    osc[i] = sc[src];
    ocl[i] = cl[src];
    *reinterpret_cast<float4*>(obx + (size_t)i * 4) = *reinterpret_cast<const float4*>(bx + (size_t)src * 4);
  }
  __syncthreads();
  if (tid == 0) a.count[b] = K;
}

static int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
extern "C" size_t dy_nms_presort_workspace(int B, int max_nms) { return (size_t)B * pow2_at_least(max_nms) * 8; }
extern "C" int dy_nms_presort(const float* cbox, const float* cscore, const float* ccls, int* count, int B, int cap, int max_nms,
                              float* obox, float* oscore, float* ocls, void* workspace, hipStream_t stream) {
  if (B < 1 || cap < 1 || max_nms < 1 || max_nms > (1 << 20) || !workspace || ((uintptr_t)cbox & 15) || ((uintptr_t)obox & 15))
    return DY_ERR_ARG;
  PresortArgs a{cbox, cscore, ccls, count, obox, oscore, ocls, (unsigned long long*)workspace, B, cap, max_nms, pow2_at_least(max_nms)};
  hipLaunchKernelGGL(nms_presort_kernel, dim3(B), dim3(1024), 0, stream, a);
  DY_CHECK_LAUNCH();
  return DY_OK;
}
