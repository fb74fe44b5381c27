// Detect's inference tail in one launch (reference nn/modules/head.py:50-74 `_inference`): per level the two final convolutions
// (cv2[l][2]: Conv2d(64, 4 * reg_max, 1) and cv3[l][2]: Conv2d(c3, nc, 1), head.py:38-42), the DFL expectation
// (nn/modules/block.py:52-55), dist2bbox(xywh=True) * stride (utils/tal.py:310-318) and sigmoid(cls), written straight as
// y (B, 4 + nc, A) fp32.  The eager path did this as two generic conv launches per level that wrote fp32 logits (64 + nc values per
// anchor: 2.5 GB at 1280x1280, batch 32, nc = 80) and dy_decode_predictions that read them back (1.0 ms of a 10.6 ms forward there).
//
// One wave owns 64 consecutive pixels of a level: the B operands (the activated outputs of cv2[l][1] / cv3[l][1], fp16 NHWC) come
// straight from global memory as 16-byte granules, the A operands (fp16 copies of the fp32 master weights, as dy_pack_weights makes
// them) from LDS.  BIT-COMPATIBLE with the eager path by construction: the k-steps follow the chunk order dy_conv_forward takes
// for that channel count (32 channels per MFMA; 16 real + 16 zero for channel counts that are no multiple of 32), logits = fp32
// accumulator + bias, softmax / expectation / box arithmetic in decode_pred_kernel's association (csrc/nms.hip), libm expf.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include "common.h"
#include "dealyolo_hip.h"

#pragma clang fp contract(off)  // as csrc/nms.hip: the decode arithmetic must round like decode_pred_kernel's (no fused multiply-adds)

struct HeadInferLevel {
  const f16* xb;      // [npix][ldb] activated input of the final box conv (64 channels)
  const f16* xc;      // [npix][ldc] activated input of the final class conv (cin_c channels)
  const float* wb;    // fp32 [64][64]
  const float* bb;    // [64]
  const float* wc;    // fp32 [nc][cin_c]
  const float* bc;    // [nc]
  int ldb, ldc, H, W, a0, nblk;
  float stride;
};
struct HeadInferArgs {
  HeadInferLevel lv[4];
  float* y;           // (B, 4 + nc, A)
  int nl, B, A, nc, cin_c;
};

// 4 x 4 transpose across the four 16-lane rows of a wave and four registers: (register t, row q) -> (register q, row t), two
// v_permlane32_swap + two v_permlane16_swap (gfx950).  The MFMA result layout gives a lane the channels of ITS row for one pixel of
// each N-tile; after the transpose register q holds ONE channel for 64 consecutive pixels (lane = pixel), i.e. one 256-byte run of a
// (B, 4+nc, A) output row per store instead of four 64-byte pieces.
// (inline asm: chained through __builtin_amdgcn_permlane{16,32}_swap, hipcc 7.2 dropped one swap of four and stored ONE register to
// all four outputs -- seen in the ISA, caught by the bit-equality test; the s_nop cover the VALU -> permlane-swap -> VALU wait states the
// hazard recogniser would have inserted around the builtin form)
static __device__ __forceinline__ void swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
static __device__ __forceinline__ void swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
static __device__ __forceinline__ void xpose4(float& r0, float& r1, float& r2, float& r3) {
  swap32(r0, r2);   // r0 = [r0.0 r0.1 r2.0 r2.1]  r2 = [r0.2 r0.3 r2.2 r2.3]
  swap32(r1, r3);
  swap16(r0, r1);   // r0 = [r0.0 r1.0 r2.0 r3.0]  r1 = [r0.1 r1.1 r2.1 r3.1]
  swap16(r2, r3);
}

// CCK: channels per k-step of the class conv (32, or 16 = half of every MFMA's K is zero padding -- what dy_conv_forward does for
// channel counts like 48 / 80); KS: k-steps; MT: 16-row tiles of class outputs.
template <int CCK, int KS, int MT>
__global__ __launch_bounds__(256, 2) void head_infer_kernel(HeadInferArgs g) {
  const HeadInferLevel& a = g.lv[blockIdx.y];
  if ((int)blockIdx.x >= a.nblk) return;
  constexpr int PB = 160;                 // box weight row pitch (bytes): 10 slots, == 2 mod 4 (conflict-free ds_read_b128 groups)
  constexpr int KC = KS * 32;             // padded K of the class conv as the MFMAs see it
  constexpr int PC = KC * 2 + 32;         // class weight row pitch: KC/8 + 2 slots, == 2 mod 4 for KC = 32, 64, 96, 128, 160
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_wb = smem;
  char* const s_wc = smem + 64 * PB;
  float* const s_bb = reinterpret_cast<float*>(smem + 64 * PB + 16 * MT * PC);
  float* const s_bc = s_bb + 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, k = i & 63;     // channel c = 16 q' + 4 m' + j'  ->  row 16 m' + 4 q' + j' : a lane ends with the 16 bins of one side
    const int row = ((c >> 2) & 3) * 16 + (c >> 4) * 4 + (c & 3);
    *reinterpret_cast<f16*>(s_wb + row * PB + k * 2) = (f16)a.wb[i];
  }
  for (int i = tid; i < 16 * MT * KC; i += 256) {
    const int c = i / KC, kk = i - c * KC;                 // natural row order: MFMA row 4 q + j of tile m = channel 16 m + 4 q + j
    const int ks = kk >> 5, r = kk & 31;
    const int ch = ks * CCK + r;                           // real input channel of this K slot (r >= CCK: padding)
    const float v = (c < g.nc && r < CCK && ch < g.cin_c) ? a.wc[(size_t)c * g.cin_c + ch] : 0.f;
    *reinterpret_cast<f16*>(s_wc + c * PC + kk * 2) = (f16)v;
  }
  if (tid < 64) s_bb[tid] = a.bb[tid];
  if (tid < 16 * MT) s_bc[tid] = tid < g.nc ? a.bc[tid] : 0.f;
  __syncthreads();
  const int hw = a.H * a.W;
  const long npix = (long)g.B * hw;
  const float invw = 1.0f / (float)a.W;
  const int no = 4 + g.nc;
  const bool cvalid = CCK == 32 || q < 2;                  // lane groups 2, 3 carry the zero half of a 16-channel k-step
  for (long base = ((long)blockIdx.x * 4 + wave) * 64; base < npix; base += (long)a.nblk * 256) {
    // where this lane's OUTPUT pixel (base + lane, after the transposes) goes
    const long opix = base + lane;
    const bool ovalid = opix < npix;
    const int ob = (int)((ovalid ? opix : 0) / hw), orr = (int)((ovalid ? opix : 0) - (long)ob * hw);
    float* const yout = g.y + ((size_t)ob * no) * g.A + a.a0 + orr;
    // ---------------------------------------------------------------- box: 64 -> 64, DFL, xywh * stride
    {
      half8 bf[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long pix = base + t * 16 + p;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          if (pix < npix) bf[t][ks] = *reinterpret_cast<const half8*>(a.xb + pix * a.ldb + ks * 32 + q * 8);
          else bf[t][ks] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      f32x4 acc[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const half8 a0f = *reinterpret_cast<const half8*>(s_wb + (m * 16 + p) * PB + q * 16);
        const half8 a1f = *reinterpret_cast<const half8*>(s_wb + (m * 16 + p) * PB + 64 + q * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0f, bf[t][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1f, bf[t][1], acc[m][t], 0, 0, 0);
        }
      }
      float z[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long pix = base + t * 16 + p;
        float v[16];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[m * 4 + j] = acc[m][t][j] + s_bb[q * 16 + m * 4 + j];
        float mx = v[0];
#pragma unroll
        for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
        float den4[4], num4[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {  // decode_pred_kernel's quarters: four consecutive bins, then (q0 + q1) + (q2 + q3)
          const float e0 = expf(v[gq * 4] - mx), e1 = expf(v[gq * 4 + 1] - mx), e2 = expf(v[gq * 4 + 2] - mx), e3 = expf(v[gq * 4 + 3] - mx);
          const float k0 = (float)(gq * 4);
          den4[gq] = (e0 + e1) + (e2 + e3);
          num4[gq] = e0 * k0 + e1 * (k0 + 1.f) + e2 * (k0 + 2.f) + e3 * (k0 + 3.f);
        }
        const float den = (den4[0] + den4[1]) + (den4[2] + den4[3]), num = (num4[0] + num4[1]) + (num4[2] + num4[3]);
        const float e = num / den;
        const long pc = pix < npix ? pix : npix - 1;
        const int b = (int)(pc / hw), r = (int)(pc - (long)b * hw);
        const int iy = (int)(((float)r + 0.5f) * invw), ix = r - iy * a.W;
        const float anc = (q & 1) ? (iy + 0.5f) : (ix + 0.5f);
        const float mine = q < 2 ? anc - e : anc + e;                 // side q: left, top, right, bottom
        const float other = __shfl_xor(mine, 32, 64);                  // the opposite side (q ^ 2) of the same pixel
        // rows 0, 1: centre x / y = (lo + hi) / 2; rows 2, 3: width / height = hi - lo  (decode_pred_kernel's expressions)
        z[t] = q < 2 ? (mine + other) * 0.5f * a.stride : (mine - other) * a.stride;
      }
      xpose4(z[0], z[1], z[2], z[3]);
      if (ovalid) {
#pragma unroll
        for (int c = 0; c < 4; ++c) yout[(size_t)c * g.A] = z[c];
      }
    }
    // ---------------------------------------------------------------- classes: cin_c -> nc, sigmoid
    {
      half8 bf[4][KS];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const long pix = base + t * 16 + p;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int ch = ks * CCK + q * 8;
          if (pix < npix && cvalid && ch < g.cin_c) bf[t][ks] = *reinterpret_cast<const half8*>(a.xc + pix * a.ldc + ch);
          else bf[t][ks] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
      // one 16-channel tile at a time: its four accumulators, the sigmoid, the transpose and the stores before the next tile's MFMAs
      // (all MT tiles live at once cost 80 accumulator registers for nc = 80: one wave per SIMD, nothing to hide the loads behind)
#pragma unroll 1
      for (int m = 0; m < MT; ++m) {
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const half8 af = *reinterpret_cast<const half8*>(s_wc + (m * 16 + p) * PC + ks * 64 + q * 16);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[t][ks], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float bj = s_bc[m * 16 + q * 4 + j];
          float sgm[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) sgm[t] = 1.f / (1.f + expf(-(acc[t][j] + bj)));
          xpose4(sgm[0], sgm[1], sgm[2], sgm[3]);  // sgm[q'] = channel 16 m + 4 q' + j of pixel base + lane
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int c = m * 16 + qq * 4 + j;
            if (c < g.nc && ovalid) yout[(size_t)(4 + c) * g.A] = sgm[qq];
          }
        }
      }
    }
  }
}

// channels per k-step and k-steps dy_conv_forward's geometry gives a 1x1 conv over cin channels (csrc/conv.hip pick_cc): whole
// 32-channel steps where cin is a multiple of 32, otherwise 16-channel chunks (one half-empty MFMA each)
static bool head_infer_shape(int cin, int* cck, int* ks) {
  if (cin >= 32 && cin % 32 == 0 && cin <= 128) { *cck = 32; *ks = cin / 32; return true; }
  if (cin % 16 == 0 && cin / 16 <= 5) { *cck = 16; *ks = cin / 16; return true; }
  return false;
}
extern "C" int dy_head_infer_supported(int cin_box, int cout_box, int cin_cls, int nc) {
  int cck, ks;
  return (cin_box == 64 && cout_box == 64 && nc >= 1 && nc <= 80 && head_infer_shape(cin_cls, &cck, &ks)) ? 1 : 0;
}

template <int CCK, int KS>
static int head_infer_mt(int mt, dim3 grid, size_t lds, hipStream_t s, const HeadInferArgs& g) {
  switch (mt) {
#define DY_MT(M) case M: { static bool set = false; auto k = head_infer_kernel<CCK, KS, M>; \
    if (!set) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) != hipSuccess) return DY_ERR_LAUNCH; set = true; } \
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, g); break; }
    DY_MT(1) DY_MT(2) DY_MT(3) DY_MT(4) DY_MT(5)
#undef DY_MT
    default: return DY_ERR_ARG;
  }
  DY_CHECK_LAUNCH();
  return DY_OK;
}

extern "C" int dy_head_infer_levels(int nl, const void* const* x_box, const int* ld_box, const float* const* w_box, const float* const* b_box,
                                    const void* const* x_cls, const int* ld_cls, const float* const* w_cls, const float* const* b_cls,
                                    const int* h, const int* w, const float* stride, int n, int cin_cls, int nc, float* y,
                                    hipStream_t stream) {
  int cck, ks;
  if (nl < 1 || nl > 4 || n < 1 || !y || !head_infer_shape(cin_cls, &cck, &ks) || nc < 1 || nc > 80) return DY_ERR_ARG;
  HeadInferArgs g{};
  int a0 = 0, gx = 1;
  for (int l = 0; l < nl; ++l) {
    if (!x_box[l] || !x_cls[l] || !w_box[l] || !b_box[l] || !w_cls[l] || !b_cls[l] || h[l] < 1 || w[l] < 1) return DY_ERR_ARG;
    if ((ld_box[l] & 7) || (ld_cls[l] & 7) || ((uintptr_t)x_box[l] & 15) || ((uintptr_t)x_cls[l] & 15)) return DY_ERR_ALIGN;
    // persistent-ish: every workgroup first stages both weight sets as fp16 (4,096 + nc x cin fp32 loads), so a few long-lived workgroups per
    // CU beat one per 256 pixels (DY_HEAD_INFER_WGS: the cap per level)
    static const int cap = getenv("DY_HEAD_INFER_WGS") ? atoi(getenv("DY_HEAD_INFER_WGS")) : 512;  // 4096 / 2048 / 1024 / 512 / 256: 768 / 708 / 673 / 656 / 840 us at 1280x1280, batch 32, nc 80
    long blocks = ((long)n * h[l] * w[l] + 255) / 256;
    if (blocks > cap) blocks = cap;
    g.lv[l] = HeadInferLevel{(const f16*)x_box[l], (const f16*)x_cls[l], w_box[l], b_box[l], w_cls[l], b_cls[l], ld_box[l], ld_cls[l],
                             h[l], w[l], a0, (int)blocks, stride[l]};
    a0 += h[l] * w[l];
    gx = (int)blocks > gx ? (int)blocks : gx;
  }
  g.y = y; g.nl = nl; g.B = n; g.A = a0; g.nc = nc; g.cin_c = cin_cls;
  const int mt = (nc + 15) / 16;
  const size_t lds = 64 * 160 + (size_t)16 * mt * (ks * 64 + 32) + (64 + 16 * mt) * 4;
  const dim3 grid(gx, nl);
  if (cck == 32) {
    if (ks == 1) return head_infer_mt<32, 1>(mt, grid, lds, stream, g);
    if (ks == 2) return head_infer_mt<32, 2>(mt, grid, lds, stream, g);
    if (ks == 3) return head_infer_mt<32, 3>(mt, grid, lds, stream, g);
    if (ks == 4) return head_infer_mt<32, 4>(mt, grid, lds, stream, g);
  } else {
    if (ks == 1) return head_infer_mt<16, 1>(mt, grid, lds, stream, g);
    if (ks == 3) return head_infer_mt<16, 3>(mt, grid, lds, stream, g);
    if (ks == 5) return head_infer_mt<16, 5>(mt, grid, lds, stream, g);
  }
  return DY_ERR_ARG;
}
