// Shared device helpers for the DEAL-YOLO gfx950 kernels (wave64, MFMA 16x16x32 f16).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define DY_OK 0
#define DY_ERR_ARG (-1)      // unsupported shape / bad argument
#define DY_ERR_LAUNCH (-2)   // hipGetLastError() != success after launch
#define DY_ERR_ALIGN (-3)    // pointer / stride alignment the kernel relies on is violated

#define LDS_AS __attribute__((address_space(3)))

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 16 lanes that share (lane >> 4)
// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in every lane.  Four v_add_f32 with a DPP operand -- quad
// swaps, then row_half_mirror (lane i <-> 7 - i of its half row: the other quad, whose lanes all hold their quad's sum) and row_mirror
// (i <-> 15 - i: the other half row) -- instead of four __shfl_xor, which the compiler lowers to ds_bpermute / ds_swizzle: each of those
// is an LDS-pipe round trip, and the statistics tail of the conv kernel does 128 of them per lane on 8 waves at once.
template <int CTRL>
static __device__ __forceinline__ float dpp_read(float x) {  // x of the lane the DPP control names (0 where there is none)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
#define DY_DPP_QUAD_1032 0xB1   // quad_perm [1, 0, 3, 2]: lane ^ 1
#define DY_DPP_QUAD_2301 0x4E   // quad_perm [2, 3, 0, 1]: lane ^ 2
#define DY_DPP_HALF_MIRROR 0x141
#define DY_DPP_ROW_MIRROR 0x140
#define DY_DPP_ROW_SHL(n) (0x100 + (n))  // lane + n inside its row of 16
static __device__ __forceinline__ float quad_sum(float v) {  // over the 4 lanes of a quad
  v += dpp_read<DY_DPP_QUAD_1032>(v);
  v += dpp_read<DY_DPP_QUAD_2301>(v);
  return v;
}
static __device__ __forceinline__ float half_row_sum(float v) {  // over the 8 lanes of a half row
  v = quad_sum(v);
  v += dpp_read<DY_DPP_HALF_MIRROR>(v);
  return v;
}
static __device__ __forceinline__ float quad16_sum(float v) {
  v = half_row_sum(v);
  v += dpp_read<DY_DPP_ROW_MIRROR>(v);
  return v;
}
static __device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
static __device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + __expf(-z)); }

// ---- BatchNorm + activation element math shared by the element-wise kernels (bn_act.hip) and the kernels that apply it while
// staging (conv_wgrad.hip): one definition, so a fused pass produces the bits of the stand-alone pass.
#ifndef DY_ACT_SILU
#define DY_ACT_NONE 0
#define DY_ACT_SILU 1
#define DY_ACT_LEAKY 2
#endif
template <int ACT>
static __device__ __forceinline__ float act_fwd_t(float z) {
  // forward keeps the correctly rounded division: the 1-ulp v_rcp_f32 variant is as accurate for any single value, but the
  // perturbation it puts on every activation was enough to flip a task-aligned top-10 choice in the 64x64 golden case
  // (first-layer gradient 3.7e-2 -> 1.3e-1 off the reference); the backward factor below may use v_rcp_f32
  if (ACT == DY_ACT_SILU) return z / (1.f + __expf(-z));
  if (ACT == DY_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
  return z;
}
// forward activation of a PAIR of values with the division written out: q = z / den through v_rcp + one Newton step on the
// reciprocal + one residual correction of the quotient (Markstein's sequence: the correctly rounded quotient whenever nothing
// over- or underflows on the way -- den is in [1, 2^126], so nothing does), all in packed fp32; the compiler's own division adds
// v_div_scale / v_div_fmas / v_div_fixup for ranges that cannot occur here.  Same bits as act_fwd_t for the values that matter,
// about half the instructions.  The forward apply pass of the accumulator path uses it (DY_SILU_FAST=0 switches back).
template <int ACT>
static __device__ __forceinline__ f32x2 act_fwd2_fast(f32x2 z) {
  if (ACT == DY_ACT_SILU) {
    f32x2 t = z * (f32x2){-1.4426950408889634f, -1.4426950408889634f};
    t = (f32x2){fminf(t[0], 126.f), fminf(t[1], 126.f)};  // exp2 stays finite: den = inf would turn the Newton step into 0 * inf
    const f32x2 one = {1.f, 1.f};
    const f32x2 den = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + one;
    f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    r = __builtin_elementwise_fma(__builtin_elementwise_fma(-den, r, one), r, r);
    const f32x2 q = z * r;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-den, q, z), r, q);
  }
  if (ACT == DY_ACT_LEAKY) return (f32x2){z[0] > 0.f ? z[0] : 0.1f * z[0], z[1] > 0.f ? z[1] : 0.1f * z[1]};
  return z;
}
// y = fp16(SiLU(raw * scale + shift)) for one 8-channel granule, bit for bit what bn_act_apply_kernel<DY_ACT_SILU, ., true> stores:
// consumers that read a Conv's RAW output and apply BatchNorm + SiLU themselves (csrc/head_rows.hip, the loss's logit recompute)
// see the tensor the apply launch would have written.
static __device__ __forceinline__ half8 bn_silu_apply8(const half8& xv, const float* sc, const float* sh) {
  half8 out;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 z = act_fwd2_fast<DY_ACT_SILU>(__builtin_elementwise_fma((f32x2){(float)xv[j], (float)xv[j + 1]}, (f32x2){sc[j], sc[j + 1]},
                                                                         (f32x2){sh[j], sh[j + 1]}));
    const half2_ oh = __builtin_convertvector(z, half2_);
    out[j] = oh[0];
    out[j + 1] = oh[1];
  }
  return out;
}
// d act / dz for a PAIR of values: everything except v_exp / v_rcp is a 2-wide packed fp32 instruction (v_pk_fma_f32 ...), which
// is what makes the BatchNorm backward cheap enough to run inside a staging loop.  sigmoid' form: s + z*s*(1-s), s*(1-s) = s - s^2.
template <int ACT>
static __device__ __forceinline__ f32x2 act_grad2(f32x2 z) {
  if (ACT == DY_ACT_SILU) {
    const f32x2 t = z * (f32x2){-1.4426950408889634f, -1.4426950408889634f};
    const f32x2 den = (f32x2){__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + (f32x2){1.f, 1.f};
    const f32x2 s = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    const f32x2 q = __builtin_elementwise_fma(-s, s, s);
    return __builtin_elementwise_fma(z, q, s);
  }
  if (ACT == DY_ACT_LEAKY) return (f32x2){z[0] > 0.f ? 1.f : 0.1f, z[1] > 0.f ? 1.f : 0.1f};
  return (f32x2){1.f, 1.f};
}
template <int ACT>
static __device__ __forceinline__ float act_grad_t(float z) {
  return act_grad2<ACT>((f32x2){z, z})[0];
}
// d(raw conv output) of one 8-channel granule: dx = sc*g - (kb*x + kc), g = dy * act'(x*sc + sh)   (bn_act_bwd_apply_kernel)
template <int ACT>
static __device__ __forceinline__ half8 bn_bwd_apply8(const half8& dv, const half8& xv, const float* sc, const float* sh,
                                                      const float* kb, const float* kc) {
  half8 out;
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 x = {(float)xv[j], (float)xv[j + 1]}, d = {(float)dv[j], (float)dv[j + 1]};
    const f32x2 s2 = {sc[j], sc[j + 1]}, h2 = {sh[j], sh[j + 1]}, b2 = {kb[j], kb[j + 1]}, c2 = {kc[j], kc[j + 1]};
    const f32x2 ad = act_grad2<ACT>(__builtin_elementwise_fma(x, s2, h2));
    const f32x2 o = __builtin_elementwise_fma(s2 * ad, d, -__builtin_elementwise_fma(b2, x, c2));
    const half2_ oh = __builtin_convertvector(o, half2_);
    out[j] = oh[0];
    out[j + 1] = oh[1];
  }
  return out;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#define DY_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return DY_ERR_LAUNCH;           \
  } while (0)
