// Shared device helpers for the DEAL-YOLO gfx950 kernels (wave64, MFMA 16x16x32 f16).
#pragma once
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
static __device__ __forceinline__ float quad16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
static __device__ __forceinline__ float silu_f(float z) { return z / (1.f + __expf(-z)); }
static __device__ __forceinline__ float sigmoid_f(float z) { return 1.f / (1.f + __expf(-z)); }

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#define DY_CHECK_LAUNCH()                                  \
  do {                                                     \
    hipError_t e__ = hipGetLastError();                    \
    if (e__ != hipSuccess) return DY_ERR_LAUNCH;           \
  } while (0)
