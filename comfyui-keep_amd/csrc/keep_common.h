// Shared helpers for libkeep_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/keep_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void keep_set_error(const char* fmt, ...);

#define KEEP_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      keep_set_error(__VA_ARGS__);         \
      return KEEP_EINVAL;                  \
    }                                      \
  } while (0)

#define KEEP_LAUNCH_CHECK(name)                                            \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      keep_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
      return KEEP_EHIP;                                                    \
    }                                                                      \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case KEEP_ACT_RELU: return v > 0.f ? v : 0.f;
    case KEEP_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
    case KEEP_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case KEEP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

__device__ __forceinline__ float pro_apply(float v, int act) {
  if (act == KEEP_PRO_SWISH) return v * (1.0f / (1.0f + expf(-v)));
  if (act == KEEP_PRO_RELU) return v > 0.f ? v : 0.f;
  return v;
}
