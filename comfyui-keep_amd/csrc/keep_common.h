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

// The product library reads NO environment variable: kernel-selection overrides arrive through `flags` of the argument structs
// (KEEP_CONV_* / KEEP_ATTN_*, include/keep_hip.h), deployment settings through their fields (plan_ref_images).  Developer A/B switches
// (tools/dev/README.md) exist only in builds with -DKEEP_DEV_KNOBS (tools/dev/build_ab.sh; implied by -DKEEP_X3_ABLATE).
#if defined(KEEP_X3_ABLATE) && !defined(KEEP_DEV_KNOBS)
#define KEEP_DEV_KNOBS 1
#endif
#ifdef KEEP_DEV_KNOBS
#include <stdlib.h>
#define KEEP_DEV_ENV(name) getenv(name)
#else
#define KEEP_DEV_ENV(name) ((const char*)nullptr)
#endif

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

__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case KEEP_ACT_RELU: return relu_keep_nan(v);   // (NaN < 0 is false: a NaN stays a NaN)
    case KEEP_ACT_LRELU02: return v < 0.f ? 0.2f * v : v;
    case KEEP_ACT_LRELU01: return v < 0.f ? 0.1f * v : v;
    case KEEP_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case KEEP_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case KEEP_ACT_SILU: return v / (1.0f + expf(-v));
    default: return v;
  }
}

// Abramowitz-Stegun 7.1.26: |erf error| <= 1.5e-7 (below bf16 and fp32-accumulation noise); ~12 instructions instead of
// the ~40 of the exact erff -- the GELU epilogue of a 128x128 tile is 64 evaluations per thread.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
  const float y = 1.0f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}

__device__ __forceinline__ float act_apply_fast(float v, int act) {
  switch (act) {
    case KEEP_ACT_RELU: return relu_keep_nan(v);   // (NaN < 0 is false: a NaN stays a NaN)
    case KEEP_ACT_LRELU02: return v < 0.f ? 0.2f * v : v;
    case KEEP_ACT_LRELU01: return v < 0.f ? 0.1f * v : v;
    case KEEP_ACT_GELU: return 0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752440f));
    case KEEP_ACT_SIGMOID: return __frcp_rn(1.0f + __expf(-v));
    case KEEP_ACT_SILU: return v * __frcp_rn(1.0f + __expf(-v));
    default: return v;
  }
}

// KEEP_MMA_X3 policy forms: x * sigmoid(x) as v_exp_f32 + v_rcp_f32 (1 ulp each; <= ~3e-7 relative overall -- the same grade
// as the policy's 2^-22 products) in 6 VALU instructions instead of the 23 of expf + IEEE division.  exp2(+big) = inf ->
// rcp = 0 -> x * 0 = -0 for x -> -inf; exp2(-big) = 0 -> x for x -> +inf; NaN propagates.  KEEP_X3_EXACT_ACT=1 selects the
// library forms (p.fast = 0).
__device__ __forceinline__ float swish_x3(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}
__device__ __forceinline__ float pro_apply_x3(float v, int act) {
  if (act == KEEP_PRO_SWISH) return swish_x3(v);
  if (act == KEEP_PRO_RELU) return relu_keep_nan(v);
  return v;
}

__device__ __forceinline__ float pro_apply(float v, int act) {
  if (act == KEEP_PRO_SWISH) return v * (1.0f / (1.0f + expf(-v)));
  if (act == KEEP_PRO_RELU) return relu_keep_nan(v);
  return v;
}
