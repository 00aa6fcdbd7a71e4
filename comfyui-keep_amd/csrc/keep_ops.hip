// HBM-bound kernels of the KEEP hot path: normalisation statistics, LayerNorm, GEGLU, argmax + codebook gather,
// Kalman blend, bilinear flow warp, convex flow upsampling, layout / elementwise helpers.  All are one read of the
// inputs + one write of the outputs with coalesced channel-contiguous (NHWC) accesses; reductions are wave-level
// (64-lane shuffles) with an LDS step across waves.
#include <math.h>

#include "keep_common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------------------------------------ channel stats
// grid (P, N), block 256.  Thread t owns channel(s) c = t % C (+256 stride when C > 256) and walks the chunk's
// pixels with stride 256/C: consecutive lanes read consecutive channels of one pixel (coalesced).
__global__ __launch_bounds__(256) void chan_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW,
                                                         int C, int ld, int P) {
  __shared__ float red[2][256];
  const int n = blockIdx.y, pch = blockIdx.x;
  const int per = (HW + P - 1) / P;
  const int p0 = pch * per, p1 = min(HW, p0 + per);
  const int tid = threadIdx.x;
  const float* xb = x + (long)n * HW * ld;
  for (int cbase = 0; cbase < C; cbase += 256) {
    const int cw = min(256, C - cbase);       // channels handled in this pass
    const int lanes_per_pix = cw;             // threads [0, cw*rows) active
    const int rows = 256 / cw > 0 ? 256 / cw : 1;
    float s = 0.f, ss = 0.f;
    if (tid < rows * cw) {
      const int c = cbase + tid % cw;
      for (int px = p0 + tid / cw; px < p1; px += rows) {
        const float v = xb[(long)px * ld + c];
        s += v;
        ss += v * v;
      }
    }
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
    if (tid < cw) {
      float a = 0.f, b2 = 0.f;
      for (int r = 0; r < rows; ++r) {
        a += red[0][r * cw + tid];
        b2 += red[1][r * cw + tid];
      }
      float* dst = part + (((long)n * P + pch) * C + cbase + tid) * 2;
      dst[0] = a;
      dst[1] = b2;
    }
    __syncthreads();
    (void)lanes_per_pix;
  }
}

extern "C" int32_t keep_chan_stats(const float* x, float* part, int32_t N, int32_t HW, int32_t C, int32_t ld, int32_t P,
                                   void* stream) {
  KEEP_REQUIRE(x && part && N > 0 && HW > 0 && C > 0 && ld >= C && P > 0 && P <= HW, "keep_chan_stats: bad args");
  hipLaunchKernelGGL(chan_stats_kernel, dim3(P, N), dim3(256), 0, (hipStream_t)stream, x, part, HW, C, ld, P);
  KEEP_LAUNCH_CHECK("keep_chan_stats");
  return KEEP_OK;
}

// grid (G, N), block 256: reduce P x (C/G) partials (float2 loads, fp64 accumulation), write scale/shift for the
// group's channels.  P is up to 4096 per image with the per-wave partials of the persistent halo kernel.
__global__ __launch_bounds__(256) void norm_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ scale,
                                                            float* __restrict__ shift, int HW, int C, int G, int P,
                                                            float eps) {
  __shared__ double red[2][4];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  const int total = P * cpg;
  const float2* base = reinterpret_cast<const float2*>(part) + (long)n * P * C + g * cpg;
  double s = 0.0, ss = 0.0;
  for (int i = tid; i < total; i += 256) {
    const int pch = i / cpg, cc = i - pch * cpg;
    const float2 v = base[(long)pch * C + cc];
    s += (double)v.x;
    ss += (double)v.y;
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s;
    red[1][tid >> 6] = ss;
  }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const double cnt = (double)HW * cpg;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int cc = tid; cc < cpg; cc += 256) {
    const int c = g * cpg + cc;
    const float ga = gamma ? gamma[c] : 1.f;
    const float be = beta ? beta[c] : 0.f;
    const float sc = ga * rstd;
    scale[(long)n * C + c] = sc;
    shift[(long)n * C + c] = be - (float)mean * sc;
  }
}

extern "C" int32_t keep_norm_finalize(const float* part, const float* gamma, const float* beta, float* scale,
                                      float* shift, int32_t N, int32_t HW, int32_t C, int32_t G, int32_t P, float eps,
                                      void* stream) {
  KEEP_REQUIRE(part && scale && shift && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && P > 0,
               "keep_norm_finalize: bad args (C=%d G=%d)", C, G);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, part, gamma, beta, scale, shift,
                     HW, C, G, P, eps);
  KEEP_LAUNCH_CHECK("keep_norm_finalize");
  return KEEP_OK;
}

// Small maps (H*W <= 4096): one block per (image, group) reduces the whole group in one go and writes scale/shift
// directly -- one launch instead of partial-stats + finalize, and 32*N blocks instead of a handful.
__global__ __launch_bounds__(256) void group_stats_small_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ scale,
                                                                float* __restrict__ shift, int HW, int C, int G, float eps) {
  __shared__ double red[2][4];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  const float* xb = x + (long)n * HW * C + g * cpg;
  double s = 0.0, ss = 0.0;
  if ((cpg & 3) == 0) {
    const int c4 = cpg >> 2;
    for (int i = tid; i < HW * c4; i += 256) {
      const int px = i / c4, c = (i - px * c4) << 2;
      const float4 v = *reinterpret_cast<const float4*>(xb + (long)px * C + c);
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      ss += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  } else {
    for (int i = tid; i < HW * cpg; i += 256) {
      const int px = i / cpg, c = i - px * cpg;
      const float v = xb[(long)px * C + c];
      s += (double)v;
      ss += (double)v * v;
    }
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s;
    red[1][tid >> 6] = ss;
  }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  ss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const double cnt = (double)HW * cpg;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int cc = tid; cc < cpg; cc += 256) {
    const int c = g * cpg + cc;
    const float ga = gamma ? gamma[c] : 1.f;
    const float be = beta ? beta[c] : 0.f;
    const float sc = ga * rstd;
    scale[(long)n * C + c] = sc;
    shift[(long)n * C + c] = be - (float)mean * sc;
  }
}

extern "C" int32_t keep_group_stats(const float* x, const float* gamma, const float* beta, float* scale, float* shift,
                                    int32_t N, int32_t HW, int32_t C, int32_t G, float eps, void* stream) {
  KEEP_REQUIRE(x && scale && shift && N > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "keep_group_stats: bad args");
  KEEP_REQUIRE((uintptr_t)x % 16 == 0 && C % 4 == 0, "keep_group_stats: needs 16-byte aligned x and C %% 4 == 0");
  hipLaunchKernelGGL(group_stats_small_kernel, dim3(G, N), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, scale,
                     shift, HW, C, G, eps);
  KEEP_LAUNCH_CHECK("keep_group_stats");
  return KEEP_OK;
}

__global__ void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                  const float* __restrict__ shift, float* __restrict__ out, long total, long per_n, int C,
                                  int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / per_n;
    const int c = (int)(i % C);
    float v = x[i] * scale[n * C + c] + shift[n * C + c];
    out[i] = act_apply(v, act);
  }
}

// C % 4 == 0 and 16-byte aligned tensors: one float4 per lane (the scalar form above moves 4 bytes per lane and access)
__global__ __launch_bounds__(256) void affine_act4_kernel(const float4* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, float4* __restrict__ out, long total4,
                                                          long per_n4, int C4, int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long n = i / per_n4;
    const int c = (int)(i % C4) * 4;
    const float4 v = x[i];
    const float4 sc = *reinterpret_cast<const float4*>(scale + n * C4 * 4 + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + n * C4 * 4 + c);
    out[i] = make_float4(act_apply(v.x * sc.x + sh.x, act), act_apply(v.y * sc.y + sh.y, act), act_apply(v.z * sc.z + sh.z, act),
                         act_apply(v.w * sc.w + sh.w, act));
  }
}

extern "C" int32_t keep_affine_act(const float* x, const float* scale, const float* shift, float* out, int32_t N,
                                   int32_t HW, int32_t C, int32_t act, void* stream) {
  KEEP_REQUIRE(x && scale && shift && out && N > 0 && HW > 0 && C > 0, "keep_affine_act: bad args");
  const long total = (long)N * HW * C;
  if (C % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)scale % 16 == 0 && (uintptr_t)shift % 16 == 0) {
    int blocks4 = cdiv(total / 4, 256);
    if (blocks4 > 16384) blocks4 = 16384;
    hipLaunchKernelGGL(affine_act4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, (const float4*)x, scale, shift,
                       (float4*)out, total / 4, (long)HW * C / 4, C / 4, act);
    KEEP_LAUNCH_CHECK("keep_affine_act");
    return KEEP_OK;
  }
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(affine_act_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift, out, total,
                     (long)HW * C, C, act);
  KEEP_LAUNCH_CHECK("keep_affine_act");
  return KEEP_OK;
}

// out_bf16 = bf16( act_pro( x*scale[n,c] + shift[n,c] ) ): the normalise+activate pass that feeds the 3x3 halo
// convolution (8 channels per thread: two float4 loads, one 16-byte store; RNE via v_cvt_pk_bf16_f32).
typedef __attribute__((ext_vector_type(8))) __bf16 ops_bf16x8;
template <bool IN_BF16, int ACT>
__global__ __launch_bounds__(256) void norm_act_bf16_kernel(const void* __restrict__ xin, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, ops_bf16x8* __restrict__ out,
                                                            long total8, long per_n8, int C8) {
  // two independent 8-element groups per thread and iteration: twice the bytes in flight per wave (the kernel is a pure
  // HBM stream: 4 or 2 bytes in, 2 bytes out per element)
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total8; i0 += 2 * stride) {
    float v[2][8];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = i0 + u * stride;
      ok[u] = i < total8;
      if (!ok[u]) continue;
      if (IN_BF16) {
        const ops_bf16x8 h = reinterpret_cast<const ops_bf16x8*>(xin)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = (float)h[j];
      } else {
        const float4 a = reinterpret_cast<const float4*>(xin)[2 * i];
        const float4 b = reinterpret_cast<const float4*>(xin)[2 * i + 1];
        v[u][0] = a.x; v[u][1] = a.y; v[u][2] = a.z; v[u][3] = a.w; v[u][4] = b.x; v[u][5] = b.y; v[u][6] = b.z; v[u][7] = b.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      const long i = i0 + u * stride;
      if (scale) {
        const long n = i / per_n8;
        const int c = (int)(i % C8) * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(scale + n * C8 * 8 + c);
        const float4 s1 = *reinterpret_cast<const float4*>(scale + n * C8 * 8 + c + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(shift + n * C8 * 8 + c);
        const float4 h1 = *reinterpret_cast<const float4*>(shift + n * C8 * 8 + c + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[u][j] = v[u][j] * sc[j] + sh[j];
      }
      ops_bf16x8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = v[u][j];
        if (ACT == KEEP_PRO_SWISH) t = t * __frcp_rn(1.0f + __expf(-t));
        else if (ACT == KEEP_PRO_RELU) t = relu_keep_nan(t);
        h[j] = (__bf16)t;
      }
      out[i] = h;
    }
  }
}

extern "C" int32_t keep_norm_act_bf16(const void* x, const float* scale, const float* shift, void* out, int32_t N,
                                      int32_t HW, int32_t C, int32_t act, int32_t in_dtype, void* stream) {
  KEEP_REQUIRE(x && out && N > 0 && HW > 0 && C > 0 && C % 8 == 0, "keep_norm_act_bf16: bad args (C=%d)", C);
  KEEP_REQUIRE((scale == nullptr) == (shift == nullptr), "keep_norm_act_bf16: scale/shift must pair");
  KEEP_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "keep_norm_act_bf16: 16-byte alignment");
  KEEP_REQUIRE(in_dtype == KEEP_F32 || in_dtype == KEEP_BF16, "keep_norm_act_bf16: bad in_dtype %d", in_dtype);
  const long total8 = (long)N * HW * C / 8;
  int blocks = cdiv(total8, 256);
  if (blocks > 16384) blocks = 16384;
#define KEEP_NA_LAUNCH(INB, ACTV)                                                                                       \
  hipLaunchKernelGGL((norm_act_bf16_kernel<INB, ACTV>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale, shift, \
                     (ops_bf16x8*)out, total8, (long)HW * C / 8, C / 8)
  KEEP_REQUIRE(act == KEEP_PRO_NONE || act == KEEP_PRO_SWISH || act == KEEP_PRO_RELU, "keep_norm_act_bf16: bad act %d", act);
  if (in_dtype == KEEP_BF16) {
    if (act == KEEP_PRO_SWISH) KEEP_NA_LAUNCH(true, KEEP_PRO_SWISH);
    else if (act == KEEP_PRO_RELU) KEEP_NA_LAUNCH(true, KEEP_PRO_RELU);
    else KEEP_NA_LAUNCH(true, KEEP_PRO_NONE);
  } else {
    if (act == KEEP_PRO_SWISH) KEEP_NA_LAUNCH(false, KEEP_PRO_SWISH);
    else if (act == KEEP_PRO_RELU) KEEP_NA_LAUNCH(false, KEEP_PRO_RELU);
    else KEEP_NA_LAUNCH(false, KEEP_PRO_NONE);
  }
#undef KEEP_NA_LAUNCH
  KEEP_LAUNCH_CHECK("keep_norm_act_bf16");
  return KEEP_OK;
}

__global__ void gm_join_kernel(const float* __restrict__ a, const float* __restrict__ sa, const float* __restrict__ ha,
                               const float* __restrict__ b, const float* __restrict__ sb, const float* __restrict__ hb,
                               float* __restrict__ out, long total, long per_n, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / per_n;
    const int c = (int)(i % C);
    float xa = a[i];
    if (sa) xa = xa * sa[n * C + c] + ha[n * C + c];
    float yb = b[i] * sb[n * C + c] + hb[n * C + c];
    yb = relu_keep_nan(yb);
    const float v = xa + yb;
    out[i] = relu_keep_nan(v);
  }
}

__global__ __launch_bounds__(256) void gm_join4_kernel(const float4* __restrict__ a, const float* __restrict__ sa,
                                                       const float* __restrict__ ha, const float4* __restrict__ b,
                                                       const float* __restrict__ sb, const float* __restrict__ hb,
                                                       float4* __restrict__ out, long total4, long per_n4, int C4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long n = i / per_n4;
    const long o = n * C4 * 4 + (i % C4) * 4;
    float4 xa = a[i];
    if (sa) {
      const float4 s = *reinterpret_cast<const float4*>(sa + o), h = *reinterpret_cast<const float4*>(ha + o);
      xa = make_float4(xa.x * s.x + h.x, xa.y * s.y + h.y, xa.z * s.z + h.z, xa.w * s.w + h.w);
    }
    const float4 yb = b[i];
    const float4 s = *reinterpret_cast<const float4*>(sb + o), h = *reinterpret_cast<const float4*>(hb + o);
    // (relu_keep_nan: v < 0 ? 0 : v -- fmaxf(NaN, 0) would be 0 and hide an upstream overflow)
    const float r0 = relu_keep_nan(yb.x * s.x + h.x), r1 = relu_keep_nan(yb.y * s.y + h.y);
    const float r2 = relu_keep_nan(yb.z * s.z + h.z), r3 = relu_keep_nan(yb.w * s.w + h.w);
    out[i] = make_float4(relu_keep_nan(xa.x + r0), relu_keep_nan(xa.y + r1), relu_keep_nan(xa.z + r2), relu_keep_nan(xa.w + r3));
  }
}

extern "C" int32_t keep_gm_join(const float* a, const float* sa, const float* ha, const float* b, const float* sb,
                                const float* hb, float* out, int32_t N, int32_t HW, int32_t C, void* stream) {
  KEEP_REQUIRE(a && b && sb && hb && out && N > 0 && HW > 0 && C > 0, "keep_gm_join: bad args");
  KEEP_REQUIRE((sa == nullptr) == (ha == nullptr), "keep_gm_join: sa/ha must pair");
  const long total = (long)N * HW * C;
  const bool al = (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)sb % 16 == 0 &&
                  (uintptr_t)hb % 16 == 0 && (!sa || ((uintptr_t)sa % 16 == 0 && (uintptr_t)ha % 16 == 0));
  if (C % 4 == 0 && al) {
    int blocks4 = cdiv(total / 4, 256);
    if (blocks4 > 16384) blocks4 = 16384;
    hipLaunchKernelGGL(gm_join4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, (const float4*)a, sa, ha,
                       (const float4*)b, sb, hb, (float4*)out, total / 4, (long)HW * C / 4, C / 4);
    KEEP_LAUNCH_CHECK("keep_gm_join");
    return KEEP_OK;
  }
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gm_join_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, sa, ha, b, sb, hb, out, total,
                     (long)HW * C, C);
  KEEP_LAUNCH_CHECK("keep_gm_join");
  return KEEP_OK;
}

__global__ void absmax_zero_kernel(unsigned* p, int n);      // (defined with the range probe below)

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; C <= 1024 (C/64 <= 16 values per lane kept in registers); two-pass mean/variance.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ res,
                                                        float* __restrict__ out, const float* __restrict__ pos,
                                                        int pos_rows, float* __restrict__ out2, int M, int C, float eps,
                                                        unsigned* __restrict__ amax_bits = nullptr, int rows_per_img = 1) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool valid = row < M;
  if (!valid && !amax_bits) return;
  unsigned mx = 0u;
  if (valid) {
    const float* xr = x + (long)row * C;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + j * 64;
      v[j] = c < C ? xr[c] : 0.f;
      s += v[j];
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + j * 64;
      const float d = c < C ? v[j] - mean : 0.f;
      q += d * d;
    }
    const float var = wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = lane + j * 64;
      if (c < C) {
        const float y = (v[j] - mean) * rstd * gamma[c] + beta[c];
        const float o = res ? y + res[(long)row * C + c] : y;
        out[(long)row * C + c] = o;
        mx = max(mx, __float_as_uint(o) & 0x7fffffffu);
        if (out2) out2[(long)row * C + c] = y + pos[(long)(row % pos_rows) * C + c];
      }
    }
  }
  if (amax_bits) {      // keep_layernorm_amax: the consumer's x3 range scale (max |out| per image) without a probe launch; see absmax_kernel.
    // One atomic per BLOCK when its four rows lie in one image (same-line atomics retire at ~12 ns each), else one per wave.
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    __shared__ unsigned wave_m[4];
    const int row0 = blockIdx.x * 4, row3 = min(row0 + 3, M - 1);
    const bool one_img = row0 / rows_per_img == row3 / rows_per_img;
    if (lane == 0) wave_m[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (one_img) {
      if (threadIdx.x == 0) {
        mx = max(max(wave_m[0], wave_m[1]), max(wave_m[2], wave_m[3]));
        unsigned* dst = amax_bits + row0 / rows_per_img;
        if (mx > *reinterpret_cast<volatile unsigned*>(dst)) atomicMax(dst, mx);
      }
    } else if (valid && lane == 0) {
      unsigned* dst = amax_bits + row / rows_per_img;
      if (mx > *reinterpret_cast<volatile unsigned*>(dst)) atomicMax(dst, mx);
    }
  }
}

// C == 128 (the GMFlow token stream, 2.5 M rows per call at 16 clips): 16 lanes per row, two float4 per lane -- the generic
// kernel spends a whole wave and 16 predicated iterations on 512 bytes.  Same two-pass arithmetic.
__global__ __launch_bounds__(256) void layernorm128_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ res,
                                                           float* __restrict__ out, long M, float eps) {
  const int sub = threadIdx.x & 15;                       // float4 columns sub and sub + 16
  const float4 g0 = *reinterpret_cast<const float4*>(gamma + sub * 4), g1 = *reinterpret_cast<const float4*>(gamma + 64 + sub * 4);
  const float4 b0 = *reinterpret_cast<const float4*>(beta + sub * 4), b1 = *reinterpret_cast<const float4*>(beta + 64 + sub * 4);
  for (long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4); row < M; row += (long)gridDim.x * 16) {
    const float* xr = x + row * 128;
    const float4 a = *reinterpret_cast<const float4*>(xr + sub * 4), b = *reinterpret_cast<const float4*>(xr + 64 + sub * 4);
    float s = (a.x + a.y) + (a.z + a.w) + (b.x + b.y) + (b.z + b.w);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / 128.0f);
    const float d[8] = {a.x - mean, a.y - mean, a.z - mean, a.w - mean, b.x - mean, b.y - mean, b.z - mean, b.w - mean};
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) q += d[j] * d[j];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 128.0f) + eps);
    float4 y0 = make_float4(d[0] * rstd * g0.x + b0.x, d[1] * rstd * g0.y + b0.y, d[2] * rstd * g0.z + b0.z, d[3] * rstd * g0.w + b0.w);
    float4 y1 = make_float4(d[4] * rstd * g1.x + b1.x, d[5] * rstd * g1.y + b1.y, d[6] * rstd * g1.z + b1.z, d[7] * rstd * g1.w + b1.w);
    if (res) {
      const float4 r0 = *reinterpret_cast<const float4*>(res + row * 128 + sub * 4);
      const float4 r1 = *reinterpret_cast<const float4*>(res + row * 128 + 64 + sub * 4);
      y0.x += r0.x; y0.y += r0.y; y0.z += r0.z; y0.w += r0.w;
      y1.x += r1.x; y1.y += r1.y; y1.z += r1.z; y1.w += r1.w;
    }
    *reinterpret_cast<float4*>(out + row * 128 + sub * 4) = y0;
    *reinterpret_cast<float4*>(out + row * 128 + 64 + sub * 4) = y1;
  }
}

extern "C" int32_t keep_layernorm(const float* x, const float* gamma, const float* beta, const float* res, float* out,
                                  const float* pos, int32_t pos_rows, float* out2, int32_t M, int32_t C, float eps,
                                  void* stream) {
  KEEP_REQUIRE(x && gamma && beta && out && M > 0 && C > 0 && C <= 1024, "keep_layernorm: bad args (C=%d)", C);
  KEEP_REQUIRE(!out2 || (pos && pos_rows > 0), "keep_layernorm: out2 requires pos");
  if (C == 128 && !out2 && M >= 4096 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)gamma % 16 == 0 &&
      (uintptr_t)beta % 16 == 0 && (!res || (uintptr_t)res % 16 == 0)) {
    int blocks = cdiv(M, 16);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(layernorm128_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, res, out, (long)M, eps);
    KEEP_LAUNCH_CHECK("keep_layernorm(C=128)");
    return KEEP_OK;
  }
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, res, out, pos,
                     pos_rows > 0 ? pos_rows : 1, out2, M, C, eps);
  KEEP_LAUNCH_CHECK("keep_layernorm");
  return KEEP_OK;
}

// keep_layernorm (without pos / out2) + amax[n] = max |out| over image n's rows_per_image rows: the range scale of the x3 GEMM that
// consumes the output, fused (bit patterns of |x| order like the values: an unsigned atomicMax is exact and order-independent).
extern "C" int32_t keep_layernorm_amax(const float* x, const float* gamma, const float* beta, const float* res, float* out, int32_t M,
                                       int32_t C, float eps, int32_t rows_per_image, float* amax, int32_t zeroed, void* stream) {
  KEEP_REQUIRE(x && gamma && beta && out && amax && M > 0 && C > 0 && C <= 1024 && rows_per_image > 0 && M % rows_per_image == 0,
               "keep_layernorm_amax: bad args (M=%d C=%d rows_per_image=%d)", M, C, rows_per_image);
  hipStream_t st = (hipStream_t)stream;
  const int N = M / rows_per_image;
  if (!zeroed) hipLaunchKernelGGL(absmax_zero_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(amax), N);
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, x, gamma, beta, res, out, (const float*)nullptr, 1,
                     (float*)nullptr, M, C, eps, reinterpret_cast<unsigned*>(amax), rows_per_image);
  KEEP_LAUNCH_CHECK("keep_layernorm_amax");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ GEGLU
__global__ void geglu_kernel(const float* __restrict__ x, float* __restrict__ out, long total, int F) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / F;
    const int j = (int)(i - m * F);
    const float h = x[m * 2 * F + j];
    const float g = x[m * 2 * F + F + j];
    out[i] = h * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)));
  }
}

extern "C" int32_t keep_geglu(const float* x, float* out, int32_t M, int32_t F, void* stream) {
  KEEP_REQUIRE(x && out && M > 0 && F > 0, "keep_geglu: bad args");
  const long total = (long)M * F;
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(geglu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, total, F);
  KEEP_LAUNCH_CHECK("keep_geglu");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ argmax + gather
// one wave per token: 64 lanes scan the logits row (coalesced), wave arg-max with lowest-index tie-break,
// then the wave copies the chosen codebook row.
__global__ __launch_bounds__(256) void argmax_gather_kernel(const float* __restrict__ logits,
                                                            const float* __restrict__ codebook,
                                                            const int* __restrict__ force_idx, int* __restrict__ idx,
                                                            float* __restrict__ margin, float* __restrict__ out, int M,
                                                            int ncodes, int dim, int* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* lr = logits + (long)row * ncodes;
  float best = -INFINITY, second = -INFINITY;
  int bi = 0x7fffffff;
  for (int j = lane; j < ncodes; j += 64) {
    const float v = lr[j];
    if (v > best) {
      second = best;
      best = v;
      bi = j;
    } else if (v > second) {
      second = v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o);
    const float os = __shfl_xor(second, o);
    const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) {
      second = fmaxf(best, os);
      best = ob;
      bi = oi;
    } else {
      second = fmaxf(second, ob);
    }
  }
  // The arg-max is where a NaN / inf (an fp16-range overflow of the x3 policy upstream) would become a finite, plausible,
  // WRONG code: v > best is false for every NaN, so an all-NaN row keeps bi = 0x7fffffff, and a row with some NaNs picks
  // among the rest.  Scan result not finite, or any NaN seen in the row -> the row is flagged: status bit, NaN-filled output.
  bool bad = !(fabsf(best) <= 3.0e38f) || bi < 0 || bi >= ncodes;
  {
    bool nan_seen = false;
    for (int j = lane; j < ncodes; j += 64) nan_seen |= (lr[j] != lr[j]);
    bad |= (__ballot(nan_seen) != 0ull);
  }
  int sel = bi;
  if (force_idx) {
    sel = force_idx[row];
    bad = false;                             // injected indices (parity tests): the logits do not decide anything
  }
  if (sel < 0 || sel >= ncodes) sel = 0;     // never a wild gather
  if (lane == 0) {
    if (idx) idx[row] = sel;
    if (margin) margin[row] = best - second;
    if (bad && status) atomicOr(status, KEEP_STATUS_NONFINITE_LOGITS);
  }
  const float* cb = codebook + (long)sel * dim;
  const float nanv = __builtin_nanf("");
  for (int d = lane; d < dim; d += 64) out[(long)row * dim + d] = bad ? nanv : cb[d];
}

extern "C" int32_t keep_argmax_gather(const float* logits, const float* codebook, const int32_t* force_idx, int32_t* idx,
                                      float* margin, float* out, int32_t M, int32_t ncodes, int32_t dim, int32_t* status,
                                      void* stream) {
  KEEP_REQUIRE(logits && codebook && out && M > 0 && ncodes > 0 && dim > 0, "keep_argmax_gather: bad args");
  hipLaunchKernelGGL(argmax_gather_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, logits, codebook,
                     force_idx, idx, margin, out, M, ncodes, dim, status);
  KEEP_LAUNCH_CHECK("keep_argmax_gather");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ per-pixel class arg-max
// out[m] = argmax_c x[m, c], c < C (lowest index on ties), x rows of pitch ld floats: ParseNet's out.argmax(dim=1)
// (face_restoration_helper.py:424) on the channels-last logits; one thread per pixel, 16-byte loads when ld % 4 == 0.
__global__ __launch_bounds__(256) void channel_argmax_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long M,
                                                             int C, int ld) {
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const float* r = x + m * ld;
    float best = r[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const float v = r[c];
      if (v > best) {
        best = v;
        bi = c;
      }
    }
    out[m] = (unsigned char)bi;
  }
}

extern "C" int32_t keep_channel_argmax(const float* x, uint8_t* out, int64_t M, int32_t C, int32_t ld, void* stream) {
  KEEP_REQUIRE(x && out && M > 0 && C > 0 && C <= 256 && ld >= C, "keep_channel_argmax: bad args");
  int blocks = cdiv(M, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(channel_argmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)M, C, ld);
  KEEP_LAUNCH_CHECK("keep_channel_argmax");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ detector helpers
// nn.MaxPool2d(kernel 3, stride 2, padding 1) of torchvision's ResNet stem on NHWC maps (padding = -inf); C % 4 == 0.
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float4* __restrict__ x, float4* __restrict__ out, int N, int H, int W,
                                                         int Ho, int Wo, int C4) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = x[(((long)n * H + iy) * W + ix) * C4 + c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    out[i] = m;
  }
}

extern "C" int32_t keep_maxpool3s2(const float* x, float* out, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
  KEEP_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0,
               "keep_maxpool3s2: bad args (C %% 4 == 0, 16-byte aligned tensors)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(out), N, H, W, Ho, Wo, C / 4);
  KEEP_LAUNCH_CHECK("keep_maxpool3s2");
  return KEEP_OK;
}

// Depthwise 3x3 convolution (padding 1, stride 1 | 2) + bias + activation on NHWC maps: the first half of every conv_dw block of
// the retinaface_mobile0.25 trunk (retinaface_net.py:25-34), BatchNorm folded into w / bias by the host.  Pure streaming work
// (9 MACs per loaded float): one thread owns 4 channels of one output pixel, the 9 taps are float4 loads that neighbouring
// threads (consecutive channel groups, then consecutive pixels) coalesce; the input row is re-read from L2 by the 3 output rows
// that touch it.  Taps accumulate ky-major, kx-minor (the order oracle/facelib_oracle.py restates).
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float4* __restrict__ x, const float4* __restrict__ w,
                                                        const float4* __restrict__ bias, float4* __restrict__ out, int N, int H, int W,
                                                        int Ho, int Wo, int C4, int stride, int act) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float4 a = bias ? bias[c] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * stride - 1 + ky;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * stride - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = x[(((long)n * H + iy) * W + ix) * C4 + c];
        const float4 k = w[(ky * 3 + kx) * C4 + c];
        a.x = fmaf(v.x, k.x, a.x); a.y = fmaf(v.y, k.y, a.y); a.z = fmaf(v.z, k.z, a.z); a.w = fmaf(v.w, k.w, a.w);
      }
    }
    out[i] = make_float4(act_apply(a.x, act), act_apply(a.y, act), act_apply(a.z, act), act_apply(a.w, act));
  }
}

extern "C" int32_t keep_dwconv3x3(const float* x, const float* w, const float* bias, float* out, int32_t N, int32_t H, int32_t W,
                                  int32_t C, int32_t stride, int32_t act, void* stream) {
  KEEP_REQUIRE(x && w && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2) &&
                   act >= KEEP_ACT_NONE && act <= KEEP_ACT_SILU && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0 &&
                   (uintptr_t)out % 16 == 0 && (uintptr_t)bias % 16 == 0,
               "keep_dwconv3x3: bad args (C %% 4 == 0, stride 1 | 2, 16-byte aligned tensors)");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(dwconv3x3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<const float4*>(w), reinterpret_cast<const float4*>(bias), reinterpret_cast<float4*>(out), N, H, W,
                     Ho, Wo, C / 4, stride, act);
  KEEP_LAUNCH_CHECK("keep_dwconv3x3");
  return KEEP_OK;
}

// ---- YOLOv5-face helpers (wm_facelib/detection/yolov5face/models/common.py, yolo.py): all HBM-bound float4 passes
// nn.MaxPool2d(k, stride, pad, ceil_mode) on a channel slice (padding = -inf; a ceil-mode window may hang over the right / bottom edge)
__global__ __launch_bounds__(256) void maxpool2d_kernel(const float* __restrict__ x, float* __restrict__ out, int N, int H, int W, int C4,
                                                        int in_ld, int out_ld, int k, int stride, int pad, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * stride - pad + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * stride - pad + kx;
        if (ix < 0 || ix >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + iy) * W + ix) * in_ld + c * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    *reinterpret_cast<float4*>(out + (((long)n * Ho + oy) * Wo + ox) * out_ld + c * 4) = m;
  }
}

extern "C" int32_t keep_maxpool2d(const float* x, float* out, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_ld, int32_t out_ld,
                                  int32_t k, int32_t stride, int32_t pad, int32_t Ho, int32_t Wo, void* stream) {
  KEEP_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && in_ld >= C && out_ld >= C && in_ld % 4 == 0 && out_ld % 4 == 0 &&
                   k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && 2 * pad <= k && Ho > 0 && Wo > 0 &&
                   (long)(Ho - 1) * stride - pad < H && (long)(Wo - 1) * stride - pad < W && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0,
               "keep_maxpool2d: bad args (C, in_ld, out_ld %% 4 == 0, k <= 15, every window meets the map, 16-byte aligned slices)");
  const long total = (long)N * Ho * Wo * (C / 4);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(maxpool2d_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, N, H, W, C / 4, in_ld, out_ld, k, stride, pad,
                     Ho, Wo);
  KEEP_LAUNCH_CHECK("keep_maxpool2d");
  return KEEP_OK;
}

__global__ __launch_bounds__(256) void slice_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int H, int W, int C4,
                                                         int src_ld, int dst_ld, int up) {
  const long total = (long)N * H * W * C4;
  const int Hs = H >> up, Ws = W >> up;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    *reinterpret_cast<float4*>(dst + (((long)n * H + y) * W + x) * dst_ld + c * 4) =
        *reinterpret_cast<const float4*>(src + (((long)n * Hs + (y >> up)) * Ws + (x >> up)) * src_ld + c * 4);
  }
}

extern "C" int32_t keep_slice_copy(const float* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t src_ld, int32_t dst_ld,
                                   int32_t up, void* stream) {
  KEEP_REQUIRE(src && dst && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && src_ld >= C && dst_ld >= C && src_ld % 4 == 0 &&
                   dst_ld % 4 == 0 && (up == 0 || (up == 1 && H % 2 == 0 && W % 2 == 0)) && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
               "keep_slice_copy: bad args (C, lds %% 4 == 0, up in {0, 1} with an even destination, 16-byte aligned slices)");
  const long total = (long)N * H * W * (C / 4);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(slice_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, N, H, W, C / 4, src_ld, dst_ld, up);
  KEEP_LAUNCH_CHECK("keep_slice_copy");
  return KEEP_OK;
}

// channel_shuffle(cat(a, b), 2): a thread interleaves 4 channels of a with 4 of b -> 8 consecutive output channels
__global__ __launch_bounds__(256) void channel_shuffle2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                               long rows, int h4, int a_ld, int b_ld) {
  const long total = rows * h4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % h4);
    const long r = i / h4;
    const float4 va = *reinterpret_cast<const float4*>(a + r * a_ld + c * 4);
    const float4 vb = *reinterpret_cast<const float4*>(b + r * b_ld + c * 4);
    float4* o = reinterpret_cast<float4*>(out + (r * h4 + c) * 8);
    o[0] = make_float4(va.x, vb.x, va.y, vb.y);
    o[1] = make_float4(va.z, vb.z, va.w, vb.w);
  }
}

extern "C" int32_t keep_channel_shuffle2(const float* a, const float* b, float* out, int64_t rows, int32_t half, int32_t a_ld, int32_t b_ld,
                                         void* stream) {
  KEEP_REQUIRE(a && b && out && rows > 0 && half > 0 && half % 4 == 0 && a_ld >= half && b_ld >= half && a_ld % 4 == 0 && b_ld % 4 == 0 &&
                   (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 && (uintptr_t)out % 16 == 0,
               "keep_channel_shuffle2: bad args (half, lds %% 4 == 0, 16-byte aligned)");
  const long total = (long)rows * (half / 4);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(channel_shuffle2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)rows, half / 4, a_ld, b_ld);
  KEEP_LAUNCH_CHECK("keep_channel_shuffle2");
  return KEEP_OK;
}

// Detect.forward (inference), one level: a thread decodes one (image, anchor, pixel) row of 16 values
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float* __restrict__ raw, float* __restrict__ pred, int N, int ny, int nx,
                                                          float stride, const float* __restrict__ anchors_wh, int row0, int rows_total) {
  const long total = (long)N * 3 * ny * nx;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long t = i;
    const int x = (int)(t % nx); t /= nx;
    const int y = (int)(t % ny); t /= ny;
    const int a = (int)(t % 3);
    const int n = (int)(t / 3);
    const float4* src = reinterpret_cast<const float4*>(raw + ((((long)n * ny + y) * nx + x) * 3 + a) * 16);
    const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
    const float aw = anchors_wh[2 * a], ah = anchors_wh[2 * a + 1];
    const float gx = (float)x, gy = (float)y;
    auto sg = [](float v) { return 1.0f / (1.0f + expf(-v)); };
    const float sw = sg(r0.z) * 2.0f, sh = sg(r0.w) * 2.0f;
    float4 o0, o1, o2, o3;
    o0.x = (sg(r0.x) * 2.0f - 0.5f + gx) * stride;       // (y * 2 - 0.5 + grid) * stride, yolo.py:58
    o0.y = (sg(r0.y) * 2.0f - 0.5f + gy) * stride;
    o0.z = sw * sw * aw;                                  // (y * 2) ** 2 * anchor_grid, yolo.py:59
    o0.w = sh * sh * ah;
    o1.x = sg(r1.x);                                      // objectness
    o1.y = r1.y * aw + gx * stride;                       // landmarks: raw * anchor_grid + grid * stride, yolo.py:60-74
    o1.z = r1.z * ah + gy * stride;
    o1.w = r1.w * aw + gx * stride;
    o2.x = r2.x * ah + gy * stride;
    o2.y = r2.y * aw + gx * stride;
    o2.z = r2.z * ah + gy * stride;
    o2.w = r2.w * aw + gx * stride;
    o3.x = r3.x * ah + gy * stride;
    o3.y = r3.y * aw + gx * stride;
    o3.z = r3.z * ah + gy * stride;
    o3.w = sg(r3.w);                                      // class score
    float4* dst = reinterpret_cast<float4*>(pred + ((long)n * rows_total + row0 + ((long)a * ny + y) * nx + x) * 16);
    dst[0] = o0; dst[1] = o1; dst[2] = o2; dst[3] = o3;
  }
}

extern "C" int32_t keep_yolo_decode(const float* raw, float* pred, int32_t N, int32_t ny, int32_t nx, float stride, const float* anchors_wh,
                                    int32_t row0, int32_t rows_total, void* stream) {
  KEEP_REQUIRE(raw && pred && anchors_wh && N > 0 && ny > 0 && nx > 0 && row0 >= 0 && (long)row0 + 3L * ny * nx <= rows_total &&
                   (uintptr_t)raw % 16 == 0 && (uintptr_t)pred % 16 == 0,
               "keep_yolo_decode: bad args (the level's rows lie inside pred, 16-byte aligned)");
  const long total = (long)N * 3 * ny * nx;
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(yolo_decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, raw, pred, N, ny, nx, stride, anchors_wh, row0,
                     rows_total);
  KEEP_LAUNCH_CHECK("keep_yolo_decode");
  return KEEP_OK;
}

// RetinaFace post-processing on the device (retinaface.py:208-256 per frame; retinaface_utils.py:254-294): per anchor the face
// score softmax(cls)[1], and -- only for anchors above the confidence threshold -- the decoded box and five landmarks in pixels,
// appended to the frame's compact list (one atomic slot per survivor).  heads: [N, P, 32] rows of the fused head convolution, per
// pixel [cls a0 a1 (2 each) | box a0 a1 (4 each) | landmarks a0 a1 (10 each)]; anchor index = 2 * pixel + a; priors [2P, 4]
// (cx, cy, w, h).  dets: [N, cap, 16] = x1 y1 x2 y2 score lm x 10 anchor-index (as float: < 2^24).  Only survivors cross PCIe:
// a 640 x 1138 frame has 60 160 anchors (3.9 MB of head outputs), a handful above 0.97.  The arithmetic is the host decoder's
// (engine/retinaface.py:decode_boxes / decode_landmarks), float32, in the same order.
__global__ __launch_bounds__(256) void retina_decode_kernel(const float* __restrict__ heads, const float* __restrict__ priors,
                                                            float* __restrict__ dets, int* __restrict__ counts, int N, int P, int cap,
                                                            float var0, float var1, float sx, float sy, float thr) {
  const long total = (long)N * P * 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int a = (int)(i & 1);
    const long pix = i >> 1;
    const int n = (int)(pix / P);
    const int p = (int)(pix - (long)n * P);
    const float* row = heads + pix * 32;
    const float c0 = row[2 * a], c1 = row[2 * a + 1];
    const float m = fmaxf(c0, c1);
    const float e0 = expf(c0 - m), e1 = expf(c1 - m);
    const float score = e1 / (e0 + e1);
    if (!(score > thr)) continue;
    const int slot = atomicAdd(counts + n, 1);
    if (slot >= cap) continue;                 // (the count still says how many there were: the host falls back when it exceeds cap)
    const int ai = 2 * p + a;
    const float4 pr = *reinterpret_cast<const float4*>(priors + 4L * ai);
    const float* lb = row + 4 + 4 * a;
    float bx = pr.x + lb[0] * var0 * pr.z, by = pr.y + lb[1] * var0 * pr.w;
    float bw = pr.z * expf(lb[2] * var1), bh = pr.w * expf(lb[3] * var1);
    bx -= bw / 2;
    by -= bh / 2;
    bw += bx;
    bh += by;
    float* d = dets + ((long)n * cap + slot) * 16;
    d[0] = bx * sx; d[1] = by * sy; d[2] = bw * sx; d[3] = bh * sy;
    d[4] = score;
    const float* lm = row + 12 + 10 * a;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      d[5 + 2 * k] = (pr.x + lm[2 * k] * var0 * pr.z) * sx;
      d[6 + 2 * k] = (pr.y + lm[2 * k + 1] * var0 * pr.w) * sy;
    }
    d[15] = (float)ai;
  }
}

extern "C" int32_t keep_retina_decode(const float* heads, const float* priors, float* dets, int32_t* counts, int32_t N, int32_t P,
                                      int32_t cap, float var0, float var1, float scale_x, float scale_y, float conf_threshold,
                                      void* stream) {
  KEEP_REQUIRE(heads && priors && dets && counts && N > 0 && P > 0 && cap > 0 && 2L * P < (1L << 24) && (uintptr_t)priors % 16 == 0,
               "keep_retina_decode: bad args (2 P < 2^24 anchors, 16-byte aligned priors)");
  const long total = (long)N * P * 2;
  int blocks = cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(retina_decode_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, heads, priors, dets, counts, N, P, cap, var0,
                     var1, scale_x, scale_y, conf_threshold);
  KEEP_LAUNCH_CHECK("keep_retina_decode");
  return KEEP_OK;
}

// RetinaFace.detect_faces, the rest of the post-processing on the device (retinaface.py:240-246; py_cpu_nms =
// torchvision.ops.nms, retinaface_utils.py:39-47): one block per frame orders the frame's survivors by descending score (a frame with two
// equal scores is handed back to the host: see below), suppresses greedily with
// the float32 IoU arithmetic of engine/retinaface.py:nms (areas (x2 - x1) * (y2 - y1), inter / (a_i + a_j - inter) > threshold, no
// contraction) and writes the kept rows, in order, to out[n, 0 .. out_counts[n]).  counts[n] > cap (the compact list overflowed), or two
// survivors with the same score (out_counts[n] = -2): the caller finishes that frame on the host.  Bitonic sort of (key, row) pairs in LDS, cap <= 4096.
#define NMS_MAX 4096
// order != NULL (keep_retina_nms_ordered): the caller ordered the frame's survivors itself (order[f, i] = row of rank i) -- the frames the
// sorting form hands back -- and only the suppression and the compaction run here.
__global__ __launch_bounds__(1024) void retina_nms_kernel(const float* __restrict__ dets, const int* __restrict__ counts,
                                                          float* __restrict__ out, int* __restrict__ out_counts, int cap, float thr,
                                                          const int* __restrict__ order) {
  __shared__ unsigned long long keys[NMS_MAX];
  __shared__ unsigned short rows[NMS_MAX];
  __shared__ float4 box[NMS_MAX];
  __shared__ float area[NMS_MAX];
  __shared__ unsigned char alive[NMS_MAX];
  __shared__ int scan[1024];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int cnt = counts[f];
  if (cnt > cap) {
    if (tid == 0) out_counts[f] = -1;
    return;
  }
  const int n = cnt;
  const float* fd = dets + (long)f * cap * 16;
  int npad = 1;
  while (npad < n) npad <<= 1;
  if (order) {
    for (int i = tid; i < n; i += 1024) rows[i] = (unsigned short)order[(long)f * cap + i];
    npad = 0;                              // (no sort, no tie check)
  }
  for (int i = tid; i < npad; i += 1024) {
    unsigned long long k = ~0ULL;
    if (i < n) {
      const unsigned sb = __float_as_uint(fd[i * 16 + 4]);                // scores are positive: their bits order like the values
      const unsigned an = (unsigned)fd[i * 16 + 15];
      k = ((unsigned long long)(~sb) << 32) | (unsigned long long)(0xFFFFFFFFu - an);
    }
    keys[i] = k;
    rows[i] = (unsigned short)i;
  }
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < npad; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const unsigned long long a = keys[i], b = keys[l];
          if ((a > b) == up) {
            keys[i] = b; keys[l] = a;
            const unsigned short r = rows[i]; rows[i] = rows[l]; rows[l] = r;
          }
        }
      }
      __syncthreads();
    }
  // Equal score BIT PATTERNS (a softmax saturating at exactly 1.0f): `scores.argsort()[::-1]` orders them by numpy's introsort, which is
  // not stable above 16 elements -- which of two overlapping equal-score boxes survives is numpy's to decide.  Such a frame is handed
  // back (out_counts = -2: the caller orders and suppresses the frame's device-decoded survivors with numpy itself) instead of being
  // ordered by a rule of our own.
  scan[tid] = 0;
  __syncthreads();
  for (int i = tid; i + 1 < n && !order; i += 1024)
    if ((keys[i] >> 32) == (keys[i + 1] >> 32)) scan[0] = 1;          // (benign race: every writer stores 1)
  __syncthreads();
  if (scan[0]) {
    if (tid == 0) out_counts[f] = -2;      // (-1: the compact list overflowed; -2: equal scores -- the survivors in `dets` are valid)
    return;
  }
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const float4 b = *reinterpret_cast<const float4*>(fd + (int)rows[i] * 16);
    box[i] = b;
    area[i] = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
    alive[i] = 1;
  }
  __syncthreads();
  for (int i = 0; i < n; ++i) {
    if (!alive[i]) continue;               // (uniform: alive[i] was last written before the previous barrier)
    const float4 bi = box[i];
    const float ai = area[i];
    for (int j = i + 1 + tid; j < n; j += 1024) {
      if (!alive[j]) continue;
      const float4 bj = box[j];
      const float w = fmaxf(__fsub_rn(fminf(bi.z, bj.z), fmaxf(bi.x, bj.x)), 0.f);
      const float h = fmaxf(__fsub_rn(fminf(bi.w, bj.w), fmaxf(bi.y, bj.y)), 0.f);
      const float inter = __fmul_rn(w, h);
      if (__fdiv_rn(inter, __fsub_rn(__fadd_rn(ai, area[j]), inter)) > thr) alive[j] = 0;
    }
    __syncthreads();
  }
  // order-preserving compaction: thread t owns candidates 4 t .. 4 t + 3
  int c = 0;
  for (int q = 0; q < 4; ++q) {
    const int i = tid * 4 + q;
    c += (i < n && alive[i]) ? 1 : 0;
  }
  scan[tid] = c;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int pos = scan[tid] - c;
  float* fo = out + (long)f * cap * 16;
  for (int q = 0; q < 4; ++q) {
    const int i = tid * 4 + q;
    if (i < n && alive[i]) {
      const float4* src = reinterpret_cast<const float4*>(fd + (int)rows[i] * 16);
      float4* dst = reinterpret_cast<float4*>(fo + pos * 16);
      dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
      ++pos;
    }
  }
  if (tid == 1023) out_counts[f] = scan[1023];
}

extern "C" int32_t keep_retina_nms(const float* dets, const int32_t* counts, float* out, int32_t* out_counts, int32_t N, int32_t cap,
                                   float iou_threshold, void* stream) {
  KEEP_REQUIRE(dets && counts && out && out_counts && N > 0 && cap > 0 && cap <= NMS_MAX && (uintptr_t)dets % 16 == 0 && (uintptr_t)out % 16 == 0,
               "keep_retina_nms: bad args (cap <= 4096 rows per frame, 16-byte aligned lists)");
  hipLaunchKernelGGL(retina_nms_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, dets, counts, out, out_counts, cap, iou_threshold,
                     (const int*)nullptr);
  KEEP_LAUNCH_CHECK("keep_retina_nms");
  return KEEP_OK;
}

extern "C" int32_t keep_retina_nms_ordered(const float* dets, const int32_t* counts, const int32_t* order, float* out, int32_t* out_counts,
                                           int32_t N, int32_t cap, float iou_threshold, void* stream) {
  KEEP_REQUIRE(dets && counts && order && out && out_counts && N > 0 && cap > 0 && cap <= NMS_MAX && (uintptr_t)dets % 16 == 0 &&
                   (uintptr_t)out % 16 == 0,
               "keep_retina_nms_ordered: bad args (cap <= 4096 rows per frame, 16-byte aligned lists)");
  hipLaunchKernelGGL(retina_nms_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, dets, counts, out, out_counts, cap, iou_threshold, order);
  KEEP_LAUNCH_CHECK("keep_retina_nms_ordered");
  return KEEP_OK;
}

// out[n, y, x, :] = a[n, y, x, :] + b[n, floor(y * hb / H), floor(x * wb / W), :]: the FPN top-down step
// `a + F.interpolate(b, size=a.shape[2:], mode='nearest')` (retinaface_net.py:86-92); C % 4 == 0.  (torch's nearest index is
// floor(dst * scale) with scale = in / out as a float; the integer form below equals it for every size with in <= out.)
__global__ __launch_bounds__(256) void upsample_add_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                           float4* __restrict__ out, int N, int H, int W, int hb, int wb, int C4) {
  const long total = (long)N * H * W * C4;
  const float sy = (float)hb / (float)H, sx = (float)wb / (float)W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int n = (int)(t / H);
    const int by = min((int)floorf((float)y * sy), hb - 1), bx = min((int)floorf((float)x * sx), wb - 1);
    const float4 u = a[i], v = b[(((long)n * hb + by) * wb + bx) * C4 + c];
    out[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}

extern "C" int32_t keep_upsample_add(const float* a, const float* b, float* out, int32_t N, int32_t H, int32_t W, int32_t hb,
                                     int32_t wb, int32_t C, void* stream) {
  KEEP_REQUIRE(a && b && out && N > 0 && H > 0 && W > 0 && hb > 0 && wb > 0 && C > 0 && C % 4 == 0 && (uintptr_t)a % 16 == 0 &&
                   (uintptr_t)b % 16 == 0 && (uintptr_t)out % 16 == 0,
               "keep_upsample_add: bad args (C %% 4 == 0, 16-byte aligned tensors)");
  const long total = (long)N * H * W * (C / 4);
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(upsample_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                     reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out), N, H, W, hb, wb, C / 4);
  KEEP_LAUNCH_CHECK("keep_upsample_add");
  return KEEP_OK;
}

// x = act(x) in place (KEEP_ACT_*): the ReLU that follows a Bottleneck's residual sum (the convolution epilogue adds the
// residual AFTER its own activation, so `relu(conv3 + identity)` is the fused sum plus this pass)
__global__ __launch_bounds__(256) void act_inplace_kernel(float4* __restrict__ x, long n4, int act) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = x[i];
    v.x = act_apply(v.x, act); v.y = act_apply(v.y, act); v.z = act_apply(v.z, act); v.w = act_apply(v.w, act);
    x[i] = v;
  }
}

extern "C" int32_t keep_act_inplace(float* x, int64_t n, int32_t act, void* stream) {
  KEEP_REQUIRE(x && n > 0 && n % 4 == 0 && (uintptr_t)x % 16 == 0, "keep_act_inplace: bad args (n %% 4 == 0, 16-byte aligned)");
  int blocks = cdiv(n / 4, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(act_inplace_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(x), (long)(n / 4), act);
  KEEP_LAUNCH_CHECK("keep_act_inplace");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ non-finite flag
// One pass over a tensor; a block raises the status bit at most once.  (exponent all ones <=> NaN or +-inf)
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, long n, int* __restrict__ status) {
  bool bad = false;
  const long n4 = n >> 2;
  const uint4* x4 = reinterpret_cast<const uint4*>(x);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const uint4 v = x4[i];
    bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) | ((v.z & 0x7f800000u) == 0x7f800000u) |
           ((v.w & 0x7f800000u) == 0x7f800000u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3))
    bad |= ((__float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u);
  if (__syncthreads_or(bad) && threadIdx.x == 0) atomicOr(status, KEEP_STATUS_NONFINITE_TENSOR);
}

extern "C" int32_t keep_nonfinite_flag(const float* x, int64_t n, int32_t* status, void* stream) {
  KEEP_REQUIRE(x && status && n > 0 && (uintptr_t)x % 16 == 0, "keep_nonfinite_flag: bad args (x must be 16-byte aligned)");
  int blocks = cdiv(n >> 2, 256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (long)n, status);
  KEEP_LAUNCH_CHECK("keep_nonfinite_flag");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ VQ nearest code
// VQ:37-48: idx[m] = argmin_j (|z_m|^2 + |e_j|^2) - 2 z_m . e_j -- a [M x dim] x [dim x ncodes] GEMM with an arg-min epilogue.
// One block (4 waves) per 64 tokens walks the codebook in tiles of 128 codes; per 32-dim K chunk the token rows and the code
// rows are staged in LDS with coalesced 16-byte loads (33-float pitch: the per-lane ds_read_b32 of the 32x32x2 f32 MFMA
// operand layout is conflict-free), z.e runs on v_mfma_f32_32x32x2_f32 (EXACT fp32 products -- an index search must not see
// rounded operands), |e|^2 / |z|^2 are reduced from the staged values with lane shuffles, each lane keeps the running
// (distance, index) minimum of its code column per token row, and the minimum over columns / waves is a shuffle + LDS
// reduction with lowest-index ties (torch.argmin returns the first minimum).
#define VQ_TM 64
#define VQ_TN 128
#define VQ_KC 32
#define VQ_P 33
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                         int* __restrict__ idx, int M, int ncodes, int dim) {
  __shared__ float zs[VQ_TM * VQ_P];
  __shared__ float es[VQ_TN * VQ_P];
  __shared__ float z2s[VQ_TM], e2s[VQ_TN];
  __shared__ float rbest[4][VQ_TM];
  __shared__ int ribest[4][VQ_TM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
  const int m0 = blockIdx.x * VQ_TM;
  const int c4 = (tid & 7) * 4, r8 = tid >> 3;            // staging: 8 threads x float4 per 32-dim row chunk, 32 rows per pass
  float best[2][16];
  int bidx[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      best[i][r] = INFINITY;
      bidx[i][r] = 0x7fffffff;
    }
  float z2p[2] = {0.f, 0.f};
  for (int n0 = 0; n0 < ncodes; n0 += VQ_TN) {
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float e2p[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < dim; k0 += VQ_KC) {
      __syncthreads();                                    // previous chunk's fragments consumed
#pragma unroll
      for (int j = 0; j < 2; ++j) {                       // token rows r8, r8 + 32
        const int row = r8 + j * 32, m = m0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && k0 + c4 < dim) v = *reinterpret_cast<const float4*>(z + (long)m * dim + k0 + c4);
        float* d = &zs[row * VQ_P + c4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        if (n0 == 0) z2p[j] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {                       // code rows r8 + 32 j
        const int row = r8 + j * 32, code = n0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (code < ncodes && k0 + c4 < dim) v = *reinterpret_cast<const float4*>(cb + (long)code * dim + k0 + c4);
        float* d = &es[row * VQ_P + c4];
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        e2p[j] += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < VQ_KC; ks += 2) {
        const float b = es[(wave * 32 + l31) * VQ_P + ks + lhi];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float a = zs[(i * 32 + l31) * VQ_P + ks + lhi];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
      }
    }
    // |e|^2 (and |z|^2 on the first tile): the 8 threads of a row hold its partial sums
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = e2p[j];
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      if ((tid & 7) == 0) e2s[r8 + j * 32] = s;
    }
    if (n0 == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float s = z2p[j];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if ((tid & 7) == 0) z2s[r8 + j * 32] = s;
      }
    }
    __syncthreads();
    const int code = n0 + wave * 32 + l31;
    const float e2 = e2s[wave * 32 + l31];
    if (code < ncodes) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          const float dist = (z2s[row] + e2) - 2.f * acc[i][r];     // VQ:43-44 association
          if (dist < best[i][r]) {                                    // codes ascend per lane: strict < keeps the first
            best[i][r] = dist;
            bidx[i][r] = code;
          }
        }
    }
  }
  // minimum over the 32 code columns of the half-wave, then over the 4 waves; ties -> lowest index
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = best[i][r];
      int bi = bidx[i][r];
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(bi, o);
        if (ov < v || (ov == v && oi < bi)) {
          v = ov;
          bi = oi;
        }
      }
      if (l31 == 0) {
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        rbest[wave][row] = v;
        ribest[wave][row] = bi;
      }
    }
  __syncthreads();
  if (tid < VQ_TM && m0 + tid < M) {
    float v = rbest[0][tid];
    int bi = ribest[0][tid];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float ov = rbest[w][tid];
      const int oi = ribest[w][tid];
      if (ov < v || (ov == v && oi < bi)) {
        v = ov;
        bi = oi;
      }
    }
    idx[m0 + tid] = bi;
  }
}

extern "C" int32_t keep_vq_nearest(const float* z, const float* codebook, int32_t* idx, int32_t M, int32_t ncodes,
                                   int32_t dim, void* stream) {
  KEEP_REQUIRE(z && codebook && idx && M > 0 && ncodes > 0 && dim > 0, "keep_vq_nearest: bad args");
  KEEP_REQUIRE(dim % 4 == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)codebook % 16 == 0,
               "keep_vq_nearest: dim %% 4 == 0 and 16-byte aligned z / codebook (float4 staging), got dim=%d", dim);
  hipLaunchKernelGGL(vq_nearest_kernel, dim3(cdiv(M, VQ_TM)), dim3(256), 0, (hipStream_t)stream, z, codebook, idx, M, ncodes, dim);
  KEEP_LAUNCH_CHECK("keep_vq_nearest");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ Kalman blend
__global__ void kalman_update_kernel(const float* __restrict__ zc, const float* __restrict__ zp,
                                     const float* __restrict__ g, float* __restrict__ out, long total, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float gg = g[i / C];
    out[i] = (1.0f - gg) * zc[i] + gg * zp[i];
  }
}

extern "C" int32_t keep_kalman_update(const float* z_code, const float* z_prime, const float* gain, float* out, int32_t N,
                                      int32_t HW, int32_t C, void* stream) {
  KEEP_REQUIRE(z_code && z_prime && gain && out && N > 0 && HW > 0 && C > 0, "keep_kalman_update: bad args");
  const long total = (long)N * HW * C;
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(kalman_update_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z_code, z_prime, gain, out,
                     total, C);
  KEEP_LAUNCH_CHECK("keep_kalman_update");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ flow warp
// grid_sample(bilinear, zeros, align_corners=True): with the reference's normalisation 2v/(W-1)-1 and the
// un-normalisation ((g+1)/2)*(W-1) the sample position is v = x + flow_x up to fp rounding; we follow the same
// fp32 operation sequence so borderline floor() decisions match.
__global__ void flow_warp_kernel(const float* __restrict__ x, const float* __restrict__ flow, float* __restrict__ out,
                                 int N, int H, int W, int C) {
  const long npix = (long)N * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / ((long)H * W));
    const int r = (int)(i - (long)n * H * W);
    const int oy = r / W, ox = r - oy * W;
    const float fx = flow[i * 2 + 0], fy = flow[i * 2 + 1];
    const float wm1 = (float)max(W - 1, 1), hm1 = (float)max(H - 1, 1);
    const float gx = 2.0f * ((float)ox + fx) / wm1 - 1.0f;
    const float gy = 2.0f * ((float)oy + fy) / hm1 - 1.0f;
    const float sx = ((gx + 1.f) / 2.f) * (float)(W - 1);
    const float sy = ((gy + 1.f) / 2.f) * (float)(H - 1);
    const float x0f = floorf(sx), y0f = floorf(sy);
    const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = sx - x0f, ty = sy - y0f;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    // a non-finite flow must not read as "sample outside the image" (zeros): it poisons the pixel, like torch's grid_sample;
    // its int conversions are meaningless (x0 + 1 may wrap), so no tap is read at all
    const bool fin = fabsf(fx) <= 3.0e38f && fabsf(fy) <= 3.0e38f;
    const bool vx0 = fin && x0 >= 0 && x0 < W, vx1 = fin && x0 >= -1 && x0 < W - 1;
    const bool vy0 = fin && y0 >= 0 && y0 < H, vy1 = fin && y0 >= -1 && y0 < H - 1;
    const float* xb = x + (long)n * H * W * C;
    const float poison = fin ? 0.f : __builtin_nanf("");
    for (int c = 0; c < C; ++c) {
      float acc = poison;
      if (vy0 && vx0) acc += xb[((long)y0 * W + x0) * C + c] * w00;
      if (vy0 && vx1) acc += xb[((long)y0 * W + x1) * C + c] * w01;
      if (vy1 && vx0) acc += xb[((long)y1 * W + x0) * C + c] * w10;
      if (vy1 && vx1) acc += xb[((long)y1 * W + x1) * C + c] * w11;
      out[i * C + c] = acc;
    }
  }
}

extern "C" int32_t keep_flow_warp(const float* x, const float* flow, float* out, int32_t N, int32_t H, int32_t W,
                                  int32_t C, void* stream) {
  KEEP_REQUIRE(x && flow && out && N > 0 && H > 0 && W > 0 && C > 0, "keep_flow_warp: bad args");
  const long npix = (long)N * H * W;
  int blocks = cdiv(npix, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(flow_warp_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, flow, out, N, H, W, C);
  KEEP_LAUNCH_CHECK("keep_flow_warp");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ convex upsample
// one thread per (n, y, x, sub-pixel ky,kx): softmax over the 9 taps of mask[..., tap*k*k + ky*k + kx], weighted sum
// of k*flow over the zero-padded 3x3 neighbourhood (unfold order tap = (dy+1)*3 + (dx+1)).
__global__ void convex_upsample_kernel(const float* __restrict__ mask, const float* __restrict__ flow,
                                       float* __restrict__ out, int N, int H, int W, int k) {
  const int kk = k * k;
  const long total = (long)N * H * W * kk;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int sub = (int)(i % kk);
    const long pix = i / kk;
    const int n = (int)(pix / ((long)H * W));
    const int r = (int)(pix - (long)n * H * W);
    const int y = r / W, x = r - y * W;
    const int ky = sub / k, kx = sub - ky * k;
    const float* mp = mask + pix * (9L * kk) + sub;
    float mv[9], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      mv[t] = mp[(long)t * kk];
      mx = fmaxf(mx, mv[t]);
    }
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      mv[t] = expf(mv[t] - mx);
      den += mv[t];
    }
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float* fp = flow + (((long)n * H + yy) * W + xx) * 2;
        const float wgt = mv[t] / den;
        ax += wgt * ((float)k * fp[0]);
        ay += wgt * ((float)k * fp[1]);
      }
    }
    const long oy = (long)y * k + ky, ox = (long)x * k + kx;
    float* op = out + (((long)n * H * k + oy) * ((long)W * k) + ox) * 2;
    op[0] = ax;
    op[1] = ay;
  }
}

extern "C" int32_t keep_convex_upsample(const float* mask, const float* flow, float* out, int32_t N, int32_t H, int32_t W,
                                        int32_t k, void* stream) {
  KEEP_REQUIRE(mask && flow && out && N > 0 && H > 0 && W > 0 && k > 0, "keep_convex_upsample: bad args");
  const long total = (long)N * H * W * k * k;
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(convex_upsample_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, flow, out, N, H, W, k);
  KEEP_LAUNCH_CHECK("keep_convex_upsample");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ layout helpers
// [N,C,HW] -> [N,HW,C] through a 64-pixel LDS tile so both sides are coalesced (C small: 2..3 on this path).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ out, int C,
                                                           int HW, int mode) {
  extern __shared__ float tile[];  // [C][256]
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * 256;
  const int tid = threadIdx.x;
  for (int c = 0; c < C; ++c) {
    const int px = p0 + tid;
    float v = px < HW ? x[((long)n * C + c) * HW + px] : 0.f;
    if (mode == 1) {  // GF:56-57 then GM/utils.py:55-63, same op order
      const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
      const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
      v = (v + 1.f) / 2.f * 255.f;
      v = (v / 255.f - mean) / stdv;
    }
    tile[c * 256 + tid] = v;
  }
  __syncthreads();
  const int npx = min(256, HW - p0);
  for (int i = tid; i < npx * C; i += 256) {
    const int px = i / C, c = i - px * C;
    out[((long)n * HW + p0) * C + i] = tile[c * 256 + px];
  }
}

extern "C" int32_t keep_nchw_to_nhwc(const float* x, float* out, int32_t N, int32_t C, int32_t HW, int32_t mode,
                                     void* stream) {
  KEEP_REQUIRE(x && out && N > 0 && C > 0 && C <= 16 && HW > 0, "keep_nchw_to_nhwc: bad args (C=%d)", C);
  KEEP_REQUIRE(mode == 0 || (mode == 1 && C == 3), "keep_nchw_to_nhwc: mode 1 needs C == 3");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(HW, 256), N), dim3(256), C * 256 * sizeof(float),
                     (hipStream_t)stream, x, out, C, HW, mode);
  KEEP_LAUNCH_CHECK("keep_nchw_to_nhwc");
  return KEEP_OK;
}

// GMFlow's first convolution (GM/backbone.py:69: 7x7, stride 2, pad 3 on the normalised RGB frame) as a 4x4 stride-1 convolution on the
// 2x2 space-to-depth image: out[y][x] reads rows 2y-3 .. 2y+3 = s2d rows y-2 .. y+1 (sub-row dy = (ky + 1) & 1).  One thread per
// s2d pixel: 12 values (dy, dx, c) + 4 zero channels = one 64-byte row, so the layer runs on the 16-channel MFMA kernels
// instead of the element-wise gather of a 147-deep K (engine/weights.py packs the matching [64,4,4,16] weights).
__global__ __launch_bounds__(256) void rgb_s2d_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (i >= (long)H2 * W2) return;
  const int Y = (int)(i / W2), X = (int)(i - (long)Y * W2);
  float v[16];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float mean = c == 0 ? 0.485f : (c == 1 ? 0.456f : 0.406f);
    const float stdv = c == 0 ? 0.229f : (c == 1 ? 0.224f : 0.225f);
    const float* src = x + ((long)n * 3 + c) * H * W + (long)(2 * Y) * W + 2 * X;
    const float2 r0 = *reinterpret_cast<const float2*>(src), r1 = *reinterpret_cast<const float2*>(src + W);
    const float raw[4] = {r0.x, r0.y, r1.x, r1.y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // the op order of nchw_to_nhwc_kernel mode 1 (GF:56-57 then GM/utils.py:55-63)
      float t = (raw[q] + 1.f) / 2.f * 255.f;
      t = (t / 255.f - mean) / stdv;
      v[q * 3 + c] = t;
    }
  }
  v[12] = v[13] = v[14] = v[15] = 0.f;
  float4* dst = reinterpret_cast<float4*>(out + (((long)n * H2 + Y) * W2 + X) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
}

extern "C" int32_t keep_rgb_s2d(const float* x, float* out, int32_t N, int32_t H, int32_t W, void* stream) {
  KEEP_REQUIRE(x && out && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && (uintptr_t)x % 8 == 0 && (uintptr_t)out % 16 == 0,
               "keep_rgb_s2d: bad args (H=%d W=%d must be even, x 8-byte / out 16-byte aligned)", H, W);
  hipLaunchKernelGGL(rgb_s2d_kernel, dim3(cdiv((long)(H / 2) * (W / 2), 256), N), dim3(256), 0, (hipStream_t)stream, x, out, H, W);
  KEEP_LAUNCH_CHECK("keep_rgb_s2d");
  return KEEP_OK;
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int C,
                                                           int HW) {
  extern __shared__ float tile[];  // [256][C | 1]: odd pitch, the per-pixel column reads below are bank-conflict free
  const int n = blockIdx.y;
  const int p0 = blockIdx.x * 256;
  const int tid = threadIdx.x;
  const int npx = min(256, HW - p0);
  const int P = C | 1;
  for (int i = tid; i < npx * C; i += 256) tile[(i / C) * P + (i % C)] = x[((long)n * HW + p0) * C + i];
  __syncthreads();
  if (tid < npx)
    for (int c = 0; c < C; ++c) out[((long)n * C + c) * HW + p0 + tid] = tile[tid * P + c];
}

extern "C" int32_t keep_nhwc_to_nchw(const float* x, float* out, int32_t N, int32_t C, int32_t HW, void* stream) {
  KEEP_REQUIRE(x && out && N > 0 && C > 0 && C <= 32 && HW > 0, "keep_nhwc_to_nchw: bad args (C=%d)", C);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(HW, 256), N), dim3(256), (C | 1) * 256 * sizeof(float),
                     (hipStream_t)stream, x, out, C, HW);
  KEEP_LAUNCH_CHECK("keep_nhwc_to_nchw");
  return KEEP_OK;
}

__global__ void add_bcast_kernel(const float* __restrict__ a, const float* __restrict__ t, float* __restrict__ out,
                                 long total, long tsize, float alpha) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    out[i] = a[i] + alpha * t[i % tsize];
}

extern "C" int32_t keep_add_bcast(const float* a, const float* t, float* out, int64_t total, int64_t tsize, float alpha,
                                  void* stream) {
  KEEP_REQUIRE(a && t && out && total > 0 && tsize > 0, "keep_add_bcast: bad args");
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(add_bcast_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, t, out, (long)total,
                     (long)tsize, alpha);
  KEEP_LAUNCH_CHECK("keep_add_bcast");
  return KEEP_OK;
}

__global__ void concat2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long M,
                               int C1, int C2, int ld) {
  const long total = M * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / ld;
    const int c = (int)(i - m * ld);
    out[i] = c < C1 ? a[m * C1 + c] : (c < C1 + C2 ? b[m * C2 + (c - C1)] : 0.f);      // columns >= C1 + C2: zero padding
  }
}

extern "C" int32_t keep_concat2(const float* a, const float* b, float* out, int64_t M, int32_t C1, int32_t C2, int32_t out_ld,
                                void* stream) {
  KEEP_REQUIRE(a && b && out && M > 0 && C1 > 0 && C2 > 0 && out_ld >= C1 + C2, "keep_concat2: bad args");
  const long total = (long)M * out_ld;
  int blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(concat2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)M, C1, C2, out_ld);
  KEEP_LAUNCH_CHECK("keep_concat2");
  return KEEP_OK;
}

// img_util.py:66-90: clamp(-1,1) -> (x+1)/2 -> *255 -> round half to even -> uint8, RGB -> BGR
__global__ void tensor2img_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = x[i * 3 + c];
      v = fminf(fmaxf(v, -1.f), 1.f);
      v = (v - (-1.f)) / 2.f;
      v = rintf(v * 255.0f);
      out[i * 3 + (2 - c)] = (uint8_t)v;
    }
  }
}

extern "C" int32_t keep_tensor2img(const float* x, uint8_t* out, int64_t npix, void* stream) {
  KEEP_REQUIRE(x && out && npix > 0, "keep_tensor2img: bad args");
  int blocks = cdiv(npix, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(tensor2img_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)npix);
  KEEP_LAUNCH_CHECK("keep_tensor2img");
  return KEEP_OK;
}

// keep_processor.py:258-259: float32(u8 / 255.) (float64 divide, then cast) -> (x - 0.5) / 0.5, BGR -> RGB
__global__ void img2tensor_kernel(const uint8_t* __restrict__ x, float* __restrict__ out, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = (float)((double)x[i * 3 + (2 - c)] / 255.0);
      out[i * 3 + c] = (v - 0.5f) / 0.5f;
    }
  }
}

extern "C" int32_t keep_img2tensor(const uint8_t* x, float* out, int64_t npix, void* stream) {
  KEEP_REQUIRE(x && out && npix > 0, "keep_img2tensor: bad args");
  int blocks = cdiv(npix, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(img2tensor_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)npix);
  KEEP_LAUNCH_CHECK("keep_img2tensor");
  return KEEP_OK;
}

// modules/utils.py:cv2_to_comfy_image (reference utils.py:162-166) on the device: uint8 BGR -> float32 RGB, float32(u8) / 255.0f as ONE
// correctly rounded IEEE division (numpy's `rgb.astype(np.float32) / 255.0`; a multiply by the rounded reciprocal -- what a tensor / scalar
// division compiles to in torch -- differs from it in the last bit for 126 of the 256 values).
__global__ void bgr_u8_to_comfy_kernel(const uint8_t* __restrict__ x, float* __restrict__ out, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[i * 3 + c] = __fdiv_rn((float)x[i * 3 + (2 - c)], 255.0f);
  }
}

extern "C" int32_t keep_bgr_u8_to_comfy(const uint8_t* x, float* out, int64_t npix, void* stream) {
  KEEP_REQUIRE(x && out && npix > 0, "keep_bgr_u8_to_comfy: bad args");
  int blocks = cdiv(npix, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(bgr_u8_to_comfy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)npix);
  KEEP_LAUNCH_CHECK("keep_bgr_u8_to_comfy");
  return KEEP_OK;
}

// modules/utils.py:comfy_image_to_cv2 (reference utils.py:155-160) on the device: float32 RGB -> uint8 BGR, `(x * 255).astype(np.uint8)`:
// ONE float32 multiply (numpy keeps float32 for a float32 array times a Python int), then C's truncating float -> integer conversion as
// numpy performs it on x86-64: through a 32-bit integer (cvttps2dq) whose low byte is kept, so 256.0 wraps to 0 and -1.0 to 255; NaN,
// +-inf and anything beyond the int32 range give the "integer indefinite" value 0x80000000, i.e. byte 0.  (v_cvt_i32_f32 SATURATES
// instead: the range test below is what keeps the two equal outside [0, 1] as well.)  Four pixels (48 B in, 12 B out) per thread.
__global__ __launch_bounds__(256) void comfy_to_bgr_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long npix) {
  auto cvt = [](float f) -> unsigned {
    const float v = __fmul_rn(f, 255.0f);
    return (fabsf(v) < 2147483648.0f) ? ((unsigned)(int)v & 255u) : 0u;      // (NaN fails the comparison)
  };
  const long nq = npix >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (long)gridDim.x * blockDim.x) {
    const float4* p = reinterpret_cast<const float4*>(x + q * 12);
    const float4 a = p[0], b = p[1], c = p[2];           // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
    uint3 o;
    o.x = cvt(a.z) | (cvt(a.y) << 8) | (cvt(a.x) << 16) | (cvt(b.y) << 24);          // B0 G0 R0 B1
    o.y = cvt(b.x) | (cvt(a.w) << 8) | (cvt(c.x) << 16) | (cvt(b.w) << 24);          // G1 R1 B2 G2
    o.z = cvt(b.z) | (cvt(c.w) << 8) | (cvt(c.z) << 16) | (cvt(c.y) << 24);          // R2 B3 G3 R3
    unsigned* d = reinterpret_cast<unsigned*>(out + q * 12);
    d[0] = o.x; d[1] = o.y; d[2] = o.z;
  }
  if (blockIdx.x == 0 && threadIdx.x < (npix & 3)) {      // the last one to three pixels
    const long i = (nq << 2) + threadIdx.x;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[i * 3 + ch] = (uint8_t)cvt(x[i * 3 + (2 - ch)]);
  }
}

extern "C" int32_t keep_comfy_to_bgr_u8(const float* x, uint8_t* out, int64_t npix, void* stream) {
  KEEP_REQUIRE(x && out && npix > 0, "keep_comfy_to_bgr_u8: bad args");
  KEEP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 3) == 0, "keep_comfy_to_bgr_u8: x must be 16-byte and out 4-byte aligned");
  int blocks = cdiv(npix >> 2, 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(comfy_to_bgr_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, (long)npix);
  KEEP_LAUNCH_CHECK("keep_comfy_to_bgr_u8");
  return KEEP_OK;
}

// keep_geglu + amax[n] = max |out| over image n (grid (bx, N): a block stays inside one image, one atomic per block)
__global__ __launch_bounds__(256) void geglu_amax_kernel(const float* __restrict__ x, float* __restrict__ out, long per_img, int F,
                                                         unsigned* __restrict__ amax_bits) {
  const int n = blockIdx.y;
  const float* xi = x + (long)n * per_img * 2;
  float* oi = out + (long)n * per_img;
  unsigned mx = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long)gridDim.x * 256) {
    const long m = i / F;
    const int j = (int)(i - m * F);
    const float h = xi[m * 2 * F + j];
    const float g = xi[m * 2 * F + F + j];
    const float o = h * (0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)));
    oi[i] = o;
    mx = max(mx, __float_as_uint(o) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  __shared__ unsigned wave_m[4];
  if ((threadIdx.x & 63) == 0) wave_m[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = max(max(wave_m[0], wave_m[1]), max(wave_m[2], wave_m[3]));
    if (mx > *reinterpret_cast<volatile unsigned*>(amax_bits + n)) atomicMax(amax_bits + n, mx);
  }
}

extern "C" int32_t keep_geglu_amax(const float* x, float* out, int32_t N, int32_t rows_per_image, int32_t F, float* amax, int32_t zeroed,
                                   void* stream) {
  KEEP_REQUIRE(x && out && amax && N > 0 && rows_per_image > 0 && F > 0, "keep_geglu_amax: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (!zeroed) hipLaunchKernelGGL(absmax_zero_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(amax), N);
  const long per_img = (long)rows_per_image * F;
  const long work = (per_img + 2047) / 2048;
  const long cap = 2048 / N > 1 ? 2048 / N : 1;
  const int bx = (int)(work < 1 ? 1 : (work > cap ? cap : work));
  hipLaunchKernelGGL(geglu_amax_kernel, dim3(bx, N), dim3(256), 0, st, x, out, per_img, F, reinterpret_cast<unsigned*>(amax));
  KEEP_LAUNCH_CHECK("keep_geglu_amax");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ range probe (KEEP_MMA_X3)
// amax[n] = max |x| over image n: the fp16 split of an un-normalised tensor needs its range (keep_conv2d x3_in_amax,
// keep_attention q/k/v_amax).  |x| as raw bits is monotonic in the value, so a plain unsigned atomicMax is exact and
// order-independent (NaN bit patterns compare above inf and surface as NaN in amax -> the consumers propagate them).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, unsigned* __restrict__ amax_bits, long R, int C,
                                                     long ld, long img_stride) {
  const int n = blockIdx.y;
  const float* xi = x + (long)n * img_stride;
  unsigned m = 0u;
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && (img_stride % 4 == 0) && ((uintptr_t)x % 16 == 0);
  if (vec && ld == C) {                // contiguous rows: one flat stream, four independent 16-byte loads per iteration
    const uint4* x4 = reinterpret_cast<const uint4*>(xi);
    const long total = R * (C >> 2), stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < total; i += 4 * stride) {
      const uint4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
      m = max(max(m, a.x & 0x7fffffffu), max(a.y & 0x7fffffffu, max(a.z & 0x7fffffffu, a.w & 0x7fffffffu)));
      m = max(max(m, b.x & 0x7fffffffu), max(b.y & 0x7fffffffu, max(b.z & 0x7fffffffu, b.w & 0x7fffffffu)));
      m = max(max(m, c.x & 0x7fffffffu), max(c.y & 0x7fffffffu, max(c.z & 0x7fffffffu, c.w & 0x7fffffffu)));
      m = max(max(m, d.x & 0x7fffffffu), max(d.y & 0x7fffffffu, max(d.z & 0x7fffffffu, d.w & 0x7fffffffu)));
    }
    for (; i < total; i += stride) {
      const uint4 v = x4[i];
      m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
  } else if (vec) {                    // channel slice of wider rows: (row, column) walked without a division per element
    const int c4n = C >> 2;
    const long total = R * c4n, stride = (long)gridDim.x * 256;
    const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
    long r = i0 / c4n;
    int c = (int)(i0 - r * c4n);
    const long dr = stride / c4n;
    const int dc = (int)(stride - dr * c4n);
    for (long i = i0; i < total; i += stride) {
      const uint4 v = *reinterpret_cast<const uint4*>(xi + r * ld + (c << 2));
      m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
      r += dr;
      c += dc;
      if (c >= c4n) {
        c -= c4n;
        ++r;
      }
    }
  } else {
    const long total = R * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const long r = i / C;
      const int c = (int)(i - r * C);
      m = max(m, __float_as_uint(xi[r * ld + c]) & 0x7fffffffu);
    }
  }
  // one atomic per BLOCK: the N result words share a cache line, and same-line atomics retire at ~12 ns each -- with one per wave
  // a 67 MB probe over 16 images (8192 waves) took 141 us, 100 of them in the atomic queue
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  __shared__ unsigned wave_m[4];
  if ((threadIdx.x & 63) == 0) wave_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wave_m[0], wave_m[1]), max(wave_m[2], wave_m[3]));
    if (m > *reinterpret_cast<volatile unsigned*>(amax_bits + n)) atomicMax(amax_bits + n, m);   // see wave_amax_commit
  }
}

// (a kernel, not hipMemsetAsync: a memset node inside a captured hipGraph raced with the atomics that follow it)
__global__ void absmax_zero_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

extern "C" int32_t keep_absmax(const float* x, float* amax, int32_t N, int64_t R, int32_t C, int64_t ld, int64_t img_stride, int32_t zeroed,
                               void* stream) {
  KEEP_REQUIRE(x && amax && N > 0 && R > 0 && C > 0 && ld >= C, "keep_absmax: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (!zeroed)   // the caller may hand in slots of an arena it zero-fills once per forward pass
    hipLaunchKernelGGL(absmax_zero_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(amax), N);
  KEEP_LAUNCH_CHECK("keep_absmax(zero)");
  // ~8 float4 per thread, at most ~1024 blocks (= atomics) over all images
  const long work = (R * C / 4 + 2047) / 2048;
  const long cap = 1024 / N > 1 ? 1024 / N : 1;
  int bx = (int)(work < 1 ? 1 : (work > cap ? cap : work));
  hipLaunchKernelGGL(absmax_kernel, dim3(bx, N), dim3(256), 0, st, x, reinterpret_cast<unsigned*>(amax), (long)R, C, (long)ld,
                     (long)img_stride);
  KEEP_LAUNCH_CHECK("keep_absmax");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ K0: need_upscale
// keep_arch.py:1020-1023: F.interpolate(x, scale_factor=4, mode='bilinear') (align_corners=False) on the planar clip
// frames -- torch's area-pixel source index max(0, (dst + 0.5)/scale - 0.5), neighbours clamped to the last row / column,
// weights combined in torch's own order h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).  HBM-bound: one read + scale^2 writes.
__global__ __launch_bounds__(256) void bilinear_upscale_kernel(const float* __restrict__ x, float* __restrict__ out, int planes,
                                                               int H, int W, int scale) {
  const int Ho = H * scale, Wo = W * scale;
  const long total = (long)planes * Ho * Wo;
  const float rs = 1.0f / (float)scale;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const long t = i / Wo;
    const int oy = (int)(t % Ho);
    const long pl = t / Ho;
    const float sy = fmaxf(0.f, ((float)oy + 0.5f) * rs - 0.5f), sx = fmaxf(0.f, ((float)ox + 0.5f) * rs - 0.5f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float h1 = sy - (float)y0, w1 = sx - (float)x0, h0 = 1.f - h1, w0 = 1.f - w1;
    const float* src = x + pl * (long)H * W;
    out[i] = h0 * (w0 * src[(long)y0 * W + x0] + w1 * src[(long)y0 * W + x1]) +
             h1 * (w0 * src[(long)y1 * W + x0] + w1 * src[(long)y1 * W + x1]);
  }
}

extern "C" int32_t keep_bilinear_upscale(const float* x, float* out, int32_t planes, int32_t H, int32_t W, int32_t scale,
                                         void* stream) {
  KEEP_REQUIRE(x && out && planes > 0 && H > 0 && W > 0 && scale >= 1 && scale <= 8, "keep_bilinear_upscale: bad args");
  const long total = (long)planes * H * scale * W * scale;
  int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
  hipLaunchKernelGGL(bilinear_upscale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, out, planes, H, W, scale);
  KEEP_LAUNCH_CHECK("keep_bilinear_upscale");
  return KEEP_OK;
}
