// keep_conv2d: implicit-GEMM convolution / linear layer on the CDNA4 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD -- the precision policy that holds the <=1e-3 parity bound).
//
// GEMM view:  M = N*Ho*Wo output pixels (rows), Ncol = Cout, K = KH*KW*Cin (looped tap-major, Cin in chunks of 16).
// Data layout: activations NHWC, weights [Cout][KH*KW][Cin]: both operands are K-contiguous in HBM.
// Block = 256 threads = 4 waves (one per SIMD); wave tile = (TM*32) x (TN*32) built from 32x32x2 MFMAs.
// LDS tiles are K-major ([k][row], pitch = rows+2): a 32x32x2 operand fragment is one conflict-free ds_read_b32
// per lane (lanes 0-31 -> 32 consecutive rows, lanes 32-63 -> next k), and the staging writes of one wave hit 32
// distinct banks per half-wave (pitch*chunk == 32/threads_per_row mod 32).  Global->LDS is register-staged because
// the GroupNorm/InstanceNorm affine + swish/ReLU prologue and the zero padding are applied on the way in; tile k+1
// is fetched into registers while tile k is on the matrix cores (one barrier per K step).
// Small-spatial layers (16x16 .. 64x64 maps) under-fill 256 CUs with output tiles alone -> deterministic split-K
// over the (tap, Cin-chunk) steps into a caller workspace + a reduce/epilogue kernel.
#include "keep_common.h"

#define BK 16

struct ConvP {
  const float* in;
  const float* w;
  const float* bias;
  float* out;
  const float* pro_scale;
  const float* pro_shift;
  const float* res;
  const float* aux;
  float* ws;
  int N, H, W, Cin, Cout, KH, KW, stride, pad_t, pad_l, Ho, Wo;
  int in_ld, out_ld, res_ld;
  int upsample, pro_act, epi_act;
  float aux_w;
  int split_k;
  int M;        // N*Ho*Wo
  int cchunks;  // ceil(Cin/BK)
  int nsteps;   // KH*KW*cchunks
  int vec_ok;   // Cin%4==0 && in_ld%4==0 -> float4 loads
};

__device__ __forceinline__ float epilogue_one(const ConvP& p, float v, long m, int co) {
  if (p.bias) v += p.bias[co];
  v = act_apply(v, p.epi_act);
  if (p.res) {
    float r = p.res[m * p.res_ld + co];
    if (p.aux) {
      float a = p.aux[m * (long)p.Cout + co];
      v = r + p.aux_w * (r * a + v);
    } else {
      v = v + r;
    }
  }
  return v;
}

template <int WGM, int WGN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvP p) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr int LDA = BM + 2;
  constexpr int LDB = BN + 2;
  constexpr int A_TPR = 256 / BM;  // threads per A row (pixel)
  constexpr int A_CPT = BK / A_TPR;  // channels per thread
  constexpr int B_TPR = 256 / BN;
  constexpr int B_CPT = BK / B_TPR;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(A_TPR >= 1 && B_TPR >= 1 && A_CPT >= 1 && B_CPT >= 1, "tile config");

  __shared__ float As[2][BK * LDA];
  __shared__ float Bs[2][BK * LDB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN;
  const int wn = wave % WGN;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int z = blockIdx.z;

  // split-K range of (tap, cin-chunk) steps
  const int per = (p.nsteps + p.split_k - 1) / p.split_k;
  const int s_begin = z * per;
  const int s_end = min(p.nsteps, s_begin + per);

  // ---- A staging role: one pixel row, A_CPT channels
  const int a_row = tid / A_TPR;
  const int a_kq = (tid % A_TPR) * A_CPT;
  const long a_m = m0 + a_row;
  const bool a_mvalid = a_m < p.M;
  int a_n = 0, a_oy = 0, a_ox = 0;
  if (a_mvalid) {
    int hw = p.Ho * p.Wo;
    a_n = (int)(a_m / hw);
    int r = (int)(a_m - (long)a_n * hw);
    a_oy = r / p.Wo;
    a_ox = r - a_oy * p.Wo;
  }
  const int Hv = p.upsample ? 2 * p.H : p.H;  // virtual (post-upsample) input extent
  const int Wv = p.upsample ? 2 * p.W : p.W;
  const float* a_scale = p.pro_scale ? p.pro_scale + (long)a_n * p.Cin : nullptr;
  const float* a_shift = p.pro_shift ? p.pro_shift + (long)a_n * p.Cin : nullptr;

  // ---- B staging role: one output channel row, B_CPT k's
  const int b_row = tid / B_TPR;
  const int b_kq = (tid % B_TPR) * B_CPT;
  const int b_co = n0 + b_row;
  const bool b_valid = b_co < p.Cout;
  const long w_rowoff = (long)b_co * p.KH * p.KW * p.Cin;

  float a_reg[A_CPT];
  float b_reg[B_CPT];

  auto fetch = [&](int s) {
    const int tap = s / p.cchunks;
    const int c0 = (s - tap * p.cchunks) * BK;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
    // ---- A
    const int iy = a_oy * p.stride - p.pad_t + kh;
    const int ix = a_ox * p.stride - p.pad_l + kw;
    const bool pix_ok = a_mvalid && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
    const int sy = p.upsample ? (iy >> 1) : iy;
    const int sx = p.upsample ? (ix >> 1) : ix;
    const int ca = c0 + a_kq;
    if (pix_ok) {
      const float* src = p.in + (((long)a_n * p.H + sy) * p.W + sx) * p.in_ld + ca;
      if (p.vec_ok && (A_CPT % 4 == 0) && ca + A_CPT <= p.Cin) {
#pragma unroll
        for (int j = 0; j < A_CPT; j += 4) {
          float4 v = *reinterpret_cast<const float4*>(src + j);
          a_reg[j] = v.x;
          if (j + 1 < A_CPT) a_reg[j + 1] = v.y;
          if (j + 2 < A_CPT) a_reg[j + 2] = v.z;
          if (j + 3 < A_CPT) a_reg[j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < A_CPT; ++j) a_reg[j] = (ca + j < p.Cin) ? src[j] : 0.f;
      }
      if (a_scale) {
#pragma unroll
        for (int j = 0; j < A_CPT; ++j) {
          if (ca + j < p.Cin) {
            float v = a_reg[j] * a_scale[ca + j] + a_shift[ca + j];
            a_reg[j] = pro_apply(v, p.pro_act);
          }
        }
      } else if (p.pro_act != KEEP_PRO_NONE) {
#pragma unroll
        for (int j = 0; j < A_CPT; ++j)
          if (ca + j < p.Cin) a_reg[j] = pro_apply(a_reg[j], p.pro_act);
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_CPT; ++j) a_reg[j] = 0.f;
    }
    // ---- B
    const int cb = c0 + b_kq;
    if (b_valid) {
      const float* src = p.w + w_rowoff + (long)tap * p.Cin + cb;
      if ((p.Cin % 4 == 0) && (B_CPT % 4 == 0) && cb + B_CPT <= p.Cin) {
#pragma unroll
        for (int j = 0; j < B_CPT; j += 4) {
          float4 v = *reinterpret_cast<const float4*>(src + j);
          b_reg[j] = v.x;
          if (j + 1 < B_CPT) b_reg[j + 1] = v.y;
          if (j + 2 < B_CPT) b_reg[j + 2] = v.z;
          if (j + 3 < B_CPT) b_reg[j + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < B_CPT; ++j) b_reg[j] = (cb + j < p.Cin) ? src[j] : 0.f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < B_CPT; ++j) b_reg[j] = 0.f;
    }
  };

  auto stage = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_CPT; ++j) As[buf][(a_kq + j) * LDA + a_row] = a_reg[j];
#pragma unroll
    for (int j = 0; j < B_CPT; ++j) Bs[buf][(b_kq + j) * LDB + b_row] = b_reg[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31;
  const int lhi = lane >> 5;
  const int a_frag0 = wm * TM * 32 + l31;
  const int b_frag0 = wn * TN * 32 + l31;

  if (s_begin < s_end) {
    fetch(s_begin);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
      const bool more = (s + 1 < s_end);
      if (more) fetch(s + 1);
      const float* Ab = As[buf];
      const float* Bb = Bs[buf];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = Ab[(kk * 2 + lhi) * LDA + a_frag0 + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = Bb[(kk * 2 + lhi) * LDB + b_frag0 + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (more) stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int co = n0 + wn * TN * 32 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const long m = m0 + wm * TM * 32 + i * 32 + row;
        if (m < p.M && co < p.Cout) {
          float v = acc[i][j][r];
          if (p.split_k > 1) {
            p.ws[((long)z * p.M + m) * p.Cout + co] = v;
          } else {
            p.out[m * p.out_ld + co] = epilogue_one(p, v, m, co);
          }
        }
      }
    }
  }
}

__global__ void conv_splitk_reduce_kernel(ConvP p) {
  const long total = (long)p.M * p.Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / p.Cout;
    const int co = (int)(i - m * p.Cout);
    float v = 0.f;
    for (int z = 0; z < p.split_k; ++z) v += p.ws[(long)z * total + i];
    p.out[m * p.out_ld + co] = epilogue_one(p, v, m, co);
  }
}

extern "C" int32_t keep_conv2d(const keep_conv2d_args* a, void* stream) {
  KEEP_REQUIRE(a != nullptr, "keep_conv2d: null args");
  if (a->dtype != KEEP_F32) {
    keep_set_error("keep_conv2d: dtype %d not supported (fp32 only in this build)", a->dtype);
    return KEEP_EUNSUP;
  }
  KEEP_REQUIRE(a->in && a->weight && a->out, "keep_conv2d: null tensor pointer");
  KEEP_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 && a->KH > 0 && a->KW > 0 &&
                   a->stride > 0 && a->Ho > 0 && a->Wo > 0,
               "keep_conv2d: non-positive dimension");
  KEEP_REQUIRE(a->in_ld >= a->Cin && a->out_ld >= a->Cout, "keep_conv2d: ld smaller than channel count");
  KEEP_REQUIRE((a->pro_scale == nullptr) == (a->pro_shift == nullptr), "keep_conv2d: pro_scale/pro_shift must pair");
  KEEP_REQUIRE(!a->aux || a->residual, "keep_conv2d: aux epilogue requires residual");
  KEEP_REQUIRE(!a->residual || a->res_ld >= a->Cout, "keep_conv2d: res_ld smaller than Cout");
  KEEP_REQUIRE(a->split_k >= 1, "keep_conv2d: split_k must be >= 1");
  KEEP_REQUIRE(a->split_k == 1 || a->workspace, "keep_conv2d: split_k>1 requires a workspace");
  {
    const int Hv = a->upsample ? 2 * a->H : a->H, Wv = a->upsample ? 2 * a->W : a->W;
    // the last tap of the last output must not start beyond one row/col of padding logic: (Ho-1)*s - pt < Hv
    KEEP_REQUIRE((long)(a->Ho - 1) * a->stride - a->pad_t < Hv && (long)(a->Wo - 1) * a->stride - a->pad_l < Wv,
                 "keep_conv2d: output extent %dx%d inconsistent with input %dx%d", a->Ho, a->Wo, Hv, Wv);
  }
  ConvP p;
  p.in = (const float*)a->in;
  p.w = a->weight;
  p.bias = a->bias;
  p.out = (float*)a->out;
  p.pro_scale = a->pro_scale;
  p.pro_shift = a->pro_shift;
  p.res = (const float*)a->residual;
  p.aux = (const float*)a->aux;
  p.ws = a->workspace;
  p.N = a->N; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.KH = a->KH; p.KW = a->KW;
  p.stride = a->stride; p.pad_t = a->pad_t; p.pad_l = a->pad_l; p.Ho = a->Ho; p.Wo = a->Wo;
  p.in_ld = a->in_ld; p.out_ld = a->out_ld; p.res_ld = a->res_ld;
  p.upsample = a->upsample; p.pro_act = a->pro_act; p.epi_act = a->epi_act; p.aux_w = a->aux_w;
  p.split_k = a->split_k;
  const long M = (long)a->N * a->Ho * a->Wo;
  KEEP_REQUIRE(M < (1L << 31), "keep_conv2d: M too large");
  p.M = (int)M;
  p.cchunks = (a->Cin + BK - 1) / BK;
  p.nsteps = a->KH * a->KW * p.cchunks;
  if (p.split_k > p.nsteps) p.split_k = p.nsteps;
  p.vec_ok = (a->Cin % 4 == 0 && a->in_ld % 4 == 0 && ((uintptr_t)a->in % 16 == 0)) ? 1 : 0;
  KEEP_REQUIRE((uintptr_t)a->weight % 16 == 0, "keep_conv2d: weight pointer must be 16-byte aligned");

  hipStream_t st = (hipStream_t)stream;
  dim3 block(256);
  if (a->Cout <= 32) {
    dim3 grid(cdiv(M, 128), cdiv(a->Cout, 32), p.split_k);
    hipLaunchKernelGGL((conv_f32_kernel<4, 1, 1, 1>), grid, block, 0, st, p);
  } else if (a->Cout <= 64 || M <= 4096) {
    dim3 grid(cdiv(M, 64), cdiv(a->Cout, 64), p.split_k);
    hipLaunchKernelGGL((conv_f32_kernel<2, 2, 1, 1>), grid, block, 0, st, p);
  } else {
    dim3 grid(cdiv(M, 128), cdiv(a->Cout, 128), p.split_k);
    hipLaunchKernelGGL((conv_f32_kernel<2, 2, 2, 2>), grid, block, 0, st, p);
  }
  KEEP_LAUNCH_CHECK("keep_conv2d");
  if (p.split_k > 1) {
    const long total = M * a->Cout;
    int blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    KEEP_LAUNCH_CHECK("keep_conv2d(split-K reduce)");
  }
  return KEEP_OK;
}
