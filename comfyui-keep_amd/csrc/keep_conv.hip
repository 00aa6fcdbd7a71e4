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
#include <stdlib.h>

#include "keep_conv_common.h"

// PLAIN (host-checked): no prologue, no upsample, Cin % 16 == 0 and 16-byte aligned rows: the K loop carries none of
// those uniform branches (token GEMMs, 1x1 convs, plain strided convs).
template <int WGM, int WGN, int TM, int TN, bool PLAIN = false>
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

  constexpr int MAIN_F = 2 * BK * (LDA + LDB);
  constexpr int EPI_F = 4 * (TM * 32) * (TN * 32 + 4);
  __shared__ __attribute__((aligned(16))) float smem_f[MAIN_F > EPI_F ? MAIN_F : EPI_F];
  float(*As)[BK * LDA] = reinterpret_cast<float(*)[BK * LDA]>(smem_f);
  float(*Bs)[BK * LDB] = reinterpret_cast<float(*)[BK * LDB]>(smem_f + 2 * BK * LDA);

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

  // Raw operands of the NEXT K step live in registers while the current step is on the matrix cores; the
  // normalisation affine + activation is applied only when they are written to LDS (stage), so the global loads
  // stay in flight across the whole MFMA loop instead of being waited for right after issue.
  float a_reg[A_CPT];
  float a_sc[A_CPT];
  float a_sh[A_CPT];
  float b_reg[B_CPT];
  bool a_ok = false;
  int a_ca = 0;

  auto fetch = [&](int s) {
    if (!PLAIN && p.flatk_f32) {
      // K = (kh, kw, c) flattened: the RGB first convolutions (Cin = 3) take ceil(9*3/16) = 2 K steps instead of 9 (7x7:
      // 10 instead of 49) that are 13/16 padding.  Host guarantees no prologue.
      const int ktot = p.KH * p.KW * p.Cin;
      const int ka = s * BK + a_kq;
      a_ok = a_mvalid;
      a_ca = 0;
#pragma unroll
      for (int j = 0; j < A_CPT; ++j) {
        const int kk = ka + j;
        float v = 0.f;
        if (a_mvalid && kk < ktot) {
          const int tp = kk / p.Cin, c = kk - tp * p.Cin;
          const int kh2 = tp / p.KW, kw2 = tp - kh2 * p.KW;
          int iy = a_oy * p.stride - p.pad_t + kh2;
          int ix = a_ox * p.stride - p.pad_l + kw2;
          KEEP_REFLECT(iy, ix, Hv, Wv)
          if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
            const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
            v = p.in[(((long)a_n * p.H + sy) * p.W + sx) * p.in_ld + c];
          }
        }
        a_reg[j] = v;
      }
      const int kb = s * BK + b_kq;
#pragma unroll
      for (int j = 0; j < B_CPT; ++j) b_reg[j] = (b_valid && kb + j < ktot) ? p.w[w_rowoff + kb + j] : 0.f;
      return;
    }
    const int tap = s / p.cchunks;
    const int c0 = (s - tap * p.cchunks) * BK;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
    // ---- A
    int iy = a_oy * p.stride - p.pad_t + kh;
    int ix = a_ox * p.stride - p.pad_l + kw;
    KEEP_REFLECT(iy, ix, Hv, Wv)
    a_ok = a_mvalid && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv;
    const int sy = (!PLAIN && p.upsample) ? (iy >> 1) : iy;
    const int sx = (!PLAIN && p.upsample) ? (ix >> 1) : ix;
    const int ca = c0 + a_kq;
    a_ca = ca;
    if (a_ok) {
      const float* src = p.in + (((long)a_n * p.H + sy) * p.W + sx) * p.in_ld + ca;
      if ((PLAIN || p.vec_ok) && (A_CPT % 4 == 0) && (PLAIN || ca + A_CPT <= p.Cin)) {
#pragma unroll
        for (int j = 0; j < A_CPT; j += 4) {
          float4 v = *reinterpret_cast<const float4*>(src + j);
          a_reg[j] = v.x;
          if (j + 1 < A_CPT) a_reg[j + 1] = v.y;
          if (j + 2 < A_CPT) a_reg[j + 2] = v.z;
          if (j + 3 < A_CPT) a_reg[j + 3] = v.w;
        }
        if (!PLAIN && a_scale) {
#pragma unroll
          for (int j = 0; j < A_CPT; j += 4) {
            float4 sc = *reinterpret_cast<const float4*>(a_scale + ca + j);
            float4 sh = *reinterpret_cast<const float4*>(a_shift + ca + j);
            a_sc[j] = sc.x; a_sh[j] = sh.x;
            if (j + 1 < A_CPT) { a_sc[j + 1] = sc.y; a_sh[j + 1] = sh.y; }
            if (j + 2 < A_CPT) { a_sc[j + 2] = sc.z; a_sh[j + 2] = sh.z; }
            if (j + 3 < A_CPT) { a_sc[j + 3] = sc.w; a_sh[j + 3] = sh.w; }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < A_CPT; ++j) {
          const bool cok = ca + j < p.Cin;
          a_reg[j] = cok ? src[j] : 0.f;
          if (a_scale) {
            a_sc[j] = cok ? a_scale[ca + j] : 0.f;
            a_sh[j] = cok ? a_shift[ca + j] : 0.f;
          }
        }
      }
    }
    // ---- B
    const int cb = c0 + b_kq;
    if (b_valid) {
      const float* src = p.w + w_rowoff + (long)tap * p.Cin + cb;
      if ((PLAIN || p.Cin % 4 == 0) && (B_CPT % 4 == 0) && (PLAIN || cb + B_CPT <= p.Cin)) {
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
    for (int j = 0; j < A_CPT; ++j) {
      float v = 0.f;
      if (a_ok && (PLAIN || p.flatk_f32 || a_ca + j < p.Cin)) {
        v = a_reg[j];
        if (!PLAIN && !p.flatk_f32) {
          if (a_scale) v = v * a_sc[j] + a_sh[j];
          v = pro_apply(v, p.pro_act);
        }
      }
      As[buf][(a_kq + j) * LDA + a_row] = v;
    }
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
  if (p.vec_epi) {
    staged_epilogue<WGM, WGN, TM, TN>(p, acc, smem_f, m0, n0, wm, wn, lane, wave, z);
    return;
  }
  float cs[TN], css[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { cs[j] = 0.f; css[j] = 0.f; }
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
            v = epilogue_one(p, v, m, co);
            p.out[m * p.out_ld + co] = v;
            cs[j] += v;
            css[j] += v * v;
          }
        }
      }
    }
  }
  if (p.stats) {   // host guarantees split_k == 1 and H*W %% BM == 0 (a tile never straddles two images)
    __shared__ float red[4 * TM * 0 + WGM * WGN * TN * 32][2];
    const int hw_o = p.Ho * p.Wo;
    emit_tile_stats<WGM, WGN, TN>(p, red, cs, css, wm, wn, lane, (int)(m0 / hw_o), (int)((m0 % hw_o) / (WGM * TM * 32)), n0);
  }
}

// ------------------------------------------------------------------------------------------------ bf16 MFMA
// Same implicit GEMM, operands rounded to bf16 when staged (v_mfma_f32_32x32x16_bf16, fp32 accumulate, 16x the f32
// MFMA rate).  Activations stay fp32 in HBM: the normalisation affine + activation run in fp32 on the way in and
// only the MFMA operand is rounded (RNE, v_cvt_pk_bf16_f32); weights come from a bf16 copy.  K step = 64 channels of
// one tap.  LDS tiles are row-major [row][64 bf16] with a 144-byte pitch (9 x 16-B slots, odd): the 32x32x16
// fragment (8 consecutive k per lane) is one ds_read_b128, conflict-free within each 16-lane service group, and the
// staging ds_write_b128 of 8 lanes covers 8 distinct slots.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define BK16 64     // default K step; small-M / deep-K layers use BKT = 256 (see conv_bf16_kernel)

__device__ __forceinline__ float pro_apply_fast(float v, int act) {
  if (act == KEEP_PRO_SWISH) return v * __frcp_rn(1.0f + __expf(-v));
  if (act == KEEP_PRO_RELU) return relu_keep_nan(v);
  return v;
}

// BKT = K step (channels of one tap per barrier pair).  64 with two LDS buffers is the default; layers with few
// output tiles and a deep K (16x16 / 32x32 maps, token GEMMs) are latency-bound -- one global round trip per step with
// only a handful of MFMAs to hide it -- so they run BKT = 256 with a single LDS buffer: 4x the bytes in flight per
// round trip and 4x fewer barriers for the same registers a 4-deep prefetch ring would need.
// PLAIN (host-checked): no flattened K, no prologue, no upsample, Cin % 8 == 0 and 16-byte aligned rows -- the fetch /
// stage code of every K step then carries none of those uniform branches (token GEMMs, 1x1 convs, plain strided convs).
template <int WGM, int WGN, int TM, int TN, int BKT, int NS, bool PLAIN = false>
__global__ __launch_bounds__(256) void conv_bf16_kernel(ConvP p) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr int PITCH16 = BKT + 8;          // bf16 elements per LDS row: (BKT+8)*2 B is an odd number of 16-B slots
  constexpr int GPR = BKT / 8;              // 8-channel groups per row
  constexpr int RPI = 256 / GPR;            // rows covered per staging iteration
  constexpr int NBUF = (BKT == 64) ? 2 : 1;
  constexpr int A_IT = BM / RPI;            // 16-byte (8 x bf16) pieces per thread per K step
  constexpr int B_IT = BN / RPI;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(A_IT >= 1 && B_IT >= 1, "tile config");

  constexpr int MAIN_B = NBUF * (BM + BN) * PITCH16 * 2;           // bytes
  constexpr int EPI_B = 4 * (TM * 32) * (TN * 32 + 4) * 4;         // bytes
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[MAIN_B > EPI_B ? MAIN_B : EPI_B];
  __bf16(*As)[BM * PITCH16] = reinterpret_cast<__bf16(*)[BM * PITCH16]>(smem_b);
  __bf16(*Bs)[BN * PITCH16] = reinterpret_cast<__bf16(*)[BN * PITCH16]>(smem_b + NBUF * BM * PITCH16 * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN;
  const int wn = wave % WGN;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int z = blockIdx.z;
  const int cchunks = (p.Cin + BKT - 1) / BKT;
  const int nsteps = p.flatk ? (p.KH * p.KW * p.Cin + BKT - 1) / BKT : p.KH * p.KW * cchunks;
  const int per = (nsteps + p.split_k - 1) / p.split_k;
  const int s_begin = z * per;
  const int s_end = min(nsteps, s_begin + per);

  // staging roles: piece = tid + it*256 -> row = piece / GPR, 8-channel group = piece % GPR (= tid % GPR, constant)
  const int grp = tid % GPR;
  const int row0 = tid / GPR;  // + it*RPI
  const int Hv = p.upsample ? 2 * p.H : p.H;
  const int Wv = p.upsample ? 2 * p.W : p.W;
  const int hw = p.Ho * p.Wo;
  int a_n[A_IT], a_oy[A_IT], a_ox[A_IT];
  bool a_mv[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const long m = m0 + row0 + it * RPI;
    a_mv[it] = m < p.M;
    a_n[it] = 0; a_oy[it] = 0; a_ox[it] = 0;
    if (a_mv[it]) {
      a_n[it] = (int)(m / hw);
      const int r = (int)(m - (long)a_n[it] * hw);
      a_oy[it] = r / p.Wo;
      a_ox[it] = r - a_oy[it] * p.Wo;
    }
  }
  const long wrow_stride = (long)p.KH * p.KW * p.Cin;

  // NS register slots of pending K steps.  NS = 3 is a ring (loads of step t+2 issued while step t is on the matrix
  // cores, waited for when step t+1 is staged).  MEASURED (c512@16x16, B=4): the ring is 2x SLOWER than NS = 1 -- these
  // layers are bound by operand bytes through the CU's L1 (64x64 tiles: 21 FLOP per L2 byte at ~12 B/clk/CU), not by
  // load latency, and the ring's registers halve the resident blocks.  All dispatches use NS = 1; the fix for the
  // small maps is operand reuse (the halo kernel), not deeper prefetch.
  float a_raw[NS][A_IT][8];
  bool a_ok[NS][A_IT];
  uint4 b_raw[NS][B_IT];
  int a_c[NS];   // first channel of this thread's group in the pending step
  // all rows of the tile in one image -> this thread's 8 scale/shift values are the same for all its pieces and
  // are prefetched with the operands (the common case: H*W is a multiple of the tile height)
  const long m_last = (m0 + BM - 1 < p.M) ? (m0 + BM - 1) : (long)p.M - 1;
  const bool uni_n = p.pro_scale && ((m0 / hw) == (m_last / hw));
  const long uni_off = (m0 / hw) * (long)p.Cin;
  float u_sc[NS][8], u_sh[NS][8];

  auto fetch = [&](auto slot_c, int s) {
    constexpr int SL = decltype(slot_c)::value;
    const int tap = s / cchunks;
    const int c0 = (s - tap * cchunks) * BKT;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
    const int ca = c0 + grp * 8;
    a_c[SL] = ca;
    if (!PLAIN && p.flatk) {
      // K = (kh, kw, c) flattened (Cin = 3: 27 or 147 real k's instead of 9 / 49 nearly empty 64-wide chunks)
      const int kbase = s * BKT + grp * 8;
      const int ktot = p.KH * p.KW * p.Cin;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        a_ok[SL][it] = a_mv[it] && kbase < ktot;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = kbase + j;
          float v = 0.f;
          if (a_ok[SL][it] && k < ktot) {
            const int tp = k / p.Cin, c = k - tp * p.Cin;
            const int kh2 = tp / p.KW, kw2 = tp - kh2 * p.KW;
            const int iy = a_oy[it] * p.stride - p.pad_t + kh2;
            const int ix = a_ox[it] * p.stride - p.pad_l + kw2;
            if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
              const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
              v = p.in[(((long)a_n[it] * p.H + sy) * p.W + sx) * p.in_ld + c];
            }
          }
          a_raw[SL][it][j] = v;
        }
      }
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int co = n0 + row0 + it * RPI;
        unsigned short t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          t[j] = (co < p.Cout && kbase + j < ktot) ? p.wb[(long)co * ktot + kbase + j] : (unsigned short)0;
        b_raw[SL][it] = make_uint4(t[0] | ((unsigned)t[1] << 16), t[2] | ((unsigned)t[3] << 16),
                               t[4] | ((unsigned)t[5] << 16), t[6] | ((unsigned)t[7] << 16));
      }
      a_c[SL] = 0;     // stage(): every element already validated, no per-channel mask / affine (host forbids a prologue)
      return;
    }
    if (!PLAIN && uni_n) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool cok = ca + j < p.Cin;
        u_sc[SL][j] = cok ? p.pro_scale[uni_off + ca + j] : 0.f;
        u_sh[SL][j] = cok ? p.pro_shift[uni_off + ca + j] : 0.f;
      }
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = a_oy[it] * p.stride - p.pad_t + kh;
      const int ix = a_ox[it] * p.stride - p.pad_l + kw;
      a_ok[SL][it] = a_mv[it] && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv && ca < p.Cin;
      if (a_ok[SL][it]) {
        const int sy = (!PLAIN && p.upsample) ? (iy >> 1) : iy;
        const int sx = (!PLAIN && p.upsample) ? (ix >> 1) : ix;
        const float* src = p.in + (((long)a_n[it] * p.H + sy) * p.W + sx) * p.in_ld + ca;
        if (PLAIN || (p.vec_ok && ca + 8 <= p.Cin)) {
          const float4 v0 = *reinterpret_cast<const float4*>(src);
          const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
          a_raw[SL][it][0] = v0.x; a_raw[SL][it][1] = v0.y; a_raw[SL][it][2] = v0.z; a_raw[SL][it][3] = v0.w;
          a_raw[SL][it][4] = v1.x; a_raw[SL][it][5] = v1.y; a_raw[SL][it][6] = v1.z; a_raw[SL][it][7] = v1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) a_raw[SL][it][j] = (ca + j < p.Cin) ? src[j] : 0.f;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int co = n0 + row0 + it * RPI;
      b_raw[SL][it] = make_uint4(0u, 0u, 0u, 0u);
      if (co < p.Cout && ca < p.Cin) {
        const unsigned short* src = p.wb + (long)co * wrow_stride + (long)tap * p.Cin + ca;
        if (PLAIN || (p.Cin & 7) == 0) {
          b_raw[SL][it] = *reinterpret_cast<const uint4*>(src);
        } else {
          unsigned short t[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) t[j] = (ca + j < p.Cin) ? src[j] : (unsigned short)0;
          b_raw[SL][it] = make_uint4(t[0] | ((unsigned)t[1] << 16), t[2] | ((unsigned)t[3] << 16),
                                 t[4] | ((unsigned)t[5] << 16), t[6] | ((unsigned)t[7] << 16));
        }
      }
    }
  };

  auto stage = [&](auto slot_c, int buf) {
    constexpr int SL = decltype(slot_c)::value;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      bf16x8 h;
      if (a_ok[SL][it]) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a_raw[SL][it][j];
        if (PLAIN) {
        } else if (uni_n) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] * u_sc[SL][j] + u_sh[SL][j];
        } else if (p.pro_scale) {
          const float* sc = p.pro_scale + (long)a_n[it] * p.Cin + a_c[SL];
          const float* sh = p.pro_shift + (long)a_n[it] * p.Cin + a_c[SL];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (a_c[SL] + j < p.Cin) v[j] = v[j] * sc[j] + sh[j];
        }
        if (!PLAIN && p.pro_act != KEEP_PRO_NONE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = pro_apply_fast(v[j], p.pro_act);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (__bf16)((PLAIN || p.flatk || a_c[SL] + j < p.Cin) ? v[j] : 0.f);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (__bf16)0.f;
      }
      *reinterpret_cast<bf16x8*>(&As[buf][(row0 + it * RPI) * PITCH16 + grp * 8]) = h;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      *reinterpret_cast<uint4*>(&Bs[buf][(row0 + it * RPI) * PITCH16 + grp * 8]) = b_raw[SL][it];
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
  const int a_f0 = (wm * TM * 32 + l31) * PITCH16 + lhi * 8;
  const int b_f0 = (wn * TN * 32 + l31) * PITCH16 + lhi * 8;

  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, (NS > 1 ? 1 : 0)>;
  using IC2 = std::integral_constant<int, (NS > 2 ? 2 : 0)>;
  auto mma_step = [&](int buf) {
    const __bf16* Ab = As[buf];
    const __bf16* Bb = Bs[buf];
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ++ks) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + a_f0 + i * 32 * PITCH16 + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + b_f0 + j * 32 * PITCH16 + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  if (s_begin < s_end) {
    if (NS == 3) {
      // ---- 3-slot ring, two LDS buffers
      fetch(IC0{}, s_begin);
      if (s_begin + 1 < s_end) fetch(IC1{}, s_begin + 1);
      stage(IC0{}, 0);
      __syncthreads();
      int buf = 0;
      for (int s = s_begin; s < s_end; s += 3) {
        if (s + 2 < s_end) fetch(IC2{}, s + 2);
        mma_step(buf);
        if (s + 1 < s_end) stage(IC1{}, buf ^ 1);
        __syncthreads();
        buf ^= 1;
        if (s + 1 >= s_end) break;
        if (s + 3 < s_end) fetch(IC0{}, s + 3);
        mma_step(buf);
        if (s + 2 < s_end) stage(IC2{}, buf ^ 1);
        __syncthreads();
        buf ^= 1;
        if (s + 2 >= s_end) break;
        if (s + 4 < s_end) fetch(IC1{}, s + 4);
        mma_step(buf);
        if (s + 3 < s_end) stage(IC0{}, buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    } else {
      fetch(IC0{}, s_begin);
      stage(IC0{}, 0);
      __syncthreads();
      int buf = 0;
      for (int s = s_begin; s < s_end; ++s) {
        const bool more = (s + 1 < s_end);
        if (more) fetch(IC0{}, s + 1);
        mma_step(buf);
        if (NBUF == 2) {
          if (more) stage(IC0{}, buf ^ 1);
          __syncthreads();
          buf ^= 1;
        } else {
          __syncthreads();           // single buffer: everyone is done reading before it is overwritten
          if (more) {
            stage(IC0{}, 0);
            __syncthreads();
          }
        }
      }
    }
  }

  if (p.vec_epi) {
    staged_epilogue<WGM, WGN, TM, TN>(p, acc, reinterpret_cast<float*>(smem_b), m0, n0, wm, wn, lane, wave, z);
    return;
  }
  float cs[TN], css[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { cs[j] = 0.f; css[j] = 0.f; }
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
            v = epilogue_one(p, v, m, co);
            p.out[m * p.out_ld + co] = v;
            cs[j] += v;
            css[j] += v * v;
          }
        }
      }
    }
  }
  if (p.stats) {   // host guarantees split_k == 1 and H*W %% BM == 0 (a tile never straddles two images)
    __shared__ float red[4 * TM * 0 + WGM * WGN * TN * 32][2];
    const int hw_o = p.Ho * p.Wo;
    emit_tile_stats<WGM, WGN, TN>(p, red, cs, css, wm, wn, lane, (int)(m0 / hw_o), (int)((m0 % hw_o) / (WGM * TM * 32)), n0);
  }
}

// ------------------------------------------------------------------------------------------------ 3x3 halo kernel
// The hot convolution (3x3, stride 1, pad 1; 85 % of the network's FLOPs) as a direct conv from an LDS-resident
// spatial tile instead of an im2col gather: a block owns an 8 x 32 pixel output tile x 64 output channels; per
// 32-channel chunk of Cin it stages the 10 x 34 input halo ONCE (21.8 KB bf16) plus the 9 x 64 x 32 weight slab
// (36.9 KB) and runs all 9 taps x 2 k-substeps from LDS: 72 MFMA 32x32x16 per wave per barrier pair, the 9-fold tap
// reuse is served by LDS (ds_read_b128, conflict-free with the 80-byte pixel pitch) instead of 9 trips to L1/L2.
// Per-CU global traffic: ~25 B/clk at full MFMA rate vs ~95 B/clk for the gather kernel on fp32 activations.
// The input is either fp32 (rounded to bf16 while staging) or a bf16 tensor that already carries the
// normalisation + activation (keep_norm_act_bf16), so no transcendental sits between load and LDS.
// The NEXT chunk's halo + weights are prefetched into registers while the current chunk is on the matrix cores;
// 73 KB of LDS -> 2 blocks per CU so one block's staging overlaps the other's MFMA phase.
// `upsample` folds nearest x2 into the halo addressing; split-K splits the Cin chunks (small maps at small batch).
// TW = 32: 8 x 32 output tile (maps >= 32 wide); TW = 16: 16 x 16 tile (the 16x16 latent maps: a whole image per block).
template <bool IN_BF16, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(ConvP p, int tiles_x, int tiles_y, int ncb) {
  constexpr int HALO_TW = TW, HALO_TH = 256 / TW, HALO_W = TW + 2, HALO_PIX = (HALO_TH + 2) * HALO_W;
  constexpr int RPT = 32 / TW;               // output rows covered by one 32-pixel MFMA block (1 or 2)
  // one LDS object: [halo | weights] during the main loop, [4 waves x 64 pixels x 68 floats] in the epilogue
  __shared__ __attribute__((aligned(16))) __bf16 lds_all[HALO_MAXPIX * HPITCH + 9 * 64 * HPITCH];
  __bf16* Hs = lds_all;
  __bf16* Ws = lds_all + HALO_MAXPIX * HPITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int total = gridDim.x;
  const int lid = xcd_remap(blockIdx.x, total);
  const int cb = lid % ncb;
  int t = lid / ncb;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int n = t / tiles_y;
  const int oy0 = ty * HALO_TH, ox0 = tx * HALO_TW;
  const int n0 = cb * 64;
  const int z = blockIdx.z;

  const int nchunks = p.Cin >> 5;
  const int per = (nchunks + p.split_k - 1) / p.split_k;
  const int ch_begin = z * per;
  const int ch_end = min(nchunks, ch_begin + per);

  const int Hv = p.upsample ? 2 * p.H : p.H;
  const int Wv = p.upsample ? 2 * p.W : p.W;

  // ---- chunk-invariant staging geometry
  const int g = tid & 3;                 // 8-channel group within the 32-channel chunk
  int h_off[HALO_IT];                    // element offset of (pixel, group) inside image n, -1 = zero padding / unused
#pragma unroll
  for (int it = 0; it < HALO_IT; ++it) {
    const int hp = (tid >> 2) + it * 64;
    h_off[it] = -1;
    if (hp < HALO_PIX) {
      const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        h_off[it] = (sy * p.W + sx) * p.in_ld + g * 8;
      }
    }
  }
  const long w_base = ((long)(n0 + (tid >> 2)) * 9) * p.Cin + g * 8;   // + tap*Cin + c0
  const long img_off = (long)n * p.H * p.W * p.in_ld;
  const unsigned short* in16 = reinterpret_cast<const unsigned short*>(p.in) + img_off;
  const float* in32 = p.in + img_off;

  // fp32 input may carry the previous normalisation: v = act(x*scale[n,c] + shift[n,c]) applied once per halo element
  // while staging (n is block-uniform, the thread's 8 channels are fixed within a chunk)
  const bool has_pro = !IN_BF16 && (p.pro_scale != nullptr || p.pro_act != KEEP_PRO_NONE);
  float u_sc[8], u_sh[8];
  uint4 hreg[HALO_IT];                   // bf16 input: one 16-B piece each
  float4 hlo[IN_BF16 ? 1 : HALO_IT], hhi[IN_BF16 ? 1 : HALO_IT];   // fp32 input: 8 floats per piece
  uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;   // named: a 9-element array is left in scratch by hipcc

  auto fetch = [&](int ch) {
    const int c0 = ch << 5;
    if (!IN_BF16 && p.pro_scale) {
      const float4 s0 = *reinterpret_cast<const float4*>(p.pro_scale + (long)n * p.Cin + c0 + g * 8);
      const float4 s1 = *reinterpret_cast<const float4*>(p.pro_scale + (long)n * p.Cin + c0 + g * 8 + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(p.pro_shift + (long)n * p.Cin + c0 + g * 8);
      const float4 h1 = *reinterpret_cast<const float4*>(p.pro_shift + (long)n * p.Cin + c0 + g * 8 + 4);
      u_sc[0] = s0.x; u_sc[1] = s0.y; u_sc[2] = s0.z; u_sc[3] = s0.w; u_sc[4] = s1.x; u_sc[5] = s1.y; u_sc[6] = s1.z; u_sc[7] = s1.w;
      u_sh[0] = h0.x; u_sh[1] = h0.y; u_sh[2] = h0.z; u_sh[3] = h0.w; u_sh[4] = h1.x; u_sh[5] = h1.y; u_sh[6] = h1.z; u_sh[7] = h1.w;
    }
#pragma unroll
    for (int it = 0; it < HALO_IT; ++it) {
      if (IN_BF16) {
        hreg[it] = make_uint4(0u, 0u, 0u, 0u);
        if (h_off[it] >= 0) hreg[it] = *reinterpret_cast<const uint4*>(in16 + h_off[it] + c0);
      } else {
        hlo[IN_BF16 ? 0 : it] = make_float4(0.f, 0.f, 0.f, 0.f);
        hhi[IN_BF16 ? 0 : it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h_off[it] >= 0) {
          const float* src = in32 + h_off[it] + c0;
          hlo[IN_BF16 ? 0 : it] = *reinterpret_cast<const float4*>(src);
          hhi[IN_BF16 ? 0 : it] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#define KEEP_WLOAD(TAP, R) R = *reinterpret_cast<const uint4*>(p.wb + w_base + (long)(TAP) * p.Cin + c0);
    KEEP_TAPS(KEEP_WLOAD)
#undef KEEP_WLOAD
  };

  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < HALO_IT; ++it) {
      const int hp = (tid >> 2) + it * 64;
      if (hp < HALO_PIX) {
        if (IN_BF16) {
          *reinterpret_cast<uint4*>(&Hs[hp * HPITCH + g * 8]) = hreg[it];
        } else {
          const float4 a = hlo[IN_BF16 ? 0 : it], b = hhi[IN_BF16 ? 0 : it];
          float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          if (has_pro && h_off[it] >= 0) {           // zero padding stays zero: it is applied AFTER the activation
            if (p.pro_scale) {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[j] = v[j] * u_sc[j] + u_sh[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pro_apply_fast(v[j], p.pro_act);
          }
          bf16x8 h;
#pragma unroll
          for (int j = 0; j < 8; ++j) h[j] = (__bf16)v[j];
          *reinterpret_cast<bf16x8*>(&Hs[hp * HPITCH + g * 8]) = h;
        }
      }
    }
#define KEEP_WSTORE(TAP, R) *reinterpret_cast<uint4*>(&Ws[((TAP) * 64 + (tid >> 2)) * HPITCH + g * 8]) = R;
    KEEP_TAPS(KEEP_WSTORE)
#undef KEEP_WSTORE
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // MFMA block i of this wave covers output rows (2*wave+i)*RPT .. +RPT-1; lane l31 -> (row l31/TW, column l31%TW)
  const int a_base = (((2 * wave) * RPT + l31 / TW) * HALO_W + (l31 % TW)) * HPITCH + lhi * 8;
  const int b_base = l31 * HPITCH + lhi * 8;                           // + (tap*64 + j*32)*HPITCH + ks*16

  if (ch_begin < ch_end) {
    fetch(ch_begin);
    stage();
    __syncthreads();
    for (int ch = ch_begin; ch < ch_end; ++ch) {
      const bool more = ch + 1 < ch_end;
      if (more) fetch(ch + 1);
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
              af[i] = *reinterpret_cast<const bf16x8*>(&Hs[a_base + ((i * RPT + kh) * HALO_W + kw) * HPITCH + ks * 16]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              bfr[j] = *reinterpret_cast<const bf16x8*>(&Ws[b_base + ((kh * 3 + kw) * 64 + j * 32) * HPITCH + ks * 16]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
          }
        }
      }
      __syncthreads();           // every wave is done with this chunk's LDS image
      if (more) {
        stage();
        __syncthreads();
      }
    }
  }

  // ---- epilogue through LDS: the MFMA C/D layout gives a lane ONE output channel x 16 pixels (4-byte strided
  // stores, 64 store instructions per wave, issue-bound: measured 2.1-2.5 TB/s).  Each wave parks its 64-pixel x
  // 64-channel tile in LDS and reads it back channel-contiguous: 16 bytes per lane, 4 full 256-byte pixel rows per
  // store instruction, float4 bias / residual / aux accesses, and the GroupNorm partial sums become lane-local
  // (4 fixed channels per lane).
  constexpr int EP = 68;                                   // floats per staged pixel row (64 + 4: conflict-free)
  float* et = reinterpret_cast<float*>(lds_all) + wave * 64 * EP;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int xc = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        et[(i * 32 + xc) * EP + j * 32 + l31] = acc[i][j][r];
      }
  // wave-local hand-off (same wave wrote and reads its region): LDS ops of one wave complete in order
  __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0)
  const int c4 = (lane & 15) * 4;                          // this lane's 4 channels within the 64
  const int prow = lane >> 4;                              // pixel sub-row 0..3
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
  const int co = n0 + c4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && p.split_k == 1) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const int px = it * 4 + prow;                          // 0..63 : MFMA block px>>5, pixel px&31 inside it
    const int oy = oy0 + (2 * wave + (px >> 5)) * RPT + (px & 31) / TW;
    const long m = ((long)n * p.Ho + oy) * p.Wo + ox0 + (px & 31) % TW;
    float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
    if (p.split_k > 1) {
      *reinterpret_cast<float4*>(p.ws + ((long)z * p.M + m) * p.Cout + co) = v;
      continue;
    }
    float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = act_apply_fast(e[q], p.epi_act);
    if (p.res) {
      const float4 r4 = *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
      const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
      if (p.aux) {
        const float4 a4 = *reinterpret_cast<const float4*>(p.aux + m * (long)p.Cout + co);
        const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = rr[q] + p.aux_w * (rr[q] * aa[q] + e[q]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] += rr[q];
      }
    }
      *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s4[q] += e[q];
      ss4[q] += e[q] * e[q];
    }
  }
  if (p.stats) {   // split_k == 1: combine the 4 pixel sub-rows (lanes +16, +32), then the 4 waves through LDS
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s4[q] += __shfl_xor(s4[q], 16);
      s4[q] += __shfl_xor(s4[q], 32);
      ss4[q] += __shfl_xor(ss4[q], 16);
      ss4[q] += __shfl_xor(ss4[q], 32);
    }
    __syncthreads();                                       // every wave is done with its staged tile
    float* red = reinterpret_cast<float*>(lds_all);        // [4 waves][64 ch][2]
    if (lane < 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        red[(wave * 64 + c4 + q) * 2 + 0] = s4[q];
        red[(wave * 64 + c4 + q) * 2 + 1] = ss4[q];
      }
    }
    __syncthreads();
    if (tid < 64) {
      float a = 0.f, b2 = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        a += red[(wv * 64 + tid) * 2 + 0];
        b2 += red[(wv * 64 + tid) * 2 + 1];
      }
      float* dst = p.stats + (((long)n * p.stats_P + (ty * tiles_x + tx)) * p.Cout + n0 + tid) * 2;
      dst[0] = a;
      dst[1] = b2;
    }
  }
}

// ------------------------------------------------------------------------------------------------ persistent work items
// (A wave-specialised producer/consumer version of the halo kernel -- 4 loader waves + 4 MFMA waves, one block per CU -- was
// measured SLOWER than v1: only 256 threads issue loads, halving the bytes in flight per CU, and the kernel is bound by the
// L2->CU operand stream.  It was removed; v3 below keeps every wave loading, staging and computing.)
// ------------------------------------------------------------------------------------------------ halo v3
// v1 made persistent: v3 keeps v1's
// two 4-wave blocks per CU -- every wave loads, converts, stages and computes -- but each block walks the work items
// (tile, cout-block, k-split) itself: the geometry / launch cost is paid once per block instead of once per tile, and
// the first chunk of the NEXT item is already in flight (in registers) while the epilogue of the current item runs,
// so output stores overlap input loads.
// SIMPLE_EPI: split_k == 1, no aux tensor, no epilogue activation (every ResBlock / Upsample conv): the store loop is
// compiled without those uniform branches and the activation switch.
template <bool IN_BF16, int TW, bool SIMPLE_EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_halo3_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HALO_TH = 256 / TW, HALO_W = TW + 2, HALO_PIX = (HALO_TH + 2) * HALO_W;
  constexpr int RPT = 32 / TW;
  __shared__ __attribute__((aligned(16))) __bf16 lds_all[HALO_MAXPIX * HPITCH + 9 * 64 * HPITCH];
  __bf16* Hs = lds_all;
  __bf16* Ws = lds_all + HALO_MAXPIX * HPITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int items_per_z = n_items / p.split_k;
  const int Hv = p.upsample ? 2 * p.H : p.H;
  const int Wv = p.upsample ? 2 * p.W : p.W;
  const int g = tid & 3;

  int h_off[HALO_IT];
  long img_off = 0, w_base = 0;
  bool w_ok = true;
  auto setup = [&](const HaloItem& it) {
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      h_off[k] = -1;
      if (hp < HALO_PIX) {
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = it.oy0 - 1 + hy, ix = it.ox0 - 1 + hx;
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
          const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          h_off[k] = (sy * p.W + sx) * p.in_ld + g * 8;
        }
      }
    }
    img_off = (long)it.n * p.H * p.W * p.in_ld;
    w_ok = (it.n0 + (tid >> 2)) < p.Cout;             // Cout % 64 == 32: the last cout-block is half empty
    w_base = w_ok ? ((long)(it.n0 + (tid >> 2)) * 9) * p.Cin + g * 8 : 0;
  };

  uint4 hreg[HALO_IT];
  float4 hlo[IN_BF16 ? 1 : HALO_IT], hhi[IN_BF16 ? 1 : HALO_IT];
  uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
  auto fetch = [&](int ch) {
    const int c0 = ch << 5;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      if (IN_BF16) {
        hreg[k] = make_uint4(0u, 0u, 0u, 0u);
        if (h_off[k] >= 0)
          hreg[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(p.in) + img_off + h_off[k] + c0);
      } else {
        hlo[IN_BF16 ? 0 : k] = make_float4(0.f, 0.f, 0.f, 0.f);
        hhi[IN_BF16 ? 0 : k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h_off[k] >= 0) {
          const float* src = p.in + img_off + h_off[k] + c0;
          hlo[IN_BF16 ? 0 : k] = *reinterpret_cast<const float4*>(src);
          hhi[IN_BF16 ? 0 : k] = *reinterpret_cast<const float4*>(src + 4);
        }
      }
    }
#define KEEP_WLOAD(TAP, R) R = w_ok ? *reinterpret_cast<const uint4*>(p.wb + w_base + (long)(TAP) * p.Cin + c0) : make_uint4(0u, 0u, 0u, 0u);
    KEEP_TAPS(KEEP_WLOAD)
#undef KEEP_WLOAD
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      if (hp < HALO_PIX) {
        if (IN_BF16) {
          *reinterpret_cast<uint4*>(&Hs[hp * HPITCH + g * 8]) = hreg[k];
        } else {
          const float4 a = hlo[IN_BF16 ? 0 : k], b = hhi[IN_BF16 ? 0 : k];
          bf16x8 h;
          h[0] = (__bf16)a.x; h[1] = (__bf16)a.y; h[2] = (__bf16)a.z; h[3] = (__bf16)a.w;
          h[4] = (__bf16)b.x; h[5] = (__bf16)b.y; h[6] = (__bf16)b.z; h[7] = (__bf16)b.w;
          *reinterpret_cast<bf16x8*>(&Hs[hp * HPITCH + g * 8]) = h;
        }
      }
    }
#define KEEP_WSTORE(TAP, R) *reinterpret_cast<uint4*>(&Ws[((TAP) * 64 + (tid >> 2)) * HPITCH + g * 8]) = R;
    KEEP_TAPS(KEEP_WSTORE)
#undef KEEP_WSTORE
  };

  f32x16 acc[2][2];
  const int a_base = (((2 * wave) * RPT + l31 / TW) * HALO_W + (l31 % TW)) * HPITCH + lhi * 8;
  const int b_base = l31 * HPITCH + lhi * 8;
  auto mma = [&]() {
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 af[2], bfr[2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            af[i] = *reinterpret_cast<const bf16x8*>(&Hs[a_base + ((i * RPT + kh) * HALO_W + kw) * HPITCH + ks * 16]);
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bfr[j] = *reinterpret_cast<const bf16x8*>(&Ws[b_base + ((kh * 3 + kw) * 64 + j * 32) * HPITCH + ks * 16]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  auto epilogue_t = [&](const HaloItem& it, auto res_c, auto o16_c) {
    constexpr bool HAS_RES = decltype(res_c)::value, OUT16 = decltype(o16_c)::value;   // uniform per launch: compiled in
    constexpr int EP = 68;
    float* et = reinterpret_cast<float*>(lds_all) + wave * 64 * EP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int co = it.n0 + c4;
    const bool cok = co < p.Cout;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && p.split_k == 1 && cok) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
    // unroll depth measured: 8 without a residual (109 us vs 113 at 4 and 119 at 16 on 128 ch @256^2), 16 with one (all
    // residual rows in flight: 142 us vs 146 / 152)
    constexpr int UNR = HAS_RES ? 16 : 8;
    auto m_of = [&](int q16) {
      const int px = q16 * 4 + prow;
      const int oy = it.oy0 + (2 * wave + (px >> 5)) * RPT + (px & 31) / TW;
      return ((long)it.n * p.Ho + oy) * p.Wo + it.ox0 + (px & 31) % TW;
    };
    // all residual rows loaded before the first store: inside the loop each load sits behind the previous store (`res` may alias
    // `out`) and its vmcnt wait exposes a memory round trip per row -- unrolling alone does not put them in flight
    float4 rpre[HAS_RES ? 16 : 1];
    if (HAS_RES && cok && p.split_k == 1) {
#pragma unroll
      for (int q16 = 0; q16 < 16; ++q16) rpre[q16] = *reinterpret_cast<const float4*>(p.res + m_of(q16) * p.res_ld + co);
    }
#pragma unroll UNR
    for (int q16 = 0; q16 < 16; ++q16) {
      if (!cok) break;
      const int px = q16 * 4 + prow;
      const long m = m_of(q16);
      const float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
      if (!SIMPLE_EPI && p.split_k > 1) {
        *reinterpret_cast<float4*>(p.ws + ((long)it.z * p.M + m) * p.Cout + co) = v;
        continue;
      }
      float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
      if (!SIMPLE_EPI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = act_apply_fast(e[q], p.epi_act);
      }
      if (HAS_RES) {
        const float4 r4 = rpre[HAS_RES ? q16 : 0];
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
        if (!SIMPLE_EPI && p.aux) {
          const float4 a4 = *reinterpret_cast<const float4*>(p.aux + m * (long)p.Cout + co);
          const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = rr[q] + p.aux_w * (rr[q] * aa[q] + e[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] += rr[q];
        }
      }
      if (OUT16) {               // ResBlock conv1 -> GroupNorm -> conv2: the only reader is the normalise pass
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        bf16x4_t h;
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = (__bf16)e[q];
        *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(p.out) + m * p.out_ld + co) = h;
      } else {
        *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += e[q];
        ss4[q] += e[q] * e[q];
      }
    }
    if (p.stats) {          // per wave: stats_P = 4 * tiles, partial index = tile*4 + wave (no cross-wave reduction)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += __shfl_xor(s4[q], 16);
        s4[q] += __shfl_xor(s4[q], 32);
        ss4[q] += __shfl_xor(ss4[q], 16);
        ss4[q] += __shfl_xor(ss4[q], 32);
      }
      if (lane < 16 && cok) {
        float* dst = p.stats + (((long)it.n * p.stats_P + (it.ty * tiles_x + it.tx) * 4 + wave) * p.Cout + co) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dst[q * 2 + 0] = s4[q];
          dst[q * 2 + 1] = ss4[q];
        }
      }
    }
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  HaloItem cur = halo_decode<TW>(p, item, items_per_z, tiles_x, tiles_y, ncb);
  setup(cur);
  if (cur.ch_begin < cur.ch_end) fetch(cur.ch_begin);
  while (true) {
    const bool valid = cur.ch_begin < cur.ch_end;
    if (valid) stage();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ch = cur.ch_begin; ch < cur.ch_end; ++ch) {
      const bool more = ch + 1 < cur.ch_end;
      if (more) fetch(ch + 1);
      mma();
      __syncthreads();
      if (more) {
        stage();
        __syncthreads();
      }
    }
    const int next_item = item + gridDim.x;
    const bool has_next = next_item < n_items;
    HaloItem nxt = cur;
    if (has_next) {
      nxt = halo_decode<TW>(p, next_item, items_per_z, tiles_x, tiles_y, ncb);
      setup(nxt);
      if (nxt.ch_begin < nxt.ch_end) fetch(nxt.ch_begin);     // in flight during the epilogue below
    }
    if (p.res)
      epilogue_t(cur, std::true_type{}, std::false_type{});          // (a bf16 output never carries a residual)
    else if (p.out_bf16)
      epilogue_t(cur, std::false_type{}, std::true_type{});
    else
      epilogue_t(cur, std::false_type{}, std::false_type{});
    if (!has_next) break;
    __syncthreads();                                            // every wave is done with its staged tile
    item = next_item;
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------ halo f32
// The fp32-parity policy's 3x3 stride-1 convolution: the persistent halo kernel above with exact-f32 operands
// (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD).  The gather kernel (conv_f32_kernel) re-reads the A tile once per tap and
// measured 72 TF of 157: with the loads ablated it ran 109 TF, with the MFMAs ablated the loads alone took half the
// kernel time -- the L2->CU operand stream and the matrix pipe took turns instead of overlapping.  Here a block stages
// the (8+2)x(32+2) (or 18x18) pixel halo of 16 channels ONCE and all 9 taps read it from LDS: 58 KB of operands per
// 288 MFMAs (18.4 k matrix cycles) per wave, ~3 B/clk/CU, so the matrix pipe is the only busy resource.
//   LDS rows (one pixel / one (tap, cout) weight row) are 16 floats at a 20-float pitch (80 B = 5 x 16-B slots, odd).
//   The 32x32x2 MFMA takes k from lanes 0-31 and k+1 from lanes 32-63; WHICH channel plays k at step j is free as long
//   as A and B agree, so lane-half h uses channel 8h + j at step j: each lane's 8 steps are 8 CONTIGUOUS floats = two
//   ds_read_b128 per operand row per tap instead of eight ds_read_b32 (the wave issues 8 LDS reads per 32 MFMAs).
//   GroupNorm affine + activation (exact expf) are applied when the raw fp32 halo registers are written to LDS: ~500
//   VALU instructions per 16-channel chunk beside 18 k cycles of MFMA, so no separate normalisation pass is needed.
#define FCSH 4                               // log2(channels per chunk)
#define FPITCH 20

// PRO: the prologue activation (KEEP_PRO_*) compiled in -- the staging step applies it to 24 values per thread and chunk.
template <int TW, int PRO>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_f32_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HALO_TH = 256 / TW, HALO_W = TW + 2, HALO_PIX = (HALO_TH + 2) * HALO_W;
  constexpr int RPT = 32 / TW;
  constexpr int MAIN_F = HALO_MAXPIX * FPITCH + 9 * 64 * FPITCH;
  constexpr int EPI_F = 4 * 64 * 68;
  __shared__ __attribute__((aligned(16))) float lds_f[MAIN_F > EPI_F ? MAIN_F : EPI_F];
  float* Hs = lds_f;
  float* Ws = lds_f + HALO_MAXPIX * FPITCH;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int items_per_z = n_items / p.split_k;
  const int Hv = p.upsample ? 2 * p.H : p.H;
  const int Wv = p.upsample ? 2 * p.W : p.W;
  const int g = tid & 3;
  const bool has_pro = p.pro_scale != nullptr || PRO != KEEP_PRO_NONE;

  int h_off[HALO_IT];
  long img_off = 0, w_base = 0, sc_off = 0;
  bool w_ok = true;
  auto setup = [&](const HaloItem& it) {
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      h_off[k] = -1;
      if (hp < HALO_PIX) {
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        int iy = it.oy0 - 1 + hy, ix = it.ox0 - 1 + hx;
        KEEP_REFLECT(iy, ix, Hv, Wv)
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
          const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          h_off[k] = (sy * p.W + sx) * p.in_ld + g * 4;
        }
      }
    }
    img_off = (long)it.n * p.H * p.W * p.in_ld;
    sc_off = (long)it.n * p.Cin + g * 4;
    w_ok = (it.n0 + (tid >> 2)) < p.Cout;
    w_base = w_ok ? ((long)(it.n0 + (tid >> 2)) * 9) * p.Cin + g * 4 : 0;
  };

  float4 hreg[HALO_IT];
  float4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int ch) {
    const int c0 = ch << FCSH;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      hreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h_off[k] >= 0) hreg[k] = *reinterpret_cast<const float4*>(p.in + img_off + h_off[k] + c0);
    }
#define KEEP_WLOADF(TAP, R) R = w_ok ? *reinterpret_cast<const float4*>(p.w + w_base + (long)(TAP) * p.Cin + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    KEEP_TAPS(KEEP_WLOADF)
#undef KEEP_WLOADF
    if (p.pro_scale) {
      sc4 = *reinterpret_cast<const float4*>(p.pro_scale + sc_off + c0);
      sh4 = *reinterpret_cast<const float4*>(p.pro_shift + sc_off + c0);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      if (hp < HALO_PIX) {
        float4 v = hreg[k];
        if (has_pro && h_off[k] >= 0) {      // zero padding applies to the normalised + activated tensor
          v.x = pro_apply(v.x * sc4.x + sh4.x, PRO);
          v.y = pro_apply(v.y * sc4.y + sh4.y, PRO);
          v.z = pro_apply(v.z * sc4.z + sh4.z, PRO);
          v.w = pro_apply(v.w * sc4.w + sh4.w, PRO);
        }
        *reinterpret_cast<float4*>(&Hs[hp * FPITCH + g * 4]) = v;
      }
    }
#define KEEP_WSTOREF(TAP, R) *reinterpret_cast<float4*>(&Ws[((TAP) * 64 + (tid >> 2)) * FPITCH + g * 4]) = R;
    KEEP_TAPS(KEEP_WSTOREF)
#undef KEEP_WSTOREF
  };

  f32x16 acc[2][2];
  const int a_base = (((2 * wave) * RPT + l31 / TW) * HALO_W + (l31 % TW)) * FPITCH + lhi * 8;
  const int b_base = l31 * FPITCH + lhi * 8;
  // (An explicit register double buffer of the fragments -- tap t+1 read while tap t multiplies, issue order pinned with
  // sched_group_barrier -- measured the same: the second wave on the SIMD already covers the ds_read latency.)
  auto mma = [&]() {
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        f32x4 af[2][2], bfr[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float* src = &Hs[a_base + ((i * RPT + kh) * HALO_W + kw) * FPITCH];
          af[i][0] = *reinterpret_cast<const f32x4*>(src);
          af[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float* src = &Ws[b_base + ((kh * 3 + kw) * 64 + j * 32) * FPITCH];
          bfr[j][0] = *reinterpret_cast<const f32x4*>(src);
          bfr[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
        }
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][k2 >> 2][k2 & 3], bfr[j][k2 >> 2][k2 & 3], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  auto epilogue_t = [&](const HaloItem& it, auto simple_c) {
    constexpr bool SIMPLE = decltype(simple_c)::value;      // split_k == 1, no aux, no activation
    constexpr int EP = 68;
    float* et = lds_f + wave * 64 * EP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int co = it.n0 + c4;
    const bool cok = co < p.Cout;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && p.split_k == 1 && cok) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
    auto m_of = [&](int q16) {
      const int px = q16 * 4 + prow;
      const int oy = it.oy0 + (2 * wave + (px >> 5)) * RPT + (px & 31) / TW;
      return ((long)it.n * p.Ho + oy) * p.Wo + it.ox0 + (px & 31) % TW;
    };
    // simple form: the residual rows of a group of four iterations are loaded before the group's first store (inside the loop every
    // load sits behind the previous store -- `res` may alias `out` -- and its vmcnt wait exposes a memory round trip per row)
    float4 rpre[4];
#pragma unroll 4
    for (int q16 = 0; q16 < 16; ++q16) {
      if (!cok) break;
      if (SIMPLE && p.res && (q16 & 3) == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) rpre[u] = *reinterpret_cast<const float4*>(p.res + m_of(q16 + u) * p.res_ld + co);
      }
      const int px = q16 * 4 + prow;
      const long m = m_of(q16);
      const float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
      if (!SIMPLE && p.split_k > 1) {
        *reinterpret_cast<float4*>(p.ws + ((long)it.z * p.M + m) * p.Cout + co) = v;
        continue;
      }
      float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
      if (!SIMPLE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = act_apply(e[q], p.epi_act);
      }
      if (p.res) {
        const float4 r4 = SIMPLE ? rpre[q16 & 3] : *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
        const float rr[4] = {r4.x, r4.y, r4.z, r4.w};
        if (!SIMPLE && p.aux) {
          const float4 a4 = *reinterpret_cast<const float4*>(p.aux + m * (long)p.Cout + co);
          const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = rr[q] + p.aux_w * (rr[q] * aa[q] + e[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] += rr[q];
        }
      }
      *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += e[q];
        ss4[q] += e[q] * e[q];
      }
    }
    if (p.stats) {          // per wave: stats_P = 4 * tiles, partial index = tile*4 + wave
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += __shfl_xor(s4[q], 16);
        s4[q] += __shfl_xor(s4[q], 32);
        ss4[q] += __shfl_xor(ss4[q], 16);
        ss4[q] += __shfl_xor(ss4[q], 32);
      }
      if (lane < 16 && cok) {
        float* dst = p.stats + (((long)it.n * p.stats_P + (it.ty * tiles_x + it.tx) * 4 + wave) * p.Cout + co) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dst[q * 2 + 0] = s4[q];
          dst[q * 2 + 1] = ss4[q];
        }
      }
    }
  };

  const bool simple_epi = p.split_k == 1 && !p.aux && p.epi_act == KEEP_ACT_NONE;
  int item = blockIdx.x;
  if (item >= n_items) return;
  HaloItem cur = halo_decode<TW, FCSH>(p, item, items_per_z, tiles_x, tiles_y, ncb);
  setup(cur);
  if (cur.ch_begin < cur.ch_end) fetch(cur.ch_begin);
  while (true) {
    const bool valid = cur.ch_begin < cur.ch_end;
    if (valid) stage();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ch = cur.ch_begin; ch < cur.ch_end; ++ch) {
      const bool more = ch + 1 < cur.ch_end;
      if (more) fetch(ch + 1);
      mma();
      __syncthreads();
      if (more) {
        stage();
        __syncthreads();
      }
    }
    const int next_item = item + gridDim.x;
    const bool has_next = next_item < n_items;
    HaloItem nxt = cur;
    if (has_next) {
      nxt = halo_decode<TW, FCSH>(p, next_item, items_per_z, tiles_x, tiles_y, ncb);
      setup(nxt);
      if (nxt.ch_begin < nxt.ch_end) fetch(nxt.ch_begin);     // in flight during the epilogue below
    }
    if (simple_epi)
      epilogue_t(cur, std::true_type{});
    else
      epilogue_t(cur, std::false_type{});
    if (!has_next) break;
    __syncthreads();
    item = next_item;
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------ 3x3, Cin <= 4
// The first convolutions (RGB -> 64 channels at 512x512: VQ conv_in of the LQ encoder over all B*T frames and of the HQ
// encoder every frame).  K = 27: on the generic tile kernel a block is one K step wrapped in prologue + epilogue
// (731 us for 16 frames, 1.5 TB/s on an op that only has to write 1.07 GB).  Here: persistent blocks (2 per CU), the
// weights [Cout_blk 64][K 32] stay in LDS, per 8x32-pixel item the 10x34x3 fp32 halo is loaded once (coalesced 4-byte
// loads of contiguous rows), every thread expands ITS pixel into one 32-wide bf16 im2col row in LDS, and the wave runs
// 8 MFMAs (64 pixels x 64 couts, K = 32) before the same LDS-staged float4 epilogue + GroupNorm partials as the halo kernel.
#define C3_K 32
#define C3_PITCH 40                          // bf16 per im2col / weight row (80 B: conflict-free ds_read_b128)

__global__ __launch_bounds__(256, 2) void conv3x3_c3_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HW_ = 34, HROWS = 10;
  constexpr int EPI_B = 4 * 64 * 68 * 4;                                    // staging tile; the im2col rows alias it
  constexpr int A_B = 256 * C3_PITCH * 2;
  static_assert(A_B <= EPI_B, "im2col rows fit the staging area");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[EPI_B + 64 * C3_PITCH * 2 + HROWS * HW_ * 4 * 4];
  float* et_base = reinterpret_cast<float*>(lds_raw);
  __bf16* As = reinterpret_cast<__bf16*>(lds_raw);                          // [256 px][C3_PITCH]
  __bf16* Ws = reinterpret_cast<__bf16*>(lds_raw + EPI_B);                  // [64][C3_PITCH], resident across items
  float* Hs = reinterpret_cast<float*>(lds_raw + EPI_B + 64 * C3_PITCH * 2);   // [10][34 * Cin]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Cin = p.Cin, K = 9 * Cin;                                        // K <= 36 > 32 only for Cin = 4: host limits Cin <= 3
  const int rowf = HW_ * Cin;                                                // floats per halo row

  int cur_cb = -1;
  auto load_weights = [&](int cb) {                                          // [64 couts][K] bf16, zero padded to 32
    for (int i = tid; i < 64 * C3_K; i += 256) {
      const int co = i >> 5, k = i & 31;
      unsigned short v = 0;
      if (k < K && cb * 64 + co < p.Cout) v = p.wb[(long)(cb * 64 + co) * K + k];
      reinterpret_cast<unsigned short*>(Ws)[co * C3_PITCH + k] = v;
    }
  };

  f32x16 acc[2][2];
  const bool has_act = p.epi_act != KEEP_ACT_NONE;
  const int py = tid >> 5, px = tid & 31;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int lid = xcd_remap(item, n_items);
    const int cb = lid % ncb;
    int t = lid / ncb;
    const int tx = t % tiles_x; t /= tiles_x;
    const int ty = t % tiles_y;
    const int n = t / tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 32, n0 = cb * 64;
    __syncthreads();                                                         // previous item's staging tile fully stored
    if (cb != cur_cb) {
      load_weights(cb);
      cur_cb = cb;
    }
    // ---- halo rows: (oy0-1 .. oy0+8) x (ox0-1 .. ox0+32) x Cin, contiguous in memory per row
    const float* img = p.in + (long)n * p.H * p.W * p.in_ld;
    for (int i = tid; i < HROWS * rowf; i += 256) {
      const int hy = i / rowf, r = i - hy * rowf;
      const int hx = r / Cin, c = r - hx * Cin;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      float v = 0.f;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) v = img[((long)iy * p.W + ix) * p.in_ld + c];
      Hs[i] = v;
    }
    __syncthreads();
    // ---- im2col row of this thread's pixel: k = (kh*3 + kw)*Cin + c
    {
      float vals[C3_K];
#pragma unroll
      for (int k = 0; k < C3_K; ++k) vals[k] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kwc = 0; kwc < 9; ++kwc)                                    // (kw, c) run is contiguous in the halo row for Cin = 3
          if (kwc < 3 * Cin && kh * 3 * Cin + kwc < C3_K) vals[kh * 3 * Cin + kwc] = Hs[(py + kh) * rowf + px * Cin + kwc];
      bf16x8 h[4];
#pragma unroll
      for (int k = 0; k < C3_K; ++k) h[k >> 3][k & 7] = (__bf16)vals[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<bf16x8*>(&As[tid * C3_PITCH + q * 8]) = h[q];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8*>(&As[(wave * 64 + i * 32 + l31) * C3_PITCH + ks * 16 + lhi * 8]);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(&Ws[(j * 32 + l31) * C3_PITCH + ks * 16 + lhi * 8]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                                         // A reads done: the area becomes the staging tile
    // ---- epilogue: wave tile = rows 2*wave, 2*wave+1 of the item (64 pixels) x 64 couts
    constexpr int EP = 68;
    float* et = et_base + wave * 64 * EP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int co = n0 + c4;
    const bool cok = co < p.Cout;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && cok) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll 8
    for (int q16 = 0; q16 < 16; ++q16) {
      if (!cok) break;
      const int pxl = q16 * 4 + prow;
      const int oy = oy0 + 2 * wave + (pxl >> 5);
      const long m = ((long)n * p.Ho + oy) * p.Wo + ox0 + (pxl & 31);
      const float4 v = *reinterpret_cast<const float4*>(et + pxl * EP + c4);
      float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
      if (has_act) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = act_apply_fast(e[q], p.epi_act);
      }
      *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += e[q];
        ss4[q] += e[q] * e[q];
      }
    }
    if (p.stats) {          // per wave: stats_P = Ho*Wo/64, partial index = tile*4 + wave
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += __shfl_xor(s4[q], 16);
        s4[q] += __shfl_xor(s4[q], 32);
        ss4[q] += __shfl_xor(ss4[q], 16);
        ss4[q] += __shfl_xor(ss4[q], 32);
      }
      if (lane < 16 && cok) {
        float* dst = p.stats + (((long)n * p.stats_P + (ty * tiles_x + tx) * 4 + wave) * p.Cout + co) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          dst[q * 2 + 0] = s4[q];
          dst[q * 2 + 1] = ss4[q];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ 3x3, Cout <= 4
// The generator's output convolution (64 -> 3 channels at 512x512, VQ:241): on a 32-wide MFMA tile 29 of 32 output
// columns are padding (the 128x32 gather tile measured 6.7 TF, 1 ms per 8 frames).  With <= 4 output channels the op
// is 9*Cin FMAs per pixel per channel -- plain fp32 VALU work on an LDS halo: one thread per output pixel, 8x32-pixel
// tile per block, 16-channel chunks of the (8+2)x(32+2) halo staged once (GroupNorm affine + activation applied
// while staging), weights read through the scalar cache (uniform addresses), exact fp32 in both precision policies.
#define SC_TH 8
#define SC_TW 32
#define SC_CH 16
#define SC_PITCH 20

__global__ __launch_bounds__(256) void conv3x3_cout4_kernel(ConvP p) {
  constexpr int HW_ = SC_TW + 2, HPIX = (SC_TH + 2) * HW_;
  __shared__ __attribute__((aligned(16))) float Hs[HPIX * SC_PITCH];
  const int tid = threadIdx.x;
  const int tiles_x = p.Wo / SC_TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, n = blockIdx.y;
  const int oy0 = ty * SC_TH, ox0 = tx * SC_TW;
  const int py = tid >> 5, px = tid & 31;
  const int g = tid & 3;
  const float* in = p.in + (long)n * p.H * p.W * p.in_ld;
  const bool has_pro = p.pro_scale != nullptr || p.pro_act != KEEP_PRO_NONE;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < p.Cin; c0 += SC_CH) {
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.pro_scale) {
      sc4 = *reinterpret_cast<const float4*>(p.pro_scale + (long)n * p.Cin + c0 + g * 4);
      sh4 = *reinterpret_cast<const float4*>(p.pro_shift + (long)n * p.Cin + c0 + g * 4);
    }
    __syncthreads();                                  // previous chunk fully consumed
    for (int hp = tid >> 2; hp < HPIX; hp += 64) {
      const int hy = hp / HW_, hx = hp - hy * HW_;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
        v = *reinterpret_cast<const float4*>(in + ((long)iy * p.W + ix) * p.in_ld + c0 + g * 4);
        if (has_pro) {                                // zero padding applies to the normalised tensor
          v.x = pro_apply(v.x * sc4.x + sh4.x, p.pro_act);
          v.y = pro_apply(v.y * sc4.y + sh4.y, p.pro_act);
          v.z = pro_apply(v.z * sc4.z + sh4.z, p.pro_act);
          v.w = pro_apply(v.w * sc4.w + sh4.w, p.pro_act);
        }
      }
      *reinterpret_cast<float4*>(&Hs[hp * SC_PITCH + g * 4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      const float* hrow = &Hs[((py + kh) * HW_ + px + kw) * SC_PITCH];
      float x[SC_CH];
#pragma unroll
      for (int q = 0; q < SC_CH / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(hrow + q * 4);
        x[q * 4 + 0] = v.x; x[q * 4 + 1] = v.y; x[q * 4 + 2] = v.z; x[q * 4 + 3] = v.w;
      }
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        if (co < p.Cout) {                            // uniform branch; weight addresses are uniform -> scalar loads
          const float* wrow = p.w + ((long)co * 9 + tap) * p.Cin + c0;
#pragma unroll
          for (int c = 0; c < SC_CH; ++c) acc[co] = fmaf(x[c], wrow[c], acc[co]);
        }
      }
    }
  }
  const long m = ((long)n * p.Ho + oy0 + py) * p.Wo + ox0 + px;
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    if (co < p.Cout) {
      float v = acc[co] + (p.bias ? p.bias[co] : 0.f);
      p.out[m * p.out_ld + co] = act_apply(v, p.epi_act);
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

// The same reduction four channels per thread (Cout, out_ld, res_ld multiples of 4, 16-byte aligned tensors: p.vec_epi): float4 loads of
// the partials in z order, the element-wise epilogue of epilogue_one per lane -- bit-identical to the scalar kernel (every output element
// sees the same operations in the same order), no 64-bit division per element.
__global__ void conv_splitk_reduce4_kernel(ConvP p) {
  const int c4n = p.Cout >> 2;
  const long total4 = (long)p.M * c4n;
  const long total = (long)p.M * p.Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / c4n;
    const int co = (int)(i - m * c4n) << 2;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < p.split_k; ++z) {
      const float4 t = *reinterpret_cast<const float4*>(p.ws + (long)z * total + m * p.Cout + co);
      v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    float e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e[q] = epilogue_one(p, v[q], m, co + q);
    *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
  }
}

// ------------------------------------------------------------------------------------------------ plan + dispatch
// ONE place decides which kernel a keep_conv2d call runs on, its tile, the split-K factor and the layout of the epilogue
// statistics: plan_conv().  keep_conv2d_plan() exposes that decision to the host, which sizes the workspace / statistics
// buffers from it and never re-derives kernel internals (a retune here cannot silently corrupt a caller).
int keep_conv2d_x3_halo(const keep_conv2d_args* a, ConvP& p, hipStream_t st);
bool keep_conv_x3_up2_ok(const keep_conv2d_args* a);
int keep_conv2d_x3_gather(const keep_conv2d_args* a, ConvP& p, int tile, hipStream_t st);
bool keep_conv_x3_gather_is_gemm(const keep_conv2d_args* a);
int keep_conv2d_x3_c3(const keep_conv2d_args* a, ConvP& p, hipStream_t st);
bool keep_conv_x3_halo_ok(const keep_conv2d_args* a);
bool keep_conv_x3_stream_ok(const keep_conv2d_args* a, const ConvP& p, int split_k);
bool keep_conv_x3_gather_ok(const keep_conv2d_args* a, const ConvP& p);
bool keep_gemm_x3l_ok(const keep_conv2d_args* a);
int keep_gemm_x3l_waves(const keep_conv2d_args* a);
int keep_conv2d_x3_gemm_lat(const keep_conv2d_args* a, ConvP& p, hipStream_t st);

enum ConvPath {
  PATH_COUT4 = 0, PATH_C3, PATH_HALO_F32, PATH_HALO_BF16, PATH_HALO_BF16_V1, PATH_GATHER_BF16, PATH_GATHER_F32, PATH_HALO_X3,
  PATH_GATHER_X3, PATH_NEEDS_PRENORM, PATH_C3_X3
};

struct ConvPlan {
  ConvPath path;
  int tile;            // gather kernels: 0 = 128x32 (4,1,1,1), 1 = 64x64 (2,2,1,1), 2 = 128x128 (2,2,2,2)
  bool plain, wide, bk256;
  int split_k;
  int stats_rows;      // output pixels per statistics partial; 0 = this call cannot emit statistics
  int wants_bf16_input, out_bf16_ok;
  bool amax_ok;          // this path can fill x3_out_amax (x3 kernels, single pass)
  char kernel[64];
};

static int n_cu_cached() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

static const int kTargetWaves = 1024;
// Parity policies (exact f32, x3): every choice that changes the ORDER of a floating-point sum -- the split-K factor, and with
// it the statistics partition -- is made from the per-image geometry and this fixed reference batch, never from the actual N.
// A clip's result is then bit-identical whatever its batch-mates are (KeepNet.clips_per_call follows free HBM; round 2's plans
// followed N and a clip's code indices could differ between a batch of 8 and a batch of 15).  16 images = the engine's design
// point (16 clips per call): maps of 32x32 and larger fill the chip without split-K at that count, the 16x16 stages split 4 ways
// at every batch size.  Measured (round 3, x3): reference 16 -> 235 frames/s at B = 16 and 93 at B = 1; 8 -> 230 / 101; 2 -> - /
// 109: KEEP_PLAN_REF_IMAGES selects a latency profile (a deployment-wide setting like the precision policy -- results are
// invariant within one setting).
static long plan_ref_images(const keep_conv2d_args* a) { return a->plan_ref_images > 0 ? a->plan_ref_images : 16; }
// gather kernels: launches of at most this many output rows use 64x64 tiles (more blocks), larger ones 128x128.  A tuning
// rule the HOST never mirrors (keep_conv2d_plan reports what follows from it): KEEP_CONV_SMALL_TILES forces the small tile (tests).
static long small_m_threshold(const keep_conv2d_args* a) { return (a->flags & KEEP_CONV_SMALL_TILES) ? (1L << 62) : 4096; }   // below this many matrix-core waves a launch cannot fill 256 CUs x 4 SIMDs -> split K

static int validate_conv(const keep_conv2d_args* a) {
  KEEP_REQUIRE(a != nullptr, "keep_conv2d: null args");
  if (a->dtype != KEEP_F32 && !(a->dtype == KEEP_BF16 && a->mma == KEEP_MMA_BF16)) {
    keep_set_error("keep_conv2d: dtype %d not supported (fp32 input, or bf16 input with KEEP_MMA_BF16)", a->dtype);
    return KEEP_EUNSUP;
  }
  KEEP_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0 && a->KH > 0 && a->KW > 0 &&
                   a->stride > 0 && a->Ho > 0 && a->Wo > 0,
               "keep_conv2d: non-positive dimension");
  KEEP_REQUIRE(a->in_ld >= (a->in2 ? a->in2_cin1 : a->Cin) && a->out_ld >= a->Cout, "keep_conv2d: ld smaller than channel count");
  if (a->in2 && a->mma != KEEP_MMA_X3) {
    keep_set_error("keep_conv2d: in2 (K-concatenated input) is a KEEP_MMA_X3 feature");
    return KEEP_EUNSUP;
  }
  KEEP_REQUIRE((a->pro_scale == nullptr) == (a->pro_shift == nullptr), "keep_conv2d: pro_scale/pro_shift must pair");
  KEEP_REQUIRE(!a->aux || a->residual, "keep_conv2d: aux epilogue requires residual");
  KEEP_REQUIRE(!a->residual || a->res_ld >= a->Cout, "keep_conv2d: res_ld smaller than Cout");
  KEEP_REQUIRE(a->split_k >= 0, "keep_conv2d: split_k must be >= 0 (0 = let the library choose)");
  {
    const int Hv = a->upsample ? 2 * a->H : a->H, Wv = a->upsample ? 2 * a->W : a->W;
    KEEP_REQUIRE((long)(a->Ho - 1) * a->stride - a->pad_t < Hv && (long)(a->Wo - 1) * a->stride - a->pad_l < Wv,
                 "keep_conv2d: output extent %dx%d inconsistent with input %dx%d", a->Ho, a->Wo, Hv, Wv);
  }
  KEEP_REQUIRE(a->mma == KEEP_MMA_F32 || a->mma == KEEP_MMA_BF16 || a->mma == KEEP_MMA_X3, "keep_conv2d: bad mma %d", a->mma);
  KEEP_REQUIRE(a->upsample == 0 || a->upsample == 1 || (a->upsample == KEEP_UPSAMPLE_X2_PHASES && a->mma == KEEP_MMA_X3),
               "keep_conv2d: upsample must be 0, 1 or KEEP_UPSAMPLE_X2_PHASES (KEEP_MMA_X3 only), got %d", a->upsample);
  KEEP_REQUIRE(a->pad_mode == KEEP_PAD_ZERO || a->pad_mode == KEEP_PAD_REFLECT, "keep_conv2d: bad pad_mode %d", a->pad_mode);
  if (a->pad_mode == KEEP_PAD_REFLECT) {
    const int Hv = a->upsample ? 2 * a->H : a->H, Wv = a->upsample ? 2 * a->W : a->W;
    if (a->mma == KEEP_MMA_BF16 || a->dtype != KEEP_F32 || a->pad_t != a->pad_l || a->pad_t >= Hv || a->pad_t >= Wv ||
        a->pad_t != a->KH / 2 || a->KH != a->KW) {
      keep_set_error("keep_conv2d: KEEP_PAD_REFLECT needs fp32 tensors, KEEP_MMA_F32 / KEEP_MMA_X3, a square odd kernel and pad_t == pad_l == KH/2 < min(H, W)");
      return KEEP_EUNSUP;
    }
  }
  KEEP_REQUIRE((long)a->N * a->Ho * a->Wo < (1L << 31), "keep_conv2d: M too large");
  if (a->ln_gamma && (a->mma != KEEP_MMA_X3 || a->KH != 1 || a->KW != 1 || a->stride != 1 || a->upsample)) {
    keep_set_error("keep_conv2d: ln_gamma (LayerNorm epilogue) is a feature of the KEEP_MMA_X3 GEMM form (1x1, stride 1)");
    return KEEP_EUNSUP;
  }
  return KEEP_OK;
}

// Pointer-independent geometry checks use the pointers only for their alignment; the plan query passes the same struct
// the launch will see (NULL optional tensors stay NULL), so plan and launch always agree.
static int plan_conv(const keep_conv2d_args* a, ConvP& p, ConvPlan& pl) {
  memset(&pl, 0, sizeof(pl));
  const long M_real = (long)a->N * a->Ho * a->Wo;
  // rows the HEURISTICS below see (tile, split-K): per-image rows x the fixed reference batch under the parity policies,
  // the real row count under the bf16 speed policy; launches and buffer sizes always use the real M (p.M)
  const long M = a->mma == KEEP_MMA_BF16 ? M_real : plan_ref_images(a) * (long)a->Ho * a->Wo;
  p.in = (const float*)a->in;
  p.w = a->weight;
  p.wb = (const unsigned short*)a->weight_bf16;
  p.wx3 = (const unsigned short*)a->weight_x3;
  p.acc_scale = a->x3_acc_scale;
  p.in_amax = a->x3_in_amax;
  p.in2 = (const float*)a->in2;
  p.cin1 = a->in2_cin1;
  p.out_amax = nullptr;
  p.reflect = a->pad_mode == KEEP_PAD_REFLECT ? 1 : 0;
  p.tile_cols = 0;
  p.reverse = 0;
  p.ln_gamma = a->ln_gamma;
  p.ln_beta = a->ln_beta;
  p.ln_eps = a->ln_eps;
  p.kslice_steps = 0;
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
  p.M = (int)M_real;
  p.cchunks = (a->Cin + BK - 1) / BK;
  p.nsteps = a->KH * a->KW * p.cchunks;
  p.in_bf16 = (a->dtype == KEEP_BF16) ? 1 : 0;
  // fast-math activations: bf16 policy always; x3 policy unless KEEP_X3_EXACT_ACT is set (forms of x3 grade, keep_common.h)
  p.fast = (a->mma == KEEP_MMA_BF16 || (a->mma == KEEP_MMA_X3 && !(a->flags & KEEP_CONV_X3_EXACT_ACT))) ? 1 : 0;
  p.out_bf16 = (a->out_dtype == KEEP_BF16) ? 1 : 0;
  p.vec_ok = (a->Cin % 4 == 0 && a->in_ld % 4 == 0 && ((uintptr_t)a->in % 16 == 0)) ? 1 : 0;
  p.vec_epi = (a->Cout % 4 == 0 && a->out_ld % 4 == 0 && (uintptr_t)a->out % 16 == 0 &&
               (!a->residual || (a->res_ld % 4 == 0 && (uintptr_t)a->residual % 16 == 0)) &&
               (!a->aux || (uintptr_t)a->aux % 16 == 0) && (!a->bias || (uintptr_t)a->bias % 16 == 0) &&
               (!a->workspace || (uintptr_t)a->workspace % 16 == 0)) ? 1 : 0;
  p.flatk = 0;
  p.flatk_f32 = 0;
  p.stats = a->stats_out;
  p.stats_P = a->stats_P;
  const bool no_pro = !a->pro_scale && a->pro_act == KEEP_PRO_NONE;
  const bool is33s1 = a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad_t == 1 && a->pad_l == 1;
  const bool same_size = a->Ho == (a->upsample ? 2 * a->H : a->H) && a->Wo == (a->upsample ? 2 * a->W : a->W);
  const bool tileable = (a->Ho % 8 == 0 && a->Wo % 32 == 0) || (a->Ho % 16 == 0 && a->Wo % 16 == 0);
  const bool pro_al = !a->pro_scale || ((uintptr_t)a->pro_scale % 16 == 0 && (uintptr_t)a->pro_shift % 16 == 0);
  const bool epi_al = (a->out_ld % 4 == 0) && ((uintptr_t)a->out % 16 == 0) &&
                      (!a->residual || (a->res_ld % 4 == 0 && (uintptr_t)a->residual % 16 == 0)) &&
                      (!a->aux || (uintptr_t)a->aux % 16 == 0) && (!a->bias || (uintptr_t)a->bias % 16 == 0);
  pl.wide = (a->Ho % 8 == 0 && a->Wo % 32 == 0);
  const int ncb = (a->Cout + 63) / 64;
  int mma = a->mma;
  int auto_split = 1;

  // ---- <= 4 output channels: exact-fp32 VALU kernel in every precision policy
  const bool reflect = a->pad_mode == KEEP_PAD_REFLECT;
  if (a->Cout <= 4 && is33s1 && !reflect && !a->upsample && a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && a->Cin % SC_CH == 0 &&
      a->in_ld % 4 == 0 && (uintptr_t)a->in % 16 == 0 && a->Ho == a->H && a->Wo == a->W && a->Ho % SC_TH == 0 &&
      a->Wo % SC_TW == 0 && !a->residual && !a->aux && a->split_k <= 1 && pro_al && !(a->flags & KEEP_CONV_NO_COUT4)) {
    pl.path = PATH_COUT4;
    pl.split_k = 1;
    snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_cout4_kernel");
    return KEEP_OK;
  }
  // ---- RGB first convolutions, bf16 policy: persistent im2col-in-LDS kernel
  if (mma == KEEP_MMA_BF16 && is33s1 && !a->upsample && a->Cin <= 3 && a->Cout % 4 == 0 && a->Cout >= 32 && a->dtype == KEEP_F32 &&
      a->out_dtype != KEEP_BF16 && a->Ho == a->H && a->Wo == a->W && a->Ho % 8 == 0 && a->Wo % 32 == 0 && no_pro && !a->residual &&
      !a->aux && a->split_k <= 1 && a->out_ld % 4 == 0 && (uintptr_t)a->out % 16 == 0 && (!a->bias || (uintptr_t)a->bias % 16 == 0) &&
      !(a->flags & KEEP_CONV_NO_C3)) {
    pl.path = PATH_C3;
    pl.split_k = 1;
    pl.stats_rows = 64;
    snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_c3_kernel");
    return KEEP_OK;
  }
  // ---- split fp16: halo / gather kernels where the geometry fits, the exact-f32 kernels otherwise (same parity grade)
  if (mma == KEEP_MMA_X3) {
    KEEP_REQUIRE(a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16, "keep_conv2d: KEEP_MMA_X3 takes and writes fp32 tensors");
    const bool have_w = a->weight_x3 != nullptr && (uintptr_t)a->weight_x3 % 16 == 0 && a->x3_acc_scale > 0.f;
    // RGB first convolutions: persistent im2col-in-LDS kernel, weights split on the fly from the fp32 tensor
    if (is33s1 && !reflect && !a->upsample && a->Cin <= 3 && a->Cout % 4 == 0 && a->Cout >= 32 && a->Ho == a->H && a->Wo == a->W &&
        a->Ho % 8 == 0 && a->Wo % 32 == 0 && no_pro && !a->residual && !a->aux && a->split_k <= 1 && a->out_ld % 4 == 0 &&
        (uintptr_t)a->out % 16 == 0 && (!a->bias || (uintptr_t)a->bias % 16 == 0) && a->weight && !(a->flags & KEEP_CONV_NO_C3)) {
      pl.path = PATH_C3_X3;
      pl.split_k = 1;
      pl.stats_rows = 64;
      pl.amax_ok = true;
      snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_c3_x3_kernel");
      return KEEP_OK;
    }
    if (a->upsample == KEEP_UPSAMPLE_X2_PHASES) {      // weight_x3 holds the four phase kernels: only the phase form of the halo kernel can run it
      if (!(have_w && is33s1 && keep_conv_x3_up2_ok(a))) {
        keep_set_error("keep_conv2d: upsample = KEEP_UPSAMPLE_X2_PHASES needs KEEP_MMA_X3 phase weights, a 3x3 stride-1 pad-1 convolution without "
                       "prologue / activation / aux / split-K, H %% 8 == 0, W %% 32 == 0, Cin %% 16 == 0 and Cout %% 64 == 0");
        return KEEP_EUNSUP;
      }
      pl.path = PATH_HALO_X3;
      pl.split_k = 1;
      pl.stats_rows = 256;
      pl.amax_ok = true;
      snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_halo_x3_kernel<32, x2 phases>");
      return KEEP_OK;
    }
    if (have_w && is33s1 && keep_conv_x3_halo_ok(a) && !(a->flags & KEEP_CONV_NO_HALO_X3)) {
      pl.path = PATH_HALO_X3;
      const long items = (M / 256) * ncb;
      auto_split = items >= 256 ? 1 : (int)max(1L, min(min(512L / items, (long)a->Cin / 32), 16L));
      pl.split_k = a->split_k > 0 ? a->split_k : auto_split;
      if (pl.split_k > a->Cin / 16) pl.split_k = a->Cin / 16;
      pl.stats_rows = 256;      // one statistics partial per 256-pixel tile (the block adds its four waves' sums in LDS)
      pl.amax_ok = pl.split_k == 1;
      if (keep_conv_x3_stream_ok(a, p, pl.split_k))      // the streaming form (keep_conv_x3s.hip): what rocprofv3 prints
        snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_halo_x3s_kernel");
      else
        snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_halo_x3_kernel<%d>", pl.wide ? 32 : 16);
      return KEEP_OK;
    }
    if (a->in2) {      // K-concatenated input: GEMM form of the x3 gather kernel only
      const bool ok2 = have_w && keep_conv_x3_gather_ok(a, p) && keep_conv_x3_gather_is_gemm(a) && no_pro && !a->x3_in_amax &&
                       a->in2_cin1 > 0 && a->in2_cin1 < a->Cin && a->in2_cin1 % 32 == 0 && (a->Cin - a->in2_cin1) % 4 == 0 &&
                       a->in_ld >= a->in2_cin1 && (uintptr_t)a->in2 % 16 == 0 && !is33s1 && !(a->flags & KEEP_CONV_NO_GATHER_X3);
      if (!ok2) {
        keep_set_error("keep_conv2d: in2 (K-concatenated input) needs KEEP_MMA_X3, a 1x1 stride-1 convolution without prologue / range probe and in2_cin1 %% 32 == 0");
        return KEEP_EUNSUP;
      }
    }
    // GEMM form with few rows per image (token GEMMs of the frame recurrence): the latency form, K cut into canonical slices
    // (keep_gemm_x3l.hip).  A per-image rule -- the sums it defines are the same at every batch size.
    if (have_w && keep_gemm_x3l_ok(a) && keep_conv_x3_gather_ok(a, p) && keep_conv_x3_gather_is_gemm(a) &&
        !(a->flags & (KEEP_CONV_NO_GATHER_X3 | KEEP_CONV_NO_GEMM_LAT))) {
      pl.path = PATH_GATHER_X3;
      pl.tile = 4;
      pl.plain = no_pro;
      pl.split_k = 1;
      pl.stats_rows = 0;
      pl.amax_ok = true;
      snprintf(pl.kernel, sizeof(pl.kernel), "gemm_x3l_kernel<%d>", keep_gemm_x3l_waves(a));
      return KEEP_OK;
    }
    if (have_w && keep_conv_x3_gather_ok(a, p) && !(a->flags & KEEP_CONV_NO_GATHER_X3)) {
      pl.path = PATH_GATHER_X3;
      pl.tile = (a->Cout <= 64 || M <= small_m_threshold(a)) ? 1 : 2;
      pl.plain = no_pro;
      const int steps = a->KH * a->KW * ((a->Cin + 31) / 32);
      const long blocks = pl.tile == 1 ? (long)cdiv(M, 64) * cdiv(a->Cout, 64) : (long)cdiv(M, 128) * cdiv(a->Cout, 128);
      const long waves = blocks * 4;
      auto_split = (waves >= kTargetWaves || steps < 8) ? 1 : (int)max(1L, min(min(4L * kTargetWaves / waves, (long)steps / 2), 32L));
      pl.split_k = a->split_k > 0 ? a->split_k : auto_split;
      if (pl.split_k > steps) pl.split_k = steps;
      pl.stats_rows = pl.tile == 1 ? 64 : 128;
      // a wave's rows must lie in one image: Ho*Wo a multiple of the wave tile (32 or 64 rows)
      pl.amax_ok = pl.split_k == 1 && ((long)a->Ho * a->Wo) % (pl.tile == 1 ? 32 : 64) == 0;
      if (a->ln_gamma) {      // LayerNorm in the epilogue: one wave holds whole 128-channel rows (tile <4,1,1,4>), full row tiles only
        if (!(a->ln_beta && keep_conv_x3_gather_is_gemm(a) && no_pro && a->Cout == 128 && a->out_ld == 128 && M_real % 128 == 0 &&
              a->split_k <= 1 && a->epi_act == KEEP_ACT_NONE && !a->aux && !a->stats_out && a->out_dtype == KEEP_F32 &&
              (uintptr_t)a->ln_gamma % 16 == 0 && (uintptr_t)a->ln_beta % 16 == 0 && (!a->residual || a->res_ld % 4 == 0))) {
          keep_set_error("keep_conv2d: ln_gamma (LayerNorm epilogue) needs the KEEP_MMA_X3 GEMM form (1x1, stride 1, no prologue / activation / "
                         "aux / split-K / statistics), Cout == out_ld == 128, fp32 output and N*Ho*Wo %% 128 == 0");
          return KEEP_EUNSUP;
        }
        pl.tile = 3;
        pl.split_k = 1;
        pl.stats_rows = 0;
        pl.amax_ok = ((long)a->Ho * a->Wo) % 32 == 0;
        snprintf(pl.kernel, sizeof(pl.kernel), "conv_x3_kernel<4, 1, 1, 4, true, true> + LayerNorm");
        return KEEP_OK;
      }
      snprintf(pl.kernel, sizeof(pl.kernel), "conv_x3_kernel<%s, %s, %s>", pl.tile == 1 ? "2, 2, 1, 1" : "2, 2, 2, 2",
               pl.plain ? "true" : "false", keep_conv_x3_gather_is_gemm(a) ? "true" : "false");
      return KEEP_OK;
    }
    if (a->ln_gamma) {
      keep_set_error("keep_conv2d: ln_gamma (LayerNorm epilogue) is a feature of the KEEP_MMA_X3 GEMM form");
      return KEEP_EUNSUP;
    }
    mma = KEEP_MMA_F32;
  }
  KEEP_REQUIRE(mma == KEEP_MMA_F32 || (a->weight_bf16 && (uintptr_t)a->weight_bf16 % 16 == 0) || a->weight == nullptr,
               "keep_conv2d: KEEP_MMA_BF16 needs a 16-byte aligned weight_bf16");
  // ---- fp32 policy: persistent LDS-halo kernel on f32 MFMA
  if (mma == KEEP_MMA_F32 && a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && is33s1 && (a->Cin % 16 == 0) &&
      (a->Cout % 32 == 0) && tileable && same_size && pro_al && (a->in_ld % 4 == 0) && ((uintptr_t)a->in % 16 == 0) &&
      ((uintptr_t)a->weight % 16 == 0) && epi_al && (!a->workspace || (uintptr_t)a->workspace % 16 == 0) &&
      !(a->flags & KEEP_CONV_NO_HALO_F32)) {
    pl.path = PATH_HALO_F32;
    const long items = (M / 256) * ncb;
    auto_split = items >= 256 ? 1 : (int)max(1L, min(min(512L / items, (long)a->Cin / 32), 16L));
    pl.split_k = a->split_k > 0 ? a->split_k : auto_split;
    if (pl.split_k > a->Cin / 16) pl.split_k = a->Cin / 16;
    pl.stats_rows = 64;
    snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_halo_f32_kernel<%d>", pl.wide ? 32 : 16);
    return KEEP_OK;
  }
  // ---- bf16 policy: LDS-halo kernel (persistent v3; v1 when the prologue is fused into its staging step)
  static const int halo_ver = KEEP_DEV_ENV("KEEP_HALO_VER") ? atoi(KEEP_DEV_ENV("KEEP_HALO_VER")) : 3;
  const bool halo_geom = mma == KEEP_MMA_BF16 && is33s1 && (a->Cin % 32 == 0) && (a->Cout % 32 == 0) && tileable && same_size &&
                         (a->in_ld % 8 == 0) && ((uintptr_t)a->in % 16 == 0) && epi_al;
  if (halo_geom) {
    const bool v3 = halo_ver == 3 && no_pro;
    const bool out16_ok = no_pro && !a->residual && a->split_k <= 1 && a->Cout % 64 == 0 && halo_ver == 3;
    pl.out_bf16_ok = out16_ok ? 1 : 0;
    // an fp32 input with a prologue: the host should run keep_norm_act_bf16 first and come back with a bf16 tensor
    pl.wants_bf16_input = (!no_pro && halo_ver == 3) ? 1 : 0;
    if (p.in_bf16 && !no_pro) {      // a bf16 tensor that still carries a prologue: only the two-pass form exists
      pl.path = PATH_NEEDS_PRENORM;
      pl.split_k = 1;
      snprintf(pl.kernel, sizeof(pl.kernel), "(keep_norm_act_bf16 first)");
      return KEEP_OK;
    }
    const bool ok = (a->dtype == KEEP_F32 || no_pro) && pro_al && (a->out_dtype != KEEP_BF16 || out16_ok) &&
                    (a->Cout % 64 == 0 || v3);
    if (ok) {
      pl.path = v3 ? PATH_HALO_BF16 : PATH_HALO_BF16_V1;
      const long waves = (M / 256) * ncb * 4;
      auto_split = (a->out_dtype == KEEP_BF16 || waves >= kTargetWaves) ? 1
                   : (int)max(1L, min(min((long)kTargetWaves / waves, (long)a->Cin / 64), 16L));
      pl.split_k = a->split_k > 0 ? a->split_k : auto_split;
      if (pl.split_k > a->Cin / 32) pl.split_k = a->Cin / 32;
      pl.stats_rows = v3 ? 64 : 256;
      snprintf(pl.kernel, sizeof(pl.kernel), "conv3x3_halo3_kernel<%s, %d>", p.in_bf16 ? "true" : "false", pl.wide ? 32 : 16);
      return KEEP_OK;
    }
  }
  if (p.in_bf16) {
    keep_set_error("keep_conv2d: bf16 input tensors are only accepted by the 3x3 stride-1 halo path "
                   "(Cin%%32, Cout%%32, tileable map, no prologue)");
    return KEEP_EUNSUP;
  }
  // ---- gather kernels
  pl.tile = a->Cout <= 32 ? 0 : ((a->Cout <= 64 || M <= small_m_threshold(a)) ? 1 : 2);
  const long blocks = pl.tile == 0 ? (long)cdiv(M, 128) * cdiv(a->Cout, 32)
                      : (pl.tile == 1 ? (long)cdiv(M, 64) * cdiv(a->Cout, 64) : (long)cdiv(M, 128) * cdiv(a->Cout, 128));
  const long waves = blocks * 4;
  pl.stats_rows = pl.tile == 1 ? 64 : 128;
  if (mma == KEEP_MMA_BF16) {
    pl.path = PATH_GATHER_BF16;
    p.flatk = (a->Cin < 8 && no_pro) ? 1 : 0;
    pl.plain = !p.flatk && p.vec_ok && no_pro && !a->upsample && (a->Cin % 8 == 0) && !(a->flags & KEEP_CONV_NO_PLAIN);
    int steps = p.flatk ? (a->KH * a->KW * a->Cin + BK16 - 1) / BK16 : a->KH * a->KW * ((a->Cin + BK16 - 1) / BK16);
    pl.bk256 = a->bk256 && !p.flatk && pl.tile == 1;
    if (pl.bk256) {
      steps = a->KH * a->KW * ((a->Cin + 255) / 256);
      auto_split = (int)max(1L, min(min(512L / max(blocks, 1L), (long)steps / 2), 16L));
    } else {
      auto_split = (waves >= kTargetWaves || steps < 8) ? 1 : (int)max(1L, min(min(4L * kTargetWaves / waves, (long)steps / 2), 32L));
    }
    if (a->out_dtype == KEEP_BF16) auto_split = 1;
    pl.out_bf16_ok = (p.vec_epi && !a->residual) ? 1 : 0;
    pl.split_k = a->split_k > 0 ? a->split_k : auto_split;
    if (pl.split_k > steps) pl.split_k = steps;
    const char* t = pl.tile == 0 ? "4, 1, 1, 1" : (pl.tile == 1 ? "2, 2, 1, 1" : "2, 2, 2, 2");
    snprintf(pl.kernel, sizeof(pl.kernel), "conv_bf16_kernel<%s, %s, %s>", t, pl.bk256 ? "256, 1" : "64, 1",
             (pl.plain && !pl.bk256 && pl.tile != 0) ? "true" : "false");
    return KEEP_OK;
  }
  pl.path = PATH_GATHER_F32;
  p.flatk_f32 = (a->Cin < 8 && a->dtype == KEEP_F32 && no_pro && !(a->flags & KEEP_CONV_NO_FLATK_F32)) ? 1 : 0;
  if (p.flatk_f32) p.nsteps = (a->KH * a->KW * a->Cin + BK - 1) / BK;
  pl.plain = p.vec_ok && a->Cin % 16 == 0 && no_pro && !a->upsample && pl.tile != 0 && !(a->flags & KEEP_CONV_NO_PLAIN);
  auto_split = (waves >= kTargetWaves || p.nsteps < 8) ? 1 : (int)max(1L, min(min((long)kTargetWaves / waves, (long)p.nsteps / 4), 32L));
  pl.split_k = a->split_k > 0 ? a->split_k : auto_split;
  if (pl.split_k > p.nsteps) pl.split_k = p.nsteps;
  snprintf(pl.kernel, sizeof(pl.kernel), "conv_f32_kernel<%s>", pl.tile == 0 ? "4, 1, 1, 1" : (pl.tile == 1 ? "2, 2, 1, 1" : "2, 2, 2, 2"));
  return KEEP_OK;
}

// The struct the caller was compiled against may be an older (shorter) layout of this ABI version: `struct_size` says which.
// Sizes the library does not know are rejected; the fields a shorter known layout lacks read as zero.
static int conv_args_in(const keep_conv2d_args* src, keep_conv2d_args& a) {
  KEEP_REQUIRE(src != nullptr, "keep_conv2d: null args");
  const uint32_t sz = src->struct_size;
  if (sz < KEEP_CONV2D_ARGS_V12_SIZE || sz > sizeof(keep_conv2d_args) || sz % 8 != 0) {
    keep_set_error("keep_conv2d: args.struct_size = %u, this library (ABI v%d) accepts %d..%zu -- set it to sizeof(keep_conv2d_args) "
                   "of the header the caller was built with", sz, KEEP_ABI_VERSION, KEEP_CONV2D_ARGS_V12_SIZE, sizeof(keep_conv2d_args));
    return KEEP_EINVAL;
  }
  memset(&a, 0, sizeof(a));
  memcpy(&a, src, sz);
  return KEEP_OK;
}

extern "C" int32_t keep_sizeof_conv2d_args(void) { return (int32_t)sizeof(keep_conv2d_args); }

extern "C" int32_t keep_conv2d_plan(const keep_conv2d_args* a_in, keep_conv2d_plan_out* out) {
  KEEP_REQUIRE(out != nullptr, "keep_conv2d_plan: null output");
  keep_conv2d_args a_local;
  int rc = conv_args_in(a_in, a_local);
  if (rc != KEEP_OK) return rc;
  const keep_conv2d_args* a = &a_local;
  rc = validate_conv(a);
  if (rc != KEEP_OK) return rc;
  ConvP p;
  ConvPlan pl;
  rc = plan_conv(a, p, pl);
  if (rc != KEEP_OK) return rc;
  memset(out, 0, sizeof(*out));
  const long hw_o = (long)a->Ho * a->Wo;
  out->split_k = pl.split_k;
  out->workspace_bytes = pl.split_k > 1 ? (int64_t)pl.split_k * a->N * hw_o * a->Cout * 4 : 0;
  // statistics ride on the epilogue of a single-pass launch whose tiles never straddle two images
  out->stats_rows = (pl.split_k == 1 && pl.stats_rows > 0 && hw_o % pl.stats_rows == 0 && a->out_ld == a->Cout) ? pl.stats_rows : 0;
  out->stats_P = out->stats_rows ? (int32_t)(hw_o / out->stats_rows) : 0;
  out->wants_bf16_input = pl.wants_bf16_input;
  out->out_bf16_ok = pl.out_bf16_ok;
  out->out_amax_ok = pl.amax_ok ? 1 : 0;
  out->path = (int32_t)pl.path;
  strncpy(out->kernel, pl.kernel, sizeof(out->kernel) - 1);
  return KEEP_OK;
}

extern "C" int32_t keep_conv2d(const keep_conv2d_args* a_in, void* stream) {
  keep_conv2d_args a_local;
  int rc = conv_args_in(a_in, a_local);
  if (rc != KEEP_OK) return rc;
  const keep_conv2d_args* a = &a_local;
  rc = validate_conv(a);
  if (rc != KEEP_OK) return rc;
  KEEP_REQUIRE(a->in && a->weight && a->out, "keep_conv2d: null tensor pointer");
  KEEP_REQUIRE((uintptr_t)a->weight % 16 == 0, "keep_conv2d: weight pointer must be 16-byte aligned");
  KEEP_REQUIRE(!a->pro_scale || ((uintptr_t)a->pro_scale % 16 == 0 && (uintptr_t)a->pro_shift % 16 == 0),
               "keep_conv2d: pro_scale/pro_shift must be 16-byte aligned");
  ConvP p;
  ConvPlan pl;
  rc = plan_conv(a, p, pl);
  if (rc != KEEP_OK) return rc;
  if (pl.path == PATH_NEEDS_PRENORM) {
    keep_set_error("keep_conv2d: a bf16 input with a prologue must go through keep_norm_act_bf16 first (keep_conv2d_plan: wants_bf16_input)");
    return KEEP_EUNSUP;
  }
  p.split_k = pl.split_k;
  KEEP_REQUIRE(p.split_k == 1 || a->workspace, "keep_conv2d: split_k>1 requires a workspace (keep_conv2d_plan gives its size)");
  const long M = p.M;
  const long hw_o = (long)a->Ho * a->Wo;
  if (p.stats) {
    KEEP_REQUIRE(p.split_k == 1, "keep_conv2d: stats_out requires split_k == 1");
    KEEP_REQUIRE(pl.stats_rows > 0 && hw_o % pl.stats_rows == 0 && a->stats_P == hw_o / pl.stats_rows,
                 "keep_conv2d: stats_P=%d must equal Ho*Wo/%d (keep_conv2d_plan)", a->stats_P, pl.stats_rows);
  }
  hipStream_t st = (hipStream_t)stream;
  if (a->x3_out_amax) {
    KEEP_REQUIRE(pl.amax_ok, "keep_conv2d: x3_out_amax is not available for this call (keep_conv2d_plan: out_amax_ok)");
    if (!a->x3_out_amax_zeroed) {
      hipLaunchKernelGGL(zero_u32_kernel, dim3(cdiv(a->N, 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(a->x3_out_amax), a->N);
      KEEP_LAUNCH_CHECK("keep_conv2d(zero x3_out_amax)");
    }
    p.out_amax = reinterpret_cast<unsigned*>(a->x3_out_amax);
  }
  if (p.out_bf16)
    KEEP_REQUIRE(p.vec_epi && p.split_k == 1 && !a->residual,
                 "keep_conv2d: bf16 output needs Cout/out_ld %% 4 == 0, aligned pointers, split_k == 1, no residual");
  dim3 block(256);
  const int tw = pl.wide ? 32 : 16, th = 256 / tw;
  const int tiles_x = a->Wo / tw, tiles_y = a->Ho / th, ncb = (a->Cout + 63) / 64;
  const int n_cu = n_cu_cached();
  switch (pl.path) {
    case PATH_COUT4: {
      dim3 grid((a->Ho / SC_TH) * (a->Wo / SC_TW), a->N);
      hipLaunchKernelGGL(conv3x3_cout4_kernel, grid, block, 0, st, p);
      KEEP_LAUNCH_CHECK("keep_conv2d(cout<=4)");
      return KEEP_OK;
    }
    case PATH_C3: {
      const int tx = a->Wo / 32, ty = a->Ho / 8;
      const int n_items = a->N * tx * ty * ncb;
      hipLaunchKernelGGL(conv3x3_c3_kernel, dim3(n_items < 2 * n_cu ? n_items : 2 * n_cu), block, 0, st, p, tx, ty, ncb, n_items);
      KEEP_LAUNCH_CHECK("keep_conv2d(Cin<=3)");
      return KEEP_OK;
    }
    case PATH_C3_X3:
      rc = keep_conv2d_x3_c3(a, p, st);
      if (rc != KEEP_OK) return rc;
      break;
    case PATH_HALO_X3:
      rc = keep_conv2d_x3_halo(a, p, st);
      if (rc != KEEP_OK) return rc;
      break;
    case PATH_GATHER_X3:
      rc = pl.tile == 4 ? keep_conv2d_x3_gemm_lat(a, p, st) : keep_conv2d_x3_gather(a, p, pl.tile, st);
      if (rc != KEEP_OK) return rc;
      break;
    case PATH_HALO_F32: {
      const int n_items = a->N * tiles_x * tiles_y * ncb * p.split_k;
      dim3 gridf(n_items < 2 * n_cu ? n_items : 2 * n_cu);
#define KEEP_LAUNCH_HF(TWV)                                                                                                  \
  if (a->pro_act == KEEP_PRO_SWISH)                                                                                          \
    hipLaunchKernelGGL((conv3x3_halo_f32_kernel<TWV, KEEP_PRO_SWISH>), gridf, block, 0, st, p, tiles_x, tiles_y, ncb, n_items); \
  else if (a->pro_act == KEEP_PRO_RELU)                                                                                      \
    hipLaunchKernelGGL((conv3x3_halo_f32_kernel<TWV, KEEP_PRO_RELU>), gridf, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);  \
  else                                                                                                                       \
    hipLaunchKernelGGL((conv3x3_halo_f32_kernel<TWV, KEEP_PRO_NONE>), gridf, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
      if (pl.wide) {
        KEEP_LAUNCH_HF(32)
      } else {
        KEEP_LAUNCH_HF(16)
      }
#undef KEEP_LAUNCH_HF
      KEEP_LAUNCH_CHECK("keep_conv2d(halo f32)");
      break;
    }
    case PATH_HALO_BF16: {
      const int n_items = a->N * tiles_x * tiles_y * ncb * p.split_k;
      dim3 grid3(n_items < 2 * n_cu ? n_items : 2 * n_cu);
      const bool simple = p.split_k == 1 && !a->aux && a->epi_act == KEEP_ACT_NONE;
#define KEEP_LAUNCH_H3(INB, TWV)                                                                                          \
  if (simple)                                                                                                             \
    hipLaunchKernelGGL((conv3x3_halo3_kernel<INB, TWV, true>), grid3, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);   \
  else                                                                                                                    \
    hipLaunchKernelGGL((conv3x3_halo3_kernel<INB, TWV, false>), grid3, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
      if (p.in_bf16 && pl.wide) {
        KEEP_LAUNCH_H3(true, 32)
      } else if (p.in_bf16) {
        KEEP_LAUNCH_H3(true, 16)
      } else if (pl.wide) {
        KEEP_LAUNCH_H3(false, 32)
      } else {
        KEEP_LAUNCH_H3(false, 16)
      }
#undef KEEP_LAUNCH_H3
      KEEP_LAUNCH_CHECK("keep_conv2d(halo v3)");
      break;
    }
    case PATH_HALO_BF16_V1: {
      dim3 grid(a->N * tiles_x * tiles_y * ncb, 1, p.split_k);
      if (p.in_bf16 && pl.wide)
        hipLaunchKernelGGL((conv3x3_halo_kernel<true, 32>), grid, block, 0, st, p, tiles_x, tiles_y, ncb);
      else if (p.in_bf16)
        hipLaunchKernelGGL((conv3x3_halo_kernel<true, 16>), grid, block, 0, st, p, tiles_x, tiles_y, ncb);
      else if (pl.wide)
        hipLaunchKernelGGL((conv3x3_halo_kernel<false, 32>), grid, block, 0, st, p, tiles_x, tiles_y, ncb);
      else
        hipLaunchKernelGGL((conv3x3_halo_kernel<false, 16>), grid, block, 0, st, p, tiles_x, tiles_y, ncb);
      KEEP_LAUNCH_CHECK("keep_conv2d(halo v1)");
      break;
    }
    case PATH_GATHER_BF16: {
      if (pl.tile == 0) {
        dim3 grid(cdiv(M, 128), cdiv(a->Cout, 32), p.split_k);
        hipLaunchKernelGGL((conv_bf16_kernel<4, 1, 1, 1, 64, 1>), grid, block, 0, st, p);
      } else if (pl.tile == 1) {
        dim3 grid(cdiv(M, 64), cdiv(a->Cout, 64), p.split_k);
        if (pl.bk256)
          hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 1, 1, 256, 1>), grid, block, 0, st, p);
        else if (pl.plain)
          hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 1, 1, 64, 1, true>), grid, block, 0, st, p);
        else
          hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 1, 1, 64, 1>), grid, block, 0, st, p);
      } else {
        dim3 grid(cdiv(M, 128), cdiv(a->Cout, 128), p.split_k);
        if (pl.plain)
          hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 2, 2, 64, 1, true>), grid, block, 0, st, p);
        else
          hipLaunchKernelGGL((conv_bf16_kernel<2, 2, 2, 2, 64, 1>), grid, block, 0, st, p);
      }
      KEEP_LAUNCH_CHECK("keep_conv2d(gather bf16)");
      break;
    }
    case PATH_GATHER_F32: {
      if (pl.tile == 0) {
        dim3 grid(cdiv(M, 128), cdiv(a->Cout, 32), p.split_k);
        hipLaunchKernelGGL((conv_f32_kernel<4, 1, 1, 1>), grid, block, 0, st, p);
      } else if (pl.tile == 1) {
        dim3 grid(cdiv(M, 64), cdiv(a->Cout, 64), p.split_k);
        if (pl.plain)
          hipLaunchKernelGGL((conv_f32_kernel<2, 2, 1, 1, true>), grid, block, 0, st, p);
        else
          hipLaunchKernelGGL((conv_f32_kernel<2, 2, 1, 1>), grid, block, 0, st, p);
      } else {
        dim3 grid(cdiv(M, 128), cdiv(a->Cout, 128), p.split_k);
        if (pl.plain)
          hipLaunchKernelGGL((conv_f32_kernel<2, 2, 2, 2, true>), grid, block, 0, st, p);
        else
          hipLaunchKernelGGL((conv_f32_kernel<2, 2, 2, 2>), grid, block, 0, st, p);
      }
      KEEP_LAUNCH_CHECK("keep_conv2d(gather f32)");
      break;
    }
    case PATH_NEEDS_PRENORM:
      break;
  }
  if (p.split_k > 1) {
    const long total = M * a->Cout;
    if (p.vec_epi) {
      int blocks = cdiv(total / 4, 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(conv_splitk_reduce4_kernel, dim3(blocks), dim3(256), 0, st, p);
      KEEP_LAUNCH_CHECK("keep_conv2d(split-K reduce)");
      return KEEP_OK;
    }
    int blocks = cdiv(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
    KEEP_LAUNCH_CHECK("keep_conv2d(split-K reduce)");
  }
  return KEEP_OK;
}
