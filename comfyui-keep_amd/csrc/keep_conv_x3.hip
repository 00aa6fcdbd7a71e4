// keep_conv2d, KEEP_MMA_X3: the parity-grade fast policy -- split-operand fp16 on the 2.5 PFLOP/s matrix pipe.
//
// gfx950 has no TF32-like mode: exact-f32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the 16-bit rate (157 TF).  Here every
// fp32 operand x is written as x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (both RNE; x - hi is exact in fp32), and a
// product a*b is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with three v_mfma_f32_32x32x16_f16 into ONE fp32
// accumulator (fp16 x fp16 products are exact in fp32; accumulation is the matrix pipe's fp32).  hi carries 11 significand
// bits, lo the next 11: the representation error is <= 2^-22 |x| (or 2^-25 absolute once lo is subnormal, |x| < 2^-3), the
// dropped a_lo*b_lo term is <= 2^-22 |a b| -- fp32-grade products at 3 MFMAs instead of 16: ceiling 2.5 PF / 3 = 833 TF.
// (bf16 halves would keep fp32's exponent range but only 8+8 bits: 2^-16 per product, 60x worse than this.)
//   Range: fp16 tops out at 65504.  Weights are pre-multiplied by a power of two 2^e on the host (exact) so that the
//   largest one sits just below 2^15 and the small ones keep a normal `lo`; the accumulators are multiplied by 2^-e
//   (`acc_scale`, exact) before bias / activation.  Activations are split as they are (post-GroupNorm values are O(1));
//   an activation beyond 65504 becomes inf and propagates to the output, which the host checks (engine/net.py) and
//   re-runs on the exact-f32 kernels -- never silently wrong.
//   MFMA f16 on gfx950 keeps subnormal inputs (tests/test_gpu_kernels.py::test_x3_subnormal_lo pins that).
// Weight layout (host-packed, engine/weights.py:split_x3): [Cout][KH*KW][Cin/16][hi x16 | lo x16] fp16 -- the 64 bytes a
// (cout, tap, 16-channel chunk) row needs are contiguous, 4 bytes per weight like the fp32 blob.
#include <stdlib.h>

#include "keep_conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const float (&v)[4], f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)(v[j] - (float)h);
  }
}

// x * sigmoid(x) to ~1 ulp in half the instructions of expf + an IEEE division (this runs 24 times per thread and
// 16-channel chunk in the halo staging step): exp(-x) = 2^n * 2^f with the product -x*log2(e) carried in two floats
// (v_exp_f32 is only accurate to an ulp for small |f|), then 1/(1+e) = v_rcp_f32 + one Newton step.
__device__ __forceinline__ float swish_x3(float x) {
  const float nx = -x;
  const float t = nx * 1.44269504088896341f;
  const float tl = fmaf(nx, 1.44269504088896341f, -t) + nx * 1.92596299112661746e-8f;
  const float n = rintf(t);
  float e = __builtin_amdgcn_exp2f((t - n) + tl);
  e = ldexpf(e, (int)n);                       // x << 0: e = inf -> sigmoid 0;  x >> 0: e = 0 -> sigmoid 1
  const float d = 1.0f + e;
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  return (d < 3.0e38f) ? x * r : 0.0f * x;     // (inf * 0 would be NaN in the Newton step; 0*x keeps NaN inputs NaN)
}

template <int PRO>
__device__ __forceinline__ float pro_x3(float v) {
  if (PRO == KEEP_PRO_SWISH) return swish_x3(v);
  if (PRO == KEEP_PRO_RELU) return v > 0.f ? v : 0.f;
  return v;
}

// sum over the 16 lanes of a DPP row, result in every lane of the row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror,
// row_mirror (each a single v_add_f32 with a DPP modifier)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

#define MMA_X3(ACC, AH, AL, BH, BL)                                              \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);            \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);            \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);

// ------------------------------------------------------------------------------------------------ 3x3 halo, split fp16
// The persistent LDS-halo kernel (keep_conv.hip: conv3x3_halo3_kernel / conv3x3_halo_f32_kernel) with split operands:
// a block walks (8x32 | 16x16 pixel tile) x 64-cout work items; per 16-channel chunk it stages the (8+2)x(32+2) fp32 halo ONCE --
// GroupNorm affine + exact swish applied, then split -- as rows [hi x16 | lo x16 | pad] at an 80-byte pitch (ds_read_b128
// conflict-free), plus the 9 x 64 weight rows in the same format, and all 9 taps read them from LDS:
//   per tap and wave: 4 A + 4 B fragment reads (hi, lo of 2 pixel blocks / 2 cout blocks) feed 12 MFMAs -- 0.67 LDS reads
//   per MFMA against 1.0 in the plain bf16 kernel -- and a chunk costs the same 64 B per pixel of global traffic as a
//   32-channel bf16 chunk but 1.5x the matrix time: the x3 kernel is further from the L2->CU and LDS limits than the
//   bf16 kernel by construction.  73 KB of LDS -> 2 blocks per CU: one block's staging VALU (24 swish + split per thread
//   and chunk) overlaps the other's 108 MFMAs per wave.
#define XPITCH 40   // fp16 elements per LDS row: 16 hi + 16 lo + 8 pad (80 B)

// EXP (dev builds with -DKEEP_X3_ABLATE only, 0 in the product): phase ablations -- 1: no LDS fragment reads in the MFMA loop,
// 2: no MFMAs, 3: no staging (LDS keeps stale data), 4: no global stores in the epilogue, 5: no operand fetch.
template <int TW, int PRO, bool SIMPLE_EPI, int EXP = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_x3_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HALO_TH = 256 / TW, HALO_W = TW + 2, HALO_PIX = (HALO_TH + 2) * HALO_W;
  constexpr int RPT = 32 / TW;
  constexpr int MAIN_B = (HALO_MAXPIX + 9 * 64) * XPITCH * 2;
  constexpr int EPI_B = 4 * 64 * 68 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[MAIN_B > EPI_B ? MAIN_B : EPI_B];
  _Float16* Hs = reinterpret_cast<_Float16*>(lds_raw);
  _Float16* Ws = Hs + HALO_MAXPIX * XPITCH;

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
  float in_s = 1.f, in_inv = 1.f;        // range scale of the item being FETCHED / staged (image it.n)
  auto setup = [&](const HaloItem& it) {
    if (p.in_amax) x3_range_scale(p.in_amax[it.n], in_s, in_inv);
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      h_off[k] = -1;
      if (hp < HALO_PIX) {
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = it.oy0 - 1 + hy, ix = it.ox0 - 1 + hx;
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
          const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          h_off[k] = (sy * p.W + sx) * p.in_ld + g * 4;
        }
      }
    }
    img_off = (long)it.n * p.H * p.W * p.in_ld;
    sc_off = (long)it.n * p.Cin + g * 4;
    w_ok = (it.n0 + (tid >> 2)) < p.Cout;
    w_base = w_ok ? ((long)(it.n0 + (tid >> 2)) * 9) * p.Cin * 2 + g * 8 : 0;   // fp16 elements; + (tap*Cin + c0)*2
  };

  float4 hreg[HALO_IT];
  uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int ch) {
    const int c0 = ch << 4;
    if (EXP == 5 && ch > 0) return;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      hreg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h_off[k] >= 0) hreg[k] = *reinterpret_cast<const float4*>(p.in + img_off + h_off[k] + c0);
    }
#define KEEP_WLOADX(TAP, R) R = w_ok ? *reinterpret_cast<const uint4*>(p.wx3 + w_base + ((long)(TAP) * p.Cin + c0) * 2) : make_uint4(0u, 0u, 0u, 0u);
    KEEP_TAPS(KEEP_WLOADX)
#undef KEEP_WLOADX
    if (p.pro_scale) {
      sc4 = *reinterpret_cast<const float4*>(p.pro_scale + sc_off + c0);
      sh4 = *reinterpret_cast<const float4*>(p.pro_shift + sc_off + c0);
    }
  };
  auto stage = [&]() {
    if (EXP == 3) return;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      if (hp < HALO_PIX) {
        float v[4] = {hreg[k].x, hreg[k].y, hreg[k].z, hreg[k].w};
        if (has_pro && h_off[k] >= 0) {      // zero padding applies to the normalised + activated tensor
          v[0] = pro_x3<PRO>(v[0] * sc4.x + sh4.x);
          v[1] = pro_x3<PRO>(v[1] * sc4.y + sh4.y);
          v[2] = pro_x3<PRO>(v[2] * sc4.z + sh4.z);
          v[3] = pro_x3<PRO>(v[3] * sc4.w + sh4.w);
        }
        if (p.in_amax) {
          v[0] *= in_s; v[1] *= in_s; v[2] *= in_s; v[3] *= in_s;
        }
        f16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<f16x4*>(&Hs[hp * XPITCH + g * 4]) = hi;
        *reinterpret_cast<f16x4*>(&Hs[hp * XPITCH + 16 + g * 4]) = lo;
      }
    }
#define KEEP_WSTOREX(TAP, R) *reinterpret_cast<uint4*>(&Ws[((TAP) * 64 + (tid >> 2)) * XPITCH + g * 8]) = R;
    KEEP_TAPS(KEEP_WSTOREX)
#undef KEEP_WSTOREX
  };

  f32x16 acc[2][2];
  const int a_base = (((2 * wave) * RPT + l31 / TW) * HALO_W + (l31 % TW)) * XPITCH + lhi * 8;
  const int b_base = l31 * XPITCH + lhi * 8;
  auto mma = [&]() {
    f16x8 ah[2], al[2], bh[2], bl[2];
    if (EXP == 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(&Hs[a_base + i * XPITCH]);
        al[i] = *reinterpret_cast<const f16x8*>(&Hs[a_base + i * XPITCH + 16]);
        bh[i] = *reinterpret_cast<const f16x8*>(&Ws[b_base + i * 32 * XPITCH]);
        bl[i] = *reinterpret_cast<const f16x8*>(&Ws[b_base + i * 32 * XPITCH + 16]);
      }
    }
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if (EXP != 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const _Float16* src = &Hs[a_base + ((i * RPT + kh) * HALO_W + kw) * XPITCH];
            ah[i] = *reinterpret_cast<const f16x8*>(src);
            al[i] = *reinterpret_cast<const f16x8*>(src + 16);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const _Float16* src = &Ws[b_base + ((kh * 3 + kw) * 64 + j * 32) * XPITCH];
            bh[j] = *reinterpret_cast<const f16x8*>(src);
            bl[j] = *reinterpret_cast<const f16x8*>(src + 16);
          }
        }
        if (EXP == 2) {      // keep the reads alive without the matrix pipe
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            acc[i][0][0] += (float)ah[i][0] + (float)al[i][0];
            acc[i][1][0] += (float)bh[i][0] + (float)bl[i][0];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              MMA_X3(acc[i][j], ah[i], al[i], bh[j], bl[j])
            }
        }
      }
    }
  };
  auto epilogue_t = [&](const HaloItem& it, float item_inv, auto res_c) {
    constexpr bool HAS_RES = decltype(res_c)::value;
    constexpr int EP = 68;
    float* et = reinterpret_cast<float*>(lds_raw) + wave * 64 * EP;
    const float asc = p.acc_scale * item_inv;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r] * asc;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int co = it.n0 + c4;
    const bool cok = co < p.Cout;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && p.split_k == 1 && cok) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
    constexpr int UNR = HAS_RES ? 16 : 8;
#pragma unroll UNR
    for (int q16 = 0; q16 < 16; ++q16) {
      if (!cok) break;
      const int px = q16 * 4 + prow;
      const int oy = it.oy0 + (2 * wave + (px >> 5)) * RPT + (px & 31) / TW;
      const long m = ((long)it.n * p.Ho + oy) * p.Wo + it.ox0 + (px & 31) % TW;
      const float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
      if (!SIMPLE_EPI && p.split_k > 1) {
        *reinterpret_cast<float4*>(p.ws + ((long)it.z * p.M + m) * p.Cout + co) = v;
        continue;
      }
      float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
      if (!SIMPLE_EPI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = act_apply(e[q], p.epi_act);
      }
      if (HAS_RES) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
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
      if (EXP != 4 || e[0] == 1.2345e-30f) *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
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

  int item = blockIdx.x;
  if (item >= n_items) return;
  if (EXP == 6 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(54);
  if (EXP == 7 && blockIdx.x >= gridDim.x / 2) {
    __builtin_amdgcn_s_sleep(100);
  }
  if (EXP == 8 && (blockIdx.x & 1)) __builtin_amdgcn_s_sleep(54);
  HaloItem cur = halo_decode<TW, 4>(p, item, items_per_z, tiles_x, tiles_y, ncb);
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
    const float cur_inv = in_inv;                               // setup(nxt) below moves in_s / in_inv on to the next item
    HaloItem nxt = cur;
    if (has_next) {
      nxt = halo_decode<TW, 4>(p, next_item, items_per_z, tiles_x, tiles_y, ncb);
      setup(nxt);
      if (nxt.ch_begin < nxt.ch_end) fetch(nxt.ch_begin);     // in flight during the epilogue below
    }
    if (p.res)
      epilogue_t(cur, cur_inv, std::true_type{});
    else
      epilogue_t(cur, cur_inv, std::false_type{});
    if (!has_next) break;
    __syncthreads();
    item = next_item;
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------ 3x3 halo, pipelined
// conv3x3_halo_x3_kernel above runs [stage | barrier | MFMA | barrier] per 16-channel chunk with two blocks per CU meant to
// cover each other's staging.  Measured (128 ch @256^2, N=4): 249 us = 93 us of MFMAs + 156 us of everything else -- the
// phases ADD: the matrix pipe idles whenever a block stages (ablations in DESIGN.md), and staggering the blocks changes
// nothing.  What overlaps on CDNA4 is work INSIDE one wave's instruction stream: an MFMA occupies the matrix pipe for 32
// cycles and the wave keeps issuing ~5 other instructions under it.  This kernel is built around that:
//   * ONE 4-wave block per CU (one wave per SIMD, up to 512 VGPRs each), two LDS buffers (2 x 73 KB);
//   * a flat software pipeline over (item, chunk) steps: while step s multiplies from buffer s&1, the SAME waves convert +
//     write the operands of step s+1 into the other buffer and issue the global loads of step s+2, one piece per tap
//     (9 taps x 12 MFMAs per step: piece t is staged right before the MFMAs of tap t and its registers are re-loaded
//     right after), so every staging instruction sits in an MFMA shadow and loads have a whole step to land;
//   * one raw s_barrier per step (LDS hand-off only: no vmcnt drain -- the loads of step s+2 stay in flight across it);
//   * operand roles swapped (A = weights, B = pixels): a lane's accumulators are 4 consecutive couts of ONE pixel, so the
//     epilogue stores 16-byte pieces straight from registers -- no LDS round trip, no extra barrier.
// Used for wide maps (8x32-pixel tiles) without split-K when there are at least two items per CU; everything else stays on
// the kernel above.
template <int PRO, bool SIMPLE_EPI, int SCHED = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_halo_x3p_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HALO_W = 34, HALO_PIX = 340;
  constexpr int HBUF = (HALO_MAXPIX + 1) * XPITCH;     // fp16 elements: halo rows of one buffer (+ 1 dummy row for idle lanes)
  constexpr int BUF = HBUF + 9 * 64 * XPITCH;          // one buffer: [halo | weights]
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  _Float16* const lds0 = reinterpret_cast<_Float16*>(lds_dyn);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int g = tid & 3;
  const int nch = p.Cin >> 4;
  const bool has_pro = p.pro_scale != nullptr || PRO != KEEP_PRO_NONE;

  // ---- pipeline positions (plain scalars, no aggregates: they must live in registers):
  //   f_* : item / chunk being FETCHED (step s+2), s_* : being STAGED (s+1), c_* : being MULTIPLIED (s); ch < 0 = none
  int f_ch = -1, f_n = 0, f_oy0 = 0, f_ox0 = 0, f_n0 = 0, f_tile = 0;
  unsigned f_mask = 0u;          // bit k: halo piece k of this thread is inside the image
  bool f_wok = false;            // this thread's weight row exists (Cout % 64 == 32: last cout block half empty)
  float f_ins = 1.f, f_inv = 1.f;
  int s_ch = -1, s_n = 0, s_oy0 = 0, s_ox0 = 0, s_n0 = 0, s_tile = 0;
  unsigned s_mask = 0u;
  bool s_wok = false;
  float s_ins = 1.f, s_inv = 1.f;
  int c_ch = -1, c_n = 0, c_oy0 = 0, c_ox0 = 0, c_n0 = 0, c_tile = 0;
  float c_inv = 1.f;
  int f_item = blockIdx.x;
  int h_off0 = 0, h_off1 = 0, h_off2 = 0, h_off3 = 0, h_off4 = 0, h_off5 = 0;
  long img_off = 0, w_base = 0, sc_off = 0;
  const int Hv = p.upsample ? 2 * p.H : p.H, Wv = p.upsample ? 2 * p.W : p.W;
#define KEEP_P_HOFF(K, DST)                                                                  \
  {                                                                                          \
    DST = 0;                                                                                 \
    const int hp = (tid >> 2) + (K) * 64;                                                    \
    if (hp < HALO_PIX) {                                                                     \
      const int hy = hp / HALO_W, hx = hp - hy * HALO_W;                                     \
      const int iy = f_oy0 - 1 + hy, ix = f_ox0 - 1 + hx;                                    \
      if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {                                        \
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;        \
        DST = (sy * p.W + sx) * p.in_ld + g * 4;                                             \
        f_mask |= 1u << (K);                                                                 \
      }                                                                                      \
    }                                                                                        \
  }
#define KEEP_P_DECODE()                                                                      \
  {                                                                                          \
    f_ch = -1; f_mask = 0u; f_wok = false; f_ins = 1.f; f_inv = 1.f;                         \
    h_off0 = h_off1 = h_off2 = h_off3 = h_off4 = h_off5 = 0;                                 \
    img_off = 0; w_base = 0; sc_off = 0;                                                     \
    if (f_item < n_items) {                                                                  \
      const HaloItem it = halo_decode<32, 4>(p, f_item, n_items, tiles_x, tiles_y, ncb);     \
      f_ch = 0;                                                                              \
      f_n = it.n; f_oy0 = it.oy0; f_ox0 = it.ox0; f_n0 = it.n0; f_tile = it.ty * tiles_x + it.tx; \
      if (p.in_amax) x3_range_scale(p.in_amax[it.n], f_ins, f_inv);                          \
      KEEP_P_HOFF(0, h_off0) KEEP_P_HOFF(1, h_off1) KEEP_P_HOFF(2, h_off2)                   \
      KEEP_P_HOFF(3, h_off3) KEEP_P_HOFF(4, h_off4) KEEP_P_HOFF(5, h_off5)                   \
      img_off = (long)it.n * p.H * p.W * p.in_ld;                                            \
      sc_off = (long)it.n * p.Cin + g * 4;                                                   \
      f_wok = (it.n0 + (tid >> 2)) < p.Cout;                                                 \
      w_base = f_wok ? ((long)(it.n0 + (tid >> 2)) * 9) * p.Cin * 2 + g * 8 : 0;             \
    }                                                                                        \
  }
  KEEP_P_DECODE()

  // ---- staging registers: halo piece k, weight row of tap t, prologue affine of the chunk.
  // The global loads are issued through `asm volatile` and awaited with explicit counted s_waitcnt: left to hipcc, all 17
  // loads of a step sink to the end of the loop body and are drained with vmcnt(0) before the next step starts (measured:
  // 11 k cycles per step instead of 3.5 k).  FIFO discipline: every step issues exactly KEEP_P_NLOADS loads in a fixed
  // order (sc, sh, then per tap: halo piece, weight row) and every register is re-loaded right after its last use, so
  // whenever a register is consumed at most KEEP_P_NLOADS - 2 younger loads are outstanding -> `s_waitcnt vmcnt(15)`.
  // (Loads / stores the compiler issues itself in the epilogue are younger still: they can only make a wait longer.)
  typedef float pf32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
#define KEEP_P_GLOAD(R, PTR) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(R) : "v"(PTR))
#define KEEP_P_GWAIT2(R0, R1) asm volatile("s_waitcnt vmcnt(15)" : "+v"(R0), "+v"(R1))
#define KEEP_P_GWAIT1(R0) asm volatile("s_waitcnt vmcnt(15)" : "+v"(R0))
  pf32x4 hr0, hr1, hr2, hr3, hr4, hr5;
  pu32x4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
  pf32x4 sc_f = {1.f, 1.f, 1.f, 1.f}, sh_f = {0.f, 0.f, 0.f, 0.f};      // chunk being fetched
  pf32x4 sc_s = sc_f, sh_s = sh_f;                                        // chunk being staged
  hr0 = hr1 = hr2 = hr3 = hr4 = hr5 = sh_f;
  wr0 = wr1 = wr2 = wr3 = wr4 = wr5 = wr6 = wr7 = wr8 = (pu32x4){0u, 0u, 0u, 0u};

  f32x16 acc[2][2];            // [cout block i][pixel block j]: lane = pixel l31 of row 2*wave+j, regs = couts
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int a_base = ((2 * wave) * HALO_W + l31) * XPITCH + lhi * 8;       // pixel fragment: + ((j + kh)*HALO_W + kw)*XPITCH
  const int b_base = l31 * XPITCH + lhi * 8;                               // weight fragment: + (tap*64 + i*32)*XPITCH
  const int w_dst = (tid >> 2) * XPITCH + g * 8;                           // weight row of this thread: + tap*64*XPITCH

  f16x8 whA[2], wlA[2], phA[2], plA[2], whB[2], wlB[2], phB[2], plB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    whA[i] = *reinterpret_cast<const f16x8*>(&lds0[HBUF + b_base + i * 32 * XPITCH]);
    wlA[i] = whA[i];
    phA[i] = *reinterpret_cast<const f16x8*>(&lds0[a_base + i * HALO_W * XPITCH]);
    plA[i] = phA[i];
  }
  int buf = 0;
  while (true) {
    _Float16* const Hc = lds0 + buf * BUF;                 // operands of step s
    _Float16* const Wc = Hc + HBUF;
    _Float16* const Hn = lds0 + (buf ^ 1) * BUF;           // being written: step s+1
    _Float16* const Wn = Hn + HBUF;
    if (c_ch == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    // the prologue affine of the chunk staged in this step was fetched one step ago; all addresses are valid even when there
    // is nothing to fetch / stage (offset 0 of the tensors, a dummy LDS row): the step body is branch-free
    if (SCHED != 3) KEEP_P_GWAIT2(sc_f, sh_f);
    sc_s = sc_f;
    sh_s = sh_f;
    const int fc0 = (f_ch < 0 ? 0 : f_ch) << 4;
    if (SCHED != 3) {   // always two loads (a dummy address without a prologue tensor): the FIFO depth per step is a constant
      const float* ps = p.pro_scale ? p.pro_scale + sc_off + fc0 : p.in;
      const float* ph = p.pro_scale ? p.pro_shift + sc_off + fc0 : p.in;
      KEEP_P_GLOAD(sc_f, ps);
      KEEP_P_GLOAD(sh_f, ph);
    }
    const float* const fin = p.in + img_off + fc0;
    const unsigned short* const fw = p.wx3 + w_base + (long)fc0 * 2;
    const unsigned wmask = s_wok ? 0xffffffffu : 0u;        // (a select on the whole uint4 goes through scratch)
#define KEEP_P_STAGE_H(K, HR)                                                                       \
  {                                                                                                 \
    const int hp = (tid >> 2) + (K) * 64;                                                           \
    float v[4] = {HR[0], HR[1], HR[2], HR[3]};                                                      \
    if (has_pro) {                                                                                  \
      const bool aff = p.pro_scale != nullptr;                                                      \
      v[0] = pro_x3<PRO>(aff ? v[0] * sc_s[0] + sh_s[0] : v[0]);                                    \
      v[1] = pro_x3<PRO>(aff ? v[1] * sc_s[1] + sh_s[1] : v[1]);                                    \
      v[2] = pro_x3<PRO>(aff ? v[2] * sc_s[2] + sh_s[2] : v[2]);                                    \
      v[3] = pro_x3<PRO>(aff ? v[3] * sc_s[3] + sh_s[3] : v[3]);                                    \
    }                                                                                               \
    const float mk = ((s_mask >> (K)) & 1u) ? s_ins : 0.f;   /* zero padding applies AFTER the prologue */ \
    v[0] *= mk; v[1] *= mk; v[2] *= mk; v[3] *= mk;                                                 \
    f16x4 hi, lo;                                                                                   \
    split4(v, hi, lo);                                                                              \
    _Float16* dst = &Hn[(hp < HALO_PIX ? hp : HALO_PIX) * XPITCH + g * 4];                          \
    *reinterpret_cast<f16x4*>(dst) = hi;                                                            \
    *reinterpret_cast<f16x4*>(dst + 16) = lo;                                                       \
  }
  // operand fragments of tap T+1 are read from LDS while tap T multiplies (two register sets, A / B): with one wave per
  // SIMD nothing else covers the ds_read latency
#define KEEP_P_FRAGS(SET, HB, WB, T, KH, KW)                                                        \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
    const _Float16* src = &(HB)[a_base + ((j + (KH)) * HALO_W + (KW)) * XPITCH];                    \
    ph##SET[j] = *reinterpret_cast<const f16x8*>(src);                                              \
    pl##SET[j] = *reinterpret_cast<const f16x8*>(src + 16);                                         \
  }                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
    const _Float16* src = &(WB)[b_base + ((T) * 64 + i * 32) * XPITCH];                             \
    wh##SET[i] = *reinterpret_cast<const f16x8*>(src);                                              \
    wl##SET[i] = *reinterpret_cast<const f16x8*>(src + 16);                                         \
  }
#define KEEP_P_TAP(T, WR, SET, NEXT_FRAGS, STAGE_FETCH_H)                                           \
  {                                                                                                 \
    NEXT_FRAGS                                                                                      \
    STAGE_FETCH_H                                                                                   \
    if (SCHED != 3) {                                                                               \
      *reinterpret_cast<uint4*>(&Wn[w_dst + (T) * 64 * XPITCH]) =                                   \
          make_uint4(WR[0] & wmask, WR[1] & wmask, WR[2] & wmask, WR[3] & wmask);                   \
      KEEP_P_GLOAD(WR, fw + (long)(T) * p.Cin * 2);                                                 \
    }                                                                                               \
    /* term-major order: consecutive MFMAs hit DIFFERENT accumulators (a filler instruction between two MFMAs on the */ \
    /* same accumulator costs ~43 cycles on gfx950, between different ones ~6) */                   \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl##SET[i], ph##SET[j], acc[i][j], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh##SET[i], pl##SET[j], acc[i][j], 0, 0, 0); \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh##SET[i], ph##SET[j], acc[i][j], 0, 0, 0); \
    if (SCHED == 1) {   /* one tap = one scheduling region: LDS reads first, then MFMAs evenly spaced by the VALU work */ \
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                                            \
      _Pragma("unroll") for (int q = 0; q < 12; ++q) {                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x002, (PRO == KEEP_PRO_SWISH ? 8 : 3), 0);            \
      }                                                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
    if (SCHED == 2) __builtin_amdgcn_sched_barrier(0);                                              \
  }
#define KEEP_P_SFH(K, HR, WR, HOFF) if (SCHED != 3) { KEEP_P_GWAIT2(HR, WR); KEEP_P_STAGE_H(K, HR) KEEP_P_GLOAD(HR, fin + HOFF); }
    KEEP_P_TAP(0, wr0, A, KEEP_P_FRAGS(B, Hc, Wc, 1, 0, 1), KEEP_P_SFH(0, hr0, wr0, h_off0))
    KEEP_P_TAP(1, wr1, B, KEEP_P_FRAGS(A, Hc, Wc, 2, 0, 2), KEEP_P_SFH(1, hr1, wr1, h_off1))
    KEEP_P_TAP(2, wr2, A, KEEP_P_FRAGS(B, Hc, Wc, 3, 1, 0), KEEP_P_SFH(2, hr2, wr2, h_off2))
    KEEP_P_TAP(3, wr3, B, KEEP_P_FRAGS(A, Hc, Wc, 4, 1, 1), KEEP_P_SFH(3, hr3, wr3, h_off3))
    KEEP_P_TAP(4, wr4, A, KEEP_P_FRAGS(B, Hc, Wc, 5, 1, 2), KEEP_P_SFH(4, hr4, wr4, h_off4))
    KEEP_P_TAP(5, wr5, B, KEEP_P_FRAGS(A, Hc, Wc, 6, 2, 0), KEEP_P_SFH(5, hr5, wr5, h_off5))
    KEEP_P_TAP(6, wr6, A, KEEP_P_FRAGS(B, Hc, Wc, 7, 2, 1), if (SCHED != 3) { KEEP_P_GWAIT1(wr6); })
    KEEP_P_TAP(7, wr7, B, KEEP_P_FRAGS(A, Hc, Wc, 8, 2, 2), if (SCHED != 3) { KEEP_P_GWAIT1(wr7); })
    KEEP_P_TAP(8, wr8, A, , if (SCHED != 3) { KEEP_P_GWAIT1(wr8); })
    // LDS hand-off only: this wave's LDS writes are done (lgkmcnt) and every wave has finished reading buffer `buf`;
    // the global loads issued above stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    KEEP_P_FRAGS(A, Hn, Wn, 0, 0, 0)                       // tap 0 of the next step, from the buffer just completed
    if (c_ch == nch - 1) {
      // ---- epilogue straight from the accumulators: lane = one pixel, 4 consecutive couts per 16-byte store
      const float asc = p.acc_scale * c_inv;
      float cs[2][16], css[2][16];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { cs[i][r] = 0.f; css[i][r] = 0.f; }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long m = ((long)c_n * p.Ho + c_oy0 + 2 * wave + j) * p.Wo + c_ox0 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int co = c_n0 + i * 32 + 8 * g4 + 4 * lhi;
            if (co < p.Cout) {                     // Cout % 32 == 0: a 4-channel group is inside or outside as a whole
              float e[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) e[q] = acc[i][j][g4 * 4 + q] * asc;
              if (p.bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co);
                e[0] += b4.x; e[1] += b4.y; e[2] += b4.z; e[3] += b4.w;
              }
              if (!SIMPLE_EPI) {
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = act_apply(e[q], p.epi_act);
              }
              if (p.res) {
                const float4 r4 = *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
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
              *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                cs[i][g4 * 4 + q] += e[q];
                css[i][g4 * 4 + q] += e[q] * e[q];
              }
            }
          }
        }
      }
      if (p.stats) {        // per wave: sum over its 64 pixels = over the 32 lanes of a half-wave (+ the j loop above)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float a = cs[i][r], b2 = css[i][r];
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) {
              a += __shfl_xor(a, o);
              b2 += __shfl_xor(b2, o);
            }
            cs[i][r] = a;
            css[i][r] = b2;
          }
        if (l31 == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int co = c_n0 + i * 32 + 8 * g4 + 4 * lhi;
              if (co < p.Cout) {
                float* dst = p.stats + (((long)c_n * p.stats_P + c_tile * 4 + wave) * p.Cout + co) * 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  dst[q * 2 + 0] = cs[i][g4 * 4 + q];
                  dst[q * 2 + 1] = css[i][g4 * 4 + q];
                }
              }
            }
        }
      }
    }
    // ---- advance the pipeline: C <- S <- F <- next
    c_ch = s_ch; c_n = s_n; c_oy0 = s_oy0; c_ox0 = s_ox0; c_n0 = s_n0; c_tile = s_tile; c_inv = s_inv;
    s_ch = f_ch; s_n = f_n; s_oy0 = f_oy0; s_ox0 = f_ox0; s_n0 = f_n0; s_tile = f_tile; s_mask = f_mask; s_wok = f_wok;
    s_ins = f_ins; s_inv = f_inv;
    if (f_ch >= 0) {
      if (f_ch + 1 < nch) {
        f_ch += 1;
      } else {
        f_item += gridDim.x;
        KEEP_P_DECODE()
      }
    }
    buf ^= 1;
    if (c_ch < 0 && s_ch < 0 && f_ch < 0) break;
  }
#undef KEEP_P_GLOAD
#undef KEEP_P_GWAIT1
#undef KEEP_P_GWAIT2
#undef KEEP_P_SFH
#undef KEEP_P_FRAGS
#undef KEEP_P_TAP
#undef KEEP_P_STAGE_H
#undef KEEP_P_DECODE
#undef KEEP_P_HOFF
}

// ------------------------------------------------------------------------------------------------ 3x3 halo, ping-pong
// Measured on the kernels above (128 ch @256^2, N=4): the MFMA + fragment-read loop alone runs at the chip's practical
// ceiling (1.7-1.9 PF raw at the clocks dense MFMA sustains), but converting + writing the next chunk's operands to LDS
// does not overlap with it across two co-resident blocks (249 us = 142 + 107), and folding that work into the MFMA wave's
// own instruction stream (conv3x3_halo_x3p_kernel) starves the single wave.  What the hardware does overlap is one wave
// on the matrix pipe with ANOTHER wave of the same SIMD on VALU / LDS-write -- provided the two really are in opposite
// phases.  This kernel enforces that: one 512-thread block per CU = two 4-wave groups, each with its own LDS image and its own
// stream of work items, running the same [multiply | refill] cycle exactly half a period apart; every phase change of
// either group is one block-wide s_barrier, so the groups cannot drift into step:
//     interval   I0        I1        I2        I3        I4
//     group 0    fill c0   MFMA c0   fill c1   MFMA c1   fill c2 ...
//     group 1    -         fill c0   MFMA c0   fill c1   MFMA c1 ...
//   fill = [epilogue of the finished item] + global loads of the next chunk + affine/swish/split + LDS writes: all of it
//   sits in the shadow of the other group's 108 MFMAs per wave; nothing is carried in registers across a barrier except
//   the accumulators, so the MFMA phase affords two fragment register sets (reads of tap t+1 under the MFMAs of tap t).
//   Epilogue straight from the accumulators (operand roles swapped: A = weights, B = pixels -> a lane holds 4 consecutive
//   couts of one pixel = 16-byte stores), no LDS round trip.
#ifdef KEEP_X3_ABLATE
__device__ unsigned long long keep_dbg_clk[8192];      // dev builds: per-interval time stamps of block 0 (tools/dev/)
extern "C" int keep_debug_read_clk(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(keep_dbg_clk), (size_t)n * 8);
}
#define KEEP_DBG_STAMP(SLOT)                                                                   \
  if (blockIdx.x == 0 && lane == 0 && gw == 0 && iv < 256) keep_dbg_clk[((iv * 2 + grp) * 4) + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define KEEP_DBG_STAMP(SLOT)
#endif
template <int PRO, bool SIMPLE_EPI>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_x3g_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HALO_W = 34, HALO_PIX = 340;
  constexpr int HBUF = (HALO_MAXPIX + 1) * XPITCH;     // + 1 dummy row (idle staging lanes write there)
  constexpr int REGION = HBUF + 9 * 64 * XPITCH;       // fp16 elements per group
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int g = gtid & 3;
  const int nch = p.Cin >> 4;
  _Float16* const Hs = reinterpret_cast<_Float16*>(lds_dyn) + grp * REGION;
  _Float16* const Ws = Hs + HBUF;
  const bool has_pro = p.pro_scale != nullptr || PRO != KEEP_PRO_NONE;
  const int Hv = p.upsample ? 2 * p.H : p.H, Wv = p.upsample ? 2 * p.W : p.W;

  // ---- this group's work items: first, first + stride, ...; both groups run the same number of barrier intervals
  const int stride = 2 * gridDim.x;
  const int first = 2 * blockIdx.x + grp, first_o = 2 * blockIdx.x + (grp ^ 1);
  const int n_my = first < n_items ? (n_items - 1 - first) / stride + 1 : 0;
  const int n_other = first_o < n_items ? (n_items - 1 - first_o) / stride + 1 : 0;
  const int my_steps = n_my * nch;
  const int T = (n_my > n_other ? n_my : n_other) * nch;

  // geometry of the item being refilled (f_*) and of the item being multiplied / stored (c_*)
  int f_n = 0, f_oy0 = 0, f_ox0 = 0, f_n0 = 0, f_tile = 0;
  float f_ins = 1.f, f_inv = 1.f;
  int c_n = 0, c_oy0 = 0, c_ox0 = 0, c_n0 = 0, c_tile = 0;
  float c_inv = 1.f;
  int h_off[HALO_IT];
  long img_off = 0, w_base = 0, sc_off = 0;
  bool w_ok = false;
  auto decode = [&](int item) {
    const HaloItem it = halo_decode<32, 4>(p, item, n_items, tiles_x, tiles_y, ncb);
    f_n = it.n; f_oy0 = it.oy0; f_ox0 = it.ox0; f_n0 = it.n0; f_tile = it.ty * tiles_x + it.tx;
    f_ins = 1.f; f_inv = 1.f;
    if (p.in_amax) x3_range_scale(p.in_amax[it.n], f_ins, f_inv);
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (gtid >> 2) + k * 64;
      h_off[k] = -1;
      if (hp < HALO_PIX) {
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int iy = it.oy0 - 1 + hy, ix = it.ox0 - 1 + hx;
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
          const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          h_off[k] = (sy * p.W + sx) * p.in_ld + g * 4;
        }
      }
    }
    img_off = (long)it.n * p.H * p.W * p.in_ld;
    sc_off = (long)it.n * p.Cin + g * 4;
    w_ok = (it.n0 + (gtid >> 2)) < p.Cout;
    w_base = w_ok ? ((long)(it.n0 + (gtid >> 2)) * 9) * p.Cin * 2 + g * 8 : 0;
  };
  // The operands of chunk c+1 are requested at the START of the interval that multiplies chunk c (`issue`: global ->
  // registers, asm volatile so that hipcc cannot sink the loads to their use) and converted + written to LDS in the
  // following refill interval (`refill`), when they have long landed.
  typedef float gf32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));
  gf32x4 hreg[HALO_IT];
  gu32x4 wr[9];
  gf32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  unsigned r_mask = 0u;          // bit k: halo piece k of the data in flight is inside the image
  bool r_wok = false;
  float r_ins = 1.f;
#pragma unroll
  for (int k = 0; k < HALO_IT; ++k) hreg[k] = sh4;
#pragma unroll
  for (int t = 0; t < 9; ++t) wr[t] = (gu32x4){0u, 0u, 0u, 0u};
#define KEEP_G_GLOAD(R, PTR) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(R) : "v"(PTR))
  auto issue = [&](int ch) {     // for chunk `ch` of the item last decoded
    const int c0 = ch << 4;
    r_mask = 0u;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const float* src = p.in + img_off + (h_off[k] >= 0 ? h_off[k] : 0) + c0;       // always a valid address
      KEEP_G_GLOAD(hreg[k], src);
      if (h_off[k] >= 0) r_mask |= 1u << k;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const unsigned short* src = p.wx3 + w_base + ((long)t * p.Cin + c0) * 2;
      KEEP_G_GLOAD(wr[t], src);
    }
    if (p.pro_scale) {
      const float* ps = p.pro_scale + sc_off + c0;
      const float* ph = p.pro_shift + sc_off + c0;
      KEEP_G_GLOAD(sc4, ps);
      KEEP_G_GLOAD(sh4, ph);
    }
    r_wok = w_ok;
    r_ins = f_ins;
  };
#undef KEEP_G_GLOAD
  auto refill = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(hreg[0]), "+v"(hreg[1]), "+v"(hreg[2]), "+v"(hreg[3]), "+v"(hreg[4]), "+v"(hreg[5]),
                 "+v"(sc4), "+v"(sh4));
    asm volatile("" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wr[4]), "+v"(wr[5]), "+v"(wr[6]), "+v"(wr[7]),
                 "+v"(wr[8]));
    const unsigned wmask = r_wok ? 0xffffffffu : 0u;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (gtid >> 2) + k * 64;
      float v[4] = {hreg[k][0], hreg[k][1], hreg[k][2], hreg[k][3]};
      if (has_pro) {
        const bool aff = p.pro_scale != nullptr;
        v[0] = pro_x3<PRO>(aff ? v[0] * sc4[0] + sh4[0] : v[0]);
        v[1] = pro_x3<PRO>(aff ? v[1] * sc4[1] + sh4[1] : v[1]);
        v[2] = pro_x3<PRO>(aff ? v[2] * sc4[2] + sh4[2] : v[2]);
        v[3] = pro_x3<PRO>(aff ? v[3] * sc4[3] + sh4[3] : v[3]);
      }
      const float mk = ((r_mask >> k) & 1u) ? r_ins : 0.f;   // zero padding applies AFTER the prologue
      v[0] *= mk; v[1] *= mk; v[2] *= mk; v[3] *= mk;
      f16x4 hi, lo;
      split4(v, hi, lo);
      _Float16* dst = &Hs[(hp < HALO_PIX ? hp : HALO_PIX) * XPITCH + g * 4];
      *reinterpret_cast<f16x4*>(dst) = hi;
      *reinterpret_cast<f16x4*>(dst + 16) = lo;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
      *reinterpret_cast<uint4*>(&Ws[(t * 64 + (gtid >> 2)) * XPITCH + g * 8]) =
          make_uint4(wr[t][0] & wmask, wr[t][1] & wmask, wr[t][2] & wmask, wr[t][3] & wmask);
  };

  f32x16 acc[2][2];            // [cout block i][pixel block j]: lane = pixel l31 of tile row 2*gw+j, registers = couts
  const int a_base = ((2 * gw) * HALO_W + l31) * XPITCH + lhi * 8;
  const int b_base = l31 * XPITCH + lhi * 8;
#define KEEP_G_FRAGS(SET, T, KH, KW)                                                               \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
    const _Float16* src = &Hs[a_base + ((j + (KH)) * HALO_W + (KW)) * XPITCH];                     \
    ph##SET[j] = *reinterpret_cast<const f16x8*>(src);                                             \
    pl##SET[j] = *reinterpret_cast<const f16x8*>(src + 16);                                        \
  }                                                                                                \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
    const _Float16* src = &Ws[b_base + ((T) * 64 + i * 32) * XPITCH];                              \
    wh##SET[i] = *reinterpret_cast<const f16x8*>(src);                                             \
    wl##SET[i] = *reinterpret_cast<const f16x8*>(src + 16);                                        \
  }
  // term-major: consecutive MFMAs hit different accumulators
#define KEEP_G_MMA(SET)                                                                            \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl##SET[i], ph##SET[j], acc[i][j], 0, 0, 0); \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh##SET[i], pl##SET[j], acc[i][j], 0, 0, 0); \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh##SET[i], ph##SET[j], acc[i][j], 0, 0, 0);
  auto mma = [&]() {
    f16x8 whA[2], wlA[2], phA[2], plA[2], whB[2], wlB[2], phB[2], plB[2];
    __builtin_amdgcn_s_setprio(3);        // the refilling group shares this SIMD's issue port: matrix instructions first
    KEEP_G_FRAGS(A, 0, 0, 0)
    KEEP_G_FRAGS(B, 1, 0, 1) KEEP_G_MMA(A)
    KEEP_G_FRAGS(A, 2, 0, 2) KEEP_G_MMA(B)
    KEEP_G_FRAGS(B, 3, 1, 0) KEEP_G_MMA(A)
    KEEP_G_FRAGS(A, 4, 1, 1) KEEP_G_MMA(B)
    KEEP_G_FRAGS(B, 5, 1, 2) KEEP_G_MMA(A)
    KEEP_G_FRAGS(A, 6, 2, 0) KEEP_G_MMA(B)
    KEEP_G_FRAGS(B, 7, 2, 1) KEEP_G_MMA(A)
    KEEP_G_FRAGS(A, 8, 2, 2) KEEP_G_MMA(B)
    KEEP_G_MMA(A)
    __builtin_amdgcn_s_setprio(0);
  };
#undef KEEP_G_FRAGS
#undef KEEP_G_MMA
  auto epilogue = [&]() {
    const float asc = p.acc_scale * c_inv;
    const long m0 = ((long)c_n * p.Ho + c_oy0 + 2 * gw) * p.Wo + c_ox0 + l31;       // pixel of j = 0; j = 1 is one row below
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int co = c_n0 + i * 32 + 8 * g4 + 4 * lhi;
        const bool cok = co < p.Cout;              // Cout % 32 == 0: a 4-channel group is inside or outside as a whole
        float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
        if (cok) {
          float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + co);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const long m = m0 + (long)j * p.Wo;
            float e[4] = {acc[i][j][g4 * 4 + 0] * asc + b4.x, acc[i][j][g4 * 4 + 1] * asc + b4.y,
                          acc[i][j][g4 * 4 + 2] * asc + b4.z, acc[i][j][g4 * 4 + 3] * asc + b4.w};
            if (!SIMPLE_EPI) {
#pragma unroll
              for (int q = 0; q < 4; ++q) e[q] = act_apply(e[q], p.epi_act);
            }
            if (p.res) {
              const float4 r4 = *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
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
            *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              s4[q] += e[q];
              ss4[q] += e[q] * e[q];
            }
          }
        }
        if (p.stats) {
          // per 16-lane DPP row: 32 output pixels (16 lanes x the two tile rows above) -> one partial per row, stats_P =
          // Ho*Wo/32.  Four DPP adds per value (quad swaps, half-row mirror, row mirror) instead of five LDS-routed
          // shuffles: measured, the shuffle form made the epilogue 20 k cycles per item (the MFMA interval is 6 k).
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            s4[q] = row16_sum(s4[q]);
            ss4[q] = row16_sum(ss4[q]);
          }
          if ((lane & 15) == 0 && cok) {
            float* dst = p.stats + (((long)c_n * p.stats_P + (c_tile * 4 + gw) * 2 + (l31 >> 4)) * p.Cout + co) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              dst[q * 2 + 0] = s4[q];
              dst[q * 2 + 1] = ss4[q];
            }
          }
        }
      }
    }
  };

  // ---- interval loop: group-local interval k = iv - grp; k == 0: first fill, odd k: multiply step (k-1)/2, even k: refill
  for (int iv = 0; iv < 2 * T + 2; ++iv) {
    const int k = iv - grp;
    KEEP_DBG_STAMP(0)
    if (k == 0) {
      if (my_steps > 0) {
        decode(first);
        issue(0);
        refill();
      }
    } else if (k >= 1 && k <= 2 * T) {
      const int s = (k - 1) >> 1;
      if (s < my_steps) {
        const int it_idx = s / nch, ch = s - it_idx * nch;
        if ((k - 1) & 1) {                       // ---- refill interval (the other group multiplies)
          if (ch == nch - 1) epilogue();
          if (s + 1 < my_steps) refill();
        } else {                                 // ---- multiply interval
          if (ch == 0) {
            c_n = f_n; c_oy0 = f_oy0; c_ox0 = f_ox0; c_n0 = f_n0; c_tile = f_tile; c_inv = f_inv;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
          if (s + 1 < my_steps) {                // request the next chunk now: it lands while this one multiplies
            if (ch == nch - 1) {
              decode(first + (it_idx + 1) * stride);
              issue(0);
            } else {
              issue(ch + 1);
            }
          }
          mma();
        }
      }
    }
    KEEP_DBG_STAMP(1)
    if (iv < 2 * T + 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    KEEP_DBG_STAMP(2)
  }
}

// ------------------------------------------------------------------------------------------------ gather GEMM, split fp16
// Everything that is not a 3x3 stride-1 convolution on a tileable map: token GEMMs, 1x1 convs, stride-2 convs.  Implicit
// GEMM like conv_bf16_kernel: K step = 32 channels of one tap; LDS rows [hi x32 | lo x32 | pad x8] at a 144-byte pitch (9 slots:
// ds_read_b128 conflict-free); two LDS buffers, the next step's operands in registers while the current one multiplies.
// A is split while staging (fp32 activations, optional GroupNorm affine + activation first); B comes pre-split.
#define XBK 32
#define XP (2 * XBK + 8)

template <int WGM, int WGN, int TM, int TN, bool PLAIN>
__global__ __launch_bounds__(256) void conv_x3_kernel(ConvP p) {
  constexpr int BM = WGM * TM * 32;
  constexpr int BN = WGN * TN * 32;
  constexpr int A_IT = BM / 64;             // (row, 8-channel group) pieces per thread: 4 groups per row
  constexpr int B_IT = BN / 32;             // (row, 16-byte piece) per thread: 8 pieces per row
  static_assert(WGM * WGN == 4 && A_IT >= 1 && B_IT >= 1, "tile config");
  constexpr int MAIN_B = 2 * (BM + BN) * XP * 2;
  constexpr int EPI_B = 4 * (TM * 32) * (TN * 32 + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[MAIN_B > EPI_B ? MAIN_B : EPI_B];
  _Float16(*As)[BM * XP] = reinterpret_cast<_Float16(*)[BM * XP]>(smem_b);
  _Float16(*Bs)[BN * XP] = reinterpret_cast<_Float16(*)[BN * XP]>(smem_b + 2 * BM * XP * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN;
  const int wn = wave % WGN;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int z = blockIdx.z;
  const int cchunks = (p.Cin + XBK - 1) / XBK;
  const int nsteps = p.KH * p.KW * cchunks;
  const int per = (nsteps + p.split_k - 1) / p.split_k;
  const int s_begin = z * per;
  const int s_end = min(nsteps, s_begin + per);

  const int a_grp = tid & 3, a_row0 = tid >> 2;          // + it*64
  const int b_pc = tid & 7, b_row0 = tid >> 3;           // + it*32
  const int hw = p.Ho * p.Wo;
  int a_n[A_IT], a_oy[A_IT], a_ox[A_IT];
  bool a_mv[A_IT];
  float a_s[A_IT];                                    // range scale of the row's image (x3_in_amax), 1 without
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const long m = m0 + a_row0 + it * 64;
    a_mv[it] = m < p.M;
    a_n[it] = 0; a_oy[it] = 0; a_ox[it] = 0;
    a_s[it] = 1.f;
    if (a_mv[it]) {
      a_n[it] = (int)(m / hw);
      const int r = (int)(m - (long)a_n[it] * hw);
      a_oy[it] = r / p.Wo;
      a_ox[it] = r - a_oy[it] * p.Wo;
      if (p.in_amax) {
        float inv;
        x3_range_scale(p.in_amax[a_n[it]], a_s[it], inv);
      }
    }
  }
  const long wrow_stride = (long)p.KH * p.KW * p.Cin * 2;     // fp16 elements per cout row
  // b piece -> LDS column: 16-channel chunk c = b_pc >> 2, part = b_pc & 3 (0,1: hi ch 0-7 / 8-15; 2,3: lo)
  const int b_col = ((b_pc & 3) >> 1) * XBK + (b_pc >> 2) * 16 + (b_pc & 1) * 8;

  float a_raw[A_IT][8];
  bool a_ok[A_IT];
  uint4 b_raw[B_IT];
  int a_c = 0;
  const long m_last = (m0 + BM - 1 < p.M) ? (m0 + BM - 1) : (long)p.M - 1;
  const bool uni_n = !PLAIN && p.pro_scale && ((m0 / hw) == (m_last / hw));
  const long uni_off = (m0 / hw) * (long)p.Cin;
  float u_sc[8], u_sh[8];

  auto fetch = [&](int s) {
    const int tap = s / cchunks;
    const int c0 = (s - tap * cchunks) * XBK;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
    const int ca = c0 + a_grp * 8;
    a_c = ca;
    if (uni_n && ca < p.Cin) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        u_sc[j] = p.pro_scale[uni_off + ca + j];
        u_sh[j] = p.pro_shift[uni_off + ca + j];
      }
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int iy = a_oy[it] * p.stride - p.pad_t + kh;
      const int ix = a_ox[it] * p.stride - p.pad_l + kw;
      a_ok[it] = a_mv[it] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ca < p.Cin;
      if (a_ok[it]) {
        const float* src = p.in + (((long)a_n[it] * p.H + iy) * p.W + ix) * p.in_ld + ca;
        const float4 v0 = *reinterpret_cast<const float4*>(src);
        const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
        a_raw[it][0] = v0.x; a_raw[it][1] = v0.y; a_raw[it][2] = v0.z; a_raw[it][3] = v0.w;
        a_raw[it][4] = v1.x; a_raw[it][5] = v1.y; a_raw[it][6] = v1.z; a_raw[it][7] = v1.w;
      }
    }
    const bool cb_ok = c0 + (b_pc >> 2) * 16 < p.Cin;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int co = n0 + b_row0 + it * 32;
      b_raw[it] = make_uint4(0u, 0u, 0u, 0u);
      if (co < p.Cout && cb_ok)
        b_raw[it] = *reinterpret_cast<const uint4*>(p.wx3 + (long)co * wrow_stride + ((long)tap * p.Cin + c0) * 2 + b_pc * 8);
    }
  };

  auto stage = [&](int buf) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      f16x8 hi, lo;
      if (a_ok[it]) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = a_raw[it][j];
        if (!PLAIN) {
          if (uni_n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * u_sc[j] + u_sh[j];
          } else if (p.pro_scale) {
            const float* sc = p.pro_scale + (long)a_n[it] * p.Cin + a_c;
            const float* sh = p.pro_shift + (long)a_n[it] * p.Cin + a_c;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
          }
          if (p.pro_act != KEEP_PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = pro_apply(v[j], p.pro_act);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float vs = v[j] * a_s[it];
          const _Float16 h = (_Float16)vs;
          hi[j] = h;
          lo[j] = (_Float16)(vs - (float)h);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { hi[j] = (_Float16)0.f; lo[j] = (_Float16)0.f; }
      }
      _Float16* dst = &As[buf][(a_row0 + it * 64) * XP + a_grp * 8];
      *reinterpret_cast<f16x8*>(dst) = hi;
      *reinterpret_cast<f16x8*>(dst + XBK) = lo;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
      *reinterpret_cast<uint4*>(&Bs[buf][(b_row0 + it * 32) * XP + b_col]) = b_raw[it];
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
  const int a_f0 = (wm * TM * 32 + l31) * XP + lhi * 8;
  const int b_f0 = (wn * TN * 32 + l31) * XP + lhi * 8;

  auto mma_step = [&](int buf) {
    const _Float16* Ab = As[buf];
    const _Float16* Bb = Bs[buf];
#pragma unroll
    for (int ks = 0; ks < XBK / 16; ++ks) {
      f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(Ab + a_f0 + i * 32 * XP + ks * 16);
        al[i] = *reinterpret_cast<const f16x8*>(Ab + a_f0 + i * 32 * XP + ks * 16 + XBK);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const f16x8*>(Bb + b_f0 + j * 32 * XP + ks * 16);
        bl[j] = *reinterpret_cast<const f16x8*>(Bb + b_f0 + j * 32 * XP + ks * 16 + XBK);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          MMA_X3(acc[i][j], ah[i], al[i], bh[j], bl[j])
        }
    }
  };

  if (s_begin < s_end) {
    fetch(s_begin);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int s = s_begin; s < s_end; ++s) {
      const bool more = (s + 1 < s_end);
      if (more) fetch(s + 1);
      mma_step(buf);
      if (more) stage(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }
  const float asc = p.acc_scale;
  if (p.in_amax) {          // undo the per-image range scale: accumulator register r of tile i holds output row ...
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float sr = 1.f, inv = 1.f;
        if (m < p.M) x3_range_scale(p.in_amax[m / hw], sr, inv);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j][r] *= asc * inv;
      }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= asc;
  }
  staged_epilogue<WGM, WGN, TM, TN>(p, acc, reinterpret_cast<float*>(smem_b), m0, n0, wm, wn, lane, wave, z);
}

// ------------------------------------------------------------------------------------------------ dispatch
static int x3_num_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}

// Geometry the x3 kernels accept (everything else of a KEEP_MMA_X3 call runs on the exact-f32 kernels: same parity grade).
bool keep_conv_x3_halo_ok(const keep_conv2d_args* a) {
  return a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad_t == 1 &&
         a->pad_l == 1 && (a->Cin % 16 == 0) && (a->Cout % 32 == 0) &&
         ((a->Ho % 8 == 0 && a->Wo % 32 == 0) || (a->Ho % 16 == 0 && a->Wo % 16 == 0)) &&
         a->Ho == (a->upsample ? 2 * a->H : a->H) && a->Wo == (a->upsample ? 2 * a->W : a->W) &&
         (!a->pro_scale || ((uintptr_t)a->pro_scale % 16 == 0 && (uintptr_t)a->pro_shift % 16 == 0)) &&
         (a->in_ld % 4 == 0) && ((uintptr_t)a->in % 16 == 0) && (a->out_ld % 4 == 0) && ((uintptr_t)a->out % 16 == 0) &&
         (!a->residual || (a->res_ld % 4 == 0 && (uintptr_t)a->residual % 16 == 0)) &&
         (!a->aux || (uintptr_t)a->aux % 16 == 0) && (!a->bias || (uintptr_t)a->bias % 16 == 0) &&
         (!a->workspace || (uintptr_t)a->workspace % 16 == 0);
}

bool keep_conv_x3_gather_ok(const keep_conv2d_args* a, const ConvP& p) {
  return a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && !a->upsample && (a->Cin % 16 == 0) && (a->in_ld % 4 == 0) &&
         ((uintptr_t)a->in % 16 == 0) && p.vec_epi;
}

// the pipelined kernel wants wide tiles, no split-K and at least two work items per CU (one block per CU walks them)
// 0: two blocks per CU (conv3x3_halo_x3_kernel), 1: in-stream pipelined (x3p, dev A/B only), 2: ping-pong groups (x3g)
int keep_conv_x3_halo_variant(const keep_conv2d_args* a, int split_k) {
  const bool wide = (a->Ho % 8 == 0 && a->Wo % 32 == 0);
  if (!wide || split_k != 1 || getenv("KEEP_NO_HALO_X3P")) return 0;
  const long n_items = (long)a->N * (a->Wo / 32) * (a->Ho / 8) * ((a->Cout + 63) / 64);
  if (!(n_items >= 2L * x3_num_cu() || getenv("KEEP_X3P_ALWAYS"))) return 0;
  const char* sel = getenv("KEEP_X3_HALO");
  if (sel && sel[0] == 'b') return 0;
  if (sel && sel[0] == 'p') return 1;
  return 2;
}

int keep_conv2d_x3_halo(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  const int nchunks = a->Cin / 16;
  if (p.split_k > nchunks) p.split_k = nchunks;
  const bool wide = (a->Ho % 8 == 0 && a->Wo % 32 == 0);
  const int tw = wide ? 32 : 16, th = 256 / tw;
  const int tiles_x = a->Wo / tw, tiles_y = a->Ho / th, ncb = (a->Cout + 63) / 64;
  const int n_items = a->N * tiles_x * tiles_y * ncb * p.split_k;
  const int n_cu = x3_num_cu();
  dim3 grid(n_items < 2 * n_cu ? n_items : 2 * n_cu), block(256);
  const bool simple = p.split_k == 1 && !a->aux && a->epi_act == KEEP_ACT_NONE;
  // pipelined single-block-per-CU kernel: wide tiles, no split-K, at least two work items per CU
  const int variant = keep_conv_x3_halo_variant(a, p.split_k);
  if (variant == 2) {
    constexpr size_t kLdsG = 2 * (size_t)(HALO_MAXPIX + 1 + 9 * 64) * XPITCH * 2;
    static bool attr_g = false;
#define KEEP_G_FOREACH(X) X(KEEP_PRO_NONE, true) X(KEEP_PRO_NONE, false) X(KEEP_PRO_SWISH, true) X(KEEP_PRO_SWISH, false) \
                          X(KEEP_PRO_RELU, true) X(KEEP_PRO_RELU, false)
    if (!attr_g) {
#define KEEP_G_ATTR(PROV, SIMP)                                                                                    \
  if (hipFuncSetAttribute((const void*)conv3x3_halo_x3g_kernel<PROV, SIMP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)kLdsG) != hipSuccess) {                                                             \
    keep_set_error("keep_conv2d: hipFuncSetAttribute(halo x3g) failed");                                           \
    return KEEP_EHIP;                                                                                              \
  }
      KEEP_G_FOREACH(KEEP_G_ATTR)
#undef KEEP_G_ATTR
      attr_g = true;
    }
    const int nb = (n_items + 1) / 2;
    dim3 gridg(nb < n_cu ? nb : n_cu);
#define KEEP_G_LAUNCH(PROV, SIMP)                                                                                  \
  if (a->pro_act == PROV && simple == SIMP)                                                                        \
    hipLaunchKernelGGL((conv3x3_halo_x3g_kernel<PROV, SIMP>), gridg, dim3(512), kLdsG, st, p, tiles_x, tiles_y, ncb, n_items);
    KEEP_G_FOREACH(KEEP_G_LAUNCH)
#undef KEEP_G_LAUNCH
#undef KEEP_G_FOREACH
    KEEP_LAUNCH_CHECK("keep_conv2d(halo x3g)");
    return KEEP_OK;
  }
  if (variant == 1) {
    constexpr size_t kLds = 2 * (size_t)(HALO_MAXPIX + 1 + 9 * 64) * XPITCH * 2;
    static bool attr_set = false;
#define KEEP_P_FOREACH(X) X(KEEP_PRO_NONE, true) X(KEEP_PRO_NONE, false) X(KEEP_PRO_SWISH, true) X(KEEP_PRO_SWISH, false) \
                          X(KEEP_PRO_RELU, true) X(KEEP_PRO_RELU, false)
    if (!attr_set) {
#define KEEP_P_ATTR(PROV, SIMP)                                                                                    \
  if (hipFuncSetAttribute((const void*)conv3x3_halo_x3p_kernel<PROV, SIMP>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)kLds) != hipSuccess) {                                                              \
    keep_set_error("keep_conv2d: hipFuncSetAttribute(halo x3p) failed");                                           \
    return KEEP_EHIP;                                                                                              \
  }
      KEEP_P_FOREACH(KEEP_P_ATTR)
#undef KEEP_P_ATTR
      attr_set = true;
    }
    dim3 gridp(n_items < n_cu ? n_items : n_cu);
#ifdef KEEP_X3_ABLATE
    if (getenv("KEEP_X3P_SCHED") && simple && (a->pro_act == KEEP_PRO_SWISH || a->pro_act == KEEP_PRO_NONE)) {
      const int sv = atoi(getenv("KEEP_X3P_SCHED"));
#define KEEP_P_SCHED(PROV, SV)                                                                                           \
  if (a->pro_act == PROV && sv == SV) {                                                                                  \
    hipFuncSetAttribute((const void*)conv3x3_halo_x3p_kernel<PROV, true, SV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds); \
    hipLaunchKernelGGL((conv3x3_halo_x3p_kernel<PROV, true, SV>), gridp, block, kLds, st, p, tiles_x, tiles_y, ncb, n_items); \
    KEEP_LAUNCH_CHECK("keep_conv2d(halo x3p sched)");                                                                    \
    return KEEP_OK;                                                                                                      \
  }
      KEEP_P_SCHED(KEEP_PRO_NONE, 1) KEEP_P_SCHED(KEEP_PRO_SWISH, 1) KEEP_P_SCHED(KEEP_PRO_NONE, 2) KEEP_P_SCHED(KEEP_PRO_SWISH, 2)
      KEEP_P_SCHED(KEEP_PRO_NONE, 3)
#undef KEEP_P_SCHED
    }
#endif
#define KEEP_P_LAUNCH(PROV, SIMP)                                                                                  \
  if (a->pro_act == PROV && simple == SIMP)                                                                        \
    hipLaunchKernelGGL((conv3x3_halo_x3p_kernel<PROV, SIMP>), gridp, block, kLds, st, p, tiles_x, tiles_y, ncb, n_items);
    KEEP_P_FOREACH(KEEP_P_LAUNCH)
#undef KEEP_P_LAUNCH
#undef KEEP_P_FOREACH
    KEEP_LAUNCH_CHECK("keep_conv2d(halo x3p)");
    return KEEP_OK;
  }
#ifdef KEEP_X3_ABLATE
  if (getenv("KEEP_X3_EXP") && wide && simple && (a->pro_act == KEEP_PRO_SWISH || a->pro_act == KEEP_PRO_NONE)) {
    const int ex = atoi(getenv("KEEP_X3_EXP"));
#define KEEP_LAUNCH_ABL(E)                                                                                                         \
  if (ex == E) {                                                                                                                   \
    if (a->pro_act == KEEP_PRO_SWISH)                                                                                              \
      hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_SWISH, true, E>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items); \
    else                                                                                                                           \
      hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_NONE, true, E>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);  \
    KEEP_LAUNCH_CHECK("keep_conv2d(halo x3 ablation)");                                                                            \
    return KEEP_OK;                                                                                                                \
  }
    KEEP_LAUNCH_ABL(1) KEEP_LAUNCH_ABL(2) KEEP_LAUNCH_ABL(3) KEEP_LAUNCH_ABL(4) KEEP_LAUNCH_ABL(5) KEEP_LAUNCH_ABL(6) KEEP_LAUNCH_ABL(7) KEEP_LAUNCH_ABL(8)
#undef KEEP_LAUNCH_ABL
  }
#endif
#define KEEP_LAUNCH_HX2(TWV, PROV)                                                                                          \
  if (simple)                                                                                                              \
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, PROV, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);  \
  else                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, PROV, false>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
#define KEEP_LAUNCH_HX(TWV)                                       \
  if (a->pro_act == KEEP_PRO_SWISH) {                             \
    KEEP_LAUNCH_HX2(TWV, KEEP_PRO_SWISH)                          \
  } else if (a->pro_act == KEEP_PRO_RELU) {                       \
    KEEP_LAUNCH_HX2(TWV, KEEP_PRO_RELU)                           \
  } else {                                                        \
    KEEP_LAUNCH_HX2(TWV, KEEP_PRO_NONE)                           \
  }
  if (wide) {
    KEEP_LAUNCH_HX(32)
  } else {
    KEEP_LAUNCH_HX(16)
  }
#undef KEEP_LAUNCH_HX
#undef KEEP_LAUNCH_HX2
  KEEP_LAUNCH_CHECK("keep_conv2d(halo x3)");
  return KEEP_OK;
}

int keep_conv2d_x3_gather(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  const long M = p.M;
  const int steps = a->KH * a->KW * ((a->Cin + XBK - 1) / XBK);
  if (p.split_k > steps) p.split_k = steps;
  const bool plain = !a->pro_scale && a->pro_act == KEEP_PRO_NONE;
  dim3 block(256);
  if (a->Cout <= 64 || M <= 4096) {
    dim3 grid(cdiv(M, 64), cdiv(a->Cout, 64), p.split_k);
    if (plain)
      hipLaunchKernelGGL((conv_x3_kernel<2, 2, 1, 1, true>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((conv_x3_kernel<2, 2, 1, 1, false>), grid, block, 0, st, p);
  } else {
    dim3 grid(cdiv(M, 128), cdiv(a->Cout, 128), p.split_k);
    if (plain)
      hipLaunchKernelGGL((conv_x3_kernel<2, 2, 2, 2, true>), grid, block, 0, st, p);
    else
      hipLaunchKernelGGL((conv_x3_kernel<2, 2, 2, 2, false>), grid, block, 0, st, p);
  }
  KEEP_LAUNCH_CHECK("keep_conv2d(gather x3)");
  return KEEP_OK;
}
