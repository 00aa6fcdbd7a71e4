// Shared pieces of the keep_conv2d kernels (keep_conv.hip: f32 / bf16 operands; keep_conv_x3.hip: split fp16 operands):
// launch parameters, epilogues, persistent work-item decoding.
#pragma once
#include <type_traits>

#include "keep_common.h"

#define BK 16

// Cache policy (the `aux` immediate of buffer_store: 0 default, 2 = nt, streaming) of the output stores of the x3 kernels:
// the streaming halo kernel, round 3's halo kernel, the gather / GEMM kernel.  Measured per family on the whole step (DESIGN 5.4,
// tools/dev/lib_ab.py): nt on the GEMM family -0.3 %, on round 3's halo kernel another -0.3 %, on the streaming kernel 0.0 % in time but
// -6.5 % of its fetched HBM bytes (PMC: the written lines no longer evict the halo rows the neighbouring tiles re-read).
// (dev A/B) cache policy of loads that are read exactly once: the halo pieces of the streaming kernel (re-read 1.33 x by the
// neighbouring tiles: expected to lose), its residual rows, the A rows of a GEMM with one column block
#ifndef KEEP_LD_AUX_XS
#define KEEP_LD_AUX_XS 0
#endif
#ifndef KEEP_LD_AUX_RES
#define KEEP_LD_AUX_RES 0
#endif
#ifndef KEEP_LD_AUX_GEMM_A1
#define KEEP_LD_AUX_GEMM_A1 0
#endif
#ifndef KEEP_NT_C3
#define KEEP_NT_C3 0      // (dev A/B) streaming stores of the RGB x3 first-conv kernel's 64-channel output rows
#endif
#ifndef KEEP_ST_AUX_XS
#define KEEP_ST_AUX_XS 2
#endif
#ifndef KEEP_ST_AUX_HALO
#define KEEP_ST_AUX_HALO 2
#endif
#ifndef KEEP_ST_AUX_GEMM
#define KEEP_ST_AUX_GEMM 2
#endif

struct ConvP {
  const float* in;
  const float* w;
  const unsigned short* wb;  // bf16 copy of w (same layout), KEEP_MMA_BF16 only
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
  int in_bf16;  // input tensor is bf16 (halo kernel only)
  int fast;     // bf16 policy: fast-math epilogue activations
  float* stats; // optional [N][P][Cout][2] per-tile (sum, sumsq) of the epilogue output, P = stats_P tiles per image
  int stats_P;
  int out_bf16; // write the output tensor as bf16 (gather kernels' staged epilogue, persistent bf16 halo kernel)
  int vec_epi;  // Cout/out_ld/res_ld %% 4 == 0 and aligned pointers -> LDS-staged float4 epilogue
  int flatk;    // Cin < 8: K = KH*KW*Cin flattened (element-wise gather) instead of tap-major chunks
  int flatk_f32;  // the same for the f32 gather kernel (16-wide K steps)
  const unsigned short* wx3;  // KEEP_MMA_X3: weights pre-multiplied by 2^e and split into fp16 (hi, lo), [Cout][KH*KW][Cin/16][hi16|lo16]
  float acc_scale;            // KEEP_MMA_X3: 2^-e, applied to the accumulators before bias / activation
  const float* in_amax;       // KEEP_MMA_X3: per-image max |input| (NULL: inputs are split unscaled)
  const float* in2;           // KEEP_MMA_X3 GEMM form: second K-concatenated input (channels >= cin1), dense rows, or NULL
  int cin1;
  unsigned* out_amax;         // KEEP_MMA_X3: per-image max |output| as raw float bits (atomicMax), or NULL
  int reflect;                // padding pixels mirror the image (nn.ReflectionPad2d, ParseNet) instead of reading zeros
  int tile_cols;              // x3 gather kernel on a 1-D grid: column blocks per row block (0: blockIdx.x / .y are the row / column block)
  int reverse;                // x3 gather kernel: row blocks in descending order (the rows the producer wrote LAST are read first)
  const float* ln_gamma;      // x3 GEMM form, tile <4,1,1,4>: LayerNorm over the 128 output channels of a row in the epilogue (or NULL)
  const float* ln_beta;
  float ln_eps;
  int kslice_steps;           // x3 GEMM form: K steps (of 32 channels) per canonical slice of keep_gemm_x3l.hip's sums; 0 = one sequential sum
};


// nn.ReflectionPad2d index map on the (virtual, post-upsample) input extent: -1 -> 1, n -> n - 2 (pad < n)
#define KEEP_REFLECT(IY, IX, HV, WV)                                   \
  if (p.reflect) {                                                     \
    IY = IY < 0 ? -IY : (IY >= (HV) ? 2 * (HV) - 2 - IY : IY);          \
    IX = IX < 0 ? -IX : (IX >= (WV) ? 2 * (WV) - 2 - IX : IX);          \
  }

// zero-fill of the small atomicMax targets as a KERNEL node: inside a captured hipGraph a hipMemsetAsync node was observed to
// race with the atomics of the kernels that follow it (replays of one graph differed by ~2e-5); a kernel keeps stream order
static __global__ void zero_u32_kernel(unsigned* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// wave-wide max of non-negative floats via their bit patterns, then AT MOST one atomic per wave: same-address atomics retire
// at ~12 ns each in L2 (65k waves of a 1x1 convolution on 4 images: 0.77 ms of atomics around 0.15 ms of work), so a wave
// first reads the running maximum (it only grows: a stale value can cause a redundant atomic, never a missed one) and
// skips the atomic when it cannot raise it -- after the first few waves almost all do.
__device__ __forceinline__ void wave_amax_commit(unsigned* dst, float m) {
  unsigned b = __float_as_uint(m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
  if ((threadIdx.x & 63) == 0 && b > *reinterpret_cast<volatile unsigned*>(dst)) atomicMax(dst, b);
}

// KEEP_MMA_X3 range scaling: the power of two s with amax * s in [2^14, 2^15) (fp16 max 65504), and 1/s.  amax = 0 (or a
// denormal) -> 1; inf / NaN propagate through the data itself.
__device__ __forceinline__ void x3_range_scale(float amax, float& s, float& inv_s) {
  int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (!(amax > 0.f) || e == 255) {
    s = 1.f;
    inv_s = 1.f;
    return;
  }
  e = e < 15 ? 15 : e;
  s = __uint_as_float((unsigned)(268 - e) << 23);
  inv_s = __uint_as_float((unsigned)(e - 14) << 23);
}

__device__ __forceinline__ float epilogue_one(const ConvP& p, float v, long m, int co) {
  if (p.bias) v += p.bias[co];
  v = p.fast ? act_apply_fast(v, p.epi_act) : act_apply(v, p.epi_act);
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

// Epilogue statistics for the NEXT normalisation (GroupNorm / InstanceNorm): every lane owns one output channel
// (column) of its wave tile; (sum, sumsq) over the tile's rows are combined across the two half-waves by a shuffle
// and across the waves that share the columns through `red` (LDS), then written as one partial per (tile, channel).
template <int WGM, int WGN, int TN>
__device__ __forceinline__ void emit_tile_stats(const ConvP& p, float (*red)[2], const float (&cs)[TN], const float (&css)[TN],
                                                int wm, int wn, int lane, int n_img, int p_idx, int n0) {
  const int l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float s = cs[j] + __shfl_xor(cs[j], 32);
    float ss = css[j] + __shfl_xor(css[j], 32);
    if (lane < 32) {
      red[(wm * WGN + wn) * TN * 32 + j * 32 + l31][0] = s;
      red[(wm * WGN + wn) * TN * 32 + j * 32 + l31][1] = ss;
    }
  }
  __syncthreads();
  constexpr int BNc = WGN * TN * 32;
  for (int c = threadIdx.x; c < BNc; c += 256) {
    const int wn_c = c / (TN * 32), rem = c - wn_c * (TN * 32);
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int m = 0; m < WGM; ++m) {
      s += red[(m * WGN + wn_c) * TN * 32 + rem][0];
      ss += red[(m * WGN + wn_c) * TN * 32 + rem][1];
    }
    const int co = n0 + c;
    if (co < p.Cout) {
      float* dst = p.stats + (((long)n_img * p.stats_P + p_idx) * p.Cout + co) * 2;
      dst[0] = s;
      dst[1] = ss;
    }
  }
}

// Epilogue through LDS (shared by the gather kernels): the MFMA C/D layout gives a lane one output channel x 16
// rows -> 4-byte strided stores, issue-bound at ~2 TB/s.  Each wave parks its (TM*32 x TN*32) tile in LDS and reads it
// back channel-contiguous: 16 bytes per lane, full rows per store instruction, float4 bias / residual / aux, and the
// normalisation partial sums are lane-local (4 fixed channels per lane).  Needs Cout, out_ld, res_ld multiples of 4 and
// 16-byte aligned pointers (p.vec_epi); otherwise the scalar path below is used.
// SIMPLE (chosen once per call, uniform): split_k == 1, no aux tensor, no activation -- the row loop then carries neither
// those branches nor the activation switch (worth 10 % on the halo kernel, whose epilogue has the same shape).
template <int WGM, int WGN, int TM, int TN, bool SIMPLE>
__device__ __forceinline__ void staged_epilogue_impl(const ConvP& p, f32x16 (&acc)[TM][TN], float* lds, long m0, int n0,
                                                     int wm, int wn, int lane, int wave, int z) {
  constexpr int WR = TM * 32, WC = TN * 32, EP = WC + 4;
  constexpr int LPR = WC / 4;          // lanes per row (float4 each)
  constexpr int RPI = 64 / LPR;        // rows per wave-instruction
  const int l31 = lane & 31, lhi = lane >> 5;
  float* et = lds + wave * WR * EP;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): wave-local hand-off, LDS ops of one wave retire in order
  const int c4 = (lane % LPR) * 4;
  const int prow = lane / LPR;
  const int co = n0 + wn * WC + c4;
  const bool cok = co < p.Cout;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && p.split_k == 1 && cok) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
  // simple form: the residual rows of a group of four iterations are loaded before the group's first store (inside the loop every load
  // sits behind the previous store -- `res` may alias `out` -- and its vmcnt wait exposes a memory round trip per row)
  float4 rpre[4];
#pragma unroll 4
  for (int it = 0; it < WR / RPI; ++it) {
    if (SIMPLE && p.res && (it & 3) == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long mu = m0 + wm * WR + (it + u) * RPI + prow;
        rpre[u] = (it + u < WR / RPI && mu < p.M && cok) ? *reinterpret_cast<const float4*>(p.res + mu * p.res_ld + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const int px = it * RPI + prow;
    const long m = m0 + wm * WR + px;
    if (m >= p.M || !cok) continue;
    const float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
    if (!SIMPLE && p.split_k > 1) {
      *reinterpret_cast<float4*>(p.ws + ((long)z * p.M + m) * p.Cout + co) = v;
      continue;
    }
    float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
    if (!SIMPLE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
    }
    if (p.res) {
      const float4 r4 = SIMPLE ? rpre[it & 3] : *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
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
    if (p.out_bf16) {
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
      amx = fmaxf(amx, fabsf(e[q]));
    }
  }
  if (p.out_amax) {   // host guarantees split_k == 1 and H*W % (tile rows) == 0: the wave's rows lie in one image
    const long mw = m0 + wm * WR;
    if (mw < p.M) wave_amax_commit(p.out_amax + (int)(mw / ((long)p.Ho * p.Wo)), amx);
  }
  if (p.stats) {   // host guarantees split_k == 1 and H*W % BM == 0 (a tile never straddles two images)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        s4[q] += __shfl_xor(s4[q], o);
        ss4[q] += __shfl_xor(ss4[q], o);
      }
    }
    __syncthreads();                     // all waves finished reading their staged tiles
    float* red = lds;                    // [4 waves][WC][2]
    if (lane < LPR) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        red[(wave * WC + c4 + q) * 2 + 0] = s4[q];
        red[(wave * WC + c4 + q) * 2 + 1] = ss4[q];
      }
    }
    __syncthreads();
    constexpr int BNc = WGN * WC;
    const int hw_o = p.Ho * p.Wo;
    const int n_img = (int)(m0 / hw_o), p_idx = (int)((m0 % hw_o) / (WGM * WR));
    for (int c = threadIdx.x; c < BNc; c += 256) {
      const int wn_c = c / WC, rem = c - wn_c * WC;
      float a = 0.f, b2 = 0.f;
#pragma unroll
      for (int mm = 0; mm < WGM; ++mm) {
        a += red[((mm * WGN + wn_c) * WC + rem) * 2 + 0];
        b2 += red[((mm * WGN + wn_c) * WC + rem) * 2 + 1];
      }
      if (n0 + c < p.Cout) {
        float* dst = p.stats + (((long)n_img * p.stats_P + p_idx) * p.Cout + n0 + c) * 2;
        dst[0] = a;
        dst[1] = b2;
      }
    }
  }
}

template <int WGM, int WGN, int TM, int TN>
__device__ __forceinline__ void staged_epilogue(const ConvP& p, f32x16 (&acc)[TM][TN], float* lds, long m0, int n0,
                                                int wm, int wn, int lane, int wave, int z) {
  if (p.split_k == 1 && !p.aux && p.epi_act == KEEP_ACT_NONE)
    staged_epilogue_impl<WGM, WGN, TM, TN, true>(p, acc, lds, m0, n0, wm, wn, lane, wave, z);
  else
    staged_epilogue_impl<WGM, WGN, TM, TN, false>(p, acc, lds, m0, n0, wm, wn, lane, wave, z);
}


#define HALO_MAXPIX 340                      // 10 x 34 (8x32 tile) >= 18 x 18 (16x16 tile)
#define HPITCH 40                            // bf16 elements per LDS pixel/weight row (80 B)
#define HALO_IT 6                            // ceil(340*4 / 256) 16-byte pieces per thread

__device__ __forceinline__ int xcd_remap(int id, int total) {
  // blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of logical ids so that neighbouring
  // tiles (shared halo rows, shared weight slab) meet in one L2.  Bijective for any total (guide T1).
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, slot = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}


struct HaloItem {
  int n, tx, ty, oy0, ox0, n0, z, ch_begin, ch_end;
};

template <int TW, int CSH = 5>
__device__ __forceinline__ HaloItem halo_decode(const ConvP& p, int item, int items_per_z, int tiles_x, int tiles_y, int ncb) {
  HaloItem it;
  it.z = item / items_per_z;
  const int lid = xcd_remap(item - it.z * items_per_z, items_per_z);
  const int cb = lid % ncb;
  int t = lid / ncb;
  it.tx = t % tiles_x; t /= tiles_x;
  it.ty = t % tiles_y;
  it.n = t / tiles_y;
  it.oy0 = it.ty * (256 / TW);
  it.ox0 = it.tx * TW;
  it.n0 = cb * 64;
  const int nchunks = p.Cin >> CSH;
  const int per = (nchunks + p.split_k - 1) / p.split_k;
  it.ch_begin = it.z * per;
  it.ch_end = min(nchunks, it.ch_begin + per);
  return it;
}

#define KEEP_TAPS(X) X(0, wr0) X(1, wr1) X(2, wr2) X(3, wr3) X(4, wr4) X(5, wr5) X(6, wr6) X(7, wr7) X(8, wr8)
