// keep_gm_ffn_x3 -- the GMFlow feed-forward block under KEEP_MMA_X3 as ONE kernel (GM/transformer.py:137-142,182-187):
//
//     out = LayerNorm( W2 . gelu( W0 . cat[src | msg] ) ; gamma, beta, eps ) + src          src, msg, out [M, C],  C = 128
//
// W0 [8C, 2C], W2 [C, 8C], no biases.  The unfused form is two launches of conv_x3_kernel -- `mlp.0` (2C -> 8C, GELU epilogue)
// writes a [M, 8C] fp32 intermediate that `mlp.2` (8C -> C, LayerNorm epilogue) reads back: 2 x 10.2 GB of HBM traffic per layer at
// 16 clips x T = 20 (2.49 M tokens), 69 ms of a 1183 ms step at 146 / 254 TF (profiles/r04_conv_census_x3_b16.txt).  Here the
// intermediate never leaves the registers.
//
// Decomposition (one block = 8 waves = 256 tokens, one block per CU, 2 waves per SIMD):
//   * TOKENS RIDE THE N AXIS OF THE MFMAs.  A wave owns 32 tokens and keeps its whole X^T = cat[src | msg]^T, split into fp16 hi / lo,
//     as the B operand of v_mfma_f32_32x32x16_f16 in registers for the life of the block: 16 K-steps x (hi, lo) x 4 = 128 VGPRs.
//   * The hidden units are walked in chunks of 32.  GEMM 1: C1[32 hidden x 32 tokens] = W0[chunk, :] . X^T  -- A = 32 weight rows from
//     LDS (16 K-steps x hi / lo fragments), 48 MFMAs in the x3 term order of conv_x3_kernel (a_lo b_hi, a_hi b_lo, a_hi b_hi), so C1
//     carries the same sums as the unfused kernel's accumulators.
//   * An MFMA's C layout hands lane (token n, half g) the 16 hidden units {4g + (r & 3) + 8 (r >> 2)} of its token.  A contraction may
//     enumerate its K axis in any order as long as both operands agree, so those 16 values ARE the B operand of GEMM 2 for two K = 16
//     steps (k-slot e of step s <-> register r = 8 s + e): scale, GELU, split into hi / lo in place -- no shuffle, no LDS round trip.
//     The matching A operand, W2[:, chunk], is stored by the host with each group of 16 hidden units permuted
//     [0-3, 8-11, 4-7, 12-15] (engine/ops.py:ffn_w2_twin), which makes a lane's 8 k-slots one 16-byte LDS read.
//   * GEMM 2: Y^T[128 out x 32 tokens] += W2p[:, chunk] . H^T: 4 row tiles x 2 K-steps x 3 = 24 MFMAs; Y^T lives in 64 VGPRs.
//   * Weights: per chunk 32 KB of W0 + 16 KB of W2 go L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 6 x 1 KB per wave), rows at a
//     64-byte pitch with the 16-byte pieces XOR-swizzled by (row >> 2) & 3 through the SOURCE address (conflict-free ds_read_b128,
//     keep_conv_x3.hip), two stages: chunk c + 1 lands under the 72 MFMAs per wave of chunk c; one s_barrier per chunk.
//   * The GELU (16 values per lane and chunk: ~5 VALU per MFMA) is not interleaved by hand: the second wave of the SIMD issues its
//     MFMAs meanwhile (tools/dev/coissue_probe2.hip: two waves per SIMD keep the matrix pipe at its bare rate with this much VALU).
//   * Epilogue: a token's 128 outputs sit in lanes n and n + 32 (64 each): LayerNorm is 64 in-lane terms + one v_permlane32_swap,
//     two passes (mean, centred squares: nn.LayerNorm's biased variance), then + src and 16-byte stores.
// Registers: 128 (X) + 64 (Y) + 16 (C1 / H) + fragments: the kernel is written against the 256-VGPR budget of two waves per SIMD.
#include "keep_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define FFN_C 128                 // model width
#define FFN_K1 (2 * FFN_C)        // GEMM 1 reduction: cat[src | msg]
#define FFN_TOK 256               // tokens per block
#define FFN_HC 32                 // hidden units per chunk
#define FFN_W0_B (16 * 32 * 64)   // LDS bytes of a W0 chunk: [K-step 16][row 32][hi16 | lo16]
#define FFN_W2_B (2 * 128 * 64)   // LDS bytes of a W2 piece: [K-step 2][out row 128][hi16 | lo16]
#define FFN_STAGE (FFN_W0_B + FFN_W2_B)
#define FFN_LDS (2 * FFN_STAGE)

struct FfnP {
  const float* src;
  const float* msg;
  float* out;
  const unsigned short* w0;      // x3 twin of W0: [hidden][2C / 16][hi16 | lo16]
  const unsigned short* w2p;     // x3 twin of W2 with every 16-group of hidden units permuted: [C][hidden / 16][hi16 | lo16]
  const float* gamma;
  const float* beta;
  float eps, asc0, asc2;
  long M;
  int hidden;
};

#define FFN_MMA_X3(ACC, AH, AL, BH, BL)                                          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);            \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);            \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0);

__device__ __forceinline__ float ffn_xor32_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ void ffn_split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const f32x2 vs = f32x2{v[j], v[j + 1]};
    const f16x2 h = __builtin_convertvector(vs, f16x2);
    const f16x2 l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2);
    hi[j] = h.x; hi[j + 1] = h.y;
    lo[j] = l.x; lo[j + 1] = l.y;
  }
}

template <bool FAST>
__global__ __launch_bounds__(512, 1) void gm_ffn_x3_kernel(FfnP p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const long blk0 = (long)blockIdx.x * FFN_TOK;
  const long rows_here = (p.M - blk0) < FFN_TOK ? (p.M - blk0) : FFN_TOK;
  const int nchunks = p.hidden / FFN_HC;

  auto make_rsrc = [&](const void* ptr, long bytes) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  // block-relative descriptors: rows beyond M read zeros and are never written (the range check of the descriptor)
  const __amdgpu_buffer_rsrc_t src_rsrc = make_rsrc(p.src + blk0 * FFN_C, rows_here * FFN_C * 4);
  const __amdgpu_buffer_rsrc_t msg_rsrc = make_rsrc(p.msg + blk0 * FFN_C, rows_here * FFN_C * 4);
  const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(p.out + blk0 * FFN_C, rows_here * FFN_C * 4);
  const __amdgpu_buffer_rsrc_t w0_rsrc = make_rsrc(p.w0, (long)p.hidden * FFN_K1 * 4);
  const __amdgpu_buffer_rsrc_t w2_rsrc = make_rsrc(p.w2p, (long)FFN_C * p.hidden * 4);

  // ---- weight DMA: lane l of a 1 KB piece carries row (l >> 2) of 16 rows, LDS piece l & 3 <- source piece (l & 3) ^ ((l >> 4) & 3)
  const int sw = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
  const int w0_voff = (lane >> 2) * (FFN_K1 * 4) + sw;          // W0 rows are 2C x 4 B = 1 KB apart
  const int w2_voff = (lane >> 2) * (p.hidden * 4) + sw;        // W2 rows are hidden x 4 B apart
  auto dma_chunk = [&](int c, int stage) __attribute__((always_inline)) {
    unsigned char* base = lds + stage * FFN_STAGE;
    // W0: 32 pieces q = 2 j + mh (K-step j, row half mh); wave w carries q = 4 w .. 4 w + 3
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int q = wave * 4 + u;
      const int j = q >> 1, mh = q & 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w0_rsrc, (__attribute__((address_space(3))) void*)(base + q * 1024), 16, w0_voff,
                                               (c * FFN_HC + mh * 16) * (FFN_K1 * 4) + j * 64, 0, 0);
    }
    // W2: 16 pieces q = 8 s + mq (K-step s, 16 out rows mq); wave w carries q = 2 w, 2 w + 1
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = wave * 2 + u;
      const int s = q >> 3, mq = q & 7;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w2_rsrc, (__attribute__((address_space(3))) void*)(base + FFN_W0_B + q * 1024), 16, w2_voff,
                                               mq * 16 * (p.hidden * 4) + (c * FFN_HC + s * 16) * 4, 0, 0);
    }
  };
  dma_chunk(0, 0);

  // ---- X^T fragments of this wave's 32 tokens: lane (token l31, half g) holds channels 16 j + 8 g .. + 8 of K-step j
  f16x8 xh[16], xl[16];
  {
    const int row_off = (wave * 32 + l31) * (FFN_C * 4) + g * 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const __amdgpu_buffer_rsrc_t& rs = j < 8 ? src_rsrc : msg_rsrc;
      const int off = row_off + (j & 7) * 64;
      const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
      const float v[8] = {__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w),
                          __uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w)};
      ffn_split8(v, xh[j], xl[j]);
    }
  }

  f32x16 y[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) y[t][r] = 0.f;

  // fragment addresses inside a stage: row m, piece (hi: g, lo: 2 + g) ^ ((m >> 2) & 3)
  const int sw_r = (l31 >> 2) & 3;
  const int a1_hi = l31 * 64 + ((g ^ sw_r) * 16);               // + j * 2048
  const int a1_lo = l31 * 64 + (((2 + g) ^ sw_r) * 16);
  // (rows 32 t + l31 of W2: (m >> 2) & 3 is the same for every tile t)
  const int a2_hi = FFN_W0_B + l31 * 64 + ((g ^ sw_r) * 16);    // + s * 8192 + t * 2048
  const int a2_lo = FFN_W0_B + l31 * 64 + (((2 + g) ^ sw_r) * 16);

  // GEMM 1 of chunk `c` out of LDS stage `st`: C1[hidden 32 x tokens 32] over K = 2C
  auto gemm1 = [&](const unsigned char* st, f32x16& c1) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c1[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(st + a1_hi + j * 2048);
      const f16x8 al = *reinterpret_cast<const f16x8*>(st + a1_lo + j * 2048);
      FFN_MMA_X3(c1, ah, al, xh[j], xl[j])
    }
  };
  // scale, GELU, split (the lane's 16 hidden values are GEMM 2's B operand for two K-steps: k-slot e of step s = register 8 s + e),
  // then GEMM 2: Y^T[out 128 x tokens 32] += W2p[:, chunk] . H^T
  auto gelu_gemm2 = [&](const unsigned char* st, const f32x16& c1) __attribute__((always_inline)) {
    f16x8 hh[2], hl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = c1[8 * s + e] * p.asc0;
        v[e] = FAST ? act_apply_fast(a, KEEP_ACT_GELU) : act_apply(a, KEEP_ACT_GELU);
      }
      ffn_split8(v, hh[s], hl[s]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(st + a2_hi + s * 8192 + t * 2048);
        const f16x8 al = *reinterpret_cast<const f16x8*>(st + a2_lo + s * 8192 + t * 2048);
        FFN_MMA_X3(y[t], ah, al, hh[s], hl[s])
      }
  };

  // (Round 5 also built a SKEWED form -- the two waves of a SIMD half a chunk apart through a third LDS stage and two barriers per chunk, so
  // that one wave's GELU always meets the other's first product on the matrix pipe: bit-identical, and 4.5 % SLOWER (6.54 vs 6.26 ms at
  // 2.49 M tokens, profiles/r05_ffn_skew_ab.txt): the waves do not run in lockstep to begin with, and the kernel is bound by its LDS
  // fragment reads -- every wave reads the whole W0 chunk, 0.67 ds_read_b128 per MFMA -- next to the matrix pipe, not by VALU.)
  for (int c = 0; c < nchunks; ++c) {
    const int stage = c & 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's pieces of chunk c have landed (and, first time, its X rows)
    __syncthreads();                           // every wave's pieces are visible; every wave has left chunk c - 1 (stage ^ 1 is free)
    if (c + 1 < nchunks) dma_chunk(c + 1, stage ^ 1);
    const unsigned char* st = lds + stage * FFN_STAGE;
    f32x16 c1;
    gemm1(st, c1);
    gelu_gemm2(st, c1);
  }

  // ---- epilogue: LayerNorm over the token's 128 outputs (64 here, 64 in lane ^ 32), + src, store
  float sm = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      y[t][r] *= p.asc2;
      sm += y[t][r];
    }
  const float mean = ffn_xor32_sum(sm) * (1.0f / FFN_C);
  float qq = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      y[t][r] -= mean;
      qq += y[t][r] * y[t][r];
    }
  const float rstd = 1.0f / sqrtf(ffn_xor32_sum(qq) * (1.0f / FFN_C) + p.eps);
  const int tok_off = (wave * 32 + l31) * (FFN_C * 4);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * t + 8 * q + 4 * g;          // register r = 4 q + i holds out channel ch + i
      const float4 gm = *reinterpret_cast<const float4*>(p.gamma + ch);
      const float4 bt = *reinterpret_cast<const float4*>(p.beta + ch);
      const u32x4 r4 = __builtin_amdgcn_raw_buffer_load_b128(src_rsrc, tok_off + ch * 4, 0, 0);
      u32x4 o;
      o.x = __float_as_uint(y[t][4 * q + 0] * rstd * gm.x + bt.x + __uint_as_float(r4.x));
      o.y = __float_as_uint(y[t][4 * q + 1] * rstd * gm.y + bt.y + __uint_as_float(r4.y));
      o.z = __float_as_uint(y[t][4 * q + 2] * rstd * gm.z + bt.z + __uint_as_float(r4.z));
      o.w = __float_as_uint(y[t][4 * q + 3] * rstd * gm.w + bt.w + __uint_as_float(r4.w));
      __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, tok_off + ch * 4, 0, 0);
    }
}

extern "C" int32_t keep_gm_ffn_x3(const float* src, const float* msg, const void* w0_x3, float w0_acc_scale, const void* w2p_x3,
                                  float w2_acc_scale, const float* ln_gamma, const float* ln_beta, float ln_eps, float* out, int64_t M,
                                  int32_t C, int32_t hidden, int32_t exact_act, void* stream) {
  KEEP_REQUIRE(src && msg && w0_x3 && w2p_x3 && ln_gamma && ln_beta && out && M > 0, "keep_gm_ffn_x3: bad args");
  KEEP_REQUIRE(C == FFN_C && hidden > 0 && hidden % FFN_HC == 0, "keep_gm_ffn_x3: built for C = %d and hidden %% %d == 0 (got %d, %d)",
               FFN_C, FFN_HC, C, hidden);
  KEEP_REQUIRE((uintptr_t)src % 16 == 0 && (uintptr_t)msg % 16 == 0 && (uintptr_t)w0_x3 % 16 == 0 && (uintptr_t)w2p_x3 % 16 == 0 &&
                   (uintptr_t)out % 16 == 0 && (uintptr_t)ln_gamma % 16 == 0 && (uintptr_t)ln_beta % 16 == 0,
               "keep_gm_ffn_x3: 16-byte alignment");
  KEEP_REQUIRE((long)hidden * FFN_K1 * 4 < (1L << 31) && (long)FFN_C * hidden * 4 < (1L << 31), "keep_gm_ffn_x3: weight tensors beyond 2 GB");
  FfnP p;
  p.src = src; p.msg = msg; p.out = out;
  p.w0 = (const unsigned short*)w0_x3; p.w2p = (const unsigned short*)w2p_x3;
  p.gamma = ln_gamma; p.beta = ln_beta;
  p.eps = ln_eps; p.asc0 = w0_acc_scale; p.asc2 = w2_acc_scale;
  p.M = M; p.hidden = hidden;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gm_ffn_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gm_ffn_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS);
    attr_set = true;
  }
  const dim3 grid((unsigned)((M + FFN_TOK - 1) / FFN_TOK)), block(512);
  if (exact_act)
    hipLaunchKernelGGL(gm_ffn_x3_kernel<false>, grid, block, FFN_LDS, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(gm_ffn_x3_kernel<true>, grid, block, FFN_LDS, (hipStream_t)stream, p);
  KEEP_LAUNCH_CHECK("keep_gm_ffn_x3");
  return KEEP_OK;
}
