// keep_conv2d, KEEP_MMA_X3, GEMM form (1x1, stride 1) with FEW ROWS PER IMAGE (<= 256): the latency form.
//
// The token GEMMs of the frame recurrence (code transformer KA:385-439, AttnBlock projections VQ:219-243, CFA KA:519-541) have
// 256 rows per image and K = 256 .. 2048.  With one clip in flight (the literal configs[1]) conv_x3_kernel spends 18 .. 57 us
// on each of them whatever the FLOP count: 32 .. 128 blocks walk K in steps of 32 with one step of prefetch, i.e. 16 .. 64 DEPENDENT
// round trips to L2 / HBM (the weights of a frame do not fit any cache: 633 MB per frame).  Here the K axis is cut into NW
// CANONICAL SLICES, one per wave of a block: a wave requests its whole slice of A (fp32 rows, straight into the MFMA fragment
// layout: lane = row, 8 consecutive channels) and of B (the pre-split weight rows, [hi16 | lo16] per 16-channel chunk) up front --
// one round trip -- splits A in registers, runs its 3 * KS / 16 MFMAs per 32 x 32 tile from a ZERO accumulator and parks the partial
// tile in LDS; after one barrier the block adds the NW partials in slice order 0, 1, .., NW - 1 and runs the epilogue (accumulator
// scale, bias, activation, residual, fused max|out|) with 16-byte row-contiguous stores.  No LDS operand staging, no K loop barrier.
//
// Numerics: the value of an output element is defined by the slicing alone (NW = 8 for K >= 1024, 4 below; slice s covers channels
// [s K / NW, (s + 1) K / NW) in order) -- NOT by the row tile (TM), the grid order or the batch: plan_conv selects this kernel from the
// per-image geometry, so a clip's bits never depend on its batch-mates, and the launch is free to pick TM / the XCD order from the real M.
#include "keep_conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GL_G 4                      // k-steps (of 16 channels) per register group: 64 channels
#define GL_EP 36                    // floats per parked row: 32 + 4 pad
// Round 6: operands reach the fragment layout through LDS.  A fragment-layout load (lane = row, 32 bytes) touches 32 cache lines per
// instruction: 65 cycles of the CU's texture-address path instead of 16 -- ~4 of the 5.9 us of a warm launch (profiles/r05_gemm_forms.txt).
// A wave now loads its 32 rows x 64 channels of A (and the 32 rows x 4 chunks of pre-split B) with 16-byte pieces CONSECUTIVE across lanes
// (16 lanes = one 256-byte row run: 8 lines per instruction), parks them in its private LDS rows at a 272-byte pitch (68 floats: the eight
// rows a ds_read_b128 phase touches fall on disjoint banks) and reads the MFMA fragments back: same values in the same registers, same
// MFMA order -- same bits.  KEEP_GL_DIRECT=1 (dev builds) keeps the direct fragment loads.
#define GL_ROWB (GL_G * 64 + 16)    // bytes per staged row
// rows per image the family takes: the 16 x 16 token maps (code transformer, AttnBlock).  The 32 x 32 maps (CFA: 135 launches per clip)
// were measured too: -0.5 ms per clip with one clip in flight, +1.8 ms per 16-clip step (64 x 64 tiles with slice totals against the
// 128 x 128 tile of the sequential sum) -- left on conv_x3_kernel.
#ifndef GL_MAX_HW
#define GL_MAX_HW 256
#endif
#define GL_MAX_TILES 512            // launches of more 32 x 32 output tiles run on conv_x3_kernel with canonical slices (same bits)

template <int NW, int TM, bool PLAIN>
__global__ __launch_bounds__(NW * 64) void gemm_x3l_kernel(ConvP p, int m_fast) {
  constexpr int BM = TM * 32, NT = NW * 64;
  constexpr int STG_W = (TM + 1) * 32 * GL_ROWB;            // staging bytes per wave: TM x 32 rows of A + 32 rows of B
  constexpr int LDS_B = NW * STG_W > NW * BM * GL_EP * 4 ? NW * STG_W : NW * BM * GL_EP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_B];
  float* part = reinterpret_cast<float*>(lds_raw);          // the parked partial tiles reuse the staging rows (one barrier in between)
  __shared__ __attribute__((aligned(16))) float pro_s[PLAIN ? 4 : 2 * 2048];      // GroupNorm (scale | shift) of the tile's image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // the faster grid index walks the LARGER operand's tiles: each of the 8 XCDs (dealt round-robin by linear block id) fetches 1/8 of it
  const int bm = m_fast ? blockIdx.x : blockIdx.y, bn = m_fast ? blockIdx.y : blockIdx.x;
  const long m0 = (long)bm * BM;
  const int n0 = bn * 32;
  const int hw = p.Ho * p.Wo;
  const int n_img = (int)(m0 / hw);                      // hw % BM == 0 (plan): the tile lies in one image
  const int K = p.Cin, KS = K / NW, NG = KS / (16 * GL_G);
  const int k0 = wave * KS;

  const long rows_here = (p.M - m0) < BM ? (p.M - m0) : BM;
  auto make_rsrc = [&](const void* ptr, long bytes) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.in + m0 * p.in_ld, ((rows_here - 1) * p.in_ld + K) * 4);
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wx3 + (long)n0 * K * 2, 32L * K * 4);
  // piece j of a group: rows 4 j + (lane >> 4), bytes (lane & 15) * 16 of the row's 256-byte run of the group
  const int r4 = lane >> 4, c16 = (lane & 15) * 16;
  const int a_voff = r4 * p.in_ld * 4 + c16;             // + j * 4 rows (+ i * 32 rows); the group's channel offset in the scalar operand
  const int b_voff = r4 * K * 4 + c16;                   // weight row = K/16 chunks of [hi16 | lo16] = 64 B: a group is 256 B of it
  unsigned char* stg_a = lds_raw + wave * STG_W;
  unsigned char* stg_b = stg_a + TM * 32 * GL_ROWB;
  const int st_off = r4 * GL_ROWB + c16;                 // where this lane parks piece j (+ j * 4 rows)

  struct Grp {
    u32x4 a[TM][8];
    u32x4 b[8];
  };
  auto fetch = [&](int g, Grp& R) {
    const int kc = k0 + g * 16 * GL_G;                   // first channel of the group (wave-uniform)
#pragma unroll
    for (int j = 0; j < 8; ++j) R.b[j] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff + j * 4 * K * 4, kc * 4, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        R.a[i][j] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff + (i * 32 + j * 4) * p.in_ld * 4, kc * 4, 0);
  };

  Grp R0, R1;
  fetch(0, R0);
  if (NG > 1) fetch(1, R1);
  float a_s = 1.f, a_inv = 1.f;
  if (p.in_amax) x3_range_scale(p.in_amax[n_img], a_s, a_inv);
  if (!PLAIN) {      // the image's (scale, shift) rows: global -> LDS under the operand loads
    for (int c = tid; c < K; c += NT) {
      pro_s[c] = p.pro_scale ? p.pro_scale[(long)n_img * K + c] : 1.f;
      pro_s[2048 + c] = p.pro_shift ? p.pro_shift[(long)n_img * K + c] : 0.f;
    }
    __syncthreads();
  }

  f32x16 acc[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  auto compute = [&](int g, Grp& R) {
    const int kc = k0 + g * 16 * GL_G + lhi * 8;
    // park the group (this wave's rows only: LDS serves a wave's requests in order, no barrier), then read it back in the fragment layout
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(stg_b + st_off + j * 4 * GL_ROWB) = R.b[j];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(stg_a + st_off + (i * 32 + j * 4) * GL_ROWB) = R.a[i][j];
#pragma unroll
    for (int ks = 0; ks < GL_G; ++ks) {
      const f16x8 bh = *reinterpret_cast<const f16x8*>(stg_b + l31 * GL_ROWB + ks * 64 + lhi * 16);
      const f16x8 bl = *reinterpret_cast<const f16x8*>(stg_b + l31 * GL_ROWB + ks * 64 + 32 + lhi * 16);
      float sc[8], sh[8];
      if (!PLAIN) {
        const float4 s0 = *reinterpret_cast<const float4*>(&pro_s[kc + ks * 16]);
        const float4 s1 = *reinterpret_cast<const float4*>(&pro_s[kc + ks * 16 + 4]);
        const float4 h0 = *reinterpret_cast<const float4*>(&pro_s[2048 + kc + ks * 16]);
        const float4 h1 = *reinterpret_cast<const float4*>(&pro_s[2048 + kc + ks * 16 + 4]);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[8];
        {
          const float4 v0 = *reinterpret_cast<const float4*>(stg_a + (i * 32 + l31) * GL_ROWB + ks * 64 + lhi * 32);
          const float4 v1 = *reinterpret_cast<const float4*>(stg_a + (i * 32 + l31) * GL_ROWB + ks * 64 + lhi * 32 + 16);
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        }
        if (!PLAIN) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = v[j] * sc[j] + sh[j];
            if (p.pro_act != KEEP_PRO_NONE) v[j] = p.fast ? pro_apply_x3(v[j], p.pro_act) : pro_apply(v[j], p.pro_act);
          }
        }
        f16x8 ah, al;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const f32x2 vs = f32x2{v[j], v[j + 1]} * a_s;
          const f16x2 h = __builtin_convertvector(vs, f16x2);
          const f16x2 l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2);
          ah[j] = h.x; ah[j + 1] = h.y;
          al[j] = l.x; al[j + 1] = l.y;
        }
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
      }
    }
  };

  for (int g = 0; g < NG; g += 2) {
    compute(g, R0);
    if (g + 2 < NG) fetch(g + 2, R0);
    if (g + 1 < NG) {
      compute(g + 1, R1);
      if (g + 3 < NG) fetch(g + 3, R1);
    }
  }

  // park the slice's partial tile: part[wave][row][col] -- over the staging rows, once every wave is past its last fragment read
  __syncthreads();
  float* mine = part + wave * BM * GL_EP;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * GL_EP + l31] = acc[i][r];
  __syncthreads();

  // slice-ordered sum + epilogue: one float4 (row, 4 channels) per thread and pass
  const float post = p.acc_scale * a_inv;
  float amx = 0.f;
  constexpr int UNITS = BM * 8;
#pragma unroll
  for (int u0 = 0; u0 < UNITS; u0 += NT) {
    const int u = u0 + tid;
    if (UNITS % NT != 0 && u >= UNITS) break;
    const int row = u >> 3, c4 = (u & 7) * 4;
    float4 v = *reinterpret_cast<const float4*>(part + row * GL_EP + c4);
#pragma unroll
    for (int s = 1; s < NW; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(part + (s * BM + row) * GL_EP + c4);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const long m = m0 + row;
    const int co = n0 + c4;
    if (m >= p.M) continue;
    float e[4] = {v.x * post, v.y * post, v.z * post, v.w * post};
    if (p.bias) {
      const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co);
      e[0] += b4.x; e[1] += b4.y; e[2] += b4.z; e[3] += b4.w;
    }
    if (p.epi_act != KEEP_ACT_NONE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
    }
    if (p.res) {
      const float4 r4 = *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co);
      e[0] += r4.x; e[1] += r4.y; e[2] += r4.z; e[3] += r4.w;
    }
    *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(e[q]));
  }
  if (p.out_amax) wave_amax_commit(p.out_amax + n_img, amx);
}

// Geometry of the latency form (a per-image rule: nothing here looks at N).
bool keep_gemm_x3l_ok(const keep_conv2d_args* a) {
  const long hw = (long)a->Ho * a->Wo;
  const int K = a->Cin;
  const int nw = K >= 1024 ? 8 : 4;
  // (GEMMs with a GroupNorm prologue -- the AttnBlock qkv projection, 117 launches per clip -- stay on conv_x3_kernel's sequential sum: the
  // latency form gains 0.8 ms per clip there, the slice totals of the prologue form cost the 16-clip step 5.5 ms: 46.8 -> 91 us per launch)
  return !a->pro_scale && a->pro_act == KEEP_PRO_NONE && a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad_t == 0 && a->pad_l == 0 && a->Ho == a->H && a->Wo == a->W && !a->upsample &&
         hw >= 64 && hw <= GL_MAX_HW && hw % 64 == 0 && K >= 256 && K <= 2048 && K % (nw * 16 * GL_G) == 0 && a->Cout % 32 == 0 && !a->in2 && !a->aux &&
         !a->ln_gamma && a->split_k <= 1 && a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && a->in_ld % 4 == 0 && (uintptr_t)a->in % 16 == 0 &&
         a->out_ld % 4 == 0 && (uintptr_t)a->out % 16 == 0 && (!a->residual || (a->res_ld % 4 == 0 && (uintptr_t)a->residual % 16 == 0)) &&
         (!a->bias || (uintptr_t)a->bias % 16 == 0) && (long)a->in_ld * 4 * 64 + (long)K * 4 < (1L << 31) && (long)K * 4 * 32 < (1L << 31);
}

// slices of 128 channels for K = 512 .. 1024 (4 / 8 waves), 256 for K = 2048, 64 for K = 256: the throughput form folds a slice total
// every 4 K steps at least where the launches are many (conv_x3_kernel KSL: 16 adds per 24 MFMAs and wave)
int keep_gemm_x3l_waves(const keep_conv2d_args* a) { return a->Cin >= 1024 ? 8 : 4; }

int keep_conv2d_x3_gather(const keep_conv2d_args* a, ConvP& p, int tile, hipStream_t st);

// The sums are defined by the K slicing; WHICH kernel evaluates them follows the real row count (bit-neutral): the latency form while
// its 32 x 32 tiles leave CUs idle, conv_x3_kernel with canonical slices (p.kslice_steps: 64 x 64 or 128 x 128 block tiles, operands
// staged through LDS -- 4 x less L2 -> CU traffic per FLOP) beyond.
int keep_conv2d_x3_gemm_lat(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  const long M = p.M;
  const bool plain = !a->pro_scale && a->pro_act == KEEP_PRO_NONE;
  const int nw = keep_gemm_x3l_waves(a);
  const long tiles32 = (M / 32) * (a->Cout / 32);
  const bool by_tiles = (a->flags & KEEP_CONV_GEMM_LAT_TILES) || (tiles32 > GL_MAX_TILES && !(a->flags & KEEP_CONV_GEMM_LAT_WAVES));
  if (by_tiles) {
    p.kslice_steps = a->Cin / nw / 32;
    p.split_k = 1;
    return keep_conv2d_x3_gather(a, p, 1, st);      // 64 x 64 tiles at every row count: the slice totals of a 128 x 128 tile leave one block per CU
  }
  const int gm = cdiv(M, 32), gn = a->Cout / 32;
  const int m_fast = (long)M > (long)a->Cout ? 1 : 0;       // the faster index walks the LARGER operand: each XCD sees 1/8 of it
  dim3 grid(m_fast ? gm : gn, m_fast ? gn : gm);
#define KEEP_LAUNCH_GL(NWV)                                                                                \
  if (!plain)                                                                                              \
    hipLaunchKernelGGL((gemm_x3l_kernel<NWV, 1, false>), grid, dim3(NWV * 64), 0, st, p, m_fast);          \
  else                                                                                                     \
    hipLaunchKernelGGL((gemm_x3l_kernel<NWV, 1, true>), grid, dim3(NWV * 64), 0, st, p, m_fast);
  if (nw == 8) {
    KEEP_LAUNCH_GL(8)
  } else {
    KEEP_LAUNCH_GL(4)
  }
#undef KEEP_LAUNCH_GL
  KEEP_LAUNCH_CHECK("keep_conv2d(gemm x3 latency form)");
  return KEEP_OK;
}
