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
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split4(const float (&v)[4], f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)(v[j] - (float)h);
  }
}

// (A hand-rolled x*sigmoid(x) -- exp2 with a two-float range reduction + v_rcp and one Newton step -- was measured SLOWER in
// this staging step than the compiler's expf + IEEE division: 316 vs 279 us on 128 ch @256^2; the library forms stay.)
template <int PRO, bool FAST>
__device__ __forceinline__ float pro_x3(float v) {
  return FAST ? pro_apply_x3(v, PRO) : pro_apply(v, PRO);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// x + (the value 16 / 32 lanes across): gfx950 lane-swap instructions, one VALU op each (ds_bpermute is an LDS round trip)
__device__ __forceinline__ float xor16_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float xor32_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
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
// 2: no MFMAs, 3: no staging (LDS keeps stale data), 4: no global stores in the epilogue, 5: no operand fetch,
// 6: start stagger between the two blocks of a CU, 7: library expf + IEEE division in the swish prologue, 8: s_setprio(1)
// around the MFMA loop, 10: affine-only prologue (no transcendentals), 11: no epilogue, 12: weight DMA not waited for, 13: no
// weight DMA, 14: halo rows computed but not written to LDS, 15: weight DMA from one 1 KB source (L1 hits), 16: no weight DMA and the
// B fragments read from the halo rows, 17: the product kernel with the block cycle counter (KEEP_X3_CYC=1 prints it for every variant).
// WDMA: the 9 x 64 pre-split weight rows of a chunk go from L2 straight into LDS (buffer_load_dwordx4 ... lds, 1 KB per wave
// instruction, no VGPR round trip, no ds_write): rows at a 64-byte pitch, the 16-byte pieces of a row XOR-swizzled by
// (row >> 2) & 3 through the SOURCE address of each lane (an LDS-DMA destination is lane-linear), which keeps the B-fragment
// ds_read_b128 conflict-free without padding.  Issued after the barrier that ends a chunk's MFMA phase, landed (vmcnt(0)) under the
// VALU staging of the halo.
// UP2 (p.up2): nearest-x2 upsample + 3x3 convolution as FOUR 2x2-tap convolutions on the SOURCE grid, one per output parity (py, px):
//   out[2y+py, 2x+px] = sum over the 2x2 source neighbourhood of the pre-added weights (row y-1: w[0], row y: w[1]+w[2] for py = 0;
//   row y: w[0]+w[1], row y+1: w[2] for py = 1; columns alike) -- 4 of 9 taps per output instead of 9: the x2-replicated pixels of the
//   upsampled tensor multiply the same source value.  The weight tensor holds the four phase kernels as 4*Cout virtual cout rows
//   ([phase][Cout][9 taps][Cin/16][hi16|lo16], unused taps zero and never fetched); an item is (source tile, phase, cout block), its
//   MFMA loop is compiled per phase (static tap list), its epilogue scatters to the stride-2 output pixels.  Tiles, halo and
//   addressing are those of a plain 3x3 convolution on the source (p.upsample = 0, p.Ho / p.Wo = the OUTPUT extent).
template <int TW, int PRO, bool SIMPLE_EPI, int EXP = 0, bool FASTACT = true, bool WDMA = false, bool UP2 = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_x3_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  static_assert(!UP2 || (TW == 32 && WDMA && SIMPLE_EPI && PRO == KEEP_PRO_NONE), "UP2: wide tiles, DMA weights, simple epilogue, no prologue");
  constexpr int HALO_TH = 256 / TW, HALO_W = TW + 2, HALO_PIX = (HALO_TH + 2) * HALO_W;
  constexpr int RPT = 32 / TW;
  constexpr int MAIN_B = (HALO_MAXPIX + 9 * 64) * XPITCH * 2;
  constexpr int EPI_B = 4 * 64 * 68 * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[MAIN_B > EPI_B ? MAIN_B : EPI_B];
  _Float16* Hs = reinterpret_cast<_Float16*>(lds_raw);
  _Float16* Ws = Hs + HALO_MAXPIX * XPITCH;
  unsigned char* const wdma_base = lds_raw + HALO_MAXPIX * XPITCH * 2;      // == Ws as bytes
  int fetched_ch = 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int items_per_z = n_items / p.split_k;
  const int Hv = p.upsample ? 2 * p.H : p.H;
  const int Wv = p.upsample ? 2 * p.W : p.W;
  const int g = tid & 3;
  const bool has_pro = p.pro_scale != nullptr || PRO != KEEP_PRO_NONE;

  // Operand fetch through buffer descriptors: ONE buffer_load_dwordx4 per 16-byte piece, no 64-bit address arithmetic and no
  // EXEC branches -- a padding pixel / a cout row beyond Cout carries the out-of-range offset -16 and the hardware returns
  // zeros (bounds check on voffset; the channel / tap displacement rides in the scalar offset).  (The flat-load form spent
  // ~185 instructions per chunk here, 15 % of the wave's time by the s_memtime timeline.)
  int h_voff[HALO_IT];                   // byte offset of this thread's piece inside the image; < 0: zero padding
  int w_voff = -16;                      // byte offset of this thread's piece of its cout row in the split weight tensor
  int dma_voff[4] = {-16, -16, -16, -16};   // WDMA: this lane's source offset for the four 16-cout row groups of a tap
  long sc_off = 0;
  float in_s = 1.f, in_inv = 1.f;        // range scale of the item being FETCHED / staged (image it.n)
  __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wx3, 0, (UP2 ? 4 : 1) * p.Cout * 9 * p.Cin * 4, 0x00020000);
  const int cout_rows = (UP2 ? 4 : 1) * p.Cout;      // weight rows (UP2: four phase kernels)
  const int ncb_real = p.Cout >> 6;                  // UP2: cout blocks per phase (Cout % 64 == 0)
  float4 bias_nx = make_float4(0.f, 0.f, 0.f, 0.f);
  float amax_raw = 0.f;
  auto setup = [&](const HaloItem& it) {
    if (p.in_amax) amax_raw = p.in_amax[it.n];      // turned into (in_s, in_inv) at the top of the item: the load has an epilogue to land
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      h_voff[k] = -16;
      if (hp < HALO_PIX) {
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        int iy = it.oy0 - 1 + hy, ix = it.ox0 - 1 + hx;
        KEEP_REFLECT(iy, ix, Hv, Wv)
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
          const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          h_voff[k] = ((sy * p.W + sx) * p.in_ld + g * 4) * 4;
        }
      }
    }
    const unsigned long long base = (unsigned long long)(p.in + (long)it.n * p.H * p.W * p.in_ld);
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)base), bhi = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32));
    in_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)bhi << 32) | blo), 0, p.H * p.W * p.in_ld * 4, 0x00020000);
    sc_off = (long)it.n * p.Cin + g * 4;
    bias_nx = make_float4(0.f, 0.f, 0.f, 0.f);       // bias of the item's cout block for this lane's four epilogue channels: in flight
    const int n0r = UP2 ? ((it.n0 >> 6) % ncb_real) << 6 : it.n0;          // real cout of the block's first channel
    if (p.bias && p.split_k == 1 && n0r + (lane & 15) * 4 < p.Cout)      // over a whole item (loaded in the epilogue it was waited for at once)
      bias_nx = *reinterpret_cast<const float4*>(p.bias + n0r + (lane & 15) * 4);
    w_voff = (it.n0 + (tid >> 2)) < cout_rows ? ((it.n0 + (tid >> 2)) * 9 * p.Cin * 2 + g * 8) * 2 : -16;
    if (WDMA) {
      // DMA instruction q of this chunk (36 per chunk, 9 per wave: q = wave * 9 + t) fills LDS rows q*16 .. q*16+15 = tap q/4, couts
      // (q%4)*16 + lane/4; lane's physical 16-byte slot lane&3 holds logical piece (lane&3) ^ ((lane>>4)&3) of its row
      // (stored rotated by the wave index: instruction t of wave w is q = 9 w + t, its row group q & 3 = (w + t) & 3 -- slot t & 3)
      const int lp = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = it.n0 + ((wave + j) & 3) * 16 + (lane >> 2);
        dma_voff[j] = co < cout_rows ? co * 9 * p.Cin * 4 + lp * 16 : -16;
      }
    }
  };

  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define KEEP_T(IDX)                                                   \
  if (EXP == 9) {                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();       \
    tacc[IDX] += t1 - t0;                                             \
    t0 = t1;                                                          \
  }
  float4 hreg[HALO_IT];
  uint4 wr0, wr1, wr2, wr3, wr4, wr5, wr6, wr7, wr8;
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int ch) {
    const int c0 = ch << 4;
    if (EXP == 5 && ch > 0) return;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, h_voff[k], c0 * 4, 0);
      hreg[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
#define KEEP_WLOADX(TAP, R)                                                                                  \
  {                                                                                                          \
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_voff, ((TAP) * p.Cin + c0) * 4, 0);      \
    R = make_uint4(v.x, v.y, v.z, v.w);                                                                      \
  }
    if (!WDMA) {
      KEEP_TAPS(KEEP_WLOADX)
    }
#undef KEEP_WLOADX
    if (p.pro_scale) {
      sc4 = *reinterpret_cast<const float4*>(p.pro_scale + sc_off + c0);
      sh4 = *reinterpret_cast<const float4*>(p.pro_shift + sc_off + c0);
    }
    fetched_ch = ch;
  };
  int up_par = 0, up_mask = 0x1ff;       // UP2: phase (py * 2 + px) and tap set of the CURRENT item
  auto stage = [&]() {
    constexpr bool FAST = FASTACT && EXP != 7;
    if (EXP == 3) return;
    if (WDMA && !(EXP == 5 && fetched_ch > 0) && EXP != 13 && EXP != 16) {   // weights of the chunk being staged: L2 -> LDS, in flight under the halo's VALU work below
      const int c0 = fetched_ch << 4;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;            // wave-uniform (M0 / soffset operands)
        if (UP2 && !((up_mask >> (q >> 2)) & 1)) continue;                      // a tap this phase does not use: never fetched
        if (EXP == 15)      // every piece from the same 1 KB of the weight tensor: L1 hits, no L2 traffic
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(wdma_base + q * 1024), 16,
                                                   lane * 16, 0, 0, 0);
        else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(wdma_base + q * 1024), 16,
                                                 dma_voff[t & 3], ((q >> 2) * p.Cin + c0) * 4, 0, 0);
      }
    }
    KEEP_T(8)
    if (EXP == 9) {
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      KEEP_T(9)
    }
    // GroupNorm affine + activation + split of this thread's 16-byte pieces, two values per instruction where the ISA has a packed
    // form (v_pk_fma / v_pk_add / v_pk_mul / v_cvt_pk_f16_f32): VALU instructions do not overlap the matrix pipe of their SIMD
    // (tools/dev/coissue_probe.hip), so every one of them is paid in full.  Fast swish: x * rcp(1 + exp2(x * s' + t')) with the
    // -log2(e) folded into a second affine (s', t') once per chunk.
    const f32x2 sc01 = {sc4.x, sc4.y}, sc23 = {sc4.z, sc4.w}, sh01 = {sh4.x, sh4.y}, sh23 = {sh4.z, sh4.w};
    constexpr bool FOLD = PRO == KEEP_PRO_SWISH && FAST && EXP != 10;
    const f32x2 nsc01 = sc01 * -1.4426950408889634f, nsc23 = sc23 * -1.4426950408889634f;
    const f32x2 nsh01 = sh01 * -1.4426950408889634f, nsh23 = sh23 * -1.4426950408889634f;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      if (hp < HALO_PIX) {
        _Float16* dst = &Hs[hp * XPITCH + g * 4];
        if (!has_pro || h_voff[k] >= 0) {
          f32x2 v01 = {hreg[k].x, hreg[k].y}, v23 = {hreg[k].z, hreg[k].w};
          if (has_pro) {
            const f32x2 y01 = v01 * sc01 + sh01, y23 = v23 * sc23 + sh23;
            if (FOLD) {
              const f32x2 z01 = v01 * nsc01 + nsh01, z23 = v23 * nsc23 + nsh23;
              f32x2 d01 = {__builtin_amdgcn_exp2f(z01.x), __builtin_amdgcn_exp2f(z01.y)};
              f32x2 d23 = {__builtin_amdgcn_exp2f(z23.x), __builtin_amdgcn_exp2f(z23.y)};
              d01 += 1.0f;
              d23 += 1.0f;
              const f32x2 r01 = {__builtin_amdgcn_rcpf(d01.x), __builtin_amdgcn_rcpf(d01.y)};
              const f32x2 r23 = {__builtin_amdgcn_rcpf(d23.x), __builtin_amdgcn_rcpf(d23.y)};
              v01 = y01 * r01;
              v23 = y23 * r23;
            } else if (EXP == 10) {
              v01 = y01;
              v23 = y23;
            } else {
              v01 = f32x2{pro_x3<PRO, FAST>(y01.x), pro_x3<PRO, FAST>(y01.y)};
              v23 = f32x2{pro_x3<PRO, FAST>(y23.x), pro_x3<PRO, FAST>(y23.y)};
            }
          }
          if (PRO == KEEP_PRO_NONE && p.in_amax) {     // activated inputs are bounded: the host never probes them
            v01 *= in_s;
            v23 *= in_s;
          }
          const f16x2 h01 = __builtin_convertvector(v01, f16x2), h23 = __builtin_convertvector(v23, f16x2);
          const f16x2 l01 = __builtin_convertvector(v01 - __builtin_convertvector(h01, f32x2), f16x2);
          const f16x2 l23 = __builtin_convertvector(v23 - __builtin_convertvector(h23, f32x2), f16x2);
          const f16x4 hi = {h01.x, h01.y, h23.x, h23.y}, lo = {l01.x, l01.y, l23.x, l23.y};
          if (EXP != 14 || (float)hi[0] + (float)lo[1] + (float)hi[2] + (float)lo[3] == 1.2345e-30f) {
            *reinterpret_cast<f16x4*>(dst) = hi;
            *reinterpret_cast<f16x4*>(dst + 16) = lo;
          }
        } else {      // zero padding applies to the normalised + activated tensor
          const f16x4 zero = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          *reinterpret_cast<f16x4*>(dst) = zero;
          *reinterpret_cast<f16x4*>(dst + 16) = zero;
        }
      }
    }
#define KEEP_WSTOREX(TAP, R) *reinterpret_cast<uint4*>(&Ws[((TAP) * 64 + (tid >> 2)) * XPITCH + g * 8]) = R;
    if (!WDMA) {
      KEEP_TAPS(KEEP_WSTOREX)
    } else if (EXP != 12 && EXP != 13 && EXP != 16) {
      KEEP_T(0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed (the barrier publishes them)
      KEEP_T(10)
    }
#undef KEEP_WSTOREX
  };

  f32x16 acc[2][2];
  const int a_base = (((2 * wave) * RPT + l31 / TW) * HALO_W + (l31 % TW)) * XPITCH + lhi * 8;
  // WDMA: 64-byte rows, physical slot = logical piece ^ ((row >> 2) & 3); tap / cout-block offsets are multiples of 16 rows
  const int b_base = WDMA ? l31 * 32 + ((lhi ^ ((l31 >> 2) & 3)) * 8) : l31 * XPITCH + lhi * 8;
  auto mma_m = [&](auto mask_c) {
    constexpr int MASK = decltype(mask_c)::value;      // taps of this instantiation (bit kh * 3 + kw)
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
    if (EXP == 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        if (!((MASK >> (kh * 3 + kw)) & 1)) continue;
        if (EXP != 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const _Float16* src = &Hs[a_base + ((i * RPT + kh) * HALO_W + kw) * XPITCH];
            ah[i] = *reinterpret_cast<const f16x8*>(src);
            al[i] = *reinterpret_cast<const f16x8*>(src + 16);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (WDMA) {
              const int o = b_base + ((kh * 3 + kw) * 64 + j * 32) * 32;
              if (EXP == 16) {      // no weight DMA; the B fragments come from the (changing) halo rows
                bh[j] = *reinterpret_cast<const f16x8*>(&Hs[o & 8191]);
                bl[j] = *reinterpret_cast<const f16x8*>(&Hs[(o ^ 16) & 8191]);
                continue;
              }
              bh[j] = *reinterpret_cast<const f16x8*>(&Ws[o]);
              bl[j] = *reinterpret_cast<const f16x8*>(&Ws[o ^ 16]);                 // lo piece: logical + 2 -> physical slot ^ 2
            } else {
              const _Float16* src = &Ws[b_base + ((kh * 3 + kw) * 64 + j * 32) * XPITCH];
              bh[j] = *reinterpret_cast<const f16x8*>(src);
              bl[j] = *reinterpret_cast<const f16x8*>(src + 16);
            }
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
    if (EXP == 8) __builtin_amdgcn_s_setprio(0);
  };
  auto mma = [&]() {
    if (!UP2) {
      mma_m(std::integral_constant<int, 0x1ff>{});
      return;
    }
    switch (up_par) {      // (py, px): taps kh in {py, py + 1}, kw in {px, px + 1}
      case 0: mma_m(std::integral_constant<int, 0x01b>{}); break;
      case 1: mma_m(std::integral_constant<int, 0x036>{}); break;
      case 2: mma_m(std::integral_constant<int, 0x0d8>{}); break;
      default: mma_m(std::integral_constant<int, 0x1b0>{}); break;
    }
  };
  // Epilogue: the wave parks its 64 x 64 tile in LDS and reads it back channel-contiguous (16 B per lane, 4 pixels x 256 B per
  // store instruction).  Addressing is image-relative and 32-bit: per-lane byte offset once per item, the (row, column)
  // displacement of each of the 16 pixel groups in the scalar offset of a buffer store -- no 64-bit multiplies per row.
  // Channels beyond Cout carry the out-of-range offset (stores dropped, residual reads zero).  The per-channel statistics are
  // reduced across the four pixel groups of the wave with v_permlane16/32_swap (VALU) instead of ds_bpermute.
  auto make_rsrc = [&](const void* ptr, int bytes) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  auto epilogue_t = [&](const HaloItem& it, float item_inv, const float4 bias4, auto res_c) {
    constexpr bool HAS_RES = decltype(res_c)::value;
    constexpr int EP = 68;
    float* et = reinterpret_cast<float*>(lds_raw) + wave * 64 * EP;
    if (EXP == 11 && acc[0][0][0] + acc[1][1][3] + acc[0][1][7] + acc[1][0][9] != 1.2345e-30f) return 0.f;
    const float asc = p.acc_scale * item_inv;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r];      // raw: asc (a power of two) rides in the bias FMA below
    __builtin_amdgcn_s_waitcnt(0xc07f);
    KEEP_T(11)
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int e_par = UP2 ? (it.n0 >> 6) / ncb_real : 0;
    const int e_n0 = UP2 ? it.n0 - e_par * p.Cout : it.n0;                    // real first cout of the block
    const int co = e_n0 + c4;
    const bool cok = co < p.Cout;
    const int hw_o = p.Ho * p.Wo;
    // pixel of group 0 inside the image (UP2: source pixel (y, x) of phase (py, px) -> output pixel (2y + py, 2x + px))
    const int pix_b = UP2 ? (2 * (it.oy0 + 2 * wave) + (e_par >> 1)) * p.Wo + 2 * (it.ox0 + prow) + (e_par & 1)
                          : (it.oy0 + 2 * wave * RPT) * p.Wo + it.ox0 + prow;
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(p.out + (long)it.n * hw_o * p.out_ld, hw_o * p.out_ld * 4);
    const int v_out = cok ? (pix_b * p.out_ld + co) * 4 : -16;
    __amdgpu_buffer_rsrc_t res_rsrc = out_rsrc, aux_rsrc = out_rsrc;
    int v_res = -16, v_aux = -16;
    if (HAS_RES) {
      res_rsrc = make_rsrc(p.res + (long)it.n * hw_o * p.res_ld, hw_o * p.res_ld * 4);
      v_res = cok ? (pix_b * p.res_ld + co) * 4 : -16;
      if (!SIMPLE_EPI && p.aux) {
        aux_rsrc = make_rsrc(p.aux + (long)it.n * hw_o * p.Cout, hw_o * p.Cout * 4);
        v_aux = cok ? (pix_b * p.Cout + co) * 4 : -16;
      }
    }
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float amx = 0.f;
    // Residual rows of the simple form: all 16 loads are issued before the first store.  Inside the loop every load sat behind
    // the previous iteration's store (the compiler must assume `res` and `out` alias -- in place they do) and its
    // `s_waitcnt vmcnt(0)` -- loads and stores share the counter on gfx9 -- exposed a full memory round trip 16 times per item.
    // (In-place use stays correct: a thread reads exactly the 16 addresses it later writes.)
    auto dpix_of = [&](int q16) {
      const int drow = (q16 >> 3) * RPT + (TW == 32 ? 0 : ((q16 & 7) >> 2));
      const int dcol = TW == 32 ? (q16 & 7) * 4 : (q16 & 3) * 4;
      return (UP2 ? 2 : 1) * (drow * p.Wo + dcol);                            // wave-uniform
    };
    u32x4 rpre[SIMPLE_EPI && HAS_RES ? 16 : 1];
    if (SIMPLE_EPI && HAS_RES) {
#pragma unroll
      for (int q16 = 0; q16 < 16; ++q16) rpre[q16] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, v_res, dpix_of(q16) * p.res_ld * 4, 0);
    }
    // the general epilogue (activation switch, aux tensor, split-K) stays a loop: fully unrolled it is 30k instructions; its residual
    // / aux rows are loaded per group of four iterations, before the group's first store
    constexpr int UNR = SIMPLE_EPI ? 16 : 4;
    u32x4 rgrp[4], agrp[4];
#pragma unroll UNR
    for (int q16 = 0; q16 < 16; ++q16) {
      if (!SIMPLE_EPI && HAS_RES && (q16 & 3) == 0 && p.split_k == 1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          rgrp[u] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, v_res, dpix_of(q16 + u) * p.res_ld * 4, 0);
          if (p.aux) agrp[u] = __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, v_aux, dpix_of(q16 + u) * p.Cout * 4, 0);
        }
      }
      const int px = q16 * 4 + prow;
      const int dpix = dpix_of(q16);
      const float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
      if (!SIMPLE_EPI && p.split_k > 1) {
        if (cok) {
          const long m = (long)it.n * hw_o + pix_b + dpix;
          *reinterpret_cast<float4*>(p.ws + ((long)it.z * p.M + m) * p.Cout + co) = make_float4(v.x * asc, v.y * asc, v.z * asc, v.w * asc);
        }
        continue;
      }
      float e[4] = {__builtin_fmaf(v.x, asc, bias4.x), __builtin_fmaf(v.y, asc, bias4.y), __builtin_fmaf(v.z, asc, bias4.z),
                    __builtin_fmaf(v.w, asc, bias4.w)};
      if (!SIMPLE_EPI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
      }
      if (HAS_RES) {
        const u32x4 r4 = SIMPLE_EPI ? rpre[SIMPLE_EPI ? q16 : 0] : rgrp[q16 & 3];
        const float rr[4] = {__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z), __uint_as_float(r4.w)};
        if (!SIMPLE_EPI && p.aux) {
          const u32x4 a4 = agrp[q16 & 3];
          const float aa[4] = {__uint_as_float(a4.x), __uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w)};
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = rr[q] + p.aux_w * (rr[q] * aa[q] + e[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] += rr[q];
        }
      }
      if (EXP != 4 || e[0] == 1.2345e-30f) {
        u32x4 o;
        o.x = __float_as_uint(e[0]); o.y = __float_as_uint(e[1]); o.z = __float_as_uint(e[2]); o.w = __float_as_uint(e[3]);
        __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, v_out, dpix * p.out_ld * 4, KEEP_ST_AUX_HALO);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += e[q];
        ss4[q] = __builtin_fmaf(e[q], e[q], ss4[q]);      // (explicit: the general epilogue's loop form left SOME of these uncontracted -- the replica kernel of keep_conv_x3s.hip reproduces one fused form)
        amx = fmaxf(amx, fabsf(e[q]));
      }
    }
    if (p.stats) {          // one partial per 256-pixel tile (stats_P = tiles): the four waves' sums meet in LDS, added in wave order
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] = xor32_sum(xor16_sum(s4[q]));
        ss4[q] = xor32_sum(xor16_sum(ss4[q]));
      }
      if (lane < 16) {      // the head of this wave's own staging tile (its rows are consumed: the values live in registers)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          et[(c4 + q) * 2 + 0] = s4[q];
          et[(c4 + q) * 2 + 1] = ss4[q];
        }
      }
      __syncthreads();
      if (wave == 0) {
        const float* e0 = reinterpret_cast<const float*>(lds_raw);
        float a = 0.f, b2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += e0[w * 64 * EP + lane * 2 + 0];
          b2 += e0[w * 64 * EP + lane * 2 + 1];
        }
        if (e_n0 + lane < p.Cout) {      // (UP2: four partials per source tile, one per phase -- each covers 256 output pixels)
          const int part = UP2 ? (it.ty * tiles_x + it.tx) * 4 + e_par : it.ty * tiles_x + it.tx;
          float* dst = p.stats + (((long)it.n * p.stats_P + part) * p.Cout + e_n0 + lane) * 2;
          dst[0] = a;
          dst[1] = b2;
        }
      }
    }
    return amx;
  };

  int amax_n = -1;                       // image of the running max|out| of this thread
  float amax_run = 0.f;
  int item = blockIdx.x;
  if (item >= n_items) return;
  if (EXP == 6 && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(54);     // start stagger of the second block per CU
  // EXP == 9: phase timeline (s_memtime) of wave 0, summed over blocks into p.ws as u64[8]:
  // 0 stage, 1 wait at the barrier after staging, 2 fetch issue, 3 mma, 4 wait at the barrier after mma, 5 item set-up, 6 epilogue
  const unsigned long long cyc0 = EXP != 0 ? __builtin_amdgcn_s_memtime() : 0ull;           // dev builds: shader cycles and
  const unsigned long long rtc0 = EXP != 0 ? __builtin_amdgcn_s_memrealtime() : 0ull;       // 100 MHz ticks of the whole block
  HaloItem cur = halo_decode<TW, 4>(p, item, items_per_z, tiles_x, tiles_y, ncb);
  setup(cur);
  if (cur.ch_begin < cur.ch_end) fetch(cur.ch_begin);
  if (EXP == 9) t0 = __builtin_amdgcn_s_memtime();
  while (true) {
    const bool valid = cur.ch_begin < cur.ch_end;
    if (p.in_amax) x3_range_scale(amax_raw, in_s, in_inv);
    if (UP2) {
      up_par = __builtin_amdgcn_readfirstlane((cur.n0 >> 6) / ncb_real);
      up_mask = up_par == 0 ? 0x01b : up_par == 1 ? 0x036 : up_par == 2 ? 0x0d8 : 0x1b0;
    }
    if (EXP == 9) {
      __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0): operand loads landed
      KEEP_T(7)
    }
    if (valid) stage();
    KEEP_T(0)
    __syncthreads();
    KEEP_T(1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ch = cur.ch_begin; ch < cur.ch_end; ++ch) {
      const bool more = ch + 1 < cur.ch_end;
      if (more) fetch(ch + 1);
      KEEP_T(2)
      mma();
      KEEP_T(3)
      __syncthreads();
      KEEP_T(4)
      if (more) {
        if (EXP == 9) {
          __builtin_amdgcn_s_waitcnt(0x0f70);
          KEEP_T(7)
        }
        stage();
        KEEP_T(0)
        __syncthreads();
        KEEP_T(1)
      }
    }
    const int next_item = item + gridDim.x;
    const bool has_next = next_item < n_items;
    const float cur_inv = in_inv;                               // setup(nxt) below moves in_s / in_inv / the bias on to the next item
    const float4 cur_bias = bias_nx;
    HaloItem nxt = cur;
    if (has_next) {
      nxt = halo_decode<TW, 4>(p, next_item, items_per_z, tiles_x, tiles_y, ncb);
      setup(nxt);
      if (nxt.ch_begin < nxt.ch_end) fetch(nxt.ch_begin);     // in flight during the epilogue below
    }
    KEEP_T(5)
    const float amx = p.res ? epilogue_t(cur, cur_inv, cur_bias, std::true_type{}) : epilogue_t(cur, cur_inv, cur_bias, std::false_type{});
    if (p.out_amax) {       // max|out| of the image: a wave only goes to memory when it holds a value above everything it has committed
      if (cur.n != amax_n) { // (or seen) for this image -- the per-item "read the running maximum, skip if not larger" test was a dependent
        amax_n = cur.n;      // global read in every epilogue
        amax_run = 0.f;
      }
      if (__builtin_amdgcn_ballot_w64(amx > amax_run) != 0ull) {
        unsigned b = __float_as_uint(amx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
        unsigned* dst = p.out_amax + cur.n;
        unsigned seen = b;
        if (lane == 0) {
          seen = *reinterpret_cast<volatile unsigned*>(dst);
          if (b > seen) atomicMax(dst, b);
        }
        seen = max(b, (unsigned)__builtin_amdgcn_readfirstlane((int)seen));
        amax_run = fmaxf(amax_run, __uint_as_float(seen));
      }
    }
    KEEP_T(6)
    if (!has_next) break;
    __syncthreads();
    KEEP_T(4)
    item = next_item;
    cur = nxt;
  }
  if (EXP != 0 && tid == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.ws);
    atomicAdd(dst + 13, __builtin_amdgcn_s_memtime() - cyc0);
    atomicAdd(dst + 14, __builtin_amdgcn_s_memrealtime() - rtc0);
    atomicAdd(dst + 15, 1ull);
    if (blockIdx.x < 1024) {          // per-block start / end (100 MHz ticks) behind the 16 sums
      dst[16 + blockIdx.x * 2] = rtc0;
      dst[17 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
    }
  }
  if (EXP == 9 && tid == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.ws);
#pragma unroll
    for (int q = 0; q < 12; ++q) atomicAdd(dst + q, tacc[q]);
    atomicAdd(dst + 12, 1ull);
  }
#undef KEEP_T
}

// ------------------------------------------------------------------------------------------------ gather GEMM, split fp16
// Everything that is not a 3x3 stride-1 convolution on a tileable map: token GEMMs, 1x1 convs, stride-2 convs.  Implicit
// GEMM like conv_bf16_kernel: K step = 32 channels of one tap; LDS rows [hi x32 | lo x32 | pad x8] at a 144-byte pitch (9 slots:
// ds_read_b128 conflict-free); two LDS buffers, the next step's operands in registers while the current one multiplies.
// A is split while staging (fp32 activations, optional GroupNorm affine + activation first); B comes pre-split.
// Epilogue of the x3 gather kernel for FULL row tiles (m0 + BM <= M): the wave parks its tile in LDS and reads it back
// channel-contiguous like staged_epilogue_impl, with block-relative 32-bit addressing through buffer descriptors (base =
// tensor + m0 * ld; row displacement in the vector offset) -- no 64-bit multiply per row, no EXEC branch per store.
template <int WGM, int WGN, int TM, int TN, bool SIMPLE>
__device__ __forceinline__ void x3_gather_epilogue(const ConvP& p, f32x16 (&acc)[TM][TN], float* lds, long m0, int n0, int wm,
                                                   int wn, int lane, int wave, int z) {
  constexpr int WR = TM * 32, WC = TN * 32, EP = WC + 4, BM = WGM * WR;
  constexpr int LPR = WC / 4, RPI = 64 / LPR;
  const int l31 = lane & 31, lhi = lane >> 5;
  float* et = lds + wave * WR * EP;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  auto make_rsrc = [&](const void* ptr, int bytes) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const int c4 = (lane % LPR) * 4, prow = lane / LPR;
  const int co = n0 + wn * WC + c4;
  const bool cok = co < p.Cout;
  const int row0 = wm * WR + prow;                       // row of iteration 0 inside the block tile
  const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(p.out + m0 * p.out_ld, BM * p.out_ld * 4);
  const int v_out = cok ? (row0 * p.out_ld + co) * 4 : (int)0x80000000;
  __amdgpu_buffer_rsrc_t res_rsrc = out_rsrc, aux_rsrc = out_rsrc, ws_rsrc = out_rsrc;
  int v_res = (int)0x80000000, v_aux = (int)0x80000000, v_ws = (int)0x80000000;
  if (p.res) {
    res_rsrc = make_rsrc(p.res + m0 * p.res_ld, BM * p.res_ld * 4);
    v_res = cok ? (row0 * p.res_ld + co) * 4 : (int)0x80000000;
    if (!SIMPLE && p.aux) {
      aux_rsrc = make_rsrc(p.aux + m0 * p.Cout, BM * p.Cout * 4);
      v_aux = cok ? (row0 * p.Cout + co) * 4 : (int)0x80000000;
    }
  }
  if (!SIMPLE && p.split_k > 1) {
    ws_rsrc = make_rsrc(p.ws + ((long)z * p.M + m0) * p.Cout, BM * p.Cout * 4);
    v_ws = cok ? (row0 * p.Cout + co) * 4 : (int)0x80000000;
  }
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias && p.split_k == 1 && cok) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
  // LayerNorm over the row's 128 channels (tile <4,1,1,4>: the 32 lanes of a half-wave hold one whole row, 4 channels each)
  constexpr bool LN_OK = SIMPLE && WGN == 1 && TN == 4;
  const bool ln = LN_OK && p.ln_gamma != nullptr;
  float4 lng = make_float4(1.f, 1.f, 1.f, 1.f), lnb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ln) {
    lng = *reinterpret_cast<const float4*>(p.ln_gamma + c4);
    lnb = *reinterpret_cast<const float4*>(p.ln_beta + c4);
  }
  // SIMPLE form: groups of up to 8 rows, the group's residual rows loaded before its first store (inside the loop each load sat behind
  // the previous store and its `s_waitcnt vmcnt(0)` exposed a memory round trip per row: see the halo kernel's epilogue)
  constexpr int NIT = WR / RPI, GRP = NIT < 8 ? NIT : 8;
  constexpr int UNR = SIMPLE ? GRP : 4;
  u32x4 rpre[SIMPLE ? GRP : 1];
#pragma unroll UNR
  for (int it = 0; it < NIT; ++it) {
    if (SIMPLE && p.res && (it % GRP) == 0) {
#pragma unroll
      for (int u = 0; u < GRP; ++u) rpre[u] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, v_res + (it + u) * RPI * p.res_ld * 4, 0, 0);
    }
    const int px = it * RPI + prow;
    const float4 v = *reinterpret_cast<const float4*>(et + px * EP + c4);
    if (!SIMPLE && p.split_k > 1) {
      u32x4 o;
      o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
      __builtin_amdgcn_raw_buffer_store_b128(o, ws_rsrc, v_ws + it * RPI * p.Cout * 4, 0, 0);
      continue;
    }
    float e[4] = {v.x + bias4.x, v.y + bias4.y, v.z + bias4.z, v.w + bias4.w};
    if (!SIMPLE) {
#pragma unroll
      for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
    }
    if (LN_OK && ln) {      // two passes like layernorm128_kernel: mean, then the centred squares (biased variance)
      float sm = (e[0] + e[1]) + (e[2] + e[3]);
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) sm += __shfl_xor(sm, o);
      const float mean = sm * (1.0f / 128.0f);
      const float d[4] = {e[0] - mean, e[1] - mean, e[2] - mean, e[3] - mean};
      float qq = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) qq += __shfl_xor(qq, o);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / 128.0f) + p.ln_eps);
      e[0] = d[0] * rstd * lng.x + lnb.x;
      e[1] = d[1] * rstd * lng.y + lnb.y;
      e[2] = d[2] * rstd * lng.z + lnb.z;
      e[3] = d[3] * rstd * lng.w + lnb.w;
    }
    if (p.res) {
      const u32x4 r4 = SIMPLE ? rpre[SIMPLE ? it % GRP : 0] : __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, v_res + it * RPI * p.res_ld * 4, 0, 0);
      const float rr[4] = {__uint_as_float(r4.x), __uint_as_float(r4.y), __uint_as_float(r4.z), __uint_as_float(r4.w)};
      if (!SIMPLE && p.aux) {
        const u32x4 a4 = __builtin_amdgcn_raw_buffer_load_b128(aux_rsrc, v_aux + it * RPI * p.Cout * 4, 0, 0);
        const float aa[4] = {__uint_as_float(a4.x), __uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w)};
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = rr[q] + p.aux_w * (rr[q] * aa[q] + e[q]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] += rr[q];
      }
    }
    {
      u32x4 o;
      o.x = __float_as_uint(e[0]); o.y = __float_as_uint(e[1]); o.z = __float_as_uint(e[2]); o.w = __float_as_uint(e[3]);
      __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, v_out + it * RPI * p.out_ld * 4, 0, KEEP_ST_AUX_GEMM);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s4[q] += e[q];
      ss4[q] += e[q] * e[q];
      amx = fmaxf(amx, fabsf(e[q]));
    }
  }
  if (p.out_amax) wave_amax_commit(p.out_amax + (int)((m0 + wm * WR) / ((long)p.Ho * p.Wo)), amx);
  if (p.stats) {   // host guarantees split_k == 1 and H*W % BM == 0 (a tile never straddles two images)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (LPR == 8) {
        s4[q] += __shfl_xor(s4[q], 8);
        ss4[q] += __shfl_xor(ss4[q], 8);
      }
      s4[q] = xor32_sum(xor16_sum(s4[q]));
      ss4[q] = xor32_sum(xor16_sum(ss4[q]));
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
    const int n_img = (int)(m0 / hw_o), p_idx = (int)((m0 % hw_o) / BM);
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

#define XBK 32
#define XP (2 * XBK + 8)
#ifndef XG_ABL
#define XG_ABL 0      // dev (tools/dev/README.md): 1 no operand split, 2 no output stores, 3 no MFMAs, 4 no A loads, 5 no B loads
#endif

// ONE: 1x1 stride-1 unpadded convolution == a row-major GEMM: the A rows are fetched with block-relative buffer loads (rows
// beyond M read zeros through the descriptor's range check), no im2col index arithmetic.
// KSL: canonical K slices (p.kslice_steps > 0, keep_gemm_x3l.hip) -- its own instantiations: the slice totals cost TM * TN * 16 registers.
// DEEP: the 64 x 64 tile's register prefetch ring (below); the im2col form at large row counts runs without it (bit-neutral).
// KAL: KSL with slices of exactly PD = 4 K steps (K = 512, 1024: every token GEMM of the code transformer) -- slice boundaries are compile-time
// positions of the unrolled ring: the first MFMA of a slice takes the inline constant 0 as C, the fold is 16 adds, nothing is zeroed.
template <int WGM, int WGN, int TM, int TN, bool PLAIN, bool ONE = false, bool KSL = false, bool DEEP = true, bool KAL = false>
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
  // tile_cols > 0 (1-D grid): the column blocks of one row block are consecutive logical ids of ONE XCD (xcd_remap), so the A rows
  // a row block shares between its column blocks are re-read from that XCD's L2 instead of once per column block from HBM
  int bx = blockIdx.x, by = blockIdx.y;
  if (p.tile_cols > 0) {
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    bx = lid / p.tile_cols;
    by = lid - bx * p.tile_cols;
  }
  if (p.reverse) bx = (int)((p.M + BM - 1) / BM) - 1 - bx;
  const long m0 = (long)bx * BM;
  const int n0 = by * BN;
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
    if (a_mv[it] && (!ONE || p.in_amax || !PLAIN)) {      // the GEMM form needs the image index only for per-image scales
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

  // the in-flight operands of one K step (registers)
  struct StepRegs {
    float a_raw[A_IT][8];
    bool a_ok[A_IT];
    uint4 b_raw[B_IT];
    int a_c;
    float u_sc[8], u_sh[8];
  };
  const long m_last = (m0 + BM - 1 < p.M) ? (m0 + BM - 1) : (long)p.M - 1;
  const bool uni_n = !PLAIN && p.pro_scale && ((m0 / hw) == (m_last / hw));
  const long uni_off = (m0 / hw) * (long)p.Cin;

  const long rows_here = (p.M - m0) < BM ? (p.M - m0) : BM;
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
  int a_voff[A_IT];
  if (ONE) {
    const unsigned long long b = (unsigned long long)(p.in + m0 * p.in_ld);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)rows_here * p.in_ld * 4, 0x00020000);
#pragma unroll
    for (int it = 0; it < A_IT; ++it) a_voff[it] = ((a_row0 + it * 64) * p.in_ld + a_grp * 8) * 4;
  }
  // K-concatenated second input (channels >= cin1, dense rows): its own descriptor and row offsets; a K step never straddles
  // the seam (cin1 % 32 == 0)
  __amdgpu_buffer_rsrc_t a2_rsrc = a_rsrc;
  int a2_voff[A_IT];
  const int ld2 = p.Cin - p.cin1;
  if (ONE && p.in2) {
    const unsigned long long b = (unsigned long long)(p.in2 + m0 * ld2);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    a2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)rows_here * ld2 * 4, 0x00020000);
#pragma unroll
    for (int it = 0; it < A_IT; ++it) a2_voff[it] = ((a_row0 + it * 64) * ld2 + a_grp * 8) * 4;
  }
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.wx3, 0, p.Cout * p.KH * p.KW * p.Cin * 4, 0x00020000);
  int b_voff[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int co = n0 + b_row0 + it * 32;
    b_voff[it] = co < p.Cout ? (int)((long)co * wrow_stride * 2) + b_pc * 16 : (int)0x80000000;
  }

  auto fetch = [&](int s, StepRegs& R) {
    const int tap = s / cchunks;
    const int c0 = (s - tap * cchunks) * XBK;
    const int kh = tap / p.KW;
    const int kw = tap - kh * p.KW;
    const int ca = c0 + a_grp * 8;
    R.a_c = ca;
    if (uni_n && ca < p.Cin) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        R.u_sc[j] = p.pro_scale[uni_off + ca + j];
        R.u_sh[j] = p.pro_shift[uni_off + ca + j];
      }
    }
    if (ONE) {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        R.a_ok[it] = ca < p.Cin;             // rows beyond M: zeros from the range check
        if (XG_ABL == 4) {
#pragma unroll
          for (int j = 0; j < 8; ++j) R.a_raw[it][j] = (float)(j + s);
          continue;
        }
        const bool second = p.in2 && c0 >= p.cin1;                      // wave-uniform
        u32x4 v0, v1;
        if (KEEP_LD_AUX_GEMM_A1 != 0 && BN >= 128 && p.Cout <= BN && !p.in2) {      // one column block: every A row is read exactly once
          v0 = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[it], c0 * 4, KEEP_LD_AUX_GEMM_A1);
          v1 = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[it] + 16, c0 * 4, KEEP_LD_AUX_GEMM_A1);
        } else {
          v0 = second ? __builtin_amdgcn_raw_buffer_load_b128(a2_rsrc, a2_voff[it], (c0 - p.cin1) * 4, 0)
                      : __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[it], c0 * 4, 0);
          v1 = second ? __builtin_amdgcn_raw_buffer_load_b128(a2_rsrc, a2_voff[it] + 16, (c0 - p.cin1) * 4, 0)
                      : __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[it] + 16, c0 * 4, 0);
        }
        R.a_raw[it][0] = __uint_as_float(v0.x); R.a_raw[it][1] = __uint_as_float(v0.y);
        R.a_raw[it][2] = __uint_as_float(v0.z); R.a_raw[it][3] = __uint_as_float(v0.w);
        R.a_raw[it][4] = __uint_as_float(v1.x); R.a_raw[it][5] = __uint_as_float(v1.y);
        R.a_raw[it][6] = __uint_as_float(v1.z); R.a_raw[it][7] = __uint_as_float(v1.w);
      }
    } else {
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        int iy = a_oy[it] * p.stride - p.pad_t + kh;
        int ix = a_ox[it] * p.stride - p.pad_l + kw;
        KEEP_REFLECT(iy, ix, p.H, p.W)
        R.a_ok[it] = a_mv[it] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ca < p.Cin;
        if (R.a_ok[it]) {
          const float* src = p.in + (((long)a_n[it] * p.H + iy) * p.W + ix) * p.in_ld + ca;
          const float4 v0 = *reinterpret_cast<const float4*>(src);
          const float4 v1 = *reinterpret_cast<const float4*>(src + 4);
          R.a_raw[it][0] = v0.x; R.a_raw[it][1] = v0.y; R.a_raw[it][2] = v0.z; R.a_raw[it][3] = v0.w;
          R.a_raw[it][4] = v1.x; R.a_raw[it][5] = v1.y; R.a_raw[it][6] = v1.z; R.a_raw[it][7] = v1.w;
        }
      }
    }
    const bool cb_ok = c0 + (b_pc >> 2) * 16 < p.Cin;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      if (XG_ABL == 5) { R.b_raw[it] = make_uint4(s, it, s, it); continue; }
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, cb_ok ? b_voff[it] : (int)0x80000000, (tap * p.Cin + c0) * 4, 0);
      R.b_raw[it] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };

  auto stage = [&](int buf, StepRegs& R) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      f16x8 hi, lo;
      if (R.a_ok[it]) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = R.a_raw[it][j];
        if (!PLAIN) {
          if (uni_n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * R.u_sc[j] + R.u_sh[j];
          } else if (p.pro_scale) {
            const float* sc = p.pro_scale + (long)a_n[it] * p.Cin + R.a_c;
            const float* sh = p.pro_shift + (long)a_n[it] * p.Cin + R.a_c;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = v[j] * sc[j] + sh[j];
          }
          if (p.pro_act != KEEP_PRO_NONE) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p.fast ? pro_apply_x3(v[j], p.pro_act) : pro_apply(v[j], p.pro_act);
          }
        }
        if (XG_ABL == 1 && PLAIN && ONE) {
          *reinterpret_cast<float4*>(&hi) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(&lo) = make_float4(v[4], v[5], v[6], v[7]);
        } else
#pragma unroll
        for (int j = 0; j < 8; j += 2) {      // two values per instruction: v_pk_mul, v_cvt_pk_f16_f32, v_pk_add (VALU is paid in full: coissue_probe)
          const f32x2 vs = f32x2{v[j], v[j + 1]} * a_s[it];
          const f16x2 h = __builtin_convertvector(vs, f16x2);
          const f16x2 l = __builtin_convertvector(vs - __builtin_convertvector(h, f32x2), f16x2);
          hi[j] = h.x; hi[j + 1] = h.y;
          lo[j] = l.x; lo[j + 1] = l.y;
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
      *reinterpret_cast<uint4*>(&Bs[buf][(b_row0 + it * 32) * XP + b_col]) = R.b_raw[it];
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

  auto mma_step = [&](int buf, bool zc = false) {
    if (XG_ABL == 3) return;
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
          if (KAL && zc && ks == 0) {
            const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], z16, 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          } else {
            MMA_X3(acc[i][j], ah[i], al[i], bh[j], bl[j])
          }
        }
    }
  };

  if (s_begin < s_end) {
    // Register prefetch ring of PD K steps.  Large tiles (128 x 128: many blocks in flight, the K loop does not wait for its loads --
    // a two-deep ring measured 2119 vs 2045 us on 256 -> 1024 at 0.62 M rows) keep PD = 1.  The 64 x 64 tile serves the launches that
    // CANNOT fill the chip with blocks (token GEMMs of <= 4096 rows, stride-2 convolutions of one clip): their K loop is a chain of
    // dependent L2 / HBM round trips (20 us for 4096 x 512 x 512 whatever the FLOPs), and a step in flight costs only 18 VGPRs there.
    // The im2col form at LARGE row counts (16 x 512 x 512 stride-2: 656 -> 735 us with the ring: 4 -> 3 blocks per CU) keeps PD = 1 (DEEP = false).
    constexpr int PD = (TM * TN == 1 && DEEP) ? (PLAIN ? 4 : 2) : 1;      // (the prologue form carries 16 scale / shift registers per step in flight)
    StepRegs R[PD];
#pragma unroll
    for (int u = 0; u < PD; ++u)
      if (s_begin + u < s_end) fetch(s_begin + u, R[u]);
    stage(0, R[0]);
    __syncthreads();
    int buf = 0;
    // canonical K slices (p.kslice_steps > 0: the throughput form of keep_gemm_x3l.hip's sums): every slice of kslice_steps K steps is
    // accumulated from zero and the slice totals are added in order -- bit for bit what the latency form's waves + LDS reduction produce
    f32x16 tot[KSL ? TM : 1][KSL ? TN : 1];
    int ksl_left = p.kslice_steps;
    bool ksl_first = true;
    for (int s0 = s_begin; s0 < s_end; s0 += PD) {
#pragma unroll
      for (int u = 0; u < PD; ++u) {
        const int s = s0 + u;
        if (s < s_end) {      // (uniform)
          if (s + PD < s_end) fetch(s + PD, R[u]);            // R[u] held step s: staged one iteration ago
          if (KAL) {
            mma_step(buf, u == 0);
            if (u == PD - 1) {
              if (ksl_first) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                  for (int j = 0; j < TN; ++j) tot[KSL ? i : 0][KSL ? j : 0] = acc[i][j];
              } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                  for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[KSL ? i : 0][KSL ? j : 0][r] += acc[i][j][r];
              }
              ksl_first = false;
            }
          } else {
            mma_step(buf);
          }
          if (KSL && !KAL && --ksl_left == 0) {
            ksl_left = p.kslice_steps;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                  tot[KSL ? i : 0][KSL ? j : 0][r] = ksl_first ? acc[i][j][r] : tot[KSL ? i : 0][KSL ? j : 0][r] + acc[i][j][r];
                  acc[i][j][r] = 0.f;
                }
            ksl_first = false;
          }
          if (s + 1 < s_end) stage(buf ^ 1, R[(u + 1) % PD]);
          __syncthreads();
          buf ^= 1;
        }
      }
    }
    if (KSL) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = tot[KSL ? i : 0][KSL ? j : 0];
    }
  }
  if (XG_ABL == 2 && acc[0][0][0] != 1234.5f) return;
  if (XG_ABL == 3) acc[0][0][0] = (float)As[0][threadIdx.x];
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
  if (m0 + BM <= p.M && p.Cout * 4L * BM < (1L << 30) && p.out_ld * 4L * BM < (1L << 30) && p.res_ld * 4L * BM < (1L << 30)) {
    if (p.split_k == 1 && !p.aux && p.epi_act == KEEP_ACT_NONE)
      x3_gather_epilogue<WGM, WGN, TM, TN, true>(p, acc, reinterpret_cast<float*>(smem_b), m0, n0, wm, wn, lane, wave, z);
    else
      x3_gather_epilogue<WGM, WGN, TM, TN, false>(p, acc, reinterpret_cast<float*>(smem_b), m0, n0, wm, wn, lane, wave, z);
    return;
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

// ------------------------------------------------------------------------------------------------ 3x3, Cin <= 3, split fp16
// The RGB first convolutions (3 -> 64 at 512x512: VQ conv_in of the LQ encoder over all B*T frames, of the HQ encoder every
// frame) under KEEP_MMA_X3 -- the x3 form of conv3x3_c3_kernel (keep_conv.hip): persistent blocks (2 per CU), per 8x32-pixel
// item the 10x34xCin fp32 halo is loaded once, every thread expands ITS pixel into one K = 32 im2col row [hi x32 | lo x32]
// (K = 9*Cin <= 27 real taps, zero padded), the 64 x K weight rows are split on the fly from the fp32 weights (27 values per
// cout: no pre-split twin needed) and stay in LDS across items, and the wave runs 2 x 4 x 3 MFMAs before the LDS-staged float4
// epilogue with GroupNorm partials and the fused max|out|.  HBM-write bound (64 couts x 4 B per pixel); on the exact-f32
// gather kernel the same launches ran at 26-28 TFLOP/s = 1.6 TB/s.
#define C3X_K 32
#define C3X_P 72                             // fp16 per im2col / weight row: 32 hi + 32 lo + 8 pad (144 B)
__global__ __launch_bounds__(256, 2) void conv3x3_c3_x3_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  constexpr int HW_ = 34, HROWS = 10;
  constexpr int EPI_B = 4 * 64 * 68 * 4;                                    // staging tile; im2col rows and the halo alias it
  constexpr int A_B = 256 * C3X_P * 2;
  static_assert(A_B + HROWS * HW_ * 4 * 4 <= EPI_B, "im2col rows + halo fit the staging area");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[EPI_B + 64 * C3X_P * 2];
  float* et_base = reinterpret_cast<float*>(lds_raw);
  _Float16* As = reinterpret_cast<_Float16*>(lds_raw);                      // [256 px][C3X_P]
  float* Hs = reinterpret_cast<float*>(lds_raw + A_B);                      // [10][34 * Cin]
  _Float16* Ws = reinterpret_cast<_Float16*>(lds_raw + EPI_B);              // [64][C3X_P], resident across items

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int Cin = p.Cin, K = 9 * Cin;
  const int rowf = HW_ * Cin;

  int cur_cb = -1;
  auto load_weights = [&](int cb) {                                          // [64 couts][K] -> hi | lo, zero padded to 32
    for (int i = tid; i < 64 * C3X_K; i += 256) {
      const int co = i >> 5, k = i & 31;
      float v = 0.f;
      if (k < K && cb * 64 + co < p.Cout) v = p.w[(long)(cb * 64 + co) * K + k];
      const _Float16 h = (_Float16)v;
      Ws[co * C3X_P + k] = h;
      Ws[co * C3X_P + 32 + k] = (_Float16)(v - (float)h);
    }
  };

  f32x16 acc[2][2];
  const bool has_act = p.epi_act != KEEP_ACT_NONE;
  const int py = tid >> 5, px = tid & 31;
  // The halo values (<= 4 per thread) and the input range of the NEXT item are loaded while the current one multiplies and stores:
  // loaded at the top of their own item they were waited for at once (two resident blocks per CU hide little of a memory round trip).
  struct Item { int cb, tx, ty, n; };
  auto decode = [&](int item) {
    Item it;
    const int lid = xcd_remap(item, n_items);
    it.cb = lid % ncb;
    int t = lid / ncb;
    it.tx = t % tiles_x; t /= tiles_x;
    it.ty = t % tiles_y;
    it.n = t / tiles_y;
    return it;
  };
  constexpr int HV = (HROWS * HW_ * 3 + 255) / 256;                          // halo floats per thread (Cin <= 3)
  float hv[HV], amax_raw = 0.f;
  auto load_halo = [&](const Item& it) {
    const float* img = p.in + (long)it.n * p.H * p.W * p.in_ld;
    const int oy0 = it.ty * 8, ox0 = it.tx * 32;
#pragma unroll
    for (int u = 0; u < HV; ++u) {
      const int i = tid + u * 256;
      hv[u] = 0.f;
      if (i < HROWS * rowf) {
        const int hy = i / rowf, r = i - hy * rowf;
        const int hx = r / Cin, c = r - hx * Cin;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) hv[u] = img[((long)iy * p.W + ix) * p.in_ld + c];
      }
    }
    if (p.in_amax) amax_raw = p.in_amax[it.n];
  };
  int amax_n = -1;                                                           // image of the max|out| this wave has committed or seen
  float amax_seen = 0.f;
  int item = blockIdx.x;
  if (item >= n_items) return;
  Item nxt = decode(item);
  load_halo(nxt);
  for (; item < n_items; item += gridDim.x) {
    const Item cur = nxt;
    const int cb = cur.cb, tx = cur.tx, ty = cur.ty, n = cur.n;
    const int oy0 = ty * 8, ox0 = tx * 32, n0 = cb * 64;
    float in_s = 1.f, in_inv = 1.f;
    if (p.in_amax) x3_range_scale(amax_raw, in_s, in_inv);
    __syncthreads();                                                         // previous item's staging tile fully stored
    if (cb != cur_cb) {
      load_weights(cb);
      cur_cb = cb;
    }
#pragma unroll
    for (int u = 0; u < HV; ++u)
      if (tid + u * 256 < HROWS * rowf) Hs[tid + u * 256] = hv[u] * in_s;
    if (item + (int)gridDim.x < n_items) {
      nxt = decode(item + gridDim.x);
      load_halo(nxt);
    }
    __syncthreads();
    {
      float vals[C3X_K];
#pragma unroll
      for (int k = 0; k < C3X_K; ++k) vals[k] = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kwc = 0; kwc < 9; ++kwc)                                    // (kw, c) run is contiguous in the halo row for Cin = 3
          if (kwc < 3 * Cin && kh * 3 * Cin + kwc < C3X_K) vals[kh * 3 * Cin + kwc] = Hs[(py + kh) * rowf + px * Cin + kwc];
      f16x8 hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < C3X_K; ++k) {
        const _Float16 h = (_Float16)vals[k];
        hi[k >> 3][k & 7] = h;
        lo[k >> 3][k & 7] = (_Float16)(vals[k] - (float)h);
      }
      __syncthreads();                                                       // every thread has read its halo values (As aliases nothing of Hs, but keep the order simple)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<f16x8*>(&As[tid * C3X_P + q * 8]) = hi[q];
        *reinterpret_cast<f16x8*>(&As[tid * C3X_P + 32 + q * 8]) = lo[q];
      }
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
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const _Float16* src = &As[(wave * 64 + i * 32 + l31) * C3X_P + ks * 16 + lhi * 8];
        ah[i] = *reinterpret_cast<const f16x8*>(src);
        al[i] = *reinterpret_cast<const f16x8*>(src + 32);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const _Float16* src = &Ws[(j * 32 + l31) * C3X_P + ks * 16 + lhi * 8];
        bh[j] = *reinterpret_cast<const f16x8*>(src);
        bl[j] = *reinterpret_cast<const f16x8*>(src + 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          MMA_X3(acc[i][j], ah[i], al[i], bh[j], bl[j])
        }
    }
    __syncthreads();                                                         // A reads done: the area becomes the staging tile
    constexpr int EP = 68;
    float* et = et_base + wave * 64 * EP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) et[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * EP + j * 32 + l31] = acc[i][j][r] * in_inv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int co = n0 + c4;
    const bool cok = co < p.Cout;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float amx = 0.f;
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
        for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
      }
      if (KEEP_NT_C3)
        __builtin_nontemporal_store((f32x4){e[0], e[1], e[2], e[3]}, reinterpret_cast<f32x4*>(p.out + m * p.out_ld + co));
      else
        *reinterpret_cast<float4*>(p.out + m * p.out_ld + co) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += e[q];
        ss4[q] += e[q] * e[q];
        amx = fmaxf(amx, fabsf(e[q]));
      }
    }
    if (p.out_amax) {       // to memory only above what this wave has committed or seen for the image (see the halo kernel)
      if (n != amax_n) {
        amax_n = n;
        amax_seen = 0.f;
      }
      if (__builtin_amdgcn_ballot_w64(amx > amax_seen) != 0ull) {
        unsigned b = __float_as_uint(amx);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
        unsigned* dst = p.out_amax + n;
        unsigned seen = b;
        if (lane == 0) {
          seen = *reinterpret_cast<volatile unsigned*>(dst);
          if (b > seen) atomicMax(dst, b);
        }
        seen = max(b, (unsigned)__builtin_amdgcn_readfirstlane((int)seen));
        amax_seen = fmaxf(amax_seen, __uint_as_float(seen));
      }
    }
    if (p.stats) {          // per wave: stats_P = Ho*Wo/64, partial index = tile*4 + wave
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] = xor32_sum(xor16_sum(s4[q]));
        ss4[q] = xor32_sum(xor16_sum(ss4[q]));
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

int keep_conv2d_x3_c3(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  const int tx = a->Wo / 32, ty = a->Ho / 8, ncb = (a->Cout + 63) / 64;
  const int n_items = a->N * tx * ty * ncb;
  const int n_cu = x3_num_cu();
  hipLaunchKernelGGL(conv3x3_c3_x3_kernel, dim3(n_items < 2 * n_cu ? n_items : 2 * n_cu), dim3(256), 0, st, p, tx, ty, ncb, n_items);
  KEEP_LAUNCH_CHECK("keep_conv2d(Cin<=3, x3)");
  return KEEP_OK;
}

// Geometry the x3 kernels accept (everything else of a KEEP_MMA_X3 call runs on the exact-f32 kernels: same parity grade).
bool keep_conv_x3_halo_ok(const keep_conv2d_args* a) {
  return a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad_t == 1 &&
         a->pad_l == 1 && (a->Cin % 16 == 0) && (a->Cout % 32 == 0) &&
         ((a->Ho % 8 == 0 && a->Wo % 32 == 0) || (a->Ho % 16 == 0 && a->Wo % 16 == 0)) &&
         a->Ho == (a->upsample ? 2 * a->H : a->H) && a->Wo == (a->upsample ? 2 * a->W : a->W) &&
         (long)a->H * a->W * a->in_ld * 4 < (1L << 31) && (long)a->Cout * 9 * a->Cin * 4 < (1L << 31) &&   // buffer offsets
         (long)a->Ho * a->Wo * a->out_ld * 4 < (1L << 31) && (long)a->Ho * a->Wo * (a->residual ? a->res_ld : 1) * 4 < (1L << 31) &&
         (!a->pro_scale || ((uintptr_t)a->pro_scale % 16 == 0 && (uintptr_t)a->pro_shift % 16 == 0)) &&
         (a->in_ld % 4 == 0) && ((uintptr_t)a->in % 16 == 0) && (a->out_ld % 4 == 0) && ((uintptr_t)a->out % 16 == 0) &&
         (!a->residual || (a->res_ld % 4 == 0 && (uintptr_t)a->residual % 16 == 0)) &&
         (!a->aux || (uintptr_t)a->aux % 16 == 0) && (!a->bias || (uintptr_t)a->bias % 16 == 0) &&
         (!a->workspace || (uintptr_t)a->workspace % 16 == 0);
}

// KEEP_UPSAMPLE_X2_PHASES: the halo geometry on the SOURCE map (8 x 32 tiles), whole 64-cout blocks, no prologue / activation / aux / split-K
bool keep_conv_x3_up2_ok(const keep_conv2d_args* a) {
  return keep_conv_x3_halo_ok(a) && a->H % 8 == 0 && a->W % 32 == 0 && a->Cout % 64 == 0 && !a->pro_scale && a->pro_act == KEEP_PRO_NONE &&
         a->epi_act == KEEP_ACT_NONE && !a->aux && a->split_k <= 1 && a->pad_mode == KEEP_PAD_ZERO &&
         (long)4 * a->Cout * 9 * a->Cin * 4 < (1L << 31);
}

bool keep_conv_x3_gather_ok(const keep_conv2d_args* a, const ConvP& p) {
  return a->dtype == KEEP_F32 && a->out_dtype != KEEP_BF16 && !a->upsample && (a->Cin % 16 == 0) && (a->in_ld % 4 == 0) &&
         ((uintptr_t)a->in % 16 == 0) && p.vec_epi && (long)a->Cout * a->KH * a->KW * a->Cin * 4 < (1L << 31);   // weight buffer offsets
}

// 1x1 stride-1 unpadded convolution with rows short enough for block-relative 32-bit offsets: the GEMM variant of the kernel
bool keep_conv_x3_gather_is_gemm(const keep_conv2d_args* a) {
  return a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad_t == 0 && a->pad_l == 0 && a->Ho == a->H && a->Wo == a->W &&
         (long)a->in_ld * 4 * 128 < (1L << 30);
}

bool keep_conv_x3p_ok(const keep_conv2d_args* a, const ConvP& p, int split_k);
bool keep_conv_x3q_ok(const keep_conv2d_args* a, const ConvP& p, int split_k);
bool keep_conv_x3p_full_ok(const keep_conv2d_args* a, const ConvP& p, int split_k);
int keep_conv2d_x3_small_full(const keep_conv2d_args* a, ConvP& p, hipStream_t st);
int keep_conv2d_x3_partials(const keep_conv2d_args* a, ConvP& p, hipStream_t st);
bool keep_conv_x3_stream_ok(const keep_conv2d_args* a, const ConvP& p, int split_k);
int keep_conv2d_x3_stream(const keep_conv2d_args* a, ConvP& p, int n_cu, hipStream_t st);

int keep_conv2d_x3_halo(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  if (a->upsample == KEEP_UPSAMPLE_X2_PHASES) {      // four 2x2-tap phase convolutions on the source grid (kernel comment: UP2)
    const int tx = a->W / 32, ty = a->H / 8, ncbv = 4 * (a->Cout / 64);
    const int n_items = a->N * tx * ty * ncbv;
    const int n_cu = x3_num_cu();
    p.upsample = 0;                                    // the kernel addresses the source like a plain 3x3 convolution
    p.split_k = 1;
    dim3 grid(n_items < 2 * n_cu ? n_items : 2 * n_cu), block(256);
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_NONE, true, 0, true, true, true>), grid, block, 0, st, p, tx, ty, ncbv, n_items);
    KEEP_LAUNCH_CHECK("keep_conv2d(halo x3, x2 phases)");
    return KEEP_OK;
  }
  const int nchunks = a->Cin / 16;
  if (p.split_k > nchunks) p.split_k = nchunks;
  if (keep_conv_x3p_ok(a, p, p.split_k)) return keep_conv2d_x3_partials(a, p, st);                        // keep_conv_x3p.hip (few images: 64-pixel tiles, same partials)
  if (keep_conv_x3_stream_ok(a, p, p.split_k) && keep_conv_x3q_ok(a, p, p.split_k)) return keep_conv2d_x3_small_full(a, p, st);      // few items: 64-pixel blocks, the streaming kernel's values
  if (keep_conv_x3_stream_ok(a, p, p.split_k)) return keep_conv2d_x3_stream(a, p, x3_num_cu(), st);      // keep_conv_x3s.hip
  if (keep_conv_x3p_full_ok(a, p, p.split_k)) return keep_conv2d_x3_partials(a, p, st);                  // few items, this kernel's epilogue (aux tensor / 16-wide maps): 64-pixel blocks, same values
  const bool wide = (a->Ho % 8 == 0 && a->Wo % 32 == 0);
  const int tw = wide ? 32 : 16, th = 256 / tw;
  const int tiles_x = a->Wo / tw, tiles_y = a->Ho / th, ncb = (a->Cout + 63) / 64;
  const int n_items = a->N * tiles_x * tiles_y * ncb * p.split_k;
  const int n_cu = x3_num_cu();
  const int per_cu = KEEP_DEV_ENV("KEEP_X3_BLOCKS_PER_CU") ? atoi(KEEP_DEV_ENV("KEEP_X3_BLOCKS_PER_CU")) : 2;      // dev: occupancy scaling probe
  dim3 grid(n_items < per_cu * n_cu ? n_items : per_cu * n_cu), block(256);
  const bool simple = p.split_k == 1 && !a->aux && a->epi_act == KEEP_ACT_NONE;
  // pipelined single-block-per-CU kernel: wide tiles, no split-K, at least two work items per CU
#ifdef KEEP_X3_ABLATE
  if (KEEP_DEV_ENV("KEEP_X3_EXP") && wide && simple && (a->pro_act == KEEP_PRO_SWISH || a->pro_act == KEEP_PRO_NONE)) {
    const int ex = atoi(KEEP_DEV_ENV("KEEP_X3_EXP"));
static unsigned long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 128 + 1024 * 16);
    const bool cyc = KEEP_DEV_ENV("KEEP_X3_CYC") != nullptr;
    if (cyc) (void)hipMemsetAsync(dbg, 0, 128, st);
    ConvP q = p;
    q.ws = reinterpret_cast<float*>(dbg);
    auto report = [&](int e) {      // shader cycles and 100 MHz ticks per block -> the effective shader clock of this variant
      if (!cyc) return;
      unsigned long long h[16];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost);
      const double nb = (double)h[15];
      fprintf(stderr, "[x3 cycles] exp %d  blocks %.0f  cycles/block %.0f  us/block %.1f  clock %.0f MHz\n", e, nb, h[13] / nb,
              h[14] / nb / 100.0, (double)h[13] / ((double)h[14] / 100.0));
      static unsigned long long se[2048];
      const int nblk = (int)grid.x < 1024 ? (int)grid.x : 1024;
      (void)hipMemcpy(se, dbg + 16, nblk * 16, hipMemcpyDeviceToHost);
      unsigned long long t_min = ~0ull, t_max = 0;
      for (int b = 0; b < nblk; ++b) {
        if (se[2 * b] < t_min) t_min = se[2 * b];
        if (se[2 * b + 1] > t_max) t_max = se[2 * b + 1];
      }
      double dur_x[8] = {0}, st_x[8] = {0}, en_x[8] = {0}, dmin = 1e30, dmax = 0;
      int cnt_x[8] = {0};
      for (int b = 0; b < nblk; ++b) {
        const double d = (se[2 * b + 1] - se[2 * b]) / 100.0;
        dur_x[b & 7] += d; st_x[b & 7] += (se[2 * b] - t_min) / 100.0; en_x[b & 7] += (se[2 * b + 1] - t_min) / 100.0; cnt_x[b & 7]++;
        if (d < dmin) dmin = d;
        if (d > dmax) dmax = d;
      }
      fprintf(stderr, "[x3 blocks] span %.1f us  block life min %.1f max %.1f us | per XCD (start, life, end):", (t_max - t_min) / 100.0, dmin, dmax);
      for (int x = 0; x < 8; ++x) fprintf(stderr, "  %.0f/%.0f/%.0f", st_x[x] / cnt_x[x], dur_x[x] / cnt_x[x], en_x[x] / cnt_x[x]);
      fprintf(stderr, "\n");
    };
#define KEEP_LAUNCH_ABL(E)                                                                                                         \
  if (ex == E) {                                                                                                                   \
    if (a->pro_act == KEEP_PRO_SWISH)                                                                                              \
      hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_SWISH, true, E, true, true>), grid, block, 0, st, q, tiles_x, tiles_y, ncb, n_items); \
    else                                                                                                                           \
      hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_NONE, true, E, true, true>), grid, block, 0, st, q, tiles_x, tiles_y, ncb, n_items);  \
    KEEP_LAUNCH_CHECK("keep_conv2d(halo x3 ablation)");                                                                            \
    report(E);                                                                                                                     \
    return KEEP_OK;                                                                                                                \
  }
    KEEP_LAUNCH_ABL(1) KEEP_LAUNCH_ABL(2) KEEP_LAUNCH_ABL(3) KEEP_LAUNCH_ABL(4) KEEP_LAUNCH_ABL(5) KEEP_LAUNCH_ABL(6) KEEP_LAUNCH_ABL(7)
    KEEP_LAUNCH_ABL(8) KEEP_LAUNCH_ABL(10) KEEP_LAUNCH_ABL(11) KEEP_LAUNCH_ABL(12) KEEP_LAUNCH_ABL(13) KEEP_LAUNCH_ABL(14) KEEP_LAUNCH_ABL(15) KEEP_LAUNCH_ABL(16)
    KEEP_LAUNCH_ABL(17)
    if (ex == 9) {       // phase timeline: one instrumented launch, cycle sums printed to stderr
      (void)hipMemsetAsync(dbg, 0, 128, st);
      if (a->pro_act == KEEP_PRO_SWISH)
        hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_SWISH, true, 9, true, true>), grid, block, 0, st, q, tiles_x, tiles_y, ncb, n_items);
      else
        hipLaunchKernelGGL((conv3x3_halo_x3_kernel<32, KEEP_PRO_NONE, true, 9, true, true>), grid, block, 0, st, q, tiles_x, tiles_y, ncb, n_items);
      unsigned long long h[16];
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost);
      const double nb = (double)h[12], tot = (double)(h[0] + h[1] + h[2] + h[3] + h[4] + h[5] + h[6] + h[7] + h[8] + h[9] + h[10] + h[11]);
      fprintf(stderr, "[x3 timeline] stage split: DMA issue %.1f%%  wait halo regs %.1f%%  VALU+ds_write %.1f%%  wait DMA %.1f%%  | epilogue: park in LDS %.1f%%  rest %.1f%%\n",
              100.0 * h[8] / tot, 100.0 * h[9] / tot, 100.0 * h[0] / tot, 100.0 * h[10] / tot, 100.0 * h[11] / tot, 100.0 * h[6] / tot);
      fprintf(stderr, "[x3 timeline] blocks %.0f  cycles/block %.0f | stage %.1f%%  sync-after-stage %.1f%%  fetch-issue %.1f%%  mma %.1f%%  "
              "sync-after-mma %.1f%%  item-setup %.1f%%  epilogue %.1f%%  wait-loads %.1f%%\n", nb, tot / nb, 100.0 * h[0] / tot, 100.0 * h[1] / tot,
              100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot, 100.0 * h[5] / tot, 100.0 * h[6] / tot, 100.0 * h[7] / tot);
      return KEEP_OK;
    }
#undef KEEP_LAUNCH_ABL
  }
#endif
  static const bool wdma = !KEEP_DEV_ENV("KEEP_X3_NO_WDMA");      // weights by LDS-DMA (default); the VGPR-staged form stays for A/B runs
#define KEEP_LAUNCH_HX2(TWV, PROV)                                                                                          \
  if (wdma && simple)                                                                                                      \
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, PROV, true, 0, true, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);  \
  else if (wdma)                                                                                                           \
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, PROV, false, 0, true, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items); \
  else if (simple)                                                                                                         \
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, PROV, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);  \
  else                                                                                                                     \
    hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, PROV, false>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
#define KEEP_LAUNCH_HX(TWV)                                       \
  if (a->pro_act == KEEP_PRO_SWISH && !p.fast) {                  \
    if (simple)                                                   \
      hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, KEEP_PRO_SWISH, true, 0, false>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);  \
    else                                                          \
      hipLaunchKernelGGL((conv3x3_halo_x3_kernel<TWV, KEEP_PRO_SWISH, false, 0, false>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items); \
  } else if (a->pro_act == KEEP_PRO_SWISH) {                      \
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

// The GroupNorm partials of the 128 x 128 tile (x3_gather_epilogue<2, 2, 2, 2>: one partial per 128 output rows) from the WRITTEN output, in
// that epilogue's order: a wave's lane sums its rows it * 4 + prow (it = 0 .. 15) of 4 columns sequentially, the four prow lanes meet as
// (p0 + p1) + (p2 + p3), the two row halves of the block as (0 + upper) + lower.  Lets a launch with few rows run the 64 x 64 tile
// (4 x the blocks, same output bits) and still hand the plan's statistics partition to keep_norm_finalize.  grid (rows / 128, cols / 128).
__global__ __launch_bounds__(256) void conv_x3_gather_stats_replica_kernel(ConvP p) {
  __shared__ float red[4 * 64 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.x * 128;
  const int n0 = blockIdx.y * 128;
  const int c4 = (lane & 15) * 4, prow = lane >> 4;
  const int co = n0 + wn * 64 + c4;
  const bool cok = co < p.Cout;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const long row = m0 + wm * 64 + it * 4 + prow;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cok) v = *reinterpret_cast<const float4*>(p.out + row * p.out_ld + co);
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s4[q] += e[q];
      ss4[q] += e[q] * e[q];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    s4[q] = xor32_sum(xor16_sum(s4[q]));
    ss4[q] = xor32_sum(xor16_sum(ss4[q]));
  }
  if (lane < 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[(wave * 64 + c4 + q) * 2 + 0] = s4[q];
      red[(wave * 64 + c4 + q) * 2 + 1] = ss4[q];
    }
  }
  __syncthreads();
  const int hw_o = p.Ho * p.Wo;
  const int n_img = (int)(m0 / hw_o), p_idx = (int)((m0 % hw_o) / 128);
  for (int c = threadIdx.x; c < 128; c += 256) {
    const int wn_c = c >> 6, rem = c & 63;
    float a = 0.f, b2 = 0.f;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      a += red[((mm * 2 + wn_c) * 64 + rem) * 2 + 0];
      b2 += red[((mm * 2 + wn_c) * 64 + rem) * 2 + 1];
    }
    if (n0 + c < p.Cout) {
      float* dst = p.stats + (((long)n_img * p.stats_P + p_idx) * p.Cout + n0 + c) * 2;
      dst[0] = a;
      dst[1] = b2;
    }
  }
}

#ifndef KEEP_GATHER_SMALL_ROWS
#define KEEP_GATHER_SMALL_ROWS 4096      // rows in flight up to which a launch planned for the 128 x 128 tile runs the 64 x 64 tile (dev A/B: -D)
#endif
// tile: plan_conv's choice (1: 64x64 block tiles, 2: 128x128, 3: 128x128 as four 32-row waves with the LayerNorm epilogue) --
// the launch never re-derives it
int keep_conv2d_x3_gather(const keep_conv2d_args* a, ConvP& p, int tile, hipStream_t st) {
  const long M = p.M;
  // the plan's tile follows the reference batch because the statistics partition does; a launch WITHOUT statistics may take the small
  // tile when the real row count is small (one clip in flight: 4 x the blocks) -- the K order of a row's sum does not depend on the tile.
  // WITH statistics (the encoder's stride-2 convolutions): the small tile too, the 128-row partials then come from the replica kernel above.
  float* replica_stats = nullptr;
  if (tile == 2 && p.stats && M <= KEEP_GATHER_SMALL_ROWS && M % 128 == 0 && ((long)p.Ho * p.Wo) % 128 == 0 && p.split_k == 1 && p.vec_epi &&
      p.Cout % 4 == 0 && !p.out_bf16 && !(a->flags & KEEP_CONV_NO_SMALL_PARTIALS)) {
    replica_stats = p.stats;
    p.stats = nullptr;
    tile = 1;
  }
  if (tile == 2 && !p.stats && M <= KEEP_GATHER_SMALL_ROWS) tile = 1;
  const int big_tile = tile >= 2;
  const int steps = a->KH * a->KW * ((a->Cin + XBK - 1) / XBK);
  if (p.split_k > steps) p.split_k = steps;
  const bool plain = !a->pro_scale && a->pro_act == KEEP_PRO_NONE;
  dim3 block(256);
  // 1x1 stride-1 unpadded convolutions (token GEMMs): block-relative buffer-load fetch, no im2col index arithmetic
  const bool one = keep_conv_x3_gather_is_gemm(a);
#define KEEP_LAUNCH_GX(A, B, C, D)                                                                 \
  if (p.kslice_steps == 4 && plain && C * D == 1 && steps % 4 == 0)                                \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, true, true, true, true, C * D == 1>), grid, block, 0, st, p); \
  else if (p.kslice_steps > 0 && plain)                                                            \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, true, true, true>), grid, block, 0, st, p);     \
  else if (p.kslice_steps > 0)                                                                     \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, false, true, true>), grid, block, 0, st, p);    \
  else if (plain && one)                                                                           \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, true, true>), grid, block, 0, st, p);           \
  else if (plain && C * D == 1 && M > 262144)                                                      \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, true, false, false, false>), grid, block, 0, st, p); \
  else if (plain)                                                                                  \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, true, false>), grid, block, 0, st, p);          \
  else if (one)                                                                                    \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, false, true>), grid, block, 0, st, p);          \
  else                                                                                             \
    hipLaunchKernelGGL((conv_x3_kernel<A, B, C, D, false, false>), grid, block, 0, st, p);
  // several column blocks and many row blocks: row-block-major order on a 1-D grid (KEEP_X3_GEMM_2D=1: the 2-D grid, for A/B runs)
  const int bt = big_tile ? 128 : 64;
  const long gx = cdiv(M, bt), gy = cdiv(a->Cout, bt);
  const bool rowmajor = gy > 1 && gx >= 1024 && gx * gy < (1L << 30) && !KEEP_DEV_ENV("KEEP_X3_GEMM_2D");
  p.tile_cols = rowmajor ? (int)gy : 0;
  {
    static const int rev = KEEP_DEV_ENV("KEEP_X3_GEMM_REVERSE") ? atoi(KEEP_DEV_ENV("KEEP_X3_GEMM_REVERSE")) : 0;      // dev A/B
    p.reverse = rev;
  }
  dim3 grid(rowmajor ? (unsigned)(gx * gy) : (unsigned)gx, rowmajor ? 1u : (unsigned)gy, p.split_k);
  if (tile == 3) {
    hipLaunchKernelGGL((conv_x3_kernel<4, 1, 1, 4, true, true>), grid, block, 0, st, p);
  } else if (!big_tile) {
    KEEP_LAUNCH_GX(2, 2, 1, 1)
  } else {
    KEEP_LAUNCH_GX(2, 2, 2, 2)
  }
#undef KEEP_LAUNCH_GX
  KEEP_LAUNCH_CHECK("keep_conv2d(gather x3)");
  if (replica_stats) {
    p.stats = replica_stats;
    hipLaunchKernelGGL(conv_x3_gather_stats_replica_kernel, dim3((unsigned)(M / 128), (unsigned)cdiv(a->Cout, 128)), dim3(256), 0, st, p);
    KEEP_LAUNCH_CHECK("keep_conv2d(gather x3, statistics replica)");
  }
  return KEEP_OK;
}
