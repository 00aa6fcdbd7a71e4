// keep_conv2d, KEEP_MMA_X3, 3x3 stride 1 on SMALL maps with a split-K plan, FEW images per launch: the partial producer in small tiles.
//
// The 16 x 16 stages of the frame recurrence (VQ ResBlocks of 512 channels, KA:1062-1127) run split 4 .. 8 ways over the input channels
// at every batch size (plan_conv: a per-image rule), conv3x3_halo_x3_kernel<16> writing one partial per (256-pixel tile, 64 couts, z)
// block and conv_splitk_reduce_kernel adding them in z order.  With ONE clip in flight that is 32 blocks of four waves walking 8
// chunks x 108 MFMAs each: 40 us for 1.2 GFLOP on a chip with 1024 SIMDs.  This kernel produces THE SAME partials -- the same channel
// chunks per z (halo_decode's rule), per chunk the same prologue arithmetic (GroupNorm affine, x3-grade swish, range scale, split) and
// the nine taps in the same order with the same three-MFMA product into a zero-initialised accumulator: bit-identical values in the
// same workspace layout, the reducer is unchanged -- from blocks of 64 pixels (4 x 16) x 64 couts: four times the blocks, each wave
// ONE 32 x 32 tile (27 MFMAs per chunk).  Operand staging is the halo kernel's: the fp32 halo (6 x 18 pixels x 16 channels) goes
// register -> VALU -> LDS rows [hi16 | lo16 | pad], the 9 x 64 pre-split weight rows go L2 -> register -> LDS in the DMA kernel's image (64-byte rows, XOR-swizzled
// 16-byte slots); three operand stages in LDS: a chunk costs one barrier and its loads are issued two chunks ahead (a wave's 27
// MFMAs per chunk are shorter than an HBM round trip of the weights: with two stages every chunk waited ~1.5 us for its DMA).
// keep_conv2d_x3_halo picks it from the REAL item count (bit-neutral); large batches keep the 256-pixel kernel.
#include "keep_conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define XPP 40                         // fp16 per halo row: 16 hi + 16 lo + 8 pad (80 B), as in keep_conv_x3.hip
#define XP_HW 18                       // halo width of a 16-column tile
#define XP_HPIX (6 * XP_HW)            // 4 + 2 rows
#define XP_HBYTES (128 * XPP * 2)      // 10240: 128 rows per stage, rows 108 .. 127 parked but never read (every thread converts two pieces: one basic block)
#define XP_WBYTES (9 * 64 * 64)        // 36864: 9 taps x 64 couts x 64 B

#ifndef XQ_SCHED
#define XQ_SCHED 1      // round 6: conversion of chunk c + 1 scheduled into the MFMA chain of chunk c (0: the compiler's own order)
#endif
#define XP_ST 3                        // operand stages in LDS: the loads of chunk c + 2 are issued before the MFMAs of chunk c
// NCH > 0: the block's chunk count, known at compile time -- the ring is straight-line code and the compiler's wait counts are exact (inside
// a loop it orders every LDS read of a stage behind vmcnt(0): the DMA that filled the stage came through the back edge); 0: any count.
template <int PRO, bool HAS_SC, int NCH>
__global__ __launch_bounds__(256) void conv3x3_x3p_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  // one LDS object per weight stage: the compiler's wait-count pass orders a ds_read behind every LDS-DMA write that MAY alias it -- with
  // one array the fragment reads of stage u waited (vmcnt(1)) for the DMA of stage u + 2, i.e. for the whole prefetch distance
  __shared__ __attribute__((aligned(1024))) unsigned char ws0[XP_WBYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char ws1[XP_WBYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char ws2[XP_WBYTES];
  __shared__ __attribute__((aligned(16))) unsigned char hs_raw[XP_ST * XP_HBYTES];
  static_assert(XP_ST == 3, "three weight stages");
#define XP_WS(i) ((i) == 0 ? ws0 : ((i) == 1 ? ws1 : ws2))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ph = wave >> 1, chf = wave & 1;      // pixel rows {2 ph, 2 ph + 1} of the tile, cout half
  const int g = tid & 3;
  // item -> (z, image, tile, cout block): the pixel tiles of one (cout block, z) -- they share the weight rows -- are neighbours in one XCD
  const int lid = xcd_remap(blockIdx.x, n_items);
  const int npt = p.N * tiles_y * tiles_x;
  const int pt = lid % npt, cbz = lid / npt;
  const int cb = cbz % ncb, z = cbz / ncb;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int oy0 = ty * 4, ox0 = tx * 16, n0 = cb * 64;
  const int nchunks = p.Cin >> 4;
  const int per = (nchunks + p.split_k - 1) / p.split_k;      // halo_decode<TW, 4>
  const int ch_begin = z * per, ch_end = min(nchunks, ch_begin + per);
  constexpr bool has_pro = HAS_SC || PRO != KEEP_PRO_NONE;

  float in_s = 1.f, in_inv = 1.f;
  if (p.in_amax) x3_range_scale(p.in_amax[n], in_s, in_inv);
  int h_voff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int hp = (tid >> 2) + k * 64;
    h_voff[k] = -16;
    if (hp < XP_HPIX) {
      const int hy = hp / XP_HW, hx = hp - hy * XP_HW;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) h_voff[k] = ((iy * p.W + ix) * p.in_ld + g * 4) * 4;
    }
  }
  auto make_rsrc = [&](const void* ptr, int bytes) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(p.in + (long)n * p.H * p.W * p.in_ld, p.H * p.W * p.in_ld * 4);
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wx3, p.Cout * 9 * p.Cin * 4);
  const long sc_off = (long)n * p.Cin + g * 4;
  // weight DMA (conv3x3_halo_x3_kernel, WDMA): instruction q = 9 wave + t fills LDS rows q * 16 .. + 15 = tap q / 4, couts (q % 4) * 16 + lane / 4;
  // the lane's physical slot lane & 3 holds logical piece (lane & 3) ^ ((lane >> 4) & 3) of its 64-byte row
  int dma_voff[4];
  {
    const int lp = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = n0 + ((wave + j) & 3) * 16 + (lane >> 2);
      dma_voff[j] = co < p.Cout ? co * 9 * p.Cin * 4 + lp * 16 : -16;
    }
  }
  float4 hregs[XP_ST][2];
  u32x4 wregs[XP_ST][9];      // weight pieces in flight (register -> ds_write_b128 in stage(): plain loads, so the compiler's wait counts are exact --
                              // behind LDS-DMA writes it ordered every LDS access with vmcnt(0), i.e. behind the whole prefetch distance)
  float4 sc4s[XP_ST], sh4s[XP_ST];
#pragma unroll
  for (int i = 0; i < XP_ST; ++i) {
    sc4s[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    sh4s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto fetch = [&](int ch, int buf) {
    float4 (&hreg)[2] = hregs[buf];
    float4 &sc4 = sc4s[buf], &sh4 = sh4s[buf];
    const int c0 = ch << 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, h_voff[k], c0 * 4, 0);
      hreg[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    if (HAS_SC) {
      sc4 = *reinterpret_cast<const float4*>(p.pro_scale + sc_off + c0);
      sh4 = *reinterpret_cast<const float4*>(p.pro_shift + sc_off + c0);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {      // the piece LDS-DMA instruction q = 9 wave + t of conv3x3_halo_x3_kernel would move for this lane
      const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;
      wregs[buf][t] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dma_voff[t & 3], ((q >> 2) * p.Cin + c0) * 4, 0);
    }
  };
  // GroupNorm affine + activation + range scale + split of this thread's pieces: the expressions of conv3x3_halo_x3_kernel::stage
  auto stage = [&](int buf) {
    const float4 (&hreg)[2] = hregs[buf];
    const float4 sc4 = sc4s[buf], sh4 = sh4s[buf];
    {
      unsigned char* wbase = XP_WS(buf);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;
        *reinterpret_cast<u32x4*>(wbase + q * 1024 + lane * 16) = wregs[buf][t];
      }
    }
    _Float16* Hs = reinterpret_cast<_Float16*>(hs_raw + buf * XP_HBYTES);
    const f32x2 sc01 = {sc4.x, sc4.y}, sc23 = {sc4.z, sc4.w}, sh01 = {sh4.x, sh4.y}, sh23 = {sh4.z, sh4.w};
    const f32x2 nsc01 = sc01 * -1.4426950408889634f, nsc23 = sc23 * -1.4426950408889634f;
    const f32x2 nsh01 = sh01 * -1.4426950408889634f, nsh23 = sh23 * -1.4426950408889634f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hp = (tid >> 2) + k * 64;
      if (hp < XP_HPIX) {
        _Float16* dst = &Hs[hp * XPP + g * 4];
        if (!has_pro || h_voff[k] >= 0) {
          f32x2 v01 = {hreg[k].x, hreg[k].y}, v23 = {hreg[k].z, hreg[k].w};
          if (has_pro) {
            const f32x2 y01 = v01 * sc01 + sh01, y23 = v23 * sc23 + sh23;
            if (PRO == KEEP_PRO_SWISH) {
              const f32x2 z01 = v01 * nsc01 + nsh01, z23 = v23 * nsc23 + nsh23;
              f32x2 d01 = {__builtin_amdgcn_exp2f(z01.x), __builtin_amdgcn_exp2f(z01.y)};
              f32x2 d23 = {__builtin_amdgcn_exp2f(z23.x), __builtin_amdgcn_exp2f(z23.y)};
              d01 += 1.0f;
              d23 += 1.0f;
              const f32x2 r01 = {__builtin_amdgcn_rcpf(d01.x), __builtin_amdgcn_rcpf(d01.y)};
              const f32x2 r23 = {__builtin_amdgcn_rcpf(d23.x), __builtin_amdgcn_rcpf(d23.y)};
              v01 = y01 * r01;
              v23 = y23 * r23;
            } else {
              v01 = y01;
              v23 = y23;
            }
          }
          if (PRO == KEEP_PRO_NONE && p.in_amax) {
            v01 *= in_s;
            v23 *= in_s;
          }
          const f16x2 h01 = __builtin_convertvector(v01, f16x2), h23 = __builtin_convertvector(v23, f16x2);
          const f16x2 l01 = __builtin_convertvector(v01 - __builtin_convertvector(h01, f32x2), f16x2);
          const f16x2 l23 = __builtin_convertvector(v23 - __builtin_convertvector(h23, f32x2), f16x2);
          const f16x4 hi = {h01.x, h01.y, h23.x, h23.y}, lo = {l01.x, l01.y, l23.x, l23.y};
          *reinterpret_cast<f16x4*>(dst) = hi;
          *reinterpret_cast<f16x4*>(dst + 16) = lo;
        } else {      // zero padding applies to the normalised + activated tensor
          const f16x4 zero = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          *reinterpret_cast<f16x4*>(dst) = zero;
          *reinterpret_cast<f16x4*>(dst + 16) = zero;
        }
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int a_base = ((2 * ph + (l31 >> 4)) * XP_HW + (l31 & 15)) * XPP + lhi * 8;
  const int b_base = l31 * 32 + ((lhi ^ ((l31 >> 2) & 3)) * 8);
  auto mma = [&](int buf) {
    const _Float16* Ws = reinterpret_cast<const _Float16*>(XP_WS(buf));
    const _Float16* Hs = reinterpret_cast<const _Float16*>(hs_raw + buf * XP_HBYTES);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const _Float16* src = &Hs[a_base + (kh * XP_HW + kw) * XPP];
        const f16x8 ah = *reinterpret_cast<const f16x8*>(src), al = *reinterpret_cast<const f16x8*>(src + 16);
        const int o = b_base + ((kh * 3 + kw) * 64 + chf * 32) * 32;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(&Ws[o]), bl = *reinterpret_cast<const f16x8*>(&Ws[o ^ 16]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      }
  };

  // Round 6: a wave's 27 MFMAs per chunk are ONE dependent chain (~32 cycles each); `fused(bc, bn)` issues them with the conversion + LDS writes of
  // the chunk in stage bn cut into 25 steps, one per MFMA gap (the expressions of `stage`, a padded piece selected to zero instead of branched
  // around; every gap pinned by a sched_barrier), and the next tap's fragments requested under the current tap's MFMAs.
  auto fused = [&](int bc, int bn) __attribute__((always_inline)) {
    const _Float16* Ws = reinterpret_cast<const _Float16*>(XP_WS(bc));
    const _Float16* Hs = reinterpret_cast<const _Float16*>(hs_raw + bc * XP_HBYTES);
    _Float16* Hn = reinterpret_cast<_Float16*>(hs_raw + bn * XP_HBYTES);
    unsigned char* wn = XP_WS(bn);
    const float4 sc4 = sc4s[bn], sh4 = sh4s[bn];
    const f32x2 sc01 = {sc4.x, sc4.y}, sc23 = {sc4.z, sc4.w}, sh01 = {sh4.x, sh4.y}, sh23 = {sh4.z, sh4.w};
    const f32x2 nsc01 = sc01 * -1.4426950408889634f, nsc23 = sc23 * -1.4426950408889634f;
    const f32x2 nsh01 = sh01 * -1.4426950408889634f, nsh23 = sh23 * -1.4426950408889634f;
    f32x2 v01[2], v23[2], y01[2], y23[2], d01[2], d23[2];
    f16x2 h01[2], h23[2];
    auto step = [&](int i) __attribute__((always_inline)) {
      if (i < 16) {
        const int k = i >> 3, j = i & 7;
        if (j == 0) {
          v01[k] = f32x2{hregs[bn][k].x, hregs[bn][k].y};
          v23[k] = f32x2{hregs[bn][k].z, hregs[bn][k].w};
          if (has_pro) {
            y01[k] = v01[k] * sc01 + sh01;
            y23[k] = v23[k] * sc23 + sh23;
          }
        } else if (j == 1) {
          if (PRO == KEEP_PRO_SWISH) {
            d01[k] = v01[k] * nsc01 + nsh01;
            d23[k] = v23[k] * nsc23 + nsh23;
          }
        } else if (j == 2) {
          if (PRO == KEEP_PRO_SWISH) {
            d01[k] = f32x2{__builtin_amdgcn_exp2f(d01[k].x), __builtin_amdgcn_exp2f(d01[k].y)};
            d23[k] = f32x2{__builtin_amdgcn_exp2f(d23[k].x), __builtin_amdgcn_exp2f(d23[k].y)};
          }
        } else if (j == 3) {
          if (PRO == KEEP_PRO_SWISH) {
            d01[k] += 1.0f;
            d23[k] += 1.0f;
          }
        } else if (j == 4) {
          if (PRO == KEEP_PRO_SWISH) {
            d01[k] = f32x2{__builtin_amdgcn_rcpf(d01[k].x), __builtin_amdgcn_rcpf(d01[k].y)};
            d23[k] = f32x2{__builtin_amdgcn_rcpf(d23[k].x), __builtin_amdgcn_rcpf(d23[k].y)};
          }
        } else if (j == 5) {
          if (PRO == KEEP_PRO_SWISH) {
            v01[k] = y01[k] * d01[k];
            v23[k] = y23[k] * d23[k];
          } else if (has_pro) {
            v01[k] = y01[k];
            v23[k] = y23[k];
          }
          if (PRO == KEEP_PRO_NONE && p.in_amax) {
            v01[k] *= in_s;
            v23[k] *= in_s;
          }
        } else if (j == 6) {
          h01[k] = __builtin_convertvector(v01[k], f16x2);
          h23[k] = __builtin_convertvector(v23[k], f16x2);
        } else {
          const f16x2 l01 = __builtin_convertvector(v01[k] - __builtin_convertvector(h01[k], f32x2), f16x2);
          const f16x2 l23 = __builtin_convertvector(v23[k] - __builtin_convertvector(h23[k], f32x2), f16x2);
          const bool pad = has_pro && h_voff[k] < 0;      // zero padding applies to the normalised + activated tensor
          const f16x4 zero = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
          const f16x4 hi = {h01[k].x, h01[k].y, h23[k].x, h23[k].y}, lo = {l01.x, l01.y, l23.x, l23.y};
          _Float16* dst = &Hn[((tid >> 2) + k * 64) * XPP + g * 4];
          *reinterpret_cast<f16x4*>(dst) = pad ? zero : hi;
          *reinterpret_cast<f16x4*>(dst + 16) = pad ? zero : lo;
        }
      } else if (i < 25) {
        const int t = i - 16;
        const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;
        *reinterpret_cast<u32x4*>(wn + q * 1024 + lane * 16) = wregs[bn][t];
      }
    };
    const _Float16* src0 = &Hs[a_base];
    f16x8 ah = *reinterpret_cast<const f16x8*>(src0), al = *reinterpret_cast<const f16x8*>(src0 + 16);
    f16x8 bh = *reinterpret_cast<const f16x8*>(&Ws[b_base + (chf * 32) * 32]), bl = *reinterpret_cast<const f16x8*>(&Ws[(b_base + (chf * 32) * 32) ^ 16]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      f16x8 nah = ah, nal = al, nbh = bh, nbl = bl;
      if (t < 8) {
        const int kh = (t + 1) / 3, kw = (t + 1) % 3;
        const _Float16* src = &Hs[a_base + (kh * XP_HW + kw) * XPP];
        nah = *reinterpret_cast<const f16x8*>(src);
        nal = *reinterpret_cast<const f16x8*>(src + 16);
        const int o = b_base + ((t + 1) * 64 + chf * 32) * 32;
        nbh = *reinterpret_cast<const f16x8*>(&Ws[o]);
        nbl = *reinterpret_cast<const f16x8*>(&Ws[o ^ 16]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);      // conv3x3_halo_x3_kernel's term order
      step(3 * t);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      step(3 * t + 1);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      step(3 * t + 2);
      __builtin_amdgcn_sched_barrier(0);
      ah = nah; al = nal; bh = nbh; bl = nbl;
    }
  };

// (a __syncthreads() carries vmcnt(0): it would wait for the loads of chunk c + 2 as well; LDS writes are complete at lgkmcnt(0))
#define XP_BARRIER()                  \
  __builtin_amdgcn_s_waitcnt(0xc07f); \
  __builtin_amdgcn_s_barrier();
  if (NCH > 0 || ch_begin < ch_end) {
    const bool two = NCH > 0 ? NCH > 1 : ch_begin + 1 < ch_end;
    fetch(ch_begin, 0);
    if (two) fetch(ch_begin + 1, 1);
    stage(0);
    XP_BARRIER()
    if (NCH > 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {      // chunk ch_begin + c lives in stage c % XP_ST
        if (c + 2 < NCH) fetch(ch_begin + c + 2, (c + 2) % XP_ST);
        if (XQ_SCHED && c + 1 < NCH) {
          fused(c % XP_ST, (c + 1) % XP_ST);
          XP_BARRIER()
        } else {
          mma(c % XP_ST);
          if (c + 1 < NCH) {
            stage((c + 1) % XP_ST);
            XP_BARRIER()
          }
        }
      }
    } else {
      for (int c0 = ch_begin; c0 < ch_end; c0 += XP_ST) {
#pragma unroll
        for (int u = 0; u < XP_ST; ++u) {      // chunk c0 + u lives in stage u (the ring advances by XP_ST per outer iteration)
          const int ch = c0 + u;
          if (ch < ch_end) {
            if (ch + 2 < ch_end) fetch(ch + 2, (u + 2) % XP_ST);
            mma(u);
            if (ch + 1 < ch_end) {
              stage((u + 1) % XP_ST);
            }
            XP_BARRIER()
          }
        }
      }
    }
  }
  const float asc = p.acc_scale * in_inv;
  const int hw_o = p.Ho * p.Wo;
  if (p.split_k == 1) {
    // Round 6 -- un-split plans the streaming kernel does not take (an aux tensor: CFT's shift convolution, KA:470-472; 16-wide maps), few
    // items: the WHOLE convolution from these blocks, conv3x3_halo_x3_kernel's epilogue arithmetic (e = fma(acc, scale, bias); activation;
    // residual, or res + aux_w (res aux + e)) on rows of 32 channels, per-image max|out|; its GroupNorm partials come from
    // conv_stats_replica_kernel (8 x 32 tiles: the launcher admits statistics only there).  Same operand arithmetic and MFMA order as the
    // partials above: bit-equal to the 256-pixel kernel (tests/test_gpu_kernels.py).
    XP_BARRIER()                                                          // every wave is past its last fragment read: the halo stages become the row staging
    float* et = reinterpret_cast<float*>(hs_raw) + wave * 32 * 36;      // 4 waves x 4608 B <= 3 stages x 8640 B
#pragma unroll
    for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[r];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int c4 = (lane & 7) * 4, co4 = n0 + chf * 32 + c4;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + co4);
    float amx = 0.f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rr = it * 8 + (lane >> 3);
      const long m = (long)n * hw_o + (oy0 + 2 * ph + (rr >> 4)) * p.Wo + ox0 + (rr & 15);
      const float4 v = *reinterpret_cast<const float4*>(et + rr * 36 + c4);
      float e[4] = {__builtin_fmaf(v.x, asc, bias4.x), __builtin_fmaf(v.y, asc, bias4.y), __builtin_fmaf(v.z, asc, bias4.z),
                    __builtin_fmaf(v.w, asc, bias4.w)};
      if (p.epi_act != KEEP_ACT_NONE) {
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
      }
      if (p.res) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.res + m * p.res_ld + co4);
        const float rr4[4] = {r4.x, r4.y, r4.z, r4.w};
        if (p.aux) {
          const float4 a4 = *reinterpret_cast<const float4*>(p.aux + m * (long)p.Cout + co4);
          const float aa[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = rr4[q] + p.aux_w * (rr4[q] * aa[q] + e[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] += rr4[q];
        }
      }
      *reinterpret_cast<float4*>(p.out + m * p.out_ld + co4) = make_float4(e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) amx = fmaxf(amx, fabsf(e[q]));
    }
    if (p.out_amax) wave_amax_commit(p.out_amax + n, amx);
    return;
  }
#undef XP_BARRIER
  // the partial of this (tile, cout block, z): ws[z][m][co], scaled like the halo kernel's split-K epilogue (v * asc)
  const int co = n0 + chf * 32 + l31;
  if (co < p.Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const long m = (long)n * hw_o + (oy0 + 2 * ph + (rr >> 4)) * p.Wo + ox0 + (rr & 15);
      p.ws[((long)z * p.M + m) * p.Cout + co] = acc[r] * asc;
    }
  }
}

#ifndef XQ_DIST
#define XQ_DIST 2      // chunks between a chunk's loads and its conversion (3: the register ring holds three chunks in flight)
#endif
#ifndef XQ_ABL
#define XQ_ABL 0      // dev: 1 no staging after the first chunk, 2 no MFMAs, 3 no operand fetch after the prologue, 4 no fragment reads
#endif
// ------------------------------------------------------------------------------------------------ un-split plans: the whole convolution
// The same blocks for plans WITHOUT split-K on wide maps (32 x 32 .. 128 x 128 of one clip: 16 .. 128 items of conv3x3_halo_x3s_kernel on
// 256 CUs, each wave 108 MFMAs per chunk): this kernel reproduces THAT kernel's values -- its conversion arithmetic (fma(v, sc, sh);
// exp2(fma(y, -log2 e, pad)); + 1; rcp; y * r; hi = f16(y), lo = f16(y - hi)), its product order (a_lo b_hi, a_hi b_hi, a_hi b_lo),
// chunks and taps in order, e = fma(acc, acc_scale, bias) (+ residual), the per-image max|out| (a maximum: order-free) -- and
// conv_stats_replica_kernel (keep_conv_x3s.hip) re-reads the written tile in the streaming kernel's epilogue order for the GroupNorm
// partials.  Bit-equal outputs and statistics; the launch follows the REAL item count.
template <int PRO, bool AFF, int NCH>
__global__ __launch_bounds__(256) void conv3x3_x3q_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  static_assert(AFF || PRO == KEEP_PRO_NONE, "an activation prologue comes with its GroupNorm affine");
  __shared__ __attribute__((aligned(1024))) unsigned char ws0[XP_WBYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char ws1[XP_WBYTES];
  __shared__ __attribute__((aligned(1024))) unsigned char ws2[XP_WBYTES];
  // one LDS object per halo stage as well (round 6): the conversion of chunk c + 1 is scheduled INTO the MFMA sequence of chunk c (below), which
  // needs the compiler to know that its ds_writes and the fragment reads of chunk c touch different memory
  // (128 rows, not 108: every thread converts and parks two pieces -- rows 108 .. 127 are never read -- so that the conversion is one basic block)
  __shared__ __attribute__((aligned(16))) unsigned char hs0[128 * XPP * 2];
  __shared__ __attribute__((aligned(16))) unsigned char hs1[128 * XPP * 2];
  __shared__ __attribute__((aligned(16))) unsigned char hs2[128 * XPP * 2];
#define XP_HS(i) ((i) == 0 ? hs0 : ((i) == 1 ? hs1 : hs2))
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int ph = wave >> 1, chf = wave & 1;
  const int g = tid & 3;
  const int lid = xcd_remap(blockIdx.x, n_items);
  const int npt = p.N * tiles_y * tiles_x;
  const int pt = lid % npt, cb = lid / npt;
  const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
  const int oy0 = ty * 4, ox0 = tx * 16, n0 = cb * 64;
  const int nch = p.Cin >> 4;

  float in_s = 1.f, in_inv = 1.f;
  if (p.in_amax) x3_range_scale(p.in_amax[n], in_s, in_inv);
  const float rs = (PRO == KEEP_PRO_NONE && p.in_amax) ? in_s : 1.f;
  const int Hv = p.upsample ? 2 * p.H : p.H, Wv = p.upsample ? 2 * p.W : p.W;      // nearest x2 (VQ:149-150) folded into the halo addresses like the streaming kernel's
  int h_voff[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int hp = (tid >> 2) + k * 64;
    h_voff[k] = -16;
    if (hp < XP_HPIX) {
      const int hy = hp / XP_HW, hx = hp - hy * XP_HW;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
        const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
        h_voff[k] = ((sy * p.W + sx) * p.in_ld + g * 4) * 4;
      }
    }
  }
  auto make_rsrc = [&](const void* ptr, int bytes) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t in_rsrc = make_rsrc(p.in + (long)n * p.H * p.W * p.in_ld, p.H * p.W * p.in_ld * 4);
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wx3, p.Cout * 9 * p.Cin * 4);
  const long sc_off = (long)n * p.Cin + g * 4;
  int dma_voff[4];
  {
    const int lp = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = n0 + ((wave + j) & 3) * 16 + (lane >> 2);
      dma_voff[j] = co < p.Cout ? co * 9 * p.Cin * 4 + lp * 16 : -16;
    }
  }
  float4 hregs[XP_ST][2];
  u32x4 wregs[XP_ST][9];
  float4 sc4s[XP_ST], sh4s[XP_ST];
#pragma unroll
  for (int i = 0; i < XP_ST; ++i) {
    sc4s[i] = make_float4(1.f, 1.f, 1.f, 1.f);
    sh4s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto fetch = [&](int ch, int buf) {
    const int c0 = ch << 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, h_voff[k], c0 * 4, 0);
      hregs[buf][k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    if (AFF) {
      sc4s[buf] = *reinterpret_cast<const float4*>(p.pro_scale + sc_off + c0);
      sh4s[buf] = *reinterpret_cast<const float4*>(p.pro_shift + sc_off + c0);
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;
      wregs[buf][t] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, dma_voff[t & 3], ((q >> 2) * p.Cin + c0) * 4, 0);
    }
  };
  auto stage = [&](int buf) {      // conv3x3_halo_x3s_kernel::conv_step, all steps of a piece at once
    {
      unsigned char* wbase = XP_WS(buf);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;
        *reinterpret_cast<u32x4*>(wbase + q * 1024 + lane * 16) = wregs[buf][t];
      }
    }
    _Float16* Hs = reinterpret_cast<_Float16*>(XP_HS(buf));
    const float csc[4] = {sc4s[buf].x, sc4s[buf].y, sc4s[buf].z, sc4s[buf].w}, csh[4] = {sh4s[buf].x, sh4s[buf].y, sh4s[buf].z, sh4s[buf].w};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int hp = (tid >> 2) + k * 64;
      {      // (no `hp < XP_HPIX` test: rows beyond the halo hold zeros nobody reads)
        const float hv[4] = {hregs[buf][k].x, hregs[buf][k].y, hregs[buf][k].z, hregs[buf][k].w};
        const float padf = (AFF && h_voff[k] < 0) ? 1e30f : 0.f;
        float cv[4], cw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cv[j] = AFF ? __builtin_fmaf(hv[j], csc[j], csh[j]) : hv[j] * rs;
          if (PRO == KEEP_PRO_SWISH) {
            cw[j] = __builtin_fmaf(cv[j], -1.4426950408889634f, padf);
            cw[j] = __builtin_amdgcn_exp2f(cw[j]);
            cw[j] += 1.0f;
            cw[j] = __builtin_amdgcn_rcpf(cw[j]);
            cv[j] *= cw[j];
          } else if (AFF) {
            cv[j] *= rs;
            cv[j] = padf != 0.f ? 0.f : cv[j];
          }
        }
        const f16x2 h01 = __builtin_convertvector(f32x2{cv[0], cv[1]}, f16x2), h23 = __builtin_convertvector(f32x2{cv[2], cv[3]}, f16x2);
        const float l0 = __builtin_fmaf((float)h01.x, -1.0f, cv[0]), l1 = __builtin_fmaf((float)h01.y, -1.0f, cv[1]);
        const float l2 = __builtin_fmaf((float)h23.x, -1.0f, cv[2]), l3 = __builtin_fmaf((float)h23.y, -1.0f, cv[3]);
        const f16x2 q01 = __builtin_convertvector(f32x2{l0, l1}, f16x2), q23 = __builtin_convertvector(f32x2{l2, l3}, f16x2);
        const f16x4 hi = {h01.x, h01.y, h23.x, h23.y}, lo = {q01.x, q01.y, q23.x, q23.y};
        _Float16* dst = &Hs[hp * XPP + g * 4];
        *reinterpret_cast<f16x4*>(dst) = hi;
        *reinterpret_cast<f16x4*>(dst + 16) = lo;
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int a_base = ((2 * ph + (l31 >> 4)) * XP_HW + (l31 & 15)) * XPP + lhi * 8;
  const int b_base = l31 * 32 + ((lhi ^ ((l31 >> 2) & 3)) * 8);
  auto mma = [&](int buf) {
    const _Float16* Ws = reinterpret_cast<const _Float16*>(XP_WS(buf));
    const _Float16* Hs = reinterpret_cast<const _Float16*>(XP_HS(buf));
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const _Float16* src = &Hs[a_base + (kh * XP_HW + kw) * XPP];
        const f16x8 ah = *reinterpret_cast<const f16x8*>(src), al = *reinterpret_cast<const f16x8*>(src + 16);
        const int o = b_base + ((kh * 3 + kw) * 64 + chf * 32) * 32;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(&Ws[o]), bl = *reinterpret_cast<const f16x8*>(&Ws[o ^ 16]);
        if (XQ_ABL == 2) {
          acc[0] += (float)al[0] + (float)bh[0] + (float)ah[1] + (float)bl[1];
        } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);      // the streaming kernel's term order
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        }
      }
  };
  // Round 6: the 27 MFMAs of a chunk form ONE dependent chain (a wave owns one 32 x 32 accumulator): ~32 cycles each in which the wave can issue
  // whatever does not depend on them (up to five single-issue instructions per MFMA ride for free).  `fused(bc, bn)` issues the chunk
  // in stage bc and, one step per MFMA gap, the conversion + LDS writes of the chunk in stage bn (the expressions of `stage`, cut into 25 steps of
  // <= 4 VALU / 4 transcendentals / one or two LDS writes; every gap is pinned by a sched_barrier) -- the staging no longer queues behind the chain.
  auto fused = [&](int bc, int bn) __attribute__((always_inline)) {
    const _Float16* Ws = reinterpret_cast<const _Float16*>(XP_WS(bc));
    const _Float16* Hs = reinterpret_cast<const _Float16*>(XP_HS(bc));
    _Float16* Hn = reinterpret_cast<_Float16*>(XP_HS(bn));
    unsigned char* wn = XP_WS(bn);
    const float csc[4] = {sc4s[bn].x, sc4s[bn].y, sc4s[bn].z, sc4s[bn].w}, csh[4] = {sh4s[bn].x, sh4s[bn].y, sh4s[bn].z, sh4s[bn].w};
    float cv[2][4], cw[2][4], lo4[2][4];
    f16x2 h01[2], h23[2];
    auto step = [&](int i) __attribute__((always_inline)) {
      if (i < 16) {
        const int k = i >> 3, j = i & 7;
        const float hv[4] = {hregs[bn][k].x, hregs[bn][k].y, hregs[bn][k].z, hregs[bn][k].w};
        const float padf = (AFF && h_voff[k] < 0) ? 1e30f : 0.f;
        if (j == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) cv[k][q] = AFF ? __builtin_fmaf(hv[q], csc[q], csh[q]) : hv[q] * rs;
        } else if (j == 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (PRO == KEEP_PRO_SWISH) cw[k][q] = __builtin_fmaf(cv[k][q], -1.4426950408889634f, padf);
            else if (AFF) {
              cv[k][q] *= rs;
              cv[k][q] = padf != 0.f ? 0.f : cv[k][q];
            }
          }
        } else if (j == 2) {
          if (PRO == KEEP_PRO_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cw[k][q] = __builtin_amdgcn_exp2f(cw[k][q]);
          }
        } else if (j == 3) {
          if (PRO == KEEP_PRO_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cw[k][q] += 1.0f;
          }
        } else if (j == 4) {
          if (PRO == KEEP_PRO_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cw[k][q] = __builtin_amdgcn_rcpf(cw[k][q]);
          }
        } else if (j == 5) {
          if (PRO == KEEP_PRO_SWISH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cv[k][q] *= cw[k][q];
          }
        } else if (j == 6) {
          h01[k] = __builtin_convertvector(f32x2{cv[k][0], cv[k][1]}, f16x2);
          h23[k] = __builtin_convertvector(f32x2{cv[k][2], cv[k][3]}, f16x2);
          lo4[k][0] = __builtin_fmaf((float)h01[k].x, -1.0f, cv[k][0]);
          lo4[k][1] = __builtin_fmaf((float)h01[k].y, -1.0f, cv[k][1]);
        } else {
          lo4[k][2] = __builtin_fmaf((float)h23[k].x, -1.0f, cv[k][2]);
          lo4[k][3] = __builtin_fmaf((float)h23[k].y, -1.0f, cv[k][3]);
          const f16x2 q01 = __builtin_convertvector(f32x2{lo4[k][0], lo4[k][1]}, f16x2), q23 = __builtin_convertvector(f32x2{lo4[k][2], lo4[k][3]}, f16x2);
          const f16x4 hi = {h01[k].x, h01[k].y, h23[k].x, h23[k].y}, lo = {q01.x, q01.y, q23.x, q23.y};
          _Float16* dst = &Hn[((tid >> 2) + k * 64) * XPP + g * 4];
          *reinterpret_cast<f16x4*>(dst) = hi;
          *reinterpret_cast<f16x4*>(dst + 16) = lo;
        }
      } else if (i < 25) {
        const int t = i - 16;
        const int q = __builtin_amdgcn_readfirstlane(wave) * 9 + t;
        *reinterpret_cast<u32x4*>(wn + q * 1024 + lane * 16) = wregs[bn][t];
      }
    };
    const _Float16* src0 = &Hs[a_base];
    f16x8 ah = *reinterpret_cast<const f16x8*>(src0), al = *reinterpret_cast<const f16x8*>(src0 + 16);
    f16x8 bh = *reinterpret_cast<const f16x8*>(&Ws[b_base + (chf * 32) * 32]), bl = *reinterpret_cast<const f16x8*>(&Ws[(b_base + (chf * 32) * 32) ^ 16]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      f16x8 nah = ah, nal = al, nbh = bh, nbl = bl;
      if (t < 8) {      // the next tap's fragments, requested under this tap's MFMAs
        const int kh = (t + 1) / 3, kw = (t + 1) % 3;
        const _Float16* src = &Hs[a_base + (kh * XP_HW + kw) * XPP];
        nah = *reinterpret_cast<const f16x8*>(src);
        nal = *reinterpret_cast<const f16x8*>(src + 16);
        const int o = b_base + ((t + 1) * 64 + chf * 32) * 32;
        nbh = *reinterpret_cast<const f16x8*>(&Ws[o]);
        nbl = *reinterpret_cast<const f16x8*>(&Ws[o ^ 16]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);      // the streaming kernel's term order
      step(3 * t);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      step(3 * t + 1);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      step(3 * t + 2);
      __builtin_amdgcn_sched_barrier(0);
      ah = nah; al = nal; bh = nbh; bl = nbl;
    }
  };
#define XP_BARRIER()                  \
  __builtin_amdgcn_s_waitcnt(0xc07f); \
  __builtin_amdgcn_s_barrier();
  {
    const bool two = NCH > 0 ? NCH > 1 : nch > 1;
    fetch(0, 0);
    if (two) fetch(1, 1);
    if (XQ_DIST == 3 && NCH > 2) fetch(2, 2);
    stage(0);
    XP_BARRIER()
    if (NCH > 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c + XQ_DIST < NCH && XQ_ABL != 3) fetch(c + XQ_DIST, (c + XQ_DIST) % XP_ST);
        if (XQ_SCHED && XQ_ABL == 0 && c + 1 < NCH) {
          fused(c % XP_ST, (c + 1) % XP_ST);
          XP_BARRIER()
        } else {
          mma(c % XP_ST);
          if (c + 1 < NCH) {
            if (XQ_ABL != 1) stage((c + 1) % XP_ST);
            XP_BARRIER()
          }
        }
      }
    } else {
      for (int c0 = 0; c0 < nch; c0 += XP_ST) {
#pragma unroll
        for (int u = 0; u < XP_ST; ++u) {
          const int ch = c0 + u;
          if (ch < nch) {
            if (ch + 2 < nch) fetch(ch + 2, (u + 2) % XP_ST);
            mma(u);
            if (ch + 1 < nch) stage((u + 1) % XP_ST);
            XP_BARRIER()
          }
        }
      }
    }
  }
  // ---- epilogue: the wave parks its 32 x 32 tile (over halo stage 0 + 1: every wave is past its last fragment read) and writes rows of 32 channels
  XP_BARRIER()
#undef XP_BARRIER
  float* et = reinterpret_cast<float*>(ws0) + wave * 32 * 36;         // 4 waves x 4608 B of weight stage 0 (every wave is past its last fragment read)
#pragma unroll
  for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[r];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const float asc = p.acc_scale * in_inv;
  const int hw_o = p.Ho * p.Wo;
  const int c4 = (lane & 7) * 4, co = n0 + chf * 32 + c4;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + co);
  float amx = 0.f;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int rr = it * 8 + (lane >> 3);
    const long m = (long)n * hw_o + (oy0 + 2 * ph + (rr >> 4)) * p.Wo + ox0 + (rr & 15);
    const float4 v = *reinterpret_cast<const float4*>(et + rr * 36 + c4);
    float e[4] = {__builtin_fmaf(v.x, asc, bias4.x), __builtin_fmaf(v.y, asc, bias4.y), __builtin_fmaf(v.z, asc, bias4.z),
                  __builtin_fmaf(v.w, asc, bias4.w)};
    if (p.epi_act != KEEP_ACT_NONE) {      // (uniform; before the residual, the streaming kernel's order and forms)
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
  if (p.out_amax) wave_amax_commit(p.out_amax + n, amx);
}

// Geometry / epilogue this producer covers (everything else of a split-K plan stays on conv3x3_halo_x3_kernel: same partials).
bool keep_conv_x3p_ok(const keep_conv2d_args* a, const ConvP& p, int split_k) {
  const long items_old = (long)a->N * ((a->Ho * a->Wo) / 256) * ((a->Cout + 63) / 64) * split_k;
  return split_k > 1 && !a->upsample && a->pad_mode == KEEP_PAD_ZERO && a->Ho % 4 == 0 && a->Wo % 16 == 0 && (a->Ho * a->Wo) % 256 == 0 &&
         a->Cout % 64 == 0 && a->Cin % 16 == 0 && items_old <= 128 && a->workspace &&
         (a->pro_act == KEEP_PRO_NONE || (a->pro_act == KEEP_PRO_SWISH && p.fast)) && !(a->flags & KEEP_CONV_NO_SMALL_PARTIALS);
}

void keep_conv_stats_replica(const ConvP& p, int n_img, hipStream_t st);      // keep_conv_x3s.hip

// Un-split plans conv3x3_halo_x3_kernel would run (keep_conv2d_x3_halo has already offered them to the streaming kernel's family) with few
// items: the same 64-pixel blocks with the full epilogue.  Statistics only on 8 x 32-tile maps (the replica kernel's partition).
bool keep_conv_x3p_full_ok(const keep_conv2d_args* a, const ConvP& p, int split_k) {
  const long items_old = (long)a->N * ((a->Ho * a->Wo) / 256) * ((a->Cout + 63) / 64);
  return split_k == 1 && !a->upsample && a->pad_mode == KEEP_PAD_ZERO && a->Ho % 4 == 0 && a->Wo % 16 == 0 && (a->Ho * a->Wo) % 256 == 0 &&
         a->Cout % 64 == 0 && a->Cin % 16 == 0 && a->Cin >= 32 && items_old <= 64 && (!a->aux || a->residual) && a->out_ld % 4 == 0 &&
         (!a->residual || a->res_ld % 4 == 0) && (!p.stats || (a->Ho % 8 == 0 && a->Wo % 32 == 0 && a->out_ld == a->Cout)) &&
         (a->pro_act == KEEP_PRO_NONE || (a->pro_act == KEEP_PRO_SWISH && p.fast)) && !(a->flags & KEEP_CONV_NO_SMALL_PARTIALS);
}

int keep_conv2d_x3_partials(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  const int tiles_x = a->Wo / 16, tiles_y = a->Ho / 4, ncb = a->Cout / 64;
  const int n_items = a->N * tiles_x * tiles_y * ncb * p.split_k;
  const bool sc = a->pro_scale != nullptr;
  const int nchunks = a->Cin / 16, per = (nchunks + p.split_k - 1) / p.split_k;
  const int nch = (nchunks % p.split_k == 0 && (per == 4 || per == 8 || per == 16)) ? per : 0;      // every z walks `per` chunks
#define KEEP_LAUNCH_XP(PROV, SCV)                                                                                                     \
  {                                                                                                                                   \
    if (nch == 4) hipLaunchKernelGGL((conv3x3_x3p_kernel<PROV, SCV, 4>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items);        \
    else if (nch == 8) hipLaunchKernelGGL((conv3x3_x3p_kernel<PROV, SCV, 8>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items);   \
    else if (nch == 16) hipLaunchKernelGGL((conv3x3_x3p_kernel<PROV, SCV, 16>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items); \
    else hipLaunchKernelGGL((conv3x3_x3p_kernel<PROV, SCV, 0>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items);                 \
  }
  if (a->pro_act == KEEP_PRO_SWISH && sc) KEEP_LAUNCH_XP(KEEP_PRO_SWISH, true)
  else if (a->pro_act == KEEP_PRO_SWISH) KEEP_LAUNCH_XP(KEEP_PRO_SWISH, false)
  else if (sc) KEEP_LAUNCH_XP(KEEP_PRO_NONE, true)
  else KEEP_LAUNCH_XP(KEEP_PRO_NONE, false)
#undef KEEP_LAUNCH_XP
  KEEP_LAUNCH_CHECK("keep_conv2d(halo x3, small-tile partials)");
  if (p.split_k == 1 && p.stats) keep_conv_stats_replica(p, a->N, st);      // (the full-epilogue form: keep_conv_x3p_full_ok)
  return KEEP_OK;
}

// Un-split plans on wide maps (what conv3x3_halo_x3s_kernel takes) with few items: the 64-pixel blocks, same values.
#ifndef XQ_MAX_ITEMS
#define XQ_MAX_ITEMS 64      // items of the 256-pixel streaming kernel up to which the 64-pixel blocks take the map
#endif
bool keep_conv_x3q_ok(const keep_conv2d_args* a, const ConvP& p, int split_k) {
  const long items_s = (long)a->N * (a->Ho / 8) * (a->Wo / 32) * ((a->Cout + 63) / 64);
  const bool aff = a->pro_scale != nullptr;
  // round 6: an epilogue activation (CFT's scale.0|shift.0 + LeakyReLU, KA:468-469) and the 9-tap nearest-x2 form (the 16 -> 32 Upsample) too:
  // one uniform branch / two shifts in the halo addresses, the streaming kernel's own
  return split_k == 1 && a->upsample != KEEP_UPSAMPLE_X2_PHASES && a->pad_mode == KEEP_PAD_ZERO && a->Ho % 8 == 0 && a->Wo % 32 == 0 && a->Cout % 64 == 0 &&
         a->Cin % 16 == 0 && a->Cin >= 32 && items_s <= XQ_MAX_ITEMS && !a->aux &&
         ((a->pro_act == KEEP_PRO_SWISH && aff && p.fast) || a->pro_act == KEEP_PRO_NONE) && (!p.stats || a->out_ld == a->Cout) &&
         !(a->flags & (KEEP_CONV_NO_SMALL_PARTIALS | KEEP_CONV_NO_STREAM));
}

int keep_conv2d_x3_small_full(const keep_conv2d_args* a, ConvP& p, hipStream_t st) {
  const int tiles_x = a->Wo / 16, tiles_y = a->Ho / 4, ncb = a->Cout / 64;
  const int n_items = a->N * tiles_x * tiles_y * ncb;
  const bool aff = a->pro_scale != nullptr;
  const int nchunks = a->Cin / 16;
  const int nch = (nchunks == 4 || nchunks == 8 || nchunks == 16 || nchunks == 32) ? nchunks : 0;
#define KEEP_LAUNCH_XQ(PROV, AFFV)                                                                                                    \
  {                                                                                                                                   \
    if (nch == 4) hipLaunchKernelGGL((conv3x3_x3q_kernel<PROV, AFFV, 4>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items);        \
    else if (nch == 8) hipLaunchKernelGGL((conv3x3_x3q_kernel<PROV, AFFV, 8>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items);   \
    else if (nch == 16) hipLaunchKernelGGL((conv3x3_x3q_kernel<PROV, AFFV, 16>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items); \
    else if (nch == 32) hipLaunchKernelGGL((conv3x3_x3q_kernel<PROV, AFFV, 32>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items); \
    else hipLaunchKernelGGL((conv3x3_x3q_kernel<PROV, AFFV, 0>), dim3(n_items), dim3(256), 0, st, p, tiles_x, tiles_y, ncb, n_items);                 \
  }
  if (a->pro_act == KEEP_PRO_SWISH) KEEP_LAUNCH_XQ(KEEP_PRO_SWISH, true)
  else if (aff) KEEP_LAUNCH_XQ(KEEP_PRO_NONE, true)
  else KEEP_LAUNCH_XQ(KEEP_PRO_NONE, false)
#undef KEEP_LAUNCH_XQ
  KEEP_LAUNCH_CHECK("keep_conv2d(halo x3, small tiles)");
  if (p.stats) keep_conv_stats_replica(p, a->N, st);
  return KEEP_OK;
}
