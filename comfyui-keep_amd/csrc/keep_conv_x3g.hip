// keep_conv2d, KEEP_MMA_X3, 1x1 stride-1 convolutions with many rows (token GEMMs of GMFlow -- gmflow/transformer.py:60-185 q/k/v,
// merge, mlp -- and the 1x1 shortcuts of the VQGAN ResBlocks, vqgan_arch.py:166-181): the STREAMING form of conv_x3_kernel<.., ONE>
// (keep_conv_x3.hip; same arithmetic in the same order per accumulator, same operand formats: results are bit-identical).
//
// What the block-per-tile kernel loses on these shapes (round-4 ablations on 622 592 x 256 -> 1024, tools/dev/README.md "XG_ABL"):
// 2057 us as is, 1872 us WITHOUT its MFMAs, 1208 us without its output stores -- the K loop of a tile is 4-8 steps long, so a block
// spends its life in a prologue (first fetch: a memory round trip), a latency-bound K loop and an epilogue (accumulators -> LDS ->
// 64 KB of stores) that nothing overlaps, and because every tile costs the same, the two blocks of a CU -- and all 256 CUs -- go
// through these phases TOGETHER: the chip alternates between a load burst, a short matrix phase and a store burst.
//
// This kernel keeps two blocks per CU resident for the whole launch; a block walks tiles t, t + grid, ... and overlaps three tiles:
//   * the MFMAs of tile t accumulate into one of TWO accumulator sets;
//   * the epilogue of tile t - 1 (scale, bias, activation, 4x4 lane transposes by DPP so that a lane owns 4 consecutive output
//     columns, 16-byte stores) runs out of the other set in the gaps BETWEEN those MFMAs: no LDS, no barrier, and the stores leave
//     the CU spread over the next tile's K loop instead of as a burst;
//   * the first K step of tile t + 1 is fetched under the last step of tile t.
// The MFMAs of a step are issued term-major (a_lo*b_hi over the TM x TN accumulators, then a_hi*b_lo, then a_hi*b_hi) so that
// consecutive MFMAs never share an accumulator (a filler between two MFMAs on one accumulator is a +43-cycle cliff,
// MI355X_MICROARCH.md) while every accumulator still sees its three terms in the order of conv_x3_kernel; each gap carries at most
// one piece of epilogue (one value, half a transpose, or one store) and one piece of operand staging (two values split into hi / lo
// halves, or one ds_write_b128), pinned with sched_barrier(0); scalar fp32 only (-fno-slp-vectorize: packed fp32 math is an
// anti-lever beside MFMAs).  One fragment register set: a fragment is re-read for the second 16-channel half in the gap behind
// its last use (b_hi, needed by the first and the last term, has two).
#include <stdlib.h>

#include "keep_conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GX_BK 32                  // channels per K step
#define GX_XP (2 * GX_BK + 8)     // halves per LDS row: [hi x32 | lo x32 | pad] -- conv_x3_kernel's row format
#ifndef GX_ABL                    // dev builds (tools/dev/README.md): 1 = epilogue after the K loop instead of inside it
#define GX_ABL 0
#endif

__device__ __forceinline__ float gx_quad_x1(float v) {      // value of lane ^ 1
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float gx_quad_x2(float v) {      // value of lane ^ 2
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// WGM x WGN waves of TM x TN 32x32 accumulators; U: K steps per unrolled round (the previous tile's epilogue is spread over the
// first round: U <= K steps per tile); GELU: the fast-math GELU epilogue (the activation is a compile-time choice: a runtime switch
// in each of the 64 value pieces was 260 instructions per MFMA gap and 2000 spilled registers)
template <int WGM, int WGN, int TM, int TN, int U, bool GELU>
__global__ __launch_bounds__(256, 2) void gemm_x3s_kernel(ConvP p, int tile_cols, int n_tiles) {
  constexpr int BM = WGM * TM * 32, BN = WGN * TN * 32;
  constexpr int A_IT = BM / 64, B_IT = BN / 32;
  constexpr int A_BUF = BM * GX_XP, B_BUF = BN * GX_XP, B_OFF = 2 * A_BUF;      // halves
  constexpr int MF = TM * TN * 6;            // MFMAs (= gaps) per K step
  constexpr int NGRP = TM * TN * 4;          // 4-row x 4-column groups per lane and tile
  constexpr int NP = NGRP * 7;               // epilogue pieces per tile: 4 values, 2 transpose halves, 1 store per group
  constexpr int NSP = A_IT * 4 + B_IT;       // staging pieces per K step
  constexpr int NGAP = U * MF;
  static_assert(WGM * WGN == 4 && NSP <= MF, "tile config");
  __shared__ __attribute__((aligned(16))) _Float16 sm[2 * (BM + BN) * GX_XP];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN, l31 = lane & 31, lhi = lane >> 5;
  const int G = gridDim.x;
  int tM = xcd_remap(blockIdx.x, G);      // consecutive tiles (the column blocks of a row block) meet in one XCD's L2
  if (tM >= n_tiles) return;
  const int nsteps = p.Cin / GX_BK;
  const int hw = p.Ho * p.Wo;
  const int ld2 = p.Cin - p.cin1;

  auto make_rsrc = [&](const void* ptr, long bytes) __attribute__((always_inline)) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  };

  // ---- operand fetch (state F: the tile whose K steps are being requested)
  const int a_grp = tid & 3, a_row0 = tid >> 2;          // rows a_row0 + it * 64, channels a_grp * 8 .. + 8 of the step
  const int b_pc = tid & 7, b_row0 = tid >> 3;           // rows b_row0 + it * 32, 16-byte piece b_pc of the step's 128 B
  // b piece -> LDS column: 16-channel chunk c = b_pc >> 2, part = b_pc & 3 (0,1: hi ch 0-7 / 8-15; 2,3: lo)
  const int b_col = ((b_pc & 3) >> 1) * GX_BK + (b_pc >> 2) * 16 + (b_pc & 1) * 8;
  int a_voff[A_IT], a2_voff[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    a_voff[it] = ((a_row0 + it * 64) * p.in_ld + a_grp * 8) * 4;
    a2_voff[it] = ((a_row0 + it * 64) * ld2 + a_grp * 8) * 4;
  }
  const int a_wr = a_row0 * GX_XP + a_grp * 8;
  const int b_wr = B_OFF + b_row0 * GX_XP + b_col;
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wx3, (long)p.Cout * p.Cin * 4);
  __amdgpu_buffer_rsrc_t a_rsrc = w_rsrc, a2_rsrc = w_rsrc;
  int b_voff[B_IT];
  float a_s = 1.f;                    // range scale of the tile's image (x3_in_amax), 1 without
  float a_raw[A_IT][8];
  u32x4 b_raw[B_IT];

  auto setF = [&](int t) __attribute__((always_inline)) {
    const int bx = t / tile_cols, by = t - bx * tile_cols;
    const long m0 = (long)bx * BM;
    const long rows = (p.M - m0) < BM ? (p.M - m0) : BM;
    a_rsrc = make_rsrc(p.in + m0 * p.in_ld, rows * p.in_ld * 4);
    if (p.in2) a2_rsrc = make_rsrc(p.in2 + m0 * ld2, rows * ld2 * 4);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int co = by * BN + b_row0 + it * 32;
      b_voff[it] = co < p.Cout ? co * p.Cin * 4 + b_pc * 16 : (int)0x80000000;
    }
    if (p.in_amax) {
      float inv;
      x3_range_scale(p.in_amax[m0 / hw], a_s, inv);
    }
  };
  auto fetch = [&](int s) __attribute__((always_inline)) {
    const int c0 = s * GX_BK;
    const bool second = p.in2 && c0 >= p.cin1;            // uniform: a K step never straddles the seam (cin1 % 32 == 0)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const u32x4 v0 = second ? __builtin_amdgcn_raw_buffer_load_b128(a2_rsrc, a2_voff[it], (c0 - p.cin1) * 4, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[it], c0 * 4, 0);
      const u32x4 v1 = second ? __builtin_amdgcn_raw_buffer_load_b128(a2_rsrc, a2_voff[it] + 16, (c0 - p.cin1) * 4, 0)
                              : __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[it] + 16, c0 * 4, 0);
      a_raw[it][0] = __uint_as_float(v0.x); a_raw[it][1] = __uint_as_float(v0.y);
      a_raw[it][2] = __uint_as_float(v0.z); a_raw[it][3] = __uint_as_float(v0.w);
      a_raw[it][4] = __uint_as_float(v1.x); a_raw[it][5] = __uint_as_float(v1.y);
      a_raw[it][6] = __uint_as_float(v1.z); a_raw[it][7] = __uint_as_float(v1.w);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) b_raw[it] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff[it], c0 * 4, 0);
  };

  // ---- staging pieces: the fetched step -> LDS buffer wb (0 | 1), the A values split into hi / lo halves on the way
  unsigned s_hi[4], s_lo[4];
  auto stage_piece = [&](int sp, int wb) __attribute__((always_inline)) {
    if (sp < A_IT * 4) {
      const int it = sp >> 2, q = sp & 3;
      const float v0 = a_raw[it][2 * q] * a_s, v1 = a_raw[it][2 * q + 1] * a_s;
      const f16x2 h = __builtin_convertvector(f32x2{v0, v1}, f16x2);
      const float w0 = __builtin_fmaf((float)h.x, -1.0f, v0), w1 = __builtin_fmaf((float)h.y, -1.0f, v1);      // v_fma_mix_f32
      const f16x2 l = __builtin_convertvector(f32x2{w0, w1}, f16x2);
      s_hi[q] = __builtin_bit_cast(unsigned, h);
      s_lo[q] = __builtin_bit_cast(unsigned, l);
      if (q == 3) {
        _Float16* dst = sm + wb * A_BUF + a_wr + it * 64 * GX_XP;
        *reinterpret_cast<uint4*>(dst) = make_uint4(s_hi[0], s_hi[1], s_hi[2], s_hi[3]);
        *reinterpret_cast<uint4*>(dst + GX_BK) = make_uint4(s_lo[0], s_lo[1], s_lo[2], s_lo[3]);
      }
    } else {
      const int it = sp - A_IT * 4;
      const u32x4 v = b_raw[it];
      *reinterpret_cast<uint4*>(sm + wb * B_BUF + b_wr + it * 32 * GX_XP) = make_uint4(v.x, v.y, v.z, v.w);
    }
  };

  // ---- epilogue pieces (state E: the tile whose accumulators are finished)
  const bool odd1 = (lane & 1) != 0, odd2 = (lane & 2) != 0;
  __amdgpu_buffer_rsrc_t o_rsrc = w_rsrc;
  int o_voff[TN];                      // lane part of the store offset per column block of the wave; 0x80000000 for lanes beyond Cout
#pragma unroll
  for (int j = 0; j < TN; ++j) o_voff[j] = (int)0x80000000;
  float e_scale = 1.f;
  float biasE[TN], biasM[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { biasE[j] = 0.f; biasM[j] = 0.f; }
  const int ld4 = p.out_ld * 4;
  float ev[4];
  auto epi_piece = [&](int q, f32x16 (&acc)[TM][TN]) __attribute__((always_inline)) {
    const int e = q / 7, k = q - e * 7;
    const int g = e & 3, j = (e >> 2) % TN, i = (e >> 2) / TN;
    if (k < 4) {
      float v = __builtin_fmaf(acc[i][j][4 * g + k], e_scale, biasE[j]);
      if (GELU) v = act_apply_fast(v, KEEP_ACT_GELU);
      ev[k] = v;
    } else if (k == 4) {      // 4x4 transpose over (4 lanes) x (4 registers), first half: exchange with lane ^ 1
      const float r01 = gx_quad_x1(odd1 ? ev[0] : ev[1]);
      const float r23 = gx_quad_x1(odd1 ? ev[2] : ev[3]);
      ev[0] = odd1 ? r01 : ev[0]; ev[1] = odd1 ? ev[1] : r01;
      ev[2] = odd1 ? r23 : ev[2]; ev[3] = odd1 ? ev[3] : r23;
    } else if (k == 5) {      // second half: exchange with lane ^ 2 -- lane 4c + r now holds row r, columns 4c .. 4c + 3
      const float r02 = gx_quad_x2(odd2 ? ev[0] : ev[2]);
      const float r13 = gx_quad_x2(odd2 ? ev[1] : ev[3]);
      ev[0] = odd2 ? r02 : ev[0]; ev[2] = odd2 ? ev[2] : r02;
      ev[1] = odd2 ? r13 : ev[1]; ev[3] = odd2 ? ev[3] : r13;
    } else {
      // (no SGPR in the soffset field: with one, hipcc drops the wait state between a 128-bit store and the next write of its data
      // registers -- "no hazard" per the ISA notes -- and on gfx950 the last dword of lanes 12-15 of each row was then read late:
      // one stale value per ~25 000, round 4)
      u32x4 o;
      o.x = __float_as_uint(ev[0]); o.y = __float_as_uint(ev[1]); o.z = __float_as_uint(ev[2]); o.w = __float_as_uint(ev[3]);
      __builtin_amdgcn_raw_buffer_store_b128(o, o_rsrc, o_voff[j] + (i * 32 + 8 * g) * ld4, 0, 0);
    }
  };
  auto setE = [&](int t) __attribute__((always_inline)) {      // after tile t's K loop: its accumulators become the pending ones
    const int bx = t / tile_cols, by = t - bx * tile_cols;
    const long m0 = (long)bx * BM;
    const long rows = (p.M - m0) < BM ? (p.M - m0) : BM;
    o_rsrc = make_rsrc(p.out + m0 * p.out_ld, rows * (long)ld4);
    // accumulator register r of a 32x32 tile holds row (r & 3) + 8 * (r >> 2) + 4 * lhi; after the transposes the lane's row is lane & 3
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = by * BN + wn * TN * 32 + j * 32 + (l31 >> 2) * 4;
      o_voff[j] = col < p.Cout ? (wm * TM * 32 + 4 * lhi + (lane & 3)) * ld4 + col * 4 : (int)0x80000000;
    }
    e_scale = p.acc_scale;
    if (p.in_amax) {
      float sr, inv;
      x3_range_scale(p.in_amax[m0 / hw], sr, inv);
      e_scale *= inv;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) biasE[j] = biasM[j];
  };

  // ---- one K step: MF MFMAs, term-major, with the scheduled pieces in the gaps
  const int a_f0 = (wm * TM * 32 + l31) * GX_XP + lhi * 8;
  const int b_f0 = B_OFF + (wn * TN * 32 + l31) * GX_XP + lhi * 8;
  int buf = 0;
  auto step = [&](int u, bool do_epi, bool do_stage, f32x16 (&accM)[TM][TN], f32x16 (&accE)[TM][TN]) __attribute__((always_inline)) {
    const _Float16* Ab = sm + buf * A_BUF + a_f0;
    const _Float16* Bb = sm + buf * B_BUF + b_f0;
    f16x8 ah[TM], al[TM], bl[TN], bh[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      al[i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * GX_XP + GX_BK);
      ah[i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * GX_XP);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bh[0][j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * GX_XP);
      bl[j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * GX_XP + GX_BK);
      bh[1][j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * GX_XP + 16);
    }
    auto gap = [&](int m) __attribute__((always_inline)) {
      if (GX_ABL != 1 && do_epi) {
        const int gg = u * MF + m;
#pragma unroll
        for (int q = gg * NP / NGAP; q < (gg + 1) * NP / NGAP; ++q) epi_piece(q, accE);
      }
      if (do_stage && m >= MF - NSP) stage_piece(m - (MF - NSP), buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[ks][j], accM[i][j], 0, 0, 0);
          if (ks == 0 && j == TN - 1) al[i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * GX_XP + 16 + GX_BK);
          gap(ks * (MF / 2) + i * TN + j);
        }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accM[i][j], 0, 0, 0);
          if (ks == 0 && i == TM - 1) bl[j] = *reinterpret_cast<const f16x8*>(Bb + j * 32 * GX_XP + 16 + GX_BK);
          gap(ks * (MF / 2) + TM * TN + j * TM + i);
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          accM[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[ks][j], accM[i][j], 0, 0, 0);
          if (ks == 0 && j == TN - 1) ah[i] = *reinterpret_cast<const f16x8*>(Ab + i * 32 * GX_XP + 16);
          gap(ks * (MF / 2) + 2 * TM * TN + i * TN + j);
        }
    }
  };

  // ---- one tile: its K loop (+ the pending tile's epilogue in the first round), then it becomes the pending tile
  bool pend = false;
  auto pass = [&](f32x16 (&accM)[TM][TN], f32x16 (&accE)[TM][TN]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accM[i][j][r] = 0.f;
    if (p.bias) {      // this tile's bias values: read a whole K loop before their first use
      const int by = tM % tile_cols;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int co = by * BN + wn * TN * 32 + j * 32 + l31;
        biasM[j] = co < p.Cout ? p.bias[co] : 0.f;
      }
    }
    const bool has_next = tM + G < n_tiles;
    for (int s0 = 0; s0 < nsteps; s0 += U) {
      const bool do_epi = pend && s0 == 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = s0 + u;
        if (s < nsteps) {
          const bool more = s + 1 < nsteps;
          if (more) {
            fetch(s + 1);
          } else if (has_next) {
            setF(tM + G);
            fetch(0);
          }
          step(u, do_epi, more || has_next, accM, accE);
          __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's ds_writes of the next step have landed
          __builtin_amdgcn_s_barrier();
          buf ^= 1;
        }
      }
    }
    if (GX_ABL == 1 && pend) {
#pragma unroll
      for (int q = 0; q < NP; ++q) epi_piece(q, accE);
    }
    setE(tM);
    pend = true;
    tM += G;
  };

  setF(tM);
  fetch(0);
#pragma unroll
  for (int sp = 0; sp < NSP; ++sp) stage_piece(sp, 0);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  f32x16 acc0[TM][TN], acc1[TM][TN];
  while (true) {
    pass(acc0, acc1);
    if (tM >= n_tiles) {
#pragma unroll
      for (int q = 0; q < NP; ++q) epi_piece(q, acc0);
      break;
    }
    pass(acc1, acc0);
    if (tM >= n_tiles) {
#pragma unroll
      for (int q = 0; q < NP; ++q) epi_piece(q, acc1);
      break;
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
bool keep_conv_x3_gather_is_gemm(const keep_conv2d_args* a);

static bool gx_off() { return getenv("KEEP_X3_NO_GEMM_STREAM") != nullptr; }      // (read per call: the A/B test toggles it)

// the launches this kernel takes from conv_x3_kernel<.., PLAIN, ONE> (keep_conv2d_x3_gather): plain 1x1 GEMMs with enough row
// tiles to keep two resident blocks per CU busy for several tiles each, and the simple epilogue (bias, activation)
bool keep_conv_x3_gemm_stream_ok(const keep_conv2d_args* a, const ConvP& p, int split_k) {
  if (gx_off() || !keep_conv_x3_gather_is_gemm(a)) return false;
  if (a->pro_scale || a->pro_act != KEEP_PRO_NONE || split_k != 1 || a->aux || a->residual || a->stats_out || a->x3_out_amax)
    return false;
  if (a->epi_act != KEEP_ACT_NONE && !(a->epi_act == KEEP_ACT_GELU && p.fast)) return false;
  if (a->Cin % GX_BK != 0 || a->Cin < 2 * GX_BK || a->Cout % 4 != 0 || a->out_ld % 4 != 0 || (uintptr_t)a->out % 16 != 0) return false;
  if (a->in2 && (a->in2_cin1 % GX_BK != 0)) return false;
  const long M = (long)a->N * a->Ho * a->Wo;
  const int bn = a->Cout <= 64 ? 64 : 128;
  const long tiles = ((M + 127) / 128) * ((a->Cout + bn - 1) / bn);
  if (tiles < 2048 || tiles >= (1L << 30)) return false;
  if (a->x3_in_amax && ((long)a->Ho * a->Wo) % 128 != 0) return false;      // a row tile lies in one image: one range scale per tile
  if (128L * a->in_ld * 4 >= (1L << 31) || 128L * a->out_ld * 4 >= (1L << 31) || (long)a->Cout * a->Cin * 4 >= (1L << 31)) return false;
  return true;
}

int keep_conv2d_x3_gemm_stream(const keep_conv2d_args* a, ConvP& p, int n_cu, hipStream_t st) {
  const bool narrow = a->Cout <= 64;
  const int bn = narrow ? 64 : 128;
  const int tile_cols = (a->Cout + bn - 1) / bn;
  const long tiles = (((long)p.M + 127) / 128) * tile_cols;
  const int nsteps = a->Cin / GX_BK;
  const int grid = (int)(tiles < 2L * n_cu ? tiles : 2L * n_cu);
  p.tile_cols = tile_cols;
#define KEEP_LAUNCH_GS2(WGM, WGN, TM, TN, UU)                                                                                    \
  if (a->epi_act == KEEP_ACT_GELU)                                                                                               \
    hipLaunchKernelGGL((gemm_x3s_kernel<WGM, WGN, TM, TN, UU, true>), dim3(grid), dim3(256), 0, st, p, tile_cols, (int)tiles);  \
  else                                                                                                                           \
    hipLaunchKernelGGL((gemm_x3s_kernel<WGM, WGN, TM, TN, UU, false>), dim3(grid), dim3(256), 0, st, p, tile_cols, (int)tiles);
#define KEEP_LAUNCH_GS(WGM, WGN, TM, TN)   \
  if (nsteps >= 8) {                       \
    KEEP_LAUNCH_GS2(WGM, WGN, TM, TN, 8)   \
  } else if (nsteps >= 4) {                \
    KEEP_LAUNCH_GS2(WGM, WGN, TM, TN, 4)   \
  } else {                                 \
    KEEP_LAUNCH_GS2(WGM, WGN, TM, TN, 2)   \
  }
  if (narrow) {
    KEEP_LAUNCH_GS(4, 1, 1, 2)
  } else {
    KEEP_LAUNCH_GS(2, 2, 2, 2)
  }
#undef KEEP_LAUNCH_GS
#undef KEEP_LAUNCH_GS2
  KEEP_LAUNCH_CHECK("keep_conv2d(gemm x3s)");
  return KEEP_OK;
}
