// keep_conv2d, KEEP_MMA_X3, 3x3 stride-1 convolutions on 8x32-pixel tiles: the STREAMING form of conv3x3_halo_x3_kernel
// (keep_conv_x3.hip; same arithmetic, same operand formats, same epilogue contract -- VQGAN ResBlock convolutions,
// /root/reference/modules/deps/wm_basicsr/archs/vqgan_arch.py:155-181: GroupNorm -> swish -> conv3x3 (+ residual)).
//
// What round 3's kernel lost: its wave spent 27 % of its time turning fp32 halo values into split fp16 rows (affine, swish,
// hi / lo split: ~36 VALU instructions per 16-byte piece) in a phase of its own, behind a barrier, while the matrix pipe of its
// SIMD waited for the partner wave -- SIMD 80 % busy, 57 % matrix.  tools/dev/coissue_probe2.hip (round 4) measures what the
// hardware offers instead: behind each v_mfma_f32_32x32x16_f16 of a wave, up to FIVE single-issue instructions of the SAME wave
// are free (17.7 ns per MFMA with 0..5 v_fma_f32; 6: 21.0 ns), two such waves per SIMD run at the bare MFMA rate with 4.8
// fillers per gap (fma + cvt + exp | rcp + ds_read_b128 + ds_write_b64: 18.0 vs 17.4 ns), ONE v_pk_fma_f32 per gap costs +4 ns
// (packed fp32 math is an anti-lever beside MFMAs), and a filler between two MFMAs on the same accumulator is a +43-cycle cliff
// (MI355X_MICROARCH.md).  Round 3's "VALU is paid in full" was the two-DIFFERENT-waves case at 8 VALU per gap.
//
// So this kernel is ONE software pipeline per wave over the block's whole list of (item, 16-channel chunk) steps:
//   step s:   MFMAs of chunk s                      (108, term-major: a_lo*b_hi, a_hi*b_hi, a_hi*b_lo over the four accumulators,
//                                                    so consecutive MFMAs never share an accumulator)
//             + conversion of chunk s+1             (13 steps of <= 4 scalar-f32 VALU / <= 2 transcendentals per piece, one step per
//                                                    MFMA gap, pinned with sched_barrier(0); the packed rows go straight to the OTHER
//                                                    halo buffer: ds_write_b64 x 2 per piece)
//             + request of chunk s+2                (buffer_load_dwordx4 of a piece as soon as its raw registers are consumed)
//             + fragment reads of the next tap      (ONE register set: a fragment is re-read in the gap behind its last use)
//   barrier;  weights of chunk s+1: L2 -> LDS by LDS-DMA;  [last chunk of an item: epilogue];  barrier.
// Items follow each other without a seam: the first chunk of the next item is converted under the last chunk of the current one,
// its weights land under the epilogue.  LDS: two halo buffers of 340 rows x 64 B ([hi16 | lo16], 16-byte pieces XOR-swizzled by
// (x >> 2) & 3 of the pixel's halo column -- conflict-free ds_read_b128 with an ADDITIVE tap displacement) + 576 weight rows x 64 B
// (as keep_conv_x3.hip: WDMA) + 2 x 512 B that catch the writes of the lanes without a sixth piece = 81408 B: two blocks per CU.
// The epilogue parks 16 pixel rows at a time in the halo buffer the last chunk just released (the other one already holds the next
// item's first chunk).
#include <stdlib.h>

#include "keep_conv_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define XS_HW 34                              // halo width (32 + 2)
#define XS_PIX 340                            // 10 x 34
#define XS_HBUF (XS_PIX * 64 + 512)           // bytes per halo buffer: 340 rows + 512 B that catch the ds_writes of (thread, piece 5) pairs beyond the halo
#define XS_WOFF (2 * XS_HBUF)                 // weight rows
#define XS_LDS (XS_WOFF + 576 * 64)           // 81408 B: two blocks per CU
#define XS_EP 68                              // floats per parked epilogue row
#ifndef XS_ABL                                // dev builds: phase ablations (tools/dev/README.md); 0 = the product
#define XS_ABL 0
#endif

__device__ __forceinline__ float xs_xor16_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float xs_xor32_sum(float x) {
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// TL: dev builds (-DKEEP_X3_ABLATE, KEEP_X3_EXP=21): wave 0's s_memtime timeline summed into p.ws
template <int PRO, bool AFF, bool TL = false>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_x3s_kernel(ConvP p, int tiles_x, int tiles_y, int ncb, int n_items) {
  static_assert(AFF || PRO == KEEP_PRO_NONE, "an activation prologue comes with its GroupNorm affine");
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[XS_LDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5, g = tid & 3;
  const int nch = p.Cin >> 4;                  // chunks per item (>= 2: host)
  const int Hv = p.upsample ? 2 * p.H : p.H, Wv = p.upsample ? 2 * p.W : p.W;
  if ((int)blockIdx.x >= n_items) return;

  auto make_rsrc = [&](const void* ptr, int bytes) __attribute__((always_inline)) {
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);   // every offset out of range: zeros, no traffic
  const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wx3, p.Cout * 9 * p.Cin * 4);
  const __amdgpu_buffer_rsrc_t sc_rsrc = AFF ? make_rsrc(p.pro_scale, p.N * p.Cin * 4) : null_rsrc;
  const __amdgpu_buffer_rsrc_t sh_rsrc = AFF ? make_rsrc(p.pro_shift, p.N * p.Cin * 4) : null_rsrc;

  // ---- per-thread constants
  // halo piece k of this thread: pixel hp = tid / 4 + 64 k (row-major in the 10 x 34 halo), channels 4 g .. 4 g + 3 of the chunk.
  // LDS row = 64 B: logical 16-byte pieces [hi 0-7 | hi 8-15 | lo 0-7 | lo 8-15], physical piece = logical ^ ((hx >> 2) & 3).
  int wr_addr[HALO_IT];                        // byte address (inside a halo buffer) of this thread's 8 hi bytes; lo = addr ^ 32
#pragma unroll
  for (int k = 0; k < HALO_IT; ++k) {
    const int hp = (tid >> 2) + k * 64;
    const int hx = hp % XS_HW;
    wr_addr[k] = hp < XS_PIX ? hp * 64 + (((g >> 1) ^ ((hx >> 2) & 3)) << 4) + (g & 1) * 8 : XS_PIX * 64 + lane * 8;
  }
  // A fragments: lane -> pixel (2 wave + i + kh, l31 + kw) of the halo; the swizzle depends on l31 + kw only: one base per kw,
  // everything else of (tap, i) is an immediate offset
  int rd_a[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) rd_a[kw] = ((2 * wave) * XS_HW + l31 + kw) * 64 + ((lhi ^ (((l31 + kw) >> 2) & 3)) << 4);
  // B fragments (weight rows, WDMA layout of keep_conv_x3.hip): physical slot = logical piece ^ ((row >> 2) & 3)
  const int rd_b = XS_WOFF + l31 * 64 + ((lhi ^ ((l31 >> 2) & 3)) << 4);

  // ---- pipeline state.  F: the chunk to REQUEST next; C: the chunk to convert / whose weights to fetch next; M: the chunk being multiplied
  HaloItem itF = halo_decode<32, 4>(p, blockIdx.x, n_items, tiles_x, tiles_y, ncb), itC = itF, itM = itF;
  int chF = 0, chC = 0, chM = 0, itemF = blockIdx.x;
  bool okF = true, okC = false;
  int h_voff[HALO_IT];                         // F: byte offset of piece k inside the image; < 0: zero padding
  int sc_voff = 0;                             // F: byte offset of this thread's 4 channels in pro_scale / pro_shift (chunk 0)
  __amdgpu_buffer_rsrc_t in_rsrc = null_rsrc;  // F
  float amaxF = 0.f;                           // F: max |input| of the image (raw inputs only)
  unsigned padmask = 0;                        // C: bit k = piece k is zero padding
  int dma_voff[4] = {-16, -16, -16, -16};      // C: this lane's source offsets of the 16-cout row groups (j - wave) & 3, j = 0..3
  float in_sC = 1.f, in_invC = 1.f, in_invM = 1.f;
  bool after_epi = false;                      // the previous step ended with an epilogue
  float4 biasM = make_float4(0.f, 0.f, 0.f, 0.f);      // M: bias of the item's cout block, this lane's four epilogue channels
  const int items_per_z = n_items;

  auto setup_F = [&]() __attribute__((always_inline)) {                       // itF -> h_voff, in_rsrc, sc_voff, amaxF
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) {
      const int hp = (tid >> 2) + k * 64;
      h_voff[k] = -16;
      if (hp < XS_PIX) {
        const int hy = hp / XS_HW, hx = hp - hy * XS_HW;
        int iy = itF.oy0 - 1 + hy, ix = itF.ox0 - 1 + hx;
        KEEP_REFLECT(iy, ix, Hv, Wv)
        if (iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) {
          const int sy = p.upsample ? (iy >> 1) : iy, sx = p.upsample ? (ix >> 1) : ix;
          h_voff[k] = ((sy * p.W + sx) * p.in_ld + g * 4) * 4;
        }
      }
    }
    in_rsrc = make_rsrc(p.in + (long)itF.n * p.H * p.W * p.in_ld, p.H * p.W * p.in_ld * 4);
    sc_voff = (itF.n * p.Cin + g * 4) * 4;
    if (p.in_amax) amaxF = p.in_amax[itF.n];
  };
  auto cross_C = [&]() __attribute__((always_inline)) {                       // C enters the item F points at: its padding mask, weight rows, range scale
    padmask = 0;
#pragma unroll
    for (int k = 0; k < HALO_IT; ++k) padmask |= (h_voff[k] < 0 ? 1u : 0u) << k;
    const int lp = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = itF.n0 + ((j - wave) & 3) * 16 + (lane >> 2);
      dma_voff[j] = co < p.Cout ? co * 9 * p.Cin * 4 + lp * 16 : -16;
    }
    in_sC = 1.f;
    in_invC = 1.f;
    if (p.in_amax) x3_range_scale(amaxF, in_sC, in_invC);
  };
  auto advance_F = [&]() __attribute__((always_inline)) {                     // F moves on by one chunk (possibly into the block's next item)
    if (++chF < nch) return;
    chF = 0;
    itemF += gridDim.x;
    okF = itemF < n_items;
    if (okF) {
      itF = halo_decode<32, 4>(p, itemF, items_per_z, tiles_x, tiles_y, ncb);
      setup_F();
    }
  };

  // ---- registers of the pipeline
  float4 hreg[HALO_IT];                        // raw pieces of the chunk to convert next (then: in flight for the one after)
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float cv[4] = {0.f, 0.f, 0.f, 0.f}, cw[4] = {0.f, 0.f, 0.f, 0.f};
  f16x2 chi[2] = {f16x2{(_Float16)0.f, (_Float16)0.f}, f16x2{(_Float16)0.f, (_Float16)0.f}};
  f16x2 clo[2] = {f16x2{(_Float16)0.f, (_Float16)0.f}, f16x2{(_Float16)0.f, (_Float16)0.f}};
  float padf = 0.f;                            // piece in flight: 1e30 for a zero-padding piece
  f32x16 acc[2][2];

  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define XS_T(IDX)                                                     \
  if (TL) {                                                           \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();       \
    tacc[IDX] += t1 - t0;                                             \
    t0 = t1;                                                          \
  }

  // Conversion of piece k = q / 18 of the chunk in `hreg`, step q % 18 -- at most two plain VALU instructions or two transcendentals per
  // step, one step per MFMA gap (the budget per gap is five issue slots including the fragment read and its wait: coissue_probe2);
  // hb: byte offset of the halo buffer the packed rows go to.  The request of the same piece of the chunk after it (F) rides in step 1:
  // the raw registers are dead from there on.  Zero padding applies to the normalised + activated tensor: under the swish the exp2
  // argument of a padding piece gets +1e30 (exp2 -> inf, rcp -> 0, y * 0 = 0) -- no select instructions.
  auto conv_step = [&](int q, int hb) __attribute__((always_inline)) {
    constexpr bool SW = PRO == KEEP_PRO_SWISH, RL = PRO == KEEP_PRO_RELU;
    const int k = q / 18, st = q % 18;
    const float rs = (PRO == KEEP_PRO_NONE && p.in_amax) ? in_sC : 1.f;
    const float csc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, csh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    if (st == 0) {
      cv[0] = AFF ? __builtin_fmaf(hreg[k].x, csc[0], csh[0]) : hreg[k].x * rs;
      cv[1] = AFF ? __builtin_fmaf(hreg[k].y, csc[1], csh[1]) : hreg[k].y * rs;
    }
    if (st == 1) {
      cv[2] = AFF ? __builtin_fmaf(hreg[k].z, csc[2], csh[2]) : hreg[k].z * rs;
      cv[3] = AFF ? __builtin_fmaf(hreg[k].w, csc[3], csh[3]) : hreg[k].w * rs;
      if (XS_ABL != 11) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(okF ? in_rsrc : null_rsrc, h_voff[k], chF * 64, KEEP_LD_AUX_XS);
        hreg[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
      }
      if (AFF && k == HALO_IT - 1 && XS_ABL != 11) {           // the affine of this chunk has been read for the last time: the next chunk's
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(okF ? sc_rsrc : null_rsrc, sc_voff, chF * 64, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(okF ? sh_rsrc : null_rsrc, sc_voff, chF * 64, 0);
        sc4 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        sh4 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
      }
    }
    if (st == 2 && AFF) padf = (padmask & (1u << k)) ? 1e30f : 0.f;
    if (XS_ABL == 10 && st >= 3 && st <= 12) return;
    if (SW) {
      if (st == 3) { cw[0] = __builtin_fmaf(cv[0], -1.4426950408889634f, padf); cw[1] = __builtin_fmaf(cv[1], -1.4426950408889634f, padf); }
      if (st == 4) { cw[2] = __builtin_fmaf(cv[2], -1.4426950408889634f, padf); cw[3] = __builtin_fmaf(cv[3], -1.4426950408889634f, padf); }
      if (st == 5) { cw[0] = __builtin_amdgcn_exp2f(cw[0]); cw[1] = __builtin_amdgcn_exp2f(cw[1]); }
      if (st == 6) { cw[2] = __builtin_amdgcn_exp2f(cw[2]); cw[3] = __builtin_amdgcn_exp2f(cw[3]); }
      if (st == 7) { cw[0] += 1.0f; cw[1] += 1.0f; }
      if (st == 8) { cw[2] += 1.0f; cw[3] += 1.0f; }
      if (st == 9) { cw[0] = __builtin_amdgcn_rcpf(cw[0]); cw[1] = __builtin_amdgcn_rcpf(cw[1]); }
      if (st == 10) { cw[2] = __builtin_amdgcn_rcpf(cw[2]); cw[3] = __builtin_amdgcn_rcpf(cw[3]); }
      if (st == 11) { cv[0] *= cw[0]; cv[1] *= cw[1]; }
      if (st == 12) { cv[2] *= cw[2]; cv[3] *= cw[3]; }
    } else if (AFF) {                          // ReLU / plain affine: padding by select (NaN stays NaN: relu_keep_nan)
      if (st == 3 && RL) { cv[0] = relu_keep_nan(cv[0]); }
      if (st == 4 && RL) { cv[1] = relu_keep_nan(cv[1]); }
      if (st == 5 && RL) { cv[2] = relu_keep_nan(cv[2]); }
      if (st == 6 && RL) { cv[3] = relu_keep_nan(cv[3]); }
      if (st == 7 && !RL) { cv[0] *= rs; cv[1] *= rs; }
      if (st == 8 && !RL) { cv[2] *= rs; cv[3] *= rs; }
      if (st == 11) { cv[0] = padf != 0.f ? 0.f : cv[0]; cv[1] = padf != 0.f ? 0.f : cv[1]; }
      if (st == 12) { cv[2] = padf != 0.f ? 0.f : cv[2]; cv[3] = padf != 0.f ? 0.f : cv[3]; }
    }
    if (st == 13) {
      chi[0] = __builtin_convertvector(f32x2{cv[0], cv[1]}, f16x2);
      chi[1] = __builtin_convertvector(f32x2{cv[2], cv[3]}, f16x2);
    }
    if (st == 14) { cw[0] = __builtin_fmaf((float)chi[0].x, -1.0f, cv[0]); cw[1] = __builtin_fmaf((float)chi[0].y, -1.0f, cv[1]); }      // v_fma_mix_f32: v - float(hi)
    if (st == 15) { cw[2] = __builtin_fmaf((float)chi[1].x, -1.0f, cv[2]); cw[3] = __builtin_fmaf((float)chi[1].y, -1.0f, cv[3]); }
    if (st == 16) {
      clo[0] = __builtin_convertvector(f32x2{cw[0], cw[1]}, f16x2);
      clo[1] = __builtin_convertvector(f32x2{cw[2], cw[3]}, f16x2);
      if (XS_ABL != 12 || cw[0] == 1.2345e-30f) *reinterpret_cast<uint2*>(lds_raw + wr_addr[k] + hb) = make_uint2(__builtin_bit_cast(unsigned, chi[0]), __builtin_bit_cast(unsigned, chi[1]));
    }
    if (st == 17 && (XS_ABL != 12 || cw[1] == 1.2345e-30f))
      *reinterpret_cast<uint2*>(lds_raw + (wr_addr[k] ^ 32) + hb) = make_uint2(__builtin_bit_cast(unsigned, clo[0]), __builtin_bit_cast(unsigned, clo[1]));
  };
  // weights of chunk C, taps 3 gq .. 3 gq + 2: L2 -> LDS, 12 pieces of 1 KB (piece q = 16 weight rows: tap q / 4, couts 16 (q % 4) + lane / 4),
  // three per wave: q = 12 gq + 3 wave + u -> row group (u - wave) & 3 = dma_voff[u]
  auto dma_group = [&](int gq) __attribute__((always_inline)) {
    const int w3 = __builtin_amdgcn_readfirstlane(wave) * 3;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = 12 * gq + w3 + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(lds_raw + XS_WOFF + q * 1024), 16, dma_voff[u],
                                               ((q >> 2) * p.Cin + chC * 16) * 4, 0, 0);
    }
  };
  // fragments of one tap: 0 a_lo0, 1 a_lo1, 2 b_hi0, 3 b_hi1, 4 a_hi0, 5 a_hi1, 6 b_lo0, 7 b_lo1
  auto frag_of = [&](int t, int f, int hb) __attribute__((always_inline)) -> f16x8 {
    const int kh = t / 3, kw = t - kh * 3;
    if (f == 0 || f == 1 || f == 4 || f == 5) {
      const int i = f & 1;
      const int a = (f < 4 ? (rd_a[kw] ^ 32) : rd_a[kw]) + hb + ((i + kh) * XS_HW) * 64;      // hb and the tap displacement are multiples of 64
      return *reinterpret_cast<const f16x8*>(lds_raw + a);
    }
    const int j = f & 1;
    const int o = (f >= 4 ? (rd_b ^ 32) : rd_b) + (t * 64 + j * 32) * 64;
    return *reinterpret_cast<const f16x8*>(lds_raw + o);
  };
  // MFMAs of the chunk in halo buffer hbM + everything that rides in their shadow (chunk C -> halo buffer hbC, request of chunk F)
  auto mma_step = [&](int hbM, int hbC) __attribute__((always_inline)) {
    constexpr int NCONV = 18 * HALO_IT;        // 108 conversion steps: one per MFMA gap
    f16x8 fr[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) fr[f] = frag_of(0, f, hbM);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int m = 0; m < 12; ++m) {
        const int term = m >> 2, i = (m >> 1) & 1, j = m & 1;
        const int ia = term == 0 ? i : 4 + i;
        const int ib = term == 2 ? 6 + j : 2 + j;
        if (XS_ABL == 1) {
          asm volatile("" ::"v"(fr[ia]), "v"(fr[ib]));
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[ia], fr[ib], acc[i][j], 0, 0, 0);
        }
        if (t < 8 && XS_ABL != 5) {            // re-read behind the last use: a_lo0 m=1, a_lo1 3, b_hi0 6, b_hi1 7, a_hi0 9, b_lo0 10, a_hi1 / b_lo1 11
          if (m == 1) fr[0] = frag_of(t + 1, 0, hbM);
          if (m == 3) fr[1] = frag_of(t + 1, 1, hbM);
          if (m == 6) fr[2] = frag_of(t + 1, 2, hbM);
          if (m == 7) fr[3] = frag_of(t + 1, 3, hbM);
          if (m == 9) fr[4] = frag_of(t + 1, 4, hbM);
          if (m == 10) fr[6] = frag_of(t + 1, 6, hbM);
          if (m == 11) {
            fr[5] = frag_of(t + 1, 5, hbM);
            fr[7] = frag_of(t + 1, 7, hbM);
          }
        }
        if (t * 12 + m < NCONV && XS_ABL != 2) conv_step(t * 12 + m, hbC);
        __builtin_amdgcn_sched_barrier(0);
      }
      // The weight stage is a ring of three tap groups: once every wave is past taps 0-2 (their fragment reads completed before the MFMAs
      // that used them), the next chunk's taps 0-2 go into the same rows; likewise taps 3-5.  Taps 6-8 follow after the step's last
      // barrier and are published by the next step's first ring barrier -- hence the counted vmcnt in front of it: the vector-memory
      // instructions younger than those pieces are at least this step's first two raw-piece requests (gaps 1 and 19).
      if (t == 2) {
        // (behind an epilogue also its 16 output stores are younger: gfx9 counts loads and stores in one in-order counter, and
        // waiting for the stores' acknowledgements here parked every wave of the block for microseconds once per item)
        if (after_epi) {
          asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        if (XS_ABL != 6) asm volatile("s_barrier" ::: "memory");
        if (okC && XS_ABL != 4) dma_group(0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t == 5) {
        if (XS_ABL != 6) asm volatile("s_barrier" ::: "memory");
        if (okC && XS_ABL != 4) dma_group(1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- epilogue of item itM: 16 pixel rows at a time through the halo buffer the item's last chunk just released
  auto epilogue_t = [&](int hb, auto res_c) __attribute__((always_inline)) {
    constexpr bool HAS_RES = decltype(res_c)::value;
    float* et = reinterpret_cast<float*>(lds_raw + hb) + wave * 16 * XS_EP;
    const float asc = p.acc_scale * in_invM;
    const int c4 = (lane & 15) * 4, prow = lane >> 4;
    const int co = itM.n0 + c4;
    const bool cok = co < p.Cout;
    const int hw_o = p.Ho * p.Wo;
    const int pix_b = (itM.oy0 + 2 * wave) * p.Wo + itM.ox0 + prow;
    const __amdgpu_buffer_rsrc_t out_rsrc = make_rsrc(p.out + (long)itM.n * hw_o * p.out_ld, hw_o * p.out_ld * 4);
    const int v_out = cok ? (pix_b * p.out_ld + co) * 4 : -16;
    __amdgpu_buffer_rsrc_t res_rsrc = out_rsrc;
    int v_res = -16;
    if (HAS_RES) {
      res_rsrc = make_rsrc(p.res + (long)itM.n * hw_o * p.res_ld, hw_o * p.res_ld * 4);
      v_res = cok ? (pix_b * p.res_ld + co) * 4 : -16;
    }
    const float4 bias4 = biasM;
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
    float amx = 0.f;
    auto dpix_of = [&](int q16) __attribute__((always_inline)) { return (q16 >> 3) * p.Wo + (q16 & 7) * 4; };      // pixel group q16 of the wave: row q16 / 8, columns 4 (q16 % 8) ..
    // all 16 residual rows before the first store (a load behind a store waits for the store's acknowledgement: vmcnt is shared; in
    // place a thread reads exactly what it later writes) -- the fragment and conversion registers of the MFMA loop are free here
    u32x4 rpre[HAS_RES ? 16 : 1];
    if (HAS_RES) {
#pragma unroll
      for (int q16 = 0; q16 < 16; ++q16) rpre[q16] = __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, v_res, dpix_of(q16) * p.res_ld * 4, KEEP_LD_AUX_RES);
    }
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {           // rows 16 rd .. 16 rd + 15 of the wave's 64 pixels: accumulator tile rd / 2, registers 8 (rd % 2) ..
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int r = (rd & 1) * 8 + r8;
          if (XS_ABL != 8 || acc[rd >> 1][j][r] == 1.2345e-30f) et[((r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi) * XS_EP + j * 32 + l31] = acc[rd >> 1][j][r];
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): wave-local hand-off
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(et + (u * 4 + prow) * XS_EP + c4);
        float e[4] = {__builtin_fmaf(v.x, asc, bias4.x), __builtin_fmaf(v.y, asc, bias4.y), __builtin_fmaf(v.z, asc, bias4.z),
                      __builtin_fmaf(v.w, asc, bias4.w)};
        if (p.epi_act != KEEP_ACT_NONE) {       // (uniform; the activation comes before the residual: keep_conv_common.h epilogue_one)
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = p.fast ? act_apply_fast(e[q], p.epi_act) : act_apply(e[q], p.epi_act);
        }
        if (HAS_RES) {
          const u32x4 r4 = rpre[HAS_RES ? rd * 4 + u : 0];
          e[0] += __uint_as_float(r4.x); e[1] += __uint_as_float(r4.y);
          e[2] += __uint_as_float(r4.z); e[3] += __uint_as_float(r4.w);
        }
        u32x4 o;
        o.x = __float_as_uint(e[0]); o.y = __float_as_uint(e[1]); o.z = __float_as_uint(e[2]); o.w = __float_as_uint(e[3]);
        if (XS_ABL != 7 || e[0] == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b128(o, out_rsrc, v_out, dpix_of(rd * 4 + u) * p.out_ld * 4, KEEP_ST_AUX_XS);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s4[q] += e[q];
          ss4[q] = __builtin_fmaf(e[q], e[q], ss4[q]);
          amx = fmaxf(amx, fabsf(e[q]));
        }
      }
    }
    if (p.stats && XS_ABL != 9) {              // one partial per 256-pixel tile: the four waves' sums meet in LDS, added in wave order
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] = xs_xor32_sum(xs_xor16_sum(s4[q]));
        ss4[q] = xs_xor32_sum(xs_xor16_sum(ss4[q]));
      }
      if (lane < 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          et[(c4 + q) * 2 + 0] = s4[q];
          et[(c4 + q) * 2 + 1] = ss4[q];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (not __syncthreads: its fence would wait for the stores above)
      if (wave == 0) {
        const float* e0 = reinterpret_cast<const float*>(lds_raw + hb);
        float a = 0.f, b2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += e0[w * 16 * XS_EP + lane * 2 + 0];
          b2 += e0[w * 16 * XS_EP + lane * 2 + 1];
        }
        if (itM.n0 + lane < p.Cout) {
          float* dst = p.stats + (((long)itM.n * p.stats_P + itM.ty * tiles_x + itM.tx) * p.Cout + itM.n0 + lane) * 2;
          dst[0] = a;
          dst[1] = b2;
        }
      }
    }
    return amx;
  };

  // ---- prologue: chunk 0 of the block's first item converted in the open, chunk 1 requested
  setup_F();
#pragma unroll
  for (int k = 0; k < HALO_IT; ++k) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, h_voff[k], 0, KEEP_LD_AUX_XS);
    hreg[k] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  }
  if (AFF) {
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(sc_rsrc, sc_voff, 0, 0);
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(sh_rsrc, sc_voff, 0, 0);
    sc4 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    sh4 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
  }
  itC = itF;
  chC = 0;
  okC = true;
  cross_C();
  advance_F();                                 // F: chunk 1 of the same item (nch >= 2)
  dma_group(0);
  dma_group(1);
  dma_group(2);
#pragma unroll
  for (int k = 0; k < HALO_IT; ++k) {
#pragma unroll
    for (int st = 0; st < 18; ++st) conv_step(k * 18 + st, 0);
  }
  itM = itC;
  chM = 0;
  in_invM = in_invC;
  chC = 1;                                     // C: chunk 1 (same item)
  advance_F();                                 // F: chunk 2, or the next item's chunk 0
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int amax_n = -1;                             // image of the max |out| this wave has committed or seen
  float amax_run = 0.f;
  const unsigned long long cyc0 = TL ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rtc0 = TL ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if (TL) t0 = cyc0;
  // one pipeline step; HB: byte offset of the halo buffer holding chunk M (compile time: folded into the DS offsets).  true: done
  auto step = [&](auto hb_c) __attribute__((always_inline)) -> bool {
    constexpr int HB = decltype(hb_c)::value;
    const bool last = chM == nch - 1;          // the item's last chunk: its epilogue follows
    if (last) {                                // the bias of the item's cout block lands under the MFMAs
      const int co = itM.n0 + (lane & 15) * 4;
      biasM = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && co < p.Cout) biasM = *reinterpret_cast<const float4*>(p.bias + co);
    }
    mma_step(HB, XS_HBUF - HB);
    XS_T(0)
    // this wave's weight pieces (taps 0-5 of the next chunk) and converted rows are in LDS.  Younger than the last of those pieces are
    // the raw-piece requests of gaps 73 and 91 (and the two affine loads of gap 91): they may stay in flight -- vmcnt(0) here exposed
    // most of an HBM round trip per step
    if (AFF) {
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");              // every wave is done with halo buffer HB and weight taps 6-8; the other buffer and the next taps 0-5 are complete
    XS_T(1)
    if (okC && XS_ABL != 4) dma_group(2);
    XS_T(2)
    if (last) {
      const float amx = XS_ABL == 3 ? acc[0][0][0] + acc[1][1][5] : (p.res ? epilogue_t(HB, std::true_type{}) : epilogue_t(HB, std::false_type{}));
      if (p.out_amax) {     // max|out| of the image: a wave goes to memory only above everything it has committed or seen for this image
        if (itM.n != amax_n) {
          amax_n = itM.n;
          amax_run = 0.f;
        }
        if (__builtin_amdgcn_ballot_w64(amx > amax_run) != 0ull) {
          unsigned b = __float_as_uint(amx);
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
          unsigned* dst = p.out_amax + itM.n;
          unsigned seen = b;
          if (lane == 0) {
            seen = *reinterpret_cast<volatile unsigned*>(dst);
            if (b > seen) atomicMax(dst, b);
          }
          seen = max(b, (unsigned)__builtin_amdgcn_readfirstlane((int)seen));
          amax_run = fmaxf(amax_run, __uint_as_float(seen));
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      XS_T(3)
    }
    after_epi = last;
    if (!okC) return true;
    // M <- C
    if (chC == 0) {
      itM = itC;
      in_invM = in_invC;
    }
    chM = chC;
    // C <- F (the padding mask / weight rows of a new item come from F's registers: F is still inside that item, nch >= 2)
    okC = okF;
    if (okF) {
      if (chF == 0) {
        itC = itF;
        cross_C();
      }
      chC = chF;
      advance_F();                             // F <- the chunk after it
    }
    XS_T(4)
    if (last) {                                // the parked rows / statistics of the epilogue live in buffer HB: the next step converts into it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
    XS_T(6)
    return false;
  };
  while (true) {
    if (step(std::integral_constant<int, 0>{})) break;
    if (step(std::integral_constant<int, XS_HBUF>{})) break;
  }
  if (TL && tid == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.ws);
#pragma unroll
    for (int q = 0; q < 12; ++q) atomicAdd(dst + q, tacc[q]);
    atomicAdd(dst + 12, 1ull);
    atomicAdd(dst + 13, __builtin_amdgcn_s_memtime() - cyc0);
    atomicAdd(dst + 14, __builtin_amdgcn_s_memrealtime() - rtc0);
    if (blockIdx.x < 1024) {                   // per-block start / end (100 MHz ticks) behind the 16 sums
      dst[16 + blockIdx.x * 2] = rtc0;
      dst[17 + blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime();
    }
  }
#undef XS_T
}

// ------------------------------------------------------------------------------------------------ statistics of a written tile
// GroupNorm partials of an output tensor another kernel wrote (keep_conv_x3p.hip: the 64-pixel blocks of one clip in flight), reduced in
// EXACTLY the order of the epilogue above -- a block per (8 x 32 tile, 64 couts), lane (prow, c4) walks the wave's 16 pixel groups in
// order, the two lane-swap sums, the four waves in order -- so that the partials are the bits conv3x3_halo_x3s_kernel would have written.
__global__ __launch_bounds__(256) void conv_stats_replica_kernel(ConvP p, int tiles_x, int tiles_y, int ncb) {
  __shared__ float red[4 * 64 * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cb = blockIdx.x % ncb;
  int t = blockIdx.x / ncb;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y, n = t / tiles_y;
  const int c4 = (lane & 15) * 4, prow = lane >> 4, n0 = cb * 64;
  const int co = n0 + c4;
  const int hw_o = p.Ho * p.Wo;
  const int pix_b = (ty * 8 + 2 * wave) * p.Wo + tx * 32 + prow;
  const float* src = p.out + ((long)n * hw_o + pix_b) * p.out_ld + co;
  float s4[4] = {0.f, 0.f, 0.f, 0.f}, ss4[4] = {0.f, 0.f, 0.f, 0.f};
  if (co < p.Cout) {
#pragma unroll
    for (int g16 = 0; g16 < 16; ++g16) {
      const float4 v = *reinterpret_cast<const float4*>(src + (long)((g16 >> 3) * p.Wo + (g16 & 7) * 4) * p.out_ld);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s4[q] += e[q];
        ss4[q] = __builtin_fmaf(e[q], e[q], ss4[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    s4[q] = xs_xor32_sum(xs_xor16_sum(s4[q]));
    ss4[q] = xs_xor32_sum(xs_xor16_sum(ss4[q]));
  }
  if (lane < 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[(wave * 64 + c4 + q) * 2 + 0] = s4[q];
      red[(wave * 64 + c4 + q) * 2 + 1] = ss4[q];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float a = 0.f, b2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[(w * 64 + lane) * 2 + 0];
      b2 += red[(w * 64 + lane) * 2 + 1];
    }
    if (n0 + lane < p.Cout) {
      float* dst = p.stats + (((long)n * p.stats_P + ty * tiles_x + tx) * p.Cout + n0 + lane) * 2;
      dst[0] = a;
      dst[1] = b2;
    }
  }
}

void keep_conv_stats_replica(const ConvP& p, int n_img, hipStream_t st) {
  const int tiles_x = p.Wo / 32, tiles_y = p.Ho / 8, ncb = (p.Cout + 63) / 64;
  hipLaunchKernelGGL(conv_stats_replica_kernel, dim3(n_img * tiles_x * tiles_y * ncb), dim3(256), 0, st, p, tiles_x, tiles_y, ncb);
}

// ------------------------------------------------------------------------------------------------ dispatch
// Geometry of the streaming kernel inside what keep_conv_x3_halo_ok() already admits: 8x32 tiles, whole-item K ranges of at least two
// chunks, no split-K / aux tensor (bias / activation / residual / statistics / max|out| epilogue), the fast activation forms.
// (Round 4 also tried drawing the items from a global ticket counter instead of the static stride -- block lifetimes of one launch
// spread by +-20 %, bimodal -- with the ticket prefetched one item ahead: bit-identical results, 1.5 % (64 ch @512^2) to 5 % (128 ch
// @256^2) SLOWER: the launch is bound by chip-wide throughput, early finishers hand their share to the rest; removed.)  Everything else stays on
// conv3x3_halo_x3_kernel.  flags & KEEP_CONV_NO_STREAM: round 3's kernel everywhere (kernel-vs-kernel tests, A/B runs).
bool keep_conv_x3_stream_ok(const keep_conv2d_args* a, const ConvP& p, int split_k) {
  const bool off = (a->flags & KEEP_CONV_NO_STREAM) != 0;
  const bool simple = split_k == 1 && !a->aux;      // (an epilogue activation is one uniform branch per row here: ParseNet's LeakyReLU)
  const bool aff = a->pro_scale != nullptr;
  return !off && simple && a->Ho % 8 == 0 && a->Wo % 32 == 0 && a->Cin >= 32 && a->upsample != KEEP_UPSAMPLE_X2_PHASES &&
         (a->pro_act == KEEP_PRO_NONE || (aff && (a->pro_act == KEEP_PRO_RELU || (a->pro_act == KEEP_PRO_SWISH && p.fast)))) &&
         (long)a->N * a->Cin * 4 < (1L << 31);
}

int keep_conv2d_x3_stream(const keep_conv2d_args* a, ConvP& p, int n_cu, hipStream_t st) {
  const int tiles_x = a->Wo / 32, tiles_y = a->Ho / 8, ncb = (a->Cout + 63) / 64;
  const int n_items = a->N * tiles_x * tiles_y * ncb;
  dim3 grid(n_items < 2 * n_cu ? n_items : 2 * n_cu), block(256);
  const bool aff = a->pro_scale != nullptr;
  if (KEEP_DEV_ENV("KEEP_X3_OCC")) {      // dev: resident blocks per CU as the runtime sees them
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv3x3_halo_x3s_kernel<KEEP_PRO_SWISH, true>, 256, 0);
    fprintf(stderr, "[x3s] occupancy: %d blocks per CU (LDS %d B per block)\n", nb, XS_LDS);
  }
#ifdef KEEP_X3_ABLATE
  if (KEEP_DEV_ENV("KEEP_X3_EXP") && atoi(KEEP_DEV_ENV("KEEP_X3_EXP")) == 21 && a->pro_act == KEEP_PRO_SWISH) {      // timeline of wave 0
    static unsigned long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc(&dbg, 128 + 1024 * 16);
    (void)hipMemsetAsync(dbg, 0, 128, st);
    ConvP q = p;
    q.ws = reinterpret_cast<float*>(dbg);
    hipLaunchKernelGGL((conv3x3_halo_x3s_kernel<KEEP_PRO_SWISH, true, true>), grid, block, 0, st, q, tiles_x, tiles_y, ncb, n_items);
    unsigned long long h[16];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, dbg, 128, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int i = 0; i < 12; ++i) tot += (double)h[i];
    fprintf(stderr, "[x3s timeline] abl %d clock %.0f MHz | blocks %.0f cycles/block %.0f (whole %.0f) | mma+conv %.1f%%  sync %.1f%% | dma issue %.1f%%  epilogue %.1f%%  advance %.1f%%  dma wait %.1f%%  sync %.1f%%\n",
            XS_ABL, (double)h[13] / ((double)h[14] / 100.0), (double)h[12], tot / h[12], (double)h[13] / h[12], 100.0 * h[0] / tot, 100.0 * h[1] / tot, 100.0 * h[2] / tot, 100.0 * h[3] / tot, 100.0 * h[4] / tot,
            100.0 * h[5] / tot, 100.0 * h[6] / tot);
    {      // block lifetimes of the launch (100 MHz ticks): spread and histogram
      static unsigned long long se[2048];
      const int nblk = (int)grid.x < 1024 ? (int)grid.x : 1024;
      (void)hipMemcpy(se, dbg + 16, nblk * 16, hipMemcpyDeviceToHost);
      unsigned long long t_min = ~0ull, t_max = 0;
      for (int b = 0; b < nblk; ++b) {
        if (se[2 * b] < t_min) t_min = se[2 * b];
        if (se[2 * b + 1] > t_max) t_max = se[2 * b + 1];
      }
      double lsum = 0, lmin = 1e30, lmax = 0;
      for (int b = 0; b < nblk; ++b) {
        const double l = (se[2 * b + 1] - se[2 * b]) / 100.0;
        lsum += l;
        if (l < lmin) lmin = l;
        if (l > lmax) lmax = l;
      }
      int hist[10] = {0};
      for (int b = 0; b < nblk; ++b) hist[(int)(9.999 * ((se[2 * b + 1] - se[2 * b]) / 100.0 - lmin) / (lmax - lmin + 1e-9))]++;
      fprintf(stderr, "[x3s blocks] span %.1f us | block life min %.1f mean %.1f max %.1f us | histogram (min..max, 10 bins):", (t_max - t_min) / 100.0, lmin,
              lsum / nblk, lmax);
      for (int i = 0; i < 10; ++i) fprintf(stderr, " %d", hist[i]);
      fprintf(stderr, "\n");
    }
    return KEEP_OK;
  }
#endif
  if (a->pro_act == KEEP_PRO_SWISH)
    hipLaunchKernelGGL((conv3x3_halo_x3s_kernel<KEEP_PRO_SWISH, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
  else if (a->pro_act == KEEP_PRO_RELU)
    hipLaunchKernelGGL((conv3x3_halo_x3s_kernel<KEEP_PRO_RELU, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
  else if (aff)
    hipLaunchKernelGGL((conv3x3_halo_x3s_kernel<KEEP_PRO_NONE, true>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
  else
    hipLaunchKernelGGL((conv3x3_halo_x3s_kernel<KEEP_PRO_NONE, false>), grid, block, 0, st, p, tiles_x, tiles_y, ncb, n_items);
  KEEP_LAUNCH_CHECK("keep_conv2d(halo x3, streaming)");
  return KEEP_OK;
}
