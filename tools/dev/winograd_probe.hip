// dev / evidence, NOT product ("build a microkernel for 64 -> 64 @512^2 only if a variant passes the existing assertions
// unchanged" -- oracle/numerics_gate.py: Winograd F(2x2, 3x3) in fp32 is a go).  A stride-1 3x3 convolution, zero padding 1, NHWC fp32 in / out,
// GroupNorm-affine + swish prologue (the dominant kernel's instantiation), bias epilogue, under the x3 precision policy -- every product of the
// 16 element-wise GEMMs is a_hi b_hi + a_hi b_lo + a_lo b_hi on v_mfma_f32_32x32x16_f16, fp32 accumulation -- but on Winograd-transformed operands:
// 16 products per 2 x 2 outputs and input channel instead of 36 (Lavin & Gray 2016; the transform matrices of oracle/numerics_gate.py).
//
//   block = 256 threads, PERSISTENT: grid (G, cout blocks), block (bx, cb) walks the 16 x 16 output tiles bx, bx + G, ... (8 x 8 = 64 Winograd tiles x 64 output
//     channels each; 16 input channels per chunk); the pipeline runs across tiles: the last chunk of a tile carries the activation + transform of the next tile's first
//   halo (18 x 18 pixels x 16 ch, fp32) -> registers -> affine + swish -> LDS `raw` (96-byte pixel stride: the 4 x 4 window reads are conflict-free)
//   thread (tile, 4 channels): V = B^T d B in fp32 (32 packed adds), split hi = f16(V), lo = f16(V - hi), LDS `vbuf` [position 16][tile 64][64-byte row:
//     hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15, 16-byte slots XOR-swizzled by (tile >> 2) & 3] -- two stages
//   U = G g G^T is transformed, scaled and split ON THE HOST once per layer and laid out in MFMA B-fragment order: a wave's fragment is one coalesced
//     1 KB read L2 -> registers, no LDS (tools/dev/winograd_probe.py:pack_weights)
//   wave l owns the four positions (i, l), i = 0..3, of all 64 tiles x 64 couts: 16 accumulators of 32 x 32 (256 accumulation registers), 48 MFMAs per chunk
//   epilogue: column transform t = A^T m in registers (the wave holds a whole column), t through LDS, wave (a, b) forms Y[a][b] = (t A)[a][b] of every
//     tile, * acc_scale + bias, stores pixel (2 ty + a, 2 tx + b) -- 128 contiguous bytes per lane group
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define WG_HW 18
#define WG_HPIX (WG_HW * WG_HW)
#define WG_RAWP 24      // floats per halo pixel in LDS (16 used)
#ifndef WG_TR4
#define WG_TR4 0        // transform: 0 = two passes over two channels per thread; 1 = one pass over four (spills at 256 accumulation + 256 other registers: slower)
#endif
#ifndef WG_MLATE
#define WG_MLATE 12     // first of the 24 activation groups (first half of a chunk) that carries MFMAs
#endif
#ifndef WG_BLATE
#define WG_BLATE (WG_TR4 ? 2 : 6)      // first of the transform groups (second half) that carries MFMAs (same-box sweep: 2 / 4 / 6 / 7 -> 584 / ~560 / 517 / 540 us at 8 x 512^2 x 64)
#endif
#ifndef WG_TL
#define WG_TL 0      // 1: phase timeline of wave 0 (s_memtime) into WgArgs::dbg
#endif
#define WG_T(q)                                            \
  if (WG_TL) {                                             \
    const unsigned long long tn = __builtin_amdgcn_s_memtime(); \
    tacc[q] += tn - tprev;                                 \
    tprev = tn;                                            \
  }
#ifndef WG_SCHED
#define WG_SCHED 2      // 2: MFMAs and conversion arithmetic interleaved by hand (one step per MFMA gap, pinned by sched_barriers); 1: sched_group_barrier hints; 0: the compiler's order
#endif

struct WgArgs {
  const float* x;
  const u32x4* u;
  const float* bias;
  const float* scale;
  const float* shift;
  float* y;
  int N, H, W, Cin, Cout;
  float acc_scale;
  unsigned long long* dbg;      // WG_TL builds: cycles of wave 0 per phase, summed over blocks
};

static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int NCH, bool SWISH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void wg_f22_x3_kernel(WgArgs p) {
  static_assert(NCH % 2 == 0, "chunk c lives in stage c & 1: a tile's first chunk is always stage 0");
  __shared__ __attribute__((aligned(1024))) unsigned char vbuf[2 * 65536];
  __shared__ __attribute__((aligned(16))) float raw[(WG_HPIX + 1) * WG_RAWP];      // (+ 1: the pixel threads without a sixth item park it on)
#define WG_V(i) (vbuf + (i) * 65536)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lhi = lane >> 5;
  const int tiles_x = p.W >> 4, tiles_y = p.H >> 4;
  const int npt = p.N * tiles_y * tiles_x, G = gridDim.x, cb = blockIdx.y;
  const int g = tid & 3;
  int pt = blockIdx.x;
  if (pt >= npt) return;

  // halo item j of this thread: pixel (hy, hx) of the 18 x 18 halo, its four channels g * 4 ..; float offset in `raw`
  struct Geo {
    int n, oy0, ox0;      // (uniform)
  };
  unsigned hoff[6];      // byte offset of item j in the whole input tensor for the tile whose chunks are being FETCHED (0xfffffff0: zero padding / no item -- the
                         // buffer load returns zeros); recomputed when the fetches cross into the next tile
  auto geo = [&](int t, Geo& q) {
    const bool ok = t < npt;
    const int tx = t % tiles_x, r = t / tiles_x;
    const int ty = r % tiles_y;
    q.n = ok ? r / tiles_y : 0;
    q.oy0 = ty * 16;
    q.ox0 = tx * 16;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int hp = (tid >> 2) + 64 * j, hy = hp / WG_HW, hx = hp - hy * WG_HW;
      const int iy = q.oy0 - 1 + hy, ix = q.ox0 - 1 + hx;
      hoff[j] = (ok && hp < WG_HPIX && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? (unsigned)((((q.n * p.H + iy) * p.W + ix) * p.Cin + g * 4) * 4) : 0xfffffff0u;
    }
  };
  auto hlds = [&](int j) { return (j < 5 || tid < 16 ? ((tid >> 2) + 64 * j) : WG_HPIX) * WG_RAWP + g * 4; };      // float offset of item j in `raw`
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long)p.N * p.H * p.W * p.Cin * 4), 0x00020000);
  f32x4 hreg[6];
  f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
  unsigned hpad = 0;      // bit j: item j of the chunk in `hreg` is zero padding
  auto fetch_halo = [&](const Geo& q, int c) __attribute__((always_inline)) {
    const int c0 = c << 4;
    hpad = 0;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (j < 5 || wave == 0) hreg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)hoff[j], c0 * 4, 0));
      hpad |= (hoff[j] == 0xfffffff0u ? 1u : 0u) << j;
    }
    sc4 = ld4(p.scale + (long)q.n * p.Cin + c0 + g * 4);
    sh4 = ld4(p.shift + (long)q.n * p.Cin + c0 + g * 4);
  };
  unsigned apad = 0;      // the padding bits of the chunk being activated (latched before the next fetch overwrites hpad)
  auto activate_item = [&](int j) __attribute__((always_inline)) {
    f32x4 v = hreg[j];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf(v[q], sc4[q], sh4[q]);
    if (SWISH) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float e = __builtin_amdgcn_exp2f(v[q] * -1.4426950408889634f);
        e += 1.0f;
        v[q] *= __builtin_amdgcn_rcpf(e);
      }
    }
    if ((apad >> j) & 1) v = f32x4{0.f, 0.f, 0.f, 0.f};      // zero padding applies to the activated tensor
    *reinterpret_cast<f32x4*>(&raw[hlds(j)]) = v;
  };

  // weight fragments of one chunk: position i of this wave's column, cout half ch, hi / lo
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.u), 0, (p.Cout >> 6) * NCH * 16 * 2 * 2 * 1024, 0x00020000);
  u32x4 bf[2][2][2];      // positions i and i + 1 resident (slot i & 1); position i + 2 is requested when the MFMAs of position i have been issued
  auto fetch_b = [&](int c, int i) __attribute__((always_inline)) {
    const int pidx = i * 4 + wave;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl)      // one VGPR of address (lane * 16), the fragment's 1 KB block as a scalar offset: no per-fragment pointer lives in registers
        bf[i & 1][ch][hl] = __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, lane * 16, ((((cb * NCH + c) * 16 + pidx) * 2 + ch) * 2 + hl) * 1024, 0);
  };

  const int tile = tid >> 2, ty8 = tile >> 3, tx8 = tile & 7;
  const int wbase = ((2 * ty8) * WG_HW + 2 * tx8) * WG_RAWP + g * 4;
  const int vrow = tile * 64 + ((((g >> 1) ^ ((tile >> 2) & 3))) * 16) + (g & 1) * 8;      // byte offset of this thread's 4 hi halves inside a position
  const int arow = l31 * 64 + ((lhi ^ ((l31 >> 2) & 3)) * 16);                            // byte offset of this lane's hi fragment inside a (position, tile half)

  f32x16 acc[4][2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int th = 0; th < 2; ++th)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][th][ch][r] = 0.f;
  };

  // ---- the steps that ride in the MFMA gaps: activation (first half of a chunk) and transform (second half) of the NEXT chunk
  f32x4 av[5], ae[5];
  auto act_step = [&](int j, int st) __attribute__((always_inline)) {
    if (st == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) av[j][q] = __builtin_fmaf(hreg[j][q], sc4[q], sh4[q]);
    } else if (st == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ae[j][q] = av[j][q] * -1.4426950408889634f;
    } else if (st == 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ae[j][q] = __builtin_amdgcn_exp2f(ae[j][q]);
    } else if (st == 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ae[j][q] += 1.0f;
    } else if (st == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) ae[j][q] = __builtin_amdgcn_rcpf(ae[j][q]);
    } else if (st == 5) {
#pragma unroll
      for (int q = 0; q < 4; ++q) av[j][q] = SWISH ? av[j][q] * ae[j][q] : av[j][q];
      if ((apad >> j) & 1) av[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      *reinterpret_cast<f32x4*>(&raw[hlds(j)]) = av[j];
    }
  };
  f16x8 fah[2][2], fal[2][2];      // (positions i and i + 1: index i & 1)
  auto load_a = [&](int cur, int i) __attribute__((always_inline)) {
    const unsigned char* vb = WG_V(cur) + (i * 4 + wave) * 4096;
#pragma unroll
    for (int th = 0; th < 2; ++th) {
      fah[i & 1][th] = *reinterpret_cast<const f16x8*>(vb + th * 2048 + arow);
      fal[i & 1][th] = *reinterpret_cast<const f16x8*>(vb + th * 2048 + (arow ^ 32));
    }
  };
  auto mfma_one = [&](int i, int kk) __attribute__((always_inline)) {      // kk = term * 4 + th * 2 + ch: four independent accumulators between dependent issues
    const int term = kk >> 2, th = (kk >> 1) & 1, ch = kk & 1;
    const f16x8 bh = __builtin_bit_cast(f16x8, bf[i & 1][ch][0]), bl = __builtin_bit_cast(f16x8, bf[i & 1][ch][1]);
    if (term == 0) acc[i][th][ch] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[i & 1][th], bh, acc[i][th][ch], 0, 0, 0);
    else if (term == 1) acc[i][th][ch] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i & 1][th], bh, acc[i][th][ch], 0, 0, 0);
    else acc[i][th][ch] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[i & 1][th], bl, acc[i][th][ch], 0, 0, 0);
  };
#if WG_TR4
  // one pass, four channels per thread (twice the live registers of the two-pass form, half its instructions)
  f32x4 dwin[4][2], wrow[4][4];      // (window column cc in dwin[.][cc & 1]: one column of lookahead)
  auto load_dcol = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dwin[rr][cc & 1] = ld4(&raw[wbase + (rr * WG_HW + cc) * WG_RAWP]);
  };
  auto load_d = [&](int) __attribute__((always_inline)) {
    load_dcol(0);
    load_dcol(1);
  };
  const int vrow_lo = vrow ^ 32;
  auto tr_step = [&](int nb, int t) __attribute__((always_inline)) {      // t = 0 .. 19
    if (t < 4) {      // B^T d: rows of the window
      wrow[0][t] = dwin[0][t & 1] - dwin[2][t & 1];
      wrow[1][t] = dwin[1][t & 1] + dwin[2][t & 1];
      wrow[2][t] = dwin[2][t & 1] - dwin[1][t & 1];
      wrow[3][t] = dwin[1][t & 1] - dwin[3][t & 1];
      if (t + 2 < 4) load_dcol(t + 2);
    } else {          // (. B) for position (i, l), split, park
      const int i = (t - 4) >> 2, l = (t - 4) & 3;
      const f32x4 v = l == 0 ? wrow[i][0] - wrow[i][2] : (l == 1 ? wrow[i][1] + wrow[i][2] : (l == 2 ? wrow[i][2] - wrow[i][1] : wrow[i][1] - wrow[i][3]));
      const f16x2 h01 = __builtin_convertvector(f32x2{v[0], v[1]}, f16x2), h23 = __builtin_convertvector(f32x2{v[2], v[3]}, f16x2);
      const f32x2 r01 = f32x2{v[0], v[1]} - __builtin_convertvector(h01, f32x2), r23 = f32x2{v[2], v[3]} - __builtin_convertvector(h23, f32x2);
      const f16x2 q01 = __builtin_convertvector(r01, f16x2), q23 = __builtin_convertvector(r23, f16x2);
      unsigned char* dst = WG_V(nb) + (i * 4 + l) * 4096;
      *reinterpret_cast<f16x4*>(dst + vrow) = f16x4{h01.x, h01.y, h23.x, h23.y};
      *reinterpret_cast<f16x4*>(dst + vrow_lo) = f16x4{q01.x, q01.y, q23.x, q23.y};
    }
  };
#define WG_TSTEPS 20
#define WG_TGROUPS 5
#else
  // two passes of two channels each (h = 0, 1: channels 4 g + 2 h, + 1): half the live registers of a four-channel window
  f32x2 dwin[4][2], wrow[4][4];      // (window column cc in dwin[.][cc & 1]: one column of lookahead)
  auto load_dcol = [&](int h, int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) dwin[rr][cc & 1] = *reinterpret_cast<const f32x2*>(&raw[wbase + (rr * WG_HW + cc) * WG_RAWP + 2 * h]);
  };
  auto load_d = [&](int h) __attribute__((always_inline)) {
    load_dcol(h, 0);
    load_dcol(h, 1);
  };
  auto tr_step = [&](int nb, int tt) __attribute__((always_inline)) {      // tt = 0 .. 39: pass h = tt / 20, step t = tt % 20
    const int h = tt / 20, t = tt % 20;
    if (t < 4) {      // B^T d: rows of the window
      wrow[0][t] = dwin[0][t & 1] - dwin[2][t & 1];
      wrow[1][t] = dwin[1][t & 1] + dwin[2][t & 1];
      wrow[2][t] = dwin[2][t & 1] - dwin[1][t & 1];
      wrow[3][t] = dwin[1][t & 1] - dwin[3][t & 1];
      if (t + 2 < 4) load_dcol(h, t + 2);
      else if (h == 0) load_dcol(1, t - 2);      // (the second pass's first two columns behind the first pass's last two)
    } else {          // (. B) for position (i, l), split, park
      const int i = (t - 4) >> 2, l = (t - 4) & 3;
      const f32x2 v = l == 0 ? wrow[i][0] - wrow[i][2] : (l == 1 ? wrow[i][1] + wrow[i][2] : (l == 2 ? wrow[i][2] - wrow[i][1] : wrow[i][1] - wrow[i][3]));
      const f16x2 hi = __builtin_convertvector(v, f16x2);
      const f16x2 lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x2), f16x2);
      unsigned char* dst = WG_V(nb) + (i * 4 + l) * 4096;
      *reinterpret_cast<f16x2*>(dst + vrow + 4 * h) = hi;
      *reinterpret_cast<f16x2*>(dst + (vrow ^ 32) + 4 * h) = lo;
    }
  };

#define WG_TSTEPS 40
#define WG_TGROUPS 10
#endif
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = WG_TL ? __builtin_amdgcn_s_memtime() : 0ull;
  // ---- prologue: the first tile's chunk 0 into stage 0, its chunk 1 in flight
  Geo gc, gn, gf;      // the tile being computed, the next one, the one whose chunks are being fetched
  geo(pt, gc);
  gf = gc;
  fetch_halo(gf, 0);
  fetch_b(0, 0);
  fetch_b(0, 1);
  apad = hpad;
  if (wave == 0) activate_item(5);
#pragma unroll
  for (int j = 0; j < 5; ++j) activate_item(j);
  __syncthreads();
  fetch_halo(gf, 1);
  load_d(0);
#pragma unroll
  for (int t = 0; t < WG_TSTEPS; ++t) tr_step(0, t);
  __syncthreads();
  zero_acc();
  WG_T(0)

  while (true) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int cur = c & 1, nxt = cur ^ 1;
      // first half: positions 0, 1 of chunk c with the activation of the next chunk (raw is free: its transform ended before the last barrier)
      apad = hpad;
      if (wave == 0) activate_item(5);      // (16 threads' sixth item: its own small block)
      // 24 groups of activation steps; the MFMAs ride in groups WG_MLATE .. 23 -- the weight fragments of a position are requested when the position two
      // before it has been issued, and L2 / MALL answers in 1 000 - 2 500 cycles next to the streaming input: the later the MFMAs, the less they wait
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        if (k == WG_MLATE - 2) load_a(cur, 0);
        if (k == WG_MLATE + (24 - WG_MLATE) / 4) load_a(cur, 1);
        if (k >= WG_MLATE) {
#pragma unroll
          for (int m = ((k - WG_MLATE) * 24) / (24 - WG_MLATE); m < ((k - WG_MLATE + 1) * 24) / (24 - WG_MLATE); ++m) {
            mfma_one(m / 12, m % 12);
            if (m == 11) fetch_b(c, 2);
            if (m == 23) fetch_b(c, 3);
          }
        }
#pragma unroll
        for (int a = (k * 35) / 24; a < ((k + 1) * 35) / 24; ++a) act_step(a % 5, a / 5);
        __builtin_amdgcn_sched_barrier(0);
      }
      WG_T(1)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();      // raw holds the next chunk
      WG_T(2)
      // second half: positions 2, 3 with the transform of the next chunk into the other stage; the loads of the chunk after it
      if (c + 2 == NCH) {      // the fetches cross into the next tile of this block (beyond the last: every load out of range -- one wasted activation + transform)
        geo(pt + G, gn);
        gf = gn;
      }
      fetch_halo(gf, (c + 2) % NCH);
      load_d(0);
      __builtin_amdgcn_sched_barrier(0);
      // ten groups (per pass: the window rows, then one output row i = four positions each): the steps of a group are independent chains the
      // scheduler interleaves (one wave per SIMD: a lone cvt -> sub -> cvt chain would sit out its own latency), 2-3 MFMAs per group
#pragma unroll
      for (int gi = 0; gi < WG_TGROUPS; ++gi) {
        if (gi == WG_BLATE - 1) load_a(cur, 2);
        if (gi == WG_BLATE + (WG_TGROUPS - WG_BLATE) / 2 - 1) load_a(cur, 3);      // (one group ahead of position 3's first MFMA)
        if (gi >= WG_BLATE) {
#pragma unroll
          for (int k = ((gi - WG_BLATE) * 24) / (WG_TGROUPS - WG_BLATE); k < ((gi - WG_BLATE + 1) * 24) / (WG_TGROUPS - WG_BLATE); ++k) {
            mfma_one(2 + k / 12, k % 12);
            if (k == 11) fetch_b((c + 1) % NCH, 0);
            if (k == 23) fetch_b((c + 1) % NCH, 1);
          }
        }
#pragma unroll
        for (int t = gi * 4; t < gi * 4 + 4; ++t) tr_step(nxt, t);
        __builtin_amdgcn_sched_barrier(0);
        if (gi == 0) { WG_T(6) }
        if (gi == WG_TGROUPS / 2 - 1) { WG_T(7) }
      }
      WG_T(3)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();      // stage nxt holds the next chunk; every wave is past its reads of stage cur and of raw
      WG_T(4)
    }
    // ---- epilogue of the tile through stage 1 (stage 0 already holds the next tile's first chunk), one tile half per round.
    // Column transform in registers: t[a] = (A^T m)[a] for this wave's column l: t0 = m0 + m1 + m2, t1 = m1 - m2 - m3
    {
      float* ex = reinterpret_cast<float*>(WG_V(1));      // [wave l][a][ch][r4][lane][4] floats = 4 x 16 KB
      const int a_w = wave >> 1, bcol = wave & 1;
      float* yout = p.y + (((long)gc.n * p.H + gc.oy0 + a_w) * p.W + gc.ox0 + 8 * lhi + bcol) * p.Cout + cb * 64 + l31;
      const int row_st = 2 * p.W * p.Cout, col_st = 2 * p.Cout;      // one Winograd tile down / right
#pragma unroll
      for (int th = 0; th < 2; ++th) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              f32x4 t;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int r = r4 * 4 + q;
                t[q] = a == 0 ? (acc[0][th][ch][r] + acc[1][th][ch][r]) + acc[2][th][ch][r] : (acc[1][th][ch][r] - acc[2][th][ch][r]) - acc[3][th][ch][r];
              }
              *reinterpret_cast<f32x4*>(ex + (((((wave * 2 + a) * 2 + ch) * 4 + r4) * 64 + lane) * 4)) = t;
            }
        __syncthreads();
        auto ld_t = [&](int l, int ch, int r4) { return *reinterpret_cast<const f32x4*>(ex + (((((l * 2 + a_w) * 2 + ch) * 4 + r4) * 64 + lane) * 4)); };
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const float bias = p.bias ? p.bias[cb * 64 + ch * 32 + l31] : 0.f;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            f32x4 yv;
            if (bcol == 0) yv = (ld_t(0, ch, r4) + ld_t(1, ch, r4)) + ld_t(2, ch, r4);
            else yv = (ld_t(1, ch, r4) - ld_t(2, ch, r4)) - ld_t(3, ch, r4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = r4 * 4 + q;
              // row of the 32 x 32 accumulator = tile inside the half: m = (r & 3) + 8 (r >> 2) + 4 lhi -> tile row th * 4 + (r >> 2), tile column (r & 3) + 4 lhi
              yout[(th * 4 + (r >> 2)) * row_st + (r & 3) * col_st + ch * 32] = __builtin_fmaf(yv[q], p.acc_scale, bias);
            }
          }
        }
        __syncthreads();
      }
    }
    zero_acc();
    WG_T(5)
    pt += G;
    if (pt >= npt) break;
    gc = gn;
  }
  if (WG_TL) {
    if (tid == 0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) atomicAdd(p.dbg + q, tacc[q]);
      atomicAdd(p.dbg + 8, 1ull);
    }
  }
}

extern "C" int wg_probe_run(const float* x, const void* u, const float* bias, const float* scale, const float* shift, float* y, int N, int H, int W, int Cin,
                            int Cout, float acc_scale, int swish, void* stream, unsigned long long* dbg, int blocks) {
  if (H % 16 || W % 16 || Cout % 64 || (Cin != 64 && Cin != 128) || !scale || !shift || (long)N * H * W * Cin * 4 >= (1L << 32)) return 1;
  WgArgs a{x, reinterpret_cast<const u32x4*>(u), bias, scale, shift, y, N, H, W, Cin, Cout, acc_scale, dbg};
  if (WG_TL && !dbg) return 3;
  const int ncb = Cout / 64, npt = N * (H / 16) * (W / 16);
  int gx = (blocks > 0 ? blocks : 256) / ncb;      // one block per CU (160 KB of LDS): the cout blocks share the CUs
  if (gx > npt) gx = npt;
  const dim3 grid(gx, ncb), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 64) {
    if (swish) hipLaunchKernelGGL((wg_f22_x3_kernel<4, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wg_f22_x3_kernel<4, false>), grid, block, 0, st, a);
  } else {
    if (swish) hipLaunchKernelGGL((wg_f22_x3_kernel<8, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wg_f22_x3_kernel<8, false>), grid, block, 0, st, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
