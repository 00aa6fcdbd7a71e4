// keep_attention: fused softmax(scale * Q K^T + mask) V on the CDNA4 matrix cores (fp32 in / fp32 accumulate),
// flash style -- the L x L score matrix (64 MB per GMFlow pair in the reference) never leaves the CU.
//
// One wave owns 32 query rows.  Per 32-key tile:
//   S^T (keys x queries) = K . Q^T   with v_mfma_f32_32x32x2_f32, A = K tile, B = Q tile, both K-major in LDS
//   C/D layout: lane (q = lane&31, h = lane>>5) holds keys kr(r,h) = (r&3) + 8*(r>>2) + 4*h, r = 0..15, of ITS query
//   -> the softmax row reduction is 15 in-lane ops + one cross-half shuffle (no LDS, no 5-step butterfly);
//   P.V: the MFMA k index is free to be ANY key order as long as A and B agree, so step r uses k=0 <-> key kr(r,0)
//   and k=1 <-> key kr(r,1): the A operand of step r is simply the lane's own p[r] register -- P is never moved,
//   transposed or staged -- and B is V[kr(r,h)][dv = lane&31].
//   O (queries x dv) comes out with rows = queries spread over registers, so the per-query rescale factors are
//   fetched with 16 wave shuffles per tile.
// Token addressing modes (plain / sparse-causal keys / shifted windows with region mask) are pure index math at
// tile-load time, so the window partition, roll, key concatenation and batch swap of the reference cost no HBM pass.
#include <math.h>
#include <stdlib.h>

#include "keep_common.h"

// (dev A/B) cache policy of the x3 attention kernel's output rows: -DKEEP_NT_ATTN_O=1 = streaming stores
#ifndef KEEP_NT_ATTN_O
#define KEEP_NT_ATTN_O 0
#endif
#if KEEP_NT_ATTN_O
#define KEEP_ATTN_O_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define KEEP_ATTN_O_STORE(ptr, val) (*(ptr) = (val))
#endif


struct AttnP {
  const float* q;
  const float* k;
  const float* v;
  const unsigned short* q16;  // bf16 inputs (attn_bf16in_kernel)
  const unsigned short* k16;
  const unsigned short* v16;
  float* o;
  long q_bs, q_ts, q_hs, k_bs, k_ts, k_hs, v_bs, v_ts, v_hs, o_bs, o_ts, o_hs;
  int B, H, Lq, Lk, D, Dv;
  float scale;
  int mode, T, seg_len;
  int img_h, img_w, ksplit, shift, kv_rot, n_img;
  int nslices;  // dv slices per head
  unsigned flags;  // keep_attention_args.flags (KEEP_ATTN_NO_*)
#ifdef KEEP_X3_ABLATE
  int abl;               // dev builds: phase ablation selector (KEEP_ATTN_EXP)
#endif
  const _Float16* kv_pack;   // KEEP_MMA_X3: packed K / V^T tile images (attn_pack_kv_x3_kernel) or NULL
  const float* q_amax;   // KEEP_MMA_X3 range probes ([B] each) or NULL
  const float* k_amax;
  const float* v_amax;
};

// KEEP_MMA_X3 range scaling (same rule as keep_conv_common.h): power of two s with amax * s in [2^14, 2^15), and 1/s
__device__ __forceinline__ void attn_range_scale(float amax, float& s, float& inv_s) {
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

// window-mode: token t of window-batch bw -> (image, pixel index)
__device__ __forceinline__ void win_decode(const AttnP& p, int bw, int t, int rot, int& img, int& pix) {
  const int k2 = p.ksplit * p.ksplit;
  img = bw / k2;
  const int widx = bw - img * k2;
  const int wy = widx / p.ksplit, wx = widx - wy * p.ksplit;
  const int wh = p.img_h / p.ksplit, ww = p.img_w / p.ksplit;
  // t / ww without the ~40-instruction integer division (exact for t < 65536, enforced on the host): everything else in
  // this function is block-uniform and lands on the scalar unit
  const int ty = (int)(((float)t + 0.5f) * (1.0f / (float)ww)), tx = t - ty * ww;
  int y = wy * wh + ty + p.shift;
  int x = wx * ww + tx + p.shift;
  if (y >= p.img_h) y -= p.img_h;
  if (x >= p.img_w) x -= p.img_w;
  pix = y * p.img_w + x;
  if (rot) {
    img += rot;
    if (img >= p.n_img) img -= p.n_img;
  }
}

// region id of a window-local token in the ROLLED frame (GM/transformer.py:24-35)
__device__ __forceinline__ int win_region(const AttnP& p, int bw, int t) {
  const int k2 = p.ksplit * p.ksplit;
  const int widx = bw % k2;
  const int wy = widx / p.ksplit, wx = widx - wy * p.ksplit;
  const int wh = p.img_h / p.ksplit, ww = p.img_w / p.ksplit;
  const int ty = (int)(((float)t + 0.5f) * (1.0f / (float)ww)), tx = t - ty * ww;
  const int yr = wy * wh + ty, xr = wx * ww + tx;
  const int sh = wh / 2, sw = ww / 2;
  const int ih = yr < p.img_h - wh ? 0 : (yr < p.img_h - sh ? 1 : 2);
  const int iw = xr < p.img_w - ww ? 0 : (xr < p.img_w - sw ? 1 : 2);
  return ih * 3 + iw;
}

// Block-uniform part of the window index math (mode 2): every token of a block lives in one window of one image, so the
// divisions by ksplit happen once per block; per token only t / ww remains (exact float reciprocal for t < 65536).
struct WinCtx {
  int img, img_kv, y0, x0, wh, ww;
  float inv_ww;
};
__device__ __forceinline__ WinCtx win_ctx(const AttnP& p, int bw) {
  WinCtx c;
  const int k2 = p.ksplit * p.ksplit;
  c.img = bw / k2;
  const int widx = bw - c.img * k2;
  const int wy = widx / p.ksplit, wx = widx - wy * p.ksplit;
  c.wh = p.img_h / p.ksplit;
  c.ww = p.img_w / p.ksplit;
  c.y0 = wy * c.wh;
  c.x0 = wx * c.ww;
  c.img_kv = c.img + p.kv_rot;
  if (c.img_kv >= p.n_img) c.img_kv -= p.n_img;
  c.inv_ww = 1.0f / (float)c.ww;
  return c;
}
// token t of the window -> pixel index in the un-rolled image and region id in the rolled frame (GM/transformer.py:24-35)
__device__ __forceinline__ void win_token(const AttnP& p, const WinCtx& c, int t, int& pix, int& region) {
  const int ty = (int)(((float)t + 0.5f) * c.inv_ww);
  const int tx = t - ty * c.ww;
  const int yr = c.y0 + ty, xr = c.x0 + tx;
  int y = yr + p.shift, x = xr + p.shift;
  if (y >= p.img_h) y -= p.img_h;
  if (x >= p.img_w) x -= p.img_w;
  pix = y * p.img_w + x;
  const int sh = c.wh / 2, sw = c.ww / 2;
  const int ih = yr < p.img_h - c.wh ? 0 : (yr < p.img_h - sh ? 1 : 2);
  const int iw = xr < p.img_w - c.ww ? 0 : (xr < p.img_w - sw ? 1 : 2);
  region = ih * 3 + iw;
}

__device__ __forceinline__ long q_offset(const AttnP& p, int b, int t, long bs, long ts) {
  if (p.mode == 2) {
    int img, pix;
    win_decode(p, b, t, 0, img, pix);
    return (long)img * bs + (long)pix * ts;
  }
  return (long)b * bs + (long)t * ts;
}

__device__ __forceinline__ long kv_offset(const AttnP& p, int b, int t, long bs, long ts) {
  if (p.mode == 2) {
    int img, pix;
    win_decode(p, b, t, p.kv_rot, img, pix);
    return (long)img * bs + (long)pix * ts;
  }
  if (p.mode == 1) {
    const int f = b % p.T;
    int src, tt;
    if (t < p.seg_len) {
      src = b - f;
      tt = t;
    } else {
      src = (f == 0) ? b : b - 1;
      tt = t - p.seg_len;
    }
    return (long)src * bs + (long)tt * ts;
  }
  return (long)b * bs + (long)t * ts;
}

// Loads one [32 x ncols] fp32 tile (rows = key/query tokens, cols contiguous) global -> LDS (row pitch `pitch`),
// float4 along the contiguous axis; row offsets (window / sparse-causal index math) are computed once per float4.
template <int NT, bool IS_Q>
__device__ __forceinline__ void load_tile_direct(const AttnP& p, const float* base, long bs, long ts, long hoff,
                                                 int b, int t0, int tmax, int nrows, int c0, int ncols, int cmax,
                                                 float* dst, int pitch, int tid, bool vec) {
  if (vec) {
    const int c4n = ncols >> 2;
    for (int i = tid; i < nrows * c4n; i += NT) {
      const int row = i / c4n, c = (i - row * c4n) << 2;
      const int t = t0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < tmax && c0 + c < cmax) {
        const long off = (IS_Q ? q_offset(p, b, t, bs, ts) : kv_offset(p, b, t, bs, ts)) + hoff + c0 + c;
        v = *reinterpret_cast<const float4*>(base + off);
      }
      float* d = dst + row * pitch + c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    for (int i = tid; i < nrows * ncols; i += NT) {
      const int row = i / ncols, c = i - row * ncols;
      const int t = t0 + row;
      float v = 0.f;
      if (t < tmax && c0 + c < cmax)
        v = base[(IS_Q ? q_offset(p, b, t, bs, ts) : kv_offset(p, b, t, bs, ts)) + hoff + c0 + c];
      dst[row * pitch + c] = v;
    }
  }
}

// WAVES waves x 32 queries per block; QK depth processed in chunks of DC = min(D,128) so the Q tile of the
// block fits LDS for every head size (D = 512 for the VQGAN AttnBlock): LDS = (WAVES*32 + 32)*(DC+1) + 32*DVS floats.
// With a single chunk (D <= 128) and 4 waves, the NEXT key tile's K and V are prefetched into registers while the
// current tile is on the matrix cores.
template <int WAVES, int DVT>
__global__ __launch_bounds__(64 * WAVES, 2) void attn_f32_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int DVS = DVT * 32;    // dv slice handled by this block
  constexpr int NT = 64 * WAVES;
  constexpr bool CAN_PREFETCH = (WAVES == 4);
  constexpr int PF = 4;            // float4 per thread per prefetched tile (32 x 128 floats / 256 threads)
  const int DC = p.D < 128 ? p.D : 128;
  const int nch = p.D / DC;
  const int QP = DC + 1;           // odd pitch: 32 rows -> 32 distinct banks
  float* Qs = smem;                              // [WAVES*32][QP]
  float* Ks = Qs + WAVES * 32 * QP;              // [32][QP]
  float* Vs = Ks + 32 * QP;                      // [32][DVS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z;
  const int head = blockIdx.y / p.nslices;
  const int dv0 = (blockIdx.y - head * p.nslices) * DVS;
  const int q0 = blockIdx.x * (WAVES * 32);
  const long qh = (long)head * p.q_hs, kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  const bool qk_vec = (DC % 4 == 0) && (p.q_ts % 4 == 0) && (p.q_bs % 4 == 0) && (p.q_hs % 4 == 0) &&
                      (p.k_ts % 4 == 0) && (p.k_bs % 4 == 0) && (p.k_hs % 4 == 0) &&
                      ((uintptr_t)p.q % 16 == 0) && ((uintptr_t)p.k % 16 == 0);
  const bool v_vec = (p.Dv % 4 == 0) && (p.v_ts % 4 == 0) && (p.v_bs % 4 == 0) && (p.v_hs % 4 == 0) &&
                     ((uintptr_t)p.v % 16 == 0);
  const bool prefetch = CAN_PREFETCH && nch == 1 && qk_vec && v_vec;

  if (nch == 1)
    load_tile_direct<NT, true>(p, p.q, p.q_bs, p.q_ts, qh, b, q0, p.Lq, WAVES * 32, 0, DC, p.D, Qs, QP, tid, qk_vec);

  const int my_q = q0 + wave * 32 + l31;  // query owned by this lane in the S^T layout
  int my_region = 0;
  if (p.mode == 2 && p.shift > 0 && my_q < p.Lq) my_region = win_region(p, b, my_q);

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[DVT];
#pragma unroll
  for (int j = 0; j < DVT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;

  const float* Qw = Qs + wave * 32 * QP;
  const float* kp = Ks + l31 * QP + lhi;
  const float* qp = Qw + l31 * QP + lhi;
  const int ntiles = (p.Lk + 31) / 32;

  // ---- register prefetch state (single-chunk path)
  float4 kreg[PF], vreg[PF];
  const int kc4n = DC >> 2, vc4n = DVS >> 2;
  auto pf_issue = [&](int kt) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = tid + u * NT;
      kreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      vreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 32 * kc4n) {
        const int row = i / kc4n, c = (i - row * kc4n) << 2;
        const int t = kt * 32 + row;
        if (t < p.Lk) kreg[u] = *reinterpret_cast<const float4*>(p.k + kv_offset(p, b, t, p.k_bs, p.k_ts) + kh + c);
      }
      if (i < 32 * vc4n) {
        const int row = i / vc4n, c = (i - row * vc4n) << 2;
        const int t = kt * 32 + row;
        if (t < p.Lk && dv0 + c < p.Dv)
          vreg[u] = *reinterpret_cast<const float4*>(p.v + kv_offset(p, b, t, p.v_bs, p.v_ts) + vh + dv0 + c);
      }
    }
  };
  auto pf_commit = [&]() {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = tid + u * NT;
      if (i < 32 * kc4n) {
        const int row = i / kc4n, c = (i - row * kc4n) << 2;
        float* d = Ks + row * QP + c;
        d[0] = kreg[u].x; d[1] = kreg[u].y; d[2] = kreg[u].z; d[3] = kreg[u].w;
      }
      if (i < 32 * vc4n) {
        const int row = i / vc4n, c = (i - row * vc4n) << 2;
        *reinterpret_cast<float4*>(Vs + row * DVS + c) = vreg[u];
      }
    }
  };

  if (prefetch) {
    pf_issue(0);
    pf_commit();
  }

  for (int kt = 0; kt < ntiles; ++kt) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;

    if (prefetch) {
      __syncthreads();                         // tile kt (and Q) visible in LDS
      if (kt + 1 < ntiles) pf_issue(kt + 1);   // next tile's loads fly during the MFMAs below
      for (int d2 = 0; d2 < DC; d2 += 2) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[d2], qp[d2], s, 0, 0, 0);
    } else {
      for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();                       // previous chunk / tile fully consumed
        if (nch > 1)
          load_tile_direct<NT, true>(p, p.q, p.q_bs, p.q_ts, qh, b, q0, p.Lq, WAVES * 32, ch * DC, DC, p.D, Qs, QP,
                                     tid, qk_vec);
        load_tile_direct<NT, false>(p, p.k, p.k_bs, p.k_ts, kh, b, kt * 32, p.Lk, 32, ch * DC, DC, p.D, Ks, QP, tid,
                                    qk_vec);
        if (ch == 0)
          load_tile_direct<NT, false>(p, p.v, p.v_bs, p.v_ts, vh, b, kt * 32, p.Lk, 32, dv0, DVS, p.Dv, Vs, DVS, tid,
                                      v_vec);
        __syncthreads();
        for (int d2 = 0; d2 < DC; d2 += 2) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[d2], qp[d2], s, 0, 0, 0);
      }
    }

    // ---- scale, mask, online softmax (row = this lane's query)
    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float val = s[r] * p.scale;
      if (p.mode == 2 && p.shift > 0 && key < p.Lk) {
        if (win_region(p, b, key) != my_region) val += -100.0f;
      }
      if (key >= p.Lk) val = -INFINITY;
      s[r] = val;
      mloc = fmaxf(mloc, val);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = expf(m_run - m_new);  // first tile: exp(-inf) = 0
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[r] - m_new);
      s[r] = pv;
      lsum += pv;
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;

    // ---- rescale O: its rows (queries) sit at (r&3)+8*(r>>2)+4*lhi -> fetch that query's alpha
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float ar = __shfl(alpha, qrow);
#pragma unroll
      for (int j = 0; j < DVT; ++j) o[j][r] *= ar;
    }
    // ---- O += P . V   (step r: k=0 <-> key kr(r,0), k=1 <-> key kr(r,1); A operand = own p[r])
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int krow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float* vp = Vs + krow * DVS + l31;
#pragma unroll
      for (int j = 0; j < DVT; ++j) o[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[r], vp[j * 32], o[j], 0, 0, 0);
    }
    if (prefetch && kt + 1 < ntiles) {
      __syncthreads();                         // every wave is done reading tile kt
      pf_commit();
    }
  }

  // ---- normalise and store: O rows = queries (r&3)+8*(r>>2)+4*lhi, cols = dv0 + j*32 + l31
  const float inv_l = 1.0f / l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float il = __shfl(inv_l, qrow);
    const int t = q0 + wave * 32 + qrow;
    if (t < p.Lq) {
      const long base = q_offset(p, b, t, p.o_bs, p.o_ts) + (long)head * p.o_hs;
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const int dv = dv0 + j * 32 + l31;
        if (dv < p.Dv) p.o[base + dv] = o[j][r] * il;
      }
    }
  }
}

template <int WAVES, int DVT>
static int launch_attn(const AttnP& p, hipStream_t st) {
  const int DC = p.D < 128 ? p.D : 128;
  const size_t lds = (size_t)((WAVES * 32 + 32) * (DC + 1) + 32 * DVT * 32) * sizeof(float);
  static bool attr_set = false;  // per instantiation; idempotent, benign if raced
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_f32_kernel<WAVES, DVT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, WAVES * 32), p.H * p.nslices, p.B);
  hipLaunchKernelGGL((attn_f32_kernel<WAVES, DVT>), grid, dim3(64 * WAVES), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ bf16 MFMA variant
// Same algorithm with v_mfma_f32_32x32x16_bf16 (Q, K, V and P rounded to bf16, fp32 accumulate, fp32 softmax).
//   S^T = K . Q^T : fragments are 8 consecutive d per lane (one ds_read_b128) from row-major bf16 tiles, pitch DC+8.
//   P . V : a 16-key MFMA step s takes the lane's OWN registers p[8s..8s+7] (keys (j&3) + 16s + 8(j>>2) + 4h of
//   its query) as the A operand, so P never moves; the matching B operand is one 16-byte read of V^T stored with the
//   keys of each row permuted into exactly that order: pos(key) = ((key>>4)*2 + ((key>>2)&1))*8 + (((key>>3)&1)<<2 | key&3).
typedef __attribute__((ext_vector_type(8))) __bf16 abf16x8;

__device__ __forceinline__ int vt_pos(int key) {
  return (((key >> 4) << 1) + ((key >> 2) & 1)) * 8 + ((((key >> 3) & 1) << 2) | (key & 3));
}

template <int WAVES, int DVT>
__global__ __launch_bounds__(64 * WAVES) void attn_bf16_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem16[];
  constexpr int DVS = DVT * 32;
  constexpr int NT = 64 * WAVES;
  constexpr int VP = 40;                 // V^T row pitch in bf16 (32 keys + 8 pad = 80 B)
  const int DC = p.D < 128 ? p.D : 128;
  const int nch = p.D / DC;
  const int QP = DC + 8;                 // bf16 elements; (DC+8)*2 B is an odd number of 16-B slots for DC % 16 == 0
  __bf16* Qs = smem16;                               // [WAVES*32][QP]
  __bf16* Ks = Qs + WAVES * 32 * QP;                 // [32][QP]
  __bf16* Vt = Ks + 32 * QP;                         // [DVS][VP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z;
  const int head = blockIdx.y / p.nslices;
  const int dv0 = (blockIdx.y - head * p.nslices) * DVS;
  const int q0 = blockIdx.x * (WAVES * 32);
  const long qh = (long)head * p.q_hs, kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  const int g8n = DC >> 3;               // 8-column groups per row

  // fp32 rows -> bf16 LDS rows, 8 columns per piece
  auto stage_rows = [&](const float* base, long bs, long ts, long hoff, bool is_q, int t0, int tmax, int nrows,
                        int c0, __bf16* dst) {
    for (int i = tid; i < nrows * g8n; i += NT) {
      const int row = i / g8n, c = (i - row * g8n) << 3;
      const int t = t0 + row;
      abf16x8 h;
      if (t < tmax) {
        const float* src = base + (is_q ? q_offset(p, b, t, bs, ts) : kv_offset(p, b, t, bs, ts)) + hoff + c0 + c;
        const float4 a = *reinterpret_cast<const float4*>(src);
        const float4 c4 = *reinterpret_cast<const float4*>(src + 4);
        h[0] = (__bf16)a.x; h[1] = (__bf16)a.y; h[2] = (__bf16)a.z; h[3] = (__bf16)a.w;
        h[4] = (__bf16)c4.x; h[5] = (__bf16)c4.y; h[6] = (__bf16)c4.z; h[7] = (__bf16)c4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (__bf16)0.f;
      }
      *reinterpret_cast<abf16x8*>(dst + row * QP + c) = h;
    }
  };
  // V tile -> V^T with permuted key order; lane <-> dv column (coalesced 4-byte reads), key pairs packed per dword
  auto stage_vt = [&](int kt) {
    for (int i = tid; i < 16 * DVS; i += NT) {
      const int pair = i / DVS, dv = i - pair * DVS;
      const int k0 = pair * 2;
      const int t = kt * 32 + k0;
      float v0 = 0.f, v1 = 0.f;
      if (dv0 + dv < p.Dv) {
        if (t < p.Lk) v0 = p.v[kv_offset(p, b, t, p.v_bs, p.v_ts) + vh + dv0 + dv];
        if (t + 1 < p.Lk) v1 = p.v[kv_offset(p, b, t + 1, p.v_bs, p.v_ts) + vh + dv0 + dv];
      }
      __bf16* d = Vt + dv * VP + vt_pos(k0);         // vt_pos(k0+1) = vt_pos(k0) + 1 for even k0
      d[0] = (__bf16)v0;
      d[1] = (__bf16)v1;
    }
  };

  if (nch == 1) stage_rows(p.q, p.q_bs, p.q_ts, qh, true, q0, p.Lq, WAVES * 32, 0, Qs);

  const int my_q = q0 + wave * 32 + l31;
  int my_region = 0;
  if (p.mode == 2 && p.shift > 0 && my_q < p.Lq) my_region = win_region(p, b, my_q);

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[DVT];
#pragma unroll
  for (int j = 0; j < DVT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;

  const __bf16* kp = Ks + l31 * QP + lhi * 8;
  const __bf16* qp = Qs + (wave * 32 + l31) * QP + lhi * 8;
  const int ntiles = (p.Lk + 31) / 32;

  for (int kt = 0; kt < ntiles; ++kt) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();
      if (nch > 1) stage_rows(p.q, p.q_bs, p.q_ts, qh, true, q0, p.Lq, WAVES * 32, ch * DC, Qs);
      stage_rows(p.k, p.k_bs, p.k_ts, kh, false, kt * 32, p.Lk, 32, ch * DC, Ks);
      if (ch == 0) stage_vt(kt);
      __syncthreads();
      for (int d = 0; d < DC; d += 16)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const abf16x8*>(kp + d),
                                                    *reinterpret_cast<const abf16x8*>(qp + d), s, 0, 0, 0);
    }

    float mloc = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      float val = s[r] * p.scale;
      if (p.mode == 2 && p.shift > 0 && key < p.Lk) {
        if (win_region(p, b, key) != my_region) val += -100.0f;
      }
      if (key >= p.Lk) val = -INFINITY;
      s[r] = val;
      mloc = fmaxf(mloc, val);
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __expf(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __expf(s[r] - m_new);
      s[r] = pv;
      lsum += pv;
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;

#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float ar = __shfl(alpha, qrow);
#pragma unroll
      for (int j = 0; j < DVT; ++j) o[j][r] *= ar;
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      abf16x8 pa;
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[j] = (__bf16)s[st * 8 + j];
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const abf16x8 vb = *reinterpret_cast<const abf16x8*>(Vt + (j * 32 + l31) * VP + (st * 2 + lhi) * 8);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, o[j], 0, 0, 0);
      }
    }
  }

  const float inv_l = 1.0f / l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float il = __shfl(inv_l, qrow);
    const int t = q0 + wave * 32 + qrow;
    if (t < p.Lq) {
      const long base = q_offset(p, b, t, p.o_bs, p.o_ts) + (long)head * p.o_hs;
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const int dv = dv0 + j * 32 + l31;
        if (dv < p.Dv) p.o[base + dv] = o[j][r] * il;
      }
    }
  }
}

template <int WAVES, int DVT>
static int launch_attn_bf16(const AttnP& p, hipStream_t st) {
  const int DC = p.D < 128 ? p.D : 128;
  const size_t lds = (size_t)((WAVES * 32 + 32) * (DC + 8) + DVT * 32 * 40) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bf16_kernel<WAVES, DVT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, WAVES * 32), p.H * p.nslices, p.B);
  hipLaunchKernelGGL((attn_bf16_kernel<WAVES, DVT>), grid, dim3(64 * WAVES), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(bf16)");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ split fp16 (x3) variant
// KEEP_MMA_X3 (see keep_conv_x3.hip): Q, K, V and P are each split into two fp16 halves (x = hi + lo) and every product
// runs as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 -- fp32-grade scores and outputs at 3 MFMAs per 32x32x16 step
// instead of 8 f32 MFMAs (32x32x2) at 1/16 of the rate.  Same algorithm as attn_bf16_kernel (S^T = K.Q^T so the softmax
// reduction is in-lane; P feeds P.V from the lane's own registers against a key-permuted V^T); softmax in exact fp32 (expf).
//   LDS rows: Q / K  [hi x DC | lo x DC | pad x 8] fp16 (pitch 2*DC+8: an odd number of 16-byte slots for DC % 8 == 0),
//             V^T    [32 keys hi (permuted) | 32 keys lo | pad x 8] per dv column (pitch 72).
//   The next key tile's K and V are fetched into registers while the current tile is on the matrix cores (single D chunk).
typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 af16x2 __attribute__((ext_vector_type(2)));

// PF: register prefetch of the next key tile (4-wave blocks); QREG: D <= 128, the wave's Q fragments (hi + lo, 8 x 16-d
// steps) live in registers for the whole kernel -- loaded straight from global, never staged -- so a block's LDS is one
// K tile + one V^T tile (<= 35 KB) and two blocks share a CU.  D > 128 (VQGAN AttnBlock D = 512, CFA D = 256; 16x16 / 32x32
// token maps): Q and K are re-staged through LDS per 128-wide chunk.
// PACKED (WAVES 4, D = 128 or 256, any Dv): K and V^T tiles come pre-split from attn_pack_kv_x3_kernel as the exact LDS image
// [32 x (hi128|lo128|pad8) | 128 x (32 hi permuted | 32 lo | pad8)] -- 2208 16-byte pieces per key tile copied global -> register
// -> LDS: 9 wide loads and 9 ds_write_b128 per thread and tile instead of 20 loads (16 of them 4-byte), 128 split
// operations and 20 narrow LDS writes (the phase ablation: loads 55 %, commit 25 % of the unpacked kernel's time).
template <int WAVES, int DVT, int NQ, bool PACKED = false>
__global__ __launch_bounds__(64 * WAVES, ((WAVES == 4 && NQ != 16) ? 2 : 1)) void attn_x3_kernel(AttnP p) {
#ifdef KEEP_X3_ABLATE
  const int abl = p.abl;
#else
  constexpr int abl = 0;
#endif
  constexpr bool QREG = NQ > 0;             // NQ 16-wide d steps of Q live in registers (NQ = 16: D <= 256, one block per CU)
  constexpr int NQA = QREG ? NQ : 1;
  extern __shared__ __attribute__((aligned(16))) _Float16 smemx[];
  constexpr int DVS = DVT * 32;
  constexpr int NT = 64 * WAVES;
  constexpr int VP = 72;
  constexpr bool PF = (WAVES == 4) && QREG;
  constexpr int KPF = PF ? (32 * 4 * NQA + NT - 1) / NT : 1;   // float4 pieces of a 32 x (16*NQ) K tile per thread
  constexpr int VPF = PF ? (16 * DVS + NT - 1) / NT : 1;       // (key pair, dv) items of a 32 x DVS V tile per thread
  const int DCMAX = QREG ? 16 * NQA : 128;
  const int DC = p.D < DCMAX ? p.D : DCMAX;
  const int nch = p.D / DC;
  const int QP = 2 * DC + 8;
  _Float16* Ks = smemx;                               // [32][QP]
  _Float16* Vt = Ks + 32 * QP;                        // [DVS][VP]
  _Float16* Qs = Vt + DVS * VP;                       // [WAVES*32][QP], chunked path only
  // Window mode (GMFlow swin attention, mode 2) on the prefetching variant: the window-token -> pixel map and the shifted-
  // window region ids of ALL keys are computed once per block into LDS (tpix: Lk ints; treg: 32 nibbles per key tile) instead
  // of ~36 index computations (float reciprocal division, wrap, compares) per thread and key tile -- they were 4x the MFMA
  // issue time of this kernel (121 TFLOP/s on the 1024-token windows).
  const bool tab = PF && p.mode == 2 && p.Lk <= 4096;
  int* tpix = reinterpret_cast<int*>(Qs);
  unsigned* treg = reinterpret_cast<unsigned*>(tpix + ((p.Lk + 31) & ~31));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // XCD-aware block order: the hardware deals linear block ids round-robin to the 8 XCDs, so the query blocks of ONE
  // (batch, head) -- which all stream the same K / V -- would land on 8 different L2s and fetch K / V 8 times from HBM
  // (19.5 GB per GMFlow window-attention call: 14 TB/s of demand).  Give each XCD a contiguous range of logical ids:
  // the query blocks of one (batch, head) become neighbours in one L2.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int lid = bx + gx * (by + gy * bz);
    const int qd = total >> 3, rm = total & 7, xcd = lid & 7, slot = lid >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
    bx = logical % gx;
    const int rest = logical / gx;
    by = rest % gy;
    bz = rest / gy;
  }
  const int b = bz;
  const int head = by / p.nslices;
  const int dv0 = (by - head * p.nslices) * DVS;
  const int q0 = bx * (WAVES * 32);
  const long qh = (long)head * p.q_hs, kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  const int g4n = DC >> 2;
  WinCtx wc;
  if (tab) {
    wc = win_ctx(p, b);
    const int ntl = (p.Lk + 31) >> 5;
    unsigned char* treg8 = reinterpret_cast<unsigned char*>(treg + ntl * 4);
    for (int t = tid; t < ntl * 32; t += NT) {
      int pix = 0, region = 0;
      if (t < p.Lk) win_token(p, wc, t, pix, region);
      tpix[t] = pix;
      treg8[t] = (unsigned char)region;
    }
    __syncthreads();
    for (int i = tid; i < ntl * 4; i += NT) {
      unsigned w = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) w |= (unsigned)treg8[i * 8 + j] << (4 * j);
      treg[i] = w;
    }
    __syncthreads();
  }
  // range scales of un-normalised operands (mode 0, host-probed): q, k, v are multiplied by powers of two before the
  // split; the score scale and the output normalisation absorb the inverses (all exact)
  float sq = 1.f, sk = 1.f, sv = 1.f, inv_qk = 1.f, inv_sv = 1.f;
  if (p.q_amax) {
    float iq, ik;
    attn_range_scale(p.q_amax[b], sq, iq);
    attn_range_scale(p.k_amax[b], sk, ik);
    attn_range_scale(p.v_amax[b], sv, inv_sv);
    inv_qk = iq * ik;
  }

  auto split8 = [&](const float4 a, const float4 c, float sc, af16x8& hi, af16x8& lo) {
    const float f[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, c.x * sc, c.y * sc, c.z * sc, c.w * sc};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)f[j];
      hi[j] = h;
      lo[j] = (_Float16)(f[j] - (float)h);
    }
  };
  auto split_store4 = [&](const float4 v, float sc, _Float16* dst) {      // dst -> hi; lo lives DC elements further
    const float f[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
    af16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const _Float16 h = (_Float16)f[j];
      hi[j] = h;
      lo[j] = (_Float16)(f[j] - (float)h);
    }
    *reinterpret_cast<af16x4*>(dst) = hi;
    *reinterpret_cast<af16x4*>(dst + DC) = lo;
  };
  auto stage_rows = [&](const float* base, long bs, long ts, long hoff, bool is_q, int t0, int tmax, int nrows, int c0,
                        _Float16* dst) {
    for (int i = tid; i < nrows * g4n; i += NT) {
      const int row = i / g4n, c = (i - row * g4n) << 2;
      const int t = t0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < tmax)
        v = *reinterpret_cast<const float4*>(base + (is_q ? q_offset(p, b, t, bs, ts) : kv_offset(p, b, t, bs, ts)) + hoff + c0 + c);
      split_store4(v, is_q ? sq : sk, dst + row * QP + c);
    }
  };
  auto stage_vt = [&](int kt) {      // lane <-> dv column (coalesced 4-byte reads), two consecutive keys per item
    for (int i = tid; i < 16 * DVS; i += NT) {
      const int pair = i / DVS, dv = i - pair * DVS;
      const int t = kt * 32 + pair * 2;
      float v[2] = {0.f, 0.f};
      if (dv0 + dv < p.Dv) {
        if (t < p.Lk) v[0] = p.v[kv_offset(p, b, t, p.v_bs, p.v_ts) + vh + dv0 + dv];
        if (t + 1 < p.Lk) v[1] = p.v[kv_offset(p, b, t + 1, p.v_bs, p.v_ts) + vh + dv0 + dv];
      }
      af16x2 hi, lo;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float vs = v[j] * sv;
        const _Float16 h = (_Float16)vs;
        hi[j] = h;
        lo[j] = (_Float16)(vs - (float)h);
      }
      _Float16* d = Vt + dv * VP + vt_pos(pair * 2);          // vt_pos(k0+1) = vt_pos(k0) + 1 for even k0
      *reinterpret_cast<af16x2*>(d) = hi;
      *reinterpret_cast<af16x2*>(d + 32) = lo;
    }
  };
  // ---- register prefetch of the next key tile
  float4 kreg[KPF];
  float vreg[VPF][2];
  auto k_issue = [&](int kt) {
#pragma unroll
    for (int u = 0; u < KPF; ++u) {
      const int i = tid + u * NT;
      kreg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 32 * g4n) {
        const int row = i / g4n, c = (i - row * g4n) << 2;
        const int t = kt * 32 + row;
        if (t < p.Lk) {
          const long off = tab ? (long)wc.img_kv * p.k_bs + (long)tpix[t] * p.k_ts : kv_offset(p, b, t, p.k_bs, p.k_ts);
          kreg[u] = *reinterpret_cast<const float4*>(p.k + off + kh + c);
        }
      }
    }
  };
  auto k_commit = [&]() {
#pragma unroll
    for (int u = 0; u < KPF; ++u) {
      const int i = tid + u * NT;
      if (i < 32 * g4n) {
        const int row = i / g4n, c = (i - row * g4n) << 2;
        split_store4(kreg[u], sk, Ks + row * QP + c);
      }
    }
  };
  auto v_issue = [&](int kt) {
#pragma unroll
    for (int u = 0; u < VPF; ++u) {
      const int i = tid + u * NT;
      vreg[u][0] = 0.f;
      vreg[u][1] = 0.f;
      if (i < 16 * DVS) {
        const int pair = i / DVS, dv = i - pair * DVS;
        const int t = kt * 32 + pair * 2;
        if (dv0 + dv < p.Dv) {
          if (tab) {      // `pair` is wave-uniform (DVS >= 64): the two table reads are LDS broadcasts
            const long vb = (long)wc.img_kv * p.v_bs + vh + dv0 + dv;
            if (t < p.Lk) vreg[u][0] = p.v[vb + (long)tpix[t] * p.v_ts];
            if (t + 1 < p.Lk) vreg[u][1] = p.v[vb + (long)tpix[t + 1] * p.v_ts];
          } else {
            if (t < p.Lk) vreg[u][0] = p.v[kv_offset(p, b, t, p.v_bs, p.v_ts) + vh + dv0 + dv];
            if (t + 1 < p.Lk) vreg[u][1] = p.v[kv_offset(p, b, t + 1, p.v_bs, p.v_ts) + vh + dv0 + dv];
          }
        }
      }
    }
  };
  auto v_commit = [&]() {
#pragma unroll
    for (int u = 0; u < VPF; ++u) {
      const int i = tid + u * NT;
      if (i < 16 * DVS) {
        const int pair = i / DVS, dv = i - pair * DVS;
        af16x2 hi, lo;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float vs = vreg[u][j] * sv;
          const _Float16 h = (_Float16)vs;
          hi[j] = h;
          lo[j] = (_Float16)(vs - (float)h);
        }
        _Float16* d = Vt + dv * VP + vt_pos(pair * 2);
        *reinterpret_cast<af16x2*>(d) = hi;
        *reinterpret_cast<af16x2*>(d + 32) = lo;
      }
    }
  };

  // packed tile of (batch, head, key tile): [K image 32 x (2 D + 8)] [V^T slice 0: DVS x 72] [slice 1] ...; a block copies the K
  // image and ITS dv slice -- contiguous in LDS (Ks, then Vt)
  constexpr int PKK = PACKED ? 32 * (2 * 16 * NQA + 8) / 8 : 1;       // 16-byte pieces of the K image (D == 16 NQ)
  constexpr int PKV = PACKED ? DVS * 72 / 8 : 1;                      // ... of one V^T slice
  constexpr int PK16 = PKK + PKV;
  constexpr int PKN = PACKED ? (PK16 + NT - 1) / NT : 1;
  uint4 preg[PKN];
  auto pk_issue = [&](int kt) {
    const long tile = ((long)b * p.H + head) * ((p.Lk + 31) >> 5) + kt;
    const uint4* src = reinterpret_cast<const uint4*>(p.kv_pack) + tile * (PKK + p.nslices * PKV);
    const int vs = PKK + (dv0 / DVS) * PKV;
#pragma unroll
    for (int u = 0; u < PKN; ++u) {
      const int i = tid + u * NT;
      preg[u] = make_uint4(0u, 0u, 0u, 0u);
      if (i < PK16) preg[u] = src[i < PKK ? i : vs + (i - PKK)];
    }
  };
  auto pk_commit = [&]() {
#pragma unroll
    for (int u = 0; u < PKN; ++u) {
      const int i = tid + u * NT;
      if (i < PK16) reinterpret_cast<uint4*>(Ks)[i] = preg[u];
    }
  };

  const int my_q = q0 + wave * 32 + l31;
  // ---- Q fragments of this lane: row my_q, d in [16*step + 8*lhi, +8), split once
  af16x8 qfh[NQA], qfl[NQA];
  if (QREG) {
    const bool qok = my_q < p.Lq;
    const float* qrow = p.q + (qok ? q_offset(p, b, my_q, p.q_bs, p.q_ts) + qh : 0) + lhi * 8;
#pragma unroll
    for (int d8 = 0; d8 < NQA; ++d8) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (qok && d8 * 16 < DC) {
        a = *reinterpret_cast<const float4*>(qrow + d8 * 16);
        c = *reinterpret_cast<const float4*>(qrow + d8 * 16 + 4);
      }
      split8(a, c, sq, qfh[QREG ? d8 : 0], qfl[QREG ? d8 : 0]);
    }
  }
  if (PF) {
    if (PACKED) {
      pk_issue(0);
      pk_commit();
    } else {
      k_issue(0);
      v_issue(0);
      k_commit();
      v_commit();
    }
  }

  int my_region = 0;
  if (p.mode == 2 && p.shift > 0 && my_q < p.Lq) my_region = win_region(p, b, my_q);

  const float qk_scale = p.scale * inv_qk;
  const float qk_scale2 = qk_scale * 1.4426950408889634f;      // scores in the exp2 domain (PF variant)
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[DVT];
#pragma unroll
  for (int j = 0; j < DVT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;

  const _Float16* kp = Ks + l31 * QP + lhi * 8;
  const _Float16* qp = Qs + (wave * 32 + l31) * QP + lhi * 8;
  const int ntiles = (p.Lk + 31) / 32;

  for (int kt = 0; kt < ntiles; ++kt) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    if (QREG) {
      if (!PF) {
        __syncthreads();                             // previous tile fully consumed
        stage_rows(p.k, p.k_bs, p.k_ts, kh, false, kt * 32, p.Lk, 32, 0, Ks);
        stage_vt(kt);
      }
      __syncthreads();                               // tile kt visible in LDS
      if (PF && kt + 1 < ntiles && abl != 5) {
        if (PACKED) {
          pk_issue(kt + 1);                          // next tile's image flies during the MFMAs below
        } else {
          k_issue(kt + 1);
          v_issue(kt + 1);
        }
      }
#pragma unroll
      for (int d8 = 0; d8 < NQA; ++d8) {
        if (d8 * 16 < DC && abl != 1) {
          const af16x8 kh8 = *reinterpret_cast<const af16x8*>(kp + (abl == 6 ? 0 : d8 * 16));
          const af16x8 kl8 = *reinterpret_cast<const af16x8*>(kp + DC + (abl == 6 ? 0 : d8 * 16));
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qfh[QREG ? d8 : 0], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qfl[QREG ? d8 : 0], s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qfh[QREG ? d8 : 0], s, 0, 0, 0);
        }
      }
    } else {
      for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        stage_rows(p.q, p.q_bs, p.q_ts, qh, true, q0, p.Lq, WAVES * 32, ch * DC, Qs);
        stage_rows(p.k, p.k_bs, p.k_ts, kh, false, kt * 32, p.Lk, 32, ch * DC, Ks);
        if (ch == 0) stage_vt(kt);
        __syncthreads();
        for (int d = 0; d < DC; d += 16) {
          const af16x8 kh8 = *reinterpret_cast<const af16x8*>(kp + d), kl8 = *reinterpret_cast<const af16x8*>(kp + DC + d);
          const af16x8 qh8 = *reinterpret_cast<const af16x8*>(qp + d), ql8 = *reinterpret_cast<const af16x8*>(qp + DC + d);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh8, s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql8, s, 0, 0, 0);
          s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh8, s, 0, 0, 0);
        }
      }
    }

    float mloc = -INFINITY;
    if (PF) {
      // exp2 domain (one v_exp_f32 per probability instead of the 13-instruction expf: x3 grade, keep_common.h); the
      // shifted-window mask (-100 on cross-region pairs, GM/transformer.py:24-35) from the packed region table
      uint4 rw = make_uint4(0u, 0u, 0u, 0u);
      const bool masked = p.mode == 2 && p.shift > 0;
      if (masked && tab) rw = *reinterpret_cast<const uint4*>(treg + kt * 4);
      const unsigned rws[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kl = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const int key = kt * 32 + kl;
        float val = s[r] * qk_scale2;
        if (masked) {
          const int kr = tab ? (int)((rws[r >> 2] >> (4 * ((r & 3) + 4 * lhi))) & 15u) : (key < p.Lk ? win_region(p, b, key) : my_region);
          if (kr != my_region) val += -100.0f * 1.4426950408889634f;
        }
        if (key >= p.Lk) val = -INFINITY;
        s[r] = val;
        mloc = fmaxf(mloc, val);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float val = s[r] * qk_scale;
        if (p.mode == 2 && p.shift > 0 && key < p.Lk) {
          if (win_region(p, b, key) != my_region) val += -100.0f;
        }
        if (key >= p.Lk) val = -INFINITY;
        s[r] = val;
        mloc = fmaxf(mloc, val);
      }
    }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = PF ? __builtin_amdgcn_exp2f(m_run - m_new) : expf(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = abl == 3 ? s[r] - m_new : (PF ? __builtin_amdgcn_exp2f(s[r] - m_new) : expf(s[r] - m_new));
      s[r] = pv;
      lsum += pv;
    }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;

    // rescale of the running output: skipped when no query of the wave raised its maximum (alpha == 1 exactly) -- after the
    // first few key tiles that is the common case, and the 16 cross-lane broadcasts + 16*DVT multiplies go away with it
    if (!PF || __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float ar = __shfl(alpha, qrow);
#pragma unroll
        for (int j = 0; j < DVT; ++j) o[j][r] *= ar;
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      af16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)s[st * 8 + j];
        ph[j] = h;
        pl[j] = (_Float16)(s[st * 8 + j] - (float)h);
      }
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        if (abl == 2) continue;
        const _Float16* vrow = Vt + ((abl == 6 ? 0 : j) * 32 + l31) * VP + (st * 2 + lhi) * 8;
        const af16x8 vh8 = *reinterpret_cast<const af16x8*>(vrow), vl8 = *reinterpret_cast<const af16x8*>(vrow + 32);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh8, o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl8, o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh8, o[j], 0, 0, 0);
      }
    }
    if (PF && kt + 1 < ntiles) {
      __syncthreads();                               // every wave is done reading tile kt
      if (abl != 4) {
        if (PACKED) {
          pk_commit();
        } else {
          k_commit();
          v_commit();
        }
      }
    }
  }

  const float inv_l = inv_sv / l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float il = __shfl(inv_l, qrow);
    const int t = q0 + wave * 32 + qrow;
    if (t < p.Lq) {
      const long base = q_offset(p, b, t, p.o_bs, p.o_ts) + (long)head * p.o_hs;
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const int dv = dv0 + j * 32 + l31;
        if (dv < p.Dv) KEEP_ATTN_O_STORE(p.o + base + dv, o[j][r] * il);
      }
    }
  }
}

// K / V^T tile images for the PACKED variant above: one block per (key tile, head, batch) gathers the tile's 32 key rows
// (window / sparse-causal index math once per element instead of once per query block), applies the range scales, splits,
// assembles the LDS image in LDS and writes it out with 16-byte stores.
template <int DC, int DVS>
__global__ __launch_bounds__(256) void attn_pack_kv_x3_kernel(AttnP p, _Float16* out) {
  constexpr int QP = 2 * DC + 8, VP = 72, PKK = 32 * QP / 8, PKV = DVS * VP / 8;
  __shared__ __attribute__((aligned(16))) _Float16 img[32 * QP + DVS * VP];
  _Float16* Ks = img;
  _Float16* Vt = img + 32 * QP;
  const int tid = threadIdx.x;
  const int kt = blockIdx.x, head = blockIdx.y / p.nslices, slice = blockIdx.y - head * p.nslices, b = blockIdx.z;
  float sk = 1.f, sv = 1.f, tmp;
  if (p.q_amax) {
    attn_range_scale(p.k_amax[b], sk, tmp);
    attn_range_scale(p.v_amax[b], sv, tmp);
  }
  for (int i = tid; i < PKK + PKV; i += 256) reinterpret_cast<uint4*>(img)[i] = make_uint4(0u, 0u, 0u, 0u);   // pads, tail keys
  __syncthreads();
  const long kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  if (slice == 0) {                                            // the K image is shared by the dv slices: written once
    for (int i = tid; i < 32 * (DC / 4); i += 256) {           // K: 32 rows x DC/4 float4
      const int row = i / (DC / 4), c = (i - row * (DC / 4)) << 2;
      const int t = kt * 32 + row;
      if (t < p.Lk) {
        const float4 v = *reinterpret_cast<const float4*>(p.k + kv_offset(p, b, t, p.k_bs, p.k_ts) + kh + c);
        const float f[4] = {v.x * sk, v.y * sk, v.z * sk, v.w * sk};
        af16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const _Float16 h = (_Float16)f[j];
          hi[j] = h;
          lo[j] = (_Float16)(f[j] - (float)h);
        }
        *reinterpret_cast<af16x4*>(Ks + row * QP + c) = hi;
        *reinterpret_cast<af16x4*>(Ks + row * QP + DC + c) = lo;
      }
    }
  }
  const int dv0 = slice * DVS;
  for (int i = tid; i < 16 * DVS; i += 256) {                // V^T: (key pair, dv) items, lane <-> dv (coalesced reads)
    const int pair = i / DVS, dv = i - pair * DVS;
    const int t = kt * 32 + pair * 2;
    float v[2] = {0.f, 0.f};
    if (dv0 + dv < p.Dv) {                                     // dv columns beyond Dv stay zero
      if (t < p.Lk) v[0] = p.v[kv_offset(p, b, t, p.v_bs, p.v_ts) + vh + dv0 + dv];
      if (t + 1 < p.Lk) v[1] = p.v[kv_offset(p, b, t + 1, p.v_bs, p.v_ts) + vh + dv0 + dv];
    }
    af16x2 hi, lo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float vs = v[j] * sv;
      const _Float16 h = (_Float16)vs;
      hi[j] = h;
      lo[j] = (_Float16)(vs - (float)h);
    }
    _Float16* d = Vt + dv * VP + vt_pos(pair * 2);
    *reinterpret_cast<af16x2*>(d) = hi;
    *reinterpret_cast<af16x2*>(d + 32) = lo;
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(out) + (((long)b * p.H + head) * gridDim.x + kt) * (PKK + p.nslices * PKV);
  if (slice == 0)
    for (int i = tid; i < PKK; i += 256) dst[i] = reinterpret_cast<const uint4*>(img)[i];
  for (int i = tid; i < PKV; i += 256) dst[PKK + slice * PKV + i] = reinterpret_cast<const uint4*>(img)[PKK + i];
}

// packed K / V^T path: worth it when several query blocks stream the same keys
static long attn_pack_bytes(const AttnP& p, int mma) {
  if (mma != KEEP_MMA_X3 || (p.D != 128 && p.D != 256) || p.Lq < 256 || (p.flags & KEEP_ATTN_NO_PACK)) return 0;
  const int dvs = p.Dv <= 32 ? 32 : (p.Dv <= 64 ? 64 : 128);
  const int nsl = (p.Dv + dvs - 1) / dvs;
  return (long)p.B * p.H * ((p.Lk + 31) / 32) * ((32 * (2 * p.D + 8) + nsl * dvs * 72) * 2L);
}

template <int DVT, int NQ>
static int launch_attn_x3_packed(const AttnP& p, hipStream_t st) {
  constexpr int DC = 16 * NQ;
  const int ntl = (p.Lk + 31) / 32;
  hipLaunchKernelGGL((attn_pack_kv_x3_kernel<DC, DVT * 32>), dim3(ntl, p.H * p.nslices, p.B), dim3(256), 0, st, p,
                     const_cast<_Float16*>(p.kv_pack));
  KEEP_LAUNCH_CHECK("keep_attention(x3 pack)");
  size_t lds = (size_t)(32 * (2 * DC + 8) + DVT * 32 * 72) * 2;
  if (p.mode == 2 && p.Lk <= 4096) lds += (size_t)ntl * (32 * 4 + 16 + 32);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_x3_kernel<4, DVT, NQ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, 128), p.H * p.nslices, p.B);
#ifdef KEEP_X3_ABLATE
  const_cast<AttnP&>(p).abl = KEEP_DEV_ENV("KEEP_ATTN_EXP") ? atoi(KEEP_DEV_ENV("KEEP_ATTN_EXP")) : 0;
#endif
  hipLaunchKernelGGL((attn_x3_kernel<4, DVT, NQ, true>), grid, dim3(256), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(x3 packed)");
  return KEEP_OK;
}

template <int WAVES, int DVT, int NQ>
static int launch_attn_x3_t(const AttnP& p, hipStream_t st) {
  constexpr bool QREG = NQ > 0;
  const int DCMAX = QREG ? 16 * NQ : 128;
  const int DC = p.D < DCMAX ? p.D : DCMAX;
  size_t lds = (size_t)(((QREG ? 0 : WAVES * 32) + 32) * (2 * DC + 8) + DVT * 32 * 72) * 2;
  if (WAVES == 4 && QREG && p.mode == 2 && p.Lk <= 4096) {      // window tables: pixel per key, packed + byte region ids
    const size_t ntl = (size_t)(p.Lk + 31) / 32;
    lds += ntl * 32 * 4 + ntl * 16 + ntl * 32;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_x3_kernel<WAVES, DVT, NQ>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, WAVES * 32), p.H * p.nslices, p.B);
#ifdef KEEP_X3_ABLATE
  const_cast<AttnP&>(p).abl = KEEP_DEV_ENV("KEEP_ATTN_EXP") ? atoi(KEEP_DEV_ENV("KEEP_ATTN_EXP")) : 0;
#endif
  hipLaunchKernelGGL((attn_x3_kernel<WAVES, DVT, NQ>), grid, dim3(64 * WAVES), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(x3)");
  return KEEP_OK;
}

// D > 256 with at most 256 keys (the VQGAN AttnBlock: 256 tokens, one head of d = 512, VQ:226-239): the whole score row of
// a query fits in registers (8 key tiles x 16 values per lane), so the loops are turned inside out -- D chunks OUTSIDE, key
// tiles inside, S^T accumulated for all tiles -- and the 128-query Q chunk is staged once per chunk instead of once per
// (key tile, chunk): 3.3x less staging traffic than the generic chunked path, plain (not online) softmax, then P.V tile by
// tile.  Mode 0 only.
template <int DVT>
__global__ __launch_bounds__(256, 1) void attn_x3_sfull_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smemx[];
  constexpr int DVS = DVT * 32, NT = 256, VP = 72, DC = 128, QP = 2 * DC + 8, KT2 = 64;
  _Float16* Qs = smemx;                               // [128][QP]
  _Float16* Ks = Qs + 128 * QP;                       // [64][QP]: two key tiles per barrier pair
  _Float16* Vt = Ks + KT2 * QP;                       // [DVS][VP]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z;
  const int head = blockIdx.y / p.nslices;
  const int dv0 = (blockIdx.y - head * p.nslices) * DVS;
  const int q0 = blockIdx.x * 128;
  const long qh = (long)head * p.q_hs, kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  const int nch = p.D / DC;
  const int ntiles = (p.Lk + 31) / 32;                // <= 8
  float sq = 1.f, sk = 1.f, sv = 1.f, inv_qk = 1.f, inv_sv = 1.f;
  if (p.q_amax) {
    float iq, ik;
    attn_range_scale(p.q_amax[b], sq, iq);
    attn_range_scale(p.k_amax[b], sk, ik);
    attn_range_scale(p.v_amax[b], sv, inv_sv);
    inv_qk = iq * ik;
  }
  auto stage_rows = [&](const float* base, long bs, long ts, long hoff, int t0, int tmax, int nrows, int c0, float sc, _Float16* dst) {
    for (int i = tid; i < nrows * (DC / 4); i += NT) {
      const int row = i / (DC / 4), c = (i - row * (DC / 4)) << 2;
      const int t = t0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < tmax) v = *reinterpret_cast<const float4*>(base + (long)b * bs + (long)t * ts + hoff + c0 + c);
      const float f[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
      af16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)f[j];
        hi[j] = h;
        lo[j] = (_Float16)(f[j] - (float)h);
      }
      *reinterpret_cast<af16x4*>(dst + row * QP + c) = hi;
      *reinterpret_cast<af16x4*>(dst + row * QP + c + DC) = lo;
    }
  };
  f32x16 s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[k][r] = 0.f;
  const _Float16* qp = Qs + (wave * 32 + l31) * QP + lhi * 8;
  for (int ch = 0; ch < nch; ++ch) {
    __syncthreads();                                   // previous chunk fully consumed
    stage_rows(p.q, p.q_bs, p.q_ts, qh, q0, p.Lq, 128, ch * DC, sq, Qs);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      if (k2 * 2 < ntiles) {
        if (k2 > 0) __syncthreads();                   // the previous pair of key tiles is consumed
        stage_rows(p.k, p.k_bs, p.k_ts, kh, k2 * KT2, p.Lk, KT2, ch * DC, sk, Ks);
        __syncthreads();
#pragma unroll
        for (int d = 0; d < DC; d += 16) {
          const af16x8 qh8 = *reinterpret_cast<const af16x8*>(qp + d), ql8 = *reinterpret_cast<const af16x8*>(qp + DC + d);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const _Float16* kp = Ks + (t * 32 + l31) * QP + lhi * 8 + d;
            const af16x8 kh8 = *reinterpret_cast<const af16x8*>(kp), kl8 = *reinterpret_cast<const af16x8*>(kp + DC);
            s[k2 * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh8, s[k2 * 2 + t], 0, 0, 0);
            s[k2 * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql8, s[k2 * 2 + t], 0, 0, 0);
            s[k2 * 2 + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh8, s[k2 * 2 + t], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- softmax over the whole key row of this lane's query (exact fp32)
  const float qk_scale = p.scale * inv_qk;
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float val = key < p.Lk ? s[k][r] * qk_scale : -INFINITY;
      s[k][r] = val;
      m = fmaxf(m, val);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float lsum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[k][r] - m);
      s[k][r] = pv;
      lsum += pv;
    }
  lsum += __shfl_xor(lsum, 32);
  // ---- O = P . V, one 32-key tile at a time (V^T key-permuted like attn_bf16_kernel)
  f32x16 o[DVT];
#pragma unroll
  for (int j = 0; j < DVT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    if (kt < ntiles) {
      __syncthreads();
      for (int i = tid; i < 16 * DVS; i += NT) {
        const int pair = i / DVS, dv = i - pair * DVS;
        const int t = kt * 32 + pair * 2;
        float v[2] = {0.f, 0.f};
        if (dv0 + dv < p.Dv) {
          if (t < p.Lk) v[0] = p.v[(long)b * p.v_bs + (long)t * p.v_ts + vh + dv0 + dv];
          if (t + 1 < p.Lk) v[1] = p.v[(long)b * p.v_bs + (long)(t + 1) * p.v_ts + vh + dv0 + dv];
        }
        af16x2 hi, lo;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float vs = v[j] * sv;
          const _Float16 h = (_Float16)vs;
          hi[j] = h;
          lo[j] = (_Float16)(vs - (float)h);
        }
        _Float16* d = Vt + dv * VP + vt_pos(pair * 2);
        *reinterpret_cast<af16x2*>(d) = hi;
        *reinterpret_cast<af16x2*>(d + 32) = lo;
      }
      __syncthreads();
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        af16x8 ph, pl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const _Float16 h = (_Float16)s[kt][st * 8 + j];
          ph[j] = h;
          pl[j] = (_Float16)(s[kt][st * 8 + j] - (float)h);
        }
#pragma unroll
        for (int j = 0; j < DVT; ++j) {
          const _Float16* vrow = Vt + (j * 32 + l31) * VP + (st * 2 + lhi) * 8;
          const af16x8 vh8 = *reinterpret_cast<const af16x8*>(vrow), vl8 = *reinterpret_cast<const af16x8*>(vrow + 32);
          o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh8, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl8, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh8, o[j], 0, 0, 0);
        }
      }
    }
  }
  const float inv_l = inv_sv / lsum;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float il = __shfl(inv_l, qrow);
    const int t = q0 + wave * 32 + qrow;
    if (t < p.Lq) {
      const long base = (long)b * p.o_bs + (long)t * p.o_ts + (long)head * p.o_hs;
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const int dv = dv0 + j * 32 + l31;
        if (dv < p.Dv) p.o[base + dv] = o[j][r] * il;
      }
    }
  }
}

// The same case (D = 512 = 4 x 128, at most 256 keys, mode 0: the VQGAN AttnBlock on a 16 x 16 map) cut for LATENCY and block
// count: a block owns 32 queries (not 128) and ALL of Dv, so a launch has Lq/32 x H x B blocks (8 per image instead of
// 2 x 4 dv slices that each recomputed Q.K^T) and no product is computed twice:
//   phase 1  wave w multiplies ITS 128-channel chunk of Q (32 x 128) with the same chunk of every key tile -- operands staged
//            wave-privately (no block barrier in the loop) -- into 8 partial score tiles S_w^T [32 keys x 32 queries];
//   phase 2  the four partial sums of a tile are exchanged through LDS and added in wave order 0..3 by every wave (fixed
//            order: deterministic, independent of the batch) -- each wave then holds the full score rows of the 32 queries;
//   phase 3  exact fp32 softmax over the 256 keys (every wave, redundantly: 128 values per lane);
//   phase 4  wave w multiplies P with ITS 128-column slice of V (staged wave-privately, key-permuted like attn_bf16_kernel).
// 16^2 AttnBlock, B = 1: 140 us -> see DESIGN.md 5.1; the old kernel stays for the shapes this one does not take.
__global__ __launch_bounds__(256, 1) void attn_x3_sfull2_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) _Float16 smemx[];
  constexpr int DC = 128, QP = 2 * DC + 8, VP = 72, DVS = 128;
  constexpr int WREG = 2 * 32 * QP;                    // halfs per wave: Q chunk rows + one key tile's rows (>= DVS * VP for phase 4)
  static_assert(WREG >= DVS * VP, "V^T tile fits the wave's staging region");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 32;
  _Float16* Qw = smemx + wave * WREG;
  _Float16* Kw = Qw + 32 * QP;
  _Float16* Vt = Qw;                                   // phase 4 reuses the wave's region
  float* red = reinterpret_cast<float*>(smemx + 4 * WREG);      // [4 waves][32 x 32]
  const long qh = (long)head * p.q_hs, kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  const int ntiles = (p.Lk + 31) / 32;                 // <= 8
  float sq = 1.f, sk = 1.f, sv = 1.f, inv_qk = 1.f, inv_sv = 1.f;
  if (p.q_amax) {
    float iq, ik;
    attn_range_scale(p.q_amax[b], sq, iq);
    attn_range_scale(p.k_amax[b], sk, ik);
    attn_range_scale(p.v_amax[b], sv, inv_sv);
    inv_qk = iq * ik;
  }
  // 32 rows x 128 channels (this wave's chunk) of q or k by the 64 lanes of ONE wave: 16 float4 per lane, all loads of a tile
  // issued back to back into registers (the next tile's while the current one is on the matrix cores), then split into
  // (hi | lo) rows
  float4 buf[16];
  auto load_rows = [&](const float* base, long bs, long ts, long hoff, int t0, int tmax) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = lane + u * 64;
      const int row = i >> 5, c = (i & 31) << 2;
      const int t = t0 + row;
      buf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < tmax) buf[u] = *reinterpret_cast<const float4*>(base + (long)b * bs + (long)t * ts + hoff + wave * DC + c);
    }
  };
  auto split_rows = [&](float sc, _Float16* dst) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = lane + u * 64;
      const int row = i >> 5, c = (i & 31) << 2;
      const float f[4] = {buf[u].x * sc, buf[u].y * sc, buf[u].z * sc, buf[u].w * sc};
      af16x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const _Float16 h = (_Float16)f[j];
        hi[j] = h;
        lo[j] = (_Float16)(f[j] - (float)h);
      }
      *reinterpret_cast<af16x4*>(dst + row * QP + c) = hi;
      *reinterpret_cast<af16x4*>(dst + row * QP + c + DC) = lo;
    }
  };
  f32x16 s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[k][r] = 0.f;
  // ---- phase 1: partial S^T over this wave's channel chunk
  load_rows(p.q, p.q_bs, p.q_ts, qh, q0, p.Lq);
  split_rows(sq, Qw);
  load_rows(p.k, p.k_bs, p.k_ts, kh, 0, p.Lk);
  const _Float16* qp = Qw + l31 * QP + lhi * 8;
  const _Float16* kp = Kw + l31 * QP + lhi * 8;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    if (kt < ntiles) {
      split_rows(sk, Kw);
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): wave-private hand-off (LDS ops of one wave retire in order)
      if (kt + 1 < ntiles) load_rows(p.k, p.k_bs, p.k_ts, kh, (kt + 1) * 32, p.Lk);      // in flight during the MFMAs below
#pragma unroll
      for (int d = 0; d < DC; d += 16) {
        const af16x8 qh8 = *reinterpret_cast<const af16x8*>(qp + d), ql8 = *reinterpret_cast<const af16x8*>(qp + DC + d);
        const af16x8 kh8 = *reinterpret_cast<const af16x8*>(kp + d), kl8 = *reinterpret_cast<const af16x8*>(kp + DC + d);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh8, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql8, s[kt], 0, 0, 0);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh8, s[kt], 0, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);              // fragment reads done before the next tile overwrites Kw
    }
  }
  // ---- phase 2: full scores = sum over the four channel chunks, in wave order
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    if (kt < ntiles) {
      __syncthreads();                                 // the previous tile's partials are consumed
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31] = s[kt][r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = ((r & 3) + 8 * (r >> 2) + 4 * lhi) * 32 + l31;
        s[kt][r] = ((red[o] + red[1024 + o]) + red[2048 + o]) + red[3072 + o];
      }
    }
  }
  // ---- phase 3: softmax over the whole key row of this lane's query (exact fp32)
  const float qk_scale = p.scale * inv_qk;
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float val = key < p.Lk ? s[k][r] * qk_scale : -INFINITY;
      s[k][r] = val;
      m = fmaxf(m, val);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float lsum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[k][r] - m);
      s[k][r] = pv;
      lsum += pv;
    }
  lsum += __shfl_xor(lsum, 32);
  // ---- phase 4: O[:, dv slice of this wave] = P . V
  const int dv0 = wave * DVS;
  f32x16 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  if (dv0 < p.Dv) {
    // V tile = 32 keys x 128 dv of this wave's slice: 8 (key pair, 4-dv group) items per lane, 2 float4 each, reusing buf[]
    auto load_v = [&](int kt) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = lane + u * 64;
        const int pair = i >> 5, dv = (i & 31) << 2;
        const int t = kt * 32 + pair * 2;
        buf[2 * u] = make_float4(0.f, 0.f, 0.f, 0.f);
        buf[2 * u + 1] = buf[2 * u];
        if (dv0 + dv < p.Dv) {
          if (t < p.Lk) buf[2 * u] = *reinterpret_cast<const float4*>(p.v + (long)b * p.v_bs + (long)t * p.v_ts + vh + dv0 + dv);
          if (t + 1 < p.Lk) buf[2 * u + 1] = *reinterpret_cast<const float4*>(p.v + (long)b * p.v_bs + (long)(t + 1) * p.v_ts + vh + dv0 + dv);
        }
      }
    };
    load_v(0);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      if (kt < ntiles) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = lane + u * 64;
          const int pair = i >> 5, dv = (i & 31) << 2;
          const float4 va = buf[2 * u], vb = buf[2 * u + 1];
          const float fa[4] = {va.x * sv, va.y * sv, va.z * sv, va.w * sv}, fb[4] = {vb.x * sv, vb.y * sv, vb.z * sv, vb.w * sv};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            af16x2 hi, lo;
            const _Float16 ha = (_Float16)fa[j], hb = (_Float16)fb[j];
            hi[0] = ha; hi[1] = hb;
            lo[0] = (_Float16)(fa[j] - (float)ha); lo[1] = (_Float16)(fb[j] - (float)hb);
            _Float16* dd = Vt + (dv + j) * VP + vt_pos(pair * 2);     // vt_pos(k0+1) = vt_pos(k0) + 1 for even k0
            *reinterpret_cast<af16x2*>(dd) = hi;
            *reinterpret_cast<af16x2*>(dd + 32) = lo;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (kt + 1 < ntiles) load_v(kt + 1);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          af16x8 ph, pl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const _Float16 h = (_Float16)s[kt][st * 8 + j];
            ph[j] = h;
            pl[j] = (_Float16)(s[kt][st * 8 + j] - (float)h);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const _Float16* vrow = Vt + (j * 32 + l31) * VP + (st * 2 + lhi) * 8;
            const af16x8 vh8 = *reinterpret_cast<const af16x8*>(vrow), vl8 = *reinterpret_cast<const af16x8*>(vrow + 32);
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh8, o[j], 0, 0, 0);
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl8, o[j], 0, 0, 0);
            o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh8, o[j], 0, 0, 0);
          }
        }
      }
    }
  }
  const float inv_l = inv_sv / lsum;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float il = __shfl(inv_l, qrow);
    const int t = q0 + qrow;
    if (t < p.Lq && dv0 < p.Dv) {
      const long base = (long)b * p.o_bs + (long)t * p.o_ts + (long)head * p.o_hs;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int dv = dv0 + j * 32 + l31;
        if (dv < p.Dv) p.o[base + dv] = o[j][r] * il;
      }
    }
  }
}

static bool attn_sfull2_ok(const AttnP& p) {
  return p.mode == 0 && p.D == 512 && p.Lk <= 256 && p.Dv <= 512 && p.Dv % 4 == 0 && p.v_ts % 4 == 0 && p.v_bs % 4 == 0 &&
         p.v_hs % 4 == 0 && (uintptr_t)p.v % 16 == 0 && !(p.flags & KEEP_ATTN_NO_SFULL2);
}

static int launch_attn_x3_sfull2(const AttnP& p, hipStream_t st) {
  const size_t lds = (size_t)4 * (2 * 32 * (2 * 128 + 8)) * 2 + 4 * 1024 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_x3_sfull2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_x3_sfull2_kernel, dim3(cdiv(p.Lq, 32), p.H, p.B), dim3(256), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(x3, full score row, 32-query blocks)");
  return KEEP_OK;
}

template <int DVT>
static int launch_attn_x3_sfull(const AttnP& p, hipStream_t st) {
  const size_t lds = (size_t)((128 + 64) * (2 * 128 + 8) + DVT * 32 * 72) * 2;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_x3_sfull_kernel<DVT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, 128), p.H * p.nslices, p.B);
  hipLaunchKernelGGL((attn_x3_sfull_kernel<DVT>), grid, dim3(256), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(x3, full score row)");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ x3, D = Dv = 512, two passes
// The VQGAN AttnBlock on a 16 x 16 map (VQ:219-243: one head of d = 512 over 256 tokens) as TWO launches in the latency form of
// keep_gemm_x3l.hip -- every wave owns a K slice, requests its whole operand slices up front, the block adds the partial tiles in wave
// order through LDS:
//   attn_scores_x3l_kernel   S[q][key] = scale * sum_d Q K : 32 x 32 tiles, four waves x 128 channels; BOTH operands are fp32 rows split
//                            in registers (lane = token, 8 consecutive channels: the MFMA fragment layout straight from global memory);
//   attn_pv_x3l_kernel       O = softmax(S) V : 32 queries x 32 dv tiles, four waves x 64 keys; the row maximum / sum of the block's
//                            32 queries is agreed through LDS (every block sees whole score rows), P = exp(S - m) is split in
//                            registers as the A operand, V is the B operand read as a GATHER (lane = dv column, one dword per key:
//                            32 consecutive dv of one key per half-wave = whole 128-byte lines, no transpose pass).
// 64 + 128 blocks per image instead of attn_x3_sfull2_kernel's 8 (each walking 8 key tiles twice): 77 -> ~16 us with one clip in
// flight.  One decomposition per image at every batch size: bits never depend on batch-mates.
__global__ __launch_bounds__(256) void attn_scores_x3l_kernel(AttnP p, float* __restrict__ sc_out) {
  __shared__ __attribute__((aligned(16))) float part[4 * 32 * 36];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int key0 = blockIdx.x * 32, q0 = blockIdx.y * 32, bh = blockIdx.z, b = bh / p.H, head = bh - b * p.H;
  const int c0 = wave * 128 + lhi * 8;
  const float* qp = p.q + (long)b * p.q_bs + (long)head * p.q_hs + (long)(q0 + l31) * p.q_ts + c0;
  const float* kp = p.k + (long)b * p.k_bs + (long)head * p.k_hs + (long)(key0 + l31) * p.k_ts + c0;
  float4 qa[8][2], ka[8][2];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    qa[ks][0] = *reinterpret_cast<const float4*>(qp + ks * 16);
    qa[ks][1] = *reinterpret_cast<const float4*>(qp + ks * 16 + 4);
    ka[ks][0] = *reinterpret_cast<const float4*>(kp + ks * 16);
    ka[ks][1] = *reinterpret_cast<const float4*>(kp + ks * 16 + 4);
  }
  __builtin_amdgcn_sched_barrier(0);      // every load is in flight before the first split (the scheduler would interleave them to save registers)
  auto split8 = [&](const float4 a, const float4 c, af16x8& hi, af16x8& lo) {
    const float f[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)f[j];
      hi[j] = h;
      lo[j] = (_Float16)(f[j] - (float)h);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    af16x8 qh8, ql8, kh8, kl8;
    split8(qa[ks][0], qa[ks][1], qh8, ql8);
    split8(ka[ks][0], ka[ks][1], kh8, kl8);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql8, kh8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh8, kl8, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh8, kh8, acc, 0, 0, 0);
  }
  float* mine = part + wave * 32 * 36;
#pragma unroll
  for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[r];
  __syncthreads();
  const int row = tid >> 3, c4 = (tid & 7) * 4;
  float4 v = *reinterpret_cast<const float4*>(part + row * 36 + c4);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 t = *reinterpret_cast<const float4*>(part + (w * 32 + row) * 36 + c4);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  *reinterpret_cast<float4*>(sc_out + ((long)bh * p.Lq + q0 + row) * p.Lk + key0 + c4) =
      make_float4(v.x * p.scale, v.y * p.scale, v.z * p.scale, v.w * p.scale);
}

__global__ __launch_bounds__(256) void attn_pv_x3l_kernel(AttnP p, const float* __restrict__ sc_in) {
  __shared__ __attribute__((aligned(16))) float part[4 * 32 * 36];
  __shared__ float red_m[4 * 32], red_l[4 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int dv0 = blockIdx.x * 32, q0 = blockIdx.y * 32, bh = blockIdx.z, b = bh / p.H, head = bh - b * p.H;
  const int k0 = wave * 64 + lhi * 8;
  const float* sp = sc_in + ((long)bh * p.Lq + q0 + l31) * p.Lk + k0;
  const float* vp = p.v + (long)b * p.v_bs + (long)head * p.v_hs + (long)k0 * p.v_ts + dv0 + l31;
  float4 sa[4][2];
  float vb[4][8];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    sa[ks][0] = *reinterpret_cast<const float4*>(sp + ks * 16);
    sa[ks][1] = *reinterpret_cast<const float4*>(sp + ks * 16 + 4);
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 8; ++j) vb[ks][j] = vp[(long)(ks * 16 + j) * p.v_ts];
  __builtin_amdgcn_sched_barrier(0);
  // row maximum over all 256 keys: this lane's 32 values, its half-wave twin, the other three waves
  float sv[4][8];
  float m = -INFINITY;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    sv[ks][0] = sa[ks][0].x; sv[ks][1] = sa[ks][0].y; sv[ks][2] = sa[ks][0].z; sv[ks][3] = sa[ks][0].w;
    sv[ks][4] = sa[ks][1].x; sv[ks][5] = sa[ks][1].y; sv[ks][6] = sa[ks][1].z; sv[ks][7] = sa[ks][1].w;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, sv[ks][j]);
  }
  m = fmaxf(m, __shfl_xor(m, 32));
  if (lane < 32) red_m[wave * 32 + l31] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_m[l31], red_m[32 + l31]), fmaxf(red_m[64 + l31], red_m[96 + l31]));
  float lsum = 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    af16x8 ph, pl, vh, vl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pv = expf(sv[ks][j] - m);
      lsum += pv;
      const _Float16 h = (_Float16)pv;
      ph[j] = h;
      pl[j] = (_Float16)(pv - (float)h);
      const _Float16 g = (_Float16)vb[ks][j];
      vh[j] = g;
      vl[j] = (_Float16)(vb[ks][j] - (float)g);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, acc, 0, 0, 0);
  }
  lsum += __shfl_xor(lsum, 32);
  if (lane < 32) red_l[wave * 32 + l31] = lsum;
  float* mine = part + wave * 32 * 36;
#pragma unroll
  for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 36 + l31] = acc[r];
  __syncthreads();
  const int row = tid >> 3, c4 = (tid & 7) * 4;
  float4 v = *reinterpret_cast<const float4*>(part + row * 36 + c4);
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const float4 t = *reinterpret_cast<const float4*>(part + (w * 32 + row) * 36 + c4);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  const float il = 1.0f / (((red_l[row] + red_l[32 + row]) + red_l[64 + row]) + red_l[96 + row]);
  *reinterpret_cast<float4*>(p.o + (long)b * p.o_bs + (long)head * p.o_hs + (long)(q0 + row) * p.o_ts + dv0 + c4) =
      make_float4(v.x * il, v.y * il, v.z * il, v.w * il);
}

static bool attn_two_pass_ok(const AttnP& p) {
  return p.mode == 0 && p.D == 512 && p.Dv == 512 && p.Lq == 256 && p.Lk == 256 && !p.q_amax && p.o_ts % 4 == 0 && p.o_bs % 4 == 0 &&
         p.o_hs % 4 == 0 && (uintptr_t)p.o % 16 == 0 && !(p.flags & KEEP_ATTN_NO_TWO_PASS);
}
static long attn_two_pass_bytes(const AttnP& p) { return (long)p.B * p.H * p.Lq * p.Lk * 4; }

static int launch_attn_x3_two_pass(const AttnP& p, float* scores, hipStream_t st) {
  hipLaunchKernelGGL(attn_scores_x3l_kernel, dim3(p.Lk / 32, p.Lq / 32, p.B * p.H), dim3(256), 0, st, p, scores);
  KEEP_LAUNCH_CHECK("keep_attention(x3, two passes: scores)");
  hipLaunchKernelGGL(attn_pv_x3l_kernel, dim3(p.Dv / 32, p.Lq / 32, p.B * p.H), dim3(256), 0, st, p, (const float*)scores);
  KEEP_LAUNCH_CHECK("keep_attention(x3, two passes: softmax . V)");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ x3, small heads, latency form
// Multi-head attention over at most 256 keys with D = Dv = 64 (the code transformer's nn.MultiheadAttention, KA:385-439: 8 heads x 256
// tokens): attn_x3_kernel walks the 8 key tiles of a (head, 128-query) block one after the other (stage -> barrier -> MFMA -> barrier:
// 40 us for 0.13 GFLOP, 16 blocks per image).  Here a block owns 32 queries of one head and its four waves own 64 KEYS each: a wave
// requests its Q, K and V rows up front (Q / K straight into the MFMA fragment layout: lane = token, 8 consecutive d), computes its
// partial S^T = K.Q^T (two 32 x 32 tiles), the block agrees on the row maximum through LDS (one barrier), every wave exponentiates,
// multiplies its P with its V slice (V^T staged wave-privately, key-permuted like attn_bf16_kernel) and parks (O_w, l_w); after the
// second barrier O = (O_0 + O_1 + O_2 + O_3) / (l_0 + l_1 + l_2 + l_3), waves in order.  Plain (not online) softmax with the global
// row maximum: exact fp32, deterministic, one decomposition at every batch size (per-image rule) -- bits never depend on batch-mates.
__global__ __launch_bounds__(256) void attn_x3_small_kernel(AttnP p) {
  constexpr int DV = 64, VP = 72, OP = 68;
  __shared__ __attribute__((aligned(16))) _Float16 vt_s[4 * DV * VP];       // per wave: V^T tile [64 dv][32 hi | 32 lo | pad]; later O_w [32 q][68]
  __shared__ float red_m[4 * 32], red_l[4 * 32];
  static_assert(32 * OP * 4 <= DV * VP * 2, "the wave's O tile fits its V^T region");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {      // the query blocks of one (batch, head) share K / V: neighbours in one XCD's L2 (see attn_x3_kernel)
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * gridDim.z;
    const int lid = bx + gx * (by + gy * bz);
    const int qd = total >> 3, rm = total & 7, xcd = lid & 7, slot = lid >> 3;
    const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
    bx = logical % gx;
    const int rest = logical / gx;
    by = rest % gy;
    bz = rest / gy;
  }
  const int b = bz, head = by, q0 = bx * 32, key0 = wave * 64;
  float sq = 1.f, sk = 1.f, sv = 1.f, inv_qk = 1.f, inv_sv = 1.f;
  if (p.q_amax) {
    float iq, ik;
    attn_range_scale(p.q_amax[b], sq, iq);
    attn_range_scale(p.k_amax[b], sk, ik);
    attn_range_scale(p.v_amax[b], sv, inv_sv);
    inv_qk = iq * ik;
  }
  // ---- every load of the wave, issued back to back
  float4 qraw[4][2], kraw[2][4][2], vraw[2][4][2];
  {
    const int tq = q0 + l31;
    const float* qp = p.q + (long)b * p.q_bs + (long)tq * p.q_ts + (long)head * p.q_hs + lhi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qraw[ks][0] = qraw[ks][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tq < p.Lq) {
        qraw[ks][0] = *reinterpret_cast<const float4*>(qp + ks * 16);
        qraw[ks][1] = *reinterpret_cast<const float4*>(qp + ks * 16 + 4);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int tk = key0 + kt * 32 + l31;
      const float* kp = p.k + (long)b * p.k_bs + (long)tk * p.k_ts + (long)head * p.k_hs + lhi * 8;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        kraw[kt][ks][0] = kraw[kt][ks][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tk < p.Lk) {
          kraw[kt][ks][0] = *reinterpret_cast<const float4*>(kp + ks * 16);
          kraw[kt][ks][1] = *reinterpret_cast<const float4*>(kp + ks * 16 + 4);
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // (key pair, 4-dv group) items: 16 pairs x 16 groups per 32-key tile
        const int i = lane + u * 64;
        const int pair = i >> 4, dv = (i & 15) << 2;
        const int t = key0 + kt * 32 + pair * 2;
        const float* vp = p.v + (long)b * p.v_bs + (long)t * p.v_ts + (long)head * p.v_hs + dv;
        vraw[kt][u][0] = vraw[kt][u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < p.Lk) vraw[kt][u][0] = *reinterpret_cast<const float4*>(vp);
        if (t + 1 < p.Lk) vraw[kt][u][1] = *reinterpret_cast<const float4*>(vp + p.v_ts);
      }
  }
  auto split8 = [&](const float4 a, const float4 c, float sc, af16x8& hi, af16x8& lo) {
    const float f[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, c.x * sc, c.y * sc, c.z * sc, c.w * sc};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const _Float16 h = (_Float16)f[j];
      hi[j] = h;
      lo[j] = (_Float16)(f[j] - (float)h);
    }
  };
  // ---- partial scores S^T[key][query] of this wave's 64 keys
  f32x16 s[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    af16x8 qh8, ql8;
    split8(qraw[ks][0], qraw[ks][1], sq, qh8, ql8);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      af16x8 kh8, kl8;
      split8(kraw[kt][ks][0], kraw[kt][ks][1], sk, kh8, kl8);
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl8, qh8, s[kt], 0, 0, 0);
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, ql8, s[kt], 0, 0, 0);
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh8, qh8, s[kt], 0, 0, 0);
    }
  }
  const float qk_scale = p.scale * inv_qk;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float val = key < p.Lk ? s[kt][r] * qk_scale : -INFINITY;
      s[kt][r] = val;
      m = fmaxf(m, val);
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  if (lane < 32) red_m[wave * 32 + l31] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red_m[l31], red_m[32 + l31]), fmaxf(red_m[64 + l31], red_m[96 + l31]));
  float lsum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(s[kt][r] - m);
      s[kt][r] = pv;
      lsum += pv;
    }
  lsum += __shfl_xor(lsum, 32);
  if (lane < 32) red_l[wave * 32 + l31] = lsum;
  // ---- O_w = P_w . V_w
  _Float16* Vt = vt_s + wave * DV * VP;
  f32x16 o[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt) {
    if (kt) __builtin_amdgcn_s_waitcnt(0xc07f);      // the previous tile's fragment reads are done (wave-private region)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = lane + u * 64;
      const int pair = i >> 4, dv = (i & 15) << 2;
      const float4 va = vraw[kt][u][0], vb = vraw[kt][u][1];
      const float fa[4] = {va.x * sv, va.y * sv, va.z * sv, va.w * sv}, fb[4] = {vb.x * sv, vb.y * sv, vb.z * sv, vb.w * sv};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        af16x2 hi, lo;
        const _Float16 ha = (_Float16)fa[j], hb = (_Float16)fb[j];
        hi[0] = ha; hi[1] = hb;
        lo[0] = (_Float16)(fa[j] - (float)ha); lo[1] = (_Float16)(fb[j] - (float)hb);
        _Float16* dd = Vt + (dv + j) * VP + vt_pos(pair * 2);     // vt_pos(k0 + 1) = vt_pos(k0) + 1 for even k0
        *reinterpret_cast<af16x2*>(dd) = hi;
        *reinterpret_cast<af16x2*>(dd + 32) = lo;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      af16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)s[kt][st * 8 + j];
        ph[j] = h;
        pl[j] = (_Float16)(s[kt][st * 8 + j] - (float)h);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const _Float16* vrow = Vt + (j * 32 + l31) * VP + (st * 2 + lhi) * 8;
        const af16x8 vh8 = *reinterpret_cast<const af16x8*>(vrow), vl8 = *reinterpret_cast<const af16x8*>(vrow + 32);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh8, o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl8, o[j], 0, 0, 0);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh8, o[j], 0, 0, 0);
      }
    }
  }
  // ---- park O_w [32 queries][64 dv] over the wave's V^T region, then the wave-ordered sum
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float* ow = reinterpret_cast<float*>(Vt);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) ow[((r & 3) + 8 * (r >> 2) + 4 * lhi) * OP + j * 32 + l31] = o[j][r];
  __syncthreads();
  {
    const int qrow = tid >> 3, c8 = (tid & 7) * 8;
    const int t = q0 + qrow;
    float acc8[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float* src = reinterpret_cast<const float*>(vt_s + w * DV * VP) + qrow * OP + c8;
      const float4 a = *reinterpret_cast<const float4*>(src), c = *reinterpret_cast<const float4*>(src + 4);
      const float f[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) acc8[j] = w == 0 ? f[j] : acc8[j] + f[j];
    }
    const float lt = ((red_l[qrow] + red_l[32 + qrow]) + red_l[64 + qrow]) + red_l[96 + qrow];
    const float il = inv_sv / lt;
    if (t < p.Lq) {
      float* dst = p.o + (long)b * p.o_bs + (long)t * p.o_ts + (long)head * p.o_hs + c8;
      *reinterpret_cast<float4*>(dst) = make_float4(acc8[0] * il, acc8[1] * il, acc8[2] * il, acc8[3] * il);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(acc8[4] * il, acc8[5] * il, acc8[6] * il, acc8[7] * il);
    }
  }
}

static bool attn_small_ok(const AttnP& p) {
  return p.mode == 0 && p.D == 64 && p.Dv == 64 && p.Lk <= 256 && p.Lq >= 64 && p.v_ts % 4 == 0 && p.v_bs % 4 == 0 && p.v_hs % 4 == 0 &&
         (uintptr_t)p.v % 16 == 0 && p.o_ts % 4 == 0 && p.o_bs % 4 == 0 && p.o_hs % 4 == 0 && (uintptr_t)p.o % 16 == 0 &&
         !(p.flags & KEEP_ATTN_NO_SMALL);
}

static int launch_attn_x3_small(const AttnP& p, hipStream_t st) {
  hipLaunchKernelGGL(attn_x3_small_kernel, dim3(cdiv(p.Lq, 32), p.H, p.B), dim3(256), 0, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(x3, small heads, latency form)");
  return KEEP_OK;
}

template <int WAVES, int DVT>
static int launch_attn_x3(const AttnP& p, hipStream_t st) {
  if (p.D <= 128) return launch_attn_x3_t<WAVES, DVT, 8>(p, st);
  if (p.D <= 256) return launch_attn_x3_t<4, DVT, 16>(p, st);
  if (attn_sfull2_ok(p)) return launch_attn_x3_sfull2(p, st);
  if (p.mode == 0 && p.Lk <= 256 && p.D % 128 == 0) return launch_attn_x3_sfull<DVT>(p, st);
  return launch_attn_x3_t<4, DVT, 0>(p, st);
}

// ------------------------------------------------------------------------------------------------ bf16-input variant
// Q, K, V already live in HBM as bf16 (the projection GEMMs write bf16, halving the K/V traffic that every q-tile
// block re-reads).  128 queries per block (4 waves), 64 keys per iteration (two 32x32 score tiles per wave).
// Single D chunk (D <= 128, every caller but the VQGAN AttnBlock) -- the pipelined path:
//   * token offsets and the shifted-window region masks of a key tile are computed THREE tiles ahead by a rotating wave
//     into a 3-deep LDS ring (off the critical path; the masks are 64-bit ballots per region id, so a lane tests one
//     bit per score instead of reading 32 region ids);
//   * the K / V pieces of tile kt+2 are issued right after tile kt+1 was committed to LDS, so they have a whole
//     iteration (QK^T, softmax, PV of tile kt+1) to land; two barriers per tile;
//   * Q fragments stay in registers; softmax runs in the exp2 domain (scale * log2 e folded into one multiply).
// An ablation of the previous schedule on the GMFlow window shape (981 us) had shown 383 us of pure skeleton (four
// barriers + serial table math per tile), 160 us of exposed load latency and 205 us of softmax VALU.
// D > 128 (AttnBlock, D = 512): the chunked path -- tables per tile, Q/K re-staged per 128-wide chunk.
// V^T is staged key-permuted exactly like attn_bf16_kernel.
template <int DVT, bool MASKED>       // MASKED: shifted windows (mode 2, shift > 0) -- the only caller of the region mask
__global__ __launch_bounds__(256, 2) void attn_bf16in_kernel(AttnP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int DVS = DVT * 32;
  constexpr int NT = 256, KT = 64;
  constexpr int VP = KT + 8;             // V^T row pitch (bf16): 144 B = 9 slots
  constexpr int KPF = 4;                 // prefetched 16-byte K pieces per thread (64 rows x 16 groups / 256)
  constexpr int VPF = (DVS / 8 + 3) / 4; // prefetched 16-byte V pieces per thread (lane = key, 4 waves share dv groups)
  constexpr int NTAB = 3;                // table ring depth
  const int DC = p.D < 128 ? p.D : 128;
  const int nch = p.D / DC;
  const int QP = DC + 8;
  const int g8n = DC >> 3;
  long* qoff = reinterpret_cast<long*>(smem_raw);                                  // [128]
  long* koff = qoff + 128;                                                         // [NTAB][64]
  long* voff = koff + NTAB * KT;                                                   // [NTAB][64]
  unsigned long long* kmask = reinterpret_cast<unsigned long long*>(voff + NTAB * KT);   // [NTAB][16] region ballots
  __bf16* Qs = reinterpret_cast<__bf16*>(kmask + NTAB * 16);                       // [128][QP]
  __bf16* Ks = Qs + 128 * QP;                                                      // [64][QP]
  __bf16* Vt = Ks + KT * QP;                                                       // [DVS][VP]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.z;
  const int head = blockIdx.y / p.nslices;
  const int dv0 = (blockIdx.y - head * p.nslices) * DVS;
  const int q0 = blockIdx.x * 128;
  const long qh = (long)head * p.q_hs, kh = (long)head * p.k_hs, vh = (long)head * p.v_hs;
  constexpr bool use_mask = MASKED;
  WinCtx wc = {};
  if (p.mode == 2) wc = win_ctx(p, b);
  // element offset of query / key / value / output token t (window, sparse-causal or plain addressing) + region id
  auto tok_q = [&](int t, long bs, long ts) -> long {
    if (p.mode == 2) {
      int pix, reg;
      win_token(p, wc, t, pix, reg);
      return (long)wc.img * bs + (long)pix * ts;
    }
    return (long)b * bs + (long)t * ts;
  };

  if (tid < 128) {
    const int t = q0 + tid;
    qoff[tid] = (t < p.Lq) ? tok_q(t, p.q_bs, p.q_ts) + qh : -1;
  }
  // executed by ONE wave (all 64 lanes, lane <-> key of the tile)
  auto fill_tables = [&](int kt, int slot) {
    const int t = kt * KT + lane;
    const bool ok = t < p.Lk;
    int reg = 0;
    if (p.mode == 2) {
      int pix;
      win_token(p, wc, t, pix, reg);
      koff[slot * KT + lane] = ok ? (long)wc.img_kv * p.k_bs + (long)pix * p.k_ts + kh : -1;
      voff[slot * KT + lane] = ok ? (long)wc.img_kv * p.v_bs + (long)pix * p.v_ts + vh : -1;
    } else {
      koff[slot * KT + lane] = ok ? kv_offset(p, b, t, p.k_bs, p.k_ts) + kh : -1;
      voff[slot * KT + lane] = ok ? kv_offset(p, b, t, p.v_bs, p.v_ts) + vh : -1;
    }
    if (use_mask) {
      if (!ok) reg = 0;
#pragma unroll
      for (int R = 0; R < 9; ++R) {
        const unsigned long long m = __ballot(reg != R);
        if (lane == R) kmask[slot * 16 + R] = m;
      }
    }
  };
  auto stage_q = [&](int ch) {
    for (int i = tid; i < 128 * g8n; i += NT) {
      const int row = i / g8n, c = (i - row * g8n) << 3;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (qoff[row] >= 0) v = *reinterpret_cast<const uint4*>(p.q16 + qoff[row] + ch * DC + c);
      *reinterpret_cast<uint4*>(Qs + row * QP + c) = v;
    }
  };
  uint4 kpf[KPF], vpf[VPF];
  auto k_issue = [&](int ch, int slot) {
#pragma unroll
    for (int u = 0; u < KPF; ++u) {
      const int i = tid + u * NT;
      kpf[u] = make_uint4(0u, 0u, 0u, 0u);
      if (i < KT * g8n) {
        const int row = i / g8n, c = (i - row * g8n) << 3;
        const long off = koff[slot * KT + row];
        if (off >= 0) kpf[u] = *reinterpret_cast<const uint4*>(p.k16 + off + ch * DC + c);
      }
    }
  };
  auto k_commit = [&]() {
#pragma unroll
    for (int u = 0; u < KPF; ++u) {
      const int i = tid + u * NT;
      if (i < KT * g8n) {
        const int row = i / g8n, c = (i - row * g8n) << 3;
        *reinterpret_cast<uint4*>(Ks + row * QP + c) = kpf[u];
      }
    }
  };
  auto v_issue = [&](int slot) {    // lane <-> key, waves share the 8-column dv groups
    const long off = voff[slot * KT + lane];
#pragma unroll
    for (int u = 0; u < VPF; ++u) {
      const int dg = wave + u * 4;
      vpf[u] = make_uint4(0u, 0u, 0u, 0u);
      if (dg * 8 < DVS && dv0 + dg * 8 < p.Dv && off >= 0)
        vpf[u] = *reinterpret_cast<const uint4*>(p.v16 + off + dv0 + dg * 8);
    }
  };
  auto v_commit = [&]() {
    const int pos = (lane >> 5) * 32 + vt_pos(lane & 31);
#pragma unroll
    for (int u = 0; u < VPF; ++u) {
      const int dg = wave + u * 4;
      if (dg * 8 < DVS) {
        const unsigned w[4] = {vpf[u].x, vpf[u].y, vpf[u].z, vpf[u].w};
        unsigned short* dst = reinterpret_cast<unsigned short*>(Vt) + (dg * 8) * VP + pos;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j * VP] = (unsigned short)(w[j >> 1] >> ((j & 1) * 16));
      }
    }
  };

  const int my_q = q0 + wave * 32 + l31;
  int my_region = 0;
  if (use_mask && my_q < p.Lq) {
    int pix;
    win_token(p, wc, my_q, pix, my_region);
  }

  // scores live in the exp2 domain: s2 = s * scale * log2(e); the -100 of the window mask scales along
  const float scale2 = p.scale * 1.4426950408889634f;
  const float mask2 = -100.0f * 1.4426950408889634f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[DVT];
#pragma unroll
  for (int j = 0; j < DVT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;

  const __bf16* kp = Ks + l31 * QP + lhi * 8;
  const __bf16* qp = Qs + (wave * 32 + l31) * QP + lhi * 8;
  const int ntiles = (p.Lk + KT - 1) / KT;
  const bool pipelined = (nch == 1);

  // online softmax + P.V of one 64-key tile; `mbits`: bit (t*32 + (r&3) + 8*(r>>2)) set = key outside the lane's region
  // (already shifted right by 4*lhi)
  auto softmax_pv = [&](int kt, f32x16 (&s)[2], unsigned long long mbits) {
    const bool full = (kt + 1) * KT <= p.Lk;
    const unsigned mlo = (unsigned)mbits, mhi = (unsigned)(mbits >> 32);
    float mloc = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float val = s[t][r] * scale2;
        if (use_mask && (((t ? mhi : mlo) >> ((r & 3) + 8 * (r >> 2))) & 1u)) val += mask2;
        if (!full && kt * KT + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi >= p.Lk) val = -INFINITY;
        s[t][r] = val;
        mloc = fmaxf(mloc, val);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float lsum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(s[t][r] - m_new);
        s[t][r] = pv;
        lsum += pv;
      }
    lsum += __shfl_xor(lsum, 32);
    l_run = l_run * alpha + lsum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float ar = __shfl(alpha, qrow);
#pragma unroll
      for (int j = 0; j < DVT; ++j) o[j][r] *= ar;
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      abf16x8 pa;
#pragma unroll
      for (int j = 0; j < 8; ++j) pa[j] = (__bf16)s[st >> 1][(st & 1) * 8 + j];
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const abf16x8 vb =
            *reinterpret_cast<const abf16x8*>(Vt + (j * 32 + l31) * VP + (st >> 1) * 32 + ((st & 1) * 2 + lhi) * 8);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, vb, o[j], 0, 0, 0);
      }
    }
  };

  if (pipelined) {
    if (wave < NTAB && wave < ntiles) fill_tables(wave, wave);
    __syncthreads();                       // qoff + tables of tiles 0..2 visible
    stage_q(0);
    k_issue(0, 0);
    v_issue(0);
    k_commit();
    v_commit();
    __syncthreads();                       // Q + tile 0 in LDS
    if (ntiles > 1) {
      k_issue(0, 1);
      v_issue(1);
    }
    for (int kt = 0; kt < ntiles; ++kt) {
      f32x16 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      unsigned long long mbits = 0ull;
      if (use_mask) mbits = kmask[(kt % NTAB) * 16 + my_region] >> (4 * lhi);
#pragma unroll
      for (int d8 = 0; d8 < 8; ++d8) {
        if (d8 * 16 < DC) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
            s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const abf16x8*>(kp + t * 32 * QP + d8 * 16),
                                                           *reinterpret_cast<const abf16x8*>(qp + d8 * 16), s[t], 0, 0, 0);
        }
      }
      softmax_pv(kt, s, mbits);
      __syncthreads();                     // every wave is done with the LDS image (and the tables) of tile kt
      if (kt + 1 < ntiles) {
        k_commit();                        // tile kt+1: issued one iteration ago
        v_commit();
        if (kt + 2 < ntiles) {
          k_issue(0, (kt + 2) % NTAB);     // lands during the whole next iteration
          v_issue((kt + 2) % NTAB);
        }
        if (kt + 3 < ntiles && wave == (kt & 3)) fill_tables(kt + 3, kt % NTAB);
        __syncthreads();                   // image of tile kt+1 complete; tables of tile kt+3 visible before their use
      }
    }
  } else {
    for (int kt = 0; kt < ntiles; ++kt) {
      f32x16 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      unsigned long long mbits = 0ull;
      for (int ch = 0; ch < nch; ++ch) {
        __syncthreads();
        if (ch == 0 && wave == 0) fill_tables(kt, 0);
        if (ch == 0) __syncthreads();
        stage_q(ch);
        k_issue(ch, 0);
        if (ch == 0) v_issue(0);
        k_commit();
        if (ch == 0) v_commit();
        __syncthreads();
        if (ch == 0 && use_mask) mbits = kmask[my_region] >> (4 * lhi);
        for (int d = 0; d < DC; d += 16) {
          const abf16x8 qf = *reinterpret_cast<const abf16x8*>(qp + d);
#pragma unroll
          for (int t = 0; t < 2; ++t)
            s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const abf16x8*>(kp + t * 32 * QP + d),
                                                           qf, s[t], 0, 0, 0);
        }
      }
      softmax_pv(kt, s, mbits);
    }
  }

  // Output: the 32 x DVS tile of each wave goes through LDS so that global stores are 16 bytes per lane along dv
  // (4-byte stores straight from the MFMA layout are issue-bound: 64 store instructions per lane per block).
  const float inv_l = 1.0f / l_run;
  const bool vec_o = (p.Dv % 4 == 0) && (p.o_ts % 4 == 0) && (p.o_bs % 4 == 0) && (p.o_hs % 4 == 0) &&
                     ((uintptr_t)p.o % 16 == 0);
  if (vec_o) {
    constexpr int OP = DVS + 4;
    __syncthreads();                       // every wave is done with Q / K / V^T: the operand area becomes the staging area
    float* ost = reinterpret_cast<float*>(Qs) + wave * 32 * OP;
    long* ooff = qoff;                     // reuse: output row offsets of this block's 128 queries
    if (tid < 128) {
      const int t = q0 + tid;
      ooff[tid] = (t < p.Lq) ? tok_q(t, p.o_bs, p.o_ts) + (long)head * p.o_hs : -1;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
      const float il = __shfl(inv_l, qrow);
#pragma unroll
      for (int j = 0; j < DVT; ++j) ost[qrow * OP + j * 32 + l31] = o[j][r] * il;
    }
    __syncthreads();                       // staged tiles + ooff visible
    constexpr int C4 = DVS / 4;
#pragma unroll 4
    for (int i = lane; i < 32 * C4; i += 64) {
      const int row = i / C4, c4 = (i - row * C4) * 4;
      const long base = ooff[wave * 32 + row];
      if (base >= 0 && dv0 + c4 < p.Dv)
        *reinterpret_cast<float4*>(p.o + base + dv0 + c4) = *reinterpret_cast<const float4*>(ost + row * OP + c4);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qrow = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    const float il = __shfl(inv_l, qrow);
    const int t = q0 + wave * 32 + qrow;
    if (t < p.Lq) {
      const long base = tok_q(t, p.o_bs, p.o_ts) + (long)head * p.o_hs;
#pragma unroll
      for (int j = 0; j < DVT; ++j) {
        const int dv = dv0 + j * 32 + l31;
        if (dv < p.Dv) p.o[base + dv] = o[j][r] * il;
      }
    }
  }
}

template <int DVT>
static int launch_attn_bf16in(const AttnP& p, hipStream_t st) {
  const int DC = p.D < 128 ? p.D : 128;
  const size_t operands = (size_t)((128 + 64) * (DC + 8) + DVT * 32 * 72) * 2;
  const size_t staging = (size_t)4 * 32 * (DVT * 32 + 4) * 4;          // the output tiles reuse the operand area
  const size_t lds = (128 + 3 * 64 + 3 * 64) * sizeof(long) + 3 * 16 * sizeof(unsigned long long) +
                     (operands > staging ? operands : staging);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bf16in_kernel<DVT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)attn_bf16in_kernel<DVT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_attention: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid(cdiv(p.Lq, 128), p.H * p.nslices, p.B);
  if (p.mode == 2 && p.shift > 0)
    hipLaunchKernelGGL((attn_bf16in_kernel<DVT, true>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((attn_bf16in_kernel<DVT, false>), grid, dim3(256), lds, st, p);
  KEEP_LAUNCH_CHECK("keep_attention(bf16 inputs)");
  return KEEP_OK;
}

static int attn_args_in(const keep_attention_args* src, keep_attention_args& a) {
  KEEP_REQUIRE(src != nullptr, "keep_attention: null args");
  const uint32_t sz = src->struct_size;
  if (sz < KEEP_ATTENTION_ARGS_V12_SIZE || sz > sizeof(keep_attention_args) || sz % 8 != 0) {
    keep_set_error("keep_attention: args.struct_size = %u, this library (ABI v%d) accepts %d..%zu -- set it to "
                   "sizeof(keep_attention_args) of the header the caller was built with", sz, KEEP_ABI_VERSION,
                   KEEP_ATTENTION_ARGS_V12_SIZE, sizeof(keep_attention_args));
    return KEEP_EINVAL;
  }
  memset(&a, 0, sizeof(a));
  memcpy(&a, src, sz);
  return KEEP_OK;
}

extern "C" int32_t keep_sizeof_attention_args(void) { return (int32_t)sizeof(keep_attention_args); }

extern "C" int64_t keep_attention_workspace_bytes(const keep_attention_args* a_in) {
  keep_attention_args a_local;
  if (attn_args_in(a_in, a_local) != KEEP_OK) return 0;
  const keep_attention_args* a = &a_local;
  if (a->mma != KEEP_MMA_X3 || a->in_dtype == KEEP_BF16 || a->B <= 0 || a->H <= 0 || a->Lk <= 0) return 0;
  const bool x3_ok = (a->D % 16 == 0) && (a->q_ts % 4 == 0) && (a->q_bs % 4 == 0) && (a->q_hs % 4 == 0) && (a->k_ts % 4 == 0) &&
                     (a->k_bs % 4 == 0) && (a->k_hs % 4 == 0) && ((uintptr_t)a->q % 16 == 0) && ((uintptr_t)a->k % 16 == 0) &&
                     !(a->flags & KEEP_ATTN_NO_X3);
  if (!x3_ok) return 0;
  AttnP p;
  p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.D = a->D; p.Dv = a->Dv; p.flags = a->flags;
  p.mode = a->mode; p.q_amax = a->q_amax; p.o = a->o; p.o_ts = a->o_ts; p.o_bs = a->o_bs; p.o_hs = a->o_hs;
  if (attn_two_pass_ok(p)) return attn_two_pass_bytes(p);      // the score matrix between the two launches
  return attn_pack_bytes(p, a->mma);
}

extern "C" int32_t keep_attention(const keep_attention_args* a_in, void* stream) {
  keep_attention_args a_local;
  const int rc_in = attn_args_in(a_in, a_local);
  if (rc_in != KEEP_OK) return rc_in;
  const keep_attention_args* a = &a_local;
  KEEP_REQUIRE(a->q && a->k && a->v && a->o, "keep_attention: null tensor pointer");
  KEEP_REQUIRE(a->B > 0 && a->H > 0 && a->Lq > 0 && a->Lk > 0 && a->D > 0 && a->Dv > 0, "keep_attention: bad dims");
  KEEP_REQUIRE(a->D % 2 == 0 && (a->D <= 128 || a->D % 128 == 0), "keep_attention: D=%d must be even and (<= 128 or a multiple of 128)", a->D);
  KEEP_REQUIRE(a->mode >= 0 && a->mode <= 2, "keep_attention: bad mode %d", a->mode);
  KEEP_REQUIRE(a->mma == KEEP_MMA_F32 || a->mma == KEEP_MMA_BF16 || a->mma == KEEP_MMA_X3, "keep_attention: bad mma %d", a->mma);
  if (a->mode == 1)
    KEEP_REQUIRE(a->T > 0 && a->seg_len > 0 && a->Lk == 2 * a->seg_len && a->B % a->T == 0,
                 "keep_attention: sparse-causal mode needs Lk == 2*seg_len and B %% T == 0");
  if (a->mode == 2) {
    KEEP_REQUIRE(a->ksplit > 0 && a->img_h % a->ksplit == 0 && a->img_w % a->ksplit == 0, "keep_attention: bad window split");
    const int wh = a->img_h / a->ksplit, ww = a->img_w / a->ksplit;
    KEEP_REQUIRE(a->Lq == wh * ww && a->Lk == wh * ww, "keep_attention: window mode needs Lq == Lk == window size");
    KEEP_REQUIRE(a->n_img > 0 && a->B == a->n_img * a->ksplit * a->ksplit, "keep_attention: B != n_img*ksplit^2");
    KEEP_REQUIRE(a->shift >= 0 && a->shift < wh && a->shift < ww, "keep_attention: bad shift");
    KEEP_REQUIRE(a->kv_rot >= 0 && a->kv_rot < a->n_img, "keep_attention: bad kv_rot");
    KEEP_REQUIRE((long)a->img_h * a->img_w <= 65536, "keep_attention: window mode supports maps of at most 65536 tokens");
  }
  AttnP p;
  p.flags = a->flags;
  p.q = (const float*)a->q; p.k = (const float*)a->k; p.v = (const float*)a->v; p.o = a->o;
  p.q16 = (const unsigned short*)a->q; p.k16 = (const unsigned short*)a->k; p.v16 = (const unsigned short*)a->v;
  p.q_bs = a->q_bs; p.q_ts = a->q_ts; p.q_hs = a->q_hs;
  p.k_bs = a->k_bs; p.k_ts = a->k_ts; p.k_hs = a->k_hs;
  p.v_bs = a->v_bs; p.v_ts = a->v_ts; p.v_hs = a->v_hs;
  p.o_bs = a->o_bs; p.o_ts = a->o_ts; p.o_hs = a->o_hs;
  p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk; p.D = a->D; p.Dv = a->Dv;
  p.scale = a->scale; p.mode = a->mode; p.T = a->T; p.seg_len = a->seg_len;
  p.img_h = a->img_h; p.img_w = a->img_w; p.ksplit = a->ksplit; p.shift = a->shift; p.kv_rot = a->kv_rot;
  p.n_img = a->n_img;
  p.q_amax = a->q_amax; p.k_amax = a->k_amax; p.v_amax = a->v_amax;
  p.kv_pack = nullptr;
  KEEP_REQUIRE((!a->q_amax && !a->k_amax && !a->v_amax) || (a->q_amax && a->k_amax && a->v_amax && a->mode == 0 && a->mma == KEEP_MMA_X3),
               "keep_attention: q/k/v_amax come together, with KEEP_MMA_X3 and mode 0 only");
  hipStream_t st = (hipStream_t)stream;
  // dv slice per block: 32 / 64 / 128 columns
  const int dvt = a->Dv <= 32 ? 1 : (a->Dv <= 64 ? 2 : 4);
  p.nslices = cdiv(a->Dv, dvt * 32);
  if (a->in_dtype == KEEP_BF16) {
    const bool ok = a->mma == KEEP_MMA_BF16 && (a->D % 16 == 0) && (a->Dv % 8 == 0) && (a->q_ts % 8 == 0) &&
                    (a->q_bs % 8 == 0) && (a->q_hs % 8 == 0) && (a->k_ts % 8 == 0) && (a->k_bs % 8 == 0) &&
                    (a->k_hs % 8 == 0) && (a->v_ts % 8 == 0) && (a->v_bs % 8 == 0) && (a->v_hs % 8 == 0) &&
                    ((uintptr_t)a->q % 16 == 0) && ((uintptr_t)a->k % 16 == 0) && ((uintptr_t)a->v % 16 == 0);
    KEEP_REQUIRE(ok, "keep_attention: bf16 inputs need KEEP_MMA_BF16, D %% 16 == 0, Dv %% 8 == 0 and 16-byte aligned rows");
    if (dvt == 1) return launch_attn_bf16in<1>(p, st);
    if (dvt == 2) return launch_attn_bf16in<2>(p, st);
    return launch_attn_bf16in<4>(p, st);
  }
  if (a->mma == KEEP_MMA_BF16) {
    // bf16 operands: needs 16-byte-aligned fp32 rows and D a multiple of 16
    const bool ok = (a->D % 16 == 0) && (a->q_ts % 4 == 0) && (a->q_bs % 4 == 0) && (a->q_hs % 4 == 0) &&
                    (a->k_ts % 4 == 0) && (a->k_bs % 4 == 0) && (a->k_hs % 4 == 0) && ((uintptr_t)a->q % 16 == 0) &&
                    ((uintptr_t)a->k % 16 == 0);
    KEEP_REQUIRE(ok, "keep_attention: KEEP_MMA_BF16 needs D %% 16 == 0 and 16-byte aligned q/k rows");
    if (a->Lq <= 32) {
      if (dvt == 1) return launch_attn_bf16<1, 1>(p, st);
      if (dvt == 2) return launch_attn_bf16<1, 2>(p, st);
      return launch_attn_bf16<1, 4>(p, st);
    }
    if (dvt == 1) return launch_attn_bf16<4, 1>(p, st);
    if (dvt == 2) return launch_attn_bf16<4, 2>(p, st);
    return launch_attn_bf16<4, 4>(p, st);
  }
  // split fp16: fp32 tensors with 16-byte aligned rows, D a multiple of 16; everything else runs on the exact-f32 kernel
  if (a->mma == KEEP_MMA_X3 && (a->D % 16 == 0) && (a->q_ts % 4 == 0) && (a->q_bs % 4 == 0) && (a->q_hs % 4 == 0) &&
      (a->k_ts % 4 == 0) && (a->k_bs % 4 == 0) && (a->k_hs % 4 == 0) && ((uintptr_t)a->q % 16 == 0) &&
      ((uintptr_t)a->k % 16 == 0) && !(a->flags & KEEP_ATTN_NO_X3)) {
    if (attn_small_ok(p)) return launch_attn_x3_small(p, st);
    if (attn_two_pass_ok(p) && a->workspace && a->workspace_bytes >= attn_two_pass_bytes(p) && (uintptr_t)a->workspace % 16 == 0)
      return launch_attn_x3_two_pass(p, (float*)a->workspace, st);
    if (a->Lq <= 32) {
      if (dvt == 1) return launch_attn_x3<1, 1>(p, st);
      if (dvt == 2) return launch_attn_x3<1, 2>(p, st);
      return launch_attn_x3<1, 4>(p, st);
    }
    {
      const long need = attn_pack_bytes(p, a->mma);
      if (need > 0 && a->workspace && a->workspace_bytes >= need && (uintptr_t)a->workspace % 16 == 0) {
        p.kv_pack = (const _Float16*)a->workspace;
        if (a->D == 128) {
          if (dvt == 1) return launch_attn_x3_packed<1, 8>(p, st);
          if (dvt == 2) return launch_attn_x3_packed<2, 8>(p, st);
          return launch_attn_x3_packed<4, 8>(p, st);
        }
        if (dvt == 1) return launch_attn_x3_packed<1, 16>(p, st);
        if (dvt == 2) return launch_attn_x3_packed<2, 16>(p, st);
        return launch_attn_x3_packed<4, 16>(p, st);
      }
    }
    if (dvt == 1) return launch_attn_x3<4, 1>(p, st);
    if (dvt == 2) return launch_attn_x3<4, 2>(p, st);
    return launch_attn_x3<4, 4>(p, st);
  }
  // one wave per block for tiny query counts (temporal attention over T frames), else 4 (one per SIMD)
  if (a->Lq <= 32) {
    if (dvt == 1) return launch_attn<1, 1>(p, st);
    if (dvt == 2) return launch_attn<1, 2>(p, st);
    return launch_attn<1, 4>(p, st);
  }
  if (dvt == 1) return launch_attn<4, 1>(p, st);
  if (dvt == 2) return launch_attn<4, 2>(p, st);
  return launch_attn<4, 4>(p, st);
}

// ------------------------------------------------------------------------------------------------ fused GMFlow FFN
// GM/transformer.py:139-142,182: message = Linear(8C -> C)( GELU( Linear(2C -> 8C)( cat[source, message] ) ) ), no biases,
// C = 128.  As two GEMM launches the [M, 8C] intermediate costs 5 GB written + 5 GB read per layer at M = 1.2 M tokens
// (bench: 4.7 ms + 1.6 ms per layer, six layers).  Fused, the intermediate never leaves the CU -- same structure as
// the attention kernel above with W0 in the role of K and W2 in the role of V:
//   per 64-wide chunk of the hidden dimension   H^T (hidden x tokens) = W0c . X^T        (X fragments live in registers)
//                                               out (tokens x C)     += gelu(H) . W2c^T  (H feeds the MFMA from the
//   lane's own registers; W2c is staged with its hidden columns permuted into that order, vt_pos)
// Block = 128 tokens (4 waves x 32), W0c / W2c chunks in LDS (52 KB), next chunk prefetched into registers while the
// current one is on the matrix cores, 2 blocks per CU; fp32 in / fp32 out, bf16 MFMA operands, exact-erf-class GELU
// (erf_fast, 1.5e-7) in fp32.
#define MLP_C 128
__global__ __launch_bounds__(256, 2) void gm_mlp_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                        const unsigned short* __restrict__ w0,
                                                        const unsigned short* __restrict__ w2, float* __restrict__ out,
                                                        long M) {
  constexpr int K1 = 2 * MLP_C, HID = 8 * MLP_C, HC = 64;
  constexpr int P0 = K1 + 8;               // W0c row pitch (bf16): 528 B = 33 slots
  constexpr int P2 = HC + 8;               // W2c row pitch (bf16): 144 B = 9 slots
  constexpr int OP = MLP_C + 4;            // output staging pitch (floats)
  constexpr int OPER_B = (HC * P0 + MLP_C * P2) * 2;
  constexpr int STAGE_B = 4 * 32 * OP * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[OPER_B > STAGE_B ? OPER_B : STAGE_B];
  __bf16* W0s = reinterpret_cast<__bf16*>(lds_raw);            // [64][P0]
  __bf16* W2s = W0s + HC * P0;                                 // [128][P2], hidden columns permuted (vt_pos)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const long m0 = (long)blockIdx.x * 128;
  const long my_tok = m0 + wave * 32 + l31;

  // ---- X fragments: token my_tok, channels ks*16 + lhi*8 .. +8 of cat[xa | xb], ks = 0..15
  abf16x8 xf[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = ks * 16 + lhi * 8;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (my_tok < M) {
      const float* src = (c < MLP_C ? xa + my_tok * MLP_C + c : xb + my_tok * MLP_C + (c - MLP_C));
      v0 = *reinterpret_cast<const float4*>(src);
      v1 = *reinterpret_cast<const float4*>(src + 4);
    }
    xf[ks][0] = (__bf16)v0.x; xf[ks][1] = (__bf16)v0.y; xf[ks][2] = (__bf16)v0.z; xf[ks][3] = (__bf16)v0.w;
    xf[ks][4] = (__bf16)v1.x; xf[ks][5] = (__bf16)v1.y; xf[ks][6] = (__bf16)v1.z; xf[ks][7] = (__bf16)v1.w;
  }

  // ---- weight chunk staging: W0c = rows hc*64..+64 of w0 [HID][K1]; W2c = columns hc*64..+64 of w2 [C][HID]
  // (named registers: hipcc keeps small arrays that are captured by a lambda in scratch memory)
  uint4 p00, p01, p02, p03, p04, p05, p06, p07, p20, p21, p22, p23;
#define MLP_P0(X) X(0, p00) X(1, p01) X(2, p02) X(3, p03) X(4, p04) X(5, p05) X(6, p06) X(7, p07)
#define MLP_P2(X) X(0, p20) X(1, p21) X(2, p22) X(3, p23)
#define MLP_LD0(U, R) R = *reinterpret_cast<const uint4*>(w0 + (long)(hc_n * HC + ((tid + (U) * 256) >> 5)) * K1 + (((tid + (U) * 256) & 31) << 3));
#define MLP_LD2(U, R) R = *reinterpret_cast<const uint4*>(w2 + (long)((tid + (U) * 256) >> 3) * HID + hc_n * HC + (((tid + (U) * 256) & 7) << 3));
#define MLP_ST0(U, R) *reinterpret_cast<uint4*>(W0s + ((tid + (U) * 256) >> 5) * P0 + (((tid + (U) * 256) & 31) << 3)) = R;
  // hidden h8 .. h8+7 of a W2 row: the first four land at vt_pos(h8), the last four 8 positions further (other lane half)
#define MLP_ST2(U, R)                                                                                   \
  {                                                                                                     \
    const int h8 = ((tid + (U) * 256) & 7) << 3;                                                        \
    uint2* d = reinterpret_cast<uint2*>(W2s + ((tid + (U) * 256) >> 3) * P2 + (h8 >> 5) * 32 + vt_pos(h8 & 31)); \
    d[0] = make_uint2(R.x, R.y);                                                                        \
    d[2] = make_uint2(R.z, R.w);                                                                        \
  }

  f32x16 o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;

  const __bf16* a0 = W0s + l31 * P0 + lhi * 8;
  {
    const int hc_n = 0;
    MLP_P0(MLP_LD0)
    MLP_P2(MLP_LD2)
  }
  for (int hc = 0; hc < HID / HC; ++hc) {
    MLP_P0(MLP_ST0)
    MLP_P2(MLP_ST2)
    __syncthreads();                               // chunk hc in LDS
    if (hc + 1 < HID / HC) {                       // next chunk flies during the MFMAs below
      const int hc_n = hc + 1;
      MLP_P0(MLP_LD0)
      MLP_P2(MLP_LD2)
    }
    f32x16 s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const abf16x8*>(a0 + t * 32 * P0 + ks * 16), xf[ks],
                                                       s[t], 0, 0, 0);
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      abf16x8 pa;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = s[st >> 1][(st & 1) * 8 + j];
        pa[j] = (__bf16)(0.5f * v * (1.0f + erf_fast(v * 0.70710678118654752440f)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const abf16x8 wb = *reinterpret_cast<const abf16x8*>(W2s + (j * 32 + l31) * P2 + (st >> 1) * 32 + ((st & 1) * 2 + lhi) * 8);
        o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, wb, o[j], 0, 0, 0);
      }
    }
    __syncthreads();                               // every wave is done with chunk hc
  }

  // ---- output through LDS: 16-byte stores along the channel axis
  float* ost = reinterpret_cast<float*>(lds_raw) + wave * 32 * OP;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
    for (int j = 0; j < 4; ++j) ost[row * OP + j * 32 + l31] = o[j][r];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);              // wave-private tile: LDS writes of this wave landed
#pragma unroll 4
  for (int i = lane; i < 32 * (MLP_C / 4); i += 64) {
    const int row = i >> 5, c4 = (i & 31) << 2;
    const long tok = m0 + wave * 32 + row;
    if (tok < M) *reinterpret_cast<float4*>(out + tok * MLP_C + c4) = *reinterpret_cast<const float4*>(ost + row * OP + c4);
  }
}

extern "C" int32_t keep_gm_mlp(const float* a, const float* b, const void* w0_bf16, const void* w2_bf16, float* out,
                               int64_t M, int32_t C, void* stream) {
  KEEP_REQUIRE(a && b && w0_bf16 && w2_bf16 && out && M > 0, "keep_gm_mlp: bad args");
  KEEP_REQUIRE(C == MLP_C, "keep_gm_mlp: built for C = %d (got %d)", MLP_C, C);
  KEEP_REQUIRE((uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 && (uintptr_t)w0_bf16 % 16 == 0 &&
                   (uintptr_t)w2_bf16 % 16 == 0 && (uintptr_t)out % 16 == 0,
               "keep_gm_mlp: 16-byte alignment");
  hipLaunchKernelGGL(gm_mlp_kernel, dim3(cdiv(M, 128)), dim3(256), 0, (hipStream_t)stream, a, b,
                     (const unsigned short*)w0_bf16, (const unsigned short*)w2_bf16, out, (long)M);
  KEEP_LAUNCH_CHECK("keep_gm_mlp");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ streaming token GEMM
// out[M, N] = X[M, 128] . W[N, 128]^T (+ bias): the GMFlow projections (qkv / q / kv / merge, GM/transformer.py:148-176) at
// M = 1.2 M tokens are pure streaming: 2 K steps of work per 128x128 output tile, so the generic tile kernel spends its
// time in per-tile prologue / epilogue (650 us for 1.27 GB = 1.9 TB/s).  Here W (<= 384 x 128 bf16) is staged in LDS ONCE
// per persistent block; the block then walks token tiles of 128 rows: the next tile's X rows are in flight (registers)
// while the current one is multiplied, outputs leave through a wave-private LDS tile as 16-byte stores (fp32 or bf16).
#define TL_K 128
template <int NT>                                   // N = NT * 128
__global__ __launch_bounds__(256, (NT == 1 ? 2 : 1)) void token_linear_kernel(const float* __restrict__ x,
                                                                               const unsigned short* __restrict__ w,
                                                                               const float* __restrict__ bias, void* __restrict__ out,
                                                                               long M, int out_bf16, int n_tiles) {
  constexpr int N = NT * 128;
  constexpr int XP = TL_K + 8;                      // bf16 pitch: 272 B = 17 slots
  constexpr int OP = 64 + 4;                        // staging pitch (floats): 64 output columns at a time
  extern __shared__ __attribute__((aligned(16))) unsigned char tl_raw[];
  __bf16* Wsh = reinterpret_cast<__bf16*>(tl_raw);                 // [N][XP]
  __bf16* Xs = Wsh + N * XP;                                       // [128][XP]
  float* Ost = reinterpret_cast<float*>(Xs);                       // [4 waves][32][OP]: aliases the X tile (same size)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  for (int i = tid; i < N * (TL_K / 8); i += 256) {                // W -> LDS, once
    const int row = i >> 4, c8 = (i & 15) << 3;
    *reinterpret_cast<uint4*>(Wsh + row * XP + c8) = *reinterpret_cast<const uint4*>(w + (long)row * TL_K + c8);
  }

  // X tile 128 x 128 fp32 = 4096 float4: 16 per thread, row = (tid >> 5) + 8*u, float4 column = tid & 31
  float4 x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
#define TL_XR(X) X(0, x0) X(1, x1) X(2, x2) X(3, x3) X(4, x4) X(5, x5) X(6, x6) X(7, x7) X(8, x8) X(9, x9) X(10, x10) \
                 X(11, x11) X(12, x12) X(13, x13) X(14, x14) X(15, x15)
#define TL_LD(U, R)                                                                                         \
  {                                                                                                         \
    const long row = m_n + (tid >> 5) + 8 * (U);                                                            \
    R = row < M ? *reinterpret_cast<const float4*>(x + row * TL_K + ((tid & 31) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f); \
  }
#define TL_ST(U, R)                                                                                         \
  {                                                                                                         \
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;                                            \
    bf16x4_t h;                                                                                             \
    h[0] = (__bf16)R.x; h[1] = (__bf16)R.y; h[2] = (__bf16)R.z; h[3] = (__bf16)R.w;                         \
    *reinterpret_cast<bf16x4_t*>(Xs + ((tid >> 5) + 8 * (U)) * XP + ((tid & 31) << 2)) = h;                 \
  }

  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  {
    const long m_n = (long)tile * 128;
    TL_XR(TL_LD)
  }
  const __bf16* ap = Xs + (wave * 32 + l31) * XP + lhi * 8;
  const __bf16* bp = Wsh + l31 * XP + lhi * 8;
  float* ost = Ost + wave * 32 * OP;
  while (true) {
    const long m0 = (long)tile * 128;
    __syncthreads();                                 // previous tile's fragment reads are done (and W is staged)
    TL_XR(TL_ST)
    __syncthreads();
    const int next = tile + gridDim.x;
    if (next < n_tiles) {
      const long m_n = (long)next * 128;
      TL_XR(TL_LD)
    }
    abf16x8 af[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) af[ks] = *reinterpret_cast<const abf16x8*>(ap + ks * 16);
    __syncthreads();                                 // every wave holds its X fragments: the tile becomes the output staging area
#pragma unroll
    for (int nh = 0; nh < NT * 2; ++nh) {            // 64 output columns at a time
      f32x16 o[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          o[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], *reinterpret_cast<const abf16x8*>(bp + ((nh * 2 + j) * 32) * XP + ks * 16),
                                                         o[j], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
#pragma unroll
        for (int j = 0; j < 2; ++j) ost[row * OP + j * 32 + l31] = o[j][r];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);            // wave-private tile
#pragma unroll 4
      for (int i = lane; i < 32 * 16; i += 64) {
        const int row = i >> 4, c4 = (i & 15) << 2;
        const long tok = m0 + wave * 32 + row;
        if (tok < M) {
          float4 v = *reinterpret_cast<const float4*>(ost + row * OP + c4);
          if (bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + nh * 64 + c4);
            v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
          }
          if (out_bf16) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
            bf16x4_t h;
            h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
            *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(out) + tok * N + nh * 64 + c4) = h;
          } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + tok * N + nh * 64 + c4) = v;
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);            // staging tile read before the next 64 columns overwrite it
    }
    if (next >= n_tiles) break;
    tile = next;
  }
#undef TL_XR
#undef TL_LD
#undef TL_ST
}

template <int NT>
static int launch_token_linear(const float* x, const void* w, const float* bias, void* out, long M, int out_bf16,
                               hipStream_t st) {
  const size_t lds = (size_t)(NT * 128 + 128) * (TL_K + 8) * 2;     // W + X tile (the output staging aliases the X tile)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)token_linear_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      keep_set_error("keep_token_linear: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return KEEP_EHIP;
    }
    attr_set = true;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  const int n_tiles = cdiv(M, 128);
  const int per_cu = (NT == 1) ? 2 : 1;
  const int blocks = n_tiles < per_cu * n_cu ? n_tiles : per_cu * n_cu;
  hipLaunchKernelGGL(token_linear_kernel<NT>, dim3(blocks), dim3(256), lds, st, x, (const unsigned short*)w, bias, out, M,
                     out_bf16, n_tiles);
  KEEP_LAUNCH_CHECK("keep_token_linear");
  return KEEP_OK;
}

extern "C" int32_t keep_token_linear(const float* x, const void* w_bf16, const float* bias, void* out, int64_t M, int32_t K,
                                     int32_t N, int32_t out_dtype, void* stream) {
  KEEP_REQUIRE(x && w_bf16 && out && M > 0, "keep_token_linear: bad args");
  KEEP_REQUIRE(K == TL_K && (N == 128 || N == 256 || N == 384), "keep_token_linear: built for K = 128, N in {128,256,384} (got K=%d N=%d)", K, N);
  KEEP_REQUIRE(out_dtype == KEEP_F32 || out_dtype == KEEP_BF16, "keep_token_linear: bad out_dtype");
  KEEP_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)w_bf16 % 16 == 0 && (uintptr_t)out % 16 == 0 &&
                   (!bias || (uintptr_t)bias % 16 == 0),
               "keep_token_linear: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  const int ob = out_dtype == KEEP_BF16 ? 1 : 0;
  if (N == 128) return launch_token_linear<1>(x, w_bf16, bias, out, (long)M, ob, st);
  if (N == 256) return launch_token_linear<2>(x, w_bf16, bias, out, (long)M, ob, st);
  return launch_token_linear<3>(x, w_bf16, bias, out, (long)M, ob, st);
}
