// Paste-back compositing on the device: the per-face work of FaceRestoreHelper.paste_faces_to_input_image
// (/root/reference/modules/deps/wm_facelib/utils/face_restoration_helper.py:346-475, use_parse=True branch) without the
// full-frame cv2 passes: parse mask -> two 101-tap separable Gaussians (:433-434) -> [border zeroing, /255 (:435-437) folded
// into the sampler] -> inverse-affine bilinear warp of mask (:441, float weights) and restored face (:382, 15-bit integer
// weights) -> float32 blend (:463) restricted to the face's bounding box -> clip / round-half-even / uint8 (:465-468).
// Arithmetic follows OpenCV's published algorithms operation by operation (fixed-point warp coordinates: AB_BITS 10, INTER_BITS
// 5; every float product / sum rounded separately, no FMA contraction: __fmul_rn / __fadd_rn) so that the result equals the
// numpy restatement oracle/paste_oracle.py bit for bit.  All kernels are HBM / latency bound (a 1080p frame is 6 MB).
#include "keep_common.h"

// HIP's __fmul_rn / __fadd_rn are plain operators: without this the compiler contracts them into FMAs (hipcc defaults to
// -ffp-contract=fast) and the results drift from the separately rounded restatement by an ulp.
#pragma clang fp contract(off)
// (the header intrinsics are inline functions compiled under the header's contraction mode: own helpers below the pragma)
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }
__device__ __forceinline__ double dmul_rn(double a, double b) { return a * b; }
__device__ __forceinline__ double dadd_rn(double a, double b) { return a + b; }

#define PASTE_MAXTAPS 1024     // (a 4 KB tap table per block; a face of ~10 000 px would need 1001 taps)

// ---- separable filter, BORDER_REFLECT_101, float32, taps accumulated in index order.  src: float [n,H,W], or class map
// uint8 [n,H,W] looked up through lut (MASK_COLORMAP, :428-429) when lut != nullptr.
__device__ __forceinline__ int reflect101(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * (n - 1) - i : i;
}

__global__ void sep_rows_kernel(const float* __restrict__ src, const uint8_t* __restrict__ cls, const float* __restrict__ lut,
                                float* __restrict__ dst, int H, int W, const float* __restrict__ kern, int ntap) {
  __shared__ float ks[PASTE_MAXTAPS];
  for (int i = threadIdx.x; i < ntap; i += blockDim.x) ks[i] = kern[i];
  __syncthreads();
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)gridDim.y * H * W;
  (void)total;
  const long base = (long)blockIdx.y * H * W;
  if (idx >= (long)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
  const int r = ntap >> 1;
  float acc = 0.f;
  for (int i = 0; i < ntap; ++i) {
    const int xx = reflect101(x - r + i, W);
    const float v = cls ? lut[cls[base + (long)y * W + xx]] : src[base + (long)y * W + xx];
    acc = add_rn(acc, mul_rn(ks[i], v));
  }
  dst[base + idx] = acc;
}

__global__ void sep_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W,
                                const float* __restrict__ kern, int ntap) {
  __shared__ float ks[PASTE_MAXTAPS];
  for (int i = threadIdx.x; i < ntap; i += blockDim.x) ks[i] = kern[i];
  __syncthreads();
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long base = (long)blockIdx.y * H * W;
  if (idx >= (long)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
  const int r = ntap >> 1;
  float acc = 0.f;
  for (int i = 0; i < ntap; ++i) {
    const int yy = reflect101(y - r + i, H);
    acc = add_rn(acc, mul_rn(ks[i], src[base + (long)yy * W + x]));
  }
  dst[base + idx] = acc;
}

extern "C" int32_t keep_sep_filter(const float* src, const uint8_t* classes, const float* lut, float* tmp, float* dst, int32_t n,
                                   int32_t H, int32_t W, const float* kern, int32_t ntap, void* stream) {
  KEEP_REQUIRE((src != nullptr) != (classes != nullptr), "keep_sep_filter: exactly one of src / classes");
  KEEP_REQUIRE(!classes || lut, "keep_sep_filter: a class map needs its lookup table");
  KEEP_REQUIRE(tmp && dst && kern && n > 0 && H > 1 && W > 1, "keep_sep_filter: bad arguments");
  KEEP_REQUIRE(ntap >= 1 && ntap <= PASTE_MAXTAPS && (ntap & 1) && ntap / 2 < H && ntap / 2 < W,
               "keep_sep_filter: odd tap count <= %d and smaller than twice the image, got %d", PASTE_MAXTAPS, ntap);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv((long)H * W, 256), n);
  hipLaunchKernelGGL(sep_rows_kernel, grid, dim3(256), 0, st, src, classes, lut, tmp, H, W, kern, ntap);
  hipLaunchKernelGGL(sep_cols_kernel, grid, dim3(256), 0, st, tmp, dst, H, W, kern, ntap);
  KEEP_LAUNCH_CHECK("keep_sep_filter");
  return KEEP_OK;
}

// ---- uint8 frame -> float32 accumulator, and back (clip [0,255], round half to even)
__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ x, float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)x[i];
}
__global__ void f32_round_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)rintf(fminf(fmaxf(x[i], 0.f), 255.f));
}
extern "C" int32_t keep_u8_to_f32(const uint8_t* x, float* out, int64_t n, void* stream) {
  KEEP_REQUIRE(x && out && n > 0, "keep_u8_to_f32: bad arguments");
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  KEEP_LAUNCH_CHECK("keep_u8_to_f32");
  return KEEP_OK;
}
extern "C" int32_t keep_f32_round_u8(const float* x, uint8_t* out, int64_t n, void* stream) {
  KEEP_REQUIRE(x && out && n > 0, "keep_f32_round_u8: bad arguments");
  hipLaunchKernelGGL(f32_round_u8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, (long)n);
  KEEP_LAUNCH_CHECK("keep_f32_round_u8");
  return KEEP_OK;
}

// ---- crop warp: cv2.warpAffine(frame uint8 [H,W,3], M, (dw, dh), INTER_LINEAR, BORDER_CONSTANT, borderValue) -- the call that
// PRODUCES the aligned 512x512 crops (face_restoration_helper.py:316-318, borderValue (135, 133, 132)).  Same fixed-point
// coordinates and 15-bit weights as the face sampler below; a tap outside the source reads the border colour.
struct WarpP {
  const uint8_t* src;
  uint8_t* dst;
  double m00, m01, m02, m10, m11, m12;   // destination -> source map
  int H, W, dh, dw;
  int b0, b1, b2;
};

__global__ void warp_affine_u8_kernel(WarpP p) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= p.dw || y >= p.dh) return;
  const long adelta = llrint(dmul_rn(dmul_rn(p.m00, (double)x), 1024.0));
  const long bdelta = llrint(dmul_rn(dmul_rn(p.m10, (double)x), 1024.0));
  const long X0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m01, (double)y), p.m02), 1024.0)) + 16;
  const long Y0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m11, (double)y), p.m12), 1024.0)) + 16;
  const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
  syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
  const int i00 = (32 - fx) * (32 - fy) * 32, i01 = fx * (32 - fy) * 32, i10 = (32 - fx) * fy * 32, i11 = fx * fy * 32;
  const int bv[3] = {p.b0, p.b1, p.b2};
  auto at = [&](int yy, int xx, int c) -> int {
    if (xx < 0 || xx >= p.W || yy < 0 || yy >= p.H) return bv[c];
    return p.src[((long)yy * p.W + xx) * 3 + c];
  };
  uint8_t* d = p.dst + ((long)y * p.dw + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    d[c] = (uint8_t)((at(sy, sx, c) * i00 + at(sy, sx + 1, c) * i01 + at(sy + 1, sx, c) * i10 + at(sy + 1, sx + 1, c) * i11 + (1 << 14)) >> 15);
}

extern "C" int32_t keep_warp_affine_u8(const uint8_t* src, int32_t H, int32_t W, uint8_t* dst, int32_t dh, int32_t dw,
                                       const double* dst_to_src, int32_t border_b, int32_t border_g, int32_t border_r, void* stream) {
  KEEP_REQUIRE(src && dst && dst_to_src && H > 0 && W > 0 && dh > 0 && dw > 0 && H < 32768 && W < 32768, "keep_warp_affine_u8: bad arguments");
  WarpP p;
  p.src = src; p.dst = dst;
  p.m00 = dst_to_src[0]; p.m01 = dst_to_src[1]; p.m02 = dst_to_src[2];
  p.m10 = dst_to_src[3]; p.m11 = dst_to_src[4]; p.m12 = dst_to_src[5];
  p.H = H; p.W = W; p.dh = dh; p.dw = dw; p.b0 = border_b & 255; p.b1 = border_g & 255; p.b2 = border_r & 255;
  hipLaunchKernelGGL(warp_affine_u8_kernel, dim3(cdiv(dw, 32), cdiv(dh, 8)), dim3(256), 0, (hipStream_t)stream, p);
  KEEP_LAUNCH_CHECK("keep_warp_affine_u8");
  return KEEP_OK;
}

// ---- use_parse=False soft mask (face_restoration_helper.py:386-415): coverage of the warped face square, rectangular erosions,
// Gaussian edge.  coverage: cv2.warpAffine(np.ones(face_size, float32), M, (W, H)) -- float weights of the quantised position,
// BORDER_CONSTANT 0 -- written for the whole frame.
struct CoverP {
  float* dst;
  double m00, m01, m02, m10, m11, m12;
  int H, W, fh, fw;
};
__global__ void warp_ones_kernel(CoverP p) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= p.W || y >= p.H) return;
  const long adelta = llrint(dmul_rn(dmul_rn(p.m00, (double)x), 1024.0));
  const long bdelta = llrint(dmul_rn(dmul_rn(p.m10, (double)x), 1024.0));
  const long X0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m01, (double)y), p.m02), 1024.0)) + 16;
  const long Y0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m11, (double)y), p.m12), 1024.0)) + 16;
  const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
  syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
  const float ax = div_rn((float)fx, 32.f), ay = div_rn((float)fy, 32.f);
  const float w00 = mul_rn(sub_rn(1.f, ax), sub_rn(1.f, ay)), w01 = mul_rn(ax, sub_rn(1.f, ay));
  const float w10 = mul_rn(sub_rn(1.f, ax), ay), w11 = mul_rn(ax, ay);
  auto one = [&](int yy, int xx) -> float { return (xx >= 0 && xx < p.fw && yy >= 0 && yy < p.fh) ? 1.f : 0.f; };
  float v = mul_rn(one(sy, sx), w00);
  v = add_rn(v, mul_rn(one(sy, sx + 1), w01));
  v = add_rn(v, mul_rn(one(sy + 1, sx), w10));
  v = add_rn(v, mul_rn(one(sy + 1, sx + 1), w11));
  p.dst[(long)y * p.W + x] = v;
}
extern "C" int32_t keep_warp_ones(float* dst, int32_t H, int32_t W, int32_t fh, int32_t fw, const double* dst_to_src, void* stream) {
  KEEP_REQUIRE(dst && dst_to_src && H > 0 && W > 0 && fh > 0 && fw > 0, "keep_warp_ones: bad arguments");
  CoverP p;
  p.dst = dst;
  p.m00 = dst_to_src[0]; p.m01 = dst_to_src[1]; p.m02 = dst_to_src[2];
  p.m10 = dst_to_src[3]; p.m11 = dst_to_src[4]; p.m12 = dst_to_src[5];
  p.H = H; p.W = W; p.fh = fh; p.fw = fw;
  hipLaunchKernelGGL(warp_ones_kernel, dim3(cdiv(W, 32), cdiv(H, 8)), dim3(256), 0, (hipStream_t)stream, p);
  KEEP_LAUNCH_CHECK("keep_warp_ones");
  return KEEP_OK;
}

// ---- draw_box (face_restoration_helper.py:393-400,467-475): mask_border = ones(face_size) with the rectangle (bt, bt) .. (fw - bt - 1,
// fh - bt - 1) zeroed (cv2.rectangle, filled, corners inclusive), warped like the coverage mask above; pixels where it exceeds 0.5 are
// painted (0, 255, 0) in the rounded uint8 frame.  One thread per pixel of the face's bounding box.
struct BoxP {
  uint8_t* frame;
  double m00, m01, m02, m10, m11, m12;
  int H, W, fh, fw, bt, x0, y0, x1, y1;
};
__global__ void draw_box_kernel(BoxP p) {
  const int x = p.x0 + blockIdx.x * 32 + (threadIdx.x & 31), y = p.y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= p.x1 || y >= p.y1) return;
  const long adelta = llrint(dmul_rn(dmul_rn(p.m00, (double)x), 1024.0));
  const long bdelta = llrint(dmul_rn(dmul_rn(p.m10, (double)x), 1024.0));
  const long X0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m01, (double)y), p.m02), 1024.0)) + 16;
  const long Y0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m11, (double)y), p.m12), 1024.0)) + 16;
  const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
  syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
  const float ax = div_rn((float)fx, 32.f), ay = div_rn((float)fy, 32.f);
  const float w00 = mul_rn(sub_rn(1.f, ax), sub_rn(1.f, ay)), w01 = mul_rn(ax, sub_rn(1.f, ay));
  const float w10 = mul_rn(sub_rn(1.f, ax), ay), w11 = mul_rn(ax, ay);
  auto border = [&](int yy, int xx) -> float {
    if (xx < 0 || xx >= p.fw || yy < 0 || yy >= p.fh) return 0.f;
    return (xx >= p.bt && xx <= p.fw - p.bt - 1 && yy >= p.bt && yy <= p.fh - p.bt - 1) ? 0.f : 1.f;
  };
  float v = mul_rn(border(sy, sx), w00);
  v = add_rn(v, mul_rn(border(sy, sx + 1), w01));
  v = add_rn(v, mul_rn(border(sy + 1, sx), w10));
  v = add_rn(v, mul_rn(border(sy + 1, sx + 1), w11));
  if (v > 0.5f) {
    uint8_t* d = p.frame + ((long)y * p.W + x) * 3;
    d[0] = 0; d[1] = 255; d[2] = 0;
  }
}
extern "C" int32_t keep_draw_box(uint8_t* frame, int32_t H, int32_t W, int32_t fh, int32_t fw, int32_t thickness, const double* dst_to_src,
                                 int32_t x0, int32_t y0, int32_t x1, int32_t y1, void* stream) {
  KEEP_REQUIRE(frame && dst_to_src && H > 0 && W > 0 && fh > 0 && fw > 0 && thickness >= 1 && x0 >= 0 && y0 >= 0 && x1 <= W && y1 <= H,
               "keep_draw_box: bad arguments");
  if (x1 <= x0 || y1 <= y0) return KEEP_OK;
  BoxP p;
  p.frame = frame;
  p.m00 = dst_to_src[0]; p.m01 = dst_to_src[1]; p.m02 = dst_to_src[2];
  p.m10 = dst_to_src[3]; p.m11 = dst_to_src[4]; p.m12 = dst_to_src[5];
  p.H = H; p.W = W; p.fh = fh; p.fw = fw; p.bt = thickness; p.x0 = x0; p.y0 = y0; p.x1 = x1; p.y1 = y1;
  hipLaunchKernelGGL(draw_box_kernel, dim3(cdiv(x1 - x0, 32), cdiv(y1 - y0, 8)), dim3(256), 0, (hipStream_t)stream, p);
  KEEP_LAUNCH_CHECK("keep_draw_box");
  return KEEP_OK;
}

// cv2.erode(img, np.ones((k, k))): minimum over the k x k window anchored at k/2 (offsets -k/2 .. k-1-k/2), pixels outside the image
// never win (constant border +inf).  A minimum is separable: rows, then columns.
__global__ void erode_pass_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int k, int along_x) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (long)y * W);
  const int a = k >> 1;
  float m = INFINITY;
  for (int i = 0; i < k; ++i) {
    const int xx = along_x ? x - a + i : x, yy = along_x ? y : y - a + i;
    if (xx >= 0 && xx < W && yy >= 0 && yy < H) m = fminf(m, src[(long)yy * W + xx]);
  }
  dst[idx] = m;
}
extern "C" int32_t keep_erode_rect(const float* src, float* tmp, float* dst, int32_t H, int32_t W, int32_t k, void* stream) {
  KEEP_REQUIRE(src && tmp && dst && H > 0 && W > 0 && k >= 1 && k <= 4096, "keep_erode_rect: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(erode_pass_kernel, dim3(cdiv((long)H * W, 256)), dim3(256), 0, st, src, tmp, H, W, k, 1);
  hipLaunchKernelGGL(erode_pass_kernel, dim3(cdiv((long)H * W, 256)), dim3(256), 0, st, tmp, dst, H, W, k, 0);
  KEEP_LAUNCH_CHECK("keep_erode_rect");
  return KEEP_OK;
}

// ---- one face: warp (face uint8 x3, mask float) + blend into the float frame, over the face's bounding box
struct PasteP {
  float* acc;            // [H,W,3] float32 frame, in place
  const uint8_t* face;   // [fh,fw,3]
  const float* mask;     // [fh,fw] blurred parse mask on the 0..255 scale (before border zeroing and /255)
  double m00, m01, m02, m10, m11, m12;   // destination -> source map (inverse of the matrix given to cv2.warpAffine)
  int H, W, fh, fw, x0, y0, bw, bh, border;
  const float* frame_mask;   // use_parse=False: the soft mask already in FRAME space [H,W] (FH:411-415); then `mask` is unused
};

__device__ __forceinline__ float mask_at(const PasteP& p, int sy, int sx) {
  if (sx < 0 || sx >= p.fw || sy < 0 || sy >= p.fh) return 0.f;                                   // BORDER_CONSTANT 0
  if (sy < p.border || sy >= p.fh - p.border || sx < p.border || sx >= p.fw - p.border) return 0.f;   // :435-436
  return div_rn(p.mask[(long)sy * p.fw + sx], 255.0f);                                          // :437
}
__device__ __forceinline__ int face_at(const PasteP& p, int sy, int sx, int c) {
  if (sx < 0 || sx >= p.fw || sy < 0 || sy >= p.fh) return 0;
  return p.face[((long)sy * p.fw + sx) * 3 + c];
}

__global__ void paste_face_kernel(PasteP p) {
  const int bx = blockIdx.x * 32 + (threadIdx.x & 31), by = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (bx >= p.bw || by >= p.bh) return;
  const int x = p.x0 + bx, y = p.y0 + by;
  // imgwarp.cpp WarpAffineInvoker: per-column and per-row terms rounded to 1/1024 px (cvRound), sum truncated to 1/32 px
  const long adelta = llrint(dmul_rn(dmul_rn(p.m00, (double)x), 1024.0));
  const long bdelta = llrint(dmul_rn(dmul_rn(p.m10, (double)x), 1024.0));
  const long X0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m01, (double)y), p.m02), 1024.0)) + 16;
  const long Y0 = llrint(dmul_rn(dadd_rn(dmul_rn(p.m11, (double)y), p.m12), 1024.0)) + 16;
  const long X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
  long sxl = X >> 5, syl = Y >> 5;
  sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);
  syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
  const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
  // mask: float weights of the quantised position, left-to-right float sums
  const float ax = div_rn((float)fx, 32.f), ay = div_rn((float)fy, 32.f);
  const float w00 = mul_rn(sub_rn(1.f, ax), sub_rn(1.f, ay)), w01 = mul_rn(ax, sub_rn(1.f, ay));
  const float w10 = mul_rn(sub_rn(1.f, ax), ay), w11 = mul_rn(ax, ay);
  float soft;
  if (p.frame_mask) {
    soft = p.frame_mask[(long)y * p.W + x];
  } else {
    soft = mul_rn(mask_at(p, sy, sx), w00);
    soft = add_rn(soft, mul_rn(mask_at(p, sy, sx + 1), w01));
    soft = add_rn(soft, mul_rn(mask_at(p, sy + 1, sx), w10));
    soft = add_rn(soft, mul_rn(mask_at(p, sy + 1, sx + 1), w11));
  }
  const float inv = sub_rn(1.f, soft);
  // face: 15-bit integer weights, (sum + 2^14) >> 15
  const int i00 = (32 - fx) * (32 - fy) * 32, i01 = fx * (32 - fy) * 32, i10 = (32 - fx) * fy * 32, i11 = fx * fy * 32;
  float* dst = p.acc + ((long)y * p.W + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v = (face_at(p, sy, sx, c) * i00 + face_at(p, sy, sx + 1, c) * i01 + face_at(p, sy + 1, sx, c) * i10 +
                   face_at(p, sy + 1, sx + 1, c) * i11 + (1 << 14)) >> 15;
    dst[c] = add_rn(mul_rn(soft, (float)v), mul_rn(inv, dst[c]));                        // :463
  }
}

extern "C" int32_t keep_paste_face(float* frame, int32_t H, int32_t W, const uint8_t* face, const float* mask, int32_t fh, int32_t fw,
                                   const double* dst_to_src, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t mask_border,
                                   void* stream) {
  KEEP_REQUIRE(frame && face && mask && dst_to_src && H > 0 && W > 0 && fh > 1 && fw > 1, "keep_paste_face: bad arguments");
  // mask_border < 0: `mask` is a FRAME-space soft mask [H,W] (use_parse=False path), sampled at the destination pixel
  KEEP_REQUIRE(x0 >= 0 && y0 >= 0 && x1 <= W && y1 <= H, "keep_paste_face: box outside the frame");
  if (x1 <= x0 || y1 <= y0) return KEEP_OK;        // the face does not touch the frame
  PasteP p;
  p.acc = frame; p.face = face; p.mask = mask;
  p.m00 = dst_to_src[0]; p.m01 = dst_to_src[1]; p.m02 = dst_to_src[2];
  p.m10 = dst_to_src[3]; p.m11 = dst_to_src[4]; p.m12 = dst_to_src[5];
  p.H = H; p.W = W; p.fh = fh; p.fw = fw; p.x0 = x0; p.y0 = y0; p.bw = x1 - x0; p.bh = y1 - y0;
  p.border = mask_border < 0 ? 0 : mask_border;
  p.frame_mask = mask_border < 0 ? mask : nullptr;
  hipLaunchKernelGGL(paste_face_kernel, dim3(cdiv(p.bw, 32), cdiv(p.bh, 8)), dim3(256), 0, (hipStream_t)stream, p);
  KEEP_LAUNCH_CHECK("keep_paste_face");
  return KEEP_OK;
}
