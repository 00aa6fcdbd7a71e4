// YoloDetector pre- and post-processing on the device (wm_facelib/detection/yolov5face/face_detector.py:48-104,
// utils/datasets.py:5-36, utils/general.py:89-165).  Both are one pass over HBM-resident data: the letterbox reads every frame
// byte once and writes the network input once; the selection reads the prediction rows once and writes only the survivors, so
// that a handful of rows per frame cross PCIe instead of [anchors, 16] floats (8 MB for a 1080p frame).
#include <math.h>

#include "keep_common.h"

// ------------------------------------------------------------------------------------------------ letterbox
// cv2.cvtColor(BGR2RGB) -> [cv2.resize(INTER_LINEAR) to (rw, rh)] -> cv2.copyMakeBorder(top, left, value 114) -> float / 255, written
// NHWC (the engine's layout; the reference's transpose(0, 3, 1, 2) is a view of the same values).
// The resize is OpenCV's 8-bit bilinear (modules/imgproc/src/resize.cpp, resizeGeneric_ with HResizeLinear / VResizeLinear, fixed point):
//   fx = (float)((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; sx < 0: (0, fx = 0); sx >= W - 1: (W - 1, fx = 0);
//   alpha = { sat_short(rint((1.f - fx) * 2048)), sat_short(rint(fx * 2048)) }; rows likewise except that out-of-range ROWS are clamped
//   and keep their weights; horizontal pass in int: S * a0 + S' * a1; vertical: (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// scale_x / scale_y arrive as the doubles OpenCV forms: 1. / ((double)rw / W).
__device__ __forceinline__ void cv_lin_coef(int d, double scale, int size, bool clamp_pos, int& s, int& a0, int& a1) {
  float f = (float)__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5);      // (two roundings, as the host code compiles: no fma)
  int i = (int)floorf(f);
  f -= (float)i;
  if (clamp_pos) {
    if (i < 0) { i = 0; f = 0.f; }
    if (i >= size - 1) { i = size - 1; f = 0.f; }
  }
  s = i;
  a0 = max(-32768, min(32767, __float2int_rn((1.f - f) * 2048.f)));
  a1 = max(-32768, min(32767, __float2int_rn(f * 2048.f)));
}

__global__ __launch_bounds__(256) void yolo_letterbox_kernel(const uint8_t* __restrict__ frames, float* __restrict__ out, int N, int H, int W,
                                                             int rh, int rw, int top, int left, int H2, int W2, double scale_x,
                                                             double scale_y, int swap_rb) {
  const long total = (long)N * H2 * W2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % W2);
    const long t = i / W2;
    const int y = (int)(t % H2);
    const int n = (int)(t / H2);
    const int dx = x - left, dy = y - top;
    int v[3] = {114, 114, 114};
    if (dx >= 0 && dx < rw && dy >= 0 && dy < rh) {
      const uint8_t* f = frames + (long)n * H * W * 3;
      if (rw == W && rh == H) {
        const uint8_t* p = f + ((long)dy * W + dx) * 3;
        v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
      } else {
        int sx, a0, a1, sy, b0, b1;
        cv_lin_coef(dx, scale_x, W, true, sx, a0, a1);
        cv_lin_coef(dy, scale_y, H, false, sy, b0, b1);
        const int x1 = min(sx + 1, W - 1);                       // (a1 = 0 whenever sx + 1 is outside)
        const int y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
        const uint8_t* r0 = f + (long)y0 * W * 3;
        const uint8_t* r1 = f + (long)y1 * W * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int h0 = r0[sx * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
          const int h1 = r1[sx * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
          const int q = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
          v[c] = min(255, max(0, q));
        }
      }
    }
    float* o = out + i * 3;
    const int c0 = swap_rb ? 2 : 0;
    o[0] = (float)v[c0] / 255.0f;                                // pp_imgs.float() / 255.0 (face_detector.py:66-67): IEEE division
    o[1] = (float)v[1] / 255.0f;
    o[2] = (float)v[2 - c0] / 255.0f;
  }
}

extern "C" int32_t keep_yolo_letterbox_u8(const uint8_t* frames, float* out, int32_t N, int32_t H, int32_t W, int32_t rh, int32_t rw,
                                          int32_t top, int32_t left, int32_t H2, int32_t W2, int32_t swap_rb, void* stream) {
  KEEP_REQUIRE(frames && out && N > 0 && H > 0 && W > 0 && rh > 0 && rw > 0 && top >= 0 && left >= 0 && top + rh <= H2 && left + rw <= W2,
               "keep_yolo_letterbox_u8: bad args (N %d, %dx%d -> %dx%d at (%d, %d) of %dx%d)", N, H, W, rh, rw, top, left, H2, W2);
  // (an exact 2x reduction is INTER_AREA inside cv2.resize: not this kernel's arithmetic)
  KEEP_REQUIRE(!(H == 2 * rh && W == 2 * rw), "keep_yolo_letterbox_u8: an exact 2x reduction is cv2's INTER_AREA path (unsupported)");
  const double scale_x = 1.0 / ((double)rw / (double)W), scale_y = 1.0 / ((double)rh / (double)H);
  const long total = (long)N * H2 * W2;
  const int blocks = (int)min((total + 255) / 256, 256L * 64);
  hipLaunchKernelGGL(yolo_letterbox_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, frames, out, N, H, W, rh, rw, top, left, H2, W2,
                     scale_x, scale_y, swap_rb);
  KEEP_LAUNCH_CHECK("keep_yolo_letterbox_u8");
  return KEEP_OK;
}

// ------------------------------------------------------------------------------------------------ candidate selection
// non_max_suppression_face up to the NMS call (general.py:89-143) for the one-class face detectors (nc = 1, multi_label False):
// rows with objectness > thr; conf = class score * objectness (float32 product); xywh -> xyxy (x -+ w / 2); rows with conf > thr are
// appended to the frame's compact list in keep_retina_nms's layout: x1 y1 x2 y2 conf lm x 10 row-index.
__global__ __launch_bounds__(256) void yolo_select_kernel(const float* __restrict__ pred, float* __restrict__ dets, int* __restrict__ counts,
                                                          int N, int P, int cap, float thr) {
  const long total = (long)N * P;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const float4* src = reinterpret_cast<const float4*>(pred + i * 16);
    const float4 r1 = src[1];
    if (!(r1.x > thr)) continue;                          // xc = prediction[..., 4] > conf_thres
    const float4 r3 = src[3];
    const float conf = r3.w * r1.x;                       // x[:, 15:] *= x[:, 4:5]
    if (!(conf > thr)) continue;
    const int n = (int)(i / P);
    const int slot = atomicAdd(counts + n, 1);
    if (slot >= cap) continue;
    const float4 r0 = src[0], r2 = src[2];
    const float hw = r0.z / 2, hh = r0.w / 2;
    float4* d = reinterpret_cast<float4*>(dets + ((long)n * cap + slot) * 16);
    d[0] = make_float4(r0.x - hw, r0.y - hh, r0.x + hw, r0.y + hh);
    d[1] = make_float4(conf, r1.y, r1.z, r1.w);
    d[2] = r2;
    d[3] = make_float4(r3.x, r3.y, r3.z, (float)(i - (long)n * P));
  }
}

extern "C" int32_t keep_yolo_select(const float* pred, float* dets, int32_t* counts, int32_t N, int32_t P, int32_t cap, float conf_threshold,
                                    void* stream) {
  KEEP_REQUIRE(pred && dets && counts && N > 0 && P > 0 && cap > 0 && P < (1 << 24) && (uintptr_t)pred % 16 == 0 && (uintptr_t)dets % 16 == 0,
               "keep_yolo_select: bad args (N %d, P %d, cap %d)", N, P, cap);
  const long total = (long)N * P;
  const int blocks = (int)min((total + 255) / 256, 256L * 32);
  hipLaunchKernelGGL(yolo_select_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, pred, dets, counts, N, P, cap, conf_threshold);
  KEEP_LAUNCH_CHECK("keep_yolo_select");
  return KEEP_OK;
}
