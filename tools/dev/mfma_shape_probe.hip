// Micro-probe: sustained rate of v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 (same FLOPs per cycle on paper) with random
// operands, 1 and 2 waves per SIMD -- does one shape cost less power (a higher sustained clock)?
// hipcc --offload-arch=gfx950 -O3 mfma_shape_probe.hip -o mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned seed) {
  const int lane = threadIdx.x & 63;
  unsigned r = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  h8 fa[4], fb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      r = r * 1664525u + 1013904223u;
      fa[k][j] = (_Float16)(((int)(r >> 9) & 2047) * (1.0f / 1024.0f) - 1.0f);
      r = r * 1664525u + 1013904223u;
      fb[k][j] = (_Float16)(((int)(r >> 9) & 2047) * (1.0f / 1024.0f) - 1.0f);
    }
  float s = 0.f;
  if (SHAPE == 32) {
    f16v acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[t & 3], fb[(t + 1) & 3], acc[t & 3], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 16; ++q) s += acc[a][q];
  } else {
    f4v acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[t & 3], fb[(t + 1) & 3], acc[t & 7], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) s += acc[a][q];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lane;
}

template <int SHAPE>
static void run(const char* name, int threads) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 40000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<SHAPE>, dim3(256), dim3(threads), 0, 0, out, 400, 1u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(probe<SHAPE>, dim3(256), dim3(threads), 0, 0, out, iters, 7u);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * (threads / 64) * (double)iters * 8 * 32768.0;      // 8 x 32x32x16 == 16 x 16x16x32 per iteration
  printf("%-40s threads=%d  %8.3f ms  %7.0f TFLOP/s\n", name, threads, ms, flops / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<32>("v_mfma_f32_32x32x16_f16, random operands", 256);
    run<16>("v_mfma_f32_16x16x32_f16, random operands", 256);
    run<32>("v_mfma_f32_32x32x16_f16, random operands", 512);
    run<16>("v_mfma_f32_16x16x32_f16, random operands", 512);
  }
  return 0;
}
