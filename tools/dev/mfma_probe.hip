// Micro-probe: how fast can W waves per SIMD issue v_mfma_f32_32x32x16_f16 with NACC independent accumulators, with and
// without LDS fragment reads in the loop.  hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, int LDSR>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (_Float16)(0.001f * (i & 63));
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f16v acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  h8 fa[4], fb[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    fa[k] = *reinterpret_cast<const h8*>(&lds[(lane * 40 + k * 8) & 8191 & ~7]);
    fb[k] = *reinterpret_cast<const h8*>(&lds[(lane * 40 + 2048 + k * 8) & 8191 & ~7]);
  }
  for (int it = 0; it < iters; ++it) {
    if (LDSR) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        fa[k] = *reinterpret_cast<const h8*>(&lds[((lane * 40 + k * 8 + it * 16) & 4095) & ~7]);
        fb[k] = *reinterpret_cast<const h8*>(&lds[(((lane * 40 + k * 8 + it * 16) & 4095) & ~7) + 4096]);
      }
    }
#pragma unroll
    for (int t = 0; t < 12 / NACC * 1; ++t)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[(a + t) & 3], fb[(a * 3 + t) & 3], acc[a], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int LDSR>
void run(const char* name, int threads, int blocks_per_cu) {
  float* out;
  hipMalloc(&out, 256 * 8 * 1024 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(threads);
  hipLaunchKernelGGL((probe<NACC, LDSR>), grid, block, 0, 0, out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NACC, LDSR>), grid, block, 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int per_it = (12 / NACC) * NACC;
  const double waves_per_simd = threads / 64.0 * blocks_per_cu / 4.0;
  const double mfma_per_simd = (double)iters * per_it * waves_per_simd;
  const double tf = 256.0 * 4 * mfma_per_simd * 32768.0 / (ms * 1e-3) / 1e12;
  printf("%-28s threads=%d blocks/CU=%d waves/SIMD=%.0f  %.3f ms  %.0f TF  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", name, threads,
         blocks_per_cu, waves_per_simd, ms, tf, ms * 1e-3 * 2.4e9 / mfma_per_simd);
  hipFree(out);
}

int main() {
  run<4, 0>("4 accs, regs only", 256, 1);
  run<4, 0>("4 accs, regs only", 256, 2);
  run<4, 0>("4 accs, regs only", 512, 1);
  run<4, 0>("4 accs, regs only", 512, 2);
  run<2, 0>("2 accs, regs only", 256, 1);
  run<2, 0>("2 accs, regs only", 512, 1);
  run<1, 0>("1 acc, regs only", 256, 1);
  run<6, 0>("6 accs, regs only", 256, 1);
  run<12, 0>("12 accs, regs only", 256, 1);
  run<4, 1>("4 accs, 8 ds_read_b128/12", 256, 1);
  run<4, 1>("4 accs, 8 ds_read_b128/12", 256, 2);
  run<4, 1>("4 accs, 8 ds_read_b128/12", 512, 1);
  run<4, 1>("4 accs, 8 ds_read_b128/12", 512, 2);
  run<12, 1>("12 accs, 8 ds_read_b128/12", 256, 1);
  return 0;
}
