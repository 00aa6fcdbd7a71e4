// Micro-probe: do matrix (MFMA) and vector (VALU) instructions of DIFFERENT waves on one SIMD overlap, and do they overlap
// inside ONE wave?  Blocks of 256 threads (one wave per SIMD) and 512 threads (two waves per SIMD); a wave's role is picked by
// its index: role M issues only v_mfma_f32_32x32x16_f16 (4 independent accumulators), role V only v_fma_f32 chains (8
// independent chains), role X interleaves both in one instruction stream.
// hipcc --offload-arch=gfx950 -O3 coissue_probe.hip -o coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode bits per wave-in-SIMD slot: 1 = MFMA, 2 = VALU, 3 = both interleaved, 0 = idle (wave exits)
template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters, int mode_slot0, int mode_slot1, int map, int n_valu) {
  const int wave = threadIdx.x >> 6;
  const int slot = map == 0 ? (wave >> 2) : (wave & 1);   // map 0: waves 0-3 / 4-7 get the two roles (wave w on SIMD w % 4); map 1: even / odd waves
  const int mode = slot == 0 ? mode_slot0 : mode_slot1;
  const int lane = threadIdx.x & 63;
  f16v acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  h8 fa, fb;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    fa[k] = (_Float16)(0.01f * ((lane + k) & 7));
    fb[k] = (_Float16)(0.02f * ((lane * 3 + k) & 7));
  }
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.001f * (lane + k);
  const float m = 1.0000001f, c = 1e-9f;
  if (mode == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[t & 3], 0, 0, 0);
    }
  } else if (mode == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(m), "v"(c));
          if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
          if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
          if (KIND == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[k]));
          if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(v[k]) : "v"(c));
          if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&v[k & 6])) : "v"(*reinterpret_cast<const double*>(&v[0])));
        }
    }
  } else if (mode == 3) {      // per MFMA: n_valu / n_mfma VALU instructions behind it in the same stream (8 MFMAs + 8*k VALU per iteration)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[t & 3], 0, 0, 0);
        if (n_valu >= 1) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (k < n_valu) v[k] = __builtin_fmaf(v[k], m, c);          // (plain C: inline asm behind an MFMA makes the compiler pad with s_nop)
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND = 0>
static float run(const char* name, int threads, int m0, int m1, int n_valu, double mfma_per_it, double valu_per_it, int map = 0) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, out, 200, m0, m1, map, n_valu);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, out, iters, m0, m1, map, n_valu);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_it = ms * 1e6 / iters;
  printf("%-58s %8.3f ms  %7.1f ns / iteration  (MFMA-only bound %.0f cycles, VALU-only bound %.0f cycles per iteration)\n", name, ms, ns_per_it,
         mfma_per_it * 32, valu_per_it * 4);
  (void)hipFree(out);
  return ms;
}

int main() {
  run("1 wave/SIMD: 8 MFMA per iteration", 256, 1, 0, 0, 8, 0);
  run("1 wave/SIMD: 64 VALU per iteration", 256, 2, 0, 0, 0, 64);
  run("2 waves/SIMD: MFMA wave + VALU wave (8 MFMA | 64 VALU)", 512, 1, 2, 0, 8, 64);
  run("2 waves/SIMD (even/odd roles): MFMA wave + VALU wave", 512, 1, 2, 0, 8, 64, 1);
  run<1>("1 wave/SIMD: 64 v_add_f32", 256, 2, 0, 0, 0, 64);
  run<1>("2 waves/SIMD: v_add wave + v_add wave", 512, 2, 2, 0, 0, 128);
  run<1>("2 waves/SIMD: MFMA wave + v_add_f32 wave", 512, 1, 2, 0, 8, 64);
  run<2>("1 wave/SIMD: 64 v_exp_f32", 256, 2, 0, 0, 0, 64);
  run<2>("2 waves/SIMD: v_exp wave + v_exp wave", 512, 2, 2, 0, 0, 128);
  run<2>("2 waves/SIMD: MFMA wave + v_exp_f32 wave", 512, 1, 2, 0, 8, 64);
  run<3>("1 wave/SIMD: 64 v_cvt_f16_f32", 256, 2, 0, 0, 0, 64);
  run<3>("2 waves/SIMD: MFMA wave + v_cvt_f16_f32 wave", 512, 1, 2, 0, 8, 64);
  run<4>("1 wave/SIMD: 64 v_mov_b32", 256, 2, 0, 0, 0, 64);
  run<4>("2 waves/SIMD: v_mov wave + v_mov wave", 512, 2, 2, 0, 0, 128);
  run<4>("2 waves/SIMD: MFMA wave + v_mov_b32 wave", 512, 1, 2, 0, 8, 64);
  run("2 waves/SIMD: MFMA wave + MFMA wave", 512, 1, 1, 0, 16, 0);
  run("2 waves/SIMD: VALU wave + VALU wave", 512, 2, 2, 0, 0, 128);
  for (int k = 1; k <= 8; ++k) {
    char name[96];
    snprintf(name, sizeof(name), "1 wave/SIMD: each MFMA followed by %d VALU (one stream)", k);
    run(name, 256, 3, 0, k, 8, 8.0 * k);
  }
  for (int k = 2; k <= 8; k += 2) {
    char name[96];
    snprintf(name, sizeof(name), "2 waves/SIMD: both interleave MFMA + %d VALU", k);
    run(name, 512, 3, 3, k, 16, 16.0 * k);
  }
  return 0;
}
