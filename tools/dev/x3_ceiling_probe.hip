// What can an x3 (split-fp16, three MFMAs per product) inner loop sustain on THIS box when nothing but the matrix pipe and its operand
// reads is in the way?  The ceiling the conv path's roofline fraction is argued against, measured instead of quoted:
//
//   variant 0  bare `v_mfma_f32_32x32x16_f16` loop, RANDOM fp16 operands resident in registers, 4 independent accumulators per wave
//              (the matrix pipe alone: clock under matrix load = the power budget)
//   variant 1  the x3 product loop of the halo kernels with its operands resident in LDS and NOTHING else: per 16-channel K step and
//              tap a wave reads 2 x (a_hi, a_lo) pixel fragments and 2 x (b_hi, b_lo) weight fragments (8 ds_read_b128, conflict-free
//              80-byte pitch) and issues 12 MFMAs term-major (a_lo.b_hi, a_hi.b_hi, a_hi.b_lo on four 32x32 accumulators) -- no
//              fp32 -> 2 x fp16 conversion, no HBM / L2 traffic, no epilogue, random data
//   variant 2  variant 1 + the conversion's arithmetic in the MFMA shadows (per gap two scalar-f32 VALU on private registers: the
//              affine + split of the streaming kernel without its memory side)
//   variant 3  the MX-fp8 lever priced (oracle/numerics_gate.py says the numerics pass): per four 16-channel K steps and tap the a_hi.b_hi term
//              stays 4 x 4 fp16 MFMAs from LDS operands (as in variant 1, without the lo fragment reads), the two LOW terms become
//              2 x 4 `v_mfma_scale_f32_32x32x64_f8f6f4` (e4m3 operands of K = 64, unit block scales) on register-resident random fp8
//              operands: 16 + 8 MFMA issues per 4 K steps instead of 48.  x3-equivalent = fp32-grade products per second x 2.
//
// Every block records s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop: clock = cycles / ticks.  Built as a shared
// object (extern "C" x3_probe_run) that bench.py loads for `roofline.practical_peak`; `tools/dev/x3_ceiling.py` prints the table that
// is committed as profiles/r06_x3_ceiling_probe.txt.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC x3_ceiling_probe.hip -o libx3_ceiling_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ _Float16 rnd_half(uint32_t i) {      // uniform in [-2, 2): every mantissa bit toggles
  return (_Float16)(((float)(hash32(i) & 0xffffu) - 32768.0f) * (1.0f / 16384.0f));
}

constexpr int A_ROWS = 340;          // the halo of an 8 x 32 tile
constexpr int B_ROWS = 9 * 64;       // 9 taps x 64 couts
constexpr int PITCH = 80;            // bytes per row: [hi16 | lo16 | pad] -- conflict-free ds_read_b128

template <int VARIANT>
__global__ __launch_bounds__(256, 2) void x3_probe_kernel(float* __restrict__ out, unsigned long long* __restrict__ clk, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[(A_ROWS + B_ROWS) * PITCH];
  for (int i = threadIdx.x; i < (A_ROWS + B_ROWS) * PITCH / 2; i += 256)
    reinterpret_cast<_Float16*>(lds)[i] = rnd_half(seed + blockIdx.x * 65536u + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, kh = lane >> 5;
  f16v acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  h8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ah[j][e] = rnd_half(seed ^ (lane * 64 + j * 8 + e));
      al[j][e] = rnd_half(seed ^ (lane * 64 + 16 + j * 8 + e)) * (_Float16)0.0009765625f;
      bh[j][e] = rnd_half(seed ^ (lane * 64 + 32 + j * 8 + e));
      bl[j][e] = rnd_half(seed ^ (lane * 64 + 48 + j * 8 + e)) * (_Float16)0.0009765625f;
    }
  }
  v8i qa[2], qb[2];      // variant 3: random fp8 (e4m3) operand bytes, K = 64 per lane pair
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qa[j][e] = (int)(hash32(seed * 3u + lane * 32 + j * 8 + e) & 0x77777777u);      // exponents below the NaN / inf patterns
      qb[j][e] = (int)(hash32(seed * 5u + lane * 32 + 16 + j * 8 + e) & 0x77777777u);
    }
  float va = 1.0f + lane * 1e-3f, vb = 0.5f;
  unsigned long long c0 = 0, t0 = 0;
  if (threadIdx.x == 0) {
    c0 = __builtin_amdgcn_s_memtime();
    t0 = __builtin_amdgcn_s_memrealtime();
  }
  if (VARIANT == 3) {
    // one iteration = one tap over FOUR 16-channel K steps: 4 x 4 fp16 MFMAs (a_hi . b_hi) + 2 x 4 fp8 MFMAs of K = 64 (the two low terms)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int arow = wave * 64 + row + (tap % 3) + 34 * (tap / 3) + ((it + ks) & 3);
          const int brow = tap * 64 + row;
          const unsigned char* pa = lds + arow * PITCH + kh * 16;
          const unsigned char* pb = lds + (A_ROWS + brow) * PITCH + kh * 16 + (ks & 1) * 32;
          ah[0] = *reinterpret_cast<const h8*>(pa);
          ah[1] = *reinterpret_cast<const h8*>(pa + 32 * PITCH);
          bh[0] = *reinterpret_cast<const h8*>(pb);
          bh[1] = *reinterpret_cast<const h8*>(pb + 32 * PITCH);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[a >> 1], bh[a & 1], acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int term = 0; term < 2; ++term)
#pragma unroll
          for (int a = 0; a < 4; ++a)
            acc[a] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa[(a >> 1) ^ term], qb[(a & 1) ^ term], acc[a], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    }
  } else
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (VARIANT >= 1) {
        // pixel rows of this wave: 64 consecutive halo pixels displaced by the tap (and by `it`, so that no read is loop-invariant)
        const int arow = wave * 64 + row + (tap % 3) + 34 * (tap / 3) + (it & 3);      // <= 296: the second fragment (+32 rows) stays inside
        const int brow = tap * 64 + row;
        const unsigned char* pa = lds + arow * PITCH + kh * 16;
        const unsigned char* pb = lds + (A_ROWS + brow) * PITCH + kh * 16;
        ah[0] = *reinterpret_cast<const h8*>(pa);
        al[0] = *reinterpret_cast<const h8*>(pa + 32);
        ah[1] = *reinterpret_cast<const h8*>(pa + 32 * PITCH);
        al[1] = *reinterpret_cast<const h8*>(pa + 32 * PITCH + 32);
        bh[0] = *reinterpret_cast<const h8*>(pb);
        bl[0] = *reinterpret_cast<const h8*>(pb + 32);
        bh[1] = *reinterpret_cast<const h8*>(pb + 32 * PITCH);
        bl[1] = *reinterpret_cast<const h8*>(pb + 32 * PITCH + 32);
      }
      // term-major: consecutive MFMAs never share an accumulator
#pragma unroll
      for (int term = 0; term < 3; ++term) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const h8 fa = term == 0 ? al[a >> 1] : ah[a >> 1];
          const h8 fb = term == 2 ? bl[a & 1] : bh[a & 1];
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[a], 0, 0, 0);
          if (VARIANT == 2) {      // two scalar-f32 VALU per gap on private registers (the conversion's affine + residual, no memory side)
            asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %1, %1, %0, %0" : "+v"(va), "+v"(vb));
          }
        }
      }
    }
  }
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2 + 0] = __builtin_amdgcn_s_memtime() - c0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - t0;
  }
  float s = va + vb;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// res[0] = ms, res[1] = raw MFMA TFLOP/s, res[2] = in-kernel clock in GHz (median over blocks), res[3] = MFMA issue cycles per MFMA and SIMD
extern "C" int x3_probe_run(int variant, int blocks_per_cu, int iters, double* res) {
  int dev = 0;
  hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * blocks_per_cu;
  float* out = nullptr;
  unsigned long long* clk = nullptr;
  if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)) != hipSuccess) return -2;
  if (hipMalloc(&clk, (size_t)blocks * 2 * sizeof(unsigned long long)) != hipSuccess) return -2;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto launch = [&](int n) {
    switch (variant) {
      case 0: hipLaunchKernelGGL(x3_probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 12345u); break;
      case 1: hipLaunchKernelGGL(x3_probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 12345u); break;
      case 2: hipLaunchKernelGGL(x3_probe_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 12345u); break;
      default: hipLaunchKernelGGL(x3_probe_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, clk, n, 12345u); break;
    }
  };
  launch(iters / 8 + 1);      // warm-up (clocks, caches)
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  launch(iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) return -3;
  unsigned long long* h = (unsigned long long*)malloc((size_t)blocks * 2 * sizeof(unsigned long long));
  hipMemcpy(h, clk, (size_t)blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double cyc = 0, ticks = 0;
  for (int b = 0; b < blocks; ++b) {
    cyc += (double)h[b * 2];
    ticks += (double)h[b * 2 + 1];
  }
  free(h);
  // variant 3: an iteration covers FOUR K steps per tap -- 16 fp16 MFMAs + 8 fp8 MFMAs of four times the K: `raw` counts what x3 would have
  // issued for the same products (48 fp16 MFMAs), so that raw / 3 stays "fp32-grade products per second x 2" across variants
  const double mfma_per_wave = (double)iters * 9 * (variant == 3 ? 48 : 12);
  const double waves = (double)blocks * 4;
  const double flops = waves * mfma_per_wave * 2.0 * 32 * 32 * 16;
  res[0] = ms;
  res[1] = flops / (ms * 1e-3) / 1e12;
  res[2] = ticks > 0 ? cyc / ticks * 0.1 : 0.0;                  // cycles per 10 ns -> GHz
  res[3] = (cyc / blocks) / (mfma_per_wave * blocks_per_cu);     // shader cycles per MFMA and SIMD (blocks_per_cu waves share a SIMD)
  hipFree(out);
  hipFree(clk);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return 0;
}
