// Micro-probe (round 4): how many single-issue instructions does ONE wave hide in the shadow of its own MFMAs?
// The round-3 probe (coissue_probe.hip) only answered the two-wave case cleanly: its one-stream mode compiled a runtime
// `k < n_valu` select around every filler, so every variant issued 16 VALU per MFMA.  Here the loop body is ONE asm block per
// variant (the compiler can neither pad it with s_nop nor reorder it): 8 x { v_mfma_f32_32x32x16_f16 on a rotating
// accumulator ; NF fillers of one kind on registers nobody else touches }.
//   kinds: fma = v_fma_f32, exp = v_exp_f32, cvt = v_cvt_pk_f16_f32 (hmm: v_cvt_pkrtz is the legacy form), pk = v_pk_fma_f32,
//          add = v_add_f32, dsr = ds_read_b128 (conflict-free, waited for once per iteration), dsw = ds_write_b64,
//          mix = the halo kernel's staging mix per MFMA gap: 1 fma + 1 cvt + (exp | rcp alternating)
// Blocks of 256 threads (one wave per SIMD) and 512 threads (two per SIMD), one block per CU, 256 blocks.
// hipcc --offload-arch=gfx950 -O3 coissue_probe2.hip -o coissue_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define M(i) "v_mfma_f32_32x32x16_f16 %[a" #i "], %[fa], %[fb], %[a" #i "]\n"
#define FMA(i) "v_fma_f32 %[v" #i "], %[v" #i "], %[m], %[c]\n"
#define ADD(i) "v_add_f32 %[v" #i "], %[v" #i "], %[c]\n"
#define EXP(i) "v_exp_f32 %[v" #i "], %[v" #i "]\n"
#define RCP(i) "v_rcp_f32 %[v" #i "], %[v" #i "]\n"
#define CVT(i) "v_cvt_pk_f16_f32 %[v" #i "], %[v" #i "], %[c]\n"
#define PKF(i) "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[pm], %[pm]\n"
#define DSR(i) "ds_read_b128 %[q" #i "], %[la] offset:" #i "*1024\n"
#define DSW(i) "ds_write_b64 %[la], %[p" #i "] offset:" #i "*1024\n"

#define F0(X)
#define F1(X) X(0)
#define F2(X) X(0) X(1)
#define F3(X) X(0) X(1) X(2)
#define F4(X) X(0) X(1) X(2) X(3)
#define F5(X) X(0) X(1) X(2) X(3) X(4)
#define F6(X) X(0) X(1) X(2) X(3) X(4) X(5)
#define F8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// the halo kernel's staging mix: per gap ~1.3 plain VALU + 0.45 transcendentals + 0.1 ds_write + 0.67 ds_read
#define MIXA FMA(0) CVT(1) EXP(2)
#define MIXB FMA(3) CVT(4) RCP(5)
#define MIXC FMA(0) FMA(3) CVT(1) EXP(2)
#define MIXD FMA(0) FMA(3) CVT(4) RCP(5)

#define OPERANDS                                                                                                          \
  : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3]), [v0] "+v"(v[0]), [v1] "+v"(v[1]),         \
    [v2] "+v"(v[2]), [v3] "+v"(v[3]), [v4] "+v"(v[4]), [v5] "+v"(v[5]), [v6] "+v"(v[6]), [v7] "+v"(v[7]),                 \
    [p0] "+v"(pk[0]), [p1] "+v"(pk[1]), [p2] "+v"(pk[2]), [p3] "+v"(pk[3]), [p4] "+v"(pk[4]), [p5] "+v"(pk[5]),           \
    [p6] "+v"(pk[6]), [p7] "+v"(pk[7]), [q0] "+v"(q[0]), [q1] "+v"(q[1]), [q2] "+v"(q[2]), [q3] "+v"(q[3]),               \
    [q4] "+v"(q[4]), [q5] "+v"(q[5]), [q6] "+v"(q[6]), [q7] "+v"(q[7])                                                    \
  : [fa] "v"(fa), [fb] "v"(fb), [m] "v"(m), [c] "v"(c), [pm] "v"(pm), [la] "v"(la)                                        \
  : "memory"

#define BODY8(G0, G1, G2, G3, G4, G5, G6, G7)                                                                             \
  asm volatile(M(0) G0 M(1) G1 M(2) G2 M(3) G3 M(0) G4 M(1) G5 M(2) G6 M(3) G7 "s_waitcnt lgkmcnt(0)\n" OPERANDS)
#define BODY(G) BODY8(G, G, G, G, G, G, G, G)

enum { K_NONE, K_FMA1, K_FMA2, K_FMA3, K_FMA4, K_FMA5, K_FMA6, K_FMA8, K_ADD4, K_EXP1, K_EXP2, K_EXP3, K_CVT2, K_CVT4, K_PK1, K_PK2, K_PK4,
       K_DSR1, K_DSR2, K_DSW1, K_DSW2, K_MIX3, K_MIX4, K_MIX3R, K_MIX4RW, K_NOMFMA_MIX3, K_COUNT };
static const char* kind_name[K_COUNT] = {"MFMA only", "1 v_fma_f32", "2 v_fma_f32", "3 v_fma_f32", "4 v_fma_f32", "5 v_fma_f32", "6 v_fma_f32",
  "8 v_fma_f32", "4 v_add_f32", "1 v_exp_f32", "2 v_exp_f32", "3 v_exp_f32", "2 v_cvt_pk_f16_f32", "4 v_cvt_pk_f16_f32", "1 v_pk_fma_f32",
  "2 v_pk_fma_f32", "4 v_pk_fma_f32", "1 ds_read_b128", "2 ds_read_b128", "1 ds_write_b64", "2 ds_write_b64", "mix: fma + cvt + exp|rcp",
  "mix: 2 fma + cvt + exp|rcp", "mix3 + 1 ds_read_b128 (2 of 3 gaps)", "mix4 + ds_read_b128 (2/3) + ds_write_b64 (1/8)",
  "mix3 WITHOUT the MFMAs (price of the fillers alone)"};
static const double kind_fill[K_COUNT] = {0, 1, 2, 3, 4, 5, 6, 8, 4, 1, 2, 3, 2, 4, 1, 2, 4, 1, 2, 1, 2, 3, 4, 3.67, 4.8, 3};

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 8 * 1024 + 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  double pk[8];
  f4v q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = 0.001f * (lane + k);
    pk[k] = 1.0 + 1e-3 * lane;
    q[k] = f4v{0.f, 0.f, 0.f, 0.f};
  }
  const float m = 1.0000001f, c = 1e-9f;
  const double pm = 1.0;
  for (int i = threadIdx.x; i < (int)sizeof(lds) / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.f;
  __syncthreads();
  const unsigned la = (unsigned)(size_t)(lds + (wave & 7) * 8192 + lane * 16) & 0xffff;      // conflict-free b128 rows; offsets i*1024 stay inside the wave's 8 KB
  for (int it = 0; it < iters; ++it) {
    if (KIND == K_NONE) BODY();
    if (KIND == K_FMA1) BODY(F1(FMA));
    if (KIND == K_FMA2) BODY(F2(FMA));
    if (KIND == K_FMA3) BODY(F3(FMA));
    if (KIND == K_FMA4) BODY(F4(FMA));
    if (KIND == K_FMA5) BODY(F5(FMA));
    if (KIND == K_FMA6) BODY(F6(FMA));
    if (KIND == K_FMA8) BODY(F8(FMA));
    if (KIND == K_ADD4) BODY(F4(ADD));
    if (KIND == K_EXP1) BODY(F1(EXP));
    if (KIND == K_EXP2) BODY(F2(EXP));
    if (KIND == K_EXP3) BODY(F3(EXP));
    if (KIND == K_CVT2) BODY(F2(CVT));
    if (KIND == K_CVT4) BODY(F4(CVT));
    if (KIND == K_PK1) BODY(F1(PKF));
    if (KIND == K_PK2) BODY(F2(PKF));
    if (KIND == K_PK4) BODY(F4(PKF));
    if (KIND == K_DSR1) BODY8(DSR(0), DSR(1), DSR(2), DSR(3), DSR(4), DSR(5), DSR(6), DSR(7));
    if (KIND == K_DSR2) BODY8(DSR(0) DSR(1), DSR(2) DSR(3), DSR(4) DSR(5), DSR(6) DSR(7), DSR(0) DSR(1), DSR(2) DSR(3), DSR(4) DSR(5), DSR(6) DSR(7));
    if (KIND == K_DSW1) BODY8(DSW(0), DSW(1), DSW(2), DSW(3), DSW(4), DSW(5), DSW(6), DSW(7));
    if (KIND == K_DSW2) BODY8(DSW(0) DSW(1), DSW(2) DSW(3), DSW(4) DSW(5), DSW(6) DSW(7), DSW(0) DSW(1), DSW(2) DSW(3), DSW(4) DSW(5), DSW(6) DSW(7));
    if (KIND == K_MIX3) BODY8(MIXA, MIXB, MIXA, MIXB, MIXA, MIXB, MIXA, MIXB);
    if (KIND == K_MIX4) BODY8(MIXC, MIXD, MIXC, MIXD, MIXC, MIXD, MIXC, MIXD);
    if (KIND == K_MIX3R) BODY8(MIXA DSR(0), MIXB DSR(1), MIXA, MIXB DSR(2), MIXA DSR(3), MIXB, MIXA DSR(4), MIXB DSR(5));
    if (KIND == K_MIX4RW) BODY8(MIXC DSR(0), MIXD DSR(1), MIXC DSW(7), MIXD DSR(2), MIXC DSR(3), MIXD, MIXC DSR(4), MIXD DSR(5));
    if (KIND == K_NOMFMA_MIX3) asm volatile(MIXA MIXB MIXA MIXB MIXA MIXB MIXA MIXB OPERANDS);
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[a][r];
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k] + (float)pk[k] + q[k][0] + q[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(int threads) {
  float* out;
  (void)hipMalloc(&out, 256 * 512 * sizeof(float));
  const int iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, out, 200);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double ns_per_mfma = ms * 1e6 / iters / 8 / (threads / 256);      // per MFMA of one SIMD
  printf("%d wave(s)/SIMD  %-52s %6.2f fillers/gap  %7.2f ns per MFMA and SIMD\n", threads / 256, kind_name[KIND], kind_fill[KIND], ns_per_mfma);
  (void)hipFree(out);
}

template <int KIND>
static void run_all() {
  if constexpr (KIND < K_COUNT) {
    run<KIND>(256);
    run<KIND>(512);
    run_all<KIND + 1>();
  }
}

int main() {
  run_all<0>();
  return 0;
}
