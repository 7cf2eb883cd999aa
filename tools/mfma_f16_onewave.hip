// One wave per SIMD: do fp32 VALU instructions of the SAME wave hide behind its v_mfma_f32_32x32x16_f16?  (The eight-wave K1
// interleaves the two waves of a SIMD; a 512-register kernel would have to do it inside one instruction stream.)
// MODE 0: dependent MFMA chain into one accumulator; 1: two accumulators alternating.  NV fma (+ NE v_exp) behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NV, int NE, int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  f32x16 acc[2];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + lane + i); b[i] = (_Float16)(0.25f * lane - i); }
  float v[8];
  for (int j = 0; j < 16; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[MODE ? (i & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[MODE ? (i & 1) : 0], 0, 0, 0);
#pragma unroll
      for (int e = 0; e < NV; ++e) { const int c = (i * (NV + NE) + e) & 7; v[c] = __builtin_fmaf(v[c], 0.999f, 1e-7f); }
#pragma unroll
      for (int e = 0; e < NE; ++e) { const int c = (i * (NV + NE) + NV + e) & 7; v[c] = __builtin_amdgcn_exp2f(v[c]); }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (NV + NE) __builtin_amdgcn_sched_group_barrier(0x002, NV + NE, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (lds[0] ? 0.f : 0.f);
}
template <int NV, int NE, int MODE> void run(float* d) {
  const int iters = 2000, blocks = 256 * 4;
  hipFuncSetAttribute((const void*)k<NV, NE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NE, MODE>), dim3(blocks), dim3(256), 100 * 1024, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NE, MODE>), dim3(blocks), dim3(256), 100 * 1024, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = 4.0 * iters * 16;      // 4 rounds of workgroups per CU, one wave per SIMD each
  printf("one wave/SIMD, f16 32x32x16, %d fma + %d exp per MFMA, %s: %.2f ms = %.1f ns per MFMA and SIMD\n", NV, NE,
         MODE ? "two accumulators" : "one dependent chain", ms, ms * 1e6 / mfma_per_simd);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  run<0, 0, 0>(d); run<0, 0, 1>(d); run<2, 0, 1>(d); run<4, 0, 1>(d); run<5, 0, 1>(d); run<6, 0, 1>(d); run<8, 0, 1>(d);
  run<4, 1, 1>(d); run<4, 2, 1>(d); run<5, 1, 1>(d); run<6, 0, 0>(d);
  return 0;
}
