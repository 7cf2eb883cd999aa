// Does fp32 VALU work hide behind BF16 MFMAs (v_mfma_f32_32x32x16_bf16) on gfx950?  Same shape as mfma_coissue.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV, int KIND>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.5f + lane + i); b[i] = (__bf16)(0.25f * lane - i); }
  float v[8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          const int c = (i * NV + e) & 7;
          if (KIND == 0) v[c] = __builtin_fmaf(v[c], 0.999f, 1e-7f);
          else v[c] = __builtin_amdgcn_exp2f(v[c]);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int KIND> void run(float* d) {
  const int iters = 2000, blocks = 256 * 2 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 16.0 * (2.0 * 32 * 32 * 16);
  printf("bf16 32x32x16, NV=%d kind=%d: %.2f ms  %.1f TFLOP/s (%.3f of 2500)\n", NV, KIND, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  run<0, 0>(d); run<1, 0>(d); run<2, 0>(d); run<4, 0>(d); run<6, 0>(d); run<8, 0>(d);
  run<1, 1>(d); run<2, 1>(d);
  return 0;
}
