// Calibration 2: do independent fp32 VALU instructions hide behind v_mfma_f32_16x16x4_f32 on gfx950?
// Stream per wave: {1 MFMA, NV independent VALU} repeated; 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NV, int KIND, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  float a = 0.5f + lane, b = 0.25f * lane;
  float v[8];
  f32x2 w[8];
  for (int i = 0; i < 8; ++i) w[i] = f32x2{0.001f * lane, 0.002f * i};
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; v[i] = 0.001f * (lane + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          const int c = (i * NV + e) & 7;
          if (KIND == 0) v[c] = __builtin_fmaf(v[c], 0.999f, 1e-7f);
          else if (KIND == 1) v[c] = __builtin_amdgcn_exp2f(v[c]);
          else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[c]) : "v"(w[(c + 1) & 7]), "v"(w[(c + 2) & 7]));
          else if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(v[(c + 1) & 7]), "v"(v[(c + 2) & 7]));
          else if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[c]) : "v"(w[(c + 1) & 7]));
          else if (KIND == 6) asm volatile("v_log_f32 %0, %0" : "+v"(v[c]));
          else if (KIND == 7) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v[c]) : "v"(v[(c + 1) & 7]));
          else if (KIND == 8) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[c]));
          else v[c] = fmaxf(v[c] * 0.5f, -v[c]);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (NV && KIND < 3) __builtin_amdgcn_sched_group_barrier(0x002, NV * (KIND == 2 ? 2 : 1), 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i] + w[i][0] + w[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int KIND, int WPS> void run(float* d) {
  const int iters = 2000, blocks = 256 * WPS * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, KIND, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, KIND, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 32.0 * 2048.0;
  printf("NV=%d kind=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s (%.3f of 157.3)\n", NV, KIND, WPS, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  run<0, 0, 2>(d); run<2, 0, 2>(d);
  run<2, 3, 2>(d); run<4, 3, 2>(d);
  run<2, 4, 2>(d); run<4, 4, 2>(d);
  run<2, 5, 2>(d); run<4, 5, 2>(d);
  run<2, 6, 2>(d); run<2, 7, 2>(d); run<4, 7, 2>(d); run<2, 8, 2>(d);
  return 0;
}
