// Does fp32 VALU work of ONE wave hide behind the fp32 MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// 512-thread workgroups = 2 waves per SIMD: waves 0-3 run a pure MFMA stream, waves 4-7 a pure VALU stream
// (kind 0: v_fma_f32, kind 1: v_exp_f32) of `nv` instructions per MFMA of the other wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int nv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float res = 0;
  if (wave < 4) {
    f32x4 acc[8];
    float a = 0.5f + lane, b = 0.25f * lane;
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][3];
  } else {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
    const int n = iters * nv;   // 32 MFMAs per iteration in the other wave -> nv*32 VALU per iteration here
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (KIND == 0) v[i] = __builtin_fmaf(v[i], 0.999f, 1e-7f);
          else v[i] = __builtin_amdgcn_exp2f(v[i]);
        }
    }
    for (int i = 0; i < 8; ++i) res += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int KIND> void run(float* d, int nv) {
  const int iters = 4000, blocks = 256 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 0, 0, d, iters, nv);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 0, 0, d, iters, nv);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 32.0 * 2048.0;
  printf("kind %d, %d VALU per MFMA in the sibling wave: %.2f ms, MFMA waves %.1f TFLOP/s (%.3f of 157.3)\n", KIND, nv, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}
int main() {
  float* d; hipMalloc(&d, 1024 * 512 * 4);
  run<0>(d, 0); run<0>(d, 1); run<0>(d, 2); run<0>(d, 4); run<0>(d, 6);
  run<1>(d, 1); run<1>(d, 2);
  return 0;
}
