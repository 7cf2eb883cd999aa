// How fast do DEPENDENT v_mfma_f32_32x32x16_bf16 chains issue on gfx950?  NACC accumulators per wave (1 = one dependent chain,
// as in the bf16x3 K1), WPS waves per SIMD, NV fp32 VALU instructions (KIND 0: fma, 1: exp2) placed behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int NV, int KIND, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[NACC];
  bf16x8 a[4], b[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) { a[j][i] = (__bf16)(0.5f + 0.01f * lane + i + j); b[j][i] = (__bf16)(0.25f * lane - i - j); }
  float v[8];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16 / NACC; ++rep) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(rep + i) & 3], b[(rep * 3 + i) & 3], acc[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          const int c = (i * NV + e + rep) & 7;
          if (KIND == 0) v[c] = __builtin_fmaf(v[c], 0.999f, 1e-7f);
          else v[c] = __builtin_amdgcn_exp2f(v[c]);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC, int NV, int KIND, int WPS> void run(float* d) {
  const int iters = 1000, blocks = 256 * WPS * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  static unsigned long long* clk = nullptr;
  if (!clk) hipMalloc(&clk, 16384 * 16);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NACC, NV, KIND, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters, clk);
  hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<NACC, NV, KIND, WPS>), dim3(blocks), dim3(256), 0, 0, d, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  unsigned long long hc[4096];
  hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
  double cs = 0, rs = 0;
  for (int i = 0; i < 2048; ++i) { cs += hc[2 * i]; rs += hc[2 * i + 1]; }
  const double ghz = cs / rs * 0.1;       // s_memrealtime ticks at 100 MHz
  const double cyc_per_mfma = cs / 2048 / ((double)iters * 16.0) / WPS;   // per SIMD: WPS waves share it
  const double nm = (double)blocks * 4 * iters * 16.0;          // MFMAs
  const double per_simd = nm / 1024.0;
  printf("NACC=%d NV=%d kind=%d waves/SIMD=%d: %.2f ms  %.0f TFLOP/s; %.1f ns per MFMA per SIMD; in-kernel clock %.2f GHz, %.1f cycles per MFMA per SIMD\n", NACC, NV, KIND, WPS, ms,
         nm * 32768.0 / ms / 1e9, ms * 1e6 / per_simd, ghz, cyc_per_mfma);
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  run<4, 0, 0, 2>(d); run<1, 0, 0, 2>(d); run<1, 0, 0, 1>(d); run<2, 0, 0, 2>(d); run<4, 0, 0, 1>(d);
  run<1, 2, 0, 2>(d); run<1, 4, 0, 2>(d); run<1, 3, 1, 2>(d); run<2, 4, 0, 2>(d); run<1, 4, 0, 1>(d);
  return 0;
}
