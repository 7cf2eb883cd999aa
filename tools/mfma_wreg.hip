// Calibration 3: the pdf-major inner loop in isolation -- W (80 VGPRs) and x (20 VGPRs) in registers, 4 accumulator
// chains, 80 MFMAs per tile; optional per-tile VALU (squaring).  Occupancy set by a dummy LDS allocation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SQ, int CH>
__global__ __launch_bounds__(256, 2) void k(float* out, int ntiles, int lds_floats, int getenv_rand) {
  extern __shared__ float dummy[];
  const int lane = threadIdx.x & 63;
  if (lds_floats > 0 && threadIdx.x == 0) dummy[0] = 1.0f;
  float wr[CH][20], xb[20];
  f32x4 gc[CH], acc[CH];
  for (int b = 0; b < CH; ++b) { gc[b] = f32x4{0.f, 1.f, 2.f, 3.f}; for (int s = 0; s < 20; ++s) wr[b][s] = 0.001f * (lane + b + s); }
  for (int s = 0; s < 20; ++s) xb[s] = 0.01f * (lane + s);
  if (getenv_rand) {   // realistic operand values: data toggling costs power
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    for (int b = 0; b < CH; ++b) for (int s = 0; s < 20; ++s) { h = h * 1664525u + 1013904223u; wr[b][s] = ((int)(h >> 8) - (1 << 23)) * (3.0f / (1 << 23)); }
    for (int s = 0; s < 20; ++s) { h = h * 1664525u + 1013904223u; xb[s] = ((int)(h >> 8) - (1 << 23)) * (2.0f / (1 << 23)); }
  }
  float vsum = 0.f;
  const bool sq = (lane >> 4) & 2;
  for (int t = 0; t < ntiles; ++t) {
    if (SQ) {
#pragma unroll
      for (int s = 0; s < 20; ++s) xb[s] = xb[s] * (sq ? -1.0f : 1.0f);
    }
#pragma unroll
    for (int s = 0; s < 20; ++s)
#pragma unroll
      for (int b = 0; b < CH; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[b][s], xb[s], s == 0 ? gc[b] : acc[b], 0, 0, 0);
    float v = 0;
    for (int b = 0; b < CH; ++b) v += acc[b][b & 3];
    vsum += v;
  }
  out[(blockIdx.x & 4095) * 256 + threadIdx.x] = vsum;
}
template <int SQ, int CH> void run(float* d, int wgs_per_cu) {
  const int ntiles = getenv("NT") ? atoi(getenv("NT")) : 2000;
  const int lds = wgs_per_cu == 1 ? 100 * 1024 : wgs_per_cu == 2 ? 64 * 1024 : wgs_per_cu == 3 ? 48 * 1024 : 8 * 1024;
  hipFuncSetAttribute((const void*)k<SQ, CH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int blocks = getenv("BL") ? atoi(getenv("BL")) : 256 * wgs_per_cu * 2;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SQ, CH>), dim3(blocks), dim3(256), lds, 0, d, ntiles, lds / 4, getenv("RAND") ? 1 : 0);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<SQ, CH>), dim3(blocks), dim3(256), lds, 0, d, ntiles, lds / 4, getenv("RAND") ? 1 : 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * ntiles * 20.0 * CH * 2048.0;
  printf("SQ=%d chains=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s (%.3f of 157.3)\n", SQ, CH, wgs_per_cu, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  if (getenv("NT")) { run<1, 4>(d, 3); run<1, 4>(d, 3); return 0; }
  run<0, 4>(d, 1); run<0, 4>(d, 2); run<0, 4>(d, 3);
  run<1, 4>(d, 1); run<1, 4>(d, 2); run<1, 4>(d, 3);
  run<0, 2>(d, 2); run<0, 8>(d, 2);
  return 0;
}
