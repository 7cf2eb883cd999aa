// Probe: v_mfma_f32_16x16x4_f32 lane maps (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C col=l&15,row=(l>>4)*4+reg)
// and bit-equality with the k-ordered fmaf chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, const float* C, float* D, int steps) {
  int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f32x4 acc;
  for (int i = 0; i < 4; ++i) acc[i] = C[(q * 4 + i) * 16 + r];
  for (int s = 0; s < steps; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * (4 * steps) + 4 * s + q], B[(4 * s + q) * 16 + r], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(q * 4 + i) * 16 + r] = acc[i];
}
int main() {
  const int steps = 20, K = 4 * steps;
  std::vector<float> A(16 * K), B(K * 16), C(256), D(256);
  srand(2);
  auto rnd = []() { return (float)(rand() % 20001 - 10000) / 3000.f; };
  for (auto& v : A) v = rnd(); for (auto& v : B) v = rnd() * 2; for (auto& v : C) v = rnd() * 30;
  float *dA, *dB, *dC, *dD;
  (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 1024); (void)hipMalloc(&dD, 1024);
  (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, steps);
  (void)hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    float c = C[i * 16 + j];
    for (int kk = 0; kk < K; ++kk) c = fmaf(A[i * K + kk], B[kk * 16 + j], c);
    if (c != D[i * 16 + j]) ++bad;
  }
  printf("mfma16x16x4 probe: not-bit-equal-to-fmaf-chain=%d of 256\n", bad);
  return 0;
}
