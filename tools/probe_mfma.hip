// Probe: verify v_mfma_f32_32x32x2_f32 operand/result lane maps and that the result is
// bit-for-bit a k-ordered fmaf chain (k = lane>>5 order 0 then 1), C as the chain start.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, const float* C, float* D, int steps) {
  int lane = threadIdx.x;
  int r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    acc[i] = C[row * 32 + r];
  }
  for (int s = 0; s < steps; ++s) {
    float a = A[r * (2 * steps) + 2 * s + h];   // A[i=r][k=2s+h]
    float b = B[(2 * s + h) * 32 + r];          // B[k][j=r]
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    D[row * 32 + r] = acc[i];
  }
}
int main() {
  const int steps = 40, K = 2 * steps;
  std::vector<float> A(32 * K), B(K * 32), C(32 * 32), D(32 * 32);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : A) v = rnd() * 3;
  for (auto& v : B) v = rnd() * 5;
  for (auto& v : C) v = rnd() * 100;
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, steps);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad_chain = 0, bad_tol = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float c = C[i * 32 + j]; double ref = c;
    for (int kk = 0; kk < K; ++kk) { c = fmaf(A[i * K + kk], B[kk * 32 + j], c); ref += (double)A[i * K + kk] * B[kk * 32 + j]; }
    if (c != D[i * 32 + j]) ++bad_chain;
    if (fabs(ref - D[i * 32 + j]) > 1e-3) ++bad_tol;
  }
  printf("mfma32x32x2 probe: not-bit-equal-to-fmaf-chain=%d of 1024, out-of-tolerance=%d\n", bad_chain, bad_tol);
  return bad_tol != 0;
}
