// Probe: v_mfma_f64_16x16x4_f64 lane maps: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C col=l&15,row=(l>>4)+4*reg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, int steps) {
  int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  f64x4 acc = {0, 0, 0, 0};
  for (int s = 0; s < steps; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[r * (4 * steps) + 4 * s + q], B[(4 * s + q) * 16 + r], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(q + 4 * i) * 16 + r] = acc[i];
}
int main() {
  const int steps = 4, K = 4 * steps;
  std::vector<double> A(16 * K), B(K * 16), D(256);
  srand(3);
  for (auto& v : A) v = (rand() % 2001 - 1000) / 64.0; for (auto& v : B) v = (rand() % 2001 - 1000) / 32.0;
  double *dA, *dB, *dD;
  (void)hipMalloc(&dA, A.size() * 8); (void)hipMalloc(&dB, B.size() * 8); (void)hipMalloc(&dD, 2048);
  (void)hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
  (void)hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double c = 0; for (int kk = 0; kk < K; ++kk) c += A[i * K + kk] * B[kk * 16 + j];
    if (c != D[i * 16 + j]) ++bad;
  }
  printf("mfma f64 16x16x4 probe: mismatches=%d of 256 (exact small-integer data)\n", bad);
  return 0;
}
