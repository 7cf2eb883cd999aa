// tools/probe_bf16x3.hip -- can fp32 log-likelihood dot products run on the bf16 matrix cores at fp32 accuracy?
//
// (1) checks the operand / result lane layout of v_mfma_f32_32x32x16_bf16 assumed by the K1 bf16x3 kernel;
// (2) evaluates s = g + sum_k w[k] * x[k]  (K = 80: the [M | -V/2] . [x | x^2] contraction of one Gaussian and one
//     frame at D = 40) three ways and compares each with an fp64 evaluation, normalised by B = |g| + sum |w x|:
//       chain   the fp32 fmaf chain the fp32-MFMA kernels compute today
//       x3_6    both operands split EXACTLY into three bf16 pieces (w = w1 + w2 + w3, x = x1 + x2 + x3); the six
//               partial products of weight >= 2^-16 (x1w3, x2w2, x3w1, x1w2, x2w1, x1w1 -- small first) accumulated in
//               the fp32 accumulator of the bf16 MFMA
//       x3_9    all nine partial products
// hipcc --offload-arch=gfx950 -O2 -o probe_bf16x3 probe_bf16x3.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float v, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)v;
  float r = v - (float)a;
  b = (__bf16)r;
  r = r - (float)b;
  c = (__bf16)r;
}

// one wave: C[32][32] = G[32] (bias per row) + W[32][K] . X[K][32], K multiple of 16
// W row-major [32][K], X column-major-by-frame: X[j][k] (frame j), out[i][j]
template <int NPROD>
__global__ void k_x3(const float* W, const float* X, const float* G, int K, float* out, float* resid) {
  const int l = threadIdx.x, rc = l & 31, kb = l >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = G[8 * (r >> 2) + 4 * kb + (r & 3)];
  // pass p = 0..NPROD-1 in small-to-large order
  const int wa[9] = {2, 1, 0, 1, 0, 0, 2, 2, 1};   // which piece of w
  const int xa[9] = {0, 1, 2, 0, 1, 0, 2, 1, 2};   // which piece of x   (first six: x1w3 x2w2 x3w1 x1w2 x2w1 x1w1)
  // order small first: products (w3x1, w2x2, w1x3), (w2x1, w1x2), (w1x1); for 9: the three tiny ones go first
  int order[9]; int n = 0;
  if (NPROD == 9) { order[n++] = 6; order[n++] = 7; order[n++] = 8; }
  for (int p = 0; p < 6; ++p) order[n++] = p;
  float worst = 0.f;
  for (int q = 0; q < n; ++q) {
    const int p = order[q];
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 a, b;
      for (int e = 0; e < 8; ++e) {
        __bf16 p0, p1, p2;
        split3(W[rc * K + k0 + 8 * kb + e], p0, p1, p2);
        if (q == 0) worst = fmaxf(worst, fabsf(W[rc * K + k0 + 8 * kb + e] - ((float)p0 + (float)p1 + (float)p2)));
        a[e] = wa[p] == 0 ? p0 : wa[p] == 1 ? p1 : p2;
        split3(X[rc * K + k0 + 8 * kb + e], p0, p1, p2);
        b[e] = xa[p] == 0 ? p0 : xa[p] == 1 ? p1 : p2;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) out[(8 * (r >> 2) + 4 * kb + (r & 3)) * 32 + rc] = acc[r];
  resid[l] = worst;
}

int main() {
  const int K = 80, TRIALS = 200;
  std::vector<float> W(32 * K), X(32 * K), G(32), out(32 * 32), res(64);
  float *dW, *dX, *dG, *dO, *dR;
  hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&dG, 128); hipMalloc(&dO, out.size() * 4); hipMalloc(&dR, 256);
  // ---- (1) layout: small integers, exact in every arithmetic ----
  srand(1);
  for (auto& v : W) v = (float)(rand() % 7 - 3);
  for (auto& v : X) v = (float)(rand() % 5 - 2);
  for (auto& v : G) v = (float)(rand() % 9 - 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dG, G.data(), 128, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_x3<6>, dim3(1), dim3(64), 0, 0, dW, dX, dG, K, dO, dR);
  hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = G[i];
    for (int k = 0; k < K; ++k) s += (double)W[i * K + k] * X[j * K + k];
    if (s != out[i * 32 + j]) ++bad;
  }
  printf("layout check (A: row = lane&31, k = 8*(lane>>5)+e; C: row = 8*(r>>2)+4*(lane>>5)+(r&3), col = lane&31): %s (%d of 1024 cells differ)\n", bad ? "WRONG" : "ok", bad);
  // ---- (2) accuracy on log-likelihood-like data ----
  double e_chain = 0, e_6 = 0, e_9 = 0, r_chain = 0, r_6 = 0, r_9 = 0, split_resid = 0; long cnt = 0;
  for (int t = 0; t < TRIALS; ++t) {
    srand(100 + t);
    auto rnd = [] { return (rand() + 0.5) / (RAND_MAX + 1.0); };
    auto gauss = [&] { return std::sqrt(-2 * std::log(rnd())) * std::cos(6.283185307179586 * rnd()); };
    // Gaussian i: mean ~ 3 N(0,1), var ~ U[0.5, 2]; W = [mean/var | -0.5/var]; frame j: x ~ mean of a random Gaussian + noise; X = [x | fl(x*x)]
    std::vector<float> mean(32 * 40), var(32 * 40);
    for (int i = 0; i < 32 * 40; ++i) { mean[i] = (float)(3 * gauss()); var[i] = (float)(0.5 + 1.5 * rnd()); }
    for (int i = 0; i < 32; ++i) {
      double gc = -0.5 * 40 * 1.8378770664093453;
      for (int d = 0; d < 40; ++d) {
        float iv = 1.0f / var[i * 40 + d], miv = mean[i * 40 + d] * iv;
        W[i * K + 2 * d] = miv; W[i * K + 2 * d + 1] = -0.5f * iv;
        gc += -0.5 * std::log((double)var[i * 40 + d]) - 0.5 * (double)miv * miv / iv;
      }
      G[i] = (float)(gc + std::log(1.0 / 64));
    }
    for (int j = 0; j < 32; ++j) {
      int src = rand() % 32; bool far = (t & 1) && (j & 1);      // half the frames of odd trials: an unrelated point (large |ll|)
      for (int d = 0; d < 40; ++d) {
        float x = far ? (float)(3 * gauss()) : (float)(mean[src * 40 + d] + std::sqrt(var[src * 40 + d]) * gauss());
        X[j * K + 2 * d] = x; X[j * K + 2 * d + 1] = x * x;
      }
    }
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dG, G.data(), 128, hipMemcpyHostToDevice);
    std::vector<float> o6(1024), o9(1024);
    hipLaunchKernelGGL(k_x3<6>, dim3(1), dim3(64), 0, 0, dW, dX, dG, K, dO, dR);
    hipMemcpy(o6.data(), dO, 4096, hipMemcpyDeviceToHost); hipMemcpy(res.data(), dR, 256, hipMemcpyDeviceToHost);
    for (float v : res) split_resid = std::max(split_resid, (double)v);
    hipLaunchKernelGGL(k_x3<9>, dim3(1), dim3(64), 0, 0, dW, dX, dG, K, dO, dR);
    hipMemcpy(o9.data(), dO, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = G[i], B = std::fabs((double)G[i]);
      float c = G[i];
      for (int d = 0; d < 40; ++d) {   // the kernels' order: M x (d, d+1) then V x^2 (d, d+1) in pairs; order does not matter for the comparison
        s += (double)W[i * K + 2 * d] * X[j * K + 2 * d] + (double)W[i * K + 2 * d + 1] * X[j * K + 2 * d + 1];
        B += std::fabs((double)W[i * K + 2 * d] * X[j * K + 2 * d]) + std::fabs((double)W[i * K + 2 * d + 1] * X[j * K + 2 * d + 1]);
        c = fmaf(W[i * K + 2 * d], X[j * K + 2 * d], c); c = fmaf(W[i * K + 2 * d + 1], X[j * K + 2 * d + 1], c);
      }
      double a = std::fabs(c - s) / B, b = std::fabs(o6[i * 32 + j] - s) / B, d9 = std::fabs(o9[i * 32 + j] - s) / B;
      e_chain = std::max(e_chain, a); e_6 = std::max(e_6, b); e_9 = std::max(e_9, d9);
      r_chain += a * a; r_6 += b * b; r_9 += d9 * d9; ++cnt;
    }
  }
  printf("split residual max |w - (w1+w2+w3)| = %.3g (0 = exact)\n", split_resid);
  printf("error / B vs fp64 over %ld cells:   max        rms     (test tolerance: 1e-6 B + 1e-5)\n", cnt);
  printf("  fp32 fmaf chain (today)        %.3e  %.3e\n", e_chain, std::sqrt(r_chain / cnt));
  printf("  bf16x3, 6 products, bf16 MFMA  %.3e  %.3e\n", e_6, std::sqrt(r_6 / cnt));
  printf("  bf16x3, 9 products, bf16 MFMA  %.3e  %.3e\n", e_9, std::sqrt(r_9 / cnt));
  return 0;
}
