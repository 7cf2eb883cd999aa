// K1 loop lab (round 3): the chain loop of the f16x2 K1 in its REGULAR form -- every wave holds NT live 32-frame tiles, no
// first-live skipping, no shared remainder tiles, walk entries alternate "last tile of a pdf" -- in several structures, to
// find what the loop itself can reach before the walk's bookkeeping is put back.  Prints ns per MFMA and SIMD next to the
// bare dependent MFMA loop of the same run, and checks one workgroup's log-likelihoods against fp64.
//
//   WAVES   8 (two per SIMD, 256 registers) or 4 (one per SIMD, 512 registers)
//   NT      32-frame tiles per wave (B fragments resident in registers)
//   ONEACC  0: f16x2 as in khg_k1_f16x2.hip.inc (residual pieces pre-scaled by 2^11, main + cross accumulators, combined by
//              16 v_fma per chain); 1: "f16x2s" -- UNSCALED residual pieces, operands scaled so that the feature columns peak at
//              2^15 and the largest weight column at 2^15, all three partial products into ONE accumulator at scale 2^S, the
//              scale folded into the log-sum-exp's constants
//   G       W tiles per LDS slot / barrier interval (two slots)
//   LSE     0: no log-sum-exp (MFMA stream + LDS traffic only)
//   PREF    A fragments of tile t + 1 requested before the chains of tile t (inside a barrier interval)
//
// hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/k1lab.hip -o tools/bin/k1lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int KS = 5, D = 40, K = 80;
constexpr int TILEB = (2 * KS + 1) * 1024, PPT = 2 * KS + 1;
constexpr float MFLOOR = -1.0e30f;    // (the product uses -3e38 and never runs an update on an empty accumulator)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- synthetic data (same generator on host and device)
__host__ __device__ inline uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__host__ __device__ inline float urand(uint32_t a, uint32_t b) { return (hash32(a * 0x9e3779b9U + hash32(b)) >> 8) * (1.0f / 16777216.0f); }   // [0, 1)
// model: W[tile][g][k]: k even: mean * invvar, k odd: -0.5 invvar; gconst[tile][g]
__host__ __device__ inline float w_val(int tile, int g, int k) {
  const int d = k >> 1;
  const float var = 0.5f + 1.5f * urand(tile * 32 + g, 1000 + d);
  const float mean = 4.0f * urand(tile * 32 + g, 2000 + d) - 2.0f;
  return (k & 1) ? -0.5f / var : mean / var;
}
__host__ __device__ inline float g_val(int tile, int g) { return -60.0f - 20.0f * urand(tile * 32 + g, 77); }
__host__ __device__ inline float x_val(int64_t frame, int d) { return 6.0f * urand((uint32_t)frame, 3000 + d) - 3.0f; }

__host__ __device__ inline void split_scaled(float v, _Float16& a, _Float16& b) { a = (_Float16)v; b = (_Float16)((v - (float)a) * 2048.0f); }
__host__ __device__ inline void split_plain(float v, _Float16& a, _Float16& b) { a = (_Float16)v; b = (_Float16)(v - (float)a); }

// ew[k], ex[k]: operand exponents (w' = w 2^ew, x' = x 2^ex); gscale = 2^S with S = ew + ex (the same for every k)
template <bool ONEACC>
__global__ void pack_w(char* wimg, int ntiles, const int* ew, float gscale) {
  const int t = blockIdx.x;
  char* img = wimg + (size_t)t * TILEB;
  for (int f = threadIdx.x; f < KS * 64; f += blockDim.x) {
    const int s = f >> 6, lane = f & 63, g = lane & 31, kb = lane >> 5;
    f16x8 p0, p1;
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * kb + e;
      const float v = ldexpf(w_val(t, g, k), ew[k]);
      _Float16 a, b;
      if (ONEACC) split_plain(v, a, b); else split_scaled(v, a, b);
      p0[e] = a; p1[e] = b;
    }
    *reinterpret_cast<f16x8*>(img + ((0 * KS + s) * 64 + lane) * 16) = p0;
    *reinterpret_cast<f16x8*>(img + ((1 * KS + s) * 64 + lane) * 16) = p1;
  }
  float* gc = reinterpret_cast<float*>(img + 2 * KS * 1024);
  for (int i = threadIdx.x; i < 256; i += blockDim.x) gc[i] = i < 32 ? g_val(t, i) * gscale : 0.0f;
}
template <bool ONEACC>
__global__ void pack_x(u32x4* xh, int64_t nxt, const int* ex) {
  const int64_t nfrag = nxt * (KS * 64);
  for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nfrag; f += (int64_t)gridDim.x * blockDim.x) {
    const int64_t xt = f / (KS * 64);
    const int w = (int)(f - xt * (KS * 64)), s = w >> 6, lane = w & 63, col = lane & 31, kb = lane >> 5;
    f16x8 p0, p1;
    for (int e = 0; e < 8; e += 2) {
      const int k = 16 * s + 8 * kb + e, d = k >> 1;
      const float x = x_val(xt * 32 + col, d), xx = x * x;
      _Float16 a, b;
      if (ONEACC) split_plain(ldexpf(x, ex[k]), a, b); else split_scaled(ldexpf(x, ex[k]), a, b);
      p0[e] = a; p1[e] = b;
      if (ONEACC) split_plain(ldexpf(xx, ex[k + 1]), a, b); else split_scaled(ldexpf(xx, ex[k + 1]), a, b);
      p0[e + 1] = a; p1[e + 1] = b;
    }
    u32x4* dst = xh + xt * (2 * KS * 64);
    dst[(0 * KS + s) * 64 + lane] = __builtin_bit_cast(u32x4, p0);
    dst[(1 * KS + s) * 64 + lane] = __builtin_bit_cast(u32x4, p1);
  }
}

struct LabArgs {
  const u32x4* xh;       // [workgroup][tile NTW][piece 2][KS][lane]
  const char* wimg;
  const int32_t* walk;   // [workgroup][ntl]: tile id | last << 31
  int ntl;
  float* ll;             // [workgroup][pdf][NTW * 32]
  float c1;              // log2(e) 2^-S
  float inv_scale;       // 2^-S
  uint64_t* stamps;      // [workgroup][2]: s_memtime ticks, s_memrealtime ticks of wave 0 (or null)
};

__device__ __forceinline__ float max3v(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float exp2_le1(float a) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(a), 0.0f, 1.0f); }
__device__ __forceinline__ float pair_max(float x) {
  auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(__uint_as_float(s[0])), "v"(__uint_as_float(s[1])));
  return r;
}
__device__ __forceinline__ float pair_sum(float x) {
  auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

#define PIN(x) asm volatile("" : "+v"(x))
#ifndef VHEAD
#define VHEAD 4
#endif
#ifndef VPER
#define VPER 5
#endif

template <int WAVES, int NT, bool ONEACC, int G, bool LSE, bool PREF, int STAG, int DBG = 0>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void lab(LabArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int SLOTB = G * TILEB;
  constexpr int NTW = WAVES * NT;
  constexpr float LN2 = 0.69314718055994530942f;
  static_assert((G * NT) % 2 == 0, "accumulator parity must repeat per barrier interval");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, col = lane & 31, kb = lane >> 5;
  const float c1 = a.c1;
  const uint64_t t0c = __builtin_readcyclecounter(), t0r = __builtin_amdgcn_s_memrealtime();

  f16x8 xb[NT][2][KS];
#pragma unroll
  for (int f = 0; f < NT; ++f) {
    const u32x4* src = a.xh + ((int64_t)blockIdx.x * NTW + wave + WAVES * f) * (2 * KS * 64) + lane;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s) xb[f][p][s] = __builtin_bit_cast(f16x8, src[(p * KS + s) * 64]);
  }
#pragma unroll
  for (int f = 0; f < NT; ++f)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s) asm volatile("" ::"v"(xb[f][p][s]));
  float* llw = a.ll + (int64_t)blockIdx.x * (a.ntl / 2) * (NTW * 32);

  const uint32_t lane_b = (uint32_t)lane * 16u;
  auto dma_tile = [&](int tile, int slot, int sub) {
    const char* src = a.wimg + (int64_t)tile * TILEB;
    const int w0 = (wave + 3 * sub) % WAVES;
#pragma unroll
    for (int i = 0; i < (PPT + WAVES - 1) / WAVES; ++i) {
      const int piece = w0 + WAVES * i;
      if (piece < PPT)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024 + lane_b),
                                         (__attribute__((address_space(3))) void*)(lds + slot * SLOTB + sub * TILEB + piece * 1024), 16, 0, 0);
    }
  };
  const int ntl = a.ntl;
  const int32_t* tl = a.walk + (int64_t)blockIdx.x * ntl;
  auto sload = [&](int i) {
    const int32_t* ptr = tl + (i < ntl ? i : ntl - 1);
    int v;
    asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(ptr));
    return v;
  };
  auto sload_wait = [&](int& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)::"memory"); };
  const int nit = (ntl + G - 1) / G;
  int eg[G], en[G];
  auto sload_group = [&](int g, int (&dst)[G]) {
#pragma unroll
    for (int t = 0; t < G; ++t) dst[t] = sload(G * g + t);
  };
  auto wait_group = [&](int (&v)[G]) {
#pragma unroll
    for (int t = 0; t < G; ++t) sload_wait(v[t]);
  };
  auto dma_group = [&](const int (&v)[G], int slot, int first) {
#pragma unroll
    for (int t = 0; t < G; ++t)
      if (first + t < ntl) dma_tile(v[t] & 0x3fffff, slot, t);
  };
  sload_group(0, eg);
  sload_group(1, en);
  wait_group(eg);
  wait_group(en);
  dma_group(eg, 0, 0);

  float m_run[NT], s_run[NT];
#pragma unroll
  for (int f = 0; f < NT; ++f) { m_run[f] = MFLOOR; s_run[f] = 0.0f; }
  // accumulators: two sets alternate; a set filled with -inf makes its log-sum-exp update a no-op (the pipeline's first step)
  f32x16 accm[2], accc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { accm[0][i] = accm[1][i] = -INFINITY; accc[0][i] = accc[1][i] = 0.0f; }
  int pend_last = 0, pend_j = 0;

  auto lse_update = [&](auto r_c, auto f_c) {
    constexpr int R = decltype(r_c)::value, F = decltype(f_c)::value;
    f32x16 c;
    if constexpr (ONEACC) c = accm[R];
    else {
#pragma unroll
      for (int i = 0; i < 16; ++i) c[i] = __builtin_fmaf(accc[R][i], 1.0f / 2048.0f, accm[R][i]);
    }
    float m = max3v(max3v(c[0], c[1], c[2]), max3v(c[3], c[4], c[5]), max3v(c[6], c[7], c[8]));
    const float m2 = max3v(max3v(c[9], c[10], c[11]), max3v(c[12], c[13], c[14]), c[15]);
    m = pair_max(max3v(m, m2, MFLOOR));
    const float mnew = __builtin_fmaxf(m_run[F], m);
    const float nm = -mnew * c1;
    float ss0 = 0.0f, ss1 = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      ss0 += exp2_le1(__builtin_fmaf(c[i], c1, nm));
      ss1 += exp2_le1(__builtin_fmaf(c[i + 1], c1, nm));
    }
    const float sc = __builtin_amdgcn_exp2f((m_run[F] - mnew) * c1);
    s_run[F] = __builtin_fmaf(s_run[F], sc, ss0 + ss1);
    m_run[F] = mnew;
  };

  // ---- the log-sum-exp update as an indexed list of single instructions (hand-placed interleave, DBG & 8) ----
  struct LseTmp { float t[5], u0, u1, m, mnew, nm, sc, a[16], e[16], ss0, ss1; };
  constexpr int LSE_NOPS = 63;
  auto lse_op = [&](auto i_c, auto r_c, auto f_c, LseTmp& q) {
    constexpr int I = decltype(i_c)::value, R = decltype(r_c)::value, F = decltype(f_c)::value;
    const f32x16& c = accm[R];
    if constexpr (I < 5) { q.t[I] = max3v(c[3 * I], c[3 * I + 1], c[3 * I + 2]); }
    else if constexpr (I == 5) { q.u0 = max3v(q.t[0], q.t[1], q.t[2]); }
    else if constexpr (I == 6) { q.u1 = max3v(q.t[3], q.t[4], c[15]); }
    else if constexpr (I == 7) { q.m = max3v(q.u0, q.u1, MFLOOR); }
    else if constexpr (I == 8) { q.m = pair_max(q.m); }
    else if constexpr (I == 9) { q.mnew = __builtin_fmaxf(m_run[F], q.m); }
    else if constexpr (I == 10) { q.nm = -q.mnew * c1; }
    else if constexpr (I == 11) { q.sc = __builtin_fmaf(m_run[F], c1, q.nm); }
    else if constexpr (I == 12) { q.sc = __builtin_amdgcn_exp2f(q.sc); q.ss0 = 0.0f; q.ss1 = 0.0f; }
    else if constexpr (I < 13 + 48) {
      // software pipeline over the 16 values: step j issues A_j (fma), E_{j-1} (exp), S_{j-2} (add); empty slots skipped
      constexpr int K0 = I - 13;                 // 0 .. 47 in issue order
      constexpr auto pick = [] {
        int n = 0;
        for (int j = 0; j < 18; ++j)
          for (int sl = 0; sl < 3; ++sl) {
            const int idx = j - sl;
            if (idx < 0 || idx > 15) continue;
            if (n == K0) return sl * 16 + idx;
            ++n;
          }
        return -1;
      }();
      constexpr int SL = pick / 16, IDX = pick % 16;
      if constexpr (SL == 0) { q.a[IDX] = __builtin_fmaf(c[IDX], c1, q.nm); }
      else if constexpr (SL == 1) { q.e[IDX] = exp2_le1(q.a[IDX]); }
      else if constexpr (IDX < 2) { if constexpr (IDX & 1) q.ss1 = q.e[IDX]; else q.ss0 = q.e[IDX]; }
      else { if constexpr (IDX & 1) { q.ss1 += q.e[IDX]; PIN(q.ss1); } else { q.ss0 += q.e[IDX]; PIN(q.ss0); } }
    }
    else if constexpr (I == 61) { q.ss0 += q.ss1; PIN(q.ss0); }
    else if constexpr (I == 62) { s_run[F] = __builtin_fmaf(s_run[F], q.sc, q.ss0); m_run[F] = q.mnew; }
  };
  auto lse_ops = [&](auto lo_c, auto hi_c, auto r_c, auto f_c, LseTmp& q) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (lse_op(std::integral_constant<int, LO + I>(), r_c, f_c, q), ...);
    }(std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>());
  };
  auto finish = [&](auto f_c, int jrow) {
    constexpr int F = decltype(f_c)::value;
    const float st = pair_sum(s_run[F]);
    const float v = __builtin_fmaf(m_run[F], a.inv_scale, __builtin_amdgcn_logf(st) * LN2);
    if (kb == 0) llw[(int64_t)jrow * (NTW * 32) + 32 * (wave + WAVES * F) + col] = v;
    m_run[F] = MFLOOR;
    s_run[F] = 0.0f;
  };
  struct Frags { f16x8 A[2][KS]; f32x16 gc; };
  auto load_frags = [&](const char* wt, Frags& fr) {
    const float* gcp = reinterpret_cast<const float*>(wt + 2 * KS * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gcp + 8 * i + 4 * kb);
      fr.gc[4 * i] = g[0]; fr.gc[4 * i + 1] = g[1]; fr.gc[4 * i + 2] = g[2]; fr.gc[4 * i + 3] = g[3];
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s) fr.A[p][s] = *reinterpret_cast<const f16x8*>(wt + ((p * KS + s) * 64 + lane) * 16);
  };
  // chain for frame tile FW into accumulator set W, with the log-sum-exp of set 1 - W (frame tile FR) inside it
  auto chain = [&](Frags& fr, auto w_c, auto fw_c, auto fr_c, auto roll_c, const char* nxt) {
    constexpr int W = decltype(w_c)::value, FW = decltype(fw_c)::value;
    constexpr bool ROLL = decltype(roll_c)::value;
    f32x16 cm, cc;
    if constexpr (!(DBG & 16)) cm = fr.gc;
#pragma unroll
    for (int i = 0; i < 16; ++i) cc[i] = 0.0f;
    if constexpr (LSE && ONEACC && (DBG & 8)) {
      // hand-placed: ops [lo, hi) of the previous chain's update behind MFMA i; gap 0 left empty (the previous chain's last MFMA
      // is still in the pipe), the 63 instructions spread evenly over gaps 1 .. 14
      LseTmp q;
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
          constexpr int s = I / 3, jj = I % 3;
          if constexpr (DBG & 16) {        // MFMA by inline asm: B fragments in AGPRs, accumulators and A in VGPRs
            if constexpr (I == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(cm) : "v"(fr.A[0][0]), "a"(xb[FW][1][0]), "v"(fr.gc));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cm) : "v"(fr.A[jj == 1 ? 1 : 0][s]), "a"(xb[FW][jj == 0 ? 1 : 0][s]));
          } else
          cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[jj == 1 ? 1 : 0][s], xb[FW][jj == 0 ? 1 : 0][s], cm, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int lo = I == 0 ? 0 : (I - 1) * LSE_NOPS / 14, hi = I == 0 ? 0 : I * LSE_NOPS / 14;
          lse_ops(std::integral_constant<int, lo>(), std::integral_constant<int, hi>(), std::integral_constant<int, 1 - W>(), fr_c, q);
          if constexpr (ROLL) {            // rolling prefetch: the next W tile's fragments into the registers this chain has finished with
            if constexpr (I == 0) {
              const float* gcp = reinterpret_cast<const float*>(nxt + 2 * KS * 1024);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gcp + 8 * i + 4 * kb);
                fr.gc[4 * i] = g[0]; fr.gc[4 * i + 1] = g[1]; fr.gc[4 * i + 2] = g[2]; fr.gc[4 * i + 3] = g[3];
              }
            }
            if constexpr (jj == 2) {
              fr.A[0][s] = *reinterpret_cast<const f16x8*>(nxt + ((0 * KS + s) * 64 + lane) * 16);
              fr.A[1][s] = *reinterpret_cast<const f16x8*>(nxt + ((1 * KS + s) * 64 + lane) * 16);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }(), ...);
      }(std::make_integer_sequence<int, 3 * KS>());
      accm[W] = cm;
      return;
    }
    if constexpr (LSE) lse_update(std::integral_constant<int, 1 - W>(), fr_c);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (ONEACC) {
        cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[0][s], xb[FW][1][s], cm, 0, 0, 0);
        cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[1][s], xb[FW][0][s], cm, 0, 0, 0);
        cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[0][s], xb[FW][0][s], cm, 0, 0, 0);
      } else {
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[0][s], xb[FW][1][s], cc, 0, 0, 0);
        cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[0][s], xb[FW][0][s], cm, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr.A[1][s], xb[FW][0][s], cc, 0, 0, 0);
      }
    }
    if constexpr (LSE) {
      __builtin_amdgcn_sched_group_barrier(0x002, VHEAD, 0);
#pragma unroll
      for (int i = 0; i < 3 * KS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
      }
    }
    if constexpr (!LSE) {            // keep every chain alive without a log-sum-exp
      asm volatile("" :: "v"(cm));
      if constexpr (!ONEACC) asm volatile("" :: "v"(cc));
    }
    accm[W] = cm;
    if constexpr (!ONEACC) accc[W] = cc;
  };

  int j = 0;
  if constexpr (DBG & 32) {
    // ROLLING form: the barrier of an interval sits in front of its LAST tile (whose fragments are already in registers), the
    // DMA of the group after next goes out right behind it, and the last chain of every tile fetches the next tile's fragments
    int e2[G];
    sload_group(2, e2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nit > 1) dma_group(en, 1, G);
    Frags fa;
    load_frags(lds, fa);
    for (int it = 0; it < nit; ++it) {
      const char* slot_base = lds + (it & 1) * SLOTB;
      const char* other_base = lds + ((it + 1) & 1) * SLOTB;
#pragma unroll
      for (int t = 0; t < G; ++t) {
        const int e = eg[t];
        const int last = e < 0 ? 1 : 0;
        if (t == G - 1) {
          wait_group(e2);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (it + 2 < nit) dma_group(e2, it & 1, G * (it + 2));
        }
        const char* nxt = t + 1 < G ? slot_base + (t + 1) * TILEB : other_base;
        [&]<int... F>(std::integer_sequence<int, F...>) {
          ([&] {
            const int par = (t * NT + F) & 1;
            auto run = [&](auto w_c) {
              chain(fa, w_c, std::integral_constant<int, F>(), std::integral_constant<int, (F + NT - 1) % NT>(), std::integral_constant<bool, F == NT - 1>(), nxt);
            };
            if (par == 0) run(std::integral_constant<int, 0>()); else run(std::integral_constant<int, 1>());
            if constexpr (F == 0) { if (pend_last) finish(std::integral_constant<int, NT - 1>(), pend_j); }
            else { if (last) finish(std::integral_constant<int, F - 1>(), j); }
          }(), ...);
        }(std::make_integer_sequence<int, NT>());
        pend_last = last; pend_j = j;
        j += last;
      }
#pragma unroll
      for (int t = 0; t < G; ++t) { eg[t] = en[t]; en[t] = e2[t]; }
      sload_group(it + 3, e2);
    }
  } else
  for (int it = 0; it < nit; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (!(DBG & 1)) __builtin_amdgcn_s_barrier();
    if constexpr (!(DBG & 2)) { if (it + 1 < nit) dma_group(en, (it + 1) & 1, G * (it + 1)); }
    // optional stagger: the second wave of every SIMD leaves the barrier late
    if constexpr (STAG > 0) {
      if (wave >= WAVES / 2) __builtin_amdgcn_s_sleep(STAG);
    }
    int ec[G];
#pragma unroll
    for (int t = 0; t < G; ++t) { ec[t] = eg[t]; eg[t] = en[t]; }
    sload_group(it + 2, en);
    const char* slot_base = lds + (it & 1) * SLOTB;
    Frags fa, fb;
    if constexpr (PREF) load_frags(slot_base, fa);
#pragma unroll
    for (int t = 0; t < G; ++t) {
      {
        const int e = ec[t];
        const int last = e < 0 ? 1 : 0;
        Frags* cur;
        if constexpr (PREF) {
          cur = (t & 1) ? &fb : &fa;
          if (t + 1 < G) load_frags(slot_base + (t + 1) * TILEB, (t & 1) ? fa : fb);
        } else {
          cur = &fa;
          if (!(DBG & 4) || (it == 0 && t == 0)) load_frags(slot_base + t * TILEB, fa);
        }
        [&]<int... F>(std::integer_sequence<int, F...>) {
          ([&] {
            constexpr int c = F;                 // chain index inside the tile; G * NT is even, so the parity of (t NT + F) is static
            const int par = (t * NT + c) & 1;
            auto run = [&](auto w_c) {
              chain(*cur, w_c, std::integral_constant<int, F>(), std::integral_constant<int, (F + NT - 1) % NT>(), std::false_type(), nullptr);
            };
            if (par == 0) run(std::integral_constant<int, 0>()); else run(std::integral_constant<int, 1>());
            // the chain just closed the log-sum-exp of the previous chain: tile (F - 1) of this W tile, or tile NT - 1 of the previous one
            if constexpr (LSE) {
              if constexpr (F == 0) { if (pend_last) finish(std::integral_constant<int, NT - 1>(), pend_j); }
              else { if (last) finish(std::integral_constant<int, F - 1>(), j); }
            }
          }(), ...);
        }(std::make_integer_sequence<int, NT>());
        pend_last = last; pend_j = j;
        j += last;
      }
    }
    wait_group(en);
  }
  if (a.stamps && threadIdx.x == 0) {
    a.stamps[2 * blockIdx.x] = __builtin_readcyclecounter() - t0c;
    a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  // drain: the last chain's log-sum-exp
  if constexpr (LSE) {
    const int par = (ntl * NT - 1) & 1;
    if (par == 0) lse_update(std::integral_constant<int, 0>(), std::integral_constant<int, NT - 1>());
    else lse_update(std::integral_constant<int, 1>(), std::integral_constant<int, NT - 1>());
    if (pend_last) finish(std::integral_constant<int, NT - 1>(), pend_j);
  } else {
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += accm[0][i] + accm[1][i] + (ONEACC ? 0.0f : accc[0][i] + accc[1][i]);
    llw[threadIdx.x] = s;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// TRANSPOSED decomposition ("labx"): the utterance's feature tiles (n x 10 KiB of B fragments) sit in LDS for the whole
// workgroup, every WAVE owns whole pdfs: it loads the pdf's two W tiles from global memory straight into registers (no LDS
// staging, no workgroup barrier in the walk, no dealing of frame tiles), then runs over the frame tiles f = 0 .. n-1 with two
// chains per f; the B fragments of tile f + 1 are fetched from LDS into the registers the second chain has finished with.
// The log-sum-exp of a chain runs inside the next chain (hand-placed); a pdf's value for tile f is complete inside chain
// (f + 1, 0), which also stores it -- no data-dependent branch in the loop.
template <bool LSE, int DBG>
__global__ __launch_bounds__(512, 2) void labx(LabArgs a, int n) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr float LN2 = 0.69314718055994530942f;
  constexpr int XTB = 2 * KS * 1024;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, col = lane & 31, kb = lane >> 5;
  const float c1 = a.c1;
  const uint64_t t0c = __builtin_readcyclecounter(), t0r = __builtin_amdgcn_s_memrealtime();
  {
    const char* src = reinterpret_cast<const char*>(a.xh) + (int64_t)blockIdx.x * n * XTB;
    for (int i = wave; i < n * 2 * KS; i += 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(lds + i * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const int npdf = a.ntl / 2;
  const int32_t* tl = a.walk + (int64_t)blockIdx.x * a.ntl;
  float* llw = a.ll + (int64_t)blockIdx.x * npdf * (n * 32);
  float* dump = a.ll + (int64_t)gridDim.x * npdf * (n * 32) + 64;     // a scratch line behind the output
  float m_run = MFLOOR, s_run = 0.0f;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = -INFINITY; acc1[i] = -INFINITY; }
  float* pend = dump;
  f16x8 B[2][KS];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int s = 0; s < KS; ++s) B[p][s] = *reinterpret_cast<const f16x8*>(lds + ((p * KS + s) * 64 + lane) * 16);

  struct Q { float t[5], u0, u1, m, mnew, nm, sc, a[16], e[16], ss0, ss1, st, v; };
  // op I of: the log-sum-exp update of accumulator c (63 ops), then -- FIN -- the finish of the pending (pdf, tile) (6 ops)
  auto op = [&](auto i_c, const f32x16& c, Q& q) {
    constexpr int I = decltype(i_c)::value;
    if constexpr (I < 5) q.t[I] = max3v(c[3 * I], c[3 * I + 1], c[3 * I + 2]);
    else if constexpr (I == 5) q.u0 = max3v(q.t[0], q.t[1], q.t[2]);
    else if constexpr (I == 6) q.u1 = max3v(q.t[3], q.t[4], c[15]);
    else if constexpr (I == 7) q.m = max3v(q.u0, q.u1, MFLOOR);
    else if constexpr (I == 8) q.m = pair_max(q.m);
    else if constexpr (I == 9) q.mnew = __builtin_fmaxf(m_run, q.m);
    else if constexpr (I == 10) q.nm = -q.mnew * c1;
    else if constexpr (I == 11) q.sc = __builtin_fmaf(m_run, c1, q.nm);
    else if constexpr (I == 12) q.sc = __builtin_amdgcn_exp2f(q.sc);
    else if constexpr (I < 13 + 48) {
      constexpr int K0 = I - 13;
      constexpr auto pick = [] {
        int nn = 0;
        for (int j = 0; j < 18; ++j)
          for (int sl = 0; sl < 3; ++sl) {
            const int idx = j - sl;
            if (idx < 0 || idx > 15) continue;
            if (nn == K0) return sl * 16 + idx;
            ++nn;
          }
        return -1;
      }();
      constexpr int SL = pick / 16, IDX = pick % 16;
      if constexpr (SL == 0) q.a[IDX] = __builtin_fmaf(c[IDX], c1, q.nm);
      else if constexpr (SL == 1) q.e[IDX] = exp2_le1(q.a[IDX]);
      else if constexpr (IDX < 2) { if constexpr (IDX & 1) q.ss1 = q.e[IDX]; else q.ss0 = q.e[IDX]; }
      else { if constexpr (IDX & 1) { q.ss1 += q.e[IDX]; PIN(q.ss1); } else { q.ss0 += q.e[IDX]; PIN(q.ss0); } }
    }
    else if constexpr (I == 61) { q.ss0 += q.ss1; PIN(q.ss0); }
    else if constexpr (I == 62) { s_run = __builtin_fmaf(s_run, q.sc, q.ss0); m_run = q.mnew; }
    else if constexpr (I == 63) q.st = pair_sum(s_run);
    else if constexpr (I == 64) q.st = __builtin_amdgcn_logf(q.st);
    else if constexpr (I == 65) q.v = q.st * LN2;
    else if constexpr (I == 66) q.v = __builtin_fmaf(m_run, a.inv_scale, q.v);
    else if constexpr (I == 67) { if (kb == 0) pend[col] = q.v; }
    else if constexpr (I == 68) { m_run = MFLOOR; s_run = 0.0f; }
  };
  auto ops = [&](auto lo_c, auto hi_c, const f32x16& c, Q& q) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    [&]<int... I>(std::integer_sequence<int, I...>) { (op(std::integral_constant<int, LO + I>(), c, q), ...); }
    (std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>());
  };
  // one chain: acc = gc + A . B (15 MFMAs), NOPS instructions of the list above on accumulator `prev` spread over gaps 1 .. 14;
  // ROLL: B[.][s] <- tile at nxt once step s is done
  auto chain = [&](f32x16& acc, f32x16& gc, f16x8 (&A)[2][KS], const f32x16& prev, auto nops_c, auto roll_c, const char* nxt, auto wroll_c, const char* wn) {
    constexpr int NOPS = decltype(nops_c)::value;
    constexpr bool ROLL = decltype(roll_c)::value, WROLL = decltype(wroll_c)::value;
    Q q;
    f32x16 cm = gc;
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        constexpr int s = I / 3, jj = I % 3;
        cm = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[jj == 1 ? 1 : 0][s], B[jj == 0 ? 1 : 0][s], cm, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LSE) {
          constexpr int lo = I == 0 ? 0 : (I - 1) * NOPS / 14, hi = I == 0 ? 0 : I * NOPS / 14;
          ops(std::integral_constant<int, lo>(), std::integral_constant<int, hi>(), prev, q);
        }
        if constexpr (ROLL && jj == 2) {
          B[0][s] = *reinterpret_cast<const f16x8*>(nxt + ((0 * KS + s) * 64 + lane) * 16);
          B[1][s] = *reinterpret_cast<const f16x8*>(nxt + ((1 * KS + s) * 64 + lane) * 16);
        }
        if constexpr (WROLL) {            // the NEXT pdf's W tile into the registers this (last) pass over it has finished with
          if constexpr (I == 0) {
            const float* g = reinterpret_cast<const float*>(wn + 2 * KS * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const f32x4 x0 = *reinterpret_cast<const f32x4*>(g + 8 * i + 4 * kb);
#pragma unroll
              for (int jx = 0; jx < 4; ++jx) gc[4 * i + jx] = x0[jx];
            }
          }
          if constexpr (jj == 2) {
            A[0][s] = *reinterpret_cast<const f16x8*>(wn + ((0 * KS + s) * 64 + lane) * 16);
            A[1][s] = *reinterpret_cast<const f16x8*>(wn + ((1 * KS + s) * 64 + lane) * 16);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }(), ...);
    }(std::make_integer_sequence<int, 3 * KS>());
    if constexpr (!LSE) asm volatile("" :: "v"(cm));
    acc = cm;
  };

  f16x8 A0[2][KS], A1[2][KS];
  f32x16 gc0, gc1;
  auto load_w = [&](const char* w, f16x8 (&A)[2][KS], f32x16& gc) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int s = 0; s < KS; ++s) A[p][s] = *reinterpret_cast<const f16x8*>(w + ((p * KS + s) * 64 + lane) * 16);
    const float* g = reinterpret_cast<const float*>(w + 2 * KS * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(g + 8 * i + 4 * kb);
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) gc[4 * i + jx] = x0[jx];
    }
  };
  auto tile_ptr = [&](int idx) { return a.wimg + (int64_t)(__builtin_amdgcn_readfirstlane(tl[idx < a.ntl ? idx : a.ntl - 1]) & 0x3fffff) * TILEB; };
  if constexpr (DBG & 1) { if (wave < npdf) { load_w(tile_ptr(2 * wave), A0, gc0); load_w(tile_ptr(2 * wave + 1), A1, gc1); } }
  for (int u = wave; u < npdf; u += 8) {
    if constexpr (!(DBG & 1)) { load_w(tile_ptr(2 * u), A0, gc0); load_w(tile_ptr(2 * u + 1), A1, gc1); }
    const char* wn0 = tile_ptr(2 * (u + 8));
    const char* wn1 = tile_ptr(2 * (u + 8) + 1);
    float tch[4];
    if constexpr (DBG & 2) {       // pull the next pdf's tiles from beyond L2 into L2: one dword per 128-byte line
      tch[0] = *reinterpret_cast<const float*>(wn0 + lane * 128);
      tch[1] = *reinterpret_cast<const float*>(wn0 + 8192 + (lane & 31) * 128 / 2);
      tch[2] = *reinterpret_cast<const float*>(wn1 + lane * 128);
      tch[3] = *reinterpret_cast<const float*>(wn1 + 8192 + (lane & 31) * 128 / 2);
    }
    float* llrow = llw + (int64_t)u * (n * 32) + col;
    for (int f = 0; f < n - ((DBG & 1) ? 1 : 0); ++f) {
      const char* nxt = lds + (f + 1 < n ? f + 1 : 0) * XTB;
      chain(acc0, gc0, A0, acc1, std::integral_constant<int, 69>(), std::false_type(), nullptr, std::false_type(), nullptr);
      pend = llrow + 32 * f - col;
      chain(acc1, gc1, A1, acc0, std::integral_constant<int, 63>(), std::true_type(), nxt, std::false_type(), nullptr);
    }
    if constexpr (DBG & 1) {       // last frame tile: the next pdf's W tiles roll in
      chain(acc0, gc0, A0, acc1, std::integral_constant<int, 69>(), std::false_type(), nullptr, std::true_type(), wn0);
      pend = llrow + 32 * (n - 1) - col;
      chain(acc1, gc1, A1, acc0, std::integral_constant<int, 63>(), std::true_type(), lds, std::true_type(), wn1);
    }
    if constexpr (DBG & 2) asm volatile("" :: "v"(tch[0]), "v"(tch[1]), "v"(tch[2]), "v"(tch[3]));
  }
  if (a.stamps && threadIdx.x == 0) {
    a.stamps[2 * blockIdx.x] = __builtin_readcyclecounter() - t0c;
    a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
  }
  if constexpr (LSE) {     // drain: the last chain's update and the last finish
    Q q;
    ops(std::integral_constant<int, 0>(), std::integral_constant<int, 69>(), acc1, q);
  } else {
    float sx = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) sx += acc0[i] + acc1[i];
    dump[threadIdx.x] = sx;
  }
}

// bare dependent MFMA loop (one wave per SIMD), the clock-limited floor of this box
__global__ __launch_bounds__(256, 1) void bare(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[2];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.5f + 0.01f * lane + i); b[i] = (_Float16)(0.25f * lane - i); }
  for (int j = 0; j < 16; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 1], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

struct Data {
  char* wimg[2];        // [ONEACC]
  u32x4* xh[2];
  int32_t* walk;
  float* ll;
  int *ew[2], *ex[2];
  float c1[2], inv[2], gscale[2];
  std::vector<int32_t> walk_h;
  int nwg, ntl, ntiles_w;
  uint64_t* stamps;
  int64_t nxt;
};


static void check_ll(Data& d, int nwg, int NTW) {
  // fp64 check of workgroup 0 and the last one
  std::vector<float> got((size_t)(d.ntl / 2) * NTW * 32);
  double worst = 0;
  for (int wg : {0, nwg - 1}) {
    CK(hipMemcpy(got.data(), d.ll + (size_t)wg * got.size(), got.size() * 4, hipMemcpyDeviceToHost));
    for (int j = 0; j < d.ntl / 2; j += 7)
      for (int t = 0; t < NTW * 32; t += 5) {
        const int64_t frame = ((int64_t)wg * NTW + t / 32) * 32 + (t & 31);
        double x[D];
        for (int dd = 0; dd < D; ++dd) x[dd] = x_val(frame, dd);
        double c[64], mx = -1e300, B = 0;
        for (int h = 0; h < 2; ++h) {
          const int tile = d.walk_h[(size_t)wg * d.ntl + 2 * j + h] & 0x3fffff;
          for (int g = 0; g < 32; ++g) {
            double s = g_val(tile, g), b = std::fabs(s);
            for (int dd = 0; dd < D; ++dd) {
              const double xx = (double)(float)(x[dd] * x[dd]);
              s += (double)w_val(tile, g, 2 * dd) * x[dd] + (double)w_val(tile, g, 2 * dd + 1) * xx;
              b += std::fabs((double)w_val(tile, g, 2 * dd) * x[dd]) + std::fabs((double)w_val(tile, g, 2 * dd + 1) * xx);
            }
            c[32 * h + g] = s; mx = s > mx ? s : mx; B = b > B ? b : B;
          }
        }
        double se = 0;
        for (int g = 0; g < 64; ++g) se += std::exp(c[g] - mx);
        const double want = mx + std::log(se);
        const double err = std::fabs(want - got[(size_t)j * NTW * 32 + t]) / B;
        worst = err > worst ? err : worst;
      }
  }
  printf("   max |err| / B vs fp64 %.2e", worst);
}

template <bool LSE, int DBG>
double runx(Data& d, int n, const char* name, bool check) {
  const int nwg = (int)(d.nxt / n);
  LabArgs a{d.xh[1], d.wimg[1], d.walk, d.ntl, d.ll, d.c1[1], d.inv[1], d.stamps};
  const size_t ldsb = (size_t)n * 2 * KS * 1024;
  auto kern = labx<LSE, DBG>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)kern));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemset(d.ll, 0, (size_t)nwg * (d.ntl / 2) * n * 32 * 4));
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), ldsb, 0, a, n);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), ldsb, 0, a, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double mfma = (double)nwg * n * d.ntl * 3 * KS;
  const double ns = best * 1e6 / (mfma / 1024.0);
  std::vector<uint64_t> st(2 * (size_t)nwg);
  CK(hipMemcpy(st.data(), d.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < nwg; ++i) { cyc += (double)st[2 * i]; rt += (double)st[2 * i + 1]; }
  const double ghz = cyc / rt * 0.1, cpm = cyc / nwg / ((double)n * d.ntl * 3 * KS / 4.0);
  printf("%-34s regs %3d  %7.3f ms  %6.2f ns/MFMA/SIMD  clock %.2f GHz  %5.1f cyc/MFMA/SIMD (pipe %.2f)", name, fa.numRegs, best, ns, ghz, cpm, 32.0 / cpm);
  if (check && LSE) check_ll(d, nwg, n);
  printf("\n");
  fflush(stdout);
  return ns;
}

template <int WAVES, int NT, bool ONEACC, int G, bool LSE, bool PREF, int STAG, int DBG = 0>
double run(Data& d, const char* name, bool check) {
  constexpr int NTW = WAVES * NT;
  const int nwg = (int)(d.nxt / NTW);
  LabArgs a{d.xh[ONEACC], d.wimg[ONEACC], d.walk, d.ntl, d.ll, d.c1[ONEACC], d.inv[ONEACC], d.stamps};
  const size_t ldsb = 2 * G * TILEB;
  auto kern = lab<WAVES, NT, ONEACC, G, LSE, PREF, STAG, DBG>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)kern));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipMemset(d.ll, 0, (size_t)nwg * (d.ntl / 2) * NTW * 32 * 4));
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(WAVES * 64), ldsb, 0, a);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(WAVES * 64), ldsb, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double mfma = (double)nwg * NTW * d.ntl * 3 * KS;
  const double ns = best * 1e6 / (mfma / 1024.0);
  std::vector<uint64_t> st(2 * (size_t)nwg);
  CK(hipMemcpy(st.data(), d.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < nwg; ++i) { cyc += (double)st[2 * i]; rt += (double)st[2 * i + 1]; }
  const double ghz = cyc / rt * 0.1, cpm = cyc / nwg / ((double)NTW * d.ntl * 3 * KS / 4.0);
  printf("%-34s regs %3d  %7.3f ms  %6.2f ns/MFMA/SIMD  clock %.2f GHz  %5.1f cyc/MFMA/SIMD (pipe %.2f)", name, fa.numRegs, best, ns, ghz, cpm, 32.0 / cpm);
  if (check && LSE) check_ll(d, nwg, NTW);
  printf("\n");
  fflush(stdout);
  return ns;
}

int main(int argc, char** argv) {
  Data d;
  d.ntiles_w = 10000; d.ntl = 192;
  d.nxt = 4096 * 16;          // 32-frame tiles: 4096 workgroups of 16 tiles (or 2731 of 24, ...)
  // exponents.  f16x2: balanced, both operands peak near sqrt(max |w| max |x|).  f16x2s: feature columns peak in [2^14, 2^15),
  // the largest weight column in [2^14, 2^15), S = ew + ex the same for all k.
  std::vector<int> ew[2], ex[2];
  double pmax = 0;
  std::vector<double> wmax(K, 0), xmax(K, 0);
  for (int k = 0; k < K; ++k) {
    for (int t = 0; t < 200; ++t) for (int g = 0; g < 32; ++g) wmax[k] = std::fmax(wmax[k], std::fabs(w_val(t, g, k)));
    wmax[k] *= 1.05;            // the sample's maximum, with a margin
    xmax[k] = (k & 1) ? 9.0 : 3.0;
    pmax = std::fmax(pmax, wmax[k] * xmax[k]);
  }
  ew[0].resize(K); ex[0].resize(K); ew[1].resize(K); ex[1].resize(K);
  int S = 1000;
  for (int k = 0; k < K; ++k) {
    const int e = (int)std::lrint(0.5 * std::log2(wmax[k] / xmax[k]));
    ew[0][k] = -e; ex[0][k] = e;
    ex[1][k] = 14 - (int)std::floor(std::log2(xmax[k]));               // x' in [2^14, 2^15)
    const int cand = 14 - (int)std::floor(std::log2(wmax[k])) + ex[1][k];
    S = cand < S ? cand : S;
  }
  for (int k = 0; k < K; ++k) ew[1][k] = S - ex[1][k];
  printf("f16x2s: S = %d (max_k max|w| max|x| = %.1f)\n", S, pmax);
  const float L2E = 1.44269504088896340736f;
  d.c1[0] = L2E; d.inv[0] = 1.0f; d.gscale[0] = 1.0f;
  d.c1[1] = std::ldexp(L2E, -S); d.inv[1] = std::ldexp(1.0f, -S); d.gscale[1] = std::ldexp(1.0f, S);
  for (int o = 0; o < 2; ++o) {
    CK(hipMalloc(&d.wimg[o], (size_t)d.ntiles_w * TILEB));
    CK(hipMalloc(&d.xh[o], (size_t)d.nxt * 2 * KS * 1024));
    CK(hipMalloc(&d.ew[o], K * 4)); CK(hipMalloc(&d.ex[o], K * 4));
    CK(hipMemcpy(d.ew[o], ew[o].data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.ex[o], ex[o].data(), K * 4, hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(pack_w<false>, dim3(d.ntiles_w), dim3(256), 0, 0, d.wimg[0], d.ntiles_w, d.ew[0], d.gscale[0]);
  hipLaunchKernelGGL(pack_w<true>, dim3(d.ntiles_w), dim3(256), 0, 0, d.wimg[1], d.ntiles_w, d.ew[1], d.gscale[1]);
  hipLaunchKernelGGL(pack_x<false>, dim3(4096), dim3(256), 0, 0, d.xh[0], d.nxt, d.ex[0]);
  hipLaunchKernelGGL(pack_x<true>, dim3(4096), dim3(256), 0, 0, d.xh[1], d.nxt, d.ex[1]);
  CK(hipDeviceSynchronize());
  const int maxwg = (int)(d.nxt / 4);
  d.walk_h.resize((size_t)maxwg * d.ntl);
  for (int wg = 0; wg < maxwg; ++wg)
    for (int i = 0; i < d.ntl; ++i) {
      // a "pdf" = two consecutive tiles 2p, 2p + 1 of the image
      const int p = hash32(wg * 977 + (i >> 1) * 31 + 5) % (d.ntiles_w / 2);
      d.walk_h[(size_t)wg * d.ntl + i] = (2 * p + (i & 1)) | ((i & 1) ? (int32_t)0x80000000 : 0);
    }
  CK(hipMalloc(&d.walk, d.walk_h.size() * 4));
  CK(hipMemcpy(d.walk, d.walk_h.data(), d.walk_h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&d.ll, (size_t)d.nxt * 32 * (d.ntl / 2) * 4 + 4096));
  CK(hipMalloc(&d.stamps, (size_t)maxwg * 16));

  {  // bare MFMA loop
    float* o; CK(hipMalloc(&o, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4000;
    hipLaunchKernelGGL(bare, dim3(1024), dim3(256), 0, 0, o, iters);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bare, dim3(1024), dim3(256), 0, 0, o, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("bare dependent f16 MFMA loop, one wave per SIMD: %.2f ns/MFMA/SIMD\n", ms * 1e6 / (4.0 * iters * 16));
  }
  const int sel = argc > 1 ? atoi(argv[1]) : -1;
  auto want = [&](int g) { return sel < 0 || sel == g; };
  //        WAVES NT ONEACC G  LSE   PREF  STAG
  if (want(0)) {
    run<8, 2, false, 6, true, false, 0>(d, "8w nt2 f16x2  G6", true);
    run<8, 2, false, 6, false, false, 0>(d, "8w nt2 f16x2  G6 noLSE", false);
    run<8, 2, true, 6, true, false, 0>(d, "8w nt2 f16x2s G6", true);
    run<8, 2, true, 6, false, false, 0>(d, "8w nt2 f16x2s G6 noLSE", false);
  }
  if (want(1)) {
    run<8, 2, true, 6, true, true, 0>(d, "8w nt2 f16x2s G6 pref", true);
    run<8, 2, true, 6, true, false, 4>(d, "8w nt2 f16x2s G6 stag4", true);
    run<8, 2, true, 6, true, true, 4>(d, "8w nt2 f16x2s G6 pref stag4", true);
    run<8, 2, true, 4, true, false, 0>(d, "8w nt2 f16x2s G4", true);
  }
  if (want(2)) {
    run<8, 3, true, 6, true, false, 0>(d, "8w nt3 f16x2s G6", true);
    run<8, 3, true, 6, true, true, 0>(d, "8w nt3 f16x2s G6 pref", true);
    run<8, 1, true, 6, true, false, 0>(d, "8w nt1 f16x2s G6", true);
  }
  if (want(3)) {
    run<4, 4, true, 6, true, false, 0>(d, "4w nt4 f16x2s G6", true);
    run<4, 4, true, 6, true, true, 0>(d, "4w nt4 f16x2s G6 pref", true);
    run<4, 4, true, 6, false, true, 0>(d, "4w nt4 f16x2s G6 pref noLSE", false);
    run<4, 4, false, 6, true, true, 0>(d, "4w nt4 f16x2  G6 pref", true);
  }
  if (want(5)) {
    run<8, 2, true, 6, true, false, 0, 8>(d, "8w nt2 f16x2s hand", true);
    run<8, 3, true, 6, true, false, 0, 8>(d, "8w nt3 f16x2s hand", true);
    run<8, 2, true, 6, true, true, 0, 8>(d, "8w nt2 f16x2s hand pref", true);
    run<4, 4, true, 6, true, true, 0, 8>(d, "4w nt4 f16x2s hand pref", true);
    run<4, 4, true, 6, true, false, 0, 8>(d, "4w nt4 f16x2s hand", true);
    run<8, 2, true, 6, true, false, 0, 40>(d, "8w nt2 f16x2s hand roll", true);
    run<8, 3, true, 6, true, false, 0, 40>(d, "8w nt3 f16x2s hand roll", true);
    run<8, 1, true, 6, true, false, 0, 40>(d, "8w nt1 f16x2s hand roll", true);
    run<8, 1, true, 6, true, false, 0, 8>(d, "8w nt1 f16x2s hand", true);
    run<4, 4, true, 6, true, true, 0, 24>(d, "4w nt4 f16x2s hand asm pref", true);
    run<4, 4, true, 6, true, false, 0, 24>(d, "4w nt4 f16x2s hand asm", true);
    run<4, 3, true, 6, true, false, 0, 24>(d, "4w nt3 f16x2s hand asm", true);
    run<4, 2, true, 6, true, false, 0, 24>(d, "4w nt2 f16x2s hand asm", true);
  }
  if (want(7)) {
    runx<true, 1>(d, 14, "x: n=14 wroll", true);
    runx<true, 1>(d, 10, "x: n=10 wroll", true);
    runx<true, 3>(d, 10, "x: n=10 wroll touch", true);
    runx<true, 1>(d, 5, "x: n=5 wroll", true);
    runx<true, 3>(d, 5, "x: n=5 wroll touch", true);
    runx<true, 0>(d, 14, "x: n=14", true);
    runx<true, 0>(d, 10, "x: n=10", true);
    runx<true, 0>(d, 8, "x: n=8", true);
    runx<true, 0>(d, 5, "x: n=5", true);
    runx<false, 0>(d, 10, "x: n=10 noLSE", false);
  }
  if (want(6)) {
    run<8, 2, true, 6, false, false, 0>(d, "8w nt2 noLSE", false);
    run<8, 2, true, 6, false, false, 0, 1>(d, "8w nt2 noLSE nobarrier", false);
    run<8, 2, true, 6, false, false, 0, 2>(d, "8w nt2 noLSE noDMA", false);
    run<8, 2, true, 6, false, false, 0, 3>(d, "8w nt2 noLSE nobarrier noDMA", false);
    run<8, 2, true, 6, false, false, 0, 4>(d, "8w nt2 noLSE noAread", false);
    run<8, 2, true, 6, false, false, 0, 7>(d, "8w nt2 noLSE nobar noDMA noAread", false);
    run<8, 2, true, 6, false, true, 0>(d, "8w nt2 noLSE pref", false);
    run<4, 4, true, 6, false, false, 0>(d, "4w nt4 noLSE", false);
    run<4, 4, true, 6, false, false, 0, 1>(d, "4w nt4 noLSE nobarrier", false);
    run<4, 4, true, 6, false, false, 0, 7>(d, "4w nt4 noLSE nobar noDMA noAread", false);
    run<4, 2, true, 6, false, true, 0>(d, "4w nt2 noLSE pref", false);
    run<4, 2, true, 6, false, false, 0>(d, "4w nt2 noLSE", false);
  }
  if (want(4)) {
    run<4, 2, true, 6, true, true, 0>(d, "4w nt2 f16x2s G6 pref", true);
    run<4, 3, true, 6, true, true, 0>(d, "4w nt3 f16x2s G6 pref", true);   // G NT even
  }
  return 0;
}
