// kaldi_hmm_gmm_amd/csrc/khg_capi.hip -- C-ABI implementation (include/khg_hip.h):
// host-side planning (model tile image, per-utterance pdf lists, in-arc CSR, K1 chunks) and
// the launches of K1 / K2 / K3.  gfx950 only.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>   // DeviceRadixSort: the stable (pdf, frame) sort of K3's bucketing
#include <dlfcn.h>             // RCCL is bound at run time (C1)

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <deque>
#include <utility>
#include <vector>

#include "../../include/khg_hip.h"

#include "khg_k1_loglikes.hip.inc"
#include "khg_k1_pdfmajor.hip.inc"
#include "khg_k1_bf16x3.hip.inc"
#include "khg_k1_f16x2.hip.inc"
#include "khg_k1_wide.hip.inc"
#include "khg_k1_f16x2s.hip.inc"
#include "khg_k2_viterbi.hip.inc"
#include "khg_k3_accstats.hip.inc"
#include "khg_k4_mstep.hip.inc"

// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
extern "C" const char* khg_last_error(void) { return g_err.c_str(); }
int khg_set_error(int code, const std::string& msg) { g_err = msg; return code; }  // shared with khg_host.cpp
extern "C" int khg_version(void) { return 100; }

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return khg_set_error(KHG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

struct khg_timing { std::string name; hipEvent_t e0, e1; };
struct khg_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  static constexpr int NSIDE = 4;
  hipStream_t sides[NSIDE] = {nullptr, nullptr, nullptr, nullptr};  // the serial faithful-decoder kernels run here, beside the main stream's next K1
  int next_side = 0;
  hipStream_t comm_stream = nullptr;   // C1 pieces run here while K3 continues on `stream` (khg_acc_stats_reduce)
  hipEvent_t ev_k3 = nullptr, ev_c1 = nullptr;
  bool own_stream = false;
  int32_t* err_flag_d = nullptr;
  float* dump_d = nullptr;          // 256 floats nobody reads (K1 f16x2s: where the pipeline's first, empty value goes)
  bool timing = false;
  std::vector<khg_timing> timings;
  int opt[KHG_OPT_COUNT] = {};    // khg_ctx_set_option (KHG_OPT_*); the environment variables of include/khg_hip.h only seed the defaults, once, at khg_ctx_create
};
// scoped HIP-event pair around a kernel launch, on the launching stream (only when enabled)
struct KernelTimer {
  khg_ctx* c; size_t idx = 0; bool on; hipStream_t s;
  KernelTimer(khg_ctx* ctx, const char* name, hipStream_t st = nullptr) : c(ctx), on(ctx->timing), s(st ? st : ctx->stream) {
    if (!on) return;
    khg_timing t; t.name = name;
    (void)hipEventCreate(&t.e0); (void)hipEventCreate(&t.e1);
    (void)hipEventRecord(t.e0, s);
    c->timings.push_back(t); idx = c->timings.size() - 1;
  }
  ~KernelTimer() { if (on) (void)hipEventRecord(c->timings[idx].e1, s); }
};

template <class T>
static int dev_alloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return KHG_OK;
}
template <class T>
static int dev_upload(khg_ctx* ctx, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIPCHK(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return KHG_OK;
}
#define DEVFREE(p) do { if (p) { (void)hipFree((void*)(p)); (p) = nullptr; } } while (0)

static void ctx_defaults_from_env(khg_ctx* c);
extern "C" int khg_ctx_create(int device, void* stream, khg_ctx** out) {
  if (!out) return khg_set_error(KHG_E_ARG, "khg_ctx_create: out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return khg_set_error(KHG_E_HIP, "khg_ctx_create: no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n) return khg_set_error(KHG_E_ARG, "khg_ctx_create: bad device index");
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
    return khg_set_error(KHG_E_UNSUPPORTED, std::string("khg_ctx_create: built for gfx950, device is ") + prop.gcnArchName);
  khg_ctx* c = new khg_ctx();
  c->device = device;
  if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
  else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  for (auto& s : c->sides) HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  ctx_defaults_from_env(c);
  int rc = dev_alloc(&c->err_flag_d, 1);
  if (!rc) rc = dev_alloc(&c->dump_d, 256);
  if (rc) { delete c; return rc; }
  HIPCHK(hipMemsetAsync(c->err_flag_d, 0, sizeof(int32_t), c->stream));
  *out = c;
  return KHG_OK;
}
extern "C" int khg_ctx_destroy(khg_ctx* c) {
  if (!c) return KHG_OK;
  (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->sides) (void)hipStreamSynchronize(s);
  DEVFREE(c->err_flag_d); DEVFREE(c->dump_d);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  for (auto& s : c->sides) (void)hipStreamDestroy(s);
  if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
  if (c->ev_k3) (void)hipEventDestroy(c->ev_k3);
  if (c->ev_c1) (void)hipEventDestroy(c->ev_c1);
  delete c;
  return KHG_OK;
}
extern "C" int khg_ctx_set_timing(khg_ctx* c, int on) {
  if (!c) return khg_set_error(KHG_E_ARG, "ctx is NULL");
  c->timing = on != 0;
  return KHG_OK;
}
// drains the recorded (kernel name, milliseconds) pairs; names are '\n'-separated
extern "C" int khg_ctx_get_timings(khg_ctx* c, char* names, int64_t names_cap, float* ms, int32_t cap, int32_t* n_out) {
  if (!c || !n_out) return khg_set_error(KHG_E_ARG, "bad arguments");
  HIPCHK(hipStreamSynchronize(c->stream));
  for (auto& s : c->sides) HIPCHK(hipStreamSynchronize(s));
  int n = 0; std::string all;
  for (auto& t : c->timings) {
    float v = 0.0f;
    (void)hipEventElapsedTime(&v, t.e0, t.e1);
    if (n < cap && ms) ms[n] = v;
    all += t.name; all += '\n';
    (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1);
    ++n;
  }
  c->timings.clear();
  if (names && names_cap > 0) { size_t k = std::min<size_t>(all.size(), (size_t)names_cap - 1); memcpy(names, all.data(), k); names[k] = 0; }
  *n_out = n;
  return KHG_OK;
}
// valid range of every option (inclusive)
static const struct { int lo, hi; } k_opt_range[KHG_OPT_COUNT] = {
  {KHG_K1_AUTO, KHG_K1_F16X2S}, {0, 3}, {0, 6}, {0, 1 << 20}, {-1, 1}, {0, 255}, {0, 1}, {0, 4}, {0, 2}, {0, 1}, {0, 2}, {0, 2}, {0, 2}, {0, 64}, {0, 1}, {0, 1}};
extern "C" int khg_ctx_set_option(khg_ctx* c, int opt, int value) {
  if (!c || opt < 0 || opt >= KHG_OPT_COUNT) return khg_set_error(KHG_E_ARG, "khg_ctx_set_option: bad arguments");
  if (value < k_opt_range[opt].lo || value > k_opt_range[opt].hi)
    return khg_set_error(KHG_E_ARG, "khg_ctx_set_option: option " + std::to_string(opt) + " takes values " + std::to_string(k_opt_range[opt].lo) + " .. " + std::to_string(k_opt_range[opt].hi));
  c->opt[opt] = value;
  return KHG_OK;
}
extern "C" int khg_ctx_get_option(const khg_ctx* c, int opt, int* value) {
  if (!c || !value || opt < 0 || opt >= KHG_OPT_COUNT) return khg_set_error(KHG_E_ARG, "khg_ctx_get_option: bad arguments");
  *value = c->opt[opt];
  return KHG_OK;
}
extern "C" int khg_ctx_set_k1_form(khg_ctx* c, int form) { return khg_ctx_set_option(c, KHG_OPT_K1_FORM, form); }
// Defaults from the environment, read ONCE per context (A/B runs of an unmodified caller): NAME=value, value an integer or one of the words
// listed.  Everything else goes through khg_ctx_set_option.
static void ctx_defaults_from_env(khg_ctx* c) {
  static const struct { const char* name; int opt; const char* words; } tab[] = {
    {"KHG_K1", KHG_OPT_K1_FORM, "auto=0,bf16x3=1,pdf=2,fp32=2,utt=3,f16x2=4,f16x2s=5"},
    {"KHG_K1_ORDER", KHG_OPT_K1_ORDER, "desc=0,none=1,asc=2,tiles=3"},
    {"KHG_K1_NF", KHG_OPT_K1_NF, ""}, {"KHG_K1P_TS", KHG_OPT_K1P_TS, ""}, {"KHG_K1_INTERLEAVE", KHG_OPT_K1_INTERLEAVE, ""},
    {"KHG_K1B_DBG", KHG_OPT_K1_DBG, ""}, {"KHG_K2_INORDER", KHG_OPT_K2_INORDER, ""}, {"KHG_K2_KS", KHG_OPT_K2_KS, ""},
    {"KHG_K2_SERIAL", KHG_OPT_K2_SERIAL, ""}, {"KHG_K2_PROF", KHG_OPT_K2_PROF, ""},
    {"KHG_K3_BUCKET", KHG_OPT_K3_BUCKET, "sort=0,atomic=1,count=2"}, {"KHG_K3_FORM", KHG_OPT_K3_FORM, "auto=0,block=1,valu=2"},
    {"KHG_K3_VALU", KHG_OPT_K3_FORM, "1=2"}, {"KHG_K3_PHASEB", KHG_OPT_K3_PHASE_B, "f64=0,f32=1,f16=2"},
    {"KHG_K3_NY", KHG_OPT_K3_NY, ""}, {"KHG_DEBUG", KHG_OPT_DEBUG, ""}, {"KHG_K3_PHASEA", KHG_OPT_K3_PHASE_A, "auto=0,f16=0,f32=1"}};
  c->opt[KHG_OPT_K1_INTERLEAVE] = -1;
  c->opt[KHG_OPT_K1P_TS] = 1024;
  c->opt[KHG_OPT_K3_PHASE_B] = 2;        // both phases of K3's wave form on the fp16 matrix cores where they apply (else the fp64 pipe)
  for (const auto& t : tab) {
    const char* e = getenv(t.name);
    if (!e || !*e) continue;
    int v = atoi(e);
    bool word = false;
    for (const char* w = t.words; *w;) {                 // "word=value,word=value"
      const char* eq = strchr(w, '=');
      const size_t n = (size_t)(eq - w);
      if (strlen(e) == n && strncmp(e, w, n) == 0) { v = atoi(eq + 1); word = true; break; }
      const char* comma = strchr(eq, ',');
      if (!comma) break;
      w = comma + 1;
    }
    if (!word && *t.words && !(e[0] >= '0' && e[0] <= '9') && e[0] != '-') continue;      // an unknown word: ignored
    if (v >= k_opt_range[t.opt].lo && v <= k_opt_range[t.opt].hi) c->opt[t.opt] = v;
  }
}
static int check_err_flag(khg_ctx* c, const char* where);
extern "C" int khg_ctx_sync(khg_ctx* c) {
  if (!c) return khg_set_error(KHG_E_ARG, "ctx is NULL");
  return check_err_flag(c, "khg_ctx_sync");   // synchronises the stream, then reports deferred kernel errors
}
// read-and-clear the device error word; maps bits to the reference's exceptions
static int check_err_flag(khg_ctx* c, const char* where) {
  int32_t f = 0;
  for (auto& s : c->sides) HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipMemcpyAsync(&f, c->err_flag_d, sizeof(f), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (f) {
    HIPCHK(hipMemsetAsync(c->err_flag_d, 0, sizeof(int32_t), c->stream));
    if (f & 1) return khg_set_error(KHG_E_RUNTIME, std::string(where) + ": Invalid answer (overflow or invalid variances/features?)");
    if (f & 2) return khg_set_error(KHG_E_RUNTIME, std::string(where) + ": internal queue overflow in the faithful decoder");
    if (f & 8) return khg_set_error(KHG_E_RUNTIME, std::string(where) + ": internal error: K3 work items exceed their bound");
    return khg_set_error(KHG_E_RUNTIME, std::string(where) + ": pdf-id out of range (graph/model mismatch)");
  }
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
// K0: pack the K1 tile image from the row-major parameters (one workgroup per W tile).
// planes q = 0..3 of [32][ROW]: q=0/1: means_invvars at even/odd d, q=2/3: -0.5*inv_vars (exact
// scaling) at even/odd d, element s of a row <-> d = 2s + (q&1); then gconst[32].  Padding rows:
// W = 0, gconst = -inf (they contribute exp2(-inf) = 0 to the log-sum-exp).  Also writes the
// row-major -0.5*inv_vars copy K3 uses.
template <int KQ>
__global__ __launch_bounds__(256) void k0_pack_tiles(const float* __restrict__ gconsts, const float* __restrict__ miv,
                                                      const float* __restrict__ iv, const int32_t* __restrict__ gauss_off,
                                                      const int32_t* __restrict__ pdf_tile_off, const int32_t* __restrict__ tile_pdf,
                                                      int D, float* __restrict__ wimg, float* __restrict__ nhiv) {
  constexpr int ROW = khg_row_floats(KQ), TILE = khg_tile_floats(KQ);
  const int t = blockIdx.x, p = tile_pdf[t];
  const int g_first = gauss_off[p] + 32 * (t - pdf_tile_off[p]);
  const int nrow = min(32, gauss_off[p + 1] - g_first);
  float* img = wimg + (size_t)t * TILE;
  for (int i = threadIdx.x; i < TILE; i += 256) {
    float v = 0.0f;
    if (i < 4 * 32 * ROW) {
      const int q = i / (32 * ROW), r = (i / ROW) & 31, s = i % ROW;
      const int d = 2 * s + (q & 1);
      if (r < nrow && d < D) {
        const size_t src = (size_t)(g_first + r) * D + d;
        v = (q & 2) ? -0.5f * iv[src] : miv[src];
      }
    } else if (i < 4 * 32 * ROW + 32) {
      const int r = i - 4 * 32 * ROW;
      v = r < nrow ? gconsts[g_first + r] : -INFINITY;
    }
    img[i] = v;
  }
  for (int i = threadIdx.x; i < nrow * D; i += 256) nhiv[(size_t)g_first * D + i] = -0.5f * iv[(size_t)g_first * D + i];
}

// A model image that K1 launches on several contexts' streams read and that is re-packed in place when the parameters or the
// scale exponents change: the pack waits for every recorded reader, a reader on another stream waits for the pack.
struct ImgSync {
  hipEvent_t packed = nullptr;
  hipStream_t pack_stream = nullptr;
  std::vector<std::pair<hipStream_t, hipEvent_t>> readers;
  int before_pack(hipStream_t s) {
    for (auto& r : readers)
      if (r.first != s) HIPCHK(hipStreamWaitEvent(s, r.second, 0));
    return KHG_OK;
  }
  int after_pack(hipStream_t s) {
    if (!packed) HIPCHK(hipEventCreateWithFlags(&packed, hipEventDisableTiming));
    HIPCHK(hipEventRecord(packed, s));
    pack_stream = s;
    return KHG_OK;
  }
  int before_read(hipStream_t s) {
    if (packed && pack_stream != s) HIPCHK(hipStreamWaitEvent(s, packed, 0));
    return KHG_OK;
  }
  int after_read(hipStream_t s) {
    for (auto& r : readers)
      if (r.first == s) { HIPCHK(hipEventRecord(r.second, s)); return KHG_OK; }
    hipEvent_t e = nullptr;
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(e, s));
    readers.emplace_back(s, e);
    return KHG_OK;
  }
  void destroy() {
    if (packed) (void)hipEventDestroy(packed);
    for (auto& r : readers) (void)hipEventDestroy(r.second);
    packed = nullptr; readers.clear();
  }
};

// ------------------------------------------------------------------------------------------
struct khg_model {
  khg_ctx* ctx = nullptr;
  int32_t P = 0, D = 0, KQ = 0, ntiles = 0;
  int64_t sumG = 0;
  std::vector<int32_t> gauss_off, pdf_tile_off;
  float* wimg_d = nullptr;
  int32_t* pdf_tile_off_d = nullptr;
  int32_t* gauss_off_d = nullptr;
  float *gconsts_d = nullptr, *miv_d = nullptr, *iv_d = nullptr, *nhiv_d = nullptr;
  float* weights_d = nullptr;   // only the device M-step needs them (khg_model_set_weights)
  bool has_weights = false;
  int32_t wimg_tiles = 0;       // tiles wimg_d was allocated for
  char* wimgb_d = nullptr;      // bf16x3 K1 image (khg_k1_bf16x3.hip.inc), k1b_tile_bytes(KS) per 32-Gaussian tile
  int32_t wimgb_tiles = 0, KS = 0;
  bool wimgb_valid = false;       // packed from the current parameters (lazily: only the bf16x3 form reads it)
  // f16x2 K1 image (khg_k1_f16x2.hip.inc): packed lazily by khg_loglikes with the scale exponents of the utterance set
  char* wimgh_d = nullptr;
  int32_t wimgh_tiles = 0;
  std::vector<int32_t> wimgh_ex;   // exponents the image was packed with (empty: stale)
  ImgSync wimgh_sync, wimgb_sync;
  // f16x2s K1 image (khg_k1_f16x2s.hip.inc): packed lazily with the weight exponents ew[k] = S - ex[k] and gconst 2^S
  char* wimgs_d = nullptr;
  int32_t wimgs_tiles = 0;
  std::vector<int32_t> wimgs_key;  // [ex[0..K) of the set, S] the image was packed with (empty: stale)
  float* ubound_d = nullptr; int32_t ubound_tiles = 0; bool ubound_valid = false;   // BAND form of K1: per-pdf upper bound of the log-likelihood (k1s_ubound)
  std::vector<int32_t> xs_ex_seen; // element-wise minimum of the feature exponents of the sets scored so far (f16x2s)
  ImgSync wimgs_sync;
  std::vector<float> wmax;         // per k = 2 d + kind: max |W[.][k]| of the current parameters (empty: not computed)
  // K3's fp16 phase A (k3_accumulate_wave<NB, true>): scale exponents derived from the model alone; cleared with wmax
  std::vector<float> k3_xb;        // per dim: max over the Gaussians of |mean| + 8 sigma (empty: not computed)
  std::vector<int32_t> k3_ex;      // [80] per k = 2 d + kind
  int32_t k3_S = 0;
  bool k3_f16_ok = false;          // the model side of the form's domain holds
  int32_t* k3_ex_d = nullptr;
  float gcmax = 0.0f;              // max |gconst| over the finite ones (valid with wmax)
  int32_t* tile_pdf_d = nullptr;   // tile -> pdf map of the current layout
  K4Res* k4_res_d = nullptr;       // per-pdf results of the M-step in progress (khg_model_mle_update*)
  int32_t k4_res_P = 0;
};

// (Re)build everything derived from gauss_off + the row-major parameters in HBM: the tile offsets, the K1
// tile image and the -0.5*inv_vars copy K3 reads -- packed ON THE DEVICE (k0_pack_tiles), no host-side
// 113 MB image, no extra copies.  Used by khg_model_create and after the device M-step.
static int model_pack(khg_ctx* ctx, khg_model* m) {
  const int P = m->P, D = m->D;
  m->pdf_tile_off.resize((size_t)P + 1);
  int nt = 0;
  for (int p = 0; p < P; ++p) { m->pdf_tile_off[p] = nt; nt += (m->gauss_off[p + 1] - m->gauss_off[p] + 31) / 32; }
  m->pdf_tile_off[P] = nt;
  m->ntiles = nt;
  const int TILE = m->KQ ? khg_tile_floats(m->KQ) : 0;
  if (!m->pdf_tile_off_d) { int rc = dev_alloc(&m->pdf_tile_off_d, (size_t)P + 1); if (rc) return rc; }
  if (!m->gauss_off_d) { int rc = dev_alloc(&m->gauss_off_d, (size_t)P + 1); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(m->pdf_tile_off_d, m->pdf_tile_off.data(), sizeof(int32_t) * ((size_t)P + 1), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(m->gauss_off_d, m->gauss_off.data(), sizeof(int32_t) * ((size_t)P + 1), hipMemcpyHostToDevice, ctx->stream));
  if (!m->nhiv_d) { int rc = dev_alloc(&m->nhiv_d, (size_t)m->sumG * D); if (rc) return rc; }
  if (m->KQ == 0) {
    // any-dimension model (D > 80): no tile images; K3 still reads -0.5 * inv_vars
    m->KS = 0;
    m->wimgb_valid = false;
    m->wimgh_ex.clear(); m->wimgs_key.clear(); m->ubound_valid = false; m->wmax.clear(); m->k3_xb.clear();
    const int64_t n = m->sumG * D;
    hipLaunchKernelGGL(k0_nhalf, dim3((int)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, ctx->stream, m->iv_d, n, m->nhiv_d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    return KHG_OK;
  }
  if (!m->wimg_d || m->wimg_tiles < nt) {
    DEVFREE(m->wimg_d);
    int rc = dev_alloc(&m->wimg_d, (size_t)nt * TILE);
    if (rc) return rc;
    m->wimg_tiles = nt;
  }
  m->KS = m->KQ == 10 ? 5 : 10;
  m->wimgb_valid = false;
  std::vector<int32_t> tile_pdf((size_t)nt);   // tile -> pdf map for the pack kernel
  for (int p = 0; p < P; ++p)
    for (int t = m->pdf_tile_off[p]; t < m->pdf_tile_off[p + 1]; ++t) tile_pdf[(size_t)t] = p;
  DEVFREE(m->tile_pdf_d);
  m->wimgh_ex.clear();
  m->wimgs_key.clear(); m->ubound_valid = false;
  m->wmax.clear(); m->k3_xb.clear();
  int rc = dev_upload(ctx, &m->tile_pdf_d, tile_pdf);
  int32_t* tile_pdf_d = m->tile_pdf_d;
  if (!rc) {
    KernelTimer kt(ctx, "k0_pack_tiles");
    if (m->KQ == 10) hipLaunchKernelGGL(k0_pack_tiles<10>, dim3(nt), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, tile_pdf_d, D, m->wimg_d, m->nhiv_d);
    else hipLaunchKernelGGL(k0_pack_tiles<20>, dim3(nt), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, tile_pdf_d, D, m->wimg_d, m->nhiv_d);
  }
  if (!rc) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // tile_pdf and the host offset vectors are free after this
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  return rc;
}

extern "C" int khg_model_create(khg_ctx* ctx, int32_t P, int32_t D, const int32_t* gauss_off,
                                const float* gconsts, const float* miv, const float* iv, khg_model** out) {
  if (!ctx || !out || P <= 0 || D <= 0 || !gauss_off || !gconsts || !miv || !iv)
    return khg_set_error(KHG_E_ARG, "khg_model_create: bad arguments");
  if (D > KHG_MAX_DIM) return khg_set_error(KHG_E_UNSUPPORTED, "khg_model_create: feature dim > " + std::to_string(KHG_MAX_DIM) + " is not supported (a 64-frame chunk of rows must fit LDS)");
  if (gauss_off[0] != 0) return khg_set_error(KHG_E_ARG, "khg_model_create: gauss_off[0] != 0");
  for (int p = 0; p < P; ++p)
    if (gauss_off[p + 1] <= gauss_off[p]) return khg_set_error(KHG_E_ARG, "khg_model_create: every pdf needs >= 1 Gaussian");
  khg_model* m = new khg_model();
  m->ctx = ctx; m->P = P; m->D = D;
  m->KQ = (D <= 40) ? 10 : (D <= 80) ? 20 : 0;      // 0: no tile image; K1 / K3 run their any-dimension forms (k1w_loglikes, k3_accumulate<0>)
  m->gauss_off.assign(gauss_off, gauss_off + P + 1);
  m->sumG = gauss_off[P];
  // the row-major parameters go up as they are (K3 and the device M-step read them)
  auto up = [&](float** dst, const float* src, size_t n) -> int {
    int r = dev_alloc(dst, n);
    if (r) return r;
    HIPCHK(hipMemcpyAsync(*dst, src, sizeof(float) * n, hipMemcpyHostToDevice, ctx->stream));
    return KHG_OK;
  };
  int rc = up(&m->gconsts_d, gconsts, (size_t)m->sumG);
  if (!rc) rc = up(&m->miv_d, miv, (size_t)m->sumG * D);
  if (!rc) rc = up(&m->iv_d, iv, (size_t)m->sumG * D);
  if (!rc) rc = model_pack(ctx, m);   // ends with a stream sync: the caller's arrays are free after this
  if (rc) { khg_model_destroy(m); return rc; }
  *out = m;
  return KHG_OK;
}
extern "C" int khg_model_destroy(khg_model* m) {
  if (!m) return KHG_OK;
  m->wimgh_sync.destroy(); m->wimgb_sync.destroy(); m->wimgs_sync.destroy();
  DEVFREE(m->wimg_d); DEVFREE(m->wimgb_d); DEVFREE(m->wimgh_d); DEVFREE(m->wimgs_d); DEVFREE(m->ubound_d); DEVFREE(m->k3_ex_d); DEVFREE(m->tile_pdf_d); DEVFREE(m->k4_res_d); DEVFREE(m->pdf_tile_off_d); DEVFREE(m->gauss_off_d);
  DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d); DEVFREE(m->weights_d);
  delete m;
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
struct khg_tm {
  khg_ctx* ctx = nullptr;
  int32_t num_tids = 0, max_pdf = -1;
  std::vector<int32_t> id2pdf;
  int32_t* id2pdf_d = nullptr;
  float* trans_cost_d = nullptr;
  bool has_trans_cost = false;
};
extern "C" int khg_tm_create(khg_ctx* ctx, int32_t num_tids, const int32_t* id2pdf, khg_tm** out) {
  if (!ctx || !out || num_tids <= 0 || !id2pdf) return khg_set_error(KHG_E_ARG, "khg_tm_create: bad arguments");
  khg_tm* t = new khg_tm();
  t->ctx = ctx; t->num_tids = num_tids;
  t->id2pdf.assign(id2pdf, id2pdf + num_tids + 1);
  for (int i = 1; i <= num_tids; ++i) {
    if (id2pdf[i] < 0) { delete t; return khg_set_error(KHG_E_ARG, "khg_tm_create: negative pdf-id"); }
    t->max_pdf = std::max(t->max_pdf, id2pdf[i]);
  }
  int rc = dev_upload(ctx, &t->id2pdf_d, t->id2pdf);
  if (!rc) rc = dev_alloc(&t->trans_cost_d, (size_t)num_tids + 1);
  if (!rc) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e)); }
  if (rc) { khg_tm_destroy(t); return rc; }
  *out = t;
  return KHG_OK;
}
extern "C" int khg_tm_set_trans_cost(khg_tm* t, const float* cost) {
  if (!t) return khg_set_error(KHG_E_ARG, "tm is NULL");
  if (!cost) { t->has_trans_cost = false; return KHG_OK; }
  HIPCHK(hipMemcpyAsync(t->trans_cost_d, cost, sizeof(float) * ((size_t)t->num_tids + 1), hipMemcpyHostToDevice, t->ctx->stream));
  HIPCHK(hipStreamSynchronize(t->ctx->stream));
  t->has_trans_cost = true;
  return KHG_OK;
}
extern "C" int khg_tm_destroy(khg_tm* t) {
  if (!t) return KHG_OK;
  DEVFREE(t->id2pdf_d); DEVFREE(t->trans_cost_d);
  delete t;
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
struct khg_utts {
  khg_ctx* ctx = nullptr;
  int32_t n_utt = 0, D = 0;
  int64_t N = 0;  // total frames
  bool has_graphs = false;
  std::vector<int64_t> frame_off, state_off, pdf_off, ll_off, bp_off, path_off, words_off;
  std::vector<int32_t> pdfs;
  int32_t max_states = 0, max_inarcs = 0, max_indeg = 0, max_outdeg = 0;
  bool same_col = true;          // every state's in-arcs read one score row (reorder = true training graphs)
  int32_t pdfs_checked_P = -1;   // model size the pdf lists were last validated against
  bool has_eps = false;
  // device
  const float* feats_d = nullptr; bool own_feats = false;
  int64_t *frame_off_d = nullptr, *state_off_d = nullptr, *pdf_off_d = nullptr, *ll_off_d = nullptr;
  int32_t *pdfs_d = nullptr, *start_d = nullptr;
  KwChunk* wchunks_d = nullptr;   // 64-frame chunks of the any-dimension K1 (khg_k1_wide.hip.inc)
  int32_t n_wchunks = 0;
  int64_t *in_off_d = nullptr, *out_off_d = nullptr;
  int32_t *in_src_d = nullptr, *in_col_d = nullptr, *in_tid_d = nullptr, *in_olabel_d = nullptr, *out_inidx_d = nullptr;
  float *in_w_d = nullptr, *final_d = nullptr;
  // K1
  K1Chunk* chunks_d = nullptr; int32_t n_chunks = 0; int32_t chunk_kq = 0;
  int64_t* tile_off_d = nullptr; int32_t* tiles_d = nullptr;
  std::vector<int32_t> pdf_first;  // per (utterance, listed pdf): first frame at which any state emitting it can hold a token
  std::vector<int32_t> pdf_last;   // ... last frame at which an arc carrying it can still lead to a final state by the utterance's end (-1: never)
  // BAND form of the default K1 (khg_loglikes_band): what khg_align needs to recompute the utterances whose beam certificate fails
  int ll_mode = 0;                 // how the resident scores were computed: 0 every cell, 1 from the first needed tile, 2 band
  K1sArgs band_args; khg_model* band_model = nullptr; int band_ks = 0; size_t band_lds = 0;
  int tiles_reach = -1;            // whether the walk lists carry those first frames (reachable-only K1) or zeros
  std::vector<int32_t> tiles_pto;  // the model tile layout (pdf_tile_off) the walk lists were built for
  int64_t* tile2_off_d = nullptr; int32_t* tiles2_d = nullptr; std::vector<int32_t> tiles2_pto; int tiles2_reach = -1;   // bf16x3: pair walk
  float* ll_d = nullptr; int64_t ll_total = 0; bool ll_valid = false;
  // K1, pdf-major form: repacked features (once), work plan (per reachable flag)
  float* xpl_d = nullptr; int64_t* utt_xtile_off_d = nullptr; int32_t xpl_kq = 0;
  K1pEntry* p_ents_d = nullptr; K1pSlice* p_slices_d = nullptr; int32_t p_nslices = 0; int p_reach = -1; int32_t p_P = -1;
  int32_t p_grp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // slices per block count (index 1..8); the plan also depends on the model's gauss_off
  std::vector<int32_t> p_goff;
  // K1, bf16x3 form: B fragments of the features (once), workgroup chunks
  k1b_u32x4* xb3_d = nullptr; int64_t* utt_x32_off_d = nullptr; int32_t xb3_ks = 0;
  K1bChunk* bchunks_d = nullptr; int32_t n_bchunks = 0, bchunk_nt = 0;
  int32_t* x32_utt_d = nullptr; int64_t n_x32 = 0;        // 32-frame tile -> utterance
  // K1, f16x2 form: B fragments packed with the scale exponents xh_ex (per k = 2 d + kind)
  k1b_u32x4* xh_d = nullptr; int32_t xh_ks = 0; std::vector<int32_t> xh_ex; int32_t* xh_ex_d = nullptr;
  std::vector<float> xmax;         // per feature dimension: max |x| over the set (empty: not computed)
  // K1, f16x2s form: B fragments packed once with the set's own exponents xs_ex (feature columns peak in [2^14, 2^15)),
  // workgroup chunks of <= k1s_nmax tiles, one unit per (utterance, listed pdf)
  k1b_u32x4* xs_d = nullptr; int32_t xs_ks = 0; std::vector<int32_t> xs_ex; int32_t* xs_ex_d = nullptr;
  K1sChunk* schunks_d = nullptr; int32_t n_schunks = 0, schunk_nmax = 0;
  K1sUnit* sunits_d = nullptr; std::vector<int32_t> sunits_pto; int sunits_reach = -1;
  // K2 scratch / outputs
  uint8_t* bp_d = nullptr; int64_t *bp_off_d = nullptr, *path_off_d = nullptr, *words_off_d = nullptr;
  double* layer_best_d = nullptr; int32_t* layer_cnt_d = nullptr; int32_t* path_d = nullptr;
  unsigned char* k2_gscratch_d = nullptr; size_t k2_gscratch_bytes = 0;   // K2 tables of graphs too large for LDS
  int32_t* k2_order_d = nullptr;   // utterances by length, longest first: the DP kernel's launch order
  int32_t *ali_d = nullptr, *words_d = nullptr, *num_words_d = nullptr, *status_d = nullptr;
  float* like_d = nullptr;
  bool ali_valid = false;
  hipEvent_t ev_dp = nullptr, ev_ali = nullptr;   // K2-DP done (main) -> faithful kernel (side) -> alignment complete
  bool ali_pending = false;
  // K3 scratch
  int32_t *pdf_count_d = nullptr, *pdf_cursor_d = nullptr, *frame_ids_d = nullptr;
  uint32_t *sort_keys_d = nullptr, *sort_keys_out_d = nullptr, *sort_vals_d = nullptr; void* sort_tmp_d = nullptr; size_t sort_tmp_bytes = 0;
  int32_t* cs_hist_d = nullptr; size_t cs_hist_n = 0; int64_t* cs_tot_d = nullptr; size_t cs_tot_n = 0;   // counting-sort bucketing (k3_cs_*)
  double *k3_part_d = nullptr, *k3_llpart_d = nullptr; size_t k3_part_n = 0, k3_llpart_n = 0;   // wave-form K3: slice images / per-pdf log-likes
  void* k3_items_d = nullptr; int32_t* k3_item_off_d = nullptr; size_t k3_items_n = 0, k3_item_off_n = 0;   // K3 work items (k3_make_items)
  int64_t* pdf_start_d = nullptr; unsigned long long* tid_count_d = nullptr;
  int32_t k3_P = 0, k3_tids = 0;
};

static void plan_ll(khg_utts* u) {
  u->ll_off.assign(u->n_utt + 1, 0);
  for (int i = 0; i < u->n_utt; ++i) {
    int64_t T = u->frame_off[i + 1] - u->frame_off[i];
    int64_t tpad = (T + 31) & ~int64_t(31);
    u->ll_off[i + 1] = u->ll_off[i] + (u->pdf_off[i + 1] - u->pdf_off[i]) * tpad;
  }
  u->ll_total = u->ll_off[u->n_utt];
}

extern "C" int khg_utts_create(khg_ctx* ctx, const khg_tm* tm, int32_t n_utt, int32_t D,
                               const int64_t* frame_off, const float* feats_h, const float* feats_dv,
                               const int64_t* state_off, const int32_t* start, const int64_t* arc_off,
                               const int32_t* ilabel, const int32_t* olabel, const float* weight,
                               const int32_t* nextstate, const float* final_w, khg_utts** out) {
  if (!ctx || !out || n_utt <= 0 || D <= 0 || !frame_off || (!feats_h && !feats_dv))
    return khg_set_error(KHG_E_ARG, "khg_utts_create: bad arguments");
  if (frame_off[0] != 0) return khg_set_error(KHG_E_ARG, "khg_utts_create: frame_off[0] != 0");
  for (int i = 0; i < n_utt; ++i)
    if (frame_off[i + 1] < frame_off[i]) return khg_set_error(KHG_E_ARG, "khg_utts_create: frame_off not monotone");
  khg_utts* u = new khg_utts();
  u->ctx = ctx; u->n_utt = n_utt; u->D = D;
  u->frame_off.assign(frame_off, frame_off + n_utt + 1);
  u->N = frame_off[n_utt];
  int rc = KHG_OK;
  auto fail = [&](int code, const std::string& msg) { khg_utts_destroy(u); return khg_set_error(code, msg); };
  if (feats_dv) { u->feats_d = feats_dv; u->own_feats = false; }
  else {
    float* p = nullptr;
    rc = dev_alloc(&p, (size_t)u->N * D);
    if (rc) { khg_utts_destroy(u); return rc; }
    u->feats_d = p; u->own_feats = true;
    if (u->N) {
      hipError_t e = hipMemcpyAsync(p, feats_h, sizeof(float) * (size_t)u->N * D, hipMemcpyHostToDevice, ctx->stream);
      if (e != hipSuccess) return fail(KHG_E_HIP, hipGetErrorString(e));
    }
  }
  rc = dev_upload(ctx, &u->frame_off_d, u->frame_off);
  if (rc) { khg_utts_destroy(u); return rc; }
  u->pdf_off.assign(n_utt + 1, 0);

  if (state_off && state_off[n_utt] > 0) {
    if (!tm || !start || !arc_off || !ilabel || !olabel || !weight || !nextstate || !final_w)
      return fail(KHG_E_ARG, "khg_utts_create: graph arrays / tm missing");
    u->has_graphs = true;
    u->state_off.assign(state_off, state_off + n_utt + 1);
    const int64_t NS = state_off[n_utt], NA = arc_off[NS];
    std::vector<int64_t> in_off(NS + 1, 0), out_off(arc_off, arc_off + NS + 1);
    std::vector<int32_t> in_src(NA), in_col(NA), in_tid(NA), in_ol(NA), out_inidx(NA);
    std::vector<float> in_w(NA);
    std::vector<int32_t> tmp_pdfs, cursor;
    u->bp_off.assign(n_utt + 1, 0); u->path_off.assign(n_utt + 1, 0); u->words_off.assign(n_utt + 1, 0);
    for (int i = 0; i < n_utt; ++i) {
      const int64_t s0 = state_off[i], S = state_off[i + 1] - s0;
      const int64_t a0 = arc_off[s0], a1 = arc_off[s0 + S], A = a1 - a0;
      const int64_t T = frame_off[i + 1] - frame_off[i];
      if (S < 0 || A < 0) return fail(KHG_E_ARG, "khg_utts_create: offsets not monotone");
      if (start[i] >= S) return fail(KHG_E_ARG, "khg_utts_create: start state out of range");
      if (S > 65535) return fail(KHG_E_UNSUPPORTED, "khg_utts_create: more than 65535 states in one decoding graph");
      u->max_states = std::max<int64_t>(u->max_states, S);
      u->max_inarcs = std::max<int64_t>(u->max_inarcs, A);
      for (int64_t s = 0; s < S; ++s) u->max_outdeg = std::max<int32_t>(u->max_outdeg, (int32_t)(arc_off[s0 + s + 1] - arc_off[s0 + s]));
      // pdf list of this utterance = distinct id2pdf[ilabel] over its arcs
      tmp_pdfs.clear();
      int64_t nwords = 0;
      for (int64_t a = a0; a < a1; ++a) {
        int l = ilabel[a];
        if (l < 0 || l > tm->num_tids)
          return fail(KHG_E_RUNTIME, "AddTransitionProbs: invalid symbol " + std::to_string(l) + " on graph input side.");
        if (l >= 1) tmp_pdfs.push_back(tm->id2pdf[l]); else u->has_eps = true;
        if (nextstate[a] < 0 || nextstate[a] >= S) return fail(KHG_E_ARG, "khg_utts_create: nextstate out of range");
        if (olabel[a] != 0) ++nwords;
      }
      std::sort(tmp_pdfs.begin(), tmp_pdfs.end());
      tmp_pdfs.erase(std::unique(tmp_pdfs.begin(), tmp_pdfs.end()), tmp_pdfs.end());
      if (tmp_pdfs.size() > 32767) return fail(KHG_E_UNSUPPORTED, "khg_utts_create: more than 32767 distinct pdfs on one decoding graph");
      u->pdf_off[i + 1] = u->pdf_off[i] + (int64_t)tmp_pdfs.size();
      u->pdfs.insert(u->pdfs.end(), tmp_pdfs.begin(), tmp_pdfs.end());
      {
        // first frame each listed pdf can be needed at: a token can sit in state s after no fewer than
        // dmin[s] emitting arcs (0-1 BFS from the start state), so the score of an arc's pdf is first
        // read at frame dmin[src].  K1 may skip (pdf, frame) cells before that (khg_loglikes_reachable).
        std::vector<int32_t> dmin((size_t)S, INT32_MAX);
        std::vector<int32_t> dq;
        if (start[i] >= 0) {
          std::deque<int32_t> q;
          dmin[start[i]] = 0; q.push_back(start[i]);
          while (!q.empty()) {
            const int s = q.front(); q.pop_front();
            for (int64_t a = arc_off[s0 + s]; a < arc_off[s0 + s + 1]; ++a) {
              const int d = nextstate[a], wgt = ilabel[a] >= 1 ? 1 : 0;
              if (dmin[s] + wgt < dmin[d]) {
                dmin[d] = dmin[s] + wgt;
                if (wgt) q.push_back(d); else q.push_front(d);
              }
            }
          }
        }
        // ... and the last: from state d a final state is no fewer than dfin[d] emitting arcs away (0-1 BFS over the reversed
        // graph from the final states), so an arc into d consumed at frame t leaves T - 1 - t frames, enough iff t <= T - 1 - dfin[d].
        // A token past that can never reach a final state: the BAND form of K1 does not compute what only such tokens read.
        std::vector<int32_t> dfin((size_t)S, INT32_MAX);
        {
          std::vector<int64_t> roff((size_t)S + 1, 0);
          for (int64_t a = a0; a < a1; ++a) roff[(size_t)nextstate[a] + 1]++;
          for (int64_t s = 0; s < S; ++s) roff[(size_t)s + 1] += roff[(size_t)s];
          std::vector<int32_t> rsrc((size_t)A), rw((size_t)A), rc_((size_t)S, 0);
          for (int64_t s = 0; s < S; ++s)
            for (int64_t a = arc_off[s0 + s]; a < arc_off[s0 + s + 1]; ++a) {
              const size_t pos = (size_t)(roff[(size_t)nextstate[a]] + rc_[(size_t)nextstate[a]]++);
              rsrc[pos] = (int32_t)s; rw[pos] = ilabel[a] >= 1 ? 1 : 0;
            }
          std::deque<int32_t> q;
          for (int64_t s = 0; s < S; ++s)
            if (final_w[s0 + s] != std::numeric_limits<float>::infinity()) { dfin[(size_t)s] = 0; q.push_back((int32_t)s); }
          while (!q.empty()) {
            const int d = q.front(); q.pop_front();
            for (int64_t k = roff[(size_t)d]; k < roff[(size_t)d + 1]; ++k) {
              const int s = rsrc[(size_t)k], wgt = rw[(size_t)k];
              if (dfin[(size_t)d] + wgt < dfin[(size_t)s]) {
                dfin[(size_t)s] = dfin[(size_t)d] + wgt;
                if (wgt) q.push_back(s); else q.push_front(s);
              }
            }
          }
        }
        const size_t base = u->pdf_first.size();
        u->pdf_first.resize(base + tmp_pdfs.size(), INT32_MAX);
        u->pdf_last.resize(base + tmp_pdfs.size(), -1);
        for (int64_t s = 0; s < S; ++s) {
          if (dmin[s] == INT32_MAX) continue;
          for (int64_t a = arc_off[s0 + s]; a < arc_off[s0 + s + 1]; ++a) {
            if (ilabel[a] < 1) continue;
            const size_t j = (size_t)(std::lower_bound(tmp_pdfs.begin(), tmp_pdfs.end(), tm->id2pdf[ilabel[a]]) - tmp_pdfs.begin());
            u->pdf_first[base + j] = std::min(u->pdf_first[base + j], dmin[s]);
            const int df = dfin[(size_t)nextstate[a]];
            if (df != INT32_MAX) u->pdf_last[base + j] = std::max<int32_t>(u->pdf_last[base + j], (int32_t)std::max<int64_t>(-1, T - 1 - df));
          }
        }
      }
      // in-arc CSR: stable bucketing by destination (ties in the DP then resolve to the lowest
      // original arc index, like a strict '<' scan over arcs in file order)
      for (int64_t a = a0; a < a1; ++a) in_off[s0 + nextstate[a] + 1]++;
      in_off[s0] = a0;
      for (int64_t s = 0; s < S; ++s) {
        if (in_off[s0 + s + 1] > 254) return fail(KHG_E_UNSUPPORTED, "khg_utts_create: a state has more than 254 incoming arcs");
        u->max_indeg = std::max<int32_t>(u->max_indeg, (int32_t)in_off[s0 + s + 1]);
        in_off[s0 + s + 1] += in_off[s0 + s];
      }
      cursor.assign(S, 0);
      for (int64_t s = 0; s < S; ++s) {
        for (int64_t a = arc_off[s0 + s]; a < arc_off[s0 + s + 1]; ++a) {
          int d = nextstate[a];
          int64_t pos = in_off[s0 + d] + cursor[d]++;
          in_src[pos] = (int32_t)s;
          in_tid[pos] = ilabel[a];
          in_ol[pos] = olabel[a];
          in_w[pos] = weight[a];
          int col = -1;
          if (ilabel[a] >= 1)
            col = (int)(std::lower_bound(tmp_pdfs.begin(), tmp_pdfs.end(), tm->id2pdf[ilabel[a]]) - tmp_pdfs.begin());
          in_col[pos] = col;
          out_inidx[a] = (int32_t)(pos - a0);
        }
      }
      for (int64_t s = 0; s < S && u->same_col; ++s)
        for (int64_t k = in_off[s0 + s] + 1; k < in_off[s0 + s + 1]; ++k)
          if (in_col[k] != in_col[in_off[s0 + s]]) { u->same_col = false; break; }
      // generic path: one byte per (layer, state); fast path: one dword per (group of eight layers, lane), whole waves
      u->bp_off[i + 1] = u->bp_off[i] + std::max<int64_t>((T + 1) * std::max<int64_t>((S + 15) & ~int64_t(15), 512), ((T >> 3) + 1) * (S + 256) * 4);
      u->path_off[i + 1] = u->path_off[i] + T + S + 8;
      u->words_off[i + 1] = u->words_off[i] + nwords;
    }
    in_off[NS] = NA;
    std::vector<int32_t> startv(start, start + n_utt);
    std::vector<float> finalv(final_w, final_w + NS);
    rc = dev_upload(ctx, &u->state_off_d, u->state_off);
    if (!rc) rc = dev_upload(ctx, &u->start_d, startv);
    if (!rc) rc = dev_upload(ctx, &u->in_off_d, in_off);
    if (!rc) rc = dev_upload(ctx, &u->out_off_d, out_off);
    if (!rc) rc = dev_upload(ctx, &u->in_src_d, in_src);
    if (!rc) rc = dev_upload(ctx, &u->in_col_d, in_col);
    if (!rc) rc = dev_upload(ctx, &u->in_tid_d, in_tid);
    if (!rc) rc = dev_upload(ctx, &u->in_olabel_d, in_ol);
    if (!rc) rc = dev_upload(ctx, &u->out_inidx_d, out_inidx);
    if (!rc) rc = dev_upload(ctx, &u->in_w_d, in_w);
    if (!rc) rc = dev_upload(ctx, &u->final_d, finalv);
    if (!rc) rc = dev_upload(ctx, &u->bp_off_d, u->bp_off);
    if (!rc) rc = dev_upload(ctx, &u->path_off_d, u->path_off);
    if (!rc) rc = dev_upload(ctx, &u->words_off_d, u->words_off);
    if (!rc) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e)); }
    if (rc) { khg_utts_destroy(u); return rc; }
  }
  plan_ll(u);
  { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) return fail(KHG_E_HIP, hipGetErrorString(e)); }
  *out = u;
  return KHG_OK;
}

extern "C" int khg_utts_set_pdf_list(khg_utts* u, int32_t n, const int32_t* pdfs) {
  if (!u || n <= 0 || !pdfs) return khg_set_error(KHG_E_ARG, "khg_utts_set_pdf_list: bad arguments");
  if (u->has_graphs) return khg_set_error(KHG_E_ARG, "khg_utts_set_pdf_list: set has graphs; its pdf lists come from them");
  u->pdfs.clear();
  u->pdf_first.clear(); u->pdf_last.clear();     // no graphs behind an explicit list: every frame is needed
  for (int i = 0; i < u->n_utt; ++i) { u->pdf_off[i + 1] = u->pdf_off[i] + n; u->pdfs.insert(u->pdfs.end(), pdfs, pdfs + n); }
  plan_ll(u);
  DEVFREE(u->pdf_off_d); DEVFREE(u->pdfs_d); DEVFREE(u->ll_off_d); DEVFREE(u->ll_d); DEVFREE(u->chunks_d); DEVFREE(u->wchunks_d);
  DEVFREE(u->tile_off_d); DEVFREE(u->tiles_d); u->tiles_pto.clear(); u->tiles_reach = -1;
  DEVFREE(u->tile2_off_d); DEVFREE(u->tiles2_d); u->tiles2_pto.clear(); u->tiles2_reach = -1;
  DEVFREE(u->p_ents_d); DEVFREE(u->p_slices_d); u->p_reach = -1;
  // the default K1's per-set unit table is indexed through pdf_off, and the id range check is cached per model size: both are stale now
  DEVFREE(u->sunits_d); u->sunits_pto.clear(); u->sunits_reach = -1;
  u->pdfs_checked_P = -1;
  u->ll_valid = false;
  return KHG_OK;
}

// Borrowed device features were rewritten in place: drop everything derived from them (column maxima, the fp16 / bf16 planes of the
// split K1 forms); the next khg_loglikes re-packs.  K3 and the fp32 K1 forms read feats_d live.
extern "C" int khg_utts_features_changed(khg_utts* u) {
  if (!u) return khg_set_error(KHG_E_ARG, "khg_utts_features_changed: bad arguments");
  u->xmax.clear();
  u->xs_ks = 0; u->xs_ex.clear();
  u->xh_ks = 0; u->xh_ex.clear();
  u->xb3_ks = 0;
  u->ll_valid = false;
  return KHG_OK;
}

extern "C" int khg_utts_destroy(khg_utts* u) {
  if (!u) return KHG_OK;
  if (u->own_feats) DEVFREE(u->feats_d);
  DEVFREE(u->frame_off_d); DEVFREE(u->state_off_d); DEVFREE(u->pdf_off_d); DEVFREE(u->ll_off_d);
  DEVFREE(u->pdfs_d); DEVFREE(u->wchunks_d); DEVFREE(u->start_d); DEVFREE(u->in_off_d); DEVFREE(u->out_off_d);
  DEVFREE(u->in_src_d); DEVFREE(u->in_col_d); DEVFREE(u->in_tid_d); DEVFREE(u->in_olabel_d); DEVFREE(u->out_inidx_d);
  DEVFREE(u->in_w_d); DEVFREE(u->final_d); DEVFREE(u->chunks_d); DEVFREE(u->ll_d); DEVFREE(u->tile_off_d); DEVFREE(u->tiles_d);
  DEVFREE(u->xpl_d); DEVFREE(u->utt_xtile_off_d); DEVFREE(u->p_ents_d); DEVFREE(u->p_slices_d);
  DEVFREE(u->xb3_d); DEVFREE(u->utt_x32_off_d); DEVFREE(u->bchunks_d); DEVFREE(u->x32_utt_d); DEVFREE(u->xh_d); DEVFREE(u->xh_ex_d); DEVFREE(u->tile2_off_d); DEVFREE(u->tiles2_d);
  DEVFREE(u->xs_d); DEVFREE(u->xs_ex_d); DEVFREE(u->schunks_d); DEVFREE(u->sunits_d);
  DEVFREE(u->bp_d); DEVFREE(u->bp_off_d); DEVFREE(u->path_off_d); DEVFREE(u->words_off_d);
  DEVFREE(u->layer_best_d); DEVFREE(u->layer_cnt_d); DEVFREE(u->path_d); DEVFREE(u->k2_gscratch_d); DEVFREE(u->k2_order_d);
  DEVFREE(u->ali_d); DEVFREE(u->words_d); DEVFREE(u->num_words_d); DEVFREE(u->status_d); DEVFREE(u->like_d);
  DEVFREE(u->pdf_count_d); DEVFREE(u->pdf_cursor_d); DEVFREE(u->frame_ids_d); DEVFREE(u->pdf_start_d); DEVFREE(u->tid_count_d);
  DEVFREE(u->sort_keys_d); DEVFREE(u->sort_keys_out_d); DEVFREE(u->sort_vals_d); DEVFREE(u->sort_tmp_d); DEVFREE(u->cs_hist_d); DEVFREE(u->cs_tot_d);
  DEVFREE(u->k3_part_d); DEVFREE(u->k3_llpart_d); DEVFREE(u->k3_items_d); DEVFREE(u->k3_item_off_d);
  if (u->ev_dp) (void)hipEventDestroy(u->ev_dp);
  if (u->ev_ali) (void)hipEventDestroy(u->ev_ali);
  delete u;
  return KHG_OK;
}
extern "C" int khg_utts_num_pdfs(const khg_utts* u, int64_t* pdf_off) {
  if (!u || !pdf_off) return khg_set_error(KHG_E_ARG, "bad arguments");
  std::copy(u->pdf_off.begin(), u->pdf_off.end(), pdf_off);
  return KHG_OK;
}
extern "C" int khg_utts_pdfs(const khg_utts* u, int32_t* pdfs) {
  if (!u || !pdfs) return khg_set_error(KHG_E_ARG, "bad arguments");
  std::copy(u->pdfs.begin(), u->pdfs.end(), pdfs);
  return KHG_OK;
}

// the main stream must not touch ali / status / the ll buffer while the side-stream decoder runs
static int wait_ali(khg_ctx* ctx, khg_utts* u) {
  if (u->ali_pending) { HIPCHK(hipStreamWaitEvent(ctx->stream, u->ev_ali, 0)); u->ali_pending = false; }
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
// K1
template <int KQ, int NF, int WPS>
static void launch_k1(const K1Args& a, int nchunks, bool aligned, hipStream_t s) {
  if (aligned) hipLaunchKernelGGL((k1_loglikes<KQ, NF, true, WPS>), dim3(nchunks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((k1_loglikes<KQ, NF, false, WPS>), dim3(nchunks), dim3(256), 0, s, a);
}
// 16-frame tiles per wave: 6 x 20 B-operand VGPRs fit 2 waves/SIMD at D <= 40 (KHG_K1_NF=5 selects the smaller chunk)
static int k1_nf(const khg_ctx* ctx, int KQ) { return KQ != 10 ? 5 : ctx->opt[KHG_OPT_K1_NF] == 5 ? 5 : 6; }

// K1 in pdf-major form: plan (entries grouped by pdf, cut into workgroup slices) + repacked features.
// D > 80: k1w_loglikes over 64-frame chunks (every cell of every listed pdf; khg_k1_wide.hip.inc)
static int loglikes_wide(khg_ctx* ctx, const khg_model* m, khg_utts* u) {
  if (!u->wchunks_d) {
    std::vector<KwChunk> ch;
    for (int i = 0; i < u->n_utt; ++i) {
      const int64_t T = u->frame_off[i + 1] - u->frame_off[i];
      if (T <= 0 || u->pdf_off[i + 1] == u->pdf_off[i]) continue;
      for (int64_t t0 = 0; t0 < T; t0 += 64) ch.push_back(KwChunk{i, (int32_t)t0, (int32_t)std::min<int64_t>(64, T - t0), 0});
    }
    u->n_wchunks = (int)ch.size();
    int rc = dev_upload(ctx, &u->wchunks_d, ch);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  KwArgs a;
  a.feats = u->feats_d; a.frame_off = u->frame_off_d; a.chunks = u->wchunks_d; a.pdf_off = u->pdf_off_d; a.pdfs = u->pdfs_d;
  a.gauss_off = m->gauss_off_d; a.gconsts = m->gconsts_d; a.means_invvars = m->miv_d; a.nhalf_inv_vars = m->nhiv_d;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.err_flag = ctx->err_flag_d; a.D = m->D;
  const size_t lds = sizeof(float) * 64 * (size_t)(m->D | 1);
  if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k1w_loglikes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (u->n_wchunks > 0) {
    KernelTimer kt(ctx, "k1_loglikes");
    hipLaunchKernelGGL(k1w_loglikes, dim3(u->n_wchunks), dim3(256), lds, ctx->stream, a);
  }
  HIPCHK(hipGetLastError());
  u->ll_valid = true;
  return KHG_OK;
}

static int loglikes_pdf_major(khg_ctx* ctx, const khg_model* m, khg_utts* u, bool reachable_only) {
  int rc = KHG_OK;
  const int KH = 2 * m->KQ;
  if (!u->xpl_d || u->xpl_kq != m->KQ) {
    // x tiles: ceil(T/16) per utterance, parity planes [2][16][KH] each
    DEVFREE(u->xpl_d); DEVFREE(u->utt_xtile_off_d);
    std::vector<int64_t> xoff((size_t)u->n_utt + 1, 0);
    for (int i = 0; i < u->n_utt; ++i) xoff[(size_t)i + 1] = xoff[(size_t)i] + (u->frame_off[i + 1] - u->frame_off[i] + 15) / 16;
    const int64_t nx = xoff[(size_t)u->n_utt];
    std::vector<int32_t> xutt((size_t)nx);
    for (int i = 0; i < u->n_utt; ++i)
      for (int64_t t = xoff[(size_t)i]; t < xoff[(size_t)i + 1]; ++t) xutt[(size_t)t] = i;
    int32_t* xutt_d = nullptr;
    rc = dev_upload(ctx, &u->utt_xtile_off_d, xoff);
    if (!rc) rc = dev_upload(ctx, &xutt_d, xutt);
    if (!rc) rc = dev_alloc(&u->xpl_d, (size_t)std::max<int64_t>(nx, 1) * 2 * 16 * KH);
    if (!rc && nx > 0) {
      const int gb = (int)std::min<int64_t>(65535, (nx * (2 * 16 * KH / 4) + 255) / 256);
      if (m->KQ == 10) hipLaunchKernelGGL(k1p_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_xtile_off_d, xutt_d, nx, u->D, u->xpl_d);
      else hipLaunchKernelGGL(k1p_pack_x<20>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_xtile_off_d, xutt_d, nx, u->D, u->xpl_d);
      hipError_t e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    }
    DEVFREE(xutt_d);
    if (rc) return rc;
    u->xpl_kq = m->KQ;
  }
  if (!u->p_ents_d || u->p_reach != (int)reachable_only || u->p_P != m->P || u->p_goff != m->gauss_off) {
    DEVFREE(u->p_ents_d); DEVFREE(u->p_slices_d);
    // entries grouped by pdf (counting sort keeps utterance order inside a pdf)
    std::vector<int64_t> cnt((size_t)m->P + 1, 0);
    for (int32_t p : u->pdfs) cnt[(size_t)p + 1]++;
    for (int p = 0; p < m->P; ++p) cnt[(size_t)p + 1] += cnt[(size_t)p];
    std::vector<K1pEntry> ents(u->pdfs.size());
    std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
    for (int i = 0; i < u->n_utt; ++i) {
      const int n16 = (int)((u->frame_off[i + 1] - u->frame_off[i] + 15) / 16);
      for (int64_t k = u->pdf_off[i]; k < u->pdf_off[i + 1]; ++k) {
        int ef = 0;
        if (reachable_only) ef = (int)std::min<int64_t>(n16, (int64_t)u->pdf_first[(size_t)k] / 16);
        ents[(size_t)cur[(size_t)u->pdfs[(size_t)k]]++] = K1pEntry{i, (int32_t)(k - u->pdf_off[i]), ef, n16 - ef};
      }
    }
    // slices: <= TS tiles and <= K1P_MAXENT entries of one pdf each
    const int TS = std::max(1, ctx->opt[KHG_OPT_K1P_TS]);
    std::vector<K1pSlice> slices;
    for (int p = 0; p < m->P; ++p) {
      int64_t e = cnt[(size_t)p];
      const int64_t e_end = cnt[(size_t)p + 1];
      int off = 0;                             // tiles of entry e already given out
      while (e < e_end) {
        while (e < e_end && ents[(size_t)e].nt - off <= 0) { ++e; off = 0; }
        if (e >= e_end) break;
        K1pSlice s{p, (int32_t)e, 0, off, 0};
        int64_t ee = e;
        int o = off;
        while (ee < e_end && s.ntiles < TS && s.nent < K1P_MAXENT) {
          const int avail = ents[(size_t)ee].nt - o;
          if (avail <= 0) { ++s.nent; ++ee; o = 0; continue; }
          const int take = std::min(avail, TS - s.ntiles);
          s.ntiles += take;
          ++s.nent;
          if (take == avail) { ++ee; o = 0; } else { o += take; break; }
        }
        if (s.ntiles > 0) slices.push_back(s);
        e = ee; off = o;
      }
    }
    // group the slices by the pdf's number of 16-Gaussian blocks: one launch (kernel instantiation) per count
    auto nblk_of = [&](const K1pSlice& s) { return (m->gauss_off[s.pdf + 1] - m->gauss_off[s.pdf] + 15) / 16; };
    std::stable_sort(slices.begin(), slices.end(), [&](const K1pSlice& x, const K1pSlice& y) { return nblk_of(x) < nblk_of(y); });
    for (int k = 0; k < 10; ++k) u->p_grp[k] = 0;
    for (const auto& s : slices) u->p_grp[nblk_of(s)]++;          // counts per block count 1..4
    if (ents.size() >= (size_t)INT32_MAX) return khg_set_error(KHG_E_UNSUPPORTED, "khg_loglikes: too many (utterance, pdf) entries");
    rc = dev_upload(ctx, &u->p_ents_d, ents);
    if (!rc) rc = dev_upload(ctx, &u->p_slices_d, slices);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u->p_nslices = (int32_t)slices.size();
    if (ctx->opt[KHG_OPT_DEBUG]) { long long tt = 0; for (auto& s : slices) tt += s.ntiles; fprintf(stderr, "[khg] pdf-major plan: %zu entries, %zu slices, %lld tiles\n", ents.size(), slices.size(), tt); }
    u->p_reach = (int)reachable_only;
    u->p_P = m->P;
    u->p_goff = m->gauss_off;
  }
  K1pArgs a;
  a.xpl = u->xpl_d; a.utt_xtile_off = u->utt_xtile_off_d; a.frame_off = u->frame_off_d;
  a.ents = u->p_ents_d; a.slices = u->p_slices_d; a.wimg = m->wimg_d; a.pdf_tile_off = m->pdf_tile_off_d;
  a.gauss_off = m->gauss_off_d; a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.err_flag = ctx->err_flag_d;
  if (u->p_nslices > 0) {
    KernelTimer kt(ctx, "k1_loglikes");
    int first = 0;
    for (int nb = 1; nb <= 8; ++nb) {
      const int n = u->p_grp[nb];
      if (n == 0) continue;
      a.slice0 = first;
      first += n;
#define K1P_LAUNCH(KQ_, NB_, WPS_) hipLaunchKernelGGL((k1p_loglikes<KQ_, NB_, WPS_>), dim3(n), dim3(256), 0, ctx->stream, a)
      if (m->KQ == 10) {
        switch (nb) { case 1: K1P_LAUNCH(10, 1, 2); break; case 2: K1P_LAUNCH(10, 2, 2); break; case 3: K1P_LAUNCH(10, 3, 2); break; case 4: K1P_LAUNCH(10, 4, 2); break;
                      case 5: K1P_LAUNCH(10, 5, 2); break; case 6: K1P_LAUNCH(10, 6, 2); break; case 7: K1P_LAUNCH(10, 7, 2); break; default: K1P_LAUNCH(10, 8, 2); break; }
      } else {
        switch (nb) { case 1: K1P_LAUNCH(20, 1, 1); break; case 2: K1P_LAUNCH(20, 2, 1); break; case 3: K1P_LAUNCH(20, 3, 1); break; case 4: K1P_LAUNCH(20, 4, 1); break;
                      case 5: K1P_LAUNCH(20, 5, 1); break; case 6: K1P_LAUNCH(20, 6, 1); break; case 7: K1P_LAUNCH(20, 7, 1); break; default: K1P_LAUNCH(20, 8, 1); break; }
      }
#undef K1P_LAUNCH
    }
    HIPCHK(hipGetLastError());
  }
  u->ll_valid = true;
  return KHG_OK;
}

// per-utterance W-tile walk for this model's tile layout (it only changes when the number of Gaussians of some pdf
// crosses a multiple of 32): for every pdf on the utterance's list its 32-Gaussian tiles in order.  Entry = tile id
// (bits 0-21) | first needed 16-frame tile of the pdf, clamped to 127 (bits 22-28; 0 unless reachable_only) |
// last-tile-of-pdf flag (bit 31).  Shared by the utterance-major fp32 kernel and the bf16x3 kernel.
static int ensure_walk(khg_ctx* ctx, const khg_model* m, khg_utts* u, bool reachable_only) {
  if (u->tiles_pto == m->pdf_tile_off && u->tiles_reach == (int)reachable_only) return KHG_OK;
  if (m->ntiles >= (1 << 22)) return khg_set_error(KHG_E_UNSUPPORTED, "khg_loglikes: more than 4M W tiles");
  DEVFREE(u->tile_off_d); DEVFREE(u->tiles_d);
  std::vector<int64_t> toff((size_t)u->n_utt + 1, 0);
  std::vector<int32_t> tiles;
  for (int i = 0; i < u->n_utt; ++i) {
    for (int64_t k = u->pdf_off[i]; k < u->pdf_off[i + 1]; ++k) {
      const int p = u->pdfs[(size_t)k];
      int ef = 0;
      if (reachable_only) ef = (int)std::min<int64_t>(127, (int64_t)u->pdf_first[(size_t)k] / 16);
      for (int t = m->pdf_tile_off[p]; t < m->pdf_tile_off[p + 1]; ++t)
        tiles.push_back(t | (ef << 22) | (t + 1 == m->pdf_tile_off[p + 1] ? (int32_t)0x80000000 : 0));
    }
    toff[(size_t)i + 1] = (int64_t)tiles.size();
  }
  int rc = dev_upload(ctx, &u->tile_off_d, toff);
  if (!rc) rc = dev_upload(ctx, &u->tiles_d, tiles);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  u->tiles_pto = m->pdf_tile_off;
  u->tiles_reach = (int)reachable_only;
  return KHG_OK;
}

// 32-frame tile layout of the set shared by the bf16x3 / f16x2 / f16x2s forms: tile offsets per utterance, tile -> utterance.
static int ensure_x32_layout(khg_ctx* ctx, khg_utts* u) {
  if (u->utt_x32_off_d) return KHG_OK;
  std::vector<int64_t> xoff((size_t)u->n_utt + 1, 0);
  for (int i = 0; i < u->n_utt; ++i) xoff[(size_t)i + 1] = xoff[(size_t)i] + (u->frame_off[i + 1] - u->frame_off[i] + 31) / 32;
  const int64_t nx = xoff[(size_t)u->n_utt];
  std::vector<int32_t> xutt((size_t)nx);
  for (int i = 0; i < u->n_utt; ++i)
    for (int64_t t = xoff[(size_t)i]; t < xoff[(size_t)i + 1]; ++t) xutt[(size_t)t] = i;
  int rc = dev_upload(ctx, &u->utt_x32_off_d, xoff);
  if (!rc) rc = dev_upload(ctx, &u->x32_utt_d, xutt);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  u->n_x32 = nx;
  return KHG_OK;
}
// Workgroup chunks of <= `per` 32-frame tiles: an utterance is cut into equal parts (a 17-tile utterance becomes 9 + 8 tiles, not
// 16 + 1), and the chunks are launched longest first (duration ~ frame tiles x pdfs): the workgroups still running when the grid
// drains are then the short ones (a launch of 12 500 utterances -- the 8-GPU shard -- is ~49 rounds of workgroups whose durations
// differ 4x).  K1bChunk and K1sChunk have the same layout.
template <class Chunk>
static void plan_x32_chunks(const khg_utts* u, int per, int order, std::vector<Chunk>* ch) {
  ch->clear();
  for (int i = 0; i < u->n_utt; ++i) {
    const int n32 = (int)((u->frame_off[i + 1] - u->frame_off[i] + 31) / 32);
    if (u->pdf_off[i + 1] == u->pdf_off[i]) continue;
    const int nch = (n32 + per - 1) / per;
    for (int c = 0; c < nch; ++c) {
      const int t0 = (int)((int64_t)n32 * c / nch), t1 = (int)((int64_t)n32 * (c + 1) / nch);
      if (t1 > t0) ch->push_back(Chunk{i, t0, t1 - t0, 0});
    }
  }
  // KHG_OPT_K1_ORDER (experiments): 0 frame tiles x pdfs descending (default), 1 utterance order, 2 ascending, 3 frame tiles descending
  auto cost = [&](const Chunk& c) { return (int64_t)c.ntiles * (order == 3 ? 1 : (u->pdf_off[c.utt + 1] - u->pdf_off[c.utt])); };
  if (order == 2) std::stable_sort(ch->begin(), ch->end(), [&](const Chunk& a, const Chunk& b) { return cost(a) < cost(b); });
  else if (order != 1) std::stable_sort(ch->begin(), ch->end(), [&](const Chunk& a, const Chunk& b) { return cost(a) > cost(b); });
}
static int ensure_x32(khg_ctx* ctx, khg_utts* u, int NTMAX) {
  int rc = ensure_x32_layout(ctx, u);
  if (rc) return rc;
  if (u->bchunks_d && u->bchunk_nt == NTMAX) return KHG_OK;
  DEVFREE(u->bchunks_d);
  std::vector<K1bChunk> ch;
  plan_x32_chunks(u, 8 * NTMAX, ctx->opt[KHG_OPT_K1_ORDER], &ch);
  rc = dev_upload(ctx, &u->bchunks_d, ch);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  u->n_bchunks = (int32_t)ch.size(); u->bchunk_nt = NTMAX;
  return KHG_OK;
}

// K1 on the bf16 matrix cores (khg_k1_bf16x3.hip.inc): B fragments of the features (once per set), chunks, walk.
static int loglikes_bf16x3(khg_ctx* ctx, const khg_model* mc, khg_utts* u, bool reachable_only) {
  khg_model* m = const_cast<khg_model*>(mc);
  const int KS = m->KS, NTMAX = KS == 5 ? 2 : 1;
  int rc = ensure_x32(ctx, u, NTMAX);
  if (rc) return rc;
  if (!m->wimgb_valid) {          // the bf16x3 image of the current parameters
    if (!m->wimgb_d || m->wimgb_tiles < m->ntiles) {
      DEVFREE(m->wimgb_d);
      rc = dev_alloc(&m->wimgb_d, (size_t)m->ntiles * k1b_tile_bytes(KS));
      if (rc) return rc;
      m->wimgb_tiles = m->ntiles;
    }
    rc = m->wimgb_sync.before_pack(ctx->stream);
    if (rc) return rc;
    KernelTimer kt(ctx, "k0b_pack_tiles");
    if (KS == 5) hipLaunchKernelGGL(k0b_pack_tiles<5>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, m->D, m->wimgb_d, ctx->err_flag_d);
    else hipLaunchKernelGGL(k0b_pack_tiles<10>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, m->D, m->wimgb_d, ctx->err_flag_d);
    HIPCHK(hipGetLastError());
    rc = m->wimgb_sync.after_pack(ctx->stream);
    if (rc) return rc;
    m->wimgb_valid = true;
  }
  if (!u->xb3_d || u->xb3_ks != KS) {
    DEVFREE(u->xb3_d);
    const int64_t nx = u->n_x32;
    rc = dev_alloc(&u->xb3_d, (size_t)std::max<int64_t>(nx, 1) * 3 * KS * 64);
    if (!rc && nx > 0) {
      const int gb = (int)std::min<int64_t>(65535, (nx * KS * 64 + 255) / 256);
      if (KS == 5) hipLaunchKernelGGL(k1b_pack_x<5>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, u->D, u->xb3_d);
      else hipLaunchKernelGGL(k1b_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, u->D, u->xb3_d);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    }
    if (rc) return rc;
    u->xb3_ks = KS;
  }
  rc = ensure_walk(ctx, m, u, reachable_only);
  if (rc) return rc;
  K1bArgs a;
  a.xb = u->xb3_d; a.utt_xtile_off = u->utt_x32_off_d; a.frame_off = u->frame_off_d; a.chunks = u->bchunks_d;
  a.wimg = m->wimgb_d; a.utt_tile_off = u->tile_off_d; a.utt_tiles = u->tiles_d;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.err_flag = ctx->err_flag_d;
  a.dbg = ctx->opt[KHG_OPT_K1_DBG];
  if (u->n_bchunks > 0) {
    const size_t lds = (size_t)k1b_ring(KS) * k1b_group(KS) * k1b_tile_bytes(KS);
    const void* fn = KS == 5 ? (const void*)k1b_loglikes<5, 2> : (const void*)k1b_loglikes<10, 1>;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    rc = m->wimgb_sync.before_read(ctx->stream);
    if (rc) return rc;
    {
      KernelTimer kt(ctx, "k1_loglikes");
      if (KS == 5) hipLaunchKernelGGL((k1b_loglikes<5, 2>), dim3(u->n_bchunks), dim3(512), lds, ctx->stream, a);
      else hipLaunchKernelGGL((k1b_loglikes<10, 1>), dim3(u->n_bchunks), dim3(512), lds, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    rc = m->wimgb_sync.after_read(ctx->stream);
    if (rc) return rc;
  }
  u->ll_valid = true;
  return KHG_OK;
}

// column maxima of |a[n][D]| -> host
static int absmax_cols(khg_ctx* ctx, const float* a_d, int64_t n, int D, std::vector<float>* out) {
  uint32_t* m_d = nullptr;
  int rc = dev_alloc(&m_d, 128);
  if (rc) return rc;
  std::vector<uint32_t> h(128, 0);
  hipError_t e = hipMemsetAsync(m_d, 0, 128 * sizeof(uint32_t), ctx->stream);
  if (e == hipSuccess && n > 0) {
    const int gb = (int)std::min<int64_t>(4096, (n + 1) / 2);
    KernelTimer kt(ctx, "k1_absmax");
    hipLaunchKernelGGL(k1h_absmax, dim3(gb), dim3(256), 0, ctx->stream, a_d, n, D, m_d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(h.data(), m_d, 128 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  DEVFREE(m_d);
  if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  out->resize((size_t)D);
  for (int d = 0; d < D; ++d) memcpy(&(*out)[(size_t)d], &h[(size_t)d], sizeof(float));
  return KHG_OK;
}

// f16x2 scale exponents (khg_k1_f16x2.hip.inc): per k = 2 d + kind, x' = x 2^e and w' = w 2^-e.  `fits`: every scaled
// operand stays <= 2^15 (fp16 overflows at 65504).
static const float K1H_LIMIT = 32768.0f;
static bool k1h_fits(const std::vector<int32_t>& ex, const std::vector<float>& xk, const std::vector<float>& wk) {
  for (size_t k = 0; k < ex.size(); ++k) {
    if (!(std::ldexp(xk[k], ex[k]) <= K1H_LIMIT) || !(std::ldexp(wk[k], -ex[k]) <= K1H_LIMIT)) return false;
  }
  return true;
}
static void k1h_balance(const std::vector<float>& xk, const std::vector<float>& wk, std::vector<int32_t>* ex) {
  ex->assign(xk.size(), 0);
  for (size_t k = 0; k < xk.size(); ++k) {
    const bool hx = xk[k] > 0.0f && std::isfinite(xk[k]), hw = wk[k] > 0.0f && std::isfinite(wk[k]);
    double e = 0.0;
    if (hx && hw) e = 0.5 * (std::log2((double)wk[k]) - std::log2((double)xk[k]));
    else if (hx) e = -std::log2((double)xk[k]);      // only one side has values: bring its maximum to ~1
    else if (hw) e = std::log2((double)wk[k]);
    (*ex)[k] = (int32_t)std::lrint(std::min(120.0, std::max(-120.0, e)));
  }
}

// Exact maxima behind the split forms' domain check and the f16x2 scales: per feature dimension over the set (once), per
// k = 2 d + kind and over the gconsts for the model (once per parameter version).  -> xk[k] = max |X[k][.]|.
static int k1_maxima(khg_ctx* ctx, khg_model* m, khg_utts* u, std::vector<float>* xk) {
  const int D = m->D, K = 16 * m->KS;
  int rc = KHG_OK;
  if (u->xmax.empty()) { rc = absmax_cols(ctx, u->feats_d, u->N, D, &u->xmax); if (rc) return rc; }
  if (m->wmax.empty()) {
    std::vector<float> a, b, g;
    rc = absmax_cols(ctx, m->miv_d, m->sumG, D, &a);
    if (!rc) rc = absmax_cols(ctx, m->iv_d, m->sumG, D, &b);
    if (!rc) rc = absmax_cols(ctx, m->gconsts_d, m->sumG, 1, &g);
    if (rc) return rc;
    m->wmax.assign((size_t)K, 0.0f);
    for (int d = 0; d < D; ++d) { m->wmax[(size_t)2 * d] = a[(size_t)d]; m->wmax[(size_t)2 * d + 1] = 0.5f * b[(size_t)d]; }
    m->gcmax = g[0];
  }
  xk->assign((size_t)K, 0.0f);
  for (int d = 0; d < D; ++d) { (*xk)[(size_t)2 * d] = u->xmax[(size_t)d]; (*xk)[(size_t)2 * d + 1] = u->xmax[(size_t)d] * u->xmax[(size_t)d]; }
  return KHG_OK;
}
// The split forms (f16x2, bf16x3) fold the log-sum-exp's subtraction into an fma (k1_exp2_le1): valid while every
// log-likelihood term sum stays below 2^28 in magnitude (khg_k1_f16x2.hip.inc, "Domain").
static bool k1_split_domain(const khg_model* m, const std::vector<float>& xk) {
  double bound = (double)m->gcmax;
  for (size_t k = 0; k < xk.size(); ++k) bound += (double)m->wmax[k] * (double)xk[k];
  return bound <= 268435456.0;      // also false for NaN / inf
}

// K1 on the fp16 matrix cores (khg_k1_f16x2.hip.inc).  -> KHG_OK, an error, or +1: outside the split forms' domain (the
// caller runs an fp32-MFMA form).
#ifndef K1H_NT10
#define K1H_NT10 2      // 32-frame tiles per wave at D <= 80 (12 spilled registers; 1: none)
#endif
static int loglikes_f16x2(khg_ctx* ctx, khg_model* m, khg_utts* u, bool reachable_only) {
  const int KS = m->KS, NTMAX = KS == 5 ? 2 : K1H_NT10, D = m->D, K = 16 * KS;
  std::vector<float> xk;
  int rc = k1_maxima(ctx, m, u, &xk);
  if (rc) return rc;
  if (!k1_split_domain(m, xk)) return 1;
  rc = ensure_x32(ctx, u, NTMAX);
  if (rc) return rc;
  // the set's planes are kept while the model still fits their scales
  const bool have_x = u->xh_d && u->xh_ks == KS && (int)u->xh_ex.size() == K;
  if (!have_x || !k1h_fits(u->xh_ex, xk, m->wmax)) {
    std::vector<int32_t> ex;
    k1h_balance(xk, m->wmax, &ex);
    if (!k1h_fits(ex, xk, m->wmax)) return 1;
    const int64_t nx = u->n_x32;
    if (!u->xh_d || u->xh_ks != KS) {
      DEVFREE(u->xh_d);
      rc = dev_alloc(&u->xh_d, (size_t)std::max<int64_t>(nx, 1) * 2 * KS * 64);
      if (rc) return rc;
    }
    DEVFREE(u->xh_ex_d);
    rc = dev_upload(ctx, &u->xh_ex_d, ex);
    if (rc) return rc;
    if (nx > 0) {
      KernelTimer kt(ctx, "k1h_pack_x");
      const int gb = (int)std::min<int64_t>(65535, (nx * KS * 64 + 255) / 256);
      if (KS == 5) hipLaunchKernelGGL(k1h_pack_x<5>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xh_ex_d, u->xh_d);
      else hipLaunchKernelGGL(k1h_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xh_ex_d, u->xh_d);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));    // `ex` (pageable) is free after this
    u->xh_ks = KS; u->xh_ex = ex;
  }
  if (m->wimgh_ex != u->xh_ex) {
    if (!m->wimgh_d || m->wimgh_tiles < m->ntiles) {
      DEVFREE(m->wimgh_d);
      rc = dev_alloc(&m->wimgh_d, (size_t)m->ntiles * k1h_tile_bytes(KS));
      if (rc) return rc;
      m->wimgh_tiles = m->ntiles;
    }
    rc = m->wimgh_sync.before_pack(ctx->stream);
    if (rc) return rc;
    KernelTimer kt(ctx, "k0h_pack_tiles");
    if (KS == 5) hipLaunchKernelGGL(k0h_pack_tiles<5>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, u->xh_ex_d, m->wimgh_d);
    else hipLaunchKernelGGL(k0h_pack_tiles<10>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, u->xh_ex_d, m->wimgh_d);
    HIPCHK(hipGetLastError());
    rc = m->wimgh_sync.after_pack(ctx->stream);
    if (rc) return rc;
    m->wimgh_ex = u->xh_ex;
  }
  rc = ensure_walk(ctx, m, u, reachable_only);
  if (rc) return rc;
  K1hArgs a;
  a.xh = u->xh_d; a.utt_xtile_off = u->utt_x32_off_d; a.frame_off = u->frame_off_d; a.chunks = u->bchunks_d;
  a.wimg = m->wimgh_d; a.utt_tile_off = u->tile_off_d; a.utt_tiles = u->tiles_d;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.err_flag = ctx->err_flag_d;
  a.dbg = ctx->opt[KHG_OPT_K1_DBG];
  a.tbuf = nullptr;
#ifdef K1H_TIMING
  uint64_t* tbuf_d = nullptr;
  if (u->n_bchunks > 0) {
    rc = dev_alloc(&tbuf_d, (size_t)u->n_bchunks * 64);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(tbuf_d, 0, (size_t)u->n_bchunks * 64 * sizeof(uint64_t), ctx->stream));
    a.tbuf = tbuf_d;
  }
#endif
  if (u->n_bchunks > 0) {
    const size_t lds = (size_t)k1h_lds_bytes(KS);
    const void* fn = KS == 5 ? (const void*)k1h_loglikes<5, 2> : (const void*)k1h_loglikes<10, K1H_NT10>;
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    rc = m->wimgh_sync.before_read(ctx->stream);
    if (rc) return rc;
    {
      KernelTimer kt(ctx, "k1_loglikes");
      if (KS == 5) hipLaunchKernelGGL((k1h_loglikes<5, 2>), dim3(u->n_bchunks), dim3(512), lds, ctx->stream, a);
      else hipLaunchKernelGGL((k1h_loglikes<10, K1H_NT10>), dim3(u->n_bchunks), dim3(512), lds, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    rc = m->wimgh_sync.after_read(ctx->stream);
    if (rc) return rc;
  }
#ifdef K1H_TIMING
  if (tbuf_d) {      // measurement build only: per-wave cycle breakdown, averaged by the number of frame tiles the wave owns
    std::vector<uint64_t> h((size_t)u->n_bchunks * 64);
    HIPCHK(hipMemcpyAsync(h.data(), tbuf_d, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    double sum[3][8] = {};
    long cnt[3] = {0, 0, 0};
    for (size_t w = 0; w < h.size() / 8; ++w) {
      const uint64_t* o = &h[w * 8];
      if (o[5] == 0 || o[6] > 2) continue;
      for (int i = 0; i < 8; ++i) sum[o[6]][i] += (double)o[i];
      ++cnt[o[6]];
    }
    for (int nt = 0; nt < 3; ++nt)
      if (cnt[nt]) fprintf(stderr, "k1h timing NT=%d waves=%ld: per wave cycles(100MHz-or-core clock units): barrier %.0f frag-wait %.0f chains %.0f tail %.0f total %.0f; tiles %.1f intervals %.1f\n",
                           nt, cnt[nt], sum[nt][0] / cnt[nt], sum[nt][1] / cnt[nt], sum[nt][2] / cnt[nt], sum[nt][3] / cnt[nt], sum[nt][5] / cnt[nt], sum[nt][4] / cnt[nt], sum[nt][7] / cnt[nt]);
    DEVFREE(tbuf_d);
  }
#endif
  u->ll_valid = true;
  return KHG_OK;
}

// K1 on the fp16 matrix cores, one accumulator per chain, transposed decomposition (khg_k1_f16x2s.hip.inc; the default).
// -> KHG_OK, an error, or +1: outside this form's domain (the caller tries the two-accumulator f16x2 form next).
static int loglikes_f16x2s(khg_ctx* ctx, khg_model* m, khg_utts* u, int reach) {
  const bool reachable_only = reach != 0;
  const int KS = m->KS, D = m->D, K = 16 * KS, NMAX = k1s_nmax(KS);
  std::vector<float> xk;
  int rc = k1_maxima(ctx, m, u, &xk);
  if (rc) return rc;
  if (!k1_split_domain(m, xk)) return 1;
  // exponents: every feature column peaks in [2^14, 2^15) (the set's own property), the largest weight column too (S)
  std::vector<int32_t> ex((size_t)K, 0), ew((size_t)K, 0);
  for (int k = 0; k < K; ++k) if (xk[(size_t)k] > 0.0f) ex[(size_t)k] = 14 - std::ilogb(xk[(size_t)k]);
  // Several utterance sets score against one model (batches of a shard, two contexts): the image is keyed by the feature
  // exponents, so per-set exponents would re-pack the 100 MB image on every alternating call.  The model keeps the element-wise
  // minimum of the exponents of the sets it has scored (a smaller exponent never overflows fp16; the absolute part of the error
  // bound is re-checked below for the exponents actually used) and every set packs its planes with those.
  if (m->xs_ex_seen.size() == ex.size()) {
    for (int k = 0; k < K; ++k) ex[(size_t)k] = std::min(ex[(size_t)k], m->xs_ex_seen[(size_t)k]);
  }
  m->xs_ex_seen = ex;
  int S = INT_MAX;
  for (int k = 0; k < K; ++k) if (m->wmax[(size_t)k] > 0.0f) S = std::min(S, 14 - std::ilogb(m->wmax[(size_t)k]) + ex[(size_t)k]);
  if (S == INT_MAX) S = 0;
  if (S < -100 || S > 100) return 1;
  double floor_sum = 0.0;      // the absolute part of the error bound, at the column maxima (khg_k1_f16x2s.hip.inc)
  for (int k = 0; k < K; ++k) {
    ew[(size_t)k] = S - ex[(size_t)k];
    floor_sum += std::ldexp((double)m->wmax[(size_t)k], ew[(size_t)k]) + std::ldexp((double)xk[(size_t)k], ex[(size_t)k]);
  }
  floor_sum = std::ldexp(floor_sum, -25 - S);
  if (!(floor_sum <= 2.0e-6)) return 1;
  rc = ensure_x32_layout(ctx, u);
  if (rc) return rc;
  if (!u->schunks_d || u->schunk_nmax != NMAX) {
    DEVFREE(u->schunks_d);
    std::vector<K1sChunk> ch;
    plan_x32_chunks(u, NMAX, ctx->opt[KHG_OPT_K1_ORDER], &ch);
    rc = dev_upload(ctx, &u->schunks_d, ch);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u->n_schunks = (int32_t)ch.size(); u->schunk_nmax = NMAX;
  }
  // the set's B fragments: packed once (the exponents depend on the features alone)
  if (!u->xs_d || u->xs_ks != KS || u->xs_ex != ex) {
    const int64_t nx = u->n_x32;
    if (!u->xs_d || u->xs_ks != KS) {
      DEVFREE(u->xs_d);
      rc = dev_alloc(&u->xs_d, (size_t)std::max<int64_t>(nx, 1) * 2 * KS * 64);
      if (rc) return rc;
    }
    DEVFREE(u->xs_ex_d);
    rc = dev_upload(ctx, &u->xs_ex_d, ex);
    if (rc) return rc;
    if (nx > 0) {
      KernelTimer kt(ctx, "k1s_pack_x");
      const int gb = (int)std::min<int64_t>(65535, (nx * KS * 64 + 255) / 256);
      if (KS == 5) hipLaunchKernelGGL(k1s_pack_x<5>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xs_ex_d, u->xs_d);
      else hipLaunchKernelGGL(k1s_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xs_ex_d, u->xs_d);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));    // `ex` (pageable) is free after this
    u->xs_ks = KS; u->xs_ex = ex;
  }
  // the model's image for these exponents
  std::vector<int32_t> key(ex);
  key.push_back(S);
  if (m->wimgs_key != key) {
    rc = m->wimgs_sync.before_pack(ctx->stream);
    if (rc) return rc;
    if (!m->wimgs_d || m->wimgs_tiles < m->ntiles) {
      HIPCHK(hipStreamSynchronize(ctx->stream));     // (the waits on other streams' readers were enqueued above)
      DEVFREE(m->wimgs_d);
      rc = dev_alloc(&m->wimgs_d, (size_t)m->ntiles * k1s_tile_bytes(KS));
      if (rc) return rc;
      m->wimgs_tiles = m->ntiles;
    }
    int32_t* ew_d = nullptr;
    rc = dev_upload(ctx, &ew_d, ew);
    if (rc) return rc;
    const float gscale = std::ldexp(1.0f, S);
    {
      KernelTimer kt(ctx, "k0s_pack_tiles");
      if (KS == 5) hipLaunchKernelGGL(k0s_pack_tiles<5>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, ew_d, gscale, m->wimgs_d);
      else hipLaunchKernelGGL(k0s_pack_tiles<10>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, ew_d, gscale, m->wimgs_d);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);     // `ew` (pageable) and ew_d are free after this
    DEVFREE(ew_d);
    if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    rc = m->wimgs_sync.after_pack(ctx->stream);
    if (rc) return rc;
    m->wimgs_key = key;
  }
  // models of small pdfs (all <= 8 / <= 16 Gaussians, D <= 40): 4 / 2 pdfs share one MFMA tile (k1s_loglikes_packed)
  int maxG = 0;
  for (int p = 0; p < m->P; ++p) maxG = std::max(maxG, m->gauss_off[p + 1] - m->gauss_off[p]);
  const int pack = (KS != 5 || (ctx->opt[KHG_OPT_K1_DBG] & 16)) ? 1 : maxG <= 8 ? 4 : maxG <= 16 ? 2 : 1;
  const bool band = reach == 2 && pack == 1 && u->pdf_last.size() == u->pdfs.size();     // (the packed kernel keeps the front-only form)
  // one unit per (utterance, listed pdf): first W tile, number of W tiles, first (and, BAND form, last) needed 32-frame tile
  // (KHG_K1B_DBG bits 32 / 64, A/B only: no shifted tiles / only the 16-frame shift)
  const int shift_mode = (ctx->opt[KHG_OPT_K1_DBG] & 32) ? 0 : (ctx->opt[KHG_OPT_K1_DBG] & 64) ? 1 : 2;
  // khg_loglikes_reachable (no band): a pdf's tiles run to the utterance's end, which is a band whose last frame is T - 1 -- the same
  // shifted tiles save the same tile (every second pdf); the kernel takes its band path, the fill only meets padding frames
  const bool tail_shift = !band && reachable_only && pack == 1 && shift_mode != 0 && u->pdf_first.size() == u->pdfs.size();
  const int units_key = (band ? 2 : (int)reachable_only) + 4 * shift_mode;
  if (u->sunits_pto != m->pdf_tile_off || u->sunits_reach != units_key) {
    DEVFREE(u->sunits_d);
    std::vector<K1sUnit> units(u->pdfs.size());
    std::vector<int32_t> unit_T;                     // tail_shift: the utterance length of every unit
    if (tail_shift) {
      unit_T.resize(u->pdfs.size());
      for (int i = 0; i < u->n_utt; ++i)
        for (int64_t k = u->pdf_off[(size_t)i]; k < u->pdf_off[(size_t)i + 1]; ++k) unit_T[(size_t)k] = (int32_t)std::min<int64_t>(INT32_MAX, u->frame_off[(size_t)i + 1] - u->frame_off[(size_t)i]);
    }
    for (size_t k = 0; k < u->pdfs.size(); ++k) {
      const int p = u->pdfs[k];
      const int nt = m->pdf_tile_off[p + 1] - m->pdf_tile_off[p];
      if (nt > (int)K1S_NT_MASK) return khg_set_error(KHG_E_UNSUPPORTED, "khg_loglikes: a pdf of more than 65 504 Gaussians");
      const uint32_t need = reachable_only ? (uint32_t)std::min<int64_t>(255, (int64_t)u->pdf_first[k] / 32) : 0u;
      // last needed tile: 255 = no limit (also a pdf no accepting path reads, last = -1: tile 0 ... nothing past it is computed
      // only when last >= 0; a never-needed pdf keeps last tile 0 so that the kernel's [first, last] range is at most one tile)
      uint32_t last = 255u, shift = 0u;
      if (band || tail_shift) {
        const int32_t pl = band ? u->pdf_last[k] : unit_T[k] - 1;
        last = pl < 0 ? 0u : (uint32_t)std::min<int32_t>(255, pl / 32);
        // the band ends earlier inside its tile than it starts: tiles that start at its first frame cover it with one tile fewer
        if (pl >= 0 && need < 255u && last < 255u && last > need && (pl % 32) < (u->pdf_first[k] % 32)) shift = (uint32_t)(u->pdf_first[k] % 32);
        if (shift_mode == 0) shift = 0u;
        else if (shift_mode == 1) shift = (shift >= 16u && (pl % 32) < 16) ? 16u : 0u;
      }
      units[k] = K1sUnit{m->pdf_tile_off[p], (uint32_t)nt | (shift << 11) | (need << 16) | (last << 24)};
    }
    rc = dev_upload(ctx, &u->sunits_d, units);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u->sunits_pto = m->pdf_tile_off;
    u->sunits_reach = units_key;
  }
  // BAND form: the per-pdf upper bounds the skipped tiles are filled with, indexed by a pdf's first W tile; per parameter version
  if ((band || tail_shift) && !m->ubound_valid) {
    if (!m->ubound_d || m->ubound_tiles < m->ntiles) {
      DEVFREE(m->ubound_d);
      rc = dev_alloc(&m->ubound_d, (size_t)m->ntiles);
      if (rc) return rc;
      m->ubound_tiles = m->ntiles;
    }
    hipLaunchKernelGGL(k1s_ubound, dim3(m->P), dim3(64), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, D, m->ubound_d);
    HIPCHK(hipGetLastError());
    m->ubound_valid = true;
  }
  K1sArgs a;
  a.xs = u->xs_d; a.utt_xtile_off = u->utt_x32_off_d; a.frame_off = u->frame_off_d; a.chunks = u->schunks_d;
  a.wimg = m->wimgs_d; a.pdf_off = u->pdf_off_d; a.units = u->sunits_d;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.dump = ctx->dump_d; a.err_flag = ctx->err_flag_d;
  a.c1 = std::ldexp(1.44269504088896340736f, -S);
  a.inv_scale = std::ldexp(1.0f, -S);
  a.mfloor = -3.0e38f / std::max(1.0f, a.c1);
  a.ubound = (band || tail_shift) ? m->ubound_d : nullptr; a.repair_status = nullptr; a.repair_bit = 0;
  u->ll_mode = band ? 2 : (reachable_only ? 1 : 0);
  if (u->n_schunks > 0) {
    rc = m->wimgs_sync.before_read(ctx->stream);
    if (rc) return rc;
    const size_t lds = (size_t)NMAX * k1s_xtile_bytes(KS) + 64;     // + the work-item counter
    if (band) { u->band_args = a; u->band_model = m; u->band_ks = KS; u->band_lds = lds; }
    const void* fn = pack == 4 ? (const void*)k1s_loglikes_packed<5, 4> : pack == 2 ? (const void*)k1s_loglikes_packed<5, 2>
                     : KS == 5 ? (const void*)k1s_loglikes<5> : (const void*)k1s_loglikes<10>;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
      KernelTimer kt(ctx, "k1_loglikes");
      if (pack == 4) hipLaunchKernelGGL((k1s_loglikes_packed<5, 4>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
      else if (pack == 2) hipLaunchKernelGGL((k1s_loglikes_packed<5, 2>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
      else if (KS == 5) hipLaunchKernelGGL((k1s_loglikes<5>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
      else hipLaunchKernelGGL((k1s_loglikes<10>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    rc = m->wimgs_sync.after_read(ctx->stream);
    if (rc) return rc;
  }
  u->ll_valid = true;
  return KHG_OK;
}

static int loglikes_impl(khg_ctx* ctx, const khg_model* m, khg_utts* u, int reach) {
  if (!ctx || !m || !u) return khg_set_error(KHG_E_ARG, "khg_loglikes: bad arguments");
  if (u->pdf_first.size() != u->pdfs.size()) reach = 0;
  const bool reachable_only = reach != 0;
  u->ll_mode = reachable_only ? 1 : 0;
  if (m->D != u->D) return khg_set_error(KHG_E_RUNTIME, "Dim mismatch: data dim = " + std::to_string(u->D) + " vs. model dim = " + std::to_string(m->D));
  if (u->pdfs_checked_P != m->P) {     // once per (set, model size): 7 M entries at the bench size, 1.5 ms of host time per call
    for (int32_t p : u->pdfs)
      if (p < 0 || p >= m->P) return khg_set_error(KHG_E_RUNTIME, "Likely graph/model mismatch, e.g. using wrong HCLG.fst (pdf-id " + std::to_string(p) + ")");
    u->pdfs_checked_P = m->P;
  }
  int rc = wait_ali(ctx, u);
  if (rc) return rc;
  if (!u->pdf_off_d) {
    rc = dev_upload(ctx, &u->pdf_off_d, u->pdf_off);
    if (!rc) rc = dev_upload(ctx, &u->pdfs_d, u->pdfs);
    if (!rc) rc = dev_upload(ctx, &u->ll_off_d, u->ll_off);
    if (!rc) rc = dev_alloc(&u->ll_d, (size_t)u->ll_total);
    if (rc) return rc;
  }
  {
    // which K1: bf16x3 (default: the bf16 matrix cores at fp32 accuracy), or one of the fp32-MFMA forms -- pdf-major (pdfs of
    // <= 128 Gaussians) / utterance-major -- whose per-Gaussian fmaf chain is pinned bit for bit by the tests
    int form = ctx->opt[KHG_OPT_K1_FORM];
    if (form == KHG_K1_AUTO) form = KHG_K1_F16X2S;
    if (u->N == 0 || u->pdfs.empty()) { u->ll_valid = true; return KHG_OK; }
    if (m->KQ == 0) return loglikes_wide(ctx, m, u);      // D > 80: one form
    if (form == KHG_K1_F16X2S) {
      rc = loglikes_f16x2s(ctx, const_cast<khg_model*>(m), u, reach);
      if (rc <= 0) return rc;
      form = KHG_K1_F16X2;           // the absolute part of its error bound is too large for this model: two accumulators
    }
    if (form == KHG_K1_F16X2) {
      rc = loglikes_f16x2(ctx, const_cast<khg_model*>(m), u, reachable_only);
      if (rc <= 0) return rc;
      form = KHG_K1_FP32_PDF;        // magnitudes outside the split forms' domain
    }
    if (form == KHG_K1_BF16X3) {
      std::vector<float> xk;
      rc = k1_maxima(ctx, const_cast<khg_model*>(m), u, &xk);
      if (rc) return rc;
      if (k1_split_domain(m, xk)) return loglikes_bf16x3(ctx, m, u, reachable_only);
      form = KHG_K1_FP32_PDF;
    }
    int maxG = 0;
    for (int p = 0; p < m->P; ++p) maxG = std::max(maxG, m->gauss_off[p + 1] - m->gauss_off[p]);
    if (form == KHG_K1_FP32_PDF && maxG <= 128) return loglikes_pdf_major(ctx, m, u, reachable_only);
  }
  if (!u->chunks_d || u->chunk_kq != m->KQ * 16 + k1_nf(ctx, m->KQ)) {
    DEVFREE(u->chunks_d);
    const int maxtiles = 4 * k1_nf(ctx, m->KQ);
    std::vector<K1Chunk> ch;
    for (int i = 0; i < u->n_utt; ++i) {
      int64_t T = u->frame_off[i + 1] - u->frame_off[i];
      if (T <= 0 || u->pdf_off[i + 1] == u->pdf_off[i]) continue;
      int n16 = (int)((T + 15) / 16);
      int nchunks = (n16 + maxtiles - 1) / maxtiles;
      int per = (n16 + nchunks - 1) / nchunks;
      for (int c = 0; c < nchunks; ++c) {
        int t0 = c * per * 16;
        int nfr = (int)std::min<int64_t>((int64_t)per * 16, T - t0);
        if (nfr <= 0) break;
        ch.push_back(K1Chunk{i, t0, nfr, 0});
      }
    }
    u->n_chunks = (int)ch.size();
    u->chunk_kq = m->KQ * 16 + k1_nf(ctx, m->KQ);
    rc = dev_upload(ctx, &u->chunks_d, ch);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));  // ch is a local
  }
  rc = ensure_walk(ctx, m, u, reachable_only);
  if (rc) return rc;
  K1Args a;
  a.feats = u->feats_d; a.frame_off = u->frame_off_d; a.chunks = u->chunks_d; a.wimg = m->wimg_d;
  a.utt_tile_off = u->tile_off_d; a.utt_tiles = u->tiles_d;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.err_flag = ctx->err_flag_d; a.D = m->D;
  a.interleave = reachable_only ? 1 : 0;
  if (ctx->opt[KHG_OPT_K1_INTERLEAVE] >= 0) a.interleave = ctx->opt[KHG_OPT_K1_INTERLEAVE];
  const bool aligned = (m->D % 4 == 0) && ((reinterpret_cast<uintptr_t>(u->feats_d) & 15) == 0);
  if (u->n_chunks > 0) {
    KernelTimer kt(ctx, "k1_loglikes");
    if (m->KQ == 10 && k1_nf(ctx, 10) == 6) launch_k1<10, 6, 2>(a, u->n_chunks, aligned, ctx->stream);
    else if (m->KQ == 10) launch_k1<10, 5, 2>(a, u->n_chunks, aligned, ctx->stream);
    else launch_k1<20, 5, 1>(a, u->n_chunks, aligned, ctx->stream);
    HIPCHK(hipGetLastError());
  }
  u->ll_valid = true;
  return KHG_OK;
}
extern "C" int khg_utts_pdf_first(const khg_utts* u, int32_t* first) {
  if (!u || !first) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (u->pdf_first.size() != u->pdfs.size()) std::fill(first, first + u->pdfs.size(), 0);
  else std::copy(u->pdf_first.begin(), u->pdf_first.end(), first);
  return KHG_OK;
}
extern "C" int khg_loglikes(khg_ctx* ctx, const khg_model* m, khg_utts* u) { return loglikes_impl(ctx, m, u, 0); }
extern "C" int khg_loglikes_reachable(khg_ctx* ctx, const khg_model* m, khg_utts* u) { return loglikes_impl(ctx, m, u, 1); }
extern "C" int khg_loglikes_band(khg_ctx* ctx, const khg_model* m, khg_utts* u) { return loglikes_impl(ctx, m, u, 2); }
extern "C" int khg_utts_pdf_last(const khg_utts* u, int32_t* last) {
  if (!u || !last) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (u->pdf_last.size() != u->pdfs.size()) std::fill(last, last + u->pdfs.size(), INT32_MAX);
  else std::copy(u->pdf_last.begin(), u->pdf_last.end(), last);
  return KHG_OK;
}
extern "C" int khg_loglikes_layout(const khg_utts* u, int64_t* ll_off, int64_t* total) {
  if (!u) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (ll_off) std::copy(u->ll_off.begin(), u->ll_off.end(), ll_off);
  if (total) *total = u->ll_total;
  return KHG_OK;
}
extern "C" int khg_loglikes_download(khg_ctx* ctx, const khg_utts* u, float* ll) {
  if (!ctx || !u || !ll) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (!u->ll_valid) return khg_set_error(KHG_E_ARG, "khg_loglikes_download: call khg_loglikes first");
  int rc = check_err_flag(ctx, "khg_loglikes");
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ll, u->ll_d, sizeof(float) * (size_t)u->ll_total, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_loglikes_upload(khg_ctx* ctx, khg_utts* u, const float* ll) {
  if (!ctx || !u || !ll) return khg_set_error(KHG_E_ARG, "bad arguments");
  int rc = wait_ali(ctx, u);
  if (rc) return rc;
  if (!u->pdf_off_d) {
    rc = dev_upload(ctx, &u->pdf_off_d, u->pdf_off);
    if (!rc) rc = dev_upload(ctx, &u->pdfs_d, u->pdfs);
    if (!rc) rc = dev_upload(ctx, &u->ll_off_d, u->ll_off);
    if (!rc) rc = dev_alloc(&u->ll_d, (size_t)u->ll_total);
    if (rc) return rc;
  }
  HIPCHK(hipMemcpyAsync(u->ll_d, ll, sizeof(float) * (size_t)u->ll_total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  u->ll_mode = 0; u->band_model = nullptr;      // the caller's scores: every cell as given
  u->ll_valid = true;
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
// K2
extern "C" void khg_align_config_default(khg_align_config* c) {
  c->beam = 200.0f; c->retry_beam = 0.0f; c->careful = 0; c->acoustic_scale = 1.0f;
  c->max_active = INT32_MAX; c->min_active = 20; c->beam_delta = 0.5f; c->hash_ratio = 2.0f;
  c->like_scale = 0.0f;
}

static int ensure_ali(khg_ctx* ctx, khg_utts* u) {
  if (!u->ali_d) { int rc = dev_alloc(&u->ali_d, (size_t)u->N); if (rc) return rc; }
  return KHG_OK;
}

extern "C" int khg_align(khg_ctx* ctx, const khg_tm* tm, khg_utts* u, const khg_align_config* cfg,
                         int32_t* ali_h, int32_t* words_h, int64_t* words_off_h, int64_t words_cap,
                         float* like_h, int32_t* status_h) {
  if (!ctx || !tm || !u || !cfg) return khg_set_error(KHG_E_ARG, "khg_align: bad arguments");
  if (!u->has_graphs) return khg_set_error(KHG_E_ARG, "khg_align: the utterance set has no decoding graphs");
  if (!u->ll_valid) return khg_set_error(KHG_E_ARG, "khg_align: call khg_loglikes first");
  // decoder-wrappers.cc:29-33
  if ((cfg->retry_beam != 0 && cfg->retry_beam <= cfg->beam) || cfg->beam <= 0.0)
    return khg_set_error(KHG_E_RUNTIME, "Beams do not make sense: beam " + std::to_string(cfg->beam) + ", retry-beam " + std::to_string(cfg->retry_beam));
  // faster-decoder.cc:24-27
  if (!(cfg->hash_ratio >= 1.0) || !(cfg->max_active > 1) || !(cfg->min_active >= 0 && cfg->min_active < cfg->max_active))
    return khg_set_error(KHG_E_RUNTIME, "FasterDecoderOptions assertion failed");
  int rc = wait_ali(ctx, u);
  if (!rc) rc = ensure_ali(ctx, u);
  if (rc) return rc;
  if (!u->bp_d) {
    rc = dev_alloc(&u->bp_d, (size_t)u->bp_off[u->n_utt]);
    if (!rc) rc = dev_alloc(&u->layer_best_d, (size_t)(u->N + u->n_utt));
    if (!rc) rc = dev_alloc(&u->layer_cnt_d, (size_t)(u->N + u->n_utt));
    if (!rc) rc = dev_alloc(&u->path_d, (size_t)u->path_off[u->n_utt]);
    if (!rc) rc = dev_alloc(&u->words_d, (size_t)u->words_off[u->n_utt]);
    if (!rc) rc = dev_alloc(&u->num_words_d, (size_t)u->n_utt);
    if (!rc) rc = dev_alloc(&u->status_d, (size_t)u->n_utt);
    if (!rc) rc = dev_alloc(&u->like_d, (size_t)u->n_utt);
    if (rc) return rc;
  }
  HIPCHK(hipMemsetAsync(u->ali_d, 0, sizeof(int32_t) * (size_t)u->N, ctx->stream));
  K2Args a;
  a.frame_off = u->frame_off_d; a.state_off = u->state_off_d; a.start = u->start_d;
  a.in_off = u->in_off_d; a.in_src = u->in_src_d; a.in_col = u->in_col_d; a.in_tid = u->in_tid_d;
  a.in_olabel = u->in_olabel_d; a.in_w = u->in_w_d; a.out_off = u->out_off_d; a.out_inidx = u->out_inidx_d;
  a.final_w = u->final_d; a.trans_cost = tm->has_trans_cost ? tm->trans_cost_d : nullptr;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d;
  a.bp = u->bp_d; a.bp_off = u->bp_off_d; a.layer_best = u->layer_best_d; a.layer_cnt = u->layer_cnt_d;
  a.path = u->path_d; a.path_off = u->path_off_d;
  a.ali = u->ali_d; a.words = u->words_d; a.words_off = u->words_off_d; a.num_words = u->num_words_d;
  a.like = u->like_d; a.status = u->status_d; a.err_flag = ctx->err_flag_d;
  a.prof = nullptr;
  // launch order of the DP kernel: longest utterances first (built once per set)
  if (!u->k2_order_d && u->n_utt > 0) {
    std::vector<int32_t> ord((size_t)u->n_utt);
    for (int i = 0; i < u->n_utt; ++i) ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) {
      return u->frame_off[x + 1] - u->frame_off[x] > u->frame_off[y + 1] - u->frame_off[y];
    });
    int rc2 = dev_upload(ctx, &u->k2_order_d, ord);
    if (rc2) return rc2;
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  a.order = ctx->opt[KHG_OPT_K2_INORDER] ? nullptr : u->k2_order_d;
  const bool k2prof = ctx->opt[KHG_OPT_K2_PROF] != 0;
  if (k2prof) { HIPCHK(hipMalloc(reinterpret_cast<void**>(&a.prof), sizeof(long long) * 8 * (size_t)u->n_utt)); HIPCHK(hipMemset(a.prof, 0, sizeof(long long) * 8 * (size_t)u->n_utt)); }
  a.beam = cfg->beam; a.retry_beam = cfg->retry_beam; a.acoustic_scale = cfg->acoustic_scale;
  a.like_scale = cfg->like_scale != 0.0f ? cfg->like_scale : cfg->acoustic_scale;
  a.beam_delta = cfg->beam_delta; a.hash_ratio = cfg->hash_ratio;
  a.max_active = cfg->max_active; a.min_active = cfg->min_active;
  a.max_states = u->max_states; a.max_inarcs = u->max_inarcs;
  const size_t S = (size_t)u->max_states, A = (size_t)u->max_inarcs;
  size_t max_npdf = 0;
  for (int i = 0; i < u->n_utt; ++i) max_npdf = std::max<size_t>(max_npdf, (size_t)(u->pdf_off[i + 1] - u->pdf_off[i]));
  // threads: one destination state each (up to 1024), KS states per thread beyond that
  int nthr = (int)std::min<size_t>(1024, (S + 63) / 64 * 64);
  const int ks_force = ctx->opt[KHG_OPT_K2_KS];   // experiment: states per thread on the register-resident path
  if (ks_force == 2 || ks_force == 4) nthr = (int)std::min<size_t>(1024, ((S + ks_force - 1) / ks_force + 63) / 64 * 64);
  const size_t nwave = nthr / 64;
  // register-resident path for the whole batch: in-degree <= 3 (up to 4 states per thread) or <= 6 (one state per thread)
  const bool deg6 = !u->has_eps && u->max_indeg > 3 && u->max_indeg <= 6 && S <= 1024;
  const bool fast = deg6 || (!u->has_eps && u->max_indeg <= 3 && S <= 4096);
  const int KSsel = !fast ? 0 : ((ks_force == 2 || ks_force == 4) && !deg6 && S <= (size_t)1024 * ks_force ? ks_force : (S <= 1024 ? 1 : (S <= 2048 ? 2 : 4)));
  const size_t NSl = fast ? KSsel : 1;
  // trace-back block: fast = five groups of eight layers, one dword per lane and state slot; generic = 33 layers of bytes
  // (fast: also the waves' strips of parked layer minima / counts, 2.5 KB each, in the same area during the forward pass)
  const size_t tb_bytes = fast ? std::max<size_t>(5 * (size_t)nthr * NSl * 4, 2560 * nwave) : (K2_FB + 1) * ((S + 15) & ~size_t(15));
  // cur | nxt | reductions | arcs | in_off | wave minima/counts | flags | [align] | max(score block (generic), trace-back block)
  size_t lds_dp = 16 * S + 8 * K2_MAXW + 8 * A + 4 * (S + 1) + 8 * K2_FB * nwave + 32 + 8 * K2_MAXW + 16 +
                  std::max<size_t>(fast ? 0 : 4 * K2_SB * (max_npdf | 1), tb_bytes) + 64;
  size_t HB = std::max<size_t>(2 * S, 1000);
  size_t lds_f = 32 * S + 8 * HB + 4 * (S + A) + 4 * S + 4 * (S + 1) + 16 * A + A + 64;
  // The order-faithful decoder for the utterances the DP cannot certify: the wave-parallel form with all its tables in LDS; with the
  // graph tables in an HBM scratch slice per utterance (> ~1600 states on a chain graph); the one-lane form beyond that.
  // KHG_K2_SERIAL = 1: always the one-lane form; 2: the HBM-graph wave form wherever its per-frame tables fit (tests, A/B).
  const int odeg_w = u->max_outdeg <= 8 ? std::max(1, (int)u->max_outdeg) : 0;     // 0: exact slot prefix sums
  const bool use_pos = u->has_eps || S > 1000;
  const size_t lds_w_mut = 16 * S + 8 * S + 4 * 4 * S + 4 * S + 4 * max_npdf + (odeg_w ? 0 : 4 * A + 4 * S) + (use_pos ? 4 * S : 0) +
                           (u->has_eps ? 4 * (S + A + 1) : 0) + 8 + 8 * ((std::max(A, S * (size_t)odeg_w) + 63) / 64 + 1);
  const size_t lds_w_graph = 8 * (S + 1) + 5 * 4 * A + (u->has_eps ? 4 * (S + 1) + 4 * A : 0) + A + S + 64;
  const int fmode = ctx->opt[KHG_OPT_K2_SERIAL];
  const bool wave_lds = fmode == 0 && S <= 65535 && lds_w_mut + lds_w_graph <= 160 * 1024;
  const bool wave_gm = fmode != 1 && !wave_lds && S <= 65535 && lds_w_mut <= 160 * 1024;
  const bool lane_gm = !wave_lds && !wave_gm && lds_f > 160 * 1024;
  // Graphs whose DP tables exceed the 160 KB of LDS (a large decoding graph, not a training graph): the generic DP runs with its
  // tables carved out of the same HBM scratch slice.
  const bool gmem = lds_dp > 160 * 1024;
  a.gscratch = nullptr; a.gscratch_stride = 0;
  if (gmem) // (the generic DP's carve-up: no register-resident path)
    lds_dp = 16 * S + 8 * K2_MAXW + 8 * A + 4 * (S + 1) + 8 * K2_FB * nwave + 32 + 8 * K2_MAXW + 16 +
             std::max<size_t>(4 * K2_SB * (max_npdf | 1), (K2_FB + 1) * ((S + 15) & ~size_t(15))) + 64;
  if (gmem || wave_gm || lane_gm) {
    const size_t stride = (std::max(gmem ? lds_dp : 0, std::max(wave_gm ? lds_w_graph : 0, lane_gm ? lds_f : 0)) + 255) & ~size_t(255);
    const size_t need = stride * (size_t)u->n_utt;
    if (need > u->k2_gscratch_bytes) {
      DEVFREE(u->k2_gscratch_d);
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&u->k2_gscratch_d), need));
      u->k2_gscratch_bytes = need;
    }
    a.gscratch = u->k2_gscratch_d; a.gscratch_stride = (int64_t)stride;
  }
  if (gmem) {
    KernelTimer kt(ctx, "k2_viterbi_dp");
    hipLaunchKernelGGL((k2_viterbi_dp<1, 1, false, true>), dim3(u->n_utt), dim3(nthr), 0, ctx->stream, a);
  } else {
    // in-degree <= 2 (a linear transcript's chain of HMM states: self-loop + forward arc): the two-slot instantiation, a sixth fewer
    // instructions per layer than the three-slot one (the layer loop is bound by VALU issue; every slot is evaluated, empty or not)
    const bool deg2 = fast && !deg6 && KSsel == 1 && u->max_indeg <= 2 && ctx->opt[KHG_OPT_K2_KS] != 3;
    // (KHG_K2_KS = 3: the general three-slot kernel, for the A/B)
    const bool sc2 = deg2 && u->same_col;     // ... and one score row per state: one score block / cost conversion per state
    const bool sc3 = fast && !deg6 && !deg2 && KSsel == 1 && u->same_col && ctx->opt[KHG_OPT_K2_KS] != 3;   // three slots, one score row per state
    // Two / four states per thread (graphs of more than 1024 / 2048 states): a block of 1024 threads leaves 128 registers per lane;
    // the three-slot form needs ~180 at two states per thread (84 registers spilled at the one-state kernels' budget of 96: a
    // transcript of > 340 phones ran 9x slower per frame than one of 330), the two-slot forms of chain graphs fit (round 4)
    const bool deg2m = fast && !deg6 && KSsel > 1 && u->max_indeg <= 2;
#define K2_DP_CASES(X)                                                                                         \
    if (deg6) X((k2_viterbi_dp<1, 6, true>));                                                                  \
    else if (sc2) X((k2_viterbi_dp<1, 2, true, false, true>));                                                 \
    else if (deg2) X((k2_viterbi_dp<1, 2, true>));                                                             \
    else if (sc3) X((k2_viterbi_dp<1, 3, true, false, true>));                                                 \
    else if (KSsel == 1) X((k2_viterbi_dp<1, 3, true>));                                                       \
    else if (KSsel == 2 && deg2m && u->same_col) X((k2_viterbi_dp<2, 2, true, false, true>));                  \
    else if (KSsel == 2 && deg2m) X((k2_viterbi_dp<2, 2, true>));                                              \
    else if (KSsel == 2) X((k2_viterbi_dp<2, 3, true>));                                                       \
    else if (KSsel == 4 && deg2m && u->same_col) X((k2_viterbi_dp<4, 2, true, false, true>));                  \
    else if (KSsel == 4 && deg2m) X((k2_viterbi_dp<4, 2, true>));                                              \
    else if (KSsel == 4) X((k2_viterbi_dp<4, 3, true>));                                                       \
    else X((k2_viterbi_dp<1, 1, false>));
#define K2_SET_LDS(FN) HIPCHK(hipFuncSetAttribute((const void*)FN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dp))
#define K2_LAUNCH(FN) hipLaunchKernelGGL(FN, dim3(u->n_utt), dim3(nthr), lds_dp, ctx->stream, a)
    if (lds_dp > 48 * 1024) { K2_DP_CASES(K2_SET_LDS) }
    KernelTimer kt(ctx, "k2_viterbi_dp");
    K2_DP_CASES(K2_LAUNCH)
#undef K2_LAUNCH
#undef K2_SET_LDS
#undef K2_DP_CASES
  }
  HIPCHK(hipGetLastError());
  if (!u->ev_dp) { HIPCHK(hipEventCreateWithFlags(&u->ev_dp, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&u->ev_ali, hipEventDisableTiming)); }
  HIPCHK(hipEventRecord(u->ev_dp, ctx->stream));
  hipStream_t side = ctx->sides[ctx->next_side];
  ctx->next_side = (ctx->next_side + 1) % khg_ctx::NSIDE;
  HIPCHK(hipStreamWaitEvent(side, u->ev_dp, 0));
  if (u->ll_mode == 2 && u->band_model && u->n_schunks > 0) {
    // BAND form of K1: the utterances the DP could not certify are about to be decoded by the order-faithful kernel, which reads
    // every cell a token reaches -- also the ones the band left at their upper bound.  Recompute exactly those utterances (from
    // their first needed tile on, no upper limit); a workgroup of any other utterance returns at once.
    K1sArgs ra = u->band_args;
    ra.repair_status = u->status_d; ra.repair_bit = K2_ST_NEED_FALLBACK;
    khg_model* bm = u->band_model;
    rc = bm->wimgs_sync.before_read(side);
    if (rc) return rc;
    {
      KernelTimer kt(ctx, "k1_band_repair", side);
      const size_t lds_r = u->band_lds + 4 * 513;          // + the list of flagged chunks
      const unsigned gr = (unsigned)std::min<int64_t>(256, ((int64_t)u->n_schunks + 511) / 512);
      if (u->band_ks == 5) {
        HIPCHK(hipFuncSetAttribute((const void*)k1s_repair<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
        hipLaunchKernelGGL((k1s_repair<5>), dim3(gr), dim3(512), lds_r, side, ra, (int)u->n_schunks);
      } else {
        HIPCHK(hipFuncSetAttribute((const void*)k1s_repair<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
        hipLaunchKernelGGL((k1s_repair<10>), dim3(gr), dim3(512), lds_r, side, ra, (int)u->n_schunks);
      }
    }
    HIPCHK(hipGetLastError());
    rc = bm->wimgs_sync.after_read(side);
    if (rc) return rc;
  }
  {
    KernelTimer kt(ctx, "k2_viterbi_faithful", side);
    if (wave_gm) {
      if (lds_w_mut > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful_wave<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w_mut));
      hipLaunchKernelGGL(k2_viterbi_faithful_wave<true>, dim3(u->n_utt), dim3(64), lds_w_mut, side, a, u->has_eps ? 1 : 0, odeg_w, (int)max_npdf);
    } else if (wave_lds) {
      const size_t lds_w = lds_w_mut + lds_w_graph;
      if (lds_w > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful_wave<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
      hipLaunchKernelGGL(k2_viterbi_faithful_wave<false>, dim3(u->n_utt), dim3(64), lds_w, side, a, u->has_eps ? 1 : 0, odeg_w, (int)max_npdf);
    } else if (lane_gm) {
      hipLaunchKernelGGL(k2_viterbi_faithful<true>, dim3(u->n_utt), dim3(64), 0, side, a);
    } else {
      if (lds_f > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      hipLaunchKernelGGL(k2_viterbi_faithful<false>, dim3(u->n_utt), dim3(64), lds_f, side, a);
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(u->ev_ali, side));
  u->ali_pending = true;
  u->ali_valid = true;
  if (k2prof) {  // diagnostics: average s_memtime ticks per phase of k2_viterbi_dp
    std::vector<long long> pr(8 * (size_t)u->n_utt);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(pr.data(), a.prof, pr.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(a.prof);
    double ph[4] = {0, 0, 0, 0}, sT = 0, sS = 0, sf = 0; int n = 0;
    for (int i = 0; i < u->n_utt; ++i) if (pr[i * 8 + 4]) { for (int k = 0; k < 4; ++k) ph[k] += (double)(pr[i * 8 + k + 1] - pr[i * 8 + k]); sT += pr[i * 8 + 5]; sS += pr[i * 8 + 6]; sf += pr[i * 8 + 7]; ++n; }
    if (n) fprintf(stderr, "[KHG_K2_PROF] %d utts, avg T %.1f S %.1f fast %.2f threads %d lds %zu | ticks: setup %.0f forward %.0f traceback %.0f replay %.0f\n",
                   n, sT / n, sS / n, sf / n, nthr, lds_dp, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
  }
  if (!ali_h && !like_h && !status_h && !words_h) return KHG_OK;   // asynchronous: errors surface at khg_ctx_sync / downloads
  rc = wait_ali(ctx, u);
  if (rc) return rc;
  rc = check_err_flag(ctx, "khg_align");  // synchronises
  if (rc) return rc;
  if (ali_h) HIPCHK(hipMemcpyAsync(ali_h, u->ali_d, sizeof(int32_t) * (size_t)u->N, hipMemcpyDeviceToHost, ctx->stream));
  if (like_h) HIPCHK(hipMemcpyAsync(like_h, u->like_d, sizeof(float) * (size_t)u->n_utt, hipMemcpyDeviceToHost, ctx->stream));
  if (status_h) HIPCHK(hipMemcpyAsync(status_h, u->status_d, sizeof(int32_t) * (size_t)u->n_utt, hipMemcpyDeviceToHost, ctx->stream));
  if (words_h && words_off_h) {
    std::vector<int32_t> w((size_t)u->words_off[u->n_utt]), nw((size_t)u->n_utt);
    if (!w.empty()) HIPCHK(hipMemcpyAsync(w.data(), u->words_d, sizeof(int32_t) * w.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(nw.data(), u->num_words_d, sizeof(int32_t) * nw.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int64_t o = 0;
    for (int i = 0; i < u->n_utt; ++i) {
      words_off_h[i] = o;
      int64_t n = std::min<int64_t>(nw[i], u->words_off[i + 1] - u->words_off[i]);
      if (o + n > words_cap) return khg_set_error(KHG_E_ARG, "khg_align: words_cap too small");
      std::copy(w.begin() + u->words_off[i], w.begin() + u->words_off[i] + n, words_h + o);
      o += n;
    }
    words_off_h[u->n_utt] = o;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

extern "C" int khg_ali_upload(khg_ctx* ctx, khg_utts* u, const int32_t* ali) {
  if (!ctx || !u || !ali) return khg_set_error(KHG_E_ARG, "bad arguments");
  int rc = wait_ali(ctx, u);
  if (!rc) rc = ensure_ali(ctx, u);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(u->ali_d, ali, sizeof(int32_t) * (size_t)u->N, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  u->ali_valid = true;
  return KHG_OK;
}

extern "C" int khg_ali_download(khg_ctx* ctx, khg_utts* u, int32_t* ali) {
  if (!ctx || !u || !ali) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (!u->ali_valid) return khg_set_error(KHG_E_ARG, "khg_ali_download: no resident alignment");
  int rc = wait_ali(ctx, u);
  if (!rc) rc = check_err_flag(ctx, "khg_align");
  if (rc) return rc;
  if (u->N) HIPCHK(hipMemcpyAsync(ali, u->ali_d, sizeof(int32_t) * (size_t)u->N, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
// accumulators + K3
struct khg_accs {
  khg_ctx* ctx = nullptr;
  int64_t sumG = 0; int32_t D = 0, num_tids = 0;
  int64_t n = 0, cap = 0;
  double* buf_d = nullptr;
  float* wire_d = nullptr; int64_t wire_cap = 0;   // fp32 wire image of the block (khg_accs_allreduce_f32 only)
  double* occ() const { return buf_d; }
  double* mean() const { return buf_d + sumG; }
  double* var() const { return buf_d + sumG + sumG * D; }
  double* trans() const { return buf_d + sumG + 2 * sumG * D; }
  double* scalars() const { return trans() + num_tids + 1; }
};
extern "C" int khg_accs_create(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_accs** out) {
  if (!ctx || !m || !tm || !out) return khg_set_error(KHG_E_ARG, "khg_accs_create: bad arguments");
  khg_accs* a = new khg_accs();
  a->ctx = ctx; a->sumG = m->sumG; a->D = m->D; a->num_tids = tm->num_tids;
  a->n = a->sumG * (1 + 2 * (int64_t)a->D) + a->num_tids + 1 + 8;
  int rc = dev_alloc(&a->buf_d, (size_t)a->n);
  if (rc) { delete a; return rc; }
  a->cap = a->n;
  *out = a;
  return khg_accs_zero(ctx, a);
}
extern "C" int khg_accs_destroy(khg_accs* a) { if (a) { DEVFREE(a->buf_d); DEVFREE(a->wire_d); delete a; } return KHG_OK; }
extern "C" int khg_accs_zero(khg_ctx* ctx, khg_accs* a) {
  if (!ctx || !a) return khg_set_error(KHG_E_ARG, "bad arguments");
  HIPCHK(hipMemsetAsync(a->buf_d, 0, sizeof(double) * (size_t)a->n, ctx->stream));
  return KHG_OK;
}
extern "C" int khg_accs_size(const khg_accs* a, int64_t* n) { if (!a || !n) return khg_set_error(KHG_E_ARG, "bad arguments"); *n = a->n; return KHG_OK; }
extern "C" int khg_accs_device_ptr(const khg_accs* a, void** p) { if (!a || !p) return khg_set_error(KHG_E_ARG, "bad arguments"); *p = a->buf_d; return KHG_OK; }
extern "C" int khg_accs_download(khg_ctx* ctx, const khg_accs* a, double* buf) {
  if (!ctx || !a || !buf) return khg_set_error(KHG_E_ARG, "bad arguments");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(buf, a->buf_d, sizeof(double) * (size_t)a->n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_accs_upload(khg_ctx* ctx, khg_accs* a, const double* buf) {
  if (!ctx || !a || !buf) return khg_set_error(KHG_E_ARG, "bad arguments");
  HIPCHK(hipMemcpyAsync(a->buf_d, buf, sizeof(double) * (size_t)a->n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

static int accs_allreduce_pieces(khg_ctx* ctx, khg_accs* a, const khg_model* m, int first_pdf, int n_pdf, void* comm, hipStream_t st);
static int ctx_comm_stream(khg_ctx* ctx);
// K3, optionally with C1 pipelined behind it: the pdfs are cut into `nparts` ranges; the accumulate kernels of range i + 1 run on
// the context's stream while the all-reduce of range i's accumulator rows runs on the context's communication stream.
// K3's phase A on the fp16 matrix cores (khg_k3_accstats.hip.inc, k3_accumulate_wave<NB, true>): the scale exponents come from the
// MODEL alone -- every rank of a sharded run derives the same ones, so the statistics do not depend on the sharding: per
// dimension the features are expected inside xb = max_g (|mean| + 8 sigma); x' = x 2^ex peaks in [2^12, 2^13) there (fp16 overflows
// at ~8 xb), fl(x^2)' in [2^9, 2^10) (same limit), and the largest weight column peaks in [2^14, 2^15) (S).  *use = false (the fp32
// phase A runs) when the model side of the f16x2s domain fails (khg_k1_f16x2s.hip.inc: the absolute part of the error bound,
// evaluated at xb, above 4e-6; |S| > 40; the log-sum-exp's 2^28 bound at 16 xb) or when a feature of THIS set overflows fp16.
static int k3_phase_a_scales(khg_ctx* ctx, khg_model* m, khg_utts* u, bool* use) {
  *use = false;
  const int D = m->D, K = 80;
  if (m->KQ != 10 || D > 40) return KHG_OK;
  std::vector<float> xk;
  int rc = k1_maxima(ctx, m, u, &xk);          // the set's column maxima (cached) and the model's (wmax, gcmax; cached per version)
  if (rc) return rc;
  if (m->k3_xb.empty()) {
    uint32_t* b_d = nullptr;
    rc = dev_alloc(&b_d, 64);
    if (rc) return rc;
    std::vector<uint32_t> hb(64, 0);
    hipError_t e = hipMemsetAsync(b_d, 0, 64 * sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) {
      const int64_t n = m->sumG;
      hipLaunchKernelGGL(k3_model_xbound, dim3((int)std::min<int64_t>(2048, (n * D + 255) / 256)), dim3(256), 0, ctx->stream, m->miv_d, m->iv_d, n, D, b_d);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hb.data(), b_d, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    DEVFREE(b_d);
    if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    m->k3_xb.assign((size_t)D, 0.0f);
    for (int d = 0; d < D; ++d) memcpy(&m->k3_xb[(size_t)d], &hb[(size_t)d], sizeof(float));
    m->k3_ex.assign((size_t)K, 0);
    bool ok = true;
    for (int d = 0; d < D; ++d) {
      const float xb = m->k3_xb[(size_t)d];
      if (!(xb > 0.0f) || !(xb < 1.0e18f)) { ok = false; break; }
      m->k3_ex[(size_t)2 * d] = 12 - std::ilogb(xb);               // xb 2^ex in [2^12, 2^13): fp16 overflows beyond 8 xb
      m->k3_ex[(size_t)2 * d + 1] = 9 - std::ilogb(xb * xb);        // xb^2 2^ex in [2^9, 2^10): beyond 8 xb as well
    }
    int S = INT_MAX;
    if (ok) {
      for (int k = 0; k < 2 * D; ++k) if (m->wmax[(size_t)k] > 0.0f) S = std::min(S, 14 - std::ilogb(m->wmax[(size_t)k]) + m->k3_ex[(size_t)k]);
      if (S == INT_MAX) S = 0;
      if (S < -40 || S > 40) ok = false;
    }
    if (ok) {
      double floor_sum = 0.0, bound = (double)m->gcmax;
      for (int k = 0; k < 2 * D; ++k) {
        const double xbk = (k & 1) ? (double)m->k3_xb[(size_t)(k >> 1)] * (double)m->k3_xb[(size_t)(k >> 1)] : (double)m->k3_xb[(size_t)(k >> 1)];
        floor_sum += std::ldexp((double)m->wmax[(size_t)k], S - m->k3_ex[(size_t)k]) + std::ldexp(xbk, m->k3_ex[(size_t)k]);
        bound += (double)m->wmax[(size_t)k] * xbk * ((k & 1) ? 128.0 : 16.0);
      }
      // (4e-6: the headroom for features outside the model's envelope costs two bits against K1s, whose planes peak at 2^14 by
      //  construction; the fp32 chain this replaces carries ~7e-7 B, i.e. ~1e-4 at the same shapes)
      if (!(std::ldexp(floor_sum, -25 - S) <= 4.0e-6) || !(bound <= 268435456.0)) ok = false;
      if (ctx->opt[KHG_OPT_DEBUG]) fprintf(stderr, "[khg] K3 fp16 phase A: S %d, floor %.3g, bound %.3g -> %s\n", S, std::ldexp(floor_sum, -25 - S), bound, ok ? "on" : "off");
    }
    m->k3_S = ok ? S : 0;
    m->k3_f16_ok = ok;
    if (ok) {
      if (!m->k3_ex_d) { rc = dev_alloc(&m->k3_ex_d, (size_t)K); if (rc) return rc; }
      HIPCHK(hipMemcpyAsync(m->k3_ex_d, m->k3_ex.data(), sizeof(int32_t) * (size_t)K, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
    }
  }
  if (!m->k3_f16_ok) return KHG_OK;
  for (int k = 0; k < 2 * D; ++k)
    if (!(std::ldexp((double)xk[(size_t)k], m->k3_ex[(size_t)k]) < 65504.0)) return KHG_OK;     // a feature beyond 64 xb: fp32 phase A for this set
  *use = true;
  return KHG_OK;
}

static int acc_stats_impl(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc, void* comm, int nparts) {
  if (!ctx || !m || !tm || !u || !acc) return khg_set_error(KHG_E_ARG, "khg_acc_stats: bad arguments");
  if (!u->ali_valid) return khg_set_error(KHG_E_ARG, "khg_acc_stats: no resident alignment (khg_align or khg_ali_upload first)");
  if (m->D != u->D || acc->D != m->D || acc->sumG != m->sumG || acc->num_tids != tm->num_tids)
    return khg_set_error(KHG_E_RUNTIME, "khg_acc_stats: accumulator / model / feature dimensions do not match");
  if (tm->max_pdf >= m->P) return khg_set_error(KHG_E_RUNTIME, "khg_acc_stats: transition model refers to pdf-ids the model does not have");
  int rc = wait_ali(ctx, u);
  if (rc) return rc;
  if (!u->frame_ids_d || u->k3_P != m->P || u->k3_tids != tm->num_tids) {
    DEVFREE(u->pdf_count_d); DEVFREE(u->pdf_cursor_d); DEVFREE(u->pdf_start_d); DEVFREE(u->tid_count_d); DEVFREE(u->frame_ids_d);
    rc = dev_alloc(&u->pdf_count_d, (size_t)m->P);
    if (!rc) rc = dev_alloc(&u->pdf_cursor_d, (size_t)m->P);
    if (!rc) rc = dev_alloc(&u->pdf_start_d, (size_t)m->P + 1);
    if (!rc) rc = dev_alloc(&u->tid_count_d, (size_t)tm->num_tids + 1);
    if (!rc) rc = dev_alloc(&u->frame_ids_d, (size_t)u->N);
    if (rc) return rc;
    u->k3_P = m->P; u->k3_tids = tm->num_tids;
  }
  HIPCHK(hipMemsetAsync(u->pdf_count_d, 0, sizeof(int32_t) * (size_t)m->P, ctx->stream));
  HIPCHK(hipMemsetAsync(u->tid_count_d, 0, sizeof(unsigned long long) * ((size_t)tm->num_tids + 1), ctx->stream));
  K3Args a;
  a.feats = u->feats_d; a.ali = u->ali_d; a.id2pdf = tm->id2pdf_d; a.num_tids = tm->num_tids;
  a.N = u->N; a.P = m->P; a.D = m->D;
  a.gauss_off = m->gauss_off_d; a.gconsts = m->gconsts_d; a.means_invvars = m->miv_d; a.inv_vars = m->iv_d; a.nhalf_inv_vars = m->nhiv_d;
  a.pdf_count = u->pdf_count_d; a.pdf_start = u->pdf_start_d; a.pdf_cursor = u->pdf_cursor_d;
  a.frame_ids = u->frame_ids_d; a.tid_count = u->tid_count_d;
  a.occ = acc->occ(); a.mean_acc = acc->mean(); a.var_acc = acc->var(); a.trans_acc = acc->trans(); a.scalars = acc->scalars();
  a.weight = weight; a.err_flag = ctx->err_flag_d; a.part = nullptr; a.ll_part = nullptr; a.pdf0 = 0; a.npdf = m->P; a.items = nullptr; a.item_off = nullptr;
  a.pa_ex = nullptr; a.pa_S = 0; a.pa_scale = 1.0f; a.pa_inv = 1.0f; a.pa_c1 = 1.44269504088896340736f;
  nparts = std::max(1, std::min(nparts, m->P));
  if (comm && nparts > 1) { rc = ctx_comm_stream(ctx); if (rc) return rc; }
  if (u->N > 0) {
    const int gb = (int)std::min<int64_t>(4096, (u->N + 255) / 256);
    {
      KernelTimer kt(ctx, "k3_bucket");
      // KHG_OPT_K3_BUCKET = 1: cursor-bump scatter (bucket order depends on the atomics)
      if (ctx->opt[KHG_OPT_K3_BUCKET] == 1 || u->N >= (int64_t)INT_MAX) {
        hipLaunchKernelGGL(k3_count, dim3(gb), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(k3_scan, dim3(1), dim3(1024), 0, ctx->stream, a);
        hipLaunchKernelGGL(k3_scatter, dim3(gb), dim3(256), 0, ctx->stream, a);
      } else if (ctx->opt[KHG_OPT_K3_BUCKET] == 2 && m->P <= K3_CS_MAXP) {
        // the library's own stable counting sort (khg_k3_accstats.hip.inc: k3_cs_*; opt-in: 1.27 ms against the radix sort's 0.80 at the
        // bench size): blocks of CB consecutive frames
        const int nw = m->P + 1 <= 7168 ? 4 : 2;                      // waves of a placing block: nw x (P + 1 + 1024) counters of LDS
        int64_t CB = 32768;
        while (CB > 64 * nw * 4 && (u->N + CB - 1) / CB < 1024) CB /= 2;   // enough blocks to fill the chip on small sets
        const int nblk = (int)((u->N + CB - 1) / CB);
        const size_t hist_n = (size_t)nblk * ((size_t)m->P + 1);
        if (u->cs_hist_n < hist_n) { DEVFREE(u->cs_hist_d); rc = dev_alloc(&u->cs_hist_d, hist_n); if (rc) return rc; u->cs_hist_n = hist_n; }
        if (u->cs_tot_n < (size_t)m->P + 1) { DEVFREE(u->cs_tot_d); rc = dev_alloc(&u->cs_tot_d, (size_t)m->P + 1); if (rc) return rc; u->cs_tot_n = (size_t)m->P + 1; }
        K3CsArgs c{u->cs_hist_d, u->cs_tot_d, (int32_t)CB, nblk};
        const bool ldst = tm->num_tids <= K3_LDS_TIDS;
        const size_t lds_h = sizeof(unsigned int) * ((size_t)m->P + 1 + (ldst ? (size_t)tm->num_tids + 1 : 0));
        const size_t lds_p = sizeof(unsigned int) * (size_t)nw * ((size_t)m->P + 1 + 1024);     // per-wave counters + the run-head hash tags
        if (lds_h > 48 * 1024) {
          HIPCHK(hipFuncSetAttribute((const void*)k3_cs_hist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
          HIPCHK(hipFuncSetAttribute((const void*)k3_cs_hist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
        }
        if (lds_p > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k3_cs_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        if (ldst) hipLaunchKernelGGL(k3_cs_hist<true>, dim3(nblk), dim3(256), lds_h, ctx->stream, a, c);
        else hipLaunchKernelGGL(k3_cs_hist<false>, dim3(nblk), dim3(256), lds_h, ctx->stream, a, c);
        hipLaunchKernelGGL(k3_cs_scan, dim3((m->P + 256) / 256), dim3(256), 0, ctx->stream, a, c);
        hipLaunchKernelGGL(k3_cs_starts, dim3(1), dim3(1024), 0, ctx->stream, a, c);
        hipLaunchKernelGGL(k3_cs_place, dim3(nblk), dim3(64 * nw), lds_p, ctx->stream, a, c);
      } else {
        // stable sort of (pdf, frame) pairs: frames of a pdf stay in frame order; the bucket boundaries are read
        // off the sorted keys
        int bits = 1;
        while ((1 << bits) <= m->P) ++bits;            // keys are 0..P
        if (!u->sort_keys_d) {
          rc = dev_alloc(&u->sort_keys_d, (size_t)u->N);
          if (!rc) rc = dev_alloc(&u->sort_keys_out_d, (size_t)u->N);
          if (!rc) rc = dev_alloc(&u->sort_vals_d, (size_t)u->N);
          if (rc) return rc;
        }
        size_t need = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, u->sort_keys_d, u->sort_keys_out_d, u->sort_vals_d,
                                                  reinterpret_cast<uint32_t*>(u->frame_ids_d), (int)u->N, 0, bits, ctx->stream));
        if (need > u->sort_tmp_bytes) {
          DEVFREE(u->sort_tmp_d);
          HIPCHK(hipMalloc(&u->sort_tmp_d, need));
          u->sort_tmp_bytes = need;
        }
        if (tm->num_tids <= K3_LDS_TIDS) hipLaunchKernelGGL(k3_sort_keys<true>, dim3(std::min(gb, 1024)), dim3(256), 0, ctx->stream, a, u->sort_keys_d, u->sort_vals_d);
        else hipLaunchKernelGGL(k3_sort_keys<false>, dim3(std::min(gb, 1024)), dim3(256), 0, ctx->stream, a, u->sort_keys_d, u->sort_vals_d);
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(u->sort_tmp_d, need, u->sort_keys_d, u->sort_keys_out_d, u->sort_vals_d,
                                                  reinterpret_cast<uint32_t*>(u->frame_ids_d), (int)u->N, 0, bits, ctx->stream));
        hipLaunchKernelGGL(k3_bounds, dim3((m->P + 256) / 256), dim3(256), 0, ctx->stream, a, u->sort_keys_out_d);
      }
    }
    // Work items of the accumulate kernels (k3_make_items): ny_base slices per pdf, more for a pdf whose bucket is far above the
    // average slice (2 x; silence in real transcripts).  -> the number of blocks to launch for a range of np pdfs (an upper bound
    // from N and P alone: the bucket sizes stay on the device) in *extra_blocks; parked: the slices park images (wave forms).
    int64_t k3_extra_blocks = 0;
    auto make_items = [&](int ny_base, bool parked, size_t nsum1) -> int {
      const int64_t avg = u->N / std::max(1, m->P);
      const int target = (int)std::min<int64_t>(1 << 30, std::max<int64_t>(512, 2 * avg / ny_base));
      const int64_t per_t = u->N / target;
      k3_extra_blocks = std::min<int64_t>(per_t + m->P, 2 * per_t) + 1;
      const int64_t max_items = (int64_t)m->P * ny_base + k3_extra_blocks;
      const int64_t max_slots = !parked ? INT_MAX : ny_base > 1 ? max_items : 2 * per_t + 1;
      if (max_items >= INT_MAX) return khg_set_error(KHG_E_UNSUPPORTED, "khg_acc_stats: too many work items");
      if (u->k3_items_n < (size_t)max_items) {
        DEVFREE(u->k3_items_d);
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&u->k3_items_d), sizeof(K3Item) * (size_t)max_items));
        u->k3_items_n = (size_t)max_items;
      }
      if (u->k3_item_off_n < (size_t)m->P + 1) {
        DEVFREE(u->k3_item_off_d);
        int rc2 = dev_alloc(&u->k3_item_off_d, (size_t)m->P + 1);
        if (rc2) return rc2;
        u->k3_item_off_n = (size_t)m->P + 1;
      }
      if (parked && u->k3_part_n < (size_t)max_slots * nsum1) {
        DEVFREE(u->k3_part_d);
        int rc2 = dev_alloc(&u->k3_part_d, (size_t)max_slots * nsum1);
        if (rc2) return rc2;
        u->k3_part_n = (size_t)max_slots * nsum1;
      }
      hipLaunchKernelGGL(k3_make_items, dim3(1), dim3(1024), 0, ctx->stream, a, ny_base, target, (int)max_items, (int)std::min<int64_t>(max_slots, INT_MAX),
                         reinterpret_cast<K3Item*>(u->k3_items_d), u->k3_item_off_d);
      a.items = reinterpret_cast<const K3Item*>(u->k3_items_d); a.item_off = u->k3_item_off_d;
      return KHG_OK;
    };
    int maxG = 0;
    for (int p = 0; p < m->P; ++p) maxG = std::max(maxG, m->gauss_off[p + 1] - m->gauss_off[p]);
    const int64_t avg_chunks = (u->N / std::max(1, m->P) + K3_CHUNK - 1) / K3_CHUNK;
    const int k3form = ctx->opt[KHG_OPT_K3_FORM];     // 1: the chunk-per-block MFMA form for every shape; 2: the VALU form
    // the chunk-per-block MFMA form holds 16 * 4 * NBW Gaussians: NBW <= 2 at D <= 80, <= 4 at D <= 40 (the accumulators are registers)
    const bool use_mfma = (maxG <= 128 || (maxG <= 256 && m->KQ == 10)) && k3form != 2 && m->KQ != 0;
    const bool use_wave = use_mfma && m->KQ == 10 && maxG <= 64 && k3form != 1;
    if (use_wave) {
      // wave-local form: W in LDS + per-wave planes during the tile loop, the fp64 fold image afterwards
      const int nb = (maxG + 15) / 16;
      // phase A on the fp16 matrix cores where the model-derived scales hold (KHG_K3_PHASEA=f32 keeps the fp32 chain)
      bool f16a = false;
      // (pdfs of <= 32 Gaussians keep the fp32 chain: 40 MFMAs per tile are not worth the split, and the two-waves-per-SIMD
      //  instantiations have no registers for the fp16 W pieces)
      if (nb >= 3 && ctx->opt[KHG_OPT_K3_PHASE_A] == 0 && ctx->opt[KHG_OPT_K3_PHASE_B] != 1) { rc = k3_phase_a_scales(ctx, const_cast<khg_model*>(m), u, &f16a); if (rc) return rc; }
      if (f16a) {
        a.pa_ex = m->k3_ex_d; a.pa_S = m->k3_S;
        a.pa_scale = std::ldexp(1.0f, m->k3_S); a.pa_inv = std::ldexp(1.0f, -m->k3_S); a.pa_c1 = std::ldexp(1.44269504088896340736f, -m->k3_S);
      }
      // phase B on the fp16 matrix cores as well (k3_accumulate_wave16; KHG_OPT_K3_PHASE_B = 2): where phase A's split planes exist
      // and the weight is an ordinary number
      const bool f16b = f16a && ctx->opt[KHG_OPT_K3_PHASE_B] == 2 && std::isfinite(weight) && std::fabs(weight) > 1.0e-30f && std::fabs(weight) < 1.0e30f;
      if (f16b) { a.pb_SG = 13 - std::ilogb(std::fabs(weight)); a.pb_gscale = std::ldexp(1.0f, a.pb_SG); }
      // fp32 phase A: W + the waves' planes; fp16 phase A: the waves' planes + their split planes; then the fold image
      const size_t lds = std::max<size_t>(f16a ? sizeof(float) * (4 * 4 * 16 * 20) + 2 * (size_t)(4 * 2 * 16 * K3_XH_ROW)
                                               : sizeof(float) * ((size_t)nb * 20 * 64 + 4 * 4 * 16 * 20),
                                          sizeof(double) * ((size_t)nb * 16 * 80 + (size_t)nb * 16));
      const int64_t avg_tiles = (u->N / std::max(1, m->P) + 15) / 16;
      int ny = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(32, (avg_tiles + 15) / 16), (4096 + m->P - 1) / m->P));
      if (ctx->opt[KHG_OPT_K3_NY] > 0) ny = ctx->opt[KHG_OPT_K3_NY];
      // per-pdf log-like partials (always) and, with several blocks per pdf, the slice images they park
      const size_t nsum1 = (size_t)nb * 16 * 80 + (size_t)nb * 16 + 1;
      if (u->k3_llpart_n < (size_t)m->P) {
        DEVFREE(u->k3_llpart_d);
        rc = dev_alloc(&u->k3_llpart_d, (size_t)m->P);
        if (rc) return rc;
        u->k3_llpart_n = (size_t)m->P;
      }
      rc = make_items(ny, true, nsum1);
      if (rc) return rc;
      HIPCHK(hipMemsetAsync(u->k3_llpart_d, 0, sizeof(double) * (size_t)m->P, ctx->stream));
      a.ll_part = u->k3_llpart_d;
      a.part = u->k3_part_d;
      for (int part = 0; part < nparts; ++part) {
      const int p0 = (int)((int64_t)m->P * part / nparts), np = (int)((int64_t)m->P * (part + 1) / nparts) - p0;
      a.pdf0 = p0; a.npdf = np;
      const unsigned nblk = (unsigned)((int64_t)np * ny + k3_extra_blocks);
      {
      KernelTimer kt(ctx, "k3_accumulate");
      // phase B on the fp64 matrix pipe (default: products exact, N ranks sum to the one-rank statistics to 1e-12) or, with
      // KHG_K3_PHASEB=f32, on the fp32 pipe with 256-frame fp32 partial sums (k3_accumulate_wave32: 13 % faster, ~1e-6)
      const bool exact_b = ctx->opt[KHG_OPT_K3_PHASE_B] != 1;       // (2 = the fp16 matrix cores where they apply, else fp64)
      // k3_accumulate_wave32: the workgroup's fp64 image + W + two x planes per wave
      const size_t lds32 = sizeof(double) * ((size_t)nb * 16 * 80 + (size_t)nb * 16) + sizeof(float) * ((size_t)nb * 20 * 64 + 4 * 2 * 16 * 20);
      // k3_accumulate_wave16: per wave two tiles' phase-A planes + the phase-B planes (halves), then the split W operands
      const size_t lds16 = std::max<size_t>(2 * (size_t)4 * (2 * (2 * 16 * K3_XH_ROW) + 2 * 2 * 40 * 36) + (size_t)nb * 3 * 2 * 64 * 16,
                                            sizeof(double) * ((size_t)nb * 16 * 80 + (size_t)nb * 16));
#define K3_WAVE_LAUNCH(NBV)                                                                                            \
  do {                                                                                                                  \
    if (!exact_b && lds32 > 48 * 1024)                                                                                  \
      HIPCHK(hipFuncSetAttribute((const void*)k3_accumulate_wave32<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32)); \
    if (f16b) {                                                                                                          \
      HIPCHK(hipFuncSetAttribute((const void*)k3_accumulate_wave16<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16)); \
      hipLaunchKernelGGL((k3_accumulate_wave16<NBV>), dim3(nblk), dim3(256), lds16, ctx->stream, a);                     \
    } else if (exact_b && f16a) hipLaunchKernelGGL((k3_accumulate_wave<NBV, true>), dim3(nblk), dim3(256), lds, ctx->stream, a);    \
    else if (exact_b) hipLaunchKernelGGL((k3_accumulate_wave<NBV>), dim3(nblk), dim3(256), lds, ctx->stream, a);         \
    else hipLaunchKernelGGL((k3_accumulate_wave32<NBV>), dim3(nblk), dim3(256), lds32, ctx->stream, a);                  \
    hipLaunchKernelGGL((k3_wave_finalize<NBV>), dim3(np), dim3(256), 0, ctx->stream, a);                                 \
  } while (0)
      switch (nb) {
        case 1: K3_WAVE_LAUNCH(1); break;
        case 2: K3_WAVE_LAUNCH(2); break;
        case 3: K3_WAVE_LAUNCH(3); break;
        default: K3_WAVE_LAUNCH(4); break;
      }
#undef K3_WAVE_LAUNCH
      }
      if (comm && nparts > 1) { rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm, nullptr); if (rc) return rc; }
      }
      a.pdf0 = 0; a.npdf = m->P;
      hipLaunchKernelGGL(k3_wave_scalars, dim3(1), dim3(1024), 0, ctx->stream, a);
    } else if (use_mfma) {
      // fp32 + fp64 MFMA form; fewer, longer blocks: the fp64 accumulators stay in registers per block
      const size_t lds = sizeof(float) * ((size_t)4 * K3_CHUNK * 2 * m->KQ + 5 * K3_CHUNK);   // 4 planes [64][KH] + reductions
      // slices per pdf: every block ends with one fp64 atomic per accumulator cell (G*(2D+1) of them), so
      // use as few blocks as still fill the chip (~4096 = 256 CUs x 8 blocks x 2 rounds)
      int ny = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(32, avg_chunks), (4096 + m->P - 1) / m->P));
      if (ctx->opt[KHG_OPT_K3_NY] > 0) ny = ctx->opt[KHG_OPT_K3_NY];
      rc = make_items(ny, false, 0);
      if (rc) return rc;
      for (int part = 0; part < nparts; ++part) {
        const int p0 = (int)((int64_t)m->P * part / nparts), np = (int)((int64_t)m->P * (part + 1) / nparts) - p0;
        a.pdf0 = p0; a.npdf = np;
        const unsigned nblk = (unsigned)((int64_t)np * ny + k3_extra_blocks);
        {
          KernelTimer kt(ctx, "k3_accumulate");
          if (m->KQ == 10 && maxG <= 64) hipLaunchKernelGGL((k3_accumulate_mfma<10, 1>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 10 && maxG <= 128) hipLaunchKernelGGL((k3_accumulate_mfma<10, 2>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 10 && maxG <= 192) hipLaunchKernelGGL((k3_accumulate_mfma<10, 3>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 10) hipLaunchKernelGGL((k3_accumulate_mfma<10, 4>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (maxG <= 64) hipLaunchKernelGGL((k3_accumulate_mfma<20, 1>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else hipLaunchKernelGGL((k3_accumulate_mfma<20, 2>), dim3(nblk), dim3(256), lds, ctx->stream, a);
        }
        if (comm && nparts > 1) { rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm, nullptr); if (rc) return rc; }
      }
      a.pdf0 = 0;
    } else {
      const size_t lds = sizeof(float) * (size_t)K3_CHUNK * ((size_t)(m->KQ ? 4 * m->KQ : (m->D | 1)) + (maxG | 1) + 4);
      if (lds > 160 * 1024) return khg_set_error(KHG_E_UNSUPPORTED, "khg_acc_stats: pdf too large for the LDS chunk buffers");
      const void* k3fn = m->KQ == 10 ? (const void*)k3_accumulate<10> : m->KQ == 20 ? (const void*)k3_accumulate<20> : (const void*)k3_accumulate<0>;
      if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(k3fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int ny = (int)std::max<int64_t>(1, std::min<int64_t>(64, (avg_chunks + 3) / 4));
      rc = make_items(ny, false, 0);
      if (rc) return rc;
      for (int part = 0; part < nparts; ++part) {
        const int p0 = (int)((int64_t)m->P * part / nparts), np = (int)((int64_t)m->P * (part + 1) / nparts) - p0;
        a.pdf0 = p0; a.npdf = np;
        const unsigned nblk = (unsigned)((int64_t)np * ny + k3_extra_blocks);
        {
          KernelTimer kt(ctx, "k3_accumulate");
          if (m->KQ == 10) hipLaunchKernelGGL(k3_accumulate<10>, dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 20) hipLaunchKernelGGL(k3_accumulate<20>, dim3(nblk), dim3(256), lds, ctx->stream, a);
          else hipLaunchKernelGGL(k3_accumulate<0>, dim3(nblk), dim3(256), lds, ctx->stream, a);
        }
        if (comm && nparts > 1) { rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm, nullptr); if (rc) return rc; }
      }
      a.pdf0 = 0;
    }
    HIPCHK(hipGetLastError());
  } else if (comm && nparts > 1) {
    // a rank without frames launches nothing but takes part in the same collectives, in the same order, as every other rank: the
    // sequence is a function of (P, nparts) only
    for (int part = 0; part < nparts; ++part) {
      const int p0 = (int)((int64_t)m->P * part / nparts), np = (int)((int64_t)m->P * (part + 1) / nparts) - p0;
      rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm, nullptr);
      if (rc) return rc;
    }
  }
  if (comm) {
    // the rest of the block: all of it when nothing was pipelined, else the transition counts and the scalars; then the kernels'
    // stream waits for the communication stream
    if (nparts > 1) rc = accs_allreduce_pieces(ctx, acc, m, -1, 0, comm, nullptr);
    else rc = khg_accs_allreduce(ctx, acc, comm);
    if (rc) return rc;
  }
  return KHG_OK;   // asynchronous: kernel-side errors surface at khg_ctx_sync / khg_accs_download
}
extern "C" int khg_acc_stats(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc) {
  return acc_stats_impl(ctx, m, tm, u, weight, acc, nullptr, 1);
}
extern "C" int khg_acc_stats_reduce(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc, void* comm, int32_t nparts) {
  return acc_stats_impl(ctx, m, tm, u, weight, acc, comm, nparts <= 0 ? 4 : nparts);
}

// ------------------------------------------------------------------------------------------
// C1: the cross-GPU sum of the accumulator block, RCCL called directly.  RCCL is bound at
// run time from whatever copy the process already holds (torch bundles one with the same SONAME; two
// copies in one process would each want their own view of the devices), so the library has no link-time
// dependency on it and a one-GPU user never loads it.
namespace {
struct KhgNcclId { char internal[KHG_COMM_ID_BYTES]; };   // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
struct RcclApi {
  void* h = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, KhgNcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;      // optional (sharded M-step)
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;            // optional (khg_comm_info)
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static int rccl_bind() {
  if (g_rccl.AllReduce) return KHG_OK;
  void* h = nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy already in the process
  for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return khg_set_error(KHG_E_UNSUPPORTED, std::string("RCCL not available: ") + dlerror());
  RcclApi a; a.h = h;
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  a.Reduce = reinterpret_cast<decltype(a.Reduce)>(dlsym(h, "ncclReduce"));
  a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(h, "ncclBroadcast"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
  a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(dlsym(h, "ncclGetVersion"));
  if (!a.AllReduce || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !a.GroupStart || !a.GroupEnd)
    return khg_set_error(KHG_E_UNSUPPORTED, "RCCL library lacks ncclAllReduce / ncclCommInitRank");
  g_rccl = a;
  return KHG_OK;
}
static int rccl_fail(const char* what, int r) {
  return khg_set_error(KHG_E_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
constexpr int kNcclSum = 0, kNcclInt8 = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;   // rccl.h: ncclRedOp_t / ncclDataType_t
__global__ __launch_bounds__(256) void c1_narrow(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) dst[i] = (float)src[i];
}
__global__ __launch_bounds__(256) void c1_widen(const float* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) dst[i] = (double)src[i];
}
}  // namespace

static int ctx_comm_stream(khg_ctx* ctx) {
  if (!ctx->comm_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  if (!ctx->ev_k3) HIPCHK(hipEventCreateWithFlags(&ctx->ev_k3, hipEventDisableTiming));
  if (!ctx->ev_c1) HIPCHK(hipEventCreateWithFlags(&ctx->ev_c1, hipEventDisableTiming));
  return KHG_OK;
}
// The rows of pdfs [first_pdf, first_pdf + n_pdf) of the block -- occ, mean_acc, var_acc: three contiguous pieces -- or, with
// first_pdf < 0, the transition counts and scalars behind them, summed over the ranks in ONE RCCL group.  st == nullptr: the
// pieces run on the context's communication stream BEHIND everything enqueued on its kernel stream so far, and the kernel stream
// then waits for them only when the tail (first_pdf < 0) has gone out: the pipelined form of khg_acc_stats_reduce.
static int accs_allreduce_pieces(khg_ctx* ctx, khg_accs* a, const khg_model* m, int first_pdf, int n_pdf, void* comm, hipStream_t st) {
  int rc = rccl_bind();
  if (rc) return rc;
  const bool piped = st == nullptr;
  if (piped) {
    rc = ctx_comm_stream(ctx);
    if (rc) return rc;
    st = ctx->comm_stream;
    HIPCHK(hipEventRecord(ctx->ev_k3, ctx->stream));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_k3, 0));
  }
  struct Piece { double* p; size_t n; } pc[3];
  int npc = 0;
  if (first_pdf < 0) {
    pc[npc++] = Piece{a->trans(), (size_t)a->num_tids + 1 + 8};
  } else {
    if (first_pdf + n_pdf > m->P || n_pdf < 0) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce_range: pdf range outside the model");
    const int64_t g0 = m->gauss_off[first_pdf], g1 = m->gauss_off[first_pdf + n_pdf];
    if (g1 > g0) {
      pc[npc++] = Piece{a->occ() + g0, (size_t)(g1 - g0)};
      pc[npc++] = Piece{a->mean() + g0 * a->D, (size_t)((g1 - g0) * a->D)};
      pc[npc++] = Piece{a->var() + g0 * a->D, (size_t)((g1 - g0) * a->D)};
    }
  }
  {
    KernelTimer kt(ctx, "c1_allreduce", st);
    int r = g_rccl.GroupStart();
    for (int i = 0; i < npc && !r; ++i) r = g_rccl.AllReduce(pc[i].p, pc[i].p, pc[i].n, kNcclFloat64, kNcclSum, comm, st);
    const int r2 = g_rccl.GroupEnd();
    if (r || r2) return rccl_fail("ncclAllReduce (range)", r ? r : r2);
  }
  if (piped && first_pdf < 0) {
    HIPCHK(hipEventRecord(ctx->ev_c1, st));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_c1, 0));
  }
  return KHG_OK;
}
extern "C" int khg_accs_allreduce_range(khg_ctx* ctx, khg_accs* a, const khg_model* m, int32_t first_pdf, int32_t n_pdf, void* comm) {
  if (!ctx || !a || !m) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce_range: bad arguments");
  if (a->sumG != m->sumG || a->D != m->D) return khg_set_error(KHG_E_RUNTIME, "khg_accs_allreduce_range: accumulator / model layouts differ");
  if (!comm) return KHG_OK;
  return accs_allreduce_pieces(ctx, a, m, first_pdf, n_pdf, comm, ctx->stream);
}

extern "C" int khg_comm_unique_id(void* id_out) {
  if (!id_out) return khg_set_error(KHG_E_ARG, "khg_comm_unique_id: id_out is NULL");
  int rc = rccl_bind();
  if (rc) return rc;
  int r = g_rccl.GetUniqueId(id_out);
  return r ? rccl_fail("ncclGetUniqueId", r) : KHG_OK;
}
extern "C" int khg_comm_create(khg_ctx* ctx, int32_t nranks, int32_t rank, const void* id, void** comm_out) {
  if (!ctx || !id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return khg_set_error(KHG_E_ARG, "khg_comm_create: bad arguments");
  int rc = rccl_bind();
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  KhgNcclId uid;
  memcpy(uid.internal, id, sizeof(uid.internal));
  void* comm = nullptr;
  int r = g_rccl.CommInitRank(&comm, nranks, uid, rank);
  if (r) return rccl_fail("ncclCommInitRank", r);
  *comm_out = comm;
  return KHG_OK;
}
// what RCCL itself says about a communicator: the number of ranks it spans, this process's rank in it, the library's version code
extern "C" int khg_comm_info(void* comm, int32_t* nranks, int32_t* rank, int32_t* version) {
  int rc = rccl_bind();
  if (rc) return rc;
  int v = 0;
  if (nranks) { *nranks = 0; if (comm && g_rccl.CommCount) { int r = g_rccl.CommCount(comm, &v); if (r) return rccl_fail("ncclCommCount", r); *nranks = v; } }
  if (rank) { *rank = -1; if (comm && g_rccl.CommUserRank) { int r = g_rccl.CommUserRank(comm, &v); if (r) return rccl_fail("ncclCommUserRank", r); *rank = v; } }
  if (version) { *version = 0; if (g_rccl.GetVersion) { int r = g_rccl.GetVersion(&v); if (r) return rccl_fail("ncclGetVersion", r); *version = v; } }
  return KHG_OK;
}
extern "C" int khg_comm_destroy(void* comm) {
  if (!comm) return KHG_OK;
  int rc = rccl_bind();
  if (rc) return rc;
  int r = g_rccl.CommDestroy(comm);
  return r ? rccl_fail("ncclCommDestroy", r) : KHG_OK;
}
extern "C" int khg_accs_allreduce(khg_ctx* ctx, khg_accs* a, void* comm) {
  if (!ctx || !a) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce: bad arguments");
  if (!comm) return KHG_OK;
  int rc = rccl_bind();
  if (rc) return rc;
  KernelTimer kt(ctx, "c1_allreduce");
  int r = g_rccl.AllReduce(a->buf_d, a->buf_d, (size_t)a->n, kNcclFloat64, kNcclSum, comm, ctx->stream);
  return r ? rccl_fail("ncclAllReduce", r) : KHG_OK;
}
extern "C" int khg_accs_allreduce_f32(khg_ctx* ctx, khg_accs* a, void* comm) {
  if (!ctx || !a) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce_f32: bad arguments");
  if (comm) { int rc = rccl_bind(); if (rc) return rc; }
  if (a->wire_cap < a->n) {
    DEVFREE(a->wire_d);
    int rc = dev_alloc(&a->wire_d, (size_t)a->n);
    if (rc) return rc;
    a->wire_cap = a->n;
  }
  KernelTimer kt(ctx, "c1_allreduce_f32");
  const int gb = (int)std::min<int64_t>(8192, (a->n + 255) / 256);
  hipLaunchKernelGGL(c1_narrow, dim3(gb), dim3(256), 0, ctx->stream, a->buf_d, a->wire_d, a->n);
  if (comm) {
    int r = g_rccl.AllReduce(a->wire_d, a->wire_d, (size_t)a->n, kNcclFloat32, kNcclSum, comm, ctx->stream);
    if (r) return rccl_fail("ncclAllReduce", r);
  }
  hipLaunchKernelGGL(c1_widen, dim3(gb), dim3(256), 0, ctx->stream, a->wire_d, a->buf_d, a->n);
  HIPCHK(hipGetLastError());
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
// K4: device M-step
extern "C" int khg_model_set_weights(khg_ctx* ctx, khg_model* m, const float* weights) {
  if (!ctx || !m || !weights) return khg_set_error(KHG_E_ARG, "khg_model_set_weights: bad arguments");
  if (!m->weights_d) { int rc = dev_alloc(&m->weights_d, (size_t)m->sumG); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(m->weights_d, weights, sizeof(float) * (size_t)m->sumG, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  m->has_weights = true;
  return KHG_OK;
}
extern "C" int khg_model_num_gauss(const khg_model* m, int64_t* total, int32_t* gauss_off) {
  if (!m) return khg_set_error(KHG_E_ARG, "khg_model_num_gauss: model is NULL");
  if (total) *total = m->sumG;
  if (gauss_off) std::memcpy(gauss_off, m->gauss_off.data(), sizeof(int32_t) * ((size_t)m->P + 1));
  return KHG_OK;
}
extern "C" int khg_model_download(khg_ctx* ctx, const khg_model* m, float* weights, float* gconsts, float* miv, float* iv) {
  if (!ctx || !m) return khg_set_error(KHG_E_ARG, "khg_model_download: bad arguments");
  if (weights && !m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_download: the model has no weights (khg_model_set_weights)");
  const size_t G = (size_t)m->sumG, n = G * m->D;
  if (weights) HIPCHK(hipMemcpyAsync(weights, m->weights_d, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream));
  if (gconsts) HIPCHK(hipMemcpyAsync(gconsts, m->gconsts_d, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream));
  if (miv) HIPCHK(hipMemcpyAsync(miv, m->miv_d, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (iv) HIPCHK(hipMemcpyAsync(iv, m->iv_d, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

// The device M-step in two halves, so that it can be SHARDED over ranks by pdf range:
//   rows:    k4_mle_update on pdfs [p0, p0 + np): their parameter rows are rewritten in place (old layout), one K4Res per pdf;
//   finish:  totals in pdf order, compaction when some pdf lost Gaussians, the K1 / K3 images -- on the complete rows + results.
static int mle_update_rows(khg_ctx* ctx, khg_model* m, const khg_accs* acc, const khg_mle_options* o, uint16_t flags, int p0, int np) {
  if (!ctx || !m || !acc || !o) return khg_set_error(KHG_E_ARG, "khg_model_mle_update: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_mle_update: the model has no weights (khg_model_set_weights)");
  if (acc->D != m->D || acc->sumG != m->sumG)
    return khg_set_error(KHG_E_RUNTIME, "khg_model_mle_update: accumulator / model dimensions do not match");
  if (flags & ~0x7) return khg_set_error(KHG_E_RUNTIME, "Flags in argument do not match the active accumulators");   // mle-diag-gmm.cc:252
  if (p0 < 0 || np < 0 || p0 + np > m->P) return khg_set_error(KHG_E_ARG, "khg_model_mle_update: pdf range outside the model");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  const int P = m->P, D = m->D;
  int maxG = 0;
  for (int p = 0; p < P; ++p) maxG = std::max(maxG, m->gauss_off[p + 1] - m->gauss_off[p]);
  const size_t lds = sizeof(double) * (256 + (size_t)maxG) + sizeof(float) * 5 * (size_t)maxG;
  if (lds > 60 * 1024) return khg_set_error(KHG_E_UNSUPPORTED, "khg_model_mle_update: more than ~2000 Gaussians in one pdf");
  if (!m->k4_res_d || m->k4_res_P != P) {
    DEVFREE(m->k4_res_d);
    int rc = dev_alloc(&m->k4_res_d, (size_t)P);
    if (rc) return rc;
    m->k4_res_P = P;
  }
  K4Args a;
  a.gauss_off = m->gauss_off_d; a.D = D;
  a.occ = acc->occ(); a.macc = acc->mean(); a.vacc = acc->var();
  a.w = m->weights_d; a.gc = m->gconsts_d; a.miv = m->miv_d; a.iv = m->iv_d;
  a.res = m->k4_res_d;
  a.min_w = o->min_gaussian_weight; a.min_occ = o->min_gaussian_occupancy; a.min_var = o->min_variance;
  double* floor_d = nullptr;
  a.var_floor = nullptr;
  if (o->variance_floor_vector) {
    std::vector<double> fv(o->variance_floor_vector, o->variance_floor_vector + D);
    int rcf = dev_upload(ctx, &floor_d, fv);
    if (!rcf) { hipError_t ef = hipStreamSynchronize(ctx->stream); if (ef != hipSuccess) rcf = khg_set_error(KHG_E_HIP, hipGetErrorString(ef)); }
    if (rcf) { DEVFREE(floor_d); return rcf; }
    a.var_floor = floor_d;
  }
  a.remove_low = o->remove_low_count_gaussians; a.flags = flags; a.pdf0 = p0;
  if (np > 0) {
    KernelTimer kt(ctx, "k4_mle_update");
    hipLaunchKernelGGL(k4_mle_update, dim3(np), dim3(256), lds, ctx->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && floor_d) e = hipStreamSynchronize(ctx->stream);
  DEVFREE(floor_d);
  if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  return KHG_OK;
}
static int mle_update_finish(khg_ctx* ctx, khg_model* m, float* objf_change, float* count, int32_t* floored_elems, int32_t* floored_gauss,
                             int32_t* removed) {
  if (!ctx || !m || !m->k4_res_d || m->k4_res_P != m->P) return khg_set_error(KHG_E_ARG, "khg_model_mle_update_finish: no update in progress");
  const int P = m->P, D = m->D;
  std::vector<K4Res> res((size_t)P);
  hipError_t e = hipMemcpyAsync(res.data(), m->k4_res_d, sizeof(K4Res) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  int rc = KHG_OK;
  // totals in pdf order, float, as MleAmDiagGmmUpdate adds them (csrc/mle-am-diag-gmm.cc:177-193)
  float tot_obj = 0.0f, tot_count = 0.0f;
  int tfe = 0, tfg = 0, trm = 0;
  std::vector<int32_t> new_off((size_t)P + 1);
  int out = 0;
  for (int p = 0; p < P; ++p) {
    const K4Res& r = res[(size_t)p];
    if (r.bad) return khg_set_error(KHG_E_RUNTIME, "pdf " + std::to_string(p) + ": not a number in gconst computation");
    tot_obj += r.obj_change; tot_count += r.count; tfe += r.floored_elems; tfg += r.floored_gauss; trm += r.removed;
    new_off[(size_t)p] = out;
    out += r.newG;
  }
  new_off[(size_t)P] = out;
  if (trm > 0) {
    // some pdf shrank: move every pdf's rows to the new offsets in fresh arrays
    int32_t* new_off_d = nullptr;
    float *w2 = nullptr, *gc2 = nullptr, *miv2 = nullptr, *iv2 = nullptr;
    rc = dev_upload(ctx, &new_off_d, new_off);
    if (!rc) rc = dev_alloc(&w2, (size_t)out);
    if (!rc) rc = dev_alloc(&gc2, (size_t)out);
    if (!rc) rc = dev_alloc(&miv2, (size_t)out * D);
    if (!rc) rc = dev_alloc(&iv2, (size_t)out * D);
    if (!rc) {
      KernelTimer kt(ctx, "k4_compact");
      hipLaunchKernelGGL(k4_compact, dim3(P), dim3(256), 0, ctx->stream, m->gauss_off_d, new_off_d, D, m->weights_d, m->gconsts_d,
                         m->miv_d, m->iv_d, w2, gc2, miv2, iv2);
    }
    if (!rc) {
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    }
    DEVFREE(new_off_d);
    if (rc) { DEVFREE(w2); DEVFREE(gc2); DEVFREE(miv2); DEVFREE(iv2); return rc; }
    DEVFREE(m->weights_d); DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d);
    m->weights_d = w2; m->gconsts_d = gc2; m->miv_d = miv2; m->iv_d = iv2;
    m->gauss_off = new_off;
    m->sumG = out;
  }
  rc = model_pack(ctx, m);   // new K1 tile image + -0.5*inv_vars from the updated parameters
  if (rc) return rc;
  if (objf_change) *objf_change = tot_obj;
  if (count) *count = tot_count;
  if (floored_elems) *floored_elems = tfe;
  if (floored_gauss) *floored_gauss = tfg;
  if (removed) *removed = trm;
  return KHG_OK;
}
extern "C" int khg_model_mle_update(khg_ctx* ctx, khg_model* m, const khg_accs* acc, const khg_mle_options* o, uint16_t flags,
                                    float* objf_change, float* count, int32_t* floored_elems, int32_t* floored_gauss,
                                    int32_t* removed) {
  int rc = mle_update_rows(ctx, m, acc, o, flags, 0, m ? m->P : 0);
  if (rc) return rc;
  return mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
}
extern "C" int khg_model_mle_update_range(khg_ctx* ctx, khg_model* m, const khg_accs* acc, const khg_mle_options* o, uint16_t flags,
                                          int32_t first_pdf, int32_t n_pdf) {
  return mle_update_rows(ctx, m, acc, o, flags, first_pdf, n_pdf);
}
extern "C" int khg_model_mle_update_finish(khg_ctx* ctx, khg_model* m, float* objf_change, float* count, int32_t* floored_elems,
                                           int32_t* floored_gauss, int32_t* removed) {
  return mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
}
// the rows an update of pdfs [first_pdf, first_pdf + n_pdf) rewrote + its per-pdf results (32 bytes each), to / from the host: the
// exchange step of the sharded M-step for callers whose ranks cannot share device buffers (the tests' gloo ranks on one GPU)
static int mle_rows_copy(khg_ctx* ctx, khg_model* m, int p0, int np, float* w, float* gc, float* miv, float* iv, void* res, bool up) {
  if (!ctx || !m || p0 < 0 || np < 0 || p0 + np > m->P || !m->k4_res_d || m->k4_res_P != m->P)
    return khg_set_error(KHG_E_ARG, "khg_model_mle_rows: bad arguments (or no update in progress)");
  const size_t g0 = (size_t)m->gauss_off[p0], ng = (size_t)m->gauss_off[p0 + np] - g0, D = (size_t)m->D;
  auto cp = [&](float* host, float* dev, size_t n) -> hipError_t {
    if (!host || n == 0) return hipSuccess;
    return up ? hipMemcpyAsync(dev, host, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream)
              : hipMemcpyAsync(host, dev, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  };
  HIPCHK(cp(w, m->weights_d + g0, ng));
  HIPCHK(cp(gc, m->gconsts_d + g0, ng));
  HIPCHK(cp(miv, m->miv_d + g0 * D, ng * D));
  HIPCHK(cp(iv, m->iv_d + g0 * D, ng * D));
  if (res && np > 0)
    HIPCHK(up ? hipMemcpyAsync(m->k4_res_d + p0, res, sizeof(K4Res) * (size_t)np, hipMemcpyHostToDevice, ctx->stream)
              : hipMemcpyAsync(res, m->k4_res_d + p0, sizeof(K4Res) * (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_model_mle_rows_download(khg_ctx* ctx, khg_model* m, int32_t first_pdf, int32_t n_pdf, float* weights, float* gconsts,
                                           float* means_invvars, float* inv_vars, void* results) {
  return mle_rows_copy(ctx, m, first_pdf, n_pdf, weights, gconsts, means_invvars, inv_vars, results, false);
}
extern "C" int khg_model_mle_rows_upload(khg_ctx* ctx, khg_model* m, int32_t first_pdf, int32_t n_pdf, const float* weights,
                                         const float* gconsts, const float* means_invvars, const float* inv_vars, const void* results) {
  return mle_rows_copy(ctx, m, first_pdf, n_pdf, const_cast<float*>(weights), const_cast<float*>(gconsts), const_cast<float*>(means_invvars),
                       const_cast<float*>(inv_vars), const_cast<void*>(results), true);
}
// SURVEY.md 8f-3 as written: the block is REDUCED by pdf range to its owner (rank r owns pdfs [P r / N, P (r + 1) / N)) instead of
// all-reduced, every rank updates its own pdfs, the updated rows and per-pdf results are broadcast from their owners, and every rank
// finishes (compaction, images) on the complete model: (N - 1) / N x (207 + 105) MB per rank on the wire instead of
// 2 (N - 1) / N x 207 MB at 5000 x 64 x 40.  `acc` holds this rank's LOCAL sums (no khg_accs_allreduce before); on return its
// occupancies are summed over the ranks, its mean / variance rows are complete only for the rank's own pdfs, and its transition counts
// and scalars are untouched (khg_accs_allreduce_range with first_pdf < 0 sums those).
extern "C" int khg_model_mle_update_sharded(khg_ctx* ctx, khg_model* m, khg_accs* acc, const khg_mle_options* o, uint16_t flags,
                                            void* comm, int32_t nranks, int32_t rank, float* objf_change, float* count,
                                            int32_t* floored_elems, int32_t* floored_gauss, int32_t* removed) {
  if (!ctx || !m || !acc || !o || nranks < 1 || rank < 0 || rank >= nranks) return khg_set_error(KHG_E_ARG, "khg_model_mle_update_sharded: bad arguments");
  if (!comm) {                      // (a one-rank communicator still goes through RCCL: reductions and broadcasts to itself)
    int rc = mle_update_rows(ctx, m, acc, o, flags, 0, m->P);
    return rc ? rc : mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
  }
  int rc = rccl_bind();
  if (rc) return rc;
  if (!g_rccl.Reduce || !g_rccl.Broadcast) return khg_set_error(KHG_E_UNSUPPORTED, "RCCL library lacks ncclReduce / ncclBroadcast");
  if (acc->D != m->D || acc->sumG != m->sumG) return khg_set_error(KHG_E_RUNTIME, "khg_model_mle_update_sharded: accumulator / model dimensions do not match");
  const int P = m->P;
  const int64_t D = m->D;
  auto range = [&](int r, int* p0, int* np) { *p0 = (int)((int64_t)P * r / nranks); *np = (int)((int64_t)P * (r + 1) / nranks) - *p0; };
  {
    KernelTimer kt(ctx, "c1_reduce_by_pdf_range");
    int r = g_rccl.GroupStart();
    // occupancies: all of them to everybody (1 / (2 D + 1) of the block; the mixing-up targets need every pdf's) ...
    if (!r && acc->sumG > 0) r = g_rccl.AllReduce(acc->occ(), acc->occ(), (size_t)acc->sumG, kNcclFloat64, kNcclSum, comm, ctx->stream);
    for (int o2 = 0; o2 < nranks && !r; ++o2) {      // ... first- and second-order sums: each pdf range to its owner only
      int p0, np;
      range(o2, &p0, &np);
      const int64_t g0 = m->gauss_off[p0], ng = m->gauss_off[p0 + np] - g0;
      if (ng == 0) continue;
      r = g_rccl.Reduce(acc->mean() + g0 * D, acc->mean() + g0 * D, (size_t)(ng * D), kNcclFloat64, kNcclSum, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Reduce(acc->var() + g0 * D, acc->var() + g0 * D, (size_t)(ng * D), kNcclFloat64, kNcclSum, o2, comm, ctx->stream);
    }
    const int r2 = g_rccl.GroupEnd();
    if (r || r2) return rccl_fail("ncclReduce (sharded M-step)", r ? r : r2);
  }
  int p0, np;
  range(rank, &p0, &np);
  rc = mle_update_rows(ctx, m, acc, o, flags, p0, np);
  if (rc) return rc;
  {
    KernelTimer kt(ctx, "c1_broadcast_rows");
    int r = g_rccl.GroupStart();
    for (int o2 = 0; o2 < nranks && !r; ++o2) {
      int q0, nq;
      range(o2, &q0, &nq);
      const int64_t g0 = m->gauss_off[q0], ng = m->gauss_off[q0 + nq] - g0;
      if (nq > 0) r = g_rccl.Broadcast(m->k4_res_d + q0, m->k4_res_d + q0, sizeof(K4Res) * (size_t)nq, kNcclInt8, o2, comm, ctx->stream);
      if (ng == 0) continue;
      if (!r) r = g_rccl.Broadcast(m->weights_d + g0, m->weights_d + g0, (size_t)ng, kNcclFloat32, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Broadcast(m->gconsts_d + g0, m->gconsts_d + g0, (size_t)ng, kNcclFloat32, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Broadcast(m->miv_d + g0 * D, m->miv_d + g0 * D, (size_t)(ng * D), kNcclFloat32, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Broadcast(m->iv_d + g0 * D, m->iv_d + g0 * D, (size_t)(ng * D), kNcclFloat32, o2, comm, ctx->stream);
    }
    const int r2 = g_rccl.GroupEnd();
    if (r || r2) return rccl_fail("ncclBroadcast (sharded M-step)", r ? r : r2);
  }
  return mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
}

extern "C" int khg_model_split(khg_ctx* ctx, khg_model* m, const int32_t* targets, float perturb, const float* randn, int64_t n_randn) {
  if (!ctx || !m || !targets) return khg_set_error(KHG_E_ARG, "khg_model_split: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_split: the model has no weights (khg_model_set_weights)");
  const int P = m->P, D = m->D;
  std::vector<int32_t> new_off((size_t)P + 1, 0);
  std::vector<int64_t> rand_off((size_t)P + 1, 0);
  for (int p = 0; p < P; ++p) {
    const int cur = m->gauss_off[p + 1] - m->gauss_off[p];
    if (cur == 0 && targets[p] > 0) return khg_set_error(KHG_E_RUNTIME, "khg_model_split: pdf " + std::to_string(p) + " has no component to split");
    if (targets[p] < cur)   // csrc/diag-gmm.cc:782-786
      return khg_set_error(KHG_E_RUNTIME, "Cannot split from " + std::to_string(cur) + " to " + std::to_string(targets[p]) + " components");
    new_off[(size_t)p + 1] = new_off[(size_t)p] + targets[p];
    rand_off[(size_t)p + 1] = rand_off[(size_t)p] + (targets[p] - cur);
  }
  const int64_t nnew = rand_off[(size_t)P], out = new_off[(size_t)P];
  if (nnew == 0) return KHG_OK;
  if (!randn) return khg_set_error(KHG_E_ARG, "khg_model_split: randn_h is NULL");
  if (n_randn < nnew * D)
    return khg_set_error(KHG_E_ARG, "khg_model_split: randn_h holds " + std::to_string(n_randn) + " deviates, " + std::to_string(nnew * D) + " are needed (new components x dim)");
  std::vector<float> rv(randn, randn + (size_t)nnew * D);
  int32_t *new_off_d = nullptr, *bad_d = nullptr;
  int64_t* rand_off_d = nullptr;
  float *rand_d = nullptr, *w2 = nullptr, *gc2 = nullptr, *miv2 = nullptr, *iv2 = nullptr;
  int rc = dev_upload(ctx, &new_off_d, new_off);
  if (!rc) rc = dev_upload(ctx, &rand_off_d, rand_off);
  if (!rc) rc = dev_upload(ctx, &rand_d, rv);
  if (!rc) rc = dev_alloc(&bad_d, 1);
  if (!rc) rc = dev_alloc(&w2, (size_t)out);
  if (!rc) rc = dev_alloc(&gc2, (size_t)out);
  if (!rc) rc = dev_alloc(&miv2, (size_t)out * D);
  if (!rc) rc = dev_alloc(&iv2, (size_t)out * D);
  int32_t bad = 0;
  if (!rc) {
    hipError_t e = hipMemsetAsync(bad_d, 0, sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) {
      KernelTimer kt(ctx, "k4_split");
      hipLaunchKernelGGL(k4_split, dim3(P), dim3(256), 0, ctx->stream, m->gauss_off_d, new_off_d, D, m->weights_d, m->miv_d, m->iv_d, w2, gc2,
                         miv2, iv2, rand_d, rand_off_d, perturb, bad_d);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  DEVFREE(new_off_d); DEVFREE(rand_off_d); DEVFREE(rand_d); DEVFREE(bad_d);
  if (!rc && bad) rc = khg_set_error(KHG_E_RUNTIME, "khg_model_split: not a number in gconst computation");
  if (rc) { DEVFREE(w2); DEVFREE(gc2); DEVFREE(miv2); DEVFREE(iv2); return rc; }
  DEVFREE(m->weights_d); DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d);
  m->weights_d = w2; m->gconsts_d = gc2; m->miv_d = miv2; m->iv_d = iv2;
  m->gauss_off = new_off;
  m->sumG = out;
  return model_pack(ctx, m);
}

// AmDiagGmm::MergeByCount's per-pdf DiagGmm::Merge (csrc/am-diag-gmm.cc:91-108, csrc/diag-gmm.cc:557-759) on the device model:
// pdf p keeps targets[p] components (1 <= targets[p] <= its count).  Nothing crosses PCIe but the offsets.
extern "C" int khg_model_merge(khg_ctx* ctx, khg_model* m, const int32_t* targets) {
  if (!ctx || !m || !targets) return khg_set_error(KHG_E_ARG, "khg_model_merge: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_merge: the model has no weights (khg_model_set_weights)");
  const int P = m->P, D = m->D;
  std::vector<int32_t> new_off((size_t)P + 1, 0);
  std::vector<int64_t> delta_off((size_t)P + 1, 0);
  bool any = false;
  for (int p = 0; p < P; ++p) {
    const int cur = m->gauss_off[p + 1] - m->gauss_off[p];
    if (targets[p] <= 0 || cur < targets[p])   // csrc/diag-gmm.cc:558-562
      return khg_set_error(KHG_E_RUNTIME, "Invalid argument for target number of Gaussians (=" + std::to_string(targets[p]) + "), #Gauss = " + std::to_string(cur));
    new_off[(size_t)p + 1] = new_off[(size_t)p] + targets[p];
    const bool greedy = targets[p] < cur && targets[p] > 1;
    delta_off[(size_t)p + 1] = delta_off[(size_t)p] + (greedy ? (int64_t)cur * cur : 0);
    any = any || targets[p] < cur;
  }
  if (!any) return KHG_OK;
  const int64_t out = new_off[(size_t)P], old = m->sumG;
  K4MergeArgs a{};
  int32_t *new_off_d = nullptr, *idx_d = nullptr, *bad_d = nullptr;
  int64_t* delta_off_d = nullptr;
  float *scratch = nullptr, *delta_d = nullptr, *w2 = nullptr, *gc2 = nullptr, *miv2 = nullptr, *iv2 = nullptr;
  int rc = dev_upload(ctx, &new_off_d, new_off);
  if (!rc) rc = dev_upload(ctx, &delta_off_d, delta_off);
  if (!rc) rc = dev_alloc(&idx_d, (size_t)2 * old);
  if (!rc) rc = dev_alloc(&bad_d, 1);
  if (!rc) rc = dev_alloc(&scratch, (size_t)old * (2 + 4 * (size_t)D));
  if (!rc) rc = dev_alloc(&delta_d, (size_t)std::max<int64_t>(1, delta_off[(size_t)P]));
  if (!rc) rc = dev_alloc(&w2, (size_t)out);
  if (!rc) rc = dev_alloc(&gc2, (size_t)out);
  if (!rc) rc = dev_alloc(&miv2, (size_t)out * D);
  if (!rc) rc = dev_alloc(&iv2, (size_t)out * D);
  int32_t bad = 0;
  if (!rc) {
    a.old_off = m->gauss_off_d; a.new_off = new_off_d; a.D = D;
    a.w = m->weights_d; a.gc = m->gconsts_d; a.miv = m->miv_d; a.iv = m->iv_d;
    a.w2 = w2; a.gc2 = gc2; a.miv2 = miv2; a.iv2 = iv2;
    a.wk = scratch; a.logdet = scratch + old;
    a.mean = scratch + 2 * old; a.m2 = a.mean + old * D; a.mivk = a.m2 + old * D; a.ivk = a.mivk + old * D;
    a.gone = idx_d; a.keep = idx_d + old; a.delta = delta_d; a.delta_off = delta_off_d; a.bad = bad_d;
    hipError_t e = hipMemsetAsync(bad_d, 0, sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) {
      KernelTimer kt(ctx, "k4_merge");
      hipLaunchKernelGGL(k4_merge, dim3(P), dim3(256), 0, ctx->stream, a);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  DEVFREE(new_off_d); DEVFREE(delta_off_d); DEVFREE(idx_d); DEVFREE(bad_d); DEVFREE(scratch); DEVFREE(delta_d);
  if (!rc && (bad & 2)) rc = khg_set_error(KHG_E_RUNTIME, "khg_model_merge: no pair of components left to merge (max_i != max_j && max_i != -1 && max_j != -1)");
  if (!rc && (bad & 1)) rc = khg_set_error(KHG_E_RUNTIME, "khg_model_merge: not a number in gconst computation");
  if (rc) { DEVFREE(w2); DEVFREE(gc2); DEVFREE(miv2); DEVFREE(iv2); return rc; }
  DEVFREE(m->weights_d); DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d);
  m->weights_d = w2; m->gconsts_d = gc2; m->miv_d = miv2; m->iv_d = iv2;
  m->gauss_off = new_off;
  m->sumG = out;
  return model_pack(ctx, m);
}

// After khg_model_mle_update removed Gaussians the accumulator block is laid out for fewer rows.
extern "C" int khg_accs_relayout(khg_ctx* ctx, khg_accs* a, const khg_model* m) {
  if (!ctx || !a || !m) return khg_set_error(KHG_E_ARG, "khg_accs_relayout: bad arguments");
  if (m->D != a->D) return khg_set_error(KHG_E_RUNTIME, "khg_accs_relayout: dimension mismatch");
  const int64_t n = m->sumG * (1 + 2 * (int64_t)a->D) + a->num_tids + 1 + 8;
  if (n > a->cap) {
    DEVFREE(a->buf_d);
    int rc = dev_alloc(&a->buf_d, (size_t)n);
    if (rc) return rc;
    a->cap = n;
  }
  a->sumG = m->sumG; a->n = n;
  return khg_accs_zero(ctx, a);
}
extern "C" int khg_accs_download_trans(khg_ctx* ctx, const khg_accs* a, double* trans, double* scalars) {
  if (!ctx || !a) return khg_set_error(KHG_E_ARG, "khg_accs_download_trans: bad arguments");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  if (trans) HIPCHK(hipMemcpyAsync(trans, a->trans(), sizeof(double) * ((size_t)a->num_tids + 1), hipMemcpyDeviceToHost, ctx->stream));
  if (scalars) HIPCHK(hipMemcpyAsync(scalars, a->scalars(), sizeof(double) * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_accs_download_range(khg_ctx* ctx, const khg_accs* a, int64_t first, int64_t count, double* dst) {
  if (!ctx || !a || !dst || first < 0 || count < 0 || first + count > a->n) return khg_set_error(KHG_E_ARG, "khg_accs_download_range: bad arguments");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  if (count) HIPCHK(hipMemcpyAsync(dst, a->buf_d + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

extern "C" int khg_model_scale_weights(khg_ctx* ctx, khg_model* m, int32_t n, const int32_t* pdfs, float scale) {
  if (!ctx || !m || n < 0 || (n > 0 && !pdfs)) return khg_set_error(KHG_E_ARG, "khg_model_scale_weights: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_scale_weights: the model has no weights (khg_model_set_weights)");
  if (n == 0) return KHG_OK;
  std::vector<int32_t> v(pdfs, pdfs + n);
  for (int32_t p : v)
    if (p < 0 || p >= m->P) return khg_set_error(KHG_E_ARG, "khg_model_scale_weights: pdf-id out of range");
  int32_t *pdfs_d = nullptr, *bad_d = nullptr;
  int rc = dev_upload(ctx, &pdfs_d, v);
  if (!rc) rc = dev_alloc(&bad_d, 1);
  int32_t bad = 0;
  if (!rc) {
    hipError_t e = hipMemsetAsync(bad_d, 0, sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k4_scale_weights, dim3(n), dim3(64), 0, ctx->stream, pdfs_d, m->gauss_off_d, m->D, scale, m->weights_d,
                         m->gconsts_d, m->miv_d, m->iv_d, bad_d);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  DEVFREE(pdfs_d); DEVFREE(bad_d);
  if (rc) return rc;
  if (bad) return khg_set_error(KHG_E_RUNTIME, "khg_model_scale_weights: not a number in gconst computation");
  return model_pack(ctx, m);
}
