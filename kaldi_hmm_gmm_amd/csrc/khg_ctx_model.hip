// kaldi_hmm_gmm_amd/csrc/khg_ctx_model.hip -- C-ABI (include/khg_hip.h): errors, the per-device context and its options, the
// acoustic-model handle (row-major parameters + the fp32 K1 tile image, packed on the device) and the transition table.  gfx950 only.
#include "khg_internal.hpp"

#include <mutex>
#include "khg_k1_loglikes.hip.inc"     // the tile geometry (khg_row_floats / khg_tile_floats) k0_pack_tiles shares with K1

// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
extern "C" const char* khg_last_error(void) { return g_err.c_str(); }
int khg_set_error(int code, const std::string& msg) { g_err = msg; return code; }  // shared with khg_host.cpp
extern "C" int khg_version(void) { return 100; }

// ---- the small-set scratch arena (KhgArena, khg_internal.hpp) ----------------------------------------------------------------
static std::mutex g_ctx_mu;
static std::vector<khg_ctx*> g_ctxs;          // live contexts: khg_arena_release finds the owner of a pointer here
static constexpr size_t KHG_ARENA_BYTES = size_t(24) << 20;
void* arena_alloc(khg_ctx* ctx, size_t bytes) {
  KhgArena& a = ctx->arena;
  if (!a.dev) {
    if (a.cap == SIZE_MAX) return nullptr;                       // could not be made: do not try again
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, KHG_ARENA_BYTES) != hipSuccess || hipHostMalloc(&h, KHG_ARENA_BYTES, hipHostMallocDefault) != hipSuccess) {
      if (d) (void)hipFree(d);
      (void)hipGetLastError();
      a.cap = SIZE_MAX;
      return nullptr;
    }
    a.dev = static_cast<char*>(d); a.host = static_cast<char*>(h); a.cap = KHG_ARENA_BYTES; a.top = a.base = 256;      // (no allocation shares the block's own address: a handle destroyed after its context must not hipFree it)
  }
  bytes = (std::max<size_t>(bytes, 1) + 255) & ~size_t(255);
  if (a.top + bytes > a.cap) return nullptr;
  void* p = a.dev + a.top;
  a.blocks.emplace_back(a.top, a.top + bytes);
  a.top += bytes;
  return p;
}
void arena_mark_dirty(khg_ctx* ctx, const void* dev_ptr, size_t bytes) {
  KhgArena& a = ctx->arena;
  const size_t lo = (size_t)((const char*)dev_ptr - a.dev), hi = lo + bytes;
  // allocations are 256-byte aligned: a range that starts within 256 bytes of the previous one's end belongs to the next allocation
  // (nothing un-staged lies between them) and joins it; anything else -- an allocation a kernel may already have written -- is left out
  if (!a.dirty.empty() && lo >= a.dirty.back().second && lo - a.dirty.back().second < 256) a.dirty.back().second = hi;
  else a.dirty.emplace_back(lo, hi);
}
int arena_flush(khg_ctx* ctx) {
  KhgArena& a = ctx->arena;
  if (a.dirty.empty()) return KHG_OK;
  for (const auto& r : a.dirty)
    HIPCHK(hipMemcpyAsync(a.dev + r.first, a.host + r.first, r.second - r.first, hipMemcpyHostToDevice, ctx->stream));
  a.dirty.clear();
  return KHG_OK;
}
bool khg_ctx_alive(const khg_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  return std::find(g_ctxs.begin(), g_ctxs.end(), ctx) != g_ctxs.end();
}
// Scratch blocks of destroyed contexts that utterance sets still point into (a handle may be destroyed after its context): kept until
// the last of those sets lets go, so that no later context's block can take the same addresses while stale pointers exist.
static std::vector<KhgArena> g_orphans;
bool khg_arena_release(void* p) {
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  for (size_t z = 0; z < g_orphans.size(); ++z) {
    KhgArena& a = g_orphans[z];
    if (!a.owns(p)) continue;
    const size_t off = (size_t)((char*)p - a.dev);
    for (size_t i = a.blocks.size(); i-- > 0;)
      if (a.blocks[i].first == off) { a.blocks.erase(a.blocks.begin() + (long)i); break; }
    if (a.blocks.empty()) { (void)hipFree(a.dev); (void)hipHostFree(a.host); g_orphans.erase(g_orphans.begin() + (long)z); }
    return true;
  }
  for (khg_ctx* c : g_ctxs) {
    KhgArena& a = c->arena;
    if (!a.owns(p)) continue;
    // (the owner waited for its streams before releasing.)  Blocks are a stack: the top falls back to the end of the last live one
    const size_t off = (size_t)((char*)p - a.dev);
    for (size_t i = a.blocks.size(); i-- > 0;)
      if (a.blocks[i].first == off) { a.blocks.erase(a.blocks.begin() + (long)i); break; }
    a.top = a.blocks.empty() ? a.base : a.blocks.back().second;
    for (size_t i = a.dirty.size(); i-- > 0;) {                                      // staged bytes of dead allocations
      if (a.dirty[i].first >= a.top) a.dirty.erase(a.dirty.begin() + (long)i);
      else if (a.dirty[i].second > a.top) a.dirty[i].second = a.top;
    }
    return true;
  }
  return false;
}
void khg_dev_free(void* p) {
  if (p && !khg_arena_release(p)) (void)hipFree(p);
}

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
  if (!rc && hipHostMalloc(reinterpret_cast<void**>(&c->err_host), 256, hipHostMallocDefault) != hipSuccess) rc = khg_set_error(KHG_E_HIP, "khg_ctx_create: hipHostMalloc failed");
  if (rc) { delete c; return rc; }
  HIPCHK(hipMemsetAsync(c->err_flag_d, 0, sizeof(int32_t), c->stream));
  { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctxs.push_back(c); }
  *out = c;
  return KHG_OK;
}
extern "C" int khg_ctx_destroy(khg_ctx* c) {
  if (!c) return KHG_OK;
  (void)hipStreamSynchronize(c->stream);
  for (auto& s : c->sides) (void)hipStreamSynchronize(s);
  {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end());
    if (c->arena.dev && !c->arena.blocks.empty()) {        // small sets outlive the context: their scratch stays until they go (g_orphans)
      c->arena.dirty.clear();
      g_orphans.push_back(std::move(c->arena));
      c->arena = KhgArena();
    }
  }
  DEVFREE(c->err_flag_d); DEVFREE(c->dump_d);
  if (c->err_host) (void)hipHostFree(c->err_host);
  if (c->arena.dev) (void)hipFree(c->arena.dev);
  if (c->arena.host) (void)hipHostFree(c->arena.host);
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
  {KHG_K1_AUTO, KHG_K1_F16X2S}, {0, 4}, {0, 6}, {0, 1 << 20}, {-1, 1}, {0, 255}, {0, 1}, {0, 4}, {0, 3}, {0, 1}, {0, 2}, {0, 2}, {0, 2}, {0, 64}, {0, 1}, {0, 1}, {0, 1}};
extern "C" int khg_ctx_set_option(khg_ctx* c, int opt, int value) {
  if (!c || opt < 0 || opt >= KHG_OPT_COUNT) return khg_set_error(KHG_E_ARG, "khg_ctx_set_option: bad arguments");
  if (value < k_opt_range[opt].lo || value > k_opt_range[opt].hi || (opt == KHG_OPT_K1_FORM && value == 1))      // (1: the removed bf16x3 form)
    return khg_set_error(KHG_E_ARG, "khg_ctx_set_option: option " + std::to_string(opt) + " takes values " + std::to_string(k_opt_range[opt].lo) + " .. " + std::to_string(k_opt_range[opt].hi));
  c->opt[opt] = value;
  return KHG_OK;
}
extern "C" int khg_ctx_get_option(const khg_ctx* c, int opt, int* value) {
  if (c && value && (opt == KHG_INFO_SCRATCH_BYTES || opt == KHG_INFO_SCRATCH_BLOCKS)) {
    *value = opt == KHG_INFO_SCRATCH_BYTES ? (c->arena.dev ? (int)c->arena.top : 0) : (int)c->arena.blocks.size();
    return KHG_OK;
  }
  if (!c || !value || opt < 0 || opt >= KHG_OPT_COUNT) return khg_set_error(KHG_E_ARG, "khg_ctx_get_option: bad arguments");
  *value = c->opt[opt];
  return KHG_OK;
}
extern "C" int khg_ctx_set_k1_form(khg_ctx* c, int form) { return khg_ctx_set_option(c, KHG_OPT_K1_FORM, form); }
// Defaults from the environment, read ONCE per context (A/B runs of an unmodified caller): NAME=value, value an integer or one of the words
// listed.  Everything else goes through khg_ctx_set_option.
static void ctx_defaults_from_env(khg_ctx* c) {
  static const struct { const char* name; int opt; const char* words; } tab[] = {
    {"KHG_K1", KHG_OPT_K1_FORM, "auto=0,pdf=2,fp32=2,utt=3,f16x2=4,f16x2s=5"},
    {"KHG_K1_ORDER", KHG_OPT_K1_ORDER, "desc=0,none=1,asc=2,tiles=3,xcd=4"},
    {"KHG_K1_NF", KHG_OPT_K1_NF, ""}, {"KHG_K1P_TS", KHG_OPT_K1P_TS, ""}, {"KHG_K1_INTERLEAVE", KHG_OPT_K1_INTERLEAVE, ""},
    {"KHG_K1B_DBG", KHG_OPT_K1_DBG, ""}, {"KHG_K2_INORDER", KHG_OPT_K2_INORDER, ""}, {"KHG_K2_KS", KHG_OPT_K2_KS, ""},
    {"KHG_K2_SERIAL", KHG_OPT_K2_SERIAL, ""}, {"KHG_K2_PROF", KHG_OPT_K2_PROF, ""},
    {"KHG_K3_BUCKET", KHG_OPT_K3_BUCKET, "sort=0,atomic=1,count=2"}, {"KHG_K3_FORM", KHG_OPT_K3_FORM, "auto=0,block=1,valu=2"},
    {"KHG_K3_VALU", KHG_OPT_K3_FORM, "1=2"}, {"KHG_K3_PHASEB", KHG_OPT_K3_PHASE_B, "f64=0,f32=1,f16=2"},
    {"KHG_K3_NY", KHG_OPT_K3_NY, ""}, {"KHG_DEBUG", KHG_OPT_DEBUG, ""}, {"KHG_K3_PHASEA", KHG_OPT_K3_PHASE_A, "auto=0,f16=0,f32=1"},
    {"KHG_K2_SPLIT", KHG_OPT_K2_SPLIT, "on=0,off=1"}};
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
extern "C" int khg_ctx_sync(khg_ctx* c) {
  if (!c) return khg_set_error(KHG_E_ARG, "ctx is NULL");
  return check_err_flag(c, "khg_ctx_sync");   // synchronises the stream, then reports deferred kernel errors
}
// read-and-clear the device error word; maps bits to the reference's exceptions
int check_err_flag(khg_ctx* c, const char* where) {
  // every download of the library passes here first: uploads still staged in the arena's mirror reach the device before any
  // plain copy reads (or a later flush overwrites) the bytes they belong to
  { int rf = arena_flush(c); if (rf) return rf; }
  for (int i = 0; i < khg_ctx::NSIDE; ++i)
    if (c->side_dirty[i]) { HIPCHK(hipStreamSynchronize(c->sides[i])); c->side_dirty[i] = false; }
  HIPCHK(hipMemcpyAsync(c->err_host, c->err_flag_d, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->pageable_pending = false;
  const int32_t f = *c->err_host;
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

// -0.5 * inv_vars for K3 when no tile image is packed
__global__ void k0_nhalf(const float* __restrict__ iv, int64_t n, float* __restrict__ nhiv) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) nhiv[i] = -0.5f * iv[i];
}

// ------------------------------------------------------------------------------------------
// What the library derives from the parameters once per PARAMETER VERSION, in ONE pass over the model's rows (round 5; four passes --
// three k1h_absmax launches and k3_model_xbound, each with its own scratch, download and wait -- plus k1s_ubound before):
//   maxima[0 .. D)          max |means_invvars[.][d]|      } the column maxima behind the split K1 forms' scale exponents and
//   maxima[512 .. 512 + D)  max |inv_vars[.][d]|           } domain checks (k1h_absmax's rule: -inf skipped, NaN / inf -> +inf)
//   maxima[1024 .. 1024+D)  max_g |mean| + 8 sigma         the feature envelope K3's fp16 phase A scales by (k3_model_xbound's rule)
//   maxima[1536]            max |gconst|
//   ubound[first tile of p] log sum_g exp(gconst_g + 0.5 sum_d mi^2 / iv) + margin: the BAND form's fill (k1s_ubound's rule, fp64,
//                           every sum in a fixed order: the value does not depend on the launch)
// One workgroup walks pdfs p = block, block + grid, ...; a thread owns one dimension of one of the 256 / D rows of a pass (two
// dimensions when D > 256); a row's fp64 terms are summed by its first thread, a pdf's components by wave 0.
struct K0StatsArgs {
  const float *gconsts, *miv, *iv;
  const int32_t *gauss_off, *pdf_tile_off;     // pdf_tile_off / ubound may be NULL (no tile image: D > 80)
  int P, D;
  uint32_t* maxima;                            // [1537], zeroed by the caller
  float* ubound;
};
__device__ __forceinline__ float k0_absmax_step(float m, float v) {
  if (v == -INFINITY) return m;
  return (fabsf(v) <= 3.0e38f) ? fmaxf(m, fabsf(v)) : INFINITY;
}
__global__ __launch_bounds__(256) void k0_model_stats(K0StatsArgs a) {
  constexpr int NPASS = 8;                   // passes of 256 / D rows whose terms are parked before one barrier and one round of row sums
  __shared__ double s_term[NPASS][256];
  __shared__ double s_c[1024];
  __shared__ unsigned s_bmax;                // max over the pdf's components of |gconst| + 0.5 sum mi^2 / iv (float bits): the magnitude K1's rounding scales with
  const int D = a.D, Dc = D < 256 ? D : 256, RP = 256 / Dc, t = threadIdx.x;
  const int r = t / Dc, d0 = t - r * Dc;
  const bool act = r < RP;
  float m_miv[2] = {0.0f, 0.0f}, m_iv[2] = {0.0f, 0.0f}, m_xb[2] = {0.0f, 0.0f}, m_gc = 0.0f;
  for (int p = blockIdx.x; p < a.P; p += gridDim.x) {
    const int g0 = a.gauss_off[p], g1 = a.gauss_off[p + 1];
    double lm = -INFINITY, ls = 0.0;             // wave 0's per-lane running (max, sum) over this pdf's components
    if (t == 0) s_bmax = 0u;
    __syncthreads();
    for (int gb = g0; gb < g1; gb += 1024) {     // <= 1024 components at a time through s_c
      const int ge = min(g1, gb + 1024);
      for (int g = gb; g < ge; g += RP * NPASS) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          double term = 0.0;
          const int row = g + ps * RP + r;
          if (act && row < ge) {
            int j = 0;
            for (int d = d0; d < D; d += Dc, ++j) {
              const float mi = a.miv[(size_t)row * D + d], v = a.iv[(size_t)row * D + d];
              m_miv[j] = k0_absmax_step(m_miv[j], mi);
              m_iv[j] = k0_absmax_step(m_iv[j], v);
              float b = fabsf(mi / v) + 8.0f * rsqrtf(v);
              if (!(b < 3.0e38f)) b = INFINITY;
              m_xb[j] = fmaxf(m_xb[j], b);
              term += 0.5 * (double)mi * (double)mi / (double)v;
            }
            if (d0 == 0) m_gc = k0_absmax_step(m_gc, a.gconsts[row]);
          }
          s_term[ps][t] = term;
        }
        __syncthreads();
        for (int idx = t; idx < NPASS * RP; idx += 256) {      // one thread per parked row: pass idx / RP, row slot idx % RP, its Dc terms in order
          const int ps = idx / RP, rr = idx - ps * RP, row = g + ps * RP + rr;
          if (row < ge) {
            double c = 0.0;
            for (int k2 = 0; k2 < Dc; ++k2) c += s_term[ps][rr * Dc + k2];
            s_c[row - gb] = c + (double)a.gconsts[row];
            const float bs = (float)(c + fabs((double)a.gconsts[row]));
            if (bs > 0.0f && bs < 3.0e38f) atomicMax(&s_bmax, __float_as_uint(bs));
          }
        }
        __syncthreads();
      }
      if (t < 64) {
        for (int i = t; i < ge - gb; i += 64) {
          const double c = s_c[i];
          if (c > lm) { ls = ls * exp(lm - c) + 1.0; lm = c; }
          else if (c > -INFINITY) ls += exp(c - lm);
        }
      }
      __syncthreads();
    }
    if (t < 64 && a.ubound && a.pdf_tile_off) {
      for (int o = 32; o > 0; o >>= 1) {
        const double m2 = __shfl_xor(lm, o), s2 = __shfl_xor(ls, o);
        const double mm = fmax(lm, m2);
        if (mm > -INFINITY) ls = ls * exp(lm - mm) + s2 * exp(m2 - mm);
        lm = mm;
      }
      if (t == 0 && a.pdf_tile_off[p + 1] > a.pdf_tile_off[p]) {
        double v = lm + log(ls);
        if (!(v == v) || v == -INFINITY) v = 0.0;         // an all-dead pdf scores -inf everywhere (an error in K1 either way): any finite fill
        if (v > 3.0e38) v = 3.0e38;
        float vf = (float)v;
        if ((double)vf < v) vf = nextafterf(vf, INFINITY);        // rounded up: still a bound
        // + what K1's own rounding can add to a value near the bound: its error scales with the magnitude of the cancelling terms
        // (tests hold K1 to 1e-5 + 1e-6 B; 2^-20 B here), which for un-normalised features dwarfs the fixed margin
        a.ubound[a.pdf_tile_off[p]] = vf + 1.0e-3f + 1.0e-5f * fabsf(vf) + ldexpf(__uint_as_float(s_bmax), -20);
      }
    }
  }
  if (act) {
    int j = 0;
    for (int d = d0; d < D; d += Dc, ++j) {
      if (m_miv[j] > 0.0f) atomicMax(a.maxima + d, __float_as_uint(m_miv[j]));
      if (m_iv[j] > 0.0f) atomicMax(a.maxima + 512 + d, __float_as_uint(m_iv[j]));
      if (m_xb[j] > 0.0f) atomicMax(a.maxima + 1024 + d, __float_as_uint(m_xb[j]));
    }
    if (d0 == 0 && m_gc > 0.0f) atomicMax(a.maxima + 1536, __float_as_uint(m_gc));
  }
}

// The host side: runs once per parameter version (wmax empty = stale), fills wmax / gcmax (K1's column maxima), k3_xb (K3's feature
// envelope; the exponents derived from it stay in khg_k3.hip) and, where the model has a tile layout, ubound.
int model_stats(khg_ctx* ctx, khg_model* m) {
  if (!m->wmax.empty()) return KHG_OK;
  const int D = m->D, K = m->KS > 0 ? 16 * m->KS : 2 * D;
  if (!m->stats_d) { int rc = dev_alloc(&m->stats_d, 1537); if (rc) return rc; }
  if (m->KQ != 0 && (!m->ubound_d || m->ubound_tiles < m->ntiles)) {
    DEVFREE(m->ubound_d);
    int rc = dev_alloc(&m->ubound_d, (size_t)m->ntiles);
    if (rc) return rc;
    m->ubound_tiles = m->ntiles;
  }
  HIPCHK(hipMemsetAsync(m->stats_d, 0, 1537 * sizeof(uint32_t), ctx->stream));
  K0StatsArgs a{m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->KQ != 0 ? m->pdf_tile_off_d : nullptr, m->P, D, m->stats_d, m->KQ != 0 ? m->ubound_d : nullptr};
  {
    KernelTimer kt(ctx, "k0_model_stats");
    KHG_LAUNCH(ctx, k0_model_stats, dim3((unsigned)std::min(m->P, 2048)), dim3(256), 0, ctx->stream, a);
  }
  HIPCHK(hipGetLastError());
  std::vector<uint32_t> h(1537);
  HIPCHK(hipMemcpyAsync(h.data(), m->stats_d, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  auto f = [&](size_t i) { float v; memcpy(&v, &h[i], sizeof(float)); return v; };
  m->wmax.assign((size_t)std::max(K, 2 * D), 0.0f);
  for (int d = 0; d < D; ++d) { m->wmax[(size_t)2 * d] = f((size_t)d); m->wmax[(size_t)2 * d + 1] = 0.5f * f(512 + (size_t)d); }
  m->gcmax = f(1536);
  m->k3_xb_raw.assign((size_t)D, 0.0f);
  for (int d = 0; d < D; ++d) m->k3_xb_raw[(size_t)d] = f(1024 + (size_t)d);
  m->ubound_valid = m->KQ != 0;
  return KHG_OK;
}

// (Re)build everything derived from gauss_off + the row-major parameters in HBM: the tile offsets, the K1
// tile image and the -0.5*inv_vars copy K3 reads -- packed ON THE DEVICE (k0_pack_tiles), no host-side
// 113 MB image, no extra copies.  Used by khg_model_create and after the device M-step.
int model_pack(khg_ctx* ctx, khg_model* m) {
  const int P = m->P, D = m->D;
  ++m->version;
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
    m->wimgh_ex.clear(); m->wimgs_key.clear(); m->ubound_valid = false; m->wmax.clear(); m->k3_xb.clear();
    const int64_t n = m->sumG * D;
    KHG_LAUNCH(ctx, k0_nhalf, dim3((int)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, ctx->stream, m->iv_d, n, m->nhiv_d);
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
    if (m->KQ == 10) KHG_LAUNCH(ctx, k0_pack_tiles<10>, dim3(nt), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, tile_pdf_d, D, m->wimg_d, m->nhiv_d);
    else KHG_LAUNCH(ctx, k0_pack_tiles<20>, dim3(nt), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, tile_pdf_d, D, m->wimg_d, m->nhiv_d);
  }
  if (!rc) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // tile_pdf and the host offset vectors are free after this
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  return rc;
}

// live model handles by serial: an utterance set scored in the BAND form keeps (serial, version) of its model and khg_align asks here
// whether that model still exists before the repair launch dereferences it
static std::mutex g_model_mu;
static std::vector<khg_model*> g_models;
static uint64_t g_model_serial = 0;
khg_model* khg_model_lookup(uint64_t serial) {
  std::lock_guard<std::mutex> lk(g_model_mu);
  for (khg_model* m : g_models) if (m->serial == serial) return m;
  return nullptr;
}

extern "C" int khg_model_create(khg_ctx* ctx, int32_t P, int32_t D, const int32_t* gauss_off,
                                const float* gconsts, const float* miv, const float* iv, khg_model** out) {
  if (ctx_dead(ctx) || !out || P <= 0 || D <= 0 || !gauss_off || !gconsts || !miv || !iv)
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
  { std::lock_guard<std::mutex> lk(g_model_mu); m->serial = ++g_model_serial; g_models.push_back(m); }
  *out = m;
  return KHG_OK;
}
// Everything derived from the parameters and cached per parameter version (the split K1 forms' images and their keys, the BAND form's
// upper bounds, the column maxima behind the scale exponents, K3's phase-A scales) is dropped and the version moves on, exactly as
// after an in-place update: the next khg_loglikes / khg_acc_stats derive them again.
extern "C" int khg_model_invalidate(khg_model* m) {
  if (!m) return khg_set_error(KHG_E_ARG, "khg_model_invalidate: model is NULL");
  ++m->version;
  m->wimgh_ex.clear(); m->wimgs_key.clear(); m->ubound_valid = false; m->wmax.clear(); m->k3_xb.clear();
  return KHG_OK;
}
extern "C" int khg_model_destroy(khg_model* m) {
  if (!m) return KHG_OK;
  { std::lock_guard<std::mutex> lk(g_model_mu); g_models.erase(std::remove(g_models.begin(), g_models.end(), m), g_models.end()); }
  m->wimgh_sync.destroy(); m->wimgs_sync.destroy();
  DEVFREE(m->wimg_d); DEVFREE(m->wimgh_d); DEVFREE(m->wimgs_d); DEVFREE(m->ubound_d); DEVFREE(m->stats_d); DEVFREE(m->k3_ex_d); DEVFREE(m->tile_pdf_d); DEVFREE(m->k4_res_d); DEVFREE(m->pdf_tile_off_d); DEVFREE(m->gauss_off_d);
  DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d); DEVFREE(m->weights_d);
  delete m;
  return KHG_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int khg_tm_create(khg_ctx* ctx, int32_t num_tids, const int32_t* id2pdf, khg_tm** out) {
  if (ctx_dead(ctx) || !out || num_tids <= 0 || !id2pdf) return khg_set_error(KHG_E_ARG, "khg_tm_create: bad arguments");
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
