// kaldi_hmm_gmm_amd/csrc/khg_internal.hpp -- what the translation units of libkhg_hip.so share: the handles behind include/khg_hip.h
// (khg_ctx / khg_model / khg_tm / khg_utts / khg_accs), the error and allocation helpers and the few functions one unit calls in
// another.  The units, by handle:  khg_ctx_model.hip (context, model image, transition table), khg_utts.hip (utterance sets: features +
// graphs), khg_k1.hip (log-likelihoods), khg_k2.hip (Viterbi alignment), khg_k3.hip (accumulators + statistics), khg_c1.hip (RCCL
// exchange), khg_k4.hip (device M-step).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/khg_hip.h"
#include "khg_dev_types.hip.inc"

int khg_set_error(int code, const std::string& msg);      // khg_ctx_model.hip; shared with khg_host.cpp

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return khg_set_error(KHG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
  } while (0)

struct khg_timing { std::string name; hipEvent_t e0, e1; };
// Scratch of the SMALL utterance sets (one call of the reference's per-utterance API = one set of one utterance: create, K1 + K2 or
// K3, destroy): one device block mirrored by one pinned host block of the same size.  Allocation bumps a pointer (no hipMalloc /
// hipFree per call), an upload is a memcpy into the mirror at the allocation's own offset -- all uploads staged since the last
// launch go to the device in one copy per run of neighbouring allocations (arena_flush) --, a download lands in the mirror.  The block rewinds when its last live
// allocation is released.  Requests that do not fit fall back to hipMalloc.
struct KhgArena {
  char *dev = nullptr, *host = nullptr;
  size_t cap = 0, top = 0, base = 0;          // base: bytes reserved for the context itself (the error word)
  std::vector<std::pair<size_t, size_t>> blocks;  // live allocations [offset, end), ascending: a release pops the dead tail, so per-call
                                                  // sets made beside a longer-lived small set reuse the same bytes instead of filling the block
  std::vector<std::pair<size_t, size_t>> dirty;   // staged mirror ranges not yet copied to the device (neighbours merged)
  bool owns(const void* p) const { return dev && (const char*)p >= dev && (const char*)p < dev + cap; }
  char* mirror(const void* dev_ptr) const { return host + ((const char*)dev_ptr - dev); }
};
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
  int32_t* err_host = nullptr;      // 256 pinned bytes: [0] landing word of check_err_flag, [64..128) the 8 scalars of khg_accs_download_trans
  bool side_dirty[NSIDE] = {false, false, false, false};   // work was enqueued on the side stream since it was last waited for
  float* dump_d = nullptr;          // 256 floats nobody reads (K1 f16x2s: where the pipeline's first, empty value goes)
  KhgArena arena;
  bool pageable_pending = false;    // a hipMemcpyAsync from pageable host memory may still be reading its source (sync_pageable)
  bool timing = false;
  std::vector<khg_timing> timings;
  int opt[KHG_OPT_COUNT] = {};    // khg_ctx_set_option (KHG_OPT_*); the environment variables of include/khg_hip.h only seed the defaults, once, at khg_ctx_create
};
int arena_flush(khg_ctx* ctx);                             // khg_ctx_model.hip: staged uploads -> device, one copy on the context's stream
void* arena_alloc(khg_ctx* ctx, size_t bytes);             // 256-byte aligned; NULL when the block is full (or could not be made)
bool khg_arena_release(void* p);                           // true when p came from some context's arena
void khg_dev_free(void* p);                                // arena or hipFree
void arena_mark_dirty(khg_ctx* ctx, const void* dev_ptr, size_t bytes);
bool khg_ctx_alive(const khg_ctx* ctx);                    // khg_ctx_model.hip: the context has not been destroyed (handles may outlive it)
// First argument check of every entry point that takes a context: null, or destroyed while a caller (a host object's cached device
// handle, say) still held the pointer -> KHG_E_ARG instead of a use after free.
static inline bool ctx_dead(const khg_ctx* ctx) { return !ctx || !khg_ctx_alive(ctx); }
// A SMALL utterance set (scratch in its creator's arena) belongs to that context: staged uploads are flushed by launches on it and
// khg_utts_destroy waits on its streams only.  Entry points that launch on / copy from a set take this check (-> KHG_E_ARG).
int utts_foreign_ctx(const khg_ctx* ctx, const struct khg_utts* u, const char* where);     // khg_utts.hip
// Every kernel launch of the library goes through this: uploads staged in the arena since the last launch reach the device first.
#define KHG_LAUNCH(ctx_, ...) do { (void)arena_flush(ctx_); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
// scoped HIP-event pair around a kernel launch, on the launching stream (only when enabled); staged uploads go out first
struct KernelTimer {
  khg_ctx* c; size_t idx = 0; bool on; hipStream_t s;
  KernelTimer(khg_ctx* ctx, const char* name, hipStream_t st = nullptr) : c(ctx), on(ctx->timing), s(st ? st : ctx->stream) {
    (void)arena_flush(ctx);
    if (!on) return;
    khg_timing t; t.name = name;
    (void)hipEventCreate(&t.e0); (void)hipEventCreate(&t.e1);
    (void)hipEventRecord(t.e0, s);
    c->timings.push_back(t); idx = c->timings.size() - 1;
  }
  ~KernelTimer() { if (on) (void)hipEventRecord(c->timings[idx].e1, s); }
};

template <class T>
inline int dev_alloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T)));
  return KHG_OK;
}
template <class T>
inline int dev_upload(khg_ctx* ctx, T** p, const std::vector<T>& v) {
  int rc = dev_alloc(p, v.size());
  if (rc) return rc;
  if (!v.empty()) { HIPCHK(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream)); ctx->pageable_pending = true; }
  return KHG_OK;
}
// after uploads from host vectors that are about to go out of scope: waits only if a pageable copy was issued since the last wait
inline int sync_pageable(khg_ctx* ctx) {
  if (!ctx->pageable_pending) return KHG_OK;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->pageable_pending = false;
  return KHG_OK;
}
#define DEVFREE(p) do { if (p) { khg_dev_free((void*)(p)); (p) = nullptr; } } while (0)

// kernel-side plan / result records the handles only hold pointers to (defined beside their kernels, khg_k*.hip.inc)
struct KwChunk; struct K1Chunk; struct K1pEntry; struct K1pSlice; struct K1bChunk; struct K1sChunk; struct K1sUnit; struct K4Res;

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
  uint64_t serial = 0;       // unique per handle: khg_model_lookup(serial) tells a live handle from a destroyed one (or its address reused)
  uint64_t version = 0;      // bumped whenever the parameters or their layout change in place (model_pack)
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
  int32_t KS = 0;
  // f16x2 K1 image (khg_k1_f16x2.hip.inc): packed lazily by khg_loglikes with the scale exponents of the utterance set
  char* wimgh_d = nullptr;
  int32_t wimgh_tiles = 0;
  std::vector<int32_t> wimgh_ex;   // exponents the image was packed with (empty: stale)
  ImgSync wimgh_sync;
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
  std::vector<float> k3_xb_raw;    // the same as model_stats read it (valid with wmax); k3_xb / k3_ex / k3_S are derived from it in khg_k3.hip
  uint32_t* stats_d = nullptr;     // landing block of k0_model_stats
  std::vector<int32_t> k3_ex;      // [80] per k = 2 d + kind
  int32_t k3_S = 0;
  bool k3_f16_ok = false;          // the model side of the form's domain holds
  int32_t* k3_ex_d = nullptr;
  float gcmax = 0.0f;              // max |gconst| over the finite ones (valid with wmax)
  int32_t* tile_pdf_d = nullptr;   // tile -> pdf map of the current layout
  K4Res* k4_res_d = nullptr;       // per-pdf results of the M-step in progress (khg_model_mle_update*)
  int32_t k4_res_P = 0;
};

struct khg_tm {
  khg_ctx* ctx = nullptr;
  int32_t num_tids = 0, max_pdf = -1;
  std::vector<int32_t> id2pdf;
  int32_t* id2pdf_d = nullptr;
  float* trans_cost_d = nullptr;
  bool has_trans_cost = false;
};

struct khg_utts {
  khg_ctx* ctx = nullptr;
  bool small = false;            // scratch from the context's arena (KhgArena): few utterances, one call each
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
  void* band_args = nullptr;       // K1sArgs of the band launch (khg_k1.hip owns it)
  khg_model* band_model = nullptr; int band_ks = 0; size_t band_lds = 0; uint64_t band_serial = 0, band_version = 0; std::vector<int32_t> band_key;   // ... and which model / parameter version / fp16 image they belong to int band_ks = 0; size_t band_lds = 0;
  int tiles_reach = -1;            // whether the walk lists carry those first frames (reachable-only K1) or zeros
  std::vector<int32_t> tiles_pto;  // the model tile layout (pdf_tile_off) the walk lists were built for
  float* ll_d = nullptr; int64_t ll_total = 0; bool ll_valid = false;
  // K1, pdf-major form: repacked features (once), work plan (per reachable flag)
  float* xpl_d = nullptr; int64_t* utt_xtile_off_d = nullptr; int32_t xpl_kq = 0;
  K1pEntry* p_ents_d = nullptr; K1pSlice* p_slices_d = nullptr; int32_t p_nslices = 0; int p_reach = -1; int32_t p_P = -1;
  int32_t p_grp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // slices per block count (index 1..8); the plan also depends on the model's gauss_off
  std::vector<int32_t> p_goff;
  // K1, split forms: the set's 32-frame tile layout, workgroup chunks
  int64_t* utt_x32_off_d = nullptr;
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
  char* out_blk_d = nullptr; size_t out_blk_bytes = 0;   // small sets: [status | like | num_words | words | ali] in one arena block = one download
  hipEvent_t ev_dp = nullptr, ev_ali = nullptr;   // K2-DP done (main) -> faithful kernel (side) -> alignment complete
  bool ali_pending = false;
  // split mode of khg_align (KHG_OPT_K2_SPLIT): the order-faithful decoders' alignments land in ali2_d; unc_d flags the utterances the
  // DP left to them, unc_cnt_h (pinned, device-visible) counts them and their frames.  ali_split: ali2_d has not been merged into
  // ali_d yet (wait_ali does it; khg_acc_stats first accumulates the certified utterances).
  int32_t *ali2_d = nullptr, *unc_d = nullptr, *unc_cnt_h = nullptr, *unc_cnt_dev = nullptr;
  int64_t* sub_off_d = nullptr;     // [U + 1] first position of an uncertified utterance's frames in the second K3 pass
  bool ali_split = false;
  // K3 scratch
  int32_t *pdf_count_d = nullptr, *pdf_cursor_d = nullptr, *frame_ids_d = nullptr;
  uint32_t *sort_keys_d = nullptr, *sort_keys_out_d = nullptr, *sort_vals_d = nullptr; void* sort_tmp_d = nullptr; size_t sort_tmp_bytes = 0;
  int32_t* cs_hist_d = nullptr; size_t cs_hist_n = 0; int64_t* cs_tot_d = nullptr; size_t cs_tot_n = 0;   // counting-sort bucketing (k3_cs_*)
  double *k3_part_d = nullptr, *k3_llpart_d = nullptr; size_t k3_part_n = 0, k3_llpart_n = 0;   // wave-form K3: slice images / per-pdf log-likes
  void* k3_items_d = nullptr; int32_t* k3_item_off_d = nullptr; size_t k3_items_n = 0, k3_item_off_n = 0;   // K3 work items (k3_make_items)
  int64_t* pdf_start_d = nullptr; unsigned long long* tid_count_d = nullptr;
  int32_t k3_P = 0, k3_tids = 0;
};

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

// allocations / uploads owned by an utterance set: the context's arena for small sets, hipMalloc otherwise
constexpr size_t KHG_ARENA_MAX_REQ = size_t(4) << 20;
template <class T>
inline int u_alloc(khg_utts* u, T** p, size_t n) {
  if (n == 0) n = 1;
  if (u->small && n * sizeof(T) <= KHG_ARENA_MAX_REQ) {
    *p = static_cast<T*>(arena_alloc(u->ctx, n * sizeof(T)));
    if (*p) return KHG_OK;
  }
  return dev_alloc(p, n);
}
template <class T>
inline int u_upload(khg_ctx* ctx, khg_utts* u, T** p, const std::vector<T>& v) {
  if (u->small && v.size() * sizeof(T) <= KHG_ARENA_MAX_REQ) {
    *p = static_cast<T*>(arena_alloc(u->ctx, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (*p) {
      if (!v.empty()) { memcpy(u->ctx->arena.mirror(*p), v.data(), v.size() * sizeof(T)); arena_mark_dirty(u->ctx, *p, v.size() * sizeof(T)); }
      return KHG_OK;
    }
  }
  return dev_upload(ctx, p, v);
}

// ---- functions one unit calls in another ---------------------------------------------------------------------------------
int check_err_flag(khg_ctx* c, const char* where);        // khg_ctx_model.hip: read-and-clear the device error word (synchronises)
int model_pack(khg_ctx* ctx, khg_model* m);               // khg_ctx_model.hip: everything derived from gauss_off + the row-major parameters
int model_stats(khg_ctx* ctx, khg_model* m);              // khg_ctx_model.hip: per parameter version, one pass: column maxima, feature envelope, band upper bounds
int wait_ali(khg_ctx* ctx, khg_utts* u);                  // khg_utts.hip: the main stream waits for the side-stream decoder
void k1_free_band(khg_utts* u);                           // khg_k1.hip: the BAND form's saved launch arguments
int k1_maxima(khg_ctx* ctx, khg_model* m, khg_utts* u, std::vector<float>* xk);   // khg_k1.hip: column maxima of features / parameters
khg_model* khg_model_lookup(uint64_t serial);             // khg_ctx_model.hip: the live handle with this serial, or NULL
int k1_band_check(khg_ctx* ctx, khg_utts* u);             // khg_k1.hip: the band's model is alive and unchanged (re-scores when only its image was re-packed)
int k1_band_repair(khg_ctx* ctx, khg_utts* u, int32_t* status_d, int repair_bit, hipStream_t side);   // khg_k1.hip
int accs_allreduce_pieces(khg_ctx* ctx, khg_accs* a, const khg_model* m, int first_pdf, int n_pdf, void* comm, hipStream_t st);   // khg_c1.hip
int ctx_comm_stream(khg_ctx* ctx);                        // khg_c1.hip
