// kaldi_hmm_gmm_amd/csrc/khg_utts.hip -- C-ABI (include/khg_hip.h): utterance sets = features + decoding graphs resident in HBM;
// host-side planning of the per-utterance pdf lists, first / last useful frames and the in-arc CSR K2 reads.  gfx950 only.
#include "khg_internal.hpp"

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
  if (ctx_dead(ctx) || !out || n_utt <= 0 || D <= 0 || !frame_off || (!feats_h && !feats_dv))
    return khg_set_error(KHG_E_ARG, "khg_utts_create: bad arguments");
  if (frame_off[0] != 0) return khg_set_error(KHG_E_ARG, "khg_utts_create: frame_off[0] != 0");
  for (int i = 0; i < n_utt; ++i)
    if (frame_off[i + 1] < frame_off[i]) return khg_set_error(KHG_E_ARG, "khg_utts_create: frame_off not monotone");
  khg_utts* u = new khg_utts();
  u->ctx = ctx; u->n_utt = n_utt; u->D = D;
  u->frame_off.assign(frame_off, frame_off + n_utt + 1);
  u->N = frame_off[n_utt];
  // the reference's per-utterance call pattern (one utterance per set, a new set per call): scratch from the context's arena, one
  // staged copy instead of a hipMalloc + pageable copy per table
  u->small = n_utt <= 16 && u->N <= 16384 && u->N * (int64_t)D <= (int64_t)(512 << 10) && (!state_off || (state_off[n_utt] <= 32768 && arc_off && arc_off[state_off[n_utt]] <= 65536));
  int rc = KHG_OK;
  auto fail = [&](int code, const std::string& msg) { khg_utts_destroy(u); return khg_set_error(code, msg); };
  if (feats_dv) { u->feats_d = feats_dv; u->own_feats = false; }
  else {
    float* p = nullptr;
    rc = u_alloc(u, &p, (size_t)u->N * D);
    if (rc) { khg_utts_destroy(u); return rc; }
    u->feats_d = p; u->own_feats = true;
    if (u->N) {
      const size_t nb = sizeof(float) * (size_t)u->N * D;
      if (ctx->arena.owns(p)) { memcpy(ctx->arena.mirror(p), feats_h, nb); arena_mark_dirty(ctx, p, nb); }
      else {
        hipError_t e = hipMemcpyAsync(p, feats_h, nb, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(KHG_E_HIP, hipGetErrorString(e));
        ctx->pageable_pending = true;
      }
    }
    if (u->small && u->N > 0) {      // the column maxima the split K1 forms scale by: a pass over 48 kB here instead of a kernel + a download
      u->xmax.assign((size_t)D, 0.0f);
      for (int64_t t = 0; t < u->N; ++t) {
        const float* x = feats_h + (size_t)t * D;
        for (int d = 0; d < D; ++d) {        // k1h_absmax's rule: -inf skipped, NaN / +inf come out as +inf
          const float v = x[d], m = u->xmax[(size_t)d];
          if (v == -std::numeric_limits<float>::infinity()) continue;
          u->xmax[(size_t)d] = (std::fabs(v) <= 3.0e38f) ? std::fmax(m, std::fabs(v)) : std::numeric_limits<float>::infinity();
        }
      }
    }
  }
  rc = u_upload(ctx, u, &u->frame_off_d, u->frame_off);
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
    rc = u_upload(ctx, u, &u->state_off_d, u->state_off);
    if (!rc) rc = u_upload(ctx, u, &u->start_d, startv);
    if (!rc) rc = u_upload(ctx, u, &u->in_off_d, in_off);
    if (!rc) rc = u_upload(ctx, u, &u->out_off_d, out_off);
    if (!rc) rc = u_upload(ctx, u, &u->in_src_d, in_src);
    if (!rc) rc = u_upload(ctx, u, &u->in_col_d, in_col);
    if (!rc) rc = u_upload(ctx, u, &u->in_tid_d, in_tid);
    if (!rc) rc = u_upload(ctx, u, &u->in_olabel_d, in_ol);
    if (!rc) rc = u_upload(ctx, u, &u->out_inidx_d, out_inidx);
    if (!rc) rc = u_upload(ctx, u, &u->in_w_d, in_w);
    if (!rc) rc = u_upload(ctx, u, &u->final_d, finalv);
    if (!rc) rc = u_upload(ctx, u, &u->bp_off_d, u->bp_off);
    if (!rc) rc = u_upload(ctx, u, &u->path_off_d, u->path_off);
    if (!rc) rc = u_upload(ctx, u, &u->words_off_d, u->words_off);
    if (rc) { khg_utts_destroy(u); return rc; }
  }
  plan_ll(u);
  rc = arena_flush(ctx);
  if (!rc) rc = sync_pageable(ctx);        // the caller's arrays and the vectors above are free after this
  if (rc) { khg_utts_destroy(u); return rc; }
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
  u->ll_valid = false;
  return KHG_OK;
}

int utts_foreign_ctx(const khg_ctx* ctx, const khg_utts* u, const char* where) {
  if (u && u->small && u->ctx != ctx)
    return khg_set_error(KHG_E_ARG, std::string(where) + ": a small utterance set (<= 16 utterances: scratch in its context's arena) must be used with the context that created it");
  return KHG_OK;
}

extern "C" int khg_utts_destroy(khg_utts* u) {
  if (!u) return KHG_OK;
  if (u->small && u->ctx && khg_ctx_alive(u->ctx)) {      // (a handle may be destroyed after its context: nothing is in flight then)
    // arena scratch is reused by the next set at once (hipFree would have waited for the device): nothing of this set may be in flight
    khg_ctx* c = u->ctx;
    for (int i = 0; i < khg_ctx::NSIDE; ++i)
      if (c->side_dirty[i]) { (void)hipStreamSynchronize(c->sides[i]); c->side_dirty[i] = false; }
    (void)hipStreamSynchronize(c->stream);
    c->pageable_pending = false;
  }
  if (u->out_blk_d) {                      // ali / words / like / status live inside one block (small sets)
    u->ali_d = nullptr; u->words_d = nullptr; u->num_words_d = nullptr; u->status_d = nullptr; u->like_d = nullptr;
    DEVFREE(u->out_blk_d);
  }
  if (u->own_feats) DEVFREE(u->feats_d);
  DEVFREE(u->frame_off_d); DEVFREE(u->state_off_d); DEVFREE(u->pdf_off_d); DEVFREE(u->ll_off_d);
  DEVFREE(u->pdfs_d); DEVFREE(u->wchunks_d); DEVFREE(u->start_d); DEVFREE(u->in_off_d); DEVFREE(u->out_off_d);
  DEVFREE(u->in_src_d); DEVFREE(u->in_col_d); DEVFREE(u->in_tid_d); DEVFREE(u->in_olabel_d); DEVFREE(u->out_inidx_d);
  DEVFREE(u->in_w_d); DEVFREE(u->final_d); DEVFREE(u->chunks_d); DEVFREE(u->ll_d); DEVFREE(u->tile_off_d); DEVFREE(u->tiles_d);
  DEVFREE(u->xpl_d); DEVFREE(u->utt_xtile_off_d); DEVFREE(u->p_ents_d); DEVFREE(u->p_slices_d);
  DEVFREE(u->utt_x32_off_d); DEVFREE(u->bchunks_d); DEVFREE(u->x32_utt_d); DEVFREE(u->xh_d); DEVFREE(u->xh_ex_d);
  DEVFREE(u->xs_d); DEVFREE(u->xs_ex_d); DEVFREE(u->schunks_d); DEVFREE(u->sunits_d);
  DEVFREE(u->bp_d); DEVFREE(u->bp_off_d); DEVFREE(u->path_off_d); DEVFREE(u->words_off_d);
  DEVFREE(u->ali2_d); DEVFREE(u->unc_d); DEVFREE(u->sub_off_d);
  if (u->unc_cnt_h) { (void)hipHostFree(u->unc_cnt_h); u->unc_cnt_h = nullptr; }
  DEVFREE(u->layer_best_d); DEVFREE(u->layer_cnt_d); DEVFREE(u->path_d); DEVFREE(u->k2_gscratch_d); DEVFREE(u->k2_order_d);
  DEVFREE(u->ali_d); DEVFREE(u->words_d); DEVFREE(u->num_words_d); DEVFREE(u->status_d); DEVFREE(u->like_d);
  DEVFREE(u->pdf_count_d); DEVFREE(u->pdf_cursor_d); DEVFREE(u->frame_ids_d); DEVFREE(u->pdf_start_d); DEVFREE(u->tid_count_d);
  DEVFREE(u->sort_keys_d); DEVFREE(u->sort_keys_out_d); DEVFREE(u->sort_vals_d); DEVFREE(u->sort_tmp_d); DEVFREE(u->cs_hist_d); DEVFREE(u->cs_tot_d);
  DEVFREE(u->k3_part_d); DEVFREE(u->k3_llpart_d); DEVFREE(u->k3_items_d); DEVFREE(u->k3_item_off_d);
  k1_free_band(u);
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
// split mode: the alignments the order-faithful decoders wrote to their own buffer go into the set's alignment
__global__ void k2_merge_fallback(const int32_t* __restrict__ unc, const int64_t* __restrict__ frame_off, const int32_t* __restrict__ ali2,
                                  int32_t* __restrict__ ali, int n_utt) {
  for (int u = blockIdx.x; u < n_utt; u += gridDim.x) {
    if (!unc[u]) continue;
    const int64_t f0 = frame_off[u], f1 = frame_off[u + 1];
    for (int64_t f = f0 + threadIdx.x; f < f1; f += blockDim.x) ali[f] = ali2[f];
  }
}
int wait_ali(khg_ctx* ctx, khg_utts* u) {
  if (u->ali_pending) { HIPCHK(hipStreamWaitEvent(ctx->stream, u->ev_ali, 0)); u->ali_pending = false; }
  if (u->ali_split) {
    u->ali_split = false;
    KHG_LAUNCH(ctx, k2_merge_fallback, dim3((unsigned)std::min(u->n_utt, 16384)), dim3(64), 0, ctx->stream, u->unc_d, u->frame_off_d, u->ali2_d, u->ali_d, u->n_utt);
    HIPCHK(hipGetLastError());
  }
  return KHG_OK;
}
