// kaldi_hmm_gmm_amd/csrc/khg_k2.hip -- C-ABI (include/khg_hip.h): K2, Viterbi forced alignment (khg_align): kernel selection by graph
// shape, LDS budgets, the exact DP on the main stream and the order-faithful decoder on a side stream.  gfx950 only.
#include "khg_internal.hpp"

#include "khg_k2_viterbi.hip.inc"

// ------------------------------------------------------------------------------------------
// K2
extern "C" void khg_align_config_default(khg_align_config* c) {
  c->beam = 200.0f; c->retry_beam = 0.0f; c->careful = 0; c->acoustic_scale = 1.0f;
  c->max_active = INT32_MAX; c->min_active = 20; c->beam_delta = 0.5f; c->hash_ratio = 2.0f;
  c->like_scale = 0.0f;
}

// The resident alignment.  Small sets with graphs keep everything khg_align hands back -- [status | like | num_words | words | ali] --
// in ONE block (of the context's arena: one download into its pinned mirror instead of five pageable copies).
static int ensure_ali(khg_ctx* ctx, khg_utts* u) {
  if (u->ali_d) return KHG_OK;
  if (u->small && u->has_graphs) {
    const size_t nu = (size_t)u->n_utt, nw = (size_t)u->words_off[u->n_utt];
    const size_t bytes = 4 * (3 * nu + nw + (size_t)u->N) + 64;
    int rc = u_alloc(u, &u->out_blk_d, bytes);
    if (rc) return rc;
    u->out_blk_bytes = bytes;
    int32_t* p = reinterpret_cast<int32_t*>(u->out_blk_d);
    u->status_d = p; u->like_d = reinterpret_cast<float*>(p + nu); u->num_words_d = p + 2 * nu; u->words_d = p + 3 * nu; u->ali_d = p + 3 * nu + nw;
    return KHG_OK;
  }
  return u_alloc(u, &u->ali_d, (size_t)u->N);
}

extern "C" int khg_align(khg_ctx* ctx, const khg_tm* tm, khg_utts* u, const khg_align_config* cfg,
                         int32_t* ali_h, int32_t* words_h, int64_t* words_off_h, int64_t words_cap,
                         float* like_h, int32_t* status_h) {
  if (ctx_dead(ctx) || !tm || !u || !cfg) return khg_set_error(KHG_E_ARG, "khg_align: bad arguments");
  { int rf = utts_foreign_ctx(ctx, u, "khg_align"); if (rf) return rf; }
  if (!u->has_graphs) return khg_set_error(KHG_E_ARG, "khg_align: the utterance set has no decoding graphs");
  if (!u->ll_valid) return khg_set_error(KHG_E_ARG, "khg_align: call khg_loglikes first");
  // decoder-wrappers.cc:29-33
  if ((cfg->retry_beam != 0 && cfg->retry_beam <= cfg->beam) || cfg->beam <= 0.0)
    return khg_set_error(KHG_E_RUNTIME, "Beams do not make sense: beam " + std::to_string(cfg->beam) + ", retry-beam " + std::to_string(cfg->retry_beam));
  // faster-decoder.cc:24-27
  if (!(cfg->hash_ratio >= 1.0) || !(cfg->max_active > 1) || !(cfg->min_active >= 0 && cfg->min_active < cfg->max_active))
    return khg_set_error(KHG_E_RUNTIME, "FasterDecoderOptions assertion failed");
  int rc = wait_ali(ctx, u);
  if (!rc) rc = k1_band_check(ctx, u);      // BAND scores: their model must be alive and unchanged (khg_k1.hip)
  if (!rc) rc = ensure_ali(ctx, u);
  if (rc) return rc;
  if (!u->bp_d) {
    rc = u_alloc(u, &u->bp_d, (size_t)u->bp_off[u->n_utt]);
    if (!rc) rc = u_alloc(u, &u->layer_best_d, (size_t)(u->N + u->n_utt));
    if (!rc) rc = u_alloc(u, &u->layer_cnt_d, (size_t)(u->N + u->n_utt));
    if (!rc) rc = u_alloc(u, &u->path_d, (size_t)u->path_off[u->n_utt]);
    if (!rc && !u->out_blk_d) {
      rc = u_alloc(u, &u->words_d, (size_t)u->words_off[u->n_utt]);
      if (!rc) rc = u_alloc(u, &u->num_words_d, (size_t)u->n_utt);
      if (!rc) rc = u_alloc(u, &u->status_d, (size_t)u->n_utt);
      if (!rc) rc = u_alloc(u, &u->like_d, (size_t)u->n_utt);
    }
    if (rc) return rc;
  }
  rc = arena_flush(ctx);      // an alignment staged by khg_ali_upload must not land on top of the cleared block
  if (rc) return rc;
  HIPCHK(hipMemsetAsync(u->ali_d, 0, sizeof(int32_t) * (size_t)u->N, ctx->stream));
  K2Args a;
  a.frame_off = u->frame_off_d; a.state_off = u->state_off_d; a.start = u->start_d;
  a.in_off = u->in_off_d; a.in_src = u->in_src_d; a.in_col = u->in_col_d; a.in_tid = u->in_tid_d;
  a.in_olabel = u->in_olabel_d; a.in_w = u->in_w_d; a.out_off = u->out_off_d; a.out_inidx = u->out_inidx_d;
  a.final_w = u->final_d; a.trans_cost = tm->has_trans_cost ? tm->trans_cost_d : nullptr;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d;
  a.bp = u->bp_d; a.bp_off = u->bp_off_d; a.layer_best = u->layer_best_d; a.layer_cnt = u->layer_cnt_d;
  a.path = u->path_d; a.path_off = u->path_off_d;
  a.ali = u->ali_d; a.ali_fb = u->ali_d; a.unc = nullptr; a.unc_cnt = nullptr; a.words = u->words_d; a.words_off = u->words_off_d; a.num_words = u->num_words_d;
  a.like = u->like_d; a.status = u->status_d; a.err_flag = ctx->err_flag_d;
  a.prof = nullptr;
  // launch order of the DP kernel: longest utterances first (built once per set)
  if (!u->k2_order_d && u->n_utt > 1) {
    std::vector<int32_t> ord((size_t)u->n_utt);
    for (int i = 0; i < u->n_utt; ++i) ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int32_t x, int32_t y) {
      return u->frame_off[x + 1] - u->frame_off[x] > u->frame_off[y + 1] - u->frame_off[y];
    });
    int rc2 = u_upload(ctx, u, &u->k2_order_d, ord);
    if (!rc2) rc2 = sync_pageable(ctx);
    if (rc2) return rc2;
  }
  a.order = ctx->opt[KHG_OPT_K2_INORDER] ? nullptr : u->k2_order_d;
  // Split mode: nobody waits for the results here and the set is large -- the order-faithful decoders (side stream) write to ali2_d, the DP
  // kernel leaves an uncertified utterance's range of ali_d zero, flags and counts it: khg_acc_stats can then accumulate the certified
  // utterances while those decoders still run (khg_k3.hip); wait_ali merges.
  const bool split = !(ali_h || like_h || status_h || words_h) && u->n_utt > 64 && !u->small && ctx->opt[KHG_OPT_K2_SPLIT] == 0;
  if (split) {
    if (!u->ali2_d) {
      rc = u_alloc(u, &u->ali2_d, (size_t)u->N);
      if (!rc) rc = u_alloc(u, &u->unc_d, (size_t)u->n_utt + 2);       // flags | [utterances, frames] the DP could not certify
      if (rc) return rc;
      u->unc_cnt_dev = u->unc_d + u->n_utt;
      HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&u->unc_cnt_h), 64, 0));   // pinned landing place of the two counters
    }
    HIPCHK(hipMemsetAsync(u->unc_d, 0, sizeof(int32_t) * ((size_t)u->n_utt + 2), ctx->stream));
    a.ali_fb = u->ali2_d; a.unc = u->unc_d; a.unc_cnt = u->unc_cnt_dev;
  }
  const bool k2prof = ctx->opt[KHG_OPT_K2_PROF] != 0;
  if (k2prof) { HIPCHK(hipMalloc(reinterpret_cast<void**>(&a.prof), sizeof(long long) * 16 * (size_t)u->n_utt)); HIPCHK(hipMemset(a.prof, 0, sizeof(long long) * 16 * (size_t)u->n_utt)); }   // [U][8] DP stamps | [U][8] chain-decoder stamps
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
  // The chain form (k2_viterbi_faithful_chain: no epsilon-input arcs, <= 1000 states, out-degree <= 4 -- a linear transcript's training
  // graph): a third of the wave form's latency per frame.  KHG_K2_SERIAL = 3: the general wave form also where the chain form applies.
  const int odeg_c = (int)std::max<int32_t>(1, u->max_outdeg);
  const size_t S4 = (S + 3) & ~size_t(3);
  // the frame loop's tables: token costs x 2 + state keys (24) | first / winner (8) | token states x 2 (4): 36 per state; per out-arc slot:
  // parked cost (8) + record (8) + info (4, GetCutoff's array over it) + ordinal (1); the score row.  Over them, set-up and tail only:
  // in-arc offsets (4 per state) + sources (2 per arc) + the trace-back's 9 rows.
  const size_t lds_chain = std::max<size_t>(36 * S4 + 21 * S4 * (size_t)odeg_c + 4 * max_npdf,
                                            4 * (S4 + 1) + 2 * A + 16 + 9 * ((S + 15) & ~size_t(15))) + 128;
  const bool chain = fmode == 0 && !u->has_eps && S <= 1000 && u->max_outdeg <= 4 && lds_chain <= 64 * 1024 && max_npdf <= 32767;
  const bool wave_lds = (fmode == 0 || fmode == 3) && S <= 65535 && lds_w_mut + lds_w_graph <= 160 * 1024;
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
      { int rg = u_alloc(u, &u->k2_gscratch_d, need); if (rg) return rg; }
      u->k2_gscratch_bytes = need;
    }
    a.gscratch = u->k2_gscratch_d; a.gscratch_stride = (int64_t)stride;
  }
  if (gmem) {
    KernelTimer kt(ctx, "k2_viterbi_dp");
    KHG_LAUNCH(ctx, (k2_viterbi_dp<1, 1, false, true>), dim3(u->n_utt), dim3(nthr), 0, ctx->stream, a);
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
#define K2_LAUNCH(FN) KHG_LAUNCH(ctx, FN, dim3(u->n_utt), dim3(nthr), lds_dp, ctx->stream, a)
    if (lds_dp > 48 * 1024) { K2_DP_CASES(K2_SET_LDS) }
    KernelTimer kt(ctx, "k2_viterbi_dp");
    K2_DP_CASES(K2_LAUNCH)
#undef K2_LAUNCH
#undef K2_SET_LDS
#undef K2_DP_CASES
  }
  HIPCHK(hipGetLastError());
  // The order-faithful decoder for what the DP could not certify runs on a side stream, beside the main stream's next K1 -- unless
  // the caller waits for the results right here (host outputs): then nothing can overlap it, and it follows the DP on the main
  // stream without the two cross-stream events (the per-utterance call pattern: ~30 us of a ~300 us call).
  const bool sync_call = ali_h || like_h || status_h || words_h;
  hipStream_t side = ctx->stream;
  if (!sync_call) {
    if (!u->ev_dp) { HIPCHK(hipEventCreateWithFlags(&u->ev_dp, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&u->ev_ali, hipEventDisableTiming)); }
    if (split) HIPCHK(hipMemcpyAsync(u->unc_cnt_h, u->unc_cnt_dev, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));   // read by khg_acc_stats behind ev_dp
    HIPCHK(hipEventRecord(u->ev_dp, ctx->stream));
    side = ctx->sides[ctx->next_side];
    ctx->side_dirty[ctx->next_side] = true;
    ctx->next_side = (ctx->next_side + 1) % khg_ctx::NSIDE;
    HIPCHK(hipStreamWaitEvent(side, u->ev_dp, 0));
  }
  rc = k1_band_repair(ctx, u, u->status_d, K2_ST_NEED_FALLBACK, side);      // khg_k1.hip (BAND form of K1 only)
  if (rc) return rc;
  {
    KernelTimer kt(ctx, "k2_viterbi_faithful", side);
    if (chain) {
#define K2_CHAIN(OD)                                                                                                            \
  do {                                                                                                                          \
    if (lds_chain > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful_chain<OD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_chain)); \
    KHG_LAUNCH(ctx, k2_viterbi_faithful_chain<OD>, dim3(u->n_utt), dim3(64), lds_chain, side, a, (int)max_npdf);                \
  } while (0)
      switch (odeg_c) { case 1: K2_CHAIN(1); break; case 2: K2_CHAIN(2); break; case 3: K2_CHAIN(3); break; default: K2_CHAIN(4); break; }
#undef K2_CHAIN
    } else if (wave_gm) {
      if (lds_w_mut > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful_wave<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w_mut));
      KHG_LAUNCH(ctx, k2_viterbi_faithful_wave<true>, dim3(u->n_utt), dim3(64), lds_w_mut, side, a, u->has_eps ? 1 : 0, odeg_w, (int)max_npdf);
    } else if (wave_lds) {
      const size_t lds_w = lds_w_mut + lds_w_graph;
      if (lds_w > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful_wave<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
      KHG_LAUNCH(ctx, k2_viterbi_faithful_wave<false>, dim3(u->n_utt), dim3(64), lds_w, side, a, u->has_eps ? 1 : 0, odeg_w, (int)max_npdf);
    } else if (lane_gm) {
      KHG_LAUNCH(ctx, k2_viterbi_faithful<true>, dim3(u->n_utt), dim3(64), 0, side, a);
    } else {
      if (lds_f > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k2_viterbi_faithful<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
      KHG_LAUNCH(ctx, k2_viterbi_faithful<false>, dim3(u->n_utt), dim3(64), lds_f, side, a);
    }
  }
  HIPCHK(hipGetLastError());
  if (!sync_call) { HIPCHK(hipEventRecord(u->ev_ali, side)); u->ali_pending = true; }
  u->ali_split = split;
  u->ali_valid = true;
  if (k2prof) {  // diagnostics: average s_memtime ticks per phase of k2_viterbi_dp
    std::vector<long long> pr(16 * (size_t)u->n_utt);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipStreamSynchronize(side));
    HIPCHK(hipMemcpy(pr.data(), a.prof, pr.size() * 8, hipMemcpyDeviceToHost));
    (void)hipFree(a.prof);
    double ph[4] = {0, 0, 0, 0}, sT = 0, sS = 0, sf = 0; int n = 0;
    for (int i = 0; i < u->n_utt; ++i) if (pr[i * 8 + 4]) { for (int k = 0; k < 4; ++k) ph[k] += (double)(pr[i * 8 + k + 1] - pr[i * 8 + k]); sT += pr[i * 8 + 5]; sS += pr[i * 8 + 6]; sf += pr[i * 8 + 7]; ++n; }
    if (n) fprintf(stderr, "[KHG_K2_PROF] %d utts, avg T %.1f S %.1f fast %.2f threads %d lds %zu | ticks: setup %.0f forward %.0f traceback %.0f replay %.0f\n",
                   n, sT / n, sS / n, sf / n, nthr, lds_dp, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
    const long long* pc = pr.data() + 8 * (size_t)u->n_utt;     // chain decoder: 0 start, 1 set-up done, 2 frames done (all attempts), 3 finished, 4 frames decoded
    double cs[3] = {0, 0, 0}, cf = 0, cp[3] = {0, 0, 0}; int nc = 0;
    for (int i = 0; i < u->n_utt; ++i) if (pc[i * 8 + 3]) { for (int k = 0; k < 3; ++k) { cs[k] += (double)(pc[i * 8 + k + 1] - pc[i * 8 + k]); cp[k] += (double)pc[i * 8 + 5 + k]; } cf += (double)pc[i * 8 + 4]; ++nc; }
    if (nc) fprintf(stderr, "[KHG_K2_PROF] chain decoder: %d utts, avg frames decoded %.1f | ticks: setup %.0f frames %.0f (%.1f per frame: GetCutoff %.1f, pass 1 %.1f, pass 2 %.1f, rest = row flush + score staging) finish %.0f\n",
                    nc, cf / nc, cs[0] / nc, cs[1] / nc, cs[1] / std::max(1.0, cf), cp[0] / std::max(1.0, cf), cp[1] / std::max(1.0, cf), cp[2] / std::max(1.0, cf), cs[2] / nc);
  }
  if (!ali_h && !like_h && !status_h && !words_h) return KHG_OK;   // asynchronous: errors surface at khg_ctx_sync / downloads
  rc = wait_ali(ctx, u);
  if (rc) return rc;
  if (u->out_blk_d && ctx->arena.owns(u->out_blk_d)) {
    // one block, one copy into its pinned mirror; the error word's own copy and wait follow it on the stream
    char* hm = ctx->arena.mirror(u->out_blk_d);
    HIPCHK(hipMemcpyAsync(hm, u->out_blk_d, u->out_blk_bytes, hipMemcpyDeviceToHost, ctx->stream));
    rc = check_err_flag(ctx, "khg_align");  // synchronises
    if (rc) return rc;
    const size_t nu = (size_t)u->n_utt, nwt = (size_t)u->words_off[u->n_utt];
    const int32_t* hp = reinterpret_cast<const int32_t*>(hm);
    if (status_h) memcpy(status_h, hp, 4 * nu);
    if (like_h) memcpy(like_h, hp + nu, 4 * nu);
    if (ali_h && u->N) memcpy(ali_h, hp + 3 * nu + nwt, 4 * (size_t)u->N);
    if (words_h && words_off_h) {
      const int32_t *nw = hp + 2 * nu, *w = hp + 3 * nu;
      int64_t o = 0;
      for (int i = 0; i < u->n_utt; ++i) {
        words_off_h[i] = o;
        const int64_t n = std::min<int64_t>(nw[i], u->words_off[i + 1] - u->words_off[i]);
        if (o + n > words_cap) return khg_set_error(KHG_E_ARG, "khg_align: words_cap too small");
        std::copy(w + u->words_off[i], w + u->words_off[i] + n, words_h + o);
        o += n;
      }
      words_off_h[u->n_utt] = o;
    }
    return KHG_OK;
  }
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
  if (ctx_dead(ctx) || !u || !ali) return khg_set_error(KHG_E_ARG, "bad arguments");
  { int rf = utts_foreign_ctx(ctx, u, "khg_ali_upload"); if (rf) return rf; }
  int rc = wait_ali(ctx, u);
  if (!rc) rc = ensure_ali(ctx, u);
  if (rc) return rc;
  if (ctx->arena.owns(u->ali_d)) {          // staged: goes out with the next launch's flush
    if (u->N) { memcpy(ctx->arena.mirror(u->ali_d), ali, sizeof(int32_t) * (size_t)u->N); arena_mark_dirty(ctx, u->ali_d, sizeof(int32_t) * (size_t)u->N); }
  } else {
    HIPCHK(hipMemcpyAsync(u->ali_d, ali, sizeof(int32_t) * (size_t)u->N, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  u->ali_valid = true;
  return KHG_OK;
}

extern "C" int khg_ali_download(khg_ctx* ctx, khg_utts* u, int32_t* ali) {
  if (ctx_dead(ctx) || !u || !ali) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (!u->ali_valid) return khg_set_error(KHG_E_ARG, "khg_ali_download: no resident alignment");
  { int rf = utts_foreign_ctx(ctx, u, "khg_ali_download"); if (rf) return rf; }
  int rc = wait_ali(ctx, u);
  if (!rc) rc = arena_flush(ctx);           // small sets: an uploaded alignment may still be staged in the pinned mirror
  if (!rc) rc = check_err_flag(ctx, "khg_align");
  if (rc) return rc;
  if (u->N) HIPCHK(hipMemcpyAsync(ali, u->ali_d, sizeof(int32_t) * (size_t)u->N, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
