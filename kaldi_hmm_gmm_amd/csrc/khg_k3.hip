// kaldi_hmm_gmm_amd/csrc/khg_k3.hip -- C-ABI (include/khg_hip.h): the accumulator block and K3, sufficient statistics
// (khg_acc_stats / khg_acc_stats_reduce): bucketing of frames by pdf, work items, form selection, the launches.  gfx950 only.
#include "khg_internal.hpp"

#include <hipcub/hipcub.hpp>   // DeviceRadixSort: the stable (pdf, frame) sort of K3's bucketing

#include "khg_k3_accstats.hip.inc"

// ------------------------------------------------------------------------------------------
// accumulators + K3
extern "C" int khg_accs_create(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_accs** out) {
  if (ctx_dead(ctx) || !m || !tm || !out) return khg_set_error(KHG_E_ARG, "khg_accs_create: bad arguments");
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
  if (ctx_dead(ctx) || !a) return khg_set_error(KHG_E_ARG, "bad arguments");
  HIPCHK(hipMemsetAsync(a->buf_d, 0, sizeof(double) * (size_t)a->n, ctx->stream));
  return KHG_OK;
}
extern "C" int khg_accs_size(const khg_accs* a, int64_t* n) { if (!a || !n) return khg_set_error(KHG_E_ARG, "bad arguments"); *n = a->n; return KHG_OK; }
extern "C" int khg_accs_device_ptr(const khg_accs* a, void** p) { if (!a || !p) return khg_set_error(KHG_E_ARG, "bad arguments"); *p = a->buf_d; return KHG_OK; }
extern "C" int khg_accs_download(khg_ctx* ctx, const khg_accs* a, double* buf) {
  if (ctx_dead(ctx) || !a || !buf) return khg_set_error(KHG_E_ARG, "bad arguments");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(buf, a->buf_d, sizeof(double) * (size_t)a->n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_accs_upload(khg_ctx* ctx, khg_accs* a, const double* buf) {
  if (ctx_dead(ctx) || !a || !buf) return khg_set_error(KHG_E_ARG, "bad arguments");
  HIPCHK(hipMemcpyAsync(a->buf_d, buf, sizeof(double) * (size_t)a->n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

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
  const int D = m->D, K = 8 * m->KQ;           // 80 exponents at D <= 40 (the wave forms), 160 at D <= 80 (k3_accumulate_block16)
  if (m->KQ == 0) return KHG_OK;
  std::vector<float> xk;
  int rc = k1_maxima(ctx, m, u, &xk);          // the set's column maxima (cached) and the model's (wmax, gcmax; cached per version)
  if (rc) return rc;
  if (m->k3_xb.empty()) {
    if (m->wmax.empty() || (int)m->k3_xb_raw.size() != D) { m->wmax.clear(); rc = model_stats(ctx, m); if (rc) return rc; }
    m->k3_xb = m->k3_xb_raw;          // (read with the column maxima: k0_model_stats, one pass per parameter version)
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

// One pass of K3 over the set's resident alignment: all N frames (nsub < 0), or -- the second pass of the split mode -- the nsub frames
// of the utterances the DP left to the order-faithful decoders (flagged in u->unc_d; their alignments are merged into ali_d by now).
static int acc_stats_pass(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc, void* comm, int nparts, int64_t nsub) {
  int rc = KHG_OK;
  const int64_t Neff = nsub < 0 ? u->N : nsub;
  HIPCHK(hipMemsetAsync(u->pdf_count_d, 0, sizeof(int32_t) * (size_t)m->P, ctx->stream));
  HIPCHK(hipMemsetAsync(u->tid_count_d, 0, sizeof(unsigned long long) * ((size_t)tm->num_tids + 1), ctx->stream));
  K3Args a;
  a.feats = u->feats_d; a.ali = u->ali_d; a.id2pdf = tm->id2pdf_d; a.num_tids = tm->num_tids;
  a.N = Neff; a.P = m->P; a.D = m->D;
  a.gauss_off = m->gauss_off_d; a.gconsts = m->gconsts_d; a.means_invvars = m->miv_d; a.inv_vars = m->iv_d; a.nhalf_inv_vars = m->nhiv_d;
  a.pdf_count = u->pdf_count_d; a.pdf_start = u->pdf_start_d; a.pdf_cursor = u->pdf_cursor_d;
  a.frame_ids = u->frame_ids_d; a.tid_count = u->tid_count_d;
  a.occ = acc->occ(); a.mean_acc = acc->mean(); a.var_acc = acc->var(); a.trans_acc = acc->trans(); a.scalars = acc->scalars();
  a.weight = weight; a.err_flag = ctx->err_flag_d; a.part = nullptr; a.ll_part = nullptr; a.pdf0 = 0; a.npdf = m->P; a.items = nullptr; a.item_off = nullptr;
  a.pa_ex = nullptr; a.pa_S = 0; a.pa_scale = 1.0f; a.pa_inv = 1.0f; a.pa_c1 = 1.44269504088896340736f;
  nparts = std::max(1, std::min(nparts, m->P));
  if (comm && nparts > 1) { rc = ctx_comm_stream(ctx); if (rc) return rc; }
  if (Neff > 0) {
    const int gb = (int)std::min<int64_t>(4096, (Neff + 255) / 256);
    {
      KernelTimer kt(ctx, nsub < 0 ? "k3_bucket" : "k3_bucket_pass2");
      // KHG_OPT_K3_BUCKET = 1: cursor-bump scatter (bucket order depends on the atomics)
      if (nsub < 0 && (ctx->opt[KHG_OPT_K3_BUCKET] == 1 || u->N >= (int64_t)INT_MAX)) {
        KHG_LAUNCH(ctx, k3_count, dim3(gb), dim3(256), 0, ctx->stream, a);
        KHG_LAUNCH(ctx, k3_scan, dim3(1), dim3(1024), 0, ctx->stream, a);
        KHG_LAUNCH(ctx, k3_scatter, dim3(gb), dim3(256), 0, ctx->stream, a);
      } else if (nsub < 0 && ctx->opt[KHG_OPT_K3_BUCKET] == 2 && m->P <= K3_CS_MAXP) {
        // the library's own stable counting sort (khg_k3_accstats.hip.inc: k3_cs_*; opt-in: 1.27 ms against the radix sort's 0.80 at the
        // bench size): blocks of CB consecutive frames
        const int nw = m->P + 1 <= 7168 ? 4 : 2;                      // waves of a placing block: nw x (P + 1 + 1024) counters of LDS
        int64_t CB = 32768;
        while (CB > 64 * nw * 4 && (u->N + CB - 1) / CB < 1024) CB /= 2;   // enough blocks to fill the chip on small sets
        const int nblk = (int)((u->N + CB - 1) / CB);
        const size_t hist_n = (size_t)nblk * ((size_t)m->P + 1);
        if (u->cs_hist_n < hist_n) { DEVFREE(u->cs_hist_d); rc = u_alloc(u, &u->cs_hist_d, hist_n); if (rc) return rc; u->cs_hist_n = hist_n; }
        if (u->cs_tot_n < (size_t)m->P + 1) { DEVFREE(u->cs_tot_d); rc = u_alloc(u, &u->cs_tot_d, (size_t)m->P + 1); if (rc) return rc; u->cs_tot_n = (size_t)m->P + 1; }
        K3CsArgs c{u->cs_hist_d, u->cs_tot_d, (int32_t)CB, nblk};
        const bool ldst = tm->num_tids <= K3_LDS_TIDS;
        const size_t lds_h = sizeof(unsigned int) * ((size_t)m->P + 1 + (ldst ? (size_t)tm->num_tids + 1 : 0));
        const size_t lds_p = sizeof(unsigned int) * (size_t)nw * ((size_t)m->P + 1 + 1024);     // per-wave counters + the run-head hash tags
        if (lds_h > 48 * 1024) {
          HIPCHK(hipFuncSetAttribute((const void*)k3_cs_hist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
          HIPCHK(hipFuncSetAttribute((const void*)k3_cs_hist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h));
        }
        if (lds_p > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k3_cs_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
        if (ldst) KHG_LAUNCH(ctx, k3_cs_hist<true>, dim3(nblk), dim3(256), lds_h, ctx->stream, a, c);
        else KHG_LAUNCH(ctx, k3_cs_hist<false>, dim3(nblk), dim3(256), lds_h, ctx->stream, a, c);
        KHG_LAUNCH(ctx, k3_cs_scan, dim3((m->P + 256) / 256), dim3(256), 0, ctx->stream, a, c);
        KHG_LAUNCH(ctx, k3_cs_starts, dim3(1), dim3(1024), 0, ctx->stream, a, c);
        KHG_LAUNCH(ctx, k3_cs_place, dim3(nblk), dim3(64 * nw), lds_p, ctx->stream, a, c);
      } else {
        // stable sort of (pdf, frame) pairs: frames of a pdf stay in frame order; the bucket boundaries are read
        // off the sorted keys
        int bits = 1;
        while ((1 << bits) <= m->P) ++bits;            // keys are 0..P
        if (!u->sort_keys_d) {
          rc = u_alloc(u, &u->sort_keys_d, (size_t)u->N);
          if (!rc) rc = u_alloc(u, &u->sort_keys_out_d, (size_t)u->N);
          if (!rc) rc = u_alloc(u, &u->sort_vals_d, (size_t)u->N);
          if (rc) return rc;
        }
        size_t need = 0;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need, u->sort_keys_d, u->sort_keys_out_d, u->sort_vals_d,
                                                  reinterpret_cast<uint32_t*>(u->frame_ids_d), (int)Neff, 0, bits, ctx->stream));
        if (need > u->sort_tmp_bytes) {
          DEVFREE(u->sort_tmp_d);
          { int rt = u_alloc(u, reinterpret_cast<char**>(&u->sort_tmp_d), need); if (rt) return rt; }
          u->sort_tmp_bytes = need;
        }
        if (nsub >= 0) {
          // the (pdf, frame) pairs of the flagged utterances only, utterances in order: positions from an exclusive sum of their lengths
          if (!u->sub_off_d) { rc = u_alloc(u, &u->sub_off_d, (size_t)u->n_utt + 1); if (rc) return rc; }
          KHG_LAUNCH(ctx, k3_sub_scan, dim3(1), dim3(1024), 0, ctx->stream, u->unc_d, u->frame_off_d, u->n_utt, u->sub_off_d);
          KHG_LAUNCH(ctx, k3_sub_keys, dim3((unsigned)std::min(u->n_utt, 8192)), dim3(64), 0, ctx->stream, a, u->unc_d, u->frame_off_d, u->sub_off_d, u->n_utt, u->sort_keys_d, u->sort_vals_d);
        } else if (tm->num_tids <= K3_LDS_TIDS) KHG_LAUNCH(ctx, k3_sort_keys<true>, dim3(std::min(gb, 1024)), dim3(256), 0, ctx->stream, a, u->sort_keys_d, u->sort_vals_d);
        else KHG_LAUNCH(ctx, k3_sort_keys<false>, dim3(std::min(gb, 1024)), dim3(256), 0, ctx->stream, a, u->sort_keys_d, u->sort_vals_d);
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(u->sort_tmp_d, need, u->sort_keys_d, u->sort_keys_out_d, u->sort_vals_d,
                                                  reinterpret_cast<uint32_t*>(u->frame_ids_d), (int)Neff, 0, bits, ctx->stream));
        KHG_LAUNCH(ctx, k3_bounds, dim3((m->P + 256) / 256), dim3(256), 0, ctx->stream, a, u->sort_keys_out_d);
      }
    }
    // Work items of the accumulate kernels (k3_make_items): ny_base slices per pdf, more for a pdf whose bucket is far above the
    // average slice (2 x; silence in real transcripts).  -> the number of blocks to launch for a range of np pdfs (an upper bound
    // from N and P alone: the bucket sizes stay on the device) in *extra_blocks; parked: the slices park images (wave forms).
    int64_t k3_extra_blocks = 0;
    // Work items of one CLASS of pdfs (cls_lo < Gaussians <= cls_hi; the others get none): the form is chosen per pdf, so one pdf
    // that Split (csrc/diag-gmm.cc:780-851) pushed past 64 Gaussians does not take every other pdf off the wave form.  Two classes
    // keep their item lists side by side in the same buffers (second = true: the upper halves).
    auto make_items = [&](int ny_base, bool parked, size_t nsum1, int cls_lo, int cls_hi, bool second) -> int {
      const int64_t avg = Neff / std::max(1, m->P);
      const int target = (int)std::min<int64_t>(1 << 30, std::max<int64_t>(512, 2 * avg / ny_base));
      const int64_t per_t = Neff / target;
      k3_extra_blocks = std::min<int64_t>(per_t + m->P, 2 * per_t) + 1;
      const int64_t max_items = (int64_t)m->P * ny_base + k3_extra_blocks;
      const int64_t max_slots = !parked ? INT_MAX : ny_base > 1 ? max_items : 2 * per_t + 1;
      if (max_items >= INT_MAX / 2) return khg_set_error(KHG_E_UNSUPPORTED, "khg_acc_stats: too many work items");
      if (u->k3_items_n < (size_t)max_items) {
        DEVFREE(u->k3_items_d);
        { int ri = u_alloc(u, reinterpret_cast<K3Item**>(&u->k3_items_d), 2 * (size_t)max_items); if (ri) return ri; }
        u->k3_items_n = (size_t)max_items;
      }
      if (u->k3_item_off_n < (size_t)m->P + 1) {
        DEVFREE(u->k3_item_off_d);
        int rc2 = u_alloc(u, &u->k3_item_off_d, 2 * ((size_t)m->P + 1));
        if (rc2) return rc2;
        u->k3_item_off_n = (size_t)m->P + 1;
      }
      if (parked && u->k3_part_n < (size_t)max_slots * nsum1) {
        DEVFREE(u->k3_part_d);
        int rc2 = u_alloc(u, &u->k3_part_d, (size_t)max_slots * nsum1);
        if (rc2) return rc2;
        u->k3_part_n = (size_t)max_slots * nsum1;
      }
      K3Item* items = reinterpret_cast<K3Item*>(u->k3_items_d) + (second ? u->k3_items_n : 0);
      int32_t* item_off = u->k3_item_off_d + (second ? u->k3_item_off_n : 0);
      KHG_LAUNCH(ctx, k3_make_items, dim3(1), dim3(1024), 0, ctx->stream, a, ny_base, target, (int)max_items, (int)std::min<int64_t>(max_slots, INT_MAX),
                         items, item_off, cls_lo, cls_hi);
      a.items = items; a.item_off = item_off;
      return KHG_OK;
    };
    const int k3form = ctx->opt[KHG_OPT_K3_FORM];     // 1: the chunk-per-block MFMA form for every shape; 2: the VALU form
    // ---- the wave-local form (pdfs of <= 64 Gaussians at D <= 40): W in LDS + per-wave planes during the tile loop, the fp64 fold image afterwards ----
    auto run_wave = [&](int cls_lo, int cls_hi, int maxG, void* comm_) -> int {
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
      const int64_t avg_tiles = (Neff / std::max(1, m->P) + 15) / 16;
      int ny = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(32, (avg_tiles + 15) / 16), (4096 + m->P - 1) / m->P));
      if (ctx->opt[KHG_OPT_K3_NY] > 0) ny = ctx->opt[KHG_OPT_K3_NY];
      // per-pdf log-like partials (always) and, with several blocks per pdf, the slice images they park
      const size_t nsum1 = (size_t)nb * 16 * 80 + (size_t)nb * 16 + 1;
      if (u->k3_llpart_n < (size_t)m->P) {
        DEVFREE(u->k3_llpart_d);
        rc = u_alloc(u, &u->k3_llpart_d, (size_t)m->P);
        if (rc) return rc;
        u->k3_llpart_n = (size_t)m->P;
      }
      rc = make_items(ny, true, nsum1, cls_lo, cls_hi, false);
      if (rc) return rc;
      HIPCHK(hipMemsetAsync(u->k3_llpart_d, 0, sizeof(double) * (size_t)m->P, ctx->stream));
      a.ll_part = u->k3_llpart_d;
      a.part = u->k3_part_d;
      const int nparts_ = comm_ ? nparts : 1;
      for (int part = 0; part < nparts_; ++part) {
      const int p0 = (int)((int64_t)m->P * part / nparts_), np = (int)((int64_t)m->P * (part + 1) / nparts_) - p0;
      a.pdf0 = p0; a.npdf = np;
      const unsigned nblk = (unsigned)((int64_t)np * ny + k3_extra_blocks);
      {
      KernelTimer kt(ctx, nsub < 0 ? "k3_accumulate" : "k3_accumulate_pass2");
      // phase B on the fp64 matrix pipe (products exact, N ranks sum to the one-rank statistics to 1e-12) or on the fp16 matrix
      // cores (default where they apply)
      const bool exact_b = true;
      // k3_accumulate_wave16: per wave two tiles' phase-A planes + the phase-B planes (halves), then the split W operands
      const size_t lds16 = std::max<size_t>(2 * (size_t)4 * (2 * (2 * 16 * K3_XH_ROW) + 2 * 2 * 40 * 36) + (size_t)nb * 3 * 2 * 64 * 16,
                                            sizeof(double) * ((size_t)nb * 16 * 80 + (size_t)nb * 16));
#define K3_WAVE_LAUNCH(NBV)                                                                                            \
  do {                                                                                                                  \
    if (f16b) {                                                                                                          \
      HIPCHK(hipFuncSetAttribute((const void*)k3_accumulate_wave16<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16)); \
      KHG_LAUNCH(ctx, (k3_accumulate_wave16<NBV>), dim3(nblk), dim3(256), lds16, ctx->stream, a);                     \
    } else if (exact_b && f16a) KHG_LAUNCH(ctx, (k3_accumulate_wave<NBV, true>), dim3(nblk), dim3(256), lds, ctx->stream, a);    \
    else KHG_LAUNCH(ctx, (k3_accumulate_wave<NBV>), dim3(nblk), dim3(256), lds, ctx->stream, a);                      \
    KHG_LAUNCH(ctx, (k3_wave_finalize<NBV>), dim3(np), dim3(256), 0, ctx->stream, a);                                 \
  } while (0)
      switch (nb) {
        case 1: K3_WAVE_LAUNCH(1); break;
        case 2: K3_WAVE_LAUNCH(2); break;
        case 3: K3_WAVE_LAUNCH(3); break;
        default: K3_WAVE_LAUNCH(4); break;
      }
#undef K3_WAVE_LAUNCH
      }
      if (comm_ && nparts_ > 1) { rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm_, nullptr); if (rc) return rc; }
      }
      a.pdf0 = 0; a.npdf = m->P;
      KHG_LAUNCH(ctx, k3_wave_scalars, dim3(1), dim3(1024), 0, ctx->stream, a, cls_lo, cls_hi);
      return KHG_OK;
    };
    // ---- fp32 + fp64 MFMA form; fewer, longer blocks: the fp64 accumulators stay in registers per block ----
    auto run_mfma = [&](int cls_lo, int cls_hi, int maxG, int n_cls, bool second, void* comm_) -> int {
      const int64_t avg_chunks = (Neff / std::max(1, m->P) + K3_CHUNK - 1) / K3_CHUNK;
      const size_t lds = sizeof(float) * ((size_t)4 * K3_CHUNK * 2 * m->KQ + 5 * K3_CHUNK);   // 4 planes [64][KH] + reductions
      // slices per pdf: every block ends with one fp64 atomic per accumulator cell (G*(2D+1) of them), so
      // use as few blocks as still fill the chip (~4096 = 256 CUs x 8 blocks x 2 rounds)
      // (a class of a few pdfs -- the ones a split has grown -- is cut finer: one block walking a whole bucket is a latency of its own)
      int ny = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(32, avg_chunks), (4096 + n_cls - 1) / std::max(1, n_cls)));
      if (ctx->opt[KHG_OPT_K3_NY] > 0) ny = ctx->opt[KHG_OPT_K3_NY];
      rc = make_items(ny, false, 0, cls_lo, cls_hi, second);
      if (rc) return rc;
      // Both phases on the fp16 matrix cores (k3_accumulate_block16) where the model-derived scales hold and the weight is an ordinary
      // number (the conditions of the wave form's fp16 phases): <= 128 Gaussians at D <= 80, <= 192 at D <= 40.  KHG_K3_PHASEA=f32 or
      // KHG_K3_PHASEB=f64 keep the fp32 / fp64 MFMA form.
      bool b16 = false;
      if (ctx->opt[KHG_OPT_K3_PHASE_A] == 0 && ctx->opt[KHG_OPT_K3_PHASE_B] == 2 && ((m->KQ == 20 && maxG <= 128) || (m->KQ == 10 && maxG <= 192)) &&
          std::isfinite(weight) && std::fabs(weight) > 1.0e-30f && std::fabs(weight) < 1.0e30f) {
        rc = k3_phase_a_scales(ctx, const_cast<khg_model*>(m), u, &b16);
        if (rc) return rc;
      }
      if (b16) {
        a.pa_ex = m->k3_ex_d; a.pa_S = m->k3_S;
        a.pa_scale = std::ldexp(1.0f, m->k3_S); a.pa_inv = std::ldexp(1.0f, -m->k3_S); a.pa_c1 = std::ldexp(1.44269504088896340736f, -m->k3_S);
        a.pb_SG = 13 - std::ilogb(std::fabs(weight)); a.pb_gscale = std::ldexp(1.0f, a.pb_SG);
      }
      // split planes of a chunk in both layouts + the softmax's exchange: [2][64][KPAD + 8] + [2][2 XP][72] halves, 5 x 64 floats
      const size_t kpad = ((size_t)8 * m->KQ + 31) / 32 * 32;
      // ... or, before the chunk loop, the pdf's parameter rows [2][lanes' Gaussians][D | 1] + the columns' exponents
      const size_t gp_b16 = maxG <= 64 ? 64 : (maxG <= 128 ? 128 : 192);
      const size_t lds_b16 = std::max<size_t>(2 * (2 * (size_t)K3_CHUNK * (kpad + 8) + 2 * (size_t)8 * m->KQ * (K3_CHUNK + 8)) + sizeof(float) * 9 * K3_CHUNK,
                                              sizeof(float) * (2 * gp_b16 * (size_t)(m->D | 1) + 8 * (size_t)m->KQ));
      const int nparts_ = comm_ ? nparts : 1;
      for (int part = 0; part < nparts_; ++part) {
        const int p0 = (int)((int64_t)m->P * part / nparts_), np = (int)((int64_t)m->P * (part + 1) / nparts_) - p0;
        a.pdf0 = p0; a.npdf = np;
        const unsigned nblk = (unsigned)((int64_t)std::min(np, n_cls) * ny + k3_extra_blocks);
        {
          KernelTimer kt(ctx, nsub < 0 ? "k3_accumulate" : "k3_accumulate_pass2");
          if (b16) {
#define K3_B16(KQV, NBWV, NWVV)                                                                                                 \
  do {                                                                                                                          \
    HIPCHK(hipFuncSetAttribute((const void*)k3_accumulate_block16<KQV, NBWV, NWVV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b16)); \
    KHG_LAUNCH(ctx, (k3_accumulate_block16<KQV, NBWV, NWVV>), dim3(nblk), dim3(64 * NWVV), lds_b16, ctx->stream, a);            \
  } while (0)
            // (one Gaussian block per wave: < 256 registers, so a block of eight waves runs two per SIMD)
            if (maxG <= 64) { if (m->KQ == 20) K3_B16(20, 1, 4); else K3_B16(10, 1, 4); }
            else if (maxG <= 128) { if (m->KQ == 20) K3_B16(20, 1, 8); else K3_B16(10, 1, 8); }
            else K3_B16(10, 3, 4);         // (four blocks per wave -- 193..256 Gaussians -- spill at 512 registers: the fp32 / fp64 form keeps them)
#undef K3_B16
          } else
          if (m->KQ == 10 && maxG <= 64) KHG_LAUNCH(ctx, (k3_accumulate_mfma<10, 1>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 10 && maxG <= 128) KHG_LAUNCH(ctx, (k3_accumulate_mfma<10, 2>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 10 && maxG <= 192) KHG_LAUNCH(ctx, (k3_accumulate_mfma<10, 3>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 10) KHG_LAUNCH(ctx, (k3_accumulate_mfma<10, 4>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (maxG <= 64) KHG_LAUNCH(ctx, (k3_accumulate_mfma<20, 1>), dim3(nblk), dim3(256), lds, ctx->stream, a);
          else KHG_LAUNCH(ctx, (k3_accumulate_mfma<20, 2>), dim3(nblk), dim3(256), lds, ctx->stream, a);
        }
        if (comm_ && nparts_ > 1) { rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm_, nullptr); if (rc) return rc; }
      }
      a.pdf0 = 0; a.npdf = m->P;
      return KHG_OK;
    };
    // ---- the VALU form: any number of Gaussians, any dimension ----
    auto run_valu = [&](int cls_lo, int cls_hi, int maxG, int n_cls, bool second, void* comm_) -> int {
      const int64_t avg_chunks = (Neff / std::max(1, m->P) + K3_CHUNK - 1) / K3_CHUNK;
      const size_t lds = sizeof(float) * (size_t)K3_CHUNK * ((size_t)(m->KQ ? 4 * m->KQ : (m->D | 1)) + (maxG | 1) + 4);
      if (lds > 160 * 1024) return khg_set_error(KHG_E_UNSUPPORTED, "khg_acc_stats: pdf too large for the LDS chunk buffers");
      const void* k3fn = m->KQ == 10 ? (const void*)k3_accumulate<10> : m->KQ == 20 ? (const void*)k3_accumulate<20> : (const void*)k3_accumulate<0>;
      if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(k3fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int ny = (int)std::max<int64_t>(1, std::min<int64_t>(64, (avg_chunks + 3) / 4));
      rc = make_items(ny, false, 0, cls_lo, cls_hi, second);
      if (rc) return rc;
      const int nparts_ = comm_ ? nparts : 1;
      for (int part = 0; part < nparts_; ++part) {
        const int p0 = (int)((int64_t)m->P * part / nparts_), np = (int)((int64_t)m->P * (part + 1) / nparts_) - p0;
        a.pdf0 = p0; a.npdf = np;
        const unsigned nblk = (unsigned)((int64_t)std::min(np, n_cls) * ny + k3_extra_blocks);
        {
          KernelTimer kt(ctx, nsub < 0 ? "k3_accumulate" : "k3_accumulate_pass2");
          if (m->KQ == 10) KHG_LAUNCH(ctx, k3_accumulate<10>, dim3(nblk), dim3(256), lds, ctx->stream, a);
          else if (m->KQ == 20) KHG_LAUNCH(ctx, k3_accumulate<20>, dim3(nblk), dim3(256), lds, ctx->stream, a);
          else KHG_LAUNCH(ctx, k3_accumulate<0>, dim3(nblk), dim3(256), lds, ctx->stream, a);
        }
        if (comm_ && nparts_ > 1) { rc = accs_allreduce_pieces(ctx, acc, m, p0, np, comm_, nullptr); if (rc) return rc; }
      }
      a.pdf0 = 0; a.npdf = m->P;
      return KHG_OK;
    };
    // Form per pdf: the wave form takes every pdf of <= 64 Gaussians (D <= 40); what is left -- a few pdfs a split grew, or a model
    // of wide pdfs throughout -- goes to the chunk-per-block MFMA form (<= 256 Gaussians at D <= 40, <= 128 at D <= 80) or the VALU form.
    int maxG = 0, maxG_lo = 0, n_hi = 0;
    for (int p = 0; p < m->P; ++p) {
      const int Gp = m->gauss_off[p + 1] - m->gauss_off[p];
      maxG = std::max(maxG, Gp);
      if (Gp <= 64) maxG_lo = std::max(maxG_lo, Gp); else ++n_hi;
    }
    const bool mfma_ok = (maxG <= 128 || (maxG <= 256 && m->KQ == 10)) && k3form != 2 && m->KQ != 0;
    const bool wave_ok = m->KQ == 10 && k3form == 0 && maxG_lo > 0;
    if (wave_ok && n_hi == 0) rc = run_wave(0, INT_MAX, maxG, comm);
    else if (wave_ok) {
      rc = run_wave(0, 64, maxG_lo, nullptr);                  // (with an exchange: its pieces follow the second class's kernels)
      if (!rc) rc = mfma_ok ? run_mfma(64, INT_MAX, maxG, n_hi, true, comm) : run_valu(64, INT_MAX, maxG, n_hi, true, comm);
    } else if (mfma_ok) rc = run_mfma(0, INT_MAX, maxG, m->P, false, comm);
    else rc = run_valu(0, INT_MAX, maxG, m->P, false, comm);
    if (rc) return rc;
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
static int acc_stats_impl(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc, void* comm, int nparts) {
  if (ctx_dead(ctx) || !m || !tm || !u || !acc) return khg_set_error(KHG_E_ARG, "khg_acc_stats: bad arguments");
  { int rf = utts_foreign_ctx(ctx, u, "khg_acc_stats"); if (rf) return rf; }
  if (!u->ali_valid) return khg_set_error(KHG_E_ARG, "khg_acc_stats: no resident alignment (khg_align or khg_ali_upload first)");
  if (m->D != u->D || acc->D != m->D || acc->sumG != m->sumG || acc->num_tids != tm->num_tids)
    return khg_set_error(KHG_E_RUNTIME, "khg_acc_stats: accumulator / model / feature dimensions do not match");
  if (tm->max_pdf >= m->P) return khg_set_error(KHG_E_RUNTIME, "khg_acc_stats: transition model refers to pdf-ids the model does not have");
  // Split mode (khg_k2.hip): the order-faithful decoders may still be running on their side stream, writing to their own buffer.  The DP
  // kernel counted what it left to them: wait for IT (the host joins the stream once, right behind K2), and if there is anything,
  // accumulate the certified utterances now -- their ranges of ali_d are final, the others' are zero -- and the rest in a second pass
  // once the decoders are done.  Statistics are additive (csrc/mle-am-diag-gmm.cc:41-52); the second pass is the same kernels over
  // the flagged utterances' frames.
  int64_t nsub = -1;
  if (u->ali_pending && u->ali_split && u->N > 0) {
    HIPCHK(hipEventSynchronize(u->ev_dp));
    if (u->unc_cnt_h[0] == 0) u->ali_split = false;         // nothing to merge, nothing to wait for but the (empty) side stream
    else nsub = u->unc_cnt_h[1];
  }
  int rc = nsub < 0 ? wait_ali(ctx, u) : KHG_OK;
  if (!rc) rc = arena_flush(ctx);
  if (rc) return rc;
  if (!u->frame_ids_d || u->k3_P != m->P || u->k3_tids != tm->num_tids) {
    DEVFREE(u->pdf_count_d); DEVFREE(u->pdf_cursor_d); DEVFREE(u->pdf_start_d); DEVFREE(u->tid_count_d); DEVFREE(u->frame_ids_d);
    rc = u_alloc(u, &u->pdf_count_d, (size_t)m->P);
    if (!rc) rc = u_alloc(u, &u->pdf_cursor_d, (size_t)m->P);
    if (!rc) rc = u_alloc(u, &u->pdf_start_d, (size_t)m->P + 1);
    if (!rc) rc = u_alloc(u, &u->tid_count_d, (size_t)tm->num_tids + 1);
    if (!rc) rc = u_alloc(u, &u->frame_ids_d, (size_t)u->N);
    if (rc) return rc;
    u->k3_P = m->P; u->k3_tids = tm->num_tids;
  }
  if (nsub < 0) return acc_stats_pass(ctx, m, tm, u, weight, acc, comm, nparts, -1);
  rc = acc_stats_pass(ctx, m, tm, u, weight, acc, nullptr, 1, -1);          // the exchange (if any) follows the second pass
  if (!rc) rc = wait_ali(ctx, u);                                             // decoders done -> their alignments merged into ali_d
  if (!rc) rc = acc_stats_pass(ctx, m, tm, u, weight, acc, comm, nparts, nsub);
  return rc;
}
extern "C" int khg_acc_stats(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc) {
  return acc_stats_impl(ctx, m, tm, u, weight, acc, nullptr, 1);
}
extern "C" int khg_acc_stats_reduce(khg_ctx* ctx, const khg_model* m, const khg_tm* tm, khg_utts* u, float weight, khg_accs* acc, void* comm, int32_t nparts) {
  return acc_stats_impl(ctx, m, tm, u, weight, acc, comm, nparts <= 0 ? 4 : nparts);
}
