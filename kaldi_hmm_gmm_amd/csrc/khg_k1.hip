// kaldi_hmm_gmm_amd/csrc/khg_k1.hip -- C-ABI (include/khg_hip.h): K1, the log-likelihoods (khg_loglikes / _reachable / _band): form
// selection, scale exponents and domain checks of the split forms, per-set plans (chunks, units, walks) and the launches.  gfx950 only.
#include "khg_internal.hpp"

#include "khg_k1_loglikes.hip.inc"
#include "khg_k1_pdfmajor.hip.inc"
#include "khg_k1_split_common.hip.inc"
#include "khg_k1_f16x2.hip.inc"
#include "khg_k1_wide.hip.inc"
#include "khg_k1_f16x2s.hip.inc"

// ------------------------------------------------------------------------------------------
// K1
template <int KQ, int NF, int WPS>
static void launch_k1(khg_ctx* ctx, const K1Args& a, int nchunks, bool aligned, hipStream_t s) {
  if (aligned) KHG_LAUNCH(ctx, (k1_loglikes<KQ, NF, true, WPS>), dim3(nchunks), dim3(256), 0, s, a);
  else KHG_LAUNCH(ctx, (k1_loglikes<KQ, NF, false, WPS>), dim3(nchunks), dim3(256), 0, s, a);
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
    int rc = u_upload(ctx, u, &u->wchunks_d, ch);
    if (rc) return rc;
    { int rs = sync_pageable(ctx); if (rs) return rs; }
  }
  KwArgs a;
  a.feats = u->feats_d; a.frame_off = u->frame_off_d; a.chunks = u->wchunks_d; a.pdf_off = u->pdf_off_d; a.pdfs = u->pdfs_d;
  a.gauss_off = m->gauss_off_d; a.gconsts = m->gconsts_d; a.means_invvars = m->miv_d; a.nhalf_inv_vars = m->nhiv_d;
  a.ll = u->ll_d; a.ll_off = u->ll_off_d; a.err_flag = ctx->err_flag_d; a.D = m->D;
  const size_t lds = sizeof(float) * 64 * (size_t)(m->D | 1);
  if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k1w_loglikes, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (u->n_wchunks > 0) {
    KernelTimer kt(ctx, "k1_loglikes");
    KHG_LAUNCH(ctx, k1w_loglikes, dim3(u->n_wchunks), dim3(256), lds, ctx->stream, a);
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
    rc = u_upload(ctx, u, &u->utt_xtile_off_d, xoff);
    if (!rc) rc = dev_upload(ctx, &xutt_d, xutt);
    if (!rc) rc = u_alloc(u, &u->xpl_d, (size_t)std::max<int64_t>(nx, 1) * 2 * 16 * KH);
    if (!rc && nx > 0) {
      const int gb = (int)std::min<int64_t>(65535, (nx * (2 * 16 * KH / 4) + 255) / 256);
      if (m->KQ == 10) KHG_LAUNCH(ctx, k1p_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_xtile_off_d, xutt_d, nx, u->D, u->xpl_d);
      else KHG_LAUNCH(ctx, k1p_pack_x<20>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_xtile_off_d, xutt_d, nx, u->D, u->xpl_d);
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
    rc = u_upload(ctx, u, &u->p_ents_d, ents);
    if (!rc) rc = u_upload(ctx, u, &u->p_slices_d, slices);
    if (rc) return rc;
    { int rs = sync_pageable(ctx); if (rs) return rs; }
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
#define K1P_LAUNCH(KQ_, NB_, WPS_) KHG_LAUNCH(ctx, (k1p_loglikes<KQ_, NB_, WPS_>), dim3(n), dim3(256), 0, ctx->stream, a)
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
// last-tile-of-pdf flag (bit 31).  Shared by the utterance-major fp32 kernel and the f16x2 kernel.
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
  int rc = u_upload(ctx, u, &u->tile_off_d, toff);
  if (!rc) rc = u_upload(ctx, u, &u->tiles_d, tiles);
  if (rc) return rc;
  { int rs = sync_pageable(ctx); if (rs) return rs; }
  u->tiles_pto = m->pdf_tile_off;
  u->tiles_reach = (int)reachable_only;
  return KHG_OK;
}

// 32-frame tile layout of the set shared by the f16x2 / f16x2s forms: tile offsets per utterance, tile -> utterance.
static int ensure_x32_layout(khg_ctx* ctx, khg_utts* u) {
  if (u->utt_x32_off_d) return KHG_OK;
  std::vector<int64_t> xoff((size_t)u->n_utt + 1, 0);
  for (int i = 0; i < u->n_utt; ++i) xoff[(size_t)i + 1] = xoff[(size_t)i] + (u->frame_off[i + 1] - u->frame_off[i] + 31) / 32;
  const int64_t nx = xoff[(size_t)u->n_utt];
  std::vector<int32_t> xutt((size_t)nx);
  for (int i = 0; i < u->n_utt; ++i)
    for (int64_t t = xoff[(size_t)i]; t < xoff[(size_t)i + 1]; ++t) xutt[(size_t)t] = i;
  int rc = u_upload(ctx, u, &u->utt_x32_off_d, xoff);
  if (!rc) rc = u_upload(ctx, u, &u->x32_utt_d, xutt);
  if (rc) return rc;
  { int rs = sync_pageable(ctx); if (rs) return rs; }
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
  // KHG_OPT_K1_ORDER (experiments): 0 frame tiles x pdfs descending (default), 1 utterance order, 2 ascending, 3 frame tiles descending,
  // 4 the chunks of ONE utterance eight positions apart: workgroups go to the eight XCDs round-robin, so siblings -- which walk the same
  // pdf list, i.e. the same W tiles, at about the same time -- share an L2 (the second one's W loads need not come from HBM)
  if (order == 4) {
    std::vector<int32_t> first(1, 0);                       // chunks are grouped by utterance: first[i] = first chunk of group i
    for (size_t i = 1; i < ch->size(); ++i) if ((*ch)[i].utt != (*ch)[i - 1].utt) first.push_back((int32_t)i);
    first.push_back((int32_t)ch->size());
    const int ng = (int)first.size() - 1;
    std::vector<int32_t> grp((size_t)ng);
    for (int i = 0; i < ng; ++i) grp[(size_t)i] = i;
    auto gcost = [&](int g) { int64_t t = 0; for (int k = first[(size_t)g]; k < first[(size_t)g + 1]; ++k) t += (*ch)[(size_t)k].ntiles; return t * (u->pdf_off[(*ch)[(size_t)first[(size_t)g]].utt + 1] - u->pdf_off[(*ch)[(size_t)first[(size_t)g]].utt]); };
    auto gn = [&](int g) { return first[(size_t)g + 1] - first[(size_t)g]; };
    std::stable_sort(grp.begin(), grp.end(), [&](int a, int b) { return gn(a) != gn(b) ? gn(a) > gn(b) : gcost(a) > gcost(b); });
    std::vector<Chunk> out;
    out.reserve(ch->size());
    for (int b = 0; b < ng; b += 8) {                       // a band of eight utterances: round r = chunk r of each
      const int nb = std::min(8, ng - b);
      int rounds = 0;
      for (int i = 0; i < nb; ++i) rounds = std::max(rounds, gn(grp[(size_t)(b + i)]));
      for (int r = 0; r < rounds; ++r)
        for (int i = 0; i < nb; ++i) {
          const int g = grp[(size_t)(b + i)];
          if (r < gn(g)) out.push_back((*ch)[(size_t)(first[(size_t)g] + r)]);
          else out.push_back(Chunk{(*ch)[(size_t)first[(size_t)g]].utt, 0, 0, 0});      // (an empty workgroup keeps the band's XCD alignment)
        }
    }
    ch->swap(out);
    return;
  }
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
  rc = u_upload(ctx, u, &u->bchunks_d, ch);
  if (rc) return rc;
  { int rs = sync_pageable(ctx); if (rs) return rs; }
  u->n_bchunks = (int32_t)ch.size(); u->bchunk_nt = NTMAX;
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
    KHG_LAUNCH(ctx, k1h_absmax, dim3(gb), dim3(256), 0, ctx->stream, a_d, n, D, m_d);
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
int k1_maxima(khg_ctx* ctx, khg_model* m, khg_utts* u, std::vector<float>* xk) {
  const int D = m->D, K = 16 * m->KS;
  int rc = KHG_OK;
  if (u->xmax.empty()) { rc = absmax_cols(ctx, u->feats_d, u->N, D, &u->xmax); if (rc) return rc; }
  if (m->wmax.empty()) { rc = model_stats(ctx, m); if (rc) return rc; }      // one pass per parameter version (khg_ctx_model.hip)
  xk->assign((size_t)K, 0.0f);
  for (int d = 0; d < D; ++d) { (*xk)[(size_t)2 * d] = u->xmax[(size_t)d]; (*xk)[(size_t)2 * d + 1] = u->xmax[(size_t)d] * u->xmax[(size_t)d]; }
  return KHG_OK;
}
// The split forms (f16x2, f16x2s) fold the log-sum-exp's subtraction into an fma (k1_exp2_le1): valid while every
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
      rc = u_alloc(u, &u->xh_d, (size_t)std::max<int64_t>(nx, 1) * 2 * KS * 64);
      if (rc) return rc;
    }
    DEVFREE(u->xh_ex_d);
    rc = u_upload(ctx, u, &u->xh_ex_d, ex);
    if (rc) return rc;
    if (nx > 0) {
      KernelTimer kt(ctx, "k1h_pack_x");
      const int gb = (int)std::min<int64_t>(65535, (nx * KS * 64 + 255) / 256);
      if (KS == 5) KHG_LAUNCH(ctx, k1h_pack_x<5>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xh_ex_d, u->xh_d);
      else KHG_LAUNCH(ctx, k1h_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xh_ex_d, u->xh_d);
      HIPCHK(hipGetLastError());
    }
    { int rs = sync_pageable(ctx); if (rs) return rs; }    // `ex` (pageable) is free after this
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
    if (KS == 5) KHG_LAUNCH(ctx, k0h_pack_tiles<5>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, u->xh_ex_d, m->wimgh_d);
    else KHG_LAUNCH(ctx, k0h_pack_tiles<10>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, u->xh_ex_d, m->wimgh_d);
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
      if (KS == 5) KHG_LAUNCH(ctx, (k1h_loglikes<5, 2>), dim3(u->n_bchunks), dim3(512), lds, ctx->stream, a);
      else KHG_LAUNCH(ctx, (k1h_loglikes<10, K1H_NT10>), dim3(u->n_bchunks), dim3(512), lds, ctx->stream, a);
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
  // (When the shared exponents push the absolute part of the error bound past its limit -- a set with far larger features scored
  //  against this model earlier -- and the set's OWN exponents hold it, the model follows this set: one re-pack instead of the
  //  slower two-accumulator form for every set from here on.)
  const std::vector<int32_t> ex_own = ex;
  int S = 0;
  auto scales = [&]() -> bool {         // S, ew and the floor for the current `ex`
    S = INT_MAX;
    for (int k = 0; k < K; ++k) if (m->wmax[(size_t)k] > 0.0f) S = std::min(S, 14 - std::ilogb(m->wmax[(size_t)k]) + ex[(size_t)k]);
    if (S == INT_MAX) S = 0;
    if (S < -100 || S > 100) return false;
    double floor_sum = 0.0;      // the absolute part of the error bound, at the column maxima (khg_k1_f16x2s.hip.inc)
    for (int k = 0; k < K; ++k) {
      ew[(size_t)k] = S - ex[(size_t)k];
      floor_sum += std::ldexp((double)m->wmax[(size_t)k], ew[(size_t)k]) + std::ldexp((double)xk[(size_t)k], ex[(size_t)k]);
    }
    return std::ldexp(floor_sum, -25 - S) <= 2.0e-6;
  };
  bool shared = m->xs_ex_seen.size() == ex.size();
  if (shared) for (int k = 0; k < K; ++k) ex[(size_t)k] = std::min(ex[(size_t)k], m->xs_ex_seen[(size_t)k]);
  if (!scales()) {
    if (!shared || ex == ex_own) return 1;
    ex = ex_own;
    if (!scales()) return 1;
  }
  m->xs_ex_seen = ex;
  rc = ensure_x32_layout(ctx, u);
  if (rc) return rc;
  if (!u->schunks_d || u->schunk_nmax != NMAX) {
    DEVFREE(u->schunks_d);
    std::vector<K1sChunk> ch;
    // (one or two utterances -- the per-utterance call pattern -- cannot fill the chip with whole utterances: short chunks spread them
    //  over more workgroups; the W tiles are re-read per chunk from the cache, a band that crosses chunks keeps its aligned tiles)
    // (D > 40: the W image does not fit the Infinity Cache and an utterance is two or three chunks -- their order puts siblings on one XCD:
    //  K1 55.6 -> 52.2 ms at 10 000 x 128 x 80 / 20 000 utterances; at D <= 40 it changes nothing: 56.9 ms either way)
    const int order = (ctx->opt[KHG_OPT_K1_ORDER] == 0 && KS == 10 && !u->small) ? 4 : ctx->opt[KHG_OPT_K1_ORDER];
    plan_x32_chunks(u, (u->small && u->n_utt <= 2) ? std::min(NMAX, 3) : NMAX, order, &ch);
    rc = u_upload(ctx, u, &u->schunks_d, ch);
    if (rc) return rc;
    { int rs = sync_pageable(ctx); if (rs) return rs; }
    u->n_schunks = (int32_t)ch.size(); u->schunk_nmax = NMAX;
  }
  // the set's B fragments: packed once (the exponents depend on the features alone)
  if (!u->xs_d || u->xs_ks != KS || u->xs_ex != ex) {
    const int64_t nx = u->n_x32;
    if (!u->xs_d || u->xs_ks != KS) {
      DEVFREE(u->xs_d);
      rc = u_alloc(u, &u->xs_d, (size_t)std::max<int64_t>(nx, 1) * 2 * KS * 64);
      if (rc) return rc;
    }
    DEVFREE(u->xs_ex_d);
    rc = u_upload(ctx, u, &u->xs_ex_d, ex);
    if (rc) return rc;
    if (nx > 0) {
      KernelTimer kt(ctx, "k1s_pack_x");
      const int gb = (int)std::min<int64_t>(65535, (nx * KS * 64 + 255) / 256);
      if (KS == 5) KHG_LAUNCH(ctx, k1s_pack_x<5>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xs_ex_d, u->xs_d);
      else KHG_LAUNCH(ctx, k1s_pack_x<10>, dim3(gb), dim3(256), 0, ctx->stream, u->feats_d, u->frame_off_d, u->utt_x32_off_d, u->x32_utt_d, nx, D, u->xs_ex_d, u->xs_d);
      HIPCHK(hipGetLastError());
    }
    { int rs = sync_pageable(ctx); if (rs) return rs; }    // `ex` (pageable) is free after this
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
      if (KS == 5) KHG_LAUNCH(ctx, k0s_pack_tiles<5>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, ew_d, gscale, m->wimgs_d);
      else KHG_LAUNCH(ctx, k0s_pack_tiles<10>, dim3(m->ntiles), dim3(256), 0, ctx->stream, m->gconsts_d, m->miv_d, m->iv_d, m->gauss_off_d, m->pdf_tile_off_d, m->tile_pdf_d, D, ew_d, gscale, m->wimgs_d);
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
    rc = u_upload(ctx, u, &u->sunits_d, units);
    if (rc) return rc;
    { int rs = sync_pageable(ctx); if (rs) return rs; }
    u->sunits_pto = m->pdf_tile_off;
    u->sunits_reach = units_key;
  }
  // BAND form: the per-pdf upper bounds the skipped tiles are filled with, indexed by a pdf's first W tile; per parameter version
  if ((band || tail_shift) && !m->ubound_valid) {       // (filled by model_stats with the column maxima; only a stale flag gets here)
    m->wmax.clear();
    rc = model_stats(ctx, m);
    if (rc) return rc;
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
    if (band) {
      if (!u->band_args) u->band_args = new K1sArgs();
      *static_cast<K1sArgs*>(u->band_args) = a; u->band_model = m; u->band_ks = KS; u->band_lds = lds;
      u->band_serial = m->serial; u->band_version = m->version; u->band_key = m->wimgs_key;
    }
    const void* fn = pack == 4 ? (const void*)k1s_loglikes_packed<5, 4> : pack == 2 ? (const void*)k1s_loglikes_packed<5, 2>
                     : KS == 5 ? (const void*)k1s_loglikes<5> : (const void*)k1s_loglikes<10>;
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    {
      KernelTimer kt(ctx, "k1_loglikes");
      if (pack == 4) KHG_LAUNCH(ctx, (k1s_loglikes_packed<5, 4>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
      else if (pack == 2) KHG_LAUNCH(ctx, (k1s_loglikes_packed<5, 2>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
      else if (KS == 5) KHG_LAUNCH(ctx, (k1s_loglikes<5>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
      else KHG_LAUNCH(ctx, (k1s_loglikes<10>), dim3(u->n_schunks), dim3(512), lds, ctx->stream, a);
    }
    HIPCHK(hipGetLastError());
    rc = m->wimgs_sync.after_read(ctx->stream);
    if (rc) return rc;
  }
  u->ll_valid = true;
  return KHG_OK;
}

static int loglikes_impl(khg_ctx* ctx, const khg_model* m, khg_utts* u, int reach) {
  if (ctx_dead(ctx) || !m || !u) return khg_set_error(KHG_E_ARG, "khg_loglikes: bad arguments");
  { int rf = utts_foreign_ctx(ctx, u, "khg_loglikes"); if (rf) return rf; }
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
    rc = u_upload(ctx, u, &u->pdf_off_d, u->pdf_off);
    if (!rc) rc = u_upload(ctx, u, &u->pdfs_d, u->pdfs);
    if (!rc) rc = u_upload(ctx, u, &u->ll_off_d, u->ll_off);
    if (!rc) rc = u_alloc(u, &u->ll_d, (size_t)u->ll_total);
    if (rc) return rc;
  }
  {
    // which K1: f16x2s (default: the fp16 matrix cores at fp32 accuracy) -> f16x2 -> one of the fp32-MFMA forms -- pdf-major (pdfs of
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
    rc = u_upload(ctx, u, &u->chunks_d, ch);
    if (rc) return rc;
    { int rs = sync_pageable(ctx); if (rs) return rs; }  // ch is a local
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
    if (m->KQ == 10 && k1_nf(ctx, 10) == 6) launch_k1<10, 6, 2>(ctx, a, u->n_chunks, aligned, ctx->stream);
    else if (m->KQ == 10) launch_k1<10, 5, 2>(ctx, a, u->n_chunks, aligned, ctx->stream);
    else launch_k1<20, 5, 1>(ctx, a, u->n_chunks, aligned, ctx->stream);
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
  if (ctx_dead(ctx) || !u || !ll) return khg_set_error(KHG_E_ARG, "bad arguments");
  if (!u->ll_valid) return khg_set_error(KHG_E_ARG, "khg_loglikes_download: call khg_loglikes first");
  { int rf = utts_foreign_ctx(ctx, u, "khg_loglikes_download"); if (rf) return rf; }
  int rc = check_err_flag(ctx, "khg_loglikes");
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(ll, u->ll_d, sizeof(float) * (size_t)u->ll_total, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_loglikes_upload(khg_ctx* ctx, khg_utts* u, const float* ll) {
  if (ctx_dead(ctx) || !u || !ll) return khg_set_error(KHG_E_ARG, "bad arguments");
  { int rf = utts_foreign_ctx(ctx, u, "khg_loglikes_upload"); if (rf) return rf; }
  int rc = wait_ali(ctx, u);
  if (rc) return rc;
  if (!u->pdf_off_d) {
    rc = u_upload(ctx, u, &u->pdf_off_d, u->pdf_off);
    if (!rc) rc = u_upload(ctx, u, &u->pdfs_d, u->pdfs);
    if (!rc) rc = u_upload(ctx, u, &u->ll_off_d, u->ll_off);
    if (!rc) rc = u_alloc(u, &u->ll_d, (size_t)u->ll_total);
    if (rc) return rc;
  }
  rc = arena_flush(ctx);      // (staged uploads first: none may land on the score block after this copy)
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(u->ll_d, ll, sizeof(float) * (size_t)u->ll_total, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  u->ll_mode = 0; u->band_model = nullptr;      // the caller's scores: every cell as given
  u->ll_valid = true;
  return KHG_OK;
}

// BAND form: khg_align's repair launch reads the model the scores were computed with -- its fp16 image, its upper bounds -- through
// pointers saved at khg_loglikes_band.  Before anything is launched: that handle must still be alive (by serial, not by address) and
// at the same parameter version; and if only its IMAGE was re-packed since (another set lowered the shared feature exponents), this
// set's planes no longer match it: the set is scored again, in the band form, against the current image.
int k1_band_check(khg_ctx* ctx, khg_utts* u) {
  if (u->ll_mode != 2 || !u->band_args) return KHG_OK;
  khg_model* bm = khg_model_lookup(u->band_serial);
  if (!bm || bm != u->band_model || bm->version != u->band_version) {
    u->ll_valid = false;
    return khg_set_error(KHG_E_ARG, "khg_align: the model these scores were computed with (khg_loglikes_band) was destroyed or updated since; call khg_loglikes_band again");
  }
  if (bm->wimgs_key != u->band_key) return loglikes_impl(ctx, bm, u, 2);
  return KHG_OK;
}

void k1_free_band(khg_utts* u) {
  delete static_cast<K1sArgs*>(u->band_args);
  u->band_args = nullptr; u->band_model = nullptr;
}

// BAND form of K1: the utterances the DP could not certify are about to be decoded by the order-faithful kernel, which reads
// every cell a token reaches -- also the ones the band left at their upper bound.  Recompute exactly those utterances (from
// their first needed tile on, no upper limit) on `side`; a workgroup of any other utterance returns at once.
int k1_band_repair(khg_ctx* ctx, khg_utts* u, int32_t* status_d, int repair_bit, hipStream_t side) {
  if (u->ll_mode != 2 || !u->band_model || !u->band_args || u->n_schunks <= 0) return KHG_OK;
  K1sArgs ra = *static_cast<K1sArgs*>(u->band_args);
  ra.repair_status = status_d; ra.repair_bit = repair_bit;
  khg_model* bm = u->band_model;
  int rc = bm->wimgs_sync.before_read(side);
  if (rc) return rc;
  {
    KernelTimer kt(ctx, "k1_band_repair", side);
    const size_t lds_r = u->band_lds + 4 * 513;          // + the list of flagged chunks
    const unsigned gr = (unsigned)std::min<int64_t>(256, ((int64_t)u->n_schunks + 511) / 512);
    if (u->band_ks == 5) {
      HIPCHK(hipFuncSetAttribute((const void*)k1s_repair<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
      KHG_LAUNCH(ctx, (k1s_repair<5>), dim3(gr), dim3(512), lds_r, side, ra, (int)u->n_schunks);
    } else {
      HIPCHK(hipFuncSetAttribute((const void*)k1s_repair<10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));
      KHG_LAUNCH(ctx, (k1s_repair<10>), dim3(gr), dim3(512), lds_r, side, ra, (int)u->n_schunks);
    }
  }
  HIPCHK(hipGetLastError());
  return bm->wimgs_sync.after_read(side);
}
