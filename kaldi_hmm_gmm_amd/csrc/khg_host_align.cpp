// Implementation of khg_host_align.hpp: see the header for what each piece mirrors in the reference.
#include "khg_host_align.hpp"

#include <cstdio>

namespace khg {

std::string FasterDecoderOptions::ToString() const {
  char buf[256];
  std::snprintf(buf, sizeof(buf), "FasterDecoderOptions(beam=%g, max_active=%d, min_active=%d, beam_delta=%g, hash_ratio=%g)", (double)beam, max_active,
                min_active, (double)beam_delta, (double)hash_ratio);
  return buf;
}

DecodableAmDiagGmmUnmapped::DecodableAmDiagGmmUnmapped(std::shared_ptr<AmDiagGmm> am, const float* feats, int64_t T, int D)
    : am_(std::move(am)), feats_(feats, feats + (size_t)T * D), T_(T), D_(D) {
  KHG_REQUIRE(am_ != nullptr, "DecodableAmDiagGmm: no model");
}
const std::vector<float>& DecodableAmDiagGmmUnmapped::Scores() const {
  if (ll_.empty() && T_ > 0) {
    KHG_REQUIRE(am_->Dim() == D_, "Dim mismatch: data dim = " + std::to_string(D_) + " vs. model dim = " + std::to_string(am_->Dim()));
    std::vector<int32_t> go, pdfs((size_t)am_->NumPdfs());
    std::vector<float> gc, miv, iv;
    am_->Flat(&go, &gc, nullptr, &miv, &iv);
    for (int p = 0; p < am_->NumPdfs(); ++p) pdfs[(size_t)p] = p;
    ll_ = GpuLoglikes(am_->NumPdfs(), D_, go.data(), gc.data(), miv.data(), iv.data(), feats_.data(), T_, pdfs.data(), am_->NumPdfs());
  }
  return ll_;
}
float DecodableAmDiagGmmUnmapped::ZeroBased(int frame, int state) const {
  KHG_REQUIRE(frame >= 0 && frame < NumFramesReady(), "frame < NumFramesReady() assertion failed");
  KHG_REQUIRE(state >= 0 && state < am_->NumPdfs(), "Likely graph/model mismatch, e.g. using wrong HCLG.fst");
  return Scores()[(size_t)state * (size_t)T_ + (size_t)frame];
}
bool DecodableAmDiagGmmUnmapped::IsLastFrame(int frame) const {
  KHG_REQUIRE(frame < NumFramesReady(), "frame < NumFramesReady() assertion failed");
  return frame == NumFramesReady() - 1;
}

namespace {
struct ModelH { khg_model* h = nullptr; ~ModelH() { if (h) khg_model_destroy(h); } };
struct TmH { khg_tm* h = nullptr; ~TmH() { if (h) khg_tm_destroy(h); } };
struct UttsH { khg_utts* h = nullptr; ~UttsH() { if (h) khg_utts_destroy(h); } };
std::string G(double x) { char b[64]; std::snprintf(b, sizeof(b), "%g", x); return b; }
}  // namespace

std::vector<AlignResult> AlignBatch(const AmDiagGmm& am, const TransitionModel& tm, const GraphsCsr& g, const std::vector<const float*>& feats,
                                    const std::vector<int64_t>& nframes, const AlignConfig& config, float acoustic_scale, const float* trans_cost,
                                    const FasterDecoderOptions* dopts, bool return_scores, float like_scale) {
  KHG_REQUIRE(!((config.retry_beam != 0 && config.retry_beam <= config.beam) || config.beam <= 0.0f),
              "Beams do not make sense: beam " + G(config.beam) + ", retry-beam " + G(config.retry_beam));   // csrc/decoder-wrappers.cc:29-33
  const int n_utt = (int)feats.size(), D = am.Dim();
  KHG_REQUIRE((int)nframes.size() == n_utt && (int)g.start.size() == n_utt && (int)g.state_off.size() == n_utt + 1, "AlignBatch: one graph and one feature matrix per utterance");
  khg_ctx* ctx = DefaultCtx();
  std::vector<int32_t> go;
  std::vector<float> gc, miv, iv;
  am.Flat(&go, &gc, nullptr, &miv, &iv);
  ModelH dm; TmH dt; UttsH us;
  CApi(khg_model_create(ctx, am.NumPdfs(), D, go.data(), gc.data(), miv.data(), iv.data(), &dm.h));
  std::vector<int32_t> id2pdf(tm.id2pdf().begin(), tm.id2pdf().end());
  CApi(khg_tm_create(ctx, tm.NumTransitionIds(), id2pdf.data(), &dt.h));
  CApi(khg_tm_set_trans_cost(dt.h, trans_cost));
  std::vector<int64_t> frame_off((size_t)n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) frame_off[(size_t)u + 1] = frame_off[(size_t)u] + nframes[(size_t)u];
  std::vector<float> all((size_t)std::max<int64_t>(frame_off[(size_t)n_utt], 1) * D);
  for (int u = 0; u < n_utt; ++u)
    if (nframes[(size_t)u] > 0) std::memcpy(all.data() + (size_t)frame_off[(size_t)u] * D, feats[(size_t)u], sizeof(float) * (size_t)nframes[(size_t)u] * D);
  CApi(khg_utts_create(ctx, dt.h, n_utt, D, frame_off.data(), all.data(), nullptr, g.state_off.data(), g.start.data(), g.arc_off.data(), g.ilabel.data(),
                       g.olabel.data(), g.weight.data(), g.nextstate.data(), g.final_w.data(), &us.h));
  // only the cells a decoder token can read; with a wide beam (few failed beam certificates to repair) also not the cells that only
  // tokens past any accepting path read (khg_loglikes_band: identical alignments, ~13 % fewer cells on chain graphs)
  if (config.beam >= 100.0f) CApi(khg_loglikes_band(ctx, dm.h, us.h));
  else CApi(khg_loglikes_reachable(ctx, dm.h, us.h));
  khg_align_config c;
  khg_align_config_default(&c);
  c.beam = config.beam; c.retry_beam = config.retry_beam; c.careful = config.careful ? 1 : 0; c.acoustic_scale = acoustic_scale;
  c.like_scale = like_scale;
  if (dopts) { c.max_active = dopts->max_active; c.min_active = dopts->min_active; c.beam_delta = dopts->beam_delta; c.hash_ratio = dopts->hash_ratio; }
  const int64_t N = frame_off[(size_t)n_utt], wcap = N + 16 * (int64_t)n_utt + 1024;
  std::vector<int32_t> ali((size_t)std::max<int64_t>(N, 1)), words((size_t)wcap), status((size_t)n_utt);
  std::vector<int64_t> woff((size_t)n_utt + 1, 0);
  std::vector<float> like((size_t)n_utt);
  CApi(khg_align(ctx, dt.h, us.h, &c, ali.data(), words.data(), woff.data(), wcap, like.data(), status.data()));
  std::vector<float> scores;
  std::vector<int64_t> ll_off((size_t)n_utt + 1, 0), pdf_off((size_t)n_utt + 1, 0);
  std::vector<int32_t> pdfs;
  if (return_scores) {
    int64_t total = 0;
    CApi(khg_loglikes_layout(us.h, ll_off.data(), &total));
    scores.resize((size_t)std::max<int64_t>(total, 1));
    CApi(khg_loglikes_download(ctx, us.h, scores.data()));
    CApi(khg_utts_num_pdfs(us.h, pdf_off.data()));
    pdfs.resize((size_t)std::max<int64_t>(pdf_off[(size_t)n_utt], 1));
    CApi(khg_utts_pdfs(us.h, pdfs.data()));
  }
  std::vector<AlignResult> out((size_t)n_utt);
  for (int u = 0; u < n_utt; ++u) {
    AlignResult& r = out[(size_t)u];
    r.status = status[(size_t)u];
    r.ok = (r.status & KHG_ALIGN_ERROR) == 0;
    r.retried = (r.status & KHG_ALIGN_RETRIED) != 0;
    r.num_frames = (int)nframes[(size_t)u];
    if (r.ok) {
      r.alignment.assign(ali.begin() + frame_off[(size_t)u], ali.begin() + frame_off[(size_t)u + 1]);
      r.words.assign(words.begin() + woff[(size_t)u], words.begin() + woff[(size_t)u + 1]);
      r.like = like[(size_t)u];
    }
    if (return_scores) {
      const int64_t T = nframes[(size_t)u], tpad = (T + 31) & ~int64_t(31);
      const int npdf = (int)(pdf_off[(size_t)u + 1] - pdf_off[(size_t)u]);
      r.pdfs.assign(pdfs.begin() + pdf_off[(size_t)u], pdfs.begin() + pdf_off[(size_t)u + 1]);
      r.loglikes.resize((size_t)npdf * (size_t)T);
      for (int j = 0; j < npdf; ++j)
        if (T > 0) std::memcpy(r.loglikes.data() + (size_t)j * T, scores.data() + ll_off[(size_t)u] + (size_t)j * tpad, sizeof(float) * (size_t)T);
    }
  }
  return out;
}

}  // namespace khg
