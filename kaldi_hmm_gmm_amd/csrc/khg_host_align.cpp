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
    std::vector<int32_t> pdfs((size_t)am_->NumPdfs());
    for (int p = 0; p < am_->NumPdfs(); ++p) pdfs[(size_t)p] = p;
    ll_ = GpuLoglikesOn(am_->DeviceModel(DefaultCtx()), D_, feats_.data(), T_, pdfs.data(), am_->NumPdfs());
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
  // the model and the transition table live on the device across calls: cached on the host objects, uploaded again only when they
  // changed (AmDiagGmm::DeviceModel / TransitionModel::DeviceTm) -- the scripts call this once per utterance
  struct { khg_model* h; } dm{am.DeviceModel(ctx)};
  struct { khg_tm* h; } dt{tm.DeviceTm(ctx)};
  UttsH us;
  CApi(khg_tm_set_trans_cost(dt.h, trans_cost));
  std::vector<int64_t> frame_off((size_t)n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) frame_off[(size_t)u + 1] = frame_off[(size_t)u] + nframes[(size_t)u];
  std::vector<float> all;
  const float* fp = nullptr;
  if (n_utt == 1) fp = feats[0];            // one utterance: its matrix as it is
  else {
    all.resize((size_t)std::max<int64_t>(frame_off[(size_t)n_utt], 1) * D);
    for (int u = 0; u < n_utt; ++u)
      if (nframes[(size_t)u] > 0) std::memcpy(all.data() + (size_t)frame_off[(size_t)u] * D, feats[(size_t)u], sizeof(float) * (size_t)nframes[(size_t)u] * D);
    fp = all.data();
  }
  static const float kNoFrames[1] = {0.0f};
  if (!fp) fp = kNoFrames;
  CApi(khg_utts_create(ctx, dt.h, n_utt, D, frame_off.data(), fp, nullptr, g.state_off.data(), g.start.data(), g.arc_off.data(), g.ilabel.data(),
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

// scripts/gmm_acc_stats_ali.py:46-58 through K3, into the accumulators' device block (khg_host_gmm.hpp)
double AccumAmDiagGmm::AccumulateAli(const AmDiagGmm& model, const TransitionModel& tm, const float* feats, const int64_t* frame_off, int n_utt,
                                     const int32_t* ali, float weight) {
  KHG_REQUIRE(n_utt >= 1 && frame_off && frame_off[0] == 0, "AccumulateAli: bad arguments");
  const int64_t N = frame_off[n_utt];
  if (N == 0) return 0.0;
  KHG_REQUIRE(NumAccs() == model.NumPdfs(), "gmm_accs.NumAccs() == am_gmm.NumPdfs() assertion failed");
  const int nt = tm.NumTransitionIds();
  for (int64_t t = 0; t < N; ++t) KHG_REQUIRE(ali[t] >= 1 && ali[t] <= nt, "gmm_acc_stats_ali: transition-id out of range");
  return AccumulateOnDevice(model, tm.DeviceTm(DefaultCtx()), nt, feats, frame_off, n_utt, ali, weight);
}

// One K3 call into the device-resident block (made for this model version and this many transition-ids); -> the call's own
// sum of weight * log-like.  `dt` == nullptr: the per-frame entry points' table, transition-id = pdf + 1, owned by the block.
double AccumAmDiagGmm::AccumulateOnDevice(const AmDiagGmm& model, khg_tm* dt, int nt, const float* feats, const int64_t* frame_off, int n_utt,
                                          const int32_t* ali, float weight) {
  const int D = model.Dim();
  khg_ctx* ctx = DefaultCtx();
  khg_model* dm = model.DeviceModel(ctx);
  const uint64_t mv = model.Version();
  if (dev_ && (dev_->ctx != ctx || dev_->model_version != mv || dev_->num_tids != nt)) {
    // another model (or the same one after an update that may have moved its layout): what is pending belongs to the old layout
    Flush();
    dev_.reset();
  }
  if (!dev_) {
    auto d = std::make_shared<Dev>();
    d->ctx = ctx; d->model_version = mv; d->num_tids = nt; d->D = D;
    d->gauss_off.assign((size_t)model.NumPdfs() + 1, 0);
    for (int p = 0; p < model.NumPdfs(); ++p) {
      const int G = model.GetPdf(p)->NumGauss();
      KHG_REQUIRE(accs_[(size_t)p]->NumGauss() == G && accs_[(size_t)p]->Dim() == D, "gmm_accs was not initialised for this model (AccumAmDiagGmm.init)");
      d->gauss_off[(size_t)p + 1] = d->gauss_off[(size_t)p] + G;
    }
    if (!dt) {
      std::vector<int32_t> id2pdf((size_t)nt + 1, 0);
      for (int p = 0; p < nt; ++p) id2pdf[(size_t)p + 1] = p;
      CApi(khg_tm_create(ctx, nt, id2pdf.data(), &d->pdf_tm));
    }
    CApi(khg_accs_create(ctx, dm, dt ? dt : d->pdf_tm, &d->h));
    dev_ = d;
  }
  if (!dt) dt = dev_->pdf_tm;
  KHG_REQUIRE(dt != nullptr, "AccumAmDiagGmm: the device block was made for a transition model, not for per-frame calls");
  UttsH us;
  CApi(khg_utts_create(ctx, nullptr, n_utt, D, frame_off, feats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &us.h));
  CApi(khg_ali_upload(ctx, us.h, ali));
  CApi(khg_acc_stats(ctx, dm, dt, us.h, weight, dev_->h));
  dev_->pending = true;
  double sc[8];
  CApi(khg_accs_download_trans(ctx, dev_->h, nullptr, sc));      // the 8 scalars: [frames, log-like, ...] running totals of the block
  const double ll = sc[1] - dev_->seen_ll;
  dev_->seen_frames = sc[0]; dev_->seen_ll = sc[1];
  return ll;
}

// csrc/mle-am-diag-gmm.cc:41-52 (AccumulateForGmm): one frame for one pdf -- the loop body of the reference's own
// scripts/gmm_acc_stats_ali.py:46-56.  The same device-resident block as AccumulateAli (a set of one frame, transition-id = pdf + 1):
// no model upload, no statistics download per frame; the return value is the frame's log-likelihood.
float AccumAmDiagGmm::AccumulateForGmm(const AmDiagGmm& model, const float* data, size_t n, int i, float weight) {
  Chk(i);
  KHG_REQUIRE(NumAccs() == model.NumPdfs(), "gmm_accs.NumAccs() == am_gmm.NumPdfs() assertion failed");
  KHG_REQUIRE((int)n == model.Dim(), "data.size() == Dim() assertion failed");
  if (weight == 0.0f) return model.GetPdf(i)->LogLikelihood(data, n);      // nothing to add; the reference still returns the likelihood
  const int64_t fo[2] = {0, 1};
  const int32_t tid = i + 1;
  const int nt = model.NumPdfs();
  if (dev_ && !dev_->pdf_tm && dev_->num_tids == nt) { Flush(); dev_.reset(); }   // a block AccumulateAli made for a transition model with exactly as many ids has no pdf table
  const double wll = AccumulateOnDevice(model, nullptr, nt, data, fo, 1, &tid, weight);
  return (float)(wll / (double)weight);
}

}  // namespace khg
