// Implementation of khg_host_fst.hpp: see the header for what each piece mirrors in the reference.
#include "khg_host_fst.hpp"

#include <cstdio>

namespace khg {

std::string StdArc::ToString() const {
  char b[96];
  std::snprintf(b, sizeof(b), "StdArc(%d, %d, %g, %d)", ilabel, olabel, (double)weight, nextstate);
  return b;
}

GraphsCsr ConcatGraphs(const std::vector<const StdVectorFst*>& fsts) {
  GraphsCsr c;
  c.state_off.push_back(0);
  c.arc_off.push_back(0);
  for (const StdVectorFst* f : fsts) {
    c.state_off.push_back(c.state_off.back() + f->NumStates());
    c.start.push_back(f->Start());
    for (int s = 0; s < f->NumStates(); ++s) {
      for (const StdArc& a : f->Arcs(s)) {
        c.ilabel.push_back(a.ilabel); c.olabel.push_back(a.olabel); c.weight.push_back(a.weight); c.nextstate.push_back(a.nextstate);
      }
      c.arc_off.push_back((int64_t)c.ilabel.size());
      c.final_w.push_back(f->Final(s));
    }
  }
  return c;
}

void ModifyGraphForCarefulAlignment(StdVectorFst* fst) {
  const int S = fst->NumStates();
  if (S == 0) return;       // "Empty FST input." -- left as it is
  const float inf = std::numeric_limits<float>::infinity();
  std::vector<std::vector<StdArc>> rhs = fst->arcs();           // the right copy, before the left one gains its epsilons
  for (auto& arcs : rhs) for (StdArc& a : arcs) a.nextstate += S;
  const int pre_initial = 2 * S;
  for (int s = 0; s < S; ++s)
    if (fst->finals()[(size_t)s] != inf) {                       // Concat: a final state's weight moves onto an epsilon arc
      fst->arcs()[(size_t)s].push_back(StdArc{0, 0, fst->finals()[(size_t)s], pre_initial});
      fst->finals()[(size_t)s] = inf;
    }
  for (auto& arcs : rhs) { fst->arcs().push_back(std::move(arcs)); fst->finals().push_back(inf); }
  fst->arcs().push_back({StdArc{0, 0, 0.0f, fst->Start() + S}});   // the pre-initial state of the right copy: final, epsilon to its start
  fst->finals().push_back(0.0f);
}

void AddTransitionProbs(const TransitionModel& tm, const std::vector<int>& disambig, float transition_scale, float self_loop_scale, StdVectorFst* fst) {
  for (size_t i = 1; i < disambig.size(); ++i) KHG_REQUIRE(disambig[i - 1] < disambig[i], "IsSortedAndUniq(disambig_syms) assertion failed");
  const std::vector<float> cost = tm.ScaledTransCost(transition_scale, self_loop_scale);
  const int nt = tm.NumTransitionIds();
  for (int s = 0; s < fst->NumStates(); ++s)
    for (StdArc& a : fst->MutableArcs(s)) {
      if (a.ilabel >= 1 && a.ilabel <= nt) a.weight = a.weight + cost[(size_t)a.ilabel];
      else if (a.ilabel != 0 && !std::binary_search(disambig.begin(), disambig.end(), a.ilabel))
        throw Error("AddTransitionProbs: invalid symbol " + std::to_string(a.ilabel) + " on graph input side.");
    }
}

bool LinearLattice::GetLinearSymbolSequence(std::vector<int>* il, std::vector<int>* ol, LatticeWeight* total) const {
  il->clear(); ol->clear();
  *total = LatticeWeight();
  if (start < 0) return false;
  LatticeWeight w = final_w;
  for (const LatticeArc& a : arcs) {
    w.value1 += a.weight.value1; w.value2 += a.weight.value2;
    if (a.ilabel) il->push_back(a.ilabel);
    if (a.olabel) ol->push_back(a.olabel);
  }
  *total = w;
  return true;
}

void FasterDecoder::SetOptions(const FasterDecoderOptions& c) {
  KHG_REQUIRE(c.hash_ratio >= 1.0f && c.max_active > 1 && c.min_active >= 0 && c.min_active < c.max_active,
              "FasterDecoder: bad options (hash_ratio >= 1, max_active > 1, 0 <= min_active < max_active)");
  cfg_ = c;
}
void FasterDecoder::InitDecoding() {
  KHG_REQUIRE(fst_ && fst_->Start() >= 0, "start_state != fst::kNoStateId assertion failed");
  has_res_ = false;
  nframes_ = 0;
}
namespace {
struct TmH { khg_tm* h = nullptr; ~TmH() { if (h) khg_tm_destroy(h); } };
struct UttsH { khg_utts* h = nullptr; ~UttsH() { if (h) khg_utts_destroy(h); } };
}  // namespace

AlignResult AlignDecodable(const StdVectorFst& fst, const DecodableInterface& dec, const AlignConfig& config, float like_scale,
                           const FasterDecoderOptions* dopts) {
  KHG_REQUIRE(!((config.retry_beam != 0 && config.retry_beam <= config.beam) || config.beam <= 0.0f), "Beams do not make sense");
  const int64_t T = dec.NumFramesReady();
  AlignResult r;
  r.num_frames = (int)T;
  const GraphsCsr g = ConcatGraphs({&fst});
  // the indices the decoder can ask for = the non-epsilon input labels of the graph; "pdf" j of the synthetic table is index j + 1
  int max_index = 0;
  for (int32_t l : g.ilabel) {
    KHG_REQUIRE(l >= 0, "AlignDecodable: negative input label on the graph");
    max_index = std::max(max_index, (int)l);
  }
  if (fst.Start() == kNoStateId || max_index == 0 || T <= 0) {       // nothing K2 could decode: same outcome as the GMM path
    r.status = KHG_ALIGN_ERROR;
    return r;
  }
  KHG_REQUIRE(max_index <= dec.NumIndices(), "AlignDecodable: the graph carries index " + std::to_string(max_index) + " but the decodable has " +
                                                 std::to_string(dec.NumIndices()));
  std::vector<int32_t> id2pdf((size_t)max_index + 1);
  id2pdf[0] = -1;
  for (int i = 1; i <= max_index; ++i) id2pdf[(size_t)i] = i - 1;
  khg_ctx* ctx = DefaultCtx();
  TmH dt; UttsH us;
  CApi(khg_tm_create(ctx, max_index, id2pdf.data(), &dt.h));
  const int64_t frame_off[2] = {0, T};
  const std::vector<float> no_feats((size_t)T, 0.0f);                // K2 reads scores, never features
  CApi(khg_utts_create(ctx, dt.h, 1, 1, frame_off, no_feats.data(), nullptr, g.state_off.data(), g.start.data(), g.arc_off.data(), g.ilabel.data(),
                       g.olabel.data(), g.weight.data(), g.nextstate.data(), g.final_w.data(), &us.h));
  int64_t pdf_off[2] = {0, 0}, ll_off[2] = {0, 0}, total = 0;
  CApi(khg_utts_num_pdfs(us.h, pdf_off));
  const int n = (int)pdf_off[1];
  r.pdfs.resize((size_t)n);
  CApi(khg_utts_pdfs(us.h, r.pdfs.data()));
  CApi(khg_loglikes_layout(us.h, ll_off, &total));
  const int64_t tpad = (T + 31) & ~int64_t(31);
  KHG_REQUIRE(total >= (int64_t)n * tpad, "AlignDecodable: unexpected score layout");
  std::vector<float> scores((size_t)total, 0.0f);
  r.loglikes.resize((size_t)n * (size_t)T);
  for (int j = 0; j < n; ++j)
    for (int64_t t = 0; t < T; ++t) {
      const float s = dec.LogLikelihood((int)t, r.pdfs[(size_t)j] + 1);
      scores[(size_t)(ll_off[0] + (int64_t)j * tpad + t)] = s;
      r.loglikes[(size_t)j * (size_t)T + (size_t)t] = s;
    }
  CApi(khg_loglikes_upload(ctx, us.h, scores.data()));
  khg_align_config c;
  khg_align_config_default(&c);
  c.beam = config.beam; c.retry_beam = config.retry_beam; c.careful = config.careful ? 1 : 0;
  c.acoustic_scale = 1.0f;                      // 1.0f * s == s: the decodable scaled its scores itself
  c.like_scale = like_scale;
  if (dopts) { c.max_active = dopts->max_active; c.min_active = dopts->min_active; c.beam_delta = dopts->beam_delta; c.hash_ratio = dopts->hash_ratio; }
  const int64_t wcap = T + 1040;
  std::vector<int32_t> ali((size_t)T), words((size_t)wcap);
  int64_t woff[2] = {0, 0};
  float like = 0.0f;
  int32_t status = 0;
  CApi(khg_align(ctx, dt.h, us.h, &c, ali.data(), words.data(), woff, wcap, &like, &status));
  r.status = status;
  r.ok = (status & KHG_ALIGN_ERROR) == 0;
  r.retried = (status & KHG_ALIGN_RETRIED) != 0;
  if (r.ok) {
    r.alignment = std::move(ali);
    r.words.assign(words.begin() + woff[0], words.begin() + woff[1]);
    r.like = like;
  }
  return r;
}

void FasterDecoder::AdvanceDecoding(const std::shared_ptr<DecodableInterface>& dec, int max_num_frames) {
  KHG_REQUIRE(dec != nullptr, "FasterDecoder: no decodable");
  KHG_REQUIRE(!(max_num_frames >= 0 && max_num_frames < dec->NumFramesReady()),
              "FasterDecoder.advanced_decoding: partial decoding (max_num_frames) is not supported on the HIP path");
  KHG_REQUIRE(nframes_ >= 0, "num_frames_decoded_ >= 0 assertion failed: call init_decoding() first");
  AlignConfig cfg;
  cfg.beam = cfg_.beam; cfg.retry_beam = 0.0f;
  dec_ = dec;
  ac_.clear();
  if (auto gmm = std::dynamic_pointer_cast<DecodableAmDiagGmmScaled>(dec)) {       // K1 + K2
    const GraphsCsr g = ConcatGraphs({fst_.get()});
    res_ = AlignBatch(*gmm->am(), *gmm->tm(), g, {gmm->feats().data()}, {(int64_t)gmm->NumFramesReady()}, cfg, gmm->scale(), nullptr, &cfg_, true)[0];
    if (res_.ok) {
      const TransitionModel& tm = *gmm->tm();
      const size_t T = res_.alignment.size();
      std::vector<int> col((size_t)tm.NumPdfs(), -1);
      for (size_t i = 0; i < res_.pdfs.size(); ++i) col[(size_t)res_.pdfs[i]] = (int)i;
      for (size_t i = 0; i < T; ++i) {
        const int c = col[(size_t)tm.TransitionIdToPdf(res_.alignment[i])];
        KHG_REQUIRE(c >= 0, "FasterDecoder: alignment uses a pdf outside the utterance's list");
        ac_.push_back((double)(-(gmm->scale() * res_.loglikes[(size_t)c * T + i])));
      }
    }
  } else {                                                                          // sampled scores + K2
    res_ = AlignDecodable(*fst_, *dec, cfg, 0.0f, &cfg_);
    if (res_.ok) {
      const size_t T = res_.alignment.size();
      for (size_t i = 0; i < T; ++i) {
        const auto it = std::lower_bound(res_.pdfs.begin(), res_.pdfs.end(), res_.alignment[i] - 1);
        KHG_REQUIRE(it != res_.pdfs.end() && *it == res_.alignment[i] - 1, "FasterDecoder: alignment uses an index outside the graph's");
        ac_.push_back((double)(-res_.loglikes[(size_t)(it - res_.pdfs.begin()) * T + i]));
      }
    }
  }
  has_res_ = true;
  nframes_ = dec->NumFramesReady();
}

bool FasterDecoder::GetBestPath(LinearLattice* lat, bool use_final_probs) const {
  *lat = LinearLattice();
  if (!ReachedFinal()) return false;     // the reference would fall back to the best non-final token; the HIP kernels keep no such token
  const std::vector<int32_t>& ali = res_.alignment;
  const int T = (int)ali.size(), S = fst_->NumStates();
  const double INF = std::numeric_limits<double>::infinity();
  const std::vector<double>& ac = ac_;
  // cheapest path through the graph with exactly this input-label sequence (= the decoder's best path); layers keep their states
  // in insertion order, a later candidate replaces an earlier one only when strictly cheaper
  struct Back { int prev = -1, arc = -1; };       // arc = index into Arcs(prev)
  struct Layer {
    std::vector<double> cost; std::vector<int> keys;
    explicit Layer(int S) : cost((size_t)S, std::numeric_limits<double>::infinity()) {}
    void Set(int s, double v) { if (cost[(size_t)s] == std::numeric_limits<double>::infinity()) keys.push_back(s); cost[(size_t)s] = v; }
  };
  std::vector<std::vector<Back>> bp_emit((size_t)T + 1, std::vector<Back>()), bp_eps((size_t)T + 1, std::vector<Back>());
  auto closure = [&](Layer& layer, std::vector<Back>& bp) {
    bp.assign((size_t)S, Back());
    std::vector<int> stack = layer.keys;
    while (!stack.empty()) {
      const int s = stack.back(); stack.pop_back();
      const double c = layer.cost[(size_t)s];
      const auto& arcs = fst_->Arcs(s);
      for (size_t k = 0; k < arcs.size(); ++k)
        if (arcs[k].ilabel == 0) {
          const double v = c + (double)arcs[k].weight;
          if (v < layer.cost[(size_t)arcs[k].nextstate]) {
            layer.Set(arcs[k].nextstate, v);
            bp[(size_t)arcs[k].nextstate] = Back{s, (int)k};
            stack.push_back(arcs[k].nextstate);
          }
        }
    }
  };
  Layer layer(S);
  layer.Set(fst_->Start(), 0.0);
  closure(layer, bp_eps[0]);
  for (int i = 0; i < T; ++i) {
    Layer nxt(S);
    bp_emit[(size_t)i + 1].assign((size_t)S, Back());
    for (int s : layer.keys) {
      const double c = layer.cost[(size_t)s];
      const auto& arcs = fst_->Arcs(s);
      for (size_t k = 0; k < arcs.size(); ++k)
        if (arcs[k].ilabel == ali[(size_t)i]) {
          const double v = c + (double)arcs[k].weight + ac[(size_t)i];
          if (v < nxt.cost[(size_t)arcs[k].nextstate]) {
            nxt.Set(arcs[k].nextstate, v);
            bp_emit[(size_t)i + 1][(size_t)arcs[k].nextstate] = Back{s, (int)k};
          }
        }
    }
    closure(nxt, bp_eps[(size_t)i + 1]);
    layer = std::move(nxt);
  }
  double best = INF;
  int bs = -1;
  for (int s : layer.keys)
    if (fst_->IsFinal(s) && layer.cost[(size_t)s] + (double)fst_->Final(s) < best) { best = layer.cost[(size_t)s] + (double)fst_->Final(s); bs = s; }
  if (bs < 0) return false;
  struct Step { int state, arc; bool emitting; double acost; };
  std::vector<Step> path;
  int s = bs;
  for (int i = T; i >= 0; --i) {
    while (bp_eps[(size_t)i][(size_t)s].prev >= 0) {            // epsilon hops inside layer i
      const Back b = bp_eps[(size_t)i][(size_t)s];
      path.push_back(Step{b.prev, b.arc, false, 0.0});
      s = b.prev;
    }
    if (i > 0) {
      const Back b = bp_emit[(size_t)i][(size_t)s];
      KHG_REQUIRE(b.prev >= 0, "FasterDecoder: broken back-pointer chain");
      path.push_back(Step{b.prev, b.arc, true, ac[(size_t)i - 1]});
      s = b.prev;
    }
  }
  std::reverse(path.begin(), path.end());
  lat->start = 0;
  LatticeWeight carry;
  for (const Step& st : path) {
    const StdArc& a = fst_->Arcs(st.state)[(size_t)st.arc];
    const LatticeWeight w{(double)a.weight + carry.value1, (st.emitting ? st.acost : 0.0) + carry.value2};
    if (a.ilabel == 0 && a.olabel == 0) { carry = w; continue; }      // RemoveEpsLocal on a linear lattice: fold true epsilons forward
    carry = LatticeWeight();
    lat->arcs.push_back(LatticeArc{a.ilabel, a.olabel, w, (int)lat->arcs.size() + 1});
  }
  const double fw = use_final_probs ? (double)fst_->Final(bs) : 0.0;
  lat->final_w = LatticeWeight{fw + carry.value1, carry.value2};
  return true;
}

}  // namespace khg
