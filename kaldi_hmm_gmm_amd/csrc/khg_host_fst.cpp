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
void FasterDecoder::AdvanceDecoding(const std::shared_ptr<DecodableAmDiagGmmScaled>& dec, int max_num_frames) {
  KHG_REQUIRE(dec != nullptr, "FasterDecoder: the HIP path needs a DecodableAmDiagGmmScaled");
  KHG_REQUIRE(!(max_num_frames >= 0 && max_num_frames < dec->NumFramesReady()),
              "FasterDecoder.advanced_decoding: partial decoding (max_num_frames) is not supported on the HIP path");
  KHG_REQUIRE(nframes_ >= 0, "num_frames_decoded_ >= 0 assertion failed: call init_decoding() first");
  AlignConfig cfg;
  cfg.beam = cfg_.beam; cfg.retry_beam = 0.0f;
  dec_ = dec;
  const GraphsCsr g = ConcatGraphs({fst_.get()});
  res_ = AlignBatch(*dec->am(), *dec->tm(), g, {dec->feats().data()}, {(int64_t)dec->NumFramesReady()}, cfg, dec->scale(), nullptr, &cfg_, true)[0];
  has_res_ = true;
  nframes_ = dec->NumFramesReady();
}

bool FasterDecoder::GetBestPath(LinearLattice* lat, bool use_final_probs) const {
  *lat = LinearLattice();
  if (!ReachedFinal()) return false;     // the reference would fall back to the best non-final token; the HIP kernels keep no such token
  const std::vector<int32_t>& ali = res_.alignment;
  const TransitionModel& tm = *dec_->tm();
  const int T = (int)ali.size(), S = fst_->NumStates();
  const double INF = std::numeric_limits<double>::infinity();
  std::vector<int> col((size_t)tm.NumPdfs(), -1);
  for (size_t i = 0; i < res_.pdfs.size(); ++i) col[(size_t)res_.pdfs[i]] = (int)i;
  std::vector<double> ac((size_t)T);
  for (int i = 0; i < T; ++i) {
    const int c = col[(size_t)tm.TransitionIdToPdf(ali[(size_t)i])];
    KHG_REQUIRE(c >= 0, "FasterDecoder: alignment uses a pdf outside the utterance's list");
    ac[(size_t)i] = (double)(-(dec_->scale() * res_.loglikes[(size_t)c * T + i]));
  }
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
