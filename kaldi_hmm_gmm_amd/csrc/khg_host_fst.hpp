// The decoding-graph container of the host API and what the alignment entry points do with it, in C++.
//
// The reference takes fst::VectorFst<fst::StdArc> from the separate kaldifst package (scripts/gmm_align_compiled.py:6,14), which
// is not available offline; this is the minimal tropical-semiring VectorFst with the kaldifst method names the reference's scripts
// and tests use.  On top of it (reference, /root/reference/kaldi-hmm-gmm/csrc/): AddTransitionProbs (hmm-utils.cc:465-493),
// ModifyGraphForCarefulAlignment (decoder-wrappers.cc:111-140), and FasterDecoder's host side (faster-decoder.cc:33-53 decode /
// :346-423 ReachedFinal / GetBestPath; python/csrc/faster-decoder.cc:14-53) over K1 + K2.
#pragma once
#include "khg_host_align.hpp"

namespace khg {

constexpr int kNoStateId = -1;

struct StdArc {
  int ilabel = 0, olabel = 0;
  float weight = 0.0f;
  int nextstate = 0;
  std::string ToString() const;
};

class StdVectorFst {
 public:
  int AddState() { arcs_.emplace_back(); final_.push_back(std::numeric_limits<float>::infinity()); return (int)arcs_.size() - 1; }
  int NumStates() const { return (int)arcs_.size(); }
  int Start() const { return start_; }
  void SetStart(int s) { start_ = s; }
  void AddArc(int state, const StdArc& a) { KHG_REQUIRE(state >= 0 && state < NumStates(), "add_arc: bad state"); arcs_[(size_t)state].push_back(a); }
  void SetFinal(int state, float w) { KHG_REQUIRE(state >= 0 && state < NumStates(), "set_final: bad state"); final_[(size_t)state] = w; }
  float Final(int state) const { KHG_REQUIRE(state >= 0 && state < NumStates(), "final: bad state"); return final_[(size_t)state]; }
  bool IsFinal(int state) const { return Final(state) != std::numeric_limits<float>::infinity(); }     // +inf == TropicalWeight::Zero()
  const std::vector<StdArc>& Arcs(int state) const { KHG_REQUIRE(state >= 0 && state < NumStates(), "arcs: bad state"); return arcs_[(size_t)state]; }
  std::vector<StdArc>& MutableArcs(int state) { KHG_REQUIRE(state >= 0 && state < NumStates(), "arcs: bad state"); return arcs_[(size_t)state]; }
  int64_t NumArcs() const { int64_t n = 0; for (auto& a : arcs_) n += (int64_t)a.size(); return n; }
  std::vector<std::vector<StdArc>>& arcs() { return arcs_; }
  const std::vector<std::vector<StdArc>>& arcs() const { return arcs_; }
  std::vector<float>& finals() { return final_; }
  const std::vector<float>& finals() const { return final_; }

 private:
  std::vector<std::vector<StdArc>> arcs_;
  std::vector<float> final_;
  int start_ = kNoStateId;
};

// the CSR block khg_utts_create consumes, for a list of graphs
GraphsCsr ConcatGraphs(const std::vector<const StdVectorFst*>& fsts);
// csrc/decoder-wrappers.cc:111-140 (+ OpenFst Concat): in place
void ModifyGraphForCarefulAlignment(StdVectorFst* fst);
// csrc/hmm-utils.cc:465-493: arc.weight (x)= -scaled transition log-prob, in place
void AddTransitionProbs(const TransitionModel& tm, const std::vector<int>& disambig_syms, float transition_scale, float self_loop_scale, StdVectorFst* fst);

// AlignUtteranceWrapper / FasterDecoder::Decode for ANY DecodableInterface (csrc/decoder-wrappers.cc:16-108 takes a
// DecodableInterface*, python/csrc/decodable-itf.cc:16-53 lets Python subclass it): the scores of every (frame, transition-id on
// the graph) are sampled through the interface into K2's score matrix (khg_loglikes_upload) and K2 decodes them unscaled -- the
// decodable already applied its own scale; `like` is divided by like_scale.  r.pdfs lists index - 1 for the sampled indices,
// r.loglikes their [n][T] scores.
AlignResult AlignDecodable(const StdVectorFst& fst, const DecodableInterface& decodable, const AlignConfig& config, float like_scale,
                           const FasterDecoderOptions* decoder_opts);

struct LatticeWeight {        // kaldifst LatticeWeight (graph cost, acoustic cost); Times adds component-wise
  double value1 = 0.0, value2 = 0.0;
};
struct LatticeArc {
  int ilabel = 0, olabel = 0;
  LatticeWeight weight;
  int nextstate = 0;
};
// The linear fst::VectorFst<LatticeArc> FasterDecoder::GetBestPath returns: state i has the single arc arcs[i] to state i + 1; the
// last state is final with `final`.
struct LinearLattice {
  std::vector<LatticeArc> arcs;
  LatticeWeight final_w;
  int start = -1;
  int NumStates() const { return start < 0 ? 0 : (int)arcs.size() + 1; }
  // kaldifst GetLinearSymbolSequence -> ok; ilabels != 0, olabels != 0, total weight
  bool GetLinearSymbolSequence(std::vector<int>* ilabels, std::vector<int>* olabels, LatticeWeight* total) const;
};

// python/csrc/faster-decoder.cc:33-53 on the GPU path: Decode runs K1 + K2 for the utterance of a DecodableAmDiagGmmScaled (any other
// DecodableInterface: its sampled scores + K2, AlignDecodable above) with the options' beam / max_active / min_active / beam_delta /
// hash_ratio (no retry); GetBestPath rebuilds the linear lattice of
// csrc/faster-decoder.cc:355-423 from the alignment: arc weights (graph cost, acoustic cost) per token, final weight, true epsilons
// removed.  Whole utterances only: AdvanceDecoding with a frame limit is not supported.
class FasterDecoder {
 public:
  FasterDecoder(std::shared_ptr<StdVectorFst> fst, const FasterDecoderOptions& config) : fst_(std::move(fst)) { SetOptions(config); }
  void SetOptions(const FasterDecoderOptions& config);
  void InitDecoding();
  void Decode(const std::shared_ptr<DecodableInterface>& decodable) { InitDecoding(); AdvanceDecoding(decodable, -1); }
  void AdvanceDecoding(const std::shared_ptr<DecodableInterface>& decodable, int max_num_frames);
  int NumFramesDecoded() const { return nframes_; }
  bool ReachedFinal() const { return has_res_ && res_.ok; }
  bool GetBestPath(LinearLattice* lat, bool use_final_probs) const;

 private:
  std::shared_ptr<StdVectorFst> fst_;
  FasterDecoderOptions cfg_;
  std::shared_ptr<DecodableInterface> dec_;
  AlignResult res_;
  std::vector<double> ac_;           // acoustic cost of each aligned frame: -(score the decoder read for (frame, alignment[frame]))
  bool has_res_ = false;
  int nframes_ = -1;
};

}  // namespace khg
