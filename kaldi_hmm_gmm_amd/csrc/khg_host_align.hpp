// Host-side C++ of the alignment API: AlignConfig, FasterDecoderOptions, DecodableAmDiagGmmUnmapped / Scaled and the batched
// AlignUtteranceWrapper over resident graphs.
//
// Mirrors (reference, /root/reference/kaldi-hmm-gmm/): csrc/decoder-wrappers.{h,cc} (AlignConfig :23-37, AlignUtteranceWrapper
// :16-108), csrc/faster-decoder.h:24-63 (FasterDecoderOptions), csrc/decodable-am-diag-gmm.h:30-103; pybind surface
// python/csrc/{decoder-wrappers,faster-decoder,decodable-am-diag-gmm}.cc.  The work is K1 (log-likes) + K2 (Viterbi) through the
// C-ABI, for one utterance or a whole batch; graphs come as the CSR block of khg_utts_create (the reference holds them in
// kaldifst's fst::VectorFst<StdArc>, a third-party container that stays on the Python side here).
#pragma once
#include <limits>

#include "khg_host_gmm.hpp"
#include "khg_host_hmm.hpp"

namespace khg {

struct AlignConfig {                 // csrc/decoder-wrappers.h:23-37
  float beam = 200.0f, retry_beam = 0.0f;
  bool careful = false;
};

struct FasterDecoderOptions {        // csrc/faster-decoder.h:24-63
  float beam = 16.0f;
  int32_t max_active = std::numeric_limits<int32_t>::max(), min_active = 20;
  float beam_delta = 0.5f, hash_ratio = 2.0f;
  std::string ToString() const;
};

struct GraphsCsr {                   // fst::VectorFst<StdArc> per utterance, concatenated as khg_utts_create takes them
  std::vector<int64_t> state_off, arc_off;
  std::vector<int32_t> start, ilabel, olabel, nextstate;
  std::vector<float> weight, final_w;
};

// csrc/decodable-itf.h: what a decoder asks of an acoustic model (1-based index, 0-based frame).  The HIP kernels read scores from
// K1's matrices, so the classes below are what the alignment entry points accept; the interface itself is there for callers
// that score frames from Python or C++ through the reference's protocol.
class DecodableInterface {
 public:
  virtual ~DecodableInterface() = default;
  virtual float LogLikelihood(int frame, int index) const = 0;
  virtual bool IsLastFrame(int frame) const = 0;
  virtual int NumFramesReady() const { throw Error("NumFramesReady() not implemented for this decodable type."); }
  virtual int NumIndices() const = 0;
};

// csrc/decodable-am-diag-gmm.h:30-78: (frame, pdf-id + 1) -> log-likelihood.  Scores for every pdf are produced by one K1 launch
// on first use and kept (the reference's one-frame cache).
// NOTE (DecodableInterface above): LogLikelihood is declared const here; the reference's is non-const because of that cache.
class DecodableAmDiagGmmUnmapped : public DecodableInterface {
 public:
  DecodableAmDiagGmmUnmapped(std::shared_ptr<AmDiagGmm> am, const float* feats, int64_t T, int D);
  float LogLikelihood(int frame, int index) const override { return ZeroBased(frame, index - 1); }
  float ZeroBased(int frame, int state) const;
  int NumFramesReady() const override { return (int)T_; }
  int NumIndices() const override { return am_->NumPdfs(); }
  bool IsLastFrame(int frame) const override;
  const std::shared_ptr<AmDiagGmm>& am() const { return am_; }
  const std::vector<float>& feats() const { return feats_; }
  int Dim() const { return D_; }

 protected:
  const std::vector<float>& Scores() const;
  std::shared_ptr<AmDiagGmm> am_;
  std::vector<float> feats_;
  int64_t T_;
  int D_;
  mutable std::vector<float> ll_;    // [num_pdfs][T], filled on first use
};

// csrc/decodable-am-diag-gmm.h:83-103: scale * LL(frame, TransitionIdToPdf(tid))
class DecodableAmDiagGmmScaled : public DecodableAmDiagGmmUnmapped {
 public:
  DecodableAmDiagGmmScaled(std::shared_ptr<AmDiagGmm> am, std::shared_ptr<TransitionModel> tm, const float* feats, int64_t T, int D, float scale)
      : DecodableAmDiagGmmUnmapped(std::move(am), feats, T, D), tm_(std::move(tm)), scale_(scale) {}
  float LogLikelihood(int frame, int tid) const override { return scale_ * ZeroBased(frame, tm_->TransitionIdToPdf(tid)); }
  int NumIndices() const override { return tm_->NumTransitionIds(); }
  const std::shared_ptr<TransitionModel>& tm() const { return tm_; }
  float scale() const { return scale_; }

 private:
  std::shared_ptr<TransitionModel> tm_;
  float scale_;
};

struct AlignResult {
  bool ok = false, retried = false;
  int status = 0;
  std::vector<int32_t> alignment, words;
  float like = 0.0f;
  int num_frames = 0;
  std::vector<float> loglikes;       // [npdf][T] of the utterance's own pdf list (return_scores)
  std::vector<int32_t> pdfs;
};

// Batched AlignUtteranceWrapper: all utterances in one K1 + K2 pass.  The graphs carry their final arc weights unless trans_cost
// (per-tid additive cost, TransitionModel::ScaledTransCost) is given, in which case it is added on the device.  config.careful
// only tells K2 the graphs were doubled (the caller applied ModifyGraphForCarefulAlignment / khg_careful_graph).
std::vector<AlignResult> AlignBatch(const AmDiagGmm& am, const TransitionModel& tm, const GraphsCsr& graphs, const std::vector<const float*>& feats,
                                    const std::vector<int64_t>& nframes, const AlignConfig& config, float acoustic_scale, const float* trans_cost,
                                    const FasterDecoderOptions* decoder_opts, bool return_scores, float like_scale = 0.0f);
// (like_scale: divisor of `like` when it is not the score scale -- a decodable whose own scale differs from the wrapper's
// acoustic_scale argument, csrc/decoder-wrappers.cc:95; 0 = acoustic_scale)

}  // namespace khg
