// Host-side C++ classes behind the reference's names: HmmState / HmmTopology, TransitionModelTuple / TransitionModel,
// MleTransitionUpdateConfig, GetPdfsForPhones.
//
// Mirrors (reference, /root/reference/kaldi-hmm-gmm/): csrc/hmm-topology.{h,cc}, csrc/transition-model.{h,cc},
// csrc/transition-information.h; pybind surface python/csrc/{hmm-topology,transition-model,transition-information}.cc.
// The integer tables are built here; the probability update is khg_transition_mle_update (khg_host.cpp) or, for
// share_for_pdfs, MleUpdateShared below in the reference's own arithmetic (csrc/transition-model.cc:531-655).
#pragma once
#include <map>
#include <set>
#include <utility>

#include "khg_host_gmm.hpp"

namespace khg {

constexpr int kNoPdf = -1;

struct HmmState {          // csrc/hmm-topology.h:71-102
  int forward_pdf_class = kNoPdf, self_loop_pdf_class = kNoPdf;
  std::vector<std::pair<int, float>> transitions;      // (destination state, probability)
  HmmState() = default;
  explicit HmmState(int fwd) : forward_pdf_class(fwd), self_loop_pdf_class(fwd) {}
  HmmState(int fwd, int sl) : forward_pdf_class(fwd), self_loop_pdf_class(sl) {}
  bool operator==(const HmmState& o) const {
    return forward_pdf_class == o.forward_pdf_class && self_loop_pdf_class == o.self_loop_pdf_class && transitions == o.transitions;
  }
  std::string ToString() const;                        // python/csrc/hmm-topology.cc:22-36
};

std::string FormatG(double x);                         // C++ ostream << float: %g

class HmmTopology {
 public:
  using Entry = std::vector<HmmState>;
  void Read(const std::string& text);                  // csrc/hmm-topology.cc:23-160 (text mode), then Check()
  std::string ToString() const;                        // :162-218 (text mode)
  const std::vector<int>& phones() const { return phones_; }
  const std::vector<int>& phone2idx() const { return phone2idx_; }
  const std::vector<Entry>& entries() const { return entries_; }
  void SetState(std::vector<int> phones, std::vector<int> phone2idx, std::vector<Entry> entries) {   // unpickling / binary read
    phones_ = std::move(phones); phone2idx_ = std::move(phone2idx); entries_ = std::move(entries);
  }
  bool IsHmm() const;                                  // :284-301
  const Entry& TopologyForPhone(int phone) const;      // :303-310
  int NumPdfClasses(int phone) const;                  // :429-440
  std::vector<int> GetPhoneToNumPdfClasses() const;    // :442-451
  int MinLength(int phone) const;                      // :453-492
  void Check() const;                                  // :312-427

 private:
  std::vector<int> phones_, phone2idx_;
  std::vector<Entry> entries_;
};

struct TransitionModelTuple {     // csrc/transition-model.h:103-126
  int phone = 0, hmm_state = 0, forward_pdf = 0, self_loop_pdf = 0;
  bool operator==(const TransitionModelTuple& o) const {
    return phone == o.phone && hmm_state == o.hmm_state && forward_pdf == o.forward_pdf && self_loop_pdf == o.self_loop_pdf;
  }
  bool operator<(const TransitionModelTuple& o) const {
    if (phone != o.phone) return phone < o.phone;
    if (hmm_state != o.hmm_state) return hmm_state < o.hmm_state;
    if (forward_pdf != o.forward_pdf) return forward_pdf < o.forward_pdf;
    return self_loop_pdf < o.self_loop_pdf;
  }
  std::string ToString() const;
};

struct MleTransitionUpdateConfig {    // csrc/transition-model.h:80-92
  float floor = 0.01f, mincount = 5.0f;
  bool share_for_pdfs = false;
};

class TransitionInformation {      // csrc/transition-information.h:26-77
 public:
  virtual ~TransitionInformation() = default;
  virtual bool TransitionIdsEquivalent(int trans_id1, int trans_id2) const = 0;
  virtual bool TransitionIdIsStartOfPhone(int trans_id) const = 0;
  virtual int TransitionIdToPhone(int trans_id) const = 0;
  virtual bool IsFinal(int trans_id) const = 0;
  virtual bool IsSelfLoop(int trans_id) const = 0;
  virtual int TransitionIdToPdf(int trans_id) const = 0;
  virtual const std::vector<int>& TransitionIdToPdfArray() const = 0;
  virtual int NumTransitionIds() const = 0;
  virtual int NumPdfs() const = 0;
};

class TransitionModel : public TransitionInformation {
 public:
  TransitionModel() = default;
  // pdf_info[pdf] = [(phone, pdf_class)] from ContextDependency::GetPdfInfo (csrc/context-dep.cc); csrc/transition-model.cc:120-252,
  // then ComputeDerived (:254-303), InitializeProbs (:318-337), Check (:396-419)
  TransitionModel(const std::vector<std::vector<std::pair<int, int>>>& pdf_info, std::shared_ptr<HmmTopology> topo);
  // what Read (csrc/transition-model.cc:85-116) rebuilds from a file: tuples + log-probs, the rest derived
  void SetFromRead(std::shared_ptr<HmmTopology> topo, std::vector<TransitionModelTuple> tuples, std::vector<float> log_probs);
  // unpickling: the 8 members as they are
  void SetState(std::vector<TransitionModelTuple> tuples, std::shared_ptr<HmmTopology> topo, std::vector<int> state2id, std::vector<int> id2state,
                std::vector<int> id2pdf, int num_pdfs, std::vector<float> log_probs, std::vector<float> nsl);
  void Check() const;
  int NumTransitionIds() const override { return (int)id2state_.size() - 1; }
  int NumTransitionStates() const { return (int)tuples_.size(); }
  int NumPdfs() const override { return num_pdfs_; }
  const std::vector<int>& TransitionIdToPdfArray() const override { return id2pdf_; }
  const std::shared_ptr<HmmTopology>& topo() const { return topo_; }
  const std::vector<TransitionModelTuple>& tuples() const { return tuples_; }
  const std::vector<int>& state2id() const { return state2id_; }
  const std::vector<int>& id2state() const { return id2state_; }
  const std::vector<int>& id2pdf() const { return id2pdf_; }
  const std::vector<float>& log_probs() const { return log_probs_; }
  const std::vector<float>& non_self_loop_log_probs() const { return nsl_; }
  void ChkTid(int tid) const { KHG_REQUIRE(tid > 0 && tid <= NumTransitionIds(), "transition-id " + std::to_string(tid) + " out of range"); }
  int TransitionIdToPdf(int tid) const override { ChkTid(tid); return id2pdf_[(size_t)tid]; }
  int TransitionIdToPhone(int tid) const override { ChkTid(tid); return tuples_[(size_t)id2state_[(size_t)tid] - 1].phone; }
  int TransitionIdToHmmState(int tid) const { ChkTid(tid); return tuples_[(size_t)id2state_[(size_t)tid] - 1].hmm_state; }
  bool TransitionIdsEquivalent(int a, int b) const override { ChkTid(a); ChkTid(b); return id2state_[(size_t)a] == id2state_[(size_t)b]; }
  bool TransitionIdIsStartOfPhone(int tid) const override { return TransitionIdToHmmState(tid) == 0; }
  bool IsSelfLoop(int tid) const override { ChkTid(tid); return IsSelfLoopRaw(tid); }
  bool IsSelfLoopRaw(int tid) const;
  bool IsFinal(int tid) const override;
  int SelfLoopOf(int trans_state) const;
  float GetTransitionLogProb(int tid) const { KHG_REQUIRE(tid >= 0 && tid <= NumTransitionIds(), "transition-id out of range"); return log_probs_[(size_t)tid]; }
  int TupleToTransitionState(int phone, int hmm_state, int pdf, int self_loop_pdf) const;      // :432-447
  int PairToTransitionId(int trans_state, int trans_index) const;                              // :385-390
  int TransitionIdToTransitionState(int tid) const { ChkTid(tid); return id2state_[(size_t)tid]; }
  float GetNonSelfLoopLogProb(int trans_state) const;                                          // :515-518
  float GetTransitionLogProbIgnoringSelfLoops(int tid) const;                                  // :520-526
  // statistics -> (objf_impr, count); csrc/transition-model.cc:657-750 / :531-655
  std::pair<float, float> MleUpdate(const double* stats, size_t n, const MleTransitionUpdateConfig& cfg);
  std::vector<uint8_t> IsSelfLoopArray() const;
  std::vector<float> ScaledTransCost(float transition_scale, float self_loop_scale) const;     // csrc/hmm-utils.cc:442-463, negated
  std::string ToString() const;                                                                // csrc/transition-model.cc:37-83 text Write
  // TransitionIdToPdf on the device (khg_tm), made when first asked for and again only when the table differs: the scripts pass the
  // same TransitionModel to every per-utterance call
  khg_tm* DeviceTm(khg_ctx* ctx) const;

 private:
  struct Dev {
    std::mutex mu; khg_ctx* ctx = nullptr; khg_tm* h = nullptr; std::vector<int> id2pdf;
    ~Dev() { if (h) khg_tm_destroy(h); }
  };
  mutable std::shared_ptr<Dev> dev_;
  void ComputeTuplesIsHmm(const std::vector<std::vector<std::pair<int, int>>>& pdf_info);
  void ComputeDerived();
  void InitializeProbs();
  void ComputeDerivedOfProbs();                        // :339-359
  std::pair<float, float> MleUpdateShared(const double* stats, const MleTransitionUpdateConfig& cfg);
  std::vector<TransitionModelTuple> tuples_;
  std::shared_ptr<HmmTopology> topo_;
  std::vector<int> state2id_, id2state_, id2pdf_;
  int num_pdfs_ = 0;
  std::vector<float> log_probs_, nsl_;
};

// csrc/transition-model.cc:752-785 -> is_unique; pdfs sorted
bool GetPdfsForPhones(const TransitionModel& tm, const std::vector<int>& phones, std::vector<int>* pdfs);

}  // namespace khg
