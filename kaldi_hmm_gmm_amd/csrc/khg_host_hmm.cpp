// Implementation of khg_host_hmm.hpp: see the header for what each class mirrors in the reference.
#include "khg_host_hmm.hpp"

#include <cstdio>
#include <sstream>

namespace khg {

std::string FormatG(double x) {
  char buf[64];
  std::snprintf(buf, sizeof(buf), "%g", x);
  return buf;
}

std::string HmmState::ToString() const {
  std::string s = "HmmState(forward_pdf_class=" + std::to_string(forward_pdf_class) + ", self_loop_pdf_class=" + std::to_string(self_loop_pdf_class) +
                  ", transitions=[";
  for (size_t i = 0; i < transitions.size(); ++i) {
    if (i) s += ", ";
    s += "(" + std::to_string(transitions[i].first) + ", " + FormatG(transitions[i].second) + ")";
  }
  return s + "])";
}

// ---- HmmTopology ------------------------------------------------------------------------------------------------------
namespace {
struct Tokens {
  std::vector<std::string> tok;
  size_t pos = 0;
  explicit Tokens(const std::string& s) {
    std::istringstream is(s);
    std::string t;
    while (is >> t) tok.push_back(t);
  }
  bool done() const { return pos >= tok.size(); }
  const std::string& next() {
    KHG_REQUIRE(pos < tok.size(), "Reading HmmTopology object, unexpected end of input");
    return tok[pos++];
  }
  void expect(const char* t) {
    const std::string& g = next();
    KHG_REQUIRE(g == t, std::string("Expected token ") + t + ", got " + g);
  }
  int next_int() {
    const std::string& t = next();
    size_t used = 0;
    int v = 0;
    try { v = std::stoi(t, &used); } catch (...) { used = 0; }
    KHG_REQUIRE(used == t.size() && !t.empty(), "Reading HmmTopology object, expected integer, got instead " + t);
    return v;
  }
  float next_float() {
    const std::string& t = next();
    size_t used = 0;
    double v = 0;
    try { v = std::stod(t, &used); } catch (...) { used = 0; }
    KHG_REQUIRE(used == t.size() && !t.empty(), "Reading HmmTopology object, expected a number, got instead " + t);
    return (float)v;
  }
};
}  // namespace

void HmmTopology::Read(const std::string& text) {
  Tokens tk(text);
  tk.expect("<Topology>");
  phones_.clear(); phone2idx_.clear(); entries_.clear();
  while (!tk.done()) {
    std::string t = tk.next();
    if (t == "</Topology>") break;
    KHG_REQUIRE(t == "<TopologyEntry>", "Reading HmmTopology object, expected </Topology> or <TopologyEntry>, got " + t);
    tk.expect("<ForPhones>");
    std::vector<int> phones;
    while (true) {
      if (!tk.done() && tk.tok[tk.pos] == "</ForPhones>") { tk.next(); break; }
      phones.push_back(tk.next_int());
    }
    Entry entry;
    t = tk.next();
    while (t != "</TopologyEntry>") {
      KHG_REQUIRE(t == "<State>", "Expected </TopologyEntry> or <State>, got instead " + t);
      const int state = tk.next_int();
      KHG_REQUIRE(state == (int)entry.size(), "States are expected to be in order from zero, expected " + std::to_string(entry.size()) + ", got " + std::to_string(state));
      t = tk.next();
      if (t == "<PdfClass>") {
        entry.emplace_back(tk.next_int());
        t = tk.next();
        KHG_REQUIRE(t != "<SelfLoopPdfClass>", "pdf classes should be defined using <PdfClass> or <ForwardPdfClass>/<SelfLoopPdfClass> pair");
      } else if (t == "<ForwardPdfClass>") {
        const int fwd = tk.next_int();
        t = tk.next();
        KHG_REQUIRE(t == "<SelfLoopPdfClass>", "Expected <SelfLoopPdfClass>, got instead " + t);
        entry.emplace_back(fwd, tk.next_int());
        t = tk.next();
      } else {
        entry.emplace_back(kNoPdf);
      }
      while (t == "<Transition>") {
        const int dst = tk.next_int();
        const float prob = tk.next_float();
        entry.back().transitions.emplace_back(dst, prob);
        t = tk.next();
      }
      KHG_REQUIRE(t != "<Final>", "You are trying to read old-format topology with new Kaldi.");
      KHG_REQUIRE(t == "</State>", "Expected </State>, got instead " + t);
      t = tk.next();
    }
    const int idx = (int)entries_.size();
    entries_.push_back(entry);
    for (size_t i = 0; i < phones.size(); ++i) {
      const int ph = phones[i];
      KHG_REQUIRE(ph > 0, "phone > 0 assertion failed");
      if ((int)phone2idx_.size() <= ph) phone2idx_.resize((size_t)ph + 1, -1);
      KHG_REQUIRE(phone2idx_[(size_t)ph] == -1, "Phone with index " + std::to_string(i) + " appears in multiple topology entries.");
      phone2idx_[(size_t)ph] = idx;
      phones_.push_back(ph);
    }
  }
  std::sort(phones_.begin(), phones_.end());
  Check();
}

std::string HmmTopology::ToString() const {
  const bool hmm = IsHmm();
  std::string out = "<Topology> \n";
  for (size_t i = 0; i < entries_.size(); ++i) {
    out += "<TopologyEntry> \n<ForPhones> \n";
    for (size_t j = 0; j < phone2idx_.size(); ++j)
      if (phone2idx_[j] == (int)i) out += std::to_string(j) + " ";
    out += "\n</ForPhones> \n";
    for (size_t j = 0; j < entries_[i].size(); ++j) {
      const HmmState& st = entries_[i][j];
      out += "<State> " + std::to_string(j) + " ";
      if (st.forward_pdf_class != kNoPdf) {
        if (hmm) out += "<PdfClass> " + std::to_string(st.forward_pdf_class) + " ";
        else out += "<ForwardPdfClass> " + std::to_string(st.forward_pdf_class) + " <SelfLoopPdfClass> " + std::to_string(st.self_loop_pdf_class) + " ";
      }
      for (auto& tr : st.transitions) out += "<Transition> " + std::to_string(tr.first) + " " + FormatG(tr.second) + " ";
      out += "</State> \n";
    }
    out += "</TopologyEntry> \n";
  }
  return out + "</Topology> \n";
}

bool HmmTopology::IsHmm() const {
  for (int ph : phones_)
    for (const HmmState& st : TopologyForPhone(ph))
      if (st.forward_pdf_class != st.self_loop_pdf_class) return false;
  return true;
}
const HmmTopology::Entry& HmmTopology::TopologyForPhone(int phone) const {
  KHG_REQUIRE(phone >= 0 && phone < (int)phone2idx_.size() && phone2idx_[(size_t)phone] != -1 && phone2idx_[(size_t)phone] < (int)entries_.size(),
              "TopologyForPhone(), phone " + std::to_string(phone) + " not covered.");
  return entries_[(size_t)phone2idx_[(size_t)phone]];
}
int HmmTopology::NumPdfClasses(int phone) const {
  int m = 0;
  for (const HmmState& st : TopologyForPhone(phone)) m = std::max(m, std::max(st.forward_pdf_class, st.self_loop_pdf_class));
  return m + 1;
}
std::vector<int> HmmTopology::GetPhoneToNumPdfClasses() const {
  KHG_REQUIRE(!phones_.empty(), "HmmTopology: no phones");
  std::vector<int> out((size_t)phones_.back() + 1, -1);
  for (int ph : phones_) out[(size_t)ph] = NumPdfClasses(ph);
  return out;
}
int HmmTopology::MinLength(int phone) const {
  const Entry& entry = TopologyForPhone(phone);
  const int big = std::numeric_limits<int32_t>::max();
  std::vector<int> ml(entry.size(), big);
  ml[0] = entry[0].forward_pdf_class == -1 ? 0 : 1;
  bool changed = true;
  while (changed) {
    changed = false;
    for (size_t s = 0; s < entry.size(); ++s)
      for (auto& tr : entry[s].transitions) {
        const int nxt = tr.first;
        if (ml[s] == big) continue;
        const int v = ml[s] + (entry[(size_t)nxt].forward_pdf_class == -1 ? 0 : 1);
        if (v < ml[(size_t)nxt]) {
          ml[(size_t)nxt] = v;
          if (nxt < (int)s) changed = true;
        }
      }
  }
  return ml.back();
}
void HmmTopology::Check() const {
  KHG_REQUIRE(!entries_.empty() && !phones_.empty() && !phone2idx_.empty(), "HmmTopology::Check(), empty object.");
  std::vector<char> seen(entries_.size(), 0);
  for (int ph : phones_) {
    KHG_REQUIRE(ph >= 0 && ph < (int)phone2idx_.size() && phone2idx_[(size_t)ph] >= 0 && phone2idx_[(size_t)ph] < (int)entries_.size(),
                "HmmTopology::Check(), phone has no valid index.");
    seen[(size_t)phone2idx_[(size_t)ph]] = 1;
  }
  for (size_t i = 0; i < entries_.size(); ++i) {
    KHG_REQUIRE(seen[i], "HmmTopoloy::Check(), entry with no corresponding phones.");
    const Entry& entry = entries_[i];
    const int n = (int)entry.size();
    KHG_REQUIRE(n > 1, "HmmTopology::Check(), cannot only have one state (i.e., must have at least one emitting state).");
    KHG_REQUIRE(entry.back().transitions.empty(), "HmmTopology::Check(), last state must have no transitions.");
    KHG_REQUIRE(entry.back().forward_pdf_class == kNoPdf, "HmmTopology::Check(), last state must not be emitting.");
    std::vector<char> has_in((size_t)n, 0);
    std::set<int> classes;
    for (int j = 0; j < n; ++j) {
      const HmmState& st = entry[(size_t)j];
      double tot = 0.0;
      if (st.forward_pdf_class != kNoPdf) { classes.insert(st.forward_pdf_class); classes.insert(st.self_loop_pdf_class); }
      std::set<int> seen_t;
      for (auto& tr : st.transitions) {
        const int dst = tr.first;
        const double p = (double)tr.second;
        tot += p;
        KHG_REQUIRE(p > 0.0, "HmmTopology::Check(), negative or zero transition prob.");
        KHG_REQUIRE(!(dst == n - 1 && st.forward_pdf_class == kNoPdf), "We do not allow any state to be nonemitting and have a transition to the final-state");
        KHG_REQUIRE(dst >= 0 && dst < n, "HmmTopology::Check(), invalid dest state " + std::to_string(dst));
        KHG_REQUIRE(!seen_t.count(dst), "HmmTopology::Check(), duplicate transition found.");
        seen_t.insert(dst);
        has_in[(size_t)dst] = 1;
      }
      if (j + 1 < n) KHG_REQUIRE(tot > 0.0, "Non-final state must have transitions out.(with nonzero probability)");
      else KHG_REQUIRE(tot == 0.0, "assertion failed: tot_prob == 0.0");
    }
    for (int j = 1; j < n; ++j) KHG_REQUIRE(has_in[(size_t)j], "HmmTopology::Check, state " + std::to_string(j) + " has no input transitions.");
    KHG_REQUIRE(!classes.empty() && *classes.begin() == 0 && *classes.rbegin() == (int)classes.size() - 1,
                "HmmTopology::Check(), pdf_classes are expected to be contiguous and start from zero.");
  }
}

// ---- TransitionModel --------------------------------------------------------------------------------------------------
std::string TransitionModelTuple::ToString() const {
  return "TransitionModelTuple(phone=" + std::to_string(phone) + ",hmm_state=" + std::to_string(hmm_state) + ",forward_pdf=" + std::to_string(forward_pdf) +
         ",self_loop_pdf=" + std::to_string(self_loop_pdf) + ")";
}

TransitionModel::TransitionModel(const std::vector<std::vector<std::pair<int, int>>>& pdf_info, std::shared_ptr<HmmTopology> topo) : topo_(std::move(topo)) {
  KHG_REQUIRE(topo_ != nullptr, "TransitionModel: no topology");
  ComputeTuplesIsHmm(pdf_info);
  ComputeDerived();
  InitializeProbs();
  Check();
}
void TransitionModel::SetFromRead(std::shared_ptr<HmmTopology> topo, std::vector<TransitionModelTuple> tuples, std::vector<float> log_probs) {
  topo_ = std::move(topo);
  tuples_ = std::move(tuples);
  ComputeDerived();
  KHG_REQUIRE((int)log_probs.size() == NumTransitionIds() + 1, "TransitionModel::Read: <LogProbs> size does not match the tuples");
  log_probs_ = std::move(log_probs);
  ComputeDerivedOfProbs();
  Check();
}
void TransitionModel::SetState(std::vector<TransitionModelTuple> tuples, std::shared_ptr<HmmTopology> topo, std::vector<int> state2id, std::vector<int> id2state,
                               std::vector<int> id2pdf, int num_pdfs, std::vector<float> log_probs, std::vector<float> nsl) {
  tuples_ = std::move(tuples); topo_ = std::move(topo); state2id_ = std::move(state2id); id2state_ = std::move(id2state); id2pdf_ = std::move(id2pdf);
  num_pdfs_ = num_pdfs; log_probs_ = std::move(log_probs); nsl_ = std::move(nsl);
}

void TransitionModel::ComputeTuplesIsHmm(const std::vector<std::vector<std::pair<int, int>>>& pdf_info) {
  KHG_REQUIRE(topo_->IsHmm(), "TransitionModel: only is_hmm topologies (PdfClass) are supported by the monophone tree");
  std::map<std::pair<int, int>, std::vector<int>> to_hmm_state;
  for (int ph : topo_->phones()) {
    const auto& entry = topo_->TopologyForPhone(ph);
    for (size_t j = 0; j < entry.size(); ++j)
      if (entry[j].forward_pdf_class != kNoPdf) to_hmm_state[{ph, entry[j].forward_pdf_class}].push_back((int)j);
  }
  tuples_.clear();
  for (size_t pdf = 0; pdf < pdf_info.size(); ++pdf)
    for (auto& pc : pdf_info[pdf]) {
      auto it = to_hmm_state.find(pc);
      KHG_REQUIRE(it != to_hmm_state.end() && !it->second.empty(), "ComputeTuplesIsHmm: no HMM state emits this pdf-class");
      for (int hs : it->second) tuples_.push_back(TransitionModelTuple{pc.first, hs, (int)pdf, (int)pdf});
    }
  std::sort(tuples_.begin(), tuples_.end());
}
void TransitionModel::ComputeDerived() {
  const int n = (int)tuples_.size();
  state2id_.assign((size_t)n + 2, 0);
  int cur = 1;
  num_pdfs_ = 0;
  for (int ts = 1; ts <= n + 1; ++ts) {
    state2id_[(size_t)ts] = cur;
    if (ts <= n) {
      const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
      num_pdfs_ = std::max(num_pdfs_, std::max(1 + t.forward_pdf, 1 + t.self_loop_pdf));
      const auto& entry = topo_->TopologyForPhone(t.phone);
      KHG_REQUIRE(t.hmm_state >= 0 && t.hmm_state < (int)entry.size(), "TransitionModel: tuple with an HMM state outside its topology entry");
      cur += (int)entry[(size_t)t.hmm_state].transitions.size();
    }
  }
  id2state_.assign((size_t)cur, 0);
  id2pdf_.assign((size_t)cur, 0);
  for (int ts = 1; ts <= n; ++ts)
    for (int tid = state2id_[(size_t)ts]; tid < state2id_[(size_t)ts + 1]; ++tid) {
      id2state_[(size_t)tid] = ts;
      const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
      id2pdf_[(size_t)tid] = IsSelfLoopRaw(tid) ? t.self_loop_pdf : t.forward_pdf;
    }
}
bool TransitionModel::IsSelfLoopRaw(int tid) const {
  const int ts = id2state_[(size_t)tid], idx = tid - state2id_[(size_t)ts];
  const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
  const auto& tr = topo_->TopologyForPhone(t.phone)[(size_t)t.hmm_state].transitions;
  return idx < (int)tr.size() && tr[(size_t)idx].first == t.hmm_state;
}
bool TransitionModel::IsFinal(int tid) const {
  ChkTid(tid);
  const int ts = id2state_[(size_t)tid], idx = tid - state2id_[(size_t)ts];
  const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
  const auto& entry = topo_->TopologyForPhone(t.phone);
  return entry[(size_t)t.hmm_state].transitions[(size_t)idx].first + 1 == (int)entry.size();
}
int TransitionModel::SelfLoopOf(int ts) const {
  KHG_REQUIRE(ts >= 1 && ts <= NumTransitionStates(), "trans_state out of range");
  const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
  const auto& tr = topo_->TopologyForPhone(t.phone)[(size_t)t.hmm_state].transitions;
  for (size_t idx = 0; idx < tr.size(); ++idx)
    if (tr[idx].first == t.hmm_state) return state2id_[(size_t)ts] + (int)idx;
  return 0;
}
void TransitionModel::InitializeProbs() {
  const int nt = NumTransitionIds();
  log_probs_.assign((size_t)nt + 1, 0.0f);
  for (int tid = 1; tid <= nt; ++tid) {
    const int ts = id2state_[(size_t)tid], idx = tid - state2id_[(size_t)ts];
    const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
    const float prob = topo_->TopologyForPhone(t.phone)[(size_t)t.hmm_state].transitions[(size_t)idx].second;
    KHG_REQUIRE(prob > 0.0f, "TransitionModel::InitializeProbs, zero probability [should remove that entry in the topology]");
    log_probs_[(size_t)tid] = std::log(prob);
  }
  ComputeDerivedOfProbs();
}
void TransitionModel::ComputeDerivedOfProbs() {
  const int n = NumTransitionStates();
  nsl_.assign((size_t)n + 1, 0.0f);
  for (int ts = 1; ts <= n; ++ts) {
    const int tid = SelfLoopOf(ts);
    if (tid == 0) { nsl_[(size_t)ts] = 0.0f; continue; }
    const float slp = std::exp(log_probs_[(size_t)tid]);      // libm's float expf / logf, as khg_transition_mle_update calls them
    float p = (float)(1.0 - (double)slp);
    if (p <= 0.0f) p = 1.0e-10f;
    nsl_[(size_t)ts] = std::log(p);
  }
}
void TransitionModel::Check() const {
  KHG_REQUIRE(NumTransitionIds() != 0 && NumTransitionStates() != 0, "TransitionModel::Check failed");
  for (int tid = 1; tid <= NumTransitionIds(); ++tid) {
    const float lp = log_probs_[(size_t)tid];
    KHG_REQUIRE(lp <= 0.0f && lp - lp == 0.0f, "TransitionModel::Check: bad log prob");
  }
}
int TransitionModel::TupleToTransitionState(int phone, int hmm_state, int pdf, int self_loop_pdf) const {
  const TransitionModelTuple t{phone, hmm_state, pdf, self_loop_pdf};
  auto it = std::lower_bound(tuples_.begin(), tuples_.end(), t);
  KHG_REQUIRE(it != tuples_.end() && *it == t, "TransitionModel::TupleToTransitionState, tuple not found. (incompatible tree and model?)");
  return (int)(it - tuples_.begin()) + 1;
}
int TransitionModel::PairToTransitionId(int ts, int idx) const {
  KHG_REQUIRE(ts > 0 && ts <= (int)tuples_.size() && idx >= 0 && idx < state2id_[(size_t)ts + 1] - state2id_[(size_t)ts], "PairToTransitionId: out of range");
  return state2id_[(size_t)ts] + idx;
}
float TransitionModel::GetNonSelfLoopLogProb(int ts) const {
  KHG_REQUIRE(ts >= 0 && ts < (int)nsl_.size(), "trans_state out of range");
  return nsl_[(size_t)ts];
}
float TransitionModel::GetTransitionLogProbIgnoringSelfLoops(int tid) const {
  KHG_REQUIRE(!IsSelfLoop(tid), "GetTransitionLogProbIgnoringSelfLoops: self-loop");
  return log_probs_[(size_t)tid] - nsl_[(size_t)id2state_[(size_t)tid]];
}

std::pair<float, float> TransitionModel::MleUpdate(const double* stats, size_t n, const MleTransitionUpdateConfig& cfg) {
  KHG_REQUIRE((int)n == NumTransitionIds() + 1, "stats.size() == NumTransitionIds() + 1 assertion failed");
  if (cfg.share_for_pdfs) return MleUpdateShared(stats, cfg);
  std::vector<int32_t> s2i(state2id_.begin(), state2id_.end()), slo((size_t)NumTransitionStates() + 1, 0);
  for (int ts = 1; ts <= NumTransitionStates(); ++ts) slo[(size_t)ts] = SelfLoopOf(ts);
  float oi = 0, cnt = 0;
  CApi(khg_transition_mle_update(NumTransitionStates(), s2i.data(), slo.data(), stats, cfg.floor, cfg.mincount, log_probs_.data(), nsl_.data(), &oi, &cnt));
  return {oi, cnt};
}

// TransitionModel::MleUpdateShared (csrc/transition-model.cc:531-655): one set of transition probabilities for all transition-states
// that share a pdf.  Arithmetic as there: counts and their total in double, the new probabilities a float vector (normalised and
// floored three times), the objective change summed in float.
std::pair<float, float> TransitionModel::MleUpdateShared(const double* st, const MleTransitionUpdateConfig& cfg) {
  std::map<int, std::set<int>> groups;          // pdf -> transition-states
  const bool hmm = topo_->IsHmm();
  for (int ts = 1; ts <= NumTransitionStates(); ++ts) {
    const TransitionModelTuple& t = tuples_[(size_t)ts - 1];
    groups[t.forward_pdf].insert(ts);
    if (!hmm) groups[t.self_loop_pdf].insert(ts);
  }
  float count_sum = 0.0f, objf_sum = 0.0f;
  const float floor = cfg.floor;
  for (auto& kv : groups) {
    const std::set<int>& tstates = kv.second;
    const int one = *tstates.begin();
    const int n = state2id_[(size_t)one + 1] - state2id_[(size_t)one];
    if (n <= 1) continue;
    std::vector<double> counts((size_t)n, 0.0);
    double pdf_tot = 0.0;
    for (int ts : tstates) {
      KHG_REQUIRE(state2id_[(size_t)ts + 1] - state2id_[(size_t)ts] == n,
                  "Mismatch in #transition indices: you cannot use the --share-for-pdfs option with this topology and sharing scheme.");
      for (int k = 0; k < n; ++k) {
        const double acc = st[(size_t)state2id_[(size_t)ts] + k];
        counts[(size_t)k] += acc;
        pdf_tot += acc;
      }
    }
    count_sum = (float)((double)count_sum + pdf_tot);          // float += double
    if (pdf_tot < (double)cfg.mincount) continue;
    std::vector<float> old_p((size_t)n), new_p((size_t)n);
    for (int k = 0; k < n; ++k) {
      old_p[(size_t)k] = std::exp(log_probs_[(size_t)state2id_[(size_t)one] + k]);       // GetTransitionProb
      new_p[(size_t)k] = (float)(counts[(size_t)k] / pdf_tot);
    }
    for (int it = 0; it < 3; ++it) {                           // keep flooring + renormalising three times
      float s = 0.0f;
      for (float x : new_p) s += x;
      for (float& x : new_p) { x = x / s; x = std::max(x, floor); }
    }
    for (int k = 0; k < n; ++k) {
      const float dlog = std::log(new_p[(size_t)k]) - std::log(old_p[(size_t)k]);
      objf_sum = (float)((double)objf_sum + counts[(size_t)k] * (double)dlog);           // float += double
    }
    for (int ts : tstates)
      for (int k = 0; k < n; ++k) {
        const float lp = std::log(new_p[(size_t)k]);
        KHG_REQUIRE(std::isfinite(lp), "Log probs is inf or NaN: error in update or bad stats?");
        log_probs_[(size_t)state2id_[(size_t)ts] + k] = lp;
      }
  }
  ComputeDerivedOfProbs();
  return {objf_sum, count_sum};
}

std::vector<uint8_t> TransitionModel::IsSelfLoopArray() const {
  std::vector<uint8_t> a((size_t)NumTransitionIds() + 1, 0);
  for (int tid = 1; tid <= NumTransitionIds(); ++tid) a[(size_t)tid] = IsSelfLoopRaw(tid) ? 1 : 0;
  return a;
}
std::vector<float> TransitionModel::ScaledTransCost(float transition_scale, float self_loop_scale) const {
  std::vector<float> out((size_t)NumTransitionIds() + 1, 0.0f);
  std::vector<int32_t> i2s(id2state_.begin(), id2state_.end());
  const std::vector<uint8_t> sl = IsSelfLoopArray();
  CApi(khg_scaled_trans_cost(NumTransitionIds(), log_probs_.data(), nsl_.data(), i2s.data(), sl.data(), transition_scale, self_loop_scale, out.data()));
  return out;
}
std::string TransitionModel::ToString() const {
  std::string out = "<TransitionModel> \n" + topo_->ToString() + "<Triples> " + std::to_string(tuples_.size()) + " \n";
  for (auto& t : tuples_) out += std::to_string(t.phone) + " " + std::to_string(t.hmm_state) + " " + std::to_string(t.forward_pdf) + " \n";
  out += "</Triples> \n<LogProbs> \n [ ";
  for (size_t i = 0; i < log_probs_.size(); ++i) { if (i) out += " "; out += FormatG(log_probs_[i]); }
  return out + " ]\n</LogProbs> \n</TransitionModel> \n";
}

bool GetPdfsForPhones(const TransitionModel& tm, const std::vector<int>& phones, std::vector<int>* pdfs) {
  for (size_t i = 1; i < phones.size(); ++i) KHG_REQUIRE(phones[i - 1] < phones[i], "IsSortedAndUniq(phones) assertion failed");
  const std::set<int> ps(phones.begin(), phones.end());
  std::set<int> out;
  for (auto& t : tm.tuples())
    if (ps.count(t.phone)) { out.insert(t.forward_pdf); out.insert(t.self_loop_pdf); }
  bool ok = true;
  for (auto& t : tm.tuples())
    if ((out.count(t.forward_pdf) || out.count(t.self_loop_pdf)) && !ps.count(t.phone)) ok = false;
  pdfs->assign(out.begin(), out.end());
  return ok;
}

khg_tm* TransitionModel::DeviceTm(khg_ctx* ctx) const {
  if (!dev_) dev_ = std::make_shared<Dev>();
  std::lock_guard<std::mutex> lk(dev_->mu);
  if (dev_->h && dev_->ctx == ctx && dev_->id2pdf == id2pdf_) return dev_->h;
  if (dev_->h) { khg_tm_destroy(dev_->h); dev_->h = nullptr; }
  std::vector<int32_t> t(id2pdf_.begin(), id2pdf_.end());
  CApi(khg_tm_create(ctx, NumTransitionIds(), t.data(), &dev_->h));
  dev_->ctx = ctx; dev_->id2pdf = id2pdf_;
  return dev_->h;
}

}  // namespace khg
