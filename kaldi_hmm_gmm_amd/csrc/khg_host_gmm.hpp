// Host-side C++ classes behind the reference's names: DiagGmm, AmDiagGmm, AccumDiagGmm, AccumAmDiagGmm,
// MleDiagGmmOptions, the GmmUpdateFlags helpers, GetSplitTargets and the MLE update entry points.
//
// Mirrors (reference, /root/reference/kaldi-hmm-gmm/):
//   csrc/diag-gmm.{h,cc}, csrc/am-diag-gmm.{h,cc}, csrc/model-common.{h,cc}, csrc/mle-diag-gmm.{h,cc},
//   csrc/mle-am-diag-gmm.{h,cc}; their pybind surface python/csrc/{diag-gmm,am-diag-gmm,model-common,mle-diag-gmm,
//   mle-am-diag-gmm}.cc is what khg_py_host.cpp binds these classes with.
// Storage is plain row-major std::vector (the reference's Eigen row-major matrices); the float arithmetic of every
// setter is one IEEE operation per element in the order the reference's expressions evaluate.  Everything that scores
// features (LogLikelihood*, ComponentPosteriors, AccumulateFromDiag) goes to the GPU through the C-ABI (K1 / K3 of
// include/khg_hip.h) -- there is no CPU evaluation here -- and the M-step is khg_mle_am_diag_gmm_update (khg_host.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <atomic>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/khg_hip.h"

namespace khg {

struct Error : std::runtime_error {      // KHG_ERR of csrc/log.h:46-53 (std::runtime_error -> RuntimeError in Python)
  using std::runtime_error::runtime_error;
};
inline void CApi(int rc) {
  if (rc != KHG_OK) { const char* m = khg_last_error(); throw Error(m ? m : "khg error"); }
}
#define KHG_REQUIRE(cond, msg) do { if (!(cond)) throw ::khg::Error(msg); } while (0)

// A sum in numpy's order (pairwise, 8 running partial sums per <= 128-element block): the Python shells these classes
// replace normalised weights with ndarray.sum(), and results are kept bit-identical to them.
template <class T>
inline T NpSum(const T* a, size_t n) {
  if (n < 8) { T r = 0; for (size_t i = 0; i < n; ++i) r += a[i]; return r; }
  if (n <= 128) {
    T r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    size_t i = 8;
    for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  size_t n2 = n / 2; n2 -= n2 % 8;
  return NpSum(a, n2) + NpSum(a + n2, n - n2);
}

// ---- the default device context of the single-object calls --------------------------------------------------------------
khg_ctx* DefaultCtx();                 // created on first use on device $KHG_DEVICE (0)
void SetDefaultCtx(khg_ctx* borrowed); // the Python side hands over the context the tests / scripts made

// K1 for one feature matrix: -> [npdf][N] log-likelihoods of every frame under the listed pdfs of a flat model
std::vector<float> GpuLoglikes(int P, int D, const int32_t* gauss_off, const float* gconsts, const float* miv, const float* iv,
                               const float* feats, int64_t N, const int32_t* pdfs, int npdf);
// the same on a model that is on the device already (AmDiagGmm::DeviceModel)
std::vector<float> GpuLoglikesOn(khg_model* model, int D, const float* feats, int64_t N, const int32_t* pdfs, int npdf);
struct GpuStats {
  std::vector<double> occ, mean_acc, var_acc;
  double total_frames = 0.0, total_log_like = 0.0;
};
// K3 with explicit per-frame pdf ids
GpuStats GpuAccStats(int P, int D, const int32_t* gauss_off, const float* gconsts, const float* miv, const float* iv,
                     const float* feats, int64_t N, const int32_t* frame_pdf, float weight);

// fills standard normal deviates (RandnVector / RandGauss): rows == 0 asks for a vector of `cols`, else for a rows x cols matrix
using RandnFn = std::function<void(float* out, size_t rows, size_t cols)>;

// ---- csrc/model-common.h:18-26 -----------------------------------------------------------------------------------------
enum GmmUpdateFlags : int { kGmmMeans = 1, kGmmVariances = 2, kGmmWeights = 4, kGmmTransitions = 8, kGmmAll = 15 };
int StrToGmmFlags(const std::string& s);       // csrc/model-common.cc:99-124
std::string GmmFlagsToStr(int flags);          // :126-145
int AugmentGmmFlags(int flags);                // :72-85
std::vector<int32_t> GetSplitTargets(const std::vector<float>& state_occs, int32_t target_components, float power,
                                     float min_count);   // :29-70
// the same with min_count as the double a Python caller passes (the comparison (n + 1) * min_count >= occ is made in double)
std::vector<int32_t> GetSplitTargetsD(const std::vector<float>& state_occs, int32_t target_components, float power, double min_count);

// Every mutation of a DiagGmm / of an AmDiagGmm's list of pdfs takes a fresh value of ONE process-wide counter, so "the largest
// version among the parts" changes whenever any part changed: what AmDiagGmm::DeviceModel keys its cached device handle by.
inline uint64_t NextVersion() { static std::atomic<uint64_t> g{1}; return ++g; }

// ---- csrc/diag-gmm.h ---------------------------------------------------------------------------------------------------
class DiagGmm {
 public:
  DiagGmm() = default;
  DiagGmm(int nmix, int dim) { Resize(nmix, dim); }
  uint64_t version() const { return version_; }
  void Touch() { version_ = NextVersion(); }
  // the weighted concatenation of several mixtures, gconsts computed (csrc/diag-gmm.cc:68-101)
  explicit DiagGmm(const std::vector<std::pair<float, const DiagGmm*>>& gmms);
  void Resize(int nmix, int dim);                              // csrc/diag-gmm.cc:30-47 (vars = 1)
  void CopyFromDiagGmm(const DiagGmm& o) { *this = o; Touch(); }
  int NumGauss() const { return G_; }
  int Dim() const { return D_; }
  bool ValidGconsts() const { return valid_gconsts_; }
  const std::vector<float>& gconsts() const { return gconsts_; }
  const std::vector<float>& weights() const { return weights_; }
  const std::vector<float>& inv_vars() const { return inv_vars_; }
  const std::vector<float>& means_invvars() const { return means_invvars_; }
  std::vector<float>& mutable_gconsts() { Touch(); return gconsts_; }
  std::vector<float>& mutable_weights() { Touch(); return weights_; }
  std::vector<float>& mutable_inv_vars() { Touch(); return inv_vars_; }
  std::vector<float>& mutable_means_invvars() { Touch(); return means_invvars_; }
  void set_valid_gconsts(bool v) { Touch(); valid_gconsts_ = v; }
  // raw replacement of all parameters (unpickling, the flat store of AmDiagGmm, M-step results)
  void SetRaw(int G, int D, const float* w, const float* iv, const float* miv, const float* gc /* may be NULL */);
  std::vector<float> GetMeans() const;                         // :956-958
  std::vector<float> GetVars() const;
  void SetWeights(const float* w, size_t n);                   // :940-1021
  void SetMeans(const float* m, size_t rows, size_t cols);
  void SetInvVars(const float* v, size_t rows, size_t cols);
  void SetInvVarsAndMeans(const float* v, const float* m, size_t rows, size_t cols);
  void SetComponentWeight(int g, float w);
  void SetComponentMean(int g, const float* v, size_t n);
  void SetComponentInvVar(int g, const float* v, size_t n);
  std::vector<float> GetComponentMean(int g) const;
  std::vector<float> GetComponentVariance(int g) const;
  void RemoveComponent(int g, bool renorm_weights);            // :868-938
  void RemoveComponents(std::vector<int> gauss, bool renorm_weights);   // :853-866
  int ComputeGconsts();                                        // :103-147 (khg_compute_gconsts) -> number of "bad" components
  void NeedGconsts() const { KHG_REQUIRE(valid_gconsts_, "Must call ComputeGconsts() before computing likelihood"); }
  // likelihoods: K1 on the GPU
  float LogLikelihood(const float* data, size_t n) const;      // :150-165
  std::vector<float> LogLikelihoods(const float* data, size_t n) const;                     // :167-176 -> [G]
  std::vector<float> LogLikelihoodsMatrix(const float* data, size_t rows, size_t cols) const;  // :177-189 -> [N][G]
  double ComponentPosteriors(const float* data, size_t n, std::vector<float>* post) const;  // :368-392, K3 -> log-like
  // Gaussian selection (csrc/diag-gmm.cc:201-365): the num_gselect best components of a frame, best first, and the log-sum of
  // their likelihoods (K1 for the scores, the selection itself on the host)
  float GaussianSelection(const float* data, size_t n, int num_gselect, std::vector<int32_t>* output) const;
  float GaussianSelectionMatrix(const float* data, size_t rows, size_t cols, int num_gselect, std::vector<std::vector<int32_t>>* output) const;
  float GaussianSelectionPreselect(const float* data, size_t n, const std::vector<int32_t>& preselect, int num_gselect, std::vector<int32_t>* output) const;
  void Split(int target_components, float perturb_factor, std::vector<int>* history, const RandnFn& randn);   // :780-851
  std::vector<int> Merge(int target_components);               // :557-759 (khg_diag_gmm_merge) -> history
  void Perturb(float perturb_factor, const RandnFn& randn);    // :463-484
  std::vector<float> Generate(const RandnFn& randn) const;     // :410-446
  void Interpolate(float rho, const DiagGmm& source, int flags);   // :486-520

 private:
  int G_ = 0, D_ = 0;
  bool valid_gconsts_ = false;
  uint64_t version_ = NextVersion();
  std::vector<float> gconsts_, weights_, inv_vars_, means_invvars_;   // [G], [G], [G][D], [G][D]
};

// ---- csrc/am-diag-gmm.h:96 ---------------------------------------------------------------------------------------------
class AmDiagGmm {
 public:
  AmDiagGmm() = default;
  // a copy shares the DiagGmm objects (as the vector of shared pointers always did) but never the cached device handle
  AmDiagGmm(const AmDiagGmm& o) : pdfs_(o.pdfs_) {}
  AmDiagGmm& operator=(const AmDiagGmm& o) { if (this != &o) { pdfs_ = o.pdfs_; struct_version_ = NextVersion(); } return *this; }
  int Dim() const { return pdfs_.empty() ? 0 : pdfs_[0]->Dim(); }
  int NumPdfs() const { return (int)pdfs_.size(); }
  int NumGauss() const { int n = 0; for (auto& p : pdfs_) n += p->NumGauss(); return n; }
  void Init(const DiagGmm& proto, int num_pdfs);
  void AddPdf(const DiagGmm& gmm);
  void CopyFromAmDiagGmm(const AmDiagGmm& o);
  const std::shared_ptr<DiagGmm>& GetPdf(int i) const {
    KHG_REQUIRE(i >= 0 && i < (int)pdfs_.size(), "pdf_index out of range");
    return pdfs_[i];
  }
  int ComputeGconsts() { int n = 0; for (auto& p : pdfs_) n += p->ComputeGconsts(); return n; }
  void SplitByCount(const std::vector<float>& state_occs, int target, float perturb, float power, double min_count,
                    const RandnFn& randn);                     // csrc/am-diag-gmm.cc:72-90
  void MergeByCount(const std::vector<float>& state_occs, int target, float power, double min_count);   // :91-108
  // flat ragged view used by the device path (gconsts must be valid)
  void Flat(std::vector<int32_t>* go, std::vector<float>* gc, std::vector<float>* w, std::vector<float>* miv, std::vector<float>* iv) const;
  void SetFlat(const int32_t* go, const float* w, const float* gc, const float* miv, const float* iv);
  std::vector<std::shared_ptr<DiagGmm>>& pdfs() { struct_version_ = NextVersion(); return pdfs_; }
  const std::vector<std::shared_ptr<DiagGmm>>& pdfs() const { return pdfs_; }
  // changes whenever a parameter of any pdf, or the list of pdfs, may have changed (get_pdf(i) hands out references: the pdfs
  // carry their own versions)
  uint64_t Version() const;
  // The model on the device, uploaded when first asked for and again only after Version() moved: the reference's scripts pass the
  // same AmDiagGmm to gmm_align_compiled / gmm_acc_stats_ali once per utterance (egs/yesno/train.py:170-202).
  khg_model* DeviceModel(khg_ctx* ctx) const;
  void DropDeviceModel() const { std::lock_guard<std::mutex> lk(dev_->mu); dev_->Release(); }

 private:
  struct Dev {
    std::mutex mu; khg_ctx* ctx = nullptr; khg_model* h = nullptr; uint64_t version = 0;
    void Release() { if (h) khg_model_destroy(h); h = nullptr; }
    ~Dev() { Release(); }
  };
  std::vector<std::shared_ptr<DiagGmm>> pdfs_;
  uint64_t struct_version_ = NextVersion();
  std::shared_ptr<Dev> dev_ = std::make_shared<Dev>();     // per object: CopyFromAmDiagGmm copies the pdfs, not the handle
};

// ---- csrc/mle-diag-gmm.h:23-45 -----------------------------------------------------------------------------------------
struct MleDiagGmmOptions {
  std::vector<double> variance_floor_vector;
  float min_gaussian_weight = 1.0e-05f;
  float min_gaussian_occupancy = 10.0f;
  double min_variance = 0.001;
  bool remove_low_count_gaussians = true;
  khg_mle_options C() const {
    khg_mle_options o;
    o.min_gaussian_weight = min_gaussian_weight; o.min_gaussian_occupancy = min_gaussian_occupancy; o.min_variance = min_variance;
    o.remove_low_count_gaussians = remove_low_count_gaussians ? 1 : 0;
    o.variance_floor_vector = variance_floor_vector.empty() ? nullptr : variance_floor_vector.data();
    return o;
  }
  std::string ToString() const;
};

struct MapDiagGmmOptions {      // csrc/mle-diag-gmm.h:47-66
  float mean_tau = 10.0f, variance_tau = 50.0f, weight_tau = 10.0f;
  std::string ToString() const;
};

// ---- csrc/mle-diag-gmm.h:68-181 ----------------------------------------------------------------------------------------
class AccumDiagGmm {
 public:
  AccumDiagGmm() = default;
  void Resize(int num_gauss, int dim, int flags);
  int NumGauss() const { return G_; }
  int Dim() const { return D_; }
  int Flags() const { return flags_; }
  std::vector<double>& occupancy() { return occ_; }
  std::vector<double>& mean_accumulator() { return mean_; }
  std::vector<double>& variance_accumulator() { return var_; }
  const std::vector<double>& occupancy() const { return occ_; }
  const std::vector<double>& mean_accumulator() const { return mean_; }
  const std::vector<double>& variance_accumulator() const { return var_; }
  void SetZero(int flags);
  void Scale(float f, int flags);
  void AccumulateForComponent(const float* data, size_t n, int comp, float weight);      // csrc/mle-diag-gmm.cc:100-121
  void AccumulateFromPosteriors(const float* data, size_t n, const float* post, size_t np);   // :123-143
  float AccumulateFromDiag(const DiagGmm& gmm, const float* data, size_t n, float weight);   // :145-158 (K3)
  void AddStatsForComponent(int g, double occ, const double* x, size_t nx, const double* x2, size_t nx2);
  void Add(float scale, const AccumDiagGmm& acc);              // :176-188
  void SmoothStats(float tau);                                 // :192-203
  void SmoothWithAccum(float tau, const AccumDiagGmm& src);    // :209-226
  void SmoothWithModel(float tau, const DiagGmm& gmm);         // :228-241
  void AddRaw(const double* occ, const double* mean, const double* var);   // device statistics of this pdf's rows

 private:
  void CheckFlags(int flags) const { KHG_REQUIRE(!(flags & ~flags_), "Flags in argument do not match the active accumulators"); }
  int G_ = 0, D_ = 0, flags_ = 0;
  std::vector<double> occ_, mean_, var_;
};

// (objf_change, count, floored_elements, floored_gaussians, removed) of csrc/mle-diag-gmm.cc:243-390
struct MleUpdateResult { float objf_change = 0, count = 0; int32_t floored_elements = 0, floored_gaussians = 0, removed = 0; };
MleUpdateResult MleDiagGmmUpdate(const MleDiagGmmOptions& cfg, const AccumDiagGmm& acc, int flags, DiagGmm* gmm);
// the flat form both updates go through (khg_mle_am_diag_gmm_update): w / miv / iv in-out (compacted), gc / new_off out
MleUpdateResult MleFlatUpdate(const MleDiagGmmOptions& cfg, int P, int D, const int32_t* gauss_off, const double* occ, const double* mean_acc,
                              const double* var_acc, int acc_flags, int flags, std::vector<float>* w, std::vector<float>* gc,
                              std::vector<float>* miv, std::vector<float>* iv, std::vector<int32_t>* new_off);
float MlObjective(const DiagGmm& gmm, const AccumDiagGmm& acc);   // :479-499
// MAP re-estimation with per-quantity smoothing counts (csrc/mle-diag-gmm.cc:392-477) -> (objf_change, count); runs on the host in
// fp64 like the reference (a speaker-adaptation-sized update: not on the EM hot path)
std::pair<float, float> MapDiagGmmUpdate(const MapDiagGmmOptions& cfg, const AccumDiagGmm& acc, int flags, DiagGmm* gmm);

// ---- csrc/mle-am-diag-gmm.h:18-97 --------------------------------------------------------------------------------------
class TransitionModel;
class AccumAmDiagGmm {
 public:
  AccumAmDiagGmm() = default;
  AccumAmDiagGmm(const AccumAmDiagGmm& o) { *this = o; }
  AccumAmDiagGmm& operator=(const AccumAmDiagGmm& o);          // deep copy of the host accumulators (the source's device sums flushed first)
  void Init(const AmDiagGmm& model, int dim /* <= 0: the model's */, int flags);
  // csrc/mle-am-diag-gmm.cc:35-39: zeroes the flagged statistics of every pdf and nothing else -- the pending device sums are folded
  // into the host accumulators first, so the unflagged parts and total_frames_ / total_log_like_ survive as in the reference
  void SetZero(int flags) { Flush(); for (auto& a : accs_) a->SetZero(flags); }
  int NumAccs() const { return (int)accs_.size(); }
  int Dim() const { return accs_.empty() ? 0 : accs_[0]->Dim(); }
  float TotStatsCount() const;
  float TotCount() const { Flush(); return (float)total_frames_; }          // csrc/mle-am-diag-gmm.h:75
  float TotLogLike() const { Flush(); return (float)total_log_like_; }      // :76
  const std::shared_ptr<AccumDiagGmm>& Acc(int i) const {
    KHG_REQUIRE(i >= 0 && i < (int)accs_.size(), "index >= 0 && index < NumAccs() assertion failed");
    Flush();
    return accs_[i];
  }
  float AccumulateForGmm(const AmDiagGmm& model, const float* data, size_t n, int gmm_index, float weight);   // .cc:41-52
  float AccumulateForGmmTwoFeats(const AmDiagGmm& model, const float* d1, size_t n1, const float* d2, size_t n2, int gmm_index,
                                 float weight);                // .cc:54-76
  void AccumulateFromPosteriors(const AmDiagGmm& model, const float* data, size_t n, int gmm_index, const float* post, size_t np);
  void AccumulateForGaussian(const AmDiagGmm& am, const float* data, size_t n, int gmm_index, int gauss_index, float weight);
  void Add(float scale, const AccumAmDiagGmm& other);          // .cc:119-128
  void Scale(float scale);
  // a downloaded device block (after any all-reduce) added into these accumulators
  void AddDeviceStats(const int32_t* gauss_off, const double* occ, const double* mean, const double* var, int D, double total_frames,
                      double total_log_like);
  std::vector<std::shared_ptr<AccumDiagGmm>>& accs() { Flush(); return accs_; }
  const std::vector<std::shared_ptr<AccumDiagGmm>>& accs() const { Flush(); return accs_; }
  double total_frames() const { Flush(); return total_frames_; }
  double total_log_like() const { Flush(); return total_log_like_; }
  void set_total_frames(double v) { Flush(); total_frames_ = v; }
  void set_total_log_like(double v) { Flush(); total_log_like_ = v; }

  // scripts/gmm_acc_stats_ali.py:46-58 for one utterance (or several, concatenated): K3 over (feats, ali) into statistics that STAY
  // ON THE DEVICE between calls -- the reference's caller invokes this once per utterance (egs/yesno/train.py:191-202), and a
  // 207 MB block (5000 x 64 x 40) cannot cross PCIe per call.  Every host-side reader above first adds the pending device sums into
  // the host accumulators (Flush: one download per EM pass).  -> the log-likelihood of these frames (tot_like_this_file).
  double AccumulateAli(const AmDiagGmm& model, const TransitionModel& tm, const float* feats, const int64_t* frame_off, int n_utt, const int32_t* ali,
                       float weight = 1.0f);
  double AccumulateOnDevice(const AmDiagGmm& model, khg_tm* dt, int num_tids, const float* feats, const int64_t* frame_off, int n_utt, const int32_t* ali,
                            float weight);                     // khg_host_align.cpp: the one K3 call behind AccumulateAli and AccumulateForGmm
  bool HasDeviceStats() const { return dev_ && dev_->pending; }
  void Flush() const;        // pending device sums -> host accumulators (then the device block is zero again)

 private:
  void Chk(int i) const { KHG_REQUIRE(i >= 0 && i < (int)accs_.size(), "gmm_index >= 0 && gmm_index < NumAccs() assertion failed"); }
  void DropDevice() const { dev_.reset(); }                    // pending device sums are DISCARDED (Init / SetZero)
  struct Dev {
    khg_ctx* ctx = nullptr; khg_accs* h = nullptr;
    uint64_t model_version = 0; int num_tids = 0, D = 0;
    std::vector<int32_t> gauss_off;
    double seen_frames = 0.0, seen_ll = 0.0;                   // the block's running totals at the last call (a call's own log-like = the difference)
    bool pending = false;
    khg_tm* pdf_tm = nullptr;                                  // the per-frame entry points' table (transition-id = pdf + 1), when the block was made by them
    ~Dev() { if (h) khg_accs_destroy(h); if (pdf_tm) khg_tm_destroy(pdf_tm); }
  };
  mutable std::vector<std::shared_ptr<AccumDiagGmm>> accs_;
  mutable double total_frames_ = 0.0, total_log_like_ = 0.0;
  mutable std::shared_ptr<Dev> dev_;
};

// csrc/mle-am-diag-gmm.cc:153-202 -> (objf_change, count); am_gmm updated in place
MleUpdateResult MleAmDiagGmmUpdate(const MleDiagGmmOptions& cfg, const AccumAmDiagGmm& acc, int flags, AmDiagGmm* am_gmm);
// csrc/mle-am-diag-gmm.cc:204-227: MapDiagGmmUpdate pdf by pdf, the float sums of its two outputs
std::pair<float, float> MapAmDiagGmmUpdate(const MapDiagGmmOptions& cfg, const AccumAmDiagGmm& acc, int flags, AmDiagGmm* am_gmm);

}  // namespace khg
