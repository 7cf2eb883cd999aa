// Implementation of khg_host_gmm.hpp: see the header for what each class mirrors in the reference.
#include "khg_host_gmm.hpp"

#include <cstdio>
#include <cstdlib>
#include <limits>

namespace khg {

// ---- default context + single-object GPU calls ------------------------------------------------------------------------
namespace {
khg_ctx* g_default_ctx = nullptr;
bool g_default_owned = false;

struct ModelH { khg_model* h = nullptr; ~ModelH() { if (h) khg_model_destroy(h); } };
struct TmH { khg_tm* h = nullptr; ~TmH() { if (h) khg_tm_destroy(h); } };
struct UttsH { khg_utts* h = nullptr; ~UttsH() { if (h) khg_utts_destroy(h); } };
struct AccsH { khg_accs* h = nullptr; ~AccsH() { if (h) khg_accs_destroy(h); } };
}  // namespace

khg_ctx* DefaultCtx() {
  if (!g_default_ctx) {
    const char* e = std::getenv("KHG_DEVICE");
    CApi(khg_ctx_create(e ? std::atoi(e) : 0, nullptr, &g_default_ctx));
    g_default_owned = true;
  }
  return g_default_ctx;
}
void SetDefaultCtx(khg_ctx* borrowed) {
  if (g_default_ctx && g_default_owned && g_default_ctx != borrowed) khg_ctx_destroy(g_default_ctx);
  g_default_ctx = borrowed;
  g_default_owned = false;
}

std::vector<float> GpuLoglikes(int P, int D, const int32_t* gauss_off, const float* gconsts, const float* miv, const float* iv,
                               const float* feats, int64_t N, const int32_t* pdfs, int npdf) {
  khg_ctx* ctx = DefaultCtx();
  ModelH m;
  CApi(khg_model_create(ctx, P, D, gauss_off, gconsts, miv, iv, &m.h));
  return GpuLoglikesOn(m.h, D, feats, N, pdfs, npdf);
}
std::vector<float> GpuLoglikesOn(khg_model* model, int D, const float* feats, int64_t N, const int32_t* pdfs, int npdf) {
  khg_ctx* ctx = DefaultCtx();
  struct { khg_model* h; } m{model};
  UttsH u;
  const int64_t fo[2] = {0, N};
  CApi(khg_utts_create(ctx, nullptr, 1, D, fo, feats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &u.h));
  CApi(khg_utts_set_pdf_list(u.h, npdf, pdfs));
  CApi(khg_loglikes(ctx, m.h, u.h));
  int64_t off[2] = {0, 0}, total = 0;
  CApi(khg_loglikes_layout(u.h, off, &total));
  std::vector<float> buf((size_t)std::max<int64_t>(total, 1));
  CApi(khg_loglikes_download(ctx, u.h, buf.data()));
  const int64_t tpad = (N + 31) & ~int64_t(31);
  std::vector<float> out((size_t)npdf * (size_t)N);
  for (int j = 0; j < npdf; ++j) std::memcpy(out.data() + (size_t)j * N, buf.data() + (size_t)j * tpad, sizeof(float) * (size_t)N);
  return out;
}

GpuStats GpuAccStats(int P, int D, const int32_t* gauss_off, const float* gconsts, const float* miv, const float* iv,
                     const float* feats, int64_t N, const int32_t* frame_pdf, float weight) {
  khg_ctx* ctx = DefaultCtx();
  ModelH m; TmH tm; UttsH u; AccsH a;
  CApi(khg_model_create(ctx, P, D, gauss_off, gconsts, miv, iv, &m.h));
  std::vector<int32_t> id2pdf((size_t)P + 1, 0);          // transition-id = pdf + 1
  for (int p = 0; p < P; ++p) id2pdf[(size_t)p + 1] = p;
  CApi(khg_tm_create(ctx, P, id2pdf.data(), &tm.h));
  const int64_t fo[2] = {0, N};
  CApi(khg_utts_create(ctx, nullptr, 1, D, fo, feats, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &u.h));
  std::vector<int32_t> ali((size_t)N);
  for (int64_t t = 0; t < N; ++t) ali[(size_t)t] = frame_pdf[t] + 1;
  CApi(khg_ali_upload(ctx, u.h, ali.data()));
  CApi(khg_accs_create(ctx, m.h, tm.h, &a.h));
  CApi(khg_acc_stats(ctx, m.h, tm.h, u.h, weight, a.h));
  int64_t n = 0;
  CApi(khg_accs_size(a.h, &n));
  std::vector<double> buf((size_t)n);
  CApi(khg_accs_download(ctx, a.h, buf.data()));
  const size_t sumG = (size_t)gauss_off[P];
  GpuStats st;
  st.occ.assign(buf.begin(), buf.begin() + sumG);
  st.mean_acc.assign(buf.begin() + sumG, buf.begin() + sumG + sumG * D);
  st.var_acc.assign(buf.begin() + sumG + sumG * D, buf.begin() + sumG + 2 * sumG * D);
  const size_t sc = sumG + 2 * sumG * D + (size_t)P + 1;
  st.total_frames = buf[sc];
  st.total_log_like = buf[sc + 1];
  return st;
}

// ---- flags / split targets --------------------------------------------------------------------------------------------
int StrToGmmFlags(const std::string& s) {
  int flags = 0;
  for (char c : s) {
    switch (c) {
      case 'm': flags |= 1; break;
      case 'v': flags |= 2; break;
      case 'w': flags |= 4; break;
      case 't': flags |= 8; break;
      case 'a': flags |= 15; break;
      default: throw Error(std::string("Invalid element '") + c + "' of GmmFlagsType option string " + s);
    }
  }
  return flags;
}
std::string GmmFlagsToStr(int flags) {
  std::string s;
  if (flags & 1) s += 'm';
  if (flags & 2) s += 'v';
  if (flags & 4) s += 'w';
  if (flags & 8) s += 't';
  return s;
}
int AugmentGmmFlags(int flags) {
  KHG_REQUIRE(!(flags & ~0xF), "(flags & ~kGmmAll) == 0 assertion failed");
  if (flags & 2) flags |= 1;
  if (flags & 1) flags |= 4;
  if (!(flags & 4)) flags |= 4;
  return flags;
}


namespace {
struct CountStats {      // csrc/model-common.cc:14-27; ties go to the entry pushed first
  double key; int64_t order; int32_t pdf, nc; double occ;
  bool operator<(const CountStats& o) const { return key < o.key || (key == o.key && order > o.order); }
};
}  // namespace

std::vector<int32_t> GetSplitTargets(const std::vector<float>& occs, int32_t target_components, float power, float min_count_f) {
  return GetSplitTargetsD(occs, target_components, power, (double)min_count_f);
}
std::vector<int32_t> GetSplitTargetsD(const std::vector<float>& occs, int32_t target_components, float power, double min_count) {
  const int32_t P = (int32_t)occs.size();
  std::priority_queue<CountStats> heap;
  int64_t order = 0;
  for (int32_t p = 0; p < P; ++p) {
    const double occ = (double)std::pow(occs[(size_t)p], power);      // float pow: occ^power as BaseFloat
    heap.push(CountStats{occ / (1 + 1.0e-10), order++, p, 1, occ});
  }
  int32_t num_gauss = P;
  while (num_gauss < target_components && !heap.empty()) {
    CountStats s = heap.top();
    if (s.occ == 0) break;
    heap.pop();
    if ((s.nc + 1) * min_count >= (double)occs[(size_t)s.pdf]) {
      s.occ = 0.0;
    } else {
      s.nc += 1;
      num_gauss += 1;
    }
    s.key = s.occ / (s.nc + 1.0e-10);
    s.order = order++;
    heap.push(s);
  }
  std::vector<int32_t> targets((size_t)P, 0);
  while (!heap.empty()) { targets[(size_t)heap.top().pdf] = heap.top().nc; heap.pop(); }
  return targets;
}

// ---- DiagGmm ----------------------------------------------------------------------------------------------------------
void DiagGmm::Resize(int nmix, int dim) {
  Touch();
  KHG_REQUIRE(nmix >= 0 && dim >= 0, "DiagGmm::Resize: negative size");
  G_ = nmix; D_ = dim;
  gconsts_.assign((size_t)nmix, 0.0f);
  weights_.assign((size_t)nmix, 0.0f);
  inv_vars_.assign((size_t)nmix * dim, 1.0f);
  means_invvars_.assign((size_t)nmix * dim, 0.0f);
  valid_gconsts_ = false;
}
DiagGmm::DiagGmm(const std::vector<std::pair<float, const DiagGmm*>>& gmms) {
  if (gmms.empty()) return;                    // an empty mixture
  int num_gauss = 0;
  const int dim = gmms[0].second->Dim();
  for (auto& p : gmms) num_gauss += p.second->NumGauss();
  Resize(num_gauss, dim);
  size_t cur = 0;
  for (auto& p : gmms) {
    KHG_REQUIRE(p.first > 0.0f, "weight > 0.0 assertion failed");
    const DiagGmm& g = *p.second;
    KHG_REQUIRE(g.Dim() == dim || g.NumGauss() == 0, "DiagGmm(gmms): the mixtures differ in dimension");
    const size_t n = (size_t)g.NumGauss();
    std::copy(g.means_invvars_.begin(), g.means_invvars_.end(), means_invvars_.begin() + cur * dim);
    std::copy(g.inv_vars_.begin(), g.inv_vars_.end(), inv_vars_.begin() + cur * dim);
    for (size_t i = 0; i < n; ++i) weights_[cur + i] = p.first * g.weights_[i];
    cur += n;
  }
  ComputeGconsts();
}
void DiagGmm::SetRaw(int G, int D, const float* w, const float* iv, const float* miv, const float* gc) {
  Touch();
  G_ = G; D_ = D;
  weights_.assign(w, w + G);
  inv_vars_.assign(iv, iv + (size_t)G * D);
  means_invvars_.assign(miv, miv + (size_t)G * D);
  if (gc) { gconsts_.assign(gc, gc + G); valid_gconsts_ = true; }
  else { gconsts_.assign((size_t)G, 0.0f); valid_gconsts_ = false; }
}
std::vector<float> DiagGmm::GetMeans() const {
  std::vector<float> m(means_invvars_.size());
  for (size_t i = 0; i < m.size(); ++i) m[i] = means_invvars_[i] / inv_vars_[i];
  return m;
}
std::vector<float> DiagGmm::GetVars() const {
  std::vector<float> v(inv_vars_.size());
  for (size_t i = 0; i < v.size(); ++i) v[i] = 1.0f / inv_vars_[i];
  return v;
}
void DiagGmm::SetWeights(const float* w, size_t n) {
  Touch();
  KHG_REQUIRE((int)n == G_, "weights_.size() == w.size() assertion failed");
  weights_.assign(w, w + n);
  valid_gconsts_ = false;
}
void DiagGmm::SetMeans(const float* m, size_t rows, size_t cols) {
  Touch();
  KHG_REQUIRE((int)rows == G_ && (int)cols == D_, "SetMeans: shape mismatch");
  for (size_t i = 0; i < means_invvars_.size(); ++i) means_invvars_[i] = m[i] * inv_vars_[i];
  valid_gconsts_ = false;
}
void DiagGmm::SetInvVars(const float* v, size_t rows, size_t cols) {
  Touch();
  KHG_REQUIRE((int)rows == G_ && (int)cols == D_, "SetInvVars: shape mismatch");
  for (size_t i = 0; i < inv_vars_.size(); ++i) {
    means_invvars_[i] = means_invvars_[i] / inv_vars_[i] * v[i];
    inv_vars_[i] = v[i];
  }
  valid_gconsts_ = false;
}
void DiagGmm::SetInvVarsAndMeans(const float* v, const float* m, size_t rows, size_t cols) {
  Touch();
  KHG_REQUIRE((int)rows == G_ && (int)cols == D_, "SetInvVarsAndMeans: shape mismatch");
  for (size_t i = 0; i < inv_vars_.size(); ++i) { inv_vars_[i] = v[i]; means_invvars_[i] = m[i] * v[i]; }
  valid_gconsts_ = false;
}
void DiagGmm::SetComponentWeight(int g, float w) {
  Touch();
  KHG_REQUIRE(w > 0.0f && g < G_ && g >= 0, "SetComponentWeight assertion failed");
  weights_[(size_t)g] = w;
  valid_gconsts_ = false;
}
void DiagGmm::SetComponentMean(int g, const float* v, size_t n) {
  Touch();
  KHG_REQUIRE(g >= 0 && g < G_ && (int)n == D_, "SetComponentMean: bad index or size");
  for (int d = 0; d < D_; ++d) means_invvars_[(size_t)g * D_ + d] = inv_vars_[(size_t)g * D_ + d] * v[d];
  valid_gconsts_ = false;
}
void DiagGmm::SetComponentInvVar(int g, const float* v, size_t n) {
  Touch();
  KHG_REQUIRE(g >= 0 && g < G_ && (int)n == D_, "SetComponentInvVar: bad index or size");
  for (int d = 0; d < D_; ++d) {
    const size_t k = (size_t)g * D_ + d;
    means_invvars_[k] = means_invvars_[k] / inv_vars_[k] * v[d];
    inv_vars_[k] = v[d];
  }
  valid_gconsts_ = false;
}
std::vector<float> DiagGmm::GetComponentMean(int g) const {
  KHG_REQUIRE(g >= 0 && g < G_, "gauss < NumGauss() assertion failed");
  std::vector<float> m((size_t)D_);
  for (int d = 0; d < D_; ++d) m[(size_t)d] = means_invvars_[(size_t)g * D_ + d] / inv_vars_[(size_t)g * D_ + d];
  return m;
}
std::vector<float> DiagGmm::GetComponentVariance(int g) const {
  KHG_REQUIRE(g >= 0 && g < G_, "gauss < NumGauss() assertion failed");
  std::vector<float> v((size_t)D_);
  for (int d = 0; d < D_; ++d) v[(size_t)d] = 1.0f / inv_vars_[(size_t)g * D_ + d];
  return v;
}
void DiagGmm::RemoveComponent(int g, bool renorm) {
  Touch();
  KHG_REQUIRE(g >= 0 && g < G_, "RemoveComponent: index out of range");
  KHG_REQUIRE(G_ != 1, "Attempting to remove the only remaining component.");
  weights_.erase(weights_.begin() + g);
  gconsts_.erase(gconsts_.begin() + g);
  means_invvars_.erase(means_invvars_.begin() + (size_t)g * D_, means_invvars_.begin() + (size_t)(g + 1) * D_);
  inv_vars_.erase(inv_vars_.begin() + (size_t)g * D_, inv_vars_.begin() + (size_t)(g + 1) * D_);
  --G_;
  if (renorm) {
    const float s = NpSum(weights_.data(), weights_.size());
    for (float& w : weights_) w = w / s;
    valid_gconsts_ = false;
  }
}
void DiagGmm::RemoveComponents(std::vector<int> gauss, bool renorm) {
  std::sort(gauss.begin(), gauss.end());
  KHG_REQUIRE(std::adjacent_find(gauss.begin(), gauss.end()) == gauss.end(), "IsSortedAndUniq(gauss) assertion failed");
  for (size_t i = 0; i < gauss.size(); ++i) RemoveComponent(gauss[i] - (int)i, renorm);
}
int DiagGmm::ComputeGconsts() {
  Touch();
  const int32_t go[2] = {0, G_};
  int32_t nb = 0;
  gconsts_.resize((size_t)G_);
  CApi(khg_compute_gconsts(1, D_, go, weights_.data(), inv_vars_.data(), means_invvars_.data(), gconsts_.data(), &nb));
  valid_gconsts_ = true;
  return nb;
}
float DiagGmm::LogLikelihood(const float* data, size_t n) const {
  NeedGconsts();
  KHG_REQUIRE((int)n == D_, "DiagGmm::LogLikelihoods, dimension mismatch " + std::to_string(n) + " vs. " + std::to_string(D_));
  const int32_t go[2] = {0, G_}, pdf = 0;
  return GpuLoglikes(1, D_, go, gconsts_.data(), means_invvars_.data(), inv_vars_.data(), data, 1, &pdf, 1)[0];
}
std::vector<float> DiagGmm::LogLikelihoodsMatrix(const float* data, size_t rows, size_t cols) const {
  KHG_REQUIRE(rows != 0, "data.rows() != 0 assertion failed");
  KHG_REQUIRE((int)cols == D_, "DiagGmm::LogLikelihoods, dimension mismatch " + std::to_string(cols) + " vs. " + std::to_string(D_));
  std::vector<int32_t> go((size_t)G_ + 1), pdfs((size_t)G_);      // every Gaussian as its own one-component pdf
  for (int g = 0; g <= G_; ++g) go[(size_t)g] = g;
  for (int g = 0; g < G_; ++g) pdfs[(size_t)g] = g;
  std::vector<float> gn = GpuLoglikes(G_, D_, go.data(), gconsts_.data(), means_invvars_.data(), inv_vars_.data(), data, (int64_t)rows,
                                      pdfs.data(), G_);   // [G][N]
  std::vector<float> out(rows * (size_t)G_);
  for (int g = 0; g < G_; ++g)
    for (size_t t = 0; t < rows; ++t) out[t * G_ + g] = gn[(size_t)g * rows + t];
  return out;
}
std::vector<float> DiagGmm::LogLikelihoods(const float* data, size_t n) const {
  KHG_REQUIRE((int)n == D_, "DiagGmm::LogLikelihoods, dimension mismatch " + std::to_string(n) + " vs. " + std::to_string(D_));
  return LogLikelihoodsMatrix(data, 1, n);
}
double DiagGmm::ComponentPosteriors(const float* data, size_t n, std::vector<float>* post) const {
  NeedGconsts();
  KHG_REQUIRE((int)n == D_, "data.size() == Dim() assertion failed");
  const int32_t go[2] = {0, G_}, fp = 0;
  GpuStats st = GpuAccStats(1, D_, go, gconsts_.data(), means_invvars_.data(), inv_vars_.data(), data, 1, &fp, 1.0f);
  post->resize((size_t)G_);
  for (int g = 0; g < G_; ++g) (*post)[(size_t)g] = (float)st.occ[(size_t)g];
  return st.total_log_like;
}
namespace {
// csrc/kaldi-math.h:59-78
inline float LogAddF(float x, float y) {
  static const float kMinLogDiffFloat = std::log(std::numeric_limits<float>::epsilon());
  float diff;
  if (x < y) { diff = x - y; x = y; } else { diff = y - x; }
  if (diff >= kMinLogDiffFloat) return x + std::log1p(std::exp(diff));
  return x;
}
// the selection step shared by the three forms: candidates (loglike, id), keep those >= the n-th largest, best first
float SelectBest(const std::vector<float>& loglikes, const std::vector<int32_t>& ids, int keep, std::vector<int32_t>* output) {
  const int n = (int)loglikes.size();
  float thresh = -std::numeric_limits<float>::infinity();
  if (keep < n) {
    std::vector<float> c = loglikes;
    std::nth_element(c.begin(), c.begin() + (n - keep), c.end());
    thresh = c[(size_t)(n - keep)];
  }
  std::vector<std::pair<float, int32_t>> pairs;
  for (int p = 0; p < n; ++p)
    if (loglikes[(size_t)p] >= thresh) pairs.emplace_back(loglikes[(size_t)p], ids[(size_t)p]);
  std::sort(pairs.begin(), pairs.end(), std::greater<std::pair<float, int32_t>>());
  float tot = -std::numeric_limits<float>::infinity();
  output->clear();
  for (int j = 0; j < keep && j < (int)pairs.size(); ++j) {
    output->push_back(pairs[(size_t)j].second);
    tot = LogAddF(tot, pairs[(size_t)j].first);
  }
  KHG_REQUIRE(!output->empty(), "!output->empty() assertion failed");
  return tot;
}
}  // namespace

float DiagGmm::GaussianSelection(const float* data, size_t n, int num_gselect, std::vector<int32_t>* output) const {
  const std::vector<float> ll = LogLikelihoods(data, n);
  std::vector<int32_t> ids((size_t)G_);
  for (int g = 0; g < G_; ++g) ids[(size_t)g] = g;
  return SelectBest(ll, ids, num_gselect, output);
}
float DiagGmm::GaussianSelectionMatrix(const float* data, size_t rows, size_t cols, int num_gselect, std::vector<std::vector<int32_t>>* output) const {
  KHG_REQUIRE(rows != 0, "num_frames != 0 assertion failed");
  const std::vector<float> mat = LogLikelihoodsMatrix(data, rows, cols);      // [N][G]: one K1 launch for all frames
  std::vector<int32_t> ids((size_t)G_);
  for (int g = 0; g < G_; ++g) ids[(size_t)g] = g;
  output->assign(rows, std::vector<int32_t>());
  double ans = 0.0;
  for (size_t t = 0; t < rows; ++t) {
    const std::vector<float> ll(mat.begin() + t * G_, mat.begin() + (t + 1) * G_);
    ans += SelectBest(ll, ids, num_gselect, &(*output)[t]);
  }
  return (float)ans;
}
float DiagGmm::GaussianSelectionPreselect(const float* data, size_t n, const std::vector<int32_t>& preselect, int num_gselect, std::vector<int32_t>* output) const {
  KHG_REQUIRE(!preselect.empty(), "preselect is empty");
  const std::vector<float> all = LogLikelihoods(data, n);
  std::vector<float> ll;
  for (int32_t g : preselect) {
    KHG_REQUIRE(g >= 0 && g < G_, "preselect: component index out of range");
    ll.push_back(all[(size_t)g]);
  }
  return SelectBest(ll, preselect, std::min(num_gselect, (int)preselect.size()), output);
}

void DiagGmm::Split(int target, float perturb_factor, std::vector<int>* history, const RandnFn& randn) {
  Touch();
  int cur = G_;
  KHG_REQUIRE(!(target < cur || cur == 0), "Cannot split from " + std::to_string(cur) + " to " + std::to_string(target) + " components");
  if (target == cur) return;
  const int D = D_;
  weights_.resize((size_t)target, 0.0f);
  means_invvars_.resize((size_t)target * D, 0.0f);
  inv_vars_.resize((size_t)target * D, 0.0f);
  std::vector<float> rv((size_t)D);
  while (cur < target) {
    int mx = 0;                                   // first maximum, like the strict '>' scan
    for (int g = 1; g < cur; ++g) if (weights_[(size_t)g] > weights_[(size_t)mx]) mx = g;
    if (history) history->push_back(mx);
    weights_[(size_t)mx] = weights_[(size_t)mx] / 2.0f;
    weights_[(size_t)cur] = weights_[(size_t)mx];
    randn(rv.data(), 0, (size_t)D);
    for (int d = 0; d < D; ++d) {
      const size_t a = (size_t)mx * D + d, b = (size_t)cur * D + d;
      const float r = rv[(size_t)d] * std::sqrt(inv_vars_[a]);
      const float dl = r * perturb_factor;
      inv_vars_[b] = inv_vars_[a];
      const float m0 = means_invvars_[a];
      means_invvars_[b] = m0 + dl;
      means_invvars_[a] = m0 - dl;
    }
    ++cur;
  }
  G_ = target;
  gconsts_.assign((size_t)target, 0.0f);
  ComputeGconsts();
}
std::vector<int> DiagGmm::Merge(int target) {
  Touch();
  int32_t G = G_, nh = 0;
  std::vector<float> w = weights_, miv = means_invvars_, iv = inv_vars_, gc((size_t)G_, 0.0f);
  std::vector<int32_t> hist((size_t)2 * std::max(G_, 1));
  CApi(khg_diag_gmm_merge(&G, D_, target, w.data(), gc.data(), miv.data(), iv.data(), hist.data(), &nh));
  if (G != G_) {
    G_ = G;
    weights_.assign(w.begin(), w.begin() + G);
    gconsts_.assign(gc.begin(), gc.begin() + G);
    means_invvars_.assign(miv.begin(), miv.begin() + (size_t)G * D_);
    inv_vars_.assign(iv.begin(), iv.begin() + (size_t)G * D_);
    valid_gconsts_ = true;
  }
  return std::vector<int>(hist.begin(), hist.begin() + nh);
}
void DiagGmm::Perturb(float perturb_factor, const RandnFn& randn) {
  Touch();
  std::vector<float> rv(means_invvars_.size());
  randn(rv.data(), (size_t)G_, (size_t)D_);
  for (size_t i = 0; i < rv.size(); ++i) {
    const float r = rv[i] * std::sqrt(inv_vars_[i]);
    means_invvars_[i] = means_invvars_[i] + r * perturb_factor;
  }
  ComputeGconsts();
}
std::vector<float> DiagGmm::Generate(const RandnFn& randn) const {
  const float tot = NpSum(weights_.data(), weights_.size());
  KHG_REQUIRE(tot > 0.0f, "tot > 0.0 assertion failed");
  float r1 = 0.0f;
  randn(&r1, 0, 1);
  const double r = (double)tot * (double)r1 * 0.99999;      // a NORMAL deviate, like the reference (csrc/diag-gmm.cc:419)
  int i = 0;
  double acc = 0.0;
  while (i < G_ && acc + (double)weights_[(size_t)i] < r) { acc += (double)weights_[(size_t)i]; ++i; }
  i = std::min(i, G_ - 1);
  std::vector<float> rv((size_t)D_), out((size_t)D_);
  randn(rv.data(), 0, (size_t)D_);
  for (int d = 0; d < D_; ++d) {
    const float t = inv_vars_[(size_t)i * D_ + d];
    out[(size_t)d] = means_invvars_[(size_t)i * D_ + d] / t + rv[(size_t)d] / std::sqrt(t);
  }
  return out;
}
void DiagGmm::Interpolate(float rho_f, const DiagGmm& src, int flags) {
  Touch();
  KHG_REQUIRE(G_ == src.G_ && D_ == src.D_, "NumGauss() == source.NumGauss() && Dim() == source.Dim() assertion failed");
  // DiagGmmNormal of both (double), csrc/diag-gmm-normal.cc:14-20
  const size_t n = inv_vars_.size();
  std::vector<double> w((size_t)G_), tw((size_t)G_), wv(n), wm(n), tv(n), tmn(n);
  for (int g = 0; g < G_; ++g) { w[(size_t)g] = weights_[(size_t)g]; tw[(size_t)g] = src.weights_[(size_t)g]; }
  for (size_t i = 0; i < n; ++i) {
    wv[i] = 1.0 / (double)inv_vars_[i]; wm[i] = (double)means_invvars_[i] * wv[i];
    tv[i] = 1.0 / (double)src.inv_vars_[i]; tmn[i] = (double)src.means_invvars_[i] * tv[i];
  }
  const double rho = (double)rho_f, om = 1.0 - rho;
  if (flags & 4) {
    for (int g = 0; g < G_; ++g) w[(size_t)g] = w[(size_t)g] * om + tw[(size_t)g] * rho;
    const double s = NpSum(w.data(), w.size());
    for (double& x : w) x = x / s;
  }
  if (flags & 1) for (size_t i = 0; i < n; ++i) wm[i] = wm[i] * om + tmn[i] * rho;
  if (flags & 2) for (size_t i = 0; i < n; ++i) wv[i] = wv[i] * om + tv[i] * rho;
  // CopyToDiagGmm(kGmmAll) (csrc/diag-gmm-normal.cc:22-48)
  for (int g = 0; g < G_; ++g) weights_[(size_t)g] = (float)w[(size_t)g];
  for (size_t i = 0; i < n; ++i) {
    inv_vars_[i] = (float)(1.0 / wv[i]);
    means_invvars_[i] = (float)wm[i] * inv_vars_[i];
  }
  ComputeGconsts();
}

// ---- AmDiagGmm --------------------------------------------------------------------------------------------------------
void AmDiagGmm::Init(const DiagGmm& proto, int num_pdfs) {
  struct_version_ = NextVersion();
  pdfs_.clear();
  for (int i = 0; i < num_pdfs; ++i) pdfs_.push_back(std::make_shared<DiagGmm>(proto));
}
void AmDiagGmm::AddPdf(const DiagGmm& gmm) {
  KHG_REQUIRE(pdfs_.empty() || gmm.Dim() == Dim(), "gmm.Dim() == this->Dim() assertion failed");
  struct_version_ = NextVersion();
  pdfs_.push_back(std::make_shared<DiagGmm>(gmm));
}
void AmDiagGmm::CopyFromAmDiagGmm(const AmDiagGmm& o) {
  std::vector<std::shared_ptr<DiagGmm>> n;
  for (auto& p : o.pdfs_) n.push_back(std::make_shared<DiagGmm>(*p));
  pdfs_.swap(n);
  struct_version_ = NextVersion();
}
uint64_t AmDiagGmm::Version() const {
  uint64_t v = struct_version_;
  for (auto& p : pdfs_) v = std::max(v, p->version());
  return v;
}
khg_model* AmDiagGmm::DeviceModel(khg_ctx* ctx) const {
  std::lock_guard<std::mutex> lk(dev_->mu);
  const uint64_t v = Version();
  if (dev_->h && dev_->ctx == ctx && dev_->version == v) return dev_->h;
  dev_->Release();                          // the old image's HBM first
  KHG_REQUIRE(NumPdfs() > 0, "AmDiagGmm: no pdfs");
  std::vector<int32_t> go;
  std::vector<float> gc, miv, iv;
  Flat(&go, &gc, nullptr, &miv, &iv);
  CApi(khg_model_create(ctx, NumPdfs(), Dim(), go.data(), gc.data(), miv.data(), iv.data(), &dev_->h));
  dev_->ctx = ctx; dev_->version = v;
  return dev_->h;
}
void AmDiagGmm::SplitByCount(const std::vector<float>& occs, int target, float perturb, float power, double min_count, const RandnFn& randn) {
  KHG_REQUIRE((int)occs.size() == NumPdfs(), "state_occs.size() == NumPdfs() assertion failed");
  const std::vector<int32_t> targets = GetSplitTargetsD(occs, target, power, min_count);
  for (int i = 0; i < NumPdfs(); ++i)
    if (pdfs_[(size_t)i]->NumGauss() < targets[(size_t)i]) pdfs_[(size_t)i]->Split(targets[(size_t)i], perturb, nullptr, randn);
}
void AmDiagGmm::MergeByCount(const std::vector<float>& occs, int target, float power, double min_count) {
  KHG_REQUIRE((int)occs.size() == NumPdfs(), "state_occs.size() == NumPdfs() assertion failed");
  const std::vector<int32_t> targets = GetSplitTargetsD(occs, target, power, min_count);
  for (int i = 0; i < NumPdfs(); ++i) {
    const int t = targets[(size_t)i] == 0 ? 1 : targets[(size_t)i];      // can't merge below 1
    if (pdfs_[(size_t)i]->NumGauss() > t) pdfs_[(size_t)i]->Merge(t);
  }
}
void AmDiagGmm::Flat(std::vector<int32_t>* go, std::vector<float>* gc, std::vector<float>* w, std::vector<float>* miv, std::vector<float>* iv) const {
  go->assign(1, 0);
  if (gc) gc->clear();
  if (w) w->clear();
  if (miv) miv->clear();
  if (iv) iv->clear();
  for (auto& p : pdfs_) {
    if (gc) { p->NeedGconsts(); gc->insert(gc->end(), p->gconsts().begin(), p->gconsts().end()); }
    go->push_back(go->back() + p->NumGauss());
    if (w) w->insert(w->end(), p->weights().begin(), p->weights().end());
    if (miv) miv->insert(miv->end(), p->means_invvars().begin(), p->means_invvars().end());
    if (iv) iv->insert(iv->end(), p->inv_vars().begin(), p->inv_vars().end());
  }
}
void AmDiagGmm::SetFlat(const int32_t* go, const float* w, const float* gc, const float* miv, const float* iv) {
  const int D = Dim();
  for (size_t i = 0; i < pdfs_.size(); ++i) {
    const int a = go[i], b = go[i + 1];
    pdfs_[i]->SetRaw(b - a, D, w + a, iv + (size_t)a * D, miv + (size_t)a * D, gc + a);
  }
}

// ---- options ----------------------------------------------------------------------------------------------------------
std::string MleDiagGmmOptions::ToString() const {
  char buf[256];
  std::snprintf(buf, sizeof(buf), "MleDiagGmmOptions(min_gaussian_weight=%g, min_gaussian_occupancy=%g, min_variance=%g, remove_low_count_gaussians=%s)",
                (double)min_gaussian_weight, (double)min_gaussian_occupancy, min_variance, remove_low_count_gaussians ? "True" : "False");
  return buf;
}

// ---- AccumDiagGmm -----------------------------------------------------------------------------------------------------
void AccumDiagGmm::Resize(int num_gauss, int dim, int flags) {
  KHG_REQUIRE(num_gauss > 0 && dim > 0, "num_comp > 0 && dim > 0 assertion failed");
  flags_ = AugmentGmmFlags(flags);
  G_ = num_gauss; D_ = dim;
  occ_.assign((size_t)G_, 0.0);
  if (flags_ & 1) mean_.assign((size_t)G_ * D_, 0.0); else mean_.clear();
  if (flags_ & 2) var_.assign((size_t)G_ * D_, 0.0); else var_.clear();
}
void AccumDiagGmm::SetZero(int flags) {
  CheckFlags(flags);
  if (flags & 4) std::fill(occ_.begin(), occ_.end(), 0.0);
  if (flags & 1) std::fill(mean_.begin(), mean_.end(), 0.0);
  if (flags & 2) std::fill(var_.begin(), var_.end(), 0.0);
}
void AccumDiagGmm::Scale(float f, int flags) {
  CheckFlags(flags);
  const double d = (double)f;
  if (flags & 4) for (double& x : occ_) x *= d;
  if (flags & 1) for (double& x : mean_) x *= d;
  if (flags & 2) for (double& x : var_) x *= d;
}
void AccumDiagGmm::AccumulateForComponent(const float* data, size_t n, int comp, float weight) {
  KHG_REQUIRE(!((flags_ & 1) && (int)n != D_), "data.size() == Dim() assertion failed");
  KHG_REQUIRE(comp < G_ && comp >= 0, "comp_index < NumGauss() assertion failed");
  const double wt = (double)weight;
  occ_[(size_t)comp] += wt;
  if (flags_ & 1) {
    for (int d = 0; d < D_; ++d) mean_[(size_t)comp * D_ + d] += (double)data[d] * wt;
    if (flags_ & 2)
      for (int d = 0; d < D_; ++d) var_[(size_t)comp * D_ + d] += (double)((data[d] * data[d]) * weight);
  }
}
void AccumDiagGmm::AccumulateFromPosteriors(const float* data, size_t n, const float* post, size_t np) {
  KHG_REQUIRE(!((flags_ & 1) && (int)n != D_), "data.size() == Dim() assertion failed");
  KHG_REQUIRE((int)np == G_, "posteriors.size() == NumGauss() assertion failed");
  for (int g = 0; g < G_; ++g) occ_[(size_t)g] += (double)post[g];
  if (flags_ & 1) {
    for (int g = 0; g < G_; ++g)
      for (int d = 0; d < D_; ++d) mean_[(size_t)g * D_ + d] += (double)(post[g] * data[d]);       // fp32 product, then widened (:135)
    if (flags_ & 2)
      for (int g = 0; g < G_; ++g)
        for (int d = 0; d < D_; ++d) var_[(size_t)g * D_ + d] += (double)(post[g] * (data[d] * data[d]));   // :138-140
  }
}
float AccumDiagGmm::AccumulateFromDiag(const DiagGmm& gmm, const float* data, size_t n, float weight) {
  KHG_REQUIRE(gmm.NumGauss() == G_ && gmm.Dim() == D_, "gmm.NumGauss() == NumGauss() assertion failed");
  KHG_REQUIRE((int)n == D_, "data.size() == Dim() assertion failed");
  gmm.NeedGconsts();
  const int32_t go[2] = {0, G_}, fp = 0;
  GpuStats st = GpuAccStats(1, D_, go, gmm.gconsts().data(), gmm.means_invvars().data(), gmm.inv_vars().data(), data, 1, &fp, weight);
  AddRaw(st.occ.data(), st.mean_acc.data(), st.var_acc.data());
  return weight != 0.0f ? (float)(st.total_log_like / (double)weight) : 0.0f;
}
void AccumDiagGmm::AddRaw(const double* occ, const double* mean, const double* var) {
  for (int g = 0; g < G_; ++g) occ_[(size_t)g] += occ[g];
  if (flags_ & 1) for (size_t i = 0; i < mean_.size(); ++i) mean_[i] += mean[i];
  if (flags_ & 2) for (size_t i = 0; i < var_.size(); ++i) var_[i] += var[i];
}
void AccumDiagGmm::AddStatsForComponent(int g, double occ, const double* x, size_t nx, const double* x2, size_t nx2) {
  KHG_REQUIRE(g < G_ && g >= 0, "g < NumGauss() assertion failed");
  occ_[(size_t)g] += occ;
  if (flags_ & 1) { KHG_REQUIRE((int)nx == D_, "x_stats.size() == Dim() assertion failed"); for (int d = 0; d < D_; ++d) mean_[(size_t)g * D_ + d] += x[d]; }
  if (flags_ & 2) { KHG_REQUIRE((int)nx2 == D_, "x2_stats.size() == Dim() assertion failed"); for (int d = 0; d < D_; ++d) var_[(size_t)g * D_ + d] += x2[d]; }
}
void AccumDiagGmm::Add(float scale, const AccumDiagGmm& acc) {
  KHG_REQUIRE(acc.G_ == G_ && acc.D_ == D_, "num_comp_ == acc.num_comp_ && dim_ == acc.dim_ assertion failed");
  const double s = (double)scale;
  for (int g = 0; g < G_; ++g) occ_[(size_t)g] += acc.occ_[(size_t)g] * s;
  if (flags_ & 1) { KHG_REQUIRE(acc.mean_.size() == mean_.size(), "accumulator flags mismatch"); for (size_t i = 0; i < mean_.size(); ++i) mean_[i] += acc.mean_[i] * s; }
  if (flags_ & 2) { KHG_REQUIRE(acc.var_.size() == var_.size(), "accumulator flags mismatch"); for (size_t i = 0; i < var_.size(); ++i) var_[i] += acc.var_[i] * s; }
}
void AccumDiagGmm::SmoothStats(float tau_f) {
  const double tau = (double)tau_f;
  for (int g = 0; g < G_; ++g) {
    const double sv = (occ_[(size_t)g] + tau) / occ_[(size_t)g];
    if (!mean_.empty()) for (int d = 0; d < D_; ++d) mean_[(size_t)g * D_ + d] *= sv;
    if (!var_.empty()) for (int d = 0; d < D_; ++d) var_[(size_t)g * D_ + d] *= sv;
    occ_[(size_t)g] = occ_[(size_t)g] + tau;
  }
}
void AccumDiagGmm::SmoothWithAccum(float tau_f, const AccumDiagGmm& src) {
  KHG_REQUIRE(src.G_ == G_ && src.D_ == D_, "src_acc.NumGauss() == num_comp_ && src_acc.Dim() == dim_ assertion failed");
  const double tau = (double)tau_f;
  for (int i = 0; i < G_; ++i) {
    const double so = src.occ_[(size_t)i];
    if (so != 0.0) {     // can only smooth where the source saw data (the reference warns otherwise)
      occ_[(size_t)i] += tau;
      if (!mean_.empty() && !src.mean_.empty()) for (int d = 0; d < D_; ++d) mean_[(size_t)i * D_ + d] += src.mean_[(size_t)i * D_ + d] * tau / so;
      if (!var_.empty() && !src.var_.empty()) for (int d = 0; d < D_; ++d) var_[(size_t)i * D_ + d] += src.var_[(size_t)i * D_ + d] * tau / so;
    }
  }
}
void AccumDiagGmm::SmoothWithModel(float tau_f, const DiagGmm& gmm) {
  KHG_REQUIRE(gmm.NumGauss() == G_ && gmm.Dim() == D_, "gmm.NumGauss() == num_comp_ && gmm.Dim() == dim_ assertion failed");
  const double tau = (double)tau_f;
  const std::vector<float> means = gmm.GetMeans(), vars = gmm.GetVars();
  for (size_t i = 0; i < means.size(); ++i) {
    const double m = (double)means[i], v = (double)vars[i];
    if (!mean_.empty()) mean_[i] += m * tau;
    if (!var_.empty()) var_[i] += (v + m * m) * tau;
  }
  for (double& o : occ_) o = o + tau;
}

// ---- M-step entry points ----------------------------------------------------------------------------------------------
namespace {
MleUpdateResult FlatUpdate(const MleDiagGmmOptions& cfg, int P, int D, const int32_t* go, const double* occ, const double* ma, const double* va,
                           int acc_flags, int flags, std::vector<float>* w, std::vector<float>* gc, std::vector<float>* miv, std::vector<float>* iv,
                           std::vector<int32_t>* new_off) {
  const khg_mle_options o = cfg.C();
  new_off->assign((size_t)P + 1, 0);
  gc->assign(w->size(), 0.0f);
  MleUpdateResult r;
  CApi(khg_mle_am_diag_gmm_update(&o, P, D, go, occ, ma, va, (uint16_t)acc_flags, (uint16_t)flags, w->data(), gc->data(), miv->data(), iv->data(),
                                  new_off->data(), &r.objf_change, &r.count, &r.floored_elements, &r.floored_gaussians, &r.removed));
  return r;
}
}  // namespace

MleUpdateResult MleFlatUpdate(const MleDiagGmmOptions& cfg, int P, int D, const int32_t* go, const double* occ, const double* ma, const double* va,
                              int acc_flags, int flags, std::vector<float>* w, std::vector<float>* gc, std::vector<float>* miv, std::vector<float>* iv,
                              std::vector<int32_t>* new_off) {
  return FlatUpdate(cfg, P, D, go, occ, ma, va, acc_flags, flags, w, gc, miv, iv, new_off);
}

MleUpdateResult MleDiagGmmUpdate(const MleDiagGmmOptions& cfg, const AccumDiagGmm& acc, int flags, DiagGmm* gmm) {
  KHG_REQUIRE(gmm->NumGauss() == acc.NumGauss() && gmm->Dim() == acc.Dim(), "diag_gmm_acc.NumGauss() == gmm->NumGauss() assertion failed");
  const int32_t go[2] = {0, gmm->NumGauss()};
  std::vector<float> w = gmm->weights(), miv = gmm->means_invvars(), iv = gmm->inv_vars(), gc;
  std::vector<int32_t> new_off;
  MleUpdateResult r = FlatUpdate(cfg, 1, gmm->Dim(), go, acc.occupancy().data(), acc.mean_accumulator().empty() ? nullptr : acc.mean_accumulator().data(),
                                 acc.variance_accumulator().empty() ? nullptr : acc.variance_accumulator().data(), acc.Flags(), flags, &w, &gc, &miv,
                                 &iv, &new_off);
  gmm->SetRaw(new_off[1], gmm->Dim(), w.data(), iv.data(), miv.data(), gc.data());
  return r;
}

float MlObjective(const DiagGmm& gmm, const AccumDiagGmm& acc) {
  double dot = 0.0;
  for (int g = 0; g < gmm.NumGauss(); ++g) dot += acc.occupancy()[(size_t)g] * (double)gmm.gconsts()[(size_t)g];
  float obj = (float)dot;
  const size_t n = gmm.inv_vars().size();
  if (acc.Flags() & 1) {
    std::vector<double> t(n);
    for (size_t i = 0; i < n; ++i) t[i] = acc.mean_accumulator()[i] * (double)gmm.means_invvars()[i];
    obj = (float)((double)obj + NpSum(t.data(), n));
  }
  if (acc.Flags() & 2) {
    std::vector<double> t(n);
    for (size_t i = 0; i < n; ++i) t[i] = acc.variance_accumulator()[i] * (double)gmm.inv_vars()[i];
    obj = (float)((double)obj - 0.5 * NpSum(t.data(), n));
  }
  return obj;
}

std::string MapDiagGmmOptions::ToString() const {
  char buf[160];
  std::snprintf(buf, sizeof(buf), "MapDiagGmmOptions(mean_tau=%g, variance_tau=%g, weight_tau=%g)", (double)mean_tau, (double)variance_tau, (double)weight_tau);
  return buf;
}

std::pair<float, float> MapDiagGmmUpdate(const MapDiagGmmOptions& cfg, const AccumDiagGmm& acc, int flags, DiagGmm* gmm) {
  KHG_REQUIRE(gmm != nullptr, "gmm != NULL assertion failed");
  KHG_REQUIRE(!(flags & ~acc.Flags()), "Flags in argument do not match the active accumulators");
  KHG_REQUIRE(acc.NumGauss() == gmm->NumGauss() && acc.Dim() == gmm->Dim(), "diag_gmm_acc.NumGauss() == gmm->NumGauss() && diag_gmm_acc.Dim() == gmm->Dim() assertion failed");
  const int G = gmm->NumGauss(), D = gmm->Dim();
  double occ_sum = 0.0;
  for (double o : acc.occupancy()) occ_sum += o;
  gmm->ComputeGconsts();
  const float obj_old = MlObjective(*gmm, acc);
  // DiagGmmNormal (csrc/diag-gmm-normal.cc:14-20): everything in double
  std::vector<double> w((size_t)G), vars((size_t)G * D), means((size_t)G * D);
  for (int g = 0; g < G; ++g) w[(size_t)g] = (double)gmm->weights()[(size_t)g];
  for (size_t i = 0; i < vars.size(); ++i) { vars[i] = 1.0 / (double)gmm->inv_vars()[i]; means[i] = (double)gmm->means_invvars()[i] * vars[i]; }
  const std::vector<double> old_means = means;          // CopyToDiagGmm's `oldg`
  const double mean_tau = cfg.mean_tau, var_tau = cfg.variance_tau, w_tau = cfg.weight_tau;
  for (int i = 0; i < G; ++i) {
    const double occ = acc.occupancy()[(size_t)i];
    w[(size_t)i] = (occ + w[(size_t)i] * w_tau) / (occ_sum + w_tau);      // the weight tau is a tau for the whole state
    double* mu = &means[(size_t)i * D];
    double* var = &vars[(size_t)i * D];
    if (occ > 0.0 && (flags & kGmmMeans)) {
      const double a = 1.0 / (occ + mean_tau), b = mean_tau / (occ + mean_tau);
      for (int d = 0; d < D; ++d) { double m = acc.mean_accumulator()[(size_t)i * D + d] * a; m += mu[d] * b; mu[d] = m; }
    }
    if (occ > 0.0 && (flags & kGmmVariances)) {
      // E((x - mu)^2) = E(x^2) + mu^2 - 2 mu E(x) around the (updated) mean, then the tau weighting
      const double c = -2.0 / occ, s1 = occ / (var_tau + occ), s2 = var_tau / (var_tau + occ);
      for (int d = 0; d < D; ++d) {
        double v = acc.variance_accumulator()[(size_t)i * D + d] / occ;
        v = v + mu[d] * mu[d];
        v = v + acc.mean_accumulator()[(size_t)i * D + d] * mu[d] * c;
        v *= s1;
        v += var[d] * s2;
        var[d] = v;
      }
    }
  }
  // DiagGmmNormal::CopyToDiagGmm(gmm, flags) (csrc/diag-gmm-normal.cc:22-48)
  if (flags & kGmmWeights) for (int g = 0; g < G; ++g) gmm->mutable_weights()[(size_t)g] = (float)w[(size_t)g];
  if (flags & kGmmVariances) {
    for (size_t i = 0; i < vars.size(); ++i) gmm->mutable_inv_vars()[i] = (float)(1.0 / vars[i]);
    if (!(flags & kGmmMeans)) for (size_t i = 0; i < vars.size(); ++i) gmm->mutable_means_invvars()[i] = (float)old_means[i] * gmm->inv_vars()[i];
  }
  if (flags & kGmmMeans) for (size_t i = 0; i < means.size(); ++i) gmm->mutable_means_invvars()[i] = (float)means[i] * gmm->inv_vars()[i];
  gmm->ComputeGconsts();
  const float obj_new = MlObjective(*gmm, acc);
  return {obj_new - obj_old, (float)occ_sum};
}

std::pair<float, float> MapAmDiagGmmUpdate(const MapDiagGmmOptions& cfg, const AccumAmDiagGmm& acc, int flags, AmDiagGmm* am) {
  KHG_REQUIRE(am != nullptr && acc.Dim() == am->Dim() && acc.NumAccs() == am->NumPdfs(),
              "am_gmm != nullptr && am_diag_gmm_acc.Dim() == am_gmm->Dim() && am_diag_gmm_acc.NumAccs() == am_gmm->NumPdfs() assertion failed");
  float obj = 0.0f, count = 0.0f;
  for (int i = 0; i < acc.NumAccs(); ++i) {
    const std::pair<float, float> r = MapDiagGmmUpdate(cfg, *acc.Acc(i), flags, am->GetPdf(i).get());
    obj += r.first;
    count += r.second;
  }
  return {obj, count};
}

// ---- AccumAmDiagGmm ---------------------------------------------------------------------------------------------------
AccumAmDiagGmm& AccumAmDiagGmm::operator=(const AccumAmDiagGmm& o) {
  if (this == &o) return *this;
  o.Flush();
  DropDevice();
  accs_.clear();
  for (auto& a : o.accs_) accs_.push_back(std::make_shared<AccumDiagGmm>(*a));
  total_frames_ = o.total_frames_; total_log_like_ = o.total_log_like_;
  return *this;
}
void AccumAmDiagGmm::Init(const AmDiagGmm& model, int dim, int flags) {
  DropDevice();
  accs_.clear();
  for (int i = 0; i < model.NumPdfs(); ++i) {
    auto a = std::make_shared<AccumDiagGmm>();
    a->Resize(model.GetPdf(i)->NumGauss(), dim > 0 ? dim : model.GetPdf(i)->Dim(), flags);
    accs_.push_back(a);
  }
}
float AccumAmDiagGmm::TotStatsCount() const {
  Flush();
  double s = 0.0;
  for (auto& a : accs_) s += NpSum(a->occupancy().data(), a->occupancy().size());
  return (float)s;
}
float AccumAmDiagGmm::AccumulateForGmmTwoFeats(const AmDiagGmm& model, const float* d1, size_t n1, const float* d2, size_t n2, int i, float weight) {
  Chk(i);
  std::vector<float> post;
  const double ll = model.GetPdf(i)->ComponentPosteriors(d1, n1, &post);
  for (float& p : post) p = p * weight;
  accs_[(size_t)i]->AccumulateFromPosteriors(d2, n2, post.data(), post.size());
  total_log_like_ += (double)((float)ll * weight);
  total_frames_ += (double)weight;
  return (float)ll;
}
void AccumAmDiagGmm::AccumulateFromPosteriors(const AmDiagGmm&, const float* data, size_t n, int i, const float* post, size_t np) {
  Chk(i);
  accs_[(size_t)i]->AccumulateFromPosteriors(data, n, post, np);
  total_frames_ += (double)NpSum(post, np);
}
void AccumAmDiagGmm::AccumulateForGaussian(const AmDiagGmm& am, const float* data, size_t n, int i, int gauss, float weight) {
  Chk(i);
  KHG_REQUIRE(gauss >= 0 && gauss < am.GetPdf(i)->NumGauss(), "gauss_index out of range");
  accs_[(size_t)i]->AccumulateForComponent(data, n, gauss, weight);
}
void AccumAmDiagGmm::Add(float scale, const AccumAmDiagGmm& other) {
  KHG_REQUIRE(NumAccs() == other.NumAccs(), "num_accs == other.NumAccs() assertion failed");
  Flush(); other.Flush();
  const double s = (double)scale;
  total_frames_ += s * other.total_frames_;
  total_log_like_ += s * other.total_log_like_;
  for (size_t i = 0; i < accs_.size(); ++i) accs_[i]->Add(scale, *other.accs_[i]);
}
void AccumAmDiagGmm::Scale(float scale) {
  Flush();
  for (auto& a : accs_) a->Scale(scale, a->Flags());
  total_frames_ *= (double)scale;
  total_log_like_ *= (double)scale;
}
void AccumAmDiagGmm::AddDeviceStats(const int32_t* go, const double* occ, const double* mean, const double* var, int D, double total_frames,
                                    double total_log_like) {
  for (size_t i = 0; i < accs_.size(); ++i) {
    const size_t lo = (size_t)go[i];
    KHG_REQUIRE(go[i + 1] - go[i] == accs_[i]->NumGauss() && accs_[i]->Dim() == D, "device statistics do not match the accumulators' layout");
    accs_[i]->AddRaw(occ + lo, mean + lo * D, var + lo * D);
  }
  total_frames_ += total_frames;
  total_log_like_ += total_log_like;
}

// pending device sums -> the host accumulators: one download of the block, added pdf by pdf; the device block is zero afterwards
void AccumAmDiagGmm::Flush() const {
  if (!dev_ || !dev_->pending) return;
  Dev& d = *dev_;
  int64_t n = 0;
  CApi(khg_accs_size(d.h, &n));
  std::vector<double> buf((size_t)n);
  // (the block is plain device memory: if its context was closed meanwhile -- khg_ctx_destroy waited for its streams -- the sums are
  //  still there and the current default context's stream can fetch them)
  //  A deferred kernel error (KHG_E_RUNTIME from the device error word: an invalid answer, a pdf-id out of range) is NOT retried --
  //  the read clears the word, a second attempt would succeed and add garbage into the host accumulators.
  khg_ctx* c = d.ctx;
  int rc = khg_accs_download(c, d.h, buf.data());
  if (rc == KHG_E_ARG) { c = DefaultCtx(); rc = khg_accs_download(c, d.h, buf.data()); }    // KHG_E_ARG: the context is gone
  CApi(rc);
  CApi(khg_accs_zero(c, d.h));
  const size_t sumG = (size_t)d.gauss_off.back(), D = (size_t)d.D;
  const size_t sc = sumG + 2 * sumG * D + (size_t)d.num_tids + 1;
  d.pending = false; d.seen_frames = 0.0; d.seen_ll = 0.0;
  const_cast<AccumAmDiagGmm*>(this)->AddDeviceStats(d.gauss_off.data(), buf.data(), buf.data() + sumG, buf.data() + sumG + sumG * D, (int)D, buf[sc], buf[sc + 1]);
}

MleUpdateResult MleAmDiagGmmUpdate(const MleDiagGmmOptions& cfg, const AccumAmDiagGmm& acc, int flags, AmDiagGmm* am) {
  acc.Flush();
  KHG_REQUIRE(acc.NumAccs() == am->NumPdfs(), "am_diag_gmm_acc.NumAccs() == am_gmm->NumPdfs() assertion failed");
  KHG_REQUIRE(acc.Dim() == am->Dim(), "accumulator / model dimension mismatch (ResizeModel path is not supported)");
  KHG_REQUIRE(acc.NumAccs() > 0, "am_diag_gmm_acc.NumAccs() > 0 assertion failed");
  const int acc_flags = acc.Acc(0)->Flags(), D = am->Dim();
  std::vector<int32_t> go, new_off;
  std::vector<float> w, miv, iv, gc;
  am->Flat(&go, nullptr, &w, &miv, &iv);
  std::vector<double> occ, ma, va;
  for (int i = 0; i < acc.NumAccs(); ++i) {
    const AccumDiagGmm& a = *acc.Acc(i);
    KHG_REQUIRE(a.NumGauss() == go[(size_t)i + 1] - go[(size_t)i], "diag_gmm_acc.NumGauss() == gmm->NumGauss() assertion failed");
    occ.insert(occ.end(), a.occupancy().begin(), a.occupancy().end());
    if (acc_flags & 1) ma.insert(ma.end(), a.mean_accumulator().begin(), a.mean_accumulator().end());
    if (acc_flags & 2) va.insert(va.end(), a.variance_accumulator().begin(), a.variance_accumulator().end());
  }
  MleUpdateResult r = FlatUpdate(cfg, am->NumPdfs(), D, go.data(), occ.data(), (acc_flags & 1) ? ma.data() : nullptr, (acc_flags & 2) ? va.data() : nullptr,
                                 acc_flags, flags, &w, &gc, &miv, &iv, &new_off);
  am->SetFlat(new_off.data(), w.data(), gc.data(), miv.data(), iv.data());
  return r;
}

}  // namespace khg
