// A plain C++17 caller of the host classes (kaldi_hmm_gmm_amd/csrc/khg_host_{gmm,hmm,align,fst}.{hpp,cpp}): no Python, no pybind11,
// no torch -- what a C++ integrator of the reference links next to libkhg_hip.so.  `--no-gpu`: only the host-side entry points
// (topology, transition model, gconsts, M-step, MAP, graph helpers).  Without it: one utterance through DiagGmm::LogLikelihood,
// AccumAmDiagGmm::AccumulateForGmm and the batched AlignUtteranceWrapper on the GPU.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../kaldi_hmm_gmm_amd/csrc/khg_host_fst.hpp"

using namespace khg;

#define CHECK(c) do { if (!(c)) { std::printf("HOST_CLIENT_FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static const char* kTopo =
    "<Topology> <TopologyEntry> <ForPhones> 1 2 </ForPhones> "
    "<State> 0 <PdfClass> 0 <Transition> 0 0.75 <Transition> 1 0.25 </State> "
    "<State> 1 <PdfClass> 1 <Transition> 1 0.75 <Transition> 2 0.25 </State> "
    "<State> 2 </State> </TopologyEntry> </Topology>";

int main(int argc, char** argv) {
  const bool no_gpu = argc > 1 && !std::strcmp(argv[1], "--no-gpu");
  try {
    // topology + transition model over a monophone tree's pdf_info (pdf -> [(phone, pdf_class)])
    auto topo = std::make_shared<HmmTopology>();
    topo->Read(kTopo);
    CHECK(topo->phones().size() == 2 && topo->IsHmm() && topo->NumPdfClasses(1) == 2 && topo->MinLength(2) == 2);
    std::vector<std::vector<std::pair<int, int>>> pdf_info = {{{1, 0}}, {{1, 1}}, {{2, 0}}, {{2, 1}}};
    TransitionModel tm(pdf_info, topo);
    CHECK(tm.NumPdfs() == 4 && tm.NumTransitionStates() == 4 && tm.NumTransitionIds() == 8);
    CHECK(tm.IsSelfLoop(1) && !tm.IsSelfLoop(2) && tm.TransitionIdToPdf(3) == 1 && tm.TransitionIdToPhone(5) == 2);
    CHECK(std::fabs(tm.GetTransitionLogProb(1) - std::log(0.75f)) < 1e-6f);
    std::vector<double> st((size_t)tm.NumTransitionIds() + 1, 0.0);
    for (int t = 1; t <= tm.NumTransitionIds(); ++t) st[(size_t)t] = tm.IsSelfLoop(t) ? 90.0 : 10.0;
    auto r = tm.MleUpdate(st.data(), st.size(), MleTransitionUpdateConfig());
    CHECK(r.second == 400.0f && std::fabs(tm.GetTransitionLogProb(1) - std::log(0.9f)) < 1e-6f);
    const std::vector<float> cost = tm.ScaledTransCost(1.0f, 0.1f);
    CHECK(cost.size() == 9 && cost[0] == 0.0f && cost[1] > 0.0f);

    // a model of 4 pdfs x 3 Gaussians x 5 dims
    std::mt19937 rng(7);
    std::normal_distribution<float> nd;
    const int G = 3, D = 5;
    AmDiagGmm am;
    for (int p = 0; p < 4; ++p) {
      DiagGmm g(G, D);
      std::vector<float> w = {0.2f, 0.3f, 0.5f}, mean((size_t)G * D), iv((size_t)G * D, 1.0f);
      for (auto& x : mean) x = 3.0f * p + nd(rng);
      g.SetWeights(w.data(), w.size());
      g.SetInvVarsAndMeans(iv.data(), mean.data(), G, D);
      CHECK(g.ComputeGconsts() == 0);
      am.AddPdf(g);
    }
    CHECK(am.NumPdfs() == 4 && am.NumGauss() == 12 && am.Dim() == D);
    // host M-step and MAP from hand-made statistics
    AccumAmDiagGmm accs;
    accs.Init(am, -1, kGmmAll);
    for (int p = 0; p < 4; ++p) {
      AccumDiagGmm& a = *accs.accs()[(size_t)p];
      const std::vector<float> mu = am.GetPdf(p)->GetMeans();
      for (int g = 0; g < G; ++g) {
        a.occupancy()[(size_t)g] = 50.0;
        for (int d = 0; d < D; ++d) {
          const double m = mu[(size_t)g * D + d] + 0.1;
          a.mean_accumulator()[(size_t)g * D + d] = 50.0 * m;
          a.variance_accumulator()[(size_t)g * D + d] = 50.0 * (m * m + 1.2);
        }
      }
    }
    AmDiagGmm am_map;
    am_map.CopyFromAmDiagGmm(am);
    MleUpdateResult ur = MleAmDiagGmmUpdate(MleDiagGmmOptions(), accs, kGmmAll & 7, &am);
    CHECK(ur.count == 600.0f && ur.objf_change > 0.0f && ur.removed == 0);
    CHECK(std::fabs(am.GetPdf(0)->GetVars()[0] - 1.2f) < 1e-3f);
    auto mr = MapAmDiagGmmUpdate(MapDiagGmmOptions(), accs, kGmmAll & 7, &am_map);
    CHECK(mr.second == 600.0f && mr.first > 0.0f);
    // the graph container and its helpers
    StdVectorFst f;
    const int s0 = f.AddState(), s1 = f.AddState(), s2 = f.AddState();
    f.SetStart(s0);
    f.AddArc(s0, StdArc{2, 7, 0.0f, s1}); f.AddArc(s1, StdArc{1, 0, 0.0f, s1}); f.AddArc(s1, StdArc{4, 0, 0.0f, s2}); f.AddArc(s2, StdArc{3, 0, 0.0f, s2});
    f.SetFinal(s2, 0.5f);
    AddTransitionProbs(tm, {}, 1.0f, 0.1f, &f);
    CHECK(f.Arcs(s0)[0].weight == cost[2] && f.Arcs(s1)[0].weight == cost[1]);
    StdVectorFst careful = f;
    ModifyGraphForCarefulAlignment(&careful);
    CHECK(careful.NumStates() == 7 && !careful.IsFinal(s2) && careful.IsFinal(6));
    const GraphsCsr csr = ConcatGraphs({&f, &careful});
    CHECK(csr.state_off.size() == 3 && csr.state_off[2] == 10 && csr.arc_off.back() == (int64_t)(f.NumArcs() + careful.NumArcs()));
    if (no_gpu) { std::printf("HOST_CLIENT_OK (host entry points)\n"); return 0; }

    // ---- on the GPU: scores, statistics, alignment of one utterance through the same classes ----
    const int T = 40;
    std::vector<float> feats((size_t)T * D);
    const std::vector<float> m0 = am.GetPdf(0)->GetMeans(), m1 = am.GetPdf(1)->GetMeans();
    for (int t = 0; t < T; ++t)
      for (int d = 0; d < D; ++d) feats[(size_t)t * D + d] = (t < T / 2 ? m0[d] : m1[d]) + 0.3f * nd(rng);
    const float ll = am.GetPdf(0)->LogLikelihood(feats.data(), D);
    CHECK(std::isfinite(ll));
    AccumAmDiagGmm acc2;
    acc2.Init(am, -1, kGmmAll);
    const float ll2 = acc2.AccumulateForGmm(am, feats.data(), D, 0, 1.0f);
    CHECK(std::fabs(ll2 - ll) < 1e-3f && std::fabs(acc2.TotStatsCount() - 1.0f) < 1e-5f);
    AlignConfig cfg;
    const std::vector<AlignResult> res = AlignBatch(am, tm, ConcatGraphs({&f}), {feats.data()}, {T}, cfg, 0.1f, nullptr, nullptr, false);
    CHECK(res.size() == 1 && res[0].ok && (int)res[0].alignment.size() == T && res[0].words.size() == 1 && res[0].words[0] == 7);
    CHECK(res[0].alignment.front() == 2 && res[0].alignment.back() == 3);      // enters phone 1's first state, ends in its second state's loop
    std::printf("HOST_CLIENT_OK like %.4f\n", res[0].like);
    return 0;
  } catch (const std::exception& e) {
    std::printf("HOST_CLIENT_FAIL exception: %s\n", e.what());
    return 1;
  }
}
