// pybind11 bindings of the alignment API (khg_host_align.hpp, khg_host_fst.hpp) with the names of python/csrc/{decoder-wrappers,
// faster-decoder,decodable-am-diag-gmm,decodable-itf,hmm-utils}.cc in /root/reference/kaldi-hmm-gmm, and of the graph container with
// the kaldifst method names the reference's scripts use (StdVectorFst, StdArc).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "khg_host_fst.hpp"

namespace py = pybind11;
using namespace khg;

namespace {
template <class T>
using Arr = py::array_t<T, py::array::c_style | py::array::forcecast>;

template <class T>
Arr<T> Vec1(const std::vector<T>& v) {
  Arr<T> a({(py::ssize_t)v.size()});
  if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), sizeof(T) * v.size());
  return a;
}
py::dict CsrToDict(const GraphsCsr& c) {
  py::dict d;
  d["state_off"] = Vec1(c.state_off); d["arc_off"] = Vec1(c.arc_off); d["start"] = Vec1(c.start);
  d["ilabel"] = Vec1(c.ilabel); d["olabel"] = Vec1(c.olabel); d["nextstate"] = Vec1(c.nextstate);
  d["weight"] = Vec1(c.weight); d["final"] = Vec1(c.final_w);
  return d;
}

// python/csrc/decodable-itf.cc:14-53: a decodable written in Python overrides these four
class PyDecodableInterface : public DecodableInterface {
 public:
  using DecodableInterface::DecodableInterface;
  float LogLikelihood(int frame, int index) const override { PYBIND11_OVERRIDE_PURE_NAME(float, DecodableInterface, "log_likelihood", LogLikelihood, frame, index); }
  bool IsLastFrame(int frame) const override { PYBIND11_OVERRIDE_PURE_NAME(bool, DecodableInterface, "is_last_frame", IsLastFrame, frame); }
  int NumFramesReady() const override { PYBIND11_OVERRIDE_NAME(int, DecodableInterface, "num_frames_ready", NumFramesReady); }
  int NumIndices() const override { PYBIND11_OVERRIDE_PURE_NAME(int, DecodableInterface, "num_indices", NumIndices); }
};

AlignConfig ConfigFrom(py::object o) {
  if (py::isinstance<AlignConfig>(o)) return o.cast<AlignConfig>();
  AlignConfig c;
  c.beam = o.attr("beam").cast<float>(); c.retry_beam = o.attr("retry_beam").cast<float>(); c.careful = o.attr("careful").cast<bool>();
  return c;
}
void CheckBeams(const AlignConfig& cfg) {      // csrc/decoder-wrappers.cc:29-33
  if ((cfg.retry_beam != 0 && cfg.retry_beam <= cfg.beam) || cfg.beam <= 0.0f) {
    char b[128];
    std::snprintf(b, sizeof(b), "Beams do not make sense: beam %g, retry-beam %g", (double)cfg.beam, (double)cfg.retry_beam);
    throw Error(b);
  }
}

py::list ResultsToList(const std::vector<AlignResult>& rs, const std::vector<int64_t>& nframes, bool return_scores) {
  py::list out;
  for (size_t u = 0; u < rs.size(); ++u) {
    const AlignResult& r = rs[u];
    py::dict d;
    d["ok"] = r.ok; d["retried"] = r.retried; d["status"] = r.status;
    d["alignment"] = py::cast(r.alignment); d["words"] = py::cast(r.words);
    d["like"] = r.ok ? (double)r.like : 0.0;
    d["num_frames"] = r.num_frames;
    if (return_scores) {
      Arr<float> m({(py::ssize_t)r.pdfs.size(), (py::ssize_t)nframes[u]});
      if (!r.loglikes.empty()) std::memcpy(m.mutable_data(), r.loglikes.data(), sizeof(float) * r.loglikes.size());
      d["loglikes"] = m;
      d["pdfs"] = Vec1(r.pdfs);
    }
    out.append(d);
  }
  return out;
}

// align_batch(am, tm, fsts, feats_list, config, acoustic_scale, trans_cost=None, decoder_opts=None, return_scores=False)
py::list AlignBatchPy(std::shared_ptr<AmDiagGmm> am, std::shared_ptr<TransitionModel> tm, std::vector<std::shared_ptr<StdVectorFst>> fsts, py::list feats_list,
                      py::object config, float acoustic_scale, py::object trans_cost, py::object decoder_opts, bool return_scores) {
  const AlignConfig cfg = ConfigFrom(config);
  CheckBeams(cfg);
  std::vector<StdVectorFst> careful;           // on copies: the batch entry point leaves the caller's graphs alone
  std::vector<const StdVectorFst*> gp;
  if (cfg.careful) {
    careful.reserve(fsts.size());
    for (auto& f : fsts) {
      careful.push_back(*f);
      if (careful.back().Start() != kNoStateId) ModifyGraphForCarefulAlignment(&careful.back());
    }
    for (auto& f : careful) gp.push_back(&f);
  } else {
    for (auto& f : fsts) gp.push_back(f.get());
  }
  const GraphsCsr csr = ConcatGraphs(gp);
  const int D = am->Dim();
  std::vector<Arr<float>> keep;
  std::vector<const float*> fp;
  std::vector<int64_t> nf;
  for (py::handle f : feats_list) {
    Arr<float> a = f.cast<Arr<float>>();
    if (D <= 0 || a.size() % D != 0) throw Error("Dim mismatch: data dim vs. model dim = " + std::to_string(D));
    keep.push_back(a);
    fp.push_back(a.data());
    nf.push_back((int64_t)(a.size() / D));
  }
  Arr<float> tc;
  const float* tcp = nullptr;
  if (!trans_cost.is_none()) {
    tc = trans_cost.cast<Arr<float>>();
    if (tc.size() != tm->NumTransitionIds() + 1) throw Error("trans_cost: one cost per transition-id (+ entry 0)");
    tcp = tc.data();
  }
  FasterDecoderOptions dopts;
  const bool has_opts = !decoder_opts.is_none();
  if (has_opts) dopts = decoder_opts.cast<FasterDecoderOptions>();
  std::vector<AlignResult> rs;
  {
    py::gil_scoped_release nogil;
    rs = AlignBatch(*am, *tm, csr, fp, nf, cfg, acoustic_scale, tcp, has_opts ? &dopts : nullptr, return_scores);
  }
  return ResultsToList(rs, nf, return_scores);
}
}  // namespace

void BindAlign(py::module_& m) {
  py::class_<AlignConfig>(m, "AlignConfig")      // csrc/decoder-wrappers.h:23-37
      .def(py::init([](float beam, float retry_beam, bool careful) { AlignConfig c; c.beam = beam; c.retry_beam = retry_beam; c.careful = careful; return c; }),
           py::arg("beam") = 200.0f, py::arg("retry_beam") = 0.0f, py::arg("careful") = false)
      .def_readwrite("beam", &AlignConfig::beam).def_readwrite("retry_beam", &AlignConfig::retry_beam).def_readwrite("careful", &AlignConfig::careful)
      .def("__str__", [](const AlignConfig& c) {
        char b[128];
        std::snprintf(b, sizeof(b), "AlignConfig(beam=%g, retry_beam=%g, careful=%s)", (double)c.beam, (double)c.retry_beam, c.careful ? "True" : "False");
        return std::string(b);
      });

  py::class_<FasterDecoderOptions>(m, "FasterDecoderOptions")      // csrc/faster-decoder.h:24-63
      .def(py::init([](float beam, int64_t max_active, int min_active, float beam_delta, float hash_ratio) {
             FasterDecoderOptions o;
             o.beam = beam; o.max_active = (int32_t)std::min<int64_t>(max_active, std::numeric_limits<int32_t>::max()); o.min_active = min_active;
             o.beam_delta = beam_delta; o.hash_ratio = hash_ratio;
             return o;
           }), py::arg("beam") = 16.0f, py::arg("max_active") = (int64_t)std::numeric_limits<int32_t>::max(), py::arg("min_active") = 20, py::arg("beam_delta") = 0.5f,
           py::arg("hash_ratio") = 2.0f)
      .def_readwrite("beam", &FasterDecoderOptions::beam).def_readwrite("max_active", &FasterDecoderOptions::max_active)
      .def_readwrite("min_active", &FasterDecoderOptions::min_active).def_readwrite("beam_delta", &FasterDecoderOptions::beam_delta)
      .def_readwrite("hash_ratio", &FasterDecoderOptions::hash_ratio)
      .def("__str__", &FasterDecoderOptions::ToString);

  py::class_<DecodableInterface, PyDecodableInterface, std::shared_ptr<DecodableInterface>>(m, "DecodableInterface")
      .def(py::init<>())
      .def("log_likelihood", &DecodableInterface::LogLikelihood, py::arg("frame"), py::arg("index"))
      .def("is_last_frame", &DecodableInterface::IsLastFrame, py::arg("frame"))
      .def("num_frames_ready", &DecodableInterface::NumFramesReady)
      .def("num_indices", &DecodableInterface::NumIndices);

  py::class_<DecodableAmDiagGmmUnmapped, DecodableInterface, std::shared_ptr<DecodableAmDiagGmmUnmapped>>(m, "DecodableAmDiagGmmUnmapped")
      .def(py::init([](std::shared_ptr<AmDiagGmm> am, Arr<float> feats, float) {
             if (feats.ndim() != 2) throw Error("feats must be a 2-D float matrix");
             return std::make_shared<DecodableAmDiagGmmUnmapped>(std::move(am), feats.data(), (int64_t)feats.shape(0), (int)feats.shape(1));
           }), py::arg("am"), py::arg("feats"), py::arg("log_sum_exp_prune") = -1.0f)
      .def("log_likelihood", &DecodableAmDiagGmmUnmapped::LogLikelihood, py::arg("frame"), py::arg("index"))
      .def("_zero_based", &DecodableAmDiagGmmUnmapped::ZeroBased)
      .def("num_frames_ready", &DecodableAmDiagGmmUnmapped::NumFramesReady)
      .def("num_indices", &DecodableAmDiagGmmUnmapped::NumIndices)
      .def("is_last_frame", &DecodableAmDiagGmmUnmapped::IsLastFrame, py::arg("frame"))
      .def_property_readonly("_am", [](DecodableAmDiagGmmUnmapped& d) { return d.am(); })
      .def_property_readonly("_feats", [](DecodableAmDiagGmmUnmapped& d) {
        Arr<float> a({(py::ssize_t)d.NumFramesReady(), (py::ssize_t)d.Dim()});
        if (!d.feats().empty()) std::memcpy(a.mutable_data(), d.feats().data(), sizeof(float) * d.feats().size());
        return a;
      });

  py::class_<DecodableAmDiagGmmScaled, DecodableAmDiagGmmUnmapped, std::shared_ptr<DecodableAmDiagGmmScaled>>(m, "DecodableAmDiagGmmScaled")
      .def(py::init([](std::shared_ptr<AmDiagGmm> am, std::shared_ptr<TransitionModel> tm, Arr<float> feats, float scale, float) {
             if (feats.ndim() != 2) throw Error("feats must be a 2-D float matrix");
             return std::make_shared<DecodableAmDiagGmmScaled>(std::move(am), std::move(tm), feats.data(), (int64_t)feats.shape(0), (int)feats.shape(1), scale);
           }), py::arg("am"), py::arg("tm"), py::arg("feats"), py::arg("scale"), py::arg("log_sum_exp_prune") = -1.0f)
      .def_property_readonly("transition_model", [](DecodableAmDiagGmmScaled& d) { return d.tm(); })
      .def_property_readonly("_tm", [](DecodableAmDiagGmmScaled& d) { return d.tm(); })
      .def_property_readonly("_scale", [](DecodableAmDiagGmmScaled& d) { return (double)d.scale(); });

  m.def("align_batch", &AlignBatchPy, py::arg("am"), py::arg("tm"), py::arg("fsts"), py::arg("feats_list"), py::arg("config"), py::arg("acoustic_scale"),
        py::arg("trans_cost") = py::none(), py::arg("decoder_opts") = py::none(), py::arg("return_scores") = false);

  // python/csrc/decoder-wrappers.cc:25-47 -> (num_done, num_error, num_retried, tot_like, frame_count, alignment, words); the counters
  // are passed by value and returned incremented
  // `decodable` is any DecodableInterface, as in the reference: a DecodableAmDiagGmmScaled runs K1 + K2 (scores scaled by the
  // decodable's own scale, `like` divided by acoustic_scale, decoder-wrappers.cc:95); anything else -- the unmapped GMM decodable, a
  // Python subclass of DecodableInterface -- has its scores sampled through log_likelihood(frame, index) and decoded by K2.
  m.def("align_utterance_wrapper", [](py::object config, const std::string&, float acoustic_scale, std::shared_ptr<StdVectorFst> fst,
                                      std::shared_ptr<DecodableInterface> decodable, int num_done, int num_error, int num_retried, double tot_like,
                                      int64_t frame_count) {
    if (!fst) throw Error("align_utterance_wrapper: fst is None");
    if (!decodable) throw Error("align_utterance_wrapper: decodable is None");
    AlignConfig cfg = ConfigFrom(config);
    CheckBeams(cfg);
    if (fst->Start() == kNoStateId)                   // "Empty decoding graph" (decoder-wrappers.cc:35-41)
      return py::tuple(py::make_tuple(num_done, num_error + 1, num_retried, tot_like, frame_count, py::list(), py::list()));
    if (cfg.careful) {
      ModifyGraphForCarefulAlignment(fst.get());      // the reference mutates the caller's fst (decoder-wrappers.cc:43-45)
      cfg.careful = false;
    }
    AlignResult r;
    if (auto dec = std::dynamic_pointer_cast<DecodableAmDiagGmmScaled>(decodable)) {
      py::gil_scoped_release nogil;
      r = AlignBatch(*dec->am(), *dec->tm(), ConcatGraphs({fst.get()}), {dec->feats().data()}, {(int64_t)dec->NumFramesReady()}, cfg, dec->scale(), nullptr,
                     nullptr, false, acoustic_scale)[0];
    } else {
      r = AlignDecodable(*fst, *decodable, cfg, acoustic_scale, nullptr);       // GIL held: the scores may come from Python
    }
    if (r.retried) num_retried += 1;
    if (!r.ok) return py::tuple(py::make_tuple(num_done, num_error + 1, num_retried, tot_like, frame_count, py::list(), py::list()));
    return py::tuple(py::make_tuple(num_done + 1, num_error, num_retried, tot_like + (double)r.like, frame_count + (int64_t)r.num_frames, py::cast(r.alignment),
                                    py::cast(r.words)));
  }, py::arg("config"), py::arg("utt"), py::arg("acoustic_scale"), py::arg("fst"), py::arg("decodable"), py::arg("num_done"), py::arg("num_error"),
     py::arg("num_retried"), py::arg("tot_like"), py::arg("frame_count"));

  // ---- the graph container (kaldifst's method names) -----------------------------------------------------------------------------
  m.attr("kNoStateId") = kNoStateId;
  py::class_<StdArc>(m, "StdArc")
      .def(py::init([](int il, int ol, double w, int ns) { return StdArc{il, ol, (float)w, ns}; }), py::arg("ilabel"), py::arg("olabel"), py::arg("weight"),
           py::arg("nextstate"))
      .def_readwrite("ilabel", &StdArc::ilabel).def_readwrite("olabel", &StdArc::olabel).def_readwrite("nextstate", &StdArc::nextstate)
      .def_property("weight", [](const StdArc& a) { return (double)a.weight; }, [](StdArc& a, double w) { a.weight = (float)w; })
      .def("__repr__", &StdArc::ToString);

  py::class_<StdVectorFst, std::shared_ptr<StdVectorFst>>(m, "StdVectorFst")
      .def(py::init<>())
      .def("add_state", &StdVectorFst::AddState)
      .def_property_readonly("num_states", &StdVectorFst::NumStates)
      .def_property("start", &StdVectorFst::Start, &StdVectorFst::SetStart)
      .def("set_start", &StdVectorFst::SetStart)
      .def("add_arc", [](StdVectorFst& f, int state, py::object arc, py::kwargs kw) {
        if (!arc.is_none()) { f.AddArc(state, arc.cast<StdArc>()); return; }
        f.AddArc(state, StdArc{kw["ilabel"].cast<int>(), kw["olabel"].cast<int>(), kw.contains("weight") ? (float)kw["weight"].cast<double>() : 0.0f,
                               kw["nextstate"].cast<int>()});
      }, py::arg("state"), py::arg("arc") = py::none())
      .def("set_final", [](StdVectorFst& f, int s, double w) { f.SetFinal(s, (float)w); }, py::arg("state"), py::arg("weight") = 0.0)
      .def("final", [](StdVectorFst& f, int s) { return (double)f.Final(s); }, py::arg("state"))
      .def("is_final", &StdVectorFst::IsFinal, py::arg("state"))
      .def("arcs", [](StdVectorFst& f, int s) { return f.Arcs(s); }, py::arg("state"))       // copies: the container owns its arcs
      .def("num_arcs", [](StdVectorFst& f, py::object s) { return s.is_none() ? f.NumArcs() : (int64_t)f.Arcs(s.cast<int>()).size(); }, py::arg("state") = py::none())
      .def("copy", [](StdVectorFst& f) { return std::make_shared<StdVectorFst>(f); })
      .def_property_readonly("_arcs", [](StdVectorFst& f) { return f.arcs(); })
      .def_property_readonly("_final", [](StdVectorFst& f) { return std::vector<double>(f.finals().begin(), f.finals().end()); })
      .def_property_readonly("_start", &StdVectorFst::Start)
      .def("to_csr", [](StdVectorFst& f) {
        const GraphsCsr c = ConcatGraphs({&f});
        py::dict d;
        d["start"] = f.Start(); d["arc_off"] = Vec1(c.arc_off); d["ilabel"] = Vec1(c.ilabel); d["olabel"] = Vec1(c.olabel); d["weight"] = Vec1(c.weight);
        d["nextstate"] = Vec1(c.nextstate); d["final"] = Vec1(c.final_w);
        return d;
      })
      .def_static("from_csr", [](int start, Arr<int64_t> arc_off, Arr<int32_t> il, Arr<int32_t> ol, Arr<float> w, Arr<int32_t> ns, Arr<float> fin) {
        auto f = std::make_shared<StdVectorFst>();
        for (py::ssize_t s = 0; s < fin.size(); ++s) {
          f->AddState();
          f->SetFinal((int)s, fin.at(s));
          for (int64_t a = arc_off.at(s); a < arc_off.at(s + 1); ++a) f->AddArc((int)s, StdArc{il.at(a), ol.at(a), w.at(a), ns.at(a)});
        }
        f->SetStart(start);
        return f;
      }, py::arg("start"), py::arg("arc_off"), py::arg("ilabel"), py::arg("olabel"), py::arg("weight"), py::arg("nextstate"), py::arg("final"));

  m.def("concat_graphs", [](std::vector<std::shared_ptr<StdVectorFst>> fsts) {
    std::vector<const StdVectorFst*> p;
    for (auto& f : fsts) p.push_back(f.get());
    return CsrToDict(ConcatGraphs(p));
  }, py::arg("fsts"));
  m.def("modify_graph_for_careful_alignment", [](StdVectorFst& f) { ModifyGraphForCarefulAlignment(&f); }, py::arg("fst"));
  // python/csrc/hmm-utils.cc:14-19: disambig_syms defaults to empty; the rest are required
  // (pybind11 lets a defaulted argument precede required ones, as the reference's binding does)
  m.def("add_transition_probs", [](const TransitionModel& tm, std::vector<int> disambig, float ts, float sl, std::shared_ptr<StdVectorFst> fst) {
    if (!fst) throw Error("add_transition_probs: fst is None");
    AddTransitionProbs(tm, disambig, ts, sl, fst.get());
  }, py::arg("trans_model"), py::arg("disambig_syms") = std::vector<int>(), py::arg("transition_scale"), py::arg("self_loop_scale"), py::arg("fst"));

  py::class_<LatticeWeight>(m, "LatticeWeight")
      .def(py::init([](double a, double b) { return LatticeWeight{a, b}; }), py::arg("value1") = 0.0, py::arg("value2") = 0.0)
      .def_readwrite("value1", &LatticeWeight::value1).def_readwrite("value2", &LatticeWeight::value2)
      .def("__repr__", [](const LatticeWeight& w) {
        return "LatticeWeight(" + py::repr(py::float_(w.value1)).cast<std::string>() + ", " + py::repr(py::float_(w.value2)).cast<std::string>() + ")";
      });
  py::class_<LatticeArc>(m, "LatticeArc")
      .def(py::init([](int il, int ol, LatticeWeight w, int ns) { return LatticeArc{il, ol, w, ns}; }), py::arg("ilabel"), py::arg("olabel"), py::arg("weight"),
           py::arg("nextstate"))
      .def_readwrite("ilabel", &LatticeArc::ilabel).def_readwrite("olabel", &LatticeArc::olabel).def_readwrite("weight", &LatticeArc::weight)
      .def_readwrite("nextstate", &LatticeArc::nextstate);
  py::class_<LinearLattice>(m, "LinearLattice")
      .def(py::init<>())
      .def_readwrite("arcs", &LinearLattice::arcs).def_readwrite("final", &LinearLattice::final_w).def_readwrite("start", &LinearLattice::start)
      .def_property_readonly("num_states", &LinearLattice::NumStates)
      .def("get_linear_symbol_sequence", [](const LinearLattice& l) {
        std::vector<int> il, ol;
        LatticeWeight w;
        const bool ok = l.GetLinearSymbolSequence(&il, &ol, &w);
        return py::make_tuple(ok, il, ol, w);
      });

  py::class_<FasterDecoder>(m, "FasterDecoder")      // python/csrc/faster-decoder.cc:33-53 (the method is spelled advanced_decoding there)
      .def(py::init<std::shared_ptr<StdVectorFst>, const FasterDecoderOptions&>(), py::arg("fst"), py::arg("config"))
      .def("set_options", &FasterDecoder::SetOptions, py::arg("config"))
      .def("init_decoding", &FasterDecoder::InitDecoding)
      .def("decode", [](FasterDecoder& d, std::shared_ptr<DecodableInterface> dec) { d.Decode(dec); }, py::arg("decodable"))
      .def("advanced_decoding", [](FasterDecoder& d, std::shared_ptr<DecodableInterface> dec, int max_num_frames) { d.AdvanceDecoding(dec, max_num_frames); },
           py::arg("decodable"), py::arg("max_num_frames") = -1)
      .def("num_frames_decoded", &FasterDecoder::NumFramesDecoded)
      .def("reached_final", &FasterDecoder::ReachedFinal)
      .def("get_best_path", [](FasterDecoder& d, bool use_final_probs) {
        LinearLattice lat;
        const bool ok = d.GetBestPath(&lat, use_final_probs);
        return py::make_tuple(ok, lat);
      }, py::arg("use_final_probs") = true);
}
