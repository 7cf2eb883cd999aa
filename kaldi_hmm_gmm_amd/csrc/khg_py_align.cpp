// pybind11 bindings of the alignment API (khg_host_align.hpp) with the names of python/csrc/{decoder-wrappers,faster-decoder,
// decodable-am-diag-gmm}.cc in /root/reference/kaldi-hmm-gmm.  The graph container (the reference's kaldifst VectorFst) is the
// Python StdVectorFst: the two places that need it (ModifyGraphForCarefulAlignment, the CSR view) call its module.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "khg_host_align.hpp"

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
template <class T>
std::vector<T> ToVec(py::handle o) {
  Arr<T> a = o.cast<Arr<T>>();
  return std::vector<T>(a.data(), a.data() + a.size());
}
GraphsCsr CsrFromDict(py::dict g) {
  GraphsCsr c;
  c.state_off = ToVec<int64_t>(g["state_off"]); c.arc_off = ToVec<int64_t>(g["arc_off"]); c.start = ToVec<int32_t>(g["start"]);
  c.ilabel = ToVec<int32_t>(g["ilabel"]); c.olabel = ToVec<int32_t>(g["olabel"]); c.nextstate = ToVec<int32_t>(g["nextstate"]);
  c.weight = ToVec<float>(g["weight"]); c.final_w = ToVec<float>(g["final"]);
  return c;
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

py::object FstModule() { return py::module_::import("kaldi_hmm_gmm_amd.fst"); }

AlignConfig ConfigFrom(py::object o) {
  if (py::isinstance<AlignConfig>(o)) return o.cast<AlignConfig>();
  AlignConfig c;
  c.beam = o.attr("beam").cast<float>(); c.retry_beam = o.attr("retry_beam").cast<float>(); c.careful = o.attr("careful").cast<bool>();
  return c;
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
py::list AlignBatchPy(std::shared_ptr<AmDiagGmm> am, std::shared_ptr<TransitionModel> tm, py::list fsts, py::list feats_list, py::object config,
                      float acoustic_scale, py::object trans_cost, py::object decoder_opts, bool return_scores) {
  const AlignConfig cfg = ConfigFrom(config);
  if ((cfg.retry_beam != 0 && cfg.retry_beam <= cfg.beam) || cfg.beam <= 0.0f) {
    char b[128];
    std::snprintf(b, sizeof(b), "Beams do not make sense: beam %g, retry-beam %g", (double)cfg.beam, (double)cfg.retry_beam);
    throw Error(b);
  }
  py::object fstmod = FstModule();
  py::list graphs = fsts;
  if (cfg.careful) {                       // on copies: the batch entry point leaves the caller's graphs alone
    graphs = py::list();
    for (py::handle f : fsts) {
      py::object c = f.attr("copy")();
      if (c.attr("start").cast<int>() != -1) fstmod.attr("modify_graph_for_careful_alignment")(c);
      graphs.append(c);
    }
  }
  const GraphsCsr csr = CsrFromDict(fstmod.attr("concat_graphs")(graphs).cast<py::dict>());
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
  m.def("align_utterance_wrapper", [](py::object config, const std::string&, float acoustic_scale, py::object fst, py::object decodable, int num_done,
                                      int num_error, int num_retried, double tot_like, int64_t frame_count) {
    if (!py::isinstance<DecodableAmDiagGmmScaled>(decodable)) throw Error("align_utterance_wrapper: the HIP path needs a DecodableAmDiagGmmScaled");
    auto dec = decodable.cast<std::shared_ptr<DecodableAmDiagGmmScaled>>();
    // the reference scales scores by the decodable's scale and `like` by acoustic_scale; the scripts pass the same value
    if (dec->scale() != acoustic_scale) throw Error("align_utterance_wrapper: decodable scale and acoustic_scale must agree on this path");
    AlignConfig cfg = ConfigFrom(config);
    if (cfg.careful && fst.attr("start").cast<int>() != -1) {
      FstModule().attr("modify_graph_for_careful_alignment")(fst);      // the reference mutates the caller's fst (decoder-wrappers.cc:43-45)
      cfg.careful = false;
    }
    py::list fsts; fsts.append(fst);
    py::list feats; feats.append(decodable.attr("_feats"));
    py::dict r = AlignBatchPy(dec->am(), dec->tm(), fsts, feats, py::cast(cfg), acoustic_scale, py::none(), py::none(), false)[0].cast<py::dict>();
    if (r["retried"].cast<bool>()) num_retried += 1;
    if (!r["ok"].cast<bool>()) return py::tuple(py::make_tuple(num_done, num_error + 1, num_retried, tot_like, frame_count, py::list(), py::list()));
    return py::tuple(py::make_tuple(num_done + 1, num_error, num_retried, tot_like + r["like"].cast<double>(), frame_count + r["num_frames"].cast<int64_t>(),
                                    py::object(r["alignment"]), py::object(r["words"])));
  }, py::arg("config"), py::arg("utt"), py::arg("acoustic_scale"), py::arg("fst"), py::arg("decodable"), py::arg("num_done") = 0, py::arg("num_error") = 0,
     py::arg("num_retried") = 0, py::arg("tot_like") = 0.0, py::arg("frame_count") = 0);
}
