// pybind11 bindings of the host-side C++ classes (khg_host_gmm.hpp, khg_host_hmm.hpp, khg_host_align.hpp) under the reference's
// Python names and signatures: python/csrc/{diag-gmm,am-diag-gmm,model-common,mle-diag-gmm,mle-am-diag-gmm,hmm-topology,
// transition-model,transition-information,decoder-wrappers,faster-decoder,decodable-am-diag-gmm}.cc in
// /root/reference/kaldi-hmm-gmm.  Part of the module _kaldi_hmm_gmm_amd (khg_pybind.cpp calls BindHost).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "khg_host_align.hpp"
#include "khg_host_gmm.hpp"
#include "khg_host_hmm.hpp"

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
Arr<T> Vec2(const std::vector<T>& v, size_t rows, size_t cols) {
  Arr<T> a({(py::ssize_t)rows, (py::ssize_t)cols});
  if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), sizeof(T) * v.size());
  return a;
}
// a writable numpy view of a member vector, kept alive by (and keeping alive) its owner
template <class T>
py::array View1(std::vector<T>& v, py::handle owner) {
  return py::array_t<T>({(py::ssize_t)v.size()}, {(py::ssize_t)sizeof(T)}, v.data(), owner);
}
template <class T>
py::array View2(std::vector<T>& v, size_t rows, size_t cols, py::handle owner) {
  if (v.empty()) rows = cols = 0;
  return py::array_t<T>({(py::ssize_t)rows, (py::ssize_t)cols}, {(py::ssize_t)(sizeof(T) * cols), (py::ssize_t)sizeof(T)}, v.data(), owner);
}
std::vector<float> FVec(const Arr<float>& a) { return std::vector<float>(a.data(), a.data() + a.size()); }

// the calls below that reach the device (K1 / K3 / K4 through the C-ABI) run without the GIL: the argument arrays are owned by the
// caller's frame for the duration of the call, and nothing inside touches a Python object
template <class F>
auto NoGil(F&& f) -> decltype(f()) {
  py::gil_scoped_release release;
  return f();
}

// the `randn` callables of the Python API: randn(d) -> d deviates, randn((rows, cols)) -> a matrix; None = numpy's global generator
RandnFn MakeRandn(py::object randn) {
  return [randn](float* out, size_t rows, size_t cols) {
    py::object arg = rows == 0 ? py::object(py::int_(cols)) : py::object(py::make_tuple(rows, cols));
    py::object r = randn.is_none() ? py::module_::import("numpy").attr("random").attr("standard_normal")(arg) : randn(arg);
    Arr<float> a = Arr<float>::ensure(r);
    const size_t n = (rows == 0 ? 1 : rows) * cols;
    if (!a || (size_t)a.size() != n) throw Error("randn: expected " + std::to_string(n) + " deviates");
    std::memcpy(out, a.data(), sizeof(float) * n);
  };
}

py::tuple UpdateTuple(const MleUpdateResult& r) { return py::make_tuple(r.objf_change, r.count, r.floored_elements, r.floored_gaussians, r.removed); }

MleDiagGmmOptions OptsFrom(py::object o) {     // a bound MleDiagGmmOptions or any object with its attributes
  if (o.is_none()) return MleDiagGmmOptions();
  if (py::isinstance<MleDiagGmmOptions>(o)) return o.cast<MleDiagGmmOptions>();
  MleDiagGmmOptions c;
  c.min_gaussian_weight = o.attr("min_gaussian_weight").cast<float>();
  c.min_gaussian_occupancy = o.attr("min_gaussian_occupancy").cast<float>();
  c.min_variance = o.attr("min_variance").cast<double>();
  c.remove_low_count_gaussians = o.attr("remove_low_count_gaussians").cast<bool>();
  if (py::hasattr(o, "variance_floor_vector") && !o.attr("variance_floor_vector").is_none()) {
    Arr<double> v = o.attr("variance_floor_vector").cast<Arr<double>>();
    c.variance_floor_vector.assign(v.data(), v.data() + v.size());
  }
  return c;
}

void BindGmm(py::module_& m) {
  m.attr("_default_context") = py::none();
  m.def("set_default_context", [m](py::object ctx) {
    // the Context object stays alive, as a module attribute, while it is the default: released with the interpreter, not
    // by a static destructor after the HIP runtime has gone
    py::module_(m).attr("_default_context") = ctx;
    SetDefaultCtx(ctx.is_none() ? nullptr : reinterpret_cast<khg_ctx*>(ctx.attr("h").cast<uintptr_t>()));
  });
  m.def("str_to_gmm_flags", &StrToGmmFlags);
  m.def("gmm_flags_to_str", &GmmFlagsToStr);
  m.def("augment_gmm_flags", &AugmentGmmFlags);
  m.def("get_split_targets", [](Arr<float> occs, int target, float power, double min_count) {
    return GetSplitTargetsD(FVec(occs), target, power, min_count);
  }, py::arg("state_occs"), py::arg("target_components"), py::arg("power"), py::arg("min_count"));

  py::class_<DiagGmm, std::shared_ptr<DiagGmm>>(m, "DiagGmm")
      // the four constructors of python/csrc/diag-gmm.cc:18-22
      .def(py::init([]() { return std::make_shared<DiagGmm>(); }))
      .def(py::init([](const DiagGmm& gmm) { return std::make_shared<DiagGmm>(gmm); }), py::arg("gmm"))
      .def(py::init([](int nmix, int dim) { return std::make_shared<DiagGmm>(nmix, dim); }), py::arg("nmix"), py::arg("dim"))
      .def(py::init([](const std::vector<std::pair<float, std::shared_ptr<DiagGmm>>>& gmms) {
             std::vector<std::pair<float, const DiagGmm*>> v;
             for (auto& p : gmms) {
               if (!p.second) throw Error("DiagGmm(gmms): a None entry");
               v.emplace_back(p.first, p.second.get());
             }
             return std::make_shared<DiagGmm>(v);
           }), py::arg("gmms"))
      .def("resize", &DiagGmm::Resize, py::arg("nmix"), py::arg("dim"))
      .def("copy_from_diag_gmm", &DiagGmm::CopyFromDiagGmm, py::arg("diaggmm"))
      .def_property_readonly("num_gauss", &DiagGmm::NumGauss)
      .def_property_readonly("dim", &DiagGmm::Dim)
      .def_property_readonly("valid_gconsts", &DiagGmm::ValidGconsts)
      .def_property_readonly("_version", &DiagGmm::version)      // mutation counter (what AmDiagGmm keys its cached device model by)
      .def_property_readonly("gconsts", [](DiagGmm& g) { return Vec1(g.gconsts()); })
      .def_property("weights", [](DiagGmm& g) { return Vec1(g.weights()); }, [](DiagGmm& g, Arr<float> w) { g.SetWeights(w.data(), (size_t)w.size()); })
      .def_property_readonly("means_invvars", [](DiagGmm& g) { return Vec2(g.means_invvars(), g.NumGauss(), g.Dim()); })
      .def_property_readonly("inv_vars", [](DiagGmm& g) { return Vec2(g.inv_vars(), g.NumGauss(), g.Dim()); })
      .def_property_readonly("means", [](DiagGmm& g) { return Vec2(g.GetMeans(), g.NumGauss(), g.Dim()); })
      .def_property_readonly("vars", [](DiagGmm& g) { return Vec2(g.GetVars(), g.NumGauss(), g.Dim()); })
      // the storage itself (what the Python classes kept in these attributes): read = copy, write = raw replacement
      .def_property("_gconsts", [](DiagGmm& g) { return Vec1(g.gconsts()); }, [](DiagGmm& g, Arr<float> a) { g.mutable_gconsts() = FVec(a); })
      .def_property("_weights", [](DiagGmm& g) { return Vec1(g.weights()); }, [](DiagGmm& g, Arr<float> a) { g.mutable_weights() = FVec(a); })
      .def_property("_inv_vars", [](DiagGmm& g) { return Vec2(g.inv_vars(), g.NumGauss(), g.Dim()); },
                    [](DiagGmm& g, Arr<float> a) {
                      if (a.ndim() != 2 || a.shape(0) != g.NumGauss() || a.shape(1) != g.Dim()) throw Error("_inv_vars: shape mismatch");
                      g.mutable_inv_vars() = FVec(a);
                    })
      .def_property("_means_invvars", [](DiagGmm& g) { return Vec2(g.means_invvars(), g.NumGauss(), g.Dim()); },
                    [](DiagGmm& g, Arr<float> a) {
                      if (a.ndim() != 2 || a.shape(0) != g.NumGauss() || a.shape(1) != g.Dim()) throw Error("_means_invvars: shape mismatch");
                      g.mutable_means_invvars() = FVec(a);
                    })
      .def_property("_valid_gconsts", &DiagGmm::ValidGconsts, &DiagGmm::set_valid_gconsts)
      .def("set_weights", [](DiagGmm& g, Arr<float> w) { g.SetWeights(w.data(), (size_t)w.size()); }, py::arg("w"))
      .def("set_means", [](DiagGmm& g, Arr<float> a) {
        if (a.ndim() != 2) throw Error("SetMeans: shape mismatch");
        g.SetMeans(a.data(), (size_t)a.shape(0), (size_t)a.shape(1));
      }, py::arg("m"))
      .def("set_invvars", [](DiagGmm& g, Arr<float> a) {
        if (a.ndim() != 2) throw Error("SetInvVars: shape mismatch");
        g.SetInvVars(a.data(), (size_t)a.shape(0), (size_t)a.shape(1));
      }, py::arg("inv_vars"))
      .def("set_invvars_and_means", [](DiagGmm& g, Arr<float> v, Arr<float> mu) {
        if (v.ndim() != 2 || mu.ndim() != 2 || v.shape(0) != mu.shape(0) || v.shape(1) != mu.shape(1)) throw Error("SetInvVarsAndMeans: shape mismatch");
        g.SetInvVarsAndMeans(v.data(), mu.data(), (size_t)v.shape(0), (size_t)v.shape(1));
      }, py::arg("inv_vars"), py::arg("means"))
      .def("set_component_weight", &DiagGmm::SetComponentWeight, py::arg("gauss"), py::arg("weight"))
      .def("set_component_mean", [](DiagGmm& g, int i, Arr<float> v) { g.SetComponentMean(i, v.data(), (size_t)v.size()); }, py::arg("gauss"), py::arg("mean"))
      .def("set_component_inv_var", [](DiagGmm& g, int i, Arr<float> v) { g.SetComponentInvVar(i, v.data(), (size_t)v.size()); }, py::arg("gauss"),
           py::arg("inv_var"))
      .def("get_component_mean", [](DiagGmm& g, int i) { return Vec1(g.GetComponentMean(i)); }, py::arg("gauss"))
      .def("get_component_variance", [](DiagGmm& g, int i) { return Vec1(g.GetComponentVariance(i)); }, py::arg("gauss"))
      .def("remove_component", &DiagGmm::RemoveComponent, py::arg("gauss"), py::arg("renorm_weights"))
      .def("remove_components", &DiagGmm::RemoveComponents, py::arg("gauss"), py::arg("renorm_weights"))
      .def("compute_gconsts", &DiagGmm::ComputeGconsts)
      .def("_need_gconsts", &DiagGmm::NeedGconsts)
      .def("_as_model", [](DiagGmm& g, bool per_component) {
        const int G = g.NumGauss();
        std::vector<int32_t> go;
        if (per_component) for (int i = 0; i <= G; ++i) go.push_back(i);
        else go = {0, G};
        return py::make_tuple(Vec1(go), Vec1(g.gconsts()), Vec2(g.means_invvars(), G, g.Dim()), Vec2(g.inv_vars(), G, g.Dim()));
      })
      .def("log_likelihood", [](DiagGmm& g, Arr<float> x) { return NoGil([&] { return g.LogLikelihood(x.data(), (size_t)x.size()); }); }, py::arg("data"),
           "Return the total loglikes in a float")
      .def("log_likelihoods", [](DiagGmm& g, Arr<float> x) { return Vec1(NoGil([&] { return g.LogLikelihoods(x.data(), (size_t)x.size()); })); }, py::arg("data"),
           "Return the loglike of each component in a 1-D tensor")
      .def("log_likelihoods_matrix", [](DiagGmm& g, Arr<float> x) {
        if (x.ndim() != 2 || x.shape(0) == 0) throw Error("data.rows() != 0 assertion failed");
        return Vec2(NoGil([&] { return g.LogLikelihoodsMatrix(x.data(), (size_t)x.shape(0), (size_t)x.shape(1)); }), (size_t)x.shape(0), (size_t)g.NumGauss());
      }, py::arg("data"), "data is a 2-D tensor of shape (N, dim); returns a 2-D tensor of shape (N, nmix) with the loglike of each component")
      .def("log_likelihoods_preselect", [](DiagGmm& g, Arr<float> x, std::vector<int64_t> idx) {
        const std::vector<float> ll = NoGil([&] { return g.LogLikelihoods(x.data(), (size_t)x.size()); });
        std::vector<float> out;
        for (int64_t i : idx) {
          if (i < 0) i += (int64_t)ll.size();
          if (i < 0 || i >= (int64_t)ll.size()) throw Error("log_likelihoods_preselect: index out of range");
          out.push_back(ll[(size_t)i]);
        }
        return Vec1(out);
      }, py::arg("data"), py::arg("indices"))
      .def("component_log_likelihood", [](DiagGmm& g, Arr<float> x, int comp) {
        if (comp < 0 || comp >= g.NumGauss()) throw Error("comp_id out of range");
        return NoGil([&] { return g.LogLikelihoods(x.data(), (size_t)x.size()); })[(size_t)comp];
      }, py::arg("data"), py::arg("comp_id"))
      .def("component_posteriors", [](DiagGmm& g, Arr<float> x) {
        std::vector<float> post;
        const double ll = NoGil([&] { return g.ComponentPosteriors(x.data(), (size_t)x.size(), &post); });
        return py::make_tuple(ll, Vec1(post));
      }, py::arg("data"))
      // -> the split history like python/csrc/diag-gmm.cc:69-77 (new component i + old count was split off component history[i]);
      // `history` (a list, extended too) and `randn` (the deviates, the reference draws them itself) are this package's additions
      .def("split", [](DiagGmm& g, int target, float perturb, py::object history, py::object randn) {
        std::vector<int> h;
        g.Split(target, perturb, &h, MakeRandn(randn));
        if (!history.is_none()) for (int x : h) history.attr("append")(x);
        return h;
      }, py::arg("target_components"), py::arg("perturb_factor"), py::arg("history") = py::none(), py::arg("randn") = py::none())
      .def("gaussian_selection_1d", [](DiagGmm& g, Arr<float> x, int num_gselect) {
        std::vector<int32_t> out;
        const float f = NoGil([&] { return g.GaussianSelection(x.data(), (size_t)x.size(), num_gselect, &out); });
        return py::make_tuple(f, out);
      }, py::arg("data"), py::arg("num_gselect"))
      .def("gaussian_selection_2d", [](DiagGmm& g, Arr<float> x, int num_gselect) {
        if (x.ndim() != 2) throw Error("data must be a 2-D float matrix");
        std::vector<std::vector<int32_t>> out;
        const float f = NoGil([&] { return g.GaussianSelectionMatrix(x.data(), (size_t)x.shape(0), (size_t)x.shape(1), num_gselect, &out); });
        return py::make_tuple(f, out);
      }, py::arg("data"), py::arg("num_gselect"))
      .def("gaussian_selection_preselect", [](DiagGmm& g, Arr<float> x, std::vector<int32_t> preselect, int num_gselect) {
        std::vector<int32_t> out;
        const float f = NoGil([&] { return g.GaussianSelectionPreselect(x.data(), (size_t)x.size(), preselect, num_gselect, &out); });
        return py::make_tuple(f, out);
      }, py::arg("data"), py::arg("preselect"), py::arg("num_gselect"))
      .def("merge", &DiagGmm::Merge, py::arg("target_components"), py::call_guard<py::gil_scoped_release>())
      .def("perturb", [](DiagGmm& g, float pf, py::object randn) { g.Perturb(pf, MakeRandn(randn)); }, py::arg("perturb_factor"), py::arg("randn") = py::none())
      .def("generate", [](DiagGmm& g, py::object randn) { return Vec1(g.Generate(MakeRandn(randn))); }, py::arg("randn") = py::none())
      .def("interpolate", [](DiagGmm& g, float rho, const DiagGmm& src, int flags) { g.Interpolate(rho, src, flags); }, py::arg("rho"), py::arg("source"),
           py::arg("flags") = (int)kGmmAll)
      // pickle: (weights, inv_vars, means_invvars); gconsts are re-derived (python/csrc/diag-gmm.cc:157-167)
      .def(py::pickle(
          [](DiagGmm& g) { return py::make_tuple(Vec1(g.weights()), Vec2(g.inv_vars(), g.NumGauss(), g.Dim()), Vec2(g.means_invvars(), g.NumGauss(), g.Dim())); },
          [](py::tuple t) {
            Arr<float> w = t[0].cast<Arr<float>>(), iv = t[1].cast<Arr<float>>(), miv = t[2].cast<Arr<float>>();
            auto g = std::make_shared<DiagGmm>();
            g->SetRaw((int)w.shape(0), iv.ndim() == 2 ? (int)iv.shape(1) : 0, w.data(), iv.data(), miv.data(), nullptr);
            g->ComputeGconsts();
            return g;
          }));

  py::class_<AmDiagGmm, std::shared_ptr<AmDiagGmm>>(m, "AmDiagGmm")
      .def(py::init<>())
      .def_property_readonly("dim", &AmDiagGmm::Dim)
      .def_property_readonly("num_pdfs", &AmDiagGmm::NumPdfs)
      .def_property_readonly("_version", &AmDiagGmm::Version)
      .def_property_readonly("num_gauss", &AmDiagGmm::NumGauss)
      .def("num_gauss_in_pdf", [](AmDiagGmm& a, int i) { return a.GetPdf(i)->NumGauss(); }, py::arg("pdf_index"))
      .def("init", &AmDiagGmm::Init, py::arg("proto"), py::arg("num_pdfs"))
      .def("add_pdf", &AmDiagGmm::AddPdf, py::arg("gmm"))
      .def("copy_from_am_diag_gmm", &AmDiagGmm::CopyFromAmDiagGmm, py::arg("other"))
      .def("get_pdf", [](AmDiagGmm& a, int i) { return a.GetPdf(i); }, py::arg("pdf_index"))      // reference-returning, like the reference's binding
      .def_property("_pdfs", [](const AmDiagGmm& a) { return a.pdfs(); }, [](AmDiagGmm& a, std::vector<std::shared_ptr<DiagGmm>> v) { a.pdfs() = std::move(v); })
      .def("compute_gconsts", &AmDiagGmm::ComputeGconsts)
      .def("log_likelihood", [](AmDiagGmm& a, int i, Arr<float> x) { return NoGil([&] { return a.GetPdf(i)->LogLikelihood(x.data(), (size_t)x.size()); }); }, py::arg("pdf_index"), py::arg("data"))
      .def("get_gaussian_mean", [](AmDiagGmm& a, int i, int g) { return Vec1(a.GetPdf(i)->GetComponentMean(g)); }, py::arg("pdf_index"), py::arg("gauss"))
      .def("get_gaussian_variance", [](AmDiagGmm& a, int i, int g) { return Vec1(a.GetPdf(i)->GetComponentVariance(g)); }, py::arg("pdf_index"), py::arg("gauss"))
      .def("set_gaussian_mean", [](AmDiagGmm& a, int i, int g, Arr<float> v) { a.GetPdf(i)->SetComponentMean(g, v.data(), (size_t)v.size()); },
           py::arg("pdf_index"), py::arg("gauss_index"), py::arg("in"))
      .def("split_pdf", [](AmDiagGmm& a, int i, int target, float pf) { a.GetPdf(i)->Split(target, pf, nullptr, MakeRandn(py::none())); },
           py::arg("pdf_idx"), py::arg("target_components"), py::arg("perturb_factor"))
      .def("split_by_count", [](AmDiagGmm& a, Arr<float> occs, int target, float pf, float power, double min_count, py::object randn) {
        a.SplitByCount(FVec(occs), target, pf, power, min_count, MakeRandn(randn));
      }, py::arg("state_occs"), py::arg("target_components"), py::arg("perturb_factor"), py::arg("power"), py::arg("min_count"), py::arg("randn") = py::none())
      .def("merge_by_count", [](AmDiagGmm& a, Arr<float> occs, int target, float power, double min_count) { a.MergeByCount(FVec(occs), target, power, min_count); },
           py::arg("state_occs"), py::arg("target_components"), py::arg("power"), py::arg("min_count"))
      .def("flat", [](AmDiagGmm& a) {
        std::vector<int32_t> go; std::vector<float> gc, w, miv, iv;
        a.Flat(&go, &gc, &w, &miv, &iv);
        const size_t G = gc.size(), D = (size_t)a.Dim();
        return py::make_tuple(Vec1(go), Vec1(gc), Vec1(w), Vec2(miv, G, D), Vec2(iv, G, D));
      })
      .def("set_flat", [](AmDiagGmm& a, Arr<int32_t> go, Arr<float> w, Arr<float> gc, Arr<float> miv, Arr<float> iv) {
        if (go.size() != a.NumPdfs() + 1) throw Error("set_flat: gauss_off has num_pdfs + 1 entries");
        const int64_t n = go.at(a.NumPdfs());
        if (w.size() < n || gc.size() < n || miv.size() < n * a.Dim() || iv.size() < n * a.Dim()) throw Error("set_flat: arrays shorter than gauss_off says");
        a.SetFlat(go.data(), w.data(), gc.data(), miv.data(), iv.data());
      }, py::arg("gauss_off"), py::arg("weights"), py::arg("gconsts"), py::arg("means_invvars"), py::arg("inv_vars"))
      // pickle: flat tuple of 3 * num_pdfs arrays (python/csrc/am-diag-gmm.cc:47-71)
      .def(py::pickle(
          [](const AmDiagGmm& a) {
            py::list out;
            for (auto& p : a.pdfs()) {
              out.append(Vec1(p->weights()));
              out.append(Vec2(p->inv_vars(), p->NumGauss(), p->Dim()));
              out.append(Vec2(p->means_invvars(), p->NumGauss(), p->Dim()));
            }
            return py::tuple(out);
          },
          [](py::tuple t) {
            auto a = std::make_shared<AmDiagGmm>();
            for (size_t i = 0; i + 2 < t.size(); i += 3) {
              Arr<float> w = t[i].cast<Arr<float>>(), iv = t[i + 1].cast<Arr<float>>(), miv = t[i + 2].cast<Arr<float>>();
              auto g = std::make_shared<DiagGmm>();
              g->SetRaw((int)w.shape(0), iv.ndim() == 2 ? (int)iv.shape(1) : 0, w.data(), iv.data(), miv.data(), nullptr);
              g->ComputeGconsts();
              a->pdfs().push_back(g);
            }
            return a;
          }));

  py::class_<MleDiagGmmOptions>(m, "MleDiagGmmOptions")
      .def(py::init([](float mgw, float mgo, double mv, bool rm, py::object vfv) {
             MleDiagGmmOptions o;
             o.min_gaussian_weight = mgw; o.min_gaussian_occupancy = mgo; o.min_variance = mv; o.remove_low_count_gaussians = rm;
             if (!vfv.is_none()) { Arr<double> v = vfv.cast<Arr<double>>(); o.variance_floor_vector.assign(v.data(), v.data() + v.size()); }
             return o;
           }), py::arg("min_gaussian_weight") = 1.0e-05f, py::arg("min_gaussian_occupancy") = 10.0f, py::arg("min_variance") = 0.001,
           py::arg("remove_low_count_gaussians") = true, py::arg("variance_floor_vector") = py::none())
      .def_readwrite("min_gaussian_weight", &MleDiagGmmOptions::min_gaussian_weight)
      .def_readwrite("min_gaussian_occupancy", &MleDiagGmmOptions::min_gaussian_occupancy)
      .def_readwrite("min_variance", &MleDiagGmmOptions::min_variance)
      .def_readwrite("remove_low_count_gaussians", &MleDiagGmmOptions::remove_low_count_gaussians)
      .def_property("variance_floor_vector",
                    [](MleDiagGmmOptions& o) -> py::object { return o.variance_floor_vector.empty() ? py::object(py::none()) : py::object(Vec1(o.variance_floor_vector)); },
                    [](MleDiagGmmOptions& o, py::object v) {
                      if (v.is_none()) { o.variance_floor_vector.clear(); return; }
                      Arr<double> a = v.cast<Arr<double>>();
                      o.variance_floor_vector.assign(a.data(), a.data() + a.size());
                    })
      .def("__str__", &MleDiagGmmOptions::ToString);

  py::class_<MapDiagGmmOptions>(m, "MapDiagGmmOptions")      // python/csrc/mle-diag-gmm.cc:40-58
      .def(py::init([](float mt, float vt, float wt) { MapDiagGmmOptions o; o.mean_tau = mt; o.variance_tau = vt; o.weight_tau = wt; return o; }),
           py::arg("mean_tau") = 10.0f, py::arg("variance_tau") = 50.0f, py::arg("weight_tau") = 10.0f)
      .def_readwrite("mean_tau", &MapDiagGmmOptions::mean_tau)
      .def_readwrite("variance_tau", &MapDiagGmmOptions::variance_tau)
      .def_readwrite("weight_tau", &MapDiagGmmOptions::weight_tau)
      .def("__str__", &MapDiagGmmOptions::ToString);

  py::class_<AccumDiagGmm, std::shared_ptr<AccumDiagGmm>>(m, "AccumDiagGmm")
      // python/csrc/mle-diag-gmm.cc:65-73: two constructors, resize(num_gauss, dim, flags); resize(gmm, flags) is this package's extra
      .def(py::init([]() { return std::make_shared<AccumDiagGmm>(); }))
      .def(py::init([](const DiagGmm& gmm, int flags) {
             auto a = std::make_shared<AccumDiagGmm>();
             a->Resize(gmm.NumGauss(), gmm.Dim(), flags);
             return a;
           }), py::arg("gmm"), py::arg("flags"))
      .def("resize", [](AccumDiagGmm& a, int num_gauss, int dim, int flags) { a.Resize(num_gauss, dim, flags); }, py::arg("num_gauss"), py::arg("dim"),
           py::arg("flags"))
      .def("resize", [](AccumDiagGmm& a, const DiagGmm& gmm, int flags) { a.Resize(gmm.NumGauss(), gmm.Dim(), flags); }, py::arg("gmm"), py::arg("flags"))
      .def_property_readonly("num_gauss", &AccumDiagGmm::NumGauss)
      .def_property_readonly("dim", &AccumDiagGmm::Dim)
      .def_property_readonly("flags", &AccumDiagGmm::Flags)
      .def_property_readonly("_flags", &AccumDiagGmm::Flags)
      .def_property_readonly("_dim", &AccumDiagGmm::Dim)
      // fp64 views of the accumulators (the reference's DoubleVector / DoubleMatrix): writable in place
      .def_property("occupancy", [](py::object self) { return View1(self.cast<AccumDiagGmm&>().occupancy(), self); },
                    [](AccumDiagGmm& a, Arr<double> v) {
                      if (v.size() != a.NumGauss()) throw Error("occupancy: size mismatch");
                      a.occupancy().assign(v.data(), v.data() + v.size());
                    })
      .def_property("mean_accumulator", [](py::object self) { auto& a = self.cast<AccumDiagGmm&>(); return View2(a.mean_accumulator(), a.NumGauss(), a.Dim(), self); },
                    [](AccumDiagGmm& a, Arr<double> v) {
                      if ((size_t)v.size() != a.mean_accumulator().size()) throw Error("mean_accumulator: size mismatch");
                      a.mean_accumulator().assign(v.data(), v.data() + v.size());
                    })
      .def_property("variance_accumulator", [](py::object self) { auto& a = self.cast<AccumDiagGmm&>(); return View2(a.variance_accumulator(), a.NumGauss(), a.Dim(), self); },
                    [](AccumDiagGmm& a, Arr<double> v) {
                      if ((size_t)v.size() != a.variance_accumulator().size()) throw Error("variance_accumulator: size mismatch");
                      a.variance_accumulator().assign(v.data(), v.data() + v.size());
                    })
      .def("set_zero", &AccumDiagGmm::SetZero, py::arg("flags"))
      .def("scale", &AccumDiagGmm::Scale, py::arg("f"), py::arg("flags"))
      .def("accumulate_for_component", [](AccumDiagGmm& a, Arr<float> x, int comp, float w) { a.AccumulateForComponent(x.data(), (size_t)x.size(), comp, w); },
           py::arg("data"), py::arg("comp_index"), py::arg("weight"))
      .def("accumulate_from_posteriors", [](AccumDiagGmm& a, Arr<float> x, Arr<float> post) {
        a.AccumulateFromPosteriors(x.data(), (size_t)x.size(), post.data(), (size_t)post.size());
      }, py::arg("data"), py::arg("gauss_posteriors"))
      .def("accumulate_from_diag", [](AccumDiagGmm& a, const DiagGmm& g, Arr<float> x, float w) { return NoGil([&] { return a.AccumulateFromDiag(g, x.data(), (size_t)x.size(), w); }); },
           py::arg("gmm"), py::arg("data"), py::arg("weight"))
      .def("add_stats_for_component", [](AccumDiagGmm& a, int g, double occ, Arr<double> x, Arr<double> x2) {
        a.AddStatsForComponent(g, occ, x.data(), (size_t)x.size(), x2.data(), (size_t)x2.size());
      }, py::arg("g"), py::arg("occ"), py::arg("x_stats"), py::arg("x2_stats"))
      .def("add", &AccumDiagGmm::Add, py::arg("scale"), py::arg("acc"))
      .def("smooth_stats", &AccumDiagGmm::SmoothStats, py::arg("tau"))
      .def("smooth_with_accum", &AccumDiagGmm::SmoothWithAccum, py::arg("tau"), py::arg("src_acc"))
      .def("smooth_with_model", &AccumDiagGmm::SmoothWithModel, py::arg("tau"), py::arg("src_gmm"))
      .def("copy", [](AccumDiagGmm& a) { return std::make_shared<AccumDiagGmm>(a); });

  m.def("mle_diag_gmm_update", [](py::object cfg, const AccumDiagGmm& acc, int flags, DiagGmm& gmm) {
          const MleDiagGmmOptions o = OptsFrom(cfg);
          return UpdateTuple(NoGil([&] { return MleDiagGmmUpdate(o, acc, flags, &gmm); }));
        },
        py::arg("config"), py::arg("diag_gmm_acc"), py::arg("flags"), py::arg("gmm"));
  m.def("ml_objective", &MlObjective, py::arg("gmm"), py::arg("diaggmm_acc"));
  m.def("map_diag_gmm_update", [](const MapDiagGmmOptions& cfg, const AccumDiagGmm& acc, int flags, DiagGmm& gmm) { return MapDiagGmmUpdate(cfg, acc, flags, &gmm); },
        py::arg("config"), py::arg("diag_gmm_acc"), py::arg("flags"), py::arg("gmm"));
  // the flat M-step both updates go through: -> (new_off, w, gc, miv, iv, objf_change, count, floored_elements, floored_gaussians, removed)
  m.def("flat_update", [](py::object opts, Arr<int32_t> go, Arr<double> occ, py::object mean_acc, py::object var_acc, int acc_flags, int flags, Arr<float> w,
                          Arr<float> miv, Arr<float> iv) {
    const int P = (int)go.size() - 1;
    if (P < 1 || miv.ndim() != 2) throw Error("flat_update: bad arguments");
    const int D = (int)miv.shape(1);
    std::vector<float> wv = FVec(w), mv = FVec(miv), ivv = FVec(iv), gc;
    std::vector<int32_t> new_off;
    Arr<double> ma, va;
    const double *map = nullptr, *vap = nullptr;
    if (!mean_acc.is_none()) { ma = mean_acc.cast<Arr<double>>(); if (ma.size()) map = ma.data(); }
    if (!var_acc.is_none()) { va = var_acc.cast<Arr<double>>(); if (va.size()) vap = va.data(); }
    const MleUpdateResult r = MleFlatUpdate(OptsFrom(opts), P, D, go.data(), occ.data(), map, vap, acc_flags, flags, &wv, &gc, &mv, &ivv, &new_off);
    const size_t n = (size_t)new_off[(size_t)P];
    wv.resize(n); gc.resize(n); mv.resize(n * D); ivv.resize(n * D);
    return py::make_tuple(Vec1(new_off), Vec1(wv), Vec1(gc), Vec2(mv, n, (size_t)D), Vec2(ivv, n, (size_t)D), r.objf_change, r.count, r.floored_elements,
                          r.floored_gaussians, r.removed);
  });

  py::class_<AccumAmDiagGmm, std::shared_ptr<AccumAmDiagGmm>>(m, "AccumAmDiagGmm")
      .def(py::init<>())
      // the two overloads of python/csrc/mle-am-diag-gmm.cc:18-23 (egs/yesno/train.py:111 calls init(model=, flags=))
      .def("init", [](AccumAmDiagGmm& a, const AmDiagGmm& model, int flags) { a.Init(model, -1, flags); }, py::arg("model"), py::arg("flags"))
      .def("init", [](AccumAmDiagGmm& a, const AmDiagGmm& model, int dim, int flags) {
        if (!(dim > 0)) throw Error("dim > 0 assertion failed");
        a.Init(model, dim, flags);
      }, py::arg("model"), py::arg("dim"), py::arg("flags"))
      .def("set_zero", &AccumAmDiagGmm::SetZero, py::arg("flags"))
      .def_property_readonly("num_accs", &AccumAmDiagGmm::NumAccs)
      .def_property_readonly("dim", &AccumAmDiagGmm::Dim)
      .def_property_readonly("tot_stats_count", &AccumAmDiagGmm::TotStatsCount)
      .def_property_readonly("tot_count", &AccumAmDiagGmm::TotCount)
      .def_property_readonly("tot_log_like", &AccumAmDiagGmm::TotLogLike)
      .def_property("_total_frames", &AccumAmDiagGmm::total_frames, &AccumAmDiagGmm::set_total_frames)
      .def_property("_total_log_like", &AccumAmDiagGmm::total_log_like, &AccumAmDiagGmm::set_total_log_like)
      // gmm_acc_stats_ali's device path (scripts/gmm_acc_stats_ali.py:46-58): K3 over (feats, ali) of one utterance into statistics that
      // stay on the device between calls; transition_accs[tid] += 1 per frame on the host, as TransitionModel.accumulate does
      // -> (log_like of these frames, transition_accs)
      .def("_acc_stats_ali", [](AccumAmDiagGmm& a, const AmDiagGmm& am, const TransitionModel& tm, Arr<float> feats, std::vector<int32_t> ali, py::object tacc) {
        if (feats.ndim() != 2 || (py::ssize_t)ali.size() != feats.shape(0)) throw Error("gmm_acc_stats_ali: feats must be 2-D and len(ali) == num_frames");
        if (feats.shape(0) > 0 && feats.shape(1) != am.Dim()) throw Error("Dim mismatch: data dim = " + std::to_string(feats.shape(1)) + " vs. model dim = " + std::to_string(am.Dim()));
        const int64_t fo[2] = {0, (int64_t)feats.shape(0)};
        const double ll = NoGil([&] { return a.AccumulateAli(am, tm, feats.data(), fo, 1, ali.data()); });
        const py::ssize_t nt = tm.NumTransitionIds() + 1;
        py::array_t<double> t;
        if (tacc.is_none()) { t = py::array_t<double>(nt); std::fill(t.mutable_data(), t.mutable_data() + nt, 0.0); }
        else {
          t = py::array_t<double>::ensure(tacc);
          if (!t || t.ndim() != 1 || t.shape(0) != nt) throw Error("transition_accs: one count per transition-id (+ entry 0)");
          if (!t.writeable() || !(t.flags() & py::array::c_style)) t = py::array_t<double, py::array::c_style>(t.attr("copy")());
        }
        double* tp = t.mutable_data();
        for (int32_t x : ali) tp[x] += 1.0;
        return py::make_tuple(ll, t);
      }, py::arg("am_gmm"), py::arg("transition_model"), py::arg("feats"), py::arg("ali"), py::arg("transition_accs") = py::none())
      .def("_flush_device_stats", [](AccumAmDiagGmm& a) { NoGil([&] { a.Flush(); return 0; }); })
      .def_property_readonly("_has_device_stats", &AccumAmDiagGmm::HasDeviceStats)
      .def("get_acc", [](AccumAmDiagGmm& a, int i) { return std::make_shared<AccumDiagGmm>(*a.Acc(i)); }, py::arg("index"))   // the binding returns a COPY
      .def_property_readonly("_accs", [](AccumAmDiagGmm& a) { return a.accs(); })
      .def("accumulate_for_gmm", [](AccumAmDiagGmm& a, const AmDiagGmm& model, Arr<float> x, int i, float w) {
        return NoGil([&] { return a.AccumulateForGmm(model, x.data(), (size_t)x.size(), i, w); });
      },
           py::arg("model"), py::arg("data"), py::arg("gmm_index"), py::arg("weight"))
      .def("accumulate_for_gmm_two_feats", [](AccumAmDiagGmm& a, const AmDiagGmm& model, Arr<float> x1, Arr<float> x2, int i, float w) {
        return NoGil([&] { return a.AccumulateForGmmTwoFeats(model, x1.data(), (size_t)x1.size(), x2.data(), (size_t)x2.size(), i, w); });
      }, py::arg("model"), py::arg("data1"), py::arg("data2"), py::arg("gmm_index"), py::arg("weight"))
      .def("accumulate_from_posteriors", [](AccumAmDiagGmm& a, const AmDiagGmm& model, Arr<float> x, int i, Arr<float> post) {
        a.AccumulateFromPosteriors(model, x.data(), (size_t)x.size(), i, post.data(), (size_t)post.size());
      }, py::arg("model"), py::arg("data"), py::arg("gmm_index"), py::arg("weight"))   // the reference names the posterior vector `weight` (mle-am-diag-gmm.cc:31-33)
      .def("accumulate_for_gaussian", [](AccumAmDiagGmm& a, const AmDiagGmm& am, Arr<float> x, int i, int g, float w) {
        a.AccumulateForGaussian(am, x.data(), (size_t)x.size(), i, g, w);
      }, py::arg("am"), py::arg("data"), py::arg("gmm_index"), py::arg("gauss_index"), py::arg("weight"))
      .def("add", &AccumAmDiagGmm::Add, py::arg("scale"), py::arg("other"))
      .def("scale", &AccumAmDiagGmm::Scale, py::arg("scale"))
      .def("add_device_stats", [](AccumAmDiagGmm& a, py::dict st, Arr<int32_t> go) {
        Arr<double> occ = st["occ"].cast<Arr<double>>(), ma = st["mean_acc"].cast<Arr<double>>(), va = st["var_acc"].cast<Arr<double>>();
        if (go.size() != a.NumAccs() + 1 || occ.size() != go.at(a.NumAccs()) || ma.ndim() != 2 || ma.size() != va.size() || ma.shape(0) != occ.size())
          throw Error("add_device_stats: the statistics do not match the accumulators' layout");
        a.AddDeviceStats(go.data(), occ.data(), ma.data(), va.data(), (int)ma.shape(1), st["total_frames"].cast<double>(), st["total_log_like"].cast<double>());
      }, py::arg("st"), py::arg("gauss_off"));

  m.def("map_am_diag_gmm_update", [](const MapDiagGmmOptions& cfg, const AccumAmDiagGmm& acc, int flags, AmDiagGmm& am) { return MapAmDiagGmmUpdate(cfg, acc, flags, &am); },
        py::arg("config"), py::arg("amdiag_gmm_acc"), py::arg("flags"), py::arg("am_gmm"));
  m.def("mle_am_diag_gmm_update", [](py::object cfg, const AccumAmDiagGmm& acc, int flags, AmDiagGmm& am) {
    const MleDiagGmmOptions o = OptsFrom(cfg);
    const MleUpdateResult r = NoGil([&] { return MleAmDiagGmmUpdate(o, acc, flags, &am); });
    return py::make_tuple(r.objf_change, r.count);
  }, py::arg("config"), py::arg("amdiag_gmm_acc"), py::arg("flags"), py::arg("am_gmm"));
}

}  // namespace

void BindHmm(py::module_& m);      // khg_py_hmm.cpp
void BindAlign(py::module_& m);    // khg_py_align.cpp

void BindHost(py::module_& m, py::object* error_class) {
  static py::object* err = error_class;
  py::register_exception_translator([](std::exception_ptr p) {
    try {
      if (p) std::rethrow_exception(p);
    } catch (const khg::Error& e) {
      if (err && *err && !err->is_none()) PyErr_SetString(err->ptr(), e.what());
      else PyErr_SetString(PyExc_RuntimeError, e.what());
    }
  });
  BindGmm(m);
  BindHmm(m);
  BindAlign(m);
}
