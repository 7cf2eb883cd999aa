// pybind11 bindings of HmmState / HmmTopology / TransitionModelTuple / TransitionModel / MleTransitionUpdateConfig
// (khg_host_hmm.hpp) with the names and pickle tuples of python/csrc/{hmm-topology,transition-model,transition-information}.cc
// in /root/reference/kaldi-hmm-gmm.  Stream I/O (text + Kaldi binary) stays in the Python shells over kaldi_io: they read the
// members through these accessors and hand a parsed object back through _set_state / _set_from_read.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

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
py::list FloatList(const std::vector<float>& v) {
  py::list l;
  for (float x : v) l.append(py::float_((double)x));
  return l;
}
std::vector<float> FloatVec(py::handle seq) {
  std::vector<float> v;
  for (py::handle x : seq) v.push_back((float)x.cast<double>());
  return v;
}
}  // namespace

void BindHmm(py::module_& m) {
  m.attr("kNoPdf") = kNoPdf;

  py::class_<HmmState>(m, "HmmState")
      .def(py::init([](int fwd, py::object sl, py::object transitions) {
             HmmState s(fwd, sl.is_none() ? fwd : sl.cast<int>());
             if (!transitions.is_none())
               for (py::handle t : transitions) {
                 py::tuple p = t.cast<py::tuple>();
                 s.transitions.emplace_back(p[0].cast<int>(), (float)p[1].cast<double>());
               }
             return s;
           }), py::arg("forward_pdf_class") = kNoPdf, py::arg("self_loop_pdf_class") = py::none(), py::arg("transitions") = py::none())
      .def_readwrite("forward_pdf_class", &HmmState::forward_pdf_class)
      .def_readwrite("self_loop_pdf_class", &HmmState::self_loop_pdf_class)
      .def_readwrite("transitions", &HmmState::transitions)
      .def("__eq__", [](const HmmState& a, py::object o) { return py::isinstance<HmmState>(o) && a == o.cast<const HmmState&>(); })
      .def("__str__", &HmmState::ToString)
      .def(py::pickle([](const HmmState& s) { return py::make_tuple(s.forward_pdf_class, s.self_loop_pdf_class, s.transitions); },
                      [](py::tuple t) {
                        HmmState s(t[0].cast<int>(), t[1].cast<int>());
                        s.transitions = t[2].cast<std::vector<std::pair<int, float>>>();
                        return s;
                      }));

  py::class_<HmmTopology, std::shared_ptr<HmmTopology>>(m, "HmmTopology")
      .def(py::init<>())
      .def("read", &HmmTopology::Read, py::arg("s"))
      .def("__str__", &HmmTopology::ToString)
      .def_property_readonly("phones", [](HmmTopology& t) { return t.phones(); })
      .def_property_readonly("is_hmm", &HmmTopology::IsHmm)
      .def_property_readonly("_phones", [](HmmTopology& t) { return t.phones(); })
      .def_property_readonly("_phone2idx", [](HmmTopology& t) { return t.phone2idx(); })
      .def_property_readonly("_entries", [](HmmTopology& t) { return t.entries(); })
      .def("_set_state", &HmmTopology::SetState, py::arg("phones"), py::arg("phone2idx"), py::arg("entries"))
      .def("topology_for_phone", [](HmmTopology& t, int ph) { return t.TopologyForPhone(ph); }, py::arg("phone"))
      .def("num_pdf_classes", &HmmTopology::NumPdfClasses, py::arg("phone"))
      .def("get_phone_to_num_pdf_classes", &HmmTopology::GetPhoneToNumPdfClasses)
      .def("min_length", &HmmTopology::MinLength, py::arg("phone"))
      .def("check", &HmmTopology::Check)
      // pickle: (phones, phone2idx, entries)  python/csrc/hmm-topology.cc:83-93
      .def(py::pickle([](HmmTopology& t) { return py::make_tuple(t.phones(), t.phone2idx(), t.entries()); },
                      [](py::tuple t) {
                        auto o = std::make_shared<HmmTopology>();
                        o->SetState(t[0].cast<std::vector<int>>(), t[1].cast<std::vector<int>>(), t[2].cast<std::vector<HmmTopology::Entry>>());
                        return o;
                      }));

  py::class_<MleTransitionUpdateConfig>(m, "MleTransitionUpdateConfig")
      .def(py::init([](float floor, float mincount, bool share) { MleTransitionUpdateConfig c; c.floor = floor; c.mincount = mincount; c.share_for_pdfs = share; return c; }),
           py::arg("floor") = 0.01f, py::arg("mincount") = 5.0f, py::arg("share_for_pdfs") = false)
      .def_readwrite("floor", &MleTransitionUpdateConfig::floor)
      .def_readwrite("mincount", &MleTransitionUpdateConfig::mincount)
      .def_readwrite("share_for_pdfs", &MleTransitionUpdateConfig::share_for_pdfs);

  py::class_<TransitionModelTuple>(m, "TransitionModelTuple")
      .def(py::init([]() { return TransitionModelTuple{0, 0, 0, 0}; }))
      .def(py::init([](int ph, int hs, int fp, int sp) { return TransitionModelTuple{ph, hs, fp, sp}; }), py::arg("phone"), py::arg("hmm_state"),
           py::arg("forward_pdf"), py::arg("self_loop_pdf"))
      .def_readwrite("phone", &TransitionModelTuple::phone)
      .def_readwrite("hmm_state", &TransitionModelTuple::hmm_state)
      .def_readwrite("forward_pdf", &TransitionModelTuple::forward_pdf)
      .def_readwrite("self_loop_pdf", &TransitionModelTuple::self_loop_pdf)
      .def("_key", [](const TransitionModelTuple& t) { return py::make_tuple(t.phone, t.hmm_state, t.forward_pdf, t.self_loop_pdf); })
      .def("__eq__", [](const TransitionModelTuple& a, py::object o) { return py::isinstance<TransitionModelTuple>(o) && a == o.cast<const TransitionModelTuple&>(); })
      .def("__lt__", [](const TransitionModelTuple& a, const TransitionModelTuple& b) { return a < b; })
      .def("__str__", &TransitionModelTuple::ToString)
      .def(py::pickle([](const TransitionModelTuple& t) { return py::make_tuple(t.phone, t.hmm_state, t.forward_pdf, t.self_loop_pdf); },
                      [](py::tuple t) { return TransitionModelTuple{t[0].cast<int>(), t[1].cast<int>(), t[2].cast<int>(), t[3].cast<int>()}; }));

  py::class_<TransitionInformation, std::shared_ptr<TransitionInformation>>(m, "TransitionInformation")      // python/csrc/transition-information.cc:11-27
      .def("transition_ids_equivalent", &TransitionInformation::TransitionIdsEquivalent, py::arg("trans_id1"), py::arg("trans_id2"))
      .def("transition_ids_is_start_of_phone", &TransitionInformation::TransitionIdIsStartOfPhone, py::arg("trans_id"))
      .def("transition_id_to_phone", &TransitionInformation::TransitionIdToPhone, py::arg("trans_id"))
      .def("is_final", &TransitionInformation::IsFinal, py::arg("trans_id"))
      .def("is_self_loop", &TransitionInformation::IsSelfLoop, py::arg("trans_id"))
      .def("transition_id_to_pdf", &TransitionInformation::TransitionIdToPdf, py::arg("trans_id"))
      .def("transition_id_to_pdf_array", [](TransitionInformation& t) { return t.TransitionIdToPdfArray(); })
      .def_property_readonly("num_transition_ids", &TransitionInformation::NumTransitionIds)
      .def_property_readonly("num_pdfs", &TransitionInformation::NumPdfs);

  py::class_<TransitionModel, TransitionInformation, std::shared_ptr<TransitionModel>>(m, "TransitionModel")
      .def(py::init([]() { return std::make_shared<TransitionModel>(); }))
      .def(py::init([](py::object ctx_dep, py::object hmm_topo) {       // (ctx_dep = None: an empty model over hmm_topo, what Read fills)
             if (ctx_dep.is_none()) {
               auto tm = std::make_shared<TransitionModel>();
               if (!hmm_topo.is_none())
                 tm->SetState({}, hmm_topo.cast<std::shared_ptr<HmmTopology>>(), {}, {}, {}, 0, {}, {});
               return tm;
             }
             auto topo = hmm_topo.cast<std::shared_ptr<HmmTopology>>();
             if (!topo) throw Error("TransitionModel: hmm_topo is required with ctx_dep");
             const std::vector<int>& phones = topo->phones();
             if (phones.empty()) throw Error("TransitionModel: empty topology");
             std::vector<int> npc((size_t)phones.back() + 1, -1);
             for (int ph : phones) npc[(size_t)ph] = topo->NumPdfClasses(ph);
             // ContextDependency::GetPdfInfo (csrc/context-dep.cc): pdf -> [(phone, pdf_class)]
             auto info = ctx_dep.attr("get_pdf_info")(phones, npc).cast<std::vector<std::vector<std::pair<int, int>>>>();
             return std::make_shared<TransitionModel>(info, topo);
           }), py::arg("ctx_dep"), py::arg("hmm_topo"))
      .def("check", &TransitionModel::Check)
      .def_property_readonly("num_transition_ids", &TransitionModel::NumTransitionIds)
      .def_property_readonly("num_transition_states", &TransitionModel::NumTransitionStates)
      .def_property_readonly("num_pdfs", &TransitionModel::NumPdfs)
      .def_property_readonly("topo", [](TransitionModel& t) { return t.topo(); })
      .def_property_readonly("phones", [](TransitionModel& t) { return t.topo()->phones(); })
      .def_property_readonly("tuples", [](TransitionModel& t) { return t.tuples(); })
      .def_property_readonly("state2id", [](TransitionModel& t) { return t.state2id(); })
      .def_property_readonly("id2state", [](TransitionModel& t) { return t.id2state(); })
      .def_property_readonly("id2pdf_id", [](TransitionModel& t) { return t.id2pdf(); })
      .def_property_readonly("log_probs", [](TransitionModel& t) { return FloatList(t.log_probs()); })
      .def_property_readonly("non_self_loop_log_probs", [](TransitionModel& t) { return FloatList(t.non_self_loop_log_probs()); })
      // the members under the names the Python class kept them in
      .def_property_readonly("_tuples", [](TransitionModel& t) { return t.tuples(); })
      .def_property_readonly("_topo", [](TransitionModel& t) { return t.topo(); })
      .def_property_readonly("_state2id", [](TransitionModel& t) { return t.state2id(); })
      .def_property_readonly("_id2state", [](TransitionModel& t) { return t.id2state(); })
      .def_property_readonly("_id2pdf", [](TransitionModel& t) { return t.id2pdf(); })
      .def_property_readonly("_num_pdfs", &TransitionModel::NumPdfs)
      .def_property_readonly("_log_probs", [](TransitionModel& t) { return Vec1(t.log_probs()); })
      .def_property_readonly("_nsl", [](TransitionModel& t) { return Vec1(t.non_self_loop_log_probs()); })
      .def("_set_from_read", [](TransitionModel& t, std::shared_ptr<HmmTopology> topo, std::vector<TransitionModelTuple> tuples, Arr<float> lp) {
        t.SetFromRead(std::move(topo), std::move(tuples), std::vector<float>(lp.data(), lp.data() + lp.size()));
      }, py::arg("topo"), py::arg("tuples"), py::arg("log_probs"))
      .def("transition_id_to_pdf", &TransitionModel::TransitionIdToPdf, py::arg("trans_id"))
      .def("transition_id_to_pdf_array", [](TransitionModel& t) { return t.id2pdf(); })
      .def("transition_id_to_phone", &TransitionModel::TransitionIdToPhone, py::arg("trans_id"))
      .def("transition_id_to_hmm_state", &TransitionModel::TransitionIdToHmmState, py::arg("trans_id"))
      .def("transition_ids_equivalent", &TransitionModel::TransitionIdsEquivalent)
      .def("transition_ids_is_start_of_phone", &TransitionModel::TransitionIdIsStartOfPhone, py::arg("trans_id"))
      .def("is_self_loop", &TransitionModel::IsSelfLoop, py::arg("trans_id"))
      .def("_is_self_loop_raw", &TransitionModel::IsSelfLoopRaw)
      .def("is_final", &TransitionModel::IsFinal, py::arg("trans_id"))
      .def("self_loop_of", &TransitionModel::SelfLoopOf, py::arg("trans_state"))
      .def("get_transition_log_prob", [](TransitionModel& t, int tid) { return (double)t.GetTransitionLogProb(tid); }, py::arg("trans_id"))
      .def("tuple_to_transition_state", &TransitionModel::TupleToTransitionState, py::arg("phone"), py::arg("hmm_state"), py::arg("pdf"), py::arg("self_loop_pdf"))
      .def("pair_to_transition_id", &TransitionModel::PairToTransitionId, py::arg("trans_state"), py::arg("trans_index"))
      .def("transition_id_to_transition_state", &TransitionModel::TransitionIdToTransitionState, py::arg("trans_id"))
      .def("get_non_self_loop_log_prob", [](TransitionModel& t, int ts) { return (double)t.GetNonSelfLoopLogProb(ts); }, py::arg("trans_state"))
      .def("get_transition_log_prob_ignoring_self_loops", [](TransitionModel& t, int tid) { return (double)t.GetTransitionLogProbIgnoringSelfLoops(tid); },
           py::arg("trans_id"))
      // statistics (csrc/transition-model.h:176-189)
      .def("init_stats", [](TransitionModel& t) {
        Arr<double> a({(py::ssize_t)t.NumTransitionIds() + 1});
        std::memset(a.mutable_data(), 0, sizeof(double) * (size_t)a.size());
        return a;
      })
      .def("accumulate", [](TransitionModel& t, double prob, int tid, py::object stats) {
        t.ChkTid(tid);
        py::array_t<double> a = py::array_t<double>::ensure(stats);      // float64 arrays are updated in place, like the reference's DoubleVector&
        if (!a || a.ndim() != 1 || a.shape(0) != t.NumTransitionIds() + 1) throw Error("stats.size() == NumTransitionIds() + 1 assertion failed");
        a.mutable_at(tid) += prob;
        return a;
      }, py::arg("prob"), py::arg("trans_id"), py::arg("stats"))
      .def("mle_update", [](TransitionModel& t, Arr<double> stats, py::object cfg) {
        MleTransitionUpdateConfig c;
        if (!cfg.is_none()) {
          if (py::isinstance<MleTransitionUpdateConfig>(cfg)) c = cfg.cast<MleTransitionUpdateConfig>();
          else { c.floor = cfg.attr("floor").cast<float>(); c.mincount = cfg.attr("mincount").cast<float>(); c.share_for_pdfs = cfg.attr("share_for_pdfs").cast<bool>(); }
        }
        auto r = t.MleUpdate(stats.data(), (size_t)stats.size(), c);
        return py::make_tuple(r.first, r.second);
      }, py::arg("stats"), py::arg("cfg") = py::none())
      .def("is_self_loop_array", [](TransitionModel& t) { return Vec1(t.IsSelfLoopArray()); })
      .def("scaled_trans_cost", [](TransitionModel& t, float ts, float sl) { return Vec1(t.ScaledTransCost(ts, sl)); }, py::arg("transition_scale"),
           py::arg("self_loop_scale"))
      .def("__str__", &TransitionModel::ToString)
      // pickle: 8-tuple, python/csrc/transition-model.cc:122-150
      .def(py::pickle(
          [](TransitionModel& t) {
            return py::make_tuple(t.tuples(), t.topo(), t.state2id(), t.id2state(), t.id2pdf(), t.NumPdfs(), FloatList(t.log_probs()),
                                  FloatList(t.non_self_loop_log_probs()));
          },
          [](py::tuple t) {
            auto tm = std::make_shared<TransitionModel>();
            tm->SetState(t[0].cast<std::vector<TransitionModelTuple>>(), t[1].cast<std::shared_ptr<HmmTopology>>(), t[2].cast<std::vector<int>>(),
                         t[3].cast<std::vector<int>>(), t[4].cast<std::vector<int>>(), t[5].cast<int>(), FloatVec(t[6]), FloatVec(t[7]));
            return tm;
          }));

  m.def("get_pdfs_for_phones", [](const TransitionModel& tm, std::vector<int> phones) {
    std::vector<int> pdfs;
    const bool ok = GetPdfsForPhones(tm, phones, &pdfs);
    return py::make_tuple(ok, pdfs);
  }, py::arg("trans_model"), py::arg("phones"));
}
