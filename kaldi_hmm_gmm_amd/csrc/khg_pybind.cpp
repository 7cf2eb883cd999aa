// kaldi_hmm_gmm_amd/csrc/khg_pybind.cpp -- the pybind11 host surface over the C-ABI (include/khg_hip.h).
//
// The reference's host boundary is ONE pybind11 module, `_kaldi_hmm_gmm` (python/csrc/kaldi-hmm-gmm.cc:35-69), whose
// classes wrap Eigen-backed C++ objects.  This module, `_kaldi_hmm_gmm_amd`, is its counterpart for the accelerated
// path: C++ classes that own the C-ABI handles (device model, transition tables, resident utterance sets, accumulator
// block, RCCL communicator) and take / return numpy arrays, plus the host-side functions of the M-step, plus (BindHost,
// khg_py_host.cpp / khg_py_hmm.cpp / khg_py_align.cpp) the C++ host classes behind the reference's names: DiagGmm, AmDiagGmm,
// AccumDiagGmm, AccumAmDiagGmm, HmmTopology, TransitionModel, AlignConfig, DecodableAmDiagGmmScaled, align_utterance_wrapper ...
// kaldi_hmm_gmm_amd/*.py re-export them.
// Errors: a non-zero C-ABI status becomes a Python RuntimeError subclass (KhgError), as KHG_ERR does in the reference
// (csrc/log.h:46-53 -> std::runtime_error -> RuntimeError).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/khg_hip.h"

namespace py = pybind11;

namespace {

py::object g_khg_error;   // kaldi_hmm_gmm_amd._lib.KhgError (set at import)

void Check(int rc) {
  if (rc == KHG_OK) return;
  const char* msg = khg_last_error();
  if (g_khg_error && !g_khg_error.is_none()) {
    PyErr_SetString(g_khg_error.ptr(), msg ? msg : "khg error");
    throw py::error_already_set();
  }
  throw std::runtime_error(msg ? msg : "khg error");
}

// The C-ABI calls that launch kernels, wait for the stream or copy between host and device run WITHOUT the GIL (two contexts on
// two streams driven from two Python threads is a supported mode); Check() needs it back: it may raise.
template <class F>
int NoGil(F&& f) {
  py::gil_scoped_release release;
  return f();
}

template <class T>
using Arr = py::array_t<T, py::array::c_style | py::array::forcecast>;

template <class T>
Arr<T> Zeros(std::vector<py::ssize_t> shape) {
  Arr<T> a(shape);
  std::memset(a.mutable_data(), 0, sizeof(T) * (size_t)a.size());
  return a;
}

struct KContext {
  khg_ctx* h = nullptr;
  int device = 0;
  KContext(int dev, py::object stream) : device(dev) {
    void* st = stream.is_none() ? nullptr : reinterpret_cast<void*>(stream.cast<uintptr_t>());
    Check(khg_ctx_create(dev, st, &h));
  }
  ~KContext() { close(); }
  void close() { if (h) { khg_ctx_destroy(h); h = nullptr; } }
  void sync() { Check(NoGil([&] { return khg_ctx_sync(h); })); }
  void set_timing(bool on) { Check(khg_ctx_set_timing(h, on ? 1 : 0)); }
  void set_k1_form(const std::string& f) {
    int v = f == "auto" ? KHG_K1_AUTO : (f == "pdf" || f == "fp32") ? KHG_K1_FP32_PDF : f == "utt" ? KHG_K1_FP32_UTT : f == "f16x2" ? KHG_K1_F16X2 : f == "f16x2s" ? KHG_K1_F16X2S : -1;
    if (v < 0) throw py::key_error(f);
    Check(khg_ctx_set_k1_form(h, v));
  }
  // khg_ctx_set_option by name ("k3_form", ...) or by number; -> the previous value
  static int opt_id(const py::object& o) {
    if (py::isinstance<py::int_>(o)) return o.cast<int>();
    static const char* names[KHG_OPT_COUNT] = {"k1_form", "k1_order", "k1_nf", "k1p_ts", "k1_interleave", "k1_dbg", "k2_inorder", "k2_ks", "k2_serial",
                                               "k2_prof", "k3_bucket", "k3_form", "k3_phase_b", "k3_ny", "debug", "k3_phase_a", "k2_split"};
    const std::string n = o.cast<std::string>();
    for (int i = 0; i < KHG_OPT_COUNT; ++i) if (n == names[i]) return i;
    if (n == "scratch_bytes") return KHG_INFO_SCRATCH_BYTES;       // read-only
    if (n == "scratch_blocks") return KHG_INFO_SCRATCH_BLOCKS;
    throw py::key_error(n);
  }
  int get_option(py::object name) { int v = 0; Check(khg_ctx_get_option(h, opt_id(name), &v)); return v; }
  int set_option(py::object name, int value) {
    const int id = opt_id(name);
    int old = 0;
    Check(khg_ctx_get_option(h, id, &old));
    Check(khg_ctx_set_option(h, id, value));
    return old;
  }
  py::list timings() {
    const int cap = 4096;
    std::vector<char> names(1 << 16);
    std::vector<float> ms(cap);
    int32_t n = 0;
    Check(khg_ctx_get_timings(h, names.data(), (int64_t)names.size(), ms.data(), cap, &n));
    py::list out;
    const char* p = names.data();
    for (int i = 0; i < n && i < cap; ++i) {
      const char* e = std::strchr(p, '\n');
      std::string nm = e ? std::string(p, e) : std::string(p);
      out.append(py::make_tuple(nm, ms[i]));
      if (!e) break;
      p = e + 1;
    }
    return out;
  }
};

struct KComm {
  void* h = nullptr;
  int nranks = 1, rank = 0;
  static py::bytes unique_id() {
    char id[KHG_COMM_ID_BYTES];
    Check(khg_comm_unique_id(id));
    return py::bytes(id, KHG_COMM_ID_BYTES);
  }
  KComm(KContext& ctx, int n, int r, py::bytes uid) : nranks(n), rank(r) {
    std::string s = uid;
    if (s.size() != KHG_COMM_ID_BYTES) throw py::value_error("Comm: the id is 128 bytes");
    Check(NoGil([&] { return khg_comm_create(ctx.h, n, r, s.data(), &h); }));      // collective: blocks until every rank has called it
  }
  py::dict info() {          // what RCCL itself reports: ncclCommCount / ncclCommUserRank / ncclGetVersion
    int32_t n = 0, r = -1, v = 0;
    Check(khg_comm_info(h, &n, &r, &v));
    py::dict d;
    d["nranks"] = n; d["rank"] = r; d["version"] = v;
    return d;
  }
  ~KComm() { close(); }
  void close() { if (h) { khg_comm_destroy(h); h = nullptr; } }
};

struct KAccs;

struct KModel {
  khg_model* h = nullptr;
  py::object ctx_obj;
  KContext* ctx;
  int num_pdfs = 0, dim = 0;
  Arr<int32_t> gauss_off;
  KModel(py::object ctx_o, Arr<int32_t> go, Arr<float> gc, Arr<float> miv, Arr<float> iv, py::object weights)
      : ctx_obj(ctx_o), ctx(ctx_o.cast<KContext*>()), gauss_off(go) {
    num_pdfs = (int)go.shape(0) - 1;
    if (miv.ndim() != 2 || iv.ndim() != 2 || miv.shape(0) != iv.shape(0) || miv.shape(1) != iv.shape(1) || num_pdfs < 1 ||
        miv.shape(0) != go.at(num_pdfs) || gc.shape(0) != miv.shape(0))
      throw py::value_error("DeviceModel: inconsistent shapes");
    dim = (int)miv.shape(1);
    Check(khg_model_create(ctx->h, num_pdfs, dim, go.data(), gc.data(), miv.data(), iv.data(), &h));
    if (!weights.is_none()) set_weights(weights.cast<Arr<float>>());
  }
  ~KModel() { close(); }
  void close() { if (h) { khg_model_destroy(h); h = nullptr; } }
  int64_t sumG() const { return gauss_off.at(num_pdfs); }
  void set_weights(Arr<float> w) {
    if (w.shape(0) != sumG()) throw py::value_error("set_weights: one weight per Gaussian");
    Check(khg_model_set_weights(ctx->h, h, w.data()));
  }
  py::dict mle_update(KAccs& accs, py::object opts, int flags);
  py::dict mle_result(float oc, float cnt, int32_t fe, int32_t fg, int32_t rm);
  py::dict mle_update_sharded(KAccs& accs, py::object opts, int flags, py::object comm);
  void mle_update_range(KAccs& accs, py::object opts, int flags, int first_pdf, int n_pdf);
  py::dict mle_rows_download(int first_pdf, int n_pdf);
  void mle_rows_upload(py::dict d);
  py::dict mle_update_finish();
  void invalidate() { Check(khg_model_invalidate(h)); }
  void scale_weights(Arr<int32_t> pdfs, float scale) { Check(khg_model_scale_weights(ctx->h, h, (int32_t)pdfs.shape(0), pdfs.data(), scale)); }
  void split(Arr<int32_t> targets, float perturb, py::object randn) {
    if (targets.shape(0) != num_pdfs) throw py::value_error("split: one target per pdf");
    Arr<float> r;
    const float* rp = nullptr;
    int64_t nr = 0;
    if (!randn.is_none()) { r = randn.cast<Arr<float>>(); rp = r.data(); nr = (int64_t)r.size(); }
    Check(NoGil([&] { return khg_model_split(ctx->h, h, targets.data(), perturb, rp, nr); }));
    Arr<int32_t> go({(py::ssize_t)num_pdfs + 1});
    Check(khg_model_num_gauss(h, nullptr, go.mutable_data()));
    gauss_off = go;
  }
  void merge(Arr<int32_t> targets) {
    if (targets.shape(0) != num_pdfs) throw py::value_error("merge: one target per pdf");
    Check(NoGil([&] { return khg_model_merge(ctx->h, h, targets.data()); }));
    Arr<int32_t> go({(py::ssize_t)num_pdfs + 1});
    Check(khg_model_num_gauss(h, nullptr, go.mutable_data()));
    gauss_off = go;
  }
  py::dict download(bool weights) {
    const py::ssize_t G = sumG();
    Arr<float> gc({G}), miv({G, (py::ssize_t)dim}), iv({G, (py::ssize_t)dim});
    py::object w = py::none();
    float* wp = nullptr;
    Arr<float> wa;
    if (weights) { wa = Arr<float>({G}); wp = wa.mutable_data(); w = wa; }
    Check(NoGil([&] { return khg_model_download(ctx->h, h, wp, gc.mutable_data(), miv.mutable_data(), iv.mutable_data()); }));
    py::dict d;
    d["gauss_off"] = py::array(gauss_off).attr("copy")();
    d["weights"] = w; d["gconsts"] = gc; d["means_invvars"] = miv; d["inv_vars"] = iv;
    return d;
  }
};

struct KTransitions {
  khg_tm* h = nullptr;
  py::object ctx_obj;
  Arr<int32_t> id2pdf;
  int num_tids = 0;
  KTransitions(py::object ctx_o, Arr<int32_t> i2p) : ctx_obj(ctx_o), id2pdf(i2p) {
    num_tids = (int)i2p.shape(0) - 1;
    Check(khg_tm_create(ctx_o.cast<KContext*>()->h, num_tids, i2p.data(), &h));
  }
  ~KTransitions() { close(); }
  void close() { if (h) { khg_tm_destroy(h); h = nullptr; } }
  void set_trans_cost(py::object cost) {
    if (cost.is_none()) { Check(khg_tm_set_trans_cost(h, nullptr)); return; }
    Arr<float> c = cost.cast<Arr<float>>();
    if (c.shape(0) != num_tids + 1) throw py::value_error("set_trans_cost: num_tids + 1 entries");
    Check(khg_tm_set_trans_cost(h, c.data()));
  }
};

struct KAccs {
  khg_accs* h = nullptr;
  py::object ctx_obj;
  KContext* ctx;
  int64_t sumG = 0, size = 0;
  int dim = 0, num_tids = 0;
  KAccs(py::object ctx_o, KModel& m, KTransitions& tm) : ctx_obj(ctx_o), ctx(ctx_o.cast<KContext*>()) {
    sumG = m.sumG(); dim = m.dim; num_tids = tm.num_tids;
    Check(khg_accs_create(ctx->h, m.h, tm.h, &h));
    Check(khg_accs_size(h, &size));
  }
  ~KAccs() { close(); }
  void close() { if (h) { khg_accs_destroy(h); h = nullptr; } }
  void zero() { Check(khg_accs_zero(ctx->h, h)); }
  uintptr_t device_ptr() { void* p = nullptr; Check(khg_accs_device_ptr(h, &p)); return reinterpret_cast<uintptr_t>(p); }
  void allreduce_range(KModel& m, int first_pdf, int n_pdf, py::object comm) {
    void* c = comm.is_none() ? nullptr : comm.cast<KComm*>()->h;
    Check(NoGil([&] { return khg_accs_allreduce_range(ctx->h, h, m.h, first_pdf, n_pdf, c); }));
  }
  void allreduce(py::object comm, bool wire_fp32) {
    void* c = comm.is_none() ? nullptr : comm.cast<KComm*>()->h;
    Check(NoGil([&] { return wire_fp32 ? khg_accs_allreduce_f32(ctx->h, h, c) : khg_accs_allreduce(ctx->h, h, c); }));
  }
  py::dict split(Arr<double> buf) {
    const int64_t G = sumG, D = dim, nt = num_tids;
    if (buf.size() < G * (1 + 2 * D) + nt + 1 + 8) throw py::value_error("split: buffer too small");
    py::array base = buf;
    auto sl = [&](int64_t first, std::vector<py::ssize_t> shape) {
      std::vector<py::ssize_t> strides(shape.size());
      py::ssize_t st = sizeof(double);
      for (int i = (int)shape.size() - 1; i >= 0; --i) { strides[i] = st; st *= shape[i]; }
      return py::array(py::dtype::of<double>(), shape, strides, buf.data() + first, base);   // a view, like the ctypes twin
    };
    py::dict d;
    int64_t o = 0;
    d["occ"] = sl(o, {G}); o += G;
    d["mean_acc"] = sl(o, {G, D}); o += G * D;
    d["var_acc"] = sl(o, {G, D}); o += G * D;
    d["trans_acc"] = sl(o, {nt + 1}); o += nt + 1;
    d["total_frames"] = buf.data()[o];
    d["total_log_like"] = buf.data()[o + 1];
    return d;
  }
  void relayout(KModel& m) {
    Check(khg_accs_relayout(ctx->h, h, m.h));
    sumG = m.sumG();
    Check(khg_accs_size(h, &size));
  }
  Arr<double> download_range(int64_t first, int64_t count) {
    Arr<double> out({(py::ssize_t)count});
    Check(NoGil([&] { return khg_accs_download_range(ctx->h, h, first, count, out.mutable_data()); }));
    return out;
  }
  py::dict download_trans() {
    Arr<double> tr({(py::ssize_t)num_tids + 1});
    double sc[8];
    Check(NoGil([&] { return khg_accs_download_trans(ctx->h, h, tr.mutable_data(), sc); }));
    py::dict d;
    d["trans_acc"] = tr; d["total_frames"] = sc[0]; d["total_log_like"] = sc[1];
    return d;
  }
  py::dict download() {
    Arr<double> buf({(py::ssize_t)size});
    Check(NoGil([&] { return khg_accs_download(ctx->h, h, buf.mutable_data()); }));
    return split(buf);
  }
  void upload(Arr<double> b) {
    if (b.size() != size) throw py::value_error("upload: wrong block size");
    Check(NoGil([&] { return khg_accs_upload(ctx->h, h, b.data()); }));
  }
};

static void ParseMleOptions(py::object opts, int dim, khg_mle_options& o, Arr<double>& vfv) {
  khg_mle_options_default(&o);
  if (!opts.is_none()) {
    o.min_gaussian_weight = opts.attr("min_gaussian_weight").cast<float>();
    o.min_gaussian_occupancy = opts.attr("min_gaussian_occupancy").cast<float>();
    o.min_variance = opts.attr("min_variance").cast<double>();
    o.remove_low_count_gaussians = opts.attr("remove_low_count_gaussians").cast<bool>() ? 1 : 0;
    if (py::hasattr(opts, "variance_floor_vector") && !opts.attr("variance_floor_vector").is_none()) {
      vfv = opts.attr("variance_floor_vector").cast<Arr<double>>();
      if (vfv.size() > 0) {
        if (vfv.size() != dim) throw py::value_error("variance_floor_vector: one floor per dimension");
        o.variance_floor_vector = vfv.data();
      }
    }
  }
}
py::dict KModel::mle_result(float oc, float cnt, int32_t fe, int32_t fg, int32_t rm) {
  if (rm) {
    Arr<int32_t> go({(py::ssize_t)num_pdfs + 1});
    Check(khg_model_num_gauss(h, nullptr, go.mutable_data()));
    gauss_off = go;
  }
  py::dict d;
  d["objf_change"] = oc; d["count"] = cnt; d["floored_elements"] = fe; d["floored_gaussians"] = fg; d["removed"] = rm;
  return d;
}
py::dict KModel::mle_update(KAccs& accs, py::object opts, int flags) {
  khg_mle_options o;
  Arr<double> vfv;                                    // keeps the floor vector alive for the call
  ParseMleOptions(opts, dim, o, vfv);
  float oc = 0, cnt = 0;
  int32_t fe = 0, fg = 0, rm = 0;
  Check(NoGil([&] { return khg_model_mle_update(ctx->h, h, accs.h, &o, (uint16_t)(flags & 0xFFFF), &oc, &cnt, &fe, &fg, &rm); }));
  return mle_result(oc, cnt, fe, fg, rm);
}
// the sharded M-step (khg_model_mle_update_sharded) and its pieces
py::dict KModel::mle_update_sharded(KAccs& accs, py::object opts, int flags, py::object comm) {
  khg_mle_options o;
  Arr<double> vfv;
  ParseMleOptions(opts, dim, o, vfv);
  KComm* c = comm.is_none() ? nullptr : comm.cast<KComm*>();
  float oc = 0, cnt = 0;
  int32_t fe = 0, fg = 0, rm = 0;
  Check(NoGil([&] { return khg_model_mle_update_sharded(ctx->h, h, accs.h, &o, (uint16_t)(flags & 0xFFFF), c ? c->h : nullptr, c ? c->nranks : 1,
                                                        c ? c->rank : 0, &oc, &cnt, &fe, &fg, &rm); }));
  return mle_result(oc, cnt, fe, fg, rm);
}
void KModel::mle_update_range(KAccs& accs, py::object opts, int flags, int first_pdf, int n_pdf) {
  khg_mle_options o;
  Arr<double> vfv;
  ParseMleOptions(opts, dim, o, vfv);
  Check(NoGil([&] { return khg_model_mle_update_range(ctx->h, h, accs.h, &o, (uint16_t)(flags & 0xFFFF), first_pdf, n_pdf); }));
}
py::dict KModel::mle_rows_download(int first_pdf, int n_pdf) {
  if (first_pdf < 0 || n_pdf < 0 || first_pdf + n_pdf > num_pdfs) throw py::value_error("mle_rows_download: pdf range outside the model");
  const py::ssize_t ng = gauss_off.at(first_pdf + n_pdf) - gauss_off.at(first_pdf);
  Arr<float> w({ng}), gc({ng}), miv({ng, (py::ssize_t)dim}), iv({ng, (py::ssize_t)dim});
  py::array_t<uint8_t> res({(py::ssize_t)n_pdf, (py::ssize_t)32});
  Check(NoGil([&] { return khg_model_mle_rows_download(ctx->h, h, first_pdf, n_pdf, w.mutable_data(), gc.mutable_data(), miv.mutable_data(),
                                                       iv.mutable_data(), res.mutable_data()); }));
  py::dict d;
  d["first_pdf"] = first_pdf; d["n_pdf"] = n_pdf; d["weights"] = w; d["gconsts"] = gc; d["means_invvars"] = miv; d["inv_vars"] = iv; d["results"] = res;
  return d;
}
void KModel::mle_rows_upload(py::dict d) {
  const int first_pdf = d["first_pdf"].cast<int>(), n_pdf = d["n_pdf"].cast<int>();
  if (first_pdf < 0 || n_pdf < 0 || first_pdf + n_pdf > num_pdfs) throw py::value_error("mle_rows_upload: pdf range outside the model");
  const py::ssize_t ng = gauss_off.at(first_pdf + n_pdf) - gauss_off.at(first_pdf);
  Arr<float> w = d["weights"].cast<Arr<float>>(), gc = d["gconsts"].cast<Arr<float>>(), miv = d["means_invvars"].cast<Arr<float>>(),
             iv = d["inv_vars"].cast<Arr<float>>();
  py::array_t<uint8_t, py::array::c_style | py::array::forcecast> res = d["results"].cast<py::array_t<uint8_t, py::array::c_style | py::array::forcecast>>();
  if (w.size() != ng || gc.size() != ng || miv.size() != ng * dim || iv.size() != ng * dim || res.size() != (py::ssize_t)n_pdf * 32)
    throw py::value_error("mle_rows_upload: array sizes do not match the pdf range");
  Check(NoGil([&] { return khg_model_mle_rows_upload(ctx->h, h, first_pdf, n_pdf, w.data(), gc.data(), miv.data(), iv.data(), res.data()); }));
}
py::dict KModel::mle_update_finish() {
  float oc = 0, cnt = 0;
  int32_t fe = 0, fg = 0, rm = 0;
  Check(NoGil([&] { return khg_model_mle_update_finish(ctx->h, h, &oc, &cnt, &fe, &fg, &rm); }));
  return mle_result(oc, cnt, fe, fg, rm);
}

struct KUtts {
  khg_utts* h = nullptr;
  py::object ctx_obj, keep, keep_model;
  KContext* ctx;
  Arr<int64_t> frame_off;
  int n_utt = 0, dim = 0;
  KUtts(py::object ctx_o, py::object tm, Arr<int64_t> fo, py::object feats, py::object dim_o, py::object graphs)
      : ctx_obj(ctx_o), ctx(ctx_o.cast<KContext*>()), frame_off(fo) {
    n_utt = (int)fo.shape(0) - 1;
    const float* feats_h = nullptr;
    const float* feats_d = nullptr;
    Arr<float> fh;
    if (py::isinstance<py::tuple>(feats)) {        // (device pointer, keep-alive object): features already in HBM
      py::tuple t = feats;
      feats_d = reinterpret_cast<const float*>(t[0].cast<uintptr_t>());
      keep = t[1];
      if (dim_o.is_none()) throw py::value_error("UtteranceSet: dim is required with device features");
      dim = dim_o.cast<int>();
    } else {
      fh = feats.cast<Arr<float>>();
      if (fh.ndim() != 2 || fh.shape(0) != fo.at(n_utt)) throw py::value_error("UtteranceSet: feats must be [frames, dim]");
      dim = (int)fh.shape(1);
      feats_h = fh.data();
    }
    const khg_tm* tmh = tm.is_none() ? nullptr : tm.cast<KTransitions*>()->h;
    if (graphs.is_none()) {
      Check(khg_utts_create(ctx->h, tmh, n_utt, dim, fo.data(), feats_h, feats_d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            nullptr, nullptr, &h));
    } else {
      py::dict g = graphs;
      auto so = g["state_off"].cast<Arr<int64_t>>(); auto st = g["start"].cast<Arr<int32_t>>();
      auto ao = g["arc_off"].cast<Arr<int64_t>>(); auto il = g["ilabel"].cast<Arr<int32_t>>();
      auto ol = g["olabel"].cast<Arr<int32_t>>(); auto w = g["weight"].cast<Arr<float>>();
      auto ns = g["nextstate"].cast<Arr<int32_t>>(); auto fin = g["final"].cast<Arr<float>>();
      Check(khg_utts_create(ctx->h, tmh, n_utt, dim, fo.data(), feats_h, feats_d, so.data(), st.data(), ao.data(), il.data(), ol.data(),
                            w.data(), ns.data(), fin.data(), &h));
    }
  }
  ~KUtts() { close(); }
  void close() { if (h) { khg_utts_destroy(h); h = nullptr; } }
  void set_pdf_list(Arr<int32_t> p) { Check(khg_utts_set_pdf_list(h, (int32_t)p.shape(0), p.data())); }
  void features_changed() { Check(khg_utts_features_changed(h)); }
  py::tuple pdf_lists() {
    Arr<int64_t> off({(py::ssize_t)n_utt + 1});
    Check(khg_utts_num_pdfs(h, off.mutable_data()));
    const py::ssize_t n = off.at(n_utt);
    Arr<int32_t> pdfs({n > 0 ? n : 1});
    Check(khg_utts_pdfs(h, pdfs.mutable_data()));
    return py::make_tuple(off, py::array(pdfs)[py::slice(0, n, 1)]);
  }
  py::object pdf_first_frames() {
    Arr<int64_t> off({(py::ssize_t)n_utt + 1});
    Check(khg_utts_num_pdfs(h, off.mutable_data()));
    const py::ssize_t n = off.at(n_utt);
    Arr<int32_t> first({n > 0 ? n : 1});
    Check(khg_utts_pdf_first(h, first.mutable_data()));
    return py::array(first)[py::slice(0, n, 1)];
  }
  py::object pdf_last_frames() {
    Arr<int64_t> off({(py::ssize_t)n_utt + 1});
    Check(khg_utts_num_pdfs(h, off.mutable_data()));
    const py::ssize_t n = off.at(n_utt);
    Arr<int32_t> last({n > 0 ? n : 1});
    Check(khg_utts_pdf_last(h, last.mutable_data()));
    return py::array(last)[py::slice(0, n, 1)];
  }
  // reachable_only: from each pdf's first readable frame (khg_loglikes_reachable); band: ... up to its last useful frame, the rest
  // filled with an upper bound (khg_loglikes_band; `model` must stay alive until the align that follows has returned)
  void loglikes(KModel& m, bool reachable_only, bool band) {
    keep_model = band ? py::cast(&m) : py::object();
    Check(NoGil([&] { return band ? khg_loglikes_band(ctx->h, m.h, h) : reachable_only ? khg_loglikes_reachable(ctx->h, m.h, h) : khg_loglikes(ctx->h, m.h, h); }));
  }
  py::tuple loglikes_layout() {
    Arr<int64_t> off({(py::ssize_t)n_utt + 1});
    int64_t tot = 0;
    Check(khg_loglikes_layout(h, off.mutable_data(), &tot));
    return py::make_tuple(off, tot);
  }
  py::list download_loglikes() {
    Arr<int64_t> off({(py::ssize_t)n_utt + 1}), poff({(py::ssize_t)n_utt + 1});
    int64_t tot = 0;
    Check(khg_loglikes_layout(h, off.mutable_data(), &tot));
    Check(khg_utts_num_pdfs(h, poff.mutable_data()));
    std::vector<float> buf((size_t)(tot > 0 ? tot : 1));
    Check(NoGil([&] { return khg_loglikes_download(ctx->h, h, buf.data()); }));
    py::list out;
    for (int u = 0; u < n_utt; ++u) {
      const int64_t T = frame_off.at(u + 1) - frame_off.at(u), tpad = (T + 31) / 32 * 32, n = poff.at(u + 1) - poff.at(u);
      Arr<float> m({(py::ssize_t)n, (py::ssize_t)T});
      for (int64_t j = 0; j < n; ++j) std::memcpy(m.mutable_data() + j * T, buf.data() + off.at(u) + j * tpad, sizeof(float) * (size_t)T);
      out.append(m);
    }
    return out;
  }
  void upload_loglikes(py::list mats) {
    Arr<int64_t> off({(py::ssize_t)n_utt + 1});
    int64_t tot = 0;
    Check(khg_loglikes_layout(h, off.mutable_data(), &tot));
    std::vector<float> buf((size_t)(tot > 0 ? tot : 1), 0.0f);
    for (int u = 0; u < n_utt && u < (int)mats.size(); ++u) {
      Arr<float> m = mats[u].cast<Arr<float>>();
      const int64_t T = frame_off.at(u + 1) - frame_off.at(u), tpad = (T + 31) / 32 * 32;
      for (py::ssize_t j = 0; j < m.shape(0); ++j) std::memcpy(buf.data() + off.at(u) + j * tpad, m.data() + j * m.shape(1), sizeof(float) * (size_t)T);
    }
    Check(NoGil([&] { return khg_loglikes_upload(ctx->h, h, buf.data()); }));
  }
  py::object align(KTransitions& tm, float beam, float retry_beam, float acoustic_scale, bool careful, int64_t max_active, int min_active,
                   float beam_delta, float hash_ratio, py::object download) {
    khg_align_config c;
    khg_align_config_default(&c);
    c.beam = beam; c.retry_beam = retry_beam; c.careful = careful ? 1 : 0; c.acoustic_scale = acoustic_scale;
    c.max_active = (int32_t)std::min<int64_t>(max_active, std::numeric_limits<int32_t>::max());
    c.min_active = min_active; c.beam_delta = beam_delta; c.hash_ratio = hash_ratio;
    const bool summary = py::isinstance<py::str>(download) && download.cast<std::string>() == "summary";
    if (!summary && !download.cast<bool>()) {
      Check(NoGil([&] { return khg_align(ctx->h, tm.h, h, &c, nullptr, nullptr, nullptr, 0, nullptr, nullptr); }));
      return py::none();
    }
    Arr<float> like({(py::ssize_t)n_utt});
    Arr<int32_t> status({(py::ssize_t)n_utt});
    py::dict d;
    if (summary) {
      Check(NoGil([&] { return khg_align(ctx->h, tm.h, h, &c, nullptr, nullptr, nullptr, 0, like.mutable_data(), status.mutable_data()); }));
      d["like"] = like; d["status"] = status;
      return d;
    }
    const int64_t N = frame_off.at(n_utt), wcap = N + 16 * (int64_t)n_utt + 1024;
    Arr<int32_t> ali({(py::ssize_t)(N > 0 ? N : 1)}), words({(py::ssize_t)wcap});
    Arr<int64_t> woff({(py::ssize_t)n_utt + 1});
    Check(NoGil([&] { return khg_align(ctx->h, tm.h, h, &c, ali.mutable_data(), words.mutable_data(), woff.mutable_data(), wcap, like.mutable_data(), status.mutable_data()); }));
    d["ali"] = py::array(ali)[py::slice(0, N, 1)];
    d["like"] = like; d["status"] = status;
    d["words"] = py::array(words)[py::slice(0, woff.at(n_utt), 1)];
    d["words_off"] = woff;
    return d;
  }
  void upload_ali(Arr<int32_t> a) {
    if (a.shape(0) != frame_off.at(n_utt)) throw py::value_error("upload_ali: one transition-id per frame");
    Check(NoGil([&] { return khg_ali_upload(ctx->h, h, a.data()); }));
  }
  py::object download_ali() {
    const int64_t N = frame_off.at(n_utt);
    Arr<int32_t> a({(py::ssize_t)(N > 0 ? N : 1)});
    Check(NoGil([&] { return khg_ali_download(ctx->h, h, a.mutable_data()); }));
    return py::array(a)[py::slice(0, N, 1)];
  }
  void acc_stats_reduce(KModel& m, KTransitions& tm, KAccs& accs, float weight, py::object comm, int nparts) {
    void* c = comm.is_none() ? nullptr : comm.cast<KComm*>()->h;
    Check(NoGil([&] { return khg_acc_stats_reduce(ctx->h, m.h, tm.h, h, weight, accs.h, c, nparts); }));
  }
  void acc_stats(KModel& m, KTransitions& tm, KAccs& accs, float weight) { Check(NoGil([&] { return khg_acc_stats(ctx->h, m.h, tm.h, h, weight, accs.h); })); }
};

}  // namespace

void BindHost(py::module_& m, py::object* error_class);      // khg_py_host.cpp: the host classes behind the reference's names

PYBIND11_MODULE(_kaldi_hmm_gmm_amd, m) {
  m.doc() = "pybind11 host surface of libkhg_hip.so (include/khg_hip.h): the accelerated EM hot path of kaldi-hmm-gmm on MI355X";
  m.def("_set_error_class", [](py::object cls) { g_khg_error = cls; });
  BindHost(m, &g_khg_error);
  m.def("version", [] { return khg_version(); });
  m.attr("ALIGN_DONE") = KHG_ALIGN_DONE; m.attr("ALIGN_ERROR") = KHG_ALIGN_ERROR; m.attr("ALIGN_RETRIED") = KHG_ALIGN_RETRIED;
  m.attr("ALIGN_EXACT_DP") = KHG_ALIGN_EXACT_DP; m.attr("ALIGN_FALLBACK") = KHG_ALIGN_FALLBACK;

  py::class_<KContext>(m, "Context")
      .def(py::init<int, py::object>(), py::arg("device") = 0, py::arg("stream") = py::none())
      .def_property_readonly("h", [](KContext& c) { return reinterpret_cast<uintptr_t>(c.h); })
      .def_readonly("device", &KContext::device)
      .def("sync", &KContext::sync).def("set_timing", &KContext::set_timing).def("timings", &KContext::timings)
      .def("set_k1_form", &KContext::set_k1_form).def("set_option", &KContext::set_option, py::arg("name"), py::arg("value"))
      .def("get_option", &KContext::get_option, py::arg("name")).def("close", &KContext::close);

  py::class_<KComm>(m, "Comm")
      .def_static("unique_id", &KComm::unique_id)
      .def(py::init<KContext&, int, int, py::bytes>(), py::arg("ctx"), py::arg("nranks"), py::arg("rank"), py::arg("uid"), py::keep_alive<1, 2>())
      .def_property_readonly("h", [](KComm& c) { return reinterpret_cast<uintptr_t>(c.h); })
      .def_readonly("nranks", &KComm::nranks).def_readonly("rank", &KComm::rank).def("close", &KComm::close).def("info", &KComm::info)
      .def_property_readonly_static("ID_BYTES", [](py::object) { return KHG_COMM_ID_BYTES; });

  py::class_<KModel>(m, "DeviceModel")
      .def(py::init<py::object, Arr<int32_t>, Arr<float>, Arr<float>, Arr<float>, py::object>(), py::arg("ctx"), py::arg("gauss_off"),
           py::arg("gconsts"), py::arg("means_invvars"), py::arg("inv_vars"), py::arg("weights") = py::none())
      .def_property_readonly("h", [](KModel& x) { return reinterpret_cast<uintptr_t>(x.h); })
      .def_readonly("ctx", &KModel::ctx_obj).def_readonly("num_pdfs", &KModel::num_pdfs).def_readonly("dim", &KModel::dim)
      .def_readonly("gauss_off", &KModel::gauss_off)
      .def("set_weights", &KModel::set_weights)
      .def("mle_update", &KModel::mle_update, py::arg("accs"), py::arg("opts") = py::none(), py::arg("flags") = 0x7)
      .def("mle_update_sharded", &KModel::mle_update_sharded, py::arg("accs"), py::arg("opts") = py::none(), py::arg("flags") = 0x7,
           py::arg("comm") = py::none())
      .def("mle_update_range", &KModel::mle_update_range, py::arg("accs"), py::arg("opts"), py::arg("flags"), py::arg("first_pdf"), py::arg("n_pdf"))
      .def("mle_rows_download", &KModel::mle_rows_download).def("mle_rows_upload", &KModel::mle_rows_upload)
      .def("mle_update_finish", &KModel::mle_update_finish)
      .def("scale_weights", &KModel::scale_weights)
      .def("invalidate", &KModel::invalidate)
      .def("split", &KModel::split, py::arg("targets"), py::arg("perturb_factor"), py::arg("randn"))
      .def("merge", &KModel::merge, py::arg("targets"))
      .def("download", &KModel::download, py::arg("weights") = true)
      .def("close", &KModel::close);

  py::class_<KTransitions>(m, "DeviceTransitions")
      .def(py::init<py::object, Arr<int32_t>>(), py::arg("ctx"), py::arg("id2pdf"))
      .def_property_readonly("h", [](KTransitions& x) { return reinterpret_cast<uintptr_t>(x.h); })
      .def_readonly("ctx", &KTransitions::ctx_obj).def_readonly("id2pdf", &KTransitions::id2pdf).def_readonly("num_tids", &KTransitions::num_tids)
      .def("set_trans_cost", &KTransitions::set_trans_cost).def("close", &KTransitions::close);

  py::class_<KAccs>(m, "DeviceAccs")
      .def(py::init<py::object, KModel&, KTransitions&>(), py::arg("ctx"), py::arg("model"), py::arg("tm"))
      .def_property_readonly("h", [](KAccs& x) { return reinterpret_cast<uintptr_t>(x.h); })
      .def_readonly("ctx", &KAccs::ctx_obj).def_readonly("sumG", &KAccs::sumG).def_readonly("dim", &KAccs::dim)
      .def_readonly("num_tids", &KAccs::num_tids).def_readonly("size", &KAccs::size)
      .def("zero", &KAccs::zero).def("device_ptr", &KAccs::device_ptr)
      .def("allreduce", &KAccs::allreduce, py::arg("comm") = py::none(), py::arg("wire_fp32") = false)
      .def("allreduce_range", &KAccs::allreduce_range, py::arg("model"), py::arg("first_pdf"), py::arg("n_pdf"), py::arg("comm") = py::none())
      .def("split", &KAccs::split).def("relayout", &KAccs::relayout).def("download_range", &KAccs::download_range)
      .def("download_occ", [](KAccs& a) { return a.download_range(0, a.sumG); })
      .def("download_trans", &KAccs::download_trans).def("download", &KAccs::download).def("upload", &KAccs::upload)
      .def("close", &KAccs::close);

  py::class_<KUtts>(m, "UtteranceSet")
      .def(py::init<py::object, py::object, Arr<int64_t>, py::object, py::object, py::object>(), py::arg("ctx"), py::arg("tm"), py::arg("frame_off"),
           py::arg("feats"), py::arg("dim") = py::none(), py::arg("graphs") = py::none())
      .def_property_readonly("h", [](KUtts& x) { return reinterpret_cast<uintptr_t>(x.h); })
      .def_readonly("ctx", &KUtts::ctx_obj).def_readonly("frame_off", &KUtts::frame_off).def_readonly("n_utt", &KUtts::n_utt)
      .def_readonly("dim", &KUtts::dim)
      .def("set_pdf_list", &KUtts::set_pdf_list).def("features_changed", &KUtts::features_changed).def("pdf_lists", &KUtts::pdf_lists).def("pdf_first_frames", &KUtts::pdf_first_frames)
      .def("pdf_last_frames", &KUtts::pdf_last_frames)
      .def("loglikes", &KUtts::loglikes, py::arg("model"), py::arg("reachable_only") = false, py::arg("band") = false)
      .def("loglikes_layout", &KUtts::loglikes_layout).def("download_loglikes", &KUtts::download_loglikes)
      .def("upload_loglikes", &KUtts::upload_loglikes)
      .def("align", &KUtts::align, py::arg("tm"), py::arg("beam") = 200.0f, py::arg("retry_beam") = 0.0f, py::arg("acoustic_scale") = 1.0f,
           py::arg("careful") = false, py::arg("max_active") = (int64_t)std::numeric_limits<int32_t>::max(), py::arg("min_active") = 20,
           py::arg("beam_delta") = 0.5f, py::arg("hash_ratio") = 2.0f, py::arg("download") = true)
      .def("upload_ali", &KUtts::upload_ali).def("download_ali", &KUtts::download_ali)
      .def("acc_stats", &KUtts::acc_stats, py::arg("model"), py::arg("tm"), py::arg("accs"), py::arg("weight") = 1.0f)
      .def("acc_stats_reduce", &KUtts::acc_stats_reduce, py::arg("model"), py::arg("tm"), py::arg("accs"), py::arg("weight") = 1.0f,
           py::arg("comm") = py::none(), py::arg("nparts") = 4)
      .def("close", &KUtts::close);

  // ---- host-side functions (no GPU): gconsts, M-step, merge, transition update, AddTransitionProbs costs ----
  m.def("compute_gconsts", [](Arr<int32_t> go, Arr<float> w, Arr<float> iv, Arr<float> miv) {
    const int P = (int)go.shape(0) - 1;
    Arr<float> gc({w.shape(0)});
    int32_t bad = 0;
    Check(khg_compute_gconsts(P, (int)iv.shape(1), go.data(), w.data(), iv.data(), miv.data(), gc.mutable_data(), &bad));
    return py::make_tuple(gc, bad);
  }, "DiagGmm::ComputeGconsts (csrc/diag-gmm.cc:103-147) over a ragged model -> (gconsts, num_bad)");
  m.def("diag_gmm_merge", [](Arr<float> w, Arr<float> miv, Arr<float> iv, int target) {
    int32_t G = (int32_t)w.shape(0), nh = 0;
    const int D = (int)miv.shape(1);
    Arr<float> w2 = py::array(w).attr("copy")(), miv2 = py::array(miv).attr("copy")(), iv2 = py::array(iv).attr("copy")();
    Arr<float> gc({(py::ssize_t)G});
    std::vector<int32_t> hist((size_t)2 * (G > 0 ? G : 1));
    Check(khg_diag_gmm_merge(&G, D, target, w2.mutable_data(), gc.mutable_data(), miv2.mutable_data(), iv2.mutable_data(), hist.data(), &nh));
    hist.resize((size_t)nh);
    py::dict d;
    d["num_gauss"] = G; d["weights"] = w2; d["gconsts"] = gc; d["means_invvars"] = miv2; d["inv_vars"] = iv2; d["history"] = hist;
    return d;
  }, "DiagGmm::Merge (csrc/diag-gmm.cc:557-759): the first num_gauss rows of the returned arrays are the merged model");
  m.def("scaled_trans_cost", [](Arr<float> lp, Arr<float> nsl, Arr<int32_t> id2state, Arr<uint8_t> isl, float ts, float sls) {
    const int nt = (int)lp.shape(0) - 1;
    Arr<float> out({(py::ssize_t)nt + 1});
    Check(khg_scaled_trans_cost(nt, lp.data(), nsl.data(), id2state.data(), isl.data(), ts, sls, out.mutable_data()));
    return out;
  }, "-GetScaledTransitionLogProb (csrc/hmm-utils.cc:442-463) for every transition-id");
}
