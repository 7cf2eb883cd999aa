// kaldi_hmm_gmm_amd/csrc/khg_k4.hip -- C-ABI (include/khg_hip.h): K4, the M-step on the device (khg_model_mle_update and its sharded
// form), mixing up / down on the handle, weight scaling, and the small accumulator downloads the host-side updates need.  gfx950 only.
#include "khg_internal.hpp"
#include "khg_rccl.hpp"

#include "khg_k4_mstep.hip.inc"

// ------------------------------------------------------------------------------------------
// K4: device M-step
extern "C" int khg_model_set_weights(khg_ctx* ctx, khg_model* m, const float* weights) {
  if (ctx_dead(ctx) || !m || !weights) return khg_set_error(KHG_E_ARG, "khg_model_set_weights: bad arguments");
  if (!m->weights_d) { int rc = dev_alloc(&m->weights_d, (size_t)m->sumG); if (rc) return rc; }
  HIPCHK(hipMemcpyAsync(m->weights_d, weights, sizeof(float) * (size_t)m->sumG, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  m->has_weights = true;
  return KHG_OK;
}
extern "C" int khg_model_num_gauss(const khg_model* m, int64_t* total, int32_t* gauss_off) {
  if (!m) return khg_set_error(KHG_E_ARG, "khg_model_num_gauss: model is NULL");
  if (total) *total = m->sumG;
  if (gauss_off) std::memcpy(gauss_off, m->gauss_off.data(), sizeof(int32_t) * ((size_t)m->P + 1));
  return KHG_OK;
}
extern "C" int khg_model_download(khg_ctx* ctx, const khg_model* m, float* weights, float* gconsts, float* miv, float* iv) {
  if (ctx_dead(ctx) || !m) return khg_set_error(KHG_E_ARG, "khg_model_download: bad arguments");
  if (weights && !m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_download: the model has no weights (khg_model_set_weights)");
  const size_t G = (size_t)m->sumG, n = G * m->D;
  if (weights) HIPCHK(hipMemcpyAsync(weights, m->weights_d, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream));
  if (gconsts) HIPCHK(hipMemcpyAsync(gconsts, m->gconsts_d, sizeof(float) * G, hipMemcpyDeviceToHost, ctx->stream));
  if (miv) HIPCHK(hipMemcpyAsync(miv, m->miv_d, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (iv) HIPCHK(hipMemcpyAsync(iv, m->iv_d, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

// The device M-step in two halves, so that it can be SHARDED over ranks by pdf range:
//   rows:    k4_mle_update on pdfs [p0, p0 + np): their parameter rows are rewritten in place (old layout), one K4Res per pdf;
//   finish:  totals in pdf order, compaction when some pdf lost Gaussians, the K1 / K3 images -- on the complete rows + results.
static int mle_update_rows(khg_ctx* ctx, khg_model* m, const khg_accs* acc, const khg_mle_options* o, uint16_t flags, int p0, int np) {
  if (ctx_dead(ctx) || !m || !acc || !o) return khg_set_error(KHG_E_ARG, "khg_model_mle_update: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_mle_update: the model has no weights (khg_model_set_weights)");
  if (acc->D != m->D || acc->sumG != m->sumG)
    return khg_set_error(KHG_E_RUNTIME, "khg_model_mle_update: accumulator / model dimensions do not match");
  if (flags & ~0x7) return khg_set_error(KHG_E_RUNTIME, "Flags in argument do not match the active accumulators");   // mle-diag-gmm.cc:252
  if (p0 < 0 || np < 0 || p0 + np > m->P) return khg_set_error(KHG_E_ARG, "khg_model_mle_update: pdf range outside the model");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  const int P = m->P, D = m->D;
  int maxG = 0;
  for (int p = 0; p < P; ++p) maxG = std::max(maxG, m->gauss_off[p + 1] - m->gauss_off[p]);
  const size_t lds = sizeof(double) * (256 + (size_t)maxG) + sizeof(float) * 5 * (size_t)maxG;
  if (lds > 60 * 1024) return khg_set_error(KHG_E_UNSUPPORTED, "khg_model_mle_update: more than ~2000 Gaussians in one pdf");
  if (!m->k4_res_d || m->k4_res_P != P) {
    DEVFREE(m->k4_res_d);
    int rc = dev_alloc(&m->k4_res_d, (size_t)P);
    if (rc) return rc;
    m->k4_res_P = P;
  }
  K4Args a;
  a.gauss_off = m->gauss_off_d; a.D = D;
  a.occ = acc->occ(); a.macc = acc->mean(); a.vacc = acc->var();
  a.w = m->weights_d; a.gc = m->gconsts_d; a.miv = m->miv_d; a.iv = m->iv_d;
  a.res = m->k4_res_d;
  a.min_w = o->min_gaussian_weight; a.min_occ = o->min_gaussian_occupancy; a.min_var = o->min_variance;
  double* floor_d = nullptr;
  a.var_floor = nullptr;
  if (o->variance_floor_vector) {
    std::vector<double> fv(o->variance_floor_vector, o->variance_floor_vector + D);
    int rcf = dev_upload(ctx, &floor_d, fv);
    if (!rcf) { hipError_t ef = hipStreamSynchronize(ctx->stream); if (ef != hipSuccess) rcf = khg_set_error(KHG_E_HIP, hipGetErrorString(ef)); }
    if (rcf) { DEVFREE(floor_d); return rcf; }
    a.var_floor = floor_d;
  }
  a.remove_low = o->remove_low_count_gaussians; a.flags = flags; a.pdf0 = p0;
  if (np > 0) {
    KernelTimer kt(ctx, "k4_mle_update");
    KHG_LAUNCH(ctx, k4_mle_update, dim3(np), dim3(256), lds, ctx->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && floor_d) e = hipStreamSynchronize(ctx->stream);
  DEVFREE(floor_d);
  if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  return KHG_OK;
}
static int mle_update_finish(khg_ctx* ctx, khg_model* m, float* objf_change, float* count, int32_t* floored_elems, int32_t* floored_gauss,
                             int32_t* removed) {
  if (ctx_dead(ctx) || !m || !m->k4_res_d || m->k4_res_P != m->P) return khg_set_error(KHG_E_ARG, "khg_model_mle_update_finish: no update in progress");
  const int P = m->P, D = m->D;
  std::vector<K4Res> res((size_t)P);
  hipError_t e = hipMemcpyAsync(res.data(), m->k4_res_d, sizeof(K4Res) * (size_t)P, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  int rc = KHG_OK;
  // totals in pdf order, float, as MleAmDiagGmmUpdate adds them (csrc/mle-am-diag-gmm.cc:177-193)
  float tot_obj = 0.0f, tot_count = 0.0f;
  int tfe = 0, tfg = 0, trm = 0;
  std::vector<int32_t> new_off((size_t)P + 1);
  int out = 0;
  for (int p = 0; p < P; ++p) {
    const K4Res& r = res[(size_t)p];
    if (r.bad) return khg_set_error(KHG_E_RUNTIME, "pdf " + std::to_string(p) + ": not a number in gconst computation");
    tot_obj += r.obj_change; tot_count += r.count; tfe += r.floored_elems; tfg += r.floored_gauss; trm += r.removed;
    new_off[(size_t)p] = out;
    out += r.newG;
  }
  new_off[(size_t)P] = out;
  if (trm > 0) {
    // some pdf shrank: move every pdf's rows to the new offsets in fresh arrays
    int32_t* new_off_d = nullptr;
    float *w2 = nullptr, *gc2 = nullptr, *miv2 = nullptr, *iv2 = nullptr;
    rc = dev_upload(ctx, &new_off_d, new_off);
    if (!rc) rc = dev_alloc(&w2, (size_t)out);
    if (!rc) rc = dev_alloc(&gc2, (size_t)out);
    if (!rc) rc = dev_alloc(&miv2, (size_t)out * D);
    if (!rc) rc = dev_alloc(&iv2, (size_t)out * D);
    if (!rc) {
      KernelTimer kt(ctx, "k4_compact");
      KHG_LAUNCH(ctx, k4_compact, dim3(P), dim3(256), 0, ctx->stream, m->gauss_off_d, new_off_d, D, m->weights_d, m->gconsts_d,
                         m->miv_d, m->iv_d, w2, gc2, miv2, iv2);
    }
    if (!rc) {
      e = hipGetLastError();
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
    }
    DEVFREE(new_off_d);
    if (rc) { DEVFREE(w2); DEVFREE(gc2); DEVFREE(miv2); DEVFREE(iv2); return rc; }
    DEVFREE(m->weights_d); DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d);
    m->weights_d = w2; m->gconsts_d = gc2; m->miv_d = miv2; m->iv_d = iv2;
    m->gauss_off = new_off;
    m->sumG = out;
  }
  rc = model_pack(ctx, m);   // new K1 tile image + -0.5*inv_vars from the updated parameters
  if (rc) return rc;
  if (objf_change) *objf_change = tot_obj;
  if (count) *count = tot_count;
  if (floored_elems) *floored_elems = tfe;
  if (floored_gauss) *floored_gauss = tfg;
  if (removed) *removed = trm;
  return KHG_OK;
}
extern "C" int khg_model_mle_update(khg_ctx* ctx, khg_model* m, const khg_accs* acc, const khg_mle_options* o, uint16_t flags,
                                    float* objf_change, float* count, int32_t* floored_elems, int32_t* floored_gauss,
                                    int32_t* removed) {
  int rc = mle_update_rows(ctx, m, acc, o, flags, 0, m ? m->P : 0);
  if (rc) return rc;
  return mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
}
extern "C" int khg_model_mle_update_range(khg_ctx* ctx, khg_model* m, const khg_accs* acc, const khg_mle_options* o, uint16_t flags,
                                          int32_t first_pdf, int32_t n_pdf) {
  return mle_update_rows(ctx, m, acc, o, flags, first_pdf, n_pdf);
}
extern "C" int khg_model_mle_update_finish(khg_ctx* ctx, khg_model* m, float* objf_change, float* count, int32_t* floored_elems,
                                           int32_t* floored_gauss, int32_t* removed) {
  return mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
}
// the rows an update of pdfs [first_pdf, first_pdf + n_pdf) rewrote + its per-pdf results (32 bytes each), to / from the host: the
// exchange step of the sharded M-step for callers whose ranks cannot share device buffers (the tests' gloo ranks on one GPU)
static int mle_rows_copy(khg_ctx* ctx, khg_model* m, int p0, int np, float* w, float* gc, float* miv, float* iv, void* res, bool up) {
  if (ctx_dead(ctx) || !m || p0 < 0 || np < 0 || p0 + np > m->P || !m->k4_res_d || m->k4_res_P != m->P)
    return khg_set_error(KHG_E_ARG, "khg_model_mle_rows: bad arguments (or no update in progress)");
  const size_t g0 = (size_t)m->gauss_off[p0], ng = (size_t)m->gauss_off[p0 + np] - g0, D = (size_t)m->D;
  auto cp = [&](float* host, float* dev, size_t n) -> hipError_t {
    if (!host || n == 0) return hipSuccess;
    return up ? hipMemcpyAsync(dev, host, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream)
              : hipMemcpyAsync(host, dev, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  };
  HIPCHK(cp(w, m->weights_d + g0, ng));
  HIPCHK(cp(gc, m->gconsts_d + g0, ng));
  HIPCHK(cp(miv, m->miv_d + g0 * D, ng * D));
  HIPCHK(cp(iv, m->iv_d + g0 * D, ng * D));
  if (res && np > 0)
    HIPCHK(up ? hipMemcpyAsync(m->k4_res_d + p0, res, sizeof(K4Res) * (size_t)np, hipMemcpyHostToDevice, ctx->stream)
              : hipMemcpyAsync(res, m->k4_res_d + p0, sizeof(K4Res) * (size_t)np, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_model_mle_rows_download(khg_ctx* ctx, khg_model* m, int32_t first_pdf, int32_t n_pdf, float* weights, float* gconsts,
                                           float* means_invvars, float* inv_vars, void* results) {
  return mle_rows_copy(ctx, m, first_pdf, n_pdf, weights, gconsts, means_invvars, inv_vars, results, false);
}
extern "C" int khg_model_mle_rows_upload(khg_ctx* ctx, khg_model* m, int32_t first_pdf, int32_t n_pdf, const float* weights,
                                         const float* gconsts, const float* means_invvars, const float* inv_vars, const void* results) {
  return mle_rows_copy(ctx, m, first_pdf, n_pdf, const_cast<float*>(weights), const_cast<float*>(gconsts), const_cast<float*>(means_invvars),
                       const_cast<float*>(inv_vars), const_cast<void*>(results), true);
}
// SURVEY.md 8f-3 as written: the block is REDUCED by pdf range to its owner (rank r owns pdfs [P r / N, P (r + 1) / N)) instead of
// all-reduced, every rank updates its own pdfs, the updated rows and per-pdf results are broadcast from their owners, and every rank
// finishes (compaction, images) on the complete model: (N - 1) / N x (207 + 105) MB per rank on the wire instead of
// 2 (N - 1) / N x 207 MB at 5000 x 64 x 40.  `acc` holds this rank's LOCAL sums (no khg_accs_allreduce before); on return its
// occupancies are summed over the ranks, its mean / variance rows are complete only for the rank's own pdfs, and its transition counts
// and scalars are untouched (khg_accs_allreduce_range with first_pdf < 0 sums those).
extern "C" int khg_model_mle_update_sharded(khg_ctx* ctx, khg_model* m, khg_accs* acc, const khg_mle_options* o, uint16_t flags,
                                            void* comm, int32_t nranks, int32_t rank, float* objf_change, float* count,
                                            int32_t* floored_elems, int32_t* floored_gauss, int32_t* removed) {
  if (ctx_dead(ctx) || !m || !acc || !o || nranks < 1 || rank < 0 || rank >= nranks) return khg_set_error(KHG_E_ARG, "khg_model_mle_update_sharded: bad arguments");
  if (!comm) {                      // (a one-rank communicator still goes through RCCL: reductions and broadcasts to itself)
    int rc = mle_update_rows(ctx, m, acc, o, flags, 0, m->P);
    return rc ? rc : mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
  }
  int rc = rccl_bind();
  if (rc) return rc;
  if (!g_rccl.Reduce || !g_rccl.Broadcast) return khg_set_error(KHG_E_UNSUPPORTED, "RCCL library lacks ncclReduce / ncclBroadcast");
  if (acc->D != m->D || acc->sumG != m->sumG) return khg_set_error(KHG_E_RUNTIME, "khg_model_mle_update_sharded: accumulator / model dimensions do not match");
  const int P = m->P;
  const int64_t D = m->D;
  auto range = [&](int r, int* p0, int* np) { *p0 = (int)((int64_t)P * r / nranks); *np = (int)((int64_t)P * (r + 1) / nranks) - *p0; };
  {
    KernelTimer kt(ctx, "c1_reduce_by_pdf_range");
    int r = g_rccl.GroupStart();
    // occupancies: all of them to everybody (1 / (2 D + 1) of the block; the mixing-up targets need every pdf's) ...
    if (!r && acc->sumG > 0) r = g_rccl.AllReduce(acc->occ(), acc->occ(), (size_t)acc->sumG, kNcclFloat64, kNcclSum, comm, ctx->stream);
    for (int o2 = 0; o2 < nranks && !r; ++o2) {      // ... first- and second-order sums: each pdf range to its owner only
      int p0, np;
      range(o2, &p0, &np);
      const int64_t g0 = m->gauss_off[p0], ng = m->gauss_off[p0 + np] - g0;
      if (ng == 0) continue;
      r = g_rccl.Reduce(acc->mean() + g0 * D, acc->mean() + g0 * D, (size_t)(ng * D), kNcclFloat64, kNcclSum, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Reduce(acc->var() + g0 * D, acc->var() + g0 * D, (size_t)(ng * D), kNcclFloat64, kNcclSum, o2, comm, ctx->stream);
    }
    const int r2 = g_rccl.GroupEnd();
    if (r || r2) return rccl_fail("ncclReduce (sharded M-step)", r ? r : r2);
  }
  int p0, np;
  range(rank, &p0, &np);
  rc = mle_update_rows(ctx, m, acc, o, flags, p0, np);
  if (rc) return rc;
  {
    KernelTimer kt(ctx, "c1_broadcast_rows");
    int r = g_rccl.GroupStart();
    for (int o2 = 0; o2 < nranks && !r; ++o2) {
      int q0, nq;
      range(o2, &q0, &nq);
      const int64_t g0 = m->gauss_off[q0], ng = m->gauss_off[q0 + nq] - g0;
      if (nq > 0) r = g_rccl.Broadcast(m->k4_res_d + q0, m->k4_res_d + q0, sizeof(K4Res) * (size_t)nq, kNcclInt8, o2, comm, ctx->stream);
      if (ng == 0) continue;
      if (!r) r = g_rccl.Broadcast(m->weights_d + g0, m->weights_d + g0, (size_t)ng, kNcclFloat32, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Broadcast(m->gconsts_d + g0, m->gconsts_d + g0, (size_t)ng, kNcclFloat32, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Broadcast(m->miv_d + g0 * D, m->miv_d + g0 * D, (size_t)(ng * D), kNcclFloat32, o2, comm, ctx->stream);
      if (!r) r = g_rccl.Broadcast(m->iv_d + g0 * D, m->iv_d + g0 * D, (size_t)(ng * D), kNcclFloat32, o2, comm, ctx->stream);
    }
    const int r2 = g_rccl.GroupEnd();
    if (r || r2) return rccl_fail("ncclBroadcast (sharded M-step)", r ? r : r2);
  }
  return mle_update_finish(ctx, m, objf_change, count, floored_elems, floored_gauss, removed);
}

extern "C" int khg_model_split(khg_ctx* ctx, khg_model* m, const int32_t* targets, float perturb, const float* randn, int64_t n_randn) {
  if (ctx_dead(ctx) || !m || !targets) return khg_set_error(KHG_E_ARG, "khg_model_split: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_split: the model has no weights (khg_model_set_weights)");
  const int P = m->P, D = m->D;
  std::vector<int32_t> new_off((size_t)P + 1, 0);
  std::vector<int64_t> rand_off((size_t)P + 1, 0);
  for (int p = 0; p < P; ++p) {
    const int cur = m->gauss_off[p + 1] - m->gauss_off[p];
    if (cur == 0 && targets[p] > 0) return khg_set_error(KHG_E_RUNTIME, "khg_model_split: pdf " + std::to_string(p) + " has no component to split");
    if (targets[p] < cur)   // csrc/diag-gmm.cc:782-786
      return khg_set_error(KHG_E_RUNTIME, "Cannot split from " + std::to_string(cur) + " to " + std::to_string(targets[p]) + " components");
    new_off[(size_t)p + 1] = new_off[(size_t)p] + targets[p];
    rand_off[(size_t)p + 1] = rand_off[(size_t)p] + (targets[p] - cur);
  }
  const int64_t nnew = rand_off[(size_t)P], out = new_off[(size_t)P];
  if (nnew == 0) return KHG_OK;
  if (!randn) return khg_set_error(KHG_E_ARG, "khg_model_split: randn_h is NULL");
  if (n_randn < nnew * D)
    return khg_set_error(KHG_E_ARG, "khg_model_split: randn_h holds " + std::to_string(n_randn) + " deviates, " + std::to_string(nnew * D) + " are needed (new components x dim)");
  std::vector<float> rv(randn, randn + (size_t)nnew * D);
  int32_t *new_off_d = nullptr, *bad_d = nullptr;
  int64_t* rand_off_d = nullptr;
  float *rand_d = nullptr, *w2 = nullptr, *gc2 = nullptr, *miv2 = nullptr, *iv2 = nullptr;
  int rc = dev_upload(ctx, &new_off_d, new_off);
  if (!rc) rc = dev_upload(ctx, &rand_off_d, rand_off);
  if (!rc) rc = dev_upload(ctx, &rand_d, rv);
  if (!rc) rc = dev_alloc(&bad_d, 1);
  if (!rc) rc = dev_alloc(&w2, (size_t)out);
  if (!rc) rc = dev_alloc(&gc2, (size_t)out);
  if (!rc) rc = dev_alloc(&miv2, (size_t)out * D);
  if (!rc) rc = dev_alloc(&iv2, (size_t)out * D);
  int32_t bad = 0;
  if (!rc) {
    hipError_t e = hipMemsetAsync(bad_d, 0, sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) {
      KernelTimer kt(ctx, "k4_split");
      KHG_LAUNCH(ctx, k4_split, dim3(P), dim3(256), 0, ctx->stream, m->gauss_off_d, new_off_d, D, m->weights_d, m->miv_d, m->iv_d, w2, gc2,
                         miv2, iv2, rand_d, rand_off_d, perturb, bad_d);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  DEVFREE(new_off_d); DEVFREE(rand_off_d); DEVFREE(rand_d); DEVFREE(bad_d);
  if (!rc && bad) rc = khg_set_error(KHG_E_RUNTIME, "khg_model_split: not a number in gconst computation");
  if (rc) { DEVFREE(w2); DEVFREE(gc2); DEVFREE(miv2); DEVFREE(iv2); return rc; }
  DEVFREE(m->weights_d); DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d);
  m->weights_d = w2; m->gconsts_d = gc2; m->miv_d = miv2; m->iv_d = iv2;
  m->gauss_off = new_off;
  m->sumG = out;
  return model_pack(ctx, m);
}

// AmDiagGmm::MergeByCount's per-pdf DiagGmm::Merge (csrc/am-diag-gmm.cc:91-108, csrc/diag-gmm.cc:557-759) on the device model:
// pdf p keeps targets[p] components (1 <= targets[p] <= its count).  Nothing crosses PCIe but the offsets.
extern "C" int khg_model_merge(khg_ctx* ctx, khg_model* m, const int32_t* targets) {
  if (ctx_dead(ctx) || !m || !targets) return khg_set_error(KHG_E_ARG, "khg_model_merge: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_merge: the model has no weights (khg_model_set_weights)");
  const int P = m->P, D = m->D;
  std::vector<int32_t> new_off((size_t)P + 1, 0);
  std::vector<int64_t> delta_off((size_t)P + 1, 0);
  bool any = false;
  for (int p = 0; p < P; ++p) {
    const int cur = m->gauss_off[p + 1] - m->gauss_off[p];
    if (targets[p] <= 0 || cur < targets[p])   // csrc/diag-gmm.cc:558-562
      return khg_set_error(KHG_E_RUNTIME, "Invalid argument for target number of Gaussians (=" + std::to_string(targets[p]) + "), #Gauss = " + std::to_string(cur));
    new_off[(size_t)p + 1] = new_off[(size_t)p] + targets[p];
    const bool greedy = targets[p] < cur && targets[p] > 1;
    delta_off[(size_t)p + 1] = delta_off[(size_t)p] + (greedy ? (int64_t)cur * cur : 0);
    any = any || targets[p] < cur;
  }
  if (!any) return KHG_OK;
  const int64_t out = new_off[(size_t)P], old = m->sumG;
  K4MergeArgs a{};
  int32_t *new_off_d = nullptr, *idx_d = nullptr, *bad_d = nullptr;
  int64_t* delta_off_d = nullptr;
  float *scratch = nullptr, *delta_d = nullptr, *w2 = nullptr, *gc2 = nullptr, *miv2 = nullptr, *iv2 = nullptr;
  int rc = dev_upload(ctx, &new_off_d, new_off);
  if (!rc) rc = dev_upload(ctx, &delta_off_d, delta_off);
  if (!rc) rc = dev_alloc(&idx_d, (size_t)2 * old);
  if (!rc) rc = dev_alloc(&bad_d, 1);
  if (!rc) rc = dev_alloc(&scratch, (size_t)old * (2 + 4 * (size_t)D));
  if (!rc) rc = dev_alloc(&delta_d, (size_t)std::max<int64_t>(1, delta_off[(size_t)P]));
  if (!rc) rc = dev_alloc(&w2, (size_t)out);
  if (!rc) rc = dev_alloc(&gc2, (size_t)out);
  if (!rc) rc = dev_alloc(&miv2, (size_t)out * D);
  if (!rc) rc = dev_alloc(&iv2, (size_t)out * D);
  int32_t bad = 0;
  if (!rc) {
    a.old_off = m->gauss_off_d; a.new_off = new_off_d; a.D = D;
    a.w = m->weights_d; a.gc = m->gconsts_d; a.miv = m->miv_d; a.iv = m->iv_d;
    a.w2 = w2; a.gc2 = gc2; a.miv2 = miv2; a.iv2 = iv2;
    a.wk = scratch; a.logdet = scratch + old;
    a.mean = scratch + 2 * old; a.m2 = a.mean + old * D; a.mivk = a.m2 + old * D; a.ivk = a.mivk + old * D;
    a.gone = idx_d; a.keep = idx_d + old; a.delta = delta_d; a.delta_off = delta_off_d; a.bad = bad_d;
    hipError_t e = hipMemsetAsync(bad_d, 0, sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) {
      KernelTimer kt(ctx, "k4_merge");
      KHG_LAUNCH(ctx, k4_merge, dim3(P), dim3(256), 0, ctx->stream, a);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  DEVFREE(new_off_d); DEVFREE(delta_off_d); DEVFREE(idx_d); DEVFREE(bad_d); DEVFREE(scratch); DEVFREE(delta_d);
  if (!rc && (bad & 2)) rc = khg_set_error(KHG_E_RUNTIME, "khg_model_merge: no pair of components left to merge (max_i != max_j && max_i != -1 && max_j != -1)");
  if (!rc && (bad & 1)) rc = khg_set_error(KHG_E_RUNTIME, "khg_model_merge: not a number in gconst computation");
  if (rc) { DEVFREE(w2); DEVFREE(gc2); DEVFREE(miv2); DEVFREE(iv2); return rc; }
  DEVFREE(m->weights_d); DEVFREE(m->gconsts_d); DEVFREE(m->miv_d); DEVFREE(m->iv_d); DEVFREE(m->nhiv_d);
  m->weights_d = w2; m->gconsts_d = gc2; m->miv_d = miv2; m->iv_d = iv2;
  m->gauss_off = new_off;
  m->sumG = out;
  return model_pack(ctx, m);
}

// After khg_model_mle_update removed Gaussians the accumulator block is laid out for fewer rows.
extern "C" int khg_accs_relayout(khg_ctx* ctx, khg_accs* a, const khg_model* m) {
  if (ctx_dead(ctx) || !a || !m) return khg_set_error(KHG_E_ARG, "khg_accs_relayout: bad arguments");
  if (m->D != a->D) return khg_set_error(KHG_E_RUNTIME, "khg_accs_relayout: dimension mismatch");
  const int64_t n = m->sumG * (1 + 2 * (int64_t)a->D) + a->num_tids + 1 + 8;
  if (n > a->cap) {
    DEVFREE(a->buf_d);
    int rc = dev_alloc(&a->buf_d, (size_t)n);
    if (rc) return rc;
    a->cap = n;
  }
  a->sumG = m->sumG; a->n = n;
  return khg_accs_zero(ctx, a);
}
extern "C" int khg_accs_download_trans(khg_ctx* ctx, const khg_accs* a, double* trans, double* scalars) {
  if (ctx_dead(ctx) || !a) return khg_set_error(KHG_E_ARG, "khg_accs_download_trans: bad arguments");
  if (!trans && scalars) {      // the per-utterance call pattern reads only the totals: one pinned copy in front of the error word's
    double* land = reinterpret_cast<double*>(reinterpret_cast<char*>(ctx->err_host) + 64);
    HIPCHK(hipMemcpyAsync(land, a->scalars(), sizeof(double) * 8, hipMemcpyDeviceToHost, ctx->stream));
    int rc = check_err_flag(ctx, "khg_acc_stats");
    if (rc) return rc;
    memcpy(scalars, land, sizeof(double) * 8);
    return KHG_OK;
  }
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  if (trans) HIPCHK(hipMemcpyAsync(trans, a->trans(), sizeof(double) * ((size_t)a->num_tids + 1), hipMemcpyDeviceToHost, ctx->stream));
  if (scalars) HIPCHK(hipMemcpyAsync(scalars, a->scalars(), sizeof(double) * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}
extern "C" int khg_accs_download_range(khg_ctx* ctx, const khg_accs* a, int64_t first, int64_t count, double* dst) {
  if (ctx_dead(ctx) || !a || !dst || first < 0 || count < 0 || first + count > a->n) return khg_set_error(KHG_E_ARG, "khg_accs_download_range: bad arguments");
  { int rc = check_err_flag(ctx, "khg_acc_stats"); if (rc) return rc; }
  if (count) HIPCHK(hipMemcpyAsync(dst, a->buf_d + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return KHG_OK;
}

extern "C" int khg_model_scale_weights(khg_ctx* ctx, khg_model* m, int32_t n, const int32_t* pdfs, float scale) {
  if (ctx_dead(ctx) || !m || n < 0 || (n > 0 && !pdfs)) return khg_set_error(KHG_E_ARG, "khg_model_scale_weights: bad arguments");
  if (!m->has_weights) return khg_set_error(KHG_E_ARG, "khg_model_scale_weights: the model has no weights (khg_model_set_weights)");
  if (n == 0) return KHG_OK;
  std::vector<int32_t> v(pdfs, pdfs + n);
  for (int32_t p : v)
    if (p < 0 || p >= m->P) return khg_set_error(KHG_E_ARG, "khg_model_scale_weights: pdf-id out of range");
  int32_t *pdfs_d = nullptr, *bad_d = nullptr;
  int rc = dev_upload(ctx, &pdfs_d, v);
  if (!rc) rc = dev_alloc(&bad_d, 1);
  int32_t bad = 0;
  if (!rc) {
    hipError_t e = hipMemsetAsync(bad_d, 0, sizeof(int32_t), ctx->stream);
    if (e == hipSuccess) {
      KHG_LAUNCH(ctx, k4_scale_weights, dim3(n), dim3(64), 0, ctx->stream, pdfs_d, m->gauss_off_d, m->D, scale, m->weights_d,
                         m->gconsts_d, m->miv_d, m->iv_d, bad_d);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, bad_d, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = khg_set_error(KHG_E_HIP, hipGetErrorString(e));
  }
  DEVFREE(pdfs_d); DEVFREE(bad_d);
  if (rc) return rc;
  if (bad) return khg_set_error(KHG_E_RUNTIME, "khg_model_scale_weights: not a number in gconst computation");
  return model_pack(ctx, m);
}
