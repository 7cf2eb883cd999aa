// kaldi_hmm_gmm_amd/csrc/khg_c1.hip -- C-ABI (include/khg_hip.h): C1, the cross-GPU sum of the accumulator block over RCCL
// (bound at run time), whole or by pdf range, and the communicator helpers.  gfx950 only.
#include "khg_internal.hpp"
#include "khg_rccl.hpp"

#include <dlfcn.h>             // RCCL is bound at run time

// ------------------------------------------------------------------------------------------
// C1: the cross-GPU sum of the accumulator block, RCCL called directly.  RCCL is bound at
// run time from whatever copy the process already holds (torch bundles one with the same SONAME; two
// copies in one process would each want their own view of the devices), so the library has no link-time
// dependency on it and a one-GPU user never loads it.
RcclApi g_rccl;
int rccl_bind() {
  if (g_rccl.AllReduce) return KHG_OK;
  void* h = nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // the copy already in the process
  for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return khg_set_error(KHG_E_UNSUPPORTED, std::string("RCCL not available: ") + dlerror());
  RcclApi a; a.h = h;
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  a.Reduce = reinterpret_cast<decltype(a.Reduce)>(dlsym(h, "ncclReduce"));
  a.Broadcast = reinterpret_cast<decltype(a.Broadcast)>(dlsym(h, "ncclBroadcast"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(h, "ncclCommCount"));
  a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  a.GetVersion = reinterpret_cast<decltype(a.GetVersion)>(dlsym(h, "ncclGetVersion"));
  if (!a.AllReduce || !a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.GetErrorString || !a.GroupStart || !a.GroupEnd)
    return khg_set_error(KHG_E_UNSUPPORTED, "RCCL library lacks ncclAllReduce / ncclCommInitRank");
  g_rccl = a;
  return KHG_OK;
}
int rccl_fail(const char* what, int r) {
  return khg_set_error(KHG_E_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}
namespace {
__global__ __launch_bounds__(256) void c1_narrow(const double* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) dst[i] = (float)src[i];
}
__global__ __launch_bounds__(256) void c1_widen(const float* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += 256LL * gridDim.x) dst[i] = (double)src[i];
}
}  // namespace

int ctx_comm_stream(khg_ctx* ctx) {
  if (!ctx->comm_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
  if (!ctx->ev_k3) HIPCHK(hipEventCreateWithFlags(&ctx->ev_k3, hipEventDisableTiming));
  if (!ctx->ev_c1) HIPCHK(hipEventCreateWithFlags(&ctx->ev_c1, hipEventDisableTiming));
  return KHG_OK;
}
// The rows of pdfs [first_pdf, first_pdf + n_pdf) of the block -- occ, mean_acc, var_acc: three contiguous pieces -- or, with
// first_pdf < 0, the transition counts and scalars behind them, summed over the ranks in ONE RCCL group.  st == nullptr: the
// pieces run on the context's communication stream BEHIND everything enqueued on its kernel stream so far, and the kernel stream
// then waits for them only when the tail (first_pdf < 0) has gone out: the pipelined form of khg_acc_stats_reduce.
int accs_allreduce_pieces(khg_ctx* ctx, khg_accs* a, const khg_model* m, int first_pdf, int n_pdf, void* comm, hipStream_t st) {
  int rc = rccl_bind();
  if (rc) return rc;
  const bool piped = st == nullptr;
  if (piped) {
    rc = ctx_comm_stream(ctx);
    if (rc) return rc;
    st = ctx->comm_stream;
    HIPCHK(hipEventRecord(ctx->ev_k3, ctx->stream));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_k3, 0));
  }
  struct Piece { double* p; size_t n; } pc[3];
  int npc = 0;
  if (first_pdf < 0) {
    pc[npc++] = Piece{a->trans(), (size_t)a->num_tids + 1 + 8};
  } else {
    if (first_pdf + n_pdf > m->P || n_pdf < 0) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce_range: pdf range outside the model");
    const int64_t g0 = m->gauss_off[first_pdf], g1 = m->gauss_off[first_pdf + n_pdf];
    if (g1 > g0) {
      pc[npc++] = Piece{a->occ() + g0, (size_t)(g1 - g0)};
      pc[npc++] = Piece{a->mean() + g0 * a->D, (size_t)((g1 - g0) * a->D)};
      pc[npc++] = Piece{a->var() + g0 * a->D, (size_t)((g1 - g0) * a->D)};
    }
  }
  {
    KernelTimer kt(ctx, "c1_allreduce", st);
    int r = g_rccl.GroupStart();
    for (int i = 0; i < npc && !r; ++i) r = g_rccl.AllReduce(pc[i].p, pc[i].p, pc[i].n, kNcclFloat64, kNcclSum, comm, st);
    const int r2 = g_rccl.GroupEnd();
    if (r || r2) return rccl_fail("ncclAllReduce (range)", r ? r : r2);
  }
  if (piped && first_pdf < 0) {
    HIPCHK(hipEventRecord(ctx->ev_c1, st));
    HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_c1, 0));
  }
  return KHG_OK;
}
extern "C" int khg_accs_allreduce_range(khg_ctx* ctx, khg_accs* a, const khg_model* m, int32_t first_pdf, int32_t n_pdf, void* comm) {
  if (ctx_dead(ctx) || !a || !m) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce_range: bad arguments");
  if (a->sumG != m->sumG || a->D != m->D) return khg_set_error(KHG_E_RUNTIME, "khg_accs_allreduce_range: accumulator / model layouts differ");
  if (!comm) return KHG_OK;
  return accs_allreduce_pieces(ctx, a, m, first_pdf, n_pdf, comm, ctx->stream);
}

extern "C" int khg_comm_unique_id(void* id_out) {
  if (!id_out) return khg_set_error(KHG_E_ARG, "khg_comm_unique_id: id_out is NULL");
  int rc = rccl_bind();
  if (rc) return rc;
  int r = g_rccl.GetUniqueId(id_out);
  return r ? rccl_fail("ncclGetUniqueId", r) : KHG_OK;
}
extern "C" int khg_comm_create(khg_ctx* ctx, int32_t nranks, int32_t rank, const void* id, void** comm_out) {
  if (ctx_dead(ctx) || !id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return khg_set_error(KHG_E_ARG, "khg_comm_create: bad arguments");
  int rc = rccl_bind();
  if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  KhgNcclId uid;
  memcpy(uid.internal, id, sizeof(uid.internal));
  void* comm = nullptr;
  int r = g_rccl.CommInitRank(&comm, nranks, uid, rank);
  if (r) return rccl_fail("ncclCommInitRank", r);
  *comm_out = comm;
  return KHG_OK;
}
// what RCCL itself says about a communicator: the number of ranks it spans, this process's rank in it, the library's version code
extern "C" int khg_comm_info(void* comm, int32_t* nranks, int32_t* rank, int32_t* version) {
  int rc = rccl_bind();
  if (rc) return rc;
  int v = 0;
  if (nranks) { *nranks = 0; if (comm && g_rccl.CommCount) { int r = g_rccl.CommCount(comm, &v); if (r) return rccl_fail("ncclCommCount", r); *nranks = v; } }
  if (rank) { *rank = -1; if (comm && g_rccl.CommUserRank) { int r = g_rccl.CommUserRank(comm, &v); if (r) return rccl_fail("ncclCommUserRank", r); *rank = v; } }
  if (version) { *version = 0; if (g_rccl.GetVersion) { int r = g_rccl.GetVersion(&v); if (r) return rccl_fail("ncclGetVersion", r); *version = v; } }
  return KHG_OK;
}
extern "C" int khg_comm_destroy(void* comm) {
  if (!comm) return KHG_OK;
  int rc = rccl_bind();
  if (rc) return rc;
  int r = g_rccl.CommDestroy(comm);
  return r ? rccl_fail("ncclCommDestroy", r) : KHG_OK;
}
extern "C" int khg_accs_allreduce(khg_ctx* ctx, khg_accs* a, void* comm) {
  if (ctx_dead(ctx) || !a) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce: bad arguments");
  if (!comm) return KHG_OK;
  int rc = rccl_bind();
  if (rc) return rc;
  KernelTimer kt(ctx, "c1_allreduce");
  int r = g_rccl.AllReduce(a->buf_d, a->buf_d, (size_t)a->n, kNcclFloat64, kNcclSum, comm, ctx->stream);
  return r ? rccl_fail("ncclAllReduce", r) : KHG_OK;
}
extern "C" int khg_accs_allreduce_f32(khg_ctx* ctx, khg_accs* a, void* comm) {
  if (ctx_dead(ctx) || !a) return khg_set_error(KHG_E_ARG, "khg_accs_allreduce_f32: bad arguments");
  if (comm) { int rc = rccl_bind(); if (rc) return rc; }
  if (a->wire_cap < a->n) {
    DEVFREE(a->wire_d);
    int rc = dev_alloc(&a->wire_d, (size_t)a->n);
    if (rc) return rc;
    a->wire_cap = a->n;
  }
  KernelTimer kt(ctx, "c1_allreduce_f32");
  const int gb = (int)std::min<int64_t>(8192, (a->n + 255) / 256);
  KHG_LAUNCH(ctx, c1_narrow, dim3(gb), dim3(256), 0, ctx->stream, a->buf_d, a->wire_d, a->n);
  if (comm) {
    int r = g_rccl.AllReduce(a->wire_d, a->wire_d, (size_t)a->n, kNcclFloat32, kNcclSum, comm, ctx->stream);
    if (r) return rccl_fail("ncclAllReduce", r);
  }
  KHG_LAUNCH(ctx, c1_widen, dim3(gb), dim3(256), 0, ctx->stream, a->wire_d, a->buf_d, a->n);
  HIPCHK(hipGetLastError());
  return KHG_OK;
}
