// RCCL entry points bound at run time (khg_c1.hip); the sharded M-step (khg_k4.hip) uses ncclReduce / ncclBroadcast through the same table
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/khg_hip.h"

struct KhgNcclId { char internal[KHG_COMM_ID_BYTES]; };   // ncclUniqueId (rccl.h: 128 opaque bytes, passed by value)
struct RcclApi {
  void* h = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, KhgNcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;      // optional (sharded M-step)
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;            // optional (khg_comm_info)
  int (*CommUserRank)(void*, int*) = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
extern RcclApi g_rccl;
int rccl_bind();
int rccl_fail(const char* what, int r);
constexpr int kNcclSum = 0, kNcclInt8 = 0, kNcclFloat32 = 7, kNcclFloat64 = 8;   // rccl.h: ncclRedOp_t / ncclDataType_t
