// TEST INFRASTRUCTURE ONLY: one-rank stand-in for the RCCL calls of gpk_comm.hip (see ../hip/hip_runtime.h).
#pragma once
#include <hip/hip_runtime.h>
#define NCCL_UNIQUE_ID_BYTES 128
typedef enum { ncclSuccess = 0, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;
struct ncclUniqueId { char internal[NCCL_UNIQUE_ID_BYTES]; };
struct hostsim_comm { int n; };
typedef hostsim_comm* ncclComm_t;
static inline const char* ncclGetErrorString(ncclResult_t) { return "hostsim nccl"; }
static inline ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 7, sizeof *id); return ncclSuccess; }
static inline ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId, int) {
  if (n != 1) return ncclInvalidUsage;  // a single process cannot host more ranks
  *c = new hostsim_comm{1};
  return ncclSuccess;
}
static inline ncclResult_t ncclCommDestroy(ncclComm_t c) { delete c; return ncclSuccess; }
static inline ncclResult_t ncclBroadcast(const void* s, void* r, size_t n, ncclDataType_t, int, ncclComm_t, hipStream_t) {
  if (s != r) std::memmove(r, s, n * 8);
  return ncclSuccess;
}
static inline ncclResult_t ncclAllGather(const void* s, void* r, size_t n, ncclDataType_t, ncclComm_t, hipStream_t) {
  std::memmove(r, s, n * 8);
  return ncclSuccess;
}
static inline ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) {
  if (s != r) std::memmove(r, s, n * 8);
  return ncclSuccess;
}
