// TEST INFRASTRUCTURE ONLY: stand-in for the RCCL calls of gpk_comm.hip (see ../hip/hip_runtime.h).  One rank: plain
// copies.  More ranks: ONE PROCESS PER RANK over a POSIX shared-memory segment named by the unique id (sim_rccl.cpp) --
// blocking collectives with the matching rules of the real library checked on every call (same operation, element count
// and root on every rank, in the same order), and every wait bounded, so that a rank left alone in a collective is an
// error after HOSTSIM_RCCL_TIMEOUT_S seconds, not a hang.
#pragma once
#include <hip/hip_runtime.h>
#define NCCL_UNIQUE_ID_BYTES 128
typedef enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;
struct ncclUniqueId { char internal[NCCL_UNIQUE_ID_BYTES]; };
struct hostsim_comm;
typedef hostsim_comm* ncclComm_t;
const char* ncclGetErrorString(ncclResult_t);
ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* c, int n, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t c);
ncclResult_t ncclBroadcast(const void* s, void* r, size_t n, ncclDataType_t, int root, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void* s, void* r, size_t n, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclAllReduce(const void* s, void* r, size_t n, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
