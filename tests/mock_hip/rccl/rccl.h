// MOCK of the RCCL calls csrc/ddt_comm.cpp makes -- TEST INFRASTRUCTURE (see ../hip/hip_runtime.h).  A collective is one
// deferred operation per rank; it executes when every rank's copy is at the head of its stream with its dependencies met.
#pragma once
#include <hip/hip_runtime.h>

typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInvalidArgument = 4, ncclInternalError = 3 };
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
struct MockComm;
typedef MockComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;

extern "C" {
const char* ncclGetErrorString(ncclResult_t r);
ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId id, int rank);
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devices);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
typedef struct ncclConfig_v21700 ncclConfig_t;
ncclResult_t ncclCommSplit(ncclComm_t comm, int color, int key, ncclComm_t* newcomm, ncclConfig_t* config);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s);
ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t t, ncclComm_t comm, hipStream_t s);
ncclResult_t ncclSend(const void* send, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s);
ncclResult_t ncclRecv(void* recv, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
}
