// rccl/rccl.h of the HOST SIMULATION (tests/hostsim): the five RCCL entry points csrc/s2d_multi.hip resolves at run time.
// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h beside it).  sim_rccl.cpp builds into a library with SONAME librccl.so.1.
#pragma once

#include <hip/hip_runtime.h>

typedef struct SimComm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;

extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t r);
}
