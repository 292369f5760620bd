// MOCK of the small part of the HIP runtime that csrc/ddt_comm.cpp uses -- TEST INFRASTRUCTURE (tests/test_comm_mock.py).
// "Devices" are host memory, streams are queues of deferred operations that a scheduler (mock_runtime.cpp) executes in an
// order chosen by a policy (several deterministic extremes + seeded random), honouring only what HIP guarantees: stream order
// and event dependencies.  A missing wait in the pipeline therefore shows up as a wrong result under some schedule.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101 };
struct MockStream;
struct MockEvent;
typedef MockStream* hipStream_t;
typedef MockEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct uint4 {
  unsigned x, y, z, w;
};
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  int multiProcessorCount, clockRate;
  size_t sharedMemPerBlock, maxSharedMemoryPerMultiProcessor;
};
enum { hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };

extern "C" {
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int device);
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 10017 };
hipError_t hipDeviceGetAttribute(int* value, hipDeviceAttribute_t attr, int device);
hipError_t hipGetLastError(void);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind,
                            hipStream_t s);
}
