/* TEST INFRASTRUCTURE: a stand-in for <hip/hip_runtime.h> that lets csrc/wfa_launch.hip -- the host pipeline behind
 * launch_alignments*: threads, flags, lanes, staging ring, result scatter -- be compiled as plain C++ with g++ under
 * ThreadSanitizer (the GPU pool has no sanitizers).  "Device" memory is host memory; a stream is a worker thread that runs its
 * queue in order, so copies and events are as asynchronous as the real ones (a host buffer that is reused before its copy has
 * run, or a result read before its event, is a data race TSan sees).  Only what wfa_launch.hip uses.  tests/hip_stub/stub.cpp. */
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
struct StubStream;
struct StubEvent;
typedef StubStream* hipStream_t;
typedef StubEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; int major, minor; size_t sharedMemPerBlock; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceGetPCIBusId(char* buf, int len, int d);
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest);
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
hipError_t hipMalloc(void** p, size_t n);
template <typename T> hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
template <typename T> hipError_t hipHostMalloc(T** p, size_t n, unsigned flags) { return hipHostMalloc(reinterpret_cast<void**>(p), n, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
/* test hook: devices the stub pretends to have */
void stub_set_device_count(int n);
